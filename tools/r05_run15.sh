set -e
mkdir -p gpurun_out/r05
timeout -k 10 400 python tools/inflight_power.py 6 > gpurun_out/r05/inflight_power.txt 2>&1 || { tail -n 30 gpurun_out/r05/inflight_power.txt; exit 1; }
grep -v amdgpu.ids gpurun_out/r05/inflight_power.txt
