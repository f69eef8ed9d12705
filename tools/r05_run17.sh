set -e
mkdir -p gpurun_out/r05
timeout -k 10 500 python tools/kernel_power.py > gpurun_out/r05/kernel_power.txt 2>&1 || { tail -n 30 gpurun_out/r05/kernel_power.txt; exit 1; }
grep -v amdgpu.ids gpurun_out/r05/kernel_power.txt
