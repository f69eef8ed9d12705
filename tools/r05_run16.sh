set -e
mkdir -p gpurun_out/r05
hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_power.hip -o /tmp/mfma_power
timeout -k 10 300 /tmp/mfma_power > gpurun_out/r05/mfma_power.txt 2>&1
cat gpurun_out/r05/mfma_power.txt
