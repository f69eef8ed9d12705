set -e
mkdir -p gpurun_out/r05
hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize tools/micro/pkfma_corun.hip -o /tmp/pkfma_corun
timeout -k 10 400 /tmp/pkfma_corun 12 > gpurun_out/r05/pkfma_corun2.txt 2>&1
grep -v "^  diag\|^      " gpurun_out/r05/pkfma_corun2.txt | tail -n 30
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r05/gpu_tests_a.log 2>&1
tail -n 5 gpurun_out/r05/gpu_tests_a.log
python bench.py > gpurun_out/r05/bench_a.json 2> gpurun_out/r05/bench_a.err
cut -c1-300 gpurun_out/r05/bench_a.json
