set -e
mkdir -p gpurun_out/r05
timeout -k 10 900 python -m pytest tests/test_multirank_gpu.py -m gpu -x -q > gpurun_out/r05/gpu_tests_i.log 2>&1 || { tail -n 60 gpurun_out/r05/gpu_tests_i.log; exit 1; }
tail -n 3 gpurun_out/r05/gpu_tests_i.log
python bench.py --no-test-step --no-cpu-baseline > gpurun_out/r05/bench_f.json 2> gpurun_out/r05/bench_f.err
wc -l gpurun_out/r05/bench_f.json
python -c "import json; d=json.load(open('gpurun_out/r05/bench_f.json')); print(d['value'], d['config']['parallelism'])"
