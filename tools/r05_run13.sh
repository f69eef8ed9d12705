set -e
mkdir -p gpurun_out/r05
timeout -k 10 900 python -m pytest tests/test_multirank_gpu.py tests/test_stereo_depth_gpu.py -m gpu -x -q > gpurun_out/r05/gpu_tests_h.log 2>&1 || { tail -n 60 gpurun_out/r05/gpu_tests_h.log; exit 1; }
tail -n 3 gpurun_out/r05/gpu_tests_h.log
ST_BENCH_WORLD1_PG=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-test-step --sustain-seconds 0 > gpurun_out/r05/bench_rccl_world1.json 2> gpurun_out/r05/bench_rccl_world1.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r05/bench_rccl_world1.json'))
print(d['value'], d['config']['parallelism'], d['config']['collectives_issued'], d['config']['ranks_seen'])
PY
