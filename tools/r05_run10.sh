set -e
mkdir -p gpurun_out/r05
timeout -k 10 900 python -m pytest tests/test_stereo_depth_gpu.py -m gpu -x -q > gpurun_out/r05/gpu_tests_f.log 2>&1 || { tail -n 40 gpurun_out/r05/gpu_tests_f.log; exit 1; }
tail -n 3 gpurun_out/r05/gpu_tests_f.log
timeout -k 10 900 python bench.py --fullres-leg --no-test-step --no-cpu-baseline > gpurun_out/r05/bench_e.json 2> gpurun_out/r05/bench_e.err || { tail -n 20 gpurun_out/r05/bench_e.err; exit 1; }
python - <<'PY'
import json
d=json.load(open('gpurun_out/r05/bench_e.json'))
print(d['value'], d['ms_per_step'])
f=d.get('secondary_full_resolution')
print(f.get('value'), f.get('ms_per_step'), json.dumps(f.get('stage_ms_per_step_serialized')), f.get('disparity_vs_oracle_pair0'), f.get('error'))
PY
