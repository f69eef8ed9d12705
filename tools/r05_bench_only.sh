set -e
mkdir -p gpurun_out/r05
python bench.py --agg3d-leg --fullres-leg > gpurun_out/r05/bench_final.json 2> gpurun_out/r05/bench_final.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r05/bench_final.json'))
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['traffic'], d['roofline']['traffic_source'])
PY
