set -e
mkdir -p gpurun_out/r05
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r05/gpu_tests_b.log 2>&1 || { tail -n 40 gpurun_out/r05/gpu_tests_b.log; exit 1; }
tail -n 4 gpurun_out/r05/gpu_tests_b.log
python tools/cv_bench.py 20 > gpurun_out/r05/cv_bench.txt 2>&1
cat gpurun_out/r05/cv_bench.txt
Q="--no-test-step --no-cpu-baseline --sustain-seconds 0 --steps 100 --warmup 20"
AB=$PWD/stereotracking_amd/lib/libstereotrack_hip_ablation.so
for i in 1 2; do
  ST_LIBRARY=$AB ST_WINO_LINEAR=1 python bench.py $Q > gpurun_out/r05/ab_linear_$i.json 2>/dev/null
  ST_LIBRARY=$AB python bench.py $Q > gpurun_out/r05/ab_xcd_$i.json 2>/dev/null
done
python - <<'PY'
import json
for n in ('ab_linear_1','ab_xcd_1','ab_linear_2','ab_xcd_2'):
    d=json.load(open(f'gpurun_out/r05/{n}.json'))
    f=d['roofline']['families']['st::wino_conv3x3_kernel']
    print(n, d['value'], 'wino family ms', f['ms_per_step'], 'frac', f['frac'], 'all mfma', d['roofline']['all_mfma_kernels']['ms_per_step'])
PY
python bench.py --agg3d-leg > gpurun_out/r05/bench_b.json 2> gpurun_out/r05/bench_b.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r05/bench_b.json'))
print(d['value'], d['ms_per_step'], d['sustained'])
print(json.dumps(d.get('secondary_agg3d'))[:1500])
print(json.dumps(d['roofline']['per_variant'])[:2500])
print(d['roofline']['frac'], d['roofline']['all_mfma_kernels'], d['roofline']['secondary_costvolume'])
print(d.get('test_step',{}).get('value'), d.get('cpu_baseline'), d.get('disparity_l1_vs_oracle'))
PY
