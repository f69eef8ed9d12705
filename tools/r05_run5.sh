set -e
mkdir -p gpurun_out/r05
AB=$PWD/stereotracking_amd/lib/libstereotrack_hip_ablation.so
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r05/gpu_tests_d.log 2>&1 || { tail -n 40 gpurun_out/r05/gpu_tests_d.log; exit 1; }
tail -n 4 gpurun_out/r05/gpu_tests_d.log
# the aggressor of the co-run test (torch bf16 GEMMs) against the three FMA forms of the tools build: form 1 must show the defect
for m in 1 0; do
  echo "== cost volume FMA form $m beside torch.matmul bf16 GEMMs (hipBLASLt / rocBLAS kernels of another library)" >> gpurun_out/r05/cv_stress_torch.txt
  ST_CV_FMA=$m timeout -k 10 300 python tools/cv_stress.py torch >> gpurun_out/r05/cv_stress_torch.txt 2>&1
done
grep -v "^rep\|amdgpu.ids" gpurun_out/r05/cv_stress_torch.txt
# the parked split tests still pass on the tools build
ST_LIBRARY=$AB timeout -k 10 600 python -m pytest tests/test_conv_gpu.py -m gpu -q -k "split" > gpurun_out/r05/gpu_tests_split_tools_build.log 2>&1 || true
tail -n 3 gpurun_out/r05/gpu_tests_split_tools_build.log
python bench.py --agg3d-leg --split-leg > gpurun_out/r05/bench_c.json 2> gpurun_out/r05/bench_c.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r05/bench_c.json'))
print(d['value'], d['ms_per_step'], d['sustained'], d['roofline']['frac'])
print(json.dumps(d.get('secondary_split_bf16x3')))
print(json.dumps(d.get('parity'))[:1200])
PY
