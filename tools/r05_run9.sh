set -e
mkdir -p gpurun_out/r05
AB=$PWD/stereotracking_amd/lib/libstereotrack_hip_ablation.so
echo "# tests/test_stereo_depth_gpu.py::test_costvolume_beside_bf16_mfma_kernels_equals_serial_run against the OLD kernel form (tools build, ST_CV_FMA=1): must FAIL" > gpurun_out/r05/corun_test_old_form.txt
ST_LIBRARY=$AB ST_CV_FMA=1 timeout -k 10 300 python -m pytest tests/test_stereo_depth_gpu.py -m gpu -q -k "beside_bf16" >> gpurun_out/r05/corun_test_old_form.txt 2>&1 || true
tail -n 3 gpurun_out/r05/corun_test_old_form.txt
timeout -k 10 900 python -m pytest tests/test_stereo_depth_gpu.py tests/test_shell_gpu.py -m gpu -x -q > gpurun_out/r05/gpu_tests_e.log 2>&1 || { tail -n 40 gpurun_out/r05/gpu_tests_e.log; exit 1; }
tail -n 3 gpurun_out/r05/gpu_tests_e.log
timeout -k 10 900 python bench.py --agg3d-leg --fullres-leg --no-test-step > gpurun_out/r05/bench_d.json 2> gpurun_out/r05/bench_d.err || { tail -n 20 gpurun_out/r05/bench_d.err; exit 1; }
python - <<'PY'
import json
d=json.load(open('gpurun_out/r05/bench_d.json'))
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['secondary_costvolume'])
print(json.dumps(d.get('secondary_full_resolution'), indent=1)[:3000])
print(json.dumps(d.get('secondary_agg3d'))[:600])
PY
