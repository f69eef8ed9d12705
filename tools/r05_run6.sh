set -e
mkdir -p gpurun_out/r05
AB=$PWD/stereotracking_amd/lib/libstereotrack_hip_ablation.so
# cost-volume FMA forms: bit-exactness of form 3 (aligned R pairs x explicit L pairs, no op_sel) and its speed against form 0
ST_LIBRARY=$AB ST_CV_FMA=3 timeout -k 10 300 python -m pytest tests/test_stereo_depth_gpu.py -m gpu -q -k "costvolume or wide_volume or full_resolution" > gpurun_out/r05/gpu_tests_cv_form3.log 2>&1 || true
tail -n 2 gpurun_out/r05/gpu_tests_cv_form3.log
rm -f gpurun_out/r05/cv_bench_forms.txt
for m in 0 3 0 3; do
  echo "== ST_CV_FMA=$m" >> gpurun_out/r05/cv_bench_forms.txt
  ST_LIBRARY=$AB ST_CV_FMA=$m python tools/cv_bench.py 40 2>/dev/null | grep costvolume >> gpurun_out/r05/cv_bench_forms.txt
done
cat gpurun_out/r05/cv_bench_forms.txt
echo "== cost volume FMA form 3 beside torch.matmul bf16 GEMMs" > gpurun_out/r05/cv_stress_torch3.txt
ST_CV_FMA=3 timeout -k 10 300 python tools/cv_stress.py torch >> gpurun_out/r05/cv_stress_torch3.txt 2>&1
grep -v "^rep\|amdgpu.ids" gpurun_out/r05/cv_stress_torch3.txt
# the aggressor of the -m gpu co-run test (tests/helpers/mfma_aggressor.hip) against the FMA forms: form 1 must FAIL beside it
rm -f gpurun_out/r05/cv_stress_micro.txt
for m in 1 0 3; do
  echo "== cost volume FMA form $m beside the register-only v_mfma_f32_16x16x32_bf16 loop (tests/helpers/mfma_aggressor.hip)" >> gpurun_out/r05/cv_stress_micro.txt
  ST_CV_FMA=$m timeout -k 10 300 python tools/cv_stress.py micro >> gpurun_out/r05/cv_stress_micro.txt 2>&1
done
grep -v "^rep\|amdgpu.ids" gpurun_out/r05/cv_stress_micro.txt
timeout -k 10 300 python -m pytest tests/test_stereo_depth_gpu.py -m gpu -q -k "beside_bf16" 2>&1 | tail -n 2
