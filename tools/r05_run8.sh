set -e
mkdir -p gpurun_out/r05
AB=$PWD/stereotracking_amd/lib/libstereotrack_hip_ablation.so
timeout -k 10 600 python -m pytest tests/test_stereo_depth_gpu.py -m gpu -x -q 2>&1 | tail -n 3
echo "# the co-run test against the OLD kernel form (tools build, ST_CV_FMA=1): must FAIL" > gpurun_out/r05/corun_test_old_form.txt
ST_LIBRARY=$AB ST_CV_FMA=1 timeout -k 10 300 python -m pytest tests/test_stereo_depth_gpu.py -m gpu -q -k "beside_bf16" >> gpurun_out/r05/corun_test_old_form.txt 2>&1 || true
tail -n 4 gpurun_out/r05/corun_test_old_form.txt
for m in 0 3; do
  echo "== cost volume FMA form $m beside aggressor variant 1" >> gpurun_out/r05/cv_stress_micro.txt
  ST_CV_FMA=$m timeout -k 10 300 python tools/cv_stress.py micro 1 >> gpurun_out/r05/cv_stress_micro.txt 2>&1
done
grep -v "^rep\|amdgpu.ids" gpurun_out/r05/cv_stress_micro.txt | tail -n 4
python tools/cv_bench.py 40 2>/dev/null | tee gpurun_out/r05/cv_bench_gated.txt
