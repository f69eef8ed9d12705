set -e
mkdir -p gpurun_out/r05
rm -f gpurun_out/r05/cv_stress_micro.txt
for v in 0 1 2 3; do
  echo "== cost volume FMA form 1 (op_sel, tools build) beside aggressor variant $v of tests/helpers/mfma_aggressor.hip" >> gpurun_out/r05/cv_stress_micro.txt
  ST_CV_FMA=1 timeout -k 10 300 python tools/cv_stress.py micro $v >> gpurun_out/r05/cv_stress_micro.txt 2>&1
done
for m in 0 3; do
  echo "== cost volume FMA form $m beside aggressor variant 3" >> gpurun_out/r05/cv_stress_micro.txt
  ST_CV_FMA=$m timeout -k 10 300 python tools/cv_stress.py micro 3 >> gpurun_out/r05/cv_stress_micro.txt 2>&1
done
grep -v "^rep\|amdgpu.ids" gpurun_out/r05/cv_stress_micro.txt
timeout -k 10 300 python -m pytest tests/test_stereo_depth_gpu.py -m gpu -q -k "beside_bf16" 2>&1 | tail -n 2
