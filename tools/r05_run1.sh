set -e
mkdir -p gpurun_out/r05
hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize tools/micro/pkfma_corun.hip -o /tmp/pkfma_corun
timeout -k 10 300 /tmp/pkfma_corun 12 > gpurun_out/r05/pkfma_corun.txt 2>&1
tail -n 45 gpurun_out/r05/pkfma_corun.txt
export ST_LIBRARY=$PWD/stereotracking_amd/lib/libstereotrack_hip_ablation.so
for m in 1 2 0; do
  echo "== cost volume FMA form $m beside the split instances 53 54 50 51" >> gpurun_out/r05/cv_stress_modes.txt
  ST_CV_FMA=$m timeout -k 10 300 python tools/cv_stress.py >> gpurun_out/r05/cv_stress_modes.txt 2>&1
done
grep -v "^rep" gpurun_out/r05/cv_stress_modes.txt
grep -c "^rep" gpurun_out/r05/cv_stress_modes.txt || true
unset ST_LIBRARY
python bench.py > gpurun_out/r05/bench_base.json 2> gpurun_out/r05/bench_base.err
cat gpurun_out/r05/bench_base.json | cut -c1-600
