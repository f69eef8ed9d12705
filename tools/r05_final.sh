# Round-5 record run on the GPU box (through gpurun): the whole GPU suite, smoke, the default bench with the secondary
# legs, then the profile set of tools/profile_round.sh.  Outputs under gpurun_out/; summaries are copied to profiles/.
set -e
mkdir -p gpurun_out/r05
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r05/smoke.log 2>&1 || { cat gpurun_out/r05/smoke.log; exit 1; }
tail -n 1 gpurun_out/r05/smoke.log
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r05/gpu_tests_final.log 2>&1 || { tail -n 40 gpurun_out/r05/gpu_tests_final.log; exit 1; }
tail -n 3 gpurun_out/r05/gpu_tests_final.log
python bench.py --agg3d-leg --fullres-leg > gpurun_out/r05/bench_final.json 2> gpurun_out/r05/bench_final.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r05/bench_final.json'))
print('headline', d['value'], d['ms_per_step'], 'sustained', d['sustained'], 'frac', d['roofline']['frac'], 'pipeline_frac', d['roofline']['pipeline_frac'])
print('test_step', d['test_step']['value'], d['test_step']['primed_loop']['value'], 'cpu', d['cpu_baseline']['value'], d['cpu_baseline']['value_1_thread'])
print('agg3d', d['secondary_agg3d'].get('value'), 'fullres', d['secondary_full_resolution'].get('value'))
PY
# stereo kernels on their own (product library), and the LDS-ring 3-D kernel of the tools build as yardstick when it is there
python tools/cv_bench.py 20 > gpurun_out/r05/cv_bench_final.txt 2>&1
L=$PWD/stereotracking_amd/lib/libstereotrack_hip_ablation.so
if [ -f $L ]; then
  echo "--- tools build, ST_A3_RING=1: the first streaming 3-D kernel (ring of four rows in LDS)" >> gpurun_out/r05/cv_bench_final.txt
  ST_A3_RING=1 ST_LIBRARY=$L python tools/cv_bench.py 20 2>&1 | grep "^agg3d\|two-call" >> gpurun_out/r05/cv_bench_final.txt
fi
grep -v amdgpu.ids gpurun_out/r05/cv_bench_final.txt
