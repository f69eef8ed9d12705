set -e
timeout -k 10 900 python -m pytest tests/test_stereo_depth_gpu.py tests/test_shell_gpu.py -x -q -m gpu 2>&1 | tail -3
L=$PWD/stereotracking_amd/lib/libstereotrack_hip_ablation.so
for i in 1 2; do
echo "--- register-carried rows (tools build, same binary)"
ST_LIBRARY=$L timeout -k 10 300 python tools/cv_bench.py 20 2>&1 | grep "^agg3d\|two-call"
echo "--- LDS ring of four rows (ST_A3_RING=1)"
ST_A3_RING=1 ST_LIBRARY=$L timeout -k 10 300 python tools/cv_bench.py 20 2>&1 | grep "^agg3d\|two-call"
done
