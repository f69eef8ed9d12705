set -e
mkdir -p gpurun_out/r05
timeout -k 10 600 python -m pytest tests/test_stereo_depth_gpu.py -m gpu -x -q 2>&1 | tail -n 3
AB=$PWD/stereotracking_amd/lib/libstereotrack_hip_ablation.so
rm -f gpurun_out/r05/cv_nwv_ab.txt
for r in 1 2; do for w in 2 4; do
  echo "== ST_CV_NWV=$w (waves per workgroup), run $r" >> gpurun_out/r05/cv_nwv_ab.txt
  ST_LIBRARY=$AB ST_CV_NWV=$w python tools/cv_bench.py 40 2>/dev/null | grep costvolume >> gpurun_out/r05/cv_nwv_ab.txt
done; done
cat gpurun_out/r05/cv_nwv_ab.txt
Q="--no-test-step --no-cpu-baseline --sustain-seconds 0 --steps 100 --warmup 20"
for i in 1 2; do for w in 2 4; do
  ST_LIBRARY=$AB ST_CV_NWV=$w python bench.py $Q > gpurun_out/r05/ab_nwv${w}_$i.json 2>/dev/null
done; done
python - <<'PY' | tee -a gpurun_out/r05/cv_nwv_ab.txt
import json
for i in (1,2):
    for w in (2,4):
        d=json.load(open(f'gpurun_out/r05/ab_nwv{w}_{i}.json'))
        c=d['roofline']['secondary_costvolume']; f=d['roofline']['secondary_costvolume_fullres']
        print(f'bench, {w} waves per workgroup, run {i}: in-flight {d["value"]:.1f} pairs/s; cost volume {c["avg_launch_us"]} us ({c["frac"]} of 8 TB/s); full-res sizing {f["launch_us"]} us ({f["frac"]})')
PY
