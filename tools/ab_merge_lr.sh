#!/bin/bash
# A/B (tools build): stage-1 of the RGB branch as ONE launch over left | right (2N images) instead of two sub-batches of N.
export ST_LIBRARY=$PWD/stereotracking_amd/lib/libstereotrack_hip_ablation.so
export ST_TUNE_CACHE=$PWD/configs/tuning/mi355x.json
COMMON="--steps 60 --warmup 15 --no-cpu-baseline --no-test-step --no-secondary-legs --sustain-seconds 0"
val() { python -c "import json,sys; l=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); f=l['roofline']['families']; print(l['value'], l['ms_per_step'], 'front', f['st::front_s2_csp_kernel']['ms_per_step'], 'tail', f['st::wino_csp_tail_kernel']['ms_per_step'])" $1; }
for i in 1 2 3; do
  python bench.py $COMMON > /tmp/a.json 2>/dev/null; echo "two sub-batches  $(val /tmp/a.json)"
  ST_MERGE_LR=1 python bench.py $COMMON > /tmp/b.json 2>/tmp/b.err || tail -3 /tmp/b.err; echo "one launch (2N)  $(val /tmp/b.json)"
done
