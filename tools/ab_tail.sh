#!/bin/bash
# A/B of the fused stage-1 CSP tail (wino_csp_tail.hip, variant 56) against the two launches it replaces, on ONE box:
# interleaved runs of the headline loop on the tools build (same code, ST_NO_FUSED_TAIL=1 switches the fusion off) and the
# per-op table of both plans.  Output: gpurun_out/r06_tail_ab.txt
set -e
OUT=gpurun_out/r06_tail_ab.txt
export ST_LIBRARY=$PWD/stereotracking_amd/lib/libstereotrack_hip_ablation.so
COMMON="--steps 60 --warmup 15 --no-cpu-baseline --no-test-step --no-secondary-legs --sustain-seconds 0"
val() { python -c "import json,sys; l=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(l['value'], l['ms_per_step'], {k: round(v['ms_per_step'],4) for k,v in l['roofline']['families'].items()})" $1; }
echo "# interleaved headline runs (tools build; fused tail on / off), pairs/s ms/step families(ms/step serialized)" > $OUT
for i in 1 2 3; do
  python bench.py $COMMON > /tmp/on.json 2>/dev/null;  echo "fused    $(val /tmp/on.json)" >> $OUT
  ST_NO_FUSED_TAIL=1 python bench.py $COMMON > /tmp/off.json 2>/dev/null; echo "unfused  $(val /tmp/off.json)" >> $OUT
done
echo "# per-op table, fused plan" >> $OUT
python tools/op_profile.py --out /tmp/op_on.txt > /dev/null 2>&1 && grep -E "stage1|total" /tmp/op_on.txt >> $OUT
echo "# per-op table, unfused plan (ST_NO_FUSED_TAIL=1)" >> $OUT
ST_NO_FUSED_TAIL=1 python tools/op_profile.py --out /tmp/op_off.txt > /dev/null 2>&1 && grep -E "stage1|total" /tmp/op_off.txt >> $OUT
cat $OUT
