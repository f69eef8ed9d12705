#!/bin/bash
# tools-only experiment: phase 0 (left + right, 16 images) run as two groups of 8 (ST_SUBBATCH, ablation build)
make -C stereotracking_amd/csrc ABLATION=1 -j16 > /dev/null 2>&1
export ST_LIBRARY=$PWD/stereotracking_amd/lib/libstereotrack_hip_ablation.so
for sb in 0 8 4; do echo "== ST_SUBBATCH=$sb"; ST_SUBBATCH=$sb python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-test-step --sustain-seconds 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"; done
