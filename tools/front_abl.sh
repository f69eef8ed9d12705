#!/bin/bash
# tools-only: where does the fused stage-1 front kernel lose its time?  (timing-only ablations, wrong results)
make -C stereotracking_amd/csrc ABLATION=1 -j16 > /dev/null 2>&1
export ST_LIBRARY=$PWD/stereotracking_amd/lib/libstereotrack_hip_ablation.so
for abl in 0 1 4 16 21 53; do echo "== ST_FF_ABL=$abl"; ST_FF_ABL=$abl python tools/front_bench.py 2>&1 | grep "N=" | cut -c1-120; done
