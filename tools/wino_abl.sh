#!/bin/bash
# tools-only: where does the Winograd kernel lose its time?  (timing-only ablations, wrong results)
make -C stereotracking_amd/csrc ABLATION=1 -j16 > /dev/null 2>&1
export ST_LIBRARY=$PWD/stereotracking_amd/lib/libstereotrack_hip_ablation.so
for abl in 0 3 4 7 8 15; do echo "== ST_WN_ABL=$abl"; ST_WN_ABL=$abl python tools/wino_bench.py 2>&1 | grep -E "92x160 128->256|92x160 128->128|46x80 128->128" | cut -c1-200; done
