#!/usr/bin/env python
"""Per-kernel MFMA-pipe utilisation from a rocprofv3 --pmc pass (north_star: "MFMA utilisation against gfx950 peaks").

  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES --kernel-trace -d <dir> -- python bench.py ...
  python tools/pmc_mfma.py <dir>/**/*_counter_collection.csv [out.json]

SQ_VALU_MFMA_BUSY_CYCLES = cycles the matrix pipes were busy, summed over all SIMDs of the chip (64 per
v_mfma_f32_32x32x2_f32, 32 per v_mfma_f32_16x16x4_f32: MI355X_MICROARCH.md).  GRBM_GUI_ACTIVE = busy cycles summed
over the 8 XCDs, i.e. kernel duration in shader cycles x 8.  With 256 CUs x 4 SIMDs:
    mfma_util = MFMA_BUSY / (1024 SIMDs x GUI_ACTIVE / 8) = MFMA_BUSY / (128 x GUI_ACTIVE)
and, for an fp32 kernel, achieved TFLOP/s = mfma_util x 64 flop/cycle/SIMD x 1024 SIMDs x clock."""
import csv
import json
import sys
from collections import defaultdict

acc = defaultdict(lambda: defaultdict(float))
calls = defaultdict(int)
with open(sys.argv[1]) as f:
    for row in csv.DictReader(f):
        k = row['Kernel_Name']
        acc[k][row['Counter_Name']] += float(row['Counter_Value'])
        if row['Counter_Name'] == 'GRBM_GUI_ACTIVE':
            calls[k] += 1
out = {}
rows = []
for k, c in acc.items():
    gui, mf = c.get('GRBM_GUI_ACTIVE', 0.0), c.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0)
    if gui <= 0:
        continue
    short = k.replace('(anonymous namespace)::', '').split('(')[0].replace('void ', '')
    util = mf / (128.0 * gui)
    rows.append((gui, short, calls[k], util, c.get('SQ_BUSY_CU_CYCLES', 0.0)))
tot_gui = sum(r[0] for r in rows)
tot_mf = sum(acc[k].get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0) for k in acc)
for gui, short, n, util, busy in sorted(rows, reverse=True):
    out[short] = dict(launches=n, gui_active_cycles_per_launch=round(gui / max(n, 1)), mfma_util=round(util, 4),
                      share_of_gpu_time=round(gui / tot_gui, 4))
    print(f'{short[:70]:70s} launches {n:5d}  share {gui / tot_gui:6.3f}  MFMA pipe busy {util:6.3f}')
out['_all_kernels'] = dict(mfma_util=round(tot_mf / (128.0 * tot_gui), 4))
print(f'all kernels: MFMA pipe busy {tot_mf / (128.0 * tot_gui):.3f} of the time the GPU was active')
if len(sys.argv) > 2:
    json.dump(out, open(sys.argv[2], 'w'), indent=1)
