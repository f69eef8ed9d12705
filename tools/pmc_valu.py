#!/usr/bin/env python
"""Where the vector lanes' time goes, per kernel: share of SIMD cycles spent in MFMA, in transcendental and in other
VALU instructions (on the fp32 matrix path VALU issue time is taken from the matrix rate).
usage: pmc_valu.py <counter_collection.csv> [more passes ...]   (rocprofv3 --pmc passes of the same command)"""
import csv
import sys
from collections import defaultdict

acc = defaultdict(lambda: defaultdict(float))
calls = defaultdict(lambda: defaultdict(int))
for path in sys.argv[1:]:
    for row in csv.DictReader(open(path)):
        k = row['Kernel_Name'].replace('(anonymous namespace)::', '').split('(')[0].replace('void ', '')
        acc[k][row['Counter_Name']] += float(row['Counter_Value'])
        calls[k][row['Counter_Name']] += 1
rows = []
for k, v in acc.items():
    n = max(calls[k].values())
    gui = v.get('GRBM_GUI_ACTIVE', 0.0)
    if not gui or 'SQ_INSTS_VALU' not in v:
        continue
    # GRBM_GUI_ACTIVE is summed over the 8 XCDs by rocprofv3: cycles per XCD = gui / 8; 128 SIMDs per XCD
    simd_cycles = gui / 8.0 * 1024.0
    mfma = v.get('SQ_INSTS_MFMA', 0.0)
    trans = v.get('SQ_INSTS_VALU_TRANS_F32', 0.0)
    valu = v['SQ_INSTS_VALU'] - mfma
    busy = v.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0)     # matrix-pipe busy cycles summed over the 1024 SIMDs (tools/pmc_mfma.py)
    rows.append((gui, k, n, mfma, valu, trans, busy, v.get('SQ_VALU_MFMA_COEXEC_CYCLES', 0.0), simd_cycles,
                 v.get('SQ_ACTIVE_INST_VALU', 0.0), v.get('SQ_INSTS_SALU', 0.0)))
tot = sum(r[0] for r in rows)
print('share = of GPU time; mfma = matrix pipe busy / SIMD cycles; valu = estimated issue cycles of the non-MFMA vector instructions / '
      'SIMD cycles (4 per wave instruction, 16 per transcendental); idle = 1 - mfma - valu')
print(f'{"kernel":56s} {"share":>6s} {"mfma":>6s} {"valu":>6s} {"(trans)":>8s} {"idle":>6s} {"valu:mfma instr":>15s}')
for gui, k, n, mfma, valu, trans, busy, coex, sc, act, salu in sorted(rows, reverse=True)[:28]:
    # issue-cycle estimates per SIMD: plain VALU 4 cycles per wave instruction, transcendental 16
    valu_c = ((valu - trans) * 4.0 + trans * 16.0) / sc
    m = busy / (gui * 128.0) if gui else 0.0
    print(f'{k[:56]:56s} {gui / tot:6.3f} {m:6.3f} {valu_c:6.3f} {trans * 16.0 / sc:8.3f} {max(0.0, 1 - m - valu_c):6.3f} '
          f'{valu / mfma if mfma else float("inf"):15.2f}')
