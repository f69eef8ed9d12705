#!/usr/bin/env python
"""Summarise a rocprofv3 --pmc pass of SQ activity counters per kernel: share of wave cycles spent issuing VALU /
MFMA / LDS / VMEM instructions and waiting.  usage: pmc_sq.py <counter_collection.csv>"""
import csv
import sys
from collections import defaultdict

acc = defaultdict(lambda: defaultdict(float))
calls = defaultdict(int)
for row in csv.DictReader(open(sys.argv[1])):
    k = row['Kernel_Name'].replace('(anonymous namespace)::', '').split('(')[0].replace('void ', '')
    acc[k][row['Counter_Name']] += float(row['Counter_Value'])
    if row['Counter_Name'] == 'SQ_WAVE_CYCLES':
        calls[k] += 1
tot = sum(v['SQ_WAVE_CYCLES'] for v in acc.values())
for k, v in sorted(acc.items(), key=lambda kv: -kv[1]['SQ_WAVE_CYCLES'])[:24]:
    w = v['SQ_WAVE_CYCLES'] or 1.0
    f = lambda n: v.get(n, 0.0) / w
    print(f'{k[:58]:58s} x{calls[k]:4d} share {v["SQ_WAVE_CYCLES"] / tot:5.3f}  active {f("SQ_ACTIVE_INST_ANY"):4.2f} '
          f'(valu {f("SQ_ACTIVE_INST_VALU"):4.2f} lds {f("SQ_ACTIVE_INST_LDS"):4.2f} vmem {f("SQ_ACTIVE_INST_VMEM"):4.2f})  '
          f'wait_any {f("SQ_WAIT_ANY"):4.2f} wait_inst {f("SQ_WAIT_INST_ANY"):4.2f} wait_lds {f("SQ_WAIT_INST_LDS"):4.2f}  '
          f'ldsconf/active {v.get("SQ_LDS_BANK_CONFLICT", 0.0) / max(v.get("SQ_LDS_IDX_ACTIVE", 1.0), 1.0):4.2f}')
