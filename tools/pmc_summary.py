#!/usr/bin/env python
"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes per kernel.

FETCH_SIZE / WRITE_SIZE are reported in KiB.  Per MI355X_MICROARCH.md §HBM, on gfx950 FETCH_SIZE
counts 128-B requests at 64 B, i.e. reports exactly half the bytes of a wide coalesced streaming
read: the `fetch_bytes_corrected` column doubles it.  The focus_pack kernel (pure streaming copy of
known size) is printed as the calibration row.
usage: pmc_summary.py <fetch_counter_collection.csv> <write_counter_collection.csv> [out.json]"""
import csv
import json
import sys
from collections import defaultdict


def load(path, counter):
    acc = defaultdict(lambda: [0, 0.0])
    with open(path) as f:
        for row in csv.DictReader(f):
            if row['Counter_Name'] != counter:
                continue
            a = acc[row['Kernel_Name']]
            a[0] += 1
            a[1] += float(row['Counter_Value'])
    return acc


fetch = load(sys.argv[1], 'FETCH_SIZE')
write = load(sys.argv[2], 'WRITE_SIZE')
out = {}
for name in sorted(set(fetch) | set(write), key=lambda n: -(fetch.get(n, [0, 0])[1] + write.get(n, [0, 0])[1])):
    fc, fv = fetch.get(name, [0, 0.0])
    wc, wv = write.get(name, [0, 0.0])
    short = name.replace('(anonymous namespace)::', '').split('(')[0].replace('void ', '')
    out[short] = dict(fetch_calls=fc, write_calls=wc,
                      fetch_bytes_raw_per_launch=round(fv * 1024 / fc) if fc else None,
                      fetch_bytes_corrected_per_launch=round(2 * fv * 1024 / fc) if fc else None,
                      write_bytes_per_launch=round(wv * 1024 / wc) if wc else None)
    print(f'{short[:60]:60s} calls {fc:5d}/{wc:5d}  fetch(raw) {fv * 1024 / max(fc, 1) / 1e6:9.2f} MB  '
          f'fetch(x2) {2 * fv * 1024 / max(fc, 1) / 1e6:9.2f} MB  write {wv * 1024 / max(wc, 1) / 1e6:9.2f} MB  per launch')
if len(sys.argv) > 3:
    json.dump(out, open(sys.argv[3], 'w'), indent=1)
