#!/usr/bin/env python
"""GPU idle analysis of a `rocprofv3 --kernel-trace --output-format csv` trace: union of kernel intervals, idle gaps,
and what ran before / after the largest gaps.  usage: trace_gaps.py <kernel_trace.csv> [tail_ms]"""
import csv
import sys

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'][:60]))
rows.sort()
tail_ms = float(sys.argv[2]) if len(sys.argv) > 2 else 60.0
# the window of `tail_ms` with the most kernel launches of the conv families = the timed in-flight loop (the roofline
# pass and the secondary legs that follow it are serialized and much sparser)
starts = [r[0] for r in rows if 'conv' in r[2]]
best, lo = (0, starts[0]), 0
for hi in range(len(starts)):
    while starts[hi] - starts[lo] > tail_ms * 1e6:
        lo += 1
    if hi - lo + 1 > best[0]:
        best = (hi - lo + 1, starts[lo])
rows = [r for r in rows if best[1] <= r[0] <= best[1] + tail_ms * 1e6]
t0 = rows[0][0]
busy, cur_s, cur_e, gaps = 0, rows[0][0], rows[0][1], []
last_name = rows[0][2]
for s, e, n in rows[1:]:
    if s > cur_e:
        busy += cur_e - cur_s
        gaps.append((s - cur_e, cur_e - t0, last_name, n))
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
    last_name = n
busy += cur_e - cur_s
span = cur_e - t0
print(f'window {span / 1e6:.2f} ms, busy {busy / 1e6:.2f} ms ({busy / span:.1%}), {len(rows)} kernels, '
      f'idle {sum(g[0] for g in gaps) / 1e6:.2f} ms in {len(gaps)} gaps')
hist = [0, 0, 0, 0]
for g in gaps:
    hist[0 if g[0] < 5e3 else 1 if g[0] < 50e3 else 2 if g[0] < 500e3 else 3] += g[0]
print('idle by gap size: <5us %.2f ms, 5-50us %.2f ms, 50-500us %.2f ms, >500us %.2f ms' % tuple(h / 1e6 for h in hist))
for g in sorted(gaps, reverse=True)[:25]:
    print(f'  gap {g[0] / 1e3:8.1f} us at t={g[1] / 1e6:8.2f} ms   after {g[2]:<50} before {g[3]}')
