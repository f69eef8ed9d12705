#!/usr/bin/env python
"""Per-op timing table of the detector graph (HIP events around every launch, see
st_detector_set_timing).  Usage: python tools/op_profile.py [--batch 8] [--steps 5] [--out file]"""
import argparse
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stereotracking_amd._lib import check  # noqa: E402
from stereotracking_amd.pipeline import StereoDensePipeline  # noqa: E402
from stereotracking_amd.synthetic import synthetic_batch, synthetic_state_dict  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--batch', type=int, default=8)
ap.add_argument('--steps', type=int, default=5)
ap.add_argument('--out', default='')
a = ap.parse_args()
dev = torch.device('cuda:0')
pipe = StereoDensePipeline(a.batch, (720, 1280), 0.5, 0.33, 1, stereo=True, max_disp=192, agg_layers=2)
pipe.load_state_dict(synthetic_state_dict(pipe.param_table(), 0), tuning_cache=os.environ.get('ST_TUNE_CACHE'))
b = synthetic_batch(list(range(a.batch)), 720, 1280, 192)
img, right = b['img'].to(dev), b['right'].to(dev)
det, lib = pipe.det, pipe.det.lib
for _ in range(2):
    pipe.run(img, right)
check(lib.st_detector_set_timing(det.handle, 1))
n = lib.st_detector_num_ops(det.handle)
ms, kind, var, macs, ph = (np.zeros(n, np.float32), np.zeros(n, np.int32), np.zeros(n, np.int32),
                           np.zeros(n, np.float64), np.zeros(n, np.int32))
tot = np.zeros(n)
p = lambda x: x.ctypes.data_as(C.c_void_p)
sm = pipe.stereo_module
sm.timing = True
agg_ms = []
for _ in range(a.steps):
    pipe.run(img, right)
    torch.cuda.synchronize()
    agg_ms.append([t for _, t in sm.pop_times()])
    check(lib.st_detector_op_times(det.handle, n, p(ms), p(kind), p(var), p(macs), p(ph)))
    tot += ms
tot /= a.steps
buf = C.create_string_buffer(512)
lines = []
tiles = {v: lib.st_conv_variant_name(v).decode() for v in range(-1, 64)}
# One row per LAUNCH.  An op computed inside the launch of the op in front of it (the fused front kernel's three ops, a
# chained 1x1 pair, the riders of a grouped Winograd launch) has only its event pair's overhead as "duration": it is
# recognised by a rate no launch of its own could have (above the fp32 MFMA peak, x2.25 for Winograd) and folded into
# the leading row, time and flops together; it is listed under it without a rate.
PEAK = 157.3
descs = []
for i in range(n):
    lib.st_detector_op_desc(det.handle, i, buf, 512)
    descs.append(buf.value.decode())
leader = list(range(n))
for i in range(1, n):
    name = tiles[int(var[i])]
    rate = 2 * macs[i] / (tot[i] * 1e-3) / 1e12 if tot[i] > 0 else 0.0
    if kind[i] == 1 and kind[i - 1] == 1 and (name == 'wino2x2g+' or rate > PEAK * (2.25 if name.startswith('wino') else 1.0)):
        leader[i] = leader[i - 1]
for i in range(n):
    if leader[i] != i:
        lines.append(f'{i:3d}         +      {tiles[int(var[i])]:>8s} {2 * macs[i] / 1e9:8.2f} GF     (in launch {leader[i]})  {descs[i]}')
        continue
    members = [j for j in range(n) if leader[j] == i]
    t = sum(tot[j] for j in members)
    fl = 2 * sum(macs[j] for j in members)
    tf = fl / (t * 1e-3) / 1e12 if t > 0 and fl > 0 else 0
    lines.append(f'{i:3d} {t * 1e3:9.1f} us  {tiles[int(var[i])]:>8s} {fl / 1e9:8.2f} GF {tf:7.1f} TF/s  {descs[i]}'
                 + (f'  [{len(members)} ops in this launch]' if len(members) > 1 else ''))
agg_ms = np.asarray(agg_ms).mean(0) if agg_ms and agg_ms[0] else np.zeros(0)
am = sm.agg_macs(a.batch, pipe.height // 4, pipe.width // 4)
for l, t in enumerate(agg_ms):
    lines.append(f'agg{l} {t * 1e3:8.1f} us  {tiles[int(sm.variant)]:>8s} {2 * am / 1e9:8.2f} GF {2 * am / (t * 1e-3) / 1e12:7.1f} TF/s  '
                 f'stereo aggregation conv3x3 48->48 @{pipe.height // 4}x{pipe.width // 4}')
lines.append(f'total {tot.sum():.3f} ms, conv {tot[kind == 1].sum():.3f} ms, {2 * macs.sum() / 1e9:.1f} GFLOP')
txt = '\n'.join(lines)
print(txt)
if a.out:
    open(a.out, 'w').write(txt + '\n')
