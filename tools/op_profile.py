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
pipe = StereoDensePipeline(a.batch, (720, 1280), 0.5, 0.33, 1, stereo=True, max_disp=192)
pipe.load_state_dict(synthetic_state_dict(pipe.param_table(), 0))
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
for _ in range(a.steps):
    pipe.run(img, right)
    torch.cuda.synchronize()
    check(lib.st_detector_op_times(det.handle, n, p(ms), p(kind), p(var), p(macs), p(ph)))
    tot += ms
tot /= a.steps
buf = C.create_string_buffer(512)
lines = []
tiles = {v: lib.st_conv_variant_name(v).decode() for v in range(-1, 64)}
for i in range(n):
    lib.st_detector_op_desc(det.handle, i, buf, 512)
    tf = 2 * macs[i] / (tot[i] * 1e-3) / 1e12 if tot[i] > 0 and macs[i] > 0 else 0
    lines.append(f'{i:3d} {tot[i] * 1e3:9.1f} us  {tiles[int(var[i])]:>8s} {2 * macs[i] / 1e9:8.2f} GF {tf:7.1f} TF/s  {buf.value.decode()}')
lines.append(f'total {tot.sum():.3f} ms, conv {tot[kind == 1].sum():.3f} ms, {2 * macs.sum() / 1e9:.1f} GFLOP')
txt = '\n'.join(lines)
print(txt)
if a.out:
    open(a.out, 'w').write(txt + '\n')
