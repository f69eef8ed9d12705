#!/usr/bin/env python
"""Micro-benchmark of the cost-volume kernel on the bench geometry (8 x 184 x 320 x 64 ch features, 48 levels)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stereotracking_amd import _lib  # noqa: E402
from stereotracking_amd._lib import check, ptr  # noqa: E402

lib = _lib.load()
N, Hf, Wf, C, D = 8, 184, 320, 64, 48
dev = torch.device('cuda:0')
fl = torch.randn(N, Hf, Wf, C, device=dev)
fr = torch.randn(N, Hf, Wf, C, device=dev)
cost = torch.empty(N, Hf, Wf, D, device=dev)
disp = torch.empty(N, Hf, Wf, device=dev)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
for name, oc, od in (('volume only', cost, None), ('fused soft-argmin only', None, disp), ('both', cost, disp)):
    for _ in range(3):
        check(lib.st_costvolume_softargmin(ptr(fl), ptr(fr), N, Hf, Wf, C, C, D, 32.0, ptr(oc), ptr(od), None))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        check(lib.st_costvolume_softargmin(ptr(fl), ptr(fr), N, Hf, Wf, C, C, D, 32.0, ptr(oc), ptr(od), None))
    e1.record()
    torch.cuda.synchronize()
    print(f'costvolume ({name}): {e0.elapsed_time(e1) / reps * 1e3:.1f} us')

# ---- 3-D aggregation layer (st_volume_agg3d: single-channel 3x3x3 over d, y, x; algorithmic traffic = volume read once +
# written once) at the bench volume and at the full-resolution sizing of SURVEY.md 8(d) (D = 192 x 720 x 1280, one pair)
import ctypes as C  # noqa: E402
w27 = (C.c_float * 27)(*[0.03 * ((i * 7) % 11 - 5) for i in range(27)])
for label, (n, hf, wf, d) in (('bench volume 8x184x320x48', (8, 184, 320, 48)), ('full-res 1x720x1280x192', (1, 720, 1280, 192))):
    vin = torch.randn(n, hf, wf, d, device=dev)
    vout = torch.empty_like(vin)
    for _ in range(3):
        check(lib.st_volume_agg3d(ptr(vin), ptr(vout), n, hf, wf, d, w27, 0.01, 1, None))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        check(lib.st_volume_agg3d(ptr(vin), ptr(vout), n, hf, wf, d, w27, 0.01, 1, None))
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / reps * 1e3
    nbytes = 2.0 * vin.numel() * 4
    print(f'agg3d ({label}): {us:.1f} us, algorithmic {nbytes / 1e6:.1f} MB -> {nbytes / us / 1e6:.2f} TB/s = {nbytes / us / 8e6:.2f} of 8 TB/s')
    del vin, vout

# ---- fused cost volume + first 3-D layer (st_costvolume_agg3d) at the full-resolution sizing of the product mode:
# 8 pairs x 736 x 1280, 8 feature channels, D = 192; algorithmic traffic = features read once + volume written once
n, h, w, c, d = 8, 736, 1280, 8, 192
gl = torch.randn(n, h, w, c, device=dev)
gr = torch.randn(n, h, w, c, device=dev)
vout = torch.empty(n, h, w, d, device=dev)
for _ in range(2):
    check(lib.st_costvolume_agg3d(ptr(gl), ptr(gr), n, h, w, c, c, d, w27, 0.01, 0, ptr(vout), None))
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(max(2, reps // 4)):
    check(lib.st_costvolume_agg3d(ptr(gl), ptr(gr), n, h, w, c, c, d, w27, 0.01, 0, ptr(vout), None))
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) / max(2, reps // 4) * 1e3
nbytes = vout.numel() * 4.0 + 2.0 * gl.numel() * 4
print(f'fused cost volume + agg3d (8 x 736 x 1280 x 192, C = 8): {us:.1f} us, algorithmic {nbytes / 1e6:.1f} MB -> '
      f'{nbytes / us / 1e6:.2f} TB/s = {nbytes / us / 8e6:.2f} of 8 TB/s; {n * h * w * d * (27 + c) * 2 / us / 1e6:.1f} TFLOP/s fp32 VALU')
vol = torch.empty_like(vout)
torch.cuda.synchronize()
e0.record()
for _ in range(2):
    check(lib.st_costvolume_softargmin(ptr(gl), ptr(gr), n, h, w, c, c, d, 1.0, ptr(vol), None, None))
    check(lib.st_volume_agg3d(ptr(vol), ptr(vout), n, h, w, d, w27, 0.01, 0, None))
e1.record()
torch.cuda.synchronize()
print(f'two-call form (volume through memory): {e0.elapsed_time(e1) / 2 * 1e3:.1f} us')

# ---- wide soft-argmin (st_softargmin on the D = 192 volume of 8 pairs: 5.8 GB read once)
disp = torch.empty(n, h, w, device=dev)
for _ in range(2):
    check(lib.st_softargmin(ptr(vout), n, h, w, d, 32.0, ptr(disp), None))
torch.cuda.synchronize()
e0.record()
for _ in range(5):
    check(lib.st_softargmin(ptr(vout), n, h, w, d, 32.0, ptr(disp), None))
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) / 5 * 1e3
print(f'soft-argmin (8 x 736 x 1280 x 192): {us:.1f} us, {vout.numel() * 4 / us / 1e6:.2f} TB/s = {vout.numel() * 4 / us / 8e6:.2f} of 8 TB/s')
