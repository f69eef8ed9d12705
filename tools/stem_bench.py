#!/usr/bin/env python
"""Micro-benchmark of the fused Focus+stem kernel on the bench geometry (8 x 3 x 736 x 1280 -> 32 ch)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stereotracking_amd import _lib  # noqa: E402
from stereotracking_amd._lib import check, ptr  # noqa: E402

lib = _lib.load()
N, H, W, C = 8, 736, 1280, 32
dev = torch.device('cuda:0')
x = torch.rand(N, 3, H, W, device=dev) * 255
w = torch.randn(C, 12, 3, 3) / 200
wp = torch.empty(lib.st_stem_packed_floats(C))
bp = torch.empty(32)
planes = int(sys.argv[2]) if len(sys.argv) > 2 else 3
check(lib.st_stem_pack_weights(ptr(w), None, None, None, None, None, 0.0, C, planes, ptr(wp), ptr(bp)))
wd, bd = wp.to(dev), bp.to(dev)
out = torch.empty(N, H // 2, W // 2, C, device=dev)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
for _ in range(3):
    check(lib.st_stem_focus_conv(ptr(x), N, H, W, planes, ptr(wd), ptr(bd), C, ptr(out), C, 0, 1, None))
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    check(lib.st_stem_focus_conv(ptr(x), N, H, W, planes, ptr(wd), ptr(bd), C, ptr(out), C, 0, 1, None))
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) / reps * 1e3
gf = 2.0 * N * (H // 2) * (W // 2) * 36 * planes * C / 1e9
print(f'stem_focus_conv: {us:.1f} us  {gf / us * 1e-3:.1f} TF/s  ({gf:.2f} GFLOP)')
