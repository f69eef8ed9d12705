#!/usr/bin/env python
"""Timing-only ablations of the fused cost-volume + 3-D layer kernel (tools build: ST_LIBRARY=..._ablation.so).
ST_CVA_MODE bits: 1 no output stores (results kept alive by an empty asm), 2 no cost FMAs, 8 no feature loads in the output steps (modes 0, 1, 2, 8, 9 are instantiated)."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault('ST_LIBRARY', os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                                 'stereotracking_amd', 'lib', 'libstereotrack_hip_ablation.so'))
from stereotracking_amd import _lib  # noqa: E402
from stereotracking_amd._lib import check, ptr  # noqa: E402

lib = _lib.load()
dev = torch.device('cuda:0')
n, h, w, c, d = 8, 736, 1280, 8, 192
gl = torch.randn(n, h, w, c, device=dev)
gr = torch.randn(n, h, w, c, device=dev)
vout = torch.empty(n, h, w, d, device=dev)
w27 = (C.c_float * 27)(*[0.03 * ((i * 7) % 11 - 5) for i in range(27)])
for mode in (sys.argv[1:] or ['0', '1', '2', '8', '9']):
    os.environ['ST_CVA_MODE'] = mode
    for _ in range(2):
        check(lib.st_costvolume_agg3d(ptr(gl), ptr(gr), n, h, w, c, c, d, w27, 0.01, 0, ptr(vout), None))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(4):
        check(lib.st_costvolume_agg3d(ptr(gl), ptr(gr), n, h, w, c, c, d, w27, 0.01, 0, ptr(vout), None))
    e1.record()
    torch.cuda.synchronize()
    print(f'mode {mode}: {e0.elapsed_time(e1) / 4 * 1e3:.1f} us')

vol = torch.empty_like(vout)
torch.cuda.synchronize()
e0.record()
for _ in range(2):
    check(lib.st_costvolume_softargmin(ptr(gl), ptr(gr), n, h, w, c, c, d, 1.0, ptr(vol), None, None))
    check(lib.st_volume_agg3d(ptr(vol), ptr(vout), n, h, w, d, w27, 0.01, 0, None))
e1.record()
torch.cuda.synchronize()
print(f'yardstick of this box, two-call form: {e0.elapsed_time(e1) / 2 * 1e3:.1f} us')
