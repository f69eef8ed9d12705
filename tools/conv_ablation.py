#!/usr/bin/env python
"""Timing-only ablations of the conv kernel on one layer shape (results are wrong by construction):
   full kernel vs. no-global-loads vs. no-loads-no-barrier.  Needs the TOOLS-ONLY library:
   make -C stereotracking_amd/csrc ABLATION=1 ; usage: conv_ablation.py"""
import ctypes as C
import os
import sys

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault('ST_LIBRARY', os.path.join(_ROOT, 'stereotracking_amd', 'lib', 'libstereotrack_hip_ablation.so'))

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stereotracking_amd import _lib  # noqa: E402
from stereotracking_amd._lib import StConvDesc, check  # noqa: E402

lib = _lib.load()
dev = torch.device('cuda:0')


def bench(N, H, W, Cin, Cout, k, stride, variant, reps=20):
    x = torch.randn(N, H, W, Cin, device=dev)
    Kpad = (k * k * Cin + 31) // 32 * 32
    wp = torch.randn((Cout + 31) // 32 * 32, Kpad, device=dev) / (k * k * Cin) ** 0.5
    bp = torch.randn((Cout + 31) // 32 * 32, device=dev)
    Ho, Wo = (H + 2 * (k // 2) - k) // stride + 1, (W + 2 * (k // 2) - k) // stride + 1
    out = torch.empty(N, Ho, Wo, Cout, device=dev)
    d = StConvDesc()
    d.in_dev = x.data_ptr(); d.N, d.Hi, d.Wi, d.Cin, d.in_ld, d.in_off = N, H, W, Cin, Cin, 0
    d.wgt_dev = wp.data_ptr(); d.bias_dev = bp.data_ptr()
    d.Cout, d.KH, d.KW, d.stride, d.pad = Cout, k, k, stride, k // 2
    d.out1_dev = out.data_ptr(); d.out1_ld, d.out1_off, d.split = Cout, 0, Cout
    d.act = 1
    s = _lib.current_stream()
    for _ in range(3):
        check(lib.st_conv2d_nhwc_variant(C.byref(d), s, variant))
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(reps):
        check(lib.st_conv2d_nhwc_variant(C.byref(d), s, variant))
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    fl = 2.0 * N * Ho * Wo * k * k * Cin * Cout
    return ms, fl / ms / 1e9


for shape in [(8, 92, 160, 128, 256, 3, 1), (8, 92, 160, 64, 64, 3, 1), (8, 46, 80, 128, 128, 3, 1), (8, 92, 160, 128, 128, 1, 1)]:
    for v in (3, 14, 22, 27, 28, 29):
        try:
            ms, tf = bench(*shape, v)
            print(f'shape {shape} variant {v:3d}: {ms * 1e3:8.1f} us  {tf:7.1f} TF/s')
        except Exception as e:  # noqa: BLE001
            print(f'shape {shape} variant {v}: {e}')
