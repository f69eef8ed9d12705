#!/usr/bin/env python
"""1x1 conv (plain GEMM) layer shapes of the path: every implicit-GEMM tile instance, time and TFLOP/s."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stereotracking_amd import _lib  # noqa: E402
from stereotracking_amd._lib import StConvDesc, check, ptr  # noqa: E402

lib = _lib.load()
dev = torch.device('cuda:0')


def bench(N, H, W, Cin, Cout, k=1, stride=1, reps=8):
    x = torch.randn(N, H, W, Cin, device=dev)
    w = torch.randn(Cout, Cin, k, k) / (k * Cin ** 0.5)
    b = torch.randn(Cout) * 0.1
    wp = torch.empty(lib.st_conv_packed_floats(Cout, Cin, k, k))
    bp = torch.empty((Cout + 31) // 32 * 32)
    check(lib.st_conv_pack_weights(ptr(w), ptr(b), None, None, None, None, 0.0, Cout, Cin, k, k, ptr(wp), ptr(bp)))
    wpd, bpd = wp.to(dev), bp.to(dev)
    Ho, Wo = (H + 2 * (k // 2) - k) // stride + 1, (W + 2 * (k // 2) - k) // stride + 1
    out = torch.empty(N, Ho, Wo, Cout, device=dev)
    d = StConvDesc()
    d.in_dev = x.data_ptr(); d.N, d.Hi, d.Wi, d.Cin, d.in_ld, d.in_off = N, H, W, Cin, Cin, 0
    d.wgt_dev = wpd.data_ptr(); d.bias_dev = bpd.data_ptr()
    d.Cout, d.KH, d.KW, d.stride, d.pad = Cout, k, k, stride, k // 2
    d.out1_dev = out.data_ptr(); d.out1_ld, d.out1_off, d.split = Cout, 0, Cout
    d.post_scale, d.act = 1.0, 1
    stream = _lib.current_stream()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    gf = 2.0 * N * Ho * Wo * k * k * Cin * Cout / 1e9
    res = []
    for v in range(22):
        if lib.st_conv2d_nhwc_variant(C.byref(d), stream, v) != 0:
            continue
        best = 1e9
        for _ in range(reps):
            e0.record()
            lib.st_conv2d_nhwc_variant(C.byref(d), stream, v)
            e1.record()
            e1.synchronize()
            best = min(best, e0.elapsed_time(e1))
        res.append((best, lib.st_conv_variant_name(v).decode()))
    res.sort()
    print(f'N={N} {H}x{W} k{k}s{stride} {Cin}->{Cout}: {gf:6.2f} GF  ' +
          '  '.join(f'{n} {t * 1e3:.0f}us/{gf / t:.0f}TF' for t, n in res[:5]) + f'  ... worst {res[-1][1]} {res[-1][0] * 1e3:.0f}us')


for s in [(8, 46, 80, 512, 256), (8, 92, 160, 256, 128), (8, 92, 160, 128, 128), (8, 46, 80, 256, 256), (8, 23, 40, 1024, 512),
          (8, 23, 40, 512, 512), (8, 92, 160, 64, 64), (8, 46, 80, 128, 128),
          (16, 368, 640, 32, 64, 3, 2), (8, 184, 320, 64, 128, 3, 2), (8, 92, 160, 128, 256, 3, 2)]:
    bench(*s)
