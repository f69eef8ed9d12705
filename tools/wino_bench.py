#!/usr/bin/env python
"""Winograd instance (variant 43) vs the best implicit-GEMM tile on the 3x3 / stride-1 layer shapes of the path."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stereotracking_amd import _lib  # noqa: E402
from stereotracking_amd._lib import StConvDesc, check, ptr  # noqa: E402

lib = _lib.load()
dev = torch.device('cuda:0')


def bench(N, H, W, Cin, Cout, res, reps=10):
    x = torch.randn(N, H, W, Cin, device=dev)
    w = torch.randn(Cout, Cin, 3, 3) / (3 * Cin ** 0.5)
    b = torch.randn(Cout) * 0.1
    wp = torch.empty(lib.st_conv_packed_floats(Cout, Cin, 3, 3))
    bp = torch.empty((Cout + 31) // 32 * 32)
    check(lib.st_conv_pack_weights(ptr(w), ptr(b), None, None, None, None, 0.0, Cout, Cin, 3, 3, ptr(wp), ptr(bp)))
    wn = torch.empty(lib.st_wino_packed_floats(Cout, Cin))
    check(lib.st_wino_pack_weights(ptr(wp), Cout, Cin, ptr(wn)))
    wpd, bpd, wnd = wp.to(dev), bp.to(dev), wn.to(dev)
    out = torch.empty(N, H, W, Cout, device=dev)
    r = torch.randn(N, H, W, Cout, device=dev) if res else None
    d = StConvDesc()
    d.in_dev = x.data_ptr(); d.N, d.Hi, d.Wi, d.Cin, d.in_ld, d.in_off = N, H, W, Cin, Cin, 0
    d.wgt_dev = wpd.data_ptr(); d.bias_dev = bpd.data_ptr(); d.wgt_wino_dev = wnd.data_ptr()
    d.Cout, d.KH, d.KW, d.stride, d.pad = Cout, 3, 3, 1, 1
    d.out1_dev = out.data_ptr(); d.out1_ld, d.out1_off, d.split = Cout, 0, Cout
    if res:
        d.res_dev = r.data_ptr(); d.res_ld, d.res_off = Cout, 0
    d.post_scale, d.act = 1.0, 1
    stream = _lib.current_stream()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    res_t = {}
    for v in list(range(22)) + [42, 43, 44]:
        if lib.st_conv2d_nhwc_variant(C.byref(d), stream, v) != 0:
            continue
        best = 1e9
        for _ in range(reps):
            e0.record()
            lib.st_conv2d_nhwc_variant(C.byref(d), stream, v)
            e1.record()
            e1.synchronize()
            best = min(best, e0.elapsed_time(e1))
        res_t[v] = best
    gf = 2.0 * N * H * W * 9 * Cin * Cout / 1e9
    bv = min((t, v) for v, t in res_t.items() if v < 43)
    print(f'N={N} {H}x{W} {Cin}->{Cout}{" +res" if res else ""}: {gf:6.2f} GF  best igemm v{bv[1]} {bv[0] * 1e3:7.1f} us '
          f'({gf / bv[0]:6.1f} TF/s)   winograd {res_t[43] * 1e3:7.1f} us ({gf / res_t[43]:6.1f} TF/s direct-equivalent)  '
          f'x{bv[0] / res_t[43]:.2f}   narrow {res_t.get(44, 0) * 1e3:7.1f} us')


for shape in [(8, 92, 160, 128, 256, False), (8, 92, 160, 128, 128, False), (8, 92, 160, 64, 64, True),
              (8, 46, 80, 128, 128, True), (8, 46, 80, 128, 256, False), (8, 23, 40, 256, 256, False),
              (8, 23, 40, 128, 256, False), (8, 23, 40, 128, 128, False), (16, 184, 320, 64, 64, False),
              (16, 184, 320, 32, 32, True), (8, 184, 320, 48, 48, False)]:
    bench(*shape)
