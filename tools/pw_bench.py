#!/usr/bin/env python
"""LDS-resident 1x1 kernel (variant 46) vs the best implicit-GEMM tile / streaming kernel on the 1x1 layer shapes of
the path (steady-state back-to-back timing, 50 launches per measurement)."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stereotracking_amd import _lib  # noqa: E402
from stereotracking_amd._lib import StConvDesc, check, ptr  # noqa: E402

lib = _lib.load()
dev = torch.device('cuda:0')


def bench(N, H, W, Cin, Cout, split, res, reps=50):
    x = torch.randn(N, H, W, Cin, device=dev)
    w = torch.randn(Cout, Cin, 1, 1) / Cin ** 0.5
    b = torch.randn(Cout) * 0.1
    wp = torch.empty(lib.st_conv_packed_floats(Cout, Cin, 1, 1))
    bp = torch.empty((Cout + 31) // 32 * 32)
    check(lib.st_conv_pack_weights(ptr(w), ptr(b), None, None, None, None, 0.0, Cout, Cin, 1, 1, ptr(wp), ptr(bp)))
    wpd, bpd = wp.to(dev), bp.to(dev)
    s = split or Cout
    out = torch.empty(N, H, W, s, device=dev)
    out2 = torch.empty(N, H, W, 2 * (Cout - s), device=dev) if split else None
    r = torch.randn(N, H, W, Cout, device=dev) if res else None
    d = StConvDesc()
    d.in_dev = x.data_ptr(); d.N, d.Hi, d.Wi, d.Cin, d.in_ld, d.in_off = N, H, W, Cin, Cin, 0
    d.wgt_dev = wpd.data_ptr(); d.bias_dev = bpd.data_ptr()
    d.Cout, d.KH, d.KW, d.stride, d.pad = Cout, 1, 1, 1, 0
    d.out1_dev = out.data_ptr(); d.out1_ld, d.out1_off, d.split = s, 0, s
    if split:
        d.out2_dev = out2.data_ptr(); d.out2_ld, d.out2_off = 2 * (Cout - s), Cout - s
    if res:
        d.res_dev = r.data_ptr(); d.res_ld, d.res_off = Cout, 0
    d.post_scale, d.act = 0.5 if res else 1.0, 1
    stream = _lib.current_stream()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t = {}
    for v in list(range(22)) + [41, 46]:
        if lib.st_conv2d_nhwc_variant(C.byref(d), stream, v) != 0:
            continue
        torch.cuda.synchronize()
        e0.record()
        for _ in range(reps):
            lib.st_conv2d_nhwc_variant(C.byref(d), stream, v)
        e1.record(); e1.synchronize()
        t[v] = e0.elapsed_time(e1) / reps
    gf = 2.0 * N * H * W * Cin * Cout / 1e9
    mb = N * H * W * (Cin + Cout * (2 if res else 1)) * 4 / 1e6
    bv = min((tt, v) for v, tt in t.items() if v != 46)
    print(f'N={N} {H}x{W} {Cin}->{Cout}{" split" if split else ""}{" +res" if res else ""}: {gf:5.2f} GF {mb:6.1f} MB  best other '
          f'v{bv[1]:<2d} {bv[0] * 1e3:6.1f} us ({gf / bv[0]:5.1f} TF/s)   resident {t[46] * 1e3:6.1f} us ({gf / t[46]:5.1f} TF/s, '
          f'{mb / t[46] / 1e3:4.2f} TB/s)  x{bv[0] / t[46]:.2f}')


for shape in [(8, 92, 160, 64, 64, None, False), (8, 92, 160, 128, 128, 64, False), (8, 92, 160, 128, 128, None, False),
              (8, 92, 160, 256, 128, 64, False), (8, 46, 80, 128, 128, None, False), (8, 46, 80, 256, 128, None, False),
              (16, 184, 320, 64, 64, None, False), (8, 184, 320, 64, 64, None, True)]:
    bench(*shape)
