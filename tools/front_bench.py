#!/usr/bin/env python
"""Fused stage-1 front kernel (st_conv3x3s2_csp_front) vs the three separate launches it replaces, at the shapes of the
path (16 / 8 images of 368x640x32 -> 184x320)."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stereotracking_amd import _lib  # noqa: E402
from stereotracking_amd._lib import StConvDesc, check, ptr  # noqa: E402

lib = _lib.load()
dev = torch.device('cuda:0')


def packed(cout, cin, k):
    w = torch.randn(cout, cin, k, k) / (k * cin ** 0.5)
    b = torch.randn(cout) * 0.1
    wp = torch.empty(lib.st_conv_packed_floats(cout, cin, k, k))
    bp = torch.empty((cout + 31) // 32 * 32)
    check(lib.st_conv_pack_weights(ptr(w), ptr(b), None, None, None, None, 0.0, cout, cin, k, k, ptr(wp), ptr(bp)))
    return wp, bp


def make(N, H, W):
    """-> (launch fused, launch the three separate convolutions, GFLOP of the three convolutions)"""
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    x = torch.randn(N, H, W, 32, device=dev)
    host = [packed(64, 32, 3), packed(64, 64, 1), packed(32, 32, 1)]
    devw = [(w.to(dev), b.to(dev)) for w, b in host]
    frags = []
    for (wp, _), (co, ci) in zip(host[1:], [(64, 64), (32, 32)]):
        f = torch.empty(lib.st_front_frag_floats(co, ci))
        check(lib.st_front_pack_frags(ptr(wp), co, ci, ptr(f)))
        frags.append(f.to(dev))
    s3 = torch.empty(N, Ho, Wo, 64, device=dev)
    main = torch.empty(N, Ho, Wo, 32, device=dev)
    cat = torch.empty(N, Ho, Wo, 64, device=dev)
    tmp = torch.empty(N, Ho, Wo, 32, device=dev)
    a, m, c = StConvDesc(), StConvDesc(), StConvDesc()
    a.in_dev = x.data_ptr(); a.N, a.Hi, a.Wi, a.Cin, a.in_ld, a.in_off = N, H, W, 32, 32, 0
    a.wgt_dev, a.bias_dev = devw[0][0].data_ptr(), devw[0][1].data_ptr()
    a.Cout, a.KH, a.KW, a.stride, a.pad, a.act, a.post_scale = 64, 3, 3, 2, 1, 1, 1.0
    a.out1_dev = s3.data_ptr(); a.out1_ld, a.out1_off, a.split = 64, 0, 64
    m.in_dev = s3.data_ptr(); m.N, m.Hi, m.Wi, m.Cin, m.in_ld, m.in_off = N, Ho, Wo, 64, 64, 0
    m.wgt_dev, m.bias_dev = devw[1][0].data_ptr(), devw[1][1].data_ptr()
    m.Cout, m.KH, m.KW, m.stride, m.pad, m.act, m.post_scale = 64, 1, 1, 1, 0, 1, 1.0
    m.out1_dev = main.data_ptr(); m.out1_ld, m.out1_off, m.split = 32, 0, 32
    m.out2_dev = cat.data_ptr(); m.out2_ld, m.out2_off = 64, 32
    c.in_dev = main.data_ptr(); c.N, c.Hi, c.Wi, c.Cin, c.in_ld, c.in_off = N, Ho, Wo, 32, 32, 0
    c.wgt_dev, c.bias_dev = devw[2][0].data_ptr(), devw[2][1].data_ptr()
    c.Cout, c.KH, c.KW, c.stride, c.pad, c.act, c.post_scale = 32, 1, 1, 1, 0, 1, 1.0
    c.out1_dev = tmp.data_ptr(); c.out1_ld, c.out1_off, c.split = 32, 0, 32
    stream = _lib.current_stream()
    keep = (x, devw, frags, s3, main, cat, tmp, a, m, c)

    def fused(_keep=keep):
        check(lib.st_conv3x3s2_csp_front(C.byref(a), C.byref(m), C.byref(c), frags[0].data_ptr(), frags[1].data_ptr(), stream))

    def three(_keep=keep):
        for d in (a, m, c):
            check(lib.st_conv2d_nhwc(C.byref(d), stream))

    return fused, three, 2.0 * N * Ho * Wo * (288 * 64 + 64 * 64 + 32 * 32) / 1e9


def bench(N, H, W, reps=10):
    fused, three, gf = make(N, H, W)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

    def timed(fn):
        fn()
        best = 1e9
        for _ in range(reps):
            e0.record(); fn(); e1.record(); e1.synchronize()
            best = min(best, e0.elapsed_time(e1))
        return best

    t_f, t_3 = timed(fused), timed(three)
    print(f'N={N} {H}x{W}: {gf:6.2f} GF  fused {t_f * 1e3:7.1f} us ({gf / t_f:6.1f} TF/s)   three launches (heuristic '
          f'tiles) {t_3 * 1e3:7.1f} us ({gf / t_3:6.1f} TF/s)')


if __name__ == '__main__':
    for shape in [(16, 368, 640), (8, 368, 640)]:
        bench(*shape)
