#!/usr/bin/env python
"""Persistent Winograd instance (tile variant 57) against the per-block instance (43) on the path's 3x3 / stride-1 layer
shapes; outputs must be bit-identical.  Median of 15 launches each, interleaved."""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault('ST_LIBRARY', os.path.join(ROOT, 'stereotracking_amd', 'lib', 'libstereotrack_hip_ablation.so'))   # variant 57 lives in the tools build
from stereotracking_amd import _lib  # noqa: E402
from stereotracking_amd._lib import StConvDesc, check, ptr  # noqa: E402

lib = _lib.load()
dev = torch.device('cuda:0')


def bench(N, H, W, Cin, Cout, res, act=1, reps=15):
    x = torch.randn(N, H, W, Cin, device=dev)
    w = torch.randn(Cout, Cin, 3, 3) / (3 * Cin ** 0.5)
    b = torch.randn(Cout) * 0.1
    wp = torch.empty(lib.st_conv_packed_floats(Cout, Cin, 3, 3))
    bp = torch.empty((Cout + 31) // 32 * 32)
    check(lib.st_conv_pack_weights(ptr(w), ptr(b), None, None, None, None, 0.0, Cout, Cin, 3, 3, ptr(wp), ptr(bp)))
    wn = torch.empty(lib.st_wino_packed_floats(Cout, Cin))
    check(lib.st_wino_pack_weights(ptr(wp), Cout, Cin, ptr(wn)))
    wpd, bpd, wnd = wp.to(dev), bp.to(dev), wn.to(dev)
    r = torch.randn(N, H, W, Cout, device=dev) if res else None
    outs = {}
    d = StConvDesc()
    d.in_dev = x.data_ptr(); d.N, d.Hi, d.Wi, d.Cin, d.in_ld, d.in_off = N, H, W, Cin, Cin, 0
    d.wgt_dev = wpd.data_ptr(); d.bias_dev = bpd.data_ptr(); d.wgt_wino_dev = wnd.data_ptr()
    d.Cout, d.KH, d.KW, d.stride, d.pad = Cout, 3, 3, 1, 1
    if res:
        d.res_dev = r.data_ptr(); d.res_ld, d.res_off = Cout, 0
    d.post_scale, d.act = 1.0, act
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t = {43: [], 57: []}
    for v in (43, 57):
        outs[v] = torch.full((N, H, W, Cout), float('nan'), device=dev)
    for it in range(reps + 2):
        for v in (43, 57):
            d.out1_dev = outs[v].data_ptr(); d.out1_ld, d.out1_off, d.split = Cout, 0, Cout
            e0.record()
            check(lib.st_conv2d_nhwc_variant(C.byref(d), None, v), f'variant {v}')
            e1.record()
            e1.synchronize()
            if it >= 2:
                t[v].append(e0.elapsed_time(e1) * 1e3)
    same = torch.equal(outs[43], outs[57])
    m = {v: sorted(t[v])[len(t[v]) // 2] for v in t}
    gf = 2.0 * N * H * W * 9 * Cin * Cout / 1e9
    print(f'N={N:2d} {H:3d}x{W:3d} {Cin:3d}->{Cout:3d}{" +res" if res else "     "}: v43 {m[43]:7.1f} us  v57 {m[57]:7.1f} us  '
          f'x{m[43] / m[57]:.3f}  ({gf / m[57] * 1e3 / 2.25:5.1f} TF/s executed)  bit-identical: {same}', flush=True)


for shape in [(8, 92, 160, 64, 64, True), (8, 46, 80, 128, 128, True), (8, 46, 80, 128, 128, False), (8, 92, 160, 64, 64, False),
              (8, 184, 320, 48, 48, False), (8, 23, 40, 256, 256, False), (8, 92, 160, 128, 256, False),
              (8, 92, 160, 128, 128, False)]:
    bench(*shape)
