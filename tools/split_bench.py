#!/usr/bin/env python
"""Layer shapes of the path: best exact-fp32 implicit-GEMM / Winograd instance against the split-operand (bf16x3)
instances 50-52, time and direct-convolution TFLOP/s.   usage: python tools/split_bench.py [reps]"""
import ctypes as C
import os
import sys

import torch

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# the split-operand (bf16x3) instances are parked in the TOOLS build of the library (round 5)
os.environ.setdefault('ST_LIBRARY', os.path.join(_ROOT, 'stereotracking_amd', 'lib', 'libstereotrack_hip_ablation.so'))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stereotracking_amd import _lib  # noqa: E402
from stereotracking_amd._lib import StConvDesc, check, ptr  # noqa: E402

lib = _lib.load()
dev = torch.device('cuda:0')
REPS = int(sys.argv[1]) if len(sys.argv) > 1 else 8


def bench(name, N, H, W, Cin, Cout, k=1, stride=1):
    x = torch.randn(N, H, W, Cin, device=dev)
    w = torch.randn(Cout, Cin, k, k) / (k * Cin ** 0.5)
    b = torch.randn(Cout) * 0.1
    wp = torch.empty(lib.st_conv_packed_floats(Cout, Cin, k, k))
    bp = torch.empty((Cout + 31) // 32 * 32)
    check(lib.st_conv_pack_weights(ptr(w), ptr(b), None, None, None, None, 0.0, Cout, Cin, k, k, ptr(wp), ptr(bp)))
    wpd, bpd = wp.to(dev), bp.to(dev)
    wn = None
    if k == 3 and stride == 1 and lib.st_wino_packed_floats(Cout, Cin):
        wn = torch.empty(lib.st_wino_packed_floats(Cout, Cin))
        check(lib.st_wino_pack_weights(ptr(wp), Cout, Cin, ptr(wn)))
        wn = wn.to(dev)
    Ho, Wo = (H + 2 * (k // 2) - k) // stride + 1, (W + 2 * (k // 2) - k) // stride + 1
    out = torch.empty(N, Ho, Wo, Cout, device=dev)
    d = StConvDesc()
    d.in_dev = x.data_ptr(); d.N, d.Hi, d.Wi, d.Cin, d.in_ld, d.in_off = N, H, W, Cin, Cin, 0
    d.wgt_dev = wpd.data_ptr(); d.bias_dev = bpd.data_ptr()
    d.Cout, d.KH, d.KW, d.stride, d.pad = Cout, k, k, stride, k // 2
    d.out1_dev = out.data_ptr(); d.out1_ld, d.out1_off, d.split = Cout, 0, Cout
    d.post_scale, d.act = 1.0, 1
    d.wgt_wino_dev = wn.data_ptr() if wn is not None else None
    stream = _lib.current_stream()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    gf = 2.0 * N * Ho * Wo * k * k * Cin * Cout / 1e9

    def t(v):
        if lib.st_conv2d_nhwc_variant(C.byref(d), stream, v) != 0:
            return None
        best = 1e9
        for _ in range(REPS):
            e0.record()
            lib.st_conv2d_nhwc_variant(C.byref(d), stream, v)
            e1.record()
            e1.synchronize()
            best = min(best, e0.elapsed_time(e1))
        return best
    fp32 = sorted((t(v), lib.st_conv_variant_name(v).decode()) for v in list(range(22)) + [43, 44, 46] if t(v) is not None)
    sp = [(t(v), lib.st_conv_variant_name(v).decode()) for v in (50, 51, 52, 53, 54, 55)]
    sp = sorted(s for s in sp if s[0] is not None)
    line = f'{name:34s} {gf:6.2f} GF  fp32 best {fp32[0][1]:11s} {fp32[0][0] * 1e3:6.1f} us {gf / fp32[0][0]:5.0f} TF/s'
    if sp:
        line += f'   split best {sp[0][1]:13s} {sp[0][0] * 1e3:6.1f} us {gf / sp[0][0]:5.0f} TF/s  x{fp32[0][0] / sp[0][0]:.2f}   (' + \
                ', '.join(f'{n} {v * 1e3:.0f}' for v, n in sp) + ')'
    print(line, flush=True)


SHAPES = [('op13 3x3s2 64->128 @184x320', 8, 184, 320, 64, 128, 3, 2), ('op22 3x3s2 128->256 @92x160', 8, 92, 160, 128, 256, 3, 2),
          ('op31 3x3s2 256->512 @46x80', 8, 46, 80, 256, 512, 3, 2), ('op49 3x3s2 128->128 @92x160', 8, 92, 160, 128, 128, 3, 2),
          ('op54 3x3s2 256->256 @46x80', 8, 46, 80, 256, 256, 3, 2),
          ('op34 1x1 1024->512 @23x40', 8, 23, 40, 1024, 512), ('op40 1x1 512->256 @46x80', 8, 46, 80, 512, 256),
          ('op23 1x1 256->256 @46x80', 8, 46, 80, 256, 256), ('op35 1x1 512->512 @23x40', 8, 23, 40, 512, 512),
          ('op32 1x1 512->256 @23x40', 8, 23, 40, 512, 256), ('op24 1x1 128->128 @46x80', 8, 46, 80, 128, 128),
          ('op45 1x1 256->128 @92x160', 8, 92, 160, 256, 128), ('op48 1x1 128->128 @92x160', 8, 92, 160, 128, 128),
          ('op17 1x1 64->64 @92x160', 8, 92, 160, 64, 64), ('op6 1x1 64->64 @184x320', 16, 184, 320, 64, 64),
          ('op62 3x3 128->256 @92x160', 8, 92, 160, 128, 256, 3, 1), ('op63 3x3 128->128 @92x160', 8, 92, 160, 128, 128, 3, 1),
          ('op65 3x3 128->256 @46x80', 8, 46, 80, 128, 256, 3, 1), ('op69 3x3 128->128 @23x40', 8, 23, 40, 128, 128, 3, 1),
          ('op16 3x3 64->64 @92x160', 8, 92, 160, 64, 64, 3, 1), ('op25 3x3 128->128 @46x80', 8, 46, 80, 128, 128, 3, 1),
          ('op37 3x3 256->256 @23x40', 8, 23, 40, 256, 256, 3, 1)]
for s in SHAPES:
    bench(*s)
