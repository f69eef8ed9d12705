#!/usr/bin/env python
"""Concurrency stress of the split-operand conv instances: three HIP streams run different layers at the same time, every
result is compared with the exact-fp32 instance's (computed alone)."""
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


class Layer:
    def __init__(self, seed, N, H, W, Cin, Cout, k=1, stride=1):
        torch.manual_seed(seed)
        self.x = torch.randn(N, H, W, Cin, device=dev)
        w = torch.randn(Cout, Cin, k, k) / (k * Cin ** 0.5)
        b = torch.randn(Cout) * 0.1
        wp = torch.empty(lib.st_conv_packed_floats(Cout, Cin, k, k))
        bp = torch.empty((Cout + 31) // 32 * 32)
        check(lib.st_conv_pack_weights(ptr(w), ptr(b), None, None, None, None, 0.0, Cout, Cin, k, k, ptr(wp), ptr(bp)))
        self.wp, self.bp = wp.to(dev), bp.to(dev)
        Ho, Wo = (H + 2 * (k // 2) - k) // stride + 1, (W + 2 * (k // 2) - k) // stride + 1
        self.out = torch.empty(N, Ho, Wo, Cout, device=dev)
        d = StConvDesc()
        d.in_dev = self.x.data_ptr(); d.N, d.Hi, d.Wi, d.Cin, d.in_ld, d.in_off = N, H, W, Cin, Cin, 0
        d.wgt_dev = self.wp.data_ptr(); d.bias_dev = self.bp.data_ptr()
        d.Cout, d.KH, d.KW, d.stride, d.pad = Cout, k, k, stride, k // 2
        d.out1_dev = self.out.data_ptr(); d.out1_ld, d.out1_off, d.split = Cout, 0, Cout
        d.post_scale, d.act = 1.0, 1
        self.d = d

    def launch(self, v, stream):
        check(lib.st_conv2d_nhwc_variant(C.byref(self.d), C.c_void_p(stream.cuda_stream), v))


layers = [Layer(1, 16, 184, 320, 64, 64), Layer(2, 8, 184, 320, 64, 64), Layer(3, 8, 92, 160, 256, 128),
          Layer(4, 16, 184, 320, 64, 64), Layer(5, 8, 46, 80, 256, 256, 3, 2), Layer(6, 8, 184, 320, 64, 128, 3, 2)]
streams = [torch.cuda.Stream() for _ in layers]
refs = []
for L in layers:
    L.launch(3, torch.cuda.current_stream())
    torch.cuda.synchronize()
    refs.append(L.out.clone())
bad = 0
for v in (54, 51, 55, 52, 53, 50):
    for rep in range(40):
        for L, s in zip(layers, streams):
            L.out.fill_(float('nan'))
        torch.cuda.synchronize()
        for _ in range(3):
            for L, s in zip(layers, streams):
                L.launch(v, s)
        torch.cuda.synchronize()
        for i, (L, r) in enumerate(zip(layers, refs)):
            e = ((L.out - r).abs().max() / r.abs().max()).item()
            if not e <= 1e-4:
                bad += 1
                print(f'variant {v} rep {rep} layer {i}: err {e:.3e}', flush=True)
print('bad results:', bad)
