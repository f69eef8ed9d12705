#!/usr/bin/env python
"""Co-residency stress: VICTIM kernels (exact-fp32 instances: Winograd 43 / 44, resident 1x1 46, implicit GEMM, cost
volume) run on some streams while the split-operand (bf16 MFMA) instances run on others; every victim result is compared
bit for bit with the result it gives alone.  (Found this way: v_pk_fma_f32 with op_sel goes wrong beside bf16 MFMAs.)"""
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
    def __init__(self, seed, N, H, W, Cin, Cout, k=1, stride=1, res=False):
        torch.manual_seed(seed)
        self.x = torch.randn(N, H, W, Cin, device=dev)
        w = torch.randn(Cout, Cin, k, k) / (k * Cin ** 0.5)
        b = torch.randn(Cout) * 0.1
        wp = torch.empty(lib.st_conv_packed_floats(Cout, Cin, k, k))
        bp = torch.empty((Cout + 31) // 32 * 32)
        check(lib.st_conv_pack_weights(ptr(w), ptr(b), None, None, None, None, 0.0, Cout, Cin, k, k, ptr(wp), ptr(bp)))
        self.wp, self.bp = wp.to(dev), bp.to(dev)
        self.wn = None
        if k == 3 and stride == 1 and lib.st_wino_packed_floats(Cout, Cin):
            wn = torch.empty(lib.st_wino_packed_floats(Cout, Cin))
            check(lib.st_wino_pack_weights(ptr(wp), Cout, Cin, ptr(wn)))
            self.wn = wn.to(dev)
        Ho, Wo = (H + 2 * (k // 2) - k) // stride + 1, (W + 2 * (k // 2) - k) // stride + 1
        self.out = torch.empty(N, Ho, Wo, Cout, device=dev)
        self.r = torch.randn(N, Ho, Wo, Cout, device=dev) if res else None
        d = StConvDesc()
        d.in_dev = self.x.data_ptr(); d.N, d.Hi, d.Wi, d.Cin, d.in_ld, d.in_off = N, H, W, Cin, Cin, 0
        d.wgt_dev = self.wp.data_ptr(); d.bias_dev = self.bp.data_ptr()
        d.wgt_wino_dev = self.wn.data_ptr() if self.wn is not None else None
        d.Cout, d.KH, d.KW, d.stride, d.pad = Cout, k, k, stride, k // 2
        d.out1_dev = self.out.data_ptr(); d.out1_ld, d.out1_off, d.split = Cout, 0, Cout
        if res:
            d.res_dev = self.r.data_ptr(); d.res_ld, d.res_off = Cout, 0
        d.post_scale, d.act = (0.5 if res else 1.0), 1
        self.d = d

    def launch(self, v, stream):
        check(lib.st_conv2d_nhwc_variant(C.byref(self.d), C.c_void_p(stream.cuda_stream), v))


victims = [(Layer(1, 8, 92, 160, 128, 128, 3), 43), (Layer(2, 8, 46, 80, 128, 256, 3), 44), (Layer(3, 8, 92, 160, 64, 64, 3, res=True), 43),
           (Layer(4, 8, 184, 320, 32, 32, 3, res=True), 43), (Layer(5, 8, 92, 160, 128, 128), 46), (Layer(6, 8, 46, 80, 256, 256, 3, 2), 7),
           (Layer(7, 8, 184, 320, 48, 48, 3), 43)]
aggr = [Layer(11, 8, 92, 160, 256, 128), Layer(12, 8, 46, 80, 512, 256)]
vs = [torch.cuda.Stream() for _ in victims]
as_ = [torch.cuda.Stream() for _ in aggr]
refs = []
for L, v in victims:
    L.launch(v, torch.cuda.current_stream())
    torch.cuda.synchronize()
    refs.append(L.out.clone())
SPLITV = [int(v) for v in (sys.argv[1:] or ['53', '54', '50', '55'])]
bad = [0] * len(victims)
for rep in range(60):
    for L, _ in victims:
        L.out.fill_(float('nan'))
    torch.cuda.synchronize()
    for k in range(2):
        for (L, s) in zip(aggr, as_):
            for v in SPLITV:
                L.launch(v, s)
        for (L, v), s in zip(victims, vs):
            L.launch(v, s)
    torch.cuda.synchronize()
    for i, ((L, v), r) in enumerate(zip(victims, refs)):
        if not torch.equal(L.out, r):
            bad[i] += 1
print('victim mismatches (of 60 reps):', {f'victim{i} variant {v}': b for i, ((L, v), b) in enumerate(zip(victims, bad))})
