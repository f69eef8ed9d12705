#!/usr/bin/env python
"""Yardstick for the 1x1 layers: the vendor fp32 GEMM (torch.mm -> hipBLASLt / rocBLAS, fp32 in, fp32 accumulate) at the
GEMM shapes of the path's pointwise convolutions (M = pixels of 8 images, K = Cin, N = Cout), next to this library's
autotuned instance with its fused bias + SiLU epilogue.  Not part of the product: no library GEMM is on the hot path."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stereotracking_amd import _lib  # noqa: E402
from stereotracking_amd._lib import StConvDesc, check, ptr  # noqa: E402


def timeit(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def main():
    torch.backends.cuda.matmul.allow_tf32 = False
    lib = _lib.load()
    dev = torch.device('cuda', 0)
    shapes = [('op35 23x40 512->512', 7360, 512, 512), ('op36 23x40 256->256', 7360, 256, 256),
              ('op34 23x40 1024->512', 7360, 1024, 512), ('op23 46x80 256->256', 29440, 256, 256),
              ('op24 46x80 128->128', 29440, 128, 128), ('op40 46x80 512->256', 29440, 512, 256),
              ('op45 92x160 256->128', 117760, 256, 128), ('op14 92x160 128->128', 117760, 128, 128)]
    for name, M, K, N in shapes:
        a = torch.randn(M, K, device=dev)
        b = torch.randn(K, N, device=dev)
        out = torch.empty(M, N, device=dev)
        t_lib = timeit(lambda: torch.mm(a, b, out=out))
        w = torch.randn(N, K, 1, 1) / K ** 0.5
        bias = torch.zeros(N)
        wp = torch.empty(lib.st_conv_packed_floats(N, K, 1, 1))
        bp = torch.empty((N + 31) // 32 * 32)
        check(lib.st_conv_pack_weights(ptr(w), ptr(bias), None, None, None, None, 0.0, N, K, 1, 1, ptr(wp), ptr(bp)))
        wpd, bpd = wp.to(dev), bp.to(dev)
        d = StConvDesc()
        d.in_dev = a.data_ptr(); d.N, d.Hi, d.Wi, d.Cin, d.in_ld, d.in_off = 1, 1, M, K, K, 0
        d.wgt_dev = wpd.data_ptr(); d.bias_dev = bpd.data_ptr()
        d.Cout, d.KH, d.KW, d.stride, d.pad = N, 1, 1, 1, 0
        d.out1_dev = out.data_ptr(); d.out1_ld, d.out1_off, d.split = N, 0, N
        d.post_scale, d.act = 1.0, 1
        best = None
        for v in list(range(22)) + [41, 46]:
            if lib.st_conv2d_nhwc_variant(C.byref(d), _lib.current_stream(), v) != 0:
                continue
            t = timeit(lambda: lib.st_conv2d_nhwc_variant(C.byref(d), _lib.current_stream(), v), 20)
            if best is None or t < best[0]:
                best = (t, v)
        gf = 2.0 * M * K * N / 1e9
        print(f'{name:24s} M={M:6d} K={K:4d} N={N:3d}  vendor GEMM {t_lib:6.1f} us = {gf / t_lib * 1e3:6.1f} TF/s   '
              f'this library (instance {best[1]:2d}, + bias + SiLU) {best[0]:6.1f} us = {gf / best[0] * 1e3:6.1f} TF/s', flush=True)


if __name__ == '__main__' and '--conv-only' not in sys.argv:
    main()


def conv_yardstick():
    """3x3 layers: torch conv2d (MIOpen, fp32, channels_last, benchmark mode) vs this library's best instance."""
    import torch.nn.functional as F
    torch.backends.cudnn.benchmark = True
    torch.backends.cudnn.allow_tf32 = False
    lib = _lib.load()
    dev = torch.device('cuda', 0)
    shapes = [('head 3x3 128->256 @92x160', 8, 92, 160, 128, 256, 1), ('head 3x3 128->128 @92x160', 8, 92, 160, 128, 128, 1),
              ('csp 3x3 128->128 @46x80', 8, 46, 80, 128, 128, 1), ('csp 3x3 64->64 @92x160', 8, 92, 160, 64, 64, 1),
              ('csp 3x3 32->32 @184x320 x16', 16, 184, 320, 32, 32, 1), ('csp 3x3 256->256 @23x40', 8, 23, 40, 256, 256, 1),
              ('down 3x3s2 64->128 @184x320', 8, 184, 320, 64, 128, 2), ('down 3x3s2 128->256 @92x160', 8, 92, 160, 128, 256, 2),
              ('down 3x3s2 256->512 @46x80', 8, 46, 80, 256, 512, 2)]
    for name, N, H, W, Cin, Cout, s in shapes:
        x = torch.randn(N, H, W, Cin, device=dev)
        w = torch.randn(Cout, Cin, 3, 3) / (3 * Cin ** 0.5)
        xt = x.permute(0, 3, 1, 2)      # NCHW view of NHWC memory = channels_last
        wt = w.to(dev).contiguous(memory_format=torch.channels_last)
        try:
            t_lib = timeit(lambda: F.conv2d(xt, wt, None, s, 1), 20)
        except Exception as e:      # MIOpen without a usable solver for the shape
            t_lib = float('nan')
            print(f'{name}: MIOpen failed: {e!r}'[:200], flush=True)
        Ho, Wo = (H + 2 - 3) // s + 1, (W + 2 - 3) // s + 1
        out = torch.empty(N, Ho, Wo, Cout, device=dev)
        bias = torch.zeros(Cout)
        wp = torch.empty(lib.st_conv_packed_floats(Cout, Cin, 3, 3))
        bp = torch.empty((Cout + 31) // 32 * 32)
        check(lib.st_conv_pack_weights(ptr(w), ptr(bias), None, None, None, None, 0.0, Cout, Cin, 3, 3, ptr(wp), ptr(bp)))
        wpd, bpd = wp.to(dev), bp.to(dev)
        d = StConvDesc()
        d.in_dev = x.data_ptr(); d.N, d.Hi, d.Wi, d.Cin, d.in_ld, d.in_off = N, H, W, Cin, Cin, 0
        d.wgt_dev = wpd.data_ptr(); d.bias_dev = bpd.data_ptr()
        d.Cout, d.KH, d.KW, d.stride, d.pad = Cout, 3, 3, s, 1
        d.out1_dev = out.data_ptr(); d.out1_ld, d.out1_off, d.split = Cout, 0, Cout
        d.post_scale, d.act = 1.0, 1
        wn = None
        if s == 1:      # Winograd-form weights for instances 43 / 44
            wn = torch.empty(lib.st_wino_packed_floats(Cout, Cin), device='cpu')
            check(lib.st_wino_pack_weights(ptr(wp), Cout, Cin, ptr(wn)))      # from the packed (BN-folded) weights
            wn = wn.to(dev)
            d.wgt_wino_dev = wn.data_ptr()
        best = None
        for v in list(range(22)) + [42, 43, 44]:
            if lib.st_conv2d_nhwc_variant(C.byref(d), _lib.current_stream(), v) != 0:
                continue
            t = timeit(lambda: lib.st_conv2d_nhwc_variant(C.byref(d), _lib.current_stream(), v), 10)
            if best is None or t < best[0]:
                best = (t, v)
        gf = 2.0 * N * Ho * Wo * 9 * Cin * Cout / 1e9
        print(f'{name:30s} {gf:6.2f} GF  MIOpen conv2d {t_lib:7.1f} us = {gf / t_lib * 1e3:6.1f} TF/s   this library (instance '
              f'{best[1]:2d}, + bias + SiLU) {best[0]:7.1f} us = {gf / best[0] * 1e3:6.1f} TF/s (direct-conv flops)', flush=True)


if __name__ == '__main__' and '--conv' in sys.argv:
    conv_yardstick()
