#!/usr/bin/env python
"""Kernel-level timing of the fused stage-1 CSP tail (st_conv3x3_csp_tail, csrc/wino_csp_tail.hip) against the two
launches it replaces (Winograd conv2 + identity -> 1x1 final conv), at the benched shapes: N = 16 (left | right RGB
branch) and N = 8 with the two-branch average, 184 x 320 maps.  HIP events around 20 back-to-back launches."""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from stereotracking_amd import _lib  # noqa: E402
from stereotracking_amd._lib import StConvDesc, check, ptr  # noqa: E402


def pack(lib, w, bias):
    cout, cin, kh, kw = w.shape
    wp = torch.empty(lib.st_conv_packed_floats(cout, cin, kh, kw), dtype=torch.float32)
    bp = torch.empty((cout + 31) // 32 * 32, dtype=torch.float32)
    check(lib.st_conv_pack_weights(ptr(w.contiguous()), ptr(bias), None, None, None, None, 1e-3, cout, cin, kh, kw, ptr(wp), ptr(bp)))
    return wp, bp


def main():
    lib = _lib.load()
    dev = torch.device('cuda:0')
    H, W = 184, 320
    g = torch.Generator().manual_seed(0)
    w2 = torch.randn(32, 32, 3, 3, generator=g) / (3.0 * 32 ** 0.5)
    wf = torch.randn(64, 64, 1, 1, generator=g) / 8.0
    b2, bf = torch.randn(32, generator=g) * 0.1, torch.randn(64, generator=g) * 0.1
    wp2, bp2 = pack(lib, w2, b2)
    wpf, bpf = pack(lib, wf, bf)
    wino = torch.empty(lib.st_wino_packed_floats(32, 32), dtype=torch.float32)
    check(lib.st_wino_pack_weights(ptr(wp2), 32, 32, ptr(wino)))
    frag = torch.empty(lib.st_csp_tail_frag_floats(), dtype=torch.float32)
    check(lib.st_csp_tail_pack_frags(ptr(wpf), ptr(frag)))
    wp2, bp2, wino, wpf, bpf, frag = (t.to(dev) for t in (wp2, bp2, wino, wpf, bpf, frag))
    for N, avg in ((16, False), (8, True)):
        tmp = torch.randn(N, H, W, 36, device=dev)
        main_ = torch.randn(N, H, W, 32, device=dev)
        cat = torch.randn(N, H, W, 64, device=dev)
        other = torch.randn(N, H, W, 64, device=dev)
        out = torch.empty(N, H, W, 64, device=dev)
        c2, f = StConvDesc(), StConvDesc()
        c2.in_dev = tmp.data_ptr(); c2.N, c2.Hi, c2.Wi, c2.Cin, c2.in_ld, c2.in_off = N, H, W, 32, 36, 4
        c2.wgt_dev = wp2.data_ptr(); c2.bias_dev = bp2.data_ptr(); c2.wgt_wino_dev = wino.data_ptr()
        c2.Cout, c2.KH, c2.KW, c2.stride, c2.pad = 32, 3, 3, 1, 1
        c2.out1_dev = cat.data_ptr(); c2.out1_ld, c2.out1_off, c2.split = 64, 0, 32
        c2.res_dev = main_.data_ptr(); c2.res_ld, c2.res_off = 32, 0
        c2.post_scale, c2.act = 1.0, 1
        f.in_dev = cat.data_ptr(); f.N, f.Hi, f.Wi, f.Cin, f.in_ld, f.in_off = N, H, W, 64, 64, 0
        f.wgt_dev = wpf.data_ptr(); f.bias_dev = bpf.data_ptr()
        f.Cout, f.KH, f.KW, f.stride, f.pad = 64, 1, 1, 1, 0
        f.out1_dev = out.data_ptr(); f.out1_ld, f.out1_off, f.split = 64, 0, 64
        if avg:
            f.res_dev = other.data_ptr(); f.res_ld, f.res_off = 64, 0
        f.post_scale, f.act = (0.5 if avg else 1.0), 1

        def timed(fn, reps=20):
            for _ in range(3):
                fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                fn()
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) / reps * 1e3

        fused = timed(lambda: check(lib.st_conv3x3_csp_tail(C.byref(c2), C.byref(f), ptr(frag), None)))
        conv2 = timed(lambda: check(lib.st_conv2d_nhwc_variant(C.byref(c2), None, 43)))
        final = timed(lambda: check(lib.st_conv2d_nhwc(C.byref(f), None)))
        gf = 2.0 * N * H * W * (9 * 32 * 32 / 2.25 + 64 * 64) / 1e9
        mb = 4.0 * N * H * W * (32 + 32 + 32 + 64 + (64 if avg else 0)) / 1e6
        print(f'N={N:2d} avg={int(avg)}: fused {fused:7.1f} us ({gf / fused * 1e3:6.1f} TF/s executed, {mb / fused * 1e3:6.0f} GB/s)   '
              f'conv2 {conv2:6.1f} + final {final:6.1f} = {conv2 + final:6.1f} us   ratio {fused / (conv2 + final):.3f}', flush=True)


if __name__ == '__main__':
    main()
