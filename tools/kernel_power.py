#!/usr/bin/env python
"""Power per kernel: representative conv layers of the path run back to back for ~2.5 s each while `rocm-smi` is
sampled - socket power, shader clock, direct-equivalent and EXECUTED TFLOP/s, and the energy per executed GFLOP above
idle.  Ranks the kernels by what the power-limited in-flight loop pays for them."""
import ctypes as C
import os
import re
import subprocess
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stereotracking_amd import _lib  # noqa: E402
from stereotracking_amd._lib import StConvDesc, check, ptr  # noqa: E402

lib = _lib.load()
dev = torch.device('cuda:0')


def smi():
    out = subprocess.run(['rocm-smi', '--showclocks', '--showpower'], capture_output=True, text=True).stdout
    sclk = re.search(r'sclk clock level: \d+: \((\d+)Mhz\)', out)
    pw = re.search(r'Power \(W\): ([0-9.]+)', out)
    return (int(sclk.group(1)) if sclk else 0, float(pw.group(1)) if pw else 0.0)


def layer(N, H, W, Cin, Cout, k, stride, res=False):
    x = torch.randn(N, H, W, Cin, device=dev)
    w = torch.randn(Cout, Cin, k, k) / (k * Cin ** 0.5)
    b = torch.randn(Cout) * 0.1
    wp = torch.empty(lib.st_conv_packed_floats(Cout, Cin, k, k))
    bp = torch.empty((Cout + 31) // 32 * 32)
    check(lib.st_conv_pack_weights(ptr(w), ptr(b), None, None, None, None, 0.0, Cout, Cin, k, k, ptr(wp), ptr(bp)))
    keep = [x, wp.to(dev), bp.to(dev)]
    Ho, Wo = (H + 2 * (k // 2) - k) // stride + 1, (W + 2 * (k // 2) - k) // stride + 1
    out = torch.empty(N, Ho, Wo, Cout, device=dev)
    d = StConvDesc()
    d.in_dev = x.data_ptr(); d.N, d.Hi, d.Wi, d.Cin, d.in_ld, d.in_off = N, H, W, Cin, Cin, 0
    d.wgt_dev = keep[1].data_ptr(); d.bias_dev = keep[2].data_ptr()
    if k == 3 and stride == 1 and lib.st_wino_packed_floats(Cout, Cin):
        wn = torch.empty(lib.st_wino_packed_floats(Cout, Cin))
        check(lib.st_wino_pack_weights(ptr(wp), Cout, Cin, ptr(wn)))
        keep.append(wn.to(dev))
        d.wgt_wino_dev = keep[-1].data_ptr()
    d.Cout, d.KH, d.KW, d.stride, d.pad = Cout, k, k, stride, k // 2
    d.out1_dev = out.data_ptr(); d.out1_ld, d.out1_off, d.split = Cout, 0, Cout
    if res:
        r = torch.randn(N, Ho, Wo, Cout, device=dev)
        keep.append(r)
        d.res_dev = r.data_ptr(); d.res_ld, d.res_off = Cout, 0
    d.post_scale, d.act = 1.0, 1
    keep.append(out)
    return d, keep, 2.0 * N * Ho * Wo * k * k * Cin * Cout / 1e9


CASES = [  # name, layer args, variant, executed / direct flops
    ('Winograd 128->256 @92x160 (head conv0, level 0)', (8, 92, 160, 128, 256, 3, 1), 43, 1 / 2.25),
    ('Winograd 64->64 +res @92x160 (stage-2 bottleneck)', (8, 92, 160, 64, 64, 3, 1, True), 43, 1 / 2.25),
    ('Winograd 32->32 +res @184x320 (stage-1 conv2)', (16, 184, 320, 32, 32, 3, 1, True), 43, 1 / 2.25),
    ('implicit GEMM 3x3/s2 64->128 @184x320 (128x128 dma)', (8, 184, 320, 64, 128, 3, 2), -1, 1.0),
    ('implicit GEMM 3x3/s2 128->256 @92x160', (8, 92, 160, 128, 256, 3, 2), -1, 1.0),
    ('resident 1x1 64->64 @184x320 (stage-1 final conv)', (16, 184, 320, 64, 64, 1, 1), 46, 1.0),
    ('implicit GEMM 1x1 128->128 @46x80 (64x64 tile)', (8, 46, 80, 128, 128, 1, 1), -1, 1.0),
    ('implicit GEMM 1x1 512->256 @46x80', (8, 46, 80, 512, 256, 1, 1), -1, 1.0),
]
idle_w = smi()[1]
print(f'idle: {idle_w:.0f} W')
stream = _lib.current_stream()
for name, args, variant, exec_frac in CASES:
    d, keep, gf = layer(*args)
    call = (lambda: lib.st_conv2d_nhwc_variant(C.byref(d), stream, variant)) if variant >= 0 else (lambda: lib.st_conv2d_nhwc(C.byref(d), stream))
    assert call() == 0, name
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); call(); e1.record(); e1.synchronize()
    per = max(e0.elapsed_time(e1), 0.01)
    nl = max(20, int(150.0 / per))          # ~150 ms of launches per sample
    ws, cs, us = [], [], []
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 2.5:
        e0.record()
        for _ in range(nl):
            call()
        e1.record()
        c, w = smi()
        e1.synchronize()
        us.append(e0.elapsed_time(e1) / nl * 1e3); ws.append(w); cs.append(c)
    ws, cs, us = ws[2:] or ws, cs[2:] or cs, us[2:] or us
    w, c, u = sum(ws) / len(ws), sum(cs) / len(cs), sum(us) / len(us)
    tf = gf / (u * 1e-6) / 1e3          # GFLOP per launch / us per launch -> TFLOP/s
    print(f'{name:54s} {u:7.1f} us  {tf:6.1f} TF/s direct ({tf * exec_frac:5.1f} executed)  {w:5.0f} W  sclk {c:.0f}  '
          f'{(w - idle_w) * u * 1e-6 / (gf * exec_frac) * 1e3:6.2f} mJ per executed GFLOP above idle')
    del keep
