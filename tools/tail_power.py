#!/usr/bin/env python
"""Socket power of the fused stage-1 CSP tail (st_conv3x3_csp_tail) against the two launches it replaces, each run back to
back for ~2.5 s while `rocm-smi` is sampled (the method of tools/kernel_power.py): time, power, clock and the ENERGY per
launch above idle - the currency the power-limited in-flight loop pays in.  N = 16 (RGB branch) and N = 8 with
the two-branch average, 184 x 320."""
import ctypes as C
import os
import re
import subprocess
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from stereotracking_amd import _lib  # noqa: E402
from stereotracking_amd._lib import StConvDesc, check, ptr  # noqa: E402

lib = _lib.load()
dev = torch.device('cuda:0')


def smi():
    out = subprocess.run(['rocm-smi', '--showclocks', '--showpower'], capture_output=True, text=True).stdout
    sclk = re.search(r'sclk clock level: \d+: \((\d+)Mhz\)', out)
    pw = re.search(r'Power \(W\): ([0-9.]+)', out)
    return (int(sclk.group(1)) if sclk else 0, float(pw.group(1)) if pw else 0.0)


def pack(w, bias):
    cout, cin, kh, kw = w.shape
    wp = torch.empty(lib.st_conv_packed_floats(cout, cin, kh, kw), dtype=torch.float32)
    bp = torch.empty((cout + 31) // 32 * 32, dtype=torch.float32)
    check(lib.st_conv_pack_weights(ptr(w.contiguous()), ptr(bias), None, None, None, None, 1e-3, cout, cin, kh, kw, ptr(wp), ptr(bp)))
    return wp, bp


def sustained(fn, seconds=2.5):
    """fn back to back in bursts of ~150 ms, `rocm-smi` sampled WHILE a burst executes (as tools/kernel_power.py);
    -> (us per call, mean W, mean MHz), the first two bursts dropped"""
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); fn(); e1.record(); e1.synchronize()
    per = max(e0.elapsed_time(e1), 0.01)
    nl = max(20, int(150.0 / per))
    ws, cs, us = [], [], []
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        e0.record()
        for _ in range(nl):
            fn()
        e1.record()
        c, w = smi()
        e1.synchronize()
        us.append(e0.elapsed_time(e1) / nl * 1e3); ws.append(w); cs.append(c)
    ws, cs, us = ws[2:] or ws, cs[2:] or cs, us[2:] or us
    return sum(us) / len(us), sum(ws) / len(ws), sum(cs) / len(cs)


def main():
    time.sleep(1.0)
    idle = sum(smi()[1] for _ in range(5)) / 5
    print(f'idle: {idle:.0f} W')
    H, W = 184, 320
    g = torch.Generator().manual_seed(0)
    w2 = torch.randn(32, 32, 3, 3, generator=g) / (3.0 * 32 ** 0.5)
    wf = torch.randn(64, 64, 1, 1, generator=g) / 8.0
    b2, bf = torch.randn(32, generator=g) * 0.1, torch.randn(64, generator=g) * 0.1
    wp2, bp2 = pack(w2, b2)
    wpf, bpf = pack(wf, bf)
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

        def fused():
            check(lib.st_conv3x3_csp_tail(C.byref(c2), C.byref(f), ptr(frag), None))

        def pair():
            check(lib.st_conv2d_nhwc_variant(C.byref(c2), None, 43))
            check(lib.st_conv2d_nhwc_variant(C.byref(f), None, 46))

        rows = []
        for name, fn in (('fused tail (variant 56)', fused), ('Winograd conv2 + resident 1x1 (43 + 46)', pair)):
            us, w, mhz = sustained(fn)
            rows.append((name, us, w, mhz, (w - idle) * us * 1e-3))
            time.sleep(0.5)
        for name, us, w, mhz, mj in rows:
            print(f'N={N:2d} avg={int(avg)}  {name:42s} {us:7.1f} us  {w:6.0f} W  sclk {mhz:5.0f}  {mj:6.1f} mJ per call above idle', flush=True)
        print(f'        energy ratio fused / pair: {rows[0][4] / rows[1][4]:.3f}   time ratio {rows[0][1] / rows[1][1]:.3f}', flush=True)


if __name__ == '__main__':
    main()
