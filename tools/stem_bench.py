#!/usr/bin/env python
"""Fused stem at the bench geometry (8 x 720 x 1280 uint8 frames -> 736 x 1280 -> 32 ch @ 368 x 640):
fp32 input via LDS-DMA (+ the st_pack_raw_frames pass that produces it) vs the raw uint8 staging."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stereotracking_amd import _lib  # noqa: E402
from stereotracking_amd._lib import check, ptr  # noqa: E402


def timeit(fn, n=50):
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
    lib = _lib.load()
    dev = torch.device('cuda', 0)
    N, h, w, H, W, cout = 8, 720, 1280, 736, 1280, 32
    frames = [torch.randint(0, 256, (3, h, w), dtype=torch.uint8, device=dev) for _ in range(N)]
    table = (C.c_void_p * N)(*[f.data_ptr() for f in frames])
    wp = torch.randn(lib.st_stem_packed_floats(cout), device=dev) / 100
    bp = torch.zeros(32, device=dev)
    x = torch.empty(N, 3, H, W, device=dev)
    out = torch.empty(N, H // 2, W // 2, cout, device=dev)
    t_pack = timeit(lambda: check(lib.st_pack_raw_frames(table, N, h, w, H, W, 114.0, ptr(x), None)))
    t_f32 = timeit(lambda: check(lib.st_stem_focus_conv(ptr(x), N, H, W, 3, ptr(wp), ptr(bp), cout, ptr(out), cout, 0, 1, None)))
    t_u8 = timeit(lambda: check(lib.st_stem_focus_conv_u8(table, N, h, w, H, W, 114.0, ptr(wp), ptr(bp), cout, ptr(out), cout, 0, 1, None)))
    print(f'pack {t_pack:.1f} us   stem(fp32, LDS-DMA) {t_f32:.1f} us   stem(uint8 frames) {t_u8:.1f} us')


if __name__ == '__main__':
    main()
