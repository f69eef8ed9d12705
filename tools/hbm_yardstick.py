#!/usr/bin/env python
"""What a plain streaming kernel reaches on this MI355X at the cost volume's traffic size (245 MB read + 90 MB written per
launch): torch elementwise kernels (vectorised, no reuse, nothing to compute) as the practical HBM yardstick next to
the 8 TB/s datasheet figure."""
import torch


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
    dev = torch.device('cuda', 0)
    n = 335_000_000 // 12          # three fp32 arrays of ~112 MB: 2 read + 1 written = 335 MB
    a, b, c = (torch.randn(n, device=dev) for _ in range(3))
    t = timeit(lambda: torch.add(a, b, out=c))
    print(f'add   (2 reads + 1 write, {12 * n / 1e6:.0f} MB): {t:7.1f} us = {12 * n / t / 1e6:6.2f} TB/s')
    t = timeit(lambda: c.copy_(a))
    print(f'copy  (1 read + 1 write,  {8 * n / 1e6:.0f} MB): {t:7.1f} us = {8 * n / t / 1e6:6.2f} TB/s')
    t = timeit(lambda: c.fill_(1.0))
    print(f'fill  (1 write,           {4 * n / 1e6:.0f} MB): {t:7.1f} us = {4 * n / t / 1e6:6.2f} TB/s')
    big = torch.randn(3 * n, device=dev)
    t = timeit(lambda: big.sum())
    print(f'sum   (1 read,            {12 * n / 1e6:.0f} MB): {t:7.1f} us = {12 * n / t / 1e6:6.2f} TB/s')


if __name__ == '__main__':
    main()
