#!/usr/bin/env python
"""Shader clock and socket power while the bench workload runs, serialized (1 context) and in flight (4 contexts): is
the in-flight loop power-limited?  The loop runs for a few seconds per setting; `rocm-smi --showclocks --showpower` is
sampled from the host meanwhile (the submit loop runs ahead of the device by a bounded number of steps).
usage: python tools/inflight_power.py [seconds]"""
import os
import re
import subprocess
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stereotracking_amd.pipeline import InflightPipelines  # noqa: E402
from stereotracking_amd.synthetic import synthetic_batch, synthetic_state_dict  # noqa: E402

SECONDS = float(sys.argv[1]) if len(sys.argv) > 1 else 6.0
dev = torch.device('cuda:0')
B = 8
batches = []
for j in range(4):
    b = synthetic_batch([100000 * j + i for i in range(B)], 720, 1280, 192)
    batches.append((b['img'].to(dev), b['right'].to(dev)))


def smi():
    out = subprocess.run(['rocm-smi', '--showclocks', '--showpower'], capture_output=True, text=True).stdout
    sclk = re.search(r'sclk clock level: \d+: \((\d+)Mhz\)', out)
    mclk = re.search(r'mclk clock level: \d+: \((\d+)Mhz\)', out)
    pw = re.search(r'Power \(W\): ([0-9.]+)', out)
    return (int(sclk.group(1)) if sclk else None, int(mclk.group(1)) if mclk else None, float(pw.group(1)) if pw else None)


print('idle:', smi())
for n_ctx in (1, 4, 1, 4):
    runner = InflightPipelines(n_ctx, B, (720, 1280), 0.5, 0.33, 1, stereo=True, max_disp=192, max_det=1000, agg_layers=2)
    runner.load_state_dict(synthetic_state_dict(runner.param_table(), seed=0))
    for i in range(10):
        runner.submit(*batches[i % 4])
    runner.synchronize()
    samples, steps = [], 0
    t0 = time.perf_counter()
    last = None
    while time.perf_counter() - t0 < SECONDS:
        for _ in range(24):                      # ~100 ms of device work queued ahead
            out, ev = runner.submit(*batches[steps % 4])
            steps += 1
        samples.append(smi())
        ev.synchronize()
    runner.synchronize()
    dt = time.perf_counter() - t0
    sc = sorted(s[0] for s in samples if s[0]); pw = sorted(s[2] for s in samples if s[2])
    print(f'{n_ctx} context(s): {B * steps / dt:7.1f} pairs/s over {dt:.1f} s | sclk MHz min / median / max {sc[0]} / {sc[len(sc) // 2]} / {sc[-1]}'
          f' | socket power W min / median / max {pw[0]:.0f} / {pw[len(pw) // 2]:.0f} / {pw[-1]:.0f} | mclk {samples[-1][1]} | {len(samples)} samples')
    del runner
    torch.cuda.empty_cache()
    time.sleep(1.0)
print(subprocess.run(['rocm-smi', '--showmaxpower'], capture_output=True, text=True).stdout[-600:])
