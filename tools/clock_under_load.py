#!/usr/bin/env python
"""Shader clock and power while the fused front kernel (dense fp32 MFMA on every CU) runs back to back: launches it
for a few seconds and samples `rocm-smi --showclocks --showpower` from a child process meanwhile; also times a single
cold launch against the steady-state average (a power-limited clock shows up as the difference)."""
import os
import subprocess
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from front_bench import make  # noqa: E402

fused, _, gf = make(16, 368, 640)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
fused(); torch.cuda.synchronize()
time.sleep(1.0)
e0.record(); fused(); e1.record(); e1.synchronize()
print(f'single launch after 1 s idle: {e0.elapsed_time(e1) * 1e3:.1f} us')
print(subprocess.run(['rocm-smi', '--showclocks', '--showpower'], capture_output=True, text=True).stdout[-1500:])
t0 = time.time()
n = 0
samples = []
while time.time() - t0 < 4.0:
    e0.record()
    for _ in range(200):
        fused()
    e1.record()
    p = subprocess.run(['rocm-smi', '--showclocks', '--showpower'], capture_output=True, text=True).stdout
    e1.synchronize()
    samples.append((e0.elapsed_time(e1) / 200 * 1e3, p))
    n += 200
for us, p in samples[:2] + samples[-2:]:
    lines = [ln for ln in p.splitlines() if 'sclk' in ln or 'Power' in ln or 'mclk' in ln]
    print(f'{us:.1f} us/launch ({gf / us * 1e3:.1f} TF/s)  ' + ' | '.join(ln.strip() for ln in lines))
