#!/usr/bin/env python
"""Is the fused front kernel limited by the data-dependent power draw of fp32 MFMA?  Same kernel, same launches
(back to back, steady state), three operand sets: random, all zeros, random with tiny magnitudes."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import front_bench as fb  # noqa: E402

e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for label, scale in (('random N(0,1) inputs', 1.0), ('all-zero inputs and weights', 0.0)):
    torch.manual_seed(0)
    orig_randn = torch.randn
    if scale == 0.0:
        torch.randn = lambda *a, **k: torch.zeros(*a, **k)
    fused, _, gf = fb.make(16, 368, 640)
    torch.randn = orig_randn
    for _ in range(20):
        fused()
    torch.cuda.synchronize()
    e0.record()
    for _ in range(200):
        fused()
    e1.record(); e1.synchronize()
    us = e0.elapsed_time(e1) / 200 * 1e3
    print(f'{label}: {us:.1f} us/launch  {gf / us * 1e3:.1f} TFLOP/s')
