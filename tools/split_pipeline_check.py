#!/usr/bin/env python
"""Does a pipeline whose plan holds split-operand instances give the same head when its contexts run DIFFERENT batches
concurrently as when each batch runs alone?  And how far is it from the exact-fp32 plan?"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# the split-operand (bf16x3) instances are parked in the TOOLS build of the library (round 5)
os.environ.setdefault('ST_LIBRARY', os.path.join(_ROOT, 'stereotracking_amd', 'lib', 'libstereotrack_hip_ablation.so'))
sys.path.insert(0, ROOT)
from stereotracking_amd.pipeline import InflightPipelines  # noqa: E402
from stereotracking_amd.synthetic import synthetic_batch, synthetic_state_dict  # noqa: E402

dev = torch.device('cuda:0')
B = 8
run_split = InflightPipelines(3, B, (720, 1280), 0.5, 0.33, 1, stereo=True, max_disp=192, agg_layers=2, split_bf16=True)
sd = synthetic_state_dict(run_split.param_table(), seed=0)
run_split.load_state_dict(sd)
print('plan', run_split.pipes[0].det.get_tuning(), 'agg', run_split.pipes[0].stereo_module.variant, flush=True)
batches = [synthetic_batch([10 * j + i for i in range(B)], 720, 1280, 192) for j in range(3)]
imgs = [(b['img'].to(dev), b['right'].to(dev)) for b in batches]
alone = []
for j, (a, r) in enumerate(imgs):
    out = run_split.pipes[0].run(a, r)
    torch.cuda.synchronize()
    alone.append((out['head'].clone(), out['disp_postp'].clone()))
for rep in range(5):
    outs = []
    for j, (a, r) in enumerate(imgs):
        o, ev = run_split.submit(a, r, post=lambda out, ctx: dict(head=out['head'].clone(), disp=out['disp_postp'].clone()))
        outs.append(o)
    run_split.synchronize()
    for j, o in enumerate(outs):
        eh = (o['head'] - alone[j][0]).abs().max().item()
        ed = (o['disp'] - alone[j][1]).abs().max().item()
        print(f'rep {rep} batch {j}: concurrent vs alone: head max|d| {eh:.3e}  disp max|d| {ed:.3e}', flush=True)
ref = InflightPipelines(1, B, (720, 1280), 0.5, 0.33, 1, stereo=True, max_disp=192, agg_layers=2, split_bf16=False)
ref.load_state_dict(sd)
for j, (a, r) in enumerate(imgs):
    out = ref.pipes[0].run(a, r)
    torch.cuda.synchronize()
    h = out['head']
    e = ((h - alone[j][0]).abs() / h.abs().clamp(min=1.0)).max().item()
    print(f'batch {j}: split plan vs exact-fp32 plan: head max rel {e:.3e}', flush=True)
