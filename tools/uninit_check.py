#!/usr/bin/env python
"""Does any kernel of the plan read workspace memory that no earlier kernel of the same forward wrote?  The workspace is
poisoned with NaN before a forward; the head / disparity must equal the un-poisoned run bit for bit."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from stereotracking_amd.pipeline import StereoDensePipeline  # noqa: E402
from stereotracking_amd.synthetic import synthetic_batch, synthetic_state_dict  # noqa: E402

dev = torch.device('cuda:0')
B = 8
for split in (False, True):
    pipe = StereoDensePipeline(B, (720, 1280), 0.5, 0.33, 1, stereo=True, max_disp=192, agg_layers=2, split_bf16=split)
    sd = synthetic_state_dict(pipe.param_table(), seed=0)
    pipe.load_state_dict(sd)
    b = synthetic_batch(list(range(B)), 720, 1280, 192)
    img, right = b['img'].to(dev), b['right'].to(dev)
    out = pipe.run(img, right)
    torch.cuda.synchronize()
    head0, disp0 = out['head'].clone(), out['disp_postp'].clone()
    for poison in (float('nan'), 1e30, -3.0):
        pipe.det._ws.view(torch.float32).fill_(poison)
        for t in pipe.stereo_module._vol or ():
            t.fill_(poison)
        out = pipe.run(img, right)
        torch.cuda.synchronize()
        h, d = out['head'], out['disp_postp']
        print(f'split={split} poison={poison}: head equal {torch.equal(h, head0)} (nan {int(torch.isnan(h).sum())}, '
              f'max|d| {(h - head0).abs().nan_to_num(9e9).max().item():.3e})  disp equal {torch.equal(d, disp0)}', flush=True)
    print('plan', pipe.det.get_tuning(), flush=True)
