#!/usr/bin/env python
"""Experiment: K steps of the bench workload on ONE stream vs alternating between TWO pipelines on two streams
(kernels of consecutive batches may overlap: launch tails and the latency-bound decode / NMS / depth kernels of
one batch are filled by the convs of the other)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stereotracking_amd.pipeline import StereoDensePipeline  # noqa: E402
from stereotracking_amd.synthetic import synthetic_batch, synthetic_state_dict  # noqa: E402

B, steps = 8, int(sys.argv[1]) if len(sys.argv) > 1 else 20
nstreams = int(sys.argv[2]) if len(sys.argv) > 2 else 2
dev = torch.device('cuda:0')
cache = os.environ.get('ST_TUNE_CACHE')
pipes = []
for i in range(nstreams):
    p = StereoDensePipeline(B, (720, 1280), 0.5, 0.33, 1, stereo=True, max_disp=192, max_det=300, agg_layers=2)
    if i == 0:
        sd = synthetic_state_dict(p.param_table(), seed=0)
    p.load_state_dict(sd, tuning_cache=cache)
    pipes.append(p)
batch = synthetic_batch(list(range(B)), 720, 1280, 192)
img, right = batch['img'].to(dev), batch['right'].to(dev)
streams = [torch.cuda.Stream() for _ in range(nstreams)]


def run(n, k):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(k):
        j = i % n
        with torch.cuda.stream(streams[j]):
            out = pipes[j].run(img, right)
            pipes[j].pack_detections(out)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / k * 1e3


for n in range(1, nstreams + 1):
    run(n, 6)
    ms = run(n, steps)
    print(f'{n} stream(s): {ms:.3f} ms/step  {B / ms * 1e3:.1f} pairs/s', flush=True)
