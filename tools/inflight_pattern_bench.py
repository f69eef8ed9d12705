#!/usr/bin/env python
"""Why does the MOT shell's chunk loop run the dense path slower than bench.py's free-running loop?  Same contexts,
same weights, four submission patterns (ms per 8-pair chunk):
  A  free-running, the same 8 pairs every step (bench.py's timed loop)
  B  free-running, 64 distinct frame slots (inputs stream from HBM instead of the Infinity Cache)
  C  B + the shell's post step (13-column records + pinned D2H)
  D  C + the shell's consume pattern: wait for the oldest chunk, ~1.9 ms of host work, one small launch on its
     stream, then resubmit that context"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from stereotracking_amd.pipeline import InflightPipelines  # noqa: E402
from stereotracking_amd.synthetic import synthetic_batch, synthetic_state_dict  # noqa: E402

dev = torch.device('cuda', 0)
B, F, NCTX = 8, 64, int(os.environ.get('INFLIGHT', 3))
runner = InflightPipelines(NCTX, B, (720, 1280), 0.5, 0.33, 1, stereo=True, max_disp=192, max_det=1000, agg_layers=2)
runner.load_state_dict(synthetic_state_dict(runner.param_table(), seed=0))
bc = synthetic_batch(list(range(B)), 720, 1280, 192)
img8, right8 = bc['img'].to(dev), bc['right'].to(dev)
img64 = img8.repeat(F // B, 1, 1, 1).contiguous()
right64 = right8.repeat(F // B, 1, 1, 1).contiguous()
pinned = [torch.empty(B, 1001, 13, pin_memory=True) for _ in range(NCTX)]


def post_shell(out, ctx):
    rec = runner.pipes[ctx].pack_detections(out, scaled='both', n_real=B)
    pinned[ctx].copy_(rec, non_blocking=True)
    return out


def free_running(chunks, distinct, post):
    for i in range(chunks):
        s = (i * B) % F
        a, b = (img64[s:s + B], right64[s:s + B]) if distinct else (img8, right8)
        runner.submit(a, b, post=post)
    runner.synchronize()


def shell_pattern(chunks, host_ms=1.9):
    jobs = []
    for i in range(min(NCTX, chunks)):
        s = (i * B) % F
        jobs.append(runner.submit(img64[s:s + B], right64[s:s + B], post=post_shell))
    for i in range(chunks):
        out, ev = jobs[i]
        ev.synchronize()
        t = time.perf_counter()
        while time.perf_counter() - t < host_ms * 1e-3:
            pass
        with torch.cuda.stream(runner.streams[i % NCTX]):
            out['disp_postp'].sum()
        if i + NCTX < chunks:
            s = ((i + NCTX) * B) % F
            jobs.append(runner.submit(img64[s:s + B], right64[s:s + B], post=post_shell))
    runner.synchronize()


def timed(fn, chunks=64):
    fn(8)
    torch.cuda.synchronize()
    t = time.perf_counter()
    fn(chunks)
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / chunks * 1e3


print('A free-running, same 8 pairs      %.3f ms/chunk' % timed(lambda n: free_running(n, False, None)))
print('B free-running, 64 distinct slots %.3f ms/chunk' % timed(lambda n: free_running(n, True, None)))
print('C B + shell post (records + D2H)  %.3f ms/chunk' % timed(lambda n: free_running(n, True, post_shell)))
print('D shell consume pattern           %.3f ms/chunk' % timed(shell_pattern))
print('D with no host work               %.3f ms/chunk' % timed(lambda n: shell_pattern(n, 0.0)))
