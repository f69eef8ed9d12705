#!/usr/bin/env python
"""Runs a few `model.test_step` calls exactly as bench.py's test_step leg does (for `rocprofv3 --kernel-trace`), and
prints the host-side timings.  Analyse the trace with tools/trace_gaps.py."""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--calls', type=int, default=3)
    ap.add_argument('--frames-per-call', type=int, default=64)
    ap.add_argument('--inflight', type=int, default=3)
    a = ap.parse_args()
    sys.argv = [sys.argv[0], '--frames-per-call', str(a.frames_per_call), '--inflight', str(a.inflight)]
    args = bench.parse()
    dev = torch.device('cuda', 0)
    torch.cuda.set_device(0)
    from stereotracking_amd.pipeline import InflightPipelines
    from stereotracking_amd.synthetic import synthetic_batch, synthetic_state_dict
    runner = InflightPipelines(1, args.batch, (720, 1280), 0.5, 0.33, 1, stereo=True, max_disp=args.max_disp,
                               max_det=args.max_det, agg_layers=args.agg_layers)
    sd = synthetic_state_dict(runner.param_table(), seed=0)
    del runner
    batch_cpu = synthetic_batch(list(range(args.batch)), 720, 1280, args.max_disp)
    t0 = time.perf_counter()
    r = bench.test_step_leg(args, sd, batch_cpu, dev, a.calls * a.frames_per_call)
    print(r, f'({time.perf_counter() - t0:.1f} s incl. build + autotune)')


if __name__ == '__main__':
    main()
