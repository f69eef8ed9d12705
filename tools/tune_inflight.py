#!/usr/bin/env python
"""Refine the tuning plan by what the BENCH measures: throughput of the in-flight pipeline.

st_detector_autotune ranks a layer's kernel instances by the wall time of one launch on an idle chip; with three
contexts in flight the chip is never idle (tools/trace_gaps.py: 100 % busy), and what an instance costs is the CU-time
it occupies, not how long its last workgroup takes.  This tool starts from the committed plan and, layer by layer,
tries the other valid instances, keeps a change when the timed in-flight loop gets faster by more than the noise, and
writes the refined plan into the tuning cache.   usage: python tools/tune_inflight.py [--passes 1] [--out cache.json]"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stereotracking_amd.pipeline import InflightPipelines, committed_tuning_plans  # noqa: E402
from stereotracking_amd.synthetic import synthetic_batch, synthetic_state_dict  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--passes', type=int, default=1)
ap.add_argument('--steps', type=int, default=24)
ap.add_argument('--reps', type=int, default=3)
ap.add_argument('--gain', type=float, default=0.004, help='relative improvement a change must show (noise floor)')
ap.add_argument('--out', default=committed_tuning_plans())
a = ap.parse_args()
dev = torch.device('cuda:0')
runner = InflightPipelines(3, 8, (720, 1280), 0.5, 0.33, 1, stereo=True, max_disp=192, agg_layers=2)
sd = synthetic_state_dict(runner.param_table(), seed=0)
runner.load_state_dict(sd)
b = synthetic_batch(list(range(8)), 720, 1280, 192)
img, right = b['img'].to(dev), b['right'].to(dev)
lib = runner.pipes[0].det.lib
nops = lib.st_detector_num_ops(runner.pipes[0].det.handle)


def set_plan(plan):
    for p in runner.pipes:
        p.det.set_tuning(plan)


def measure():
    best = 1e9
    for _ in range(a.reps):
        for _ in range(6):
            runner.submit(img, right)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            runner.submit(img, right)
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / a.steps)
    return best


def valid(op, v):
    """An instance is valid for a layer iff the detector accepts it in the plan and a step still runs."""
    plan = list(cur)
    plan[op] = v
    try:
        set_plan(plan)
        runner.submit(img, right)
        torch.cuda.synchronize()
        return True
    except Exception:
        return False


cur = runner.pipes[0].det.get_tuning()
set_plan(cur)
base = measure()
print(f'start: {base * 1e3:.4f} ms/step = {8 / base:.1f} pairs/s')
buf = C.create_string_buffer(512)
CAND = list(range(22)) + [41, 42, 43, 44, 46]
for ps in range(a.passes):
    changed = 0
    for op in range(nops):
        if cur[op] < 0 or cur[op] in (40, 45, 47):       # not a tunable conv (stem / fused front / head_pred / skipped)
            continue
        best_v, best_t = cur[op], base
        for v in CAND:
            if v == cur[op] or not valid(op, v):
                continue
            t = measure()
            if t < best_t * (1 - a.gain):
                best_v, best_t = v, t
        did = best_v != cur[op]
        if did:
            lib.st_detector_op_desc(runner.pipes[0].det.handle, op, buf, 512)
            print(f'op {op:2d} {buf.value.decode()[:70]}: {lib.st_conv_variant_name(cur[op]).decode()} -> '
                  f'{lib.st_conv_variant_name(best_v).decode()}  {base * 1e3:.4f} -> {best_t * 1e3:.4f} ms', flush=True)
            cur[op] = best_v
            changed += 1
        set_plan(cur)
        if did or op % 8 == 0:      # re-anchor against drift
            base = measure()
    print(f'pass {ps}: {changed} layers changed, {base * 1e3:.4f} ms/step = {8 / base:.1f} pairs/s', flush=True)
    if not changed:
        break
p0 = runner.pipes[0]
key = (f'v{lib.st_version()}_b{p0.batch}_{p0.height}x{p0.width}_w{p0.det.widen_factor:g}_d{p0.det.deepen_factor:g}'
       f'_s{int(p0.stereo)}_a{p0.agg_layers}_D{p0.D}_ops{nops}')
cache = json.load(open(a.out)) if os.path.exists(a.out) else {}
cache[key] = cur
cache.setdefault(key + '_agg', p0.stereo_module.variant)
json.dump(cache, open(a.out, 'w'), indent=0, sort_keys=True)
print('wrote', a.out, key)
