#!/usr/bin/env python
"""A/B of the MOT shell's loop options on ONE box: model.test_steps (primed loop) at 64 frames per call with
raw_stem on / off and queue_depth 1 / 2 / 3, `--repeats` runs each, interleaved; prints every run and the medians,
next to the bare 3- and 4-context pipeline loop of bench.py measured in the same process."""
import argparse
import os
import statistics
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--repeats', type=int, default=5)
    ap.add_argument('--calls', type=int, default=10)
    ap.add_argument('--frames-per-call', type=int, default=64)
    ap.add_argument('--inflight', type=int, nargs='+', default=[3])
    a = ap.parse_args()
    sys.argv = [sys.argv[0]]
    args = bench.parse()
    dev = torch.device('cuda', 0)
    torch.cuda.set_device(0)
    from stereotracking_amd import mot  # noqa: F401
    from stereotracking_amd.config import Config
    from stereotracking_amd.pipeline import InflightPipelines
    from stereotracking_amd.registry import MODELS
    from stereotracking_amd.structures import TrackDataSample
    from stereotracking_amd.synthetic import synthetic_batch, synthetic_state_dict
    B, F = args.batch, a.frames_per_call
    batch_cpu = synthetic_batch(list(range(B)), 720, 1280, args.max_disp)
    img, right = batch_cpu['img'].to(dev), batch_cpu['right'].to(dev)
    bare = {}
    sd = None
    for n in (3, 4):
        runner = InflightPipelines(n, B, (720, 1280), 0.5, 0.33, 1, stereo=True, max_disp=args.max_disp,
                                   max_det=args.max_det, agg_layers=args.agg_layers)
        sd = sd or synthetic_state_dict(runner.param_table(), seed=0)
        runner.load_state_dict(sd, tuning_cache=os.environ.get('ST_TUNE_CACHE'))
        for _ in range(20):
            runner.submit(img, right)
        runner.synchronize()
        vals = []
        for _ in range(a.repeats):
            t0 = time.perf_counter()
            for _ in range(a.calls * F // B):
                runner.submit(img, right)
            runner.synchronize()
            vals.append(a.calls * F / (time.perf_counter() - t0))
        bare[n] = statistics.median(vals)
        print(f'bare pipeline loop, {n} contexts: median {bare[n]:.1f} pairs/s  {[round(v) for v in vals]}', flush=True)
        del runner
    cfg = Config.fromfile(os.path.join(ROOT, 'configs', 'stereo_tracking', 'ocsort',
                                       'stereo_yolox_s_mot_airdrone_costvolume.py'))
    cfg.model.stereo['max_disp'] = args.max_disp
    cfg.model.stereo['agg_layers'] = args.agg_layers
    left = [batch_cpu['img'][i % B:i % B + 1, :, :720].to(torch.uint8).to(dev) for i in range(F)]
    rght = [batch_cpu['right'][i % B:i % B + 1, :, :720].to(torch.uint8).to(dev) for i in range(F)]
    for inflight in a.inflight:
        model = MODELS.build(dict(cfg.model, dense_batch=B, inflight=inflight, max_det=args.max_det,
                                  tuning_cache=os.environ.get('ST_TUNE_CACHE')))
        model.detector.load_state_dict({k: v for k, v in sd.items() if not k.startswith('stereo.')}, strict=False)
        model.stereo.load_state_dict({k[len('stereo.'):]: v for k, v in sd.items() if k.startswith('stereo.')})
        frame = [0]

        def data():
            samples = [TrackDataSample(dict(frame_id=frame[0] + i, ori_shape=(720, 1280), img_shape=(720, 1280),
                                            scale_factor=(1.0, 1.0))) for i in range(F)]
            frame[0] += F
            return dict(inputs=dict(img=left, right=rght), data_samples=samples)

        def run(raw, depth, primed=True):
            model.raw_stem, model.queue_depth = raw, depth
            for ent in model._dense.values():
                for p in ent[0].pipes:
                    p.disp_buffers = depth + 1
            t0 = time.perf_counter()
            if primed:
                n = sum(len(o) for o in model.test_steps(data() for _ in range(a.calls)))
            else:
                n = sum(len(model.test_step(data())) for _ in range(a.calls))
            torch.cuda.synchronize()
            return n / (time.perf_counter() - t0)

        configs = [(True, 1, True), (False, 1, True), (True, 2, True), (True, 3, True), (True, 1, False), (True, 2, False)]
        for c in configs:
            run(*c)
        res = {c: [] for c in configs}
        for _ in range(a.repeats):
            for c in configs:
                res[c].append(run(*c))
        for c in configs:
            med = statistics.median(res[c])
            print(f'shell inflight={inflight} raw_stem={c[0]!s:5} queue_depth={c[1]} {"primed loop" if c[2] else "per-call   "}: '
                  f'median {med:.1f} pairs/s = {med / bare[4]:.3f} of bare-4 ({med / bare[3]:.3f} of bare-3)  '
                  f'{[round(v) for v in res[c]]}', flush=True)
        del model


if __name__ == '__main__':
    main()
