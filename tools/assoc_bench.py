#!/usr/bin/env python
"""Batched GPU association (f-4) against the native host tracker: us per sequence-frame.
Usage: python tools/assoc_bench.py [B ...]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stereotracking_amd.batched_assoc import BatchedGpuTracker  # noqa: E402
from stereotracking_amd.synthetic import synthetic_detection_stream  # noqa: E402
from stereotracking_amd.trackers import OCSORTTracker_Disparity  # noqa: E402

CFG = dict(obj_score_thr=0.3, init_track_thr=0.7, weight_iou_with_det_scores=False, match_iou_thr=0.1,
           num_tentatives=3, vel_consist_weight=0.2, vel_delta_t=3, num_frames_retain=30)
dev = torch.device('cuda:0')
T, M = 64, 16


def frames_of(det, T, M):
    d = np.zeros((T, M, 8), np.float32)
    c = np.zeros(T, np.int32)
    for t in range(T):
        r = det[det[:, 0] == t]
        k = len(r)
        d[t, :k, 0:4] = r[:, 1:5]
        d[t, :k, 4], d[t, :k, 6], d[t, :k, 7] = r[:, 5], r[:, 6], r[:, 7]
        c[t] = k
    return d, c


base = [frames_of(synthetic_detection_stream(200 + s, T=T, K=6), T, M) for s in range(32)]
# host: one native tracker, records of one sequence per call
rec = np.zeros((T, M + 1, 13), np.float32)
d0, c0 = base[0]
rec[:, 0, 0], rec[:, 0, 1], rec[:, 0, 2] = c0, M, 1
rec[:, 1:, 8:12], rec[:, 1:, 4:8] = d0[:, :, 0:4], d0[:, :, 4:8]
trk = OCSORTTracker_Disparity(**CFG)
n, t0 = 0, time.perf_counter()
while time.perf_counter() - t0 < 1.0:
    trk.track_records(list(range(T)), rec)
    n += T
host_us = (time.perf_counter() - t0) / n * 1e6
print(f'host native tracker (1 thread): {host_us:.1f} us per sequence-frame (6 objects)')
for B in [int(a) for a in sys.argv[1:]] or [64, 256, 1024, 4096]:
    dets = torch.from_numpy(np.stack([base[b % len(base)][0] for b in range(B)], 1)).to(dev)    # (T, B, M, 8)
    counts = torch.from_numpy(np.stack([base[b % len(base)][1] for b in range(B)], 1)).to(dev)  # (T, B)
    fids = [torch.full((B,), t, dtype=torch.int32, device=dev) for t in range(T)]
    g = BatchedGpuTracker(B, max_tracks=32, max_dets=M, device=dev, **CFG)
    for rep in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for t in range(T):
            g.step(fids[t], dets[t], counts[t], check_status=False)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    assert int(g.status.max()) == 0
    print(f'B={B:5d}: {dt / T * 1e3:7.3f} ms per step = {dt / T / B * 1e6:7.3f} us per sequence-frame '
          f'({host_us / (dt / T / B * 1e6):5.1f}x one host thread)')

# ---- dense sequences: hundreds of detections per frame (random-weight head), stress thresholds -----------------------
rng = np.random.RandomState(7)
Td, n_det, Md = 12, 300, 512
pos = rng.uniform([20, 20], [1260, 700], (n_det, 2)); vel = rng.uniform(-2, 2, (n_det, 2))
size = rng.uniform(12, 60, (n_det, 2)); score = rng.uniform(0.01, 0.9, n_det) ** 2
dd, cc = np.zeros((Td, Md, 8), np.float32), np.zeros(Td, np.int32)
for t in range(Td):
    p = pos + vel * t + rng.normal(0, 0.5, (n_det, 2))
    keep = rng.uniform(size=n_det) > 0.1
    b = np.concatenate([p - size / 2, p + size / 2], 1)[keep].astype(np.float32)
    sc = (score[keep] + rng.normal(0, 0.005, keep.sum())).clip(0.011, 0.99).astype(np.float32)
    o = np.argsort(-sc, kind='stable')
    k = len(o)
    dd[t, :k, 0:4], dd[t, :k, 4], dd[t, :k, 6], dd[t, :k, 7], cc[t] = b[o], sc[o], 20.0, 1.5, k
STRESS = dict(CFG, obj_score_thr=0.02, init_track_thr=0.05)
rec = np.zeros((Td, Md + 1, 13), np.float32)
rec[:, 0, 0], rec[:, 0, 1], rec[:, 0, 2] = cc, Md, 1
rec[:, 1:, 8:12], rec[:, 1:, 4:8] = dd[:, :, 0:4], dd[:, :, 4:8]
trk = OCSORTTracker_Disparity(**STRESS)
n, t0 = 0, time.perf_counter()
while time.perf_counter() - t0 < 1.0:
    trk.track_records(list(range(Td)), rec)
    n += Td
host_us = (time.perf_counter() - t0) / n * 1e6
print(f'dense (~270 detections, ~250 tracks per frame, stress thresholds): host {host_us:.0f} us per sequence-frame')
for B in (1, 16, 256):
    g = BatchedGpuTracker(B, max_tracks=1024, max_dets=Md, device=dev, **STRESS)
    dets = torch.from_numpy(np.repeat(dd[:, None], B, 1)).to(dev)
    counts = torch.from_numpy(np.repeat(cc[:, None], B, 1)).to(dev)
    fids = [torch.full((B,), t, dtype=torch.int32, device=dev) for t in range(Td)]
    for rep in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for t in range(Td):
            g.step(fids[t], dets[t], counts[t], check_status=False)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    assert int(g.status.max()) == 0
    print(f'  B={B:4d}: {dt / Td * 1e3:8.2f} ms per step = {dt / Td / B * 1e6:9.1f} us per sequence-frame')
