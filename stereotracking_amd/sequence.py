"""Sequence-level drivers for BASELINE.json configs[2] / configs[3]:

  configs[2]  one synthetic sequence on one GPU: the dense path runs batched (B frames per launch
              plan), detections feed the CPU OC-SORT association in frame order
  configs[3]  one long sequence sharded over the ranks of a node: every rank runs the dense path on its
              contiguous chunk of frames, ONE all-gather of the fixed-size detection buffers (RCCL over
              xGMI with backend 'nccl'), then the tracker consumes all frames in order

Because the dense path is stateless per frame (reference mmtrack/models/mot/ocsort_disparity.py:73-83)
the gathered detections - and therefore the track ids - equal those of the sequential run.
"""
import numpy as np
import torch

from . import dist as sdist
from .mot import scale_bbox
from .structures import InstanceData, TrackDataSample


def synthetic_sequence(num_frames=64, num_objects=6, height=720, width=1280, max_disp=192, seed=0):
    """`num_objects` rectangles (8-60 px) moving with constant velocity + noise
    over a textured background, per-frame disparity consistent with per-object depth 5-60 m
    (disp = 0.25 * 640 / Z).  Yields dicts with left/right uint8 (3,H,W), disp float32 (H,W), gt boxes."""
    rng = np.random.RandomState(seed)
    bg = rng.randint(0, 256, size=(3, height, width + max_disp)).astype(np.uint8)
    tex = rng.randint(0, 256, size=(num_objects, 3, 64, 64)).astype(np.uint8)
    pos = rng.uniform([100, 80], [width - 100, height - 80], (num_objects, 2))
    vel = rng.uniform(-3, 3, (num_objects, 2))
    size = rng.randint(8, 61, (num_objects, 2))
    depth = rng.uniform(5, 60, num_objects)
    bg_disp = 2
    for t in range(num_frames):
        left = bg[:, :, bg_disp:bg_disp + width].copy()   # background at a small constant disparity
        right = bg[:, :, :width].copy()
        disp = np.full((height, width), float(bg_disp), np.float32)
        boxes = []
        order = np.argsort(-depth)  # far objects first, near ones overwrite them
        p = pos + vel * t + rng.normal(0, 0.3, (num_objects, 2))
        for k in order:
            w, h = int(size[k, 0]), int(size[k, 1])
            x1, y1 = int(round(p[k, 0] - w / 2)), int(round(p[k, 1] - h / 2))
            d = int(round(0.25 * 640 / depth[k]))
            if x1 - d < 0 or y1 < 0 or x1 + w > width or y1 + h > height:
                continue
            patch = tex[k][:, :h, :w]
            left[:, y1:y1 + h, x1:x1 + w] = patch
            right[:, y1:y1 + h, x1 - d:x1 - d + w] = patch
            disp[y1:y1 + h, x1:x1 + w] = d
            boxes.append((k, x1, y1, x1 + w, y1 + h, depth[k]))
        yield dict(frame_id=t, left=left, right=right, disp=disp, gt=np.array(boxes, np.float32).reshape(-1, 6))


def frames_to_batch(frames, device, use_right=True):
    """List of frame dicts -> padded float tensors in the detector's input layout."""
    from .synthetic import pad_to_divisor
    img = np.stack([pad_to_divisor(f['left'].astype(np.float32), 32, 114.0) for f in frames])
    out = dict(img=torch.from_numpy(img).to(device))
    if use_right:
        right = np.stack([pad_to_divisor(f['right'].astype(np.float32), 32, 114.0) for f in frames])
        out['right'] = torch.from_numpy(right).to(device)
    else:
        d = np.stack([np.repeat(pad_to_divisor(f['disp'], 32, 0.0)[None], 3, 0) for f in frames])
        out['disp_postp'] = torch.from_numpy(d).to(device)
    return out


def detect_shard(pipe, frames, device, check_overflow=True):
    """Dense path over this rank's frames, `pipe.batch` frames per launch plan.  `pipe` is a StereoDensePipeline
    (strictly serial batches) or an InflightPipelines runner (consecutive batches overlap on its streams).
    -> (F_pad, max_det + 1, 8) frame records (header row + SCALED boxes, what the tracker consumes), counts
    (true counts; 0 for batch padding).  Raises DetectionOverflow if a frame kept more than max_det boxes
    (check_overflow=False defers that to unpack_frame, after a collective, so every rank raises together)."""
    runner = pipe if hasattr(pipe, 'submit') else None
    one = runner.pipes[0] if runner is not None else pipe
    B = one.batch
    bufs = []

    def pack(out, n_real):
        return one.pack_detections(out, scaled=True, n_real=n_real)   # fresh tensor

    for i in range(0, len(frames), B):
        chunk = list(frames[i:i + B])
        n_real = len(chunk)
        chunk = chunk + [chunk[-1]] * (B - n_real)  # pad the last batch with a repeated frame
        batch = frames_to_batch(chunk, device, use_right=one.stereo)
        if runner is not None:   # packed under the context's stream, before that context is reused
            det, _ = runner.submit(batch['img'], right=batch.get('right'), disp_postp=batch.get('disp_postp'),
                                   post=lambda out, ctx, n=n_real: pack(out, n))
        else:
            det = pack(pipe.run(batch['img'], right=batch.get('right'), disp_postp=batch.get('disp_postp')), n_real)
        bufs.append(det)
    if runner is not None:
        runner.synchronize()
    records = torch.cat(bufs)
    counts, cap = sdist.record_counts(records)
    if check_overflow and bool((counts > cap).any()):   # one host sync per shard, after all batches were enqueued
        bad = torch.nonzero(counts > cap).flatten().tolist()
        raise sdist.DetectionOverflow(f'frames {bad} kept {counts[bad].tolist()} boxes, detection buffer has {cap} '
                                      f'rows: build the pipeline with a larger max_det')
    return records, counts.to(torch.int32)


def track_gathered(dets, counts, num_frames, tracker, model):
    """CPU association over gathered frame records in frame order -> list of InstanceData per frame
    (boxes unscaled back, as OCSORT_Disparity.predict does at ocsort_disparity.py:95-97).  `counts` is
    ignored (the records carry their own; kept in the signature for callers that still pass it)."""
    dets = dets.cpu()
    results = []
    for t in range(num_frames):
        f = sdist.unpack_frame(dets[t], frame=t)
        sample = TrackDataSample(dict(frame_id=t))
        sample.pred_det_instances = InstanceData(**f)
        trk = tracker.track(model, None, None, sample)
        trk['bboxes'] = scale_bbox(trk.bboxes, 1 / trk.scales)
        results.append(trk)
    return results


def run_sharded_sequence(pipe, frames, tracker, model, device):
    """configs[3]: shard `frames` over the ranks, detect, all-gather once, track everywhere."""
    frames = list(frames)
    T = len(frames)
    start, stop, chunk = sdist.shard_frames(T)
    B = (pipe.pipes[0] if hasattr(pipe, 'submit') else pipe).batch
    per_rank = (chunk + B - 1) // B * B   # equal padded length on every rank
    mine = frames[start:stop]
    if mine:
        dets, _ = detect_shard(pipe, mine, device, check_overflow=False)   # raised after the gather, on every rank
    else:
        dets = torch.zeros(0, pipe.max_det + 1, 8, device=device)
    pad = per_rank - dets.shape[0]
    if pad:
        dets = torch.cat([dets, dets.new_zeros(pad, pipe.max_det + 1, 8)])
    all_dets = sdist.gather_detections(dets)   # ONE collective: the records carry their own counts
    _, world = sdist.world()
    # drop the per-rank padding: rank r holds frames [r*chunk, r*chunk + chunk) in its first `chunk` slots
    idx = torch.cat([torch.arange(r * per_rank, r * per_rank + chunk) for r in range(world)])[:T]
    return track_gathered(all_dets[idx.to(all_dets.device)], None, T, tracker, model)
