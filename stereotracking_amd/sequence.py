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


def _smooth_texture(tex, k=3):
    """k x k box blur + contrast stretch of a uint8 (..., H, W) noise texture (as synthetic.synthetic_stereo_pair): white
    noise has no structure at the stride-4 scale the stereo module correlates; an image-like texture does."""
    if k <= 1:
        return tex
    t = tex.astype(np.float32)
    pad = k // 2
    tp = np.pad(t, [(0, 0)] * (t.ndim - 2) + [(pad, pad), (pad, pad)], mode='reflect')
    acc = np.zeros_like(t)
    H, W = t.shape[-2:]
    for dy in range(k):
        for dx in range(k):
            acc += tp[..., dy:dy + H, dx:dx + W]
    t = acc / (k * k)
    t = (t - t.mean()) * (k * 0.9) + 127.5
    return np.clip(np.rint(t), 0, 255).astype(np.uint8)


def synthetic_sequence(num_frames=64, num_objects=6, height=720, width=1280, max_disp=192, seed=0, smooth=3):
    """`num_objects` rectangles (8-60 px) moving with constant velocity + noise
    over a textured background, per-frame disparity consistent with per-object depth 5-60 m
    (disp = 0.25 * 640 / Z).  Yields dicts with left/right uint8 (3,H,W), disp float32 (H,W), gt boxes."""
    rng = np.random.RandomState(seed)
    bg = _smooth_texture(rng.randint(0, 256, size=(3, height, width + max_disp)).astype(np.uint8), smooth)
    tex = _smooth_texture(rng.randint(0, 256, size=(num_objects, 3, 64, 64)).astype(np.uint8), smooth)
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
    """List of frame dicts -> padded float tensors in the detector's input layout, converted on the HOST and uploaded
    as fp32 (15 MB per frame).  Kept for small tests; the sequence drivers upload raw bytes (RawFrameUploader)."""
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


def disparity_png_codes(disp):
    """float32 disparity px -> the uint16 PNG code the dataset stores (code = 16 * px, 65535 = invalid; reference
    mmtrack/datasets/transforms/loading_disparity.py:82,129-134), or None when 16 * px is not an integer <= 65534
    (such a map cannot have come from a PNG and is uploaded as fp32)."""
    c = disp.astype(np.float64) * 16.0
    if not (np.all(c == np.rint(c)) and c.min() >= 0 and c.max() <= 65534):
        return None
    return c.astype(np.uint16)


class HostSequence:
    """A sequence resident in PAGE-LOCKED host memory as the bytes a dataset decodes: left / right uint8
    (T,3,h,w) and, for the disparity-input configs, uint16 PNG codes (T,h,w).  Batches are uploaded straight
    from slices of these tensors (no staging copy)."""

    def __init__(self, frames, use_right=True):
        frames = list(frames)
        self.use_right = bool(use_right)
        self.left = torch.from_numpy(np.stack([f['left'] for f in frames])).pin_memory()
        self.right = self.codes = None
        if use_right:
            self.right = torch.from_numpy(np.stack([f['right'] for f in frames])).pin_memory()
        else:
            codes = [disparity_png_codes(f['disp']) for f in frames]
            if any(c is None for c in codes):
                raise ValueError('HostSequence: disparity maps are not PNG-representable (16 * px not integral)')
            self.codes = torch.from_numpy(np.stack(codes).view(np.int16)).pin_memory()
        self.gt = [f.get('gt') for f in frames]

    @classmethod
    def from_raw(cls, left, right=None, codes=None, gt=None, pin=True):
        """From decoded dataset bytes (datasets.load_video): left / right uint8 (T,3,h,w), codes uint16 (T,h,w) PNG
        disparity codes (65535 = invalid) - no float round trip."""
        self = cls.__new__(cls)
        mk = (lambda a: torch.from_numpy(np.ascontiguousarray(a)).pin_memory()) if pin else \
            (lambda a: torch.from_numpy(np.ascontiguousarray(a)))
        self.use_right = right is not None
        self.left = mk(np.asarray(left, np.uint8))
        self.right = mk(np.asarray(right, np.uint8)) if right is not None else None
        self.codes = mk(np.asarray(codes, np.uint16).view(np.int16)) if codes is not None else None
        if (self.right is None) == (self.codes is None):
            raise ValueError('HostSequence.from_raw takes either right images or disparity codes')
        self.gt = list(gt) if gt is not None else [None] * self.left.shape[0]
        return self

    def __len__(self):
        return self.left.shape[0]


class RawFrameUploader:
    """Host frames -> device batches in the detector's input layout, crossing PCIe as RAW BYTES (uint8 pixels /
    uint16 disparity codes: 4.7 MB per frame instead of 15.1 MB of fp32) and converted on the
    device by st_pack_raw_inputs (cast, x3 channel repeat, /16, invalid -> 0, pad 114 / 0: reference
    loading_disparity.py:82-134, transforms_disparity.py:234-249, data_preprocessor_disparity_v1.py:38-51).

    Two page-locked staging slots and two device byte buffers alternate; the H2D copies run on a stream of their
    own, ordered against the pack kernels by events, so the upload of batch i+1 overlaps the dense work of batch i
    and the host never waits for the device unless it is two batches ahead."""

    def __init__(self, batch, ori_hw, device, use_right=True, slots=2, raw_stem=None):
        self.B, (self.h, self.w) = int(batch), (int(ori_hw[0]), int(ori_hw[1]))
        # raw_stem (default on when the geometry allows): the uploaded uint8 frames are handed to the
        # pipeline as engine.RawChunk (stereo: left and right; disparity input: the image - the uint16 codes still go
        # through st_pack_raw_inputs, the disparity stem reads fp32) - the stem kernels cast + pad them while staging their windows, no pack
        # pass and no fp32 image in HBM.  Each batch then gets device byte buffers of its own (the stems read them
        # later, on a context's stream), the page-locked staging slots still alternate.
        self.raw_stem = bool(self.w % 4 == 0 and int(batch) <= 32) if raw_stem is None else bool(raw_stem)
        if self.raw_stem and not (self.w % 4 == 0 and int(batch) <= 32):
            raise ValueError('raw_stem needs width % 4 == 0 and batch <= 32')
        self.H, self.W = (self.h + 31) // 32 * 32, (self.w + 31) // 32 * 32
        self.dev, self.use_right = torch.device(device), bool(use_right)
        self.copy_stream = torch.cuda.Stream(device=self.dev)
        shape_img, shape_code = (self.B, 3, self.h, self.w), (self.B, self.h, self.w)
        self.slots = []
        for _ in range(slots):
            sl = dict(pin_left=torch.empty(shape_img, dtype=torch.uint8).pin_memory(),
                      dev_left=torch.empty(shape_img, dtype=torch.uint8, device=self.dev),
                      uploaded=torch.cuda.Event(), consumed=torch.cuda.Event(), used=False)
            if self.use_right:
                sl.update(pin_right=torch.empty(shape_img, dtype=torch.uint8).pin_memory(),
                          dev_right=torch.empty(shape_img, dtype=torch.uint8, device=self.dev))
            else:
                sl.update(pin_code=torch.empty(shape_code, dtype=torch.int16).pin_memory(),
                          dev_code=torch.empty(shape_code, dtype=torch.int16, device=self.dev))
            self.slots.append(sl)
        self._k = 0
        self._pool = None
        self.bytes_uploaded = 0

    def _stage(self, sl, frames):
        """list of frame dicts -> this slot's page-locked buffers.  The 2.8 MB per-frame copies run on a few host
        threads (memcpy releases the GIL): one thread moves ~5 GB/s, i.e. 9 ms per 8-pair batch against 4.7 ms of
        dense work."""
        n = len(frames)
        jobs = []
        for i in range(self.B):
            f = frames[min(i, n - 1)]      # a short last batch repeats its last frame (results ignored)
            jobs.append((sl['pin_left'][i], f['left']))
            if self.use_right:
                jobs.append((sl['pin_right'][i], f['right']))
            else:
                c = disparity_png_codes(f['disp'])
                if c is None:
                    raise ValueError('disparity map is not PNG-representable; use frames_to_batch')
                jobs.append((sl['pin_code'][i], c.view(np.int16)))
        if self._pool is None:
            from concurrent.futures import ThreadPoolExecutor
            self._pool = ThreadPoolExecutor(max_workers=4)
        list(self._pool.map(lambda j: j[0].copy_(torch.from_numpy(j[1])), jobs))
        return sl['pin_left'], sl.get('pin_right'), sl.get('pin_code')

    def upload(self, frames):
        """frames: list of <= B frame dicts, or (HostSequence, start, stop).  -> dict(img[, right | disp_postp]) fp32
        (B,3,H,W) device tensors (raw_stem: two engine.RawChunk of uint8 frames), produced on the CURRENT stream (a fresh allocation per batch: safe to hand to an
        in-flight context)."""
        from ._lib import check, current_stream, load, ptr
        sl = self.slots[self._k % len(self.slots)]
        self._k += 1
        if isinstance(frames, tuple):
            seq, a, b = frames
            idx = torch.arange(a, a + self.B).clamp_(max=b - 1)
            if b - a == self.B:        # straight from the resident page-locked sequence
                srcs = (seq.left[a:b], seq.right[a:b] if self.use_right else None,
                        None if self.use_right else seq.codes[a:b])
            else:                      # ragged last batch: gather (with the last frame repeated) into the slot
                if sl['used']:
                    sl['uploaded'].synchronize()
                sl['pin_left'].copy_(seq.left[idx])
                if self.use_right:
                    sl['pin_right'].copy_(seq.right[idx])
                else:
                    sl['pin_code'].copy_(seq.codes[idx])
                srcs = (sl['pin_left'], sl.get('pin_right'), sl.get('pin_code'))
        else:
            if sl['used']:
                sl['uploaded'].synchronize()      # the H2D copy that last read this staging slot has finished
            srcs = self._stage(sl, list(frames))
        cur = torch.cuda.current_stream(self.dev)
        if self.raw_stem:
            from .engine import RawChunk
            dev_bytes = {}
            for src, name in zip(srcs, ('left', 'right', 'code')):
                if src is not None:
                    dev_bytes[name] = torch.empty(tuple(src.shape), dtype=src.dtype, device=self.dev)
                    dev_bytes[name].record_stream(self.copy_stream)
            self.copy_stream.wait_stream(cur)     # a recycled block's last reader was enqueued before this point
            with torch.cuda.stream(self.copy_stream):
                for src, name in zip(srcs, ('left', 'right', 'code')):
                    if src is not None:
                        dev_bytes[name].copy_(src, non_blocking=True)
                        self.bytes_uploaded += src.numel() * src.element_size()
                sl['uploaded'].record(self.copy_stream)
            cur.wait_event(sl['uploaded'])
            sl['used'] = True
            out = dict(img=RawChunk(list(dev_bytes['left']), 114.0))
            if self.use_right:
                out['right'] = RawChunk(list(dev_bytes['right']), 114.0)
            else:        # uint16 disparity codes -> fp32 px (x3 planes): the disparity stem reads fp32
                out['disp_postp'] = torch.empty(self.B, 3, self.H, self.W, dtype=torch.float32, device=self.dev)
                check(load().st_pack_raw_inputs(None, ptr(dev_bytes['code']), self.B, self.h, self.w, self.H, self.W, 114.0,
                                                None, ptr(out['disp_postp']), None, current_stream()), 'st_pack_raw_inputs')
            return out
        with torch.cuda.stream(self.copy_stream):
            if sl['used']:
                self.copy_stream.wait_event(sl['consumed'])   # the pack kernel that last read the device bytes is done
            for src, name in zip(srcs, ('dev_left', 'dev_right', 'dev_code')):
                if src is not None:
                    sl[name].copy_(src, non_blocking=True)
                    self.bytes_uploaded += src.numel() * src.element_size()
            sl['uploaded'].record(self.copy_stream)
        cur.wait_event(sl['uploaded'])
        lib = load()
        out = dict(img=torch.empty(self.B, 3, self.H, self.W, dtype=torch.float32, device=self.dev))
        check(lib.st_pack_raw_inputs(ptr(sl['dev_left']), None, self.B, self.h, self.w, self.H, self.W, 114.0,
                                     ptr(out['img']), None, None, current_stream()), 'st_pack_raw_inputs')
        if self.use_right:
            out['right'] = torch.empty_like(out['img'])
            check(lib.st_pack_raw_inputs(ptr(sl['dev_right']), None, self.B, self.h, self.w, self.H, self.W, 114.0,
                                         ptr(out['right']), None, None, current_stream()), 'st_pack_raw_inputs')
        else:
            out['disp_postp'] = torch.empty_like(out['img'])
            check(lib.st_pack_raw_inputs(None, ptr(sl['dev_code']), self.B, self.h, self.w, self.H, self.W, 114.0,
                                         None, ptr(out['disp_postp']), None, current_stream()), 'st_pack_raw_inputs')
        sl['consumed'].record(cur)
        sl['used'] = True
        return out


def detect_shard(pipe, frames, device, check_overflow=True, uploader=None):
    """Dense path over this rank's frames, `pipe.batch` frames per launch plan.  `pipe` is a StereoDensePipeline
    (strictly serial batches) or an InflightPipelines runner (consecutive batches overlap on its streams); `frames`
    is a list of frame dicts (host numpy) or a HostSequence (page-locked, uploaded without a staging copy).  Frames
    cross PCIe as raw bytes through a RawFrameUploader (`uploader`: reuse one across calls).
    -> (F_pad, max_det + 1, 8) frame records (header row + SCALED boxes, what the tracker consumes), counts
    (true counts; 0 for batch padding).  Raises DetectionOverflow if a frame kept more than max_det boxes
    (check_overflow=False defers that to unpack_frame, after a collective, so every rank raises together)."""
    runner = pipe if hasattr(pipe, 'submit') else None
    one = runner.pipes[0] if runner is not None else pipe
    B = one.batch
    bufs = []
    if uploader is None:
        uploader = RawFrameUploader(B, (one.ori_h, one.ori_w), device, use_right=one.stereo)
    resident = isinstance(frames, HostSequence)
    if not resident and not one.stereo and any(disparity_png_codes(f['disp']) is None for f in frames):
        uploader = None        # disparity maps that no PNG could hold: fp32 upload (small tests only)

    def pack(out, n_real):
        return one.pack_detections(out, scaled=True, n_real=n_real)   # fresh tensor

    for i in range(0, len(frames), B):
        n_real = min(B, len(frames) - i)
        if uploader is not None:
            batch = uploader.upload((frames, i, i + n_real) if resident else frames[i:i + n_real])
        else:
            chunk = list(frames[i:i + n_real])
            batch = frames_to_batch(chunk + [chunk[-1]] * (B - n_real), device, use_right=one.stereo)
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


def gather_shard_records(dets, num_frames, batch, device):
    """The exchange step of configs[3]: this rank's (F_local, M + 1, C) frame records -> the records of ALL
    `num_frames` frames in frame order, on every rank.  Pads the shard to the common padded length (ceil(chunk / batch)
    * batch slots per rank, ragged last shards included), issues ONE all-gather through the process's communication
    stream (dist.DetectionGatherer) and drops the per-rank padding again."""
    T = int(num_frames)
    _, _, chunk = sdist.shard_frames(T)
    per_rank = (chunk + batch - 1) // batch * batch   # equal padded length on every rank
    pad = per_rank - dets.shape[0]
    if pad < 0:
        raise ValueError(f'shard holds {dets.shape[0]} records, more than the padded shard length {per_rank}')
    if pad:
        dets = torch.cat([dets, dets.new_zeros((pad,) + tuple(dets.shape[1:]))])
    # ONE collective (the records carry their own counts), on the process's communication stream
    all_dets, done = sdist.DetectionGatherer(device).gather(dets)
    if done is not None:
        torch.cuda.current_stream(all_dets.device).wait_event(done)
    _, world = sdist.world()
    # drop the per-rank padding: rank r holds frames [r*chunk, r*chunk + chunk) in its first `chunk` slots
    idx = torch.cat([torch.arange(r * per_rank, r * per_rank + chunk) for r in range(world)])[:T]
    return all_dets[idx.to(all_dets.device)]


def run_sharded_sequence(pipe, frames, tracker, model, device):
    """configs[3]: shard `frames` over the ranks, detect, all-gather once, track everywhere."""
    frames = list(frames)
    T = len(frames)
    start, stop, chunk = sdist.shard_frames(T)
    B = (pipe.pipes[0] if hasattr(pipe, 'submit') else pipe).batch
    mine = frames[start:stop]
    if mine:
        dets, _ = detect_shard(pipe, mine, device, check_overflow=False)   # raised after the gather, on every rank
    else:
        dets = torch.zeros(0, pipe.max_det + 1, 8, device=device)
    return track_gathered(gather_shard_records(dets, T, B, device), None, T, tracker, model)


def run_video_replicas(pipe, videos, make_tracker, model, device, metrics=None, gts=None, already_sharded=False):
    """The reference's multi-GPU mode (mmtrack/datasets/samplers/video_sampler.py:25-70): WHOLE videos are dealt to
    the ranks in contiguous blocks (dist.shard_videos = np.array_split over the video list), every rank runs its videos
    sequentially with a fresh tracker per video and there is NO communication in the loop; only the evaluation gathers
    (MOTDroneMetrics.evaluate -> all_gather_object, mot_drone_metrics.py:336-358).  `videos`: dict name -> list of
    frame dicts (or HostSequence); `gts`: optional dict name -> per-frame lists of gt instance dicts.
    -> (dict name -> per-frame track InstanceData for THIS rank's videos, scores dict or None)."""
    names = sorted(videos)
    # already_sharded: `videos` holds only THIS rank's videos (datasets.load_videos reads a rank's share of the files)
    mine = names if already_sharded else [names[i] for i in sdist.shard_videos(len(names))]
    uploader = None
    results = {}
    for name in mine:
        frames = videos[name]
        one = pipe.pipes[0] if hasattr(pipe, 'submit') else pipe
        if uploader is None:
            uploader = RawFrameUploader(one.batch, (one.ori_h, one.ori_w), device, use_right=one.stereo)
        dets, counts = detect_shard(pipe, frames, device, uploader=uploader)
        tracks = track_gathered(dets, counts, len(frames), make_tracker(), model)   # frame_id 0 resets the tracker
        results[name] = tracks
        if metrics is not None:
            for t, trk in enumerate(tracks):
                sample = TrackDataSample(dict(frame_id=t))
                sample.pred_track_instances = trk
                metrics.process(name, sample, gts[name][t] if gts is not None else None)
    scores = metrics.evaluate() if metrics is not None else None     # collective: every rank calls it
    return results, scores
