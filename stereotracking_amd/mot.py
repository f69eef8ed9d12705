"""MOT model shell + data preprocessor, registered under the reference's type strings.

  TrackDataPreprocessor_Disparity_V1   mmtrack/models/data_preprocessors/data_preprocessor_disparity_v1.py:19-84
                                       (+ data_preprocessor.py:95-158, utils/misc.py:13-64 stack_batch)
  OCSORT_Disparity                     mmtrack/models/mot/ocsort_disparity.py:16-220 (+ ocsort.py:13-114,
                                       base.py:12-145)

The dense work (detector, decode+NMS, per-box depth) is enqueued on the GPU through the C ABI; the
association step runs on the CPU (stereotracking_amd/trackers.py), as north_star prescribes.
Relaxation of the reference: `predict` accepts N >= 1 frames of ONE video in frame
order — the dense path runs batched, the tracker consumes the frames sequentially.
"""
import ctypes as C

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib
from ._lib import check, current_stream, ptr
from .registry import MODELS, TASK_UTILS
from .structures import InstanceData, TrackDataSample
from .trackers import OCSORTTracker_Disparity  # noqa: F401  (registers the tracker)
from .motion import KalmanFilter  # noqa: F401  (registers the motion model)
from . import detectors  # noqa: F401  (registers detector / backbone / neck / head)
from . import stereo as _stereo  # noqa: F401  (registers StereoCostVolume)


def stack_batch(tensors, pad_size_divisor=0, pad_value=0):
    """Right/bottom pad (T,C,H,W) tensors to a common, divisible size and stack -> (N,T,C,H,W)
    (reference mmtrack/utils/misc.py:13-64)."""
    assert isinstance(tensors, list) and tensors and all(t.ndim == 4 for t in tensors)
    H = max(t.shape[-2] for t in tensors)
    W = max(t.shape[-1] for t in tensors)
    if pad_size_divisor > 1:
        H = (H + pad_size_divisor - 1) // pad_size_divisor * pad_size_divisor
        W = (W + pad_size_divisor - 1) // pad_size_divisor * pad_size_divisor
    out = []
    for t in tensors:
        ph, pw = H - t.shape[-2], W - t.shape[-1]
        out.append(F.pad(t, [0, pw, 0, ph], value=pad_value) if (ph or pw) else t)
    return torch.stack(out, dim=0)


@MODELS.register_module(name=['TrackDataPreprocessor_Disparity_V1'])
class TrackDataPreprocessor_Disparity_V1(nn.Module):
    """H2D copy, .float(), optional BGR<->RGB / mean-std, pad to `pad_size_divisor`, stack every key of
    `inputs` to (N,T,C,H,W).  The shipped stereo config sets only pad_size_divisor=32."""

    def __init__(self, mean=None, std=None, pad_size_divisor=1, pad_value=0, pad_mask=False, mask_pad_value=0,
                 bgr_to_rgb=False, rgb_to_bgr=False, batch_augments=None, non_blocking=False, device=None):
        super().__init__()
        assert not (bgr_to_rgb and rgb_to_bgr)
        self.channel_conversion = bgr_to_rgb or rgb_to_bgr
        self._enable_normalize = mean is not None
        if self._enable_normalize:
            self.register_buffer('mean', torch.tensor(mean, dtype=torch.float32).view(1, -1, 1, 1), False)
            self.register_buffer('std', torch.tensor(std, dtype=torch.float32).view(1, -1, 1, 1), False)
        self.pad_size_divisor, self.pad_value = pad_size_divisor, pad_value
        self.non_blocking = non_blocking
        self._device = torch.device(device) if device is not None else None

    @property
    def device(self):
        if self._device is not None:
            return self._device
        return torch.device('cuda', torch.cuda.current_device()) if torch.cuda.is_available() else torch.device('cpu')

    def forward(self, data, training=False, lazy_raw=False):
        """lazy_raw (used by OCSORT_Disparity.test_step): keys whose frames are equal-sized uint8 (1,3,h,w) CUDA
        tensors, with no normalisation / channel swap configured, come back as RawFrames (converted chunk by chunk
        inside the pipelined dense path) instead of one (N,1,3,H,W) fp32 tensor; the metainfo is set as usual."""
        inputs, samples = data['inputs'], data.get('data_samples')
        dev = self.device
        out = {}
        for key, imgs in inputs.items():
            imgs = [im.to(dev, non_blocking=self.non_blocking) for im in imgs]
            pad_shapes = [tuple(im.shape[-2:]) for im in imgs]
            plain = not (self.channel_conversion and imgs[0].size(1) == 3) and not self._enable_normalize
            if (lazy_raw and plain and imgs[0].is_cuda and len(set(pad_shapes)) == 1 and
                    all(im.dtype == torch.uint8 and tuple(im.shape[:2]) == (1, 3) for im in imgs)):
                d = self.pad_size_divisor
                h, w = pad_shapes[0]
                H, W = ((h + d - 1) // d * d, (w + d - 1) // d * d) if d > 1 else (h, w)
                out[key] = RawFrames(imgs, (H, W), self.pad_value)
                if samples is not None:
                    prefix = key[:-3]
                    for sm, ps in zip(samples, pad_shapes):
                        sm.set_metainfo({f'{prefix}batch_input_shape': (H, W), f'{prefix}pad_shape': ps})
                continue
            if plain and len({tuple(im.shape[:2]) for im in imgs}) == 1:
                # cast + pad + stack as ONE pass per frame: the (N,T,C,H,W) fp32 result is allocated once, filled with
                # the pad value, and every frame is converted straight into its slot (same values as
                # .float() -> F.pad -> torch.stack, reference utils/misc.py:13-64, without two extra full-size passes)
                d = self.pad_size_divisor
                H = max(s[0] for s in pad_shapes)
                W = max(s[1] for s in pad_shapes)
                if d > 1:
                    H, W = (H + d - 1) // d * d, (W + d - 1) // d * d
                T, Cc = imgs[0].shape[:2]
                if len(set(pad_shapes)) == 1 and len({im.dtype for im in imgs}) == 1:
                    # equal-sized frames (a video): one concatenation in the source dtype, one converting copy into
                    # the padded batch, and the pad value written to the pad strips only
                    h, w = pad_shapes[0]
                    batch = torch.empty((len(imgs), T, Cc, H, W), dtype=torch.float32, device=dev)
                    if imgs[0].dtype == torch.uint8 and imgs[0].is_cuda and T == 1 and Cc == 3:
                        # frames uploaded raw (uint8): cast + pad in ONE HIP pass (st_pack_raw_inputs)
                        raw = torch.cat(imgs, dim=0)
                        check(_lib.load().st_pack_raw_inputs(ptr(raw), None, len(imgs), h, w, H, W,
                                                             float(self.pad_value), ptr(batch), None, None,
                                                             current_stream()), 'st_pack_raw_inputs')
                    else:
                        batch[..., :h, :w].copy_(torch.stack(imgs, dim=0))
                        if h < H:
                            batch[..., h:, :] = float(self.pad_value)
                        if w < W:
                            batch[..., :h, w:] = float(self.pad_value)
                else:
                    batch = torch.full((len(imgs), T, Cc, H, W), float(self.pad_value), dtype=torch.float32, device=dev)
                    for i, im in enumerate(imgs):
                        batch[i, :, :, :im.shape[-2], :im.shape[-1]].copy_(im)
                out[key] = batch
            else:
                if self.channel_conversion and imgs[0].size(1) == 3:
                    imgs = [im[:, [2, 1, 0], ...] for im in imgs]
                imgs = [im.float() for im in imgs]
                if self._enable_normalize:
                    imgs = [(im - self.mean) / self.std for im in imgs]
                out[key] = stack_batch(imgs, self.pad_size_divisor, self.pad_value)
            if samples is not None:
                prefix = key[:-3]  # 'img' -> '', 'ref_img' -> 'ref_'
                shape = tuple(out[key].shape[-2:])
                for s, ps in zip(samples, pad_shapes):
                    s.set_metainfo({f'{prefix}batch_input_shape': shape, f'{prefix}pad_shape': ps})
        return dict(inputs=out, data_samples=samples)


class RawFrames:
    """N equal-sized uint8 CUDA frames (1,3,h,w) of one input key, NOT yet converted: what test_step hands to
    predict() for frames uploaded raw.  predict() converts a chunk at a time (torch.cat of the chunk's frames +
    st_pack_raw_inputs: cast + pad in one HIP pass) inside the pipelined submit, so the conversion
    of chunk i+3 overlaps the dense work of chunks i..i+2 and no (N,1,3,H,W) fp32 copy of the whole call exists.
    Values are exactly those of TrackDataPreprocessor_Disparity_V1.forward (reference
    data_preprocessor_disparity_v1.py:21-84 + utils/misc.py:13-64)."""

    def __init__(self, frames, pad_hw, pad_value):
        self.frames, self.pad_hw, self.pad_value = frames, (int(pad_hw[0]), int(pad_hw[1])), float(pad_value)
        self.hw = tuple(frames[0].shape[-2:])
        self.device = frames[0].device

    def __len__(self):
        return len(self.frames)

    def chunk(self, s, e, B):
        """frames [s, e) (+ the last one repeated up to B) -> (B,3,H,W) fp32, padded with pad_value."""
        fr = self.frames[s:e]
        fr = fr + [fr[-1]] * (B - len(fr))
        (h, w), (H, W) = self.hw, self.pad_hw
        out = torch.empty(B, 3, H, W, dtype=torch.float32, device=self.device)
        lib = _lib.load()
        if (B <= 32 and w % 4 == 0 and W % 4 == 0 and
                all(f.is_contiguous() and f.dtype == torch.uint8 and f.data_ptr() % 4 == 0 for f in fr)):
            # the frames stay where the dataloader put them: their pointers travel in the kernel arguments
            # (no torch.cat staging copy: 73 us per input and chunk at 8 x 720 x 1280)
            ptrs = (C.c_void_p * B)(*[f.data_ptr() for f in fr])
            check(lib.st_pack_raw_frames(ptrs, B, h, w, H, W, self.pad_value, ptr(out), current_stream()),
                  'st_pack_raw_frames')
            return out
        raw = torch.cat(fr, dim=0)
        check(lib.st_pack_raw_inputs(ptr(raw), None, B, h, w, H, W, self.pad_value, ptr(out), None, None,
                                     current_stream()), 'st_pack_raw_inputs')
        return out

    def raw_chunk(self, s, e, B, runner):
        """frames [s, e) (+ the last one repeated up to B) as an engine.RawChunk - the stem kernel casts + pads them while it
        stages its input windows (st_detector_forward_phase0_raw), no fp32 image exists - or None when the frames do
        not qualify (width % 4, alignment, non-integral pad value, padded size of another plan)."""
        from .engine import RawChunk
        fr = self.frames[s:e]
        fr = fr + [fr[-1]] * (B - len(fr))
        if self.pad_hw != (runner.height, runner.width) or not RawChunk.supported(fr, self.pad_value):
            return None
        return RawChunk(fr, self.pad_value)

    def dense(self):
        """The (N,1,3,H,W) fp32 tensor the preprocessor would have produced (for callers that want it)."""
        return self.chunk(0, len(self.frames), len(self.frames))[:, None]


def pack_raw_inputs(img_u8=None, disp_u16=None, pad_size_divisor=32, img_pad=114.0):
    """Device-side input pipeline for frames uploaded raw: uint8 (N,3,h,w) image and/or
    uint16 (N,h,w) disparity PNG codes, both CUDA tensors -> dict(img, disp_postp, disp_mask) fp32 padded to
    `pad_size_divisor`, exactly what LoadDisparityFromFile + Pad_Disparity + the preprocessor produce."""
    src = img_u8 if img_u8 is not None else disp_u16
    if src is None or not src.is_cuda:
        raise RuntimeError('pack_raw_inputs needs CUDA tensors (HIP path only)')
    N, h, w = src.shape[0], src.shape[-2], src.shape[-1]
    d = pad_size_divisor
    H, W = (h + d - 1) // d * d, (w + d - 1) // d * d
    dev = src.device
    out = {}
    if img_u8 is not None:
        assert img_u8.dtype == torch.uint8 and tuple(img_u8.shape) == (N, 3, h, w)
        img_u8 = img_u8.contiguous()
        out['img'] = torch.empty(N, 3, H, W, dtype=torch.float32, device=dev)
    if disp_u16 is not None:
        assert disp_u16.dtype in (torch.uint16, torch.int16) and tuple(disp_u16.shape) == (N, h, w)
        disp_u16 = disp_u16.contiguous()
        out['disp_postp'] = torch.empty(N, 3, H, W, dtype=torch.float32, device=dev)
        out['disp_mask'] = torch.empty(N, 1, H, W, dtype=torch.float32, device=dev)
    check(_lib.load().st_pack_raw_inputs(ptr(img_u8), ptr(disp_u16), N, h, w, H, W, float(img_pad), ptr(out.get('img')),
                                         ptr(out.get('disp_postp')), ptr(out.get('disp_mask')), current_stream()),
          'st_pack_raw_inputs')
    return out


CSV_HEADER = ['frame', 'id', 'label', 'tl_x', 'tl_y', 'br_x', 'br_y', 'depth', 'gt_depth', 'score']


def append_prediction_results(file_path, results):
    """The CSV side effect of the reference's predict (mmtrack/utils/collect_results.py:1-44: one row per track,
    `frame,id,label,tl_x,tl_y,br_x,br_y,depth,gt_depth,score`, header written when the file is empty), for EVERY
    sample of a batched call (the reference handles results[0] because it never batches)."""
    import csv
    import os
    if not file_path.endswith('.csv'):
        raise ValueError('The saving format is not supported.')
    with open(file_path, 'a') as f:
        writer = csv.writer(f)
        f.seek(0, os.SEEK_END)
        if f.tell() == 0:
            writer.writerow(CSV_HEADER)
        for sample in results:
            trk = sample.pred_track_instances
            frame_id = sample.metainfo.get('frame_id')
            boxes = trk.get('bboxes').cpu().numpy()
            ids = trk.get('instances_id').cpu().numpy()
            labels = trk.get('labels').cpu().numpy()
            scores = trk.get('scores').cpu().numpy()
            # plain floats, as the reference's d_values list yields them (a 0-d tensor would print as `tensor(10.)`)
            depth = _as_float_list(trk.get('depth'), len(ids))
            gt_depth = _as_float_list(trk.get('gt_depth'), len(ids))
            for iid, label, box, d, gd, sc in zip(ids, labels, boxes, depth, gt_depth, scores):
                writer.writerow([frame_id, iid, label, *box, d, gd, sc])


def _as_float_list(v, n):
    if v is None:
        return [float('nan')] * n
    if torch.is_tensor(v):
        v = v.detach().cpu().numpy()
    return [float(x) for x in np.asarray(v, dtype=np.float64).reshape(-1)]


def save_prediction_results(file_path):
    """Decorator form, as the reference applies it to OCSORT_Disparity.predict (collect_results.py:1-44): an existing
    file is deleted when the decorator is applied."""
    import os

    def decorator(predict_func):
        if os.path.exists(file_path):
            os.remove(file_path)

        def wrapper(*args, **kwargs):
            results = predict_func(*args, **kwargs)
            append_prediction_results(file_path, results)
            return results
        return wrapper
    return decorator


def scale_bbox(bboxes, scales):
    """Scale boxes about their centres (reference mmtrack/models/trackers/utils.py:58-73)."""
    cx, cy = (bboxes[:, 0] + bboxes[:, 2]) / 2, (bboxes[:, 1] + bboxes[:, 3]) / 2
    w, h = (bboxes[:, 2] - bboxes[:, 0]) * scales, (bboxes[:, 3] - bboxes[:, 1]) * scales
    return torch.stack((cx - w / 2, cy - h / 2, cx + w / 2, cy + h / 2), dim=-1).reshape(-1, 4)


@MODELS.register_module(name=['OCSORT_Disparity'])
class OCSORT_Disparity(nn.Module):
    """MOT shell (reference mmtrack/models/mot/ocsort_disparity.py:16-220).  `predict` accepts N >= 1 frames of ONE
    video in frame order and drives the batched dense path (pipeline.InflightPipelines): the frames go through the
    HIP launch plan `dense_batch` at a time on `inflight` contexts / streams; per chunk there is ONE device->host
    copy of the fixed-size detection records and ONE batched st_box_depth launch for the tracks' depth column - no
    per-frame launches, no per-frame syncs.  The association step consumes the frames sequentially on the CPU while
    the GPU works on the following chunks.

    New (optional) constructor arguments next to the reference's: `stereo` (StereoCostVolume config), `dense_batch`,
    `inflight`, `max_det` (rows of the detection buffer: a capacity, overflow raises) and `results_device`
    ('cpu': results stay where the CPU tracker produced them - the consumers are the host-side evaluator / CSV
    writer; 'input': moved back to the device of the inputs like the reference's tensors)."""

    def __init__(self, detector=None, tracker=None, motion=None, data_preprocessor=None, init_cfg=None,
                 baseline=0.25, focal_length=640, stereo=None, dense_batch=8, inflight=3, max_det=1000,
                 results_device='cpu', autotune=True, tuning_cache=None, results_csv=None, split_bf16=None,
                 queue_depth=1):
        super().__init__()
        self.data_preprocessor = MODELS.build(data_preprocessor) if data_preprocessor is not None else None
        self.detector = MODELS.build(detector) if detector is not None else None
        self.motion = TASK_UTILS.build(motion) if motion is not None else None
        self.tracker = MODELS.build(tracker) if tracker is not None else None
        self.baseline, self.focal_length = baseline, focal_length
        if (stereo is not None and stereo.get('full_res') and 'feat_channels' not in stereo and self.detector is not None):
            # the full-resolution mode's reduce conv reads the detector's stage-1 features: make_divisible(128, widen)
            import math
            stereo = dict(stereo, feat_channels=int(math.ceil(128 * self.detector.widen_factor / 8) * 8))
        self.stereo = MODELS.build(stereo) if stereo is not None else None  # StereoCostVolume (new module)
        if self.stereo is not None and self.detector is not None:
            self.detector.__dict__['stereo'] = self.stereo   # plain reference: registered once, under the shell
        self.dense_batch, self.inflight, self.max_det = int(dense_batch), int(inflight), int(max_det)
        # chunks queued per context: with 2 a context's next chunk is already in its stream when the host learns that the
        # current one finished (the device never waits for the host's refill); the per-context staging buffers and the
        # disparity ring have queue_depth + 1 slots: one being consumed by the host + queue_depth behind it
        self.queue_depth = max(1, int(queue_depth))
        self.raw_stem = True      # uint8 frames go to the stem kernels as they are (False: st_pack_raw_frames first)
        if results_device not in ('cpu', 'input'):
            raise ValueError("results_device must be 'cpu' or 'input'")
        self.results_device = results_device
        self.autotune, self.tuning_cache = bool(autotune), tuning_cache
        self.split_bf16 = split_bf16      # None: $ST_SPLIT_BF16; True: the autotuner may pick the split-operand conv instances
        # the reference decorates predict with save_prediction_results('results.csv') unconditionally
        # (ocsort_disparity.py:49); here the side effect is opt-in: results_csv='results.csv' reproduces it
        self.results_csv = results_csv
        if results_csv is not None:
            import os
            if os.path.exists(results_csv):
                os.remove(results_csv)
        self._pre_lazy = False
        if self.data_preprocessor is not None:
            import inspect
            fwd = getattr(self.data_preprocessor, 'forward', self.data_preprocessor)
            try:
                self._pre_lazy = 'lazy_raw' in inspect.signature(fwd).parameters
            except (TypeError, ValueError):
                self._pre_lazy = False
        self.lib = _lib.load()
        self._dense = {}          # (batch, ori_h, ori_w, stereo) -> [InflightPipelines, weights version]
        self._staging = {}        # name -> pinned host buffer (grow-only): no pinned allocation on the per-chunk path
        self.timings = dict(frames=0, tracker_s=0.0, host_s=0.0, wait_s=0.0, pre_s=0.0, tail_s=0.0, submit_s=0.0, depth_s=0.0)   # cumulative host-side costs

    # ---- reference plumbing (mot/base.py:68-113) -----------------------------------------------------
    def init_weights(self):
        if self.detector is not None:
            self.detector.init_weights()

    def test_step(self, data):
        import time
        t0 = time.perf_counter()
        if self._pre_lazy:     # decided once from the preprocessor's signature (__init__), not by catching TypeError
            data = self.data_preprocessor(data, False, lazy_raw=True)
        else:                  # a preprocessor without the lazy option (e.g. mmengine's own class)
            data = self.data_preprocessor(data, False)
        self.timings['pre_s'] += time.perf_counter() - t0
        return self.forward(data['inputs'], data['data_samples'], mode='predict')

    def forward(self, inputs, data_samples=None, mode='predict', **kwargs):
        if mode == 'predict':
            return self.predict(inputs, data_samples, **kwargs)
        if mode == 'loss':
            raise NotImplementedError('training is out of scope of the HIP hot path (SURVEY.md §3.3)')
        raise NotImplementedError('tensor mode is not supported (reference mot/base.py:144-145)')

    # ---- the batched dense path behind the plugin surface ----------------------------------------------
    def _weights_version(self):
        return tuple(t._version for t in self.state_dict(keep_vars=True).values())

    def dense_runner(self, ori_hw, stereo, batch=None):
        """InflightPipelines context set for (batch, ori_hw, stereo), built from this model's config and
        loaded from its state_dict (reference keys `detector.*`, plus `stereo.agg.*` of the new module)."""
        from .pipeline import InflightPipelines
        det = self.detector
        cfg = det.test_cfg
        nms = cfg.get('nms', dict(type='nms', iou_threshold=0.65))
        if nms.get('type', 'nms') != 'nms':
            raise NotImplementedError(f"nms type {nms.get('type')} (only greedy 'nms')")
        if not cfg.get('yolox_style', False):
            raise NotImplementedError('the batched dense path implements the shipped yolox_style=True post-processing '
                                      '(no max_per_img cut); use detector.predict for other test_cfg')
        batch = int(batch or self.dense_batch)
        key = (batch, int(ori_hw[0]), int(ori_hw[1]), bool(stereo))
        ent = self._dense.get(key)
        if ent is None:
            sm = self.stereo
            runner = InflightPipelines(
                max(1, self.inflight), batch, (key[1], key[2]), det.widen_factor, det.deepen_factor,
                det.num_classes, stereo=bool(stereo), max_disp=sm.max_disp if stereo else 192,
                feat_stride=sm.feat_stride if stereo else 4, temperature=sm.temperature if stereo else 32.0,
                score_thr=cfg.get('score_thr', 0.01), iou_thr=nms.get('iou_threshold', 0.65), max_det=self.max_det,
                baseline=self.baseline, focal_length=self.focal_length,
                pad_size_divisor=getattr(self.data_preprocessor, 'pad_size_divisor', 32) or 32,
                agg_layers=sm.agg_layers if stereo else 0, agg3d_layers=sm.agg3d_layers if stereo else 0,
                split_bf16=self.split_bf16, multi_label=getattr(det, 'multi_label', True),
                rgb_only=getattr(det, 'rgb_only', False),
                full_res=bool(getattr(sm, 'full_res', False)) if stereo else False,
                full_res_channels=(sm.reduce.out_channels if stereo and getattr(sm, 'full_res', False) else 8))
            for p in runner.pipes:     # the track-box depth reads run k's disparity while later runs are in flight
                p.disp_buffers = self.queue_depth + 1
            ent = self._dense[key] = [runner, None]
        ver = self._weights_version()
        if ent[1] != ver:
            sd = {k[len('detector.'):]: v for k, v in self.state_dict().items() if k.startswith('detector.')}
            sd.update({k: v for k, v in self.state_dict().items() if k.startswith('stereo.')})
            ent[0].load_state_dict(sd, autotune=self.autotune, tuning_cache=self.tuning_cache)
            ent[1] = ver
        return ent[0]

    def _side_stream(self, dev):
        st = self._staging.get(('side_stream', dev))
        if st is None:
            st = self._staging[('side_stream', dev)] = torch.cuda.Stream(device=dev)
        return st

    def _pinned(self, name, shape, dtype=torch.float32):
        """View of a cached page-locked staging buffer.  Allocating pinned memory per chunk (hipHostMalloc) stalls the
        host until the device is idle, which serialises the in-flight contexts; these buffers are allocated once and
        grow only when a chunk needs more room."""
        n = 1
        for v in shape:
            n *= int(v)
        buf = self._staging.get(name)
        if buf is None or buf.dtype != dtype or buf.numel() < n:
            cap = max(256, 1 << max(n - 1, 1).bit_length())
            buf = self._staging[name] = torch.empty(cap, dtype=dtype, pin_memory=True)
        return buf[:n].view(*shape)

    # ---- per-box depth on the device (ocsort_disparity.py:113-175) -------------------------------------
    def bbox_postp_depth(self, pred_instances, disp, gt_depth=None):
        """disp: (1,3,H,W) disp_postp.  Returns (instances with scaled `bboxes`, `scales`, `depth`), depth dict.
        Single-frame form of the reference method (predict() uses the batched launch instead)."""
        boxes = pred_instances['bboxes'].float().contiguous()
        d_values, scales, scaled = self._box_depth(disp, boxes[None], None, self.baseline, self.focal_length)
        depth_values = dict(d_values=d_values[0])
        if gt_depth is not None:
            depth_values['gt_d_values'] = self._box_depth(gt_depth, boxes[None], None, -1.0, 1.0)[0][0]
        pred_instances['bboxes'] = scaled[0]
        pred_instances['scales'] = scales[0]
        pred_instances['depth'] = d_values[0]
        return pred_instances, depth_values

    def _box_depth(self, disp, boxes, counts, baseline, focal):
        """ONE st_box_depth launch for a whole batch: disp (N,C,H,W), boxes (N,M,4), counts (N,) int32 or None
        (= all M rows) -> depth (N,M), scales (N,M), scaled boxes (N,M,4)."""
        N, M = boxes.shape[0], boxes.shape[1]
        dev = boxes.device
        if M == 0:
            return torch.zeros(N, 0, device=dev), torch.zeros(N, 0, device=dev), torch.zeros(N, 0, 4, device=dev)
        depth = torch.empty(N, M, device=dev)      # st_box_depth defines every row (0 past the count)
        scales = torch.empty(N, M, device=dev)
        sboxes = torch.empty(N, M, 4, device=dev)
        _, Cc, H, W = disp.shape
        disp = disp.float().contiguous()
        if counts is None:
            counts = torch.full((N,), M, dtype=torch.int32, device=dev)
        check(self.lib.st_box_depth(ptr(disp), Cc * H * W, N, H, W, ptr(boxes.float().contiguous()), ptr(counts), M,
                                    float(baseline), float(focal), None, 0, current_stream(), ptr(depth), ptr(scales),
                                    ptr(sboxes)), 'st_box_depth')
        return depth, scales, sboxes

    # ---- predict (ocsort_disparity.py:50-111) ------------------------------------------------------------
    def predict(self, inputs, data_samples, **kwargs):
        """One call = begin (validate, split into dense_batch chunks) + finish (run, associate, complete the samples)."""
        return self.finish(self.begin(inputs, data_samples, **kwargs))

    def test_steps(self, data_iter):
        """The test loop over successive `test_step` inputs of one or more videos, with the contexts kept primed ACROSS
        calls: a generator that yields, per element of `data_iter`, exactly what `test_step(data)` returns (same values,
        same order - the association runs strictly in frame order), but call k+1's preprocessor and first chunks are
        submitted while call k drains, so the device never idles between calls (fill + drain cost 5 % at 64 frames per
        call).  Stands where mmengine's TestLoop calls `model.test_step(data_batch)` per batch
        (reference mot/base.py:68-113 is the per-call entry this keeps)."""
        import time
        prev = None
        for data in data_iter:
            t0 = time.perf_counter()
            if self._pre_lazy:
                data = self.data_preprocessor(data, False, lazy_raw=True)
            else:
                data = self.data_preprocessor(data, False)
            self.timings['pre_s'] += time.perf_counter() - t0
            st = self.begin(data['inputs'], data['data_samples'])
            if prev is not None:
                yield self.finish(prev, lookahead=st)
            prev = st
        if prev is not None:
            yield self.finish(prev)

    def begin(self, inputs, data_samples, **kwargs):
        """Validate one call's inputs and plan its chunks; nothing is launched yet (finish() does, or the finish() of
        the call before this one when it is passed there as `lookahead`)."""
        img, disp_postp = inputs['img'], inputs.get('disp_postp')
        depth_postp = inputs.get('depth_postp', None)

        def unwrap(t, name):      # (N,1,C,H,W) tensor -> (N,C,H,W); RawFrames stay lazy (converted per chunk)
            if t is None or isinstance(t, RawFrames):
                return t
            assert t.dim() == 5, f'The {name} must be 5D Tensor (N, T, C, H, W).'
            assert t.size(1) == 1, 'one key frame per sample (T = 1)'
            return t[:, 0]
        img = unwrap(img, 'img')
        N = len(img)
        assert len(data_samples) == N
        if not (img.device.type == 'cuda'):
            raise RuntimeError('OCSORT_Disparity runs on the HIP path only: inputs must be CUDA tensors')
        stereo = disp_postp is None
        if stereo:
            if self.stereo is None or inputs.get('right') is None:
                raise KeyError("inputs need 'disp_postp', or 'right' with a stereo module configured")
            second = unwrap(inputs['right'], 'right')
        else:
            second = unwrap(disp_postp, 'disp_postp')
            if isinstance(second, RawFrames):
                second = second.dense()[:, 0]      # a uint8 disparity is unusual: convert it eagerly
        gt = unwrap(depth_postp, 'depth_postp')
        if isinstance(gt, RawFrames):
            gt = gt.dense()[:, 0]
        metas = [s.metainfo for s in data_samples]
        pad_hw = img.pad_hw if isinstance(img, RawFrames) else tuple(img.shape[-2:])
        ori = tuple(int(v) for v in metas[0].get('ori_shape', pad_hw)[:2])
        for m in metas[1:]:
            if tuple(int(v) for v in m.get('ori_shape', ori)[:2]) != ori:
                raise NotImplementedError('one batched launch plan needs a uniform ori_shape')
        B = min(self.dense_batch, N)      # a call with fewer frames than dense_batch gets a plan of its own size
        runner = self.dense_runner(ori, stereo, B)
        return dict(img=img, second=second, gt=gt, stereo=stereo, N=N, B=B, runner=runner, dev=img.device,
                    data_samples=data_samples, kwargs=kwargs, jobs={}, submitted=0,
                    chunks=[(s, min(s + B, N)) for s in range(0, N, B)])

    @staticmethod
    def _padded(t, s, e, B):
        if isinstance(t, RawFrames):
            return t.chunk(s, e, B)
        t = t[s:e].float().contiguous()
        if e - s < B:      # last chunk: repeat its last frame (results of the padding are ignored)
            t = torch.cat([t, t[-1:].expand(B - (e - s), *t.shape[1:])])
        return t

    def _submit_next(self, st):
        """Enqueue call `st`'s next chunk on the runner's next context (round-robin)."""
        import time
        ts = time.perf_counter()
        runner, B, stereo = st['runner'], st['B'], st['stereo']
        ci = st['submitted']
        st['submitted'] += 1
        s, e = st['chunks'][ci]
        a = b = None
        if self.raw_stem and isinstance(st['img'], RawFrames):
            if stereo and isinstance(st['second'], RawFrames):
                a, b = st['img'].raw_chunk(s, e, B, runner), st['second'].raw_chunk(s, e, B, runner)
            elif not stereo:      # disparity-input configuration: the image raw, the fp32 disparity as it is
                a = st['img'].raw_chunk(s, e, B, runner)
                b = self._padded(st['second'], s, e, B) if a is not None else None
        if a is None or b is None:      # fp32 tensors (or frames the stem cannot read raw): cast + pad as a pass of its own
            a, b = self._padded(st['img'], s, e, B), self._padded(st['second'], s, e, B)
        holder = {}
        # staging buffers of a context alternate: it is resubmitted before the chunk it just finished is consumed
        turns = self._staging.setdefault(('ctx_turns', id(runner)), [0] * len(runner))

        def post(out, ctx):   # under the context's stream: pack + start the ONE device->host copy of this chunk
            slot = turns[ctx] % (self.queue_depth + 1)
            turns[ctx] += 1
            rec = runner.pipes[ctx].pack_detections(out, scaled='both', n_real=e - s)
            host = self._pinned(('records', id(runner), ctx, slot), rec.shape, rec.dtype)
            host.copy_(rec, non_blocking=True)
            # the depth of the TRACK boxes is read from this chunk's disparity after the association, when the context
            # already runs its next chunk: the stereo module's output cycles through queue_depth + 1 buffers (disp_slot;
            # the read is ordered before the buffer's next rewrite by pipe.disp_guard), the mono input `b` is a
            # tensor of this chunk's own - no private copy either way
            holder.update(ctx=ctx, disp=out['disp_postp'], disp_slot=runner.pipes[ctx].disp_slot if stereo else None, host=host, slot=slot)
            return out
        _, ev = runner.submit(a, right=b if stereo else None, disp_postp=None if stereo else b, post=post)
        self.timings['submit_s'] += time.perf_counter() - ts
        st['jobs'][ci] = dict(s=s, e=e, ev=ev, **holder)

    def finish(self, st, lookahead=None):
        """Run call `st` to completion and return its samples.  `lookahead`: the begin() state of the NEXT call on the
        same runner - as this call's contexts free up they are refilled with that call's first chunks."""
        import time
        from .dist import DetectionOverflow
        runner, B, N, dev, gt = st['runner'], st['B'], st['N'], st['dev'], st['gt']
        data_samples, kwargs, chunks, jobs = st['data_samples'], st['kwargs'], st['chunks'], st['jobs']
        if lookahead is not None and (lookahead['runner'] is not runner or lookahead is st):
            lookahead = None
        t_host0 = time.perf_counter()

        def padded(t, s, e):
            return self._padded(t, s, e, B)

        def refill():       # one context is free: this call's next chunk, else the next call's
            if st['submitted'] < len(chunks):
                self._submit_next(st)
            elif lookahead is not None and lookahead['submitted'] < min(len(lookahead['chunks']), depth):
                self._submit_next(lookahead)

        depth = self.queue_depth * len(runner)      # chunks outstanding on the device (+ the one the host consumes)
        while st['submitted'] < min(len(chunks), depth):
            self._submit_next(st)
        outs, pending = [None] * N, []
        t_tail0 = time.perf_counter()
        def finalize(entry):   # depth of the unscaled track boxes has arrived: complete the chunk's samples
            s, e, tracks_of, dh, _ev2, _keep = entry
            for i, tracks in enumerate(tracks_of):
                k = len(tracks)
                tracks['depth'] = dh[0, i, :k].clone()
                tracks['gt_depth'] = dh[-1, i, :k].clone()   # = depth when no gt depth map was given (:104)
                sample = data_samples[s + i]
                if self.results_device == 'input':
                    tracks = tracks.to(dev)
                    sample.pred_det_instances = sample.pred_det_instances.to(dev)
                sample.pred_track_instances = tracks
                outs[s + i] = sample

        for ci in range(len(chunks)):
            job = jobs.pop(ci)
            tw = time.perf_counter()
            job['ev'].synchronize()                       # the only wait of this chunk's forward pass
            self.timings['wait_s'] += time.perf_counter() - tw
            if ci == len(chunks) - 1:
                t_tail0 = time.perf_counter()
            refill()     # refill this context FIRST: the device keeps `inflight` chunks while the host associates this
            # one (its results live in buffers of their own)
            rec = job['host']
            s, e = job['s'], job['e']
            tracks_of = []
            t0 = time.perf_counter()
            if getattr(self.tracker, 'backend', None) == 'native' and not kwargs:
                # the whole chunk in ONE native call (st_tracker_track_records reads the page-locked record buffer the
                # D2H copy landed in: detections in, unscaled track rows out); per frame only views are taken
                fids = [int(data_samples[n].metainfo.get('frame_id', -1)) for n in range(s, e)]
                # (numpy copies: a torch CPU op above ~32 K elements wakes the whole intra-op thread pool - tens of
                # milliseconds on a 256-core host whose process owns a 16-core share - for a 400 KB memcpy)
                chunk_np = rec[:e - s].numpy().copy()  # the staging buffer is reused by a later chunk
                trows, tids, tcnt = self.tracker.track_records(fids, chunk_np)
                det_labels = torch.from_numpy(chunk_np[:, 1:, 5].astype(np.int64))
                det_prior = torch.from_numpy(chunk_np[:, 1:, 12].astype(np.int64))
                trk_labels = torch.from_numpy(trows[:, :, 5].astype(np.int64))
                counts_h = chunk_np[:, 0, 0].astype(np.int64).tolist()
                chunk, trows, tids = torch.from_numpy(chunk_np), torch.from_numpy(trows), torch.from_numpy(tids)
                for i, n in enumerate(range(s, e)):
                    k, m = counts_h[i], int(tcnt[i])
                    rows, tr = chunk[i, 1:1 + k], trows[i, :m]
                    data_samples[n].pred_det_instances = InstanceData(bboxes=rows[:, 0:4], scores=rows[:, 4],
                                                                      labels=det_labels[i, :k],
                                                                      prior_idx=det_prior[i, :k])   # (:107-108)
                    tracks = InstanceData()
                    tracks['bboxes'] = tr[:, 0:4]                     # already unscaled (:95-97)
                    tracks['labels'] = trk_labels[i, :m]
                    tracks['scores'] = tr[:, 4]
                    tracks['scales'] = tr[:, 7]
                    tracks['depth'] = tr[:, 6]
                    tracks.instances_id = tids[i, :m]
                    tracks_of.append(tracks)
            else:
              for n in range(s, e):
                r = rec[n - s]
                k, cap = int(r[0, 0]), int(r[0, 1])
                if k > cap:
                    raise DetectionOverflow(f'frame {n}: {k} detections kept but the detection buffer has {cap} rows; '
                                            f'build the model with a larger max_det')
                rows = r[1:1 + k].clone()      # the staging buffer is reused by a later chunk
                labels = rows[:, 5].long()
                sample = data_samples[n]
                # reference :82-86: the tracker consumes the depth-SCALED boxes + scales + depth
                sample.pred_det_instances = InstanceData(bboxes=rows[:, 8:12], scores=rows[:, 4], labels=labels,
                                                         scales=rows[:, 7], depth=rows[:, 6])
                tracks = self.tracker.track(model=self, img=None, feats=None, data_sample=sample, **kwargs)
                tracks['bboxes'] = scale_bbox(tracks.bboxes, 1 / tracks.scales)      # unscale (:95-97)
                sample.pred_det_instances = InstanceData(bboxes=rows[:, 0:4].clone(), scores=rows[:, 4].clone(),
                                                         labels=labels, prior_idx=rows[:, 12].long())   # (:107-108)
                tracks_of.append(tracks)
            self.timings['tracker_s'] += time.perf_counter() - t0
            # reference :99-104: depth (and gt depth) of the UNSCALED track boxes - ONE batched launch per chunk, on a
            # side stream (the chunk's forward pass has completed: `ev` above), reading the chunk's own disparity copy
            td = time.perf_counter()
            mt = max([len(t) for t in tracks_of] + [1])
            # the page-locked (ctx, slot) pair below was the SOURCE of an asynchronous host->device copy two rounds ago:
            # wait for that copy's event before the host overwrites the buffer (it has long completed in practice -
            # now it is ordered, not probable)
            slot_key = ('track_slot_event', id(runner), job['ctx'], job['slot'])
            prev_ev = self._staging.get(slot_key)
            if prev_ev is not None:
                prev_ev.synchronize()
            tb = self._pinned(('track_boxes', id(runner), job['ctx'], job['slot']), (B, mt, 4)).zero_()
            tc = self._pinned(('track_counts', id(runner), job['ctx'], job['slot']), (B,), torch.int32).zero_()
            for i, t in enumerate(tracks_of):
                tb[i, :len(t)] = t.bboxes
                tc[i] = len(t)
            stream = self._side_stream(dev)
            with torch.cuda.stream(stream):
                job['disp'].record_stream(stream)
                tbd, tcd = tb.to(dev, non_blocking=True), tc.to(dev, non_blocking=True)
                d = self._box_depth(job['disp'], tbd, tcd, self.baseline, self.focal_length)[0]
                cols = [d]
                if gt is not None:
                    cols.append(self._box_depth(padded(gt, s, e), tbd, tcd, -1.0, 1.0)[0])
                dh = self._pinned(('track_depth', id(runner), ci), (len(cols), B, mt))   # read at the end of the call
                dh.copy_(torch.stack(cols), non_blocking=True)
                ev2 = torch.cuda.Event()
                ev2.record(stream)
                self._staging[slot_key] = ev2      # recorded after the copies that read tb / tc
                if job['disp_slot'] is not None:   # ... and after the last read of this disparity buffer
                    runner.pipes[job['ctx']].disp_guard[job['disp_slot']] = ev2
            pending.append((s, e, tracks_of, dh, ev2, (tbd, tcd)))
            self.timings['depth_s'] += time.perf_counter() - td
            while pending and pending[0][4].query():       # earlier chunks whose track depth has landed: complete
                finalize(pending.pop(0))                   # them now, while the device works on the next chunks
        for entry in pending:
            tw = time.perf_counter()
            entry[4].synchronize()
            self.timings['wait_s'] += time.perf_counter() - tw
            finalize(entry)
        self.timings['tail_s'] += time.perf_counter() - t_tail0
        self.timings['frames'] += N
        self.timings['host_s'] += time.perf_counter() - t_host0
        if self.results_csv is not None:
            append_prediction_results(self.results_csv, outs)
        return outs
