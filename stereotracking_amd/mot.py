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

    def forward(self, data, training=False):
        inputs, samples = data['inputs'], data.get('data_samples')
        dev = self.device
        out = {}
        for key, imgs in inputs.items():
            imgs = [im.to(dev, non_blocking=self.non_blocking) for im in imgs]
            if self.channel_conversion and imgs[0].size(1) == 3:
                imgs = [im[:, [2, 1, 0], ...] for im in imgs]
            imgs = [im.float() for im in imgs]
            if self._enable_normalize:
                imgs = [(im - self.mean) / self.std for im in imgs]
            pad_shapes = [tuple(im.shape[-2:]) for im in imgs]
            out[key] = stack_batch(imgs, self.pad_size_divisor, self.pad_value)
            if samples is not None:
                prefix = key[:-3]  # 'img' -> '', 'ref_img' -> 'ref_'
                shape = tuple(out[key].shape[-2:])
                for s, ps in zip(samples, pad_shapes):
                    s.set_metainfo({f'{prefix}batch_input_shape': shape, f'{prefix}pad_shape': ps})
        return dict(inputs=out, data_samples=samples)


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


def scale_bbox(bboxes, scales):
    """Scale boxes about their centres (reference mmtrack/models/trackers/utils.py:58-73)."""
    cx, cy = (bboxes[:, 0] + bboxes[:, 2]) / 2, (bboxes[:, 1] + bboxes[:, 3]) / 2
    w, h = (bboxes[:, 2] - bboxes[:, 0]) * scales, (bboxes[:, 3] - bboxes[:, 1]) * scales
    return torch.stack((cx - w / 2, cy - h / 2, cx + w / 2, cy + h / 2), dim=-1).reshape(-1, 4)


@MODELS.register_module(name=['OCSORT_Disparity'])
class OCSORT_Disparity(nn.Module):
    def __init__(self, detector=None, tracker=None, motion=None, data_preprocessor=None, init_cfg=None,
                 baseline=0.25, focal_length=640, stereo=None):
        super().__init__()
        self.data_preprocessor = MODELS.build(data_preprocessor) if data_preprocessor is not None else None
        self.detector = MODELS.build(detector) if detector is not None else None
        self.motion = TASK_UTILS.build(motion) if motion is not None else None
        self.tracker = MODELS.build(tracker) if tracker is not None else None
        self.baseline, self.focal_length = baseline, focal_length
        self.stereo = MODELS.build(stereo) if stereo is not None else None  # StereoCostVolume (new module)
        if self.stereo is not None and self.detector is not None:
            self.detector.__dict__['stereo'] = self.stereo   # plain reference: registered once, under the shell
        self.lib = _lib.load()

    # ---- reference plumbing (mot/base.py:68-113) -----------------------------------------------------
    def init_weights(self):
        if self.detector is not None:
            self.detector.init_weights()

    def test_step(self, data):
        data = self.data_preprocessor(data, False)
        return self.forward(data['inputs'], data['data_samples'], mode='predict')

    def forward(self, inputs, data_samples=None, mode='predict', **kwargs):
        if mode == 'predict':
            return self.predict(inputs, data_samples, **kwargs)
        if mode == 'loss':
            raise NotImplementedError('training is out of scope of the HIP hot path (SURVEY.md §3.3)')
        raise NotImplementedError('tensor mode is not supported (reference mot/base.py:144-145)')

    # ---- per-box depth on the device (ocsort_disparity.py:113-175) -------------------------------------
    def bbox_postp_depth(self, pred_instances, disp, gt_depth=None):
        """disp: (1,3,H,W) disp_postp.  Returns (instances with scaled `bboxes`, `scales`, `depth`), depth dict."""
        boxes = pred_instances['bboxes'].float().contiguous()
        d_values, scales, scaled = self._box_depth(disp, boxes, self.baseline, self.focal_length)
        depth_values = dict(d_values=d_values)
        if gt_depth is not None:
            depth_values['gt_d_values'], _, _ = self._box_depth(gt_depth, boxes, -1.0, 1.0)
        pred_instances['bboxes'] = scaled
        pred_instances['scales'] = scales
        pred_instances['depth'] = d_values
        return pred_instances, depth_values

    def _box_depth(self, disp, boxes, baseline, focal):
        M = boxes.shape[0]
        dev = boxes.device
        if M == 0:
            z = boxes.new_zeros(0)
            return z, z.clone(), boxes.new_zeros(0, 4)
        _, Cc, H, W = disp.shape
        disp = disp.float().contiguous()
        counts = torch.tensor([M], dtype=torch.int32, device=dev)
        depth = torch.empty(1, M, device=dev)
        scales = torch.empty(1, M, device=dev)
        sboxes = torch.empty(1, M, 4, device=dev)
        check(self.lib.st_box_depth(ptr(disp), Cc * H * W, 1, H, W, ptr(boxes), ptr(counts), M, float(baseline),
                                    float(focal), None, 0, current_stream(), ptr(depth), ptr(scales), ptr(sboxes)),
              'st_box_depth')
        return depth[0], scales[0], sboxes[0]

    # ---- predict (ocsort_disparity.py:50-111) ------------------------------------------------------------
    def predict(self, inputs, data_samples, **kwargs):
        img, disp_postp = inputs['img'], inputs.get('disp_postp')
        disp_mask = inputs.get('disp_mask')
        depth_postp = inputs.get('depth_postp', None)
        assert img.dim() == 5, 'The img must be 5D Tensor (N, T, C, H, W).'
        assert img.size(1) == 1, 'one key frame per sample (T = 1)'
        N = img.size(0)
        assert len(data_samples) == N
        data = dict(img=img[:, 0])
        if disp_postp is not None:
            data['disp_postp'] = disp_postp[:, 0]
        elif self.stereo is not None and inputs.get('right') is not None:
            data['right'] = inputs['right'][:, 0]   # disp_postp is computed by the stereo module
        else:
            raise KeyError("inputs need 'disp_postp', or 'right' with a stereo module configured")
        if disp_mask is not None:
            data['disp_mask'] = disp_mask[:, 0]
        det_results = self.detector.predict(data, data_samples)   # batched dense path
        outs = []
        for n in range(N):                                      # sequential association, frame order
            sample = data_samples[n]
            det = det_results[n].pred_instances
            disp_n = data['disp_postp'][n:n + 1]
            gt_n = depth_postp[n] if depth_postp is not None else None
            scaled, _ = self.bbox_postp_depth(det.clone(), disp_n, gt_n)
            sample.pred_det_instances = scaled
            tracks = self.tracker.track(model=self, img=data['img'][n:n + 1], feats=None, data_sample=sample,
                                        **kwargs)
            tracks['bboxes'] = scale_bbox(tracks.bboxes, 1 / tracks.scales)      # unscale
            _, depth = self.bbox_postp_depth(tracks.clone(), disp_n, gt_n)
            tracks['depth'] = depth['d_values']
            tracks['gt_depth'] = depth.get('gt_d_values', depth['d_values'])
            sample.pred_det_instances = det.clone()
            sample.pred_track_instances = tracks
            outs.append(sample)
        return outs
