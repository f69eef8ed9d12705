"""Thin host-side owner of the HIP detector context (ctypes over include/stereotrack.h).

PyTorch is plumbing here: it owns device memory (workspace, inputs, outputs) and the stream;
every FLOP of the path runs inside libstereotrack_hip.so.  There is no CPU fallback.
"""
import ctypes as C
from collections import OrderedDict

import torch

from . import _lib
from ._lib import StDecodeDesc, StDetectorConfig, check, current_stream, ptr


def _require_cuda(t, name):
    if not (isinstance(t, torch.Tensor) and t.is_cuda):
        raise RuntimeError(f'{name} must be a CUDA (ROCm) tensor: stereotracking_amd runs only on the '
                           f'HIP path (got {type(t).__name__} on '
                           f'{getattr(t, "device", "?")})')
    if t.dtype != torch.float32 or not t.is_contiguous():
        raise RuntimeError(f'{name} must be contiguous float32')


class RawChunk:
    """One batch of RAW frames for the fused stem: `frames` = list of B contiguous uint8 CUDA tensors (3,h,w) or
    (1,3,h,w) in allocations of their own, converted (cast + pad to the detector's H x W with `pad_value`) inside the
    stem kernel's window staging (st_detector_forward_phase0_raw) - no fp32 copy of the images in HBM."""

    def __init__(self, frames, pad_value):
        self.frames, self.pad_value = list(frames), float(pad_value)
        f0 = self.frames[0]
        self.hw = (int(f0.shape[-2]), int(f0.shape[-1]))
        self.device = f0.device
        for f in self.frames:
            if not (f.is_cuda and f.dtype == torch.uint8 and f.is_contiguous() and f.numel() == 3 * self.hw[0] * self.hw[1]
                    and tuple(f.shape[-2:]) == self.hw):
                raise RuntimeError('RawChunk: frames must be contiguous uint8 CUDA tensors (3,h,w) of one size')

    def __len__(self):
        return len(self.frames)

    @staticmethod
    def supported(frames, pad_value):
        """The stem's raw form needs: width % 4 == 0, 4-byte aligned contiguous frames, an integral pad value."""
        f0 = frames[0]
        return (float(pad_value) == int(pad_value) and 0 <= pad_value <= 255 and f0.shape[-1] % 4 == 0 and len(frames) <= 32
                and all(f.is_cuda and f.dtype == torch.uint8 and f.is_contiguous() and f.data_ptr() % 4 == 0 and
                        f.shape[-3:] == f0.shape[-3:] and f.shape[-3] == 3 for f in frames))

    def record_stream(self, stream):
        for f in self.frames:
            f.record_stream(stream)

    def table(self):
        return (C.c_void_p * len(self.frames))(*[f.data_ptr() for f in self.frames])


class HipDetector:
    """Two-branch YOLOX detector context for a fixed (batch, H, W).

    Mirrors YOLODetector_Disparity_V1._forward (reference
    mmtrack/models/detectors/yolo_detector_disparity_v1.py:127-142) + bbox_head.predict
    (:121-122) as two C-ABI calls: st_detector_forward and st_decode_nms.
    """

    def __init__(self, batch, height, width, widen_factor=0.5, deepen_factor=0.33, num_classes=1,
                 bn_eps=1e-3, stereo=False, disp_replicated=None, rgb_only=False):
        """disp_replicated: the three planes of disp_postp are identical (a 3-channel repeat of one map), so the
        disparity stem may read plane 0 with plane-summed weights.  Default: True for stereo contexts (the
        disparity then comes from st_disp_upsample_pack, which writes exactly that), False otherwise.
        rgb_only: the single-branch detector of the reference's RGB configuration (backbone `mmtrack.CSPDarknet`,
        csp_darknet.py:8-13): no disparity branch in the plan or the parameter table; `disp` of the forward calls is
        ignored and may be None."""
        self.lib = _lib.load()
        self.batch, self.height, self.width = int(batch), int(height), int(width)
        self.stereo = bool(stereo)
        self.rgb_only = bool(rgb_only)
        self.widen_factor, self.deepen_factor = float(widen_factor), float(deepen_factor)
        self.disp_replicated = self.stereo if disp_replicated is None else bool(disp_replicated)
        cfg = StDetectorConfig(C.sizeof(StDetectorConfig), float(widen_factor), float(deepen_factor),
                               int(num_classes), self.batch, self.height, self.width, float(bn_eps),
                               int(self.stereo), int(self.disp_replicated), int(self.rgb_only))
        h = C.c_void_p()
        check(self.lib.st_detector_create(C.byref(cfg), C.byref(h)), 'st_detector_create')
        self.handle = h
        self.num_classes = int(num_classes)
        self.multi_label = True      # test_cfg.multi_label (several classes): False = one candidate per prior (its best class)
        self._finalized = False
        self._ws = None
        self._dec_ws = None
        self.levels = []
        for l in range(self.lib.st_detector_num_levels(self.handle)):
            hh, ww, ss, off = C.c_int(), C.c_int(), C.c_int(), C.c_size_t()
            check(self.lib.st_detector_level_info(self.handle, l, C.byref(hh), C.byref(ww), C.byref(ss),
                                                  C.byref(off)))
            self.levels.append((hh.value, ww.value, ss.value, off.value))
        self.head_floats = self.lib.st_detector_head_floats(self.handle)
        self.num_priors = sum(h_ * w_ for h_, w_, _, _ in self.levels)
        self.macs = self.lib.st_detector_macs(self.handle)

    def __del__(self):
        h = getattr(self, 'handle', None)
        if h:
            self.lib.st_detector_destroy(h)
            self.handle = None

    # ---- parameters ------------------------------------------------------------------------
    def param_table(self):
        """[(name, shape)] in reference state_dict naming (float tensors only)."""
        out = []
        name = C.create_string_buffer(256)
        shape = (C.c_int64 * 4)()
        nd = C.c_int()
        for i in range(self.lib.st_detector_num_params(self.handle)):
            check(self.lib.st_detector_param_info(self.handle, i, name, 256, shape, C.byref(nd)))
            out.append((name.value.decode(), tuple(shape[j] for j in range(nd.value))))
        return out

    def load_state_dict(self, sd, prefix=''):
        """Set every parameter from a {name: tensor} mapping (extra keys such as
        `num_batches_tracked` are ignored), fold BN in fp64 and upload."""
        for name, shape in self.param_table():
            key = prefix + name
            if key not in sd:
                raise KeyError(f'state_dict is missing "{key}"')
            t = sd[key].detach().to('cpu', torch.float32).contiguous()
            if tuple(t.shape) != shape:
                raise ValueError(f'{key}: expected shape {shape}, got {tuple(t.shape)}')
            check(self.lib.st_detector_set_param(self.handle, name.encode(), ptr(t), t.numel()),
                  f'st_detector_set_param({name})')
        check(self.lib.st_detector_finalize(self.handle), 'st_detector_finalize')
        self._finalized = True

    def autotune(self, device=None, reps=5):
        """Pick the fastest conv tile variant per layer by measurement (st_detector_autotune)."""
        device = torch.device(device) if device is not None else torch.device('cuda', torch.cuda.current_device())
        ws = self._workspace(device)
        head = torch.empty(self.head_floats, dtype=torch.float32, device=device)
        check(self.lib.st_detector_autotune(self.handle, ptr(ws), ws.numel(), ptr(head), current_stream(), int(reps)),
              'st_detector_autotune')
        torch.cuda.synchronize(device)

    def set_split(self, allow):
        """Allow / forbid (default) the split-operand (bf16x3) conv instances 50-52 in the autotune search."""
        import os
        mask = int(os.environ.get('ST_SPLIT_MASK', '0x3F'), 0) if allow else 0   # tools: restrict the instances searched
        check(self.lib.st_detector_set_split(self.handle, mask), 'st_detector_set_split')

    def get_tuning(self):
        n = self.lib.st_detector_num_ops(self.handle)
        arr = (C.c_int * n)()
        check(self.lib.st_detector_get_tuning(self.handle, arr, n), 'st_detector_get_tuning')
        return list(arr)

    def set_tuning(self, variants):
        arr = (C.c_int * len(variants))(*[int(v) for v in variants])
        check(self.lib.st_detector_set_tuning(self.handle, arr, len(variants)), 'st_detector_set_tuning')

    # ---- forward -----------------------------------------------------------------------------
    def _workspace(self, device):
        if self._ws is None or self._ws.device != device:
            nbytes = self.lib.st_detector_workspace_bytes(self.handle)
            self._ws = torch.empty(nbytes, dtype=torch.uint8, device=device)
        return self._ws

    def forward(self, img, disp, head_out=None):
        """img, disp: (N,3,H,W) float32 CUDA tensors -> flat head buffer (see include/stereotrack.h).  img may be a
        RawChunk (uint8 frames: the RGB stem casts + pads them itself, st_detector_forward_raw)."""
        raw = isinstance(img, RawChunk)
        if self.rgb_only:
            disp = None          # not an input of the single-branch plan
        for t, nm in ((img, 'img'), (disp, 'disp_postp')):
            if nm == 'disp_postp' and self.rgb_only:
                continue
            if raw and nm == 'img':
                if len(img) != self.batch:
                    raise ValueError(f'raw chunk of {len(img)} frames for a batch-{self.batch} context')
                continue
            _require_cuda(t, nm)
            if tuple(t.shape) != (self.batch, 3, self.height, self.width):
                raise ValueError(f'{nm}: expected {(self.batch, 3, self.height, self.width)}, got {tuple(t.shape)}')
        ws = self._workspace(img.device)
        if head_out is None:
            head_out = torch.empty(self.head_floats, dtype=torch.float32, device=img.device)
        if raw:
            check(self.lib.st_detector_forward_raw(self.handle, img.table(), img.hw[0], img.hw[1], img.pad_value, ptr(disp),
                                                   ptr(ws), ws.numel(), current_stream(), ptr(head_out)),
                  'st_detector_forward_raw')
            return head_out
        check(self.lib.st_detector_forward(self.handle, ptr(img), ptr(disp), ptr(ws), ws.numel(),
                                           current_stream(), ptr(head_out)), 'st_detector_forward')
        return head_out

    def forward_phase(self, phase, img=None, disp=None, right=None, head_out=None):
        dev = next(t for t in (img, disp, right, head_out) if t is not None).device
        ws = self._workspace(dev)
        for t, nm in ((img, 'img'), (disp, 'disp_postp'), (right, 'right')):
            if t is not None:
                _require_cuda(t, nm)
        if phase == 1 and head_out is None:
            head_out = torch.empty(self.head_floats, dtype=torch.float32, device=dev)
        check(self.lib.st_detector_forward_phase(self.handle, int(phase), ptr(img), ptr(disp), ptr(right), ptr(ws),
                                                 ws.numel(), current_stream(), ptr(head_out)),
              'st_detector_forward_phase')
        return head_out

    def forward_phase0_raw(self, left, right=None):
        """Phase 0 from RawChunk inputs (uint8 frames; cast + pad inside the stem kernel)."""
        if len(left) != self.batch or (right is not None and len(right) != self.batch):
            raise ValueError(f'raw chunk of {len(left)} frames for a batch-{self.batch} context')
        if right is not None and (right.hw != left.hw or right.pad_value != left.pad_value):
            raise ValueError('left and right raw chunks differ in size / pad value')
        ws = self._workspace(left.device)
        check(self.lib.st_detector_forward_phase0_raw(self.handle, left.table(), right.table() if right is not None else None,
                                                      left.hw[0], left.hw[1], left.pad_value, ptr(ws), ws.numel(),
                                                      current_stream()), 'st_detector_forward_phase0_raw')

    def tap(self, name):
        """Internal NHWC activation as a strided torch view (N,H,W,C) into the workspace."""
        p = C.c_void_p()
        n, c, h, w, ld = C.c_int(), C.c_int(), C.c_int(), C.c_int(), C.c_int()
        ws = self._ws
        check(self.lib.st_detector_tap(self.handle, name.encode(), ptr(ws), C.byref(p), C.byref(n), C.byref(c),
                                       C.byref(h), C.byref(w), C.byref(ld)), 'st_detector_tap')
        off = (p.value - ws.data_ptr()) // 4
        flat = ws.view(torch.float32)
        return flat.as_strided((n.value, h.value, w.value, c.value),
                               (h.value * w.value * ld.value, w.value * ld.value, ld.value, 1), off)

    def head_levels(self, head_out):
        """Per level (N, h*w, row) views: [cls.., x, y, w, h, obj, padding]; row = st_head_row_floats(num_classes) (8 for
        the shipped 1..3-class heads)."""
        hr = self.head_row
        return [head_out[off:off + self.batch * h * w * hr].view(self.batch, h * w, hr)
                for h, w, _, off in self.levels]

    def head_nchw(self, head_out):
        """cls_scores, bbox_preds, objectnesses as the reference's _forward returns them:
        lists of (N,nc,h,w), (N,4,h,w), (N,1,h,w)."""
        nc = self.num_classes
        cls, reg, obj = [], [], []
        for (h, w, _, _), rows in zip(self.levels, self.head_levels(head_out)):
            x = rows.view(self.batch, h, w, self.head_row).permute(0, 3, 1, 2)
            cls.append(x[:, 0:nc])
            reg.append(x[:, nc:nc + 4])
            obj.append(x[:, nc + 4:nc + 5])
        return cls, reg, obj

    # ---- decode + NMS ------------------------------------------------------------------------
    @property
    def head_row(self):
        return int(self.lib.st_head_row_floats(self.num_classes))

    def decode_desc(self, score_thr, iou_thr, max_det, ori_shape, scale_factor=(1.0, 1.0), pad_param=None,
                    nms_mask_rows=0):
        d = StDecodeDesc()
        d.struct_size = C.sizeof(StDecodeDesc)
        d.batch = self.batch
        d.num_levels = len(self.levels)
        for l, (h, w, s, off) in enumerate(self.levels):
            d.level_h[l], d.level_w[l], d.level_stride[l], d.level_offset[l] = h, w, s, off
        d.score_thr, d.iou_thr, d.max_det = float(score_thr), float(iou_thr), int(max_det)
        d.scale_x, d.scale_y = float(scale_factor[0]), float(scale_factor[1])
        # pad_param = (top, bottom, left, right); predict_by_feat subtracts [left, top, left, top]
        d.pad_left = float(pad_param[2]) if pad_param is not None else 0.0
        d.pad_top = float(pad_param[0]) if pad_param is not None else 0.0
        d.ori_h, d.ori_w = float(ori_shape[0]), float(ori_shape[1])
        d.nms_mask_rows = int(nms_mask_rows)
        d.num_classes = self.num_classes
        d.single_label = 0 if self.multi_label else 1
        return d

    def decode_nms(self, head_out, score_thr=0.01, iou_thr=0.5, max_det=1000, ori_shape=None,
                   scale_factor=(1.0, 1.0), pad_param=None, nms_mask_rows=0, out=None):
        """-> boxes (N,max_det,4), scores (N,max_det), labels (N,max_det) int64,
        prior_idx (N,max_det) int32, counts (N,) int32 — all on device, no host sync.  counts[n] is the
        TRUE number kept; counts[n] > max_det means the buffer overflowed (the reference applies no cap
        under yolox_style=True) and the caller must raise or re-run with a larger max_det.  Rows past the count are
        written by the kernel (zero, prior index -1).  `out`: the five tensors to write into (persistent buffers of a
        pipeline context: no allocation, no clearing launch); default: fresh tensors."""
        _require_cuda(head_out, 'head_out')
        ori_shape = ori_shape or (self.height, self.width)
        d = self.decode_desc(score_thr, iou_thr, max_det, ori_shape, scale_factor, pad_param, nms_mask_rows)
        nbytes = self.lib.st_decode_nms_workspace_bytes(C.byref(d))
        if nbytes == 0:
            check(-1, 'st_decode_nms_workspace_bytes')
        dev = head_out.device
        if self._dec_ws is None or self._dec_ws.numel() < nbytes or self._dec_ws.device != dev:
            self._dec_ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        N = self.batch
        if out is None:
            out = self.decode_buffers(max_det, dev)
        boxes, scores, labels, prior, counts = out
        check(self.lib.st_decode_nms(C.byref(d), ptr(head_out), ptr(self._dec_ws), self._dec_ws.numel(),
                                     current_stream(), ptr(boxes), ptr(scores), ptr(labels), ptr(prior),
                                     ptr(counts)), 'st_decode_nms')
        return boxes, scores, labels, prior, counts

    def decode_buffers(self, max_det, dev):
        N = self.batch
        return (torch.empty(N, max_det, 4, dtype=torch.float32, device=dev),
                torch.empty(N, max_det, dtype=torch.float32, device=dev),
                torch.empty(N, max_det, dtype=torch.int64, device=dev),
                torch.empty(N, max_det, dtype=torch.int32, device=dev),
                torch.empty(N, dtype=torch.int32, device=dev))


def state_dict_from_table(table, tensors):
    """OrderedDict in table order (helper for tests / synthetic weights)."""
    return OrderedDict((n, tensors[n]) for n, _ in table)
