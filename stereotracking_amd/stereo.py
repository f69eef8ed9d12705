"""StereoCostVolume — the stereo matching module BASELINE.json's north_star adds in front of the
detector's disparity branch.  The reference has no such module (disparity comes from offline
OpenCV SGBM PNGs, reproducibility.md:166-194, loaded by
mmtrack/datasets/transforms/loading_disparity.py:71-134); what it fixes is the CONSUMER contract,
which this module honours: `disp_postp` (N,3,H,W) float32 pixels, 0 = invalid / padding
(loading_disparity.py:85-86,129-134; transforms_disparity.py:234-249), channel 0 feeding
disp2depth (mmtrack/models/mot/ocsort_disparity.py:115,132-134).

Specification (frozen; the executable spec is oracle/st_oracle.c):
  features   F_L, F_R = stem+stage1 of the detector's RGB branch on left / right (C x H/4 x W/4),
             shared weights, computed as one stacked batch of 2N
  cost       cost[d,y,x] = (1/C) sum_c F_L[c,y,x] * F_R[c,y,x-d], d in [0, max_disp/4); 0 where x-d < 0
  (agg)      optional 2-D aggregation convs over d-as-channels (agg_layers; 0 in this round)
  disparity  d_lr = sum_d d * softmax_d(temperature * cost);  disp = 4 * bilinear_x4(d_lr)
             inside the original image, 0 in the padding
"""
import ctypes as C

import torch

from . import _lib
from ._lib import check, current_stream, ptr
from .registry import MODELS


@MODELS.register_module()
class StereoCostVolume:
    def __init__(self, max_disp=192, feat_stride=4, temperature=32.0, agg_layers=0):
        if feat_stride != 4:
            raise NotImplementedError('only feat_stride=4 (stage1 features) is wired up')
        if max_disp % feat_stride:
            raise ValueError('max_disp must be a multiple of feat_stride')
        if agg_layers != 0:
            raise NotImplementedError('aggregation layers are not built yet (agg_layers=0)')
        self.max_disp, self.feat_stride = int(max_disp), int(feat_stride)
        self.levels = self.max_disp // self.feat_stride
        self.temperature = float(temperature)
        self.agg_layers = 0
        self.lib = _lib.load()

    def compute(self, engine, img, right, valid_hw, disp_lr=None, disp_postp=None, cost_out=None):
        """engine: a HipDetector built with stereo=True.  img/right: (N,3,H,W) fp32 CUDA.
        Runs phase 0 (features of left+right) then cost volume + soft-argmin + upsample.
        Returns disp_postp (N,3,H,W); phase-0 activations stay in the engine workspace for phase 1."""
        if not engine.stereo:
            raise ValueError('StereoCostVolume needs a detector context built with stereo=True')
        N, H, W = engine.batch, engine.height, engine.width
        s = self.feat_stride
        dev = img.device
        engine.forward_phase(0, img=img, right=right)
        feat = engine.tap('stage1_rgb')  # (2N, H/4, W/4, C): [left | right]
        Hf, Wf, Cf, ld = feat.shape[1], feat.shape[2], feat.shape[3], feat.stride(2)
        fl = feat.data_ptr()
        fr = fl + N * Hf * Wf * ld * 4
        if disp_lr is None:
            disp_lr = torch.empty(N, Hf, Wf, dtype=torch.float32, device=dev)
        if disp_postp is None:
            disp_postp = torch.empty(N, 3, H, W, dtype=torch.float32, device=dev)
        check(self.lib.st_costvolume_softargmin(C.c_void_p(fl), C.c_void_p(fr), N, Hf, Wf, Cf, ld, self.levels,
                                                self.temperature, ptr(cost_out), ptr(disp_lr), current_stream()),
              'st_costvolume_softargmin')
        check(self.lib.st_disp_upsample_pack(ptr(disp_lr), N, Hf, Wf, s, H, W, int(valid_hw[0]), int(valid_hw[1]),
                                             ptr(disp_postp), current_stream()), 'st_disp_upsample_pack')
        return disp_postp
