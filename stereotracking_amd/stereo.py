"""StereoCostVolume — the stereo matching module BASELINE.json's north_star adds in front of the
detector's disparity branch.  The reference has no such module (disparity comes from offline
OpenCV SGBM PNGs, reproducibility.md:166-194, loaded by
mmtrack/datasets/transforms/loading_disparity.py:71-134); what it fixes is the CONSUMER contract,
which this module honours: `disp_postp` (N,3,H,W) float32 pixels, 0 = invalid / padding
(loading_disparity.py:85-86,129-134; transforms_disparity.py:234-249), channel 0 feeding
disp2depth (mmtrack/models/mot/ocsort_disparity.py:115,132-134).

Specification (frozen; the executable spec is oracle/st_oracle.c + oracle/stereo.py):
  features   F_L, F_R = stem+stage1 of the detector's RGB branch on left / right (C x H/4 x W/4),
             shared weights, computed as one stacked batch of 2N
  cost       cost[d,y,x] = (1/C) sum_c F_L[c,y,x] * F_R[c,y,x-d], d in [0, max_disp/4); 0 where x-d < 0
  aggregate  3-D: `agg3d_layers` single-channel 3x3x3 convolutions over (d, y, x) of the volume, zero padded, SiLU
             after every layer but the last (parameters `agg3d.{l}.weight` (1,1,3,3,3), `agg3d.{l}.bias` (1,);
             csrc/agg3d.hip, bit-exact against oracle_agg3d); then
             2-D: `agg_layers` convolutions over d-as-channels: cost <- conv3x3(cost; W_l, b_l), SiLU after
             every layer but the last (parameters `agg.{l}.weight` (D',D',3,3), `agg.{l}.bias` (D',))
  disparity  d_lr = sum_d d * softmax_d(temperature * cost);  disp = 4 * bilinear_x4(d_lr)
             inside the original image, 0 in the padding

FULL-RESOLUTION mode (`full_res=True`, round 5): north_star's literal sizing - "a D x H x W cost volume, its 3D ...
aggregation and soft-argmin" with D = max_disp levels at IMAGE resolution (192 x 720 x 1280: 177 M cells, 708 MB per pair):
  features   G = reduce(F): a 1x1 convolution C -> `full_res_channels` (8), bias, no activation (`reduce.weight`,
             `reduce.bias`), then bilinear x4 (align_corners=False) to H x W  (st_feat_upsample)
  cost       cost[d,Y,X] = (1/8) sum_c G_L[c,Y,X] * G_R[c,Y,X-d], d in [0, max_disp) pixels; 0 where X-d < 0
             (st_costvolume_softargmin, materialised in slabs of <= 128 levels)
  aggregate  `agg3d_layers` 3x3x3 layers over (d, Y, X) (st_volume_agg3d); no 2-D stage (a 3x3 convolution over 192
             levels-as-channels at image resolution is 625 GFLOP per pair).  With at least one layer, cost volume and
             first layer run as ONE kernel (st_costvolume_agg3d: the volume between them never reaches memory; same bits)
  disparity  disp = sum_d d * softmax_d(temperature * cost) in pixels, no upsampling step (st_softargmin), 0 outside the
             original image, three identical channels (st_disp_upsample_pack with scale 1)
A raw-pixel correlation at full resolution was withdrawn in round 3 (not a matcher); this mode correlates learned stage-1
FEATURES brought to image resolution.  It runs at ~0.53 of the default module's rate (bench.py --fullres-leg) and exists so
that the literal sizing is a tested, benched product path, not only a kernel measurement.
"""
import ctypes as C

import torch
from torch import nn

from . import _lib
from ._lib import StConvDesc, check, current_stream, ptr
from .engine import RawChunk
from .registry import MODELS


@MODELS.register_module()
class StereoCostVolume(nn.Module):
    """Parameter holder + launcher (like the detector modules: no CPU forward).  Parameters are named
    `agg.{l}.weight` / `agg.{l}.bias`, so under the MOT shell a checkpoint carries `stereo.agg.{l}.*`."""

    def __init__(self, max_disp=192, feat_stride=4, temperature=32.0, agg_layers=0, agg3d_layers=0, full_res=False,
                 full_res_channels=8, feat_channels=64):
        super().__init__()
        if feat_stride != 4:
            raise NotImplementedError('only feat_stride=4 (stage1 features) is wired up')
        if max_disp % feat_stride:
            raise ValueError('max_disp must be a multiple of feat_stride')
        self.max_disp, self.feat_stride = int(max_disp), int(feat_stride)
        self.full_res = bool(full_res)
        self.levels = self.max_disp if self.full_res else self.max_disp // self.feat_stride
        self.reduce = None
        if self.full_res:
            if agg_layers:
                raise ValueError('full_res aggregates with 3-D layers only (agg_layers must be 0)')
            if self.levels % 16 or not 16 <= self.levels <= 192:
                raise ValueError('full_res needs max_disp to be a multiple of 16 in [16, 192]')
            if full_res_channels % 4 or feat_channels % 4:
                raise ValueError('full_res_channels / feat_channels must be multiples of 4')
            if not 4 <= full_res_channels <= feat_channels:
                raise ValueError(f'full_res needs 4 <= full_res_channels <= feat_channels (got {full_res_channels} of '
                                 f'{feat_channels}): the reduce conv starts as a pass-through of the first channels.  '
                                 'The fused cost-volume + 3-D kernel takes 4, 8 or 16 channels; any other count runs '
                                 'the two-call form (st_costvolume_softargmin -> st_volume_agg3d), same results')
            self.reduce = nn.Conv2d(int(feat_channels), int(full_res_channels), 1)
            with torch.no_grad():   # until a checkpoint is loaded: the first channels pass through
                self.reduce.weight.zero_()
                idx = torch.arange(int(full_res_channels))
                self.reduce.weight[idx, idx, 0, 0] = 1.0
                self.reduce.bias.zero_()
        if self.levels % 4 and (agg_layers or agg3d_layers):
            raise ValueError('aggregation needs max_disp / feat_stride to be a multiple of 4')
        self.temperature = float(temperature)
        self.agg_layers = int(agg_layers)
        self.agg3d_layers = int(agg3d_layers)
        D = self.levels
        self.agg3d = nn.ModuleList(nn.Conv3d(1, 1, 3, padding=1) for _ in range(self.agg3d_layers))
        self.agg = nn.ModuleList(nn.Conv2d(D, D, 3, padding=1) for _ in range(self.agg_layers))
        with torch.no_grad():   # identity until a checkpoint is loaded: behaves like agg_layers=0
            for conv in self.agg:
                conv.weight.zero_()
                conv.weight[torch.arange(D), torch.arange(D), 1, 1] = 1.0
                conv.bias.zero_()
            for conv in self.agg3d:
                conv.weight.zero_()
                conv.weight[0, 0, 1, 1, 1] = 1.0
                conv.bias.zero_()
        for p in self.parameters():
            p.requires_grad_(False)
        self.lib = _lib.load()
        self._packed = None    # (device, weights version, [(wgt, bias)])
        self._red = None       # (device, weights version, packed reduce weights, bias)
        self._fr = None        # full-resolution buffers: reduced features, upsampled features, two volumes, disparity
        self.fuse_first_layer = True   # full-resolution mode: cost volume + first 3-D layer in one kernel (tools may clear it)
        # ... and, when that is the ONLY 3-D layer, optionally the soft-argmin in the same kernel (st_costvolume_agg3d_softargmin):
        # bit-equal, and the (N,H,W,D) volume - 5.8 GB per context at the bench size - is never allocated; OFF by default
        # because it is SLOWER on MI355X (3.9-4.0 ms against 2.14 + 1.09 ms per 8 pairs, profiles/r06_fullres_single_kernel_ab.txt:
        # the oracle's in-order sums are a serial chain of 2 D dependent operations that only TW = 16 lanes of a workgroup can run)
        self.fuse_softargmin = False
        self._fr_fused = False
        self._taps3d = None    # (weights version, [(27 host floats as a ctypes array, bias)])
        self._vol = None
        self.variant = -1      # conv tile variant of the aggregation layers (-1 = library default, or autotune())
        self.timing = False    # record events around every aggregation conv (bench.py roofline accounting)
        self._events = []
        self._cv_events = []

    def forward(self, *args, **kwargs):
        raise RuntimeError('StereoCostVolume has no CPU forward: use compute() with a HipDetector context')

    # ---- parameters (aggregation convs) ---------------------------------------------------------------
    def param_table(self):
        return [(n, tuple(p.shape)) for n, p in self.named_parameters()]

    def _pack3d(self):
        ver = tuple(p._version for p in self.agg3d.parameters())
        if self._taps3d is None or self._taps3d[0] != ver:
            taps = []
            for conv in self.agg3d:
                w = conv.weight.detach().to('cpu', torch.float32).reshape(27).tolist()   # (kD, kH, kW) order
                taps.append(((C.c_float * 27)(*w), float(conv.bias.detach().to('cpu', torch.float32).reshape(-1)[0])))
            self._taps3d = (ver, taps)
        return self._taps3d[1]

    def _pack(self, dev):
        ver = tuple(p._version for p in self.agg.parameters())
        if self._packed is not None and self._packed[0] == dev and self._packed[1] == ver:
            return self._packed[2]
        D = self.levels
        packed = []
        for conv in self.agg:
            w = conv.weight.detach().to('cpu', torch.float32).contiguous()
            b = conv.bias.detach().to('cpu', torch.float32).contiguous()
            nf = self.lib.st_conv_packed_floats(D, D, 3, 3)
            wp = torch.empty(nf, dtype=torch.float32)
            bp = torch.empty((D + 31) // 32 * 32, dtype=torch.float32)
            check(self.lib.st_conv_pack_weights(ptr(w), ptr(b), None, None, None, None, 0.0, D, D, 3, 3, ptr(wp),
                                                ptr(bp)), 'st_conv_pack_weights')
            wn = None   # the same weights in Winograd form (kernel instance 43), when the shape allows it
            nw = self.lib.st_wino_packed_floats(D, D)
            if nw:
                wn = torch.empty(nw, dtype=torch.float32)
                check(self.lib.st_wino_pack_weights(ptr(wp), D, D, ptr(wn)), 'st_wino_pack_weights')
                wn = wn.to(dev)
            packed.append((wp.to(dev), bp.to(dev), wn))
        self._packed = (dev, ver, packed)
        return packed

    def _agg_desc(self, l, src, dst, wp, bp, wn=None):
        N, Hf, Wf, D = src.shape
        d = StConvDesc()
        d.in_dev = src.data_ptr(); d.N, d.Hi, d.Wi, d.Cin, d.in_ld, d.in_off = N, Hf, Wf, D, D, 0
        d.wgt_dev = wp.data_ptr(); d.bias_dev = bp.data_ptr()
        d.Cout, d.KH, d.KW, d.stride, d.pad = D, 3, 3, 1, 1
        d.out1_dev = dst.data_ptr(); d.out1_ld, d.out1_off, d.split = D, 0, D
        d.act = 1 if l < self.agg_layers - 1 else 0
        d.wgt_wino_dev = wn.data_ptr() if wn is not None else None
        return d

    @staticmethod
    def _scratch_key(dev):
        """Scratch buffers (volumes, full-resolution stages) are owned by the STREAM that launches on them: a module shared
        by several engines / streams (the non-dense MOT shell path calls `detector.stereo.compute(eng, ...)` from every
        context) gets one set per (device, stream), so two batches in flight never write the same volume."""
        return (str(dev), int(torch.cuda.current_stream(dev).cuda_stream))

    def _volumes(self, dev, N, Hf, Wf):
        D = self.levels
        if self._vol is None:
            self._vol = {}
        k = self._scratch_key(dev)
        v = self._vol.get(k)
        if v is None or v[0].shape != (N, Hf, Wf, D):
            v = self._vol[k] = (torch.zeros(N, Hf, Wf, D, dtype=torch.float32, device=dev),
                                torch.zeros(N, Hf, Wf, D, dtype=torch.float32, device=dev))
        return v

    def autotune(self, dev, N, Hf, Wf, reps=5, candidates=tuple(range(22)) + (42, 43)):
        """Pick the aggregation convs' tile variant by measurement (same policy as st_detector_autotune:
        min over `reps` individually timed launches).  Returns the chosen variant id."""
        if not self.agg_layers:
            return -1
        packed = self._pack(dev)
        va, vb = self._volumes(dev, N, Hf, Wf)
        d = self._agg_desc(0, va, vb, *packed[0])
        stream = current_stream()
        best, best_ms = -1, float('inf')
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for v in candidates:
            if self.lib.st_conv2d_nhwc_variant(C.byref(d), stream, int(v)) != 0:
                continue   # tile does not divide the padded Cout
            ms = float('inf')
            for _ in range(reps):
                e0.record()
                self.lib.st_conv2d_nhwc_variant(C.byref(d), stream, int(v))
                e1.record()
                e1.synchronize()
                ms = min(ms, e0.elapsed_time(e1))
            if best < 0 or ms < best_ms * 0.97:   # 3 % hysteresis, as st_detector_autotune
                best, best_ms = int(v), ms
        self.variant = best
        return best

    def agg_macs(self, N, Hf, Wf):
        return float(N) * Hf * Wf * self.levels * self.levels * 9

    def pop_times(self):
        """[(variant, ms)] of the aggregation convs recorded since the last call (timing=True; syncs)."""
        out = []
        for v, e0, e1 in self._events:
            e1.synchronize()
            out.append((v, e0.elapsed_time(e1)))
        self._events = []
        return out

    def pop_costvolume_time(self):
        """ms of the cost-volume launches recorded since the last call (timing=True; syncs)."""
        t = 0.0
        for e0, e1 in self._cv_events:
            e1.synchronize()
            t += e0.elapsed_time(e1)
        self._cv_events = []
        return t

    # ---- compute ------------------------------------------------------------------------------------------
    def compute(self, engine, img, right, valid_hw, disp_lr=None, disp_postp=None, cost_out=None):
        """engine: a HipDetector built with stereo=True.  img/right: (N,3,H,W) fp32 CUDA, or two RawChunk (uint8 frames).
        Runs phase 0 (features of left+right) then cost volume (+ aggregation) + soft-argmin + upsample.
        Returns disp_postp (N,3,H,W); phase-0 activations stay in the engine workspace for phase 1."""
        if not engine.stereo:
            raise ValueError('StereoCostVolume needs a detector context built with stereo=True')
        N, H, W = engine.batch, engine.height, engine.width
        s = self.feat_stride
        dev = img.device
        D = self.levels
        if isinstance(img, RawChunk):      # uint8 frames: the stem converts + pads them itself
            engine.forward_phase0_raw(img, right)
        else:
            engine.forward_phase(0, img=img, right=right)
        feat = engine.tap('stage1_rgb')  # (2N, H/4, W/4, C): [left | right]
        Hf, Wf, Cf, ld = feat.shape[1], feat.shape[2], feat.shape[3], feat.stride(2)
        fl = feat.data_ptr()
        fr = fl + N * Hf * Wf * ld * 4
        if disp_lr is None:
            disp_lr = torch.empty(N, Hf, Wf, dtype=torch.float32, device=dev)
        if disp_postp is None:
            disp_postp = torch.empty(N, 3, H, W, dtype=torch.float32, device=dev)
        stream = current_stream()
        if self.full_res:
            return self._compute_full_res(feat, N, Hf, Wf, Cf, ld, valid_hw, disp_postp, cost_out, stream, dev, H, W)
        if self.timing:
            cv0, cv1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            cv0.record()
        if self.agg_layers == 0 and self.agg3d_layers == 0:
            check(self.lib.st_costvolume_softargmin(C.c_void_p(fl), C.c_void_p(fr), N, Hf, Wf, Cf, ld, D,
                                                    self.temperature, ptr(cost_out), ptr(disp_lr), stream),
                  'st_costvolume_softargmin')
            if self.timing:
                cv1.record()
                self._cv_events.append((cv0, cv1))
        else:
            packed = self._pack(dev)
            va, vb = self._volumes(dev, N, Hf, Wf)
            # materialise the D' x Hf x Wf volume (d innermost = NHWC), aggregate it with the conv kernel
            check(self.lib.st_costvolume_softargmin(C.c_void_p(fl), C.c_void_p(fr), N, Hf, Wf, Cf, ld, D,
                                                    self.temperature, ptr(va), None, stream),
                  'st_costvolume_softargmin')
            if self.timing:
                cv1.record()
                self._cv_events.append((cv0, cv1))
            for l, (w27, b3) in enumerate(self._pack3d()):      # 3-D aggregation over (d, y, x)
                check(self.lib.st_volume_agg3d(ptr(va), ptr(vb), N, Hf, Wf, D, w27, b3,
                                               1 if l < self.agg3d_layers - 1 else 0, stream), 'st_volume_agg3d')
                va, vb = vb, va
            for l, (wp, bp, wn) in enumerate(packed):
                d = self._agg_desc(l, va, vb, wp, bp, wn)
                if self.timing:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                check(self.lib.st_conv2d_nhwc_variant(C.byref(d), stream, self.variant), 'st_conv2d_nhwc(agg)')
                if self.timing:
                    e1.record()
                    self._events.append((self.variant, e0, e1))
                va, vb = vb, va
            if cost_out is not None:
                cost_out.copy_(va)
            check(self.lib.st_softargmin(ptr(va), N, Hf, Wf, D, self.temperature, ptr(disp_lr), stream),
                  'st_softargmin')
        check(self.lib.st_disp_upsample_pack(ptr(disp_lr), N, Hf, Wf, s, H, W, int(valid_hw[0]), int(valid_hw[1]),
                                             ptr(disp_postp), stream), 'st_disp_upsample_pack')
        return disp_postp

    # ---- full-resolution mode -------------------------------------------------------------------------------
    def _pack_reduce(self, dev):
        ver = tuple(p._version for p in self.reduce.parameters())
        if self._red is not None and self._red[0] == dev and self._red[1] == ver:
            return self._red[2], self._red[3]
        w = self.reduce.weight.detach().to('cpu', torch.float32).contiguous()
        b = self.reduce.bias.detach().to('cpu', torch.float32).contiguous()
        Cr, Cf = w.shape[0], w.shape[1]
        wp = torch.empty(self.lib.st_conv_packed_floats(Cr, Cf, 1, 1), dtype=torch.float32)
        bp = torch.empty((Cr + 31) // 32 * 32, dtype=torch.float32)
        check(self.lib.st_conv_pack_weights(ptr(w), ptr(b), None, None, None, None, 0.0, Cr, Cf, 1, 1, ptr(wp), ptr(bp)),
              'st_conv_pack_weights')
        self._red = (dev, ver, wp.to(dev), bp.to(dev))
        return self._red[2], self._red[3]

    def full_res_buffers(self, dev, N, Hf, Wf, need_volume=True):
        """Persistent buffers of the full-resolution mode: reduced features (2N,Hf,Wf,Cr), upsampled features
        (2N,H,W,Cr), two volumes (N,H,W,D) (cost / aggregation ping-pong) and the disparity (N,H,W)."""
        Cr, D, s = self.reduce.out_channels, self.levels, self.feat_stride
        H, W = Hf * s, Wf * s
        if self._fr is None:
            self._fr = {}
        k = self._scratch_key(dev)          # one set per (device, launching stream): see _scratch_key
        fr = self._fr.get(k)
        if fr is None or fr['red'].shape != (2 * N, Hf, Wf, Cr):
            f32 = dict(dtype=torch.float32, device=dev)
            fr = self._fr[k] = dict(red=torch.empty(2 * N, Hf, Wf, Cr, **f32), up=torch.empty(2 * N, H, W, Cr, **f32),
                                    va=None, disp=torch.empty(N, H, W, **f32))
            fr['vb'] = None
        # first volume (N,H,W,D: 5.8 GB at the bench size): not needed at all when cost volume, the only 3-D layer and the
        # soft-argmin run as one kernel; allocated on first need (`need_volume`)
        if need_volume and fr['va'] is None:
            fr['va'] = torch.empty(N, H, W, D, dtype=torch.float32, device=dev)
        # second volume: only where a layer runs volume -> volume (the first layer fused with the cost volume writes straight
        # into `va`); allocated on first need (a tool may switch `fuse_first_layer` off on a live module)
        fused = self.fuse_first_layer and self.lib.st_costvolume_agg3d_supported(Cr, D) == 1
        if need_volume and fr['vb'] is None and self.agg3d_layers > (1 if fused else 0):
            fr['vb'] = torch.empty(N, H, W, D, dtype=torch.float32, device=dev)
        return fr

    def _compute_full_res(self, feat, N, Hf, Wf, Cf, ld, valid_hw, disp_postp, cost_out, stream, dev, H, W):
        if Cf != self.reduce.in_channels:
            raise ValueError(f'full_res: the detector\'s stage-1 features have {Cf} channels, the module was built for '
                             f'feat_channels={self.reduce.in_channels}')
        s, D, Cr = self.feat_stride, self.levels, self.reduce.out_channels
        single_kernel = (self.agg3d_layers == 1 and cost_out is None and self.fuse_first_layer and self.fuse_softargmin and
                         D in (48, 96, 192) and self.lib.st_costvolume_agg3d_supported(Cr, D) == 1)
        b = self.full_res_buffers(dev, N, Hf, Wf, need_volume=not single_kernel)
        wp, bp = self._pack_reduce(dev)
        ev = []

        def mark():
            if self.timing:
                e = torch.cuda.Event(enable_timing=True)
                e.record()
                ev.append(e)
        mark()
        d = StConvDesc()                                   # G = reduce(F): 1x1, C -> Cr, bias, no activation, 2N images
        d.in_dev = feat.data_ptr(); d.N, d.Hi, d.Wi, d.Cin, d.in_ld, d.in_off = 2 * N, Hf, Wf, Cf, ld, 0
        d.wgt_dev = wp.data_ptr(); d.bias_dev = bp.data_ptr()
        d.Cout, d.KH, d.KW, d.stride, d.pad = Cr, 1, 1, 1, 0
        d.out1_dev = b['red'].data_ptr(); d.out1_ld, d.out1_off, d.split = Cr, 0, Cr
        d.act, d.post_scale = 0, 1.0
        check(self.lib.st_conv2d_nhwc(C.byref(d), stream), 'st_conv2d_nhwc(reduce)')
        check(self.lib.st_feat_upsample(ptr(b['red']), 2 * N, Hf, Wf, Cr, Cr, s, ptr(b['up']), stream), 'st_feat_upsample')
        mark()
        gl = b['up'].data_ptr()
        gr = gl + N * H * W * Cr * 4
        va, vb = b['va'], b['vb']
        layers = self._pack3d()
        # cost volume and the first 3-D layer in one pass when the kernel takes the shape (the volume between them - 6.3 GB
        # per 8 pairs at D = 192 - then never reaches memory); the two-call form otherwise.  Same bits either way.
        fused = bool(layers) and self.fuse_first_layer and self.lib.st_costvolume_agg3d_supported(Cr, D) == 1
        # ONE 3-D layer, nobody asks for the volume and `fuse_softargmin` is set (opt-in, see __init__): the soft-argmin is taken
        # inside that kernel too (round 6, st_costvolume_agg3d_softargmin: the 4 D bytes per pixel of aggregated volume are
        # neither allocated, written nor read back; bit-equal to the calls below)
        single = single_kernel and fused and len(layers) == 1
        if single:
            w27, b3 = layers[0]
            check(self.lib.st_costvolume_agg3d_softargmin(C.c_void_p(gl), C.c_void_p(gr), N, H, W, Cr, Cr, D, w27, b3, 0,
                                                          self.temperature, ptr(b['disp']), stream),
                  'st_costvolume_agg3d_softargmin')
            mark()
            mark()
            check(self.lib.st_disp_upsample_pack(ptr(b['disp']), N, H, W, 1, H, W, int(valid_hw[0]), int(valid_hw[1]),
                                                 ptr(disp_postp), stream), 'st_disp_upsample_pack')
            mark()
            if self.timing:
                self._fr_events = ev
                self._fr_fused = 'softargmin'
            return disp_postp
        if fused:
            w27, b3 = layers[0]
            check(self.lib.st_costvolume_agg3d(C.c_void_p(gl), C.c_void_p(gr), N, H, W, Cr, Cr, D, w27, b3,
                                               1 if self.agg3d_layers > 1 else 0, ptr(va), stream), 'st_costvolume_agg3d')
            layers = layers[1:]
        else:
            check(self.lib.st_costvolume_softargmin(C.c_void_p(gl), C.c_void_p(gr), N, H, W, Cr, Cr, D, self.temperature,
                                                    ptr(va), None, stream), 'st_costvolume_softargmin')
        mark()
        for l, (w27, b3) in enumerate(layers):
            check(self.lib.st_volume_agg3d(ptr(va), ptr(vb), N, H, W, D, w27, b3,
                                           1 if l < len(layers) - 1 else 0, stream), 'st_volume_agg3d')
            va, vb = vb, va
        mark()
        if cost_out is not None:
            cost_out.copy_(va)
        check(self.lib.st_softargmin(ptr(va), N, H, W, D, self.temperature, ptr(b['disp']), stream), 'st_softargmin')
        check(self.lib.st_disp_upsample_pack(ptr(b['disp']), N, H, W, 1, H, W, int(valid_hw[0]), int(valid_hw[1]),
                                             ptr(disp_postp), stream), 'st_disp_upsample_pack')
        mark()
        if self.timing:
            self._fr_events = ev
            self._fr_fused = fused
        return disp_postp

    def pop_full_res_times(self):
        """ms of the stages of the last full-resolution compute (timing=True): features (reduce + upsample), cost
        volume, 3-D aggregation, soft-argmin + pack - or, when the first 3-D layer ran fused with the cost volume:
        features, cost volume + first layer, remaining layers, soft-argmin + pack."""
        ev = getattr(self, '_fr_events', None)
        if not ev:
            return None
        ev[-1].synchronize()
        if self._fr_fused == 'softargmin':     # features | cost volume + 3-D layer + soft-argmin in ONE kernel | - | pack
            names = ('features_reduce_upsample', 'cost_volume_agg3d_softargmin', 'agg3d_rest', 'pack')
            return {n: ev[i].elapsed_time(ev[i + 1]) for i, n in enumerate(names)}
        names = (('features_reduce_upsample', 'cost_volume_agg3d_first', 'agg3d_rest', 'softargmin_pack') if self._fr_fused
                 else ('features_reduce_upsample', 'cost_volume', 'agg3d', 'softargmin_pack'))
        return {n: ev[i].elapsed_time(ev[i + 1]) for i, n in enumerate(names)}
