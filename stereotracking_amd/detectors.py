"""Host-side mirror of the reference's detector plugin classes, registered under the SAME type
strings and accepting the same config kwargs, with every FLOP delegated to the HIP library.

  YOLODetector_Disparity_V1             mmtrack/models/detectors/yolo_detector_disparity_v1.py:15-166
  YOLOXCSPDarknet_Disparity_V1_MMYOLO   mmtrack/models/backbones/csp_darknet_disparity_v1.py:16-206
  CSPDarknet (`mmtrack.CSPDarknet`)     mmtrack/models/backbones/csp_darknet.py:8-13 (mmdet's CSPDarknet reading x['img'])
  YOLODetector (`mmyolo.YOLODetector`)  the detector type of the RGB-only stereo config
                                        (configs/stereo_tracking/ocsort/yolox_s_mmyolo_mot_airdrone.py:40-42 over
                                        configs/_base_/yolox_s_8x8_mmyolo.py:17-19)
  YOLOXPAFPN / YOLOXHead / YOLOXHeadModule   mmyolo 0.2.0 (configs/_base_/yolox_s_8x8_mmyolo.py:30-69)

The nn.Modules below only HOLD parameters (named exactly like the reference state_dict, so
checkpoints load with load_state_dict) and validate config; they have no CPU forward — calling
them without a GPU raises.
"""
import torch
import torch.nn as nn

from .engine import HipDetector
from .registry import MODELS
from .structures import InstanceData


def _attach(root, dotted, tensor, is_buffer):
    """Create nested nn.Module containers for 'a.b.0.c' and register the leaf tensor."""
    parts = dotted.split('.')
    mod = root
    for p in parts[:-1]:
        if p not in mod._modules:
            mod.add_module(p, nn.Module())
        mod = mod._modules[p]
    if is_buffer:
        mod.register_buffer(parts[-1], tensor)
    else:
        mod.register_parameter(parts[-1], nn.Parameter(tensor, requires_grad=False))


class _ParamHolder(nn.Module):
    """Base of the config-only sub-modules: no forward on purpose."""

    def forward(self, *a, **k):
        raise RuntimeError(f'{type(self).__name__} has no stand-alone forward: the whole detector runs as one '
                           f'fused HIP launch plan (YOLODetector_Disparity_V1.predict / _forward)')


@MODELS.register_module(name=['YOLOXCSPDarknet_Disparity_V1_MMYOLO'])
class YOLOXCSPDarknet_Disparity_V1_MMYOLO(_ParamHolder):
    def __init__(self, out_fd=False, arch='P5', plugins=None, deepen_factor=1.0, widen_factor=1.0,
                 input_channels=3, out_indices=(2, 3, 4), frozen_stages=-1, use_depthwise=False,
                 spp_kernal_sizes=(5, 9, 13), norm_cfg=None, act_cfg=None, norm_eval=False, init_cfg=None):
        super().__init__()
        norm_cfg = norm_cfg or dict(type='BN', momentum=0.03, eps=0.001)
        act_cfg = act_cfg or dict(type='SiLU', inplace=True)
        unsupported = []
        if arch != 'P5': unsupported.append(f'arch={arch}')
        if plugins: unsupported.append('plugins')
        if use_depthwise: unsupported.append('use_depthwise=True')
        if tuple(spp_kernal_sizes) != (5, 9, 13): unsupported.append(f'spp_kernal_sizes={spp_kernal_sizes}')
        if tuple(out_indices) != (2, 3, 4): unsupported.append(f'out_indices={out_indices}')
        if input_channels != 3: unsupported.append(f'input_channels={input_channels}')
        if out_fd: unsupported.append('out_fd=True')
        if norm_cfg.get('type') not in ('BN', 'SyncBN'): unsupported.append(f'norm {norm_cfg}')
        if act_cfg.get('type') != 'SiLU': unsupported.append(f'act {act_cfg}')
        if unsupported:
            raise NotImplementedError('HIP two-branch CSPDarknet supports the shipped stereo config only; '
                                      'unsupported: ' + ', '.join(unsupported))
        self.deepen_factor, self.widen_factor = float(deepen_factor), float(widen_factor)
        self.bn_eps = float(norm_cfg.get('eps', 1e-5))


@MODELS.register_module(name=['CSPDarknet'])
class CSPDarknet(_ParamHolder):
    """`mmtrack.CSPDarknet` (reference csp_darknet.py:8-13): mmdet 3.0.0rc4's CSPDarknet whose forward takes the MOT
    shell's input dict and reads `x['img']` only - the backbone of the reference's RGB-only stereo configuration
    (yolox_s_mmyolo_mot_airdrone.py:42; kwargs merged in from _base_/yolox_s_8x8_mmyolo.py:20-28).  Same module tree and
    state_dict keys as the image branch of the two-branch class (stem, stage1..stage4; stage4 = conv, SPPBottleneck,
    CSPLayer): the HIP plan is the two-branch plan without disp_stem / disp_stage1 and without the average."""

    def __init__(self, arch='P5', deepen_factor=1.0, widen_factor=1.0, out_indices=(2, 3, 4), frozen_stages=-1,
                 use_depthwise=False, arch_ovewrite=None, spp_kernal_sizes=(5, 9, 13), conv_cfg=None, norm_cfg=None,
                 act_cfg=None, norm_eval=False, init_cfg=None):
        super().__init__()
        norm_cfg = norm_cfg or dict(type='BN', momentum=0.03, eps=0.001)
        act_cfg = act_cfg or dict(type='Swish')
        unsupported = []
        if arch != 'P5': unsupported.append(f'arch={arch}')
        if arch_ovewrite: unsupported.append('arch_ovewrite')
        if use_depthwise: unsupported.append('use_depthwise=True')
        if conv_cfg: unsupported.append(f'conv_cfg={conv_cfg}')
        if tuple(spp_kernal_sizes) != (5, 9, 13): unsupported.append(f'spp_kernal_sizes={spp_kernal_sizes}')
        if tuple(out_indices) != (2, 3, 4): unsupported.append(f'out_indices={out_indices}')
        if norm_cfg.get('type') not in ('BN', 'SyncBN'): unsupported.append(f'norm {norm_cfg}')
        if act_cfg.get('type') not in ('SiLU', 'Swish'): unsupported.append(f'act {act_cfg}')   # x * sigmoid(x) both
        if unsupported:
            raise NotImplementedError('HIP CSPDarknet supports the shipped stereo configs only; unsupported: '
                                      + ', '.join(unsupported))
        self.deepen_factor, self.widen_factor = float(deepen_factor), float(widen_factor)
        self.bn_eps = float(norm_cfg.get('eps', 1e-5))


@MODELS.register_module()
class YOLOXPAFPN(_ParamHolder):
    def __init__(self, in_channels=(256, 512, 1024), out_channels=256, deepen_factor=1.0, widen_factor=1.0,
                 num_csp_blocks=3, use_depthwise=False, freeze_all=False, norm_cfg=None, act_cfg=None,
                 init_cfg=None):
        super().__init__()
        if list(in_channels) != [256, 512, 1024] or out_channels != 256 or num_csp_blocks != 3 or use_depthwise:
            raise NotImplementedError('HIP YOLOXPAFPN supports in_channels=[256,512,1024], out_channels=256, '
                                      'num_csp_blocks=3, use_depthwise=False (scaled by widen/deepen factors)')
        self.deepen_factor, self.widen_factor = float(deepen_factor), float(widen_factor)


@MODELS.register_module()
class YOLOXHeadModule(_ParamHolder):
    def __init__(self, num_classes=80, in_channels=256, widen_factor=1.0, num_base_priors=1, feat_channels=256,
                 stacked_convs=2, featmap_strides=(8, 16, 32), use_depthwise=False, dcn_on_last_conv=False,
                 conv_bias='auto', conv_cfg=None, norm_cfg=None, act_cfg=None, init_cfg=None):
        super().__init__()
        if (in_channels != 256 or feat_channels != 256 or stacked_convs != 2 or
                tuple(featmap_strides) != (8, 16, 32) or use_depthwise or num_base_priors != 1):
            raise NotImplementedError('HIP YOLOXHeadModule supports in/feat_channels=256, stacked_convs=2, '
                                      'strides (8,16,32), no depthwise')
        self.num_classes, self.widen_factor = int(num_classes), float(widen_factor)
        self.featmap_strides = tuple(featmap_strides)


@MODELS.register_module()
class YOLOXHead(_ParamHolder):
    def __init__(self, head_module, prior_generator=None, bbox_coder=None, loss_cls=None, loss_bbox=None,
                 loss_obj=None, loss_bbox_aux=None, train_cfg=None, test_cfg=None, init_cfg=None):
        super().__init__()
        head_module = dict(head_module)
        head_module.setdefault('type', 'YOLOXHeadModule')
        self.head_module = MODELS.build(head_module)
        self.test_cfg = dict(test_cfg or {})
        self.num_classes = self.head_module.num_classes


@MODELS.register_module(name=['YOLODetector_Disparity_V1'])
class YOLODetector_Disparity_V1(nn.Module):
    """YOLOX detector over the HIP launch plan; same constructor / method surface as the reference class.  Which plan
    runs (two-branch, or the image branch alone) follows from the BACKBONE class, as in the reference; the subclass
    `YOLODetector` below is the detector type of the RGB-only config."""

    def __init__(self, backbone, neck, bbox_head, train_cfg=None, test_cfg=None, data_preprocessor=None,
                 init_cfg=None, use_syncbn=True):
        super().__init__()
        bbox_head = dict(bbox_head)
        bbox_head.setdefault('type', 'YOLOXHead')
        bbox_head['test_cfg'] = test_cfg
        self.backbone = MODELS.build(backbone)
        self.neck = MODELS.build(neck)
        self.bbox_head = MODELS.build(bbox_head)
        self.train_cfg, self.test_cfg, self.init_cfg = train_cfg, dict(test_cfg or {}), init_cfg
        w, d = self.backbone.widen_factor, self.backbone.deepen_factor
        if (self.neck.widen_factor, self.neck.deepen_factor) != (w, d) or self.bbox_head.head_module.widen_factor != w:
            raise ValueError('backbone / neck / head widen+deepen factors must agree')
        self.widen_factor, self.deepen_factor = w, d
        self.rgb_only = isinstance(self.backbone, CSPDarknet)
        self.num_classes = self.bbox_head.num_classes
        if not 1 <= self.num_classes <= 1024:
            raise ValueError('num_classes must be in [1, 1024]')
        # several classes: multi_label (the base config's value: every (prior, class) pair above score_thr is a
        # candidate) or multi_label=False (one candidate per prior: its best class) - both in the decode kernel
        self.multi_label = bool(self.test_cfg.get('multi_label', True))
        # parameter tree from the library's own table (graph shapes do not matter for the table)
        probe = HipDetector(1, 32, 32, w, d, self.num_classes, bn_eps=self.backbone.bn_eps, rgb_only=self.rgb_only)
        self._table = probe.param_table()
        del probe
        bn_prefixes = set()
        for name, shape in self._table:
            is_buf = name.endswith('running_mean') or name.endswith('running_var')
            init = torch.ones(shape) if name.endswith(('bn.weight', 'running_var')) else torch.zeros(shape)
            _attach(self, name, init, is_buf)
            if name.endswith('.bn.weight'):
                bn_prefixes.add(name[:-len('.weight')])
        for p in sorted(bn_prefixes):
            _attach(self, p + '.num_batches_tracked', torch.zeros((), dtype=torch.long), True)
        self._engines = {}
        self._uploaded = {}
        self.stereo = None  # a StereoCostVolume, set by the MOT shell when the config has `stereo=`

    # ---- plumbing -------------------------------------------------------------------------------
    @property
    def with_neck(self):
        return True

    def init_weights(self):
        """init_cfg = dict(type='ColorPretrained', checkpoint=<LOCAL file>): load a colour-image detector checkpoint and
        start the disparity branch (`disp_stem`, `disp_stage1`) from the RGB branch's weights, non-strict - reference
        yolo_detector_disparity_v1.py:144-166.  `type='Pretrained'`: the plain non-strict load.  The shipped config
        names a URL (:46), which cannot be fetched offline: that raises with the instruction to pass a local path;
        without init_cfg nothing happens (weights come from load_state_dict / load_checkpoint)."""
        cfg = self.init_cfg
        if not cfg:
            return None
        if isinstance(cfg, (list, tuple)):
            cfg = cfg[0]
        kind = cfg.get('type')
        if kind not in ('ColorPretrained', 'Pretrained'):
            return None
        from .checkpoint import _state_dict_of, color_pretrained_state_dict, load_matching, read_checkpoint
        sd = dict(_state_dict_of(read_checkpoint(cfg.get('checkpoint'), trusted=bool(cfg.get('trusted', False)))))
        if kind == 'ColorPretrained':
            sd = color_pretrained_state_dict(sd)
        self.init_report = load_matching(self, sd, strict=False)
        return self.init_report

    def _weights_version(self):
        return tuple(t._version for t in self.state_dict(keep_vars=True).values())

    def _engine(self, N, H, W, stereo=False):
        key = (N, H, W, bool(stereo))
        eng = self._engines.get(key)
        if eng is None:
            eng = HipDetector(N, H, W, self.widen_factor, self.deepen_factor, self.num_classes,
                              bn_eps=self.backbone.bn_eps, stereo=stereo, rgb_only=self.rgb_only)
            eng.multi_label = self.multi_label
            self._engines[key] = eng
        ver = self._weights_version()
        if self._uploaded.get(key) != ver:
            eng.load_state_dict(self.state_dict())
            eng.autotune()
            self._uploaded[key] = ver
        return eng

    def _run(self, batch_inputs, valid_hw=None):
        """One dense forward.  batch_inputs: {'img', 'disp_postp'} as in the reference, or
        {'img', 'right'} when a stereo module is attached: then disp_postp is computed on the GPU and
        written back into batch_inputs['disp_postp'] for the depth step that follows."""
        if not isinstance(batch_inputs, dict) or 'img' not in batch_inputs:
            raise TypeError("batch_inputs must be a dict with 'img' and 'disp_postp' (or 'right') (N,3,H,W) tensors")
        img = batch_inputs['img']
        if not img.is_cuda:
            raise RuntimeError('YOLODetector_Disparity_V1 runs on the HIP path only: inputs must be CUDA tensors')
        img = img.float().contiguous()
        N, _, H, W = img.shape
        if batch_inputs.get('disp_postp') is None:
            if self.stereo is None or batch_inputs.get('right') is None:
                if self.rgb_only:   # the single-branch detector needs the image alone
                    eng = self._engine(N, H, W)
                    return eng, eng.forward(img, None)
                raise TypeError("need 'disp_postp', or 'right' plus a StereoCostVolume module")
            eng = self._engine(N, H, W, stereo=True)
            disp = self.stereo.compute(eng, img, batch_inputs['right'].float().contiguous(), valid_hw or (H, W))
            batch_inputs['disp_postp'] = disp
            return eng, eng.forward_phase(1, disp=disp)
        eng = self._engine(N, H, W)
        if self.rgb_only:    # csp_darknet.py:11: the backbone reads x['img']; disp_postp is for the depth step only
            return eng, eng.forward(img, None)
        return eng, eng.forward(img, batch_inputs['disp_postp'].float().contiguous())

    # ---- reference API -----------------------------------------------------------------------------
    def extract_feat(self, batch_inputs):
        """-> tuple of the 3 neck outputs (N,C,h,w) (reference :77-90)."""
        eng, _ = self._run(batch_inputs)
        return tuple(eng.tap(n).permute(0, 3, 1, 2) for n in ('p3', 'p4', 'p5'))

    def _forward(self, batch_inputs, batch_data_samples=None):
        """-> (cls_scores, bbox_preds, objectnesses), lists of NCHW tensors (reference :127-142)."""
        eng, head = self._run(batch_inputs)
        return eng.head_nchw(head)

    def predict(self, batch_inputs, batch_data_samples, rescale=True):
        """Reference :92-125.  Adds `pred_instances` (bboxes, scores, labels) to every data sample."""
        ori0 = batch_data_samples[0].metainfo.get('ori_shape') if batch_data_samples else None
        eng, head = self._run(batch_inputs, ori0[:2] if ori0 is not None else None)
        N = eng.batch
        if len(batch_data_samples) != N:
            raise ValueError(f'{len(batch_data_samples)} data samples for a batch of {N}')
        metas = [s.metainfo for s in batch_data_samples]
        ori = metas[0].get('ori_shape', (eng.height, eng.width))[:2]
        sf = metas[0].get('scale_factor', (1.0, 1.0)) if rescale else (1.0, 1.0)
        pad = metas[0].get('pad_param', None) if rescale else None
        def _pad(v):   # compare by VALUE: every data sample carries its own (equal) pad_param array
            return None if v is None else tuple(float(x) for x in v)
        for m in metas[1:]:
            if (tuple(m.get('ori_shape', ori)[:2]) != tuple(ori) or
                    tuple(m.get('scale_factor', sf)) != tuple(sf) or
                    (rescale and _pad(m.get('pad_param', None)) != _pad(pad))):
                raise NotImplementedError('one HIP decode launch needs uniform ori_shape/scale_factor/pad_param')
        cfg = self.test_cfg
        nms = cfg.get('nms', dict(type='nms', iou_threshold=0.65))
        if nms.get('type', 'nms') != 'nms':
            raise NotImplementedError(f"nms type {nms.get('type')} (only greedy 'nms')")
        cap = int(cfg.get('max_det_capacity', 1000))
        boxes, scores, labels, prior, counts = eng.decode_nms(head, cfg.get('score_thr', 0.01),
                                                              nms.get('iou_threshold', 0.65), cap, ori, sf, pad)
        counts_h = counts.cpu().tolist()  # the API returns per-image variable-length results
        if not cfg.get('yolox_style', False):
            counts_h = [min(c, int(cfg.get('max_per_img', 300))) for c in counts_h]
        for n, sample in enumerate(batch_data_samples):
            k = counts_h[n]
            if k > cap:
                raise RuntimeError(f'{k} detections exceed max_det_capacity={cap}; raise test_cfg.max_det_capacity')
            sample.pred_instances = InstanceData(bboxes=boxes[n, :k], scores=scores[n, :k], labels=labels[n, :k],
                                                 prior_idx=prior[n, :k])
        return batch_data_samples

    def loss(self, *a, **k):
        raise NotImplementedError('training is out of scope of the HIP hot path (SURVEY.md §3.3)')

    def forward(self, inputs, data_samples=None, mode='tensor'):
        if mode == 'predict':
            return self.predict(inputs, data_samples)
        if mode == 'tensor':
            return self._forward(inputs, data_samples)
        raise NotImplementedError(f'mode={mode}')


@MODELS.register_module(name=['YOLODetector'])
class YOLODetector(YOLODetector_Disparity_V1):
    """`mmyolo.YOLODetector` as configs/stereo_tracking/ocsort/yolox_s_mmyolo_mot_airdrone.py:40-42 uses it (type from
    configs/_base_/yolox_s_8x8_mmyolo.py:19): the single-stage detector whose backbone `mmtrack.CSPDarknet` picks
    `x['img']` out of the MOT shell's input dict (csp_darknet.py:8-13).  Same HIP plan machinery as the two-branch class;
    with the two-branch backbone configured it behaves exactly like `YOLODetector_Disparity_V1`."""
