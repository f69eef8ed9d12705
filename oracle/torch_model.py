"""ORACLE (test infrastructure, not product code): CPU PyTorch restatement of the reference's
two-branch YOLOX detector.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
may import this module.

PARITY PINNING: the reference has no tests, golden vectors or fixtures for this path and cannot
be imported here (mmcv/mmdet/mmyolo/mmengine absent -> ordinary ModuleNotFoundError, SURVEY.md
§8c), and most of the arithmetic lives in un-vendored mmdet 3.0.0rc4 / mmyolo 0.2.0 / mmcv
2.0.0rc3 modules.  Those parts are restated from their published definitions and are therefore
"parity unpinned" at the third-party boundary; what IS pinned is the structure the reference's
own files define (module tree, channel arithmetic, forward order), each cited below.

Every module keeps the reference's attribute names so `state_dict()` keys equal the reference
checkpoint layout (`backbone.stem.conv.conv.weight`, `neck.top_down_layers.0.1.bn.running_var`,
`bbox_head.head_module.multi_level_conv_obj.2.bias`, ...).
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F


def make_divisible(x, widen_factor, divisor=8):
    """mmyolo.models.utils.make_divisible (used at csp_darknet_disparity_v1.py:108,120-121)."""
    return math.ceil(x * widen_factor / divisor) * divisor


def make_round(x, deepen_factor):
    """mmyolo.models.utils.make_round (used at csp_darknet_disparity_v1.py:122)."""
    return max(round(x * deepen_factor), 1) if x > 1 else x


class ConvModule(nn.Module):
    """mmcv ConvModule with norm_cfg=BN(eps=1e-3, momentum=0.03), act_cfg=SiLU
    (csp_darknet_disparity_v1.py:82-84,126-135): conv(bias=False) -> bn -> SiLU."""

    def __init__(self, cin, cout, k, stride=1, padding=0):
        super().__init__()
        self.conv = nn.Conv2d(cin, cout, k, stride, padding, bias=False)
        self.bn = nn.BatchNorm2d(cout, eps=1e-3, momentum=0.03)

    def forward(self, x):
        return F.silu(self.bn(self.conv(x)))


class Focus(nn.Module):
    """mmdet Focus (built at csp_darknet_disparity_v1.py:104-111): space-to-depth in the order
    top-left, bottom-left, top-right, bottom-right, then ConvModule(4c, cout, 3, 1, 1)."""

    def __init__(self, cin, cout, k=3):
        super().__init__()
        self.conv = ConvModule(cin * 4, cout, k, 1, (k - 1) // 2)

    def forward(self, x):
        tl = x[..., ::2, ::2]
        tr = x[..., ::2, 1::2]
        bl = x[..., 1::2, ::2]
        br = x[..., 1::2, 1::2]
        return self.conv(torch.cat((tl, bl, tr, br), dim=1))


class DarknetBottleneck(nn.Module):
    """mmdet DarknetBottleneck, expansion 1.0 inside CSPLayer: 1x1 -> 3x3 (+ identity)."""

    def __init__(self, c, add_identity):
        super().__init__()
        self.conv1 = ConvModule(c, c, 1)
        self.conv2 = ConvModule(c, c, 3, 1, 1)
        self.add_identity = add_identity

    def forward(self, x):
        out = self.conv2(self.conv1(x))
        return out + x if self.add_identity else out


class CSPLayer(nn.Module):
    """mmdet CSPLayer (built at csp_darknet_disparity_v1.py:145-152), expand_ratio 0.5."""

    def __init__(self, cin, cout, num_blocks, add_identity):
        super().__init__()
        mid = int(cout * 0.5)
        self.main_conv = ConvModule(cin, mid, 1)
        self.short_conv = ConvModule(cin, mid, 1)
        self.final_conv = ConvModule(2 * mid, cout, 1)
        self.blocks = nn.Sequential(*[DarknetBottleneck(mid, add_identity) for _ in range(num_blocks)])

    def forward(self, x):
        x_short = self.short_conv(x)
        x_main = self.blocks(self.main_conv(x))
        return self.final_conv(torch.cat((x_main, x_short), dim=1))


class SPPFBottleneck(nn.Module):
    """mmyolo SPPFBottleneck with a tuple of kernel sizes => parallel SPP
    (built at csp_darknet_disparity_v1.py:137-144)."""

    def __init__(self, cin, cout, kernel_sizes=(5, 9, 13)):
        super().__init__()
        mid = cin // 2
        self.conv1 = ConvModule(cin, mid, 1)
        self.poolings = nn.ModuleList([nn.MaxPool2d(k, 1, k // 2) for k in kernel_sizes])
        self.conv2 = ConvModule(mid * (len(kernel_sizes) + 1), cout, 1)

    def forward(self, x):
        x = self.conv1(x)
        x = torch.cat([x] + [p(x) for p in self.poolings], dim=1)
        return self.conv2(x)


class YOLOXCSPDarknet_Disparity_V1(nn.Module):
    """Reference backbone mmtrack/models/backbones/csp_darknet_disparity_v1.py:17-206."""

    arch = [[64, 128, 3, True, False], [128, 256, 9, True, False], [256, 512, 9, True, False],
            [512, 1024, 3, False, True]]  # :66-69

    def __init__(self, deepen_factor=0.33, widen_factor=0.5, input_channels=3, out_indices=(2, 3, 4)):
        super().__init__()
        self.out_indices = out_indices
        w, d = widen_factor, deepen_factor
        self.stem = Focus(input_channels, make_divisible(64, w))  # :104-111
        for idx, setting in enumerate(self.arch):  # base_backbone_disparity_mmyolo.py:112-118
            self.add_module(f'stage{idx + 1}', nn.Sequential(*self._stage(setting, w, d)))
        self.disp_stem = Focus(input_channels, make_divisible(64, w))  # :93
        self.disp_stage1 = nn.Sequential(*self._stage(self.arch[0], w, d))  # :96-101

    @staticmethod
    def _stage(setting, w, d):  # :113-153
        cin, cout, n, add_identity, use_spp = setting
        cin, cout, n = make_divisible(cin, w), make_divisible(cout, w), make_round(n, d)
        stage = [ConvModule(cin, cout, 3, 2, 1)]
        if use_spp:
            stage.append(SPPFBottleneck(cout, cout, (5, 9, 13)))
        stage.append(CSPLayer(cout, cout, n, add_identity))
        return stage

    def forward(self, x):  # :155-206
        o_stem = self.stage1(self.stem(x['img']))
        o_disp = self.disp_stage1(self.disp_stem(x['disp_postp']))
        y = (o_stem + o_disp) / 2.
        outs = []
        if 1 in self.out_indices:
            outs.append(y)
        for i in (2, 3, 4):
            y = getattr(self, f'stage{i}')(y)
            if i in self.out_indices:
                outs.append(y)
        return tuple(outs)

    def stage1_features(self, img):
        """stem+stage1 of the RGB branch (the stereo module's feature extractor)."""
        return self.stage1(self.stem(img))


class CSPDarknetRGB(nn.Module):
    """Reference `mmtrack.CSPDarknet` (mmtrack/models/backbones/csp_darknet.py:8-13): mmdet 3.0.0rc4's CSPDarknet with
    `forward(x) = super().forward(x['img'])` - the backbone of the RGB-only stereo config
    (configs/stereo_tracking/ocsort/yolox_s_mmyolo_mot_airdrone.py:42).  mmdet's class [upstream-memory] builds
    `stem` = Focus(3, 64w, 3) and `stage1..4` = [ConvModule(3x3, s2), (SPPBottleneck in stage4), CSPLayer] from the same
    P5 arch table as the two-branch class (:66-69 there), i.e. the image branch of it with the same state_dict keys."""

    arch = YOLOXCSPDarknet_Disparity_V1.arch

    def __init__(self, deepen_factor=0.33, widen_factor=0.5, out_indices=(2, 3, 4)):
        super().__init__()
        self.out_indices = out_indices
        self.stem = Focus(3, make_divisible(64, widen_factor))
        for idx, setting in enumerate(self.arch):
            self.add_module(f'stage{idx + 1}',
                            nn.Sequential(*YOLOXCSPDarknet_Disparity_V1._stage(setting, widen_factor, deepen_factor)))

    def forward(self, x):
        y = self.stem(x['img'])   # csp_darknet.py:11
        outs = []
        for i in (1, 2, 3, 4):
            y = getattr(self, f'stage{i}')(y)
            if i in self.out_indices:
                outs.append(y)
        return tuple(outs)

    def stage1_features(self, img):
        return self.stage1(self.stem(img))


class YOLOXPAFPN(nn.Module):
    """mmyolo 0.2.0 YOLOXPAFPN on BaseYOLONeck.forward (configs/_base_/yolox_s_8x8_mmyolo.py:30-37)."""

    def __init__(self, deepen_factor=0.33, widen_factor=0.5, in_channels=(256, 512, 1024), out_channels=256):
        super().__init__()
        ic = [make_divisible(c, widen_factor) for c in in_channels]
        oc = make_divisible(out_channels, widen_factor)
        n = make_round(3, deepen_factor)
        self.reduce_layers = nn.ModuleList([nn.Identity(), nn.Identity(), ConvModule(ic[2], ic[1], 1)])
        self.upsample_layers = nn.ModuleList([nn.Upsample(scale_factor=2, mode='nearest') for _ in range(2)])
        self.top_down_layers = nn.ModuleList([
            nn.Sequential(CSPLayer(ic[1] * 2, ic[1], n, False), ConvModule(ic[1], ic[0], 1)),  # idx 2
            CSPLayer(ic[0] * 2, ic[0], n, False),  # idx 1
        ])
        self.downsample_layers = nn.ModuleList([ConvModule(ic[i], ic[i], 3, 2, 1) for i in range(2)])
        self.bottom_up_layers = nn.ModuleList([CSPLayer(ic[i] * 2, ic[i + 1], n, False) for i in range(2)])
        self.out_layers = nn.ModuleList([ConvModule(ic[i], oc, 1) for i in range(3)])

    def forward(self, inputs):
        reduce_outs = [self.reduce_layers[i](inputs[i]) for i in range(3)]
        inner_outs = [reduce_outs[2]]
        for idx in range(2, 0, -1):
            feat_high, feat_low = inner_outs[0], reduce_outs[idx - 1]
            up = self.upsample_layers[2 - idx](feat_high)
            inner_outs.insert(0, self.top_down_layers[2 - idx](torch.cat([up, feat_low], 1)))
        outs = [inner_outs[0]]
        for idx in range(2):
            down = self.downsample_layers[idx](outs[-1])
            outs.append(self.bottom_up_layers[idx](torch.cat([down, inner_outs[idx + 1]], 1)))
        return tuple(self.out_layers[i](outs[i]) for i in range(3))


class YOLOXHeadModule(nn.Module):
    """mmyolo 0.2.0 YOLOXHeadModule (configs/_base_/yolox_s_8x8_mmyolo.py:40-51)."""

    def __init__(self, num_classes=1, in_channels=256, widen_factor=0.5, feat_channels=256, stacked_convs=2,
                 featmap_strides=(8, 16, 32)):
        super().__init__()
        cin, feat = int(in_channels * widen_factor), int(feat_channels * widen_factor)
        self.featmap_strides = featmap_strides

        def tower():
            return nn.Sequential(*[ConvModule(cin if i == 0 else feat, feat, 3, 1, 1) for i in range(stacked_convs)])

        self.multi_level_cls_convs = nn.ModuleList([tower() for _ in featmap_strides])
        self.multi_level_reg_convs = nn.ModuleList([tower() for _ in featmap_strides])
        self.multi_level_conv_cls = nn.ModuleList([nn.Conv2d(feat, num_classes, 1) for _ in featmap_strides])
        self.multi_level_conv_reg = nn.ModuleList([nn.Conv2d(feat, 4, 1) for _ in featmap_strides])
        self.multi_level_conv_obj = nn.ModuleList([nn.Conv2d(feat, 1, 1) for _ in featmap_strides])

    def forward(self, feats):
        cls, reg, obj = [], [], []
        for i, x in enumerate(feats):
            cf = self.multi_level_cls_convs[i](x)
            rf = self.multi_level_reg_convs[i](x)
            cls.append(self.multi_level_conv_cls[i](cf))
            reg.append(self.multi_level_conv_reg[i](rf))
            obj.append(self.multi_level_conv_obj[i](rf))
        return cls, reg, obj


class _Head(nn.Module):
    def __init__(self, **kw):
        super().__init__()
        self.head_module = YOLOXHeadModule(**kw)


class OracleDetector(nn.Module):
    """YOLODetector_Disparity_V1._forward, mmtrack/models/detectors/yolo_detector_disparity_v1.py:127-142."""

    def __init__(self, deepen_factor=0.33, widen_factor=0.5, num_classes=1, rgb_only=False):
        """rgb_only: the reference's second stereo config - detector `mmyolo.YOLODetector` over `mmtrack.CSPDarknet`
        (yolox_s_mmyolo_mot_airdrone.py:40-42): same neck / head, single-branch backbone."""
        super().__init__()
        self.backbone = (CSPDarknetRGB if rgb_only else YOLOXCSPDarknet_Disparity_V1)(deepen_factor, widen_factor)
        self.neck = YOLOXPAFPN(deepen_factor, widen_factor)
        self.bbox_head = _Head(num_classes=num_classes, widen_factor=widen_factor)

    def extract_feat(self, inputs):  # :77-90
        return self.neck(self.backbone(inputs))

    def forward(self, inputs):  # :127-142
        return self.bbox_head.head_module(self.extract_feat(inputs))


def head_to_rows(cls, reg, obj):
    """[(N,nc,h,w)], [(N,4,h,w)], [(N,1,h,w)] -> per level (N, h*w, nc+5) rows = the product's
    head_out layout (cls | reg | obj), flatten order permute(0,2,3,1) as in predict_by_feat."""
    rows = []
    for c, r, o in zip(cls, reg, obj):
        n = c.shape[0]
        rows.append(torch.cat([c, r, o], dim=1).permute(0, 2, 3, 1).reshape(n, -1, c.shape[1] + 5))
    return rows
