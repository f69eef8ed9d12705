# YOLOX-s detector defaults (inference part only).  Key names and nesting follow the reference's
# configs/_base_/yolox_s_8x8_mmyolo.py so child configs override them with mmengine merge rules.
img_scale = (640, 640)  # height, width
deepen_factor = 0.33
widen_factor = 0.5

_norm = dict(type='BN', momentum=0.03, eps=0.001)
_act = dict(type='SiLU', inplace=True)

model = dict(
    detector=dict(
        _scope_='mmyolo',
        type='YOLODetector',
        backbone=dict(
            type='YOLOXCSPDarknet',
            deepen_factor=deepen_factor,
            widen_factor=widen_factor,
            out_indices=(2, 3, 4),
            spp_kernal_sizes=(5, 9, 13),
            norm_cfg=_norm,
            act_cfg=_act),
        neck=dict(
            type='YOLOXPAFPN',
            deepen_factor=deepen_factor,
            widen_factor=widen_factor,
            in_channels=[256, 512, 1024],
            out_channels=256,
            norm_cfg=_norm,
            act_cfg=_act),
        bbox_head=dict(
            type='YOLOXHead',
            head_module=dict(
                type='YOLOXHeadModule',
                num_classes=80,
                in_channels=256,
                feat_channels=256,
                widen_factor=widen_factor,
                stacked_convs=2,
                featmap_strides=(8, 16, 32),
                use_depthwise=False,
                norm_cfg=_norm,
                act_cfg=_act)),
        test_cfg=dict(
            yolox_style=True,
            multi_label=True,
            score_thr=0.001,
            max_per_img=300,
            nms=dict(type='nms', iou_threshold=0.65))))
