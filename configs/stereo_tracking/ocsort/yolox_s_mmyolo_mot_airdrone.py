# RGB-only YOLOX-s + depth-guided OC-SORT on AirDrone, inference: the detector sees the image alone (backbone
# `mmtrack.CSPDarknet`, mmtrack/models/backbones/csp_darknet.py:8-13), the loaded disparity is consumed by the MOT shell
# for the per-box depth only (ocsort_disparity.py:82-83).
# Same model dict (types, kwargs, thresholds) as the reference config of the same name,
# configs/stereo_tracking/ocsort/yolox_s_mmyolo_mot_airdrone.py:29-58; dataset / training sections are out of scope
# of this repo's hot path and omitted.
_base_ = ['../../_base_/default_runtime.py', '../../_base_/yolox_s_8x8_mmyolo.py']

data_root = 'data/AirSim_drone/'
DEPTH_RANGE = 80
img_scale = (720, 1280)
num_classes = 1
classes = ['drone']

model = dict(
    type='OCSORT_Disparity',
    data_preprocessor=dict(type='TrackDataPreprocessor_Disparity_V1', pad_size_divisor=32, batch_augments=[]),
    detector=dict(
        _scope_='mmyolo',
        backbone=dict(type='mmtrack.CSPDarknet'),
        bbox_head=dict(head_module=dict(num_classes=num_classes)),
        test_cfg=dict(score_thr=0.01, nms=dict(type='nms', iou_threshold=0.5))),
    motion=dict(type='KalmanFilter'),
    tracker=dict(
        type='OCSORTTracker_Disparity',
        obj_score_thr=0.3,
        init_track_thr=0.7,
        weight_iou_with_det_scores=False,
        match_iou_thr=0.1,
        num_tentatives=3,
        vel_consist_weight=0.2,
        vel_delta_t=3,
        num_frames_retain=30))

# test-time input contract: as yolox_s_mmyolo_mot_airdrone_disp.py (img + disp_postp + disp_mask per frame); with the
# stereo module enabled (`stereo=dict(type='StereoCostVolume', ...)` in the model dict) the right image replaces
# disp_postp and the computed disparity feeds the depth step.
