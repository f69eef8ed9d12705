# Two-branch (RGB + disparity) YOLOX-s + depth-guided OC-SORT on AirDrone, inference.
# Same model dict (types, kwargs, thresholds) as the reference config of the same name,
# configs/stereo_tracking/ocsort/yolox_s_mmyolo_mot_airdrone_disp.py:29-58; dataset / training
# sections are out of scope of this repo's hot path and omitted.
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
        type='mmtrack.YOLODetector_Disparity_V1',
        backbone=dict(type='mmtrack.YOLOXCSPDarknet_Disparity_V1_MMYOLO', input_channels=3),
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

# test-time input contract (what the dataset pipeline must deliver per frame):
#   img        uint8/float (1,3,720,1280) BGR 0..255, bottom-padded to 736 rows with 114
#   disp_postp float32     (1,3,736,1280) disparity px x3 channels, 0 = invalid / padding
#   disp_mask  uint8       (1,1,736,1280)
# With the stereo module enabled the right image replaces disp_postp (see stereo_*.py).
