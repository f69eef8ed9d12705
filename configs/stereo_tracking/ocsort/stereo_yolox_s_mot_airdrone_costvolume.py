# Same tracker + detector, but disparity is COMPUTED on the GPU from the left/right pair by the
# StereoCostVolume module (the module BASELINE.json's north_star adds; the reference loads
# pre-computed SGBM disparity PNGs instead, reproducibility.md:166-194).
_base_ = ['./yolox_s_mmyolo_mot_airdrone_disp.py']

model = dict(
    stereo=dict(
        type='StereoCostVolume',
        max_disp=192,        # full-resolution disparity range
        feat_stride=4,       # correlate stage1 features: D' = 48 levels at 184x320
        temperature=32.0,    # soft-argmin sharpness
        agg_layers=2))       # 3x3 aggregation convs over the D' x H/4 x W/4 volume (d as channels)
