# The stereo module in its FULL-RESOLUTION mode: north_star's literal sizing - a D = 192 level cost volume at image
# resolution (192 x 736 x 1280 cells per pair), one 3x3x3 aggregation layer over (d, y, x), soft-argmin in pixels
# (stereotracking_amd/stereo.py).  ~3 x slower than stereo_yolox_s_mot_airdrone_costvolume.py (48 levels at 1/4
# resolution = the same 192 px range); same detector, tracker and consumer contract.
_base_ = ['./yolox_s_mmyolo_mot_airdrone_disp.py']

model = dict(
    stereo=dict(
        type='StereoCostVolume',
        max_disp=192,          # one level per pixel of disparity
        feat_stride=4,         # stage-1 features, reduced to `full_res_channels` and brought to image resolution
        temperature=32.0,
        agg_layers=0,          # no 2-D stage at image resolution
        agg3d_layers=1,
        full_res=True,
        full_res_channels=8))
