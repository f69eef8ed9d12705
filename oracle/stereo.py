"""TEST INFRASTRUCTURE — CPU oracle of the stereo module (cost volume -> 2-D aggregation -> soft-argmin ->
x4 upsample).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.

Parity status: "parity unpinned" — the reference ships NO stereo matcher (its disparity is an offline OpenCV
SGBM product, reproducibility.md:166-194), so there is no reference output to pin against; the module is
specified by BASELINE.json's north_star and frozen in stereotracking_amd/stereo.py's docstring.  What IS
pinned is the consumer contract (disp_postp layout / units / zero padding: reference
mmtrack/datasets/transforms/loading_disparity.py:85-86,129-134 and transforms_disparity.py:234-249).

Pieces: cost volume, 3-D aggregation, soft-argmin and upsample are the plain-C loops of oracle/st_oracle.c (same fmaf order as
the kernels, so those stages are compared bit-exactly); the aggregation convs are torch fp32 conv2d on CPU
(floating-point kernel => torch fp32 reference, tolerance 1e-3 as north_star states).
"""
import numpy as np
import torch
import torch.nn.functional as F

from . import c_oracle


def aggregate(cost, sd, agg_layers, prefix='stereo.'):
    """cost: float32 numpy (N,Hf,Wf,D) -> aggregated volume, same layout.
    cost <- conv3x3(cost; agg.l.weight, agg.l.bias) over d-as-channels, SiLU after all but the last layer."""
    if agg_layers == 0:
        return cost
    x = torch.from_numpy(np.ascontiguousarray(cost)).permute(0, 3, 1, 2).contiguous()
    with torch.no_grad():
        for l in range(agg_layers):
            x = F.conv2d(x, sd[f'{prefix}agg.{l}.weight'].float(), sd[f'{prefix}agg.{l}.bias'].float(), padding=1)
            if l < agg_layers - 1:
                x = F.silu(x)
    return np.ascontiguousarray(x.permute(0, 2, 3, 1).numpy())


def aggregate3d(cost, sd, agg3d_layers, prefix='stereo.'):
    """`agg3d_layers` single-channel 3x3x3 convolutions over (d, y, x), zero padded, SiLU after all but the last
    (parameters agg3d.{l}.weight (1,1,3,3,3), agg3d.{l}.bias (1,)): the plain-C loops of oracle_agg3d, bit-exact spec."""
    for l in range(agg3d_layers):
        w = sd[f'{prefix}agg3d.{l}.weight'].float().reshape(3, 3, 3).numpy()
        b = float(sd[f'{prefix}agg3d.{l}.bias'].float().reshape(-1)[0])
        cost = c_oracle.agg3d(cost, w, b, l < agg3d_layers - 1)
    return cost


def disparity(featL, featR, C_, D, temperature, sd=None, agg_layers=0, scale=4, valid_hw=None, prefix='stereo.',
              agg3d_layers=0):
    """featL/featR: float32 numpy (N,Hf,Wf,ld) stage-1 features.  cost volume -> 3-D aggregation -> 2-D aggregation ->
    soft-argmin -> upsample.  Returns (cost, disp_lr, disp_postp[N,3,Hf*scale,Wf*scale]) — zero outside valid_hw."""
    cost = aggregate3d(c_oracle.costvolume(featL, featR, C_, D), sd, agg3d_layers, prefix)
    cost = aggregate(cost, sd, agg_layers, prefix)
    lr = c_oracle.softargmin(cost, temperature)
    vh, vw = valid_hw if valid_hw is not None else (featL.shape[1] * scale, featL.shape[2] * scale)
    return cost, lr, c_oracle.disp_upsample(lr, scale, vh, vw)


def reduce_features(feat, C_, sd, prefix='stereo.'):
    """G = reduce(F): the 1x1 convolution C_ -> Cr (bias, no activation) of the full-resolution mode, float32, one
    pixel at a time as a matrix product (stereotracking_amd/stereo.py spec; parameters reduce.weight (Cr,C_,1,1))."""
    import numpy as np
    w = sd[f'{prefix}reduce.weight'].float().reshape(-1, C_).numpy()
    b = sd[f'{prefix}reduce.bias'].float().reshape(-1).numpy()
    f = np.ascontiguousarray(feat[..., :C_], np.float32)
    return (f.reshape(-1, C_) @ w.T + b).astype(np.float32).reshape(*f.shape[:-1], w.shape[0])


def disparity_fullres(featL, featR, C_, D, temperature, sd, agg3d_layers=0, scale=4, valid_hw=None, prefix='stereo.',
                      upsampled=None):
    """The stereo module's FULL-RESOLUTION mode (spec in stereotracking_amd/stereo.py): reduce (1x1, C_ -> Cr) ->
    bilinear x`scale` -> cost volume of D = max_disp levels at image resolution -> `agg3d_layers` 3x3x3 layers ->
    soft-argmin (pixels) -> 0 outside valid_hw, three identical channels.  `upsampled` = (G_L, G_R) (N,H,W,Cr): start
    from given image-resolution features (the parity tests feed the GPU's own, so that everything after the float
    matrix product is compared BIT FOR BIT).  Returns (volume, disparity (N,H,W), disp_postp (N,3,H,W))."""
    if upsampled is None:
        gl = c_oracle.feat_upsample(reduce_features(featL, C_, sd, prefix), scale)
        gr = c_oracle.feat_upsample(reduce_features(featR, C_, sd, prefix), scale)
    else:
        gl, gr = upsampled
    cost = aggregate3d(c_oracle.costvolume(gl, gr, gl.shape[-1], D), sd, agg3d_layers, prefix)
    disp = c_oracle.softargmin(cost, temperature)
    vh, vw = valid_hw if valid_hw is not None else (disp.shape[1], disp.shape[2])
    return cost, disp, c_oracle.disp_upsample(disp, 1, vh, vw)
