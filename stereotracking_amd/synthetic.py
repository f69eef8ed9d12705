"""Seeded synthetic weights and stereo inputs (there are no checkpoints or datasets offline).

Weights follow SURVEY.md §8(d): conv ~ N(0, g^2/fan_in), BN gamma/beta/running stats randomised
(never identity, so BN-folding bugs show), head obj/cls bias = -log((1-p)/p).  The gains are
chosen so activations stay O(1) through ~60 layers with *fixed* running statistics: the two
stems are calibrated analytically against the input statistics (un-normalised 0..255 BGR and
disparity in pixels, reference data_preprocessor_disparity_v1.py:21-84 applies no mean/std), and
residual bottleneck branches get a small gamma as in trained residual nets.

Inputs follow the same section: a random textured left image, a piecewise-constant disparity
field, and the right image obtained by shifting the left by that disparity, so the cost-volume
module has a known answer.
"""
import math
from collections import OrderedDict

import numpy as np
import torch

# variance of silu(x), x ~ N(0,1)  (E[silu^2] - E[silu]^2 = 0.3557 - 0.2066^2)
_SILU_VAR = 0.313


def synthetic_state_dict(param_table, seed=0, img_stats=(127.5, 73.9), disp_stats=(24.0, 20.0),
                         prior_prob=0.01, logit_std=0.6):
    """param_table: iterable of (name, shape) in reference state_dict naming.
    Returns an OrderedDict name -> float32 tensor (BN `num_batches_tracked` not included)."""
    import zlib
    g = torch.Generator()
    table = [(n, tuple(int(s) for s in shp)) for n, shp in param_table]
    shapes = dict(table)
    sd = OrderedDict()

    def randn(shape, std=1.0, mean=0.0):
        return torch.randn(shape, generator=g, dtype=torch.float32) * std + mean

    def rand(shape, lo, hi):
        return torch.rand(shape, generator=g, dtype=torch.float32) * (hi - lo) + lo

    def zero_sum(w):
        return w - w.mean(dim=(1, 2, 3), keepdim=True)

    for name, shape in table:
        # one RNG stream per tensor group, keyed by NAME: the result does not depend on table order
        g.manual_seed((zlib.crc32(name.encode()) + 1000003 * int(seed)) & 0x7FFFFFFF)
        if name.endswith('.conv.weight') and len(shape) == 4:  # ConvModule conv
            fan_in = shape[1] * shape[2] * shape[3]
            prefix = name[:-len('.conv.weight')]
            is_stem = prefix.endswith('stem.conv')
            gain = 1.0 if is_stem else math.sqrt(1.0 / _SILU_VAR)
            w = randn(shape, gain / math.sqrt(fan_in))
            # zero-sum filters: the (always positive) mean of SiLU activations does not leak into
            # the next layer's pre-activation, so fixed running statistics stay representative
            w = w - w.mean(dim=(1, 2, 3), keepdim=True)
            sd[name] = w
            cout = shape[0]
            residual_branch = '.blocks.' in prefix and prefix.endswith('.conv2')
            gamma = rand((cout,), 0.2, 0.6) if residual_branch else rand((cout,), 0.5, 1.35)
            beta = randn((cout,), 0.1)
            if is_stem:
                mean_in, std_in = disp_stats if 'disp_stem' in prefix else img_stats
                wsum = w.sum(dim=(1, 2, 3))
                wsq = (w * w).sum(dim=(1, 2, 3))
                rmean = mean_in * wsum + randn((cout,), 0.1 * std_in)
                rvar = (std_in ** 2) * wsq * rand((cout,), 0.5, 1.5)
            else:
                rmean = randn((cout,), 0.1)
                rvar = rand((cout,), 0.5, 1.5)
            sd[prefix + '.bn.weight'] = gamma
            sd[prefix + '.bn.bias'] = beta
            sd[prefix + '.bn.running_mean'] = rmean
            sd[prefix + '.bn.running_var'] = rvar
        elif '.agg.' in '.' + name and name.endswith('.weight') and len(shape) == 4:
            # cost-volume aggregation conv: centre tap passes each disparity plane through, small random
            # mixing of neighbouring planes / pixels on top (keeps the known-shift answer recoverable)
            w = randn(shape, 0.02)
            idx = torch.arange(min(shape[0], shape[1]))
            w[idx, idx, shape[2] // 2, shape[3] // 2] += 1.0
            sd[name] = w
            sd[name[:-len('.weight')] + '.bias'] = randn((shape[0],), 0.01)
        elif '.agg3d.' in '.' + name and name.endswith('.weight') and len(shape) == 5:
            # 3-D aggregation layer (1,1,3,3,3): centre tap passes the volume through, small random neighbours
            w = randn(shape, 0.03)
            w[0, 0, 1, 1, 1] += 1.0
            sd[name] = w
            sd[name[:-len('.weight')] + '.bias'] = randn((1,), 0.01)
        elif '.reduce.' in '.' + name and name.endswith('.weight') and len(shape) == 4:
            # channel reduction of the stereo module's full-resolution mode (1x1, no activation): a random projection
            sd[name] = randn(shape, 1.0 / math.sqrt(shape[1]))
            sd[name[:-len('.weight')] + '.bias'] = randn((shape[0],), 0.01)
        elif name.endswith('.weight') and len(shape) == 4:  # bare prediction Conv2d
            fan_in = shape[1] * shape[2] * shape[3]
            prefix = name[:-len('.weight')]
            if 'conv_reg' in prefix:
                w = zero_sum(randn(shape, 0.15 / math.sqrt(fan_in * _SILU_VAR)))
                # log-size rows (w, h): keep exp(pred) within ~[0.4, 2.5] x stride, i.e. 4..80 px boxes like
                # the drones of the AirDrone set (8-60 px rectangles), not 600 px outliers
                w[2:4] *= 0.3
                sd[name] = w
                sd[prefix + '.bias'] = randn((shape[0],), 0.05)
            else:  # conv_cls / conv_obj
                sd[name] = zero_sum(randn(shape, logit_std / math.sqrt(fan_in * _SILU_VAR)))
                sd[prefix + '.bias'] = torch.full((shape[0],), -math.log((1 - prior_prob) / prior_prob))
    missing = [n for n in shapes if n not in sd]
    if missing:
        raise RuntimeError(f'synthetic_state_dict: no rule for parameters {missing[:5]} ...')
    for n, t in sd.items():
        if tuple(t.shape) != shapes[n]:
            raise RuntimeError(f'synthetic_state_dict: {n} shape {tuple(t.shape)} != {shapes[n]}')
    return sd


def synthetic_stereo_pair(seed, height=720, width=1280, max_disp=192, block=32, d_lo=2, d_hi=None,
                          smooth=3):
    """One synthetic stereo pair.

    Returns dict of numpy arrays:
      left, right : uint8 (3, H, W) BGR-like texture
      disp        : float32 (H, W) ground-truth disparity of the LEFT view in px (0 = invalid:
                    pixels whose match falls outside the right image)
    Disparity is piecewise constant on `block` x `block` tiles in [d_lo, d_hi]; right[x - d] = left[x].
    """
    rng = np.random.RandomState(seed)
    d_hi = (max_disp - 1) if d_hi is None else d_hi
    d_hi = max(d_lo, min(d_hi, max_disp - 1))
    tex = rng.randint(0, 256, size=(3, height, width + max_disp)).astype(np.float32)
    if smooth > 1:  # mild box blur along x and y so stride-4 features keep the texture
        k = smooth
        pad = k // 2
        t = np.pad(tex, ((0, 0), (pad, pad), (pad, pad)), mode='reflect')
        acc = np.zeros_like(tex)
        for dy in range(k):
            for dx in range(k):
                acc += t[:, dy:dy + height, dx:dx + width + max_disp]
        tex = acc / (k * k)
        tex = (tex - tex.mean()) * (k * 0.9) + 127.5
    tex = np.clip(np.rint(tex), 0, 255).astype(np.uint8)
    bh, bw = (height + block - 1) // block, (width + block - 1) // block
    dblk = rng.randint(d_lo, d_hi + 1, size=(bh, bw))
    # keep it a multiple of 4 so the quarter-resolution module has an exact answer
    dblk = (dblk // 4) * 4
    disp = np.kron(dblk, np.ones((block, block), dtype=np.int64))[:height, :width].astype(np.int64)
    xs = np.arange(width)[None, :].repeat(height, 0)
    # right image = texture; left[x] = right[x - d]  (sample the texture at x - d + max_disp)
    right = tex[:, :, max_disp:max_disp + width]
    src_x = xs - disp + max_disp
    left = np.take_along_axis(tex, np.broadcast_to(src_x[None], (3, height, width)), axis=2)
    valid = (xs - disp) >= 0
    disp_f = np.where(valid, disp, 0).astype(np.float32)
    return dict(left=np.ascontiguousarray(left), right=np.ascontiguousarray(right), disp=disp_f)


def pad_to_divisor(arr, divisor=32, value=0):
    """Right/bottom pad the last two dims (stack_batch semantics, reference utils/misc.py:13-64)."""
    h, w = arr.shape[-2:]
    H = (h + divisor - 1) // divisor * divisor
    W = (w + divisor - 1) // divisor * divisor
    if (H, W) == (h, w):
        return arr
    out = np.full(arr.shape[:-2] + (H, W), value, dtype=arr.dtype)
    out[..., :h, :w] = arr
    return out


def synthetic_batch(seeds, height=720, width=1280, max_disp=192, device='cpu'):
    """Batch in the layout TrackDataPreprocessor_Disparity_V1 hands to the detector
    (reference data_preprocessor_disparity_v1.py:21-84 + transforms_disparity.py:234-249):
    img / right (N,3,Hp,Wp) float32 0..255 padded with 114, disp_postp (N,3,Hp,Wp) float32 px
    padded with 0 (3-channel repeat, loading_disparity.py:85-86), disp_mask (N,1,Hp,Wp)."""
    imgs, rights, disps = [], [], []
    for s in seeds:
        p = synthetic_stereo_pair(s, height, width, max_disp)
        imgs.append(pad_to_divisor(p['left'].astype(np.float32), 32, 114.0))
        rights.append(pad_to_divisor(p['right'].astype(np.float32), 32, 114.0))
        d = pad_to_divisor(p['disp'], 32, 0.0)
        disps.append(np.repeat(d[None], 3, 0))
    out = dict(img=torch.from_numpy(np.stack(imgs)), right=torch.from_numpy(np.stack(rights)),
               disp_postp=torch.from_numpy(np.stack(disps)))
    out['disp_mask'] = (out['disp_postp'][:, :1] > 0).float()
    return {k: v.to(device) for k, v in out.items()}


def synthetic_detection_stream(seed=51, T=64, K=6, occlusion=(3, 20, 28), duplicates=False):
    """Seeded synthetic DETECTION stream for the association step (SURVEY.md §8c fixture iv / §8d config 3): K objects
    with constant velocity + noise, 10 % dropped detections, one object occluded for a few frames, depth-consistent
    scales; boxes are the depth-SCALED boxes the tracker consumes.  `duplicates`: every 7th frame repeats a detection
    exactly (equal IoU => the assignment optimum is not unique and the solver's tie behaviour decides the ids).
    -> float32 rows [t, x1, y1, x2, y2, score, depth, scale]."""
    rng = np.random.RandomState(seed)
    pos = rng.uniform([100, 80], [1100, 600], (K, 2))
    vel = rng.uniform(-4, 4, (K, 2))
    size = rng.uniform(12, 50, (K, 2))
    depth = rng.uniform(5, 60, K)
    score = rng.uniform(0.35, 0.95, K)
    rows = []
    for t in range(T):
        p = pos + vel * t + rng.normal(0, 0.4, (K, 2))
        keep = (rng.uniform(size=K) > 0.1) | (t == 0)
        k_occ, t0, t1 = occlusion
        keep[k_occ] &= not (t0 <= t < t1)          # an occlusion: the track must be re-identified
        b = np.concatenate([p - size / 2, p + size / 2], 1)[keep].astype(np.float32)
        sc = (score[keep] + rng.normal(0, 0.02, keep.sum())).astype(np.float32)
        dp = (depth[keep] + rng.normal(0, 0.2, keep.sum())).astype(np.float32)
        scl = np.clip(dp * dp / 400.0, 1.0, 3.0).astype(np.float32)
        if duplicates and t % 7 == 3 and len(b):
            j = t % len(b)
            b, sc, dp, scl = (np.concatenate([a, a[j:j + 1]]) for a in (b, sc, dp, scl))
        for i in range(len(b)):
            rows.append([t, *b[i], sc[i], dp[i], scl[i]])
    return np.asarray(rows, np.float32)
