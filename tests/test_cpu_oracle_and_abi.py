"""CPU suite (-m "not gpu"), part 1: the oracle against independent formulations and the committed
golden vectors; the C ABI exports every symbol include/stereotrack.h declares (no compute calls)."""
import os
import re

import numpy as np
import pytest
import torch

from oracle import c_oracle
from oracle import depth as odepth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, 'tests', 'golden')


# ---- C ABI -------------------------------------------------------------------------------------------
def header_functions():
    src = open(os.path.join(ROOT, 'include', 'stereotrack.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(st_[a-z0-9_]+)\s*\(', src)))


def test_library_exports_every_declared_symbol(stlib):
    from stereotracking_amd import _lib
    names = header_functions()
    assert len(names) >= 30
    for n in names:
        assert hasattr(stlib, n), f'{n} declared in include/stereotrack.h but not exported'
        assert n in _lib._PROTOS, f'{n} has no ctypes prototype'
    assert stlib.st_version() == 430      # ST_VERSION of include/stereotrack.h (also keys the tuning cache)


@pytest.mark.parametrize('nc,multi_label', [(3, False), (6, True), (6, False), (80, False)])
def test_oracle_wide_head_and_single_label_decode_matches_torch_formulation(nc, multi_label):
    """Heads wider than 3 classes (rows of head_row_floats(nc) floats) and test_cfg.multi_label=False (one candidate per
    prior: the class of its largest score) through oracle_decode_nms_gen == the independent torch formulation: same
    priors, labels and order, floats to 1e-5."""
    rng = np.random.RandomState(60 + nc)
    N, H, W = 2, 96, 160
    hr = c_oracle.head_row_floats(nc)
    assert hr == (8 if nc <= 3 else (nc + 8) // 4 * 4) and hr >= nc + 5
    levels, off = [], 0
    for s_ in (8, 16, 32):
        levels.append((H // s_, W // s_, s_, off))
        off += N * (H // s_) * (W // s_) * hr
    head = np.full(off, np.nan, np.float32)
    for h, w, s_, o in levels:
        rows = head[o:o + N * h * w * hr].reshape(N, h * w, hr)
        rows[..., :nc] = rng.normal(-1.5, 2.0, rows.shape[:2] + (nc,))
        rows[..., nc:nc + 2] = rng.normal(0, 1.0, rows.shape[:2] + (2,))
        rows[..., nc + 2:nc + 4] = rng.normal(0.6, 0.7, rows.shape[:2] + (2,))
        rows[..., nc + 4] = rng.normal(-1.0, 2.0, rows.shape[:2])
    r0 = head[:N * levels[0][0] * levels[0][1] * hr].reshape(N, -1, hr)
    r0[0, 5, :nc + 5] = r0[0, 4, :nc + 5]
    r0[0, 7, :nc] = r0[0, 7, :nc].max()        # every class ties for the best score: the first one wins
    M = 1500
    b, sc, lab, pri, cnt = c_oracle.decode_nms(head, N, levels, 0.05, 0.5, M, (H - 6, W), num_classes=nc,
                                               multi_label=multi_label)
    ref = torch_decode_nms_multiclass(head, N, levels, 0.05, 0.5, (H - 6, W), nc, hr, multi_label)
    assert cnt.min() > 30
    for n in range(N):
        k = int(cnt[n])
        assert k == len(ref[n][0]) <= M
        assert np.array_equal(pri[n, :k], ref[n][0]) and np.array_equal(lab[n, :k], ref[n][1])
        assert np.abs(b[n, :k] - ref[n][2]).max() < 1e-3 and np.abs(sc[n, :k] - ref[n][3]).max() < 1e-5
        if not multi_label:
            assert len(set(pri[n, :k].tolist())) == k
    if not multi_label:
        hit = [i for i in range(int(cnt[0])) if pri[0, i] == 7]
        assert not hit or lab[0, hit[0]] == 0


def test_struct_sizes_match_the_library(stlib):
    """A struct_size mismatch is reported, not crashed on."""
    import ctypes as C
    from stereotracking_amd._lib import StDetectorConfig
    cfg = StDetectorConfig(4, 0.5, 0.33, 1, 1, 64, 64, 1e-3, 0)  # wrong struct_size
    h = C.c_void_p()
    assert stlib.st_detector_create(C.byref(cfg), C.byref(h)) == -1
    assert b'struct_size' in stlib.st_last_error()
    cfg = StDetectorConfig(C.sizeof(StDetectorConfig), 0.5, 0.33, 1, 1, 70, 64, 1e-3, 0)  # H not /32
    assert stlib.st_detector_create(C.byref(cfg), C.byref(h)) == -1


def test_param_table_equals_reference_state_dict_layout():
    from oracle.torch_model import OracleDetector
    from stereotracking_amd.engine import HipDetector
    for widen in (0.375, 0.5):
        det = HipDetector(1, 64, 96, widen, 0.33, 1)
        table = dict(det.param_table())
        sd = {k: tuple(v.shape) for k, v in OracleDetector(0.33, widen, 1).state_dict().items()
              if not k.endswith('num_batches_tracked')}
        assert table == sd
        assert 'backbone.disp_stage1.1.blocks.0.conv2.bn.running_var' in table
        assert 'neck.top_down_layers.0.1.conv.weight' in table
        assert 'bbox_head.head_module.multi_level_conv_obj.2.bias' in table
    # analytic cost of the path (33.478 GMAC per frame-pair at 736x1280, YOLOX-s)
    det = HipDetector(1, 736, 1280, 0.5, 0.33, 1)
    assert abs(det.macs / 1e9 - 33.478) < 0.01
    assert det.num_priors == 19320


def test_forward_without_gpu_fails_loudly():
    from stereotracking_amd.engine import HipDetector
    det = HipDetector(1, 64, 96, 0.375, 0.33, 1)
    x = torch.zeros(1, 3, 64, 96)
    with pytest.raises(RuntimeError, match='CUDA'):
        det.forward(x, x)


# ---- oracle: exact helpers -------------------------------------------------------------------------------
def test_oracle_expf_is_within_2ulp_of_libm():
    lib = c_oracle.load()
    x = np.concatenate([np.linspace(-87, 88, 4001), np.linspace(-1, 1, 1001)]).astype(np.float32)
    got = np.array([lib.oracle_expf(float(v)) for v in x], np.float32)
    ref = np.exp(x.astype(np.float64))
    ulp = np.abs(got.astype(np.float64) - ref) / np.spacing(ref.astype(np.float32)).astype(np.float64)
    assert ulp.max() <= 2.0
    assert lib.oracle_expf(0.0) == 1.0 and lib.oracle_expf(float(np.float32(np.log(2.0)))) == 2.0
    assert lib.oracle_expf(-200.0) == 0.0 and np.isinf(lib.oracle_expf(100.0))


# ---- oracle: decode + NMS --------------------------------------------------------------------------------
def torch_decode_nms(head, N, levels, score_thr, iou_thr, ori_shape):
    """Independent formulation with torch ops (sigmoid/exp from ATen, brute-force greedy NMS)."""
    outs = []
    for n in range(N):
        rows, priors, strides = [], [], []
        for h, w, s, off in levels:
            r = torch.from_numpy(head[off:off + N * h * w * 8].reshape(N, h * w, 8)[n])
            ys, xs = torch.meshgrid(torch.arange(h), torch.arange(w), indexing='ij')
            priors.append(torch.stack([xs.reshape(-1) * s, ys.reshape(-1) * s], -1).float())
            strides.append(torch.full((h * w,), float(s)))
            rows.append(r)
        r, p, s = torch.cat(rows), torch.cat(priors), torch.cat(strides)
        score = torch.sigmoid(r[:, 0]) * torch.sigmoid(r[:, 5])
        xy = r[:, 1:3] * s[:, None] + p
        wh = r[:, 3:5].exp() * s[:, None]
        boxes = torch.cat([xy - wh / 2, xy + wh / 2], -1)
        idx = torch.nonzero(score > score_thr)[:, 0]
        order = idx[torch.sort(score[idx], descending=True, stable=True)[1]]
        keep = []
        sup = torch.zeros(len(order), dtype=torch.bool)
        b = boxes[order]
        area = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
        for i in range(len(order)):
            if sup[i]:
                continue
            keep.append(i)
            lt = torch.max(b[i, :2], b[i + 1:, :2])
            rb = torch.min(b[i, 2:], b[i + 1:, 2:])
            wh_ = (rb - lt).clamp(min=0)
            inter = wh_[:, 0] * wh_[:, 1]
            sup[i + 1:] |= inter / (area[i] + area[i + 1:] - inter) > iou_thr
        kb = b[keep].clone()
        kb[:, 0::2] = kb[:, 0::2].clamp(0, ori_shape[1])
        kb[:, 1::2] = kb[:, 1::2].clamp(0, ori_shape[0])
        outs.append((order[keep].numpy(), kb.numpy(), score[order[keep]].numpy()))
    return outs


def test_oracle_decode_nms_matches_torch_formulation_and_golden():
    g = np.load(os.path.join(GOLD, 'decode_nms.npz'))
    levels = [tuple(int(v) for v in l) for l in g['levels']]
    N = 2
    ori = (int(g['ori_h']), int(g['ori_w']))
    b, s, l, p, c = c_oracle.decode_nms(g['head'], N, levels, float(g['score_thr']), float(g['iou_thr']), 840, ori)
    # golden: bit-exact regression pin of the oracle
    assert np.array_equal(c, g['counts'])
    k = int(c.max())
    assert np.array_equal(p[:, :k], g['prior']) and np.array_equal(b[:, :k], g['boxes'])
    assert np.array_equal(s[:, :k], g['scores'])
    # independent torch formulation: same kept set and order, floats within 1e-4 relative
    ref = torch_decode_nms(g['head'], N, levels, float(g['score_thr']), float(g['iou_thr']), ori)
    for n in range(N):
        idx, rb, rs = ref[n]
        kk = int(c[n])
        assert kk == len(idx) and np.array_equal(p[n, :kk], idx)
        assert np.allclose(b[n, :kk], rb, rtol=1e-5, atol=1e-3)
        assert np.allclose(s[n, :kk], rs, rtol=1e-5, atol=1e-7)


def test_oracle_nms_properties():
    """Kept boxes are mutually non-overlapping above thr; every dropped candidate overlaps an earlier kept one."""
    g = np.load(os.path.join(GOLD, 'decode_nms.npz'))
    levels = [tuple(int(v) for v in l) for l in g['levels']]
    thr = 0.5
    # shift every box by +2000 px (negative pad_param) so the final clamp to [0, ori] never bites and the
    # returned boxes are exactly the ones NMS ran on
    shift = (-2000.0, 0.0, -2000.0, 0.0)
    b, s, l, p, c = c_oracle.decode_nms(g['head'], 2, levels, 0.01, thr, 840, (1e6, 1e6), (1.0, 1.0), shift)
    ball, sall, _, pall, call = c_oracle.decode_nms(g['head'], 2, levels, 0.01, 2.0, 840, (1e6, 1e6), (1.0, 1.0),
                                                    shift)  # thr > 1: no suppression
    assert b[:, :int(c.min())].min() > 0

    def iou(a, bb):
        lt, rb = np.maximum(a[:2], bb[:, :2]), np.minimum(a[2:], bb[:, 2:])
        wh = np.clip(rb - lt, 0, None)
        inter = wh[:, 0] * wh[:, 1]
        return inter / ((a[2] - a[0]) * (a[3] - a[1]) + (bb[:, 2] - bb[:, 0]) * (bb[:, 3] - bb[:, 1]) - inter)

    for n in range(2):
        k, ka = int(c[n]), int(call[n])
        assert np.all(np.diff(s[n, :k]) <= 0), 'kept boxes must come in score order'
        for i in range(k):
            assert np.all(iou(b[n, i], b[n, :i]) <= thr + 1e-6)
        kept = set(p[n, :k].tolist())
        for j in range(ka):
            if int(pall[n, j]) in kept:
                continue
            earlier = [i for i in range(k) if s[n, i] >= sall[n, j]]
            assert np.any(iou(ball[n, j], b[n, earlier]) > thr - 1e-6)


# ---- oracle: stereo module -----------------------------------------------------------------------------------
def test_oracle_costvolume_softargmin_upsample_vs_numpy_torch_and_golden():
    g = np.load(os.path.join(GOLD, 'costvolume.npz'))
    fl, fr = g['featL'], g['featR']
    N, Hf, Wf, Cc = fl.shape
    D = g['cost'].shape[-1]
    cost = c_oracle.costvolume(fl, fr, Cc, D)
    assert np.array_equal(cost, g['cost'])
    ref = np.zeros_like(cost, dtype=np.float64)
    for d in range(D):
        ref[:, :, d:, d] = (fl[:, :, d:].astype(np.float64) * fr[:, :, :Wf - d]).sum(-1) / Cc
    assert np.abs(cost - ref).max() < 1e-5
    T = float(g['temperature'])
    lr = c_oracle.softargmin(cost, T)
    assert np.array_equal(lr, g['disp_lr'])
    sm = torch.softmax(torch.from_numpy(cost).double() * T, -1)
    ref_lr = (sm * torch.arange(D, dtype=torch.float64)).sum(-1).numpy()
    assert np.abs(lr - ref_lr).max() < 1e-4
    # known answer: top half shifted by 5, bottom half by 9 (away from the left border)
    # (soft-argmin of noisy features: most pixels land on the true shift, a few are pulled by a second peak)
    assert np.mean(np.abs(lr[0, :3, 16:] - 5) < 0.05) > 0.9 and np.mean(np.abs(lr[0, 3:, 16:] - 9) < 0.05) > 0.9
    assert abs(np.median(lr[0, :3, 16:]) - 5) < 1e-3 and abs(np.median(lr[0, 3:, 16:]) - 9) < 1e-3
    up = c_oracle.disp_upsample(lr, 4, Hf * 4 - 8, Wf * 4)
    assert np.array_equal(up, g['disp_postp'])
    t = torch.nn.functional.interpolate(torch.from_numpy(lr)[:, None], scale_factor=4, mode='bilinear',
                                        align_corners=False)[:, 0].numpy() * 4
    assert np.abs(up[:, 0, :Hf * 4 - 8] - t[:, :Hf * 4 - 8]).max() < 1e-4
    assert np.all(up[:, :, Hf * 4 - 8:] == 0) and np.array_equal(up[:, 0], up[:, 2])


# ---- oracle: per-box depth -------------------------------------------------------------------------------------
def test_oracle_box_depth_golden_and_branches():
    g = np.load(os.path.join(GOLD, 'box_depth.npz'))
    disp3 = np.repeat(g['disp'][None, None], 3, 1)
    d, s, sb = odepth.bbox_postp_depth(torch.from_numpy(g['boxes']), torch.from_numpy(disp3))
    d = np.array([float(v) for v in d], np.float32)
    assert np.array_equal(np.isnan(d), np.isnan(g['depth']))
    ok = ~np.isnan(d)
    assert np.array_equal(d[ok], g['depth'][ok]) and np.array_equal(s.numpy()[ok], g['scales'][ok])
    assert np.array_equal(sb.numpy()[ok], g['scaled_boxes'][ok])
    # the fixture exercises every branch of extract_depth
    assert (d == -1).sum() >= 3, 'empty window / wrapped slice / all-invalid window'
    assert np.isnan(d).sum() >= 1, 'len 1 with all corners above the median -> empty segment -> NaN'
    assert ((g['scales'] > 1) & (g['scales'] < 3)).sum() >= 2, 'unclamped d*d'
    assert (g['scales'] == 3).sum() >= 5


def test_oracle_detector_golden():
    from oracle.torch_model import OracleDetector, head_to_rows
    from stereotracking_amd.synthetic import synthetic_batch, synthetic_state_dict
    g = np.load(os.path.join(GOLD, 'detector_tiny.npz'))
    torch.set_num_threads(1)
    ora = OracleDetector(0.33, 0.375, 1).eval()
    table = [(k, tuple(v.shape)) for k, v in ora.state_dict().items() if not k.endswith('num_batches_tracked')]
    ora.load_state_dict(synthetic_state_dict(table, seed=int(g['weights_seed'])), strict=False)
    batch = synthetic_batch([int(g['input_seed'])], 48, 96, 32)
    assert abs(batch['img'].double().sum().item() - float(g['img_sum'])) < 1e-6
    with torch.no_grad():
        rows = head_to_rows(*ora(batch))
    for l, r in enumerate(rows):
        ref = g[f'head{l}']
        assert r.shape == ref.shape
        # same code, same seeds: only the BLAS/oneDNN summation order may differ between hosts
        assert np.abs(r.numpy() - ref).max() <= 1e-4 * max(1.0, np.abs(ref).max())


def torch_decode_nms_multiclass(head, N, levels, score_thr, iou_thr, ori_shape, nc, hr=8, multi_label=True):
    """Independent torch formulation of the multi-class path [upstream-memory: mmyolo predict_by_feat with
    multi_label=True / False, mmdet filter_scores_and_topk, mmcv batched_nms offset trick]."""
    outs = []
    for n in range(N):
        rows, priors, strides = [], [], []
        for h, w, s, off in levels:
            r = torch.from_numpy(head[off:off + N * h * w * hr].reshape(N, h * w, hr)[n])
            ys, xs = torch.meshgrid(torch.arange(h), torch.arange(w), indexing='ij')
            priors.append(torch.stack([xs.reshape(-1) * s, ys.reshape(-1) * s], -1).float())
            strides.append(torch.full((h * w,), float(s)))
            rows.append(r)
        r, p, s = torch.cat(rows), torch.cat(priors), torch.cat(strides)
        scores = torch.sigmoid(r[:, :nc]) * torch.sigmoid(r[:, nc + 4:nc + 5])        # (P, nc)
        xy = r[:, nc:nc + 2] * s[:, None] + p
        wh = r[:, nc + 2:nc + 4].exp() * s[:, None]
        boxes = torch.cat([xy - wh / 2, xy + wh / 2], -1)
        if not multi_label:      # predict_by_feat: scores, labels = scores.max(1, keepdim=True); then the threshold
            best, lab1 = scores.max(1)
            keep1 = torch.nonzero(best > score_thr)[:, 0]
            valid = torch.stack([keep1, lab1[keep1]], 1)
        else:
            valid = torch.nonzero(scores > score_thr)                                  # row-major (prior, class)
        sc = scores[valid[:, 0], valid[:, 1]]
        order = torch.sort(sc, descending=True, stable=True)[1]
        pri, lab, sc = valid[order, 0], valid[order, 1], sc[order]
        b = boxes[pri]
        bo = b + (lab.float() * (b.max() + 1.0))[:, None] if len(b) else b             # batched_nms offsets
        area = (bo[:, 2] - bo[:, 0]) * (bo[:, 3] - bo[:, 1])
        keep, sup = [], torch.zeros(len(pri), dtype=torch.bool)
        for i in range(len(pri)):
            if sup[i]:
                continue
            keep.append(i)
            lt = torch.max(bo[i, :2], bo[i + 1:, :2])
            rb = torch.min(bo[i, 2:], bo[i + 1:, 2:])
            wh_ = (rb - lt).clamp(min=0)
            inter = wh_[:, 0] * wh_[:, 1]
            sup[i + 1:] |= inter / (area[i] + area[i + 1:] - inter) > iou_thr
        kb = b[keep].clone()
        kb[:, 0::2] = kb[:, 0::2].clamp(0, ori_shape[1])
        kb[:, 1::2] = kb[:, 1::2].clamp(0, ori_shape[0])
        outs.append((pri[keep].numpy(), lab[keep].numpy(), kb.numpy(), sc[keep].numpy()))
    return outs


@pytest.mark.parametrize('nc', [2, 3])
def test_oracle_multiclass_decode_nms_matches_torch_formulation(nc):
    """num_classes 2..3 (rows = class logits, x, y, w, h, obj): multi_label candidates in (prior, class) order,
    class-aware NMS on offset boxes; the same priors / labels / order as the independent torch formulation, floats to
    1e-5 (libm vs the oracle's polynomial exp); and with ONE class the multi-class entry equals the single-class one."""
    rng = np.random.RandomState(30 + nc)
    N, H, W = 2, 96, 160
    levels, off = [], 0
    for s_ in (8, 16, 32):
        levels.append((H // s_, W // s_, s_, off))
        off += N * (H // s_) * (W // s_) * 8
    head = np.zeros(off, np.float32)
    for h, w, s_, o in levels:
        rows = head[o:o + N * h * w * 8].reshape(N, h * w, 8)
        rows[..., :nc] = rng.normal(-1.5, 2.0, rows.shape[:2] + (nc,))
        rows[..., nc:nc + 2] = rng.normal(0, 1.0, rows.shape[:2] + (2,))
        rows[..., nc + 2:nc + 4] = rng.normal(0.6, 0.7, rows.shape[:2] + (2,))
        rows[..., nc + 4] = rng.normal(-1.0, 2.0, rows.shape[:2])
    # exact duplicates within and across classes: ties in score (order = prior, then class) and IoU == 1
    r0 = head[:N * levels[0][0] * levels[0][1] * 8].reshape(N, -1, 8)
    r0[0, 5] = r0[0, 4]
    r0[0, 7, :nc] = r0[0, 7, 0]
    M = 600
    b, sc, lab, pri, cnt = c_oracle.decode_nms(head, N, levels, 0.05, 0.5, M, (H - 6, W), num_classes=nc)
    ref = torch_decode_nms_multiclass(head, N, levels, 0.05, 0.5, (H - 6, W), nc)
    assert cnt.min() > 30 and len(set(lab[0, :cnt[0]].tolist())) == nc
    for n in range(N):
        k = int(cnt[n])
        assert k == len(ref[n][0]) <= M
        assert np.array_equal(pri[n, :k], ref[n][0]) and np.array_equal(lab[n, :k], ref[n][1])
        assert np.abs(b[n, :k] - ref[n][2]).max() < 1e-3 and np.abs(sc[n, :k] - ref[n][3]).max() < 1e-5
        assert np.all(np.diff(sc[n, :k]) <= 0)
    # one class through the multi-class entry == the single-class path
    one = np.zeros_like(head)
    for h, w, s_, o in levels:
        src = head[o:o + N * h * w * 8].reshape(N, h * w, 8)
        dst = one[o:o + N * h * w * 8].reshape(N, h * w, 8)
        dst[..., 0] = src[..., 0]
        dst[..., 1:6] = src[..., nc:nc + 5]
    a = c_oracle.decode_nms(one, N, levels, 0.05, 0.5, M, (H - 6, W))
    lib = c_oracle.load()
    import ctypes as C
    L = len(levels)
    arr = lambda t, v: (t * L)(*v)
    bb, ss, ll, pp, cc = (np.zeros((N, M, 4), np.float32), np.zeros((N, M), np.float32), np.zeros((N, M), np.int64),
                          np.full((N, M), -1, np.int32), np.zeros(N, np.int32))
    f = C.c_float
    ptr_ = lambda x: x.ctypes.data_as(C.c_void_p)
    rc = lib.oracle_decode_nms_mc(ptr_(one), C.c_int(N), C.c_int(L), arr(C.c_int, [l[0] for l in levels]),
                                  arr(C.c_int, [l[1] for l in levels]), arr(C.c_int, [l[2] for l in levels]),
                                  arr(C.c_size_t, [l[3] for l in levels]), f(0.05), f(0.5), C.c_int(M), f(1), f(1), f(0),
                                  f(0), f(W), f(H - 6), C.c_int(1), ptr_(bb), ptr_(ss), ptr_(ll), ptr_(pp), ptr_(cc))
    assert rc == 0 and np.array_equal(cc, a[4]) and np.array_equal(pp, a[3]) and np.array_equal(bb, a[0])


def test_oracle_backbone_reproduces_the_reference_docstring_example():
    """The ONLY known answer the reference holds for this path (it ships no tests or fixtures): the docstring example
    of the backbone class, reference mmtrack/models/backbones/csp_darknet_disparity_v1.py:50-62 - a default-size model
    (deepen 1.0, widen 1.0) on a 1x3x416x416 input prints level shapes (1,256,52,52), (1,512,26,26), (1,1024,13,13).
    The oracle restatement must produce exactly these, and its arch table must be the reference's (:66-69)."""
    from oracle import torch_model as tm
    cls = next(getattr(tm, n) for n in dir(tm) if hasattr(getattr(tm, n), 'arch') and hasattr(getattr(tm, n), '_stage'))
    assert cls.arch == [[64, 128, 3, True, False], [128, 256, 9, True, False], [256, 512, 9, True, False],
                        [512, 1024, 3, False, True]]
    torch.manual_seed(0)
    model = cls(deepen_factor=1.0, widen_factor=1.0).eval()
    x = torch.rand(1, 3, 416, 416)
    with torch.no_grad():
        outs = model(dict(img=x, disp_postp=torch.rand(1, 3, 416, 416)))
    assert [tuple(o.shape) for o in outs] == [(1, 256, 52, 52), (1, 512, 26, 26), (1, 1024, 13, 13)]
    # the shipped YOLOX-s sizing (deepen 0.33, widen 0.5) at the path's 736 x 1280: SURVEY.md §3.2 shapes
    small = cls(deepen_factor=0.33, widen_factor=0.5).eval()
    with torch.no_grad():
        outs = small(dict(img=torch.rand(1, 3, 64, 96), disp_postp=torch.rand(1, 3, 64, 96)))
    assert [tuple(o.shape) for o in outs] == [(1, 128, 8, 12), (1, 256, 4, 6), (1, 512, 2, 3)]
    assert len([m for m in small.stage2.modules() if type(m).__name__ == 'DarknetBottleneck']) == 3   # round(9 * 0.33)


def test_oracle_agg3d_equals_conv3d():
    """oracle_agg3d (the 3-D aggregation's specification, oracle/st_oracle.c) against torch.nn.functional.conv3d in
    float64 on the same taps: the restatement IS a zero-padded single-channel 3x3x3 convolution + SiLU."""
    import torch.nn.functional as F
    from oracle import c_oracle
    rng = np.random.RandomState(3)
    vol = rng.randn(2, 5, 7, 12).astype(np.float32)
    w = rng.randn(3, 3, 3).astype(np.float32)
    for act in (0, 1):
        got = c_oracle.agg3d(vol, w, 0.3, act)
        x = torch.from_numpy(vol).double().permute(0, 3, 1, 2)[:, None]              # (N, 1, D, H, W)
        ref = F.conv3d(x, torch.from_numpy(w).double()[None, None], torch.tensor([0.3], dtype=torch.float64), padding=1)
        ref = (F.silu(ref) if act else ref)[:, 0].permute(0, 2, 3, 1).numpy()
        assert np.abs(got - ref).max() <= 1e-5
    ident = np.zeros((3, 3, 3), np.float32)
    ident[1, 1, 1] = 1.0
    assert np.array_equal(c_oracle.agg3d(vol, ident, 0.0, 0), vol)


def test_shipped_code_has_no_packed_fp32_operand_select():
    """Structural fence of the round-4 wrong-result finding: on MI355X a packed-fp32 VOP3P instruction whose
    source is selected by op_sel / op_sel_hi (operand broadcast) executes single steps with the bit dropped while bf16
    MFMAs of ANY kernel run on the chip (tools/micro/pkfma_corun.hip).  The product library must not contain that
    operand form at all - checked on the built gfx950 code objects, so a compiler that folds a broadcast into op_sel
    again fails here, on CPU, before anything runs."""
    import glob
    import isa_utils
    from stereotracking_amd import _lib
    if not os.path.exists(isa_utils.OBJDUMP):
        pytest.skip(f'{isa_utils.OBJDUMP} not present: the code objects cannot be disassembled here')
    if not os.path.exists(_lib.LIB_PATH):
        pytest.skip('product library not built (run __graft_entry__.build())')
    # the fence must look at the code that SHIPS: a library older than any kernel source is stale
    csrc = os.path.join(os.path.dirname(os.path.abspath(_lib.__file__)), 'csrc')
    srcs = [f for pat in ('*.hip', '*.cpp', '*.h', 'experiments/*') for f in glob.glob(os.path.join(csrc, pat))]
    newest = max(srcs, key=os.path.getmtime)
    assert os.path.getmtime(_lib.LIB_PATH) >= os.path.getmtime(newest), \
        f'{_lib.LIB_PATH} is older than {newest}: rebuild (make -C stereotracking_amd/csrc) before trusting this fence'
    code = isa_utils.disassemble_library(_lib.LIB_PATH)
    assert len(code) > 100, 'disassembly found too few kernels'
    n_packed = sum(1 for lines in code.values() for ins in lines if ins.startswith('v_pk_') and '_f32' in ins)
    assert n_packed > 0          # the Winograd transforms do use packed fp32 (default operand selection)
    bad = [(sym, ins) for sym, lines in code.items() for ins in lines if ins.startswith('v_pk_') and 'op_sel' in ins]
    assert not bad, f'{len(bad)} packed instructions with op_sel, e.g. {bad[:3]}'
    # and the product library brings no bf16 MFMA of its own (the split-operand instances are parked in the tools build)
    bf16 = [(sym, ins) for sym, lines in code.items() for ins in lines if ins.startswith('v_mfma') and 'bf16' in ins]
    assert not bf16 and not any('conv_split' in sym for sym in code), bf16[:3]
    assert _lib.load().st_split_instances_available() == 0
