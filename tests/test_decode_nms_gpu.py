"""GPU parity of decode + filter + sort + NMS (st_decode_nms) against the C oracle: BIT-EXACT
boxes, scores, kept prior indices and counts on identical head tensors."""
import numpy as np
import pytest
import torch

from oracle import c_oracle
from stereotracking_amd.engine import HipDetector

pytestmark = pytest.mark.gpu


def run_both(det, head_np, cuda, score_thr=0.01, iou_thr=0.5, max_det=None, ori_shape=None, scale=(1.0, 1.0), pad=None,
             nms_mask_rows=0):
    max_det = max_det or det.num_priors
    ori_shape = ori_shape or (det.height, det.width)
    ref = c_oracle.decode_nms(head_np, det.batch, det.levels, score_thr, iou_thr, max_det, ori_shape, scale, pad)
    got = det.decode_nms(torch.from_numpy(head_np).to(cuda), score_thr, iou_thr, max_det, ori_shape, scale, pad,
                         nms_mask_rows=nms_mask_rows)
    torch.cuda.synchronize()
    got = [g.cpu().numpy() for g in got]
    return got, ref


def assert_bit_exact(got, ref, max_det):
    gb, gs, gl, gp, gc = got
    rb, rs, rl, rp, rc = ref
    assert np.array_equal(gc, rc), f'counts differ: {gc} vs {rc}'
    for n in range(len(gc)):
        k = min(int(gc[n]), max_det)
        assert np.array_equal(gp[n, :k], rp[n, :k]), f'image {n}: kept prior indices differ'
        assert np.array_equal(gs[n, :k].view(np.uint32), rs[n, :k].view(np.uint32)), f'image {n}: scores not bit-exact'
        assert np.array_equal(gb[n, :k].view(np.uint32), rb[n, :k].view(np.uint32)), f'image {n}: boxes not bit-exact'
        assert np.all(gl[n, :k] == 0)


def random_head(det, rng, logit_mean=-2.0, logit_std=2.0, wh_std=0.8):
    head = np.zeros(det.head_floats, np.float32)
    for h, w, s, off in det.levels:
        rows = head[off:off + det.batch * h * w * 8].reshape(det.batch, h * w, 8)
        rows[..., 0] = rng.normal(logit_mean, logit_std, rows.shape[:2])
        rows[..., 5] = rng.normal(logit_mean, logit_std, rows.shape[:2])
        rows[..., 1:3] = rng.normal(0, 1.0, rows.shape[:2] + (2,))
        rows[..., 3:5] = rng.normal(0.5, wh_std, rows.shape[:2] + (2,))
        rows[..., 6:] = np.nan  # unused slots must be ignored
    return head


@pytest.fixture(scope='module')
def det_small():
    return HipDetector(2, 160, 256, 0.375, 0.33, 1)


@pytest.mark.parametrize('seed', [0, 1, 2])
def test_random_heads_bit_exact(det_small, seed, cuda):
    head = random_head(det_small, np.random.RandomState(seed))
    got, ref = run_both(det_small, head, cuda)
    assert ref[4].min() > 20, 'test should keep a meaningful number of boxes'
    assert_bit_exact(got, ref, det_small.num_priors)


def test_rescale_and_clamp(det_small, cuda):
    head = random_head(det_small, np.random.RandomState(5))
    got, ref = run_both(det_small, head, cuda, ori_shape=(100, 190), scale=(1.1, 1.2), pad=(4.0, 4.0, 6.0, 6.0))
    assert_bit_exact(got, ref, det_small.num_priors)
    k = int(got[4][0])
    assert got[0][0, :k, 0::2].max() <= 190 and got[0][0, :k, 1::2].max() <= 100


def test_empty_and_all_pass(det_small, cuda):
    head = random_head(det_small, np.random.RandomState(6), logit_mean=-12.0, logit_std=0.1)
    got, ref = run_both(det_small, head, cuda)
    assert np.all(ref[4] == 0)
    assert_bit_exact(got, ref, det_small.num_priors)
    # every prior passes the score filter (maximum candidate count), tiny boxes => nothing suppressed
    head = random_head(det_small, np.random.RandomState(7), logit_mean=6.0, logit_std=0.5, wh_std=0.01)
    for h, w, s, off in det_small.levels:
        rows = head[off:off + det_small.batch * h * w * 8].reshape(det_small.batch, h * w, 8)
        rows[..., 3:5] = -3.0
        rows[..., 1:3] *= 0.05
    got, ref = run_both(det_small, head, cuda)
    assert np.all(ref[4] == det_small.num_priors)
    assert_bit_exact(got, ref, det_small.num_priors)


def test_adversarial_ties_and_duplicates(det_small, cuda):
    """Exact score ties (tie-break = lower prior index first) and boxes with IoU exactly at the
    threshold (suppression needs IoU > thr, not >=)."""
    rng = np.random.RandomState(8)
    head = random_head(det_small, rng, logit_mean=-9.0, logit_std=0.01)
    h, w, s, off = det_small.levels[0]
    rows = head[off:off + det_small.batch * h * w * 8].reshape(det_small.batch, h * w, 8)
    # image 0: 40 priors with IDENTICAL logits and identical decoded boxes (prior-relative offsets chosen so
    # the decoded centre is the same pixel) -> only the lowest prior index survives
    for i in range(40):
        rows[0, i, 0] = rows[0, i, 5] = 3.0
        rows[0, i, 1] = (20 - i) * 1.0      # cx = (x + off)*8 with x = i  -> 160
        rows[0, i, 2] = 5.0
        rows[0, i, 3] = rows[0, i, 4] = 1.0
    # image 1: IoU exactly 0.5.  A = level-1 prior (x=1,y=1), exp(0)*16 -> [8,8,24,24];
    # B = level-0 prior (x=1,y=2), w = exp(0)*8, h = exp(ln 2)*8 -> [8,8,16,24]: inter 128, union 256.
    h1, w1, s1, off1 = det_small.levels[1]
    rows1 = head[off1:off1 + det_small.batch * h1 * w1 * 8].reshape(det_small.batch, h1 * w1, 8)
    qa = 1 * w1 + 1
    rows1[1, qa, 0] = rows1[1, qa, 5] = 4.0
    rows1[1, qa, 1:5] = (0.0, 0.0, 0.0, 0.0)
    qb = 2 * w + 1
    rows[1, qb, 0] = rows[1, qb, 5] = 3.0
    rows[1, qb, 1:5] = (0.5, 0.0, 0.0, np.float32(np.log(2.0)))
    got, ref = run_both(det_small, head, cuda)
    assert_bit_exact(got, ref, det_small.num_priors)
    assert ref[4][0] >= 1 and ref[3][0, 0] == 0
    kept1 = set(ref[3][1, :ref[4][1]].tolist())
    assert {qb, h * w + qa} <= kept1, 'IoU == thr must NOT suppress'
    # and just above the threshold it must: same geometry with iou_thr a hair below 0.5
    got, ref = run_both(det_small, head, cuda, iou_thr=0.4999)
    assert_bit_exact(got, ref, det_small.num_priors)
    assert qb not in set(ref[3][1, :ref[4][1]].tolist())


def test_max_det_truncation_reports_full_count(det_small, cuda):
    head = random_head(det_small, np.random.RandomState(9))
    got, ref = run_both(det_small, head, cuda, max_det=16)
    assert ref[4].min() > 16
    assert_bit_exact(got, ref, 16)


def test_full_size_priors(cuda):
    """BASELINE config[1] geometry: 19 320 priors per image at 736x1280."""
    det = HipDetector(2, 736, 1280, 0.5, 0.33, 1)
    assert det.num_priors == 19320
    head = random_head(det, np.random.RandomState(10), logit_mean=-3.0, logit_std=1.5)
    got, ref = run_both(det, head, cuda, max_det=2000, ori_shape=(720, 1280))
    assert_bit_exact(got, ref, 2000)


@pytest.mark.parametrize('mask_rows', [64, 128, 1024])
def test_candidates_beyond_the_iou_mask_are_resolved_on_the_fly(det_small, cuda, mask_rows):
    """The precomputed IoU bit mask covers only the first `nms_mask_rows` candidates (workspace no longer grows
    with priors^2); everything past it goes through the reduce kernel's on-the-fly path.  Results must not depend
    on where that boundary falls: bit-exact against the oracle with hundreds of candidates on both sides."""
    head = random_head(det_small, np.random.RandomState(21), logit_mean=-1.0)
    got, ref = run_both(det_small, head, cuda, score_thr=0.001, nms_mask_rows=mask_rows)
    assert ref[4].min() > 100
    assert_bit_exact(got, ref, det_small.num_priors)
    d = det_small.decode_desc(0.001, 0.5, 100, (160, 256), nms_mask_rows=64)
    small = det_small.lib.st_decode_nms_workspace_bytes(d)
    d.nms_mask_rows = 1 << 20   # clamped to the prior count (840 -> 896 rows)
    assert small < det_small.lib.st_decode_nms_workspace_bytes(d)


def test_full_size_default_workspace_is_bounded(cuda):
    det = HipDetector(8, 736, 1280, 0.5, 0.33, 1)
    d = det.decode_desc(0.01, 0.5, 1000, (720, 1280))
    assert det.lib.st_decode_nms_workspace_bytes(d) < 32 << 20      # was 373 MB with a priors^2 mask


@pytest.mark.parametrize('nc,multi_label', [(2, True), (3, True), (3, False), (5, True), (5, False), (20, True),
                                            (80, False)])
def test_multiclass_decode_nms_bit_exact(nc, multi_label, cuda):
    """Several classes (2..3 in 8-float head rows, wider heads in st_head_row_floats(nc)-float rows): multi_label
    candidates ((prior, class) pairs, filter_scores_and_topk's order) or multi_label=False (one candidate per prior:
    the class of its largest score, first maximum on ties - mmyolo predict_by_feat), and class-aware NMS by the offset
    trick of mmcv batched_nms - kept priors, LABELS, scores, boxes and counts bit-exact against the C oracle, on random
    heads with exact duplicates within and across classes, beyond-mask candidates included."""
    det = HipDetector(2, 160, 256, 0.375, 0.33, nc)
    det.multi_label = multi_label
    hr = det.head_row
    assert hr == c_oracle.head_row_floats(nc) and hr >= nc + 5 and hr % 4 == 0
    rng = np.random.RandomState(40 + nc)
    head = np.zeros(det.head_floats, np.float32)
    for h, w, s, off in det.levels:
        rows = head[off:off + det.batch * h * w * hr].reshape(det.batch, h * w, hr)
        rows[..., :nc] = rng.normal(-1.5 - 0.02 * nc, 2.0, rows.shape[:2] + (nc,))
        rows[..., nc:nc + 2] = rng.normal(0, 1.0, rows.shape[:2] + (2,))
        rows[..., nc + 2:nc + 4] = rng.normal(0.5, 0.8, rows.shape[:2] + (2,))
        rows[..., nc + 4] = rng.normal(-1.0, 2.0, rows.shape[:2])
        rows[..., nc + 5:] = np.nan
    h0, w0, _, off0 = det.levels[0]
    r0 = head[off0:off0 + det.batch * h0 * w0 * hr].reshape(det.batch, h0 * w0, hr)
    r0[0, 11, :nc + 5] = r0[0, 10, :nc + 5]          # duplicate prior rows: equal scores for every class
    r0[1, 20, :nc] = r0[1, 20, 0]                    # equal class scores on one prior: ties broken by class index
    r0[1, 21, :nc] = r0[1, 21, :nc].max()            # ... and a prior whose best score is shared by ALL classes
    M = min(det.num_priors * (nc if multi_label else 1), 12000)
    for mask_rows in (0, 128):                       # 128: most candidates are resolved on the fly by the reduce wave
        ref = c_oracle.decode_nms(head, det.batch, det.levels, 0.02, 0.5, M, (150, 256), num_classes=nc,
                                  multi_label=multi_label)
        got = det.decode_nms(torch.from_numpy(head).to(cuda), 0.02, 0.5, M, (150, 256), nms_mask_rows=mask_rows)
        torch.cuda.synchronize()
        gb, gs, gl, gp, gc = [g.cpu().numpy() for g in got]
        rb, rs, rl, rp, rc = ref
        assert np.array_equal(gc, rc) and rc.min() > 100 and rc.max() <= M
        for n in range(det.batch):
            k = int(rc[n])
            assert np.array_equal(gp[n, :k], rp[n, :k]) and np.array_equal(gl[n, :k], rl[n, :k])
            assert np.array_equal(gs[n, :k].view(np.uint32), rs[n, :k].view(np.uint32))
            assert np.array_equal(gb[n, :k].view(np.uint32), rb[n, :k].view(np.uint32))
            assert len(set(gl[n, :k].tolist())) >= min(nc, 5)
            if not multi_label:                       # one candidate per prior: no prior is kept twice
                assert len(set(gp[n, :k].tolist())) == k
            assert np.all(gp[n, k:] == -1) and not gb[n, k:].any()
