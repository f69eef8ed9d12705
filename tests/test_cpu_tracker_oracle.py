"""CPU: the association rows of SURVEY.md §8 (a-10 lapjv-based assignment, a-11 OCSORTTracker_Disparity.track, a-12
KalmanFilter) — the PRODUCT tracker (stereotracking_amd/trackers.py + the library's st_lapjv_extended) against the
independent ORACLE (oracle/tracker.py, oracle/lapjv.py: statement-by-statement restatements of the reference
classes and of lap.lapjv).  north_star asks for bit-exact track indices, so everything index-like is compared with
==, and so are the floats the tracker merely passes through (boxes, scores, depth, scales)."""
import itertools
import os
import sys

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, 'golden'))

from make_golden import SHIPPED_TRACKER, detection_stream, run_oracle_tracker  # noqa: E402
from oracle import lapjv as olap  # noqa: E402
otr = pytest.importorskip('oracle.tracker')  # noqa: E402  (the reference-tracker restatement stays in the build container)
from stereotracking_amd.motion import KalmanFilter  # noqa: E402
from stereotracking_amd.structures import InstanceData, TrackDataSample  # noqa: E402
from stereotracking_amd.trackers import OCSORTTracker_Disparity, lapjv_extended  # noqa: E402


def brute_force_optimum(cost, lim):
    """Minimum of sum(matched costs) + lim/2 per unmatched row and per unmatched column (the objective of lap's
    (r+c)^2 extension), by enumeration of every partial matching."""
    r, c = cost.shape
    best = np.inf
    for k in range(min(r, c) + 1):
        for rows in itertools.combinations(range(r), k):
            for cols in itertools.permutations(range(c), k):
                best = min(best, sum(cost[i, j] for i, j in zip(rows, cols)) + (r + c - 2 * k) * lim / 2)
    return best


def assignment_value(cost, x, lim):
    k = int((x >= 0).sum())
    return sum(cost[i, x[i]] for i in range(len(x)) if x[i] >= 0) + (sum(cost.shape) - 2 * k) * lim / 2


def test_oracle_lapjv_reaches_the_brute_force_optimum():
    rng = np.random.RandomState(5)
    for t in range(300):
        r, c = rng.randint(1, 6), rng.randint(1, 6)
        cost = rng.rand(r, c)
        if t % 3 == 0:
            cost = np.round(cost * 4) / 4          # ties
        lim = float(rng.choice([0.3, 0.7, 0.9]))
        _, x, y = olap.lapjv(cost, True, lim)
        assert abs(assignment_value(cost, x, lim) - brute_force_optimum(cost, lim)) < 1e-9
        for i, j in enumerate(x):                 # x / y are mutually consistent
            assert j < 0 or y[j] == i
        assert sum(v >= 0 for v in x) == sum(v >= 0 for v in y)
    big = np.round(rng.rand(7, 7) * 8) / 8       # one 7x7 (130 922 partial matchings)
    _, x, _ = olap.lapjv(big, True, 0.9)
    assert abs(assignment_value(big, x, 0.9) - brute_force_optimum(big, 0.9)) < 1e-9


def test_lapjv_tie_fixtures_pin_the_chosen_optimum():
    """tests/golden/lapjv_ties.npz: 222 problems with 2..34 optimal assignments each (enumerated).  The oracle's
    restatement returns the pinned one, and the PRODUCT solver (C++, st_lapjv_extended) returns the same."""
    g = np.load(os.path.join(HERE, 'golden', 'lapjv_ties.npz'))
    assert len(g['cost']) > 100 and int(g['num_optimal'].min()) >= 2
    for pad, (r, c), lim, xp in zip(g['cost'], g['shape'], g['cost_limit'], g['x']):
        cost = pad[:r, :c]
        want = xp[:r]
        _, xo, yo = olap.lapjv(cost, True, float(lim))
        assert np.array_equal(xo, want)
        xq, yq = lapjv_extended(cost, float(lim))
        assert np.array_equal(xq, want), f'product lapjv picked another optimum: {xq} vs {want}\n{cost}'
        assert np.array_equal(yq, yo)


def test_product_lapjv_equals_oracle_on_random_and_degenerate_costs():
    rng = np.random.RandomState(7)
    for t in range(3000):
        r, c = rng.randint(1, 12), rng.randint(1, 12)
        cost = rng.rand(r, c)
        m = t % 5
        if m == 1:
            cost = np.round(cost * 4) / 4
        elif m == 2:
            cost[rng.rand(r, c) < 0.3] = 0.5
        elif m == 3:
            cost[rng.rand(r, c) < 0.1] = np.nan    # NaN boxes: unmatchable in both
        elif m == 4:
            cost[:] = 1.0                          # no overlap at all: nothing may match below cost_limit
        lim = float(rng.choice([0.3, 0.7, 0.9]))
        _, x, y = olap.lapjv(cost, True, lim)
        x2, y2 = lapjv_extended(cost, lim)
        assert np.array_equal(x, x2) and np.array_equal(y, y2)
        if m == 4:
            assert (x2 == -1).all() and (y2 == -1).all()
        if m == 3:
            assert not any(j >= 0 and np.isnan(cost[i, j]) for i, j in enumerate(x2))
    x, y = lapjv_extended(np.zeros((0, 3)), 0.9)
    assert len(x) == 0 and y.tolist() == [-1, -1, -1]


class _ProductModel:
    motion = KalmanFilter()


def run_product_tracker(det, num_frames, with_state=False, backend='native', **cfg):
    trk = OCSORTTracker_Disparity(backend=backend, **cfg)
    out = []
    for t in range(num_frames):
        d = det[det[:, 0] == t]
        s = TrackDataSample(dict(frame_id=t))
        s.pred_det_instances = InstanceData(
            bboxes=torch.from_numpy(d[:, 1:5].copy()), scores=torch.from_numpy(d[:, 5].copy()),
            labels=torch.zeros(len(d), dtype=torch.long), scales=torch.from_numpy(d[:, 7].copy()),
            depth=torch.from_numpy(d[:, 6].copy()))
        r = trk.track(_ProductModel(), None, None, s)
        for i in range(len(r.instances_id)):
            out.append([t, int(r.instances_id[i]), *r.bboxes[i].tolist(), float(r.scores[i]), float(r.depth[i]),
                        float(r.scales[i])])
    rows = np.asarray(out, np.float64).reshape(-1, 9)
    return (rows, trk) if with_state else rows


def run_oracle_with_state(det, num_frames, **cfg):
    class _Model:
        motion = otr.KalmanFilter()
    trk = otr.OCSORTTracker_Disparity(**cfg)
    for t in range(num_frames):
        d = det[det[:, 0] == t]
        inst = otr.Instances(bboxes=torch.from_numpy(d[:, 1:5].copy()), scores=torch.from_numpy(d[:, 5].copy()),
                             labels=torch.zeros(len(d), dtype=torch.long), scales=torch.from_numpy(d[:, 7].copy()),
                             depth=torch.from_numpy(d[:, 6].copy()))
        trk.track(_Model(), None, None, otr.Sample(t, inst))
    return trk


VARIANT = dict(SHIPPED_TRACKER, weight_iou_with_det_scores=True, match_iou_thr=0.3, num_frames_retain=10)


@pytest.mark.parametrize('backend', ['native', 'python'])
@pytest.mark.parametrize('seed,K,dup,cfg', [(51, 6, False, SHIPPED_TRACKER), (52, 9, False, SHIPPED_TRACKER),
                                           (53, 6, True, SHIPPED_TRACKER), (54, 12, True, SHIPPED_TRACKER),
                                           (55, 8, False, VARIANT), (56, 10, True, VARIANT),
                                           (57, 40, True, SHIPPED_TRACKER), (58, 25, False, VARIANT)])
def test_product_tracker_reproduces_the_oracle_frame_by_frame(seed, K, dup, cfg, backend):
    """Same detection stream through the oracle classes and the product tracker (native C++ routine and the pure
    Python restatement): ids, boxes, scores, depth and scales of every returned track are EQUAL in every frame.  The
    Kalman states of the tracks alive at the end are bit-equal for the Python backend (same numpy / scipy calls) and
    equal to 1e-9 for the native one (plain loops instead of BLAS / LAPACK: last-bit differences, see the header of
    csrc/ocsort_tracker.cpp)."""
    T = 48
    det = detection_stream(seed, T, K, occlusion=(K // 2, 15, 23), duplicates=dup)
    ref = run_oracle_tracker(det, T, **cfg)
    got, trk = run_product_tracker(det, T, with_state=True, backend=backend, **cfg)
    assert len(ref) > T and len(set(ref[:, 1].astype(int))) >= K
    for t in range(T):
        a, b = got[got[:, 0] == t], ref[ref[:, 0] == t]
        assert a[:, 1].astype(int).tolist() == b[:, 1].astype(int).tolist(), f'frame {t}: track ids differ'
        assert np.array_equal(a, b), f'frame {t}: boxes / scores / depth / scales differ'
    otrk = run_oracle_with_state(det, T, **cfg)
    assert int(trk.num_tracks) == int(otrk.num_tracks)
    if backend == 'python':
        assert sorted(trk.tracks) == sorted(otrk.tracks)
        for tid, tr in trk.tracks.items():
            o = otrk.tracks[tid]
            assert np.array_equal(np.asarray(tr.mean, np.float64), np.asarray(o.mean, np.float64)), tid
            assert np.array_equal(tr.covariance, o.covariance), tid
            assert bool(tr.tentative) == bool(o.tentative) and bool(tr.tracked) == bool(o.tracked)
    else:
        state = trk.native_state()
        assert [t['id'] for t in state] == list(otrk.tracks)          # same tracks alive, same creation order
        for t in state:
            o = otrk.tracks[t['id']]
            assert np.allclose(t['mean'], np.asarray(o.mean, np.float64), rtol=1e-9, atol=1e-9), t['id']
            assert np.allclose(t['covariance'], o.covariance, rtol=1e-9, atol=1e-12), t['id']
            assert t['tentative'] == bool(o.tentative) and t['tracked'] == bool(o.tracked)
            assert t['last_frame'] == int(o.frame_ids[-1])


def test_tracker_golden_sequence():
    """tests/golden/tracker_sequence.npz (SURVEY.md §8c fixture iv), generated by the ORACLE tracker with the shipped
    config: the product tracker reproduces ids / boxes / scores / depth / scales frame by frame, and identities
    survive the dropped detections and the 8-frame occlusion."""
    g = np.load(os.path.join(HERE, 'golden', 'tracker_sequence.npz'))
    det, ref, T = g['detections'], g['tracks'], int(g['num_frames'])
    assert np.array_equal(det, detection_stream(51, T))               # the committed stream is the seeded one
    assert np.array_equal(ref, run_oracle_tracker(det, T, **SHIPPED_TRACKER))   # and the oracle reproduces its fixture
    for backend in ('native', 'python'):
        got = run_product_tracker(det, T, backend=backend, **SHIPPED_TRACKER)
        assert got.shape == ref.shape and np.array_equal(got, ref), backend
    ids = lambda t: set(ref[ref[:, 0] == t][:, 1].astype(int).tolist())   # noqa: E731
    assert len(ids(28)) == 6 and ids(28) == ids(63)


def test_empty_and_first_frame_semantics():
    """Reference quirks kept on purpose (ocsort_tracker_disparity.py:391-404, 588-593): frame 0 starts tracks only
    above init_track_thr and they are born confirmed; later frames start tracks for EVERY unmatched detection that
    passed the obj_score_thr / area filter (no init threshold); a frame without detections skips prediction."""
    det = np.array([[0, 10, 10, 40, 40, 0.9, 10, 1], [0, 100, 100, 130, 130, 0.5, 10, 1],
                    [2, 12, 10, 42, 40, 0.9, 10, 1], [2, 300, 300, 330, 330, 0.35, 10, 1],
                    [2, 500, 300, 505, 305, 0.99, 10, 1]], np.float32)   # last box: area 25 < 100 => dropped
    ref = run_oracle_tracker(det, 3, **SHIPPED_TRACKER)
    for backend in ('native', 'python'):
        assert np.array_equal(run_product_tracker(det, 3, backend=backend, **SHIPPED_TRACKER), ref), backend
    assert ref[ref[:, 0] == 0][:, 1].tolist() == [0.0]                  # only the 0.9 box starts a track
    assert len(ref[ref[:, 0] == 1]) == 0
    assert sorted(ref[ref[:, 0] == 2][:, 1].tolist()) == [0.0, 1.0]     # re-found + a new track at score 0.35


def test_chunked_native_tracking_equals_frame_by_frame():
    """st_tracker_track_records (a chunk of frame records in ONE native call, what OCSORT_Disparity.predict uses)
    against the per-frame track() + scale_bbox(boxes, 1 / scales) of the same tracker: ids equal, every float
    bit-equal (the un-scaling is the same fp32 single operations torch evaluates), padding frames skipped, an
    overflowing record raises."""
    from stereotracking_amd.dist import DetectionOverflow
    from stereotracking_amd.mot import scale_bbox
    T, M = 24, 12
    det = detection_stream(51, T, duplicates=True)

    class _Model:
        motion = KalmanFilter()

    one = OCSORTTracker_Disparity(**SHIPPED_TRACKER)
    chunked = OCSORTTracker_Disparity(**SHIPPED_TRACKER)
    rec = np.zeros((T + 2, M + 1, 13), np.float32)          # 2 trailing padding frames (header all zero)
    rng = np.random.RandomState(3)
    for t in range(T):
        d = det[det[:, 0] == t]
        k = len(d)
        rec[t, 0, :3] = (k, M, 1)
        rec[t, 1:1 + k, 8:12] = d[:, 1:5]                    # the depth-scaled box the tracker consumes
        rec[t, 1:1 + k, 0:4] = rng.rand(k, 4)                # unscaled box of the detection: not used by the tracker
        rec[t, 1:1 + k, 4], rec[t, 1:1 + k, 6], rec[t, 1:1 + k, 7] = d[:, 5], d[:, 6], d[:, 7]
        rec[t, 1:1 + k, 12] = np.arange(k)
    fids = list(range(T)) + [-1, -1]
    n_rows = 0
    for lo, hi in ((0, 8), (8, 16), (16, T + 2)):            # three chunks, as predict() walks a video
        rows, ids, counts = chunked.track_records(fids[lo:hi], rec[lo:hi])
        for i, t in enumerate(range(lo, hi)):
            if t >= T:
                assert counts[i] == -1
                continue
            d = det[det[:, 0] == t]
            s = TrackDataSample(dict(frame_id=t))
            s.pred_det_instances = InstanceData(bboxes=torch.from_numpy(d[:, 1:5].copy()),
                                                scores=torch.from_numpy(d[:, 5].copy()),
                                                labels=torch.zeros(len(d), dtype=torch.long),
                                                scales=torch.from_numpy(d[:, 7].copy()),
                                                depth=torch.from_numpy(d[:, 6].copy()))
            ref = one.track(_Model(), None, None, s)
            m = int(counts[i])
            assert ids[i, :m].tolist() == ref.instances_id.tolist()
            want = scale_bbox(ref.bboxes, 1 / ref.scales).numpy()
            assert np.array_equal(rows[i, :m, 0:4].view(np.uint32), want.view(np.uint32))
            assert np.array_equal(rows[i, :m, 4], ref.scores.numpy()) and np.array_equal(rows[i, :m, 6], ref.depth.numpy())
            assert np.array_equal(rows[i, :m, 7], ref.scales.numpy())
            n_rows += m
    assert n_rows > 3 * T
    rec[0, 0, 0] = M + 5                                      # a frame that kept more boxes than the record holds
    with pytest.raises(DetectionOverflow, match='max_det'):
        chunked.track_records(fids[:8], rec[:8])
