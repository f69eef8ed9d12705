"""CPU: CLEAR-MOT / Identity restatement on hand-built cases with known answers, the
depth-range filters and the MOTChallenge file formats of reference mot_drone_metrics.py:155-253."""
import os

import numpy as np
import torch

from stereotracking_amd.metrics import MOTDroneMetrics, box_iou_xywh, clear_identity
from stereotracking_amd.structures import InstanceData, TrackDataSample


def rows(frames_to_objs):
    out = []
    for f, objs in frames_to_objs.items():
        for oid, x, y in objs:
            out.append([f, oid, x, y, 20, 20, 1.0])
    return out


def test_perfect_tracking_scores_one():
    gt = rows({t: [(1, 10 + t, 10), (2, 100, 50 + t)] for t in range(1, 11)})
    pred = rows({t: [(7, 10 + t, 10), (9, 100, 50 + t)] for t in range(1, 11)})
    r = clear_identity(gt, pred)
    assert (r['TP'], r['FN'], r['FP'], r['IDSW']) == (20, 0, 0, 0)
    assert r['MOTA'] == 1.0 and abs(r['MOTP'] - 1.0) < 1e-12 and r['IDF1'] == 1.0


def test_id_switch_fp_fn_counts():
    # one gt object over 10 frames; tracker id 1 for frames 1-5, id 2 for 6-10 (1 switch); one spurious box
    # in frame 3 (FP); missed in frame 8 (FN)
    gt = rows({t: [(1, 10 + 2 * t, 10)] for t in range(1, 11)})
    pred = {t: [(1 if t <= 5 else 2, 10 + 2 * t, 10)] for t in range(1, 11) if t != 8}
    pred[3] = pred[3] + [(5, 300, 300)]
    r = clear_identity(gt, rows(pred))
    assert (r['TP'], r['FN'], r['FP'], r['IDSW']) == (9, 1, 1, 1)
    assert abs(r['MOTA'] - (9 - 1 - 1) / 10) < 1e-12
    # identity: best single tracker id covers 5 of the 10 gt detections (id 1: frames 1-5; id 2: 4 frames)
    assert (r['IDTP'], r['IDFN'], r['IDFP']) == (5, 5, 5)
    assert abs(r['IDF1'] - 5 / (5 + 2.5 + 2.5)) < 1e-12


def test_iou_threshold_and_empty_inputs():
    a = np.array([[0, 0, 10, 10.]])
    assert abs(box_iou_xywh(a, np.array([[5, 0, 10, 10.]]))[0, 0] - 1 / 3) < 1e-12
    gt = [[1, 1, 0, 0, 10, 10, 1]]
    assert clear_identity(gt, [[1, 1, 5, 0, 10, 10, 1]])['TP'] == 0       # IoU 1/3 < 0.5: FP + FN
    assert clear_identity(gt, [[1, 1, 2, 0, 10, 10, 1]])['TP'] == 1       # IoU 2/3
    r = clear_identity(gt, [])
    assert (r['TP'], r['FN'], r['FP']) == (0, 1, 0) and r['MOTA'] == 0.0
    r = clear_identity([], [[1, 1, 0, 0, 10, 10, 1]])
    assert (r['FP'], r['FN']) == (1, 0)


def test_drone_metrics_depth_filter_and_files(tmp_path):
    m = MOTDroneMetrics(depth_thr=80)
    for t in range(3):
        s = TrackDataSample(dict(frame_id=t))
        s.pred_track_instances = InstanceData(
            bboxes=torch.tensor([[10. + t, 10, 30 + t, 30], [200., 200, 240, 240], [400., 50, 420, 70]]),
            scores=torch.tensor([0.9, 0.8, 0.7]), labels=torch.zeros(3, dtype=torch.long),
            depth=torch.tensor([20.0, 95.0, -1.0]), instances_id=torch.tensor([0, 1, 2]))
        gt = [dict(instance_id=5, bbox=[10. + t, 10, 30 + t, 30], mot_conf=1, category_id=1, visibility=1.0,
                   location=[0, 0, 20.0]),
              dict(instance_id=6, bbox=[200., 200, 240, 240], mot_conf=1, category_id=1, visibility=1.0,
                   location=[0, 0, 95.0])]   # beyond depth_thr: dropped from the GT too
        m.process('seq0', s, gt)
    res = m.evaluate()
    c = res['combined']
    assert (c['TP'], c['FP'], c['FN'], c['IDSW']) == (3, 0, 0, 0) and c['MOTA'] == 1.0 and c['IDF1'] == 1.0
    m.write_motchallenge(str(tmp_path))
    pred_lines = open(os.path.join(tmp_path, 'pred', 'seq0.txt')).read().strip().split('\n')
    assert pred_lines[0] == '1,0,10.000,10.000,20.000,20.000,0.900,-1,-1,-1' and len(pred_lines) == 3
    gt_lines = open(os.path.join(tmp_path, 'gt', 'seq0.txt')).read().strip().split('\n')
    assert gt_lines[0] == '1,5,10,10,20,20,1,1,1.00000' and len(gt_lines) == 3


# ---- multi-rank evaluation: whole videos per rank, gather only at the end (reference mot_drone_metrics.py:336-358) ----
def _fill(metrics, video, seed):
    """A small deterministic video: 3 objects, an id switch and a miss, so the scores are not trivial."""
    rng = np.random.RandomState(seed)
    for t in range(6):
        boxes = torch.tensor([[10. + 3 * t, 10, 40 + 3 * t, 40], [200., 100 + 2 * t, 240, 140 + 2 * t],
                              [400., 50, 430, 80]]) + float(rng.randint(0, 3))
        ids = torch.tensor([0, 1, 2 if t < 3 else 7])          # object 2 changes its id at t = 3
        keep = [0, 1, 2] if t != 4 else [0, 2]                 # object 1 missed at t = 4
        s = TrackDataSample(dict(frame_id=t))
        s.pred_track_instances = InstanceData(bboxes=boxes[keep], scores=torch.full((len(keep),), 0.9),
                                              labels=torch.zeros(len(keep), dtype=torch.long),
                                              depth=torch.full((len(keep),), 20.0), instances_id=ids[keep])
        gt = [dict(instance_id=k, bbox=boxes[k].tolist(), location=[0, 0, 20.0]) for k in range(3)]
        metrics.process(video, s, gt)


def _metrics_worker(rank, world, port, q):
    import torch.distributed as dist
    from stereotracking_amd import dist as sdist
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    names = [f'seq{i}' for i in range(5)]
    m = MOTDroneMetrics(depth_thr=80)
    for i in sdist.shard_videos(len(names)):         # this rank's contiguous block of whole videos
        _fill(m, names[i], seed=i)
    res = m.evaluate()                                # barrier + all_gather_object + rank-0 evaluate + broadcast
    q.put((rank, res['combined'], sorted(res['per_video']), sorted(m.pred)))
    dist.barrier()
    dist.destroy_process_group()


def test_metrics_gather_world2_gloo_equals_single_process():
    import multiprocessing as mp
    import socket
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_metrics_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    ref = MOTDroneMetrics(depth_thr=80)
    for i in range(5):
        _fill(ref, f'seq{i}', seed=i)
    want = ref.evaluate()
    assert want['combined']['IDSW'] == 5 and want['combined']['FN'] == 5       # one of each per video
    for rank, combined, videos, held in got:
        assert combined == want['combined']                    # ONE set of scores, identical on every rank
        assert videos == sorted(want['per_video']) and held == videos   # after the gather every rank holds all rows


def test_prediction_results_csv_side_effect(tmp_path):
    """reference mmtrack/utils/collect_results.py:1-44: header once, one row per track, file removed on decoration."""
    from stereotracking_amd.mot import CSV_HEADER, append_prediction_results, save_prediction_results
    path = str(tmp_path / 'results.csv')
    open(path, 'w').write('stale')

    def sample(t):
        s = TrackDataSample(dict(frame_id=t))
        s.pred_track_instances = InstanceData(bboxes=torch.tensor([[1., 2, 3, 4], [5., 6, 7, 8]]),
                                              scores=torch.tensor([0.5, 0.25]), labels=torch.zeros(2, dtype=torch.long),
                                              depth=torch.tensor([10.0, -1.0]), gt_depth=torch.tensor([10.0, -1.0]),
                                              instances_id=torch.tensor([3, 4]))
        return s

    @save_prediction_results(path)
    def predict(n0, n):
        return [sample(t) for t in range(n0, n0 + n)]

    assert not os.path.exists(path)                   # deleted when the decorator was applied
    predict(0, 2)
    predict(2, 1)
    lines = open(path).read().strip().split('\n')
    assert lines[0].split(',') == CSV_HEADER and len(lines) == 1 + 3 * 2
    assert lines[1].split(',')[:7] == ['0', '3', '0', '1.0', '2.0', '3.0', '4.0']
    # depth / gt_depth / score are plain floats (the reference writes a Python list's values), never `tensor(10.)`
    assert [float(v) for v in lines[1].split(',')[7:]] == [10.0, 10.0, 0.5]
    assert [float(v) for v in lines[2].split(',')[7:]] == [-1.0, -1.0, 0.25]
    assert 'tensor' not in open(path).read()
    assert lines[-1].split(',')[0] == '2'
    import pytest
    with pytest.raises(ValueError):
        append_prediction_results(str(tmp_path / 'x.txt'), [])


# ---- HOTA / DetA / AssA (reference mot_drone_metrics.py:83-88, 291-295: the default metric list starts with HOTA) and
# ---- the CLEAR extras Frag / MT / ML (:306-308), on cases whose answers follow from the definitions by hand ----------
def test_hota_perfect_tracking_is_one_at_every_threshold():
    from stereotracking_amd.metrics import HOTA_ALPHAS, hota
    gt = rows({t: [(1, 10 + t, 10), (2, 100, 50 + t)] for t in range(1, 11)})
    pred = rows({t: [(7, 10 + t, 10), (9, 100, 50 + t)] for t in range(1, 11)})
    h = hota(gt, pred)
    assert len(HOTA_ALPHAS) == 19 and abs(HOTA_ALPHAS[0] - 0.05) < 1e-12 and abs(HOTA_ALPHAS[-1] - 0.95) < 1e-12
    for k in ('HOTA', 'DetA', 'AssA', 'LocA', 'DetRe', 'DetPr', 'AssRe', 'AssPr'):
        assert np.allclose(h[k], 1.0), k
    assert np.all(h['HOTA_TP'] == 20) and not h['HOTA_FN'].any() and not h['HOTA_FP'].any()


def test_hota_identity_switch_halves_the_association_score():
    """One gt track of 10 frames, exact boxes, predicted as id 1 (frames 1-5) then id 2 (6-10): every detection is found
    (DetA = 1); a TP of id pair (g, 1) has TPA = 5, FNA = 5 (g's detections carried by the other id), FPA = 0, so
    A(c) = 5 / 10 for all ten TPs: AssA = 0.5, HOTA = sqrt(0.5) at every threshold; AssPr = 1, AssRe = 0.5."""
    from stereotracking_amd.metrics import hota
    gt = rows({t: [(1, 10 + 2 * t, 10)] for t in range(1, 11)})
    pred = rows({t: [(1 if t <= 5 else 2, 10 + 2 * t, 10)] for t in range(1, 11)})
    h = hota(gt, pred)
    assert np.allclose(h['DetA'], 1.0) and np.allclose(h['AssA'], 0.5) and np.allclose(h['HOTA'], np.sqrt(0.5))
    assert np.allclose(h['AssPr'], 1.0) and np.allclose(h['AssRe'], 0.5)


def test_hota_missed_half_and_localisation_threshold():
    from stereotracking_amd.metrics import HOTA_ALPHAS, hota
    # found in 5 of 10 frames with the right id: DetA = 5 / 10; A(c) = 5 / (5 + 5 FNA) = 0.5; HOTA = 0.5
    gt = rows({t: [(1, 10 + 2 * t, 10)] for t in range(1, 11)})
    h = hota(gt, rows({t: [(4, 10 + 2 * t, 10)] for t in range(1, 6)}))
    assert np.allclose(h['DetA'], 0.5) and np.allclose(h['AssA'], 0.5) and np.allclose(h['HOTA'], 0.5)
    assert np.allclose(h['DetPr'], 1.0) and np.allclose(h['DetRe'], 0.5)
    # boxes of 20 x 20 shifted by 5 px: IoU = 15 * 20 / (2 * 400 - 300) = 0.6 -> matched for alpha <= 0.6 (12 of the 19
    # thresholds), unmatched (FP + FN, score 0) beyond: mean HOTA = 12 / 19, LocA = 0.6 where anything matched
    pred = rows({t: [(4, 15 + 2 * t, 10)] for t in range(1, 11)})
    h = hota(gt, pred)
    exp = (HOTA_ALPHAS <= 0.6 + 1e-9).astype(float)
    assert exp.sum() == 12 and np.allclose(h['HOTA'], exp) and np.allclose(h['DetA'], exp) and np.allclose(h['AssA'], exp)
    assert np.allclose(h['LocA'][:12], 0.6) and abs(h['HOTA'].mean() - 12 / 19) < 1e-12
    # empty sides
    h = hota(gt, [])
    assert np.all(h['HOTA_FN'] == 10) and not h['HOTA'].any()
    h = hota([], pred)
    assert np.all(h['HOTA_FP'] == 10) and not h['HOTA'].any()


def test_hota_two_objects_swapped_ids_midway():
    """Two gt tracks far apart, 8 frames; the tracker exchanges its two ids after frame 4.  Detection is perfect.  Every
    TP has TPA = 4, FNA = 4, FPA = 4: A(c) = 4 / 12, AssA = 1 / 3, HOTA = sqrt(1 / 3); CLEAR counts 2 id switches."""
    from stereotracking_amd.metrics import hota
    gt = rows({t: [(1, 10 + t, 10), (2, 300, 50 + t)] for t in range(1, 9)})
    pred = rows({t: [(1 if t <= 4 else 2, 10 + t, 10), (2 if t <= 4 else 1, 300, 50 + t)] for t in range(1, 9)})
    h = hota(gt, pred)
    assert np.allclose(h['DetA'], 1.0) and np.allclose(h['AssA'], 1 / 3) and np.allclose(h['HOTA'], np.sqrt(1 / 3))
    r = clear_identity(gt, pred)
    assert r['IDSW'] == 2 and r['MOTA'] == (16 - 0 - 2) / 16 and r['IDF1'] == 0.5


def test_clear_frag_mt_pt_ml():
    """gt 1: 10 frames, tracked in frames 1-4 and 7-10 (two segments: Frag 1, ratio 0.8 -> PT, not MT: MT needs > 0.8);
    gt 2: 10 frames, tracked in 9 (ratio 0.9 -> MT), one gap in the middle (Frag 1); gt 3: 10 frames, tracked once (ML);
    gt 4: always tracked (MT) - so that no frame is without predictions (TrackEval skips such a frame entirely, it does not
    end a tracked segment; restated the same way here)."""
    gt = rows({t: [(1, 10 + t, 10), (2, 200, 50 + t), (3, 400 + t, 300), (4, 600, 400)] for t in range(1, 11)})
    pred = {}
    for t in range(1, 11):
        objs = [(14, 600, 400)]
        if t not in (5, 6):
            objs.append((11, 10 + t, 10))
        if t != 5:
            objs.append((12, 200, 50 + t))
        if t == 3:
            objs.append((13, 400 + t, 300))
        pred[t] = objs
    r = clear_identity(gt, rows(pred))
    assert (r['MT'], r['PT'], r['ML'], r['Frag']) == (2, 1, 1, 2)
    assert (r['TP'], r['FN'], r['FP'], r['IDSW']) == (8 + 9 + 1 + 10, 2 + 1 + 9, 0, 0)


def test_drone_metrics_report_the_reference_key_set():
    """MOTDroneMetrics(metric=...) mirrors the reference constructor (mot_drone_metrics.py:83-103): default
    ['HOTA', 'CLEAR', 'Identity'], an unknown name raises KeyError; the combined result carries the keys the reference
    logs (:291-320); two videos combine by count (HOTA: TP-weighted association, summed detection counts)."""
    import pytest
    with pytest.raises(KeyError):
        MOTDroneMetrics(metric=['HOTA', 'VACE'])
    m = MOTDroneMetrics(ignore_depth=True)
    assert m.metrics == ['HOTA', 'CLEAR', 'Identity']
    # video a: perfect, 10 detections; video b: the id-switch case above (10 detections, AssA 0.5)
    m.gt['a'] = rows({t: [(1, 10 + t, 10)] for t in range(1, 11)})
    m.pred['a'] = rows({t: [(3, 10 + t, 10)] for t in range(1, 11)})
    m.gt['b'] = rows({t: [(1, 10 + 2 * t, 10)] for t in range(1, 11)})
    m.pred['b'] = rows({t: [(1 if t <= 5 else 2, 10 + 2 * t, 10)] for t in range(1, 11)})
    out = m.evaluate(distributed=False)
    c = out['combined']
    for k in ('HOTA', 'AssA', 'DetA', 'MOTA', 'MOTP', 'IDSW', 'TP', 'FP', 'FN', 'Frag', 'MT', 'ML', 'IDF1', 'IDTP', 'IDFN',
              'IDFP', 'IDP', 'IDR'):
        assert k in c, k
    assert abs(out['per_video']['a']['HOTA'] - 1.0) < 1e-12 and abs(out['per_video']['b']['HOTA'] - np.sqrt(0.5)) < 1e-12
    # combined: DetA = 1 (20 TP, nothing else); AssA = (10 * 1 + 10 * 0.5) / 20 = 0.75
    assert abs(c['DetA'] - 1.0) < 1e-12 and abs(c['AssA'] - 0.75) < 1e-12 and abs(c['HOTA'] - np.sqrt(0.75)) < 1e-12
    assert c['IDSW'] == 1 and c['MT'] == 2 and c['ML'] == 0 and c['Frag'] == 0
    m2 = MOTDroneMetrics(ignore_depth=True, metric='CLEAR')
    m2.gt, m2.pred = m.gt, m.pred
    assert 'HOTA' not in m2.evaluate(distributed=False)['combined']
