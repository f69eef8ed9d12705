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
