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
