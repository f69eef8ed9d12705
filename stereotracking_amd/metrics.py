"""MOTChallenge writer + CLEAR-MOT / Identity metrics for the AirDrone evaluation.

Behavioural spec: reference mmtrack/evaluation/metrics/mot_drone_metrics.py
  process()                 :155-221  per-frame rows, depth-range filters
                                      (pred kept iff depth_thr >= depth > 0, gt iff location[-1] <= depth_thr)
  _save_one_video_gts_preds :223-253  MOTChallenge txt formats
  compute_metrics           :255-333  the metric list (default ['HOTA', 'CLEAR', 'Identity'], :83-88) and the keys reported:
                                      HOTA / AssA / DetA; MOTA MOTP IDSW TP FP FN Frag MT ML; IDF1 IDTP IDFN IDFP IDP IDR
The reference delegates the scores to TrackEval (un-vendored, absent here): CLEAR (MOTA, MOTP, IDSW, Frag, MT, ML),
Identity (IDF1, IDP, IDR) and HOTA (HOTA, DetA, AssA over the localisation thresholds 0.05 .. 0.95) are restated from
their published definitions (Bernardin & Stiefelhagen 2008; Ristani et al. 2016; Luiten et al., IJCV 2021) the way
TrackEval computes them [upstream-memory]: CLEAR - per-frame Hungarian matching on IoU >= 0.5 with priority for the
previous frame's match; Identity - one global bipartite matching; HOTA - per-frame Hungarian matching that maximises
(global alignment score x IoU), thresholded per alpha, association accuracy per matched pair from the id-pair counts.
Parity is UNPINNED against TrackEval itself (absent); the known-answer tests are hand-derived from the definitions.
"""
import os
from collections import defaultdict

import numpy as np
from scipy.optimize import linear_sum_assignment


def box_iou_xywh(a, b):
    """IoU of (n,4) and (m,4) boxes in MOTChallenge x,y,w,h form."""
    if len(a) == 0 or len(b) == 0:
        return np.zeros((len(a), len(b)))
    ax2, ay2 = a[:, 0] + a[:, 2], a[:, 1] + a[:, 3]
    bx2, by2 = b[:, 0] + b[:, 2], b[:, 1] + b[:, 3]
    iw = np.clip(np.minimum(ax2[:, None], bx2[None]) - np.maximum(a[:, None, 0], b[None, :, 0]), 0, None)
    ih = np.clip(np.minimum(ay2[:, None], by2[None]) - np.maximum(a[:, None, 1], b[None, :, 1]), 0, None)
    inter = iw * ih
    union = (a[:, 2] * a[:, 3])[:, None] + (b[:, 2] * b[:, 3])[None] - inter
    return np.where(union > 0, inter / np.maximum(union, 1e-12), 0.0)


def clear_identity(gt_rows, pred_rows, iou_thr=0.5):
    """gt_rows / pred_rows: arrays of (frame, id, x, y, w, h, ...).  Returns a dict with the CLEAR counts
    (TP, FN, FP, IDSW, MOTA, MOTP) and Identity scores (IDTP, IDFN, IDFP, IDF1, IDP, IDR)."""
    gt_rows = np.asarray(gt_rows, dtype=np.float64)
    pred_rows = np.asarray(pred_rows, dtype=np.float64)
    if gt_rows.size == 0:
        gt_rows = np.zeros((0, 6))
    if pred_rows.size == 0:
        pred_rows = np.zeros((0, 6))
    gt_ids = {v: i for i, v in enumerate(sorted(set(gt_rows[:, 1].astype(int).tolist())))}
    tr_ids = {v: i for i, v in enumerate(sorted(set(pred_rows[:, 1].astype(int).tolist())))}
    frames = sorted(set(gt_rows[:, 0].astype(int).tolist()) | set(pred_rows[:, 0].astype(int).tolist()))
    ng, nt = len(gt_ids), len(tr_ids)
    tp = fn = fp = idsw = 0
    motp_sum = 0.0
    prev_tracker = np.full(ng, np.nan)       # last tracker id matched to each gt (ever)
    prev_step_tracker = np.full(ng, np.nan)  # tracker id matched in the previous frame
    potential = np.zeros((ng, nt))
    gt_count, tr_count = np.zeros(ng), np.zeros(nt)
    gt_frames, gt_matched, gt_frag = np.zeros(ng), np.zeros(ng), np.zeros(ng)   # MT / PT / ML and Frag bookkeeping
    eps = np.finfo(float).eps
    for f in frames:
        g = gt_rows[gt_rows[:, 0].astype(int) == f]
        p = pred_rows[pred_rows[:, 0].astype(int) == f]
        gi = np.array([gt_ids[int(v)] for v in g[:, 1]], dtype=int)
        ti = np.array([tr_ids[int(v)] for v in p[:, 1]], dtype=int)
        gt_count[gi] += 1
        tr_count[ti] += 1
        if len(g) == 0:
            fp += len(p)
            continue
        if len(p) == 0:        # (no reset of the previous-frame match here: the frame is skipped, as TrackEval does)
            fn += len(g)
            gt_frames[gi] += 1
            continue
        sim = box_iou_xywh(g[:, 2:6], p[:, 2:6])
        rr, cc = np.nonzero(sim >= iou_thr - eps)
        potential[gi[rr], ti[cc]] += 1
        score = (ti[None, :] == prev_step_tracker[gi][:, None]) * 1000.0 + sim
        score[sim < iou_thr - eps] = 0
        rows, cols = linear_sum_assignment(-score)
        ok = score[rows, cols] > eps
        rows, cols = rows[ok], cols[ok]
        mg, mt = gi[rows], ti[cols]
        prev = prev_tracker[mg]
        idsw += int(np.sum(~np.isnan(prev) & (prev != mt)))
        prev_tracker[mg] = mt
        not_tracked_before = np.isnan(prev_step_tracker)
        prev_step_tracker[:] = np.nan
        prev_step_tracker[mg] = mt
        tracked_now = ~np.isnan(prev_step_tracker)
        gt_matched += tracked_now
        gt_frag += not_tracked_before & tracked_now         # a new tracked segment of this gt starts
        gt_frames[gi] += 1
        tp += len(rows)
        fn += len(g) - len(rows)
        fp += len(p) - len(rows)
        motp_sum += float(sim[rows, cols].sum())
    # identity: one global assignment minimising IDFN + IDFP
    fn_mat = np.zeros((ng + nt, ng + nt))
    fp_mat = np.zeros((ng + nt, ng + nt))
    fp_mat[ng:, :nt] = 1e10
    fn_mat[:ng, nt:] = 1e10
    for i in range(ng):
        fn_mat[i, :nt] = gt_count[i]
        fn_mat[i, nt + i] = gt_count[i]
    for j in range(nt):
        fp_mat[:ng, j] = tr_count[j]
        fp_mat[j + ng, j] = tr_count[j]
    fn_mat[:ng, :nt] -= potential
    fp_mat[:ng, :nt] -= potential
    r, c = linear_sum_assignment(fn_mat + fp_mat)
    idfn, idfp = float(fn_mat[r, c].sum()), float(fp_mat[r, c].sum())
    idtp = float(gt_count.sum() - idfn)
    seen = gt_frames > 0
    ratio = gt_matched[seen] / gt_frames[seen]
    mt_n = int(np.sum(ratio > 0.8))
    pt_n = int(np.sum(ratio >= 0.2)) - mt_n
    return dict(TP=tp, FN=fn, FP=fp, IDSW=idsw, Frag=int(np.sum(gt_frag[gt_frag > 0] - 1)), MT=mt_n, PT=pt_n,
                ML=ng - mt_n - pt_n,
                MOTA=(tp - fp - idsw) / max(1.0, tp + fn), MOTP=motp_sum / max(1.0, tp),
                IDTP=idtp, IDFN=idfn, IDFP=idfp,
                IDF1=idtp / max(1.0, idtp + 0.5 * idfp + 0.5 * idfn),
                IDP=idtp / max(1.0, idtp + idfp), IDR=idtp / max(1.0, idtp + idfn))


HOTA_ALPHAS = np.arange(0.05, 0.99, 0.05)     # 19 localisation thresholds 0.05 .. 0.95


def hota(gt_rows, pred_rows):
    """HOTA of ONE sequence (Luiten et al., "HOTA: A Higher Order Metric for Evaluating Multi-Object Tracking", IJCV 2021;
    reference mot_drone_metrics.py:291-295 reports the averages over alpha of HOTA / AssA / DetA).  gt_rows / pred_rows:
    arrays of (frame, id, x, y, w, h, ...); similarity = box IoU.  Returns per-alpha arrays HOTA_TP / HOTA_FN / HOTA_FP,
    AssA, AssRe, AssPr, LocA, DetA, DetRe, DetPr, HOTA (each of length 19).

    Per alpha: a TP is a (gt, prediction) pair of one frame matched by the Hungarian assignment that maximises
    global_alignment(gt id, pred id) x IoU, kept if IoU >= alpha; DetA = TP / (TP + FN + FP);
    A(c) = TPA / (TPA + FNA + FPA) for a TP c with ids (g, p): TPA = TPs carrying the same id pair, FNA / FPA = the
    other detections of g / p; AssA = mean of A over the TPs; HOTA = sqrt(DetA x AssA)."""
    gt_rows = np.asarray(gt_rows, dtype=np.float64).reshape(-1, np.shape(gt_rows)[-1] if np.size(gt_rows) else 6)
    pred_rows = np.asarray(pred_rows, dtype=np.float64).reshape(-1, np.shape(pred_rows)[-1] if np.size(pred_rows) else 6)
    A = len(HOTA_ALPHAS)
    res = {k: np.zeros(A) for k in ('HOTA_TP', 'HOTA_FN', 'HOTA_FP', 'AssA', 'AssRe', 'AssPr', 'LocA')}
    eps = np.finfo(float).eps
    if len(pred_rows) == 0 or len(gt_rows) == 0:
        res['HOTA_FN'] += len(gt_rows)
        res['HOTA_FP'] += len(pred_rows)
        res['LocA'] += 1.0
        return _hota_final(res)
    gt_ids = {v: i for i, v in enumerate(sorted(set(gt_rows[:, 1].astype(int).tolist())))}
    tr_ids = {v: i for i, v in enumerate(sorted(set(pred_rows[:, 1].astype(int).tolist())))}
    ng, nt = len(gt_ids), len(tr_ids)
    frames = sorted(set(gt_rows[:, 0].astype(int).tolist()) | set(pred_rows[:, 0].astype(int).tolist()))
    per_frame = []
    potential = np.zeros((ng, nt))
    gt_count, tr_count = np.zeros((ng, 1)), np.zeros((1, nt))
    for f in frames:          # pass 1: how well could each id pair be aligned over the whole sequence
        g = gt_rows[gt_rows[:, 0].astype(int) == f]
        p = pred_rows[pred_rows[:, 0].astype(int) == f]
        gi = np.array([gt_ids[int(v)] for v in g[:, 1]], dtype=int)
        ti = np.array([tr_ids[int(v)] for v in p[:, 1]], dtype=int)
        sim = box_iou_xywh(g[:, 2:6], p[:, 2:6])
        per_frame.append((gi, ti, sim))
        if len(gi) and len(ti):
            den = sim.sum(0)[None, :] + sim.sum(1)[:, None] - sim
            siou = np.zeros_like(sim)
            m = den > eps
            siou[m] = sim[m] / den[m]
            potential[gi[:, None], ti[None, :]] += siou
        gt_count[gi] += 1
        tr_count[0, ti] += 1
    align = potential / (gt_count + tr_count - potential)
    matches = [np.zeros((ng, nt)) for _ in range(A)]
    for gi, ti, sim in per_frame:      # pass 2: one assignment per frame, thresholded per alpha
        if len(gi) == 0:
            res['HOTA_FP'] += len(ti)
            continue
        if len(ti) == 0:
            res['HOTA_FN'] += len(gi)
            continue
        rows, cols = linear_sum_assignment(-(align[gi[:, None], ti[None, :]] * sim))
        for a, alpha in enumerate(HOTA_ALPHAS):
            ok = sim[rows, cols] >= alpha - eps
            r, c = rows[ok], cols[ok]
            res['HOTA_TP'][a] += len(r)
            res['HOTA_FN'][a] += len(gi) - len(r)
            res['HOTA_FP'][a] += len(ti) - len(r)
            if len(r):
                res['LocA'][a] += sim[r, c].sum()
                matches[a][gi[r], ti[c]] += 1
    for a in range(A):
        mc = matches[a]
        tp = max(1.0, res['HOTA_TP'][a])
        res['AssA'][a] = np.sum(mc * (mc / np.maximum(1, gt_count + tr_count - mc))) / tp
        res['AssRe'][a] = np.sum(mc * (mc / np.maximum(1, gt_count))) / tp
        res['AssPr'][a] = np.sum(mc * (mc / np.maximum(1, tr_count))) / tp
    res['LocA'] = np.maximum(1e-10, res['LocA']) / np.maximum(1e-10, res['HOTA_TP'])
    return _hota_final(res)


def _hota_final(res):
    tp, fn, fp = res['HOTA_TP'], res['HOTA_FN'], res['HOTA_FP']
    res['DetRe'] = tp / np.maximum(1, tp + fn)
    res['DetPr'] = tp / np.maximum(1, tp + fp)
    res['DetA'] = tp / np.maximum(1, tp + fn + fp)
    res['HOTA'] = np.sqrt(res['DetA'] * res['AssA'])
    return res


def hota_combine(per_sequence):
    """Sequences -> the combined result the way TrackEval's COMBINED_SEQ does: detection counts summed, the association
    and localisation scores averaged with the sequences' TP counts as weights, then DetA / HOTA from the combined values."""
    seqs = list(per_sequence)
    A = len(HOTA_ALPHAS)
    res = {k: sum((s[k] for s in seqs), np.zeros(A)) for k in ('HOTA_TP', 'HOTA_FN', 'HOTA_FP')}
    for k in ('AssA', 'AssRe', 'AssPr', 'LocA'):
        num = sum((s[k] * s['HOTA_TP'] for s in seqs), np.zeros(A))
        res[k] = np.maximum(1e-10, num) / np.maximum(1e-10, res['HOTA_TP']) if k == 'LocA' else num / np.maximum(1.0, res['HOTA_TP'])
    return _hota_final(res)


class MOTDroneMetrics:
    """Collects per-frame tracks, writes MOTChallenge files, scores them (depth-range filtered)."""

    allowed_metrics = ('HOTA', 'CLEAR', 'Identity')

    def __init__(self, depth_thr=80, ignore_depth=False, iou_thr=0.5, metric=('HOTA', 'CLEAR', 'Identity')):
        self.depth_thr, self.ignore_depth, self.iou_thr = depth_thr, ignore_depth, iou_thr
        self.metrics = [metric] if isinstance(metric, str) else list(metric)      # mot_drone_metrics.py:83-103
        for m in self.metrics:
            if m not in self.allowed_metrics:
                raise KeyError(f'metric {m} is not supported.')
        self.pred = defaultdict(list)
        self.gt = defaultdict(list)

    def process(self, video, data_sample, gt_instances=None):
        """data_sample: TrackDataSample with pred_track_instances (+ metainfo frame_id);
        gt_instances: list of dicts with instance_id, bbox (xyxy), mot_conf, category_id, visibility, location."""
        frame_id = data_sample.metainfo['frame_id']
        if gt_instances is not None:
            for ins in gt_instances:
                if self.ignore_depth or ins['location'][-1] <= self.depth_thr:
                    x1, y1, x2, y2 = ins['bbox']
                    self.gt[video].append([frame_id + 1, ins['instance_id'], x1, y1, x2 - x1, y2 - y1,
                                           ins.get('mot_conf', 1), ins.get('category_id', 1),
                                           ins.get('visibility', 1.0)])
        trk = data_sample.pred_track_instances
        ids = trk['instances_id'].cpu().numpy()
        boxes = trk['bboxes'].cpu().numpy()
        scores = trk['scores'].cpu().numpy()
        depth = trk['depth'].cpu().numpy() if 'depth' in trk else np.ones(len(ids))
        for i in range(len(ids)):
            if self.ignore_depth or self.depth_thr >= depth[i] > 0:
                b = boxes[i]
                self.pred[video].append([frame_id + 1, int(ids[i]), b[0], b[1], b[2] - b[0], b[3] - b[1], scores[i]])

    def write_motchallenge(self, out_dir):
        """pred: frame,id,x,y,w,h,score,-1,-1,-1   gt: frame,id,x,y,w,h,conf,class,visibility (reference :223-253)."""
        os.makedirs(os.path.join(out_dir, 'pred'), exist_ok=True)
        os.makedirs(os.path.join(out_dir, 'gt'), exist_ok=True)
        for video, rows in self.pred.items():
            with open(os.path.join(out_dir, 'pred', video + '.txt'), 'wt') as f:
                for t in rows:
                    f.write('%d,%d,%.3f,%.3f,%.3f,%.3f,%.3f,-1,-1,-1\n' % tuple(t[:7]))
        for video, rows in self.gt.items():
            with open(os.path.join(out_dir, 'gt', video + '.txt'), 'wt') as f:
                for t in rows:
                    f.write('%d,%d,%d,%d,%d,%d,%d,%d,%.5f\n' % tuple(t[:9]))

    def gather(self):
        """Multi-rank evaluation, the reference's way (mot_drone_metrics.py:336-358: barrier, all_gather_object of the
        per-video `seq_info`, rank 0 evaluates, broadcast_object_list of the result): ranks hold DISJOINT whole videos
        (video_sampler.py:62-70 / dist.shard_videos), so the per-video row lists are merged by key.  Afterwards every
        rank holds every video's rows.  No-op without an initialised process group."""
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
            return self
        dist.barrier()                                   # wait for all processes to complete prediction (:336)
        parts = [None] * dist.get_world_size()
        dist.all_gather_object(parts, dict(pred=dict(self.pred), gt=dict(self.gt)))   # KBs of pickled rows (:340)
        pred, gt = defaultdict(list), defaultdict(list)
        for part in parts:                               # rank order = video order of shard_videos
            for v, rows in part['pred'].items():
                if v in pred:
                    raise RuntimeError(f'video {v!r} was evaluated on more than one rank: videos shard whole')
                pred[v] = rows
            for v, rows in part['gt'].items():
                gt[v] = rows
        self.pred, self.gt = pred, gt
        return self

    def evaluate(self, distributed=True):
        """Per-video and combined (count-summed, like TrackEval's COMBINED_SEQ) scores.  With a process group (and
        `distributed`): gather first, rank 0 computes, every rank returns the broadcast result (:344-358)."""
        import torch.distributed as dist
        multi = distributed and dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
        if multi:
            self.gather()
            box = [self._evaluate_local() if dist.get_rank() == 0 else None]
            dist.broadcast_object_list(box, src=0)
            return box[0]
        return self._evaluate_local()

    def _evaluate_local(self):
        per_video = {v: clear_identity(self.gt.get(v, []), self.pred.get(v, []), self.iou_thr)
                     for v in sorted(set(self.gt) | set(self.pred))}
        tot = defaultdict(float)
        for r in per_video.values():
            for k in ('TP', 'FN', 'FP', 'IDSW', 'IDTP', 'IDFN', 'IDFP'):
                tot[k] += r[k]
            tot['motp_sum'] += r['MOTP'] * r['TP']
        combined = dict(tot)
        combined['MOTA'] = (tot['TP'] - tot['FP'] - tot['IDSW']) / max(1.0, tot['TP'] + tot['FN'])
        combined['MOTP'] = tot['motp_sum'] / max(1.0, tot['TP'])
        combined['IDF1'] = tot['IDTP'] / max(1.0, tot['IDTP'] + 0.5 * tot['IDFP'] + 0.5 * tot['IDFN'])
        combined['IDP'] = tot['IDTP'] / max(1.0, tot['IDTP'] + tot['IDFP'])
        combined['IDR'] = tot['IDTP'] / max(1.0, tot['IDTP'] + tot['IDFN'])
        for k in ('Frag', 'MT', 'PT', 'ML'):
            combined[k] = float(sum(r[k] for r in per_video.values()))
        if 'HOTA' in self.metrics:      # mot_drone_metrics.py:291-295: the averages over the 19 thresholds
            hs = {v: hota(self.gt.get(v, []), self.pred.get(v, [])) for v in per_video}
            for v, h in hs.items():
                per_video[v].update(HOTA=float(h['HOTA'].mean()), DetA=float(h['DetA'].mean()), AssA=float(h['AssA'].mean()))
            hc = hota_combine(hs.values()) if hs else _hota_final({k: np.zeros(len(HOTA_ALPHAS)) for k in
                                                                   ('HOTA_TP', 'HOTA_FN', 'HOTA_FP', 'AssA', 'AssRe', 'AssPr', 'LocA')})
            combined.update(HOTA=float(hc['HOTA'].mean()), DetA=float(hc['DetA'].mean()), AssA=float(hc['AssA'].mean()),
                            LocA=float(hc['LocA'].mean()))
        return dict(per_video=per_video, combined=combined)
