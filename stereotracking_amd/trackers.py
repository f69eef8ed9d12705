"""Depth-guided OC-SORT association (host side, CPU) — the consumer of the dense path's detection
buffer.  north_star keeps this step on the CPU; it is restated from scratch so that, fed the same
detections, it produces the same track ids as the reference.

Behavioural spec (file:line in /root/reference):
  OCSORTTracker_Disparity.track            mmtrack/models/trackers/ocsort_tracker_disparity.py:345-618
    init_track / update_track              :105-146   (+ kalman_tracker_base.py:55-76, base_tracker.py:54-113)
    vel_direction(_batch), k_step_observation, last_obs   :148-185, :267-271
    ocm_assign_ids (IoU + 0.2 * normalised velocity angle, lapjv)     :187-265
    ocr_assign_ids (last-observation IoU, lapjv)                      :273-317
    online_smooth (virtual KF updates over the lost gap)              :319-343
  pop_invalid_tracks                       kalman_tracker_base.py:78-88
  lap.lapjv(cost, extend_cost=True, cost_limit=c) is un-vendored: `lapjv_extended` calls the library's host
  solver st_lapjv_extended (csrc/lapjv.cpp: the same dense Jonker-Volgenant procedure on the (n+m)^2 extension
  with c/2 off-blocks and a zero corner), so that ties between equally good assignments resolve as in `lap`.
  mmdet.bbox_overlaps is un-vendored: restated in `bbox_overlaps` (eps = 1e-6 on the union).
"""
import ctypes as C
import math

import numpy as np
import torch

from . import _lib
from .registry import MODELS
from .structures import InstanceData


def bbox_xyxy_to_cxcyah(b):
    """mmtrack/structures/bbox/transforms.py:72-86."""
    w, h = b[:, 2] - b[:, 0], b[:, 3] - b[:, 1]
    return torch.stack([(b[:, 2] + b[:, 0]) / 2, (b[:, 3] + b[:, 1]) / 2, w / h, h], -1)


def bbox_cxcyah_to_xyxy(b):
    """mmtrack/structures/bbox/transforms.py:89-101."""
    cx, cy, ratio, h = b.split((1, 1, 1, 1), dim=-1)
    w = ratio * h
    return torch.cat([cx - w / 2.0, cy - h / 2.0, cx + w / 2.0, cy + h / 2.0], dim=-1)


def bbox_overlaps(b1, b2, eps=1e-6):
    """Pairwise IoU (T,4) x (M,4) -> (T,M), fp32; mmdet.structures.bbox.bbox_overlaps, mode='iou'."""
    a1 = (b1[:, 2] - b1[:, 0]) * (b1[:, 3] - b1[:, 1])
    a2 = (b2[:, 2] - b2[:, 0]) * (b2[:, 3] - b2[:, 1])
    lt = torch.max(b1[:, None, :2], b2[None, :, :2])
    rb = torch.min(b1[:, None, 2:], b2[None, :, 2:])
    wh = (rb - lt).clamp(min=0)
    overlap = wh[..., 0] * wh[..., 1]
    union = torch.max(a1[:, None] + a2[None, :] - overlap, overlap.new_tensor([eps]))
    return overlap / union


def lapjv_extended(cost, cost_limit):
    """lap.lapjv(cost, extend_cost=True, cost_limit=cost_limit) -> (x, y): x[i] = column matched to row i or
    -1, y[j] = row matched to column j or -1 (int32).  NaN costs (a NaN box out of extract_depth's empty-segment
    branch, ocsort_disparity.py:163-165) are undefined behaviour inside lap's solver; the library makes them
    unmatchable instead."""
    cost = np.ascontiguousarray(cost, dtype=np.float64)
    n_rows, n_cols = cost.shape
    x = np.empty(n_rows, dtype=np.int32)
    y = np.empty(n_cols, dtype=np.int32)
    _lib.check(_lib.load().st_lapjv_extended(cost.ctypes.data_as(C.c_void_p), n_rows, n_cols, float(cost_limit),
                                             x.ctypes.data_as(C.c_void_p), y.ctypes.data_as(C.c_void_p)),
               'st_lapjv_extended')
    return x, y


class _Track:
    """State of one tracklet (the reference keeps the same items in an addict.Dict)."""
    __slots__ = ('memo', 'mean', 'covariance', 'tentative', 'tracked', 'obs', 'saved', 'velocity')

    def __init__(self):
        self.memo = {}           # item name -> list of (1, ...) tensors, one per frame the track was fed
        self.mean = None
        self.covariance = None
        self.tentative = True
        self.tracked = True
        self.obs = []            # per frame: associated detection box (4,) or None
        self.saved = None        # (mean, covariance) right before the track was lost
        self.velocity = None

    @property
    def last_frame(self):
        return int(self.memo['frame_ids'][-1])


@MODELS.register_module(name=['OCSORTTracker_Disparity'])
class OCSORTTracker_Disparity:
    def __init__(self, obj_score_thr=0.3, init_track_thr=0.7, weight_iou_with_det_scores=True, match_iou_thr=0.3,
                 num_tentatives=3, vel_consist_weight=0.2, vel_delta_t=3, cmc=None, momentums=None,
                 num_frames_retain=10, reid=None, backend='native', **kwargs):
        """backend='native' (default): the whole per-frame step runs in the library's host routine st_tracker_track
        (csrc/ocsort_tracker.cpp, ~20 us per frame instead of 1-3 ms of small torch / numpy calls); 'python': the
        pure-Python restatement below.  Both are held to the oracle frame by frame (tests/test_cpu_tracker_oracle.py)."""
        if backend not in ('native', 'python'):
            raise ValueError("backend must be 'native' or 'python'")
        self.backend = backend
        self._native = None
        self.obj_score_thr = obj_score_thr
        self.init_track_thr = init_track_thr
        self.weight_iou_with_det_scores = weight_iou_with_det_scores
        self.match_iou_thr = match_iou_thr
        self.num_tentatives = num_tentatives
        self.vel_consist_weight = vel_consist_weight
        self.vel_delta_t = vel_delta_t
        self.num_frames_retain = num_frames_retain
        if momentums is not None:
            raise NotImplementedError('momentum buffers are not used by the stereo configs')
        method = cmc.get('method') if cmc is not None else None
        if method is not None:
            if method != 'glme_affine':
                raise ValueError(f"Unknown cmc method '{method}', expected 'glme_affine' or None.")
            raise NotImplementedError('Mesh-Affine CMC needs OpenCV (absent); the shipped config runs with cmc=None')
        self.reset()

    # ---- bookkeeping ---------------------------------------------------------------------------
    def reset(self):
        self.num_tracks = 0
        self.tracks = {}
        if self._native is not None:
            _lib.check(_lib.load().st_tracker_reset(self._native), 'st_tracker_reset')

    def __del__(self):
        h = getattr(self, '_native', None)
        if h is not None:
            try:
                _lib.load().st_tracker_destroy(h)
            except Exception:      # interpreter shutdown: module globals are already gone
                pass
            self._native = None

    @property
    def empty(self):
        if self.backend == 'native':
            return self._native is None or _lib.load().st_tracker_num_tracks(self._native) == 0
        return not self.tracks

    @property
    def ids(self):
        if self.backend == 'native':
            return [t['id'] for t in self.native_state()]
        return list(self.tracks.keys())

    # ---- native backend --------------------------------------------------------------------------
    def _handle(self):
        if self._native is None:
            cfg = _lib.StTrackerConfig(C.sizeof(_lib.StTrackerConfig), float(self.obj_score_thr),
                                       float(self.init_track_thr), int(bool(self.weight_iou_with_det_scores)),
                                       float(self.match_iou_thr), int(self.num_tentatives),
                                       float(self.vel_consist_weight), int(self.vel_delta_t),
                                       int(self.num_frames_retain))
            h = C.c_void_p()
            _lib.check(_lib.load().st_tracker_create(C.byref(cfg), C.byref(h)), 'st_tracker_create')
            self._native = h
        return self._native

    def native_state(self):
        """Live tracks of the native backend in creation order: [{id, mean (8,), covariance (8,8), tentative,
        tracked, last_frame}] (tests / checkpointing)."""
        lib, h = _lib.load(), self._handle()
        out = []
        for i in range(lib.st_tracker_num_tracks(h)):
            tid = C.c_int64()
            mean, cov = np.zeros(8), np.zeros((8, 8))
            te, tr, lf = C.c_int(), C.c_int(), C.c_int()
            _lib.check(lib.st_tracker_get_track(h, i, C.byref(tid), mean.ctypes.data_as(C.c_void_p),
                                                cov.ctypes.data_as(C.c_void_p), C.byref(te), C.byref(tr),
                                                C.byref(lf)), 'st_tracker_get_track')
            out.append(dict(id=tid.value, mean=mean, covariance=cov, tentative=bool(te.value), tracked=bool(tr.value),
                            last_frame=lf.value))
        return out

    def track_records(self, frame_ids, records):
        """A CHUNK of frames in one native call (st_tracker_track_records): `records` = host float32 (F, M + 1, 13)
        frame records (pipeline.pack_detections(scaled='both')), `frame_ids` = F ints.
        -> (rows (F, M, 8) float32 [unscaled box, score, label, depth, scale], ids (F, M) int64, counts (F,) int32;
        -1 = padding frame).  Native backend only."""
        if self.backend != 'native':
            raise RuntimeError("track_records needs backend='native'")
        F, R, Cc = records.shape
        rec = np.ascontiguousarray(records, dtype=np.float32) if isinstance(records, np.ndarray) else records.numpy()
        fid = np.ascontiguousarray(frame_ids, dtype=np.int32)
        rows = np.zeros((F, R - 1, 8), np.float32)     # rows past a frame's count stay zero (they are sliced off)
        ids = np.zeros((F, R - 1), np.int64)
        counts = np.empty(F, np.int32)
        rc = _lib.load().st_tracker_track_records(self._handle(), fid.ctypes.data_as(C.c_void_p),
                                                  rec.ctypes.data_as(C.c_void_p), F, R, Cc,
                                                  rows.ctypes.data_as(C.c_void_p), ids.ctypes.data_as(C.c_void_p),
                                                  R - 1, counts.ctypes.data_as(C.c_void_p))
        if rc == -4:   # ST_ERR_WORKSPACE: a frame kept more boxes than the record holds
            from .dist import DetectionOverflow
            raise DetectionOverflow(_lib.load().st_last_error().decode() + '; build the model with a larger max_det')
        _lib.check(rc, 'st_tracker_track_records')
        self.num_tracks = int(_lib.load().st_tracker_next_id(self._native))
        return rows, ids, counts

    def _track_native(self, data_sample):
        det = data_sample.pred_det_instances
        dev = det.bboxes.device
        n = len(det.bboxes)
        rows = np.empty((n, 8), np.float32)
        rows[:, 0:4] = det.bboxes.detach().cpu().numpy()
        rows[:, 4] = det.scores.detach().cpu().numpy()
        labels = det.labels.detach().cpu()
        rows[:, 5] = labels.numpy()
        rows[:, 6] = det.depth.detach().cpu().numpy()
        rows[:, 7] = det.scales.detach().cpu().numpy()
        out = np.empty((n, 8), np.float32)
        ids = np.empty(n, np.int64)
        k = C.c_int()
        frame_id = int(data_sample.metainfo.get('frame_id', -1))
        _lib.check(_lib.load().st_tracker_track(self._handle(), frame_id, rows.ctypes.data_as(C.c_void_p), n,
                                                out.ctypes.data_as(C.c_void_p), ids.ctypes.data_as(C.c_void_p), n,
                                                C.byref(k)), 'st_tracker_track')
        k = k.value
        self.num_tracks = int(_lib.load().st_tracker_next_id(self._native))
        t = torch.from_numpy(out[:k])
        res = InstanceData()
        res['bboxes'] = t[:, 0:4].contiguous().to(dev)
        res['labels'] = t[:, 5].to(labels.dtype).to(dev)
        res['scores'] = t[:, 4].contiguous().to(dev)
        res['scales'] = t[:, 7].contiguous().to(dev)
        res['depth'] = t[:, 6].contiguous().to(dev)
        res.instances_id = torch.from_numpy(ids[:k]).to(labels.dtype).to(dev)
        return res

    @property
    def confirmed_ids(self):
        return [i for i, t in self.tracks.items() if not t.tentative]

    @property
    def unconfirmed_ids(self):
        return [i for i, t in self.tracks.items() if t.tentative]

    @staticmethod
    def _last_obs(track):
        for box in reversed(track.obs):
            if box is not None:
                return box
        return None

    def _k_step_observation(self, track):
        n = len(track.obs)
        if n == 0:
            return torch.tensor((-1., -1., -1., -1.))
        if n > self.vel_delta_t and track.obs[n - 1 - self.vel_delta_t] is not None:
            return track.obs[n - 1 - self.vel_delta_t]
        return self._last_obs(track)

    @staticmethod
    def _vel_direction(b1, b2):
        if b1.sum() < 0 or b2.sum() < 0:
            return torch.tensor((-1., -1.))
        cx1, cy1 = (b1[0] + b1[2]) / 2.0, (b1[1] + b1[3]) / 2.0
        cx2, cy2 = (b2[0] + b2[2]) / 2.0, (b2[1] + b2[3]) / 2.0
        speed = torch.stack([cy2 - cy1, cx2 - cx1])
        return speed / (torch.sqrt(speed[0] ** 2 + speed[1] ** 2) + 1e-6)

    @staticmethod
    def _vel_direction_batch(b1, b2):
        cx1, cy1 = (b1[:, 0] + b1[:, 2]) / 2.0, (b1[:, 1] + b1[:, 3]) / 2.0
        cx2, cy2 = (b2[:, 0] + b2[:, 2]) / 2.0, (b2[:, 1] + b2[:, 3]) / 2.0
        speed = torch.stack((cy2[None, :] - cy1[:, None], cx2[None, :] - cx1[:, None]), dim=-1)
        norm = torch.sqrt(speed[..., 0] ** 2 + speed[..., 1] ** 2) + 1e-6
        return speed / norm[..., None]

    # ---- per-object memo update ---------------------------------------------------------------------
    def _feed(self, tid, items, frame_id):
        """One detection assigned to track `tid` this frame: start the track or extend it."""
        box = items['bboxes']
        new = tid not in self.tracks
        if new:
            t = self.tracks[tid] = _Track()
            for k, v in items.items():
                t.memo[k] = [v[None]]
            t.memo['frame_ids'] = [frame_id]
            t.tentative = frame_id != 0          # tracks born on the first frame are confirmed at once
            t.mean, t.covariance = self.kf.initiate(bbox_xyxy_to_cxcyah(box[None])[0].numpy().astype(np.float64))
            t.obs = [box]
            t.tracked = True
            t.saved = None
            t.velocity = torch.tensor((-1., -1.))
            return
        t = self.tracks[tid]
        for k, v in items.items():
            t.memo[k].append(v[None])
        t.memo['frame_ids'].append(frame_id)
        if t.tentative and len(t.memo['bboxes']) >= self.num_tentatives:
            t.tentative = False
        t.mean, t.covariance = self.kf.update(t.mean, t.covariance,
                                              bbox_xyxy_to_cxcyah(box[None])[0].numpy().astype(np.float64))
        t.tracked = True
        t.obs.append(box)
        t.velocity = self._vel_direction(self._k_step_observation(t), box)

    def _pop_invalid(self, frame_id):
        dead = [i for i, t in self.tracks.items()
                if frame_id - t.last_frame >= self.num_frames_retain or (t.tentative and t.last_frame != frame_id)]
        for i in dead:
            del self.tracks[i]

    # ---- association stages ----------------------------------------------------------------------------
    def _assign(self, dists):
        if dists.size > 0:
            return lapjv_extended(dists, 1 - self.match_iou_thr)
        return (np.full(dists.shape[0], -1, np.int32), np.full(dists.shape[1], -1, np.int32))

    def ocm_assign_ids(self, ids, det_bboxes, det_scores):
        """Observation-centric momentum: cost = 1 - IoU(KF prediction, det) + w * normalised angle between the
        track's velocity direction and the direction (k-step-old observation -> det)."""
        means = np.zeros((0, 4))
        for i in ids:
            means = np.concatenate((means, self.tracks[i].mean[:4][None]), axis=0)
        track_boxes = bbox_cxcyah_to_xyxy(torch.from_numpy(means).to(det_bboxes))
        ious = bbox_overlaps(track_boxes, det_bboxes[:, :4])
        if self.weight_iou_with_det_scores:
            ious = ious * det_scores[None]
        dists = (1 - ious).numpy()
        if len(ids) > 0 and len(det_bboxes) > 0:
            vel = torch.stack([self.tracks[i].velocity.float() for i in ids])
            kobs = torch.stack([self._k_step_observation(self.tracks[i]).float() for i in ids])
            valid = (vel.sum(dim=1) != -2) & (kobs.sum(dim=1) != -4)
            to_match = self._vel_direction_batch(kobs[:, :4], det_bboxes[:, :4])
            cos = (to_match * vel[:, None, :]).sum(dim=-1).clamp(min=-1, max=1)
            norm_angle = (torch.acos(cos) - math.pi / 2.) / math.pi
            norm_angle = norm_angle * valid[:, None].int()
            dists = dists + norm_angle.numpy() * self.vel_consist_weight
        return self._assign(dists)

    def ocr_assign_ids(self, track_obs, det_bboxes, det_scores):
        """Observation-centric recovery: IoU-only matching of last observations to leftover detections."""
        ious = bbox_overlaps(track_obs[:, :4], det_bboxes[:, :4])
        if self.weight_iou_with_det_scores:
            ious = ious * det_scores[None]
        return self._assign((1 - ious).numpy())

    def _online_smooth(self, track, new_box):
        """Re-run the KF over a linear interpolation of the gap the track was lost for."""
        last = self._last_obs(track)[:4]
        gap = 0
        for b in reversed(track.obs):
            if b is not None:
                break
            gap += 1
        step = (new_box[:4] - last) / (gap + 1)
        track.mean, track.covariance = track.saved
        for i in range(gap):
            virt = bbox_xyxy_to_cxcyah((last + (i + 1) * step)[None])[0].numpy()
            track.mean, track.covariance = self.kf.update(track.mean, track.covariance, virt)

    # ---- main entry ---------------------------------------------------------------------------------------
    def track(self, model, img, feats, data_sample, data_preprocessor=None, rescale=False, **kwargs):
        if self.backend == 'native':
            return self._track_native(data_sample)
        det = data_sample.pred_det_instances
        dev = det.bboxes.device
        fields = {k: det[k].detach().cpu() for k in ('bboxes', 'labels', 'scores', 'scales', 'depth')}
        frame_id = data_sample.metainfo.get('frame_id', -1)
        if frame_id == 0:
            self.reset()
        if not hasattr(self, 'kf'):
            self.kf = model.motion

        def take(sel):
            return {k: v[sel] for k, v in fields.items()}

        if self.empty or fields['bboxes'].size(0) == 0:
            fields = take(fields['scores'] > self.init_track_thr)
            n_new = fields['bboxes'].size(0)
            ids = torch.arange(self.num_tracks, self.num_tracks + n_new).to(fields['labels'])
            self.num_tracks += n_new
        else:
            b = fields['bboxes']
            keep = (fields['scores'] > self.obj_score_thr) & ((b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1]) > 100)
            cand = take(keep)                               # detections entering association
            cand_ids = torch.full((cand['bboxes'].size(0),), -1, dtype=fields['labels'].dtype)

            # KF predict for confirmed tracks (velocity of h zeroed while lost)
            confirmed = self.confirmed_ids
            for i in confirmed:
                t = self.tracks[i]
                if t.last_frame != frame_id - 1:
                    t.mean[7] = 0
                if t.tracked:
                    t.saved = (t.mean, t.covariance)
                t.mean, t.covariance = self.kf.predict(t.mean, t.covariance)

            def split(pool, pool_ids, det_to_track, track_ids):
                """apply one assignment: returns (matched pool, matched ids, rest pool, rest ids)"""
                hit = torch.from_numpy(det_to_track > -1)
                pool_ids = pool_ids.clone()
                if hit.any():
                    pool_ids[hit] = torch.tensor(track_ids, dtype=pool_ids.dtype)[
                        torch.from_numpy(det_to_track[det_to_track > -1].astype(np.int64))]
                return ({k: v[hit] for k, v in pool.items()}, pool_ids[hit],
                        {k: v[~hit] for k, v in pool.items()}, pool_ids[~hit])

            def cat(a, bb):
                return {k: torch.cat((a[k], bb[k]), dim=0) for k in a}

            # stage 1: confirmed tracks (OCM)
            _, d2t = self.ocm_assign_ids(confirmed, cand['bboxes'], cand['scores'])
            matched, matched_ids, rest, rest_ids = split(cand, cand_ids, d2t, confirmed)
            # stage 2: tentative tracks (OCM)
            tentative = self.unconfirmed_ids
            _, d2t = self.ocm_assign_ids(tentative, rest['bboxes'], rest['scores'])
            m2, m2_ids, rest, rest_ids = split(rest, rest_ids, d2t, tentative)
            matched, matched_ids = cat(matched, m2), torch.cat((matched_ids, m2_ids))
            # stage 3: observation-centric recovery on every still-unmatched track
            all_ids = list(self.tracks.keys())
            matched_set = set(matched_ids.tolist())
            lost = [i for i in all_ids if i not in matched_set]
            if lost:
                last_obs = torch.stack([self._last_obs(self.tracks[i]) for i in lost])
                _, d2t = self.ocr_assign_ids(last_obs, rest['bboxes'], rest['scores'])
                m3, m3_ids, rest, rest_ids = split(rest, rest_ids, d2t, lost)
                matched, matched_ids = cat(matched, m3), torch.cat((matched_ids, m3_ids))
            # re-found tracks: smooth the KF over the gap; unmatched tracks: mark lost
            for i in range(len(matched_ids)):
                t = self.tracks[int(matched_ids[i])]
                if not t.tracked:
                    self._online_smooth(t, matched['bboxes'][i])
            matched_set = set(matched_ids.tolist())
            for i in all_ids:
                if i not in matched_set:
                    self.tracks[i].tracked = False
                    self.tracks[i].obs.append(None)
            fields = cat(matched, rest)
            ids = torch.cat((matched_ids, rest_ids))
            fresh = ids == -1
            n_new = int(fresh.sum())
            ids[fresh] = torch.arange(self.num_tracks, self.num_tracks + n_new).to(ids)
            self.num_tracks += n_new

        for j in range(len(ids)):
            self._feed(int(ids[j]), {k: v[j] for k, v in fields.items()}, frame_id)
        self._pop_invalid(frame_id)

        out = InstanceData()
        for k in ('bboxes', 'labels', 'scores', 'scales', 'depth'):
            out[k] = fields[k].to(dev)
        out.instances_id = ids.to(dev)
        return out
