"""ORACLE (test infrastructure): statement-by-statement restatement of the reference's depth-guided OC-SORT
association step (same classes, same methods, same statement order, same tensor dtypes / devices / in-place
updates), so that it can serve as an independent checker of the product tracker (stereotracking_amd/trackers.py, a
from-scratch re-write with different data structures).  Only tests/ and tests/golden/make_golden.py may import it.

Transcribed (file:line in /root/reference), class and method names unchanged:
  BaseTracker                mmtrack/models/trackers/base_tracker.py:10-141   (reset, update, init/update_track)
  KalmanTrackerBase          mmtrack/models/trackers/kalman_tracker_base.py:19-88
  OCSORTTracker_Disparity    mmtrack/models/trackers/ocsort_tracker_disparity.py:20-618  (cmc=None path)
  KalmanFilter               mmtrack/models/motion/kalman_filter.py:38-189
  bbox_xyxy_to_cxcyah / bbox_cxcyah_to_xyxy   mmtrack/structures/bbox/transforms.py:72-101
Un-vendored third-party pieces restated from their published behaviour [upstream-memory]:
  addict.Dict                -> `Dict` below (attribute access on a dict; only what the tracker uses)
  mmdet.structures.bbox.bbox_overlaps (mode='iou', eps=1e-6)  -> `bbox_overlaps`
  lap.lapjv                  -> oracle/lapjv.py
  mmengine InstanceData      -> `Instances` (attribute bag)

PARITY PINNING: the reference holds no fixture for this path and cannot be imported here; pinned
by its own in-tree source text for everything except the three third-party pieces above ("parity unpinned").

numpy note: the reference's environment pins scipy<=1.7.3 / Python 3.9 (requirements/runtime.txt:12,
reproducibility.md:361-365), i.e. numpy 1.x VALUE-BASED casting: `python_float * np.float32_scalar` is float64
there, float32 under numpy >= 2 (NEP 50).  The Kalman filter receives the box as a float32 array
(`bbox.squeeze(0).cpu().numpy()`, kalman_tracker_base.py:58-60), so `initiate` is written with an explicit
float() on `measurement[3]`: the oracle then reproduces the reference's float64 covariance under ANY numpy.
"""
import numpy as np
import scipy.linalg
import torch

from . import lapjv as _lap


class Dict(dict):
    """addict.Dict as the tracker uses it: keys readable / writable as attributes."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v


class Instances:
    """Attribute bag standing in for mmengine.structures.InstanceData."""

    def __init__(self, **kw):
        self.__dict__.update(kw)


class Sample:
    """TrackDataSample as the tracker reads it: .metainfo and .pred_det_instances."""

    def __init__(self, frame_id, pred_det_instances):
        self.metainfo = dict(frame_id=frame_id)
        self.pred_det_instances = pred_det_instances


def bbox_xyxy_to_cxcyah(bboxes):
    """structures/bbox/transforms.py:72-86."""
    cx = (bboxes[:, 2] + bboxes[:, 0]) / 2
    cy = (bboxes[:, 3] + bboxes[:, 1]) / 2
    w = bboxes[:, 2] - bboxes[:, 0]
    h = bboxes[:, 3] - bboxes[:, 1]
    xyah = torch.stack([cx, cy, w / h, h], -1)
    return xyah


def bbox_cxcyah_to_xyxy(bboxes):
    """structures/bbox/transforms.py:89-101."""
    cx, cy, ratio, h = bboxes.split((1, 1, 1, 1), dim=-1)
    w = ratio * h
    x1y1x2y2 = [cx - w / 2.0, cy - h / 2.0, cx + w / 2.0, cy + h / 2.0]
    return torch.cat(x1y1x2y2, dim=-1)


def bbox_overlaps(bboxes1, bboxes2, eps=1e-6):
    """mmdet 3.0.0rc4 structures/bbox/bbox_overlaps.py, mode='iou', is_aligned=False [upstream-memory]."""
    rows, cols = bboxes1.size(-2), bboxes2.size(-2)
    if rows * cols == 0:
        return bboxes1.new_zeros((rows, cols))   # upstream returns uninitialised memory of this shape
    area1 = (bboxes1[..., 2] - bboxes1[..., 0]) * (bboxes1[..., 3] - bboxes1[..., 1])
    area2 = (bboxes2[..., 2] - bboxes2[..., 0]) * (bboxes2[..., 3] - bboxes2[..., 1])
    lt = torch.max(bboxes1[..., :, None, :2], bboxes2[..., None, :, :2])
    rb = torch.min(bboxes1[..., :, None, 2:], bboxes2[..., None, :, 2:])
    wh = (rb - lt).clamp(min=0)
    overlap = wh[..., 0] * wh[..., 1]
    union = area1[..., None] + area2[..., None, :] - overlap
    eps = union.new_tensor([eps])
    union = torch.max(union, eps)
    ious = overlap / union
    return ious


class KalmanFilter:
    """motion/kalman_filter.py:38-189 (center_only / use_nsa defaults)."""

    def __init__(self, center_only=False, use_nsa=False):
        self.center_only = center_only
        self.use_nsa = use_nsa
        ndim, dt = 4, 1.
        self._motion_mat = np.eye(2 * ndim, 2 * ndim)
        for i in range(ndim):
            self._motion_mat[i, ndim + i] = dt
        self._update_mat = np.eye(ndim, 2 * ndim)
        self._std_weight_position = 1. / 20
        self._std_weight_velocity = 1. / 160

    def initiate(self, measurement):
        mean_pos = measurement
        mean_vel = np.zeros_like(mean_pos)
        mean = np.r_[mean_pos, mean_vel]
        m3 = float(measurement[3])   # numpy-1.x casting of `python_float * np.float32` (module docstring)
        std = [
            2 * self._std_weight_position * m3,
            2 * self._std_weight_position * m3, 1e-2,
            2 * self._std_weight_position * m3,
            10 * self._std_weight_velocity * m3,
            10 * self._std_weight_velocity * m3, 1e-5,
            10 * self._std_weight_velocity * m3
        ]
        covariance = np.diag(np.square(std))
        return mean, covariance

    def predict(self, mean, covariance):
        m3 = float(mean[3])
        std_pos = [
            self._std_weight_position * m3,
            self._std_weight_position * m3, 1e-2,
            self._std_weight_position * m3
        ]
        std_vel = [
            self._std_weight_velocity * m3,
            self._std_weight_velocity * m3, 1e-5,
            self._std_weight_velocity * m3
        ]
        motion_cov = np.diag(np.square(np.r_[std_pos, std_vel]))
        mean = np.dot(self._motion_mat, mean)
        covariance = np.linalg.multi_dot(
            (self._motion_mat, covariance, self._motion_mat.T)) + motion_cov
        return mean, covariance

    def project(self, mean, covariance, bbox_score=0.):
        m3 = float(mean[3])
        std = [
            self._std_weight_position * m3,
            self._std_weight_position * m3, 1e-1,
            self._std_weight_position * m3
        ]
        if self.use_nsa:
            std = [(1 - bbox_score) * x for x in std]
        innovation_cov = np.diag(np.square(std))
        mean = np.dot(self._update_mat, mean)
        covariance = np.linalg.multi_dot(
            (self._update_mat, covariance, self._update_mat.T))
        return mean, covariance + innovation_cov

    def update(self, mean, covariance, measurement, bbox_score=0.):
        projected_mean, projected_cov = \
            self.project(mean, covariance, bbox_score)
        chol_factor, lower = scipy.linalg.cho_factor(
            projected_cov, lower=True, check_finite=False)
        kalman_gain = scipy.linalg.cho_solve((chol_factor, lower),
                                             np.dot(covariance,
                                                    self._update_mat.T).T,
                                             check_finite=False).T
        innovation = measurement - projected_mean
        new_mean = mean + np.dot(innovation, kalman_gain.T)
        new_covariance = covariance - np.linalg.multi_dot(
            (kalman_gain, projected_cov, kalman_gain.T))
        return new_mean, new_covariance


class BaseTracker:
    """base_tracker.py:10-141."""

    def __init__(self, momentums=None, num_frames_retain=10):
        if momentums is not None:
            assert isinstance(momentums, dict), 'momentums must be a dict'
        self.momentums = momentums
        self.num_frames_retain = num_frames_retain
        self.fp16_enabled = False
        self.reset()

    def reset(self):
        self.num_tracks = 0
        self.tracks = dict()

    @property
    def empty(self):
        return False if self.tracks else True

    @property
    def ids(self):
        return list(self.tracks.keys())

    def update(self, **kwargs):
        memo_items = [k for k, v in kwargs.items() if v is not None]
        rm_items = [k for k in kwargs.keys() if k not in memo_items]
        for item in rm_items:
            kwargs.pop(item)
        if not hasattr(self, 'memo_items'):
            self.memo_items = memo_items
        else:
            assert memo_items == self.memo_items

        assert 'ids' in memo_items
        num_objs = len(kwargs['ids'])
        id_indice = memo_items.index('ids')
        assert 'frame_ids' in memo_items
        frame_id = int(kwargs['frame_ids'])
        if isinstance(kwargs['frame_ids'], int):
            kwargs['frame_ids'] = torch.tensor([kwargs['frame_ids']] *
                                               num_objs)
        for k, v in kwargs.items():
            if len(v) != num_objs:
                raise ValueError()

        for obj in zip(*kwargs.values()):
            id = int(obj[id_indice])
            if id in self.tracks:
                self.update_track(id, obj)
            else:
                self.init_track(id, obj)

        self.pop_invalid_tracks(frame_id)

    def pop_invalid_tracks(self, frame_id):
        invalid_ids = []
        for k, v in self.tracks.items():
            if frame_id - v['frame_ids'][-1] >= self.num_frames_retain:
                invalid_ids.append(k)
        for invalid_id in invalid_ids:
            self.tracks.pop(invalid_id)

    def update_track(self, id, obj):
        for k, v in zip(self.memo_items, obj):
            v = v[None]
            if self.momentums is not None and k in self.momentums:
                m = self.momentums[k]
                self.tracks[id][k] = (1 - m) * self.tracks[id][k] + m * v
            else:
                self.tracks[id][k].append(v)

    def init_track(self, id, obj):
        self.tracks[id] = Dict()
        for k, v in zip(self.memo_items, obj):
            v = v[None]
            if self.momentums is not None and k in self.momentums:
                self.tracks[id][k] = v
            else:
                self.tracks[id][k] = [v]


class KalmanTrackerBase(BaseTracker):
    """kalman_tracker_base.py:19-88."""

    def __init__(self, obj_score_thr=0.3, reid=None, match_iou_thr=0.7, num_tentatives=3, **kwargs):
        super().__init__(**kwargs)
        self.obj_score_thr = obj_score_thr
        self.reid = reid
        self.match_iou_thr = match_iou_thr
        self.num_tentatives = num_tentatives

    @property
    def confirmed_ids(self):
        ids = [id for id, track in self.tracks.items() if not track.tentative]
        return ids

    def init_track(self, id, obj):
        super().init_track(id, obj)
        self.tracks[id].tentative = True
        bbox = bbox_xyxy_to_cxcyah(self.tracks[id].bboxes[-1])  # size = (1, 4)
        assert bbox.ndim == 2 and bbox.shape[0] == 1
        bbox = bbox.squeeze(0).cpu().numpy()
        self.tracks[id].mean, self.tracks[id].covariance = self.kf.initiate(
            bbox)

    def update_track(self, id, obj):
        super().update_track(id, obj)
        if self.tracks[id].tentative:
            if len(self.tracks[id]['bboxes']) >= self.num_tentatives:
                self.tracks[id].tentative = False
        bbox = bbox_xyxy_to_cxcyah(self.tracks[id].bboxes[-1])  # size = (1, 4)
        assert bbox.ndim == 2 and bbox.shape[0] == 1
        bbox = bbox.squeeze(0).cpu().numpy()
        self.tracks[id].mean, self.tracks[id].covariance = self.kf.update(
            self.tracks[id].mean, self.tracks[id].covariance, bbox)

    def pop_invalid_tracks(self, frame_id):
        invalid_ids = []
        for k, v in self.tracks.items():
            # case1: disappeared frames >= self.num_frames_retrain
            case1 = frame_id - v['frame_ids'][-1] >= self.num_frames_retain
            # case2: tentative tracks but not matched in this frame
            case2 = v.tentative and v['frame_ids'][-1] != frame_id
            if case1 or case2:
                invalid_ids.append(k)
        for invalid_id in invalid_ids:
            self.tracks.pop(invalid_id)


class _Dets:
    """One group of the six parallel per-detection tensors `track()` juggles."""
    FIELDS = ('bboxes', 'labels', 'scores', 'scales', 'depth', 'ids')

    def __init__(self, **kw):
        self.__dict__.update(kw)

    def select(self, m, with_ids=True):
        return _Dets(**{k: (getattr(self, k)[m] if (with_ids or k != 'ids') else None) for k in self.FIELDS})

    def __getitem__(self, m):
        return self.select(m)

    @staticmethod
    def cat(a, b):
        return _Dets(**{k: torch.cat((getattr(a, k), getattr(b, k)), dim=0) for k in _Dets.FIELDS})


class OCSORTTracker_Disparity(KalmanTrackerBase):
    """ocsort_tracker_disparity.py:20-618 with cmc=None (the shipped config; the Mesh-Affine CMC branch needs
    OpenCV and is not transcribed)."""

    def __init__(self, obj_score_thr=0.3, init_track_thr=0.7, weight_iou_with_det_scores=True, match_iou_thr=0.3,
                 num_tentatives=3, vel_consist_weight=0.2, vel_delta_t=3, cmc=None, **kwargs):
        super().__init__(**kwargs)
        self.obj_score_thr = obj_score_thr
        self.init_track_thr = init_track_thr

        self.weight_iou_with_det_scores = weight_iou_with_det_scores
        self.match_iou_thr = match_iou_thr
        self.vel_consist_weight = vel_consist_weight
        self.vel_delta_t = vel_delta_t

        self.num_tentatives = num_tentatives
        assert cmc is None or cmc.get('method') is None

    @property
    def unconfirmed_ids(self):
        ids = [id for id, track in self.tracks.items() if track.tentative]
        return ids

    def init_track(self, id, obj):
        super().init_track(id, obj)
        if self.tracks[id].frame_ids[-1] == 0:
            self.tracks[id].tentative = False
        else:
            self.tracks[id].tentative = True
        bbox = bbox_xyxy_to_cxcyah(self.tracks[id].bboxes[-1])  # size = (1, 4)
        assert bbox.ndim == 2 and bbox.shape[0] == 1
        bbox = bbox.squeeze(0).cpu().numpy()
        self.tracks[id].mean, self.tracks[id].covariance = self.kf.initiate(
            bbox)
        # track.obs maintains the history associated detections to this track
        self.tracks[id].obs = []
        bbox_id = self.memo_items.index('bboxes')
        self.tracks[id].obs.append(obj[bbox_id])
        # a placefolder to save mean/covariance before losing tracking it
        self.tracks[id].tracked = True
        self.tracks[id].saved_attr = Dict()
        self.tracks[id].velocity = torch.tensor(
            (-1, -1)).to(obj[bbox_id].device)  # placeholder

    def update_track(self, id, obj):
        super().update_track(id, obj)
        if self.tracks[id].tentative:
            if len(self.tracks[id]['bboxes']) >= self.num_tentatives:
                self.tracks[id].tentative = False
        self.tracks[id].tracked = True
        bbox_id = self.memo_items.index('bboxes')
        self.tracks[id].obs.append(obj[bbox_id])

        bbox1 = self.k_step_observation(self.tracks[id])
        bbox2 = obj[bbox_id]
        self.tracks[id].velocity = self.vel_direction(bbox1, bbox2).to(
            obj[bbox_id].device)

    def vel_direction(self, bbox1, bbox2):
        if bbox1.sum() < 0 or bbox2.sum() < 0:
            return torch.tensor((-1, -1))
        cx1, cy1 = (bbox1[0] + bbox1[2]) / 2.0, (bbox1[1] + bbox1[3]) / 2.0
        cx2, cy2 = (bbox2[0] + bbox2[2]) / 2.0, (bbox2[1] + bbox2[3]) / 2.0
        speed = torch.tensor([cy2 - cy1, cx2 - cx1])
        norm = torch.sqrt((speed[0])**2 + (speed[1])**2) + 1e-6
        return speed / norm

    def vel_direction_batch(self, bboxes1, bboxes2):
        cx1, cy1 = (bboxes1[:, 0] + bboxes1[:, 2]) / 2.0, (bboxes1[:, 1] +
                                                           bboxes1[:, 3]) / 2.0
        cx2, cy2 = (bboxes2[:, 0] + bboxes2[:, 2]) / 2.0, (bboxes2[:, 1] +
                                                           bboxes2[:, 3]) / 2.0
        speed_diff_y = cy2[None, :] - cy1[:, None]
        speed_diff_x = cx2[None, :] - cx1[:, None]
        speed = torch.cat((speed_diff_y[..., None], speed_diff_x[..., None]),
                          dim=-1)
        norm = torch.sqrt((speed[:, :, 0])**2 + (speed[:, :, 1])**2) + 1e-6
        speed[:, :, 0] /= norm
        speed[:, :, 1] /= norm
        return speed

    def k_step_observation(self, track):
        obs_seqs = track.obs
        num_obs = len(obs_seqs)
        if num_obs == 0:
            return torch.tensor((-1, -1, -1, -1)).to(track.obs[0].device)
        elif num_obs > self.vel_delta_t:
            if obs_seqs[num_obs - 1 - self.vel_delta_t] is not None:
                return obs_seqs[num_obs - 1 - self.vel_delta_t]
            else:
                return self.last_obs(track)
        else:
            return self.last_obs(track)

    def ocm_assign_ids(self, ids, det_bboxes, det_scores, weight_iou_with_det_scores=False, match_iou_thr=0.5,
                       offset=torch.tensor([0, 0, 0, 0])):
        # get track_bboxes
        track_bboxes = np.zeros((0, 4))
        track_scales = torch.zeros((0)).to(det_bboxes)
        for id in ids:
            track_bboxes = np.concatenate(
                (track_bboxes, self.tracks[id].mean[:4][None]), axis=0)
            track_scales = torch.concat(
                (track_scales, self.tracks[id].scales[-1]), dim=0)
        track_bboxes = torch.from_numpy(track_bboxes).to(det_bboxes)
        track_bboxes = bbox_cxcyah_to_xyxy(track_bboxes)

        # compute distance
        ious = bbox_overlaps(track_bboxes + offset.to(det_bboxes), det_bboxes[:, :4])
        if weight_iou_with_det_scores:
            ious *= det_scores[np.newaxis, ]
        dists = (1 - ious).cpu().numpy()

        if len(ids) > 0 and len(det_bboxes) > 0:
            track_velocities = torch.stack(
                [self.tracks[id].velocity for id in ids]).to(det_bboxes.device)
            k_step_observations = torch.stack([
                self.k_step_observation(self.tracks[id]) for id in ids
            ]).to(det_bboxes.device)
            # valid1: if the track has previous observations to estimate speed
            # valid2: if the associated observation k steps ago is a detection
            valid1 = track_velocities.sum(dim=1) != -2
            valid2 = k_step_observations.sum(dim=1) != -4
            valid = valid1 & valid2

            vel_to_match = self.vel_direction_batch(k_step_observations[:, :4],
                                                    det_bboxes[:, :4])
            track_velocities = track_velocities[:, None, :].repeat(
                1, det_bboxes.shape[0], 1)

            angle_cos = (vel_to_match * track_velocities).sum(dim=-1)
            angle_cos = torch.clamp(angle_cos, min=-1, max=1)
            angle = torch.acos(angle_cos)  # [0, pi]
            norm_angle = (angle - np.pi / 2.) / np.pi  # [-0.5, 0.5]
            valid_matrix = valid[:, None].int().repeat(1, det_bboxes.shape[0])
            # set non-valid entries 0
            valid_norm_angle = norm_angle * valid_matrix

            dists += valid_norm_angle.cpu().numpy() * self.vel_consist_weight

        # bipartite match
        if dists.size > 0:
            cost, row, col = _lap.lapjv(
                dists, extend_cost=True, cost_limit=1 - match_iou_thr)
        else:
            row = np.zeros(len(ids)).astype(np.int32) - 1
            col = np.zeros(len(det_bboxes)).astype(np.int32) - 1
        return row, col

    def last_obs(self, track):
        for bbox in track.obs[::-1]:
            if bbox is not None:
                return bbox

    def ocr_assign_ids(self, track_obs, det_bboxes, det_scores, weight_iou_with_det_scores=False, match_iou_thr=0.5,
                       offset=torch.tensor([0, 0, 0, 0])):
        ious = bbox_overlaps(track_obs[:, :4] + offset, det_bboxes[:, :4])
        if weight_iou_with_det_scores:
            ious *= det_scores[np.newaxis, ]

        dists = (1 - ious).cpu().numpy()

        # bipartite match
        if dists.size > 0:
            cost, row, col = _lap.lapjv(
                dists, extend_cost=True, cost_limit=1 - match_iou_thr)
        else:
            row = np.zeros(len(track_obs)).astype(np.int32) - 1
            col = np.zeros(len(det_bboxes)).astype(np.int32) - 1
        return row, col

    def online_smooth(self, track, obj):
        last_match_bbox = self.last_obs(track)[:4]
        new_match_bbox = obj[:4]
        unmatch_len = 0
        for bbox in track.obs[::-1]:
            if bbox is None:
                unmatch_len += 1
            else:
                break
        bbox_shift_per_step = (new_match_bbox - last_match_bbox) / (
            unmatch_len + 1)
        track.mean = track.saved_attr.mean
        track.covariance = track.saved_attr.covariance
        for i in range(unmatch_len):
            virtual_bbox = last_match_bbox + (i + 1) * bbox_shift_per_step
            virtual_bbox = bbox_xyxy_to_cxcyah(virtual_bbox[None, :])
            virtual_bbox = virtual_bbox.squeeze(0).cpu().numpy()
            track.mean, track.covariance = self.kf.update(
                track.mean, track.covariance, virtual_bbox)

    def track(self, model, img, feats, data_sample, data_preprocessor=None, rescale=False, **kwargs):
        """ocsort_tracker_disparity.py:345-618, statement order kept.  The reference carries six parallel tensors
        (`*_bboxes, *_labels, *_scores, *_scales, *_depth, *_ids`) through every step as separate variables; here
        each such group is one `_Dets` and `group[mask]` / `_Dets.cat` stand for the six indexing / torch.cat
        statements the reference spells out (:406-421, :453-469, :481-502, :534-566, :581-586)."""
        metainfo = data_sample.metainfo
        inst = data_sample.pred_det_instances
        cur = _Dets(bboxes=inst.bboxes, labels=inst.labels, scores=inst.scores, scales=inst.scales, depth=inst.depth,
                    ids=None)
        labels = inst.labels

        self.img = img
        frame_id = metainfo.get('frame_id', -1)
        if frame_id == 0:                                                   # :385-387
            self.reset()
        if not hasattr(self, 'kf'):                                         # :388-389
            self.kf = model.motion

        if self.empty or cur.bboxes.size(0) == 0:                           # :391-404
            valid_inds = cur.scores > self.init_track_thr
            cur = cur.select(valid_inds, with_ids=False)
            num_new_tracks = cur.bboxes.size(0)
            cur.ids = torch.arange(self.num_tracks,
                                   self.num_tracks + num_new_tracks).to(labels)
            self.num_tracks += num_new_tracks
            self.last_img = img
        else:
            # 0. init                                                       :406-421
            cur.ids = torch.full((cur.bboxes.size(0), ),
                                 -1,
                                 dtype=labels.dtype,
                                 device=labels.device)
            det_inds = cur.scores > self.obj_score_thr
            valid_area = (cur.bboxes[:, 2] - cur.bboxes[:, 0]) * (cur.bboxes[:, 3] - cur.bboxes[:, 1])
            valid_area = valid_area > 100
            det_inds = det_inds & valid_area
            det = cur[det_inds]

            # 1. predict by Kalman Filter                                   :431-441
            for id in self.confirmed_ids:
                # track is lost in previous frame
                if self.tracks[id].frame_ids[-1] != frame_id - 1:
                    self.tracks[id].mean[7] = 0
                if self.tracks[id].tracked:
                    self.tracks[id].saved_attr.mean = self.tracks[id].mean
                    self.tracks[id].saved_attr.covariance = self.tracks[
                        id].covariance
                (self.tracks[id].mean,
                 self.tracks[id].covariance) = self.kf.predict(
                     self.tracks[id].mean, self.tracks[id].covariance)

            # 2. match detections and tracks' predicted locations           :448-469
            match_track_inds, raw_match_det_inds = self.ocm_assign_ids(
                self.confirmed_ids, det.bboxes, det.scores,
                self.weight_iou_with_det_scores, self.match_iou_thr)
            valid = raw_match_det_inds > -1
            det.ids[valid] = torch.tensor(
                self.confirmed_ids)[raw_match_det_inds[valid]].to(labels)
            match = det[valid]
            assert (match.ids > -1).all()
            unmatch = det[~valid]
            assert (unmatch.ids == -1).all()

            # 3. unmatched detections vs the unconfirmed tracks             :473-502
            (tentative_match_track_inds,
             tentative_match_det_inds) = self.ocm_assign_ids(
                 self.unconfirmed_ids, unmatch.bboxes, unmatch.scores,
                 self.weight_iou_with_det_scores, self.match_iou_thr)
            valid = tentative_match_det_inds > -1
            unmatch.ids[valid] = torch.tensor(self.unconfirmed_ids)[
                tentative_match_det_inds[valid]].to(labels)
            match = _Dets.cat(match, unmatch[valid])
            assert (match.ids > -1).all()
            unmatch = unmatch[~valid]
            assert (unmatch.ids == -1).all()

            all_track_ids = [id for id, _ in self.tracks.items()]           # :504-506
            unmatched_track_inds = torch.tensor(
                [ind for ind in all_track_ids if ind not in match.ids])

            if len(unmatched_track_inds) > 0:                               # :508-566
                offset = torch.zeros(4, device=cur.bboxes.device)

                # 4. still some tracks not associated yet, perform OCR
                last_observations = []
                for id in unmatched_track_inds:
                    last_box = self.last_obs(self.tracks[id.item()])
                    last_observations.append(last_box)
                last_observations = torch.stack(last_observations)

                remain_det_ids = torch.full((unmatch.bboxes.size(0), ),
                                            -1,
                                            dtype=labels.dtype,
                                            device=labels.device)

                _, ocr_match_det_inds = self.ocr_assign_ids(
                    last_observations, unmatch.bboxes, unmatch.scores,
                    self.weight_iou_with_det_scores, self.match_iou_thr, offset)

                valid = ocr_match_det_inds > -1
                remain_det_ids[valid] = unmatched_track_inds.clone()[
                    ocr_match_det_inds[valid]].to(labels)
                unmatch.ids = remain_det_ids
                ocr_match = unmatch[valid]
                assert (ocr_match.ids > -1).all()
                ocr_unmatch = unmatch[~valid]
                assert (ocr_unmatch.ids == -1).all()
                unmatch = ocr_unmatch
                match = _Dets.cat(match, ocr_match)

            # 5. summarize the track results                                :568-586
            for i in range(len(match.ids)):
                det_bbox = match.bboxes[i]
                track_id = match.ids[i].item()
                if not self.tracks[track_id].tracked:
                    # the track is lost before this step
                    self.online_smooth(self.tracks[track_id], det_bbox)

            for track_id in all_track_ids:
                if track_id not in match.ids:
                    self.tracks[track_id].tracked = False
                    self.tracks[track_id].obs.append(None)

            cur = _Dets.cat(match, unmatch)

            # 6. assign new ids                                             :588-593
            new_track_inds = cur.ids == -1
            cur.ids[new_track_inds] = torch.arange(
                self.num_tracks,
                self.num_tracks + new_track_inds.sum()).to(labels)
            self.num_tracks += new_track_inds.sum()

        self.update(                                                        # :595-602
            ids=cur.ids,
            bboxes=cur.bboxes,
            scores=cur.scores,
            labels=cur.labels,
            scales=cur.scales,
            depth=cur.depth,
            frame_ids=frame_id)

        self.last_img = img

        return Instances(bboxes=cur.bboxes, labels=cur.labels, scores=cur.scores, scales=cur.scales,
                         depth=cur.depth, instances_id=cur.ids)                # :606-616
