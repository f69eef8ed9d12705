"""Batched GPU association: OCSORTTracker_Disparity.track for MANY independent sequences per step.

One wave per sequence runs the whole association step on the device (csrc/batched_assoc.hip): Kalman prediction, the
tracks x detections cost matrix and the Kalman updates across the lanes, the Jonker-Volgenant assignment and the track
bookkeeping on lane 0; the per-sequence state (tracks, Kalman filters, observation windows) stays in device memory.
Ids / rows equal the host tracker's (st_tracker_track) on the same detections - reference
mmtrack/models/trackers/ocsort_tracker_disparity.py:345-618.  Use it when a step carries hundreds of short sequences
(multi-camera serving); for ONE video the native host tracker is faster (0.05 ms per frame) and is what
OCSORT_Disparity uses."""
import ctypes as C

import torch

from . import _lib
from ._lib import StTrackerConfig, check, current_stream, ptr


class BatchedGpuTracker:
    """`batch` sequences in lockstep.  step(frame_ids, dets, counts) -> (rows, ids, n) device tensors:
    rows (batch, max_dets, 8) = pred_track_instances rows [depth-scaled box, score, label, depth, scale] in the
    reference's output order, ids (batch, max_dets) int64, n (batch,) int32 (-1 = the sequence had no frame)."""

    def __init__(self, batch, max_tracks=128, max_dets=512, device=None, obj_score_thr=0.3, init_track_thr=0.7,
                 weight_iou_with_det_scores=True, match_iou_thr=0.3, num_tentatives=3, vel_consist_weight=0.2,
                 vel_delta_t=3, num_frames_retain=10):
        self.lib = _lib.load()
        self.batch, self.max_tracks, self.max_dets = int(batch), int(max_tracks), int(max_dets)
        self.device = torch.device(device) if device is not None else torch.device('cuda', torch.cuda.current_device())
        cfg = StTrackerConfig(C.sizeof(StTrackerConfig), float(obj_score_thr), float(init_track_thr),
                              int(bool(weight_iou_with_det_scores)), float(match_iou_thr), int(num_tentatives),
                              float(vel_consist_weight), int(vel_delta_t), int(num_frames_retain))
        h = C.c_void_p()
        check(self.lib.st_batched_tracker_create(C.byref(cfg), self.batch, self.max_tracks, self.max_dets, C.byref(h)),
              'st_batched_tracker_create')
        self.handle = h
        self.state = torch.zeros(self.lib.st_batched_tracker_state_bytes(h), dtype=torch.uint8, device=self.device)
        self.scratch = torch.empty(self.lib.st_batched_tracker_scratch_bytes(h), dtype=torch.uint8, device=self.device)
        B, M = self.batch, self.max_dets
        self.rows = torch.zeros(B, M, 8, dtype=torch.float32, device=self.device)
        self.ids = torch.zeros(B, M, dtype=torch.int64, device=self.device)
        self.n = torch.zeros(B, dtype=torch.int32, device=self.device)
        self.status = torch.zeros(B, dtype=torch.int32, device=self.device)

    def __del__(self):
        h = getattr(self, 'handle', None)
        if h:
            try:
                self.lib.st_batched_tracker_destroy(h)
            except Exception:
                pass
            self.handle = None

    def reset(self):
        self.state.zero_()

    def step(self, frame_ids, dets, counts, check_status=True):
        """frame_ids (batch,) int32, dets (batch, max_dets, 8) float32, counts (batch,) int32 - CUDA tensors.
        check_status: one host sync to raise on a capacity overflow (False: read `self.status` yourself).  A non-zero
        status is sticky per sequence: the device state of that sequence is invalid until a step with frame_id 0."""
        for t, dt, shape in ((frame_ids, torch.int32, (self.batch,)), (dets, torch.float32, (self.batch, self.max_dets, 8)),
                             (counts, torch.int32, (self.batch,))):
            if not (t.is_cuda and t.dtype == dt and tuple(t.shape) == shape and t.is_contiguous()):
                raise ValueError(f'expected a contiguous CUDA {dt} tensor of shape {shape}, got {t.dtype} {tuple(t.shape)}')
        self.status.zero_()
        check(self.lib.st_batched_tracker_step(self.handle, ptr(frame_ids), ptr(dets), ptr(counts), ptr(self.state),
                                               ptr(self.scratch), ptr(self.rows), ptr(self.ids), ptr(self.n),
                                               ptr(self.status), current_stream()), 'st_batched_tracker_step')
        if check_status:
            st = self.status.cpu()
            if int(st.max()) != 0:
                bad = torch.nonzero(st).flatten().tolist()
                raise RuntimeError(f'batched association: capacity exceeded in sequences {bad} '
                                   f'(status {st[bad].tolist()}: 1 = max_tracks={self.max_tracks}, 2 = max_dets={self.max_dets})')
        return self.rows, self.ids, self.n
