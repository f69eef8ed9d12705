"""GPU: BASELINE configs[4]'s code path on a tiny on-disk dataset in the AirDrone layout (tools/make_tiny_airdrone.py):
CocoVID json -> MOTDispDataset -> PNG decode (uint8 left / right, uint16 disparity) -> raw-byte upload
(RawFrameUploader: the u8 + u16 code path, st_pack_raw_inputs on the device) -> dense path -> a fresh tracker per
video (VideoSampler's whole-video split) -> MOTDroneMetrics incl. the depth-range filter
(reference mot_disp_dataset.py:11-104, loading_disparity.py:71-134, video_sampler.py:25-70, mot_drone_metrics.py:155-253).
Reading the files must change NOTHING: tracks from the decoded dataset equal the tracks from the in-memory frames the
files were written from, in both configurations (precomputed disparity = the reference's own; right image + stereo
module = north_star's)."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tools'))

from stereotracking_amd import datasets as ds  # noqa: E402
from stereotracking_amd.metrics import MOTDroneMetrics  # noqa: E402
from stereotracking_amd.motion import KalmanFilter  # noqa: E402
from stereotracking_amd.pipeline import StereoDensePipeline  # noqa: E402
from stereotracking_amd.sequence import run_video_replicas, synthetic_sequence  # noqa: E402
from stereotracking_amd.synthetic import synthetic_state_dict  # noqa: E402
from stereotracking_amd.trackers import OCSORTTracker_Disparity  # noqa: E402

pytestmark = pytest.mark.gpu
H, W, D, T, V = 96, 160, 32, 10, 3


class _Model:
    motion = KalmanFilter()


def make_tracker():
    return OCSORTTracker_Disparity(obj_score_thr=0.02, init_track_thr=0.03, weight_iou_with_det_scores=False,
                                   match_iou_thr=0.1, num_tentatives=3, vel_consist_weight=0.2, vel_delta_t=3,
                                   num_frames_retain=30)


@pytest.fixture(scope='module')
def dataset(tmp_path_factory):
    from make_tiny_airdrone import make
    base, _ = make(str(tmp_path_factory.mktemp('airdrone')), videos=V, frames=T, height=H, width=W, max_disp=D, objects=3)
    return ds.MOTDispDataset(ann_file='annotations/val_cocoformat_80.json', data_root=base + os.sep,
                             data_prefix=dict(img_path='val/'), depth_dir_name='depth')


@pytest.mark.parametrize('use_right,rgb_only', [(False, False), (True, False), (False, True), (True, True)])
def test_tiny_airdrone_through_reader_pipeline_tracker_metrics(dataset, use_right, rgb_only, cuda):
    """rgb_only: the detector of the reference's RGB-only stereo config (yolox_s_mmyolo_mot_airdrone.py:40-42); the
    disparity - loaded or computed - still drives the per-box depth and the depth-range filter."""
    pipe = StereoDensePipeline(4, (H, W), 0.375, 0.33, 1, stereo=use_right, max_disp=D, max_det=256, rgb_only=rgb_only)
    pipe.load_state_dict(synthetic_state_dict(pipe.param_table(), seed=9, prior_prob=0.2, logit_std=2.5), autotune=False)
    videos, gts = ds.load_videos(dataset, use_right)
    assert sorted(videos) == ['seq00', 'seq01', 'seq02'] and all(len(v) == T for v in videos.values())
    m = MOTDroneMetrics(depth_thr=40)                         # half of the objects lie beyond 40 m: the filter bites
    res, scores = run_video_replicas(pipe, videos, make_tracker, _Model(), cuda, metrics=m, gts=gts,
                                     already_sharded=True)
    n_gt_all = sum(len(g) for v in gts.values() for g in v)
    n_gt_near = sum(1 for v in gts.values() for g in v for i in g if i['location'][-1] <= 40)
    assert 0 < n_gt_near < n_gt_all
    assert scores['combined']['TP'] + scores['combined']['FN'] == n_gt_near
    assert set(scores['per_video']) == set(videos) and sum(len(t) for r in res.values() for t in r) > 0
    for rows in m.pred.values():                              # predictions beyond the depth range were dropped too
        assert all(r[0] >= 1 for r in rows)
    # the same frames straight from memory (what the files were written from): identical tracks
    mem = {}
    for v in range(V):
        frames = list(synthetic_sequence(T, 3, H, W, D, seed=v))
        if not use_right:      # the files carry an invalid (65535) patch per frame: reproduce it on the in-memory map
            codes = videos[f'seq{v:02d}'].codes.numpy().view(np.uint16)
            for t, f in enumerate(frames):
                f['disp'] = np.where(codes[t] == 65535, 0.0, f['disp']).astype(np.float32)
        mem[f'seq{v:02d}'] = frames
    res_mem, _ = run_video_replicas(pipe, mem, make_tracker, _Model(), cuda)
    for name in videos:
        for a, b in zip(res[name], res_mem[name]):
            assert a.instances_id.tolist() == b.instances_id.tolist() and torch.equal(a.bboxes, b.bboxes)
            assert torch.equal(a.depth.nan_to_num(-7.0), b.depth.nan_to_num(-7.0))
