"""CPU suite (-m "not gpu"), part 2: host logic — plugin surface (registry / config merge / model
build), containers, preprocessor, Kalman filter, LAP, OC-SORT association, frame sharding and the
world_size-2 gloo all-gather of detection buffers."""
import itertools
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CFG = os.path.join(ROOT, 'configs', 'stereo_tracking', 'ocsort', 'yolox_s_mmyolo_mot_airdrone_disp.py')


# ---- plugin surface ---------------------------------------------------------------------------------
def test_config_base_merge_and_overrides():
    from stereotracking_amd.config import Config
    cfg = Config.fromfile(CFG)
    det = cfg.model.detector
    # child overrides type + thresholds, base keys survive the merge
    assert det.type == 'mmtrack.YOLODetector_Disparity_V1'
    assert det.backbone.type == 'mmtrack.YOLOXCSPDarknet_Disparity_V1_MMYOLO'
    assert det.backbone.widen_factor == 0.5 and det.backbone.deepen_factor == 0.33
    assert det.bbox_head.head_module.num_classes == 1 and det.bbox_head.head_module.feat_channels == 256
    assert det.test_cfg == dict(yolox_style=True, multi_label=True, score_thr=0.01, max_per_img=300,
                                nms=dict(type='nms', iou_threshold=0.5))
    assert cfg.model.tracker.match_iou_thr == 0.1 and cfg.model.tracker.num_frames_retain == 30
    cfg.merge_from_dict({'model.detector.test_cfg.score_thr': 0.2})
    assert cfg.model.detector.test_cfg.score_thr == 0.2


def test_models_build_from_reference_shaped_config():
    from stereotracking_amd import mot  # noqa: F401
    from stereotracking_amd.config import Config
    from stereotracking_amd.registry import MODELS, TASK_UTILS
    cfg = Config.fromfile(CFG)
    model = MODELS.build(cfg.model)
    assert type(model).__name__ == 'OCSORT_Disparity'
    assert type(model.detector).__name__ == 'YOLODetector_Disparity_V1'
    assert type(model.tracker).__name__ == 'OCSORTTracker_Disparity' and model.tracker.init_track_thr == 0.7
    assert type(model.motion).__name__ == 'KalmanFilter'
    assert model.baseline == 0.25 and model.focal_length == 640
    keys = set(model.state_dict().keys())
    for k in ('detector.backbone.stem.conv.conv.weight', 'detector.backbone.disp_stem.conv.bn.running_mean',
              'detector.backbone.disp_stage1.1.blocks.0.conv1.bn.num_batches_tracked',
              'detector.neck.reduce_layers.2.conv.weight', 'detector.bbox_head.head_module.multi_level_conv_reg.1.bias'):
        assert k in keys, k
    for name in ('OCSORT_Disparity', 'mmtrack.YOLODetector_Disparity_V1', 'YOLOXPAFPN', 'YOLOXHead',
                 'YOLOXHeadModule', 'TrackDataPreprocessor_Disparity_V1', 'OCSORTTracker_Disparity',
                 'StereoCostVolume'):
        assert name in MODELS, name
    assert 'KalmanFilter' in TASK_UTILS
    with pytest.raises(KeyError):
        MODELS.build(dict(type='NoSuchModel'))
    with pytest.raises(NotImplementedError):
        MODELS.build(dict(type='YOLOXCSPDarknet_Disparity_V1_MMYOLO', use_depthwise=True))
    # checkpoints in the reference layout load with plain load_state_dict
    from stereotracking_amd.synthetic import synthetic_state_dict
    sd = synthetic_state_dict(model.detector._table, seed=3)
    missing, unexpected = model.detector.load_state_dict(sd, strict=False)
    assert not unexpected and all(k.endswith('num_batches_tracked') for k in missing)
    with pytest.raises(RuntimeError, match='HIP path only'):
        z = torch.zeros(1, 3, 64, 96)
        model.detector.predict(dict(img=z, disp_postp=z), [])


REF_CFG_DIR = '/root/reference/configs/stereo_tracking/ocsort'


@pytest.mark.skipif(not os.path.isdir(REF_CFG_DIR), reason='reference tree not present (GPU box)')
@pytest.mark.parametrize('name', ['yolox_s_mmyolo_mot_airdrone_disp.py', 'yolox_s_mmyolo_mot_airdrone.py'])
def test_reference_config_files_parse_verbatim_and_build(name):
    """Drop-in check of the plugin surface: the REFERENCE's own config file, read from where it lies
    (with its own `_base_` chain), parses with this repo's loader and `MODELS.build(cfg.model)` yields the HIP-backed
    OCSORT_Disparity with the shipped thresholds - no edits to the config."""
    from stereotracking_amd import mot  # noqa: F401
    from stereotracking_amd.config import Config
    from stereotracking_amd.registry import MODELS
    cfg = Config.fromfile(os.path.join(REF_CFG_DIR, name))
    ours = Config.fromfile(os.path.join(os.path.dirname(CFG), name))    # BOTH configs under configs/stereo_tracking
    rgb_only = not name.endswith('_disp.py')
    assert cfg.model.detector.test_cfg == ours.model.detector.test_cfg       # same merged thresholds
    assert dict(cfg.model.tracker) == dict(ours.model.tracker)
    assert dict(cfg.model.detector.backbone) == dict(ours.model.detector.backbone)
    model = MODELS.build(cfg.model)
    assert type(model).__name__ == 'OCSORT_Disparity'
    assert type(model.detector).__name__ == ('YOLODetector' if rgb_only else 'YOLODetector_Disparity_V1')
    # the RGB-only config: backbone mmtrack.CSPDarknet (csp_darknet.py:8-13) - no disparity-branch parameters at all
    assert type(model.detector.backbone).__name__ == ('CSPDarknet' if rgb_only else 'YOLOXCSPDarknet_Disparity_V1_MMYOLO')
    assert model.detector.rgb_only == rgb_only
    names = [n for n, _ in model.detector._table]
    assert any('disp_stem' in n or 'disp_stage1' in n for n in names) == (not rgb_only)
    assert 'backbone.stage4.1.conv2.conv.weight' in names and 'neck.out_layers.2.conv.weight' in names
    assert model.detector.widen_factor == 0.5 and model.detector.deepen_factor == 0.33
    assert model.tracker.match_iou_thr == 0.1 and model.tracker.num_frames_retain == 30
    assert type(model.data_preprocessor).__name__ == 'TrackDataPreprocessor_Disparity_V1'
    assert model.data_preprocessor.pad_size_divisor == 32


def test_structures_and_preprocessor():
    from stereotracking_amd.mot import TrackDataPreprocessor_Disparity_V1, stack_batch
    from stereotracking_amd.structures import InstanceData, TrackDataSample
    inst = InstanceData(bboxes=torch.arange(12.).view(3, 4), scores=torch.tensor([.9, .2, .5]))
    assert len(inst) == 3 and 'scores' in inst and len(inst[inst.scores > .3]) == 2
    inst['labels'] = torch.zeros(3, dtype=torch.long)
    c = inst.clone()
    c.bboxes[0, 0] = 100
    assert inst.bboxes[0, 0] == 0
    with pytest.raises(ValueError):
        inst['bad'] = torch.zeros(2)
    s = TrackDataSample(dict(frame_id=3, ori_shape=(720, 1280)))
    s.pred_det_instances = inst
    assert s.frame_id == 3 and s.metainfo['ori_shape'] == (720, 1280) and len(s.pred_det_instances) == 3
    x = stack_batch([torch.ones(1, 3, 720, 1280)], 32, 0)
    assert x.shape == (1, 1, 3, 736, 1280) and x[0, 0, :, 720:].abs().sum() == 0
    pre = TrackDataPreprocessor_Disparity_V1(pad_size_divisor=32, device='cpu')
    data = dict(inputs=dict(img=[torch.full((1, 3, 50, 70), 7, dtype=torch.uint8)],
                            disp_postp=[torch.ones(1, 3, 50, 70)]), data_samples=[s])
    out = pre(data)
    assert out['inputs']['img'].shape == (1, 1, 3, 64, 96) and out['inputs']['img'].dtype == torch.float32
    assert s.metainfo['batch_input_shape'] == (64, 96) and s.metainfo['pad_shape'] == (50, 70)


# ---- motion / association --------------------------------------------------------------------------------
def test_kalman_filter_closed_form():
    from stereotracking_amd.motion import KalmanFilter
    kf = KalmanFilter()
    z = np.array([100., 50., 0.5, 40.])
    mean, cov = kf.initiate(z)
    assert np.allclose(mean, [100, 50, .5, 40, 0, 0, 0, 0])
    assert np.allclose(np.diag(cov), np.square([4, 4, 1e-2, 4, 2.5, 2.5, 1e-5, 2.5]))
    m1, c1 = kf.predict(mean, cov)
    assert np.allclose(m1, mean)  # zero velocity
    assert np.isclose(c1[0, 0], 16 + 6.25 + 4.0) and np.isclose(c1[0, 4], 6.25) and np.isclose(c1[4, 4], 6.25 + 0.0625)
    # scalar Kalman gain on x: P/(P+R)
    z2 = np.array([110., 50., 0.5, 40.])
    m2, c2 = kf.update(m1, c1, z2)
    P, R = c1[0, 0], 4.0
    assert np.isclose(m2[0], 100 + P / (P + R) * 10) and np.isclose(c2[0, 0], P - P * P / (P + R))
    assert np.isclose(m2[4], c1[4, 0] / (P + R) * 10)


def test_lapjv_extended_is_optimal_and_respects_cost_limit():
    from stereotracking_amd.trackers import lapjv_extended
    rng = np.random.RandomState(0)
    for n, m in [(3, 3), (4, 2), (2, 5), (5, 6), (1, 1)]:
        for _ in range(5):
            cost = rng.uniform(0, 1.2, (n, m))
            limit = 0.9
            x, y = lapjv_extended(cost, limit)
            assert len(x) == n and len(y) == m
            for i, j in enumerate(x):
                if j >= 0:
                    assert y[j] == i
            got = sum(cost[i, j] for i, j in enumerate(x) if j >= 0) + limit / 2 * ((x < 0).sum() + (y < 0).sum())
            # brute force over partial matchings of the same extended objective
            best = np.inf
            for k in range(min(n, m) + 1):
                for rows in itertools.combinations(range(n), k):
                    for cols in itertools.permutations(range(m), k):
                        v = sum(cost[r, c] for r, c in zip(rows, cols)) + limit / 2 * (n + m - 2 * k)
                        best = min(best, v)
            assert np.isclose(got, best), (cost, x, y)
            assert all(cost[i, j] <= limit + 1e-12 for i, j in enumerate(x) if j >= 0)


class _Model:
    def __init__(self):
        from stereotracking_amd.motion import KalmanFilter
        self.motion = KalmanFilter()


def synthetic_stream(num_frames=64, K=6, seed=0, drop_prob=0.08, noise=0.4):
    """K rectangles with constant velocity + noise, depth-consistent scales."""
    rng = np.random.RandomState(seed)
    pos = rng.uniform([100, 80], [1100, 600], (K, 2))
    vel = rng.uniform(-4, 4, (K, 2))
    size = rng.uniform(12, 50, (K, 2))
    depth = rng.uniform(5, 60, K)
    frames = []
    for t in range(num_frames):
        p = pos + vel * t + rng.normal(0, noise, (K, 2))
        keep = (rng.uniform(size=K) > drop_prob) | (t == 0)
        b = np.concatenate([p - size / 2, p + size / 2], 1)[keep].astype(np.float32)
        frames.append(dict(bboxes=torch.from_numpy(b), scores=torch.full((len(b),), 0.9),
                           labels=torch.zeros(len(b), dtype=torch.long), scales=torch.ones(len(b)),
                           depth=torch.from_numpy(depth[keep].astype(np.float32)), gt=np.nonzero(keep)[0]))
    return frames


def run_tracker(frames, **kw):
    from stereotracking_amd.structures import InstanceData, TrackDataSample
    from stereotracking_amd.trackers import OCSORTTracker_Disparity
    args = dict(obj_score_thr=0.3, init_track_thr=0.7, weight_iou_with_det_scores=False, match_iou_thr=0.1,
                num_tentatives=3, vel_consist_weight=0.2, vel_delta_t=3, num_frames_retain=30)
    args.update(kw)
    trk = OCSORTTracker_Disparity(**args)
    model = _Model()
    out = []
    for t, f in enumerate(frames):
        s = TrackDataSample(dict(frame_id=t))
        s.pred_det_instances = InstanceData(**{k: f[k] for k in ('bboxes', 'scores', 'labels', 'scales', 'depth')})
        out.append(trk.track(model, None, None, s))
    return trk, out


def test_tracker_keeps_identities_on_a_clean_stream():
    frames = synthetic_stream(64, 6, seed=1)
    trk, out = run_tracker(frames)
    # every object keeps ONE id over the whole sequence, across the dropped frames (OCR re-association)
    gt_to_ids = {}
    for f, o in zip(frames, out):
        assert len(o) == len(f['gt'])
        for row in range(len(o)):
            j = int(torch.argmin((f['bboxes'] - o.bboxes[row]).abs().sum(1)))
            gt_to_ids.setdefault(int(f['gt'][j]), set()).add(int(o.instances_id[row]))
    assert all(len(v) == 1 for v in gt_to_ids.values()), gt_to_ids
    assert len(set.union(*gt_to_ids.values())) == 6
    assert set(out[0].keys()) >= {'bboxes', 'labels', 'scores', 'scales', 'depth', 'instances_id'}


def test_tracker_reference_semantics_first_frame_tentative_and_retain():
    f0 = dict(bboxes=torch.tensor([[10., 10, 50, 50], [200., 200, 260, 260]]), scores=torch.tensor([0.9, 0.5]),
              labels=torch.zeros(2, dtype=torch.long), scales=torch.ones(2), depth=torch.ones(2))
    far = dict(bboxes=torch.tensor([[600., 400, 650, 450]]), scores=torch.tensor([0.4]),
               labels=torch.zeros(1, dtype=torch.long), scales=torch.ones(1), depth=torch.ones(1))
    empty = {k: v[:0] for k, v in f0.items()}
    strong = dict(far, scores=torch.tensor([0.95]))
    trk, out = run_tracker([f0, far, empty, far, strong], num_frames_retain=2)
    # frame 0: only score > init_track_thr starts a track, born confirmed
    assert out[0].instances_id.tolist() == [0]
    # frame 1: unmatched det with score 0.4 (> obj_score_thr, <= init_track_thr) still starts a (tentative) track
    assert out[1].instances_id.tolist() == [1]
    # frame 2 (no detections): tentative track 1 is popped, track 0 (lost 2 >= num_frames_retain) too ->
    # frame 3 sees an EMPTY tracker, where only score > init_track_thr may start a track: 0.4 does not
    assert out[2].instances_id.tolist() == [] and out[3].instances_id.tolist() == []
    # frame 4: a strong detection starts id 2 (ids are never reused); it is tentative and unmatched... but it
    # was fed THIS frame, so it survives pop_invalid_tracks
    assert out[4].instances_id.tolist() == [2]
    trk2, _ = run_tracker([f0, far, empty, far, strong], num_frames_retain=2)
    state = trk2.native_state() if trk2.backend == 'native' else [dict(id=i, tentative=t.tentative) for i, t in trk2.tracks.items()]
    assert [t['id'] for t in state] == [2] and state[0]['tentative']
    # small boxes (area <= 100) never enter association
    tiny = dict(bboxes=torch.tensor([[10., 10, 19, 19]]), scores=torch.tensor([0.95]),
                labels=torch.zeros(1, dtype=torch.long), scales=torch.ones(1), depth=torch.ones(1))
    trk, out = run_tracker([f0, tiny])
    assert len(out[1]) == 0
    with pytest.raises(ValueError):
        from stereotracking_amd.trackers import OCSORTTracker_Disparity
        OCSORTTracker_Disparity(cmc=dict(method='bogus'))


# ---- sharding + all-gather (world_size 2, gloo) ---------------------------------------------------------------
def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _worker(rank, world, port, frames_total, q):
    import torch.distributed as dist
    from stereotracking_amd import dist as sdist
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    start, stop, chunk = sdist.shard_frames(frames_total)
    M = 5
    local = torch.zeros(chunk, M, 8)
    counts = torch.zeros(chunk, dtype=torch.int32)
    for i, t in enumerate(range(start, stop)):
        k = t % M + 1
        counts[i] = k
        local[i, :k, 4] = t            # score slot carries the global frame index
        local[i, :k, 0] = torch.arange(k)
    full, call = sdist.gather_detections(local, counts)
    q.put((rank, full[:, :, 4].max(dim=1)[0].tolist(), call.tolist(), sdist.shard_videos(5)))
    dist.barrier()
    dist.destroy_process_group()


def test_frame_sharding_and_allgather_world2_gloo():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    T = 7  # ragged: 4 + 3 frames
    procs = [ctx.Process(target=_worker, args=(r, 2, port, T, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, frame_idx, counts, vids in res:
        # every rank holds all frames in frame order; the padded 8th slot has count 0
        assert frame_idx[:T] == [float(t) for t in range(T)] and counts[:T] == [t % 5 + 1 for t in range(T)]
        assert counts[T:] == [0]
    assert res[0][3] == [0, 1, 2] and res[1][3] == [3, 4]  # reference video split (np.array_split)


def test_shard_frames_single_process():
    from stereotracking_amd import dist as sdist
    assert sdist.shard_frames(512, 3, 8) == (192, 256, 64)
    assert sdist.shard_frames(10, 3, 4) == (9, 10, 3)
    assert sdist.shard_frames(2, 3, 4) == (2, 2, 1)
    x = torch.zeros(2, 3, 8)
    assert sdist.gather_detections(x) is x


def test_stereo_config_builds_cost_volume_module_with_aggregation():
    """The stereo config (north_star's new module under configs/stereo_tracking) builds a StereoCostVolume with
    two aggregation convs; its parameters live once in the shell's state_dict (`stereo.agg.*`), start as the
    identity, and the module refuses a CPU forward (the HIP path is the only product path)."""
    from stereotracking_amd import mot  # noqa: F401
    from stereotracking_amd.config import Config
    from stereotracking_amd.registry import MODELS
    cfg = Config.fromfile(os.path.join(os.path.dirname(CFG), 'stereo_yolox_s_mot_airdrone_costvolume.py'))
    assert cfg.model.stereo.agg_layers == 2 and cfg.model.stereo.max_disp == 192
    model = MODELS.build(cfg.model)
    st = model.stereo
    assert type(st).__name__ == 'StereoCostVolume' and st.levels == 48 and model.detector.stereo is st
    assert st.param_table() == [('agg.0.weight', (48, 48, 3, 3)), ('agg.0.bias', (48,)),
                                ('agg.1.weight', (48, 48, 3, 3)), ('agg.1.bias', (48,))]
    keys = list(model.state_dict().keys())
    assert 'stereo.agg.1.bias' in keys and not any(k.startswith('detector.stereo') for k in keys)
    w = st.agg[0].weight
    assert float(w.sum()) == 48.0 and float(w[5, 5, 1, 1]) == 1.0 and float(st.agg[1].bias.abs().sum()) == 0.0
    with pytest.raises(RuntimeError, match='no CPU forward'):
        st(torch.zeros(1, 48, 4, 4))
    with pytest.raises(ValueError):
        MODELS.build(dict(type='StereoCostVolume', max_disp=40, agg_layers=1))   # 10 levels: not a multiple of 4


def _store_worker(args):
    path, k = args
    from stereotracking_amd.pipeline import _store_plans
    return _store_plans(path, {f'key{k}': [k] * 50})


def test_tuning_cache_store_is_atomic_and_merging(tmp_path):
    """plans measured by concurrent ranks / test processes are MERGED (lock + re-read + tmp file +
    os.replace); a file that does not parse is never overwritten; the committed plan file is not the write target."""
    import json
    import multiprocessing as mp
    from stereotracking_amd import pipeline as pl
    path = str(tmp_path / 'cache' / 'tuning.json')
    with mp.get_context('spawn').Pool(4) as pool:
        assert all(pool.map(_store_worker, [(path, k) for k in range(16)]))
    got = json.load(open(path))
    assert sorted(got) == sorted(f'key{k}' for k in range(16))      # nobody's key was dropped
    bad = str(tmp_path / 'bad.json')
    open(bad, 'w').write('{"truncated": [1, 2')
    assert pl._store_plans(bad, {'x': 1}) is False
    assert open(bad).read() == '{"truncated": [1, 2'
    assert pl._read_plans(bad) == ({}, False)
    os.environ.pop('ST_TUNE_CACHE', None)
    assert os.path.abspath(pl.default_tuning_cache()) != os.path.abspath(pl.committed_tuning_plans())
    assert not pl.default_tuning_cache().startswith(ROOT + os.sep + 'configs')
    plans, ok = pl._read_plans(pl.committed_tuning_plans())
    assert ok and all('gfx950_cu256' in k for k in plans)            # device identity is part of every key


# ---- configs[3] at its stated shape: 512 frames, world 8, gloo (VERDICT r3 next #8a) --------------------------------
def _records_from_stream(det, T, M):
    """Detection rows [t, box, score, depth, scale] -> (T, M + 1, 8) frame records (pack_detections layout)."""
    rec = np.zeros((T, M + 1, 8), np.float32)
    for t in range(T):
        d = det[det[:, 0] == t]
        k = len(d)
        rec[t, 0, :3] = (k, M, 1)
        rec[t, 1:1 + k, 0:4] = d[:, 1:5]
        rec[t, 1:1 + k, 4], rec[t, 1:1 + k, 6], rec[t, 1:1 + k, 7] = d[:, 5], d[:, 6], d[:, 7]
    return torch.from_numpy(rec)


def _config3_tracks(records, T):
    from stereotracking_amd.motion import KalmanFilter
    from stereotracking_amd.sequence import track_gathered
    from stereotracking_amd.trackers import OCSORTTracker_Disparity

    class _M:
        motion = KalmanFilter()
    trk = OCSORTTracker_Disparity(obj_score_thr=0.3, init_track_thr=0.7, weight_iou_with_det_scores=False,
                                  match_iou_thr=0.1, num_tentatives=3, vel_consist_weight=0.2, vel_delta_t=3,
                                  num_frames_retain=30)
    res = track_gathered(records, None, T, trk, _M())
    return [r.instances_id.tolist() for r in res], [float(r.bboxes.double().sum()) for r in res]


def _config3_worker(rank, world, port, T, B, q):
    import torch.distributed as dist
    from stereotracking_amd import dist as sdist
    from stereotracking_amd.sequence import gather_shard_records
    from stereotracking_amd.synthetic import synthetic_detection_stream
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    torch.set_num_threads(1)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    records = _records_from_stream(synthetic_detection_stream(77, T, K=8, occlusion=(2, 200, 215)), T, 16)
    start, stop, _ = sdist.shard_frames(T)
    everything = gather_shard_records(records[start:stop].clone(), T, B, torch.device('cpu'))
    ids, sums = _config3_tracks(everything, T)
    q.put((rank, start, stop, bool(torch.equal(everything, records)), ids, sums))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('T', [512, 509])
def test_config3_sharded_sequence_world8_gloo(T):
    """BASELINE configs[3] at its stated shape, on CPU: 512 recorded frame records sharded 8-way (T = 509: ragged, the
    last rank holds 61 of 64 slots, and the shards are padded to a multiple of the 8-frame launch plan) -> ONE
    all-gather (gloo) -> the tracker on every rank: the gathered records equal the unsharded ones bit for bit and the
    track ids / boxes equal the single-process run on every rank (reference sharding: video_sampler.py:25-70; gather:
    mot_drone_metrics.py:336-358)."""
    from stereotracking_amd.synthetic import synthetic_detection_stream
    world, B = 8, 8
    ref_ids, ref_sums = _config3_tracks(
        _records_from_stream(synthetic_detection_stream(77, T, K=8, occlusion=(2, 200, 215)), T, 16), T)
    assert len(set(i for f in ref_ids for i in f)) >= 8 and sum(len(f) for f in ref_ids) > 3 * T
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_config3_worker, args=(r, world, port, T, B, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=300) for _ in range(world)), key=lambda r: r[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    chunk = (T + world - 1) // world
    for rank, start, stop, equal, ids, sums in res:
        assert (start, stop) == (min(rank * chunk, T), min(rank * chunk + chunk, T))
        assert equal, f'rank {rank}: gathered records differ from the unsharded stream'
        assert ids == ref_ids and sums == ref_sums, f'rank {rank}: tracks differ from the single-process run'


def test_checkpoint_loading_and_color_pretrained_init(tmp_path):
    """load_checkpoint (the runner's `load_from`: non-strict, `module.` prefix stripped, size mismatches reported and
    skipped) and the detector's ColorPretrained init (yolo_detector_disparity_v1.py:144-166: the disparity branch starts
    from the RGB branch's stem / stage1 weights); a URL raises with the instruction to pass a local file."""
    from stereotracking_amd import mot  # noqa: F401
    from stereotracking_amd.checkpoint import load_checkpoint
    from stereotracking_amd.config import Config
    from stereotracking_amd.registry import MODELS
    cfg = Config.fromfile(os.path.join(ROOT, 'configs', 'stereo_tracking', 'ocsort', 'yolox_s_mmyolo_mot_airdrone_disp.py'))
    det_cfg = dict(cfg.model.detector)
    det_cfg.pop('init_cfg', None)
    src = MODELS.build(det_cfg)
    g = torch.Generator().manual_seed(3)
    colour = {}
    for k, v in src.state_dict().items():
        if 'disp_' in k or not v.is_floating_point():
            continue                      # a colour-image checkpoint has no disparity branch
        colour[k] = torch.randn(v.shape, generator=g)
    k_cls = 'bbox_head.head_module.multi_level_conv_cls.0.weight'
    colour[k_cls] = torch.randn(80, *colour[k_cls].shape[1:], generator=g)          # an 80-class (COCO) head
    colour['ema_something.not_in_the_model'] = torch.zeros(3)
    path = str(tmp_path / 'yolox_colour.pth')
    torch.save(dict(state_dict={'module.' + k: v for k, v in colour.items()}, meta=dict(epoch=300)), path)

    plain = MODELS.build(det_cfg)
    ck = load_checkpoint(plain, path)
    rep = ck['_load_report']
    assert ck['meta']['epoch'] == 300 and [m[0] for m in rep['mismatched']] == [k_cls]
    assert rep['unexpected'] == ['ema_something.not_in_the_model']
    assert all('disp_' in k or k == k_cls or k.endswith('num_batches_tracked') for k in rep['missing'])
    assert torch.equal(plain.state_dict()['backbone.stem.conv.conv.weight'], colour['backbone.stem.conv.conv.weight'])
    assert float(plain.state_dict()['backbone.disp_stem.conv.conv.weight'].abs().sum()) == 0     # untouched

    stripped = str(tmp_path / 'yolox_colour_plain_keys.pth')
    torch.save(dict(state_dict=colour), stripped)
    det = MODELS.build(dict(det_cfg, init_cfg=dict(type='ColorPretrained', checkpoint=stripped)))
    rep = det.init_weights()
    sd = det.state_dict()
    for a, b in (('backbone.stem.conv.conv.weight', 'backbone.disp_stem.conv.conv.weight'),
                 ('backbone.stage1.0.conv.weight', 'backbone.disp_stage1.0.conv.weight'),
                 ('backbone.stage1.1.final_conv.bn.running_var', 'backbone.disp_stage1.1.final_conv.bn.running_var')):
        assert torch.equal(sd[a], colour[a]) and torch.equal(sd[b], colour[a]), (a, b)
    assert not any('disp_' in k and not k.endswith('num_batches_tracked') for k in rep['missing'])
    assert [m[0] for m in rep['mismatched']] == [k_cls]
    assert torch.equal(sd['backbone.stage2.0.conv.weight'], colour['backbone.stage2.0.conv.weight'])

    shipped = MODELS.build(dict(cfg.model.detector))          # the shipped init_cfg names a URL
    if shipped.init_cfg:
        with pytest.raises(RuntimeError, match='local|download'):
            shipped.init_weights()


def test_full_resolution_mode_rejects_2d_aggregation_and_bad_sizes():
    from stereotracking_amd.stereo import StereoCostVolume
    with pytest.raises(ValueError, match='agg_layers'):
        StereoCostVolume(192, 4, 32.0, agg_layers=1, full_res=True)
    with pytest.raises(ValueError, match='multiple of 16'):
        StereoCostVolume(200, 4, 32.0, full_res=True)
    m = StereoCostVolume(192, 4, 32.0, agg3d_layers=1, full_res=True)
    assert m.levels == 192 and [n for n, _ in m.param_table()] == ['reduce.weight', 'reduce.bias', 'agg3d.0.weight',
                                                                    'agg3d.0.bias']


def test_detection_gatherer_single_rank_collective_gloo():
    """dist.DetectionGatherer(single_rank_collective=True) with a process group of ONE rank issues the collective (here
    through gloo / host memory; on the GPU box through RCCL: tests/test_multirank_gpu.py) and returns the records
    unchanged; without the flag, or without a process group, a single rank returns its records as they are."""
    import socket
    import torch.distributed as dist
    from stereotracking_amd.dist import DetectionGatherer
    rec = torch.arange(2 * 5 * 8, dtype=torch.float32).view(2, 5, 8)
    g0 = DetectionGatherer()
    out, ev = g0.gather(rec)
    assert out is rec and ev is None and g0.seq == 1
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    dist.init_process_group('gloo', init_method=f'tcp://127.0.0.1:{port}', rank=0, world_size=1)
    try:
        g1 = DetectionGatherer(single_rank_collective=True)
        assert g1.single_rank_collective and not g1.on_device
        out, ev = g1.gather(rec)
        assert out is not rec and torch.equal(out, rec) and g1.seq == 1
        g2 = DetectionGatherer()            # a single rank without the flag: no collective
        assert g2.gather(rec)[0] is rec
    finally:
        dist.destroy_process_group()


# ---- bench.py --gpus N without a launcher (a --gpus N run may never come back as ONE rank) -------------
def _bench(args, env_extra=None, drop=('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT')):
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, **(env_extra or {}))
    for k in drop:
        if k not in (env_extra or {}):
            env.pop(k, None)
    return subprocess.run([sys.executable, os.path.join(root, 'bench.py')] + args, cwd=root, env=env, capture_output=True,
                          text=True, timeout=300)


def test_bench_gpus_n_without_launcher_refuses_fewer_devices():
    """`python bench.py --gpus 2` with fewer than 2 visible devices on the RCCL backend is an ERROR before any rank
    starts - never a 1-rank measurement printed as if it were the N-rank one."""
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip('two devices visible: the refusal cannot be provoked')
    p = _bench(['--gpus', '2', '--steps', '1', '--warmup', '0'])
    assert p.returncode == 2 and p.stdout.strip() == ''
    assert 'device(s) visible' in p.stderr and 'refusing' in p.stderr


def test_bench_gpus_n_without_launcher_starts_n_ranks_and_relays_their_failure():
    """Without RANK / WORLD_SIZE in the environment the parent starts the N ranks itself (torch.distributed.run, one
    process per rank).  Here (no GPU) the ranks die with 'bench.py needs a GPU': the 2-rank launch must have happened, the
    parent must exit non-zero and print no JSON line.  (The successful form - n_gpus: 2 on one card over gloo - is
    tests/test_multirank_gpu.py::test_bench_gpus2_without_launcher_starts_two_ranks.)"""
    import torch
    if torch.cuda.is_available():
        pytest.skip('a GPU is present: covered by the -m gpu form of this test')
    p = _bench(['--gpus', '2', '--steps', '1', '--warmup', '0'], dict(ST_BENCH_BACKEND='gloo'))
    assert p.returncode != 0 and p.stdout.strip() == ''
    assert 'starting 2 ranks' in p.stderr and '--nproc-per-node=2' in p.stderr
    # (torch.distributed.run may end the second rank as soon as the first has failed: one or two such messages)
    assert 1 <= p.stderr.count('bench.py needs a GPU') <= 2, p.stderr[-3000:]
    assert 'the 2-rank run failed' in p.stderr


def test_bench_under_a_launcher_with_the_wrong_world_size_is_an_error():
    p = _bench(['--gpus', '2', '--steps', '1', '--warmup', '0'], dict(RANK='0', LOCAL_RANK='0', WORLD_SIZE='1'))
    assert p.returncode != 0 and p.stdout.strip() == '' and 'WORLD_SIZE=1' in p.stderr
    p = _bench(['--gpus', '1', '--steps', '1', '--warmup', '0'], dict(RANK='0', LOCAL_RANK='0', WORLD_SIZE='2'))
    assert p.returncode != 0 and p.stdout.strip() == '' and 'WORLD_SIZE=2' in p.stderr
