"""GPU: the HIP path against the committed golden vectors, and the config-built plugin surface
(MODELS.build -> OCSORT_Disparity.test_step) end to end against the oracle composition."""
import math
import os

import numpy as np
import pytest
import torch

from oracle import c_oracle
from oracle import stereo as ostereo
from oracle import depth as odepth
from oracle.torch_model import OracleDetector, head_to_rows
from parity_utils import rel_err
from stereotracking_amd import _lib
from stereotracking_amd._lib import check, current_stream, ptr
from stereotracking_amd.engine import HipDetector
from stereotracking_amd.synthetic import synthetic_batch, synthetic_state_dict

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, 'tests', 'golden')
CFG = os.path.join(ROOT, 'configs', 'stereo_tracking', 'ocsort', 'yolox_s_mmyolo_mot_airdrone_disp.py')
CFG_RGB = os.path.join(ROOT, 'configs', 'stereo_tracking', 'ocsort', 'yolox_s_mmyolo_mot_airdrone.py')
CFG_FULLRES = os.path.join(ROOT, 'configs', 'stereo_tracking', 'ocsort', 'stereo_yolox_s_mot_airdrone_costvolume_fullres.py')
CFG_STEREO = os.path.join(ROOT, 'configs', 'stereo_tracking', 'ocsort', 'stereo_yolox_s_mot_airdrone_costvolume.py')


def test_golden_detector(cuda):
    g = np.load(os.path.join(GOLD, 'detector_tiny.npz'))
    det = HipDetector(1, 64, 96, 0.375, 0.33, 1)
    det.load_state_dict(synthetic_state_dict(det.param_table(), seed=int(g['weights_seed'])))
    batch = synthetic_batch([int(g['input_seed'])], 48, 96, 32)
    head = det.forward(batch['img'].to(cuda), batch['disp_postp'].to(cuda))
    torch.cuda.synchronize()
    for l, rows in enumerate(det.head_levels(head)):
        ref = g[f'head{l}']
        err = np.abs(rows[..., :6].cpu().numpy() - ref) / np.maximum(np.abs(ref), 1.0)
        assert err.max() <= 1e-3


def test_golden_decode_nms_bit_exact(cuda):
    g = np.load(os.path.join(GOLD, 'decode_nms.npz'))
    det = HipDetector(2, 160, 256, 0.375, 0.33, 1)
    assert [tuple(l) for l in det.levels] == [tuple(int(v) for v in l) for l in g['levels']]
    b, s, l, p, c = det.decode_nms(torch.from_numpy(g['head']).to(cuda), float(g['score_thr']), float(g['iou_thr']),
                                   840, (int(g['ori_h']), int(g['ori_w'])))
    torch.cuda.synchronize()
    assert np.array_equal(c.cpu().numpy(), g['counts'])
    k = int(g['counts'].max())
    for n in range(2):
        kn = int(g['counts'][n])
        assert np.array_equal(p[n, :kn].cpu().numpy(), g['prior'][n, :kn])
        assert np.array_equal(b[n, :kn].cpu().numpy(), g['boxes'][n, :kn])
        assert np.array_equal(s[n, :kn].cpu().numpy(), g['scores'][n, :kn])
    assert k <= 840


def test_golden_costvolume_and_box_depth(cuda):
    lib = _lib.load()
    g = np.load(os.path.join(GOLD, 'costvolume.npz'))
    fl, fr = torch.from_numpy(g['featL']).to(cuda), torch.from_numpy(g['featR']).to(cuda)
    N, Hf, Wf, Cc = fl.shape
    D = g['cost'].shape[-1]
    cost = torch.empty(N, Hf, Wf, D, device=cuda)
    lr = torch.empty(N, Hf, Wf, device=cuda)
    check(lib.st_costvolume_softargmin(ptr(fl), ptr(fr), N, Hf, Wf, Cc, Cc, D, float(g['temperature']), ptr(cost),
                                       ptr(lr), current_stream()))
    up = torch.empty(N, 3, Hf * 4, Wf * 4, device=cuda)
    check(lib.st_disp_upsample_pack(ptr(lr), N, Hf, Wf, 4, Hf * 4, Wf * 4, Hf * 4 - 8, Wf * 4, ptr(up),
                                    current_stream()))
    torch.cuda.synchronize()
    assert np.array_equal(cost.cpu().numpy(), g['cost'])
    assert np.abs(lr.cpu().numpy() - g['disp_lr']).max() <= 1e-3
    assert rel_err(up.cpu().numpy(), g['disp_postp']) <= 1e-3

    g = np.load(os.path.join(GOLD, 'box_depth.npz'))
    H, W = g['disp'].shape
    M = len(g['boxes'])
    disp3 = torch.from_numpy(np.repeat(g['disp'][None, None], 3, 1)).to(cuda)
    depth, scale, sb = (torch.empty(1, M, device=cuda), torch.empty(1, M, device=cuda),
                        torch.empty(1, M, 4, device=cuda))
    boxes_dev = torch.from_numpy(g['boxes'])[None].to(cuda)  # keep alive: ptr() does not hold a reference
    counts_dev = torch.tensor([M], dtype=torch.int32, device=cuda)
    check(lib.st_box_depth(ptr(disp3), 3 * H * W, 1, H, W, ptr(boxes_dev), ptr(counts_dev), M, 0.25, 640.0, None, 0,
                           current_stream(), ptr(depth), ptr(scale), ptr(sb)))
    torch.cuda.synchronize()
    d, ref = depth[0].cpu().numpy(), g['depth']
    assert np.array_equal(np.isnan(d), np.isnan(ref)) and np.array_equal(d == -1, ref == -1)
    ok = ~np.isnan(ref)
    assert rel_err(d[ok], ref[ok]) <= 1e-3
    assert np.abs(scale[0].cpu().numpy()[ok] - g['scales'][ok]).max() <= 1e-3
    assert rel_err(sb[0].cpu().numpy()[ok], g['scaled_boxes'][ok]) <= 1e-3


# ---- plugin surface end to end --------------------------------------------------------------------------------
def build_model(cfg_path, cuda, widen=0.375, seed=5, prior_prob=0.2, autotune=True):
    from stereotracking_amd import mot  # noqa: F401
    from stereotracking_amd.config import Config
    from stereotracking_amd.registry import MODELS
    cfg = Config.fromfile(cfg_path)
    for part in ('backbone', 'neck'):
        cfg.model.detector[part]['widen_factor'] = widen
    cfg.model.detector.bbox_head.head_module['widen_factor'] = widen
    # random weights give low scores: lower the tracker's score gates so tracks are actually started/matched
    cfg.model.tracker['init_track_thr'] = 0.03
    cfg.model.tracker['obj_score_thr'] = 0.02
    model = MODELS.build(dict(cfg.model, autotune=autotune))
    # confident synthetic head so that tracks get started (score > init_track_thr = 0.7)
    table = list(model.detector._table)
    if model.stereo is not None:
        table += [('stereo.' + n, shp) for n, shp in model.stereo.param_table()]
    sd = synthetic_state_dict(table, seed=seed, prior_prob=prior_prob, logit_std=2.5)
    model.detector.load_state_dict(sd, strict=False)
    if model.stereo is not None:
        model.stereo.load_state_dict({k[len('stereo.'):]: v for k, v in sd.items() if k.startswith('stereo.')})
        assert any(k.startswith('stereo.agg.') for k in model.state_dict()) == (model.stereo.agg_layers > 0)
        assert not any(k.startswith('detector.stereo.') for k in model.state_dict())
    return model, sd, cfg


def oracle_frame(ora, img, disp, ori_hw, score_thr=0.01, iou_thr=0.5):
    """Oracle composition for one frame -> boxes, scores, depth, scales, scaled boxes."""
    H, W = img.shape[-2:]
    with torch.no_grad():
        rows = head_to_rows(*ora(dict(img=img, disp_postp=disp)))
    levels, off, flat = [], 0, []
    for r, s in zip(rows, (8, 16, 32)):
        h, w = H // s, W // s
        buf = torch.zeros(1, h * w, 8)
        buf[..., :6] = r
        levels.append((h, w, s, off))
        off += h * w * 8
        flat.append(buf.reshape(-1))
    b, s, l, p, c = c_oracle.decode_nms(torch.cat(flat).numpy(), 1, levels, score_thr, iou_thr, 1000, ori_hw)
    k = int(c[0])
    boxes = torch.from_numpy(b[0, :k])
    d, sc, sb = odepth.bbox_postp_depth(boxes, disp)
    return boxes, torch.from_numpy(s[0, :k]), p[0, :k], torch.tensor([float(v) for v in d]), sc, sb


def test_mot_shell_matches_oracle_composition(cuda):
    """model.test_step (config-built plugin surface, batched dense path underneath) frame by frame against the ORACLE
    composition: oracle detector + C decode/NMS + numpy extract_depth (evaluated live) feeding the ORACLE tracker, whose
    ids / boxes come from tests/golden/shell_sequence.npz (generated in the build container by make_golden.py: the
    restatement of the reference's tracker classes does not travel to the GPU box)."""
    from parity_utils import unscale_boxes_np
    from stereotracking_amd.structures import TrackDataSample
    g = np.load(os.path.join(GOLD, 'shell_sequence.npz'))
    model, sd, cfg = build_model(CFG, cuda)
    ora = OracleDetector(0.33, 0.375, 1).eval()
    ora.load_state_dict(sd, strict=False)
    ori = (80, 160)
    frames = [synthetic_batch([40 + (t // 2)], ori[0], ori[1], 32) for t in range(6)]  # pairs of identical frames
    ref_tracks = g['tracks']              # rows [t, id, scaled box (4), score, depth, scale] of the oracle tracker
    n_tracked = 0
    for t, fr in enumerate(frames):
        sample = TrackDataSample(dict(frame_id=t, ori_shape=ori, img_shape=ori, scale_factor=(1.0, 1.0)))
        data = dict(inputs=dict(img=[fr['img'][0:1, :, :ori[0]].to(torch.uint8)],   # un-padded uint8, as a dataset yields
                                disp_postp=[fr['disp_postp'][0:1, :, :ori[0]]],
                                disp_mask=[fr['disp_mask'][0:1, :, :ori[0]].to(torch.uint8)]),
                    data_samples=[sample])
        # the preprocessor pads with 0, the dataset pipeline is what pads the image with 114 (Pad_Disparity):
        # feed the oracle exactly what the model sees
        out = model.test_step(data)[0]
        torch.cuda.synchronize()
        img_seen = torch.nn.functional.pad(fr['img'][0:1, :, :ori[0]].to(torch.uint8).float(), [0, 0, 0, 16])
        boxes, scores, prior, depth, scales, sboxes = oracle_frame(ora, img_seen, fr['disp_postp'][0:1], ori)
        assert np.array_equal(prior, g[f'prior{t}']), 'the live oracle and the fixture disagree on the kept priors'
        det = out.pred_det_instances
        assert len(det) == len(boxes) and len(boxes) > 0
        assert np.array_equal(det.prior_idx.cpu().numpy(), prior), 'kept prior indices differ'
        assert rel_err(det.bboxes.cpu(), boxes) <= 1e-3
        assert (det.scores.cpu() - scores).abs().max() <= 1e-3
        # the ORACLE tracker fed the ORACLE detections (fixture) must give the same ids / boxes as the HIP shell
        rt = ref_tracks[ref_tracks[:, 0] == t]
        trk = out.pred_track_instances
        assert trk.instances_id.cpu().tolist() == rt[:, 1].astype(np.int64).tolist()
        n_tracked += len(rt)
        if len(rt):
            assert rel_err(trk.bboxes.cpu(), unscale_boxes_np(rt[:, 2:6], rt[:, 8])) <= 1e-3
            assert set(trk.keys()) >= {'bboxes', 'labels', 'scores', 'scales', 'depth', 'gt_depth', 'instances_id'}
    assert n_tracked > 0, 'the scenario must actually exercise the association step'


def test_rgb_only_config_shell_matches_oracle_composition(cuda):
    """The reference's second config under configs/stereo_tracking (yolox_s_mmyolo_mot_airdrone.py:29-58): detector
    `mmyolo.YOLODetector` over `mmtrack.CSPDarknet` (image branch only), the SAME OCSORT_Disparity shell, the loaded
    disparity consumed by the per-box depth alone (ocsort_disparity.py:82-83).  model.test_step frame by frame against
    the oracle composition with the branch disabled: kept priors equal, boxes / scores within 1e-3, and the depth of the
    detections (which the tracker's scaled boxes are built from) equal to oracle/depth.py on the oracle's boxes."""
    from stereotracking_amd.structures import TrackDataSample
    model, sd, cfg = build_model(CFG_RGB, cuda)
    assert type(model.detector).__name__ == 'YOLODetector' and model.detector.rgb_only
    assert not any('disp_' in k for k in model.state_dict())
    ora = OracleDetector(0.33, 0.375, 1, rgb_only=True).eval()
    missing, unexpected = ora.load_state_dict(sd, strict=False)
    assert not unexpected and all(k.endswith('num_batches_tracked') for k in missing)
    ori = (80, 160)
    frames = [synthetic_batch([60 + (t // 2)], ori[0], ori[1], 32) for t in range(6)]
    n_tracked = 0
    for t, fr in enumerate(frames):
        sample = TrackDataSample(dict(frame_id=t, ori_shape=ori, img_shape=ori, scale_factor=(1.0, 1.0)))
        data = dict(inputs=dict(img=[fr['img'][0:1, :, :ori[0]].to(torch.uint8)],
                                disp_postp=[fr['disp_postp'][0:1, :, :ori[0]]],
                                disp_mask=[fr['disp_mask'][0:1, :, :ori[0]].to(torch.uint8)]),
                    data_samples=[sample])
        out = model.test_step(data)[0]
        torch.cuda.synchronize()
        img_seen = torch.nn.functional.pad(fr['img'][0:1, :, :ori[0]].to(torch.uint8).float(), [0, 0, 0, 16])
        boxes, scores, prior, depth, scales, sboxes = oracle_frame(ora, img_seen, fr['disp_postp'][0:1], ori)
        det = out.pred_det_instances
        assert len(det) == len(boxes) and len(boxes) > 0
        assert np.array_equal(det.prior_idx.cpu().numpy(), prior), 'kept prior indices differ'
        assert rel_err(det.bboxes.cpu(), boxes) <= 1e-3
        assert (det.scores.cpu() - scores).abs().max() <= 1e-3
        trk = out.pred_track_instances
        n_tracked += len(trk)
        if len(trk):
            assert set(trk.keys()) >= {'bboxes', 'labels', 'scores', 'scales', 'depth', 'gt_depth', 'instances_id'}
            assert bool(torch.isfinite(trk.depth.cpu()).any())    # the loaded disparity reached the depth step
    assert n_tracked > 0


def test_full_resolution_stereo_config_through_the_shell(cuda):
    """configs/.../stereo_yolox_s_mot_airdrone_costvolume_fullres.py -> MODELS.build -> model.test_step on left / right
    frames: the shell's detections, depths and disparity equal what StereoDensePipeline(full_res=True) gives on the same
    frames with the same weights (the module itself is pinned against the oracle in test_stereo_depth_gpu.py)."""
    from stereotracking_amd.pipeline import StereoDensePipeline
    from stereotracking_amd.structures import TrackDataSample
    from stereotracking_amd import mot  # noqa: F401
    from stereotracking_amd.config import Config
    from stereotracking_amd.registry import MODELS
    cfg = Config.fromfile(CFG_FULLRES)
    for part in ('backbone', 'neck'):
        cfg.model.detector[part]['widen_factor'] = 0.375
    cfg.model.detector.bbox_head.head_module['widen_factor'] = 0.375
    cfg.model.stereo['max_disp'] = 32
    cfg.model.tracker['init_track_thr'], cfg.model.tracker['obj_score_thr'] = 0.03, 0.02
    model = MODELS.build(dict(cfg.model, autotune=False, dense_batch=2, inflight=2))
    assert model.stereo.full_res and model.stereo.reduce.in_channels == 48 and model.stereo.levels == 32
    table = list(model.detector._table) + [('stereo.' + n, shp) for n, shp in model.stereo.param_table()]
    sd = synthetic_state_dict(table, seed=8, prior_prob=0.2, logit_std=2.5)
    model.detector.load_state_dict(sd, strict=False)
    model.stereo.load_state_dict({k[len('stereo.'):]: v for k, v in sd.items() if k.startswith('stereo.')})
    ori = (80, 160)
    frames = [synthetic_batch([70 + t], ori[0], ori[1], 32) for t in range(4)]
    data = dict(inputs=dict(img=[f['img'][0:1, :, :ori[0]].to(torch.uint8) for f in frames],
                            right=[f['right'][0:1, :, :ori[0]].to(torch.uint8) for f in frames]),
                data_samples=[TrackDataSample(dict(frame_id=t, ori_shape=ori, img_shape=ori, scale_factor=(1.0, 1.0)))
                              for t in range(4)])
    outs = model.test_step(data)
    torch.cuda.synchronize()
    pipe = StereoDensePipeline(2, ori, 0.375, 0.33, 1, stereo=True, max_disp=32, max_det=model.max_det, agg3d_layers=1,
                               full_res=True)
    pipe.load_state_dict(sd, autotune=False)
    for c in range(2):
        l = torch.cat([torch.nn.functional.pad(frames[2 * c + i]['img'][0:1, :, :ori[0]].to(torch.uint8).float(), [0, 0, 0, 16],
                                               value=0.0) for i in range(2)]).to(cuda)
        r = torch.cat([torch.nn.functional.pad(frames[2 * c + i]['right'][0:1, :, :ori[0]].to(torch.uint8).float(), [0, 0, 0, 16],
                                               value=0.0) for i in range(2)]).to(cuda)
        ref = pipe.run(l, r)
        torch.cuda.synchronize()
        for i in range(2):
            det = outs[2 * c + i].pred_det_instances
            k = int(ref['counts'][i])
            assert len(det) == k and k > 0
            assert torch.equal(det.bboxes.cpu(), ref['boxes'][i, :k].cpu())
            assert torch.equal(det.scores.cpu(), ref['scores'][i, :k].cpu())


def test_batched_predict_equals_sequential_and_stereo_module(cuda):
    from stereotracking_amd.structures import TrackDataSample
    # Bit equality between one 3-frame call and three 1-frame calls is a property of ONE kernel plan: the model
    # builds (and would autotune) a launch plan per batch size, and two plans that pick different kernel instances for
    # a layer (Winograd vs implicit GEMM, ...) differ by fp32 summation order.  The heuristic plan is the same for both.
    model, sd, _ = build_model(CFG_STEREO, cuda, autotune=False)
    ori = (80, 160)
    fr = synthetic_batch([50, 51, 52], ori[0], ori[1], 32)

    def samples():
        return [TrackDataSample(dict(frame_id=t, ori_shape=ori, scale_factor=(1.0, 1.0))) for t in range(3)]

    inputs = dict(img=fr['img'][:, None].to(cuda), right=fr['right'][:, None].to(cuda))
    batched = model.forward(dict(inputs), samples(), mode='predict')
    model.tracker.reset()
    seq = []
    for t in range(3):
        one = {k: v[t:t + 1] for k, v in inputs.items()}
        seq += model.forward(one, samples()[t:t + 1], mode='predict')
    torch.cuda.synchronize()
    for a, b in zip(batched, seq):
        assert a.pred_det_instances.prior_idx.tolist() == b.pred_det_instances.prior_idx.tolist()
        assert torch.equal(a.pred_det_instances.bboxes, b.pred_det_instances.bboxes)
        assert a.pred_track_instances.instances_id.tolist() == b.pred_track_instances.instances_id.tolist()
    # the stereo module's disparity equals the oracle's on the same features
    ora = OracleDetector(0.33, 0.375, 1).eval()
    ora.load_state_dict(sd, strict=False)
    with torch.no_grad():
        fl = ora.backbone.stage1_features(fr['img']).permute(0, 2, 3, 1).contiguous().numpy()
        frr = ora.backbone.stage1_features(fr['right']).permute(0, 2, 3, 1).contiguous().numpy()
    assert model.stereo.agg_layers == 2   # the shipped stereo config aggregates the volume with two 3x3 convs
    ref = ostereo.disparity(fl, frr, fl.shape[-1], model.stereo.levels, model.stereo.temperature, sd,
                            model.stereo.agg_layers, valid_hw=ori)[2]
    data = dict(img=inputs['img'][:, 0], right=inputs['right'][:, 0])
    model.detector._run(data, ori)
    torch.cuda.synchronize()
    assert rel_err(data['disp_postp'].cpu().numpy(), ref) <= 1e-3


def test_frames_without_detections_and_ragged_calls(cuda):
    """Edge cases of the shell: a score threshold nothing passes (every frame keeps 0 boxes: empty detection and track
    containers, no crash in the chunked association / depth launches), a call whose frame count is not a multiple
    of the dense batch, and a one-frame call - through the raw uint8 path (RawFrames) of test_step."""
    from stereotracking_amd import mot  # noqa: F401
    from stereotracking_amd.config import Config
    from stereotracking_amd.registry import MODELS
    from stereotracking_amd.structures import TrackDataSample
    cfg = Config.fromfile(CFG_STEREO)
    for part in ('backbone', 'neck'):
        cfg.model.detector[part]['widen_factor'] = 0.375
    cfg.model.detector.bbox_head.head_module['widen_factor'] = 0.375
    cfg.model.stereo['max_disp'] = 32
    cfg.model.detector.test_cfg['score_thr'] = 0.9999
    model = MODELS.build(dict(cfg.model, autotune=False, dense_batch=4, inflight=2))
    table = list(model.detector._table) + [('stereo.' + n, shp) for n, shp in model.stereo.param_table()]
    sd = synthetic_state_dict(table, seed=5)
    model.detector.load_state_dict(sd, strict=False)
    model.stereo.load_state_dict({k[len('stereo.'):]: v for k, v in sd.items() if k.startswith('stereo.')})
    ori = (80, 160)
    fr = synthetic_batch(list(range(60, 67)), ori[0], ori[1], 32)     # 7 frames: chunks of 4 + 3
    left = [fr['img'][i:i + 1, :, :ori[0]].to(torch.uint8).to(cuda) for i in range(7)]
    right = [fr['right'][i:i + 1, :, :ori[0]].to(torch.uint8).to(cuda) for i in range(7)]
    for lo, hi in ((0, 7), (0, 1)):
        samples = [TrackDataSample(dict(frame_id=t, ori_shape=ori, scale_factor=(1.0, 1.0))) for t in range(lo, hi)]
        outs = model.test_step(dict(inputs=dict(img=left[lo:hi], right=right[lo:hi]), data_samples=samples))
        torch.cuda.synchronize()
        assert len(outs) == hi - lo
        for o in outs:
            assert len(o.pred_det_instances) == 0 and len(o.pred_track_instances) == 0
            assert tuple(o.pred_det_instances.bboxes.shape) == (0, 4)
            assert set(o.pred_track_instances.keys()) >= {'bboxes', 'labels', 'scores', 'scales', 'depth', 'gt_depth',
                                                          'instances_id'}
            assert o.metainfo['batch_input_shape'] == (96, 160) and o.metainfo['pad_shape'] == ori


@pytest.mark.parametrize('frames_per_call,inflight,queue_depth', [(12, 3, 2), (4, 2, 1), (9, 3, 2), (30, 1, 2), (8, 2, 3)])
def test_primed_loop_equals_per_call_test_step(frames_per_call, inflight, queue_depth, cuda):
    """model.test_steps(iterable) - contexts kept primed across calls: call k+1's first chunks are submitted while call
    k drains - returns, call by call, exactly what model.test_step returns for the same calls (boxes, scores, ids,
    track depth: bit-equal), over a 2-video stream whose calls are ragged against the dense batch.  The two runs also
    differ in how the uint8 frames reach the stems (a cast + pad pass vs the stem's own raw staging): same bits."""
    from stereotracking_amd.structures import TrackDataSample
    model, _, _ = build_model(CFG_STEREO, cuda, autotune=False)
    model.dense_batch, model.inflight, model.queue_depth = 4, inflight, queue_depth
    ori = (80, 160)
    T = 30
    fr = synthetic_batch(list(range(200, 200 + T)), ori[0], ori[1], 32)
    left = [fr['img'][i:i + 1, :, :ori[0]].to(torch.uint8).to(cuda) for i in range(T)]
    right = [fr['right'][i:i + 1, :, :ori[0]].to(torch.uint8).to(cuda) for i in range(T)]

    def calls():
        for lo in range(0, T, frames_per_call):
            hi = min(T, lo + frames_per_call)
            # two videos: frame ids restart at frame 17 (the tracker resets on frame_id 0)
            samples = [TrackDataSample(dict(frame_id=t if t < 17 else t - 17, ori_shape=ori, scale_factor=(1.0, 1.0)))
                       for t in range(lo, hi)]
            yield dict(inputs=dict(img=left[lo:hi], right=right[lo:hi]), data_samples=samples)

    model.raw_stem = False       # per-call test_step through st_pack_raw_frames (cast + pad as a pass of its own) ...
    ref = [model.test_step(d) for d in calls()]
    torch.cuda.synchronize()
    model.tracker.reset()
    model.raw_stem = True        # ... against the primed loop with the uint8 frames read by the stem kernels directly
    got = list(model.test_steps(calls()))
    torch.cuda.synchronize()
    assert [len(c) for c in got] == [len(c) for c in ref]
    n_tracks = 0
    for ca, cb in zip(got, ref):
        for a, b in zip(ca, cb):
            assert a.metainfo['frame_id'] == b.metainfo['frame_id']
            assert torch.equal(a.pred_det_instances.bboxes, b.pred_det_instances.bboxes)
            assert torch.equal(a.pred_det_instances.scores, b.pred_det_instances.scores)
            ta, tb = a.pred_track_instances, b.pred_track_instances
            assert ta.instances_id.tolist() == tb.instances_id.tolist()
            for k in ('bboxes', 'scores', 'depth', 'gt_depth', 'scales'):
                assert torch.equal(ta[k], tb[k]), k
            n_tracks += len(ta)
    assert n_tracks > 0
