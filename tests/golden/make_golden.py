#!/usr/bin/env python
"""Generates the golden fixtures in this directory FROM THE ORACLE (oracle/), on the CPU.

The reference cannot be imported here or on the GPU box (mmcv/mmdet/mmyolo/mmengine absent) and
ships no fixtures of its own, so these vectors pin the oracle, not the reference:
  detector_tiny.npz    head rows of the CPU PyTorch oracle, tiny two-branch YOLOX (widen .375), 1x64x96
  decode_nms.npz       C-oracle decode+NMS on a seeded random head (840 priors x 2 images)
  box_depth.npz        numpy restatement of extract_depth (reference ocsort_disparity.py:136-175) on a
                       structured disparity map, incl. the NaN / -1 / w>800-style branches
  costvolume.npz       C-oracle cost volume + soft-argmin + upsample on seeded features
  tracker_sequence.npz 64-frame synthetic detection stream (SURVEY.md §8c fixture iv: 6 objects, dropped
                       detections, an 8-frame occlusion, depth-consistent scales) and the frame-by-frame output of
                       the ORACLE tracker (oracle/tracker.py: statement-by-statement restatement of reference
                       ocsort_tracker_disparity.py:20-618, kalman_tracker_base.py:19-88, base_tracker.py:10-141,
                       kalman_filter.py:38-189) with the SHIPPED tracker config: ids, boxes, scores, depth, scales
  lapjv_ties.npz       <= 7x7 assignment problems with NON-UNIQUE optima: all optimal assignments enumerated by
                       brute force, the one oracle/lapjv.py (restatement of lap.lapjv) picks is pinned
  config2_sequence.npz BASELINE configs[2] END TO END through the oracle, at its stated size: 24 frames of the synthetic
                       1280x720 sequence (D=192, full YOLOX-s two-branch, 2 aggregation convs) -> oracle stereo module
                       -> oracle detector on the oracle's OWN disparity -> C decode+NMS -> numpy extract_depth -> ORACLE
                       tracker, once with the SHIPPED thresholds and once with stress thresholds (hundreds of
                       tracks).  Per frame: kept prior indices (in order), scores, boxes, depth, scales; per
                       tracker config: ids / boxes per frame.  The GPU test compares model.test_step against it, so
                       oracle/tracker.py never has to run on the GPU box.
  shell_sequence.npz   the 6-frame 80x160 scenario of tests/test_shell_gpu.py (disparity given as input, tiny detector):
                       oracle detector -> C decode+NMS -> numpy extract_depth -> ORACLE tracker; kept priors, track ids
                       and unscaled track boxes per frame
Nothing here is produced by product code (stereotracking_amd/ supplies only the seeded synthetic INPUTS).
Run:  python tests/golden/make_golden.py [name ...]   (deterministic; CI checks the files are reproduced)."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import c_oracle, depth as odepth  # noqa: E402
from oracle.torch_model import OracleDetector, head_to_rows  # noqa: E402
from stereotracking_amd.synthetic import synthetic_batch, synthetic_state_dict  # noqa: E402


def detector_tiny():
    torch.manual_seed(0)
    torch.set_num_threads(1)
    ora = OracleDetector(0.33, 0.375, 1).eval()
    table = [(k, tuple(v.shape)) for k, v in ora.state_dict().items() if not k.endswith('num_batches_tracked')]
    sd = synthetic_state_dict(table, seed=7)
    ora.load_state_dict(sd, strict=False)
    batch = synthetic_batch([11], 48, 96, 32)  # padded to 64 x 96
    with torch.no_grad():
        rows = head_to_rows(*ora(batch))
        feats = ora.backbone(batch)
    out = {f'head{l}': r.numpy() for l, r in enumerate(rows)}
    out.update({f'stage{i + 2}_sum': np.float64(f.double().sum().item()) for i, f in enumerate(feats)})
    out['img_sum'] = np.float64(batch['img'].double().sum().item())
    out['disp_sum'] = np.float64(batch['disp_postp'].double().sum().item())
    out['weights_seed'], out['input_seed'] = 7, 11
    return out


def levels_for(H, W, N):
    lv, off = [], 0
    for s in (8, 16, 32):
        h, w = H // s, W // s
        lv.append((h, w, s, off))
        off += N * h * w * 8
    return lv, off


def random_head(levels, total, N, rng, mean=-2.0, std=2.0):
    head = np.zeros(total, np.float32)
    for h, w, s, off in levels:
        rows = head[off:off + N * h * w * 8].reshape(N, h * w, 8)
        rows[..., 0] = rng.normal(mean, std, rows.shape[:2])
        rows[..., 5] = rng.normal(mean, std, rows.shape[:2])
        rows[..., 1:3] = rng.normal(0, 1.0, rows.shape[:2] + (2,))
        rows[..., 3:5] = rng.normal(0.5, 0.8, rows.shape[:2] + (2,))
    return head


def decode_nms():
    N, H, W = 2, 160, 256
    levels, total = levels_for(H, W, N)
    head = random_head(levels, total, N, np.random.RandomState(21))
    b, s, l, p, c = c_oracle.decode_nms(head, N, levels, 0.01, 0.5, 840, (H - 10, W), (1.0, 1.0), None)
    k = int(c.max())
    return dict(head=head, levels=np.array(levels, np.int64), boxes=b[:, :k], scores=s[:, :k], prior=p[:, :k],
                counts=c, score_thr=0.01, iou_thr=0.5, ori_h=H - 10, ori_w=W)


def box_depth():
    rng = np.random.RandomState(31)
    H, W = 96, 160
    disp = np.full((H, W), 2.0, np.float32) + rng.uniform(0, 0.5, (H, W)).astype(np.float32)
    for _ in range(8):
        y, x = rng.randint(0, H - 30), rng.randint(0, W - 40)
        h, w = rng.randint(5, 30), rng.randint(5, 40)
        disp[y:y + h, x:x + w] = rng.uniform(3, 40) + rng.uniform(0, 0.3, (h, w))
    disp[rng.uniform(size=(H, W)) < 0.05] = 0.0
    disp[80:90, 100:120] = 0.0
    disp[20:50, 20:70] = 120.0 + rng.uniform(0, 8.0, (30, 50)).astype(np.float32)  # near object: 1 < depth^2 < 3
    boxes = [(22.0, 22.0, 68.0, 48.0), (18.5, 15.0, 75.0, 55.0)]
    for _ in range(24):
        x1, y1 = rng.uniform(0, W - 6), rng.uniform(0, H - 6)
        boxes.append((x1, y1, min(W, x1 + rng.uniform(1, 70)), min(H, y1 + rng.uniform(1, 50))))
    boxes += [(10.2, 10.7, 10.9, 30.0), (0.0, 0.0, 1.9, 40.0), (-3.5, 20.0, 40.0, 50.0), (50.0, 50.0, 51.5, 51.5),
              (100.0, 80.0, 120.0, 90.0), (0.0, 0.0, 160.0, 96.0)]
    boxes = np.array(boxes, np.float32)
    disp3 = np.repeat(disp[None, None], 3, 1)
    d, s, sb = odepth.bbox_postp_depth(torch.from_numpy(boxes), torch.from_numpy(disp3))
    return dict(disp=disp, boxes=boxes, depth=np.array([float(v) for v in d], np.float32), scales=s.numpy(),
                scaled_boxes=sb.numpy())


def costvolume():
    rng = np.random.RandomState(41)
    N, Hf, Wf, Cc, D = 1, 6, 80, 24, 12
    fr = rng.normal(0, 1, (N, Hf, Wf, Cc)).astype(np.float32)
    fl = rng.normal(0, 0.2, (N, Hf, Wf, Cc)).astype(np.float32)
    fl[:, :3, 5:] += fr[:, :3, :-5]   # top half: true disparity 5
    fl[:, 3:, 9:] += fr[:, 3:, :-9]   # bottom half: true disparity 9
    cost = c_oracle.costvolume(fl, fr, Cc, D)
    lr = c_oracle.softargmin(cost, 16.0)
    up = c_oracle.disp_upsample(lr, 4, Hf * 4 - 8, Wf * 4)
    return dict(featL=fl, featR=fr, cost=cost, disp_lr=lr, disp_postp=up, temperature=16.0)


from stereotracking_amd.synthetic import synthetic_detection_stream as detection_stream  # noqa: E402,F401  (seeded INPUT generator)


SHIPPED_TRACKER = dict(obj_score_thr=0.3, init_track_thr=0.7, weight_iou_with_det_scores=False, match_iou_thr=0.1,
                       num_tentatives=3, vel_consist_weight=0.2, vel_delta_t=3, num_frames_retain=30)
"""configs/stereo_tracking/ocsort/yolox_s_mmyolo_mot_airdrone_disp.py:49-58 (the shipped values)."""


def run_oracle_tracker(det, num_frames, **cfg):
    """The ORACLE tracker (oracle/tracker.py, the restatement of the reference classes) over a detection stream.
    -> rows [t, id, x1,y1,x2,y2, score, depth, scale]."""
    from oracle import tracker as otr

    class _Model:
        motion = otr.KalmanFilter()

    trk = otr.OCSORTTracker_Disparity(**cfg)
    out = []
    for t in range(num_frames):
        d = det[det[:, 0] == t]
        inst = otr.Instances(bboxes=torch.from_numpy(d[:, 1:5].copy()), scores=torch.from_numpy(d[:, 5].copy()),
                             labels=torch.zeros(len(d), dtype=torch.long), scales=torch.from_numpy(d[:, 7].copy()),
                             depth=torch.from_numpy(d[:, 6].copy()))
        r = trk.track(_Model(), None, None, otr.Sample(t, inst))
        for i in range(len(r.instances_id)):
            out.append([t, int(r.instances_id[i]), *r.bboxes[i].tolist(), float(r.scores[i]), float(r.depth[i]),
                        float(r.scales[i])])
    return np.asarray(out, np.float64).reshape(-1, 9)


def tracker_sequence():
    """Fixture (iv): generated by the ORACLE tracker with the shipped tracker config; the product tracker must
    reproduce it frame by frame (tests/test_cpu_tracker_oracle.py)."""
    T = 64
    det = detection_stream(51, T)
    return dict(detections=det, tracks=run_oracle_tracker(det, T, **SHIPPED_TRACKER), num_frames=T)


def lapjv_ties():
    """<= 7x7 cost matrices whose optimum is NOT unique, with ALL optimal assignments enumerated by brute force and
    the one the oracle's lapjv restatement returns pinned.  Stored padded to 7x7 (NaN outside rows x cols)."""
    import itertools
    from oracle.lapjv import lapjv
    rng = np.random.RandomState(61)
    mats, shapes, limits, chosen, n_opt = [], [], [], [], []

    def all_optimal(cost, lim):
        r, c = cost.shape
        best, sols = None, []
        for k in range(min(r, c) + 1):
            for rows in itertools.combinations(range(r), k):
                for cols in itertools.permutations(range(c), k):
                    tot = sum(cost[i, j] for i, j in zip(rows, cols)) + (r - k) * lim / 2 + (c - k) * lim / 2
                    x = -np.ones(r, np.int32)
                    for i, j in zip(rows, cols):
                        x[i] = j
                    if best is None or tot < best - 1e-12:
                        best, sols = tot, [x]
                    elif abs(tot - best) <= 1e-12:
                        sols.append(x)
        return best, sols

    cases = []
    for t in range(400):
        r, c = rng.randint(2, 6), rng.randint(2, 6)
        grid = rng.choice([2, 4, 8])
        cases.append((np.round(rng.rand(r, c) * grid) / grid, float(rng.choice([0.5, 0.75, 0.9]))))
    for r, c in ((7, 7), (6, 7), (7, 5)):          # the largest sizes: duplicate rows / columns (equal-IoU boxes)
        base = np.round(rng.rand(r, c) * 8) / 8
        base[1] = base[0]
        base[:, 2] = base[:, 1]
        cases.append((base, 0.9))
    for cost, lim in cases:
        best, sols = all_optimal(cost, lim)
        if len(sols) < 2:
            continue
        opt, x, y = lapjv(cost, True, lim)
        assert any(np.array_equal(x, s) for s in sols), 'oracle lapjv returned a non-optimal assignment'
        pad = np.full((7, 7), np.nan)
        pad[:cost.shape[0], :cost.shape[1]] = cost
        xp = np.full(7, -2, np.int32)
        xp[:len(x)] = x
        mats.append(pad); shapes.append(cost.shape); limits.append(lim); chosen.append(xp); n_opt.append(len(sols))
    return dict(cost=np.asarray(mats), shape=np.asarray(shapes, np.int32), cost_limit=np.asarray(limits),
                x=np.asarray(chosen), num_optimal=np.asarray(n_opt, np.int32))


STRESS_TRACKER = dict(SHIPPED_TRACKER, obj_score_thr=0.02, init_track_thr=0.05)
"""Random-weight heads emit low scores: gates low enough that hundreds of detections per frame become tracks."""

CONFIG2 = dict(T=24, H=720, W=1280, D=192, AGG=2, objects=6, seq_seed=3, weight_seed=0, prior_prob=0.01, logit_std=0.6,
               temperature=32.0, score_thr=0.01, iou_thr=0.5, max_det=1000)


def config2_frames(cfg=CONFIG2, smooth=3):
    from stereotracking_amd.sequence import synthetic_sequence
    return list(synthetic_sequence(cfg['T'], cfg['objects'], cfg['H'], cfg['W'], cfg['D'], seed=cfg['seq_seed'],
                                   smooth=smooth))


def config2_state_dict(cfg=CONFIG2):
    """Seeded weights of the configs[2] fixture in reference state_dict naming (detector.* keys without the prefix
    + stereo.agg.*): the table comes from the ORACLE model, not from the product."""
    ora = OracleDetector(0.33, 0.5, 1).eval()
    table = [(k, tuple(v.shape)) for k, v in ora.state_dict().items() if not k.endswith('num_batches_tracked')]
    Dl = cfg['D'] // 4
    for l in range(cfg['AGG']):
        table += [(f'stereo.agg.{l}.weight', (Dl, Dl, 3, 3)), (f'stereo.agg.{l}.bias', (Dl,))]
    return synthetic_state_dict(table, seed=cfg['weight_seed'], prior_prob=cfg['prior_prob'],
                                logit_std=cfg['logit_std'])


CONFIG2_SEQUENCES = (('', 3), ('wn_', 1))
"""(key prefix, texture smoothing) of the two configs[2] input sequences of the fixture: '' = the blurred (image-like)
textures, 'wn_' = WHITE-NOISE textures - the input on which round 3's end-to-end comparison read boxes 1.5e-3 against the
fp32 oracle (the temperature-32 soft-argmin amplifies feature noise on texture-less matches).  Both stay in the suite."""
DISP_SAMPLE = 16       # the fixture keeps every 16th pixel of the full-resolution disparity (fp32 oracle and fp64)


def config2_sequence():
    """Both input sequences, each through (i) the fp32 ORACLE pipeline and (ii) the SAME arithmetic in float64
    (parity_utils.oracle_pipeline64): the float64 leg is the yardstick - the GPU-vs-fp64 distance of every float
    quantity is asserted against north_star's 1e-3 AND against the fp32 oracle's own distance from fp64."""
    import time
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    from parity_utils import make_oracle, make_oracle64, oracle_pipeline, oracle_pipeline64, rel_err
    from stereotracking_amd.synthetic import pad_to_divisor
    cfg = CONFIG2
    torch.set_num_threads(min(os.cpu_count() or 1, 8))
    sd = config2_state_dict(cfg)
    ora = make_oracle(sd)
    ora64 = make_oracle64(ora)
    H, W = cfg['H'], cfg['W']
    levels, _ = levels_for((H + 31) // 32 * 32, W, 1)
    out = {k: np.asarray(v) for k, v in cfg.items()}
    out['disp_sample'] = np.asarray(DISP_SAMPLE)
    t0 = time.time()
    for px, smooth in CONFIG2_SEQUENCES:
        dets = []
        worst = dict(box=0.0, score=0.0, disp=0.0)
        for t, f in enumerate(config2_frames(cfg, smooth)):
            img = torch.from_numpy(pad_to_divisor(f['left'].astype(np.float32), 32, 114.0))[None]
            right = torch.from_numpy(pad_to_divisor(f['right'].astype(np.float32), 32, 114.0))[None]
            r = oracle_pipeline(ora, sd, img, right, levels, (H, W), cfg['D'], cfg['temperature'], cfg['AGG'],
                                cfg['score_thr'], cfg['iou_thr'], cfg['max_det'])
            r64 = oracle_pipeline64(ora64, sd, img, right, levels, (H, W), cfg['D'], cfg['temperature'], cfg['AGG'])
            k = len(r['prior'])
            assert r['count'] == k <= cfg['max_det']
            out[f'{px}prior{t}'] = r['prior'].astype(np.int32)
            out[f'{px}boxes{t}'] = r['boxes'].astype(np.float32)
            out[f'{px}scores{t}'] = r['scores'].astype(np.float32)
            out[f'{px}depth{t}'] = r['depth'].astype(np.float32)
            out[f'{px}scales{t}'] = r['scales'].astype(np.float32)
            out[f'{px}disp_sum{t}'] = np.float64(r['disp'].double().sum().item())
            # the float64 leg: scores / boxes of the kept priors, the sampled disparity, and the fp32 oracle's distance
            out[f'{px}score64_{t}'] = r64['scores'][r['prior']]
            out[f'{px}box64_{t}'] = r64['boxes'][r['prior']]
            out[f'{px}dsamp{t}'] = r['disp'][0, 0, ::DISP_SAMPLE, ::DISP_SAMPLE].numpy().astype(np.float32)
            out[f'{px}dsamp64_{t}'] = r64['disp'][0, 0, ::DISP_SAMPLE, ::DISP_SAMPLE].numpy()
            e_disp = rel_err(r['disp'], r64['disp'])
            e_box = rel_err(r['boxes'], r64['boxes'][r['prior']])
            e_score = float(np.abs(r['scores'].astype(np.float64) - r64['scores'][r['prior']]).max())
            e_head = max(rel_err(a, b) for a, b in zip(r['rows'], r64['rows']))
            out[f'{px}err32_{t}'] = np.asarray([e_disp, e_head, e_box, e_score])   # cpu32 vs fp64, whole frame
            worst = dict(box=max(worst['box'], e_box), score=max(worst['score'], e_score), disp=max(worst['disp'], e_disp))
            sb = r['scaled_boxes'].numpy().astype(np.float32)
            for i in range(k):   # the tracker consumes the depth-SCALED boxes (ocsort_disparity.py:82-86)
                dets.append([t, *sb[i], r['scores'][i], r['depth'][i], r['scales'][i]])
            print(f'  {px or "blurred_"}frame {t}: kept {k}; cpu32-vs-fp64 disp {e_disp:.2e} head {e_head:.2e} '
                  f'box {e_box:.2e} score {e_score:.2e}  ({time.time() - t0:.0f} s)', flush=True)
        dets = np.asarray(dets, np.float32)
        out[px + 'detections'] = dets          # the oracle's detection stream (scaled boxes): input of the f-4 GPU test
        for name, tc in (('shipped', SHIPPED_TRACKER), ('stress', STRESS_TRACKER)):
            out[px + 'tracks_' + name] = run_oracle_tracker(dets, cfg['T'], **tc)
            print(f'  tracker[{px}{name}]: {len(out[px + "tracks_" + name])} track rows, '
                  f'{len(set(out[px + "tracks_" + name][:, 1].tolist()))} ids')
        print(f'  {px or "blurred"}: worst cpu32-vs-fp64 {worst}')
    return out


SHELL_TRACKER = dict(SHIPPED_TRACKER, init_track_thr=0.03, obj_score_thr=0.02)


def shell_sequence():
    """tests/test_shell_gpu.py::test_mot_shell_matches_oracle_composition's reference side, generated HERE so that
    oracle/tracker.py never runs on the GPU box: 6 frames (pairs of identical frames) of 80x160 with the disparity as
    an INPUT (the reference's own configuration), widen 0.375, confident synthetic head."""
    torch.set_num_threads(1)
    ora = OracleDetector(0.33, 0.375, 1).eval()
    table = [(k, tuple(v.shape)) for k, v in ora.state_dict().items() if not k.endswith('num_batches_tracked')]
    sd = synthetic_state_dict(table, seed=5, prior_prob=0.2, logit_std=2.5)
    ora.load_state_dict(sd, strict=False)
    ori = (80, 160)
    levels, _ = levels_for(96, 160, 1)
    out, dets = dict(num_frames=6), []
    for t in range(6):
        fr = synthetic_batch([40 + (t // 2)], ori[0], ori[1], 32)
        # what the model sees: the preprocessor pads the un-padded uint8 frame with 0 (the dataset pipeline would
        # have padded with 114 before; the test feeds un-padded frames)
        img = torch.nn.functional.pad(fr['img'][0:1, :, :ori[0]].to(torch.uint8).float(), [0, 0, 0, 16])
        disp = fr['disp_postp'][0:1]
        with torch.no_grad():
            rows = head_to_rows(*ora(dict(img=img, disp_postp=disp)))
        flat = []
        for r in rows:
            buf = torch.zeros(1, r.shape[1], 8)
            buf[..., :6] = r
            flat.append(buf.reshape(-1))
        b, sc, _, p, c = c_oracle.decode_nms(torch.cat(flat).numpy(), 1, levels, 0.01, 0.5, 1000, ori)
        k = int(c[0])
        d, scl, sb = odepth.bbox_postp_depth(torch.from_numpy(b[0, :k]), disp)
        out[f'prior{t}'], out[f'boxes{t}'], out[f'scores{t}'] = p[0, :k], b[0, :k], sc[0, :k]
        for i in range(k):
            dets.append([t, *sb[i].tolist(), sc[0, i], float(d[i]), float(scl[i])])
    tr = run_oracle_tracker(np.asarray(dets, np.float32), 6, **SHELL_TRACKER)
    out['tracks'] = tr
    assert len(tr) > 0
    return out


if __name__ == '__main__':
    c_oracle.build()
    for name, fn in (('shell_sequence', shell_sequence), ('config2_sequence', config2_sequence), ('detector_tiny', detector_tiny), ('decode_nms', decode_nms), ('box_depth', box_depth),
                     ('costvolume', costvolume), ('tracker_sequence', tracker_sequence), ('lapjv_ties', lapjv_ties)):
        if len(sys.argv) > 1 and name not in sys.argv[1:]:
            continue
        np.savez_compressed(os.path.join(HERE, name + '.npz'), **fn())
        print('wrote', name)
