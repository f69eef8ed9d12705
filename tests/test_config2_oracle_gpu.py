"""GPU: BASELINE.json configs[2] END TO END against the oracle at its stated size — a synthetic 1280x720 sequence, D=192,
full two-branch YOLOX-s + 2 aggregation convs, detector + disparity feeding the CPU OC-SORT association.

Reference side (tests/golden/config2_sequence.npz, generated in the build container by tests/golden/make_golden.py;
nothing under oracle/tracker.py runs on the GPU box): oracle stereo module -> oracle detector on the oracle's OWN
disparity -> C decode + NMS -> numpy extract_depth -> ORACLE tracker (restatement of reference
mmtrack/models/mot/ocsort_disparity.py:50-111 + trackers/ocsort_tracker_disparity.py:345-618), with the SHIPPED tracker
thresholds and with stress thresholds (~400 tracks per frame).

Product side: Config.fromfile(stereo config) -> MODELS.build -> model.test_step over the same frames, uploaded as the
dataset pipeline yields them (uint8, padded with 114 by Pad_Disparity), 8 frames per launch plan on 3 in-flight
contexts, native CPU tracker.

What is asserted, per frame:
  * kept prior SETS equal (differences must be explained by marginal decisions AND stay below 1 %), floats of the
    common detections within 1e-3 * max(1, |ref|);
  * the GPU's detection ORDER is carried through the association: `instances_id` of model.test_step against the
    oracle tracker's ids.  New ids are handed out in DETECTION order (base_tracker.py:54-91), and the detection order
    is the score order, so two correct fp32 evaluations whose scores differ by the float noise of the path hand the
    ids of two near-tied new detections out the other way round.  The criterion is therefore: ONE bijection
    gpu id <-> oracle id holds over ALL frames on identical boxes, and every id it relabels is exchanged with an id
    BORN IN THE SAME FRAME (a permutation inside one frame's batch of new ids).  SHIPPED thresholds: no row outside
    the bijection.  Stress thresholds (~250 tracks per frame): rows outside it (an association decision that hangs
    on a margin inside the float noise) are bounded to 1 %.

Round 4: TWO input sequences (blurred textures and the WHITE-NOISE textures on which round 3 read boxes 1.5e-3 against
the fp32 oracle), and a FLOAT64 leg in the fixture (parity_utils.oracle_pipeline64: the same arithmetic in double).
The float criterion is measured against float64, per quantity (boxes, scores, disparity):
    gpu-vs-fp64 <= max(1e-3, 1.25 x cpu32-vs-fp64)      and, on the blurred sequence, gpu-vs-cpu32 <= 1e-3 as before.
(The fp32 ORACLE itself is 1.03e-3 away from float64 in one white-noise frame: at temperature 32 the soft-argmin
amplifies fp32 feature noise on texture-less matches, for every fp32 evaluation.)  The score noise that legitimises an
order swap is no longer hand-set: two detections may swap only if their FLOAT64 scores are closer than the sum of the
two measured score errors (gpu-vs-fp64 + cpu32-vs-fp64) of that sequence.
Round 6: the GPU's tracks are also SCORED against the oracle's tracks as ground truth (CLEAR MOTA / Identity IDF1,
stereotracking_amd/metrics.py): 1.0 / 1.0 with the shipped thresholds on both sequences, >= 0.99 under the stress thresholds.
The record goes to gpurun_out/r06_config2_oracle_<sequence>_<thresholds>.json (copied to profiles/)."""
import os

import numpy as np
import pytest
import torch

from parity_utils import compare_kept, match_track_rows, rel_err, unscale_boxes_np, write_record

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, 'tests', 'golden', 'config2_sequence.npz')
CFG = os.path.join(ROOT, 'configs', 'stereo_tracking', 'ocsort', 'stereo_yolox_s_mot_airdrone_costvolume.py')
SEQUENCES = {'blurred': ('', 3), 'white_noise': ('wn_', 1)}     # name -> (fixture key prefix, texture smoothing)


def build(tracker_overrides, g):
    from stereotracking_amd import mot  # noqa: F401
    from stereotracking_amd.config import Config
    from stereotracking_amd.registry import MODELS
    from stereotracking_amd.synthetic import synthetic_state_dict
    cfg = Config.fromfile(CFG)
    cfg.model.stereo['max_disp'] = int(g['D'])
    cfg.model.stereo['agg_layers'] = int(g['AGG'])
    cfg.model.tracker.update(tracker_overrides)
    # tuning_cache=None: the committed plan of pipeline.default_tuning_cache() - the one bench.py runs
    model = MODELS.build(dict(cfg.model, dense_batch=8, inflight=3, max_det=int(g['max_det'])))
    table = list(model.detector._table) + [('stereo.' + n, shp) for n, shp in model.stereo.param_table()]
    # name-keyed RNG streams: the same values make_golden.config2_state_dict drew from the ORACLE's table
    sd = synthetic_state_dict(table, seed=int(g['weight_seed']), prior_prob=float(g['prior_prob']),
                              logit_std=float(g['logit_std']))
    model.detector.load_state_dict(sd, strict=False)
    model.stereo.load_state_dict({k[len('stereo.'):]: v for k, v in sd.items() if k.startswith('stereo.')})
    return model


def frames_u8(g, dev, smooth):
    from stereotracking_amd.sequence import synthetic_sequence
    from stereotracking_amd.synthetic import pad_to_divisor
    T, H, W = int(g['T']), int(g['H']), int(g['W'])
    left, right = [], []
    for f in synthetic_sequence(T, int(g['objects']), H, W, int(g['D']), seed=int(g['seq_seed']), smooth=smooth):
        left.append(torch.from_numpy(pad_to_divisor(f['left'], 32, 114))[None].to(dev))     # (1,3,736,1280) uint8
        right.append(torch.from_numpy(pad_to_divisor(f['right'], 32, 114))[None].to(dev))
    return left, right


def order_by_score(scores, priors):
    """Positions sorted by (score desc, prior index asc) - the tie rule of SURVEY.md 7 that oracle and kernel share."""
    return np.lexsort((np.asarray(priors), -np.asarray(scores, np.float64)))


@pytest.mark.parametrize('name', ['shipped', 'stress'])
@pytest.mark.parametrize('seq', ['blurred', 'white_noise'])
def test_config2_sequence_against_oracle_pipeline_and_oracle_tracker(seq, name, cuda):
    from stereotracking_amd.structures import TrackDataSample
    g0 = np.load(GOLD)
    px, smooth = SEQUENCES[seq]

    class _G:       # the fixture restricted to one sequence: per-sequence keys carry the prefix, the config does not
        def __getitem__(self, k):
            return g0[px + k] if (px + k) in g0.files else g0[k]
    g = _G()
    T, H, W = int(g['T']), int(g['H']), int(g['W'])
    over = {} if name == 'shipped' else dict(obj_score_thr=0.02, init_track_thr=0.05)
    model = build(over, g)
    left, right = frames_u8(g, cuda, smooth)
    samples = [TrackDataSample(dict(frame_id=t, ori_shape=(H, W), img_shape=(H, W), scale_factor=(1.0, 1.0)))
               for t in range(T)]
    outs = model.test_step(dict(inputs=dict(img=left, right=right), data_samples=samples))
    torch.cuda.synchronize()
    assert len(outs) == T
    # the disparity the dense path produced for the first chunk, through the same context set (the shell does not
    # return it): sampled like the fixture, against the fp32 oracle and against float64
    runner = model.dense_runner((H, W), True, 8)
    ds = int(g['disp_sample'])
    o8 = runner.pipes[0].run(torch.cat(left[:8]).float(), torch.cat(right[:8]).float())
    dsamp_gpu = o8['disp_postp'][:, 0, ::ds, ::ds].cpu().double().numpy()
    torch.cuda.synchronize()
    e64 = dict(gpu_box=0.0, cpu_box=0.0, gpu_score=0.0, cpu_score=0.0, gpu_disp=0.0, cpu_disp=0.0)
    box_err_g, box_err_c = [], []
    for t in range(8):
        d64, d32 = g[f'dsamp64_{t}'], g[f'dsamp{t}'].astype(np.float64)
        den = np.maximum(1.0, np.abs(d64))
        e64['gpu_disp'] = max(e64['gpu_disp'], float((np.abs(dsamp_gpu[t] - d64) / den).max()))
        e64['cpu_disp'] = max(e64['cpu_disp'], float((np.abs(d32 - d64) / den).max()))
    ref_tracks = g['tracks_' + name]      # rows [t, id, scaled box (4), score, depth, scale]

    rec = dict(config=f'configs[2]: {T}-frame synthetic {W}x{H} sequence, D={int(g["D"])}, full YOLOX-s two-branch, '
                      f'{int(g["AGG"])} aggregation convs, model.test_step (8 frames per plan, 3 contexts), '
                      f'{name} tracker thresholds', frames=[])
    phi, inv = {}, {}                     # gpu id -> oracle id and back: ONE bijection over the whole sequence
    tot = dict(track_rows=0, matched=0, inconsistent=0, only_gpu=0, only_oracle=0, det_sym_diff=0, det_swaps=0,
               frames_with_equal_det_order=0, frames_with_equal_ids_in_order=0)
    worst = dict(box=0.0, score=0.0, depth=0.0, track_box=0.0, gap_at_swaps=0.0, gap64_at_swaps=0.0)
    tot.update(frames_gpu_order_eq_fp64=0, frames_cpu32_order_eq_fp64=0, frames_box_over_1e3_gpu=0,
               frames_box_over_1e3_cpu32=0)
    for t in range(T):
        det, trk = outs[t].pred_det_instances, outs[t].pred_track_instances
        gp, rp = det.prior_idx.cpu().numpy(), g[f'prior{t}']
        score_of = np.zeros(int(max(gp.max(), rp.max())) + 1, np.float32)
        score_of[rp] = g[f'scores{t}']
        score_of[gp] = np.where(score_of[gp] > 0, score_of[gp], det.scores.cpu().numpy())
        ck = compare_kept(gp, rp, score_of)
        tot['det_sym_diff'] += ck['kept_set_sym_diff']
        tot['det_swaps'] += ck['positions_swapped']
        tot['frames_with_equal_det_order'] += ck['kept_equal_in_order']
        worst['gap_at_swaps'] = max(worst['gap_at_swaps'], ck['max_score_gap_at_swaps'])
        pos_r = {int(p): k for k, p in enumerate(rp)}
        common = [k for k, p in enumerate(gp) if int(p) in pos_r]
        ir = [pos_r[int(gp[k])] for k in common]
        worst['box'] = max(worst['box'], rel_err(det.bboxes[common].cpu(), g[f'boxes{t}'][ir]))
        worst['score'] = max(worst['score'], float(np.abs(det.scores[common].cpu().numpy() - g[f'scores{t}'][ir]).max()))
        # --- against float64, on the common detections --------------------------------------------------------------
        b64, s64 = g[f'box64_{t}'][ir], g[f'score64_{t}'][ir]
        eb_g, eb_c = rel_err(det.bboxes[common].cpu(), b64), rel_err(g[f'boxes{t}'][ir], b64)
        es_g = float(np.abs(det.scores[common].cpu().double().numpy() - s64).max())
        es_c = float(np.abs(g[f'scores{t}'][ir].astype(np.float64) - s64).max())
        e64.update(gpu_box=max(e64['gpu_box'], eb_g), cpu_box=max(e64['cpu_box'], eb_c),
                   gpu_score=max(e64['gpu_score'], es_g), cpu_score=max(e64['cpu_score'], es_c))
        # the whole DISTRIBUTION of the per-box distances (the maxima above are set by a handful of boxes on ambiguous
        # matches: alternative kernel plans move the white-noise maximum between 1.05e-3 and 1.9e-3)
        b64t = torch.as_tensor(b64, dtype=torch.float64)
        den = b64t.abs().clamp(min=1.0)
        box_err_g.append(((det.bboxes[common].cpu().double() - b64t).abs() / den).max(dim=1).values.numpy())
        box_err_c.append(((torch.as_tensor(g[f'boxes{t}'][ir], dtype=torch.float64) - b64t).abs() / den).max(dim=1).values.numpy())
        tot['frames_box_over_1e3_gpu'] += eb_g > 1e-3
        tot['frames_box_over_1e3_cpu32'] += eb_c > 1e-3
        pc = gp[common]
        o64 = order_by_score(s64, pc)
        og = order_by_score(det.scores[common].cpu().numpy(), pc)
        oc = order_by_score(g[f'scores{t}'][ir], pc)
        tot['frames_gpu_order_eq_fp64'] += bool(np.array_equal(og, o64))
        tot['frames_cpu32_order_eq_fp64'] += bool(np.array_equal(oc, o64))
        sw = og != oc                      # positions where the GPU's and the fp32 oracle's orders differ
        if sw.any():
            worst['gap64_at_swaps'] = max(worst['gap64_at_swaps'], float(np.abs(s64[og[sw]] - s64[oc[sw]]).max()))
        # --- tracks of this frame -------------------------------------------------------------------------------
        rt = ref_tracks[ref_tracks[:, 0] == t]
        r_ids = rt[:, 1].astype(np.int64)
        r_boxes = unscale_boxes_np(rt[:, 2:6], rt[:, 8])
        g_ids = trk.instances_id.cpu().numpy().astype(np.int64)
        g_boxes = trk.bboxes.cpu().double().numpy()
        pairs, only_g, only_r, err = match_track_rows(g_boxes, r_boxes)
        worst['track_box'] = max(worst['track_box'], err)
        tot['track_rows'] += len(r_ids)
        tot['only_gpu'] += len(only_g)
        tot['only_oracle'] += len(only_r)
        # depth of the track boxes (the second extract_depth pass, ocsort_disparity.py:99-104) against the depth the
        # oracle attached to the same box; a box edge within float noise of an integer selects another pixel window
        # (a discrete change): such rows are counted, not compared
        g_depth = trk.depth.cpu().double().numpy()
        for i, j in pairs:
            dr, dg = float(rt[j, 7]), float(g_depth[i])
            tot['depth_rows'] = tot.get('depth_rows', 0) + 1
            if np.isnan(dr) or np.isnan(dg) or dr == -1 or dg == -1:
                tot['depth_class_mismatch'] = tot.get('depth_class_mismatch', 0) + int(
                    (np.isnan(dr) != np.isnan(dg)) or ((dr == -1) != (dg == -1)))
                continue
            e = abs(dg - dr) / max(1.0, abs(dr))
            if e > 1e-3:
                tot['depth_over_tol'] = tot.get('depth_over_tol', 0) + 1
            else:
                worst['depth'] = max(worst['depth'], e)
        for i, j in pairs:
            a, b = int(g_ids[i]), int(r_ids[j])
            if phi.get(a, b) != b or inv.get(b, a) != a:
                tot['inconsistent'] += 1
                continue
            phi[a], inv[b] = b, a
            tot['matched'] += 1
        tot['frames_with_equal_ids_in_order'] += bool(len(g_ids) == len(r_ids) and np.array_equal(g_ids, r_ids))
        rec['frames'].append(dict(t=t, det_gpu=len(gp), det_oracle=len(rp), det_sym_diff=ck['kept_set_sym_diff'],
                                  det_positions_swapped=ck['positions_swapped'], tracks_gpu=len(g_ids),
                                  tracks_oracle=len(r_ids), ids_equal_in_order=bool(
                                      len(g_ids) == len(r_ids) and np.array_equal(g_ids, r_ids))))
    relabeled = {a: b for a, b in phi.items() if a != b}
    # birth frame of every id on each side; a relabeled id must be exchanged with an id born in the same frame
    birth_g, birth_r = {}, {}
    for t in range(T):
        for a in outs[t].pred_track_instances.instances_id.cpu().tolist():
            birth_g.setdefault(int(a), t)
    for row in ref_tracks:
        birth_r.setdefault(int(row[1]), int(row[0]))
    cross_frame = {a: b for a, b in relabeled.items() if birth_g.get(a) != birth_r.get(b)}
    rec['totals'] = {k: int(v) for k, v in tot.items()}
    rec['worst'] = worst
    rec['vs_fp64'] = e64
    eg, ec = np.concatenate(box_err_g), np.concatenate(box_err_c)
    rec['box_vs_fp64_distribution'] = {
        who: dict(boxes=int(len(e)), mean=float(e.mean()), p50=float(np.percentile(e, 50)), p99=float(np.percentile(e, 99)),
                  p999=float(np.percentile(e, 99.9)), over_1e3=int((e > 1e-3).sum()), max=float(e.max()))
        for who, e in (('gpu', eg), ('cpu32', ec))}
    noise = e64['gpu_score'] + e64['cpu_score']     # MEASURED: two evaluations this far from fp64 may order a pair either way
    rec['score_noise_measured'] = noise
    rec['ids_seen'] = len(phi)
    rec['ids_relabeled'] = len(relabeled)
    rec['ids_relabeled_across_birth_frames'] = len(cross_frame)
    rec['relabeled_examples'] = {str(a): b for a, b in list(relabeled.items())[:16]}
    # --- the consequence in the metric the reference REPORTS (mot_drone_metrics.py:83-88: CLEAR + Identity): the GPU's tracks
    # scored against the ORACLE's tracks as ground truth (both unscaled, x y w h).  MOTA / IDF1 are invariant under a
    # relabeling of ids, so "one id bijection" reads as 1.0 / 1.0 here; every row outside the bijection costs an FP / FN /
    # IDSW.  (the white-noise miss of the 1e-3 float bar, bounded where it matters - the tracks.)
    from stereotracking_amd.metrics import clear_identity

    def mot_rows(t, ids, boxes):
        b = np.asarray(boxes, np.float64).reshape(-1, 4)
        return np.concatenate([np.full((len(b), 1), float(t)), np.asarray(ids, np.float64).reshape(-1, 1), b[:, 0:2],
                               b[:, 2:4] - b[:, 0:2]], axis=1)
    gt_rows = np.concatenate([mot_rows(t, ref_tracks[ref_tracks[:, 0] == t][:, 1],
                                       unscale_boxes_np(ref_tracks[ref_tracks[:, 0] == t][:, 2:6],
                                                        ref_tracks[ref_tracks[:, 0] == t][:, 8])) for t in range(T)])
    pr_rows = np.concatenate([mot_rows(t, outs[t].pred_track_instances.instances_id.cpu().numpy(),
                                       outs[t].pred_track_instances.bboxes.cpu().double().numpy()) for t in range(T)])
    mot = clear_identity(gt_rows, pr_rows, iou_thr=0.5)
    rec['mot_vs_oracle_tracks'] = {k: float(mot[k]) for k in ('MOTA', 'IDF1', 'MOTP', 'IDSW', 'FP', 'FN', 'TP', 'IDP', 'IDR')}
    rec['within_1e3_of_cpu_path'] = bool(worst['box'] <= 1e-3 and worst['score'] <= 1e-3)
    write_record(f'r06_config2_oracle_{seq}_{name}.json', rec)
    print({k: v for k, v in rec.items() if k != 'frames'})

    rows = max(tot['track_rows'], 1)
    assert tot['track_rows'] > T, 'the scenario must exercise the association step'
    # floats, against float64: north_star's 1e-3, or - where the fp32 ORACLE itself is further than that from float64
    # (white noise) - no worse than 1.25 x the oracle's own distance
    for q in ('box', 'score', 'disp'):
        assert e64['gpu_' + q] <= max(1e-3, 1.25 * e64['cpu_' + q]), (q, e64)
    # ... and not only at the maximum: the whole distribution of the per-box distance to float64 is no wider than the
    # fp32 oracle's (mean and 99.9 % quantile within 1.25 x, both far inside 1e-3)
    dist = rec['box_vs_fp64_distribution']
    assert dist['gpu']['mean'] <= 1.25 * dist['cpu32']['mean'] and dist['gpu']['p999'] <= 1e-3, dist
    assert dist['gpu']['p999'] <= max(2e-4, 1.5 * dist['cpu32']['p999']), dist
    # ... and the TAIL is counted: boxes further than 1e-3 from float64.  Round 4 measured 0 (blurred) and 3 of 15 550
    # (white noise; the fp32 oracle: 0 and 1).  A change that widens the tail but keeps the maximum inside 1.25 x the
    # oracle's would pass every bound above; it does not pass this one.
    assert dist['gpu']['over_1e3'] <= (0 if seq == 'blurred' else 3), dist
    assert worst['score'] <= 1e-3 and worst['track_box'] <= 1e-3, worst
    if seq == 'blurred':
        assert worst['box'] <= 1e-3, worst                    # gpu vs the fp32 oracle directly, as in round 3
    else:
        assert worst['box'] <= 1.01 * (e64['gpu_box'] + e64['cpu_box']) + 1e-6, (worst, e64)     # triangle inequality holds
    # order swaps only between detections whose FLOAT64 scores are inside the measured noise of the two evaluations
    assert worst['gap64_at_swaps'] <= noise, (worst, noise)
    assert tot['det_sym_diff'] <= max(2, sum(f['det_oracle'] for f in rec['frames']) // 100), tot
    assert tot.get('depth_over_tol', 0) + tot.get('depth_class_mismatch', 0) <= max(2, tot.get('depth_rows', 0) // 100), tot
    if name == 'shipped':
        # the SHIPPED configuration, in the reference's own metric: GPU tracks against the oracle's tracks are a perfect
        # score on BOTH sequences - the white-noise one included, where the float bar of 1e-3 is missed (1.5e-3)
        assert mot['MOTA'] == 1.0 and mot['IDF1'] == 1.0 and mot['IDSW'] == 0 and mot['FP'] == 0 and mot['FN'] == 0, mot
    else:
        # stress thresholds (~230-440 tracks per frame): the rows outside the bijection cost at most 1 % of either score
        assert mot['MOTA'] >= 0.99 and mot['IDF1'] >= 0.99, mot
    if name == 'shipped':
        # the SHIPPED configuration: every track row inside the bijection, relabels only inside one frame's new ids
        assert tot['inconsistent'] == 0 and tot['only_gpu'] == 0 and tot['only_oracle'] == 0, tot
        assert not cross_frame, cross_frame
    else:
        assert tot['inconsistent'] + tot['only_gpu'] + tot['only_oracle'] <= rows // 100, tot
        assert len(cross_frame) <= max(2, len(phi) // 100), (len(cross_frame), len(phi))
