"""Shared helpers of the parity tests (test infrastructure; may import oracle/).

`oracle_pipeline` is the END-TO-END oracle composition of the dense path for one stereo pair: every stage
consumes the ORACLE's own upstream result (oracle features -> oracle cost volume / aggregation /
soft-argmin / upsample -> oracle detector on the oracle's disparity -> C decode + NMS -> numpy
extract_depth), i.e. what the reference's CPU path would compute, never a GPU
intermediate.  `stagewise_*` variants feed a stage the GPU's own upstream tensor instead.
"""
import json
import os

import numpy as np
import torch

from oracle import c_oracle, depth as odepth, stereo as ostereo
from oracle.torch_model import OracleDetector, head_to_rows

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def rel_err(got, ref):
    """max |got - ref| / max(1, |ref|): north_star's float tolerance, elementwise."""
    got, ref = torch.as_tensor(got).double(), torch.as_tensor(ref).double()
    return ((got - ref).abs() / ref.abs().clamp(min=1.0)).max().item()


def make_oracle(sd, widen=0.5, deepen=0.33):
    ora = OracleDetector(deepen, widen, 1).eval()
    missing, unexpected = ora.load_state_dict({k: v for k, v in sd.items() if not k.startswith('stereo.')},
                                              strict=False)
    assert not unexpected and all(k.endswith('num_batches_tracked') for k in missing)
    return ora


def rows_to_flat_head(rows, levels):
    """Oracle head rows [(1, h*w, 6)] -> the product's flat head layout for ONE image + its level table."""
    flat, lv, off = [], [], 0
    for r, (h, w, s, _) in zip(rows, levels):
        assert r.shape[0] == 1 and r.shape[1] == h * w
        pad = torch.zeros(1, h * w, 8, dtype=torch.float32)
        pad[..., :6] = r.float()
        lv.append((h, w, s, off))
        off += h * w * 8
        flat.append(pad.reshape(-1))
    return torch.cat(flat).numpy(), lv


def oracle_pipeline(ora, sd, img, right, levels, ori_hw, max_disp, temperature, agg_layers, score_thr, iou_thr,
                    max_det, disp_override=None):
    """One pair (1,3,H,W) through the whole oracle path.  `disp_override` (1,3,H,W) replaces the oracle's own
    disparity (stage-wise comparison).  Returns a dict of numpy / torch results."""
    H, W = ori_hw
    with torch.no_grad():
        fl = ora.backbone.stage1_features(img).permute(0, 2, 3, 1).contiguous().numpy()
        fr = ora.backbone.stage1_features(right).permute(0, 2, 3, 1).contiguous().numpy()
        cost, lr, disp = ostereo.disparity(fl, fr, fl.shape[-1], max_disp // 4, temperature, sd, agg_layers,
                                           valid_hw=(H, W))
        disp = torch.from_numpy(disp)
        used = disp if disp_override is None else disp_override
        rows = head_to_rows(*ora(dict(img=img, disp_postp=used)))
    head, lv = rows_to_flat_head(rows, levels)
    boxes, scores, labels, prior, counts = c_oracle.decode_nms(head, 1, lv, score_thr, iou_thr, max_det, (H, W))
    k = min(int(counts[0]), max_det)
    d_ref, s_ref, sb_ref = odepth.bbox_postp_depth(torch.from_numpy(boxes[0, :k]), used)
    return dict(disp=disp, disp_lr=lr, rows=rows, boxes=boxes[0, :k], scores=scores[0, :k], prior=prior[0, :k],
                count=int(counts[0]), depth=np.array([float(v) for v in d_ref], np.float64),
                scales=np.array([float(v) for v in s_ref], np.float64), scaled_boxes=sb_ref)


def make_oracle64(ora):
    """A float64 copy of the oracle detector (the 'exact' evaluation both fp32 evaluations are measured against)."""
    import copy
    return copy.deepcopy(ora).double()


def stereo_fp64(fl, fr, max_disp_lr, temperature, sd, agg_layers, scale, pad_hw, valid_hw, prefix='stereo.'):
    """The stereo module's arithmetic (oracle/st_oracle.c: cost volume, aggregation convs, soft-argmin, bilinear
    upsample; same formulas) evaluated in float64 torch.  fl / fr: (1,C,Hf,Wf) double -> disp (1,3,H,W) double."""
    import torch.nn.functional as F
    _, Cc, Hf, Wf = fl.shape
    cost = torch.zeros(1, max_disp_lr, Hf, Wf, dtype=torch.float64)
    for d in range(max_disp_lr):
        cost[0, d, :, d:] = (fl[0, :, :, d:] * fr[0, :, :, :Wf - d]).sum(0) / Cc
    x = cost
    for l in range(agg_layers):
        x = F.conv2d(x, sd[f'{prefix}agg.{l}.weight'].double(), sd[f'{prefix}agg.{l}.bias'].double(), padding=1)
        if l < agg_layers - 1:
            x = F.silu(x)
    p = torch.softmax(temperature * x, dim=1)
    lr = (p * torch.arange(max_disp_lr, dtype=torch.float64).view(1, -1, 1, 1)).sum(1, keepdim=True)
    up = F.interpolate(lr, scale_factor=scale, mode='bilinear', align_corners=False) * scale
    H, W = pad_hw
    out = torch.zeros(1, 1, H, W, dtype=torch.float64)
    vh, vw = valid_hw
    out[..., :vh, :vw] = up[..., :vh, :vw]
    return out.expand(1, 3, H, W).contiguous()


def oracle_pipeline64(ora64, sd, img, right, levels, ori_hw, max_disp, temperature, agg_layers):
    """One pair through the path's arithmetic in float64: -> dict(disp (1,3,H,W) f64, scores (P,) f64, boxes (P,4) f64
    for EVERY prior, decoded / clamped as oracle/st_oracle.c does)."""
    H, W = ori_hw
    with torch.no_grad():
        fl = ora64.backbone.stage1_features(img.double())
        fr = ora64.backbone.stage1_features(right.double())
        disp = stereo_fp64(fl, fr, max_disp // 4, float(temperature), sd, agg_layers, 4, tuple(img.shape[-2:]), (H, W))
        rows = head_to_rows(*ora64(dict(img=img.double(), disp_postp=disp)))
    sc, bx = [], []
    for r, (h, w, s, _) in zip(rows, levels):
        r = r[0].numpy()
        ys, xs = np.divmod(np.arange(h * w), w)
        score = (1 / (1 + np.exp(-r[:, 0]))) * (1 / (1 + np.exp(-r[:, 5])))
        cx, cy = r[:, 1] * s + xs * s, r[:, 2] * s + ys * s
        bw, bh = np.exp(r[:, 3]) * s, np.exp(r[:, 4]) * s
        b = np.stack([cx - bw / 2, cy - bh / 2, cx + bw / 2, cy + bh / 2], 1)
        b[:, 0::2] = np.clip(b[:, 0::2], 0, W)
        b[:, 1::2] = np.clip(b[:, 1::2], 0, H)
        sc.append(score)
        bx.append(b)
    return dict(disp=disp, scores=np.concatenate(sc), boxes=np.concatenate(bx), rows=rows)


def align_kept(pa, pb):
    """Two kept-prior lists (score order) -> (ia, ib): positions of the COMMON priors in each, in a's order."""
    posb = {int(p): k for k, p in enumerate(pb)}
    ia = np.array([k for k, p in enumerate(pa) if int(p) in posb], dtype=np.int64)
    ib = np.array([posb[int(pa[k])] for k in ia], dtype=np.int64)
    return ia, ib


def compare_kept(pa, pb, score_of):
    """Kept-prior lists of two runs -> record: set equality, and how far the ORDER of the common priors differs.
    `score_of[prior]` = reference score; an order swap is only legitimate between (near-)equal scores, so the
    largest reference-score gap across swapped positions is reported (`max_score_gap_at_swaps`)."""
    sa, sb = set(int(v) for v in pa), set(int(v) for v in pb)
    ia, ib = align_kept(pa, pb)
    ca = np.asarray(pa)[ia]                       # common priors in a's order
    cb = np.asarray(pb)[np.sort(ib)]              # common priors in b's order
    swapped = ca != cb
    gap = float(np.abs(score_of[ca[swapped]] - score_of[cb[swapped]]).max()) if swapped.any() else 0.0
    return dict(count_a=len(pa), count_b=len(pb), kept_sets_equal=sa == sb, kept_set_sym_diff=len(sa ^ sb),
                kept_equal_in_order=bool(len(pa) == len(pb) and np.array_equal(pa, pb)),
                positions_swapped=int(swapped.sum()), max_score_gap_at_swaps=gap)


def compare_to_oracle(out, n, ref, levels):
    """GPU pipeline result `out` (dict of device tensors), image n, against an oracle_pipeline() dict.
    Floats are compared on the common kept priors, aligned by prior index.  Returns a record (no assertions)."""
    k_gpu = int(out['counts'][n])
    gp = out['prior_idx'][n, :min(k_gpu, out['prior_idx'].shape[1])].cpu().numpy()
    rp = ref['prior']
    score_of, _ = decode_all_priors(ref['rows'], levels)
    rec = compare_kept(gp, rp, score_of)
    rec['count_gpu'], rec['count_oracle'] = rec.pop('count_a'), ref['count']
    rec.pop('count_b')
    d_gpu = out['disp_postp'][n, 0].cpu()
    d_ref = ref['disp'][0, 0]
    ad = (d_gpu - d_ref).abs()
    rec['disp_max_abs_err_px'] = ad.max().item()
    rec['disp_max_rel_err'] = (ad / d_ref.abs().clamp(min=1.0)).max().item()
    rec['disp_l1_px'] = ad.mean().item()
    rec['disp_frac_px_over_1e-3rel'] = ((ad / d_ref.abs().clamp(min=1.0)) > 1e-3).float().mean().item()
    ia, ib = align_kept(gp, rp)
    if len(ia):
        ig = torch.from_numpy(ia)
        rec['box_max_rel_err'] = rel_err(out['boxes'][n, ig].cpu(), ref['boxes'][ib])
        rec['score_max_abs_err'] = float(np.abs(out['scores'][n, ig].cpu().numpy() - ref['scores'][ib]).max())
        same_win = (out['boxes'][n, ig].cpu().int().numpy() == ref['boxes'][ib].astype(np.int32)).all(-1)
        rec['boxes_with_different_pixel_window'] = int((~same_win).sum())   # extract_depth truncates to int pixels
        dg = out['depth'][n, ig].cpu().double().numpy()[same_win]
        dr = ref['depth'][ib][same_win]
        rec['depth_class_equal'] = bool(np.array_equal(np.isnan(dg), np.isnan(dr)) and
                                        np.array_equal(dg == -1, dr == -1))
        ok = ~np.isnan(dr) & (dr != -1) & ~np.isnan(dg) & (dg != -1)
        rec['depth_max_rel_err'] = float((np.abs(dg[ok] - dr[ok]) / np.maximum(1.0, np.abs(dr[ok]))).max()) if ok.any() else 0.0
    return rec


def decode_all_priors(rows, levels):
    """Oracle head rows [(1, h*w, 6)] of ONE image -> (scores (P,), boxes (P,4)) for every prior, fp32 numpy
    (same formulas as oracle/st_oracle.c; libm exp instead of its polynomial: used for MARGIN analysis only)."""
    sc, bx = [], []
    for r, (h, w, s, _) in zip(rows, levels):
        r = r[0].float().numpy()
        ys, xs = np.divmod(np.arange(h * w), w)
        score = (1 / (1 + np.exp(-r[:, 0]))) * (1 / (1 + np.exp(-r[:, 5])))
        cx, cy = r[:, 1] * s + xs * s, r[:, 2] * s + ys * s
        bw, bh = np.exp(r[:, 3]) * s, np.exp(r[:, 4]) * s
        sc.append(score.astype(np.float32))
        bx.append(np.stack([cx - bw / 2, cy - bh / 2, cx + bw / 2, cy + bh / 2], 1).astype(np.float32))
    return np.concatenate(sc), np.concatenate(bx)


def pairwise_iou(b):
    a = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    w = np.clip(np.minimum(b[:, None, 2], b[None, :, 2]) - np.maximum(b[:, None, 0], b[None, :, 0]), 0, None)
    h = np.clip(np.minimum(b[:, None, 3], b[None, :, 3]) - np.maximum(b[:, None, 1], b[None, :, 1]), 0, None)
    inter = w * h
    return inter / (a[:, None] + a[None, :] - inter)


def explain_kept_difference(rows, levels, kept_a, kept_b, score_thr, iou_thr, eps_score=5e-5, eps_iou=5e-3):
    """Are the priors kept by only one of two runs (e.g. GPU pipeline vs end-to-end oracle) explained by decisions
    whose MARGIN is below the float tolerance of the path?

    Two correct fp32 evaluations of the same 60-layer network differ (measured on configs[1], profiles/
    r02_e2e_parity.json: head logits up to 4e-4 relative, scores 2e-5, box coordinates 0.06 px - all inside
    north_star's 1e-3 float tolerance); a score within that of score_thr or an IoU within that of iou_thr is decided
    either way, and greedy NMS then propagates the flip to every box the flipped one overlaps.  The default margins
    are ~2x the measured score error and ~2x the IoU change a 0.06 px shift causes on a 20 px box.  Seeds = priors with |score - thr| < eps_score and pairs of candidates with
    |IoU - thr| < eps_iou; the AFFECTED set is their closure under 'overlaps (IoU > thr - eps) an affected box'.
    A difference outside that closure is a genuine bug.  Returns a record; `unexplained` must be empty."""
    scores, boxes = decode_all_priors(rows, levels)
    cand = np.nonzero(scores > score_thr - eps_score)[0]
    iou = pairwise_iou(boxes[cand])
    np.fill_diagonal(iou, 0.0)
    seed = np.abs(scores[cand] - score_thr) < eps_score
    near = np.abs(iou - iou_thr) < eps_iou
    seed |= near.any(1)
    link = iou > iou_thr - eps_iou
    affected = seed.copy()
    frontier = affected.copy()
    while frontier.any():
        new = link[frontier].any(0) & ~affected
        affected |= new
        frontier = new
    pos = {int(p): k for k, p in enumerate(cand)}
    diff = sorted(set(int(v) for v in kept_a) ^ set(int(v) for v in kept_b))
    unexplained = [p for p in diff if p not in pos or not affected[pos[p]]]
    # `affected / candidates` is the vacuity check of this proof: a closure that swallows most candidates explains
    # anything, so callers assert affected_frac <= 0.10 next to `unexplained == []`.
    return dict(candidates=int(len(cand)), marginal_seeds=int(seed.sum()), affected=int(affected.sum()),
                affected_frac=float(affected.sum()) / max(1, len(cand)),
                min_score_margin=float(np.abs(scores - score_thr).min()),
                min_iou_margin=float(np.abs(iou[iou > 0] - iou_thr).min()) if (iou > 0).any() else None,
                differing_priors=diff, unexplained=unexplained)


def head_used(det, head):
    """The written part of a flat head buffer: cat over levels of rows[..., :6] (slots 6, 7 are never written)."""
    return torch.cat([r[..., :6].reshape(r.shape[0], -1) for r in det.head_levels(head)], dim=1)


def write_record(name, rec):
    """Persist a parity record where gpurun merges it back (gpurun_out/) — copied to profiles/ by hand."""
    d = os.path.join(ROOT, 'gpurun_out')
    os.makedirs(d, exist_ok=True)
    with open(os.path.join(d, name), 'w') as f:
        json.dump(rec, f, indent=1, sort_keys=True)


def unscale_boxes_np(boxes, scales):
    """scale_bbox(boxes, 1 / scales) (reference trackers/utils.py:58-73, as ocsort_disparity.py:95-97 applies it to
    the tracks), float64 numpy."""
    boxes, inv = np.asarray(boxes, np.float64), 1.0 / np.asarray(scales, np.float64)
    cx, cy = (boxes[:, 0] + boxes[:, 2]) / 2, (boxes[:, 1] + boxes[:, 3]) / 2
    w, h = (boxes[:, 2] - boxes[:, 0]) * inv, (boxes[:, 3] - boxes[:, 1]) * inv
    return np.stack([cx - w / 2, cy - h / 2, cx + w / 2, cy + h / 2], 1)


def match_track_rows(got, ref, tol=1e-3):
    """Pair the track rows of one frame of two runs by their boxes (a track's box IS the box of the detection it was
    matched to, and NMS leaves no two detections closer than IoU 0.5, so the pairing is unambiguous): mutual nearest
    neighbours in L-inf whose distance is within `tol` * max(1, |coordinate|, box extent).
    -> (pairs [(i_got, j_ref)], rows only in got, rows only in ref, largest relative error over the pairs)."""
    got, ref = np.asarray(got, np.float64).reshape(-1, 4), np.asarray(ref, np.float64).reshape(-1, 4)
    if len(got) == 0 or len(ref) == 0:
        return [], list(range(len(got))), list(range(len(ref))), 0.0
    d = np.abs(got[:, None, :] - ref[None, :, :])                      # (G, R, 4)
    ext = np.maximum(ref[:, 2] - ref[:, 0], ref[:, 3] - ref[:, 1])
    scale = np.maximum(np.maximum(np.abs(ref), 1.0), ext[:, None])    # (R, 4)
    rel = (d / scale[None]).max(-1)                                    # (G, R)
    jg, ig = rel.argmin(1), rel.argmin(0)
    pairs, worst = [], 0.0
    for i, j in enumerate(jg):
        if ig[j] == i and rel[i, j] <= tol:
            pairs.append((i, int(j)))
            worst = max(worst, float(rel[i, j]))
    pg, pr = {i for i, _ in pairs}, {j for _, j in pairs}
    return pairs, [i for i in range(len(got)) if i not in pg], [j for j in range(len(ref)) if j not in pr], worst
