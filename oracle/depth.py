"""ORACLE (test infrastructure): numpy/torch restatement of the reference's per-box depth step,
mmtrack/models/mot/ocsort_disparity.py:113-175, and scale_bbox, mmtrack/models/trackers/utils.py:58-73.
In-tree reference code, followed line by line (np.int -> int: same truncation toward zero).
Parity pinning: no fixtures exist in the reference; pinned by its own source text only."""
import warnings

import numpy as np
import torch


def disp2depth(disp, baseline=0.25, focal_length=640):
    """ocsort_disparity.py:132-134 (torch fp32)."""
    return baseline * focal_length / (disp + 1e-6)


def extract_depth(depth, bboxes):
    """ocsort_disparity.py:136-175.  depth: torch (1,1,H,W) or (H,W); bboxes: torch (M,4)."""
    depth = depth.cpu().numpy().squeeze()
    values, scales = [], []
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')  # np.mean of empty slices -> NaN, as in the reference
        for box in bboxes:
            box = box.cpu().numpy().astype(int)
            depth_box = depth[box[1]: box[3], box[0]: box[2]]
            w = box[2] - box[0]
            d_v = depth_box[(depth_box < 150) & (depth_box > 0)]
            len_d = len(d_v)
            if len_d < 1 or w > 800:
                values.append(-1)
                scales.append(1.)
                continue
            d_sorted = np.sort(d_v, axis=None)
            d_mid = d_sorted[len_d // 2]
            v_tl = np.mean(depth[box[1]: box[1] + 2, box[0]: box[0] + 2])
            v_tr = np.mean(depth[box[1]: box[1] + 2, box[2] - 2: box[2]])
            v_bl = np.mean(depth[box[3] - 2: box[3], box[0]: box[0] + 2])
            v_br = np.mean(depth[box[3] - 2: box[3], box[2] - 2: box[2]])
            w_start = min(1 - sum([v_tl, v_tr, v_bl, v_br] > d_mid) / 4, 0.4) * len_d
            w_end = w_start + 0.6 * len_d
            d_seg = d_sorted[int(w_start): int(w_end)]
            if len(d_seg) == 0:
                d_seg = d_sorted[:-1]
            d = np.mean(d_seg)
            values.append(d)
            scale = min(d * d / 1, 3.)
            scale = max(scale, 1.)
            scales.append(scale)
    return values, scales


def scale_bbox(bboxes, scales):
    """trackers/utils.py:58-73."""
    cx = (bboxes[:, 0] + bboxes[:, 2]) / 2
    cy = (bboxes[:, 1] + bboxes[:, 3]) / 2
    w = (bboxes[:, 2] - bboxes[:, 0]) * scales
    h = (bboxes[:, 3] - bboxes[:, 1]) * scales
    return torch.cat((cx[:, None] - w[:, None] / 2, cy[:, None] - h[:, None] / 2,
                      cx[:, None] + w[:, None] / 2, cy[:, None] + h[:, None] / 2), dim=-1).reshape(-1, 4)


def bbox_postp_depth(bboxes, disp, baseline=0.25, focal_length=640):
    """ocsort_disparity.py:113-130 for disp (1,3,H,W): -> d_values (list), scales (Tensor), scaled boxes."""
    depth = disp2depth(disp[:, 0:1, :, :], baseline, focal_length)
    d_value, scales = extract_depth(depth, bboxes)
    scales = torch.Tensor(scales).to(bboxes)
    return d_value, scales, scale_bbox(bboxes, scales)
