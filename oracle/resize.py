"""TEST INFRASTRUCTURE (oracle): numpy restatement of the resampling behind Resize_Disparity with a non-identity scale.

Reference: mmtrack/datasets/transforms/transforms_disparity.py:23-137 - the image goes through mmdet Resize._resize_img
(mmcv.imrescale / imresize, cv2 backend, INTER_LINEAR ['bilinear']), disp_postp / disp_mask / depth_postp through
mmcv.imrescale(..., interpolation='nearest') (:52-112).  mmcv 2.0.0rc3 and OpenCV are un-vendored and absent here:
the algorithms below restate OpenCV's published imgproc/resize.cpp [upstream-memory]; PARITY UNPINNED against cv2 itself.

  * new size       mmcv.rescale_size: (int(w * f + 0.5), int(h * f + 0.5)), f = min(long / max(h, w), short / min(h, w))
  * INTER_LINEAR   8-bit: fx = (float)((dx + 0.5) * scale - 0.5), sx = floor(fx), fx -= sx (zeroed where the tap pair leaves
                   the row), taps cvRound((1 - fx) * 2048), cvRound(fx * 2048) (half to even), int32 horizontal pass,
                   vertical pass (((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2 with the row index clamped;
                   an exact 2 x 2 decimation is computed as the box mean (a + b + c + d + 2) >> 2 (cv2 switches to its
                   fast INTER_AREA there)
  * INTER_NEAREST  sx = min(floor(dx * src_w / dst_w), src_w - 1), likewise for rows
"""
import numpy as np


def rescale_size(h, w, scale):
    """mmcv.rescale_size for a (long, short) / (w, h) scale tuple with keep_ratio: -> (new_w, new_h)."""
    f = min(max(scale) / max(h, w), min(scale) / min(h, w))
    return int(w * f + 0.5), int(h * f + 0.5)


def _taps(n_dst, n_src, zero_at_border):
    scale = np.float64(n_src) / np.float64(n_dst)
    d = np.arange(n_dst, dtype=np.float64)
    f = ((d + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int64)
    f = (f - s.astype(np.float32)).astype(np.float32)
    if zero_at_border:
        lo, hi = s < 0, s >= n_src - 1
        f = np.where(lo | hi, np.float32(0), f)
        s = np.where(lo, 0, np.where(hi, n_src - 1, s))
    a0 = np.rint((np.float32(1) - f) * np.float32(2048)).astype(np.int64)
    a1 = np.rint(f * np.float32(2048)).astype(np.int64)
    return s, a0, a1


def resize_bilinear_u8(img, h2, w2):
    """img: uint8 (h, w) or (h, w, c) -> (h2, w2[, c]) by OpenCV's 8-bit INTER_LINEAR."""
    img = np.asarray(img)
    assert img.dtype == np.uint8
    h, w = img.shape[:2]
    x = img.reshape(h, w, -1).astype(np.int64)
    if h == 2 * h2 and w == 2 * w2:
        out = (x[0::2, 0::2] + x[0::2, 1::2] + x[1::2, 0::2] + x[1::2, 1::2] + 2) >> 2
        return out.astype(np.uint8).reshape((h2, w2) + img.shape[2:])
    sx, ax0, ax1 = _taps(w2, w, True)
    sy, b0, b1 = _taps(h2, h, False)
    x1 = np.minimum(sx + 1, w - 1)
    H = x[:, sx] * ax0[None, :, None] + x[:, x1] * ax1[None, :, None]            # (h, w2, c) int
    y0, y1 = np.clip(sy, 0, h - 1), np.clip(sy + 1, 0, h - 1)
    S0, S1 = H[y0], H[y1]
    v = (((b0[:, None, None] * (S0 >> 4)) >> 16) + ((b1[:, None, None] * (S1 >> 4)) >> 16) + 2) >> 2
    return np.clip(v, 0, 255).astype(np.uint8).reshape((h2, w2) + img.shape[2:])


def resize_nearest(a, h2, w2):
    """a: (h, w[, c]) of any dtype -> (h2, w2[, c]) by OpenCV's INTER_NEAREST."""
    a = np.asarray(a)
    h, w = a.shape[:2]
    sx = np.minimum(np.floor(np.arange(w2, dtype=np.float64) * (np.float64(w) / np.float64(w2))).astype(np.int64), w - 1)
    sy = np.minimum(np.floor(np.arange(h2, dtype=np.float64) * (np.float64(h) / np.float64(h2))).astype(np.int64), h - 1)
    return a[sy][:, sx]
