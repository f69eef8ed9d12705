"""ORACLE (test infrastructure): builds oracle/st_oracle.c with gcc and binds it with ctypes.
Only tests/, __graft_entry__ (build + smoke) and bench.py's cpu_baseline leg may import this."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(_HERE, 'st_oracle.c')
OUT_DIR = os.path.join(_HERE, '_build')
LIB = os.path.join(OUT_DIR, 'libst_oracle.so')

_lib = None


def build(force=False):
    os.makedirs(OUT_DIR, exist_ok=True)
    if force or not os.path.exists(LIB) or (os.path.exists(SRC) and os.path.getmtime(LIB) < os.path.getmtime(SRC)):
        subprocess.check_call(['gcc', '-O2', '-std=c11', '-ffp-contract=off', '-fno-fast-math', '-fopenmp', '-fPIC',
                               '-shared', '-o', LIB, SRC, '-lm'])
    return LIB


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB):
            build()
        _lib = C.CDLL(LIB)
        _lib.oracle_expf.restype = C.c_float
        _lib.oracle_expf.argtypes = [C.c_float]
        _lib.oracle_sigmoidf.restype = C.c_float
        _lib.oracle_sigmoidf.argtypes = [C.c_float]
        # OpenMP threads of the element loops: a GPU box reports every CPU of its host (256) but shares 16 of them
        if 'OMP_NUM_THREADS' not in os.environ:
            _lib.oracle_set_threads(C.c_int(min(os.cpu_count() or 1, 16)))
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def set_threads(n):
    """Bound the OpenMP threads of the stereo loops (results do not depend on it); returns the count in effect."""
    lib = load()
    lib.oracle_set_threads.restype = C.c_int
    return int(lib.oracle_set_threads(C.c_int(int(n))))


def head_row_floats(num_classes):
    """floats per prior in the head buffer (the product's st_head_row_floats, restated)."""
    return 8 if num_classes <= 3 else (num_classes + 5 + 3) // 4 * 4


def decode_nms(head, N, levels, score_thr, iou_thr, max_det, ori_shape, scale_factor=(1.0, 1.0), pad_param=None,
               num_classes=1, multi_label=True):
    """head: flat float32 numpy array in the product's head_out layout; levels: [(h, w, stride, float_offset)].
    num_classes > 1: multi_label decode (or multi_label=False: one candidate per prior) + class-aware NMS
    (oracle_decode_nms_gen).
    Returns boxes (N,max_det,4), scores, labels (int64), prior_idx (int32), counts (int32)."""
    lib = load()
    head = np.ascontiguousarray(head, dtype=np.float32)
    L = len(levels)
    lh = (C.c_int * L)(*[l[0] for l in levels])
    lw = (C.c_int * L)(*[l[1] for l in levels])
    ls = (C.c_int * L)(*[l[2] for l in levels])
    lo = (C.c_size_t * L)(*[l[3] for l in levels])
    boxes = np.zeros((N, max_det, 4), np.float32)
    scores = np.zeros((N, max_det), np.float32)
    labels = np.zeros((N, max_det), np.int64)
    prior = np.full((N, max_det), -1, np.int32)
    counts = np.zeros((N,), np.int32)
    pad_left = float(pad_param[2]) if pad_param is not None else 0.0
    pad_top = float(pad_param[0]) if pad_param is not None else 0.0
    f = C.c_float
    if num_classes > 1:
        rc = lib.oracle_decode_nms_gen(_p(head), C.c_int(N), C.c_int(L), lh, lw, ls, lo, f(score_thr), f(iou_thr),
                                       C.c_int(max_det), f(scale_factor[0]), f(scale_factor[1]), f(pad_left),
                                       f(pad_top), f(ori_shape[1]), f(ori_shape[0]), C.c_int(num_classes),
                                       C.c_int(head_row_floats(num_classes)), C.c_int(1 if multi_label else 0),
                                       _p(boxes), _p(scores), _p(labels), _p(prior), _p(counts))
    else:
        rc = lib.oracle_decode_nms(_p(head), C.c_int(N), C.c_int(L), lh, lw, ls, lo, f(score_thr), f(iou_thr),
                                   C.c_int(max_det), f(scale_factor[0]), f(scale_factor[1]), f(pad_left), f(pad_top),
                                   f(ori_shape[1]), f(ori_shape[0]), _p(boxes), _p(scores), _p(labels), _p(prior),
                                   _p(counts))
    if rc != 0:
        raise RuntimeError('oracle_decode_nms failed')
    return boxes, scores, labels, prior, counts


def costvolume(featL, featR, C_, D):
    """featL/featR: float32 numpy (N,Hf,Wf,ld) NHWC -> cost (N,Hf,Wf,D)."""
    lib = load()
    featL = np.ascontiguousarray(featL, np.float32)
    featR = np.ascontiguousarray(featR, np.float32)
    N, Hf, Wf, ld = featL.shape
    out = np.zeros((N, Hf, Wf, D), np.float32)
    lib.oracle_costvolume(_p(featL), _p(featR), C.c_int(N), C.c_int(Hf), C.c_int(Wf), C.c_int(C_), C.c_int(ld),
                          C.c_int(D), _p(out))
    return out


def softargmin(cost, temperature):
    lib = load()
    cost = np.ascontiguousarray(cost, np.float32)
    D = cost.shape[-1]
    out = np.zeros(cost.shape[:-1], np.float32)
    lib.oracle_softargmin(_p(cost), C.c_longlong(out.size), C.c_int(D), C.c_float(temperature), _p(out))
    return out


def disp_upsample(lr, scale, valid_h, valid_w):
    """lr (N,Hf,Wf) -> disp_postp (N,3,Hf*scale,Wf*scale)."""
    lib = load()
    lr = np.ascontiguousarray(lr, np.float32)
    N, Hf, Wf = lr.shape
    out = np.zeros((N, 3, Hf * scale, Wf * scale), np.float32)
    lib.oracle_disp_upsample(_p(lr), C.c_int(N), C.c_int(Hf), C.c_int(Wf), C.c_int(scale), C.c_int(Hf * scale),
                             C.c_int(Wf * scale), C.c_int(valid_h), C.c_int(valid_w), _p(out))
    return out


def feat_upsample(feat, scale, C_=None):
    """feat (N,Hf,Wf,ld) float32, the first C_ channels of every pixel (default: all) -> (N,Hf*scale,Wf*scale,C_)
    (oracle_feat_upsample: bilinear, align_corners=False)."""
    lib = load()
    feat = np.ascontiguousarray(feat, np.float32)
    N, Hf, Wf, ld = feat.shape
    C_ = ld if C_ is None else int(C_)
    out = np.zeros((N, Hf * scale, Wf * scale, C_), np.float32)
    lib.oracle_feat_upsample(_p(feat), C.c_int(N), C.c_int(Hf), C.c_int(Wf), C.c_int(C_), C.c_int(ld), C.c_int(scale),
                             _p(out))
    return out


def agg3d(vol, weight, bias, act):
    """One 3-D aggregation layer (oracle_agg3d): vol (N,Hf,Wf,D) float32, weight (3,3,3) in (kD,kH,kW) order (a Conv3d
    (1,1,3,3,3) weight squeezed), bias scalar, act: SiLU or not -> (N,Hf,Wf,D)."""
    lib = load()
    vol = np.ascontiguousarray(vol, np.float32)
    w = np.ascontiguousarray(np.asarray(weight, np.float32).reshape(27))
    N, Hf, Wf, D = vol.shape
    out = np.zeros_like(vol)
    lib.oracle_agg3d(_p(vol), C.c_int(N), C.c_int(Hf), C.c_int(Wf), C.c_int(D), _p(w), C.c_float(float(bias)),
                     C.c_int(int(bool(act))), _p(out))
    return out
