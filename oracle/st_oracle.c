/*
 * ORACLE — test infrastructure, NOT product code.
 * Plain-C CPU restatement of the index-producing / byte-exact parts of the hot path:
 *   - decode + score filter + sort + greedy NMS
 *   - per-box depth extraction (extract_depth)
 *   - stereo cost volume + soft-argmin (the module the north star adds)
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may build/load this file.
 *
 * PARITY PINNING.  The reference holds no tests, golden vectors or fixtures for this path
 * (no tests/ directory) and its Python cannot be imported in the build
 * container (mmcv/mmdet/mmyolo absent: ordinary ModuleNotFoundError).  The decode/NMS arithmetic
 * lives in un-vendored third-party code (mmyolo 0.2.0 YOLOXHead.predict_by_feat /
 * YOLOXBBoxCoder.decode, mmdet 3.0.0rc4 filter_scores_and_topk / MlvlPointGenerator, mmcv
 * 2.0.0rc3 ops.nms `nms_cpu`), restated here from their published algorithms and anchored on
 * the reference's own call sites and thresholds:
 *   mmtrack/models/detectors/yolo_detector_disparity_v1.py:121-122 (bbox_head.predict),
 *   configs/stereo_tracking/ocsort/yolox_s_mmyolo_mot_airdrone_disp.py:42 (score_thr 0.01, iou 0.5),
 *   configs/_base_/yolox_s_8x8_mmyolo.py:75-81 (yolox_style, multi_label, max_per_img).
 * => decode/NMS: "parity unpinned" at the third-party boundary.
 * extract_depth follows mmtrack/models/mot/ocsort_disparity.py:113-175 line by line (in-tree).
 * The cost-volume module has no reference implementation at all: this file
 * IS its specification.
 *
 * Floating point: compile with -O2 -ffp-contract=off.  Every float operation below is a single
 * IEEE-754 binary32 operation in a fixed order so the HIP kernels can match bit for bit.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ---- exact helpers ------------------------------------------------------------------------- */

/* Cephes-style expf: range reduction by ln2 (hi/lo), degree-5 polynomial, exact ldexp.
 * Not libm's expf on purpose: this sequence is reproducible on any IEEE machine. */
/* The element loops of the stereo functions below run on OpenMP threads (every output element is computed by the same
 * sequential code, so results do not depend on the thread count); oracle_set_threads bounds them (bench.py's
 * cpu_baseline states the threads it used). */
#ifdef _OPENMP
#include <omp.h>
int oracle_set_threads(int n) { if (n > 0) omp_set_num_threads(n); return omp_get_max_threads(); }
#else
int oracle_set_threads(int n) { (void)n; return 1; }
#endif

static float st_expf(float x) {
  if (x > 88.72283f) return INFINITY;
  if (x < -103.0f) return 0.0f;
  const float n = rintf(x * 1.44269504088896341f);
  float r = fmaf(n, -0.693359375f, x);
  r = fmaf(n, 2.12194440e-4f, r);
  float p = 1.9875691500e-4f;
  p = fmaf(p, r, 1.3981999507e-3f);
  p = fmaf(p, r, 8.3334519073e-3f);
  p = fmaf(p, r, 4.1665795894e-2f);
  p = fmaf(p, r, 1.6666665459e-1f);
  p = fmaf(p, r, 5.0000001201e-1f);
  const float r2 = r * r;
  p = fmaf(p, r2, r);
  p = p + 1.0f;
  return ldexpf(p, (int)n);
}
static float st_sigmoidf(float x) { return 1.0f / (1.0f + st_expf(-x)); }

float oracle_expf(float x) { return st_expf(x); }
float oracle_sigmoidf(float x) { return st_sigmoidf(x); }

/* ---- decode + NMS -------------------------------------------------------------------------- */

typedef struct {
  float score;
  int32_t prior; /* flat index prior * nc + class: filter_scores_and_topk's nonzero() order */
  float box[4];
} Cand;

static int cand_cmp(const void* a, const void* b) {
  const Cand* x = (const Cand*)a;
  const Cand* y = (const Cand*)b;
  if (x->score > y->score) return -1; /* score descending */
  if (x->score < y->score) return 1;
  return (x->prior > y->prior) - (x->prior < y->prior); /* tie: lower prior index first */
}

/*
 * head: per level l a float[N][h_l*w_l][8] block at head + lvl_off[l]
 *       (row = cls logit, reg x, y, w, h, obj logit, 2 unused) — flatten order h-major then w,
 *       i.e. permute(0,2,3,1) as predict_by_feat does; levels concatenated 8 -> 16 -> 32.
 * Steps (mmyolo predict_by_feat, yolox_style=True, multi_label -> False for 1 class):
 *   priors (x*s, y*s), offset 0                      (MlvlPointGenerator)
 *   score = sigmoid(cls) * sigmoid(obj)
 *   xy = pred_xy * s + prior ; wh = exp(pred_wh) * s ; xyxy = xy -/+ wh/2   (YOLOXBBoxCoder.decode)
 *   keep score > score_thr, sort descending               (filter_scores_and_topk, nms_pre >= #priors)
 *   boxes = (boxes - pad) / scale_factor                  (rescale, before NMS)
 *   greedy NMS, suppress iff inter/(a_i + a_j - inter) > iou_thr, areas (x2-x1)*(y2-y1)  (mmcv nms_cpu, offset 0)
 *   clamp x to [0, ori_w], y to [0, ori_h]; no max_per_img truncation in yolox_style
 * out_count[n] is the number kept; only the first max_det are stored.
 */
/* Several classes (nc = 2..3; head row = nc class logits, x, y, w, h, obj): multi_label decode - every (prior, class)
 * pair with sigmoid(cls_c) * sigmoid(obj) > score_thr is a candidate, ties in nonzero() order (prior-major) - and
 * class-aware NMS as mmcv batched_nms does it [upstream-memory, mmcv 2.0.0rc3]: the IoU is taken on
 * boxes + label * (boxes.max() + 1) (fp32), so boxes of different classes never overlap and same-class pairs see the
 * coordinates AFTER the offset addition rounded them.  nc <= 1: the single-class path above, unchanged. */
/* Wide heads and multi_label = False.  Head row = `hr` floats per prior (8 for nc <= 3, nc + 5 rounded up to a multiple
 * of 4 beyond: st_head_row_floats).  multi_label = 0 (mmyolo predict_by_feat, `if cfg.multi_label is False`
 * [upstream-memory, mmyolo 0.2.0]): scores.max(1) over score_c = sigmoid(cls_c) * sigmoid(obj) - the FIRST maximum on
 * ties (torch.max on CPU) - and the score threshold on that single (prior, class) pair; NMS stays class-aware. */
int oracle_decode_nms_gen(const float* head, int N, int num_levels, const int* lvl_h, const int* lvl_w,
                          const int* lvl_stride, const size_t* lvl_off, float score_thr, float iou_thr,
                          int max_det, float scale_x, float scale_y, float pad_left, float pad_top,
                          float ori_w, float ori_h, int nc, int hr, int multi_label, float* out_boxes,
                          float* out_scores, int64_t* out_labels, int32_t* out_prior, int32_t* out_count) {
  if (nc < 1) nc = 1;
  if (nc + 5 > hr) return -2;
  int P = 0;
  for (int l = 0; l < num_levels; ++l) P += lvl_h[l] * lvl_w[l];
  Cand* cand = (Cand*)malloc(sizeof(Cand) * (size_t)(P > 0 ? P : 1) * nc);
  unsigned char* sup = (unsigned char*)malloc((size_t)(P > 0 ? P : 1) * nc);
  if (!cand || !sup) { free(cand); free(sup); return -1; }
  for (int n = 0; n < N; ++n) {
    int K = 0, prior = 0;
    for (int l = 0; l < num_levels; ++l) {
      const int h = lvl_h[l], w = lvl_w[l];
      const float s = (float)lvl_stride[l];
      const float* base = head + lvl_off[l] + (size_t)n * h * w * hr;
      for (int py = 0; py < h; ++py)
        for (int px = 0; px < w; ++px, ++prior) {
          const float* row = base + ((size_t)py * w + px) * hr;
          const float sobj = st_sigmoidf(row[nc + 4]);
          const float tx = row[nc] * s, ty = row[nc + 1] * s;
          const float cx = tx + (float)px * s, cy = ty + (float)py * s;
          const float bw = st_expf(row[nc + 2]) * s, bh = st_expf(row[nc + 3]) * s;
          const float hw = bw / 2.0f, hh = bh / 2.0f;
          int c_lo = 0, c_hi = nc;
          if (!multi_label) {   /* one candidate per prior: its best class, first maximum */
            float best = -1.0f;
            for (int c = 0; c < nc; ++c) {
              const float sc = st_sigmoidf(row[c]) * sobj;
              if (sc > best) { best = sc; c_lo = c; }
            }
            c_hi = c_lo + 1;
          }
          for (int c = c_lo; c < c_hi; ++c) {
            const float score = st_sigmoidf(row[c]) * sobj;
            if (!(score > score_thr)) continue;
            Cand* q = &cand[K++];
            q->score = score;
            q->prior = prior * nc + c;
            q->box[0] = ((cx - hw) - pad_left) / scale_x;
            q->box[1] = ((cy - hh) - pad_top) / scale_y;
            q->box[2] = ((cx + hw) - pad_left) / scale_x;
            q->box[3] = ((cy + hh) - pad_top) / scale_y;
          }
        }
    }
    float maxc = -INFINITY;
    for (int i = 0; i < K; ++i)
      for (int e = 0; e < 4; ++e) maxc = fmaxf(maxc, cand[i].box[e]);
    qsort(cand, (size_t)K, sizeof(Cand), cand_cmp);
    memset(sup, 0, (size_t)(K > 0 ? K : 1));
    int kept = 0;
    for (int i = 0; i < K; ++i) {
      if (sup[i]) continue;
      const float* bo = cand[i].box;
      const float offi = nc > 1 ? (float)(cand[i].prior % nc) * (maxc + 1.0f) : 0.0f;
      float bi[4];
      for (int e = 0; e < 4; ++e) bi[e] = nc > 1 ? bo[e] + offi : bo[e];
      if (kept < max_det) {
        float* ob = out_boxes + ((size_t)n * max_det + kept) * 4;
        ob[0] = fminf(fmaxf(bo[0], 0.0f), ori_w);
        ob[1] = fminf(fmaxf(bo[1], 0.0f), ori_h);
        ob[2] = fminf(fmaxf(bo[2], 0.0f), ori_w);
        ob[3] = fminf(fmaxf(bo[3], 0.0f), ori_h);
        out_scores[(size_t)n * max_det + kept] = cand[i].score;
        out_labels[(size_t)n * max_det + kept] = cand[i].prior % nc;
        out_prior[(size_t)n * max_det + kept] = cand[i].prior / nc;
      }
      ++kept;
      const float ai = (bi[2] - bi[0]) * (bi[3] - bi[1]);
      for (int j = i + 1; j < K; ++j) {
        if (sup[j]) continue;
        const float offj = nc > 1 ? (float)(cand[j].prior % nc) * (maxc + 1.0f) : 0.0f;
        float bj[4];
        for (int e = 0; e < 4; ++e) bj[e] = nc > 1 ? cand[j].box[e] + offj : cand[j].box[e];
        const float aj = (bj[2] - bj[0]) * (bj[3] - bj[1]);
        const float xx1 = fmaxf(bi[0], bj[0]), yy1 = fmaxf(bi[1], bj[1]);
        const float xx2 = fminf(bi[2], bj[2]), yy2 = fminf(bi[3], bj[3]);
        const float w = fmaxf(0.0f, xx2 - xx1), h = fmaxf(0.0f, yy2 - yy1);
        const float inter = w * h;
        const float ovr = inter / ((ai + aj) - inter);
        if (ovr > iou_thr) sup[j] = 1;
      }
    }
    out_count[n] = kept;
  }
  free(cand);
  free(sup);
  return 0;
}

int oracle_decode_nms_mc(const float* head, int N, int num_levels, const int* lvl_h, const int* lvl_w,
                         const int* lvl_stride, const size_t* lvl_off, float score_thr, float iou_thr,
                         int max_det, float scale_x, float scale_y, float pad_left, float pad_top,
                         float ori_w, float ori_h, int nc, float* out_boxes, float* out_scores,
                         int64_t* out_labels, int32_t* out_prior, int32_t* out_count) {
  if (nc + 5 > 8) return -2;
  return oracle_decode_nms_gen(head, N, num_levels, lvl_h, lvl_w, lvl_stride, lvl_off, score_thr, iou_thr, max_det,
                               scale_x, scale_y, pad_left, pad_top, ori_w, ori_h, nc, 8, 1, out_boxes, out_scores,
                               out_labels, out_prior, out_count);
}

int oracle_decode_nms(const float* head, int N, int num_levels, const int* lvl_h, const int* lvl_w,
                      const int* lvl_stride, const size_t* lvl_off, float score_thr, float iou_thr,
                      int max_det, float scale_x, float scale_y, float pad_left, float pad_top,
                      float ori_w, float ori_h, float* out_boxes, float* out_scores,
                      int64_t* out_labels, int32_t* out_prior, int32_t* out_count) {
  int P = 0;
  for (int l = 0; l < num_levels; ++l) P += lvl_h[l] * lvl_w[l];
  Cand* cand = (Cand*)malloc(sizeof(Cand) * (size_t)(P > 0 ? P : 1));
  unsigned char* sup = (unsigned char*)malloc((size_t)(P > 0 ? P : 1));
  if (!cand || !sup) { free(cand); free(sup); return -1; }
  for (int n = 0; n < N; ++n) {
    int K = 0, prior = 0;
    for (int l = 0; l < num_levels; ++l) {
      const int h = lvl_h[l], w = lvl_w[l];
      const float s = (float)lvl_stride[l];
      const float* base = head + lvl_off[l] + (size_t)n * h * w * 8;
      for (int py = 0; py < h; ++py)
        for (int px = 0; px < w; ++px, ++prior) {
          const float* row = base + ((size_t)py * w + px) * 8;
          const float score = st_sigmoidf(row[0]) * st_sigmoidf(row[5]);
          if (!(score > score_thr)) continue;
          const float tx = row[1] * s, ty = row[2] * s;
          const float cx = tx + (float)px * s, cy = ty + (float)py * s;
          const float bw = st_expf(row[3]) * s, bh = st_expf(row[4]) * s;
          const float hw = bw / 2.0f, hh = bh / 2.0f;
          Cand* c = &cand[K++];
          c->score = score;
          c->prior = prior;
          c->box[0] = ((cx - hw) - pad_left) / scale_x;
          c->box[1] = ((cy - hh) - pad_top) / scale_y;
          c->box[2] = ((cx + hw) - pad_left) / scale_x;
          c->box[3] = ((cy + hh) - pad_top) / scale_y;
        }
    }
    qsort(cand, (size_t)K, sizeof(Cand), cand_cmp);
    memset(sup, 0, (size_t)(K > 0 ? K : 1));
    int kept = 0;
    for (int i = 0; i < K; ++i) {
      if (sup[i]) continue;
      const float* bi = cand[i].box;
      if (kept < max_det) {
        float* ob = out_boxes + ((size_t)n * max_det + kept) * 4;
        ob[0] = fminf(fmaxf(bi[0], 0.0f), ori_w);
        ob[1] = fminf(fmaxf(bi[1], 0.0f), ori_h);
        ob[2] = fminf(fmaxf(bi[2], 0.0f), ori_w);
        ob[3] = fminf(fmaxf(bi[3], 0.0f), ori_h);
        out_scores[(size_t)n * max_det + kept] = cand[i].score;
        out_labels[(size_t)n * max_det + kept] = 0;
        out_prior[(size_t)n * max_det + kept] = cand[i].prior;
      }
      ++kept;
      const float ai = (bi[2] - bi[0]) * (bi[3] - bi[1]);
      for (int j = i + 1; j < K; ++j) {
        if (sup[j]) continue;
        const float* bj = cand[j].box;
        const float aj = (bj[2] - bj[0]) * (bj[3] - bj[1]);
        const float xx1 = fmaxf(bi[0], bj[0]), yy1 = fmaxf(bi[1], bj[1]);
        const float xx2 = fminf(bi[2], bj[2]), yy2 = fminf(bi[3], bj[3]);
        const float w = fmaxf(0.0f, xx2 - xx1), h = fmaxf(0.0f, yy2 - yy1);
        const float inter = w * h;
        const float ovr = inter / ((ai + aj) - inter);
        if (ovr > iou_thr) sup[j] = 1;
      }
    }
    out_count[n] = kept;
  }
  free(cand);
  free(sup);
  return 0;
}

/* ---- stereo cost volume / soft-argmin / upsample (specification of the new module) ---------- */

/*
 * featL, featR: NHWC float[N][Hf][Wf][ld], channels [0,C) used.
 * cost[n][y][x][d] = (sum_{c=0}^{C-1} fmaf(L[c], R[x-d][c], acc)) / C  for x-d >= 0, else 0.
 */
int oracle_costvolume(const float* featL, const float* featR, int N, int Hf, int Wf, int C, int ld, int D,
                      float* out_cost) {
#pragma omp parallel for collapse(2) schedule(static)
  for (int n = 0; n < N; ++n)
    for (int y = 0; y < Hf; ++y)
      for (int x = 0; x < Wf; ++x) {
        const float* l = featL + (((size_t)n * Hf + y) * Wf + x) * ld;
        float* oc = out_cost + (((size_t)n * Hf + y) * Wf + x) * D;
        for (int d = 0; d < D; ++d) {
          if (x - d < 0) { oc[d] = 0.0f; continue; }
          const float* r = featR + (((size_t)n * Hf + y) * Wf + (x - d)) * ld;
          float acc = 0.0f;
          for (int c = 0; c < C; ++c) acc = fmaf(l[c], r[c], acc);
          oc[d] = acc / (float)C;
        }
      }
  return 0;
}

/* disp[p] = sum_d d*e_d / sum_d e_d, e_d = exp(T*cost_d - max_d(T*cost_d)), sequential in d. */
int oracle_softargmin(const float* cost, long long npix, int D, float temperature, float* out_disp) {
#pragma omp parallel for schedule(static)
  for (long long p = 0; p < npix; ++p) {
    const float* c = cost + p * D;
    float m = -INFINITY;
    for (int d = 0; d < D; ++d) m = fmaxf(m, temperature * c[d]);
    float s = 0.0f, t = 0.0f;
    for (int d = 0; d < D; ++d) {
      const float e = st_expf(temperature * c[d] - m);
      s += e;
      t = fmaf((float)d, e, t);
    }
    out_disp[p] = t / s;
  }
  return 0;
}

/* bilinear x`scale` upsample (align_corners=False), times scale, zero outside (valid_h, valid_w),
 * replicated into 3 channels NCHW [N][3][H][W] — the `disp_postp` layout
 * (reference loading_disparity.py:85-86 3-channel repeat; transforms_disparity.py:234-249 pad 0). */
int oracle_disp_upsample(const float* lr, int N, int Hf, int Wf, int scale, int H, int W, int valid_h,
                         int valid_w, float* out) {
  const float inv = 1.0f / (float)scale;
#pragma omp parallel for collapse(2) schedule(static)
  for (int n = 0; n < N; ++n)
    for (int Y = 0; Y < H; ++Y)
      for (int X = 0; X < W; ++X) {
        float v = 0.0f;
        if (Y < valid_h && X < valid_w) {
          float sy = ((float)Y + 0.5f) * inv - 0.5f, sx = ((float)X + 0.5f) * inv - 0.5f;
          if (sy < 0.0f) sy = 0.0f;
          if (sx < 0.0f) sx = 0.0f;
          int y0 = (int)sy, x0 = (int)sx;
          if (y0 > Hf - 1) y0 = Hf - 1;
          if (x0 > Wf - 1) x0 = Wf - 1;
          const int y1 = y0 + 1 < Hf ? y0 + 1 : Hf - 1, x1 = x0 + 1 < Wf ? x0 + 1 : Wf - 1;
          const float ly = sy - (float)y0, lx = sx - (float)x0, hy = 1.0f - ly, hx = 1.0f - lx;
          const float* b = lr + (size_t)n * Hf * Wf;
          const float v00 = b[(size_t)y0 * Wf + x0], v01 = b[(size_t)y0 * Wf + x1];
          const float v10 = b[(size_t)y1 * Wf + x0], v11 = b[(size_t)y1 * Wf + x1];
          v = (hy * (hx * v00 + lx * v01) + ly * (hx * v10 + lx * v11)) * (float)scale;
        }
        for (int c = 0; c < 3; ++c) out[(((size_t)n * 3 + c) * H + Y) * W + X] = v;
      }
  return 0;
}

/* Bilinear x`scale` upsampling of an NHWC feature map (align_corners=False, the index arithmetic of
 * oracle_disp_upsample above, no value scaling): in [N][Hf][Wf][C] (row stride in_ld floats per pixel) -> out
 * [N][Hf*scale][Wf*scale][C] dense.  The feature side of the stereo module's FULL-RESOLUTION mode (spec:
 * stereotracking_amd/stereo.py): the reduced stage-1 features are brought to image resolution, where the D = max_disp
 * level volume of north_star's sizing is built. */
int oracle_feat_upsample(const float* in, int N, int Hf, int Wf, int C, int in_ld, int scale, float* out) {
  const int H = Hf * scale, W = Wf * scale;
  const float inv = 1.0f / (float)scale;
#pragma omp parallel for collapse(2) schedule(static)
  for (int n = 0; n < N; ++n)
    for (int Y = 0; Y < H; ++Y)
      for (int X = 0; X < W; ++X) {
        float sy = ((float)Y + 0.5f) * inv - 0.5f, sx = ((float)X + 0.5f) * inv - 0.5f;
        if (sy < 0.0f) sy = 0.0f;
        if (sx < 0.0f) sx = 0.0f;
        int y0 = (int)sy, x0 = (int)sx;
        if (y0 > Hf - 1) y0 = Hf - 1;
        if (x0 > Wf - 1) x0 = Wf - 1;
        const int y1 = y0 + 1 < Hf ? y0 + 1 : Hf - 1, x1 = x0 + 1 < Wf ? x0 + 1 : Wf - 1;
        const float ly = sy - (float)y0, lx = sx - (float)x0, hy = 1.0f - ly, hx = 1.0f - lx;
        const float* b = in + (size_t)n * Hf * Wf * in_ld;
        const float* p00 = b + ((size_t)y0 * Wf + x0) * in_ld;
        const float* p01 = b + ((size_t)y0 * Wf + x1) * in_ld;
        const float* p10 = b + ((size_t)y1 * Wf + x0) * in_ld;
        const float* p11 = b + ((size_t)y1 * Wf + x1) * in_ld;
        float* o = out + (((size_t)n * H + Y) * W + X) * C;
        for (int c = 0; c < C; ++c) o[c] = hy * (hx * p00[c] + lx * p01[c]) + ly * (hx * p10[c] + lx * p11[c]);
      }
  return 0;
}

/* ---- 3-D aggregation of the cost volume (specification; north_star: "3D/2D aggregation") -------------------------
 * One layer = a single-channel 3x3x3 convolution over (d, y, x) of the volume [N][Hf][Wf][D] with zero padding in all
 * three dimensions, optional SiLU:
 *   out[d,y,x] = act( b + sum_{j,k,i} w[i][j][k] * vol[d+i-1, y+j-1, x+k-1] )
 * w = the Conv3d weight (1,1,3,3,3) in (kD, kH, kW) order.  Accumulation order (the kernel's, bit for bit): acc = b;
 * for j (row), for k (column), for i (disparity): acc = fmaf(w[i][j][k], v, acc) with v = 0 outside the volume (the
 * fmaf is executed for padded taps, too).  SiLU = v / (1 + st_expf(-v)). */
int oracle_agg3d(const float* vol, int N, int Hf, int Wf, int D, const float* w27, float bias, int act, float* out) {
#pragma omp parallel for collapse(2) schedule(static)
  for (int n = 0; n < N; ++n)
    for (int y = 0; y < Hf; ++y)
      for (int x = 0; x < Wf; ++x)
        for (int d = 0; d < D; ++d) {
          float acc = bias;
          for (int j = 0; j < 3; ++j)
            for (int k = 0; k < 3; ++k)
              for (int i = 0; i < 3; ++i) {
                const int yy = y + j - 1, xx = x + k - 1, dd = d + i - 1;
                const float v = (yy >= 0 && yy < Hf && xx >= 0 && xx < Wf && dd >= 0 && dd < D)
                                    ? vol[(((size_t)n * Hf + yy) * Wf + xx) * D + dd] : 0.0f;
                acc = fmaf(w27[(i * 3 + j) * 3 + k], v, acc);
              }
          if (act) acc = acc / (1.0f + st_expf(-acc));
          out[(((size_t)n * Hf + y) * Wf + x) * D + d] = acc;
        }
  return 0;
}
