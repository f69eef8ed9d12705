// Decode + score filter + sort + greedy NMS, fully on device, no host sync.
//
// Restates mmyolo YOLOXHead.predict_by_feat -> YOLOXBBoxCoder.decode ->
// mmdet filter_scores_and_topk -> mmcv.ops.batched_nms/nms (un-vendored third-party code the
// reference calls through yolo_detector_disparity_v1.py:121-122 with thresholds from
// configs/stereo_tracking/ocsort/yolox_s_mmyolo_mot_airdrone_disp.py:42).
//
// Index-producing work must be bit-exact, so every float op here is a single IEEE operation in
// a fixed order (file is compiled with -ffp-contract=off; exp is the polynomial below, not a
// hardware approximation) and oracle/st_oracle.c performs the identical sequence on the CPU.
//
// Pipeline (4 launches + 1 memset, all sizes read from device memory):
//   1 decode_filter : per prior (and class: multi_label when num_classes > 1, filter_scores_and_topk's
//                     (prior, class) pairs): score = sigmoid(cls)*sigmoid(obj), box decode, rescale;
//                     wave-aggregated append of candidates with score > thr
//   2 rank_sort     : rank_i = #{j : key_j > key_i}, key = (score bits, ~prior index) -> a
//                     permutation = stable sort by score desc, prior index asc; O(K^2) compares
//                     spread over the chip (K is a few hundred in practice, <= 19320)
//   3 nms_mask      : 64x64 tiles of IoU > thr bits (upper triangle), one wave per tile; with several classes
//                     the IoU is taken on boxes + label * (max coordinate + 1), mmcv batched_nms's offset trick
//                     (boxes of different classes never overlap; same-class pairs see the rounding the
//                     offset addition causes there, too)
//   4 nms_reduce    : one wave per image walks the 64-box chunks in score order, resolves the
//                     diagonal tile serially on a 64-bit word, ORs kept rows into the
//                     `removed` bitmap, emits kept boxes (clamped) in score order
#include <algorithm>

#include "st_common.h"

namespace st {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned long long u64;

// ---- exact-arithmetic helpers (mirrored verbatim in oracle/st_oracle.c) ----------------------
__device__ __forceinline__ float st_expf(float x) {
  if (x > 88.72283f) return __builtin_inff();
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
__device__ __forceinline__ float st_sigmoidf(float x) { return 1.0f / (1.0f + st_expf(-x)); }

struct DecodeArgs {
  const float* head;
  int num_levels;
  int lvl_h[4], lvl_w[4], lvl_stride[4];
  size_t lvl_off[4];
  int lvl_start[5];  // first flat prior index of each level
  int P;             // priors per image
  int cap;           // max candidates entering NMS (nms_pre)
  int mcap;          // rows / columns of the precomputed IoU bit mask (multiple of 64); candidates past it
                     // are resolved on the fly by the reduce kernel (identical results, slower)
  int Tm;            // mcap / 64: mask words per row
  int batch;
  int nc;            // classes (head row = nc class logits, x, y, w, h, obj)
  int hr;            // floats per head row (head_row_floats(nc): 8 up to 3 classes)
  int single_label;  // several classes, multi_label=False: ONE candidate per prior, its best class
  int cand_cap;      // P * nc: rows of the candidate arrays
  float score_thr, iou_thr;
  int max_det;
  float scale_x, scale_y, pad_left, pad_top, ori_w, ori_h;
  // workspace
  int* count;        // [N]
  u64* cand_key;     // [N][P]
  f32x4* cand_box;   // [N][P]
  f32x4* s_box;      // [N][cap]
  float* s_score;    // [N][cap]
  int* s_idx;        // [N][cap]
  u64* mask;         // [N][mcap][Tm]
  float* maxc;       // [N] largest coordinate over the candidates' boxes (class offsets; nc > 1 only)
};

__global__ __launch_bounds__(256) void decode_filter_kernel(DecodeArgs a) {
  const int n = blockIdx.y;
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  const int nc = a.nc;
  float logit[3] = {0.f, 0.f, 0.f}, obj = 0.f;
  f32x4 box = {0.f, 0.f, 0.f, 0.f};
  const float* row = a.head;     // this prior's head row (class logits of wide heads are read from it below)
  if (p < a.P) {
    int l = 0;
    while (l + 1 < a.num_levels && p >= a.lvl_start[l + 1]) ++l;
    const int q = p - a.lvl_start[l];
    const int w = a.lvl_w[l], hw = a.lvl_h[l] * w;
    const int py = q / w, px = q - py * w;
    const float s = (float)a.lvl_stride[l];
    row = a.head + a.lvl_off[l] + ((size_t)n * hw + q) * a.hr;
    float r5[5];                 // x, y, w, h, obj
    if (nc <= 3) {               // 8-float rows: two 16-byte loads.  nc = 1: cls, x, y, w | h, obj, -, -
      const f32x4 v0 = *reinterpret_cast<const f32x4*>(row);
      const f32x4 v1 = *reinterpret_cast<const f32x4*>(row + 4);
      const float r8[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
      for (int c = 0; c < nc; ++c) logit[c] = r8[c];
      for (int e = 0; e < 5; ++e) r5[e] = r8[nc + e];
    } else {
      for (int e = 0; e < 5; ++e) r5[e] = row[nc + e];
    }
    obj = r5[4];
    const float cx = r5[0] * s + (float)px * s;
    const float cy = r5[1] * s + (float)py * s;
    const float bw = st_expf(r5[2]) * s;
    const float bh = st_expf(r5[3]) * s;
    const float hw2 = bw / 2.0f, hh2 = bh / 2.0f;
    box[0] = ((cx - hw2) - a.pad_left) / a.scale_x;
    box[1] = ((cy - hh2) - a.pad_top) / a.scale_y;
    box[2] = ((cx + hw2) - a.pad_left) / a.scale_x;
    box[3] = ((cy + hh2) - a.pad_top) / a.scale_y;
  }
  const float sobj = st_sigmoidf(obj);
  const int lane = threadIdx.x & 63;
  auto emit = [&](bool valid, float score, int c) {   // wave-aggregated append (called with a uniform trip count)
    const u64 ballot = __ballot(valid);
    if (ballot == 0) return;
    int base = 0;
    if (lane == __builtin_ctzll(ballot)) base = atomicAdd(&a.count[n], __builtin_popcountll(ballot));
    base = __shfl(base, __builtin_ctzll(ballot));
    if (valid) {
      const int pos = base + __builtin_popcountll(ballot & ((1ull << lane) - 1ull));
      const unsigned flat = (unsigned)p * (unsigned)nc + (unsigned)c;   // filter_scores_and_topk's nonzero() order
      a.cand_key[(size_t)n * a.cand_cap + pos] = ((u64)__float_as_uint(score) << 32) | (u64)(0xFFFFFFFFu - flat);
      a.cand_box[(size_t)n * a.cand_cap + pos] = box;
    }
  };
  if (a.single_label) {
    // multi_label=False (mmyolo predict_by_feat): scores.max(1) over score_c = sigmoid(cls_c) * sigmoid(obj), the FIRST
    // maximum on ties, then the score threshold on that one (prior, class) pair
    float best = -1.0f;
    int bc = 0;
    for (int c = 0; c < nc; ++c) {
      const float lg = nc <= 3 ? logit[c] : (p < a.P ? row[c] : 0.f);
      const float score = st_sigmoidf(lg) * sobj;
      if (score > best) { best = score; bc = c; }
    }
    emit(p < a.P && best > a.score_thr, best, bc);
    return;
  }
  for (int c = 0; c < nc; ++c) {   // uniform trip count
    const float lg = nc <= 3 ? logit[c] : (p < a.P ? row[c] : 0.f);
    const float score = st_sigmoidf(lg) * sobj;
    emit(p < a.P && score > a.score_thr, score, c);
  }
}

__global__ __launch_bounds__(256) void rank_sort_kernel(DecodeArgs a) {
  __shared__ u64 skeys[256];
  const int n = blockIdx.y;
  const int K = a.count[n];
  const u64* keys = a.cand_key + (size_t)n * a.cand_cap;
  const f32x4* boxes = a.cand_box + (size_t)n * a.cand_cap;
  if (a.nc > 1 && blockIdx.x == 0) {   // boxes.max() over the candidates (mmcv batched_nms: max_coordinate)
    __shared__ float smax[256];
    float m = -__builtin_inff();
    for (int i = threadIdx.x; i < K; i += 256) {
      const f32x4 b = boxes[i];
      m = fmaxf(m, fmaxf(fmaxf(b[0], b[1]), fmaxf(b[2], b[3])));
    }
    smax[threadIdx.x] = m;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
      if ((int)threadIdx.x < st) smax[threadIdx.x] = fmaxf(smax[threadIdx.x], smax[threadIdx.x + st]);
      __syncthreads();
    }
    if (threadIdx.x == 0) a.maxc[n] = smax[0];
  }
  for (int i0 = blockIdx.x * 256; i0 < K; i0 += gridDim.x * 256) {
    const int i = i0 + threadIdx.x;
    const u64 ki = i < K ? keys[i] : 0ull;
    int rank = 0;
    for (int j0 = 0; j0 < K; j0 += 256) {
      __syncthreads();
      skeys[threadIdx.x] = (j0 + (int)threadIdx.x < K) ? keys[j0 + threadIdx.x] : 0ull;
      __syncthreads();
      const int lim = min(256, K - j0);
      for (int j = 0; j < lim; ++j) rank += skeys[j] > ki;
    }
    if (i < K && rank < a.cap) {
      const size_t o = (size_t)n * a.cap + rank;
      a.s_box[o] = boxes[i];
      a.s_score[o] = __uint_as_float((unsigned)(ki >> 32));
      a.s_idx[o] = (int)(0xFFFFFFFFu - (unsigned)(ki & 0xFFFFFFFFull));
    }
  }
}

// mmcv nms (cpu/cuda) IoU, offset = 0:  inter / (area_i + area_j - inter) > thr
__device__ __forceinline__ bool iou_gt(const f32x4& bi, float ai, const f32x4& bj, float thr) {
  const float aj = (bj[2] - bj[0]) * (bj[3] - bj[1]);
  const float xx1 = fmaxf(bi[0], bj[0]), yy1 = fmaxf(bi[1], bj[1]);
  const float xx2 = fminf(bi[2], bj[2]), yy2 = fminf(bi[3], bj[3]);
  const float w = fmaxf(0.0f, xx2 - xx1), h = fmaxf(0.0f, yy2 - yy1);
  const float inter = w * h;
  const float ovr = inter / ((ai + aj) - inter);
  return ovr > thr;
}

// the box NMS compares: boxes + label * (max_coordinate + 1) when there are several classes (label = flat % nc)
__device__ __forceinline__ f32x4 nms_box(const DecodeArgs& a, int n, int i, f32x4 b) {
  if (a.nc > 1) {
    const float off = (float)(a.s_idx[(size_t)n * a.cap + i] % a.nc) * (a.maxc[n] + 1.0f);
    b[0] = b[0] + off; b[1] = b[1] + off; b[2] = b[2] + off; b[3] = b[3] + off;
  }
  return b;
}

__global__ __launch_bounds__(256) void nms_mask_kernel(DecodeArgs a) {
  const int n = blockIdx.y;
  const int K = min(min(a.count[n], a.cap), a.mcap);   // the mask covers the first mcap candidates
  const int T = (K + 63) >> 6;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const f32x4* boxes = a.s_box + (size_t)n * a.cap;
  u64* mask = a.mask + (size_t)n * a.mcap * a.Tm;
  const long long items = (long long)T * T;
  for (long long it = (long long)blockIdx.x * 4 + wave; it < items; it += (long long)gridDim.x * 4) {
    const int ti = (int)(it / T), tj = (int)(it - (long long)ti * T);
    if (tj < ti) continue;  // wave-uniform
    const int j = tj * 64 + lane, i = ti * 64 + lane;
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    const f32x4 bj = j < K ? nms_box(a, n, j, boxes[j]) : zero;  // column box of this lane, broadcast by readlane
    const f32x4 bi = i < K ? nms_box(a, n, i, boxes[i]) : zero;
    const float ai = (bi[2] - bi[0]) * (bi[3] - bi[1]);
    const int jn = min(64, K - tj * 64);
    u64 bits = 0;
    for (int jj = 0; jj < jn; ++jj) {  // wave-uniform trip count
      f32x4 b;
      b[0] = __shfl(bj[0], jj); b[1] = __shfl(bj[1], jj);
      b[2] = __shfl(bj[2], jj); b[3] = __shfl(bj[3], jj);
      const bool later = (ti != tj) || (jj > lane);
      if (later && iou_gt(bi, ai, b, a.iou_thr)) bits |= 1ull << jj;
    }
    if (i < K) mask[(size_t)i * a.Tm + tj] = bits;
  }
}

__global__ __launch_bounds__(64) void nms_reduce_kernel(DecodeArgs a, float* out_boxes, float* out_scores,
                                                        long long* out_labels, int* out_prior,
                                                        int* out_count) {
  __shared__ u64 removed[512];
  const int n = blockIdx.x;
  const int lane = threadIdx.x;
  const int K = min(a.count[n], a.cap);
  const int T = (K + 63) >> 6;
  const int Tm = min(T, a.Tm);   // chunks [0, Tm) have their IoU bits in the mask, chunks [Tm, T) do not
  const f32x4* boxes = a.s_box + (size_t)n * a.cap;
  const float* scores = a.s_score + (size_t)n * a.cap;
  const int* idx = a.s_idx + (size_t)n * a.cap;
  const u64* mask = a.mask + (size_t)n * a.mcap * a.Tm;
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
  for (int w = lane; w < T; w += 64) removed[w] = 0ull;
  __syncthreads();
  int outcount = 0;
  for (int c = 0; c < T; ++c) {
    u64 rem = removed[c];
    const int i = c * 64 + lane;
    const int nb = min(64, K - c * 64);
    const f32x4 bo = i < K ? boxes[i] : zero;                  // the box that is returned
    const f32x4 bi = i < K ? nms_box(a, n, i, bo) : zero;      // the box NMS compares
    const float ai = (bi[2] - bi[0]) * (bi[3] - bi[1]);
    u64 d = 0ull;
    if (c < Tm) {
      d = i < K ? mask[(size_t)i * a.Tm + c] : 0ull;
    } else {   // diagonal tile on the fly (same bits nms_mask_kernel would have written)
      for (int jj = 0; jj < nb; ++jj) {
        f32x4 b;
        b[0] = __shfl(bi[0], jj); b[1] = __shfl(bi[1], jj);
        b[2] = __shfl(bi[2], jj); b[3] = __shfl(bi[3], jj);
        if (jj > lane && i < K && iou_gt(bi, ai, b, a.iou_thr)) d |= 1ull << jj;
      }
    }
    u64 keep = 0ull;
    for (int b = 0; b < nb; ++b) {  // wave-uniform serial resolve of the diagonal tile
      const u64 db = __shfl(d, b);
      if (!((rem >> b) & 1ull)) {
        keep |= 1ull << b;
        rem |= db;
      }
    }
    // OR the rows of the boxes kept in this chunk into the later words: lane = kept row, streaming its own
    // mask row (independent loads, deep memory-level parallelism) and ds_or-ing into the LDS bitmap
    if (c < Tm && ((keep >> lane) & 1ull)) {
      const u64* row = mask + (size_t)i * a.Tm;
      for (int w = c + 1; w < Tm; ++w) {
        const u64 m = row[w];
        if (m) atomicOr(reinterpret_cast<unsigned long long*>(&removed[w]), (unsigned long long)m);
      }
    }
    // column chunks the mask does not cover: lane = column, the kept rows of this chunk are broadcast one by one
    for (int w = max(c + 1, Tm); w < T; ++w) {   // wave-uniform
      const int j = w * 64 + lane;
      const f32x4 bj = j < K ? nms_box(a, n, j, boxes[j]) : zero;
      bool sup = false;
      u64 kk = keep;
      while (kk) {
        const int b = __builtin_ctzll(kk);
        kk &= kk - 1ull;
        f32x4 r;
        r[0] = __shfl(bi[0], b); r[1] = __shfl(bi[1], b); r[2] = __shfl(bi[2], b); r[3] = __shfl(bi[3], b);
        const float ar = __shfl(ai, b);
        sup = sup || iou_gt(r, ar, bj, a.iou_thr);
      }
      const u64 bits = __ballot(sup && j < K);
      if (lane == 0 && bits) removed[w] |= bits;
    }
    if ((keep >> lane) & 1ull) {
      const int pos = outcount + __builtin_popcountll(keep & ((1ull << lane) - 1ull));
      if (pos < a.max_det) {
        f32x4 o;
        o[0] = fminf(fmaxf(bo[0], 0.0f), a.ori_w);
        o[1] = fminf(fmaxf(bo[1], 0.0f), a.ori_h);
        o[2] = fminf(fmaxf(bo[2], 0.0f), a.ori_w);
        o[3] = fminf(fmaxf(bo[3], 0.0f), a.ori_h);
        const size_t oo = (size_t)n * a.max_det + pos;
        *reinterpret_cast<f32x4*>(out_boxes + oo * 4) = o;
        out_scores[oo] = scores[i];
        out_labels[oo] = idx[i] % a.nc;
        out_prior[oo] = idx[i] / a.nc;
      }
    }
    outcount += __builtin_popcountll(keep);
    __syncthreads();
  }
  // more candidates than the NMS handles (only possible with several classes): reported as an overflow nobody can
  // mistake for a count (every consumer treats count > max_det as an error)
  if (lane == 0) out_count[n] = a.count[n] > a.cap ? 0x7FFFFFFF : outcount;
  // rows past the count are DEFINED by this kernel (zero boxes / scores / labels, prior index -1): the caller's
  // output buffers are persistent and need no clearing launch per batch
  const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
  for (int pos = min(outcount, a.max_det) + lane; pos < a.max_det; pos += 64) {
    const size_t oo = (size_t)n * a.max_det + pos;
    *reinterpret_cast<f32x4*>(out_boxes + oo * 4) = z4;
    out_scores[oo] = 0.f;
    out_labels[oo] = 0;
    out_prior[oo] = -1;
  }
}

struct DecodeLayout {
  size_t count, cand_key, cand_box, s_box, s_score, s_idx, mask, maxc, total;
  int P, cap, mcap, Tm, nc, cand_cap;
};

static size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

static int decode_layout(const StDecodeDesc& d, DecodeLayout& L) {
  ST_REQUIRE(d.struct_size == (int)sizeof(StDecodeDesc), "st_decode_nms: struct_size mismatch");
  ST_REQUIRE(d.batch > 0 && d.num_levels > 0 && d.num_levels <= 4, "st_decode_nms: bad batch/levels");
  ST_REQUIRE(d.max_det > 0, "st_decode_nms: max_det must be positive");
  ST_REQUIRE(d.scale_x > 0 && d.scale_y > 0, "st_decode_nms: scale factors must be positive");
  long long P = 0;
  for (int l = 0; l < d.num_levels; ++l) {
    ST_REQUIRE(d.level_h[l] > 0 && d.level_w[l] > 0 && d.level_stride[l] > 0, "st_decode_nms: bad level %d", l);
    P += (long long)d.level_h[l] * d.level_w[l];
  }
  ST_REQUIRE(P < (1 << 28), "st_decode_nms: too many priors");
  L.nc = d.num_classes > 0 ? d.num_classes : 1;
  ST_REQUIRE(L.nc <= 1024, "st_decode_nms: num_classes must be in [1, 1024]");
  ST_REQUIRE(P * L.nc < (1ll << 31) - 1, "st_decode_nms: priors x classes exceeds 2^31");
  L.P = (int)P;
  // (prior, class) pairs (multi_label; one class or single_label: the priors)
  L.cand_cap = d.single_label ? L.P : L.P * L.nc;
  // nms_pre default (100000) >= priors: every candidate enters NMS.  The reduce kernel's LDS bitmap holds 32768
  // candidates: one class never exceeds it at the supported sizes; with several classes more than 32768 pairs above
  // score_thr in one image are reported as an overflow (count = INT_MAX), never dropped silently
  ST_REQUIRE((L.nc > 1 && !d.single_label) || (L.P + 63) / 64 <= 512,
             "st_decode_nms: more than 32768 candidates per image not supported");
  L.cap = std::min(L.cand_cap, 32768);
  // IoU bit mask for the first `mcap` candidates in score order (a few hundred to a few thousand pass the score
  // threshold in practice); later candidates are resolved on the fly: the workspace no longer grows with P^2
  ST_REQUIRE(d.nms_mask_rows >= 0, "st_decode_nms: nms_mask_rows must be >= 0");
  L.mcap = std::min((L.cap + 63) / 64 * 64, ((d.nms_mask_rows > 0 ? d.nms_mask_rows : 4096) + 63) / 64 * 64);
  L.Tm = L.mcap / 64;
  size_t o = 0;
  L.count = o; o = align256(o + sizeof(int) * d.batch);
  L.cand_key = o; o = align256(o + sizeof(u64) * (size_t)d.batch * L.cand_cap);
  L.cand_box = o; o = align256(o + sizeof(f32x4) * (size_t)d.batch * L.cand_cap);
  L.s_box = o; o = align256(o + sizeof(f32x4) * (size_t)d.batch * L.cap);
  L.s_score = o; o = align256(o + sizeof(float) * (size_t)d.batch * L.cap);
  L.s_idx = o; o = align256(o + sizeof(int) * (size_t)d.batch * L.cap);
  L.mask = o; o = align256(o + sizeof(u64) * (size_t)d.batch * L.mcap * L.Tm);
  L.maxc = o; o = align256(o + sizeof(float) * (size_t)d.batch);
  L.total = o;
  return ST_OK;
}

}  // namespace st

extern "C" size_t st_decode_nms_workspace_bytes(const StDecodeDesc* d) {
  st::DecodeLayout L;
  if (!d || st::decode_layout(*d, L) != ST_OK) return 0;
  return L.total;
}

extern "C" int st_decode_nms(const StDecodeDesc* d, const float* head_out_dev, void* workspace_dev,
                             size_t workspace_bytes, st_stream_t stream_, float* out_boxes_dev,
                             float* out_scores_dev, int64_t* out_labels_dev, int32_t* out_prior_idx_dev,
                             int32_t* out_count_dev) {
  using namespace st;
  if (!d) return set_error(ST_ERR_INVALID, "st_decode_nms: null desc");
  ST_REQUIRE(head_out_dev && workspace_dev && out_boxes_dev && out_scores_dev && out_labels_dev &&
                 out_prior_idx_dev && out_count_dev,
             "st_decode_nms: null pointer");
  DecodeLayout L;
  ST_CHECK(decode_layout(*d, L));
  if (workspace_bytes < L.total)
    return set_error(ST_ERR_WORKSPACE, "st_decode_nms: workspace %zu < required %zu", workspace_bytes, L.total);
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  char* ws = static_cast<char*>(workspace_dev);
  DecodeArgs a{};
  a.head = head_out_dev;
  a.num_levels = d->num_levels;
  int start = 0;
  for (int l = 0; l < d->num_levels; ++l) {
    a.lvl_h[l] = d->level_h[l]; a.lvl_w[l] = d->level_w[l]; a.lvl_stride[l] = d->level_stride[l];
    a.lvl_off[l] = d->level_offset[l];
    a.lvl_start[l] = start;
    start += d->level_h[l] * d->level_w[l];
  }
  a.lvl_start[d->num_levels] = start;
  a.P = L.P; a.cap = L.cap; a.mcap = L.mcap; a.Tm = L.Tm; a.batch = d->batch; a.nc = L.nc; a.cand_cap = L.cand_cap;
  a.hr = head_row_floats(L.nc); a.single_label = (d->single_label && L.nc > 1) ? 1 : 0;
  a.score_thr = d->score_thr; a.iou_thr = d->iou_thr; a.max_det = d->max_det;
  a.scale_x = d->scale_x; a.scale_y = d->scale_y; a.pad_left = d->pad_left; a.pad_top = d->pad_top;
  a.ori_w = d->ori_w; a.ori_h = d->ori_h;
  a.count = reinterpret_cast<int*>(ws + L.count);
  a.cand_key = reinterpret_cast<u64*>(ws + L.cand_key);
  a.cand_box = reinterpret_cast<f32x4*>(ws + L.cand_box);
  a.s_box = reinterpret_cast<f32x4*>(ws + L.s_box);
  a.s_score = reinterpret_cast<float*>(ws + L.s_score);
  a.s_idx = reinterpret_cast<int*>(ws + L.s_idx);
  a.mask = reinterpret_cast<u64*>(ws + L.mask);
  a.maxc = reinterpret_cast<float*>(ws + L.maxc);

  ST_CHECK_HIP(hipMemsetAsync(a.count, 0, sizeof(int) * d->batch, stream));
  hipLaunchKernelGGL(decode_filter_kernel, dim3((L.P + 255) / 256, d->batch), dim3(256), 0, stream, a);
  hipLaunchKernelGGL(rank_sort_kernel, dim3(std::min(64, (L.cand_cap + 255) / 256), d->batch), dim3(256), 0, stream, a);
  hipLaunchKernelGGL(nms_mask_kernel, dim3(128, d->batch), dim3(256), 0, stream, a);
  hipLaunchKernelGGL(nms_reduce_kernel, dim3(d->batch), dim3(64), 0, stream, a, out_boxes_dev, out_scores_dev,
                     reinterpret_cast<long long*>(out_labels_dev), out_prior_idx_dev, out_count_dev);
  ST_CHECK_HIP(hipGetLastError());
  return ST_OK;
}
