// Per-box depth + depth-guided box scaling on the device.
//
// Restates OCSORT_Disparity.bbox_postp_depth / disp2depth / extract_depth
// (reference mmtrack/models/mot/ocsort_disparity.py:113-175) and scale_bbox
// (mmtrack/models/trackers/utils.py:58-73).  The reference copies the whole depth map to the
// host (3.77 MB + sync) and loops over boxes in numpy, twice per frame; here one workgroup per
// box streams its window straight from the disparity map (HBM/L2 -> registers -> LDS), so nothing
// leaves the GPU.
//
// numpy semantics that are reproduced on purpose (SURVEY.md §7 "hard parts"):
//   * box.astype(np.int) truncates toward zero; slice bounds follow Python slice normalisation
//     (negative indices wrap, out-of-range clamps, start >= stop => empty)
//   * valid = 0 < depth < 150 on depth = (baseline*focal) / (disp + 1e-6) (fp32, IEEE divide)
//   * len == 0 or (x2 - x1) > 800                      => depth -1, scale 1
//   * median = sorted[len // 2]; 4 corner 2x2 means on the RAW depth (empty slice => NaN,
//     NaN > median is False)
//   * w_start = min(1 - cnt/4, 0.4) * len ; w_end = w_start + 0.6 * len  (Python doubles);
//     seg = sorted[int(w_start):int(w_end)], empty => sorted[:-1], still empty => NaN
//   * scale = max(min(d*d, 3.), 1.) with Python min/max NaN behaviour (NaN propagates)
// Order statistics come from an 8-bit-per-pass radix select on the float bit patterns (valid
// depths are positive, so the unsigned order equals the float order); the two segment bounds are
// selected together (two histograms per pass); the trimmed mean is accumulated in fp64 and rounded
// once (numpy uses fp32 pairwise summation: equal to ~1e-6 relative, inside the 1e-3 float
// tolerance of the path).  Windows of up to BD_CACHE pixels (every realistic drone box) are cached in
// LDS by the counting pass, so the following passes never touch memory again.
#include <algorithm>

#include "st_common.h"

namespace st {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// One SMALL workgroup per box: drone boxes are tens of pixels wide (8-60 px objects; the bench
// workload's ~2900 boxes average 300 pixels, the largest is 2400), so a box is a few elements per lane and the kernel's
// time is the barriers and LDS round trips of its passes, not the data: what matters is how many boxes a CU works on at
// once.  512-thread workgroups (round 2): 192 us.  128 threads, 16 KiB cache, 20 barriers per box: 95 us.  Round 3:
// first radix histogram counted by the counting pass, corner loads issued up front, 14 barriers, and 17 KiB of LDS
// per box (9 boxes per CU): 58 us.  Windows above BD_CACHE pixels stream from L2 in every pass (rare, still exact).
constexpr int BD_THREADS = 128;
constexpr int BD_CACHE = 2560;   // floats of LDS window cache (10 KiB): 50 x 50 pixels

__device__ __forceinline__ int py_slice_index(int i, int len) {
  if (i < 0) {
    i += len;
    if (i < 0) i = 0;
  } else if (i > len) {
    i = len;
  }
  return i;
}

__device__ __forceinline__ float to_depth(float v, float bf, int is_depth) {
  return is_depth ? v : bf / (v + 1e-6f);
}

__device__ __forceinline__ bool depth_valid(float d) { return d < 150.0f && d > 0.0f; }

struct Window {
  const float* disp;   // image base
  const float* cache;  // LDS copy of the window's depths (window order) or nullptr
  int W, ys, xs, ye, xe, cols, total;
  float bf;
  int is_depth;
  __device__ __forceinline__ float get(int e) const {
    if (cache) return cache[e];
    return to_depth(disp[(size_t)(ys + e / cols) * W + xs + e % cols], bf, is_depth);
  }
  // visit every depth of the window once: LDS cache (linear) or, for large windows, one wave per row
  // with lanes along x (coalesced, no index division)
  template <class F>
  __device__ __forceinline__ void for_each(F f) const {
    if (cache) {
      for (int e = threadIdx.x; e < total; e += blockDim.x) f(cache[e]);
    } else {
      const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nw = blockDim.x >> 6;
      // 2 rows x 4 column chunks = 8 independent loads in flight per lane (the loop is latency-bound)
      for (int y = ys + 2 * wave; y < ye; y += 2 * nw) {
        const float* row0 = disp + (size_t)y * W;
        const bool two = y + 1 < ye;
        const float* row1 = two ? row0 + W : row0;
        for (int x = xs + lane; x < xe; x += 256) {
          float v[8];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int xx = min(x + 64 * u, xe - 1);
            v[u] = row0[xx];
            v[4 + u] = row1[xx];
          }
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            if (x + 64 * u < xe) {
              f(to_depth(v[u], bf, is_depth));
              if (two) f(to_depth(v[4 + u], bf, is_depth));
            }
          }
        }
      }
    }
  }
};

// values (bit patterns) at NR 0-based ranks among the valid depths, selected together (NR histograms per
// pass, 4 passes of 8 bits); all threads call it.
// Precondition (set up by the kernel's counting pass, one barrier before the call): hist[0] = histogram of the TOP
// byte of every valid depth - all ranks share the empty prefix, so the first radix pass needs no sweep of its own.
// A pass is  count | barrier | prefix sums -> sh | barrier | use sh, re-zero the histograms | barrier.
constexpr int BD_NR = 7;
__device__ void select_ranks(const Window& w, const int* ranks, unsigned (*hist)[256], unsigned* sh, unsigned* out) {
  unsigned prefix[BD_NR], mask = 0;
  int r[BD_NR];
#pragma unroll
  for (int q = 0; q < BD_NR; ++q) { prefix[q] = 0; r[q] = ranks[q]; }
  for (int shift = 24; shift >= 0; shift -= 8) {
    unsigned (*hc)[256] = hist;
    if (shift != 24) {
      w.for_each([&](float d) {
        if (depth_valid(d)) {
          const unsigned u = __float_as_uint(d), um = u & mask, bin = (u >> shift) & 255u;
#pragma unroll
          for (int q = 0; q < BD_NR; ++q) {
            // ranks that still share a prefix share a histogram: count once, in the first of them
            bool first = true;
#pragma unroll
            for (int q2 = 0; q2 < q; ++q2) first = first && (prefix[q2] != prefix[q]);
            if (first && um == prefix[q]) atomicAdd(&hc[q][bin], 1u);
          }
        }
      });
      __syncthreads();
    }
    // one wave per rank: 64-lane prefix sum over the 256 bins (4 bins per lane), the lane whose bins
    // straddle the rank reports (bin, count below it)
    const int wv0 = threadIdx.x >> 6, ln = threadIdx.x & 63, nwv = blockDim.x >> 6;
    for (int wv = wv0; wv < BD_NR; wv += nwv) {   // the ranks are dealt over the waves of the workgroup
      int rq = 0;
      unsigned pq = 0;
#pragma unroll
      for (int q = 0; q < BD_NR; ++q)
        if (q == wv) { rq = r[q]; pq = prefix[q]; }
      int src = wv;  // the histogram this rank's prefix was counted in
#pragma unroll
      for (int q2 = BD_NR - 1; q2 >= 0; --q2)
        if (q2 < wv && prefix[q2] == pq) src = q2;
      unsigned h[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) h[j] = hc[src][4 * ln + j];
      const unsigned tot = h[0] + h[1] + h[2] + h[3];
      unsigned incl = tot;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const unsigned t = __shfl_up(incl, off);
        if (ln >= off) incl += t;
      }
      unsigned cum = incl - tot;
      if (cum <= (unsigned)rq && (unsigned)rq < incl) {
        int b = 4 * ln;
#pragma unroll
        for (int j = 0; j < 3; ++j)
          if (cum + h[j] <= (unsigned)rq && b == 4 * ln + j) { cum += h[j]; ++b; }
        sh[2 * wv] = b;
        sh[2 * wv + 1] = cum;
      }
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < BD_NR; ++q) {
      prefix[q] |= sh[2 * q] << shift;
      r[q] -= (int)sh[2 * q + 1];
    }
    mask |= 255u << shift;
    if (shift != 0) {   // every reader of the histograms is behind the barrier above
      for (int b = threadIdx.x; b < BD_NR * 256; b += blockDim.x) hist[b >> 8][b & 255] = 0;
      __syncthreads();
    }
  }
#pragma unroll
  for (int q = 0; q < BD_NR; ++q) out[q] = prefix[q];
}

__device__ float corner_mean(const float* disp, int H, int W, int r0, int r1, int c0, int c1, float bf,
                             int is_depth) {
  r0 = py_slice_index(r0, H); r1 = py_slice_index(r1, H);
  c0 = py_slice_index(c0, W); c1 = py_slice_index(c1, W);
  float s = 0.f;
  int cnt = 0;
  for (int y = r0; y < r1; ++y)
    for (int x = c0; x < c1; ++x) {
      s += to_depth(disp[(size_t)y * W + x], bf, is_depth);
      ++cnt;
    }
  if (cnt == 0) return __builtin_nanf("");
  return s / (float)cnt;
}

__global__ __launch_bounds__(BD_THREADS) void box_depth_kernel(const float* __restrict__ disp_all, size_t img_pitch, int H,
                                                        int W, const float* __restrict__ boxes,
                                                        const int* __restrict__ counts, int max_det, float bf,
                                                        int is_depth, float* __restrict__ out_depth,
                                                        float* __restrict__ out_scale,
                                                        float* __restrict__ out_sboxes) {
  __shared__ float cache[BD_CACHE];
  __shared__ unsigned hist[BD_NR][256];
  __shared__ unsigned sh[2 * BD_NR];
  __shared__ int s_len;
  __shared__ double s_red[BD_THREADS / 64];
  __shared__ int s_cnt[4][BD_THREADS / 64];
  const int n = blockIdx.y, k = blockIdx.x;
  const int cnt_n = min(counts[n], max_det);
  if (k >= cnt_n) {  // block-uniform: rows past the count are defined (zero), never stale
    if (threadIdx.x == 0) {
      const size_t o = (size_t)n * max_det + k;
      out_depth[o] = 0.f;
      out_scale[o] = 0.f;
      const f32x4 z = {0.f, 0.f, 0.f, 0.f};
      *reinterpret_cast<f32x4*>(out_sboxes + o * 4) = z;
    }
    return;
  }
  const float* disp = disp_all + (size_t)n * img_pitch;
  const f32x4 bx = *reinterpret_cast<const f32x4*>(boxes + ((size_t)n * max_det + k) * 4);
  const int x1 = (int)bx[0], y1 = (int)bx[1], x2 = (int)bx[2], y2 = (int)bx[3];
  Window w;
  w.disp = disp; w.W = W; w.bf = bf; w.is_depth = is_depth;
  w.ys = py_slice_index(y1, H);
  w.xs = py_slice_index(x1, W);
  w.ye = max(py_slice_index(y2, H), w.ys);
  w.xe = max(py_slice_index(x2, W), w.xs);
  w.cols = w.xe - w.xs;
  w.total = (w.ye - w.ys) * w.cols;
  if ((x2 - x1) > 800) w.total = 0;  // the reference discards such boxes (`w > 800`): skip every pass
  const bool use_cache = w.total <= BD_CACHE;
  w.cache = nullptr;

  // the 4 corner means only depend on the box: their (dependent, uncached) loads are issued here and land while the
  // passes below run; they are compared with the median after the selection
  const float v_tl = corner_mean(disp, H, W, y1, y1 + 2, x1, x1 + 2, bf, is_depth);
  const float v_tr = corner_mean(disp, H, W, y1, y1 + 2, x2 - 2, x2, bf, is_depth);
  const float v_bl = corner_mean(disp, H, W, y2 - 2, y2, x1, x1 + 2, bf, is_depth);
  const float v_br = corner_mean(disp, H, W, y2 - 2, y2, x2 - 2, x2, bf, is_depth);

  // ---- pass 0: number of valid depths, histogram of their top byte (= the first radix pass of select_ranks), and the
  // LDS window cache
  if (threadIdx.x == 0) s_len = 0;
  for (int b = threadIdx.x; b < 256; b += blockDim.x) hist[0][b] = 0;
  __syncthreads();
  int local = 0;
  auto tally = [&](float d) {
    if (depth_valid(d)) {
      ++local;
      atomicAdd(&hist[0][__float_as_uint(d) >> 24], 1u);
    }
  };
  if (use_cache) {
    for (int e = threadIdx.x; e < w.total; e += blockDim.x) {
      const float d = w.get(e);
      cache[e] = d;
      tally(d);
    }
  } else {
    w.for_each(tally);
  }
  if (local) atomicAdd(&s_len, local);
  __syncthreads();
  if (use_cache) w.cache = cache;
  const int len = s_len;
  float dval, scale;
  if (len < 1 || (x2 - x1) > 800) {
    dval = -1.0f;
    scale = 1.0f;
  } else {
    // the segment [a, b) depends on the median only through cnt in {0..4}, i.e. through 3 possible
    // fractions (0.4, 0.25, 0): select the median and all 6 candidate bounds in the same 4 passes
    int cand_a[3], cand_b[3];
    const double fr[3] = {0.4, 0.25, 0.0};
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const double w_start = fr[c] * (double)len;
      const double w_end = w_start + 0.6 * (double)len;
      int a = (int)w_start, b = (int)w_end;
      if (b > len) b = len;
      if (a > len) a = len;
      if (b - a <= 0) {  // d_seg empty -> d_sorted[:-1]
        a = 0;
        b = len - 1;
      }
      cand_a[c] = a;
      cand_b[c] = b;
    }
    int ranks[BD_NR];
    unsigned bits[BD_NR];
    ranks[0] = len / 2;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      ranks[1 + 2 * c] = min(cand_a[c], len - 1);
      ranks[2 + 2 * c] = max(min(cand_b[c] - 1, len - 1), 0);
    }
    select_ranks(w, ranks, hist, sh, bits);
    const float d_mid = __uint_as_float(bits[0]);
    const int cnt = (v_tl > d_mid) + (v_tr > d_mid) + (v_bl > d_mid) + (v_br > d_mid);
    // frac = min(1 - cnt/4, 0.4): cnt <= 2 -> 0.4, cnt == 3 -> 0.25, cnt == 4 -> 0
    const int csel = cnt <= 2 ? 0 : (cnt == 3 ? 1 : 2);
    const int a = cand_a[csel], b = cand_b[csel];
    if (b - a <= 0) {
      dval = __builtin_nanf("");
    } else {
      const unsigned va_bits = bits[1 + 2 * csel], vb_bits = bits[2 + 2 * csel];
      const float va = __uint_as_float(va_bits), vb = __uint_as_float(vb_bits);
      double sum = 0.0;
      int lt_a = 0, eq_a = 0, lt_b = 0, eq_b = 0;
      w.for_each([&](float d) {
        if (depth_valid(d)) {
          lt_a += d < va; eq_a += d == va;
          lt_b += d < vb; eq_b += d == vb;
          if (d > va && d < vb) sum += (double)d;
        }
      });
      // wave reduction by shuffles, then one LDS slot per wave (fp64 sum: the order is fixed, so deterministic)
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) {
        sum += __shfl_xor(sum, off);
        lt_a += __shfl_xor(lt_a, off); eq_a += __shfl_xor(eq_a, off);
        lt_b += __shfl_xor(lt_b, off); eq_b += __shfl_xor(eq_b, off);
      }
      const int wvr = threadIdx.x >> 6;
      if ((threadIdx.x & 63) == 0) {
        s_red[wvr] = sum;
        s_cnt[0][wvr] = lt_a; s_cnt[1][wvr] = eq_a; s_cnt[2][wvr] = lt_b; s_cnt[3][wvr] = eq_b;
      }
      __syncthreads();
      if (threadIdx.x == 0) {
        for (int v = 1; v < (int)(blockDim.x >> 6); ++v) {
          s_red[0] += s_red[v];
#pragma unroll
          for (int q = 0; q < 4; ++q) s_cnt[q][0] += s_cnt[q][v];
        }
      }
      __syncthreads();
      const int n_lt_a = s_cnt[0][0], n_eq_a = s_cnt[1][0], n_lt_b = s_cnt[2][0], n_eq_b = s_cnt[3][0];
      double tot = s_red[0];
      const int ca = min(n_lt_a + n_eq_a, b) - max(n_lt_a, a);
      tot += (double)va * (double)ca;
      if (vb_bits != va_bits) {
        const int cb = min(n_lt_b + n_eq_b, b) - max(n_lt_b, a);
        tot += (double)vb * (double)cb;
      }
      dval = (float)(tot / (double)(b - a));
    }
    const float dd = dval * dval;
    scale = (3.0f < dd) ? 3.0f : dd;       // Python min(dd, 3.): NaN stays
    scale = (1.0f > scale) ? 1.0f : scale;  // Python max(scale, 1.)
  }
  if (threadIdx.x == 0) {
    const size_t o = (size_t)n * max_det + k;
    out_depth[o] = dval;
    out_scale[o] = scale;
    // scale_bbox (trackers/utils.py:58-73), fp32 tensor ops
    const float cx = (bx[0] + bx[2]) / 2.0f, cy = (bx[1] + bx[3]) / 2.0f;
    const float bw = (bx[2] - bx[0]) * scale, bh = (bx[3] - bx[1]) * scale;
    f32x4 ob = {cx - bw / 2.0f, cy - bh / 2.0f, cx + bw / 2.0f, cy + bh / 2.0f};
    *reinterpret_cast<f32x4*>(out_sboxes + o * 4) = ob;
  }
}

// Frame records for the all-gather / the one D2H copy per chunk: (N, M + 1, cols) fp32.  Row 0 = header [true count,
// M, valid-frame flag, 0 ...]; rows 1..M = x1,y1,x2,y2 (mode 1: the depth-scaled box), score, label, depth, scale
// [mode 2: + scaled box (4) + kept prior index].  One launch instead of ~10 cat / fill / cast launches.
__global__ __launch_bounds__(256) void pack_records_kernel(const float* __restrict__ boxes,
                                                           const float* __restrict__ scores,
                                                           const long long* __restrict__ labels,
                                                           const float* __restrict__ depth,
                                                           const float* __restrict__ scales,
                                                           const float* __restrict__ sboxes,
                                                           const int* __restrict__ prior,
                                                           const int* __restrict__ counts, int N, int M, int mode,
                                                           int n_real, float* __restrict__ out) {
  const int cols = mode == 2 ? 13 : 8;
  const long long total = (long long)N * (M + 1);
  for (long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x; r < total;
       r += (long long)gridDim.x * blockDim.x) {
    const int n = (int)(r / (M + 1)), k = (int)(r % (M + 1));
    float* o = out + r * cols;
    if (k == 0) {
      const bool real = n < n_real;
      o[0] = real ? (float)counts[n] : 0.f;
      o[1] = real ? (float)M : 0.f;
      o[2] = real ? 1.f : 0.f;
      for (int c = 3; c < cols; ++c) o[c] = 0.f;
      continue;
    }
    const size_t i = (size_t)n * M + (k - 1);
    const f32x4 b = *reinterpret_cast<const f32x4*>(boxes + i * 4);
    const f32x4 sb = *reinterpret_cast<const f32x4*>(sboxes + i * 4);
    const f32x4 first = mode == 1 ? sb : b;
    o[0] = first[0]; o[1] = first[1]; o[2] = first[2]; o[3] = first[3];
    o[4] = scores[i];
    o[5] = (float)labels[i];
    o[6] = depth[i];
    o[7] = scales[i];
    if (mode == 2) {
      o[8] = sb[0]; o[9] = sb[1]; o[10] = sb[2]; o[11] = sb[3];
      o[12] = (float)prior[i];   // prior index < 2^24: exact in fp32
    }
  }
}

}  // namespace st

extern "C" int st_pack_records(const float* boxes_dev, const float* scores_dev, const int64_t* labels_dev,
                               const float* depth_dev, const float* scales_dev, const float* scaled_boxes_dev,
                               const int32_t* prior_idx_dev, const int32_t* counts_dev, int N, int max_det, int mode,
                               int n_real, float* out_records_dev, st_stream_t stream_) {
  using namespace st;
  ST_REQUIRE(boxes_dev && scores_dev && labels_dev && depth_dev && scales_dev && scaled_boxes_dev && counts_dev &&
                 out_records_dev, "st_pack_records: null pointer");
  ST_REQUIRE(mode >= 0 && mode <= 2 && (mode != 2 || prior_idx_dev), "st_pack_records: bad mode");
  ST_REQUIRE(N > 0 && max_det > 0 && n_real >= 0, "st_pack_records: bad geometry");
  const long long total = (long long)N * (max_det + 1);
  const int blocks = (int)std::min<long long>((total + 255) / 256, 2048);
  hipLaunchKernelGGL(pack_records_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream_), boxes_dev,
                     scores_dev, reinterpret_cast<const long long*>(labels_dev), depth_dev, scales_dev,
                     scaled_boxes_dev, prior_idx_dev, counts_dev, N, max_det, mode, n_real, out_records_dev);
  ST_CHECK_HIP(hipGetLastError());
  return ST_OK;
}

extern "C" size_t st_box_depth_workspace_bytes(int, int, int, int) { return 0; }

extern "C" int st_box_depth(const float* disp_dev, size_t img_pitch, int N, int H, int W, const float* boxes_dev,
                            const int32_t* counts_dev, int max_det, float baseline, float focal, void*, size_t,
                            st_stream_t stream_, float* out_depth_dev, float* out_scale_dev,
                            float* out_scaled_boxes_dev) {
  using namespace st;
  ST_REQUIRE(disp_dev && boxes_dev && counts_dev && out_depth_dev && out_scale_dev && out_scaled_boxes_dev,
             "st_box_depth: null pointer");
  ST_REQUIRE(N > 0 && H > 0 && W > 0 && max_det > 0 && N <= 65535, "st_box_depth: bad geometry");
  ST_REQUIRE(img_pitch >= (size_t)H * W, "st_box_depth: img_pitch smaller than one image");
  // baseline < 0 selects "input is already a depth map" (reference passes gt depth_postp that way,
  // ocsort_disparity.py:120-122); bf = baseline * focal as a Python float product rounded to fp32
  const int is_depth = baseline < 0.f;
  const float bf = (float)((double)baseline * (double)focal);
  hipLaunchKernelGGL(box_depth_kernel, dim3(max_det, N), dim3(BD_THREADS), 0, static_cast<hipStream_t>(stream_), disp_dev,
                     img_pitch, H, W, boxes_dev, counts_dev, max_det, bf, is_depth, out_depth_dev, out_scale_dev,
                     out_scaled_boxes_dev);
  ST_CHECK_HIP(hipGetLastError());
  return ST_OK;
}
