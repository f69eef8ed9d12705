// Per-box depth + depth-guided box scaling on the device.
//
// Restates OCSORT_Disparity.bbox_postp_depth / disp2depth / extract_depth
// (reference mmtrack/models/mot/ocsort_disparity.py:113-175) and scale_bbox
// (mmtrack/models/trackers/utils.py:58-73).  The reference copies the whole depth map to the
// host (3.77 MB + sync) and loops over boxes in numpy, twice per frame; here one workgroup per
// box streams its window straight from the disparity map (HBM/L2 -> registers), so nothing
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
// depths are positive, so the unsigned order equals the float order); the trimmed mean is
// accumulated in fp64 and rounded once (numpy uses fp32 pairwise summation: equal to ~1e-6
// relative, inside the 1e-3 float tolerance of the path).
#include <algorithm>

#include "st_common.h"

namespace st {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int py_slice_index(int i, int len) {
  if (i < 0) {
    i += len;
    if (i < 0) i = 0;
  } else if (i > len) {
    i = len;
  }
  return i;
}

struct BoxWin {
  int ys, ye, xs, xe;  // normalised slice of the box window
};

__device__ __forceinline__ float depth_at(const float* disp, int W, int y, int x, float bf, int is_depth) {
  const float v = disp[(size_t)y * W + x];
  return is_depth ? v : bf / (v + 1e-6f);
}

__device__ __forceinline__ bool depth_valid(float d) { return d < 150.0f && d > 0.0f; }

// value (bit pattern) at 0-based `rank` among the valid depths of the window; all threads call it
__device__ unsigned select_rank(const float* disp, int W, const BoxWin& w, float bf, int is_depth, int rank,
                                unsigned* hist /*[256]*/, unsigned* sh /*[2]*/) {
  const int cols = w.xe - w.xs, total = (w.ye - w.ys) * cols;
  unsigned prefix = 0, prefix_mask = 0;
  int r = rank;
  for (int shift = 24; shift >= 0; shift -= 8) {
    for (int b = threadIdx.x; b < 256; b += blockDim.x) hist[b] = 0;
    __syncthreads();
    for (int e = threadIdx.x; e < total; e += blockDim.x) {
      const int yy = w.ys + e / cols, xx = w.xs + e % cols;
      const float d = depth_at(disp, W, yy, xx, bf, is_depth);
      if (depth_valid(d)) {
        const unsigned u = __float_as_uint(d);
        if ((u & prefix_mask) == prefix) atomicAdd(&hist[(u >> shift) & 255u], 1u);
      }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      unsigned cum = 0, b = 0;
      for (; b < 256; ++b) {
        if (cum + hist[b] > (unsigned)r) break;
        cum += hist[b];
      }
      sh[0] = b;
      sh[1] = cum;
    }
    __syncthreads();
    prefix |= sh[0] << shift;
    prefix_mask |= 255u << shift;
    r -= (int)sh[1];
    __syncthreads();
  }
  return prefix;
}

__device__ float corner_mean(const float* disp, int H, int W, int r0, int r1, int c0, int c1, float bf,
                             int is_depth) {
  r0 = py_slice_index(r0, H); r1 = py_slice_index(r1, H);
  c0 = py_slice_index(c0, W); c1 = py_slice_index(c1, W);
  float s = 0.f;
  int cnt = 0;
  for (int y = r0; y < r1; ++y)
    for (int x = c0; x < c1; ++x) {
      s += depth_at(disp, W, y, x, bf, is_depth);
      ++cnt;
    }
  if (cnt == 0) return __builtin_nanf("");
  return s / (float)cnt;
}

__global__ __launch_bounds__(256) void box_depth_kernel(const float* __restrict__ disp_all, size_t img_pitch, int H,
                                                        int W, const float* __restrict__ boxes,
                                                        const int* __restrict__ counts, int max_det, float bf,
                                                        int is_depth, float* __restrict__ out_depth,
                                                        float* __restrict__ out_scale,
                                                        float* __restrict__ out_sboxes) {
  __shared__ unsigned hist[256];
  __shared__ unsigned sh[2];
  __shared__ int s_len;
  __shared__ double s_red[256];
  __shared__ int s_cnt[4][256];
  const int n = blockIdx.y, k = blockIdx.x;
  const int cnt_n = min(counts[n], max_det);
  if (k >= cnt_n) return;  // block-uniform
  const float* disp = disp_all + (size_t)n * img_pitch;
  const f32x4 bx = *reinterpret_cast<const f32x4*>(boxes + ((size_t)n * max_det + k) * 4);
  const int x1 = (int)bx[0], y1 = (int)bx[1], x2 = (int)bx[2], y2 = (int)bx[3];
  BoxWin w;
  w.ys = py_slice_index(y1, H); w.ye = py_slice_index(y2, H);
  w.xs = py_slice_index(x1, W); w.xe = py_slice_index(x2, W);
  if (w.ye < w.ys) w.ye = w.ys;
  if (w.xe < w.xs) w.xe = w.xs;
  const int cols = w.xe - w.xs, total = (w.ye - w.ys) * cols;

  // ---- pass 0: number of valid depths
  if (threadIdx.x == 0) s_len = 0;
  __syncthreads();
  int local = 0;
  for (int e = threadIdx.x; e < total; e += blockDim.x) {
    const float d = depth_at(disp, W, w.ys + e / cols, w.xs + e % cols, bf, is_depth);
    local += depth_valid(d);
  }
  if (local) atomicAdd(&s_len, local);
  __syncthreads();
  const int len = s_len;
  float dval, scale;
  if (len < 1 || (x2 - x1) > 800) {
    dval = -1.0f;
    scale = 1.0f;
  } else {
    const unsigned mid_bits = select_rank(disp, W, w, bf, is_depth, len / 2, hist, sh);
    const float d_mid = __uint_as_float(mid_bits);
    int cnt = 0;
    {
      const float v_tl = corner_mean(disp, H, W, y1, y1 + 2, x1, x1 + 2, bf, is_depth);
      const float v_tr = corner_mean(disp, H, W, y1, y1 + 2, x2 - 2, x2, bf, is_depth);
      const float v_bl = corner_mean(disp, H, W, y2 - 2, y2, x1, x1 + 2, bf, is_depth);
      const float v_br = corner_mean(disp, H, W, y2 - 2, y2, x2 - 2, x2, bf, is_depth);
      cnt = (v_tl > d_mid) + (v_tr > d_mid) + (v_bl > d_mid) + (v_br > d_mid);
    }
    const double frac = 1.0 - (double)cnt / 4.0;
    const double w_start = (frac < 0.4 ? frac : 0.4) * (double)len;
    const double w_end = w_start + 0.6 * (double)len;
    int a = (int)w_start, b = (int)w_end;
    if (b > len) b = len;
    if (a > len) a = len;
    if (b - a <= 0) {  // d_seg empty -> d_sorted[:-1]
      a = 0;
      b = len - 1;
    }
    if (b - a <= 0) {
      dval = __builtin_nanf("");
    } else {
      const unsigned va_bits = select_rank(disp, W, w, bf, is_depth, a, hist, sh);
      const unsigned vb_bits = select_rank(disp, W, w, bf, is_depth, b - 1, hist, sh);
      const float va = __uint_as_float(va_bits), vb = __uint_as_float(vb_bits);
      double sum = 0.0;
      int lt_a = 0, eq_a = 0, lt_b = 0, eq_b = 0;
      for (int e = threadIdx.x; e < total; e += blockDim.x) {
        const float d = depth_at(disp, W, w.ys + e / cols, w.xs + e % cols, bf, is_depth);
        if (depth_valid(d)) {
          lt_a += d < va; eq_a += d == va;
          lt_b += d < vb; eq_b += d == vb;
          if (d > va && d < vb) sum += (double)d;
        }
      }
      s_red[threadIdx.x] = sum;
      s_cnt[0][threadIdx.x] = lt_a; s_cnt[1][threadIdx.x] = eq_a;
      s_cnt[2][threadIdx.x] = lt_b; s_cnt[3][threadIdx.x] = eq_b;
      __syncthreads();
      for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) {
          s_red[threadIdx.x] += s_red[threadIdx.x + s];
#pragma unroll
          for (int q = 0; q < 4; ++q) s_cnt[q][threadIdx.x] += s_cnt[q][threadIdx.x + s];
        }
        __syncthreads();
      }
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

}  // namespace st

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
  hipLaunchKernelGGL(box_depth_kernel, dim3(max_det, N), dim3(256), 0, static_cast<hipStream_t>(stream_), disp_dev,
                     img_pitch, H, W, boxes_dev, counts_dev, max_det, bf, is_depth, out_depth_dev, out_scale_dev,
                     out_scaled_boxes_dev);
  ST_CHECK_HIP(hipGetLastError());
  return ST_OK;
}
