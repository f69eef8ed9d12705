// Small bandwidth-bound helpers of the detector: Focus space-to-depth packing and SPP pooling.
#include <cstdint>
#include <algorithm>
#include <cstdlib>

#include "st_common.h"

namespace st {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// NCHW [N][C][H][W] -> NHWC [N][H/2][W/2][4C], channel = q*C + c with q = TL, BL, TR, BR
// (mmdet Focus order; reference csp_darknet_disparity_v1.py:104-111).
// One thread per output pixel: 2x2 float2 reads per channel (8 B/lane, coalesced along x),
// 4C contiguous floats written.
template <int C>
__global__ __launch_bounds__(256) void focus_pack_kernel(const float* __restrict__ img, int N, int H,
                                                         int W, float* __restrict__ out) {
  const int Ho = H >> 1, Wo = W >> 1;
  const long long total = (long long)N * Ho * Wo;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    const int ox = (int)(idx % Wo);
    const long long t = idx / Wo;
    const int oy = (int)(t % Ho);
    const int n = (int)(t / Ho);
    float v[4 * C];
#pragma unroll
    for (int c = 0; c < C; ++c) {
      const float* base = img + (((size_t)n * C + c) * H + 2 * oy) * W + 2 * ox;
      const f32x2 top = *reinterpret_cast<const f32x2*>(base);
      const f32x2 bot = *reinterpret_cast<const f32x2*>(base + W);
      v[0 * C + c] = top[0];  // TL
      v[1 * C + c] = bot[0];  // BL
      v[2 * C + c] = top[1];  // TR
      v[3 * C + c] = bot[1];  // BR
    }
    float* o = out + (size_t)idx * (4 * C);
#pragma unroll
    for (int g = 0; g < C; ++g) {
      f32x4 w = {v[4 * g], v[4 * g + 1], v[4 * g + 2], v[4 * g + 3]};
      *reinterpret_cast<f32x4*>(o + 4 * g) = w;
    }
  }
}

// SPP pooling: max over 5x5 / 9x9 / 13x13 windows (stride 1, implicit -inf padding).
// One thread per (pixel, 4-channel group); windows are nested so one 13x13 sweep feeds all three.
__global__ __launch_bounds__(256) void spp_pool_kernel(const float* __restrict__ x, int x_ld, int x_off,
                                                       int N, int H, int W, int C,
                                                       float* __restrict__ out, int out_ld, int out_off,
                                                       int copy_x) {
  const int C4 = C >> 2;
  const long long total = (long long)N * H * W * C4;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    const int c4 = (int)(idx % C4);
    long long t = idx / C4;
    const int px = (int)(t % W);
    t /= W;
    const int py = (int)(t % H);
    const int n = (int)(t / H);
    const float ninf = -__builtin_inff();
    f32x4 m5 = {ninf, ninf, ninf, ninf}, m9 = m5, m13 = m5;
    for (int dy = -6; dy <= 6; ++dy) {
      const int yy = py + dy;
      if ((unsigned)yy >= (unsigned)H) continue;
      const int ady = dy < 0 ? -dy : dy;
      for (int dx = -6; dx <= 6; ++dx) {
        const int xx = px + dx;
        if ((unsigned)xx >= (unsigned)W) continue;
        const int adx = dx < 0 ? -dx : dx;
        const f32x4 v = *reinterpret_cast<const f32x4*>(
            x + ((size_t)(n * H + yy) * W + xx) * x_ld + x_off + 4 * c4);
        const int rad = ady > adx ? ady : adx;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          m13[e] = fmaxf(m13[e], v[e]);
          if (rad <= 4) m9[e] = fmaxf(m9[e], v[e]);
          if (rad <= 2) m5[e] = fmaxf(m5[e], v[e]);
        }
      }
    }
    float* o = out + ((size_t)(n * H + py) * W + px) * out_ld + out_off + 4 * c4;
    if (copy_x)
      *reinterpret_cast<f32x4*>(o) = *reinterpret_cast<const f32x4*>(
          x + ((size_t)(n * H + py) * W + px) * x_ld + x_off + 4 * c4);
    *reinterpret_cast<f32x4*>(o + C) = m5;
    *reinterpret_cast<f32x4*>(o + 2 * C) = m9;
    *reinterpret_cast<f32x4*>(o + 3 * C) = m13;
  }
}

// SPP via the max-pool cascade mp9 = mp5(mp5), mp13 = mp5(mp9) (exact for max), each mp5 separable
// (row max then column max), entirely in LDS: one workgroup owns a whole H x W map of CG channels.
// 6 LDS sweeps instead of a 169-tap window per output.
template <int CG>
__global__ __launch_bounds__(256) void spp_pool_lds_kernel(const float* __restrict__ x, int x_ld, int x_off, int H,
                                                           int W, int C, float* __restrict__ out, int out_ld,
                                                           int out_off, int copy_x) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int HW = H * W, n = blockIdx.y, c0 = blockIdx.x * CG;
  float* A = sm;
  float* B = sm + HW * CG;
  const int total = HW * CG;
  for (int e = threadIdx.x; e < total; e += blockDim.x) {
    const int pix = e / CG, c = e - pix * CG;
    const float v = x[((size_t)n * HW + pix) * x_ld + x_off + c0 + c];
    A[e] = v;
    if (copy_x) out[((size_t)n * HW + pix) * out_ld + out_off + c0 + c] = v;
  }
  __syncthreads();
  for (int level = 1; level <= 3; ++level) {
    for (int e = threadIdx.x; e < total; e += blockDim.x) {  // row max (along x), A -> B
      const int pix = e / CG, c = e - pix * CG, py = pix / W, px = pix - py * W;
      float m = A[e];
#pragma unroll
      for (int d = 1; d <= 2; ++d) {
        if (px - d >= 0) m = fmaxf(m, A[(pix - d) * CG + c]);
        if (px + d < W) m = fmaxf(m, A[(pix + d) * CG + c]);
      }
      B[e] = m;
    }
    __syncthreads();
    for (int e = threadIdx.x; e < total; e += blockDim.x) {  // column max (along y), B -> A, and store
      const int pix = e / CG, c = e - pix * CG, py = pix / W;
      float m = B[e];
#pragma unroll
      for (int d = 1; d <= 2; ++d) {
        if (py - d >= 0) m = fmaxf(m, B[(pix - d * W) * CG + c]);
        if (py + d < H) m = fmaxf(m, B[(pix + d * W) * CG + c]);
      }
      A[e] = m;
      out[((size_t)n * HW + pix) * out_ld + out_off + level * C + c0 + c] = m;
    }
    __syncthreads();
  }
}

int focus_pack_launch(const float* img, int N, int C, int H, int W, float* out, hipStream_t stream) {
  ST_REQUIRE(img && out, "focus_pack: null pointer");
  ST_REQUIRE(N > 0 && H > 0 && W > 0 && (H % 2) == 0 && (W % 2) == 0, "focus_pack: H, W must be even");
  const long long total = (long long)N * (H / 2) * (W / 2);
  const int blocks = (int)std::min<long long>((total + 255) / 256, 256 * 16);
  switch (C) {
    case 1: hipLaunchKernelGGL(focus_pack_kernel<1>, dim3(blocks), dim3(256), 0, stream, img, N, H, W, out); break;
    case 3: hipLaunchKernelGGL(focus_pack_kernel<3>, dim3(blocks), dim3(256), 0, stream, img, N, H, W, out); break;
    default: return set_error(ST_ERR_INVALID, "focus_pack: C must be 1 or 3 (got %d)", C);
  }
  ST_CHECK_HIP(hipGetLastError());
  return ST_OK;
}

int spp_pool_launch(const float* x, int x_ld, int x_off, int N, int H, int W, int C, float* out,
                    int out_ld, int out_off, hipStream_t stream) {
  ST_REQUIRE(x && out, "spp_pool: null pointer");
  ST_REQUIRE(C % 4 == 0 && x_ld % 4 == 0 && x_off % 4 == 0 && out_ld % 4 == 0 && out_off % 4 == 0,
             "spp_pool: channel counts/offsets must be multiples of 4");
  ST_REQUIRE(out_off + 4 * C <= out_ld && x_off + C <= x_ld, "spp_pool: slice exceeds ld");
  const int copy_x = !(x == out && x_ld == out_ld && x_off == out_off);
  // small maps (the stride-32 level of the path): whole-map LDS cascade, 8 channels per workgroup
  // channels per workgroup: the cascade is 7 dependent LDS sweeps with barriers, i.e. latency-bound, so small
  // problems want more, smaller workgroups.  Measured on 8 x 23x40x256: 8 channels (256 workgroups) 68 us,
  // 4 channels 45 us, 2 channels 55 us (8-byte global accesses).
  int cg = 8;
  if ((long long)(C / 8) * N < 1024) cg = 4;
  const size_t lds = (size_t)2 * H * W * cg * sizeof(float);
  if (C % cg == 0 && lds <= 150 * 1024 && N <= 65535) {
    using Kern = void (*)(const float*, int, int, int, int, int, float*, int, int, int);
    const Kern kern = cg == 8 ? spp_pool_lds_kernel<8> : cg == 4 ? spp_pool_lds_kernel<4> : spp_pool_lds_kernel<2>;
    static bool attr_set[3] = {false, false, false};
    const int ai = cg == 8 ? 0 : cg == 4 ? 1 : 2;
    if (!attr_set[ai]) {
      ST_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       150 * 1024));
      attr_set[ai] = true;
    }
    hipLaunchKernelGGL(kern, dim3(C / cg, N), dim3(256), lds, stream, x, x_ld, x_off, H, W, C, out, out_ld, out_off,
                       copy_x);
    ST_CHECK_HIP(hipGetLastError());
    return ST_OK;
  }
  const long long total = (long long)N * H * W * (C / 4);
  const int blocks = (int)std::min<long long>((total + 255) / 256, 256 * 16);
  hipLaunchKernelGGL(spp_pool_kernel, dim3(blocks), dim3(256), 0, stream, x, x_ld, x_off, N, H, W, C,
                     out, out_ld, out_off, copy_x);
  ST_CHECK_HIP(hipGetLastError());
  return ST_OK;
}

}  // namespace st

extern "C" int st_focus_pack(const float* img, int N, int C, int H, int W, float* out,
                             st_stream_t stream) {
  return st::focus_pack_launch(img, N, C, H, W, out, static_cast<hipStream_t>(stream));
}

extern "C" int st_spp_pool(const float* x, int x_ld, int x_off, int N, int H, int W, int C,
                           float* out, int out_ld, int out_off, st_stream_t stream) {
  return st::spp_pool_launch(x, x_ld, x_off, N, H, W, C, out, out_ld, out_off,
                             static_cast<hipStream_t>(stream));
}

// ---- raw input packing -------------------------------------------------------------
// Device-side restatement of the test-time input pipeline, so a frame crosses PCIe as uint8 pixels and
// uint16 disparity codes (4.7 MB) instead of three fp32 tensors (15.1 MB):
//   img  : uint8 (N,3,h,w) -> fp32 (N,3,H,W), bottom/right padding = img_pad (114, Pad_Disparity,
//          reference transforms_disparity.py:234-249; then stack_batch pads with 0 - same extent here)
//   disp : uint16 PNG code (N,h,w) -> fp32 px = code/16 with 65535 -> 0 (LoadDisparityFromFile
//          ._post_processing_v2, reference loading_disparity.py:129-134), replicated to 3 channels
//          (:85-86), padding 0; disp_mask = code < 65535 (:82).
namespace st {

__global__ __launch_bounds__(256) void pack_raw_inputs_kernel(const unsigned char* __restrict__ img,
                                                              const unsigned short* __restrict__ disp, int N, int h,
                                                              int w, int H, int W, float img_pad,
                                                              float* __restrict__ img_out,
                                                              float* __restrict__ disp_out,
                                                              float* __restrict__ mask_out) {
  const long long total = (long long)N * H * W;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    const int X = (int)(idx % W);
    const long long t = idx / W;
    const int Y = (int)(t % H);
    const int n = (int)(t / H);
    const bool inside = Y < h && X < w;
    const size_t plane = (size_t)H * W;
    if (img_out) {
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const float v = inside ? (float)img[(((size_t)n * 3 + c) * h + Y) * w + X] : img_pad;
        img_out[((size_t)n * 3 + c) * plane + (size_t)Y * W + X] = v;
      }
    }
    if (disp_out) {
      float d = 0.f, m = 0.f;
      if (inside) {
        const unsigned code = disp[((size_t)n * h + Y) * w + X];
        m = code < 65535u ? 1.f : 0.f;
        d = (code == 65535u ? 0.f : (float)code) / 16.0f;
      }
#pragma unroll
      for (int c = 0; c < 3; ++c) disp_out[((size_t)n * 3 + c) * plane + (size_t)Y * W + X] = d;
      if (mask_out) mask_out[(size_t)n * plane + (size_t)Y * W + X] = m;
    }
  }
}


// 4 pixels per thread (w, W multiples of 4, 4-/8-/16-byte aligned rows): one uchar4 / ushort4 load, float4 stores.
// Per image of 3 x 720 x 1280: 2.8 MB read, 11.3 MB written - an HBM stream (bytes, not flops).
__global__ __launch_bounds__(256) void pack_raw_inputs_vec4_kernel(const unsigned char* __restrict__ img,
                                                                   const unsigned short* __restrict__ disp, int N,
                                                                   int h, int w, int H, int W, float img_pad,
                                                                   float* __restrict__ img_out,
                                                                   float* __restrict__ disp_out,
                                                                   float* __restrict__ mask_out) {
  const int W4 = W >> 2;
  const long long total = (long long)N * H * W4;
  const size_t plane = (size_t)H * W;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    const int X = (int)(idx % W4) << 2;
    const long long t = idx / W4;
    const int Y = (int)(t % H);
    const int n = (int)(t / H);
    const bool inside = Y < h && X < w;   // w % 4 == 0: a group of 4 is inside or outside as a whole
    if (img_out) {
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        float4 v = make_float4(img_pad, img_pad, img_pad, img_pad);
        if (inside) {
          const uchar4 q = *reinterpret_cast<const uchar4*>(img + (((size_t)n * 3 + c) * h + Y) * w + X);
          v = make_float4((float)q.x, (float)q.y, (float)q.z, (float)q.w);
        }
        *reinterpret_cast<float4*>(img_out + ((size_t)n * 3 + c) * plane + (size_t)Y * W + X) = v;
      }
    }
    if (disp_out) {
      float4 d = make_float4(0.f, 0.f, 0.f, 0.f), m = d;
      if (inside) {
        const ushort4 q = *reinterpret_cast<const ushort4*>(disp + ((size_t)n * h + Y) * w + X);
        const unsigned code[4] = {q.x, q.y, q.z, q.w};
        float dv[4], mv[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          mv[k] = code[k] < 65535u ? 1.f : 0.f;
          dv[k] = (code[k] == 65535u ? 0.f : (float)code[k]) / 16.0f;
        }
        d = make_float4(dv[0], dv[1], dv[2], dv[3]);
        m = make_float4(mv[0], mv[1], mv[2], mv[3]);
      }
#pragma unroll
      for (int c = 0; c < 3; ++c)
        *reinterpret_cast<float4*>(disp_out + ((size_t)n * 3 + c) * plane + (size_t)Y * W + X) = d;
      if (mask_out) *reinterpret_cast<float4*>(mask_out + (size_t)n * plane + (size_t)Y * W + X) = m;
    }
  }
}

// The same conversion for a LIST of frames that live in separate allocations (what a dataloader hands over: one
// (3, h, w) uint8 tensor per frame): the frame pointers travel in the kernel arguments, blockIdx.y = frame, so no
// concatenated staging copy exists (torch.cat of 8 frames was 73 us per input and chunk in the test_step path).
constexpr int PACK_MAX_FRAMES = 32;
struct FramePtrs { const unsigned char* p[PACK_MAX_FRAMES]; };

__global__ __launch_bounds__(256) void pack_raw_frames_vec4_kernel(const FramePtrs fp, int h, int w, int H, int W,
                                                                   float img_pad, float* __restrict__ img_out) {
  const int n = blockIdx.y;
  const unsigned char* __restrict__ img = fp.p[n];   // uniform: a scalar load from the kernel arguments
  const int W4 = W >> 2;
  const int total = H * W4;
  const size_t plane = (size_t)H * W;
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
    const int X = (idx % W4) << 2;
    const int Y = idx / W4;
    const bool inside = Y < h && X < w;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      float4 v = make_float4(img_pad, img_pad, img_pad, img_pad);
      if (inside) {
        const uchar4 q = *reinterpret_cast<const uchar4*>(img + ((size_t)c * h + Y) * w + X);
        v = make_float4((float)q.x, (float)q.y, (float)q.z, (float)q.w);
      }
      *reinterpret_cast<float4*>(img_out + ((size_t)n * 3 + c) * plane + (size_t)Y * W + X) = v;
    }
  }
}


// ---- Resize_Disparity with a non-identity scale (reference mmtrack/datasets/transforms/transforms_disparity.py:23-137:
// image through mmcv.imrescale / imresize = cv2.resize INTER_LINEAR, disparity / mask / depth through INTER_NEAREST).
// cv2 is an un-vendored dependency; its 8-bit bilinear path is restated here from the published OpenCV source
// (imgproc/resize.cpp, [upstream-memory]): coordinates fx = (float)((dx + 0.5) * scale - 0.5), sx = floor(fx), the two
// tap weights rounded to 11-bit fixed point (cvRound: half to even), a horizontal pass in int32 and the vertical pass
// ((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2; an exact 2 x 2 decimation runs as the 2 x 2 box mean
// (a + b + c + d + 2) >> 2, which is what cv2 substitutes for INTER_LINEAR there.  Nearest: sx = min(floor(dx * scale), w - 1).
// One thread per output element; `ps` / `rs` / `pls` = pixel / row / plane strides in elements (planar CHW or interleaved HWC).
struct ResizeArgs {
  const void* in;
  void* out;
  int P, h, w, h2, w2;
  long long ips, irs, ipls, ops, ors, opls;
  double sx, sy;     // src / dst
};

__device__ __forceinline__ void rs_tap(int d, double scale, int n, int& s, int& a0, int& a1) {
  float f = (float)(((double)d + 0.5) * scale - 0.5);
  int i = (int)floorf(f);
  f -= (float)i;
  if (i < 0) { f = 0.f; i = 0; }
  if (i >= n - 1) { f = 0.f; i = n - 1; }
  s = i;
  a0 = __float2int_rn((1.f - f) * 2048.f);
  a1 = __float2int_rn(f * 2048.f);
}

__global__ __launch_bounds__(256) void resize_bilinear_u8_kernel(const ResizeArgs a) {
  const unsigned char* __restrict__ in = static_cast<const unsigned char*>(a.in);
  unsigned char* __restrict__ out = static_cast<unsigned char*>(a.out);
  const long long total = (long long)a.P * a.h2 * a.w2;
  const bool box = a.h == 2 * a.h2 && a.w == 2 * a.w2;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
    const int dx = (int)(idx % a.w2);
    const int dy = (int)((idx / a.w2) % a.h2);
    const int p = (int)(idx / ((long long)a.w2 * a.h2));
    const unsigned char* src = in + p * a.ipls;
    int v;
    if (box) {
      const unsigned char* r0 = src + (2 * dy) * a.irs + (2 * dx) * a.ips;
      const unsigned char* r1 = r0 + a.irs;
      v = ((int)r0[0] + (int)r0[a.ips] + (int)r1[0] + (int)r1[a.ips] + 2) >> 2;
    } else {
      int sx, ax0, ax1;
      rs_tap(dx, a.sx, a.w, sx, ax0, ax1);
      // rows: the fraction is NOT zeroed at the border, the row index is clamped (both taps then read the same row)
      float fy = (float)(((double)dy + 0.5) * a.sy - 0.5);
      const int sy = (int)floorf(fy);
      fy -= (float)sy;
      const int b0 = __float2int_rn((1.f - fy) * 2048.f), b1 = __float2int_rn(fy * 2048.f);
      const int y0 = min(max(sy, 0), a.h - 1), y1 = min(max(sy + 1, 0), a.h - 1);
      const int x1 = min(sx + 1, a.w - 1);
      const unsigned char* r0 = src + y0 * a.irs;
      const unsigned char* r1 = src + y1 * a.irs;
      const int S0 = (int)r0[sx * a.ips] * ax0 + (int)r0[x1 * a.ips] * ax1;
      const int S1 = (int)r1[sx * a.ips] * ax0 + (int)r1[x1 * a.ips] * ax1;
      v = (((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2;
      v = min(max(v, 0), 255);
    }
    out[p * a.opls + dy * a.ors + dx * a.ops] = (unsigned char)v;
  }
}

template <typename T>
__global__ __launch_bounds__(256) void resize_nearest_kernel(const ResizeArgs a) {
  const T* __restrict__ in = static_cast<const T*>(a.in);
  T* __restrict__ out = static_cast<T*>(a.out);
  const long long total = (long long)a.P * a.h2 * a.w2;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
    const int dx = (int)(idx % a.w2);
    const int dy = (int)((idx / a.w2) % a.h2);
    const int p = (int)(idx / ((long long)a.w2 * a.h2));
    const int sx = min((int)floor((double)dx * a.sx), a.w - 1);
    const int sy = min((int)floor((double)dy * a.sy), a.h - 1);
    out[p * a.opls + dy * a.ors + dx * a.ops] = in[p * a.ipls + sy * a.irs + sx * a.ips];
  }
}

}  // namespace st

extern "C" int st_pack_raw_frames(const unsigned char* const* frames_u8_dev_ptrs_host, int N, int h, int w, int H, int W,
                                  float img_pad, float* img_out_dev, st_stream_t stream) {
  using namespace st;
  ST_REQUIRE(frames_u8_dev_ptrs_host && img_out_dev && N > 0 && N <= PACK_MAX_FRAMES && h > 0 && w > 0 && H >= h && W >= w,
             "st_pack_raw_frames: bad argument (1..%d frames)", PACK_MAX_FRAMES);
  ST_REQUIRE(w % 4 == 0 && W % 4 == 0 && (reinterpret_cast<uintptr_t>(img_out_dev) & 15) == 0,
             "st_pack_raw_frames: widths must be multiples of 4 and the output 16-byte aligned");
  FramePtrs fp;
  for (int i = 0; i < PACK_MAX_FRAMES; ++i) fp.p[i] = i < N ? frames_u8_dev_ptrs_host[i] : nullptr;
  for (int i = 0; i < N; ++i)
    ST_REQUIRE(fp.p[i] && (reinterpret_cast<uintptr_t>(fp.p[i]) & 3) == 0, "st_pack_raw_frames: frame %d null or not 4-byte aligned", i);
  const int per = std::min((H * (W / 4) + 255) / 256, 512);
  hipLaunchKernelGGL(pack_raw_frames_vec4_kernel, dim3(per, N), dim3(256), 0, static_cast<hipStream_t>(stream), fp, h, w,
                     H, W, img_pad, img_out_dev);
  ST_CHECK_HIP(hipGetLastError());
  return ST_OK;
}

extern "C" int st_pack_raw_inputs(const unsigned char* img_u8_dev, const unsigned short* disp_u16_dev, int N, int h,
                                  int w, int H, int W, float img_pad, float* img_out_dev, float* disp_postp_out_dev,
                                  float* disp_mask_out_dev, st_stream_t stream) {
  using namespace st;
  ST_REQUIRE(N > 0 && h > 0 && w > 0 && H >= h && W >= w, "st_pack_raw_inputs: bad geometry");
  ST_REQUIRE((img_u8_dev && img_out_dev) || (disp_u16_dev && disp_postp_out_dev), "st_pack_raw_inputs: nothing to do");
  ST_REQUIRE(!img_out_dev || img_u8_dev, "st_pack_raw_inputs: img output without img input");
  ST_REQUIRE(!disp_postp_out_dev || disp_u16_dev, "st_pack_raw_inputs: disparity output without disparity input");
  const auto al = [](const void* p, size_t a) { return p == nullptr || (reinterpret_cast<uintptr_t>(p) % a) == 0; };
  if (w % 4 == 0 && W % 4 == 0 && al(img_u8_dev, 4) && al(disp_u16_dev, 8) && al(img_out_dev, 16) &&
      al(disp_postp_out_dev, 16) && al(disp_mask_out_dev, 16)) {
    const long long total4 = (long long)N * H * (W / 4);
    const int blocks4 = (int)std::min<long long>((total4 + 255) / 256, 256 * 16);
    hipLaunchKernelGGL(pack_raw_inputs_vec4_kernel, dim3(blocks4), dim3(256), 0, static_cast<hipStream_t>(stream),
                       img_u8_dev, disp_u16_dev, N, h, w, H, W, img_pad, img_out_dev, disp_postp_out_dev,
                       disp_mask_out_dev);
    ST_CHECK_HIP(hipGetLastError());
    return ST_OK;
  }
  const long long total = (long long)N * H * W;
  const int blocks = (int)std::min<long long>((total + 255) / 256, 256 * 16);
  hipLaunchKernelGGL(pack_raw_inputs_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), img_u8_dev,
                     disp_u16_dev, N, h, w, H, W, img_pad, img_out_dev, disp_postp_out_dev, disp_mask_out_dev);
  ST_CHECK_HIP(hipGetLastError());
  return ST_OK;
}

extern "C" int st_resize_planes(const void* in_dev, int P, int h, int w, int interleaved, void* out_dev, int h2, int w2,
                                int elem_bytes, int bilinear, st_stream_t stream) {
  using namespace st;
  ST_REQUIRE(in_dev && out_dev && P > 0 && P <= 16 && h > 0 && w > 0 && h2 > 0 && w2 > 0, "st_resize_planes: bad argument");
  ST_REQUIRE(elem_bytes == 1 || elem_bytes == 2 || elem_bytes == 4, "st_resize_planes: elem_bytes must be 1, 2 or 4");
  ST_REQUIRE(!bilinear || elem_bytes == 1, "st_resize_planes: bilinear is the 8-bit image path (cv2 INTER_LINEAR, uint8)");
  ST_REQUIRE((long long)P * h * w < (1ll << 31) && (long long)P * h2 * w2 < (1ll << 31), "st_resize_planes: image too large");
  ResizeArgs a;
  a.in = in_dev; a.out = out_dev; a.P = P; a.h = h; a.w = w; a.h2 = h2; a.w2 = w2;
  if (interleaved) {   // [h][w][P]
    a.ips = P; a.irs = (long long)w * P; a.ipls = 1; a.ops = P; a.ors = (long long)w2 * P; a.opls = 1;
  } else {             // [P][h][w]
    a.ips = 1; a.irs = w; a.ipls = (long long)h * w; a.ops = 1; a.ors = w2; a.opls = (long long)h2 * w2;
  }
  a.sx = (double)w / (double)w2;
  a.sy = (double)h / (double)h2;
  const long long total = (long long)P * h2 * w2;
  const int blocks = (int)std::min<long long>((total + 255) / 256, 256 * 32);
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (bilinear) hipLaunchKernelGGL(resize_bilinear_u8_kernel, dim3(blocks), dim3(256), 0, s, a);
  else if (elem_bytes == 1) hipLaunchKernelGGL(resize_nearest_kernel<unsigned char>, dim3(blocks), dim3(256), 0, s, a);
  else if (elem_bytes == 2) hipLaunchKernelGGL(resize_nearest_kernel<unsigned short>, dim3(blocks), dim3(256), 0, s, a);
  else hipLaunchKernelGGL(resize_nearest_kernel<unsigned int>, dim3(blocks), dim3(256), 0, s, a);
  ST_CHECK_HIP(hipGetLastError());
  return ST_OK;
}
