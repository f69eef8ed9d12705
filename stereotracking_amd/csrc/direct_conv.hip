// Direct 3x3 / stride-1 convolution for narrow layers (Cin, Cout in {32, 48, 64}) on gfx950, tile variant 42.
//
// Targets the narrow high-resolution 3x3 ConvModules of the path: the cost-volume aggregation convs (48 -> 48,
// stereotracking_amd/stereo.py) and the DarknetBottleneck conv2 of the stage-1 CSP layers (32 -> 32 + identity;
// mmdet CSPLayer as built at reference mmtrack/models/backbones/csp_darknet_disparity_v1.py:145-153).  In the
// generic implicit-GEMM kernel these layers (a) pad Cout = 48 to 64 (32-wide MFMA tiles: 25 % wasted matrix work)
// and (b) re-gather every input pixel nine times through im2col address generation.  Here:
//   * `v_mfma_f32_16x16x4_f32` (same flop/cycle as 32x32x2): cout blocks of 16, so Cout = 48 is exact;
//   * one workgroup = 4 x 32 output pixels; its (4+2) x (32+2) x Cin input window is brought into LDS ONCE by
//     LDS-DMA (hardware zero fill = the conv padding), pixel stride Cin + 4 floats (odd float4 count: 16 lanes x
//     ds_read_b128 hit 64 distinct banks); the nine taps are nine address offsets into that window;
//   * weights stream per tap ([Cout][Cin], 9 KB for 48 x 48) through two LDS buffers, tap t+1 in flight during
//     the MFMAs of tap t (they come from L2: every workgroup reads the same 83 KB);
//   * swapped operands (A = weights, B = pixels): a lane owns 4 consecutive couts of one pixel => 16-byte NHWC
//     stores / residual loads through range-checked buffer descriptors, no branches in the epilogue.
// Same arithmetic as st_conv2d_nhwc on these shapes (shares the packed weights [CoutPad][Kpad], k = (tap, ci)).
#include <algorithm>
#include <cstdint>
#include <set>

#include "st_common.h"

namespace st {
namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int DC_TH = 4, DC_TW = 32;           // output tile: one row of 32 pixels per wave
constexpr int DC_WH = DC_TH + 2, DC_WW = DC_TW + 2;

struct DcArgs {
  const float* in;
  const float* wgt;
  const float* bias;
  float* out;
  const float* res;
  int N, H, W, in_ld, in_off, Cout, Kpad;
  int out_ld, out_off, res_ld, res_off;
  float post_scale;
  int act;
  int tiles_x, tiles_y;
  unsigned in_bytes, wgt_bytes, out_bytes, res_bytes;
};

__device__ __forceinline__ float dc_silu(float v) { return v * __builtin_amdgcn_rcpf(1.0f + __expf(-v)); }

template <int CIN, int COUT, bool RES>
__global__ __launch_bounds__(256, 2) void direct_conv3x3_kernel(const DcArgs p) {
  constexpr int PQ = CIN / 4 + 1;              // 16-byte slots per pixel (last one = padding)
  constexpr int PS = 4 * PQ;                   // pixel stride in floats
  constexpr int WIN_SLOTS = DC_WH * DC_WW * PQ;
  constexpr int WIN_DMA = (WIN_SLOTS + 255) / 256;
  constexpr int WIN_FLOATS = WIN_DMA * 256 * 4;
  constexpr int WT_SLOTS = COUT * PQ;          // one tap: [COUT][CIN + 4]
  constexpr int WT_DMA = (WT_SLOTS + 255) / 256;
  constexpr int WT_FLOATS = WT_DMA * 256 * 4;
  constexpr int CB = COUT / 16, G = CIN / 16;
  extern __shared__ float4 dc_smem4[];
  float* win = reinterpret_cast<float*>(dc_smem4);
  float* wbuf = win + WIN_FLOATS;              // two tap buffers
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i16 = lane & 15, kq = lane >> 4;
  int b = blockIdx.x;
  const int tx = b % p.tiles_x; b /= p.tiles_x;
  const int ty = b % p.tiles_y;
  const int n = b / p.tiles_y;
  const int y0 = ty * DC_TH, x0 = tx * DC_TW;

#if defined(__HIP_DEVICE_COMPILE__)  // device-only builtins; the host pass only needs the kernel stub
  const __amdgpu_buffer_rsrc_t irsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.in), 0, (int)p.in_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t wrsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.wgt), 0, (int)p.wgt_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, (int)p.out_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rrsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(RES ? p.res : p.in), 0, (int)(RES ? p.res_bytes : 0u), 0x00020000);

  // ---- input window: slot e <-> (row, col, quad); rows / cols outside the image and the pad quad read zeros
#pragma unroll
  for (int j = 0; j < WIN_DMA; ++j) {
    const int e = tid + 256 * j;
    const int pix = e / PQ, q = e - pix * PQ;
    const int row = pix / DC_WW, col = pix - row * DC_WW;
    const int gy = y0 - 1 + row, gx = x0 - 1 + col;
    const bool ok = e < WIN_SLOTS && q < CIN / 4 && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
    const unsigned off = ok ? (unsigned)((((n * p.H + gy) * p.W + gx) * p.in_ld + p.in_off + 4 * q) * 4) : 0x80000000u;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(
        irsrc, (__attribute__((address_space(3))) void*)(win + (j * 256 + wave * 64) * 4), 16, off, 0, 0, 0);
  }
  // ---- one tap of weights: slot e <-> (cout, quad); source row co of the packed matrix, columns tap*CIN + 4q
  auto wt_dma = [&](int tap, int buf) {
#pragma unroll
    for (int j = 0; j < WT_DMA; ++j) {
      const int e = tid + 256 * j;
      const int co = e / PQ, q = e - co * PQ;
      const bool ok = e < WT_SLOTS && q < CIN / 4;
      const unsigned off = ok ? (unsigned)((co * p.Kpad + tap * CIN + 4 * q) * 4) : 0x80000000u;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(
          wrsrc, (__attribute__((address_space(3))) void*)(wbuf + buf * WT_FLOATS + (j * 256 + wave * 64) * 4), 16,
          off, 0, 0, 0);
    }
  };
  wt_dma(0, 0);
  // residual of this lane's outputs: issued now, consumed in the epilogue (its HBM latency rides under the MFMAs)
  const int oy = y0 + wave;
  f32x4 rv[RES ? CB : 1][2];
  if (RES) {
#pragma unroll
    for (int pb = 0; pb < 2; ++pb) {
      const int ox = x0 + pb * 16 + i16;
      const bool ok = oy < p.H && ox < p.W;
      const int m = (n * p.H + oy) * p.W + ox;
#pragma unroll
      for (int cb = 0; cb < CB; ++cb) {
        const unsigned roff = ok ? (unsigned)((m * p.res_ld + p.res_off + cb * 16 + 4 * kq) * 4) : 0x80000000u;
        rv[RES ? cb : 0][pb] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rrsrc, roff, 0, 0));
      }
    }
  }

  f32x4 acc[CB][2];
#pragma unroll
  for (int cb = 0; cb < CB; ++cb)
#pragma unroll
    for (int pb = 0; pb < 2; ++pb) acc[cb][pb] = f32x4{0.f, 0.f, 0.f, 0.f};
  // lane = (pixel i16 of a 16-pixel block, k quarter kq): B operand = channels 16g + 4kq + step of its pixel,
  // A operand = the same channels of cout row i16 (K permutation shared by both operands)
  const float* xlane = win + ((wave * DC_WW) + i16) * PS + 4 * kq;   // tap (0,0) of pixel block 0
  const float* wlane = wbuf + i16 * PS + 4 * kq;
  __syncthreads();   // vmcnt(0) + barrier: window and tap 0 landed

#pragma unroll
  for (int tap = 0; tap < 9; ++tap) {
    const int buf = tap & 1;
    if (tap + 1 < 9) wt_dma(tap + 1, buf ^ 1);   // lands during this tap's MFMAs
    const int ky = tap / 3, kx = tap - 3 * ky;
    const float* xt = xlane + (ky * DC_WW + kx) * PS;
    const float* wt = wlane + buf * WT_FLOATS;
    f32x4 xf[2][2], wf[2][CB];
    auto read_group = [&](int g, int set) {
#pragma unroll
      for (int pb = 0; pb < 2; ++pb) xf[set][pb] = *reinterpret_cast<const f32x4*>(xt + pb * 16 * PS + 16 * g);
#pragma unroll
      for (int cb = 0; cb < CB; ++cb) wf[set][cb] = *reinterpret_cast<const f32x4*>(wt + cb * 16 * PS + 16 * g);
    };
    read_group(0, 0);
#pragma unroll
    for (int g = 0; g < G; ++g) {
      const int set = g & 1;
      if (g + 1 < G) read_group(g + 1, set ^ 1);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int cb = 0; cb < CB; ++cb)
#pragma unroll
          for (int pb = 0; pb < 2; ++pb)
            acc[cb][pb] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[set][cb][s], xf[set][pb][s], acc[cb][pb], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (tap + 1 < 9) __syncthreads();   // next tap landed, everyone done with this tap's buffer
  }

  // ---- epilogue: D[co][pixel]: lane = pixel (pb*16 + i16), couts cb*16 + 4*kq + {0..3}
#pragma unroll
  for (int pb = 0; pb < 2; ++pb) {
    const int ox = x0 + pb * 16 + i16;
    const bool ok = oy < p.H && ox < p.W;
    const int m = (n * p.H + oy) * p.W + ox;
#pragma unroll
    for (int cb = 0; cb < CB; ++cb) {
      const int co = cb * 16 + 4 * kq;
      const f32x4 bq = *reinterpret_cast<const f32x4*>(p.bias + co);
      f32x4 v;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float x = acc[cb][pb][e] + bq[e];
        if (p.act) x = dc_silu(x);
        if (RES) x = (x + rv[RES ? cb : 0][pb][e]) * p.post_scale;
        v[e] = x;
      }
      const unsigned off = ok ? (unsigned)((m * p.out_ld + p.out_off + co) * 4) : 0x80000000u;
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), orsrc, off, 0, 0);
    }
  }
#else
  (void)win; (void)wbuf; (void)i16; (void)kq; (void)n; (void)y0; (void)x0;
#endif
}

}  // namespace

// Shapes this kernel takes: 3x3 / stride 1 / pad 1, Cin and Cout in {32, 48, 64} (Cout exact, no split / upsample),
// 16-byte aligned channel slices, every tensor below 2 GiB.
bool dc_conv_applicable(const StConvDesc& d) {
  if (d.KH != 3 || d.KW != 3 || d.stride != 1 || d.pad != 1 || d.up_dev || d.out2_dev) return false;
  if (d.Cin != 32 && d.Cin != 48 && d.Cin != 64) return false;
  if (d.Cout != 32 && d.Cout != 48 && d.Cout != 64) return false;
  if ((d.in_ld | d.in_off | d.out1_ld | d.out1_off) & 3) return false;
  if ((reinterpret_cast<uintptr_t>(d.in_dev) | reinterpret_cast<uintptr_t>(d.out1_dev)) & 15) return false;
  if (d.res_dev && (((d.res_ld | d.res_off) & 3) || (reinterpret_cast<uintptr_t>(d.res_dev) & 15))) return false;
  const long long M = (long long)d.N * d.Hi * d.Wi, lim = 1ll << 31;
  if (M * d.in_ld * 4 >= lim || M * d.out1_ld * 4 >= lim) return false;
  if (d.res_dev && M * d.res_ld * 4 >= lim) return false;
  return true;
}

int dc_conv_launch(const StConvDesc& d, hipStream_t stream) {
  ST_REQUIRE(dc_conv_applicable(d), "direct conv: shape not supported by the direct 3x3 kernel");
  ST_REQUIRE(d.in_dev && d.wgt_dev && d.bias_dev && d.out1_dev, "direct conv: null pointer");
  ST_REQUIRE(d.in_off + d.Cin <= d.in_ld && d.out1_off + d.Cout <= d.out1_ld, "direct conv: channel slice exceeds ld");
  if (d.res_dev) ST_REQUIRE(d.res_off + d.Cout <= d.res_ld, "direct conv: res slice exceeds res_ld");
  const long long M = (long long)d.N * d.Hi * d.Wi;
  DcArgs a;
  a.in = d.in_dev; a.wgt = d.wgt_dev; a.bias = d.bias_dev; a.out = d.out1_dev; a.res = d.res_dev;
  a.N = d.N; a.H = d.Hi; a.W = d.Wi; a.in_ld = d.in_ld; a.in_off = d.in_off; a.Cout = d.Cout;
  a.Kpad = round_up(9 * d.Cin, 32);
  a.out_ld = d.out1_ld; a.out_off = d.out1_off; a.res_ld = d.res_ld; a.res_off = d.res_off;
  a.post_scale = d.res_dev ? d.post_scale : 1.0f;
  a.act = d.act;
  a.tiles_x = ceil_div(d.Wi, DC_TW); a.tiles_y = ceil_div(d.Hi, DC_TH);
  a.in_bytes = (unsigned)(M * d.in_ld * 4);
  a.wgt_bytes = (unsigned)((long long)round_up(d.Cout, 32) * a.Kpad * 4);
  a.out_bytes = (unsigned)(M * d.out1_ld * 4);
  a.res_bytes = d.res_dev ? (unsigned)(M * d.res_ld * 4) : 0u;
  const long long blocks = (long long)d.N * a.tiles_x * a.tiles_y;
  ST_REQUIRE(blocks < (1ll << 31), "direct conv: grid too large");
  const int pq = d.Cin / 4 + 1;
  const size_t lds = (size_t)(((DC_WH * DC_WW * pq + 255) / 256) * 256 * 4 + 2 * ((d.Cout * pq + 255) / 256) * 256 * 4) *
                     sizeof(float);
  using Kern = void (*)(const DcArgs);
  Kern kern = nullptr;
#define DC_PICK(CI, CO)                                                                          \
  if (d.Cin == CI && d.Cout == CO)                                                               \
    kern = d.res_dev ? static_cast<Kern>(direct_conv3x3_kernel<CI, CO, true>)                    \
                     : static_cast<Kern>(direct_conv3x3_kernel<CI, CO, false>);
  DC_PICK(32, 32) DC_PICK(32, 48) DC_PICK(32, 64) DC_PICK(48, 32) DC_PICK(48, 48) DC_PICK(48, 64)
  DC_PICK(64, 32) DC_PICK(64, 48) DC_PICK(64, 64)
#undef DC_PICK
  ST_REQUIRE(kern != nullptr, "direct conv: no kernel instance");
  static std::set<Kern> attr_done;   // one process per GPU: set once per kernel instance
  if (!attr_done.count(kern)) {
    ST_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int)lds));
    attr_done.insert(kern);
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(256), lds, stream, a);
  ST_CHECK_HIP(hipGetLastError());
  return ST_OK;
}

}  // namespace st
