// Fused head of a stage-1 CSP branch on gfx950:
//     3x3 / stride-2 ConvModule (32 -> 64)  ->  CSPLayer main_conv | short_conv (1x1, 64 -> 32 | 32)
//                                            ->  DarknetBottleneck conv1 (1x1, 32 -> 32) on the main half
// i.e. `stage1.0` + the first three convolutions of `stage1.1` of both branches of the two-branch backbone
// (reference mmtrack/models/backbones/csp_darknet_disparity_v1.py:113-153: ConvModule(c1, c2, 3, stride 2) followed by
// mmdet CSPLayer(c2, c2, n, add_identity)), at the one place of the network where they are HBM-bound: 184x320 pixels
// x 16 images, 32-64 channels (16 flop/B for the 1x1 convs).  Unfused this is three launches that write and re-read
// the 64-channel stride-2 output and the main half (135 MB per image through HBM); fused, the 64-channel tensor never
// exists: a workgroup computes it for a 2 x 32 pixel tile in MFMA accumulators and feeds it straight into the two
// 1x1 convs (the swapped-operand accumulator layout D[cout][pixel] IS the B-operand layout of the next MFMA, the
// trick of pointwise_conv.hip's CHAIN mode).  Outputs: main (32 ch, the bottleneck's residual), short (32 ch, into
// the CSP concat buffer), conv1(main) (32 ch, the input of the 3x3 bottleneck conv).
//
// Structure (direct_conv.hip's, at stride 2): 4 waves = 2 rows x 2 blocks of 16 pixels; the (2*2+1) x (2*32+1) x 32
// input window goes to LDS once by LDS-DMA (hardware zero fill = conv padding), pixel stride 36 floats; per-tap
// weights [64][32] triple-buffered through LDS (tap t + 2 in flight during tap t); `v_mfma_f32_16x16x4_f32`, A = weights, B = pixels.  The 1x1 weights
// (16 KB + 4 KB) are pre-packed in fragment order and loaded from L2 into registers at kernel start, so the chained
// GEMMs run on registers only.  Same arithmetic as the three separate launches up to fp32 summation order.
#include <algorithm>
#include <cstdint>
#include <cstdlib>

#include "st_common.h"

namespace st {
namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int FF_CIN = 32, FF_C2 = 64, FF_MID = 32;
constexpr int FF_TH = 2, FF_TW = 32;                          // output tile
constexpr int FF_WH = 2 * FF_TH + 1, FF_WW = 2 * FF_TW + 1;   // input window 5 x 65
constexpr int FF_PQ = FF_CIN / 4 + 1, FF_PS = 4 * FF_PQ;      // 9 slots / 36 floats per pixel
constexpr int FF_WIN_SLOTS = FF_WH * FF_WW * FF_PQ;
constexpr int FF_WIN_DMA = (FF_WIN_SLOTS + 255) / 256;
constexpr int FF_WIN_FLOATS = FF_WIN_DMA * 256 * 4;
constexpr int FF_WT_SLOTS = FF_C2 * FF_PQ;                    // one tap: [64][32 + 4]
constexpr int FF_WT_DMA = (FF_WT_SLOTS + 255) / 256;
constexpr int FF_WT_FLOATS = FF_WT_DMA * 256 * 4;
constexpr int FF_NBUF = 3;                                   // tap-weight buffers: tap t + 2 is in flight during tap t
constexpr int FF_LDS_FLOATS = FF_WIN_FLOATS + FF_NBUF * FF_WT_FLOATS;

struct FrontArgs {
  const float* in;
  const float* wgt_a;      // [64][288] packed 3x3 weights (K = (tap, ci))
  const float* bias_a;
  const float* frag_ms;    // [cb2 4][cb 4][lane 64][4]
  const float* bias_ms;
  const float* frag_c1;    // [cb3 2][cb2 2][lane 64][4]
  const float* bias_c1;
  float* out_main;
  float* out_short;
  float* out_tmp;
  int N, Hi, Wi, Ho, Wo, in_ld, in_off;
  int main_ld, main_off, short_ld, short_off, tmp_ld, tmp_off;
  int tiles_x, tiles_y;
  unsigned in_bytes, wgt_bytes, main_bytes, short_bytes, tmp_bytes;
  int abl;   // tools-only (ST_ABLATION) timing experiments, wrong results: 1 no per-tap weight DMA, 2 no per-tap barrier,
             // 4 no chained GEMMs, 8 no window DMA, 16 no stores
};

__device__ __forceinline__ float ff_silu(float v) { return v * __builtin_amdgcn_rcpf(1.0f + __expf(-v)); }

__global__ __launch_bounds__(256, 2) void front_s2_csp_kernel(const FrontArgs p) {
  extern __shared__ float4 ff_smem4[];
  float* win = reinterpret_cast<float*>(ff_smem4);
  float* wbuf = win + FF_WIN_FLOATS;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i16 = lane & 15, kq = lane >> 4;
  const int row = wave >> 1, pb = wave & 1;
  int b = blockIdx.x;
  const int tx = b % p.tiles_x; b /= p.tiles_x;
  const int ty = b % p.tiles_y;
  const int n = b / p.tiles_y;
  const int oy0 = ty * FF_TH, ox0 = tx * FF_TW;

#if defined(__HIP_DEVICE_COMPILE__)  // device-only builtins; the host pass only needs the kernel stub
  const __amdgpu_buffer_rsrc_t irsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.in), 0, (int)p.in_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t wrsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.wgt_a), 0, (int)p.wgt_bytes, 0x00020000);

  // ---- input window: slot e <-> (row, col, quad); outside the image and the pad quad read zeros.  Within a window row
  // the even columns come first, then the odd ones: the 16 pixels of a stride-2 B fragment are then 36 floats apart
  // (conflict-free ds_read_b128) instead of 72 (2-way conflicts).
#pragma unroll
  for (int j = 0; j < FF_WIN_DMA; ++j) {
#ifdef ST_ABLATION
    if (p.abl & 8) break;
#endif
    const int e = tid + 256 * j;
    const int pix = e / FF_PQ, q = e - pix * FF_PQ;
    const int r = pix / FF_WW, cc = pix - r * FF_WW;
    const int c = cc < FF_TW + 1 ? 2 * cc : 2 * (cc - FF_TW - 1) + 1;   // row = [even columns | odd columns]
    const int gy = 2 * oy0 - 1 + r, gx = 2 * ox0 - 1 + c;
    const bool ok = e < FF_WIN_SLOTS && q < FF_CIN / 4 && gy >= 0 && gy < p.Hi && gx >= 0 && gx < p.Wi;
    const unsigned off = ok ? (unsigned)((((n * p.Hi + gy) * p.Wi + gx) * p.in_ld + p.in_off + 4 * q) * 4) : 0x80000000u;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(
        irsrc, (__attribute__((address_space(3))) void*)(win + (j * 256 + wave * 64) * 4), 16, off, 0, 0, 0);
  }
  auto wt_dma = [&](int tap, int buf) {
#pragma unroll
    for (int j = 0; j < FF_WT_DMA; ++j) {
      const int e = tid + 256 * j;
      const int co = e / FF_PQ, q = e - co * FF_PQ;
      const bool ok = e < FF_WT_SLOTS && q < FF_CIN / 4;
      const unsigned off = ok ? (unsigned)((co * (9 * FF_CIN) + tap * FF_CIN + 4 * q) * 4) : 0x80000000u;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(
          wrsrc, (__attribute__((address_space(3))) void*)(wbuf + buf * FF_WT_FLOATS + (j * 256 + wave * 64) * 4), 16,
          off, 0, 0, 0);
    }
  };
  wt_dma(0, 0);
  wt_dma(1, 1);

  // ---- 1x1 weights in fragment order: 16 + 4 coalesced 1 KB loads from L2, issued now, consumed after the 3x3 loop
  f32x4 fms[4][4], fc1[2][2];
#pragma unroll
  for (int c2 = 0; c2 < 4; ++c2)
#pragma unroll
    for (int c = 0; c < 4; ++c)
      fms[c2][c] = *reinterpret_cast<const f32x4*>(p.frag_ms + ((c2 * 4 + c) * 64 + lane) * 4);
#pragma unroll
  for (int c3 = 0; c3 < 2; ++c3)
#pragma unroll
    for (int c2 = 0; c2 < 2; ++c2)
      fc1[c3][c2] = *reinterpret_cast<const f32x4*>(p.frag_c1 + ((c3 * 2 + c2) * 64 + lane) * 4);

  // biases of the three stages in accumulator layout (couts cb*16 + 4kq + e), loaded up front for the same reason
  f32x4 ba[4], bm[4], bc[2];
#pragma unroll
  for (int cb = 0; cb < 4; ++cb) {
    ba[cb] = *reinterpret_cast<const f32x4*>(p.bias_a + cb * 16 + 4 * kq);
    bm[cb] = *reinterpret_cast<const f32x4*>(p.bias_ms + cb * 16 + 4 * kq);
  }
#pragma unroll
  for (int cb = 0; cb < 2; ++cb) bc[cb] = *reinterpret_cast<const f32x4*>(p.bias_c1 + cb * 16 + 4 * kq);

  f32x4 acc[4];
#pragma unroll
  for (int cb = 0; cb < 4; ++cb) acc[cb] = f32x4{0.f, 0.f, 0.f, 0.f};
  // lane = (pixel i16 of this wave's 16-pixel block, k quarter kq)
  const float* xlane = win + ((2 * row) * FF_WW + pb * 16 + i16) * FF_PS + 4 * kq;
  const float* wlane = wbuf + i16 * FF_PS + 4 * kq;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // window, tap 0 (and the fragment loads) landed
#pragma unroll
  for (int cb = 0; cb < 4; ++cb) asm volatile("" : "+v"(ba[cb]), "+v"(bm[cb]));   // keep the loads up here
#pragma unroll
  for (int cb = 0; cb < 2; ++cb) asm volatile("" : "+v"(bc[cb]));
  __syncthreads();

#pragma unroll
  for (int tap = 0; tap < 9; ++tap) {
    const int buf = tap % FF_NBUF;
#ifdef ST_ABLATION
    if (!(p.abl & 1))
#endif
    if (tap + 2 < 9) wt_dma(tap + 2, (tap + 2) % FF_NBUF);   // two taps (64 MFMAs per wave) to land
    const int ky = tap / 3, kx = tap - 3 * ky;
    const float* xt = xlane + (ky * FF_WW + (kx == 1 ? FF_TW + 1 : kx >> 1)) * FF_PS;   // column 2 px + kx of the window
    const float* wt = wlane + buf * FF_WT_FLOATS;
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      const f32x4 xf = *reinterpret_cast<const f32x4*>(xt + 16 * g);
      f32x4 wf[4];
#pragma unroll
      for (int cb = 0; cb < 4; ++cb) wf[cb] = *reinterpret_cast<const f32x4*>(wt + cb * 16 * FF_PS + 16 * g);
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int cb = 0; cb < 4; ++cb)
          acc[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[cb][s], xf[s], acc[cb], 0, 0, 0);
    }
#ifdef ST_ABLATION
    if (!(p.abl & 2))
#endif
    if (tap + 1 < 9) {
      // this wave's share of tap + 1 landed (the FF_WT_DMA loads of tap + 2 may still be in flight)
      if (tap + 2 < 9) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(FF_WT_DMA) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    }
  }

#ifdef ST_ABLATION
  if (p.abl & 4) {
    float v = 0.f;
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) v += acc[cb][0] + acc[cb][1] + acc[cb][2] + acc[cb][3] + fms[cb][cb][0] + fc1[cb & 1][cb >> 1][0];
    if (v == 12345.678f) p.out_main[tid] = v;
    return;
  }
#endif
  // ---- stage A epilogue in registers: lane holds couts cb*16 + 4kq + e of its pixel = the B operand of the 1x1 GEMM
  f32x4 va[4];
#pragma unroll
  for (int cb = 0; cb < 4; ++cb) {
#pragma unroll
    for (int e = 0; e < 4; ++e) va[cb][e] = ff_silu(acc[cb][e] + ba[cb][e]);
  }
  // ---- main | short = SiLU(W_ms . va + b)   (K = 64: 4 cout blocks of stage A x 4 steps)
  f32x4 am[4];
#pragma unroll
  for (int c2 = 0; c2 < 4; ++c2) am[c2] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int c2 = 0; c2 < 4; ++c2)
        am[c2] = __builtin_amdgcn_mfma_f32_16x16x4f32(fms[c2][c][s], va[c][s], am[c2], 0, 0, 0);
  f32x4 vm[4];
#pragma unroll
  for (int c2 = 0; c2 < 4; ++c2) {
#pragma unroll
    for (int e = 0; e < 4; ++e) vm[c2][e] = ff_silu(am[c2][e] + bm[c2][e]);
  }
  // ---- conv1(main) = SiLU(W_c1 . vm[0..1] + b)   (K = 32)
  f32x4 ac[2];
#pragma unroll
  for (int c3 = 0; c3 < 2; ++c3) ac[c3] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int c2 = 0; c2 < 2; ++c2)
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int c3 = 0; c3 < 2; ++c3)
        ac[c3] = __builtin_amdgcn_mfma_f32_16x16x4f32(fc1[c3][c2][s], vm[c2][s], ac[c3], 0, 0, 0);

#ifdef ST_ABLATION
  if (p.abl & 16) {
    float v = 0.f;
#pragma unroll
    for (int c2 = 0; c2 < 4; ++c2) v += vm[c2][0] + vm[c2][1] + vm[c2][2] + vm[c2][3] + ac[c2 & 1][c2];
    if (v == 12345.678f) p.out_main[tid] = v;
    return;
  }
#endif
  // ---- stores: 16 bytes per (pixel, 4 couts); range-checked descriptors, no branches
  const __amdgpu_buffer_rsrc_t mrsrc = __builtin_amdgcn_make_buffer_rsrc(p.out_main, 0, (int)p.main_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t srsrc = __builtin_amdgcn_make_buffer_rsrc(p.out_short, 0, (int)p.short_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t trsrc = __builtin_amdgcn_make_buffer_rsrc(p.out_tmp, 0, (int)p.tmp_bytes, 0x00020000);
  const int oy = oy0 + row, ox = ox0 + pb * 16 + i16;
  const bool ok = oy < p.Ho && ox < p.Wo;
  const int m = (n * p.Ho + oy) * p.Wo + ox;
#pragma unroll
  for (int c2 = 0; c2 < 2; ++c2) {
    const unsigned om = ok ? (unsigned)((m * p.main_ld + p.main_off + c2 * 16 + 4 * kq) * 4) : 0x80000000u;
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, vm[c2]), mrsrc, om, 0, 0);
    const unsigned os = ok ? (unsigned)((m * p.short_ld + p.short_off + c2 * 16 + 4 * kq) * 4) : 0x80000000u;
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, vm[2 + c2]), srsrc, os, 0, 0);
    f32x4 vt;
#pragma unroll
    for (int e = 0; e < 4; ++e) vt[e] = ff_silu(ac[c2][e] + bc[c2][e]);
    const unsigned ot = ok ? (unsigned)((m * p.tmp_ld + p.tmp_off + c2 * 16 + 4 * kq) * 4) : 0x80000000u;
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, vt), trsrc, ot, 0, 0);
  }
#else
  (void)win; (void)wbuf; (void)i16; (void)kq; (void)row; (void)pb; (void)n; (void)oy0; (void)ox0;
#endif
}

}  // namespace

// Fragment-ordered copy of a packed 1x1 weight matrix [Cout][Kpad] (Cout, Cin multiples of 16) for the chained GEMMs:
// out[((c2 * (Cin/16) + c) * 64 + lane) * 4 + e] = W[c2*16 + (lane & 15)][c*16 + 4*(lane >> 4) + e]
size_t front_frag_floats(int Cout, int Cin) { return (size_t)Cout * Cin; }
int front_pack_frags(const float* packed, int Cout, int Cin, float* out) {
  ST_REQUIRE(packed && out && Cout % 16 == 0 && Cin % 16 == 0, "front_pack_frags: channel counts must be multiples of 16");
  const int Kpad = round_up(Cin, 32);
  for (int c2 = 0; c2 < Cout / 16; ++c2)
    for (int c = 0; c < Cin / 16; ++c)
      for (int l = 0; l < 64; ++l)
        for (int e = 0; e < 4; ++e)
          out[(((size_t)c2 * (Cin / 16) + c) * 64 + l) * 4 + e] =
              packed[(size_t)(c2 * 16 + (l & 15)) * Kpad + c * 16 + 4 * (l >> 4) + e];
  return ST_OK;
}

// a = the 3x3 / stride-2 ConvModule (32 -> 64; its own output tensor is never written, out1_dev is ignored),
// ms = main_conv | short_conv on a's output (64 -> 32 | 32, split store), c1 = bottleneck conv1 on the main half.
bool front_fused_applicable(const StConvDesc& a, const StConvDesc& ms, const StConvDesc& c1) {
  const auto plain = [](const StConvDesc& d) {
    return d.act == 1 && !d.res_dev && !d.up_dev && d.post_scale == 1.f;
  };
  if (!(plain(a) && plain(ms) && plain(c1))) return false;
  if (!(a.KH == 3 && a.KW == 3 && a.stride == 2 && a.pad == 1 && a.Cin == FF_CIN && a.Cout == FF_C2)) return false;
  if (!(ms.KH == 1 && ms.KW == 1 && ms.stride == 1 && ms.pad == 0 && ms.Cin == FF_C2 && ms.Cout == 2 * FF_MID &&
        ms.split == FF_MID && ms.out1_dev && ms.out2_dev)) return false;
  if (!(c1.KH == 1 && c1.KW == 1 && c1.stride == 1 && c1.pad == 0 && c1.Cin == FF_MID && c1.Cout == FF_MID &&
        c1.out1_dev && (c1.split == 0 || c1.split >= FF_MID || !c1.out2_dev))) return false;
  if (!a.in_dev || !a.wgt_dev || !a.bias_dev || !ms.bias_dev || !c1.bias_dev) return false;
  if ((a.in_ld | a.in_off | ms.out1_ld | ms.out1_off | ms.out2_ld | ms.out2_off | c1.out1_ld | c1.out1_off) & 3) return false;
  if ((reinterpret_cast<uintptr_t>(a.in_dev) | reinterpret_cast<uintptr_t>(ms.out1_dev) |
       reinterpret_cast<uintptr_t>(ms.out2_dev) | reinterpret_cast<uintptr_t>(c1.out1_dev)) & 15) return false;
  const int Ho = (a.Hi - 1) / 2 + 1, Wo = (a.Wi - 1) / 2 + 1;
  if (ms.N != a.N || ms.Hi != Ho || ms.Wi != Wo || c1.N != a.N || c1.Hi != Ho || c1.Wi != Wo) return false;
  const long long Mi = (long long)a.N * a.Hi * a.Wi, Mo = (long long)a.N * Ho * Wo, lim = 1ll << 31;
  return Mi * a.in_ld * 4 < lim && Mo * std::max(std::max(ms.out1_ld, ms.out2_ld), c1.out1_ld) * 4 < lim &&
         a.in_off + FF_CIN <= a.in_ld && ms.out1_off + FF_MID <= ms.out1_ld && ms.out2_off + FF_MID <= ms.out2_ld &&
         c1.out1_off + FF_MID <= c1.out1_ld;
}

int front_fused_launch(const StConvDesc& da, const StConvDesc& dms, const StConvDesc& dc1, const float* frag_ms_dev,
                       const float* frag_c1_dev, hipStream_t stream) {
  ST_REQUIRE(frag_ms_dev && frag_c1_dev, "fused front: null fragment weights");
  ST_REQUIRE(front_fused_applicable(da, dms, dc1),
             "fused front: needs conv3x3/s2 32->64 -> 1x1 64->32|32 (split store) -> 1x1 32->32, SiLU, no residual, "
             "16-byte aligned tensors, channel strides multiples of 4");
  const int Ho = (da.Hi - 1) / 2 + 1, Wo = (da.Wi - 1) / 2 + 1;
  const long long Mi = (long long)da.N * da.Hi * da.Wi, Mo = (long long)da.N * Ho * Wo;
  FrontArgs a;
  a.in = da.in_dev; a.wgt_a = da.wgt_dev; a.bias_a = da.bias_dev; a.frag_ms = frag_ms_dev; a.bias_ms = dms.bias_dev;
  a.frag_c1 = frag_c1_dev; a.bias_c1 = dc1.bias_dev;
  a.out_main = dms.out1_dev; a.out_short = dms.out2_dev; a.out_tmp = dc1.out1_dev;
  a.N = da.N; a.Hi = da.Hi; a.Wi = da.Wi; a.Ho = Ho; a.Wo = Wo; a.in_ld = da.in_ld; a.in_off = da.in_off;
  a.main_ld = dms.out1_ld; a.main_off = dms.out1_off; a.short_ld = dms.out2_ld; a.short_off = dms.out2_off;
  a.tmp_ld = dc1.out1_ld; a.tmp_off = dc1.out1_off;
  a.tiles_x = ceil_div(Wo, FF_TW); a.tiles_y = ceil_div(Ho, FF_TH);
  a.in_bytes = (unsigned)(Mi * da.in_ld * 4);
  a.wgt_bytes = (unsigned)(FF_C2 * 9 * FF_CIN * 4);
  a.main_bytes = (unsigned)(Mo * dms.out1_ld * 4);
  a.short_bytes = (unsigned)(Mo * dms.out2_ld * 4);
  a.tmp_bytes = (unsigned)(Mo * dc1.out1_ld * 4);
  a.abl = 0;
#ifdef ST_ABLATION
  if (const char* e = getenv("ST_FF_ABL")) a.abl = atoi(e);
#endif
  const long long blocks = (long long)da.N * a.tiles_x * a.tiles_y;
  ST_REQUIRE(blocks < (1ll << 31), "fused front: grid too large");
  constexpr int lds = FF_LDS_FLOATS * (int)sizeof(float);
  static int lds_set = 0;
  ST_ENSURE_DYNAMIC_LDS(front_s2_csp_kernel, lds, lds_set);
  hipLaunchKernelGGL(front_s2_csp_kernel, dim3((unsigned)blocks), dim3(256), lds, stream, a);
  ST_CHECK_HIP(hipGetLastError());
  return ST_OK;
}

}  // namespace st

extern "C" size_t st_front_frag_floats(int Cout, int Cin) { return st::front_frag_floats(Cout, Cin); }
extern "C" int st_front_pack_frags(const float* packed_wgt_host, int Cout, int Cin, float* out_host) {
  return st::front_pack_frags(packed_wgt_host, Cout, Cin, out_host);
}
extern "C" int st_conv3x3s2_csp_front(const StConvDesc* a, const StConvDesc* ms, const StConvDesc* c1,
                                      const float* frag_ms_dev, const float* frag_c1_dev, st_stream_t stream) {
  if (!a || !ms || !c1) return st::set_error(ST_ERR_INVALID, "st_conv3x3s2_csp_front: null desc");
  return st::front_fused_launch(*a, *ms, *c1, frag_ms_dev, frag_c1_dev, static_cast<hipStream_t>(stream));
}
