// Fused Focus (space-to-depth) + 3x3 stem ConvModule (conv + folded BN + SiLU) for gfx950.
//
// Replaces, in one kernel, the stem of the two-branch backbone: mmdet Focus slicing
// (reference mmtrack/models/backbones/csp_darknet_disparity_v1.py:104-111 builds `Focus(3, c, 3)` for the RGB
// and the disparity stem; :176-179 run them) followed by its ConvModule.  Focus(x) then a 3x3/s1/p1 conv
// over the 12 sliced channels IS a 6x6 / stride-2 / pad-2 convolution over the 3-channel NCHW image:
//   W6[co][c][2*ky+dy][2*kx+dx] = W[co][(dy + 2*dx)*3 + c][ky][kx]     (Focus order TL, BL, TR, BR)
// so the kernel reads the planar fp32 image directly (no 12-channel NHWC intermediate: 241 MB less HBM traffic
// per 16 images) and multiplies with K = 108 exactly (the generic implicit-GEMM pads K to 128).
//
// Work decomposition (one workgroup = 8 x 64 output pixels of one image, 4 waves):
//   * the (2*8+4) x (2*64+8) x 3 input window goes to LDS by LDS-DMA (buffer_load_dwordx4 ... lds, float4
//     aligned rows), out-of-image float4 are zero-filled by the buffer unit's range check;
//   * every wave owns 2 output rows = 4 blocks of 32 pixels; the WHOLE weight matrix (108 x Cout, 13.8 KB)
//     sits in LDS next to the window, so the main loop is  ds_read_b32 -> v_mfma_f32_32x32x2f32: five 4-byte
//     LDS reads per four 64-cycle MFMAs, no global loads, no barriers;
//   * workgroups are persistent with two window buffers: tile t+grid streams in while tile t computes;
//   * k order (c, ky6, kx6) with kx6 fastest: the two k of an MFMA step are x-neighbours, so lane half
//     (l >> 5) is a +1 float offset and everything else is an immediate offset of the ds_read;
//     lanes of one half read stride-2 floats (even or odd banks): conflict-free.
//   * the MFMA operands are swapped (A = weights, B = pixels): a lane ends up with 4 x 4 consecutive couts of
//     ONE pixel, so the epilogue (+ bias, SiLU) stores NHWC with 16-byte writes.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include <cstdlib>

#include "st_common.h"

namespace st {
namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int STEM_TH = 8, STEM_TW = 64;                  // output tile (STEM_TH in {4, 8})
constexpr int STEM_WB = STEM_TH / 2;                      // 32-pixel blocks per wave (4 waves x WB = TH x 2)
constexpr int STEM_IH = 2 * STEM_TH + 4;                  // 20 input rows
constexpr int STEM_IW = 2 * STEM_TW + 8;                  // 136 input cols: window starts 4 px left of the
                                                          // first needed col - 2, so rows are float4 aligned
constexpr int STEM_ICH = STEM_IH * STEM_IW;               // floats per channel plane in LDS
constexpr int stem_kp(int cin) { return cin * 18; }       // k pairs: cin * 6 * 6 / 2  (54 for the 3-plane image)
constexpr int STEM_ROW4 = STEM_IW / 4;                    // 34 float4 per staged row
constexpr int stem_tile4(int cin) { return cin * STEM_IH * STEM_ROW4; }          // 2040 float4 per 3-plane window
constexpr int stem_ndma(int cin) { return (stem_tile4(cin) + 255) / 256; }        // 8 wave-instructions of 1 KiB
constexpr int stem_buf(int cin) { return stem_ndma(cin) * 256 * 4; }              // floats per window buffer (padded)

constexpr int STEM_MAX_FRAMES = 32;

struct StemArgs {
  const float* in;     // [N][3][H][W] planar fp32
  const float* wgt;    // [108][CoutPad]
  const float* bias;   // [CoutPad]
  float* out;          // NHWC, pixel stride out_ld, channel offset out_off
  int N, H, W, Ho, Wo, Cout, CoutPad, out_ld, out_off, act;
  int tiles_x, tiles_y;
  int planes;          // planes per image in memory (3); the kernel reads the first CIN of them
  unsigned out_bytes;  // bytes addressable through `out` (range check of the epilogue stores)
  // raw-input form (U8 kernels): the N images are separate uint8 [3][h][w] frames (h <= H, w <= W, w % 4 == 0); the
  // window is converted while it is staged, pixels of the padded H x W image outside h x w read `pad_u8` - the cast +
  // pad of the data preprocessor (data_preprocessor_disparity_v1.py:38-51) without an fp32 copy of the image in HBM
  const unsigned char* frames[STEM_MAX_FRAMES];
  int h, w;
  unsigned pad4;       // pad value replicated into 4 bytes
};

__device__ __forceinline__ float stem_silu(float v) { return v * __builtin_amdgcn_rcpf(1.0f + __expf(-v)); }

// Global -> LDS of one input window with LDS-DMA (`buffer_load_dwordx4 ... lds`: no VGPR staging).  Every
// wave-instruction fills 1 KiB of the flat [3][20][136] window; lanes whose float4 lies outside the image (or
// past the window's 2040 float4) carry an out-of-range offset, which the buffer unit turns into zeros.
template <int CIN>
__device__ __forceinline__ void stem_dma(const StemArgs& p, int t, int tid, float* dst) {
  const int tx = t % p.tiles_x;
  const int t2 = t / p.tiles_x;
  const int ty = t2 % p.tiles_y, n = t2 / p.tiles_y;
  const int iy0 = 2 * ty * STEM_TH - 2, ix0 = 2 * tx * STEM_TW - 4;
#if defined(__HIP_DEVICE_COMPILE__)  // device-only builtins; the host pass only needs the kernel stub
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(p.in + (size_t)n * p.planes * p.H * p.W), 0, CIN * p.H * p.W * 4, 0x00020000);
  const int wave = tid >> 6;
#pragma unroll
  for (int j = 0; j < stem_ndma(CIN); ++j) {
    const int idx = tid + 256 * j;
    const int rowc = idx / STEM_ROW4, col4 = idx - rowc * STEM_ROW4;   // rowc = c * 20 + row
    const int c = rowc / STEM_IH, row = rowc - c * STEM_IH;
    const int gy = iy0 + row, gx = ix0 + 4 * col4;
    const bool ok = idx < stem_tile4(CIN) && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
    const unsigned off = ok ? (unsigned)(((c * p.H + gy) * p.W + gx) * 4) : 0x80000000u;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(
        rsrc, (__attribute__((address_space(3))) void*)(dst + (j * 256 + wave * 64) * 4), 16, off, 0, 0, 0);
  }
#else
  (void)p; (void)t; (void)tid; (void)dst; (void)iy0; (void)ix0; (void)n;
#endif
}

// The same window from uint8 frames: 4 pixels (one dword) per float4 slot into registers ...
template <int CIN>
__device__ __forceinline__ void stem_fetch_u8(const StemArgs& p, int t, int tid, unsigned (&q)[stem_ndma(CIN)]) {
  const int tx = t % p.tiles_x;
  const int t2 = t / p.tiles_x;
  const int ty = t2 % p.tiles_y, n = t2 / p.tiles_y;
  const int iy0 = 2 * ty * STEM_TH - 2, ix0 = 2 * tx * STEM_TW - 4;
  const unsigned char* __restrict__ f = p.frames[n];     // uniform: scalar load from the kernel arguments
#pragma unroll
  for (int j = 0; j < stem_ndma(CIN); ++j) {
    const int idx = tid + 256 * j;
    const int rowc = idx / STEM_ROW4, col4 = idx - rowc * STEM_ROW4;
    const int c = rowc / STEM_IH, row = rowc - c * STEM_IH;
    const int gy = iy0 + row, gx = ix0 + 4 * col4;
    const bool in_pad = idx < stem_tile4(CIN) && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;   // else conv zero padding
    const bool in_img = in_pad && gy < p.h && gx < p.w;
    unsigned v = in_pad ? p.pad4 : 0u;
    if (in_img) v = *reinterpret_cast<const unsigned*>(f + ((size_t)(c * p.h + gy) * p.w + gx));
    q[j] = v;
  }
}

// ... and, converted to fp32, into the LDS window (same layout the LDS-DMA produces)
template <int CIN>
__device__ __forceinline__ void stem_commit_u8(const unsigned (&q)[stem_ndma(CIN)], int tid, float* dst) {
#pragma unroll
  for (int j = 0; j < stem_ndma(CIN); ++j) {
    const unsigned v = q[j];
    f32x4 o;
    o[0] = (float)(v & 0xffu); o[1] = (float)((v >> 8) & 0xffu); o[2] = (float)((v >> 16) & 0xffu); o[3] = (float)(v >> 24);
    *reinterpret_cast<f32x4*>(dst + (size_t)(tid + 256 * j) * 4) = o;
  }
}

// Persistent workgroups (grid = resident slots) with two window buffers: the window of tile t+grid streams into
// LDS while tile t runs its MFMAs, and the weights are staged once per workgroup instead of once per tile.
template <int NB, bool VEC, int CIN, bool U8>
__global__ __launch_bounds__(256, NB == 1 ? 2 : 1) void stem_focus_conv_kernel(const StemArgs p) {
  extern __shared__ float4 stem_smem4[];
  float* smem = reinterpret_cast<float*>(stem_smem4);
  constexpr int STEM_KP = stem_kp(CIN), STEM_BUF = stem_buf(CIN);
  float* wl = smem + 2 * STEM_BUF;   // [CIN*36][NB*32]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int half = lane >> 5, l31 = lane & 31;
  const int total = p.N * p.tiles_x * p.tiles_y;
  int t = blockIdx.x;
  if (t >= total) return;   // uniform per workgroup

  unsigned q[stem_ndma(CIN)];
  if (U8) {
    stem_fetch_u8<CIN>(p, t, tid, q);
    stem_commit_u8<CIN>(q, tid, smem);
  } else {
    stem_dma<CIN>(p, t, tid, smem);
  }
  for (int idx = tid; idx < 2 * STEM_KP * NB * 32; idx += 256) wl[idx] = p.wgt[idx];
  const float* wbase = wl + half * (NB * 32) + l31;   // B operand of lane = (co = l31, k = 2s + half)
  // operands are swapped (A = weights, B = pixels), so the accumulator is C[co][pixel]: lane = pixel l31 of the
  // block, holding couts 8g + 4*half + {0..3}, g = 0..3 -> four 16-byte NHWC stores per block
  f32x4 bv[NB][4];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb)
#pragma unroll
    for (int g = 0; g < 4; ++g) bv[nb][g] = *reinterpret_cast<const f32x4*>(p.bias + nb * 32 + 8 * g + 4 * half);
  int aoff[STEM_WB];
#pragma unroll
  for (int i = 0; i < STEM_WB; ++i) {
    const int r = (STEM_WB / 2) * wave + (i >> 1), xb = i & 1;
    aoff[i] = (2 * r) * STEM_IW + 2 * (xb * 32 + l31) + half + 2;
  }
  __syncthreads();   // vmcnt(0) + barrier: first window landed, weights visible
  int cur = 0;

  // Epilogue of one (block i, column block nb, cout group g) = 4 consecutive couts of this lane's pixel:
  // + bias, SiLU, one 16-byte NHWC store.  C[co][pixel]: col = l31 (pixel), row = (r&3) + 8*(r>>2) + 4*half.
  // Branch-free: stores go through a range-checked buffer resource, lanes that must not write (outside the
  // image, padded couts, "no previous tile yet") carry an out-of-range offset and are dropped by the hardware.
  f32x16 prev[STEM_WB][NB];   // finished accumulators of the previous tile, drained inside the next MFMA loop
#pragma unroll
  for (int i = 0; i < STEM_WB; ++i)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int r = 0; r < 16; ++r) prev[i][nb][r] = 0.f;
  unsigned pix_off[STEM_WB];   // byte offset of this lane's pixel (channel out_off) per block, or out-of-range
#pragma unroll
  for (int i = 0; i < STEM_WB; ++i) pix_off[i] = 0x80000000u;
#if defined(__HIP_DEVICE_COMPILE__)
  const __amdgpu_buffer_rsrc_t orsrc =
      __builtin_amdgcn_make_buffer_rsrc(p.out, 0, (int)p.out_bytes, 0x00020000);
#endif
  auto set_prev_tile = [&](int tt) {
    const int tx = tt % p.tiles_x;
    const int t2 = tt / p.tiles_x;
    const int ty = t2 % p.tiles_y, n = t2 / p.tiles_y;
#pragma unroll
    for (int i = 0; i < STEM_WB; ++i) {
      const int oy = ty * STEM_TH + (STEM_WB / 2) * wave + (i >> 1), ox = tx * STEM_TW + (i & 1) * 32 + l31;
      const bool ok = oy < p.Ho && ox < p.Wo;
      pix_off[i] = ok ? (unsigned)((((n * p.Ho + oy) * p.Wo + ox) * p.out_ld + p.out_off) * 4) : 0x80000000u;
    }
  };
  auto store_group = [&](int q) {
    const int i = q / (NB * 4), nb = (q / 4) % NB, g = q % 4;
    const int co = nb * 32 + 8 * g + 4 * half;
    f32x4 v;
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = prev[i][nb][4 * g + e];   // bias included: the accumulator started from it
    if (p.act) {   // uniform branch (a per-value select would cost a vector instruction per output)
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = stem_silu(v[e]);
    }
#if defined(__HIP_DEVICE_COMPILE__)
    if (VEC) {   // out_ld, out_off, Cout multiples of 4: one dwordx4 per group
      const unsigned off = co < p.Cout ? pix_off[i] + co * 4 : 0x80000000u;
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((ext_vector_type(4))) unsigned, v),
                                             orsrc, off, 0, 0);
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const unsigned off = co + e < p.Cout ? pix_off[i] + (co + e) * 4 : 0x80000000u;
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v[e]), orsrc, off, 0, 0);
      }
    }
#else
    (void)v; (void)co;
#endif
  };
  constexpr int NGROUPS = STEM_WB * NB * 4;
  // store groups spread evenly over the k loop.  Measured on MI355X (8 x 736 x 1280 -> 32 ch): spread 141 us,
  // all groups in the first 16 steps 152 us, separate epilogue phase after the barrier 149 us: VALU work does
  // not co-issue with the wave's own fp32 MFMAs, so bunching it only lengthens the bubbles.
  constexpr int GSTRIDE = STEM_KP / NGROUPS > 0 ? STEM_KP / NGROUPS : 1;   // k-steps between two store groups
  constexpr int GPER = (NGROUPS + STEM_KP - 1) / STEM_KP;                  // groups per k-step when KP < NGROUPS
  auto koff = [](int s) {   // compile-time window offset of k-pair s: k order (c, ky6, kx6), kx6 fastest
    const int c = s / 18, rem = s % 18, ky6 = rem / 3, pp = rem % 3;
    return c * STEM_ICH + ky6 * STEM_IW + 2 * pp;
  };

  while (true) {
    const int tn = t + gridDim.x;
    if (tn < total) {   // lands during the MFMA phase
      if (U8) stem_fetch_u8<CIN>(p, tn, tid, q);
      else stem_dma<CIN>(p, tn, tid, smem + (cur ^ 1) * STEM_BUF);
    }
    const float* win = smem + cur * STEM_BUF;

    // ---- STEM_WB pixel blocks per wave: rows (WB/2)*wave + (i >> 1), x blocks i & 1.
    // Straight-line k loop: [ds_read operands of step s+1] [MFMAs of step s] [a slice of the PREVIOUS tile's
    // epilogue], so LDS latency and the epilogue VALU/stores sit in the shadow of the MFMAs (co-resident
    // workgroups run in lockstep: a separate epilogue phase would leave the MFMA pipe idle).
    f32x16 acc[STEM_WB][NB];
#pragma unroll
    for (int i = 0; i < STEM_WB; ++i)
#pragma unroll
      for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][nb][r] = bv[nb][r >> 2][r & 3];   // start from the bias: no add per output
    float a[2][STEM_WB], w[2][NB];
#pragma unroll
    for (int i = 0; i < STEM_WB; ++i) a[0][i] = win[aoff[i] + koff(0)];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) w[0][nb] = wbase[nb * 32];
#pragma unroll
    for (int s = 0; s < STEM_KP; ++s) {
      const int cb = s & 1, nbuf = cb ^ 1;
      if (s + 1 < STEM_KP) {
#pragma unroll
        for (int i = 0; i < STEM_WB; ++i) a[nbuf][i] = win[aoff[i] + koff(s + 1)];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) w[nbuf][nb] = wbase[(2 * (s + 1)) * (NB * 32) + nb * 32];
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < STEM_WB; ++i)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
          acc[i][nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[cb][nb], a[cb][i], acc[i][nb], 0, 0, 0);
      if (s % GSTRIDE == 0) {
#pragma unroll
        for (int u = 0; u < GPER; ++u)
          if ((s / GSTRIDE) * GPER + u < NGROUPS) store_group((s / GSTRIDE) * GPER + u);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (U8 && tn < total) stem_commit_u8<CIN>(q, tid, smem + (cur ^ 1) * STEM_BUF);   // nobody reads that buffer now
    // next window landed (vmcnt(0)) and every wave is done reading this one
    __syncthreads();
#pragma unroll
    for (int i = 0; i < STEM_WB; ++i)
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) prev[i][nb] = acc[i][nb];
    set_prev_tile(t);
    if (tn >= total) break;   // uniform: every wave of the workgroup leaves together
    t = tn;
    cur ^= 1;
  }
  // drain: the last tile's epilogue
#pragma unroll
  for (int q = 0; q < NGROUPS; ++q) store_group(q);
}

}  // namespace

int stem_focus_conv_launch(const float* in, int N, int H, int W, int used_planes, const float* wgt,
                           const float* bias, int Cout, float* out, int out_ld, int out_off, int act,
                           hipStream_t stream, const StemRawInput* raw) {
  ST_REQUIRE((in || raw) && wgt && bias && out, "stem_focus_conv: null pointer");
  if (raw) {
    ST_REQUIRE(used_planes == 3 && N <= STEM_MAX_FRAMES, "stem_focus_conv: raw frames need the 3-plane stem and N <= %d",
               STEM_MAX_FRAMES);
    ST_REQUIRE(raw->h > 0 && raw->h <= H && raw->w > 0 && raw->w <= W && raw->w % 4 == 0,
               "stem_focus_conv: raw frame %dx%d does not fit the padded %dx%d image (w %% 4 == 0)", raw->h, raw->w, H, W);
    ST_REQUIRE(raw->pad_value >= 0.f && raw->pad_value <= 255.f && raw->pad_value == (float)(int)raw->pad_value,
               "stem_focus_conv: raw frames need an integral pad value in 0..255 (got %g)", (double)raw->pad_value);
    for (int i = 0; i < N; ++i)
      ST_REQUIRE(raw->frames[i] && (reinterpret_cast<uintptr_t>(raw->frames[i]) & 3) == 0,
                 "stem_focus_conv: raw frame %d null or not 4-byte aligned", i);
  }
  ST_REQUIRE(used_planes == 3 || used_planes == 1, "stem_focus_conv: used_planes must be 3 or 1 (got %d)",
             used_planes);
  ST_REQUIRE(N > 0 && H > 0 && W > 0 && H % 2 == 0 && W % 4 == 0,
             "stem_focus_conv: H must be even and W a multiple of 4 (got %dx%d)", H, W);
  ST_REQUIRE((long long)3 * H * W * 4 < (1ll << 31), "stem_focus_conv: image exceeds 2 GiB");
  ST_REQUIRE(Cout > 0 && Cout <= 64, "stem_focus_conv: Cout must be in 1..64 (got %d)", Cout);
  ST_REQUIRE(out_off >= 0 && out_off + Cout <= out_ld, "stem_focus_conv: output slice exceeds out_ld");
  ST_REQUIRE(raw || (reinterpret_cast<uintptr_t>(in) & 15) == 0, "stem_focus_conv: input must be 16-byte aligned");
  StemArgs a;
  for (int i = 0; i < STEM_MAX_FRAMES; ++i) a.frames[i] = raw && i < N ? raw->frames[i] : nullptr;
  a.h = raw ? raw->h : H; a.w = raw ? raw->w : W;
  a.pad4 = raw ? 0x01010101u * (unsigned)(int)raw->pad_value : 0u;
  a.in = in; a.wgt = wgt; a.bias = bias; a.out = out;
  a.N = N; a.H = H; a.W = W; a.Ho = H / 2; a.Wo = W / 2; a.Cout = Cout; a.CoutPad = round_up(Cout, 32);
  a.out_ld = out_ld; a.out_off = out_off; a.act = act; a.planes = 3;
  a.tiles_x = ceil_div(a.Wo, STEM_TW); a.tiles_y = ceil_div(a.Ho, STEM_TH);
  const long long tiles = (long long)N * a.tiles_x * a.tiles_y;
  ST_REQUIRE(tiles < (1ll << 30), "stem_focus_conv: too many tiles");
  const int nb = a.CoutPad / 32;
  const size_t lds = (size_t)(2 * stem_buf(used_planes) + 2 * stem_kp(used_planes) * nb * 32) * sizeof(float);
  // 3 planes: 79.4 KB (Cout <= 32) / 93.2 KB; 1 plane: 28.8 KB / 33.4 KB
  // persistent workgroups: as many as fit a CU's 160 KB of LDS (2 for Cout <= 32), each walks t, t + grid, ...
  static int cus = 0;
  if (!cus) {
    int dev = 0;
    ST_CHECK_HIP(hipGetDevice(&dev));
    ST_CHECK_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    if (cus <= 0) cus = 256;
  }
  const long long out_bytes = (long long)N * a.Ho * a.Wo * out_ld * 4;
  ST_REQUIRE(out_bytes < (1ll << 31), "stem_focus_conv: output exceeds 2 GiB (split the batch)");
  a.out_bytes = (unsigned)out_bytes;
  const bool vec = ((out_ld | out_off | Cout) & 3) == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0;
  using Kern = void (*)(const StemArgs);
  static const Kern kerns[12] = {
      stem_focus_conv_kernel<1, false, 3, false>, stem_focus_conv_kernel<1, true, 3, false>,
      stem_focus_conv_kernel<2, false, 3, false>, stem_focus_conv_kernel<2, true, 3, false>,
      stem_focus_conv_kernel<1, false, 1, false>, stem_focus_conv_kernel<1, true, 1, false>,
      stem_focus_conv_kernel<2, false, 1, false>, stem_focus_conv_kernel<2, true, 1, false>,
      stem_focus_conv_kernel<1, false, 3, true>,  stem_focus_conv_kernel<1, true, 3, true>,
      stem_focus_conv_kernel<2, false, 3, true>,  stem_focus_conv_kernel<2, true, 3, true>};
  const int ki = (raw ? 8 : used_planes == 1 ? 4 : 0) + (nb - 1) * 2 + (vec ? 1 : 0);
  static bool attr_set[12] = {false, false, false, false, false, false, false, false, false, false, false, false};
  if (!attr_set[ki]) {
    ST_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kerns[ki]),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_set[ki] = true;
  }
  // resident workgroups per CU: LDS, and 2 waves per SIMD (NB = 1) / 1 (NB = 2) by registers
  const int per_cu = std::min((int)((160 * 1024) / lds), nb == 1 ? 2 : 1);
  const unsigned grid = (unsigned)std::min<long long>(tiles, (long long)cus * per_cu);
  hipLaunchKernelGGL(kerns[ki], dim3(grid), dim3(256), lds, stream, a);
  ST_CHECK_HIP(hipGetLastError());
  return ST_OK;
}

}  // namespace st

extern "C" size_t st_stem_packed_floats(int Cout) {
  return Cout > 0 ? (size_t)2 * st::stem_kp(3) * st::round_up(Cout, 32) : 0;
}

// w: the stem ConvModule's conv weight [Cout][12][3][3] (channels in Focus order), optional conv bias, optional
// BatchNorm running statistics (folded in fp64 exactly like st_conv_pack_weights).  used_planes = 1 packs the
// sum over the three image planes of every tap (36 x Cout): exact when the caller guarantees that the three
// planes of the image are identical (the disparity input, a 3-channel repeat of one map).
extern "C" int st_stem_pack_weights(const float* w, const float* conv_bias, const float* bn_gamma,
                                    const float* bn_beta, const float* bn_mean, const float* bn_var, double bn_eps,
                                    int Cout, int used_planes, float* wgt_out, float* bias_out) {
  if (!w || !wgt_out || !bias_out || Cout <= 0 || Cout > 64 || (used_planes != 3 && used_planes != 1))
    return st::set_error(ST_ERR_INVALID, "st_stem_pack_weights: bad argument");
  const bool has_bn = bn_gamma != nullptr;
  if (has_bn && (!bn_beta || !bn_mean || !bn_var))
    return st::set_error(ST_ERR_INVALID, "st_stem_pack_weights: incomplete BN parameters");
  const int CoutPad = st::round_up(Cout, 32);
  std::memset(wgt_out, 0, sizeof(float) * (size_t)2 * st::stem_kp(3) * CoutPad);
  std::vector<double> accum((size_t)36 * (used_planes == 1 ? 1 : 3), 0.0);
  std::memset(bias_out, 0, sizeof(float) * (size_t)CoutPad);
  for (int co = 0; co < Cout; ++co) {
    double scale = 1.0, shift = conv_bias ? (double)conv_bias[co] : 0.0;
    if (has_bn) {
      const double inv = (double)bn_gamma[co] / std::sqrt((double)bn_var[co] + bn_eps);
      scale = inv;
      shift = (double)bn_beta[co] + (shift - (double)bn_mean[co]) * inv;
    }
    bias_out[co] = (float)shift;
    std::fill(accum.begin(), accum.end(), 0.0);
    for (int c = 0; c < 3; ++c)
      for (int ky6 = 0; ky6 < 6; ++ky6)
        for (int kx6 = 0; kx6 < 6; ++kx6) {
          const int ky = ky6 >> 1, dy = ky6 & 1, kx = kx6 >> 1, dx = kx6 & 1;
          const int cf = (dy + 2 * dx) * 3 + c;  // Focus channel: TL, BL, TR, BR groups of 3
          const double v = (double)w[(((size_t)co * 12 + cf) * 3 + ky) * 3 + kx] * scale;
          accum[(size_t)((used_planes == 1 ? 0 : c) * 6 + ky6) * 6 + kx6] += v;   // fp64 sum, rounded once
        }
    for (size_t k = 0; k < accum.size(); ++k) wgt_out[k * CoutPad + co] = (float)accum[k];
  }
  return ST_OK;
}

extern "C" int st_stem_focus_conv(const float* img_dev, int N, int H, int W, int used_planes,
                                  const float* wgt_dev, const float* bias_dev, int Cout, float* out_dev, int out_ld,
                                  int out_off, int act, st_stream_t stream) {
  return st::stem_focus_conv_launch(img_dev, N, H, W, used_planes, wgt_dev, bias_dev, Cout, out_dev, out_ld, out_off,
                                    act, static_cast<hipStream_t>(stream), nullptr);
}

extern "C" int st_stem_focus_conv_u8(const unsigned char* const* frames_u8_dev_ptrs_host, int N, int h, int w, int H, int W,
                                     float pad_value, const float* wgt_dev, const float* bias_dev, int Cout,
                                     float* out_dev, int out_ld, int out_off, int act, st_stream_t stream) {
  ST_REQUIRE(frames_u8_dev_ptrs_host != nullptr, "st_stem_focus_conv_u8: null frame table");
  const st::StemRawInput raw{frames_u8_dev_ptrs_host, h, w, pad_value};
  return st::stem_focus_conv_launch(nullptr, N, H, W, 3, wgt_dev, bias_dev, Cout, out_dev, out_ld, out_off, act,
                                    static_cast<hipStream_t>(stream), &raw);
}
