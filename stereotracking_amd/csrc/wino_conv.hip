// Winograd F(2x2, 3x3) convolution on the fp32 matrix cores of gfx950, NHWC, tile variant 43.
//
// The 3x3 / stride-1 ConvModules with >= 64 channels (YOLOX head towers, CSP bottleneck conv2 layers, PAFPN blocks:
// mmdet CSPLayer / mmyolo YOLOXHeadModule as built at reference
// mmtrack/models/backbones/csp_darknet_disparity_v1.py:145-153 and configs/_base_/yolox_s_8x8_mmyolo.py:38-51) carry
// 56 % of the multiply-adds of the path.  v_mfma_f32_32x32x2_f32 runs at the fp32 VECTOR rate (157 TFLOP/s), so the
// only way past that roof is to need fewer multiplies:   Y = A^T [ (G g G^T) .* (B^T d B) ] A   computes a 2x2 output
// tile from a 4x4 input patch with 16 multiplies per (cin, cout) instead of 36 - 2.25x fewer MFMA instructions for
// the same convolution (fp32 throughout; the transforms only add / subtract / halve, error ~1e-6 relative, inside
// the 1e-3 float tolerance of the path and checked against torch conv2d in tests/test_conv_gpu.py).
//
// Mapping.  16 independent GEMMs, one per transform coordinate xi = (a, b):  M_xi[tile][co] = sum_ci V_xi[tile][ci] *
// U_xi[co][ci].  A workgroup = 4 waves = the 4 rows `a` of the transform; it owns 32 tiles (8 x 4 tiles = 16 x 8 output
// pixels) x 64 couts, i.e. per wave 4 (b) x 2 (cout blocks of 32) accumulator tiles of 32x32.
//   * raw input window (18 x 10 pixels x 32 cin per K-chunk, 23 KB) -> LDS by LDS-DMA with hardware zero fill (= the
//     conv padding and the ragged edge), double buffered, source-side XOR swizzle (2-way conflicts at most);
//   * the input transform never touches memory: a lane (tile, 4 channels) reads its 2 x 4 patch pixels with 8
//     ds_read_b128, forms its wave's row of B^T d (4 adds) and the 4 values V[a][0..3] (4 adds) in registers - those ARE
//     the A operands of the next 4 x 2 x 4 MFMAs;
//   * the transformed weights are pre-packed on the host in FRAGMENT ORDER, so every B operand is one fully coalesced
//     1 KB wave load straight from L2 into VGPRs (1 MB per layer, shared by every workgroup), prefetched one step ahead;
//   * output transform: each wave reduces its own row over b in registers (M -> 2 values), the 4 rows meet through
//     LDS (64 KB, reusing the window buffers), then bias + SiLU (+ residual) and coalesced NHWC stores.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "st_common.h"
#include "wino_pk.h"

namespace st {
namespace {

constexpr int WN_TX = 8, WN_TY = 4;                        // tiles per workgroup (x, y): 32 = one MFMA row block
constexpr int WN_WW = 2 * WN_TX + 2, WN_WH = 2 * WN_TY + 2;   // input window: 18 x 10 pixels
constexpr int WN_PIX = WN_WW * WN_WH;                      // 180
constexpr int WN_PIECES = (WN_PIX + 7) / 8;                // LDS-DMA pieces of 8 pixels x 128 B = 1 KB
constexpr int WN_WIN_FLOATS = WN_PIECES * 8 * 32;          // one window buffer (32 cin per pixel)
// couts per workgroup = 32 * CBN (CBN = cout blocks of 32 per wave): 64 for the wide layers, 32 for Cout = 32;
// Cout = 48 runs as one padded block of 64 (zero weights, stores masked)
constexpr int wn_lds_floats(int cbn) {
  return 4 * 2 * 32 * 32 * cbn > 2 * WN_WIN_FLOATS ? 4 * 2 * 32 * 32 * cbn : 2 * WN_WIN_FLOATS;   // R exchange vs windows
}

struct WinoArgs {
  const float* in;
  const float* wino;
  const float* bias;
  float* out;
  const float* res;
  int N, H, W, Cin, in_ld, in_off, Cout;
  int out_ld, out_off, res_ld, res_off;
  float post_scale;
  int act;
  int tbx, tby, ncb, nkc;
  int g_last;   // channel groups of 8 that exist in the LAST K-chunk (4 unless Cin % 32 != 0)
  unsigned nblocks, per_xcd, order, grid;   // workgroups of this problem and their order over the grid (wn_block_of)
  unsigned in_bytes, out_bytes, res_bytes, wino_bytes;
};

// CBN = cout blocks of 32 a workgroup computes; LCBN = cout blocks per group in the weight LAYOUT (>= CBN).  LCBN = 2 with
// CBN = 1 (tile variant 44) runs a 64-cout layout with 32-cout workgroups: twice the workgroups, for the small maps
// (23x40, 46x80) whose grid otherwise fills less than one round of the chip.
// KTAIL: Cin % 32 != 0 (the 48-level aggregation convs): the last K-chunk runs only the channel groups that exist
// (g_last of 4) instead of multiplying zero-padded channels - a quarter of the layer's MFMAs at Cin = 48.
template <int CBN, int LCBN, bool RES, bool KTAIL>
__device__ __forceinline__ void wino_conv3x3_body(const WinoArgs& p, int b_) {
  constexpr int WN_CB = 32 * CBN;                       // couts per workgroup
  constexpr int WN_FRAG_FLOATS = LCBN * 64 * 4;         // one (kc, g, b) step in memory: LCBN cout blocks x 64 lanes x 4
  extern __shared__ float4 wn_smem4[];
  float* smem = reinterpret_cast<float*>(wn_smem4);
  const int tid = threadIdx.x, lane = tid & 63, a = tid >> 6;   // wave = transform row a
  const int i = lane & 31, h = lane >> 5;
  const int tyi = i >> 3, txi = i & 7;
  const int cb = b_ % p.ncb; b_ /= p.ncb;
  const int bx = b_ % p.tbx; b_ /= p.tbx;
  const int by = b_ % p.tby;
  const int n = b_ / p.tby;
  const int wy0 = by * (2 * WN_TY) - 1, wx0 = bx * (2 * WN_TX) - 1;   // window origin (conv padding = 1)

#if defined(__HIP_DEVICE_COMPILE__)  // device-only builtins; the host pass only needs the kernel stub
  const __amdgpu_buffer_rsrc_t irsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.in), 0, (int)p.in_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t wrsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.wino), 0, (int)p.wino_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, (int)p.out_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rrsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(RES ? p.res : p.in), 0, (int)(RES ? p.res_bytes : 0u), 0x00020000);

  // ---- window DMA: piece j = LDS pixel slots 8j .. 8j+7; lane (pixel slot q, 16-B slot sl) fetches channel quad
  // sl ^ ((q >> 1) & 7) of window pixel q (zero outside the image / past the window)
  constexpr int NPW = (WN_PIECES + 3) / 4;   // pieces per wave
  unsigned poff[NPW];
  unsigned tail_ok = 0;                      // bit k: this lane's channel quad exists in the LAST K-chunk (Cin % 32 != 0)
#pragma unroll
  for (int k = 0; k < NPW; ++k) {
    const int j = a + 4 * k;
    const int q = 8 * j + (lane >> 3), sl = lane & 7;
    const int wr = q / WN_WW, wc = q - wr * WN_WW;
    const int y = wy0 + wr, x = wx0 + wc;
    const bool ok = j < WN_PIECES && q < WN_PIX && y >= 0 && y < p.H && x >= 0 && x < p.W;
    const int quad = sl ^ ((q >> 1) & 7);
    poff[k] = ok ? (unsigned)((((n * p.H + y) * p.W + x) * p.in_ld + p.in_off + 4 * quad) * 4) : 0x80000000u;
    if ((p.nkc - 1) * 32 + 4 * quad < p.Cin) tail_ok |= 1u << k;
  }
  auto dma_window = [&](int kc, int buf) {
    const bool last = kc == p.nkc - 1;
#pragma unroll
    for (int k = 0; k < NPW; ++k) {
      const int j = a + 4 * k;
      const unsigned off = (last && !((tail_ok >> k) & 1u)) ? 0x80000000u : poff[k];   // channels >= Cin read as zero
      if (j < WN_PIECES)   // wave-uniform
        __builtin_amdgcn_raw_ptr_buffer_load_lds(
            irsrc, (__attribute__((address_space(3))) void*)(smem + buf * WN_WIN_FLOATS + j * 256), 16, off,
            kc * 128, 0, 0);
    }
  };

  // ---- this lane's 2 x 4 patch pixels: rows (r0, r1) of wave a's row of B^T, combined as d[r0] + sgn * d[r1]
  //   a = 0: d0 - d2    a = 1: d1 + d2    a = 2: d2 - d1    a = 3: d1 - d3
  const int r0 = a == 0 ? 0 : (a == 2 ? 2 : 1);
  const int r1 = a == 0 ? 2 : (a == 1 ? 2 : (a == 2 ? 1 : 3));
  const f32x2 sgn = a == 1 ? f32x2{1.0f, 1.0f} : f32x2{-1.0f, -1.0f};
  int qoff[8], qsw[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int r = k < 4 ? r0 : r1, c = k & 3;
    const int q = (2 * tyi + r) * WN_WW + 2 * txi + c;
    qoff[k] = q * 32;
    qsw[k] = (q >> 1) & 7;
  }

  // ---- transformed-weight stream of this wave: [cb][a][kc][g][b][nb][lane][4], one step = WN_FRAG_FLOATS
  constexpr int SPLIT = LCBN / CBN;   // workgroups sharing one layout group
  const unsigned wbase = (unsigned)(((((cb / SPLIT) * 4 + a) * p.nkc) * 16) * WN_FRAG_FLOATS + (cb % SPLIT) * CBN * 256 +
                                    lane * 4) * 4u;
  const int last_step = p.nkc * 16 - 1;
  auto load_frag = [&](int step, f32x4 (&f)[CBN]) {   // step = (kc * 4 + g) * 4 + b; the prefetch past the last step
    // the step displacement is wave-uniform: it rides in the scalar offset, no vector add per load
    const int soff = (step < last_step ? step : last_step) * (WN_FRAG_FLOATS * 4);
#pragma unroll
    for (int nb = 0; nb < CBN; ++nb)
      f[nb] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, wbase, soff + 1024 * nb, 0));
  };

  f32x16 acc[4][CBN];
#pragma unroll
  for (int b = 0; b < 4; ++b)
#pragma unroll
    for (int nb = 0; nb < CBN; ++nb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[b][nb][r] = 0.f;

  dma_window(0, 0);
  f32x4 fe[CBN], fo[CBN];   // weight fragments of the even / odd steps (16 steps per chunk: static assignment)
  load_frag(0, fe);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  for (int kc = 0; kc < p.nkc; ++kc) {
    const int buf = kc & 1;
    if (kc + 1 < p.nkc) dma_window(kc + 1, buf ^ 1);   // lands during this chunk's 128 MFMAs
    const float* win = smem + buf * WN_WIN_FLOATS;
    const int gn = (KTAIL && kc == p.nkc - 1) ? p.g_last : 4;   // wave-uniform
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      if (KTAIL && g >= gn) continue;
      // raw patch -> V[a][0..3] for this lane's 4 channels (8g + 4h .. + 3)
      f32x4 d[8];
#pragma unroll
      for (int k = 0; k < 8; ++k)
        d[k] = *reinterpret_cast<const f32x4*>(win + qoff[k] + (((2 * g + h) ^ qsw[k]) << 2));
      f32x4 P[4], V[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) P[c] = wn_addsgn(d[c], sgn, d[4 + c]);
      V[0] = wn_sub_mfma(P[0], P[2]);
      V[1] = wn_add_mfma(P[1], P[2]);
      V[2] = wn_sub_mfma(P[2], P[1]);
      V[3] = wn_sub_mfma(P[1], P[3]);
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const int step = (kc * 4 + g) * 4 + b;
        if (b & 1) load_frag(step + 1, fe); else load_frag(step + 1, fo);   // next step's weights (L2, 1 KB each)
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
          for (int nb = 0; nb < CBN; ++nb)
            acc[b][nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(V[b][s], (b & 1) ? fo[nb][s] : fe[nb][s], acc[b][nb],
                                                              0, 0, 0);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's DMA of chunk kc + 1 (and the prefetch) landed
    __syncthreads();
  }

  // ---- output transform.  Row reduction over b in registers: R[0] = M0 + M1 + M2, R[1] = M1 - M2 - M3
  float* Rb = smem;   // [a][j][tile][co]: every wave is past its last window read (barrier above)
  // The packed adds below are written as instructions, which the compiler's hazard recogniser does not see as VALU
  // reads of MFMA results: cover the 32x32 MFMA write -> VALU read distance (18 wait states) by hand.
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_nop 15\n\ts_nop 3" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int nb = 0; nb < CBN; ++nb)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = (r & 3) + 8 * (r >> 2) + 4 * h;
      if (r & 1) continue;   // accumulator registers r, r + 1 as one packed pair (tile rows m, m + 1)
      const f32x2 a0{acc[0][nb][r], acc[0][nb][r + 1]}, a1{acc[1][nb][r], acc[1][nb][r + 1]};
      const f32x2 a2{acc[2][nb][r], acc[2][nb][r + 1]}, a3{acc[3][nb][r], acc[3][nb][r + 1]};
      const f32x2 R0 = wn_pk_add(wn_pk_add(a0, a1), a2);
      const f32x2 R1 = wn_pk_sub(wn_pk_sub(a1, a2), a3);
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        Rb[((a * 2 + 0) * 32 + m + e) * WN_CB + nb * 32 + i] = R0[e];
        Rb[((a * 2 + 1) * 32 + m + e) * WN_CB + nb * 32 + i] = R1[e];
      }
    }
  __syncthreads();
  // column reduction over a + epilogue: wave w takes tiles 8w .. 8w+7 (= tile row w).  A lane owns 4 consecutive couts
  // of one tile: 16-byte LDS reads, residual loads and NHWC stores (WN_CB / 4 lanes per tile, 64 / that tiles per pass)
  constexpr int LPT = WN_CB / 4;                  // lanes per tile: 16 (64 couts) or 8 (32 couts)
  constexpr int TPI = 64 / LPT;                   // tiles per pass: 4 or 8
  const int c4 = (lane % LPT) * 4, tsel = lane / LPT;
  const int co = cb * WN_CB + c4;
  const f32x4 bias4 = *reinterpret_cast<const f32x4*>(p.bias + co);
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#pragma unroll
  for (int t8 = 0; t8 < 8; t8 += TPI) {
    const int txo = t8 + tsel;
    const int t = a * 8 + txo;   // tile (tyi = a, txi = txo)
    const int oy0 = by * (2 * WN_TY) + 2 * a, ox0 = bx * (2 * WN_TX) + 2 * txo;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const f32x4 q0 = *reinterpret_cast<const f32x4*>(Rb + ((0 * 2 + j) * 32 + t) * WN_CB + c4);
      const f32x4 q1 = *reinterpret_cast<const f32x4*>(Rb + ((1 * 2 + j) * 32 + t) * WN_CB + c4);
      const f32x4 q2 = *reinterpret_cast<const f32x4*>(Rb + ((2 * 2 + j) * 32 + t) * WN_CB + c4);
      const f32x4 q3 = *reinterpret_cast<const f32x4*>(Rb + ((3 * 2 + j) * 32 + t) * WN_CB + c4);
      const f32x4 y[2] = {wn_add(wn_add(q0, q1), q2), wn_sub(wn_sub(q1, q2), q3)};
#pragma unroll
      for (int ii = 0; ii < 2; ++ii) {
        const int oy = oy0 + ii, ox = ox0 + j;
        const bool ok = oy < p.H && ox < p.W && co < p.Cout;   // Cout is a multiple of 4: a quad is in or out as a whole
        const int m = (n * p.H + oy) * p.W + ox;
        f32x4 v = wn_add(y[ii], bias4);
        if (p.act) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = wn_silu(v[e]);
        }
        if (RES) {
          const unsigned roff = ok ? (unsigned)((m * p.res_ld + p.res_off + co) * 4) : 0x80000000u;
          const f32x4 rv = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rrsrc, roff, 0, 0));
          // element by element: as an f32x4 expression this became v_pk_mul_f32 with the scalar broadcast by op_sel_hi -
          // the one packed-fp32 operand form this library no longer emits (costvolume.hip)
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = (v[e] + rv[e]) * p.post_scale;
        }
        const unsigned off = ok ? (unsigned)((m * p.out_ld + p.out_off + co) * 4) : 0x80000000u;
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), orsrc, off, 0, 0);
      }
    }
  }
#else
  (void)smem; (void)i; (void)h; (void)tyi; (void)txi; (void)cb; (void)n; (void)wy0; (void)wx0;
#endif
}

// Block order over the grid (measured A/B: profiles/r05_wino_order_ab.txt).  Workgroup ids are dealt round-robin to
// the 8 XCDs (each with an L2 of its own) and the block decomposition has the cout block fastest, so in LAUNCH order
// (order 0) the `ncb` workgroups that DMA the same 18 x 10 window sit behind different L2s: the 256-cout head conv0
// pulls 445 MB per launch through its L2s for 119 MB of input (round-4 counters: 3.7 x).  Two XCD-aware orders were
// built: order 1 (every XCD walks a contiguous range of blocks: 247 MB) and order 2 (round-robin over the WINDOWS, all
// cout blocks of a window on one XCD in consecutive slots: 283 MB).  Both make the kernels ~1 % faster ALONE and the
// pipeline 0.6-0.8 % SLOWER under the in-flight loop (1832-1837 against 1844-1847 pairs/s, two runs each on one box):
// the re-fetches of launch order are hits in the memory-side Infinity Cache, not HBM reads, and with four contexts in
// flight the chip-wide walk of launch order co-runs better.  The product ships order 0; the others stay selectable
// in the tools build (ST_WINO_ORDER).  Every order is bijective on [0, nblocks); padding ids exit.
__device__ __forceinline__ bool wn_block_of(unsigned bid, unsigned order, unsigned per_xcd, unsigned ncb, unsigned nblocks,
                                            unsigned& b) {
  if (order == 2) {
    const unsigned slot = bid >> 3;
    b = ((slot / ncb) * 8u + (bid & 7u)) * ncb + slot % ncb;
  } else if (order == 1) {
    b = (bid & 7u) * per_xcd + (bid >> 3);
  } else {
    b = bid;
  }
  return b < nblocks;
}
static unsigned wn_order() {
#ifdef ST_ABLATION
  if (const char* e = getenv("ST_WINO_ORDER")) return (unsigned)atoi(e);
#endif
  return 0u;
}
// grid size of `blocks` workgroups (ncb cout blocks per window) under an order; per_xcd is order 1's range length
static unsigned wn_grid(long long blocks, unsigned ncb, unsigned order, unsigned* per_xcd) {
  *per_xcd = (unsigned)((blocks + 7) / 8);
  if (order == 1) return 8u * *per_xcd;
  if (order == 2) return (unsigned)((blocks / ncb + 7) / 8 * 8 * ncb);
  return (unsigned)blocks;
}

template <int CBN, int LCBN, bool RES, bool KTAIL>
__global__ __launch_bounds__(256, 2) void wino_conv3x3_kernel(const WinoArgs p) {
  unsigned b;
  if (!wn_block_of(blockIdx.x, p.order, p.per_xcd, (unsigned)p.ncb, p.nblocks, b)) return;   // uniform
  wino_conv3x3_body<CBN, LCBN, RES, KTAIL>(p, (int)b);
}

// GROUPED launch: up to WN_GROUP_MAX independent layers of the same instance (same cout blocking, no residual) share
// ONE grid - an array of problem descriptors in the kernel arguments, a workgroup finds its problem by the prefix of
// block counts (a scalar search; every field of the chosen descriptor is wave-uniform and stays in SGPRs).  The YOLOX
// head runs its towers level by level (mmyolo YOLOXHeadModule.forward, configs/_base_/yolox_s_8x8_mmyolo.py:40-51), but
// the three levels are independent: the fused cls|reg conv0 of the 92x160, 46x80 and 23x40 maps are one launch, the
// six second tower convs another.  The small maps' 144 / 288 / 480 workgroups (less than ONE round of the chip's 512
// slots each) ride in the big map's grid instead of costing a launch round of their own: 11 + 12 rounds become 9.6 + 9.6.
constexpr int WN_GROUP_MAX = 6;
struct WinoGroupArgs {
  WinoArgs p[WN_GROUP_MAX];
  unsigned first[WN_GROUP_MAX + 1];   // first[k] = first grid id of problem k (a multiple of 8); first[n] = grid size
  int n;
};

template <int CBN, int LCBN>
__global__ __launch_bounds__(256, 2) void wino_conv3x3_group_kernel(const WinoGroupArgs g) {
  const unsigned bid = blockIdx.x;
  int k = 0;
#pragma unroll
  for (int q = 1; q < WN_GROUP_MAX; ++q) k += (q < g.n && bid >= g.first[q]) ? 1 : 0;
  // every problem owns a grid range of its own that starts at a multiple of 8: the local id keeps the block's XCD
  unsigned b;
  if (!wn_block_of(bid - g.first[k], g.p[k].order, g.p[k].per_xcd, (unsigned)g.p[k].ncb, g.p[k].nblocks, b)) return;
  wino_conv3x3_body<CBN, LCBN, false, false>(g.p[k], (int)b);
}


#ifdef ST_ABLATION
// ---- PERSISTENT form of the same convolution (tile variant 57, round 6; TOOLS BUILD ONLY) -------------------------------
// Measured (tools/wino_persist_bench.py, profiles/r06_wino_persist_ab.txt): bit-identical to variant 43 and NOT faster - 0.975 to
// 1.035 x on the path's eight layer shapes.  The per-workgroup prologue is therefore not what holds these launches at an
// MFMA-pipe busy of 0.44-0.53; what the tail kernel removed and this form keeps is the transformed-weight stream (1 KB per
// 4 MFMAs and wave from L2, a wait per step).  Kept as the yardstick of that statement, not shipped.
// What the counters said about every non-head Winograd launch (MFMA-pipe busy 0.44-0.53, profiles/r06_mfma_busy.txt): a
// workgroup runs two to four K-chunks between a prologue (index set-up, the first window and weight fetch, exposed) and an
// epilogue, and dies.  Here two workgroups per CU LOOP over the tile blocks: the NEXT block's first window is requested by
// LDS-DMA at the top of the current block's LAST K-chunk (into the window buffer that chunk does not read) and its first
// weight fragment in front of the output transform, so a block starts with its operands in place.  For that the transform
// exchange cannot alias the window buffers any more: it has 32 KB of its own and the 64-cout workgroup transforms its two
// cout blocks one after the other (two more barriers per block).  LDS 2 x 23.5 + 32 = 79 KB: still two workgroups per CU.
// Per accumulator the MFMA sequence, and per output the transform's additions, are those of wino_conv3x3_body: the results
// are BIT-IDENTICAL to tile variant 43 (tests/test_conv_gpu.py).
constexpr int wnp_lds_floats() { return 2 * WN_WIN_FLOATS + 4 * 2 * 32 * 32; }

template <int CBN, int LCBN, bool RES, bool KTAIL>
__global__ __launch_bounds__(256, 2) void wino_conv3x3_persist_kernel(const WinoArgs p) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int WN_FRAG_FLOATS = LCBN * 64 * 4;
  constexpr int SPLIT = LCBN / CBN;
  extern __shared__ float4 wn_smem4[];
  float* smem = reinterpret_cast<float*>(wn_smem4);
  float* Rb = smem + 2 * WN_WIN_FLOATS;          // [a][j][tile][32]: one cout block of 32 at a time
  const int tid = threadIdx.x, lane = tid & 63, a = tid >> 6;
  const int i = lane & 31, h = lane >> 5;
  const int tyi = i >> 3, txi = i & 7;
  const __amdgpu_buffer_rsrc_t irsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.in), 0, (int)p.in_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t wrsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.wino), 0, (int)p.wino_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, (int)p.out_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rrsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(RES ? p.res : p.in), 0, (int)(RES ? p.res_bytes : 0u), 0x00020000);
  constexpr int NPW = (WN_PIECES + 3) / 4;

  // window geometry of a block: byte offsets of this lane's DMA pieces (as wino_conv3x3_body)
  auto block_of = [&](unsigned b, int& cb, int& bx, int& by, int& n) {
    cb = (int)(b % (unsigned)p.ncb); b /= (unsigned)p.ncb;
    bx = (int)(b % (unsigned)p.tbx); b /= (unsigned)p.tbx;
    by = (int)(b % (unsigned)p.tby);
    n = (int)(b / (unsigned)p.tby);
  };
  unsigned tail_ok = 0;
#pragma unroll
  for (int k = 0; k < NPW; ++k) {
    const int j = a + 4 * k;
    const int q = 8 * j + (lane >> 3), sl = lane & 7;
    const int quad = sl ^ ((q >> 1) & 7);
    if ((p.nkc - 1) * 32 + 4 * quad < p.Cin) tail_ok |= 1u << k;
  }
  auto window_offsets = [&](int bx, int by, int n, unsigned (&poff)[NPW]) {
    const int wy0 = by * (2 * WN_TY) - 1, wx0 = bx * (2 * WN_TX) - 1;
#pragma unroll
    for (int k = 0; k < NPW; ++k) {
      const int j = a + 4 * k;
      const int q = 8 * j + (lane >> 3), sl = lane & 7;
      const int wr = q / WN_WW, wc = q - wr * WN_WW;
      const int y = wy0 + wr, x = wx0 + wc;
      const bool ok = j < WN_PIECES && q < WN_PIX && y >= 0 && y < p.H && x >= 0 && x < p.W;
      const int quad = sl ^ ((q >> 1) & 7);
      poff[k] = ok ? (unsigned)((((n * p.H + y) * p.W + x) * p.in_ld + p.in_off + 4 * quad) * 4) : 0x80000000u;
    }
  };
  auto dma_window = [&](const unsigned (&poff)[NPW], int kc, int buf) {
    const bool last = kc == p.nkc - 1;
#pragma unroll
    for (int k = 0; k < NPW; ++k) {
      const int j = a + 4 * k;
      const unsigned off = (last && !((tail_ok >> k) & 1u)) ? 0x80000000u : poff[k];
      if (j < WN_PIECES)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(
            irsrc, (__attribute__((address_space(3))) void*)(smem + buf * WN_WIN_FLOATS + j * 256), 16, off, kc * 128, 0, 0);
    }
  };
  const int r0 = a == 0 ? 0 : (a == 2 ? 2 : 1);
  const int r1 = a == 0 ? 2 : (a == 1 ? 2 : (a == 2 ? 1 : 3));
  const f32x2 sgn = a == 1 ? f32x2{1.0f, 1.0f} : f32x2{-1.0f, -1.0f};
  // patch pixel (row r, column c) of tile (tyi, txi): window pixel q = (2 tyi + r) * 18 + 2 txi + c, float offset 32 q, channel
  // slot (2g + h) ^ ((q >> 1) & 7); q(c = 0) is even, so columns (0, 1) and (2, 3) share a swizzle term and
  // (2g + h) ^ sw = 2g ^ (h ^ sw): two row bases + four terms instead of 8 + 8 registers held through the block loop
  const int qb0 = (2 * tyi + r0) * WN_WW + 2 * txi, qb1 = (2 * tyi + r1) * WN_WW + 2 * txi;
  const int hs00 = (h ^ ((qb0 >> 1) & 7)) << 2, hs01 = (h ^ (((qb0 >> 1) + 1) & 7)) << 2;
  const int hs10 = (h ^ ((qb1 >> 1) & 7)) << 2, hs11 = (h ^ (((qb1 >> 1) + 1) & 7)) << 2;
  const int last_step = p.nkc * 16 - 1;
  auto wbase_of = [&](int cb) {
    return (unsigned)(((((cb / SPLIT) * 4 + a) * p.nkc) * 16) * WN_FRAG_FLOATS + (cb % SPLIT) * CBN * 256 + lane * 4) * 4u;
  };
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

  unsigned blk = blockIdx.x;
  if (blk >= p.nblocks) return;
  int cb, bx, by, n;
  block_of(blk, cb, bx, by, n);
  unsigned poff[NPW];
  window_offsets(bx, by, n, poff);
  unsigned wbase = wbase_of(cb);
  int b0 = 0;                                  // window buffer of the current block's chunk 0
  dma_window(poff, 0, 0);
  f32x4 fr[4][CBN];                            // fragment ring: step s lives in slot s & 3, requested three steps ahead
  auto load_frag = [&](unsigned wb, int step, f32x4 (&f)[CBN]) {
    const int soff = (step < last_step ? step : last_step) * (WN_FRAG_FLOATS * 4);
#pragma unroll
    for (int nb = 0; nb < CBN; ++nb)
      f[nb] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, wb, soff + 1024 * nb, 0));
  };
  load_frag(wbase, 0, fr[0]);
  load_frag(wbase, 1, fr[1]);
  load_frag(wbase, 2, fr[2]);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  for (;;) {
    const unsigned nxt = blk + gridDim.x;
    const bool has_next = nxt < p.nblocks;     // uniform
    int cbn = 0, bxn = 0, byn = 0, nn = 0;
    f32x16 acc[4][CBN];
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int nb = 0; nb < CBN; ++nb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[b][nb][r] = 0.f;

    for (int kc = 0; kc < p.nkc; ++kc) {
      const int buf = (kc + b0) & 1;
      if (kc + 1 < p.nkc) {
        dma_window(poff, kc + 1, buf ^ 1);
      } else if (has_next) {                   // the next block's first window: lands during this block's last chunk
        block_of(nxt, cbn, bxn, byn, nn);
        unsigned poffn[NPW];                   // (not kept: the block recomputes its offsets when it starts)
        window_offsets(bxn, byn, nn, poffn);
        dma_window(poffn, 0, buf ^ 1);
      }
      const float* win = smem + buf * WN_WIN_FLOATS;
      const int gn = (KTAIL && kc == p.nkc - 1) ? p.g_last : 4;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        if (KTAIL && g >= gn) continue;
        f32x4 d[8];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          d[c] = *reinterpret_cast<const f32x4*>(win + qb0 * 32 + 32 * c + ((8 * g) ^ (c < 2 ? hs00 : hs01)));
          d[4 + c] = *reinterpret_cast<const f32x4*>(win + qb1 * 32 + 32 * c + ((8 * g) ^ (c < 2 ? hs10 : hs11)));
        }
        f32x4 P[4], V[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) P[c] = wn_addsgn(d[c], sgn, d[4 + c]);
        V[0] = wn_sub_mfma(P[0], P[2]);
        V[1] = wn_add_mfma(P[1], P[2]);
        V[2] = wn_sub_mfma(P[2], P[1]);
        V[3] = wn_sub_mfma(P[1], P[3]);
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          const int step = (kc * 4 + g) * 4 + b;
          load_frag(wbase, step + 3, fr[(b + 3) & 3]);      // (16 steps per chunk: the slot of a step is b & 3)
#pragma unroll
          for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int nb = 0; nb < CBN; ++nb)
              acc[b][nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(V[b][s], fr[b & 3][nb][s], acc[b][nb], 0, 0, 0);
        }
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    }
    // the next block's first weight fragment travels under the output transform
    unsigned wbasen = wbase;
    if (has_next) {
      wbasen = wbase_of(cbn);
      load_frag(wbasen, 0, fr[0]);
      load_frag(wbasen, 1, fr[1]);
      load_frag(wbasen, 2, fr[2]);
    }

    // ---- output transform, one cout block of 32 at a time
    const int c4 = (lane & 7) * 4, txo = lane >> 3;
    const int t = a * 8 + txo;
    const int oy0 = by * (2 * WN_TY) + 2 * a, ox0 = bx * (2 * WN_TX) + 2 * txo;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_nop 15\n\ts_nop 3" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int nb = 0; nb < CBN; ++nb) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = (r & 3) + 8 * (r >> 2) + 4 * h;
        if (r & 1) continue;
        const f32x2 a0{acc[0][nb][r], acc[0][nb][r + 1]}, a1{acc[1][nb][r], acc[1][nb][r + 1]};
        const f32x2 a2{acc[2][nb][r], acc[2][nb][r + 1]}, a3{acc[3][nb][r], acc[3][nb][r + 1]};
        const f32x2 R0 = wn_pk_add(wn_pk_add(a0, a1), a2);
        const f32x2 R1 = wn_pk_sub(wn_pk_sub(a1, a2), a3);
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          Rb[((a * 2 + 0) * 32 + m + e) * 32 + i] = R0[e];
          Rb[((a * 2 + 1) * 32 + m + e) * 32 + i] = R1[e];
        }
      }
      __syncthreads();
      const int co = cb * (32 * CBN) + nb * 32 + c4;
      const f32x4 bias4 = *reinterpret_cast<const f32x4*>(p.bias + co);
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const f32x4 q0 = *reinterpret_cast<const f32x4*>(Rb + ((0 * 2 + j) * 32 + t) * 32 + c4);
        const f32x4 q1 = *reinterpret_cast<const f32x4*>(Rb + ((1 * 2 + j) * 32 + t) * 32 + c4);
        const f32x4 q2 = *reinterpret_cast<const f32x4*>(Rb + ((2 * 2 + j) * 32 + t) * 32 + c4);
        const f32x4 q3 = *reinterpret_cast<const f32x4*>(Rb + ((3 * 2 + j) * 32 + t) * 32 + c4);
        const f32x4 y[2] = {wn_add(wn_add(q0, q1), q2), wn_sub(wn_sub(q1, q2), q3)};
#pragma unroll
        for (int ii = 0; ii < 2; ++ii) {
          const int oy = oy0 + ii, ox = ox0 + j;
          const bool ok = oy < p.H && ox < p.W && co < p.Cout;
          const int m = (n * p.H + oy) * p.W + ox;
          f32x4 v = wn_add(y[ii], bias4);
          if (p.act) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = wn_silu(v[e]);
          }
          if (RES) {
            const unsigned roff = ok ? (unsigned)((m * p.res_ld + p.res_off + co) * 4) : 0x80000000u;
            const f32x4 rv = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rrsrc, roff, 0, 0));
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = (v[e] + rv[e]) * p.post_scale;
          }
          const unsigned off = ok ? (unsigned)((m * p.out_ld + p.out_off + co) * 4) : 0x80000000u;
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), orsrc, off, 0, 0);
        }
      }
      if (nb + 1 < CBN) __syncthreads();       // the next cout block's exchange overwrites Rb
    }
    if (!has_next) break;
    blk = nxt; cb = cbn; bx = bxn; by = byn; n = nn;
    if (p.nkc > 1) window_offsets(bx, by, n, poff);   // for the chunks 1.. of the block (uniform)
    wbase = wbasen;
    b0 = (b0 + p.nkc) & 1;
    // (the next block's exchange writes come behind its K loop's barriers; its first window was published by the barrier
    // that closed this block's last chunk)
  }
#endif
}

#endif  // ST_ABLATION

}  // namespace

// cout blocks of 32 per workgroup for a layer: 2 (64 couts) when Cout is a multiple of 64 or fits one padded block of
// 64 (33..63, e.g. the 48-level aggregation convs), 1 (32 couts) for the other multiples of 32; 0 = not a Winograd layer
int wino_cbn(int Cout) {
  if (Cout <= 0) return 0;
  if (Cout % 64 == 0) return 2;
  if (Cout % 32 == 0) return 1;
  if (Cout > 32 && Cout < 64) return 2;
  return 0;
}
bool wino_shape_ok(int Cin, int Cout) { return Cin >= 16 && Cin % 4 == 0 && wino_cbn(Cout) != 0; }

// floats of the fragment-ordered transformed weights of a Cout x Cin 3x3 conv (Cout padded to the workgroup's cout
// block, Cin to the K-chunk of 32)
size_t wino_packed_floats(int Cout, int Cin) {
  const int cb = 32 * wino_cbn(Cout);
  return cb ? (size_t)16 * round_up(Cout, cb) * round_up(Cin, 32) : 0;
}

// packed: the direct kernels' folded fp32 weights [CoutPad32][Kpad], K index = (kh*3+kw)*Cin + ci (host memory).
// out: [cb][a][kc][g][b][nb][lane][4] with lane = (h << 5) | j: U_{a,b}[co = cb*CB + nb*32 + j][ci = kc*32 + 8g + 4h + e]
// (zero for co >= Cout or ci >= Cin), U = G g G^T evaluated in fp64 on the fp32 weights and rounded once.
int wino_pack_weights(const float* packed, int Cout, int Cin, float* out) {
  ST_REQUIRE(packed && out && wino_shape_ok(Cin, Cout), "wino_pack_weights: not a Winograd layer shape (Cin %d, Cout %d)",
             Cin, Cout);
  const int cbn = wino_cbn(Cout), CB = 32 * cbn;
  const int Kpad = round_up(9 * Cin, 32), ncb = round_up(Cout, CB) / CB, nkc = round_up(Cin, 32) / 32;
  static const double G[4][3] = {{1, 0, 0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0, 0, 1}};
  for (int cb = 0; cb < ncb; ++cb)
    for (int a = 0; a < 4; ++a)
      for (int kc = 0; kc < nkc; ++kc)
        for (int g = 0; g < 4; ++g)
          for (int b = 0; b < 4; ++b)
            for (int nb = 0; nb < cbn; ++nb)
              for (int l = 0; l < 64; ++l)
                for (int e = 0; e < 4; ++e) {
                  const int j = l & 31, h = l >> 5;
                  const int co = cb * CB + nb * 32 + j, ci = kc * 32 + 8 * g + 4 * h + e;
                  double u = 0.0;
                  if (co < Cout && ci < Cin)
                    for (int kh = 0; kh < 3; ++kh)
                      for (int kw = 0; kw < 3; ++kw)
                        u += G[a][kh] * (double)packed[(size_t)co * Kpad + (kh * 3 + kw) * Cin + ci] * G[b][kw];
                  out[(((((((size_t)cb * 4 + a) * nkc + kc) * 4 + g) * 4 + b) * cbn + nb) * 64 + l) * 4 + e] = (float)u;
                }
  return ST_OK;
}

// Shapes: 3x3 / stride 1 / pad 1, Cin a multiple of 4 (>= 16), Cout a multiple of 32 or one padded block of 64, one
// output tensor (no split / upsample), transformed weights present, every tensor below 2 GiB.
bool wino_conv_applicable(const StConvDesc& d) {
  if (!d.wgt_wino_dev) return false;
  if (d.KH != 3 || d.KW != 3 || d.stride != 1 || d.pad != 1 || d.up_dev || d.out2_dev) return false;
  if (!wino_shape_ok(d.Cin, d.Cout)) return false;
  if ((d.in_ld | d.in_off | d.out1_ld | d.out1_off) & 3) return false;
  if ((reinterpret_cast<uintptr_t>(d.in_dev) | reinterpret_cast<uintptr_t>(d.out1_dev)) & 15) return false;
  if (d.res_dev && (((d.res_ld | d.res_off) & 3) || (reinterpret_cast<uintptr_t>(d.res_dev) & 15))) return false;
  if (d.Cout % 4) return false;
  const long long M = (long long)d.N * d.Hi * d.Wi, lim = 1ll << 31;
  if (M * d.in_ld * 4 >= lim || M * d.out1_ld * 4 >= lim) return false;
  if (d.res_dev && M * d.res_ld * 4 >= lim) return false;
  return true;
}

template <int CBN, int LCBN, bool RES, bool KTAIL = false>
static int wino_launch_instance(const WinoArgs& a, unsigned blocks, hipStream_t stream) {
  constexpr int lds = wn_lds_floats(CBN) * (int)sizeof(float);
  static int lds_set = 0;
  auto kern = wino_conv3x3_kernel<CBN, LCBN, RES, KTAIL>;
  ST_ENSURE_DYNAMIC_LDS(kern, lds, lds_set);
  hipLaunchKernelGGL(kern, dim3(a.grid), dim3(256), lds, stream, a);
  return ST_OK;
}

static int wino_fill_args(const StConvDesc& d, bool narrow, WinoArgs& a, long long* blocks_out) {
  ST_REQUIRE(wino_conv_applicable(d), "winograd conv: shape not supported (3x3 s1 p1, Cin % 4 == 0, Cout a multiple of 32 "
                                      "or <= 64, transformed weights required)");
  ST_REQUIRE(d.in_dev && d.bias_dev && d.out1_dev, "winograd conv: null pointer");
  ST_REQUIRE(d.in_off + d.Cin <= d.in_ld && d.out1_off + d.Cout <= d.out1_ld, "winograd conv: channel slice exceeds ld");
  if (d.res_dev) ST_REQUIRE(d.res_off + d.Cout <= d.res_ld, "winograd conv: res slice exceeds res_ld");
  const long long M = (long long)d.N * d.Hi * d.Wi;
  const int lcbn = wino_cbn(d.Cout);
  ST_REQUIRE(!narrow || (lcbn == 2 && d.Cout % 64 == 0), "winograd conv: the narrow instance needs Cout %% 64 == 0");
  const int cbn = narrow ? 1 : lcbn, CB = 32 * cbn;
  a.in = d.in_dev; a.wino = d.wgt_wino_dev; a.bias = d.bias_dev; a.out = d.out1_dev; a.res = d.res_dev;
  a.N = d.N; a.H = d.Hi; a.W = d.Wi; a.Cin = d.Cin; a.in_ld = d.in_ld; a.in_off = d.in_off; a.Cout = d.Cout;
  a.out_ld = d.out1_ld; a.out_off = d.out1_off; a.res_ld = d.res_ld; a.res_off = d.res_off;
  a.post_scale = d.res_dev ? d.post_scale : 1.0f;
  a.act = d.act;
  a.tbx = ceil_div(d.Wi, 2 * WN_TX); a.tby = ceil_div(d.Hi, 2 * WN_TY);
  a.ncb = round_up(d.Cout, CB) / CB; a.nkc = round_up(d.Cin, 32) / 32;
  a.in_bytes = (unsigned)(M * d.in_ld * 4);
  a.out_bytes = (unsigned)(M * d.out1_ld * 4);
  a.res_bytes = d.res_dev ? (unsigned)(M * d.res_ld * 4) : 0u;
  a.wino_bytes = (unsigned)(wino_packed_floats(d.Cout, d.Cin) * 4);
  a.g_last = 4 - (a.nkc * 32 - d.Cin) / 8;          // whole groups of 8 padded channels are skipped
  const long long blocks = (long long)d.N * a.tbx * a.tby * a.ncb;
  ST_REQUIRE(blocks + 8 < (1ll << 31), "winograd conv: grid too large");
  a.nblocks = (unsigned)blocks;
  a.order = wn_order();
  a.grid = wn_grid(blocks, (unsigned)a.ncb, a.order, &a.per_xcd);
  *blocks_out = blocks;
  return ST_OK;
}

// narrow = true: 32-cout workgroups on a 64-cout layout (tile variant 44; only where the layout has 2 blocks)
int wino_conv_launch(const StConvDesc& d, hipStream_t stream, bool narrow) {
  WinoArgs a;
  long long blocks = 0;
  ST_CHECK(wino_fill_args(d, narrow, a, &blocks));
  const int lcbn = wino_cbn(d.Cout);
  const int cbn = narrow ? 1 : lcbn;
  const bool ktail = a.g_last < 4;
  int rc;
  if (ktail && cbn == 2 && !d.res_dev) rc = wino_launch_instance<2, 2, false, true>(a, (unsigned)blocks, stream);
  else if (ktail && cbn == 1 && lcbn == 1 && !d.res_dev) rc = wino_launch_instance<1, 1, false, true>(a, (unsigned)blocks, stream);
  else if (cbn == 2) rc = d.res_dev ? wino_launch_instance<2, 2, true>(a, (unsigned)blocks, stream)
                               : wino_launch_instance<2, 2, false>(a, (unsigned)blocks, stream);
  else if (lcbn == 2) rc = d.res_dev ? wino_launch_instance<1, 2, true>(a, (unsigned)blocks, stream)
                                     : wino_launch_instance<1, 2, false>(a, (unsigned)blocks, stream);
  else rc = d.res_dev ? wino_launch_instance<1, 1, true>(a, (unsigned)blocks, stream)
                      : wino_launch_instance<1, 1, false>(a, (unsigned)blocks, stream);
  ST_CHECK(rc);
  ST_CHECK_HIP(hipGetLastError());
  return ST_OK;
}


#ifdef ST_ABLATION
// tile variant 57: the persistent form (64-cout layout only: Cout % 64 == 0 or one padded block of 64)
bool wino_persist_applicable(const StConvDesc& d) { return wino_conv_applicable(d) && wino_cbn(d.Cout) == 2; }

template <int CBN, int LCBN, bool RES, bool KTAIL>
static int wino_persist_instance(const WinoArgs& a, unsigned grid, hipStream_t stream) {
  constexpr int lds = wnp_lds_floats() * (int)sizeof(float);
  static int lds_set = 0;
  auto kern = wino_conv3x3_persist_kernel<CBN, LCBN, RES, KTAIL>;
  ST_ENSURE_DYNAMIC_LDS(kern, lds, lds_set);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, stream, a);
  return ST_OK;
}

int wino_persist_launch(const StConvDesc& d, hipStream_t stream) {
  ST_REQUIRE(wino_persist_applicable(d), "winograd conv (persistent): needs a Winograd layer with the 64-cout layout");
  WinoArgs a;
  long long blocks = 0;
  ST_CHECK(wino_fill_args(d, false, a, &blocks));
  static int cus = 0;
  if (cus == 0) {
    int dev = 0, n = 0;
    ST_CHECK_HIP(hipGetDevice(&dev));
    ST_CHECK_HIP(hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev));
    cus = std::max(1, n);
  }
  const unsigned grid = (unsigned)std::min<long long>(blocks, 2ll * cus);
  const bool ktail = a.g_last < 4;
  int rc;
  if (ktail) rc = d.res_dev ? wino_persist_instance<2, 2, true, true>(a, grid, stream)
                            : wino_persist_instance<2, 2, false, true>(a, grid, stream);
  else rc = d.res_dev ? wino_persist_instance<2, 2, true, false>(a, grid, stream)
                      : wino_persist_instance<2, 2, false, false>(a, grid, stream);
  ST_CHECK(rc);
  ST_CHECK_HIP(hipGetLastError());
  return ST_OK;
}

#endif  // ST_ABLATION

// Can these layers share one grouped launch?  The wide instance <2, 2>, no residual, no K tail, at most WN_GROUP_MAX.
bool wino_group_applicable(const StConvDesc* d, int n) {
  if (n < 2 || n > WN_GROUP_MAX) return false;
  for (int k = 0; k < n; ++k)
    if (!wino_conv_applicable(d[k]) || d[k].res_dev || d[k].Cout % 64 != 0 || d[k].Cin % 32 != 0) return false;
  return true;
}

int wino_group_launch(const StConvDesc* d, int n, hipStream_t stream) {
  ST_REQUIRE(wino_group_applicable(d, n), "winograd group: layers do not qualify (3x3 s1 p1, Cout %% 64 == 0, Cin %% 32 "
                                          "== 0, no residual, 2..%d layers)", WN_GROUP_MAX);
  WinoGroupArgs g;
  std::memset(&g, 0, sizeof(g));
  g.n = n;
  long long total = 0;
  for (int k = 0; k < n; ++k) {
    long long blocks = 0;
    ST_CHECK(wino_fill_args(d[k], false, g.p[k], &blocks));
    g.first[k] = (unsigned)total;
    total += (g.p[k].grid + 7u) / 8u * 8u;       // the next problem starts at a multiple of 8 (launch order: pad, ids exit)
    ST_REQUIRE(total + 8 < (1ll << 31), "winograd group: grid too large");
  }
  for (int k = n; k <= WN_GROUP_MAX; ++k) g.first[k] = (unsigned)total;
  constexpr int lds = wn_lds_floats(2) * (int)sizeof(float);
  static int lds_set = 0;
  auto kern = wino_conv3x3_group_kernel<2, 2>;
  ST_ENSURE_DYNAMIC_LDS(kern, lds, lds_set);
  hipLaunchKernelGGL(kern, dim3((unsigned)total), dim3(256), lds, stream, g);
  ST_CHECK_HIP(hipGetLastError());
  return ST_OK;
}

}  // namespace st

extern "C" size_t st_wino_packed_floats(int Cout, int Cin) { return st::wino_packed_floats(Cout, Cin); }

extern "C" int st_wino_pack_weights(const float* packed_wgt_host, int Cout, int Cin, float* out_host) {
  return st::wino_pack_weights(packed_wgt_host, Cout, Cin, out_host);
}
