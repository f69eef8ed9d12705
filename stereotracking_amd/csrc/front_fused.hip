// Fused head of a stage-1 CSP branch on gfx950:
//     3x3 / stride-2 ConvModule (32 -> 64)  ->  CSPLayer main_conv | short_conv (1x1, 64 -> 32 | 32)
//                                            ->  DarknetBottleneck conv1 (1x1, 32 -> 32) on the main half
// i.e. `stage1.0` + the first three convolutions of `stage1.1` of both branches of the two-branch backbone
// (reference mmtrack/models/backbones/csp_darknet_disparity_v1.py:113-153: ConvModule(c1, c2, 3, stride 2) followed by
// mmdet CSPLayer(c2, c2, n, add_identity)), at the one place of the network where they are HBM-bound: 184x320 pixels
// x 16 images, 32-64 channels (16 flop/B for the 1x1 convs).  Unfused this is three launches that write and re-read
// the 64-channel stride-2 output and the main half (135 MB per image through HBM); fused, the 64-channel tensor never
// exists: a wave computes it for 16 pixels in MFMA accumulators and feeds it straight into the two 1x1 convs (the
// swapped-operand accumulator layout D[cout][pixel] IS the B-operand layout of the next MFMA, the trick of
// pointwise_conv.hip's CHAIN mode).  Outputs: main (32 ch, the bottleneck's residual), short (32 ch, into the CSP
// concat buffer), conv1(main) (32 ch, the input of the 3x3 bottleneck conv).
//
// Structure: ONE persistent workgroup of 8 waves per CU.  Every weight of the three convolutions lives in LDS for the
// whole kernel (3x3: 72 KB, XOR-swizzled rows so the A-fragment ds_read_b128 are conflict-free; 1x1: 20 KB in MFMA
// fragment order), loaded once by LDS-DMA; after that single barrier the waves never synchronise again.  A wave
// walks over tiles of 16 output pixels of one row (XCD-contiguous order: the 32 CUs of an XCD share input rows in
// their L2).  The B operands (pixels) do not go through LDS at all: a lane's fragment for tap (ky, kx) and channel
// group g is 16 contiguous bytes of the NHWC input, so the 18 fragments of a tile are 18 `buffer_load_dwordx4`
// (range-checked descriptor: out-of-image = conv padding = zeros), issued one tile AHEAD into the registers the
// current tile has just finished with - a full tile of MFMAs (~5 us) hides the HBM latency, and no window staging,
// barrier or LDS footprint per tile is left.  `v_mfma_f32_16x16x4_f32`, A = weights, B = pixels.  Same arithmetic as
// the three separate launches up to fp32 summation order.
#include <algorithm>
#include <cstdint>
#include <cstdlib>

#include "st_common.h"

namespace st {
namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int FF_CIN = 32, FF_C2 = 64, FF_MID = 32;
constexpr int FF_WAVES = 8, FF_THREADS = 64 * FF_WAVES;   // measured: 8 waves 422 us, 12 waves 430 us, 16 waves 452 us
constexpr int FF_W3_FLOATS = 9 * FF_C2 * FF_CIN;      // [tap][cout][32 ch], 16-byte quads of a row XOR-swizzled
constexpr int FF_MS_FLOATS = 2 * FF_MID * FF_C2;      // [c2 4][c 4][lane][4]
constexpr int FF_C1_FLOATS = FF_MID * FF_MID;         // [c3 2][c2 2][lane][4]
constexpr int FF_BIAS_FLOATS = 256;                   // a 64 | ms 64 | c1 32 (+ pad)
constexpr int FF_LDS_FLOATS = FF_W3_FLOATS + FF_MS_FLOATS + FF_C1_FLOATS + FF_BIAS_FLOATS;

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
  int tiles_x;             // 16-pixel tiles per output row
  int ntiles;              // N * Ho * tiles_x
  int slots_per_xcd;       // wave slots (workgroups x 8) of one XCD; gridDim.x is a multiple of 8
  unsigned in_bytes, wgt_bytes, main_bytes, short_bytes, tmp_bytes;
};

__device__ __forceinline__ float ff_silu(float v) { return v * __builtin_amdgcn_rcpf(1.0f + __expf(-v)); }

__global__ __launch_bounds__(FF_THREADS, 1) void front_s2_csp_kernel(const FrontArgs p) {
  extern __shared__ float4 ff_smem4[];
  float* w3 = reinterpret_cast<float*>(ff_smem4);
  float* wms = w3 + FF_W3_FLOATS;
  float* wc1 = wms + FF_MS_FLOATS;
  float* bias = wc1 + FF_C1_FLOATS;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform: tile bookkeeping runs on the scalar unit
  const int i16 = lane & 15, kq = lane >> 4;

#if defined(__HIP_DEVICE_COMPILE__)  // device-only builtins; the host pass only needs the kernel stub
  const __amdgpu_buffer_rsrc_t irsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.in), 0, (int)p.in_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t wrsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.wgt_a), 0, (int)p.wgt_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t mfrsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.frag_ms), 0, FF_MS_FLOATS * 4, 0x00020000);
  const __amdgpu_buffer_rsrc_t cfrsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.frag_c1), 0, FF_C1_FLOATS * 4, 0x00020000);

  // ---- one-time fill of the weight image.  LDS slot e (16 bytes) of the 3x3 part = (tap, cout, stored quad); the
  // quad it holds is (stored ^ ((cout >> 1) & 7)): with 128-byte rows, 16 lanes reading the same quad of 16
  // consecutive couts would hit 2 x 8 banks 8 times over; swizzled they cover all 64 banks once.
#pragma unroll
  for (int j = 0; j < (FF_W3_FLOATS / 4 + FF_THREADS - 1) / FF_THREADS; ++j) {
    const int e = tid + FF_THREADS * j;
    if (j * FF_THREADS + wave * 64 >= FF_W3_FLOATS / 4) break;   // whole waves only: 64 divides every part
    const int tap = e >> 9, co = (e >> 3) & 63, qd = e & 7;
    const unsigned off = (unsigned)((co * (9 * FF_CIN) + tap * FF_CIN + 4 * (qd ^ ((co >> 1) & 7))) * 4);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(
        wrsrc, (__attribute__((address_space(3))) void*)(w3 + (j * FF_THREADS + wave * 64) * 4), 16, off, 0, 0, 0);
  }
#pragma unroll
  for (int j = 0; j < (FF_MS_FLOATS / 4 + FF_THREADS - 1) / FF_THREADS; ++j) {
    if (j * FF_THREADS + wave * 64 >= FF_MS_FLOATS / 4) break;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(
        mfrsrc, (__attribute__((address_space(3))) void*)(wms + (j * FF_THREADS + wave * 64) * 4), 16,
        (unsigned)((tid + FF_THREADS * j) * 16), 0, 0, 0);
  }
  if (wave < FF_C1_FLOATS / 256)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(cfrsrc, (__attribute__((address_space(3))) void*)(wc1 + wave * 256), 16,
                                             (unsigned)(tid * 16), 0, 0, 0);
  if (tid < 64) bias[tid] = p.bias_a[tid];
  else if (tid < 128) bias[tid] = p.bias_ms[tid - 64];
  else if (tid < 160) bias[tid] = p.bias_c1[tid - 128];

  // ---- per-lane constants
  const int sw = (i16 >> 1) & 7;
  // A fragment (cb, tap, g): couts cb*16 + i16, channels 16g + 4kq..: float offset tap*2048 + cb*512 + wq[g]
  int wq[2];
#pragma unroll
  for (int g = 0; g < 2; ++g) wq[g] = i16 * 32 + (((4 * g + kq) ^ sw) << 2);

  // wave slot -> tiles: slot order is XCD-major (workgroup b runs on XCD b % 8), so in every sweep the 32 CUs of an
  // XCD take a contiguous run of tiles (a band of ~13 output rows whose input rows they share in that XCD's L2)
  const int xcd = blockIdx.x & 7, wg = blockIdx.x >> 3;
  const int slot = xcd * p.slots_per_xcd + wg * FF_WAVES + wave;
  const int stride = 8 * p.slots_per_xcd;

  // pixel-fragment loads of tile t into X[2*tap + g]; tiles past the end (and everything outside the image) read zeros.
  // Per lane: one byte offset per kx (column 2ox - 1 + kx of input row 2oy - 1, channel 4kq; 0x80000000 = does not
  // exist -> the descriptor's range check returns zeros); the row / channel-group displacement of a tap is
  // wave-uniform and rides in the scalar offset.
  unsigned xcol[3] = {0x80000000u, 0x80000000u, 0x80000000u};
  unsigned rowmask = 0;              // bit ky set = input row 2oy - 1 + ky exists
  auto locate = [&](int t) {
    const bool live = t < p.ntiles;
    const int tx = t % p.tiles_x;
    const int r = t / p.tiles_x;     // n * Ho + oy
    const int oy = r % p.Ho, n = r / p.Ho;
    const int ox = tx * 16 + i16;
    const int gy0 = 2 * oy - 1, gx0 = 2 * ox - 1;
    const unsigned base = (unsigned)((((n * p.Hi + gy0) * p.Wi + gx0) * p.in_ld + p.in_off + 4 * kq) * 4);
    rowmask = 0;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      rowmask |= (live && gy0 + k >= 0 && gy0 + k < p.Hi) ? 1u << k : 0u;
      xcol[k] = (ox < p.Wo && gx0 + k >= 0 && gx0 + k < p.Wi) ? base + (unsigned)(k * p.in_ld * 4) : 0x80000000u;
    }
  };
  const int row_bytes = p.Wi * p.in_ld * 4;
  auto xload = [&](int tap, int g) -> f32x4 {
    const int ky = tap / 3, kx = tap - 3 * ky;
    const unsigned off = ((rowmask >> ky) & 1u) ? xcol[kx] : 0x80000000u;
    // a row of -1 makes base "negative" (wraps): only ever used with ky >= 1, where base + ky * row_bytes is back in range
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(irsrc, off + (unsigned)(ky * row_bytes + 64 * g), 0, 0));
  };

  f32x4 X[18];
  int tile = slot;
  locate(tile);
#pragma unroll
  for (int k = 0; k < 18; ++k) X[k] = xload(k >> 1, k & 1);

  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // weight image (this wave's share) landed
  __syncthreads();                                    // the only barrier of the kernel

  const __amdgpu_buffer_rsrc_t mrsrc = __builtin_amdgcn_make_buffer_rsrc(p.out_main, 0, (int)p.main_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t srsrc = __builtin_amdgcn_make_buffer_rsrc(p.out_short, 0, (int)p.short_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t trsrc = __builtin_amdgcn_make_buffer_rsrc(p.out_tmp, 0, (int)p.tmp_bytes, 0x00020000);

  for (; tile < p.ntiles; tile += stride) {
    // where this tile's results go (before `locate` moves on to the next tile)
    const int tx = tile % p.tiles_x, orow = tile / p.tiles_x;
    const int ox = tx * 16 + i16;
    const bool st_ok = ox < p.Wo;
    const int m = orow * p.Wo + ox;
    locate(tile + stride);

    f32x4 acc[4];
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) acc[cb] = *reinterpret_cast<const f32x4*>(bias + cb * 16 + 4 * kq);   // bias: the
    // accumulator starts from it (no vector add per output later - VALU time adds to MFMA time)
    // 18 steps (tap, g) of 16 MFMAs.  The A fragments of step k + 1 are read from LDS before the MFMAs of step k; the
    // scheduling barrier after every step keeps the compiler from hoisting the next tile's pixel loads above the MFMAs
    // that still read the current ones (which would double the 72 fragment registers and spill).
    f32x4 wf[2][4];
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) wf[0][cb] = *reinterpret_cast<const f32x4*>(w3 + cb * 512 + wq[0]);
#pragma unroll
    for (int k = 0; k < 18; ++k) {
      const int tap = k >> 1, g = k & 1;
      if (k + 1 < 18) {
#pragma unroll
        for (int cb = 0; cb < 4; ++cb)
          wf[(k + 1) & 1][cb] = *reinterpret_cast<const f32x4*>(w3 + ((k + 1) >> 1) * 2048 + cb * 512 + wq[(k + 1) & 1]);
      }
      __builtin_amdgcn_sched_barrier(0);   // ... and keeps these LDS reads ABOVE the MFMAs they overlap with
      const f32x4 xf = X[k];
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int cb = 0; cb < 4; ++cb)
          acc[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[k & 1][cb][s], xf[s], acc[cb], 0, 0, 0);
      X[k] = xload(tap, g);   // the next tile's fragment, a whole tile of MFMAs ahead of its use
      __builtin_amdgcn_sched_barrier(0);
    }

    // ---- stage A epilogue in registers: lane holds couts cb*16 + 4kq + e of its pixel = the B operand of the 1x1 GEMM
    f32x4 va[4];
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) {
#pragma unroll
      for (int e = 0; e < 4; ++e) va[cb][e] = ff_silu(acc[cb][e]);
    }
    // ---- main | short = SiLU(W_ms . va + b)   (K = 64: 4 cout blocks of stage A x 4 steps)
    f32x4 am[4];
#pragma unroll
    for (int c2 = 0; c2 < 4; ++c2) am[c2] = *reinterpret_cast<const f32x4*>(bias + 64 + c2 * 16 + 4 * kq);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      f32x4 wf[4];
#pragma unroll
      for (int c2 = 0; c2 < 4; ++c2) wf[c2] = *reinterpret_cast<const f32x4*>(wms + ((c2 * 4 + c) * 64 + lane) * 4);
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int c2 = 0; c2 < 4; ++c2)
          am[c2] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[c2][s], va[c][s], am[c2], 0, 0, 0);
    }
    f32x4 vm[4];
#pragma unroll
    for (int c2 = 0; c2 < 4; ++c2) {
#pragma unroll
      for (int e = 0; e < 4; ++e) vm[c2][e] = ff_silu(am[c2][e]);
    }
    constexpr bool no_store = false;
    // stores: 16 bytes per (pixel, 4 couts); range-checked descriptors, no branches
    if (!no_store) {
#pragma unroll
      for (int c2 = 0; c2 < 2; ++c2) {
        const unsigned om = st_ok ? (unsigned)((m * p.main_ld + p.main_off + c2 * 16 + 4 * kq) * 4) : 0x80000000u;
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, vm[c2]), mrsrc, om, 0, 0);
        const unsigned os = st_ok ? (unsigned)((m * p.short_ld + p.short_off + c2 * 16 + 4 * kq) * 4) : 0x80000000u;
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, vm[2 + c2]), srsrc, os, 0, 0);
      }
    }
    // ---- conv1(main) = SiLU(W_c1 . vm[0..1] + b)   (K = 32)
    f32x4 ac[2];
#pragma unroll
    for (int c3 = 0; c3 < 2; ++c3) ac[c3] = *reinterpret_cast<const f32x4*>(bias + 128 + c3 * 16 + 4 * kq);
#pragma unroll
    for (int c2 = 0; c2 < 2; ++c2) {
      f32x4 wf[2];
#pragma unroll
      for (int c3 = 0; c3 < 2; ++c3) wf[c3] = *reinterpret_cast<const f32x4*>(wc1 + ((c3 * 2 + c2) * 64 + lane) * 4);
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int c3 = 0; c3 < 2; ++c3)
          ac[c3] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[c3][s], vm[c2][s], ac[c3], 0, 0, 0);
    }
#pragma unroll
    for (int c3 = 0; c3 < 2; ++c3) {
      f32x4 vt;
#pragma unroll
      for (int e = 0; e < 4; ++e) vt[e] = ff_silu(ac[c3][e]);
      const unsigned ot = st_ok && !no_store ? (unsigned)((m * p.tmp_ld + p.tmp_off + c3 * 16 + 4 * kq) * 4) : 0x80000000u;
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, vt), trsrc, ot, 0, 0);
    }
  }
#else
  (void)w3; (void)wms; (void)wc1; (void)bias; (void)i16; (void)kq; (void)wave;
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
  a.tiles_x = ceil_div(Wo, 16);
  const long long ntiles = (long long)da.N * Ho * a.tiles_x;
  ST_REQUIRE(ntiles < (1ll << 30), "fused front: too many tiles");
  a.ntiles = (int)ntiles;
  a.in_bytes = (unsigned)(Mi * da.in_ld * 4);
  a.wgt_bytes = (unsigned)(FF_C2 * 9 * FF_CIN * 4);
  a.main_bytes = (unsigned)(Mo * dms.out1_ld * 4);
  a.short_bytes = (unsigned)(Mo * dms.out2_ld * 4);
  a.tmp_bytes = (unsigned)(Mo * dc1.out1_ld * 4);
  // one persistent workgroup per CU (a multiple of 8 so that every XCD gets the same number), fewer for small inputs
  static int cus = 0;
  if (cus == 0) {
    int dev = 0, n = 0;
    ST_CHECK_HIP(hipGetDevice(&dev));
    ST_CHECK_HIP(hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev));
    cus = std::max(8, n / 8 * 8);
  }
  const int blocks = std::min(cus, std::max(8, round_up(ceil_div(a.ntiles, FF_WAVES), 8)));
  a.slots_per_xcd = blocks / 8 * FF_WAVES;
  constexpr int lds = FF_LDS_FLOATS * (int)sizeof(float);
  static int lds_set = 0;
  ST_ENSURE_DYNAMIC_LDS(front_s2_csp_kernel, lds, lds_set);
  hipLaunchKernelGGL(front_s2_csp_kernel, dim3((unsigned)blocks), dim3(FF_THREADS), lds, stream, a);
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
