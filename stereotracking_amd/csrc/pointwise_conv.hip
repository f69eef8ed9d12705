// Streaming 1x1 convolution for narrow layers (Cin in {32, 64}, Cout <= 64) on many pixels, gfx950.
//
// The 1x1 ConvModules of the high-resolution CSP stage (reference CSPLayer main/short/final convs and the
// bottleneck conv1, built at mmtrack/models/backbones/csp_darknet_disparity_v1.py:113-153, run at :176-184) move
// 256-512 B per pixel for 2*Cin*Cout flop: ~16 flop/B, below the 20 flop/B ridge of fp32 MFMA vs HBM, i.e. they are
// HBM-bound.  In the generic implicit-GEMM kernel such a layer is all prologue (K = 64 is two chunks); measured
// 3.3 TB/s.  This kernel is built for the stream instead:
//   * persistent, barrier-free waves; the whole weight matrix (<= 16 KB) is staged into LDS once per workgroup;
//   * every wave streams its own 32-pixel blocks: the pixel operand is loaded from global memory directly in MFMA
//     layout (16-byte quads, K-permuted like the weights) two blocks ahead of the one being multiplied;
//   * operands are swapped (A = weights, B = pixels): a lane owns 4 consecutive couts of one pixel, so the fused
//     epilogue (bias = folded BN, SiLU, residual + (a+b)*s, split outputs, channel-offset slices) reads the
//     residual and writes NHWC with 16-byte accesses through range-checked buffer descriptors (no branches).
// Same arithmetic as st_conv2d_nhwc for these shapes (packed weights [CoutPad][Kpad] are shared), registered as
// tile variant 41 so st_detector_autotune picks it only where it measures faster.
#include <algorithm>
#include <cstdint>
#include <type_traits>

#include "st_common.h"

namespace st {
namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));


struct PwArgs {
  const float* in;
  const float* wgt;
  const float* bias;
  float* out1;
  float* out2;
  const float* res;
  int M, in_ld, in_off, Cout, Kpad, split;
  int out1_ld, out1_off, out2_ld, out2_off, res_ld, res_off;
  float post_scale;
  int act;
  unsigned in_bytes, out1_bytes, out2_bytes, res_bytes;
  // chained second 1x1 conv on the first 32 output channels of this one (CHAIN kernels only)
  const float* wgt2;
  const float* bias2;
  float* out3;
  int Cout2, Kpad2, out3_ld, out3_off, act2;
  unsigned out3_bytes;
};

__device__ __forceinline__ float pw_silu(float v) { return v * __builtin_amdgcn_rcpf(1.0f + __expf(-v)); }

// Persistent, barrier-free waves: wave w owns the 32-pixel blocks w, w + nwaves, ...  The B operand (pixels) is
// loaded straight from global memory into MFMA layout (lane = pixel l31, half h: the 16-byte quads 2g + h of its
// pixel, K-permuted like the weights), two blocks ahead of the one being multiplied (three register sets, the
// block loop is unrolled by 3), so every wave keeps 2 x Cin x 128 B of reads in flight and nothing in the loop
// waits on a workgroup barrier: the first version streamed 128-pixel tiles through LDS by LDS-DMA and was held to
// 3.7 TB/s by the vmcnt(0) + barrier per tile, which also waits for the previous tile's stores.
template <int CIN, int NB, bool VEC, bool RES, bool CHAIN = false>
__global__ __launch_bounds__(256, 2) void pw_conv_kernel(const PwArgs p) {
  constexpr int Q = CIN / 4;     // 16-byte quads per pixel
  constexpr int G = CIN / 8;     // k-groups of 8 channels (4 MFMA steps each)
  extern __shared__ float4 pw_smem4[];
  float* wl = reinterpret_cast<float*>(pw_smem4);   // [NB*32][CIN], quads swizzled by the cout index
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int half = lane >> 5, l31 = lane & 31;
  const int nblk = (p.M + 31) >> 5;
  const int nwaves = gridDim.x * 4;
  int t = blockIdx.x * 4 + wave;

  // weights: packed [CoutPad][Kpad] (k = ci for a 1x1 conv) -> LDS once per workgroup
  for (int e = tid; e < NB * 32 * Q; e += 256) {
    const int co = e / Q, q = e % Q;
    const f32x4 v = *reinterpret_cast<const f32x4*>(p.wgt + (size_t)co * p.Kpad + 4 * q);
    *reinterpret_cast<f32x4*>(wl + (co * Q + (q ^ (co & (Q - 1)))) * 4) = v;
  }
  float* bl = wl + NB * 32 * CIN;   // bias [NB*32] in LDS (read per group in the epilogue: keeps 32 VGPRs free)
  if (tid < NB * 32) bl[tid] = p.bias[tid];
  // CHAIN: a second 1x1 conv (32 -> Cout2 <= 32) on output channels [0, 32) of this one.  The accumulator layout
  // of the swapped MFMA (lane = pixel, couts 8g + 4*half + {0..3}) IS the K-permuted B-operand layout, so the
  // activated outputs feed the second product straight from registers: no LDS round trip, no extra HBM read.
  float* wl2 = bl + NB * 32;        // [32][32], quads swizzled by the cout index (8 quads per row)
  float* bl2 = wl2 + 32 * 32;
  if (CHAIN) {
    for (int e = tid; e < 32 * 8; e += 256) {
      const int co = e >> 3, q = e & 7;
      const f32x4 v = *reinterpret_cast<const f32x4*>(p.wgt2 + (size_t)co * p.Kpad2 + 4 * q);
      *reinterpret_cast<f32x4*>(wl2 + (co * 8 + (q ^ (co & 7))) * 4) = v;
    }
    if (tid < 32) bl2[tid] = p.bias2[tid];
  }
  const int wsw = l31 & (Q - 1);
  __syncthreads();   // weights visible; the only barrier of the kernel

#if defined(__HIP_DEVICE_COMPILE__)  // device-only builtins; the host pass only needs the kernel stub
  const __amdgpu_buffer_rsrc_t irsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.in), 0, (int)p.in_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t o1rsrc = __builtin_amdgcn_make_buffer_rsrc(p.out1, 0, (int)p.out1_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t o2rsrc =
      __builtin_amdgcn_make_buffer_rsrc(p.out2 ? p.out2 : p.out1, 0, (int)(p.out2 ? p.out2_bytes : 0u), 0x00020000);
  const __amdgpu_buffer_rsrc_t rrsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(p.res ? p.res : p.in), 0, (int)(p.res ? p.res_bytes : 0u), 0x00020000);
  const __amdgpu_buffer_rsrc_t o3rsrc =
      __builtin_amdgcn_make_buffer_rsrc(CHAIN ? p.out3 : p.out1, 0, (int)(CHAIN ? p.out3_bytes : 0u), 0x00020000);

  f32x4 xr[3][G];   // pixel operands of three blocks in flight
  auto load_x = [&](auto set_tag, int blk) {
    constexpr int SET = decltype(set_tag)::value;
    const int m = blk * 32 + l31;
    const unsigned base = (blk < nblk && m < p.M) ? (unsigned)((m * p.in_ld + p.in_off + 4 * half) * 4) : 0x80000000u;
#pragma unroll
    for (int g = 0; g < G; ++g)
      xr[SET][g] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(irsrc, base, 32 * g, 0));
  };
  auto process = [&](auto set_tag, int blk) {
    constexpr int SET = decltype(set_tag)::value;
    const int m = blk * 32 + l31;
    const bool mv = m < p.M;
    f32x4 rv[RES ? NB : 1][4];
    if (RES) {   // residual of this block: in flight during the MFMAs
#pragma unroll
      for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int co = nb * 32 + 8 * g + 4 * half;
          if (VEC) {
            const unsigned off = (mv && co < p.Cout) ? (unsigned)((m * p.res_ld + p.res_off + co) * 4) : 0x80000000u;
            rv[nb][g] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rrsrc, off, 0, 0));
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const unsigned off =
                  (mv && co + e < p.Cout) ? (unsigned)((m * p.res_ld + p.res_off + co + e) * 4) : 0x80000000u;
              rv[nb][g][e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rrsrc, off, 0, 0));
            }
          }
        }
    }
    f32x16 acc[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[nb][r] = 0.f;
    // k-group g = channels 8g .. 8g+7: lane half h supplies channel 8g + 4h + step on both operands; the weight
    // fragments of group g+1 are read from LDS before group g multiplies
    f32x4 wf[2][NB];
    auto read_w = [&](int g, int set) {
      const int qq = 2 * g + half;
#pragma unroll
      for (int nb = 0; nb < NB; ++nb)
        wf[set][nb] = *reinterpret_cast<const f32x4*>(wl + ((nb * 32 + l31) * Q + (qq ^ wsw)) * 4);
    };
    read_w(0, 0);
#pragma unroll
    for (int g = 0; g < G; ++g) {
      const int set = g & 1;
      if (g + 1 < G) read_w(g + 1, set ^ 1);
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
          acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[set][nb][s], xr[SET][g][s], acc[nb], 0, 0, 0);
    }
    // ---- epilogue: C[co][pixel], lane = pixel, couts nb*32 + 8*g + 4*half + {0..3}
    f32x4 xm[CHAIN ? 4 : 1];   // activated outputs 0..31 of this pixel = B operand of the chained conv
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int co = nb * 32 + 8 * g + 4 * half;
        const f32x4 bq = *reinterpret_cast<const f32x4*>(bl + co);
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float x = acc[nb][4 * g + e] + bq[e];
          if (p.act) x = pw_silu(x);
          if (RES) x = (x + rv[RES ? nb : 0][g][e]) * p.post_scale;
          v[e] = x;
        }
        if (CHAIN && nb == 0) xm[CHAIN ? g : 0] = v;
        // branch-free split: the store into the "other" output carries an out-of-range offset and is dropped
        if (VEC) {   // every ld / off / split / Cout is a multiple of 4: one dwordx4 per group
          const bool ok = mv && co < p.Cout, first = co < p.split;
          const unsigned off1 = ok && first ? (unsigned)((m * p.out1_ld + p.out1_off + co) * 4) : 0x80000000u;
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), o1rsrc, off1, 0, 0);
          if (p.out2) {   // uniform
            const unsigned off2 =
                ok && !first ? (unsigned)((m * p.out2_ld + p.out2_off + co - p.split) * 4) : 0x80000000u;
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), o2rsrc, off2, 0, 0);
          }
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int c = co + e;
            const bool ok = mv && c < p.Cout, first = c < p.split;
            const unsigned off1 = ok && first ? (unsigned)((m * p.out1_ld + p.out1_off + c) * 4) : 0x80000000u;
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v[e]), o1rsrc, off1, 0, 0);
            if (p.out2) {
              const unsigned off2 =
                  ok && !first ? (unsigned)((m * p.out2_ld + p.out2_off + c - p.split) * 4) : 0x80000000u;
              __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v[e]), o2rsrc, off2, 0, 0);
            }
          }
        }
      }
    if (CHAIN) {
      f32x16 acc2;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc2[r] = 0.f;
#pragma unroll
      for (int g = 0; g < 4; ++g) {   // k-group g = channels 8g + 4*half + step, exactly xm[g][step]
        const f32x4 w2 = *reinterpret_cast<const f32x4*>(wl2 + (l31 * 8 + ((2 * g + half) ^ (l31 & 7))) * 4);
#pragma unroll
        for (int s2 = 0; s2 < 4; ++s2)
          acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(w2[s2], xm[CHAIN ? g : 0][s2], acc2, 0, 0, 0);
      }
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int co = 8 * g + 4 * half;
        const f32x4 bq = *reinterpret_cast<const f32x4*>(bl2 + co);
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float x = acc2[4 * g + e] + bq[e];
          v[e] = p.act2 ? pw_silu(x) : x;
        }
        if (VEC) {
          const unsigned off = (mv && co < p.Cout2) ? (unsigned)((m * p.out3_ld + p.out3_off + co) * 4) : 0x80000000u;
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), o3rsrc, off, 0, 0);
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const unsigned off =
                (mv && co + e < p.Cout2) ? (unsigned)((m * p.out3_ld + p.out3_off + co + e) * 4) : 0x80000000u;
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v[e]), o3rsrc, off, 0, 0);
          }
        }
      }
    }
  };
  using S0 = std::integral_constant<int, 0>;
  using S1 = std::integral_constant<int, 1>;
  using S2 = std::integral_constant<int, 2>;
  // every wave walks its own blocks; out-of-range prefetches read zeros and are never used
  if (CIN <= 32) {   // two blocks ahead (3 x 16 operand registers)
    load_x(S0{}, t);
    load_x(S1{}, t + nwaves);
    while (t < nblk) {
      load_x(S2{}, t + 2 * nwaves);
      process(S0{}, t);
      t += nwaves;
      if (t >= nblk) break;
      load_x(S0{}, t + 2 * nwaves);
      process(S1{}, t);
      t += nwaves;
      if (t >= nblk) break;
      load_x(S1{}, t + 2 * nwaves);
      process(S2{}, t);
      t += nwaves;
    }
  } else {           // one block ahead (2 x 32 operand registers): stays within 2 waves per SIMD
    load_x(S0{}, t);
    while (t < nblk) {
      load_x(S1{}, t + nwaves);
      process(S0{}, t);
      t += nwaves;
      if (t >= nblk) break;
      load_x(S0{}, t + nwaves);
      process(S1{}, t);
      t += nwaves;
    }
  }
#else
  (void)wsw; (void)nwaves; (void)nblk; (void)t; (void)bl; (void)bl2;
#endif
}

}  // namespace

// Shapes this kernel takes: 1x1 / stride 1 / no padding, Cin 32 or 64, Cout <= 64, no upsampled store, every
// tensor below 2 GiB (32-bit byte offsets with the top bit reserved for "out of range").
bool pw_conv_applicable(const StConvDesc& d) {
  if (d.KH != 1 || d.KW != 1 || d.stride != 1 || d.pad != 0 || d.up_dev) return false;
  if (d.Cin != 32 && d.Cin != 64) return false;
  if (d.Cout < 1 || d.Cout > 64) return false;
  if (d.in_ld % 4 || d.in_off % 4 || (reinterpret_cast<uintptr_t>(d.in_dev) & 15)) return false;
  const long long M = (long long)d.N * d.Hi * d.Wi;
  const long long lim = 1ll << 31;
  if (M * d.in_ld * 4 >= lim || M * d.out1_ld * 4 >= lim) return false;
  if (d.out2_dev && M * d.out2_ld * 4 >= lim) return false;
  if (d.res_dev && M * d.res_ld * 4 >= lim) return false;
  return true;
}

// `chain` (may be null): a second 1x1 conv whose input is exactly output channels [0, 32) of `d`'s out1 slice
// (chain->in_dev / in_ld / in_off are ignored), Cin = 32, Cout <= 32, no split / residual / upsample.
bool pw_chain_applicable(const StConvDesc& d, const StConvDesc& c) {
  if (!pw_conv_applicable(d) || !pw_conv_applicable(c)) return false;
  const int split = d.out2_dev ? d.split : d.Cout;
  if (d.res_dev || split < 32 || d.Cout <= 32 || d.Cout > 64) return false;   // the fused kernel is the NB = 2 one
  if (c.Cin != 32 || c.Cout > 32 || c.out2_dev || c.res_dev || c.up_dev) return false;
  if (c.N != d.N || c.Hi != d.Hi || c.Wi != d.Wi) return false;
  return true;
}

int pw_conv_launch(const StConvDesc& d, hipStream_t stream, const StConvDesc* chain) {
  ST_REQUIRE(pw_conv_applicable(d), "pointwise conv: shape not supported by the streaming kernel");
  if (chain) ST_REQUIRE(pw_chain_applicable(d, *chain), "pointwise conv: pair cannot be chained");
  ST_REQUIRE(d.in_dev && d.wgt_dev && d.bias_dev && d.out1_dev, "pointwise conv: null pointer");
  const int split = d.out2_dev ? d.split : d.Cout;
  ST_REQUIRE(split >= 0 && split <= d.Cout && d.out1_off + split <= d.out1_ld, "pointwise conv: bad split / out1 slice");
  if (d.out2_dev) ST_REQUIRE(d.out2_off + (d.Cout - split) <= d.out2_ld, "pointwise conv: out2 slice exceeds out2_ld");
  if (d.res_dev) ST_REQUIRE(d.res_off + d.Cout <= d.res_ld, "pointwise conv: res slice exceeds res_ld");
  const long long M = (long long)d.N * d.Hi * d.Wi;
  PwArgs a;
  a.in = d.in_dev; a.wgt = d.wgt_dev; a.bias = d.bias_dev;
  a.out1 = d.out1_dev; a.out2 = d.out2_dev; a.res = d.res_dev;
  a.M = (int)M; a.in_ld = d.in_ld; a.in_off = d.in_off; a.Cout = d.Cout; a.Kpad = round_up(d.Cin, 32);
  a.split = split;
  a.out1_ld = d.out1_ld; a.out1_off = d.out1_off; a.out2_ld = d.out2_ld; a.out2_off = d.out2_off;
  a.res_ld = d.res_ld; a.res_off = d.res_off;
  a.post_scale = d.res_dev ? d.post_scale : 1.0f;
  a.act = d.act;
  a.in_bytes = (unsigned)(M * d.in_ld * 4);
  a.out1_bytes = (unsigned)(M * d.out1_ld * 4);
  a.out2_bytes = d.out2_dev ? (unsigned)(M * d.out2_ld * 4) : 0u;
  a.res_bytes = d.res_dev ? (unsigned)(M * d.res_ld * 4) : 0u;
  auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  bool vec = ((d.out1_ld | d.out1_off | split | d.Cout) & 3) == 0 && al16(d.out1_dev);
  if (d.out2_dev) vec = vec && ((d.out2_ld | d.out2_off) & 3) == 0 && al16(d.out2_dev);
  if (d.res_dev) vec = vec && ((d.res_ld | d.res_off) & 3) == 0 && al16(d.res_dev);
  const int nb = round_up(d.Cout, 32) / 32;
  if (chain) {
    a.wgt2 = chain->wgt_dev; a.bias2 = chain->bias_dev; a.out3 = chain->out1_dev;
    a.Cout2 = chain->Cout; a.Kpad2 = 32; a.out3_ld = chain->out1_ld; a.out3_off = chain->out1_off; a.act2 = chain->act;
    ST_REQUIRE(chain->wgt_dev && chain->bias_dev && chain->out1_dev, "pointwise conv: null chain pointer");
    ST_REQUIRE(chain->out1_off + chain->Cout <= chain->out1_ld, "pointwise conv: chain output slice exceeds its ld");
    a.out3_bytes = (unsigned)(M * chain->out1_ld * 4);
    vec = vec && ((chain->out1_ld | chain->out1_off | chain->Cout) & 3) == 0 && al16(chain->out1_dev);
  } else {
    a.wgt2 = nullptr; a.bias2 = nullptr; a.out3 = nullptr;
    a.Cout2 = 0; a.Kpad2 = 0; a.out3_ld = 0; a.out3_off = 0; a.act2 = 0; a.out3_bytes = 0;
  }
  // weights + bias (+ chained weights + bias)
  const size_t lds = (size_t)(nb * 32 * d.Cin + nb * 32 + (chain ? 32 * 32 + 32 : 0)) * sizeof(float);
  static int cus = 0;
  if (!cus) {
    int dev = 0;
    ST_CHECK_HIP(hipGetDevice(&dev));
    ST_CHECK_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    if (cus <= 0) cus = 256;
  }
  using Kern = void (*)(const PwArgs);
#define PW_K(C, N) pw_conv_kernel<C, N, false, false>, pw_conv_kernel<C, N, true, false>, \
                   pw_conv_kernel<C, N, false, true>, pw_conv_kernel<C, N, true, true>
  static const Kern kerns[20] = {PW_K(32, 1), PW_K(32, 2), PW_K(64, 1), PW_K(64, 2),
                                 pw_conv_kernel<32, 2, false, false, true>, pw_conv_kernel<32, 2, true, false, true>,
                                 pw_conv_kernel<64, 2, false, false, true>, pw_conv_kernel<64, 2, true, false, true>};
#undef PW_K
  const int ki = chain ? 16 + (d.Cin == 64 ? 2 : 0) + (vec ? 1 : 0)
                       : (d.Cin == 64 ? 8 : 0) + (nb - 1) * 4 + (d.res_dev ? 2 : 0) + (vec ? 1 : 0);
  static bool attr_set[20] = {};
  if (!attr_set[ki]) {
    ST_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kerns[ki]),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_set[ki] = true;
  }
  const long long nblk = (M + 31) / 32;                     // 32-pixel blocks, one wave each
  const unsigned grid = (unsigned)std::min<long long>((nblk + 3) / 4, (long long)cus * 2);   // 2 waves per SIMD
  hipLaunchKernelGGL(kerns[ki], dim3(grid), dim3(256), lds, stream, a);
  ST_CHECK_HIP(hipGetLastError());
  return ST_OK;
}

}  // namespace st
