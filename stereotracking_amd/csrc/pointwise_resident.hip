// 1x1 convolution with the whole weight matrix resident in LDS and the pixels fed from global memory straight into
// MFMA operand registers - tile variant 46 ("pwres"), for the 1x1 ConvModules of the path whose weights fit in LDS
// (Cin in {64, 128, 256}, Cout in {64, 128}: CSPLayer main|short / final convs and bottleneck conv1 of stage 2-3
// and of the PAFPN, out_layers; mmdet CSPLayer / DarknetBottleneck as built at
// /root/reference/mmtrack/models/backbones/csp_darknet_disparity_v1.py:113-153 and mmyolo YOLOXPAFPN).
//
// Those layers are short GEMMs (1-8 GFLOP) over 30-120 k pixels: on the implicit-GEMM kernel a workgroup spends as
// long staging its 64x32 operand tiles through LDS and on barriers as on MFMAs (45-75 TF/s).  Here (the structure of
// front_fused.hip) ONE persistent workgroup per CU loads the weights once, in MFMA-fragment order (a source-side
// permutation of the LDS-DMA: A-fragment reads are then lane-contiguous, conflict-free), and after that single
// barrier its 8 waves never synchronise: a wave owns 16-pixel tiles; a lane's B fragment for channel group g is 16
// contiguous bytes of the NHWC input, loaded by `buffer_load_dwordx4` one tile ahead into the registers the current
// tile has just consumed.  Swapped operands (A = weights, B = pixels): the accumulator holds 4 consecutive output
// channels of one pixel per lane, so bias / SiLU / residual / split store are 16-byte accesses, and the activated
// accumulator IS the B operand of a chained second 1x1 conv (CSP main_conv -> bottleneck conv1).
// `v_mfma_f32_16x16x4_f32`, exact fp32; same arithmetic as the implicit-GEMM kernel up to summation order.
#include <algorithm>
#include <cstdint>

#include "st_common.h"

namespace st {
namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int PR_WAVES = 8, PR_THREADS = 64 * PR_WAVES;

struct PwrArgs {
  const float* in;
  const float* wgt;     // packed [CoutPad][Kpad]
  const float* bias;
  const float* res;
  float* out1;
  float* out2;
  const float* wgt2;    // chained conv: packed [.][Kpad2], bias2, out3
  const float* bias2;
  float* out3;
  int M, in_ld, in_off, Kpad, Kpad2;
  int split, out1_ld, out1_off, out2_ld, out2_off, res_ld, res_off, out3_ld, out3_off;
  int act, act2;
  float post_scale;
  int ntiles;
  unsigned in_bytes, wgt_bytes, wgt2_bytes, out1_bytes, out2_bytes, res_bytes, out3_bytes;
};

__device__ __forceinline__ float pr_silu(float v) { return v * __builtin_amdgcn_rcpf(1.0f + __expf(-v)); }

// KG = Cin / 16, CB = Cout / 16, CH = channel blocks of the chained conv (0 = none): it maps output channels
// [0, 16 CH) of this conv to 16 CH channels of its own.
template <int KG, int CB, int CH, bool RES>
__global__ __launch_bounds__(PR_THREADS, 1) void pw_resident_kernel(const PwrArgs p) {
  extern __shared__ float4 pr_smem4[];
  float* wl = reinterpret_cast<float*>(pr_smem4);   // [cb][g][lane][4]
  float* wl2 = wl + CB * KG * 256;                  // [c3][c2][lane][4]
  float* bl = wl2 + CH * CH * 256;                  // bias (16 CB) | bias2 (16 CH)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform: tile bookkeeping on the scalar unit
  const int i16 = lane & 15, kq = lane >> 4;

#if defined(__HIP_DEVICE_COMPILE__)  // device-only builtins; the host pass only needs the kernel stub
  const __amdgpu_buffer_rsrc_t irsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.in), 0, (int)p.in_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t wrsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.wgt), 0, (int)p.wgt_bytes, 0x00020000);
  // weight image: fragment (cb, g) of lane l = W[16 cb + (l & 15)][16 g + 4 (l >> 4) .. + 3]; one wave-DMA per fragment
  for (int f = wave; f < CB * KG; f += PR_WAVES) {
    const int cb = f / KG, g = f - cb * KG;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(wrsrc, (__attribute__((address_space(3))) void*)(wl + f * 256), 16,
                                             (unsigned)(((cb * 16 + i16) * p.Kpad + 16 * g + 4 * kq) * 4), 0, 0, 0);
  }
  if (CH > 0) {
    const __amdgpu_buffer_rsrc_t w2rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.wgt2), 0, (int)p.wgt2_bytes, 0x00020000);
    for (int f = wave; f < CH * CH; f += PR_WAVES) {
      const int c3 = f / CH, c2 = f - c3 * CH;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(w2rsrc, (__attribute__((address_space(3))) void*)(wl2 + f * 256), 16,
                                               (unsigned)(((c3 * 16 + i16) * p.Kpad2 + 16 * c2 + 4 * kq) * 4), 0, 0, 0);
    }
  }
  if (tid < 16 * CB) bl[tid] = p.bias[tid];
  else if (tid < 16 * (CB + CH)) bl[tid] = p.bias2[tid - 16 * CB];

  const int slot = blockIdx.x * PR_WAVES + wave, stride = gridDim.x * PR_WAVES;
  // B fragments of tile t: pixel 16 t + i16, channels 16 g + 4 kq .. + 3; pixels past M read zeros (never stored)
  auto xoff = [&](int t) -> unsigned {
    const int m = t * 16 + i16;
    return (t < p.ntiles && m < p.M) ? (unsigned)((m * p.in_ld + p.in_off + 4 * kq) * 4) : 0x80000000u;
  };
  f32x4 X[KG];
  int tile = slot;
  {
    const unsigned o = xoff(tile);
#pragma unroll
    for (int g = 0; g < KG; ++g)
      X[g] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(irsrc, o + 64u * g, 0, 0));
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's share of the weight image landed
  __syncthreads();                                    // the only barrier of the kernel

  const __amdgpu_buffer_rsrc_t o1rsrc = __builtin_amdgcn_make_buffer_rsrc(p.out1, 0, (int)p.out1_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t o2rsrc =
      __builtin_amdgcn_make_buffer_rsrc(p.out2 ? p.out2 : p.out1, 0, (int)(p.out2 ? p.out2_bytes : 0u), 0x00020000);
  const __amdgpu_buffer_rsrc_t rrsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(RES ? p.res : p.in), 0, (int)(RES ? p.res_bytes : 0u), 0x00020000);
  const __amdgpu_buffer_rsrc_t o3rsrc =
      __builtin_amdgcn_make_buffer_rsrc(CH > 0 ? p.out3 : p.out1, 0, (int)(CH > 0 ? p.out3_bytes : 0u), 0x00020000);

  for (; tile < p.ntiles; tile += stride) {
    const int m = tile * 16 + i16;
    const bool st_ok = m < p.M;
    const unsigned onext = xoff(tile + stride);
    f32x4 rv[RES ? CB : 1];
    if (RES) {   // residual in accumulator layout, issued now, consumed after the MFMAs
#pragma unroll
      for (int cb = 0; cb < CB; ++cb) {
        const unsigned ro = st_ok ? (unsigned)((m * p.res_ld + p.res_off + cb * 16 + 4 * kq) * 4) : 0x80000000u;
        rv[cb] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rrsrc, ro, 0, 0));
      }
    }
    f32x4 acc[CB];
#pragma unroll
    for (int cb = 0; cb < CB; ++cb) acc[cb] = *reinterpret_cast<const f32x4*>(bl + cb * 16 + 4 * kq);   // starts from
    // the bias: no vector add per output later (VALU time adds to MFMA time on the fp32 matrix path)
    // KG steps of 4 CB MFMAs; the A fragments of step g + 1 are read from LDS before the MFMAs of step g, and the
    // scheduling barriers keep (a) those reads above the MFMAs they overlap with and (b) the next tile's pixel loads
    // below the MFMAs that still read the current ones
    f32x4 wf[2][CB];
#pragma unroll
    for (int cb = 0; cb < CB; ++cb) wf[0][cb] = *reinterpret_cast<const f32x4*>(wl + (cb * KG * 64 + lane) * 4);
#pragma unroll
    for (int g = 0; g < KG; ++g) {
      if (g + 1 < KG) {
#pragma unroll
        for (int cb = 0; cb < CB; ++cb)
          wf[(g + 1) & 1][cb] = *reinterpret_cast<const f32x4*>(wl + ((cb * KG + g + 1) * 64 + lane) * 4);
      }
      __builtin_amdgcn_sched_barrier(0);
      const f32x4 xf = X[g];
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int cb = 0; cb < CB; ++cb)
          acc[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[g & 1][cb][s], xf[s], acc[cb], 0, 0, 0);
      X[g] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(irsrc, onext + 64u * g, 0, 0));
      __builtin_amdgcn_sched_barrier(0);
    }

    // epilogue: lane holds output channels 16 cb + 4 kq + e of its pixel
    f32x4 v[CB];
#pragma unroll
    for (int cb = 0; cb < CB; ++cb) {
      v[cb] = acc[cb];
      if (p.act) {   // uniform branch, not a per-value select
#pragma unroll
        for (int e = 0; e < 4; ++e) v[cb][e] = pr_silu(v[cb][e]);
      }
    }
    if (CH > 0) {   // chained conv on channels [0, 16 CH) (before any residual: CSP main_conv has none)
      f32x4 ac[CH > 0 ? CH : 1];
#pragma unroll
      for (int c3 = 0; c3 < CH; ++c3) ac[c3] = *reinterpret_cast<const f32x4*>(bl + 16 * CB + c3 * 16 + 4 * kq);
#pragma unroll
      for (int c2 = 0; c2 < CH; ++c2) {
        f32x4 w2[CH > 0 ? CH : 1];
#pragma unroll
        for (int c3 = 0; c3 < CH; ++c3) w2[c3] = *reinterpret_cast<const f32x4*>(wl2 + ((c3 * CH + c2) * 64 + lane) * 4);
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
          for (int c3 = 0; c3 < CH; ++c3)
            ac[c3] = __builtin_amdgcn_mfma_f32_16x16x4f32(w2[c3][s], v[c2][s], ac[c3], 0, 0, 0);
      }
#pragma unroll
      for (int c3 = 0; c3 < CH; ++c3) {
        f32x4 t = ac[c3];
        if (p.act2) {
#pragma unroll
          for (int e = 0; e < 4; ++e) t[e] = pr_silu(t[e]);
        }
        const unsigned o = st_ok ? (unsigned)((m * p.out3_ld + p.out3_off + c3 * 16 + 4 * kq) * 4) : 0x80000000u;
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, t), o3rsrc, o, 0, 0);
      }
    }
#pragma unroll
    for (int cb = 0; cb < CB; ++cb) {
      f32x4 t = v[cb];
      if (RES) {
#pragma unroll
        for (int e = 0; e < 4; ++e) t[e] = (t[e] + rv[cb][e]) * p.post_scale;
      }
      const int co = cb * 16;
      if (co < p.split) {   // wave-uniform
        const unsigned o = st_ok ? (unsigned)((m * p.out1_ld + p.out1_off + co + 4 * kq) * 4) : 0x80000000u;
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, t), o1rsrc, o, 0, 0);
      } else {
        const unsigned o = st_ok ? (unsigned)((m * p.out2_ld + p.out2_off + co - p.split + 4 * kq) * 4) : 0x80000000u;
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, t), o2rsrc, o, 0, 0);
      }
    }
  }
#else
  (void)wl; (void)wl2; (void)bl; (void)i16; (void)kq; (void)wave;
#endif
}

template <int KG, int CB, int CH, bool RES>
int pwr_launch_instance(const PwrArgs& a, int blocks, hipStream_t stream) {
  constexpr int lds = (CB * KG * 256 + CH * CH * 256 + 16 * (CB + CH) + 64) * (int)sizeof(float);
  static_assert(lds <= 160 * 1024, "weight image exceeds LDS");
  static int lds_set = 0;
  ST_ENSURE_DYNAMIC_LDS((pw_resident_kernel<KG, CB, CH, RES>), lds, lds_set);
  hipLaunchKernelGGL((pw_resident_kernel<KG, CB, CH, RES>), dim3((unsigned)blocks), dim3(PR_THREADS), lds, stream, a);
  ST_CHECK_HIP(hipGetLastError());
  return ST_OK;
}

bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

bool pwr_shape(int Cin, int Cout) {
  return (Cin == 64 && Cout == 64) || (Cin == 128 && Cout == 128) || (Cin == 256 && Cout == 128) ||
         (Cin == 128 && Cout == 64);
}

}  // namespace

// 1x1 / stride 1, Cin -> Cout one of 64->64, 128->64, 128->128, 256->128; split (if any) a multiple of 16; 16-byte
// aligned tensors with channel strides / offsets multiples of 4; no upsample store.
bool pwr_conv_applicable(const StConvDesc& d) {
  if (d.KH != 1 || d.KW != 1 || d.stride != 1 || d.pad != 0 || d.up_dev) return false;
  if (!pwr_shape(d.Cin, d.Cout)) return false;
  if (!d.in_dev || !d.wgt_dev || !d.bias_dev || !d.out1_dev) return false;
  const int split = d.out2_dev ? d.split : d.Cout;
  if (split < 0 || split > d.Cout || (split & 15)) return false;
  if ((d.in_ld | d.in_off | d.out1_ld | d.out1_off) & 3) return false;
  if (!al16(d.in_dev) || !al16(d.out1_dev)) return false;
  if (d.in_off + d.Cin > d.in_ld || d.out1_off + split > d.out1_ld) return false;
  const long long M = (long long)d.N * d.Hi * d.Wi, lim = 1ll << 31;
  if (M <= 0 || M >= (1ll << 30) || M * d.in_ld * 4 >= lim || M * d.out1_ld * 4 >= lim) return false;
  if (d.out2_dev && (((d.out2_ld | d.out2_off) & 3) || !al16(d.out2_dev) || d.out2_off + d.Cout - split > d.out2_ld ||
                     M * d.out2_ld * 4 >= lim)) return false;
  if (d.res_dev && (((d.res_ld | d.res_off) & 3) || !al16(d.res_dev) || d.res_off + d.Cout > d.res_ld ||
                    M * d.res_ld * 4 >= lim)) return false;
  return true;
}

// `c`: a second 1x1 conv (64 -> 64, no split / residual / upsample) whose input is exactly output channels [0, 64)
// of `d`'s out1 slice (c.in_dev / in_ld / in_off are ignored): CSP main_conv -> bottleneck conv1 with mid = 64.
bool pwr_chain_applicable(const StConvDesc& d, const StConvDesc& c) {
  if (!pwr_conv_applicable(d)) return false;
  const int split = d.out2_dev ? d.split : d.Cout;
  if (d.res_dev || d.Cout != 128 || split != 64) return false;
  if (c.KH != 1 || c.KW != 1 || c.stride != 1 || c.pad != 0 || c.up_dev || c.res_dev || c.out2_dev) return false;
  if (c.Cin != 64 || c.Cout != 64 || !c.wgt_dev || !c.bias_dev || !c.out1_dev) return false;
  if (c.N != d.N || c.Hi != d.Hi || c.Wi != d.Wi) return false;
  if (((c.out1_ld | c.out1_off) & 3) || !al16(c.out1_dev) || c.out1_off + 64 > c.out1_ld) return false;
  return (long long)d.N * d.Hi * d.Wi * c.out1_ld * 4 < (1ll << 31);
}

int pwr_conv_launch(const StConvDesc& d, hipStream_t stream, const StConvDesc* chain) {
  ST_REQUIRE(pwr_conv_applicable(d), "resident 1x1 conv: needs 1x1/s1, Cin->Cout one of 64->64, 128->64, 128->128, 256->128, "
                                     "split %% 16 == 0, 16-byte aligned tensors, no upsample store");
  if (chain) ST_REQUIRE(pwr_chain_applicable(d, *chain), "resident 1x1 conv: pair cannot be chained");
  const long long M = (long long)d.N * d.Hi * d.Wi;
  PwrArgs a{};
  a.in = d.in_dev; a.wgt = d.wgt_dev; a.bias = d.bias_dev; a.res = d.res_dev;
  a.out1 = d.out1_dev; a.out2 = d.out2_dev;
  a.M = (int)M; a.in_ld = d.in_ld; a.in_off = d.in_off; a.Kpad = round_up(d.Cin, 32);
  a.split = d.out2_dev ? d.split : d.Cout;
  a.out1_ld = d.out1_ld; a.out1_off = d.out1_off; a.out2_ld = d.out2_ld; a.out2_off = d.out2_off;
  a.res_ld = d.res_ld; a.res_off = d.res_off;
  a.act = d.act;
  a.post_scale = d.res_dev ? d.post_scale : 1.0f;
  a.ntiles = (int)((M + 15) / 16);
  a.in_bytes = (unsigned)(M * d.in_ld * 4);
  a.wgt_bytes = (unsigned)((long long)round_up(d.Cout, 32) * a.Kpad * 4);
  a.out1_bytes = (unsigned)(M * d.out1_ld * 4);
  a.out2_bytes = d.out2_dev ? (unsigned)(M * d.out2_ld * 4) : 0u;
  a.res_bytes = d.res_dev ? (unsigned)(M * d.res_ld * 4) : 0u;
  if (chain) {
    a.wgt2 = chain->wgt_dev; a.bias2 = chain->bias_dev; a.out3 = chain->out1_dev;
    a.Kpad2 = round_up(chain->Cin, 32); a.out3_ld = chain->out1_ld; a.out3_off = chain->out1_off; a.act2 = chain->act;
    a.wgt2_bytes = (unsigned)((long long)round_up(chain->Cout, 32) * a.Kpad2 * 4);
    a.out3_bytes = (unsigned)(M * chain->out1_ld * 4);
  }
  static int cus = 0;
  if (cus == 0) {
    int dev = 0, n = 0;
    ST_CHECK_HIP(hipGetDevice(&dev));
    ST_CHECK_HIP(hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev));
    cus = std::max(1, n);
  }
  const int blocks = std::min(cus, ceil_div(a.ntiles, PR_WAVES));
  const bool res = d.res_dev != nullptr;
  if (chain) return d.Cin == 128 ? pwr_launch_instance<8, 8, 4, false>(a, blocks, stream)
                                 : pwr_launch_instance<16, 8, 4, false>(a, blocks, stream);
  if (d.Cin == 64) return res ? pwr_launch_instance<4, 4, 0, true>(a, blocks, stream)
                              : pwr_launch_instance<4, 4, 0, false>(a, blocks, stream);
  if (d.Cin == 128 && d.Cout == 64) return res ? pwr_launch_instance<8, 4, 0, true>(a, blocks, stream)
                                               : pwr_launch_instance<8, 4, 0, false>(a, blocks, stream);
  if (d.Cin == 128) return res ? pwr_launch_instance<8, 8, 0, true>(a, blocks, stream)
                               : pwr_launch_instance<8, 8, 0, false>(a, blocks, stream);
  return res ? pwr_launch_instance<16, 8, 0, true>(a, blocks, stream)
             : pwr_launch_instance<16, 8, 0, false>(a, blocks, stream);
}

}  // namespace st
