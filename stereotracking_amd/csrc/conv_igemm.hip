// Exact-fp32 MFMA implicit-GEMM convolution for gfx950 (MI355X), NHWC.
//
//   out[m][j] = epilogue( sum_k A[m][k] * W[j][k] + bias[j] )
//   m = (n, oy, ox) output pixel, k = (kh, kw, ci), j = output channel.
//
// One ConvModule of the reference (Conv2d no-bias -> BatchNorm2d(eps=1e-3) ->
// SiLU; reference mmtrack/models/backbones/csp_darknet_disparity_v1.py:126-135)
// is ONE launch of this kernel: BN is folded into W/bias on the host in fp64,
// SiLU, the bottleneck residual, the two-branch average (a+b)/2
// (csp_darknet_disparity_v1.py:184), CSP/PAFPN channel concat (channel-offset
// stores) and PAFPN nearest x2 upsample (replicated stores) all live in the
// epilogue, so no separate elementwise kernel touches HBM.
//
// Machine mapping (MI355X_MICROARCH.md): v_mfma_f32_32x32x2_f32 (exact fp32,
// 64 cycles/SIMD, 157 TFLOP/s chip peak).  A wave owns a (32*TM)x(32*TN)
// output tile in 16*TM*TN accumulator registers.  A and W tiles of BK=32 are
// staged global -> registers (16 B/lane, im2col gather with zero fill) ->
// LDS rows of 36 floats (conflict-free ds_read_b128: 36*i mod 64 distinct for
// any 16 rows) and double buffered, one barrier per K-chunk.  Lane (i, h)
// reads k = 8g+4h .. 8g+4h+3 with one ds_read_b128; MFMA step s pairs
// A[i][8g+4h+s] with W[j][8g+4h+s] for h = 0,1 - a k permutation, which a
// sum over k does not care about.
#include <algorithm>

#include "st_common.h"

namespace st {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct ConvKArgs {
  const float* in;
  const float* wgt;
  const float* bias;
  float* out1;
  float* out2;
  float* up;
  const float* res;
  int Hi, Wi, Cin, in_ld, in_off;
  int Ho, Wo, HoWo, Cout;
  int KW, stride, pad;
  int K, Kpad, M;
  int out1_ld, out1_off, split;
  int out2_ld, out2_off;
  int up_ld, up_off;
  int res_ld, res_off;
  float post_scale;
  int act;
  int n_tiles;  // tiles along Cout
};

constexpr int BK = 32;
constexpr int LDK = 36;  // padded LDS row (floats)

__device__ __forceinline__ float silu_f32(float v) {
  // v * sigmoid(v); __expf = v_exp_f32(x*log2e), 1 ulp-class; v/(1+inf) -> -0 for v << 0
  return v / (1.0f + __expf(-v));
}

template <int TM, int TN, int WM, int WN>
__global__ __launch_bounds__(64 * WM * WN) void conv_igemm_kernel(ConvKArgs p) {
  constexpr int NT = 64 * WM * WN;
  constexpr int BM = 32 * TM * WM;
  constexpr int BN = 32 * TN * WN;
  constexpr int ROWS = NT / 8;  // tile rows covered by one staging pass
  constexpr int AP = BM / ROWS;
  constexpr int BP = BN / ROWS;
  static_assert(BM % ROWS == 0 && BN % ROWS == 0, "tile/thread mismatch");

  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;                 // [2][BM][LDK]
  float* Bs = smem + 2 * BM * LDK;  // [2][BN][LDK]

  // XCD-aware tile order: blocks b and b+8 share an XCD (and its L2); give each XCD a
  // contiguous run of tiles so Cout-tiles of one pixel tile and neighbouring pixel
  // tiles (3x3 halo rows) hit the same L2.  Bijective for any grid size.
  const int nblk = gridDim.x, bid = blockIdx.x;
  const int q = nblk >> 3, r8 = nblk & 7, xcd = bid & 7, slot = bid >> 3;
  const int logical = (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + slot;
  const int mt = logical / p.n_tiles;
  const int nt = logical - mt * p.n_tiles;
  const int m0 = mt * BM, n0 = nt * BN;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave - wm * WN;
  const int l31 = lane & 31, lh = lane >> 5;

  // ---- staging geometry: thread owns 4-float group kq of rows r0 + ROWS*pass
  const int kq = tid & 7, r0 = tid >> 3;
  int iy0[AP], ix0[AP], pix0[AP];
#pragma unroll
  for (int a = 0; a < AP; ++a) {
    const int m = m0 + r0 + a * ROWS;
    const bool vm = m < p.M;
    const int mm = vm ? m : 0;
    const int n = mm / p.HoWo;
    const int rem = mm - n * p.HoWo;
    const int oy = rem / p.Wo;
    const int ox = rem - oy * p.Wo;
    iy0[a] = vm ? oy * p.stride - p.pad : -(1 << 20);
    ix0[a] = ox * p.stride - p.pad;
    pix0[a] = n * p.Hi * p.Wi;
  }
  const float* wrow[BP];
#pragma unroll
  for (int b = 0; b < BP; ++b) wrow[b] = p.wgt + (size_t)(n0 + r0 + b * ROWS) * p.Kpad + kq * 4;

  // running (kh, kw, c) of this lane's 4-float group; advanced by 32 per chunk
  int kc_c = kq * 4, kc_kh = 0, kc_kw = 0, kc_k = kq * 4;
  while (kc_c >= p.Cin) {
    kc_c -= p.Cin;
    if (++kc_kw == p.KW) { kc_kw = 0; ++kc_kh; }
  }

  f32x4 areg[AP], breg[BP];
  auto load_chunk = [&](int kc) {
    const bool vk = kc_k < p.K;
#pragma unroll
    for (int a = 0; a < AP; ++a) {
      const int iy = iy0[a] + kc_kh, ix = ix0[a] + kc_kw;
      const bool v = vk && (unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi;
      f32x4 val = {0.f, 0.f, 0.f, 0.f};
      if (v) {
        const float* src =
            p.in + (size_t)(pix0[a] + iy * p.Wi + ix) * p.in_ld + p.in_off + kc_c;
        val = *reinterpret_cast<const f32x4*>(src);
      }
      areg[a] = val;
    }
#pragma unroll
    for (int b = 0; b < BP; ++b)
      breg[b] = *reinterpret_cast<const f32x4*>(wrow[b] + (size_t)kc * BK);
    // advance to the next chunk
    kc_k += BK;
    kc_c += BK;
    while (kc_c >= p.Cin) {
      kc_c -= p.Cin;
      if (++kc_kw == p.KW) { kc_kw = 0; ++kc_kh; }
    }
  };
  auto store_chunk = [&](int buf) {
    float* Ab = As + buf * BM * LDK;
    float* Bb = Bs + buf * BN * LDK;
#pragma unroll
    for (int a = 0; a < AP; ++a)
      *reinterpret_cast<f32x4*>(Ab + (r0 + a * ROWS) * LDK + kq * 4) = areg[a];
#pragma unroll
    for (int b = 0; b < BP; ++b)
      *reinterpret_cast<f32x4*>(Bb + (r0 + b * ROWS) * LDK + kq * 4) = breg[b];
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nchunks = p.Kpad / BK;
  load_chunk(0);
  store_chunk(0);
  __syncthreads();

  for (int kc = 0; kc < nchunks; ++kc) {
    const int buf = kc & 1;
    if (kc + 1 < nchunks) load_chunk(kc + 1);  // global loads in flight under the MFMAs

    const float* Ab = As + buf * BM * LDK + (wm * 32 * TM + l31) * LDK + 4 * lh;
    const float* Bb = Bs + buf * BN * LDK + (wn * 32 * TN + l31) * LDK + 4 * lh;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      f32x4 a[TM], b[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i)
        a[i] = *reinterpret_cast<const f32x4*>(Ab + i * 32 * LDK + g * 8);
#pragma unroll
      for (int j = 0; j < TN; ++j)
        b[j] = *reinterpret_cast<const f32x4*>(Bb + j * 32 * LDK + g * 8);
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][s], b[j][s], acc[i][j], 0, 0, 0);
    }

    if (kc + 1 < nchunks) store_chunk(buf ^ 1);
    __syncthreads();
  }

  // ---- epilogue: lane holds column j = lane&31, rows (r&3) + 8*(r>>2) + 4*(lane>>5)
#pragma unroll
  for (int tj = 0; tj < TN; ++tj) {
    const int j = n0 + wn * 32 * TN + tj * 32 + l31;
    const float bj = p.bias[j];  // bias is padded to the tile grid
    const bool vj = j < p.Cout;
#pragma unroll
    for (int ti = 0; ti < TM; ++ti) {
      // residual: issue all 16 loads of this 32x32 tile before the first use (one wait, not 16
      // dependent HBM round trips)
      float rv[16];
      if (p.res) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = m0 + wm * 32 * TM + ti * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
          rv[r] = (m < p.M && vj) ? p.res[(size_t)m * p.res_ld + p.res_off + j] : 0.f;
        }
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm * 32 * TM + ti * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (m < p.M && vj) {
          float v = acc[ti][tj][r] + bj;
          if (p.act) v = silu_f32(v);
          if (p.res) v = (v + rv[r]) * p.post_scale;
          if (j < p.split)
            p.out1[(size_t)m * p.out1_ld + p.out1_off + j] = v;
          else
            p.out2[(size_t)m * p.out2_ld + p.out2_off + (j - p.split)] = v;
          if (p.up) {
            const int n = m / p.HoWo;
            const int rem = m - n * p.HoWo;
            const int oy = rem / p.Wo;
            const int ox = rem - oy * p.Wo;
            const size_t W2 = 2 * (size_t)p.Wo;
            const size_t base = ((size_t)n * 2 * p.Ho + 2 * oy) * W2 + 2 * ox;
            float* u = p.up + p.up_off + j;
            u[base * p.up_ld] = v;
            u[(base + 1) * p.up_ld] = v;
            u[(base + W2) * p.up_ld] = v;
            u[(base + W2 + 1) * p.up_ld] = v;
          }
        }
      }
    }
  }
}

template <int TM, int TN, int WM, int WN>
static int launch_variant(const ConvKArgs& a, int m_tiles, hipStream_t stream) {
  constexpr int BM = 32 * TM * WM, BN = 32 * TN * WN;
  constexpr size_t lds = (size_t)2 * (BM + BN) * LDK * sizeof(float);
  static bool attr_set = false;
  auto kern = conv_igemm_kernel<TM, TN, WM, WN>;
  if (!attr_set) {
    ST_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_set = true;
  }
  dim3 grid((unsigned)(m_tiles * a.n_tiles)), block(64 * WM * WN);
  hipLaunchKernelGGL(kern, grid, block, lds, stream, a);
  ST_CHECK_HIP(hipGetLastError());
  return ST_OK;
}

// Tile variants {BM, BN, threads}: id -> <TM,TN,WM,WN> in the switch of conv2d_launch.
//  0 128x128 (4 waves 64x64)   1 128x64 (2 waves 64x64)   2 128x32 (2 waves 64x32)
//  3  64x64  (4 waves 32x32)   4  64x32 (2 waves 32x32)   5 128x64 (4 waves 32x64)
//  6 128x32  (4 waves 32x32)   7  64x128 (4 waves 32x64)  8 256x64 (4 waves 64x64)
struct ConvVariant { int bm, bn, threads; };
static const ConvVariant kVariants[] = {{128, 128, 256}, {128, 64, 128}, {128, 32, 128}, {64, 64, 256},
                                        {64, 32, 128},   {128, 64, 256}, {128, 32, 256}, {64, 128, 256},
                                        {256, 64, 256}};
constexpr int kNumVariants = 9;

int conv_variant_count() { return kNumVariants; }
bool conv_variant_valid(int id, int cout) {
  return id >= 0 && id < kNumVariants && round_up(cout, 32) % kVariants[id].bn == 0;
}
const char* conv_variant_name(int id) {
  static const char* names[] = {"128x128", "128x64w2", "128x32w2", "64x64", "64x32w2",
                                "128x64", "128x32", "64x128", "256x64"};
  return id >= 0 && id < kNumVariants ? names[id] : "-";
}

int conv2d_launch(const StConvDesc& d, hipStream_t stream, int force_variant, int* picked_variant) {
  ST_REQUIRE(d.in_dev && d.wgt_dev && d.bias_dev && d.out1_dev, "conv: null pointer");
  ST_REQUIRE(d.Cin % 4 == 0 && d.in_ld % 4 == 0 && d.in_off % 4 == 0,
             "conv: Cin/in_ld/in_off must be multiples of 4 (got %d/%d/%d)", d.Cin, d.in_ld,
             d.in_off);
  ST_REQUIRE(d.N > 0 && d.Hi > 0 && d.Wi > 0 && d.Cout > 0 && d.KH > 0 && d.KW > 0 &&
                 d.stride > 0 && d.pad >= 0,
             "conv: bad geometry");
  ST_REQUIRE(d.in_off + d.Cin <= d.in_ld, "conv: input channel slice exceeds in_ld");
  const int Ho = (d.Hi + 2 * d.pad - d.KH) / d.stride + 1;
  const int Wo = (d.Wi + 2 * d.pad - d.KW) / d.stride + 1;
  ST_REQUIRE(Ho > 0 && Wo > 0, "conv: empty output");
  const int split = (d.out2_dev ? d.split : d.Cout);
  ST_REQUIRE(split >= 0 && split <= d.Cout, "conv: bad split");
  ST_REQUIRE(d.out1_off + split <= d.out1_ld, "conv: out1 slice exceeds out1_ld");
  if (d.out2_dev)
    ST_REQUIRE(d.out2_off + (d.Cout - split) <= d.out2_ld, "conv: out2 slice exceeds out2_ld");
  if (d.up_dev) ST_REQUIRE(d.up_off + d.Cout <= d.up_ld, "conv: up slice exceeds up_ld");
  if (d.res_dev) ST_REQUIRE(d.res_off + d.Cout <= d.res_ld, "conv: res slice exceeds res_ld");
  const long long M_ll = (long long)d.N * Ho * Wo;
  ST_REQUIRE(M_ll * (long long)std::max(std::max(d.out1_ld, d.out2_ld), d.in_ld) < (1ll << 40),
             "conv: tensor too large");
  ST_REQUIRE((long long)d.N * d.Hi * d.Wi < (1ll << 31) && M_ll < (1ll << 31),
             "conv: pixel count exceeds int32");

  ConvKArgs a;
  a.in = d.in_dev; a.wgt = d.wgt_dev; a.bias = d.bias_dev;
  a.out1 = d.out1_dev; a.out2 = d.out2_dev; a.up = d.up_dev; a.res = d.res_dev;
  a.Hi = d.Hi; a.Wi = d.Wi; a.Cin = d.Cin; a.in_ld = d.in_ld; a.in_off = d.in_off;
  a.Ho = Ho; a.Wo = Wo; a.HoWo = Ho * Wo; a.Cout = d.Cout;
  a.KW = d.KW; a.stride = d.stride; a.pad = d.pad;
  a.K = d.KH * d.KW * d.Cin; a.Kpad = round_up(a.K, BK); a.M = (int)M_ll;
  a.out1_ld = d.out1_ld; a.out1_off = d.out1_off; a.split = split;
  a.out2_ld = d.out2_ld; a.out2_off = d.out2_off;
  a.up_ld = d.up_ld; a.up_off = d.up_off;
  a.res_ld = d.res_ld; a.res_off = d.res_off;
  a.post_scale = d.res_dev ? d.post_scale : 1.0f;
  a.act = d.act;

  const int cout_pad = round_up(d.Cout, 32);
  int pick = -1;
  if (force_variant >= 0) {
    ST_REQUIRE(conv_variant_valid(force_variant, d.Cout), "conv: variant %d does not divide Cout=%d",
               force_variant, d.Cout);
    pick = force_variant;
  } else {
    // untuned default: the largest tile that still gives >= 2 blocks per CU (the detector
    // replaces this guess by a measured choice, st_detector_autotune)
    static const int order[] = {0, 5, 6, 3, 4};
    long long best_blocks = -1;
    for (int id : order) {
      if (!conv_variant_valid(id, d.Cout)) continue;
      const long long blocks = (long long)ceil_div(a.M, kVariants[id].bm) * (cout_pad / kVariants[id].bn);
      if (blocks >= 512) { pick = id; break; }
      if (blocks > best_blocks) { best_blocks = blocks; pick = id; }
    }
  }
  const ConvVariant& v = kVariants[pick];
  if (picked_variant) *picked_variant = pick;
  a.n_tiles = cout_pad / v.bn;
  const int m_tiles = ceil_div(a.M, v.bm);
  switch (pick) {
    case 0: return launch_variant<2, 2, 2, 2>(a, m_tiles, stream);
    case 1: return launch_variant<2, 2, 2, 1>(a, m_tiles, stream);
    case 2: return launch_variant<2, 1, 2, 1>(a, m_tiles, stream);
    case 3: return launch_variant<1, 1, 2, 2>(a, m_tiles, stream);
    case 4: return launch_variant<1, 1, 2, 1>(a, m_tiles, stream);
    case 5: return launch_variant<1, 2, 4, 1>(a, m_tiles, stream);
    case 6: return launch_variant<1, 1, 4, 1>(a, m_tiles, stream);
    case 7: return launch_variant<1, 2, 2, 2>(a, m_tiles, stream);
    default: return launch_variant<2, 2, 4, 1>(a, m_tiles, stream);
  }
}

}  // namespace st

extern "C" int st_conv2d_nhwc(const StConvDesc* d, st_stream_t stream) {
  if (!d) return st::set_error(ST_ERR_INVALID, "st_conv2d_nhwc: null desc");
  return st::conv2d_launch(*d, static_cast<hipStream_t>(stream), -1, nullptr);
}

// test hook: force a tile variant (0..4); not part of the documented ABI surface
extern "C" int st_conv2d_nhwc_variant(const StConvDesc* d, st_stream_t stream, int variant) {
  if (!d) return st::set_error(ST_ERR_INVALID, "st_conv2d_nhwc_variant: null desc");
  return st::conv2d_launch(*d, static_cast<hipStream_t>(stream), variant, nullptr);
}
