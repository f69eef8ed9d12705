// 3-D aggregation of the stereo cost volume (north_star: "its 3D/2D aggregation"): single-channel 3x3x3 convolutions
// over (d, y, x) of the materialised volume [N][Hf][Wf][D], zero padded, optional SiLU.  The reference has no such
// function (its disparity is an offline product, reproducibility.md:166-194); the specification is
// oracle/st_oracle.c::oracle_agg3d and this kernel is BIT-EXACT against it (same fmaf order: rows j, columns k,
// disparity taps i; padded taps contribute fmaf(w, 0, acc); SiLU through the oracle's exp polynomial).
//
// HBM-bound stencil (27 FMAs per cell against 8 bytes of traffic), written as a STREAMING stencil: one workgroup owns a
// column strip of TW pixels x all D disparities and walks down a band of RY rows, keeping a ring of four pixel rows
// ((TW + 2) x D floats each) in LDS - rows y-1, y, y+1 feed output row y while row y+2 travels from memory through
// registers into the fourth slot (one barrier per row).  A volume element is fetched (TW + 2) / TW x (RY + 2) / RY
// times per layer (1.2x at the bench volume; the round-4 kernel staged three rows per output row: 3.1x).  A thread
// produces 4 consecutive disparities of one pixel from 9 aligned 16-byte LDS reads + the two neighbours across the quad
// borders, and stores 16 bytes.
#include <algorithm>

#include "st_common.h"

namespace st {
namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float a3_expf(float x) {  // same polynomial as decode_nms.hip / costvolume.hip / oracle
  if (x > 88.72283f) return __builtin_inff();
  if (x < -103.0f) return 0.0f;
  const float n = rintf(x * 1.44269504088896341f);
  float r = fmaf(n, -0.693359375f, x);
  r = fmaf(n, 2.12194440e-4f, r);
  float p = 1.9875691500e-4f;
  p = fmaf(p, r, 1.3981999507e-3f);
  p = fmaf(p, r, 8.3334519073e-3f);
  p = fmaf(p, r, 4.1665795894e-2f);
  p = fmaf(p, r, 1.6666665459e-1f);
  p = fmaf(p, r, 5.0000001201e-1f);
  const float r2 = r * r;
  p = fmaf(p, r2, r);
  p = p + 1.0f;
  return ldexpf(p, (int)n);
}

struct Agg3dArgs {
  const float* in;
  float* out;
  int N, Hf, Wf, D;
  int RY;        // rows per band
  float w[27];   // [i = kD][j = kH][k = kW]
  float bias;
  int act;
};

// TW: pixels per strip; NST: float4 staging registers per thread = ceil((TW + 2) * (D / 4) / 256)
template <int TW, int NST>
__global__ __launch_bounds__(256) void agg3d_kernel(const Agg3dArgs a) {
  extern __shared__ float4 a3_smem4[];
  float* lds = reinterpret_cast<float*>(a3_smem4);
  const int D = a.D, DQ = D >> 2;
  constexpr int TC = TW + 2;
  const int x0 = blockIdx.x * TW, y0 = blockIdx.y * a.RY, n = blockIdx.z;
  const int y1 = min(y0 + a.RY, a.Hf);
  const int tid = threadIdx.x;
  const int rowf = TC * D;          // floats per staged pixel row
  const int nrow4 = TC * DQ;        // float4 per staged pixel row
  f32x4 st[NST];
  // one pixel row gy (columns x0-1 .. x0+TW, zero outside the image) from memory into registers / registers into slot
  auto load_row_into = [&](int gy, f32x4 (&r)[NST]) {
#pragma unroll
    for (int i = 0; i < NST; ++i) {
      const int e = tid + 256 * i;
      const int q = e % DQ, c = e / DQ;
      const int gx = x0 + c - 1;
      r[i] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (e < nrow4 && gy >= 0 && gy < a.Hf && gx >= 0 && gx < a.Wf)
        r[i] = *reinterpret_cast<const f32x4*>(a.in + (((size_t)n * a.Hf + gy) * a.Wf + gx) * D + 4 * q);
    }
  };
  auto store_row_from = [&](int gy, const f32x4 (&r)[NST]) {
    float* dst = lds + (size_t)((gy + 4) & 3) * rowf;
#pragma unroll
    for (int i = 0; i < NST; ++i) {
      const int e = tid + 256 * i;
      if (e < nrow4) *reinterpret_cast<f32x4*>(dst + 4 * e) = r[i];
    }
  };
  auto load_row = [&](int gy) { load_row_into(gy, st); };
  auto store_row = [&](int gy) { store_row_from(gy, st); };
  {   // prologue: the three rows of the first output row with all their loads in flight at once (one memory round trip)
    f32x4 p0[NST], p1[NST];
    load_row_into(y0 - 1, p0);
    load_row_into(y0, p1);
    load_row(y0 + 1);
    store_row_from(y0 - 1, p0);
    store_row_from(y0, p1);
    store_row(y0 + 1);
  }
  __syncthreads();
  const int nitem = TW * DQ;
  for (int y = y0; y < y1; ++y) {
    const bool more = y + 1 < y1;
    if (more) load_row(y + 2);      // in flight while this row is computed
    for (int it = tid; it < nitem; it += 256) {
      const int q = it % DQ, px = it / DQ;
      if (x0 + px >= a.Wf) continue;
      float acc[4] = {a.bias, a.bias, a.bias, a.bias};
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const float* row = lds + (size_t)((y + j - 1 + 4) & 3) * rowf;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          const float* p = row + (size_t)(px + k) * D + 4 * q;
          const f32x4 v = *reinterpret_cast<const f32x4*>(p);
          const float vm = q > 0 ? p[-1] : 0.0f;
          const float vp = q < DQ - 1 ? p[4] : 0.0f;
          const float vals[6] = {vm, v[0], v[1], v[2], v[3], vp};
#pragma unroll
          for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int i = 0; i < 3; ++i) acc[e] = fmaf(a.w[(i * 3 + j) * 3 + k], vals[e + i], acc[e]);
        }
      }
      f32x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = a.act ? acc[e] / (1.0f + a3_expf(-acc[e])) : acc[e];
      *reinterpret_cast<f32x4*>(a.out + (((size_t)n * a.Hf + y) * a.Wf + x0 + px) * D + 4 * q) = o;
    }
    // slot (y + 2) & 3 held row y - 2: last read while row y - 1 was computed, i.e. before the previous barrier
    if (more) store_row(y + 2);
    __syncthreads();
  }
}

}  // namespace
}  // namespace st

extern "C" int st_volume_agg3d(const float* vol_in_dev, float* vol_out_dev, int N, int Hf, int Wf, int D,
                               const float* weight27_host, float bias, int act, st_stream_t stream_) {
  using namespace st;
  ST_REQUIRE(vol_in_dev && vol_out_dev && weight27_host && vol_in_dev != vol_out_dev, "st_volume_agg3d: bad pointer");
  ST_REQUIRE(N > 0 && Hf > 0 && Wf > 0 && D >= 4 && D % 4 == 0 && D <= 192,
             "st_volume_agg3d: D must be a multiple of 4 in [4, 192] (got %d)", D);
  ST_REQUIRE(((reinterpret_cast<uintptr_t>(vol_in_dev) | reinterpret_cast<uintptr_t>(vol_out_dev)) & 15) == 0,
             "st_volume_agg3d: volumes must be 16-byte aligned");
  ST_REQUIRE(Hf < 65536 && N < 65536, "st_volume_agg3d: grid too large");
  Agg3dArgs a;
  a.in = vol_in_dev; a.out = vol_out_dev; a.N = N; a.Hf = Hf; a.Wf = Wf; a.D = D;
  for (int i = 0; i < 27; ++i) a.w[i] = weight27_host[i];
  a.bias = bias; a.act = act;
  // strip width by LDS budget (4 rows x (TW + 2) x D floats): 64 pixels up to 48 levels (51 KB), 32 up to 96 (52 KB),
  // 16 beyond (55 KB at D = 192) - two to three workgroups per CU in every case
  const int TW = D <= 48 ? 64 : (D <= 96 ? 32 : 16);
  const int strips = ceil_div(Wf, TW);
  // Band height.  A workgroup's prologue (three rows) and its halo rows are overhead per band, a half-empty last launch
  // round is overhead per launch: pick the band count b (rows RY = ceil(Hf / b) >= 4) that maximises
  // (workgroups / slots rounded up to whole rounds) x RY / (RY + 3), slots = 256 CUs x workgroups per CU by LDS.
  const int lds = 4 * (TW + 2) * D * (int)sizeof(float);
  const long long slots = 256ll * std::max(1, (160 * 1024) / lds);
  int best_b = 1;
  double best_e = -1.0;
  for (int b = 1; b <= 64 && ceil_div(Hf, b) >= 4; ++b) {
    const int ry = ceil_div(Hf, b), nb = ceil_div(Hf, ry);
    const long long wgs = (long long)N * strips * nb;
    const double fill = (double)wgs / (double)(ceil_div((int)std::min<long long>(wgs, 1 << 30), (int)slots) * slots);
    const double e = fill * ry / (ry + 3.0);
    if (e > best_e + 1e-9) { best_e = e; best_b = b; }
  }
  int RY = std::min(Hf, ceil_div(Hf, best_b));
  a.RY = RY;
  const int bands = ceil_div(Hf, RY);
  ST_REQUIRE(bands < 65536, "st_volume_agg3d: grid too large");
  const dim3 grid((unsigned)strips, (unsigned)bands, (unsigned)N);
  hipStream_t stream = static_cast<hipStream_t>(stream_);
#define ST_A3_LAUNCH(TWV, NSTV)                                                                      \
  do {                                                                                               \
    auto kern = agg3d_kernel<TWV, NSTV>;                                                             \
    static int lds_set = 0;                                                                          \
    ST_ENSURE_DYNAMIC_LDS(kern, lds, lds_set);                                                       \
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, stream, a);                                       \
  } while (0)
  const int nst = ceil_div((TW + 2) * (D / 4), 256);
  if (TW == 64) {            // D <= 48: (66 * 12) / 256 -> up to 4
    if (nst <= 2) ST_A3_LAUNCH(64, 2); else ST_A3_LAUNCH(64, 4);
  } else if (TW == 32) {     // D <= 96: (34 * 24) / 256 -> 4
    ST_A3_LAUNCH(32, 4);
  } else {                   // D <= 192: (18 * 48) / 256 -> 4
    ST_A3_LAUNCH(16, 4);
  }
#undef ST_A3_LAUNCH
  ST_CHECK_HIP(hipGetLastError());
  return ST_OK;
}
