// 3-D aggregation of the stereo cost volume (north_star: "its 3D/2D aggregation"): single-channel 3x3x3 convolutions
// over (d, y, x) of the materialised volume [N][Hf][Wf][D], zero padded, optional SiLU.  The reference has no such
// function (its disparity is an offline product, reproducibility.md:166-194); the specification is
// oracle/st_oracle.c::oracle_agg3d and this kernel is BIT-EXACT against it (same fmaf order: rows j, columns k,
// disparity taps i; padded taps contribute fmaf(w, 0, acc); SiLU through the oracle's exp polynomial).
//
// HBM-bound stencil (27 FMAs per cell against 8 bytes of traffic): one workgroup owns one row segment of TW pixels
// of one image and stages the 3 x (TW + 2) pixel rows x D floats it needs in LDS once (coalesced 16-byte loads, zero
// fill = the padding in y / x); a thread produces 4 consecutive disparities of one pixel from 9 aligned 16-byte LDS
// reads + the two neighbours across the quad borders, and stores 16 bytes.  Every volume element is fetched from
// memory 3 x (TW + 2) / TW = 3.1 times per layer through L2 (its two neighbour rows are other workgroups' tiles).
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

constexpr int A3_TW = 64;   // pixels of one row per workgroup

struct Agg3dArgs {
  const float* in;
  float* out;
  int N, Hf, Wf, D;
  float w[27];   // [i = kD][j = kH][k = kW]
  float bias;
  int act;
};

__global__ __launch_bounds__(256) void agg3d_kernel(const Agg3dArgs a) {
  extern __shared__ float4 a3_smem4[];
  float* lds = reinterpret_cast<float*>(a3_smem4);
  const int D = a.D, DQ = D >> 2;
  const int x0 = blockIdx.x * A3_TW, y = blockIdx.y, n = blockIdx.z;
  const int tid = threadIdx.x;
  constexpr int TC = A3_TW + 2;
  // ---- stage rows y-1 .. y+1, columns x0-1 .. x0+TW, all D disparities (zero outside the image)
  const int nload = 3 * TC * DQ;
  for (int e = tid; e < nload; e += 256) {
    const int q = e % DQ, pc = e / DQ;
    const int c = pc % TC, r = pc / TC;
    const int gy = y + r - 1, gx = x0 + c - 1;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (gy >= 0 && gy < a.Hf && gx >= 0 && gx < a.Wf)
      v = *reinterpret_cast<const f32x4*>(a.in + (((size_t)n * a.Hf + gy) * a.Wf + gx) * D + 4 * q);
    *reinterpret_cast<f32x4*>(lds + (size_t)pc * D + 4 * q) = v;
  }
  __syncthreads();
  const int nitem = A3_TW * DQ;
  for (int it = tid; it < nitem; it += 256) {
    const int q = it % DQ, px = it / DQ;
    if (x0 + px >= a.Wf) continue;
    float acc[4] = {a.bias, a.bias, a.bias, a.bias};
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const float* p = lds + (size_t)(j * TC + px + k) * D + 4 * q;
        const f32x4 v = *reinterpret_cast<const f32x4*>(p);
        const float vm = q > 0 ? p[-1] : 0.0f;
        const float vp = q < DQ - 1 ? p[4] : 0.0f;
        const float vals[6] = {vm, v[0], v[1], v[2], v[3], vp};
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int i = 0; i < 3; ++i) acc[e] = fmaf(a.w[(i * 3 + j) * 3 + k], vals[e + i], acc[e]);
      }
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = a.act ? acc[e] / (1.0f + a3_expf(-acc[e])) : acc[e];
    *reinterpret_cast<f32x4*>(a.out + (((size_t)n * a.Hf + y) * a.Wf + x0 + px) * D + 4 * q) = o;
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
  const int lds = 3 * (A3_TW + 2) * D * (int)sizeof(float);
  static int lds_set = 0;
  ST_ENSURE_DYNAMIC_LDS(agg3d_kernel, lds, lds_set);
  hipLaunchKernelGGL(agg3d_kernel, dim3((unsigned)ceil_div(Wf, A3_TW), (unsigned)Hf, (unsigned)N), dim3(256), lds,
                     static_cast<hipStream_t>(stream_), a);
  ST_CHECK_HIP(hipGetLastError());
  return ST_OK;
}
