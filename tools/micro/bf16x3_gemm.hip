// GROUNDWORK (tools only, not on the product path): fp32 GEMM on the BF16 matrix pipes by error-free operand
// splitting ("bf16x3").  x = hi + mid + lo with three bf16 terms carries 24 mantissa bits, so
//     a * b  ~=  hi_a hi_b + hi_a mid_b + mid_a hi_b + hi_a lo_b + lo_a hi_b + mid_a mid_b      (6 of the 9 products;
// the dropped ones are below 2^-24 relative), each product exact in the bf16 MFMA's fp32 accumulator.  gfx950 runs
// v_mfma_f32_16x16x32_bf16 at 16x the rate of the fp32-input MFMA (MI355X_MICROARCH.md, Matrix cores), so 6 bf16
// instructions replace 8 fp32 ones of a quarter of the cost each... IF the split (about 5 vector instructions per
// element) hides beside the matrix pipe.  This program measures exactly that, on the shape of a pointwise conv of the
// path (117 760 pixels x 128 -> 128 channels, `neck.out_layers.0`): the structure of pointwise_resident.hip (weights
// resident in LDS, pixels register-fed, swapped operands, 32 pixels per wave step), once with v_mfma_f32_16x16x4_f32
// and once split; it prints both times and the error of both against a float64 reference.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/micro/bf16x3_gemm.hip -o /tmp/bf16x3 && /tmp/bf16x3
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int K = 128, N = 128;   // Cin, Cout
constexpr int WAVES = 8, THREADS = 64 * WAVES;
constexpr int CB = N / 16;        // cout blocks of 16

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// ---- fp32 baseline: v_mfma_f32_16x16x4_f32, A = weights [16 couts x 4 k] from LDS, B = pixels [4 k x 16 px] ----------
__global__ __launch_bounds__(THREADS, 1) void gemm_f32(const float* __restrict__ X, const float* __restrict__ W,
                                                        float* __restrict__ Y, int M, int ntiles) {
  extern __shared__ float4 smem4[];
  float* wl = reinterpret_cast<float*>(smem4);   // [cb][g16][lane][4]: W[16cb + i][16g + 4kq + e]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i16 = lane & 15, kq = lane >> 4;
  constexpr int KG = K / 16;
  for (int f = wave; f < CB * KG; f += WAVES) {
    const int cb = f / KG, g = f - cb * KG;
    *reinterpret_cast<f32x4*>(wl + (f * 64 + lane) * 4) =
        *reinterpret_cast<const f32x4*>(W + (size_t)(cb * 16 + i16) * K + 16 * g + 4 * kq);
  }
  __syncthreads();
  for (int tile = blockIdx.x * WAVES + wave; tile < ntiles; tile += gridDim.x * WAVES) {
    const int m = tile * 16 + i16;
    f32x4 xf[KG];
#pragma unroll
    for (int g = 0; g < KG; ++g)
      xf[g] = m < M ? *reinterpret_cast<const f32x4*>(X + (size_t)m * K + 16 * g + 4 * kq) : f32x4{0, 0, 0, 0};
    f32x4 acc[CB];
#pragma unroll
    for (int cb = 0; cb < CB; ++cb) acc[cb] = f32x4{0, 0, 0, 0};
#pragma unroll
    for (int g = 0; g < KG; ++g) {
      f32x4 wf[CB];
#pragma unroll
      for (int cb = 0; cb < CB; ++cb) wf[cb] = *reinterpret_cast<const f32x4*>(wl + ((cb * KG + g) * 64 + lane) * 4);
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int cb = 0; cb < CB; ++cb) acc[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[cb][s], xf[g][s], acc[cb], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);   // keep the compiler from hoisting every step's operand reads (spills)
    }
    if (m < M) {
#pragma unroll
      for (int cb = 0; cb < CB; ++cb) *reinterpret_cast<f32x4*>(Y + (size_t)m * N + cb * 16 + 4 * kq) = acc[cb];
    }
  }
}

// ---- split: three bf16 terms per operand, 6 products on v_mfma_f32_16x16x32_bf16 -------------------------------------
__device__ __forceinline__ void split3(const f32x4 a, const f32x4 b, bf16x8& hi, bf16x8& mid, bf16x8& lo) {
  const float x[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const __bf16 h = (__bf16)x[i];
    const float r = x[i] - (float)h;      // exact: h is x rounded to 8 bits
    const __bf16 m = (__bf16)r;
    const float r2 = r - (float)m;        // exact
    hi[i] = h; mid[i] = m; lo[i] = (__bf16)r2;
  }
}

// K slot (kq, e) of the 32-channel block g holds channel 32g + 16 (e / 4) + 4 kq + (e % 4): the two 16-byte loads of a
// lane (and, for a chained conv, two accumulator quads of the same lane) - the weights use the same permutation.
template <int SEPARATE>   // 1: the five small products go to accumulators of their own, added once at the end
__global__ __launch_bounds__(THREADS, 1) void gemm_split(const float* __restrict__ X, const float* __restrict__ W,
                                                          float* __restrict__ Y, int M, int ntiles) {
  extern __shared__ float4 smem4[];
  bf16x8* wl = reinterpret_cast<bf16x8*>(smem4);   // [part][cb][g32][lane]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i16 = lane & 15, kq = lane >> 4;
  constexpr int KG = K / 32;
  for (int f = wave; f < CB * KG; f += WAVES) {     // split the fp32 weights once per workgroup
    const int cb = f / KG, g = f - cb * KG;
    const float* wr = W + (size_t)(cb * 16 + i16) * K + 32 * g + 4 * kq;
    bf16x8 h, m, l;
    split3(*reinterpret_cast<const f32x4*>(wr), *reinterpret_cast<const f32x4*>(wr + 16), h, m, l);
    wl[(0 * CB * KG + f) * 64 + lane] = h;
    wl[(1 * CB * KG + f) * 64 + lane] = m;
    wl[(2 * CB * KG + f) * 64 + lane] = l;
  }
  __syncthreads();
  // a wave step = 32 pixels (two MFMA column blocks): every weight fragment read from LDS feeds two MFMAs
  for (int tile = blockIdx.x * WAVES + wave; tile < ntiles; tile += gridDim.x * WAVES) {
    f32x4 acc[2][CB], corr[SEPARATE ? 2 : 1][SEPARATE ? CB : 1];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int cb = 0; cb < CB; ++cb) {
        acc[t][cb] = f32x4{0, 0, 0, 0};
        if (SEPARATE) corr[t][cb] = f32x4{0, 0, 0, 0};
      }
#pragma unroll
    for (int g = 0; g < KG; ++g) {
      bf16x8 xh[2], xm[2], xl[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int m = tile * 32 + t * 16 + i16;
        f32x4 a{0, 0, 0, 0}, b{0, 0, 0, 0};
        if (m < M) {
          a = *reinterpret_cast<const f32x4*>(X + (size_t)m * K + 32 * g + 4 * kq);
          b = *reinterpret_cast<const f32x4*>(X + (size_t)m * K + 32 * g + 16 + 4 * kq);
        }
        split3(a, b, xh[t], xm[t], xl[t]);
      }
#pragma unroll
      for (int cb = 0; cb < CB; ++cb) {
        const bf16x8 wh = wl[((0 * CB + cb) * KG + g) * 64 + lane];
        const bf16x8 wm = wl[((1 * CB + cb) * KG + g) * 64 + lane];
        const bf16x8 wo = wl[((2 * CB + cb) * KG + g) * 64 + lane];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          f32x4& c = SEPARATE ? corr[t][cb] : acc[t][cb];
          c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wm, xm[t], c, 0, 0, 0);     // smallest terms first
          c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wo, xh[t], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, xl[t], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wm, xh[t], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, xm[t], c, 0, 0, 0);
          acc[t][cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, xh[t], acc[t][cb], 0, 0, 0);
        }
        if ((cb & 1) == 1) __builtin_amdgcn_sched_barrier(0);   // bound the fragment reads in flight (no spills)
      }
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int m = tile * 32 + t * 16 + i16;
      if (m < M) {
#pragma unroll
        for (int cb = 0; cb < CB; ++cb) {
          f32x4 v = acc[t][cb];
          if (SEPARATE) v = v + corr[t][cb];
          *reinterpret_cast<f32x4*>(Y + (size_t)m * N + cb * 16 + 4 * kq) = v;
        }
      }
    }
  }
}

int main() {
  const int M = 8 * 92 * 160;
  std::vector<float> hx((size_t)M * K), hw((size_t)N * K);
  srand(1);
  auto rnd = [] { return (float)rand() / RAND_MAX * 2.f - 1.f; };
  for (auto& v : hx) { const float a = rnd(), b = rnd(); v = 3.0f * a * fabsf(b); }   // activation-like magnitudes
  for (auto& v : hw) v = rnd() / sqrtf((float)K) * 1.7f;
  float *dx, *dw, *dy;
  CK(hipMalloc(&dx, hx.size() * 4)); CK(hipMalloc(&dw, hw.size() * 4)); CK(hipMalloc(&dy, (size_t)M * N * 4));
  CK(hipMemcpy(dx, hx.data(), hx.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dw, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
  int cus = 0;
  CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
  const int rows = 512;   // reference rows (float64 on the host)
  std::vector<double> ref((size_t)rows * N);
  for (int r = 0; r < rows; ++r) {
    const size_t m = (size_t)r * (M / rows);
    for (int n = 0; n < N; ++n) {
      double s = 0;
      for (int k = 0; k < K; ++k) s += (double)hx[m * K + k] * (double)hw[(size_t)n * K + k];
      ref[(size_t)r * N + n] = s;
    }
  }
  std::vector<float> hy((size_t)M * N);
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const double flop = 2.0 * M * K * N;
  for (int mode = 0; mode < 3; ++mode) {
    const int lds = mode == 0 ? CB * (K / 16) * 256 * 4 : 3 * CB * (K / 32) * 64 * 16;
    const int ntiles = mode == 0 ? (M + 15) / 16 : (M + 31) / 32;
    auto launch = [&] {
      if (mode == 0) hipLaunchKernelGGL(gemm_f32, dim3(cus), dim3(THREADS), lds, 0, dx, dw, dy, M, ntiles);
      else if (mode == 1) hipLaunchKernelGGL(gemm_split<0>, dim3(cus), dim3(THREADS), lds, 0, dx, dw, dy, M, ntiles);
      else hipLaunchKernelGGL(gemm_split<1>, dim3(cus), dim3(THREADS), lds, 0, dx, dw, dy, M, ntiles);
    };
    if (mode == 0) CK(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_f32), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    if (mode == 1) CK(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_split<0>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    if (mode == 2) CK(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_split<1>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    CK(hipMemset(dy, 0xFF, (size_t)M * N * 4));
    for (int i = 0; i < 5; ++i) launch();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    const int reps = 50;
    for (int i = 0; i < reps; ++i) launch();
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipMemcpy(hy.data(), dy, hy.size() * 4, hipMemcpyDeviceToHost));
    double emax = 0, esq = 0, scale = 0;
    for (int r = 0; r < rows; ++r) {
      const size_t m = (size_t)r * (M / rows);
      for (int n = 0; n < N; ++n) {
        const double d = (double)hy[m * N + n] - ref[(size_t)r * N + n];
        emax = fmax(emax, fabs(d)); esq += d * d; scale = fmax(scale, fabs(ref[(size_t)r * N + n]));
      }
    }
    const double us = ms / reps * 1e3;
    printf("%-44s %7.1f us  %6.1f TFLOP/s (fp32-equivalent)  max err %.3e  rms err %.3e  (of max |y| %.2f)\n",
           mode == 0 ? "fp32 MFMA 16x16x4 (exact fp32)" : mode == 1 ? "bf16x3, 6 products, one accumulator"
                                                                    : "bf16x3, small products in their own accumulator",
           us, flop / (us * 1e-6) / 1e12, emax / scale, sqrt(esq / (rows * (double)N)) / scale, scale);
  }
  printf("HBM floor of this layer: %.1f us at 6.29 TB/s (in %d MB + out %d MB)\n",
         ((double)M * (K + N) * 4) / 6.29e12 * 1e6, (int)((size_t)M * K * 4 >> 20), (int)((size_t)M * N * 4 >> 20));
  return 0;
}
