// Sustained fp32 MFMA rate on gfx950: pure loops of independent v_mfma_f32_16x16x4_f32 / v_mfma_f32_32x32x2_f32,
// W waves per SIMD, every CU busy, for ~0.3 ms (the duration of the conv kernels of the path).  Prints TFLOP/s.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_peak.hip -o /tmp/mfma_peak && /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ void k16(float* out, int iters, float a0) {
  f32x4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  float a = a0 + threadIdx.x, b = a0 * 0.5f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  if (s == 12345.678f) out[threadIdx.x] = s;
}

template <int NACC>
__global__ void k32(float* out, int iters, float a0) {
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i)
    for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
  float a = a0 + threadIdx.x, b = a0 * 0.5f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < NACC; ++i)
    for (int j = 0; j < 16; ++j) s += acc[i][j];
  if (s == 12345.678f) out[threadIdx.x] = s;
}

// closer to the conv kernels: 16 MFMAs per step with DISTINCT A registers (4 x f32x4) and a B register per k,
// MODE 0: operands loaded once; 1: A fragments re-read from LDS every step (double-buffered, ds_read_b128 x 4);
// 2: 1 + one global buffer-style load of the next B fragment per step
template <int MODE>
__global__ __launch_bounds__(512) void kstep(const float* in, float* out, int iters) {
  __shared__ float4 lds[64 * 4 * 18];
  const int lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 64 * 4 * 18; i += blockDim.x) lds[i] = float4{1.f + i, 2.f, 3.f, 4.f};
  __syncthreads();
  f32x4 acc[4];
  for (int i = 0; i < 4; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  f32x4 wf[2][4], xf[18];
  for (int cb = 0; cb < 4; ++cb) wf[0][cb] = *reinterpret_cast<const f32x4*>(&lds[cb * 64 + lane]);
  for (int k = 0; k < 18; ++k) xf[k] = *reinterpret_cast<const f32x4*>(in + (k * 64 + lane) * 4);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < 18; ++k) {
      if (MODE >= 1) {
#pragma unroll
        for (int cb = 0; cb < 4; ++cb)
          wf[(k + 1) & 1][cb] = *reinterpret_cast<const f32x4*>(&lds[(((k + 1) % 18) * 4 + cb) * 64 + lane]);
      } else {
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) wf[(k + 1) & 1][cb] = wf[k & 1][cb];
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int cb = 0; cb < 4; ++cb)
          acc[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[k & 1][cb][s], xf[k][s], acc[cb], 0, 0, 0);
      if (MODE >= 2) xf[k] = *reinterpret_cast<const f32x4*>(in + ((((it + 1) & 3) * 18 + k) * 64 + lane) * 4);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  float s = 0.f;
  for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  if (s == 12345.678f) out[threadIdx.x] = s;
}

// VALU next to MFMA: 18 steps x 16 MFMAs per "tile" as above (operands in registers) plus 40 SiLU evaluations
// (v_exp + v_rcp + 3 simple ops each) per tile, either as ONE block after the 288 MFMAs (an epilogue: MODE 0) or
// spread through the MFMA stream, ~2 per step (MODE 1), or none (MODE 2).
template <int MODE>
__global__ __launch_bounds__(512) void kvalu(const float* in, float* out, int iters) {
  const int lane = threadIdx.x & 63;
  f32x4 acc[4];
  for (int i = 0; i < 4; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  f32x4 wf[4], xf;
  for (int cb = 0; cb < 4; ++cb) wf[cb] = *reinterpret_cast<const f32x4*>(in + (cb * 64 + lane) * 4);
  xf = *reinterpret_cast<const f32x4*>(in + (5 * 64 + lane) * 4);
  float v[40];
  for (int i = 0; i < 40; ++i) v[i] = in[lane + i];
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < 18; ++k) {
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) {
          acc[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[cb][s], xf[s], acc[cb], 0, 0, 0);
          if (MODE == 1) {
            const int j = k * 16 + s * 4 + cb;   // 0 .. 287: one SiLU every 7th MFMA
            if (j % 7 == 0 && j / 7 < 40) {
              const float t = v[j / 7];
              v[j / 7] = t * __builtin_amdgcn_rcpf(1.0f + __expf(-t)) + 0.25f;
            }
          }
        }
    }
    if (MODE == 0) {
#pragma unroll
      for (int i = 0; i < 40; ++i) v[i] = v[i] * __builtin_amdgcn_rcpf(1.0f + __expf(-v[i])) + 0.25f;
    }
  }
  float s = 0.f;
  for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  for (int i = 0; i < 40; ++i) s += v[i];
  if (s == 12345.678f) out[threadIdx.x] = s;
}

template <typename F>
static float time_ms(F launch, int reps) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  launch();
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int i = 0; i < reps; ++i) launch();
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  return ms / reps;
}

int main() {
  int cus = 0;
  hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
  float* out;
  hipMalloc(&out, 4096);
  printf("CUs %d\n", cus);
  for (int wps = 1; wps <= 4; wps *= 2) {   // waves per SIMD
    const int threads = 256 * wps;           // one workgroup per CU
    const int iters16 = 2400 / wps, iters32 = 1200 / wps;
    {
      float ms = time_ms([&] { hipLaunchKernelGGL(k16<4>, dim3(cus), dim3(threads), 0, 0, out, iters16, 1.0f); }, 20);
      double flop = (double)cus * (threads / 64) * iters16 * 16 * 2048.0;
      printf("16x16x4 f32, 4 accumulators, %d waves/SIMD: %.1f us/launch  %.1f TFLOP/s\n", wps, ms * 1e3, flop / ms / 1e9);
    }
    {
      float ms = time_ms([&] { hipLaunchKernelGGL(k32<4>, dim3(cus), dim3(threads), 0, 0, out, iters32, 1.0f); }, 20);
      double flop = (double)cus * (threads / 64) * iters32 * 16 * 4096.0;
      printf("32x32x2 f32, 4 accumulators, %d waves/SIMD: %.1f us/launch  %.1f TFLOP/s\n", wps, ms * 1e3, flop / ms / 1e9);
    }
  }
  {
    float* in;
    hipMalloc(&in, 4 * 18 * 64 * 16);
    hipMemset(in, 0, 4 * 18 * 64 * 16);
    const int iters = 30;
    const double flop = (double)cus * 8 * iters * 18 * 16 * 2048.0;
    float ms = time_ms([&] { hipLaunchKernelGGL(kstep<0>, dim3(cus), dim3(512), 0, 0, in, out, iters); }, 20);
    printf("conv-like steps, operands in registers:           %.1f us  %.1f TFLOP/s\n", ms * 1e3, flop / ms / 1e9);
    ms = time_ms([&] { hipLaunchKernelGGL(kstep<1>, dim3(cus), dim3(512), 0, 0, in, out, iters); }, 20);
    printf("conv-like steps, A fragments from LDS every step:  %.1f us  %.1f TFLOP/s\n", ms * 1e3, flop / ms / 1e9);
    ms = time_ms([&] { hipLaunchKernelGGL(kstep<2>, dim3(cus), dim3(512), 0, 0, in, out, iters); }, 20);
    printf("conv-like steps, + next B fragment from global:    %.1f us  %.1f TFLOP/s\n", ms * 1e3, flop / ms / 1e9);
  }
  {
    float* in;
    hipMalloc(&in, 1 << 16);
    hipMemset(in, 0, 1 << 16);
    const int iters = 30;
    const double flop = (double)cus * 8 * iters * 18 * 16 * 2048.0;
    float ms = time_ms([&] { hipLaunchKernelGGL(kvalu<2>, dim3(cus), dim3(512), 0, 0, in, out, iters); }, 20);
    printf("288 MFMAs per tile, no VALU work:                      %.1f us  %.1f TFLOP/s\n", ms * 1e3, flop / ms / 1e9);
    ms = time_ms([&] { hipLaunchKernelGGL(kvalu<0>, dim3(cus), dim3(512), 0, 0, in, out, iters); }, 20);
    printf("288 MFMAs + 40 SiLU as a block after them (epilogue):  %.1f us  %.1f TFLOP/s\n", ms * 1e3, flop / ms / 1e9);
    ms = time_ms([&] { hipLaunchKernelGGL(kvalu<1>, dim3(cus), dim3(512), 0, 0, in, out, iters); }, 20);
    printf("288 MFMAs + 40 SiLU spread through the MFMA stream:    %.1f us  %.1f TFLOP/s\n", ms * 1e3, flop / ms / 1e9);
  }
  {  // a long run: 2 waves/SIMD, ~20 ms of back-to-back launches (power / clock steady state)
    float ms = time_ms([&] { hipLaunchKernelGGL(k16<4>, dim3(cus), dim3(512), 0, 0, out, 1200, 1.0f); }, 200);
    double flop = (double)cus * 8 * 1200 * 16 * 2048.0;
    printf("16x16x4 f32 sustained (200 launches): %.1f us/launch  %.1f TFLOP/s\n", ms * 1e3, flop / ms / 1e9);
  }
  return 0;
}
