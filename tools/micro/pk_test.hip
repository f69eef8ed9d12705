#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x2 pk_add(f32x2 a, f32x2 b) { f32x2 r; asm("v_pk_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ f32x2 pk_sub(f32x2 a, f32x2 b) { f32x2 r; asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ f32x2 pk_fma(f32x2 a, f32x2 b, f32x2 c) { f32x2 r; asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
__global__ void k(const float* in, float* out) {
  const int t = threadIdx.x;
  f32x2 a{in[4 * t], in[4 * t + 1]}, b{in[4 * t + 2], in[4 * t + 3]};
  f32x2 s{-1.f, -1.f};
  f32x2 r0 = pk_add(a, b), r1 = pk_sub(a, b), r2 = pk_fma(s, b, a);
  out[6 * t + 0] = r0[0]; out[6 * t + 1] = r0[1]; out[6 * t + 2] = r1[0]; out[6 * t + 3] = r1[1];
  out[6 * t + 4] = r2[0]; out[6 * t + 5] = r2[1];
}
int main() {
  float h[256], o[384], *di, *dout;
  for (int i = 0; i < 256; ++i) h[i] = 1.0f + i * 0.37f;
  hipMalloc(&di, sizeof h); hipMalloc(&dout, sizeof o);
  hipMemcpy(di, h, sizeof h, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, di, dout);
  hipMemcpy(o, dout, sizeof o, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int t = 0; t < 64; ++t) {
    float a0 = h[4 * t], a1 = h[4 * t + 1], b0 = h[4 * t + 2], b1 = h[4 * t + 3];
    float e[6] = {a0 + b0, a1 + b1, a0 - b0, a1 - b1, a0 - b0, a1 - b1};
    for (int j = 0; j < 6; ++j) if (o[6 * t + j] != e[j]) { if (bad < 6) printf("t=%d j=%d got %g want %g\n", t, j, o[6 * t + j], e[j]); ++bad; }
  }
  printf("bad = %d\n", bad);
  return 0;
}
