// What fp32 MFMA rate does the chip SUSTAIN inside its power cap?  A register-only loop of independent
// v_mfma_f32_32x32x2_f32 (2 waves per SIMD on every CU) runs for several seconds on three operand sets - random N(0,1),
// a constant, all zeros - while `rocm-smi --showclocks --showpower` is sampled from the host.  (tools/micro/mfma_peak.hip
// measures the instruction rate on constant operands for ~0.3 ms: 155 TFLOP/s; the nominal peak is 157.3 at 2.4 GHz.)
//   hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_power.hip -o /tmp/mfma_power && /tmp/mfma_power
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <cmath>
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(256) void k32(const float* __restrict__ src, float* out, int iters) {
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
  float a[16], b[16];
  const int t = blockIdx.x * 256 + threadIdx.x;
  for (int i = 0; i < 16; ++i) {
    a[i] = src[(t * 32 + i) & 0xFFFFF];
    b[i] = src[(t * 32 + 16 + i) & 0xFFFFF];
  }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[i], acc[i & 3], 0, 0, 0);
    if ((it & 255) == 255) {   // keep the accumulators finite: rescale now and then (16 VALU per 4096 MFMAs)
#pragma unroll
      for (int q = 0; q < 4; ++q)
        for (int j = 0; j < 16; ++j) acc[q][j] *= 1e-3f;
    }
  }
  float s = 0.f;
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 16; ++j) s += acc[i][j];
  if (s == 12345.678f) out[threadIdx.x] = s;
}

static void smi(int* sclk, double* watts) {
  *sclk = 0; *watts = 0;
  FILE* f = popen("rocm-smi --showclocks --showpower 2>/dev/null", "r");
  if (!f) return;
  char line[512];
  while (fgets(line, sizeof line, f)) {
    const char* p;
    if ((p = strstr(line, "sclk clock level")) && (p = strchr(p, '('))) *sclk = atoi(p + 1);
    if ((p = strstr(line, "Power (W):"))) *watts = atof(p + 10);
  }
  pclose(f);
}

int main() {
  const int N = 1 << 20;
  std::vector<float> h(N);
  float *src, *out;
  hipMalloc(&src, N * 4); hipMalloc(&out, 4096);
  const char* names[3] = {"random N(0,1) operands", "constant operands (1.5, 0.75)", "all-zero operands"};
  for (int set = 0; set < 3; ++set) {
    srand(1);
    for (int i = 0; i < N; ++i) {
      if (set == 0) {   // Box-Muller
        const double u = (rand() + 1.0) / (RAND_MAX + 2.0), v = (rand() + 1.0) / (RAND_MAX + 2.0);
        h[i] = (float)(sqrt(-2.0 * log(u)) * cos(6.283185307179586 * v));
      } else {
        h[i] = set == 1 ? ((i & 16) ? 0.75f : 1.5f) : 0.0f;
      }
    }
    hipMemcpy(src, h.data(), N * 4, hipMemcpyHostToDevice);
    const int iters = 20000;                       // 16 MFMAs x 64 cycles x 20000 = 20.5 M cycles x 2 waves per SIMD: ~17 ms per launch
    const double flop = 2.0 * 32 * 32 * 2 * 16.0 * iters * 4 /*waves*/ * 512 /*workgroups*/;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k32, dim3(512), dim3(256), 0, 0, src, out, iters);
    hipDeviceSynchronize();
    int smin = 1 << 30, smax = 0, n = 0; double wmax = 0, wsum = 0, tf_sum = 0;
    for (int rep = 0; rep < 12; ++rep) {           // ~12 x (20 launches = 0.35 s) = 4 s
      hipEventRecord(e0);
      for (int l = 0; l < 20; ++l) hipLaunchKernelGGL(k32, dim3(512), dim3(256), 0, 0, src, out, iters);
      hipEventRecord(e1);
      int sc; double w;
      smi(&sc, &w);                                // sampled while the launches run
      hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (rep >= 2) {                              // the first samples ramp up
        tf_sum += 20 * flop / (ms * 1e-3) / 1e12; ++n;
        if (sc) { smin = sc < smin ? sc : smin; smax = sc > smax ? sc : smax; }
        wsum += w; wmax = w > wmax ? w : wmax;
      }
    }
    printf("%-32s: %6.1f TFLOP/s sustained (nominal 157.3), sclk %d-%d MHz, socket power mean %.0f W, max %.0f W\n", names[set],
           tf_sum / n, smin, smax, wsum / n, wmax);
    fflush(stdout);
  }
  system("rocm-smi --showmaxpower 2>/dev/null | grep -i 'Max Graphics'");
  return 0;
}
