// Does v_pk_fma_f32 with an op_sel operand broadcast go wrong beside bf16 MFMAs of ANOTHER wave?  (DESIGN.md 5,
// round 4: the cost-volume kernel returned a different volume while conv_split_kernel ran on another stream.)
// A register-level reproducer without the library: VICTIM kernels evaluate the same fmaf chains
//   acc[k] = fma(l, broadcast(r_k), acc[k])                 (l a register pair, r_k one float of a 4-float window)
// in five encodings, AGGRESSOR kernels keep every SIMD busy with one MFMA flavour on a second stream; every victim
// launch is compared bit for bit with the same kernel's result on an idle chip, and the encodings with each other.
//   MODE 0  two v_fma_f32                                            (the library's PK = false form)
//   MODE 1  v_pk_fma_f32 ... op_sel_hi:[1,0,1] / op_sel:[0,1,0]      (what the compiler emits for f32x2{r, r})
//   MODE 2  v_mov_b32 into a (r, r) pair, v_pk_fma_f32 without op_sel
//   MODE 3  MODE 1 with the window re-read from LDS (ds_read_b128) in front of every group, as the kernel does
//   MODE 4  MODE 1 with the OTHER half of each source pair never written (the compiler's v[172:173] case)
//   MODE 5  DIAGNOSTIC form of MODE 3: l = (1, 1), window = (1, 4096, 16, 65536) constant, so an accumulator is a count -
//           a step that read the WRONG half of its source pair shows as +4096 instead of +1 (or the reverse) and the
//           failing encoding (op_sel / op_sel_hi), the failing result half and the substituted value can be read off
//   hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize tools/micro/pkfma_corun.hip -o /tmp/pkfma_corun && /tmp/pkfma_corun
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 2; } } while (0)

constexpr int NA = 12;   // accumulator pairs per lane (the kernel: 2 x 11 + 4 scalars)

template <int MODE>
__global__ __launch_bounds__(128) void victim(float* __restrict__ out, int iters) {
  __shared__ f32x4 win[128];
  const int t = threadIdx.x;
  const int g = blockIdx.x * 128 + t;
  f32x2 l = {1.0f + (float)(g & 1023) * 0.0009765625f, 0.5f + (float)(g & 511) * 0.001953125f};
  f32x4 r = {0.25f + (float)(t & 63) * 0.015625f, -0.75f, 1.125f + (float)(t & 7), 0.0625f};
  f32x2 acc[NA];
#pragma unroll
  for (int k = 0; k < NA; ++k) acc[k] = f32x2{0.f, 0.f};
  if (MODE == 5) { l = f32x2{1.f, 1.f}; r = f32x4{1.f, 4096.f, 16.f, 65536.f}; }
  for (int it = 0; it < iters; ++it) {
    // the window changes every step (same arithmetic in every MODE, ordinary compiler-generated VALU code)
    // (element by element and built with -fno-slp-vectorize: no compiler-made packed instruction in the shared part)
    if (MODE != 5) {
      r[0] = __builtin_fmaf(r[0], 0.99951171875f, 0.001f); r[1] = __builtin_fmaf(r[1], 0.99951171875f, -0.002f);
      r[2] = __builtin_fmaf(r[2], 0.99951171875f, 0.003f); r[3] = __builtin_fmaf(r[3], 0.99951171875f, -0.004f);
      l[0] = l[0] * 1.000244140625f; l[1] = l[1] * 1.000244140625f;
    }
    if (MODE == 3 || MODE == 5) {
      win[t] = r;
      __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): own write visible to own read
      r = win[t];
    }
    f32x2 p01 = {r[0], r[1]}, p23 = {r[2], r[3]};
#pragma unroll
    for (int k = 0; k < NA; ++k) {
      const int e = k & 3;   // window element this accumulator pair multiplies
      if (MODE == 0) {
        acc[k] = f32x2{__builtin_fmaf(l[0], r[e], acc[k][0]), __builtin_fmaf(l[1], r[e], acc[k][1])};
      } else if (MODE == 1 || MODE == 3 || MODE == 5) {
        if (e == 0) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(acc[k]) : "v"(l), "v"(p01));
        if (e == 1) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0]" : "+v"(acc[k]) : "v"(l), "v"(p01));
        if (e == 2) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(acc[k]) : "v"(l), "v"(p23));
        if (e == 3) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0]" : "+v"(acc[k]) : "v"(l), "v"(p23));
      } else if (MODE == 2) {
        f32x2 d;
        asm volatile("v_mov_b32 %0, %2\n\tv_mov_b32 %1, %2" : "=&v"(d[0]), "=&v"(d[1]) : "v"(r[e]));
        asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[k]) : "v"(l), "v"(d));
      } else {   // MODE 4: low half defined, high half of the pair left as whatever the register holds
        f32x2 d;
        asm volatile("v_mov_b32 %0, %1" : "=v"(d[0]) : "v"(r[e]));   // d[1] deliberately undefined
        asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(acc[k]) : "v"(l), "v"(d));
      }
    }
  }
#pragma unroll
  for (int k = 0; k < NA; ++k) {
    out[((size_t)g * NA + k) * 2 + 0] = acc[k][0];
    out[((size_t)g * NA + k) * 2 + 1] = acc[k][1];
  }
}

// AGG 0: v_mfma_f32_32x32x16_bf16, 1: v_mfma_f32_16x16x32_bf16, 2: v_mfma_f32_32x32x16_f16, 3: v_mfma_f32_32x32x2_f32,
//     4: no MFMA (v_fma_f32 loop: the same occupancy, the vector ALU busy instead)
template <int AGG>
__global__ __launch_bounds__(256) void aggressor(float* out, int iters, float a0) {
  f32x16 acc[2];
  for (int i = 0; i < 2; ++i)
    for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
  f32x4 acc4[4] = {};
  bf16x8 ab, bb;
  f16x8 ah, bh;
  for (int j = 0; j < 8; ++j) {
    ab[j] = (__bf16)(a0 + (float)((threadIdx.x + j) & 15) * 0.125f); bb[j] = (__bf16)(0.5f + (float)j * 0.0625f);
    ah[j] = (_Float16)(a0 + (float)((threadIdx.x + j) & 15) * 0.125f); bh[j] = (_Float16)(0.5f + (float)j * 0.0625f);
  }
  float af = a0 + threadIdx.x, bf = a0 * 0.5f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int rep = 0; rep < 8; ++rep) {
      if (AGG == 0) {
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bb, ab, acc[1], 0, 0, 0);
      } else if (AGG == 1) {
#pragma unroll
        for (int i = 0; i < 4; ++i) acc4[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab, bb, acc4[i], 0, 0, 0);
      } else if (AGG == 2) {
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh, ah, acc[1], 0, 0, 0);
      } else if (AGG == 3) {
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(af, bf, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(bf, af, acc[1], 0, 0, 0);
      } else {
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[0][j] = __builtin_fmaf(acc[0][j], 0.999f, af);
      }
    }
  }
  float s = 0.f;
  for (int i = 0; i < 2; ++i)
    for (int j = 0; j < 16; ++j) s += acc[i][j];
  for (int i = 0; i < 4; ++i) s += acc4[i][0] + acc4[i][1] + acc4[i][2] + acc4[i][3];
  if (s == 12345.678f) out[threadIdx.x] = s;
}

typedef void (*vk_t)(float*, int);
typedef void (*ak_t)(float*, int, float);

int main(int argc, char** argv) {
  const int reps = argc > 1 ? atoi(argv[1]) : 12;
  const int VB = 4096, VIT = 600;          // victim: 4096 workgroups x 2 waves, ~0.2 ms alone
  const size_t n = (size_t)VB * 128 * NA * 2;
  float *dout, *dagg;
  CK(hipMalloc(&dout, n * sizeof(float)));
  CK(hipMalloc(&dagg, 4096));
  hipStream_t sv, sa;
  CK(hipStreamCreateWithFlags(&sv, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking));
  vk_t vk[5] = {victim<0>, victim<1>, victim<2>, victim<3>, victim<4>};
  ak_t ak[5] = {aggressor<0>, aggressor<1>, aggressor<2>, aggressor<3>, aggressor<4>};
  const char* an[6] = {"32x32x16_bf16", "16x16x32_bf16", "32x32x16_f16", "32x32x2_f32", "v_fma_f32 loop", "(idle chip)"};
  const char* vn[5] = {"v_fma_f32 x2", "pk_fma op_sel", "pk_fma dup pair", "pk_fma op_sel, LDS-fed", "pk_fma op_sel, undefined other half"};
  std::vector<std::vector<float>> ref(5, std::vector<float>(n));
  std::vector<float> got(n);
  for (int m = 0; m < 5; ++m) {
    CK(hipMemsetAsync(dout, 0xff, n * sizeof(float), sv));
    hipLaunchKernelGGL(vk[m], dim3(VB), dim3(128), 0, sv, dout, VIT);
    CK(hipStreamSynchronize(sv));
    CK(hipMemcpy(ref[m].data(), dout, n * sizeof(float), hipMemcpyDeviceToHost));
  }
  for (int m = 1; m < 5; ++m)
    printf("idle chip: MODE %d (%s) %s MODE 0\n", m, vn[m],
           memcmp(ref[m].data(), ref[0].data(), n * sizeof(float)) ? "DIFFERS from" : "bit-equal to");
  // time the aggressor so that it covers the victim launch: ~3 ms
  printf("%-16s | %-38s | launches differing from the idle-chip result | elements wrong (worst launch)\n", "aggressor",
         "victim encoding");
  for (int a = 0; a < 6; ++a) {
    for (int m = 0; m < 5; ++m) {
      int bad_launch = 0;
      size_t worst = 0;
      for (int rep = 0; rep < reps; ++rep) {
        CK(hipMemsetAsync(dout, 0xff, n * sizeof(float), sv));
        CK(hipStreamSynchronize(sv));
        if (a < 5) hipLaunchKernelGGL(ak[a], dim3(512), dim3(256), 0, sa, dagg, a == 3 ? 3000 : 12000, 1.0f);   // 2 waves per SIMD
        hipLaunchKernelGGL(vk[m], dim3(VB), dim3(128), 0, sv, dout, VIT);
        CK(hipStreamSynchronize(sv));
        CK(hipStreamSynchronize(sa));
        CK(hipMemcpy(got.data(), dout, n * sizeof(float), hipMemcpyDeviceToHost));
        size_t wrong = 0;
        if (memcmp(got.data(), ref[m].data(), n * sizeof(float))) {
          for (size_t i = 0; i < n; ++i) wrong += memcmp(&got[i], &ref[m][i], 4) != 0;
          ++bad_launch;
          if (wrong > worst) worst = wrong;
        }
      }
      printf("%-16s | %-38s | %3d / %-3d | %zu of %zu\n", an[a], vn[m], bad_launch, reps, worst, n);
      fflush(stdout);
    }
  }
  // ---- diagnostic: which encoding, which result half, which value (MODE 5 beside the two bf16 aggressors)
  {
    const int DIT = 200;
    const float win[4] = {1.f, 4096.f, 16.f, 65536.f};
    for (int a = 0; a < 2; ++a) {
      long hist[4][2][3] = {};   // [window element e][result half][0: other half of the pair read, 1: other pair / other, 2: total wrong]
      int shown = 0, bad_launch = 0;
      for (int rep = 0; rep < reps; ++rep) {
        CK(hipMemsetAsync(dout, 0xff, n * sizeof(float), sv));
        CK(hipStreamSynchronize(sv));
        hipLaunchKernelGGL(ak[a], dim3(512), dim3(256), 0, sa, dagg, 12000, 1.0f);
        hipLaunchKernelGGL(victim<5>, dim3(VB), dim3(128), 0, sv, dout, DIT);
        CK(hipStreamSynchronize(sv));
        CK(hipStreamSynchronize(sa));
        CK(hipMemcpy(got.data(), dout, n * sizeof(float), hipMemcpyDeviceToHost));
        bool any = false;
        for (size_t i = 0; i < n; ++i) {
          const int half = (int)(i & 1), k = (int)((i >> 1) % NA), e = k & 3;
          const size_t g = (i >> 1) / NA;
          const float want = (float)DIT * win[e];
          if (got[i] == want) continue;
          any = true;
          // a step that read the other half of its pair contributes win[e ^ 1] instead of win[e]
          const double d = (double)got[i] - (double)want, unit = (double)win[e ^ 1] - (double)win[e];
          const double steps = d / unit;
          const bool other_half = steps > 0 && steps == (double)(long)steps && steps <= DIT;
          hist[e][half][other_half ? 0 : 1]++; hist[e][half][2]++;
          if (shown < 24) {
            printf("  diag %-14s rep %d wg %zu lane %3zu acc %2d elem %d (%s) half %s: got %.1f want %.1f -> %s\n", an[a], rep,
                   g / 128, g % 128, k, e, (e & 1) ? "op_sel:[0,1,0]" : "op_sel_hi:[1,0,1]", half ? "hi" : "lo", got[i], want,
                   other_half ? "read the OTHER half of the pair in" : "other error");
            if (other_half) printf("       %.0f step(s) of %d\n", steps, DIT);
            ++shown;
          }
        }
        bad_launch += any;
      }
      printf("diagnostic beside %s: %d / %d launches wrong\n", an[a], bad_launch, reps);
      for (int e = 0; e < 4; ++e)
        for (int h = 0; h < 2; ++h)
          printf("  window elem %d (%s), result half %s: %ld wrong values, %ld explained by 'the op_sel / op_sel_hi bit was ignored for some steps', %ld other\n",
                 e, (e & 1) ? "op_sel:[0,1,0]   " : "op_sel_hi:[1,0,1]", h ? "hi" : "lo", hist[e][h][2], hist[e][h][0], hist[e][h][1]);
    }
  }
  // how long the aggressors ran next to a victim (so that "co-resident" is a measured statement)
  for (int a = 0; a < 5; ++a) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, sa));
    hipLaunchKernelGGL(ak[a], dim3(512), dim3(256), 0, sa, dagg, a == 3 ? 3000 : 12000, 1.0f);
    CK(hipEventRecord(e1, sa));
    CK(hipStreamSynchronize(sa));
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("aggressor %-16s alone: %.3f ms\n", an[a], ms);
  }
  {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int m = 0; m < 5; ++m) {
      CK(hipEventRecord(e0, sv));
      hipLaunchKernelGGL(vk[m], dim3(VB), dim3(128), 0, sv, dout, VIT);
      CK(hipEventRecord(e1, sv));
      CK(hipStreamSynchronize(sv));
      float ms = 0.f;
      CK(hipEventElapsedTime(&ms, e0, e1));
      printf("victim %-38s alone: %.3f ms\n", vn[m], ms);
    }
  }
  return 0;
}
