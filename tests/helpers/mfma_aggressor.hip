// TEST INFRASTRUCTURE (not product code; built by __graft_entry__.build() into tests/helpers/libmfma_aggressor.so).
// Kernels that keep every SIMD of the chip busy with bf16 MFMAs for a few milliseconds: the aggressor of
// tests/test_stereo_depth_gpu.py::test_costvolume_beside_bf16_mfma_kernels_equals_serial_run.
// Background: on MI355X a packed-fp32 VOP3P instruction whose source is broadcast by op_sel drops
// single 16-lane passes while bf16 MFMAs of ANOTHER wave execute (tools/micro/pkfma_corun.hip); the product library
// contains no such instruction and - since the split-operand instances were parked - no bf16 MFMA of its own, so the
// regression test has to bring the aggressor along.  What makes an effective aggressor was measured with the tools
// build's old kernel form (ST_CV_FMA=1, tools/cv_stress.py micro <variant>; profiles/r05_corun_cv_stress.txt):
//   variant 0  v_mfma_f32_16x16x32_bf16 on loop-invariant registers                        234 of 240 volumes wrong
//   variant 1  the same with both operands re-read from LDS every step (as a GEMM does)     236 of 240   <- the test's
//   variant 2  v_mfma_f32_32x32x16_bf16 with operands re-read from LDS every step             0 of 240
//   variant 3  variant 2 + a global load and an LDS write per step                            0 of 240
// (the library's own conv_split_kernel, 32x32x16: 236 of 240; torch.matmul bf16 GEMMs: 0 of 240 - whether a co-running
// kernel triggers the defect depends on more than its MFMA flavour, which is why the fence is on the VICTIM's encoding)
#include <hip/hip_runtime.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int V>
__global__ __launch_bounds__(256) void bf16_mfma_busy_kernel(float* out, const float* src, int iters, float a0) {
  __shared__ __attribute__((aligned(16))) __bf16 frag[2][64 * 8 * 8];   // 2 operands x 8 steps x 64 lanes x 8 bf16
  const int t = threadIdx.x, lane = t & 63;
  for (int i = t; i < 2 * 64 * 8 * 8; i += 256)
    (&frag[0][0])[i] = (__bf16)(a0 * 0.25f + (float)((i * 7) & 31) * 0.0625f);
  __syncthreads();
  f32x4 acc4[4] = {};
  f32x16 acc16[2] = {};
  bf16x8 ab, bb;
  for (int j = 0; j < 8; ++j) {
    ab[j] = (__bf16)(a0 + (float)((t + j) & 15) * 0.125f);
    bb[j] = (__bf16)(0.5f + (float)j * 0.0625f);
  }
  float g = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int st = 0; st < 8; ++st) {
      if (V >= 1) {
        ab = *reinterpret_cast<const bf16x8*>(&frag[0][(((st + it) & 7) * 64 + lane) * 8]);
        bb = *reinterpret_cast<const bf16x8*>(&frag[1][(((st + it) & 7) * 64 + lane) * 8]);
      }
      if (V <= 1) {
#pragma unroll
        for (int i = 0; i < 4; ++i) acc4[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab, bb, acc4[i], 0, 0, 0);
      } else {
        acc16[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, acc16[0], 0, 0, 0);
        acc16[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bb, ab, acc16[1], 0, 0, 0);
      }
    }
    if (V >= 3) {
      g += src[((it * 256 + t) * 4) & 0xFFFF];                       // a global load per step (64 K floats of source)
      frag[it & 1][(((it >> 1) & 7) * 64 + lane) * 8] = (__bf16)(g * 1e-9f + 0.5f);   // and an LDS write
    }
  }
  float s = g;
  for (int i = 0; i < 4; ++i) s += acc4[i][0] + acc4[i][1] + acc4[i][2] + acc4[i][3];
  for (int i = 0; i < 2; ++i)
    for (int j = 0; j < 16; ++j) s += acc16[i][j];
  if (s == 12345.678f) out[t] = s;   // never true: keeps the loop alive
}

// 512 workgroups x 4 waves = 2 waves per SIMD on 256 CUs.  `scratch_dev`: >= 256 KB of device memory (read by variant
// 3, never written in practice).  Returns the hipError_t of the launch.
extern "C" int st_test_bf16_mfma_busy(void* scratch_dev, int iters, int variant, void* stream) {
  float* p = static_cast<float*>(scratch_dev);
  hipStream_t s = static_cast<hipStream_t>(stream);
  switch (variant) {
    case 0: hipLaunchKernelGGL(bf16_mfma_busy_kernel<0>, dim3(512), dim3(256), 0, s, p, p, iters, 1.0f); break;
    case 1: hipLaunchKernelGGL(bf16_mfma_busy_kernel<1>, dim3(512), dim3(256), 0, s, p, p, iters, 1.0f); break;
    case 2: hipLaunchKernelGGL(bf16_mfma_busy_kernel<2>, dim3(512), dim3(256), 0, s, p, p, iters, 1.0f); break;
    default: hipLaunchKernelGGL(bf16_mfma_busy_kernel<3>, dim3(512), dim3(256), 0, s, p, p, iters, 1.0f); break;
  }
  return (int)hipGetLastError();
}
