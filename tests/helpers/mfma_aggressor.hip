// TEST INFRASTRUCTURE (not product code; built by __graft_entry__.build() into tests/helpers/libmfma_aggressor.so).
// A kernel that keeps every SIMD of the chip busy with bf16 MFMAs (v_mfma_f32_16x16x32_bf16, registers only) for a few
// milliseconds: the aggressor of tests/test_stereo_depth_gpu.py::test_costvolume_beside_bf16_mfma_kernels_equals_serial_run.
// Background: on MI355X a packed-fp32 VOP3P instruction whose source is broadcast by op_sel drops
// single 16-lane passes while bf16 MFMAs of ANOTHER wave execute (tools/micro/pkfma_corun.hip); the product library
// contains no such instruction and - since the split-operand instances were parked - no bf16 MFMA of its own, so the
// regression test has to bring the aggressor along.  torch.matmul on bf16 tensors does NOT serve: measured, it never
// triggered the defect on the old kernel form (profiles/r05_corun_cv_stress.txt), this loop did in 11 of 12 launches.
#include <hip/hip_runtime.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__global__ __launch_bounds__(256) void bf16_mfma_busy_kernel(float* out, int iters, float a0) {
  f32x4 acc[4] = {};
  bf16x8 ab, bb;
  for (int j = 0; j < 8; ++j) {
    ab[j] = (__bf16)(a0 + (float)((threadIdx.x + j) & 15) * 0.125f);
    bb[j] = (__bf16)(0.5f + (float)j * 0.0625f);
  }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int rep = 0; rep < 8; ++rep)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab, bb, acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  if (s == 12345.678f) out[threadIdx.x] = s;   // never true: keeps the loop alive
}

// 512 workgroups x 4 waves = 2 waves per SIMD on 256 CUs; iters = 12000 runs ~6 ms alone.  `scratch_dev`: >= 1 KB of
// device memory (never written in practice).  Returns the hipError_t of the launch.
extern "C" int st_test_bf16_mfma_busy(void* scratch_dev, int iters, void* stream) {
  hipLaunchKernelGGL(bf16_mfma_busy_kernel, dim3(512), dim3(256), 0, static_cast<hipStream_t>(stream),
                     static_cast<float*>(scratch_dev), iters, 1.0f);
  return (int)hipGetLastError();
}
