// Packed-fp32 helpers shared by the Winograd kernels (wino_conv.hip, wino_csp_tail.hip), gfx950 only.
#pragma once
#include <hip/hip_runtime.h>

namespace st {
namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float wn_silu(float v) { return v * __builtin_amdgcn_rcpf(1.0f + __expf(-v)); }

// f32x4 arithmetic as two packed-f32 instructions (v_pk_add_f32 / v_pk_fma_f32: two IEEE fp32 results per lane and
// instruction).  fp32 MFMA runs on the same FMA lanes as the vector ALU (equal peaks; measured: VALU time adds to MFMA
// time, tools/micro/mfma_peak.hip), so every VALU instruction of the transforms is taken from the matrix rate.
typedef float f32x2 __attribute__((ext_vector_type(2)));
// (written as instructions: from <2 x float> IR the compiler scalarises most of them again)
__device__ __forceinline__ f32x2 wn_pk_add(f32x2 a, f32x2 b) {
  f32x2 r;
  asm("v_pk_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ f32x2 wn_pk_sub(f32x2 a, f32x2 b) {
  f32x2 r;
  asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ f32x2 wn_pk_fma(f32x2 a, f32x2 b, f32x2 c) {
  f32x2 r;
  asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
__device__ __forceinline__ f32x2 wn_lo(f32x4 a) { return __builtin_shufflevector(a, a, 0, 1); }
__device__ __forceinline__ f32x2 wn_hi(f32x4 a) { return __builtin_shufflevector(a, a, 2, 3); }
__device__ __forceinline__ f32x4 wn_join(f32x2 lo, f32x2 hi) { return __builtin_shufflevector(lo, hi, 0, 1, 2, 3); }
__device__ __forceinline__ f32x4 wn_add(f32x4 a, f32x4 b) {
  return wn_join(wn_pk_add(wn_lo(a), wn_lo(b)), wn_pk_add(wn_hi(a), wn_hi(b)));
}
__device__ __forceinline__ f32x4 wn_sub(f32x4 a, f32x4 b) {
  return wn_join(wn_pk_sub(wn_lo(a), wn_lo(b)), wn_pk_sub(wn_hi(a), wn_hi(b)));
}
// a + s * b with s = +-1: s * b is exact, so the single rounding of the fma is the rounding of a +- b
__device__ __forceinline__ f32x4 wn_addsgn(f32x4 a, f32x2 s, f32x4 b) {
  return wn_join(wn_pk_fma(s, wn_lo(b), wn_lo(a)), wn_pk_fma(s, wn_hi(b), wn_hi(a)));
}
// The same, for values an MFMA consumes next: a VALU write needs 2 wait states before an MFMA reads the register.
// The compiler inserts them after its own VALU instructions, but it does not look inside an asm statement.
__device__ __forceinline__ f32x4 wn_add_mfma(f32x4 a, f32x4 b) {
  f32x2 lo, hi;
  asm("v_pk_add_f32 %0, %2, %4\n\tv_pk_add_f32 %1, %3, %5\n\ts_nop 1"
      : "=&v"(lo), "=&v"(hi) : "v"(wn_lo(a)), "v"(wn_hi(a)), "v"(wn_lo(b)), "v"(wn_hi(b)));
  return wn_join(lo, hi);
}
__device__ __forceinline__ f32x4 wn_sub_mfma(f32x4 a, f32x4 b) {
  f32x2 lo, hi;
  asm("v_pk_add_f32 %0, %2, %4 neg_lo:[0,1] neg_hi:[0,1]\n\tv_pk_add_f32 %1, %3, %5 neg_lo:[0,1] neg_hi:[0,1]\n\ts_nop 1"
      : "=&v"(lo), "=&v"(hi) : "v"(wn_lo(a)), "v"(wn_hi(a)), "v"(wn_lo(b)), "v"(wn_hi(b)));
  return wn_join(lo, hi);
}

}  // namespace
}  // namespace st
