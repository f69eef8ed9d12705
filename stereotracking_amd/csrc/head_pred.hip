// The six prediction convolutions of the decoupled head (per level: conv_cls 1x1 feat -> num_classes on the cls tower,
// conv_reg 1x1 feat -> 4 and conv_obj 1x1 feat -> 1 on the reg tower; mmyolo YOLOXHeadModule.forward_single, configured at
// /root/reference/configs/_base_/yolox_s_8x8_mmyolo.py:40-51) as ONE launch for all three levels.
//
// They are 0.2 GFLOP reading 155 MB: six launches on 32-wide MFMA tiles spent 95 us at 0.1-7 TFLOP/s.  This is a
// reduction, not a GEMM: 8 lanes share a pixel, each holds feat / 8 channels of both towers (16-byte quads, interleaved so
// that one load instruction covers 128 contiguous bytes of the pixel),
// multiplies them with the 1 + 5 weight rows kept in LDS, and three xor-shuffles sum the partials; lane 0 of the
// group stores the pixel's [cls | reg(4) | obj] row of the head buffer (32-byte rows).  HBM-bound by construction.
// Summation order: per lane ascending channel fmaf chain, then the shuffle tree - fp32, within 1e-6 of the MFMA
// convolution it replaces (the decode kernel consumes whatever head it is given, bit-exactly).
#include <algorithm>
#include <cstdint>

#include "st_common.h"

namespace st {

struct HeadPredLevel {
  const float* cls;    // cls tower features, NHWC slice
  const float* reg;    // reg tower features
  const float* wc;     // packed conv_cls weights [.][Kpad], rows 0 .. nc-1
  const float* wr;     // packed conv_reg | conv_obj weights, rows 0 .. 4
  const float* bc;
  const float* br;
  float* out;          // head rows of this level: float[M][8]
  int cls_ld, cls_off, reg_ld, reg_off;
  int M;               // pixels of this level (N * h * w)
  int blk0;            // first workgroup of this level
};
struct HeadPredArgs {
  HeadPredLevel lv[3];
  int Kpad, nc, nblocks;
};

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// CPL = 16-byte channel quads per lane and tower (feat = 32 * CPL): lane `part` of a pixel's 8 lanes holds quads
// part, part + 8, ... (one load instruction = 128 contiguous bytes per pixel).  NC = num_classes (rows of conv_cls).
template <int CPL, int NC>
__global__ __launch_bounds__(256) void head_pred_kernel(const HeadPredArgs p) {
  __shared__ float4 wl4[8 * 8 * CPL];   // [row 0..7][feat] (rows >= nc + 5 unused)
  float* wl = reinterpret_cast<float*>(wl4);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int part = lane & 7, pxl = lane >> 3;
  const int li = blockIdx.x >= p.lv[2].blk0 ? 2 : (blockIdx.x >= p.lv[1].blk0 ? 1 : 0);
  const HeadPredLevel& L = p.lv[li];
  const int nb = (li == 2 ? p.nblocks : p.lv[li + 1].blk0) - L.blk0;   // workgroups of this level
  constexpr int no = NC + 5, feat = 32 * CPL;
  for (int e = tid; e < no * feat; e += 256) {
    const int r = e / feat, c = e - r * feat;
    wl[r * feat + c] = r < NC ? L.wc[r * p.Kpad + c] : L.wr[(r - NC) * p.Kpad + c];
  }
  __syncthreads();
  float bias[no];
#pragma unroll
  for (int r = 0; r < no; ++r) bias[r] = r < NC ? L.bc[r] : L.br[r - NC];

  for (int g = (blockIdx.x - L.blk0) * 4 + wave; g * 8 < L.M; g += nb * 4) {
    const int m = g * 8 + pxl;
    const bool ok = m < L.M;
    f32x4 xc[CPL], xr[CPL];
#pragma unroll
    for (int q = 0; q < CPL; ++q) {
      const int c = 4 * (8 * q + part);
      xc[q] = ok ? *reinterpret_cast<const f32x4*>(L.cls + (size_t)m * L.cls_ld + L.cls_off + c) : f32x4{0.f, 0.f, 0.f, 0.f};
      xr[q] = ok ? *reinterpret_cast<const f32x4*>(L.reg + (size_t)m * L.reg_ld + L.reg_off + c) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    float s[no];
#pragma unroll
    for (int r = 0; r < no; ++r) {
      s[r] = 0.f;
#pragma unroll
      for (int q = 0; q < CPL; ++q) {
        const f32x4 w = *reinterpret_cast<const f32x4*>(wl + r * feat + 4 * (8 * q + part));
        const f32x4 x = r < NC ? xc[q] : xr[q];
#pragma unroll
        for (int e = 0; e < 4; ++e) s[r] = fmaf(x[e], w[e], s[r]);
      }
      s[r] += __shfl_xor(s[r], 1);
      s[r] += __shfl_xor(s[r], 2);
      s[r] += __shfl_xor(s[r], 4);
      s[r] += bias[r];
    }
    if (ok && part == 0) {
      float* o = L.out + (size_t)m * 8;
      *reinterpret_cast<f32x4*>(o) = f32x4{s[0], s[1], s[2], s[3]};
#pragma unroll
      for (int r = 4; r < no; ++r) o[r] = s[r];
    }
  }
}

}  // namespace

// the shipped head: one class (configs/_base_/yolox_s_8x8_mmyolo.py:46), feat = 256 x widen_factor in {96, 128, 256}
bool head_pred_applicable(int feat, int nc) { return nc == 1 && (feat == 96 || feat == 128 || feat == 256); }

int head_pred_launch(HeadPredArgs a, int feat, hipStream_t stream) {
  ST_REQUIRE(head_pred_applicable(feat, a.nc), "head_pred: feat %d / num_classes %d not supported", feat, a.nc);
  // workgroups per level in proportion to its pixels (8 pixels per wave and iteration, 4 waves), at least one each
  long long total = 0;
  for (int l = 0; l < 3; ++l) {
    const HeadPredLevel& L = a.lv[l];
    ST_REQUIRE(L.cls && L.reg && L.wc && L.wr && L.bc && L.br && L.out && L.M > 0, "head_pred: null pointer / empty level");
    ST_REQUIRE(((L.cls_ld | L.cls_off | L.reg_ld | L.reg_off) & 3) == 0 &&
                   ((reinterpret_cast<uintptr_t>(L.cls) | reinterpret_cast<uintptr_t>(L.reg) |
                     reinterpret_cast<uintptr_t>(L.out)) & 15) == 0,
               "head_pred: tensors must be 16-byte aligned with channel strides multiples of 4");
    total += L.M;
  }
  const int budget = 2048;   // ~8 workgroups per CU: enough loads in flight for an HBM-bound sweep
  int blk = 0;
  for (int l = 0; l < 3; ++l) {
    a.lv[l].blk0 = blk;
    const long long want = (a.lv[l].M * (long long)budget + total - 1) / total;
    const long long cap = (a.lv[l].M + 31) / 32;   // one iteration per wave at least
    blk += (int)std::max<long long>(1, std::min(want, cap));
  }
  a.nblocks = blk;
  switch (feat / 32) {
    case 3: hipLaunchKernelGGL((head_pred_kernel<3, 1>), dim3(blk), dim3(256), 0, stream, a); break;
    case 4: hipLaunchKernelGGL((head_pred_kernel<4, 1>), dim3(blk), dim3(256), 0, stream, a); break;
    default: hipLaunchKernelGGL((head_pred_kernel<8, 1>), dim3(blk), dim3(256), 0, stream, a); break;
  }
  ST_CHECK_HIP(hipGetLastError());
  return ST_OK;
}

}  // namespace st
