// Tail of a stage-1 CSP branch in ONE persistent launch (tile variant 56): DarknetBottleneck conv2 (3x3, 32 -> 32,
// Winograd F(2x2,3x3), + identity) and the CSPLayer's final_conv (1x1 on cat[blocks | short], 64 -> 64), optionally with
// the two-branch average (rgb + disparity) / 2 of the fused backbone in its epilogue.  Reference modules: mmdet CSPLayer /
// DarknetBottleneck as built at mmtrack/models/backbones/csp_darknet_disparity_v1.py:145-153 and run at :176-184
// (stage1 / disp_stage1, branch fusion :104-153, 155-184).
//
// Why one kernel.  Round 5 measured the 32 -> 32 Winograd layer as the least efficient matrix kernel of the path (MFMA
// pipe busy 0.31, 17.7 mJ per executed GFLOP): a workgroup ran ONE K-chunk - 64 MFMAs per wave - between a window
// fetch, a stream of transformed weights from L2 and a residual fetch, and died.  The 64 -> 64 final conv behind it is
// HBM-bound (5 TB/s, 1386 W): it re-reads the 32 channels conv2 has just written.  Here
//   * the workgroup is PERSISTENT over tile blocks (16 x 8 output pixels each) and keeps its wave's transformed
//     weights - U[a][b] for the wave's transform row a: 4 x 32 x 32 floats - in 64 VGPRs for its whole life: no weight
//     stream at all;
//   * conv2's output never reaches memory: after the output transform (+ bias, SiLU, identity) a wave's 32 pixels x 32
//     channels go through a 4 KB wave-private LDS tile straight into the B operands of the 1x1 GEMM, the `short` half of
//     the concat is loaded from HBM directly in operand layout, the 64 x 64 weights sit in LDS in fragment order;
//   * the 1x1 GEMM runs with SWAPPED operands (A = weights, B = pixels), so a lane ends with 4 consecutive couts of one
//     pixel per accumulator quad: 16-byte NHWC stores, 16-byte loads of the other branch for the average.
// Per tile block and wave: 64 MFMAs 32x32x2 + 128 MFMAs 16x16x4 (the same matrix cycles), two workgroup barriers; the next block's input window travels
// by LDS-DMA while the current block's epilogue and 1x1 GEMM run.  LDS: window 23 KB + transform exchange 32 KB (the 1x1
// operand tiles alias the part of it only their own wave reads) + final weights 16 KB = 71 KB -> two workgroups per CU.
// Both halves repeat the arithmetic of the launches they replace instruction for instruction - conv2 that of
// wino_conv3x3_kernel<1, 1, true>, the 1x1 GEMM that of pw_resident_kernel<4, 4, 0, RES> (16x16x4 MFMA, same K walk, bias
// in the accumulator) - so the fused launch returns the unfused pair's values BIT FOR BIT (tests/test_conv_gpu.py).
#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <cstring>

#include "st_common.h"
#include "wino_pk.h"

namespace st {
namespace {

constexpr int CT_TX = 8, CT_TY = 4;                          // tiles per block (x, y): 16 x 8 output pixels
constexpr int CT_WW = 2 * CT_TX + 2, CT_WH = 2 * CT_TY + 2;  // input window 18 x 10
constexpr int CT_PIX = CT_WW * CT_WH;                        // 180
constexpr int CT_PIECES = (CT_PIX + 7) / 8;                  // 23 LDS-DMA pieces of 8 pixels x 128 B
constexpr int CT_WIN_FLOATS = CT_PIECES * 256;
constexpr int CT_RB_FLOATS = 4 * 2 * 32 * 32;                // [a][j][tile][co]
constexpr int CT_WF_FLOATS = 64 * 64;
constexpr int CT_LDS_FLOATS = CT_WIN_FLOATS + CT_RB_FLOATS + CT_WF_FLOATS + 64;
constexpr int CT_NPW = (CT_PIECES + 3) / 4;                  // DMA pieces per wave

struct TailArgs {
  const float* in;      // conv2 input (bottleneck conv1 output), 32 channels
  const float* wino;    // conv2 weights in Winograd fragment order (wino_pack_weights, 32 x 32: 64 KB)
  const float* bias2;
  const float* res;     // conv2 identity (32 channels)
  const float* sh;      // `short` half of the concat: final conv input channels [32, 64)
  const float* wf;      // final conv weights in fragment order (csp_tail_pack_frags)
  const float* biasf;
  const float* res2;    // AVG: the other branch's stage output (64 channels)
  float* out;
  int N, H, W;
  int in_ld, in_off, res_ld, res_off, sh_ld, sh_off, res2_ld, res2_off, out_ld, out_off;
  float post2, postf;
  int tbx, tby;
  unsigned nblocks;
  unsigned in_bytes, res_bytes, sh_bytes, res2_bytes, out_bytes;
};

template <bool AVG>
__global__ __launch_bounds__(256, 2) void wino_csp_tail_kernel(const TailArgs p) {
#if defined(__HIP_DEVICE_COMPILE__)
  extern __shared__ float4 ct_smem4[];
  float* smem = reinterpret_cast<float*>(ct_smem4);
  float* win = smem;
  float* Rb = smem + CT_WIN_FLOATS;
  float* Wf = Rb + CT_RB_FLOATS;
  float* Bf = Wf + CT_WF_FLOATS;
  const int tid = threadIdx.x, lane = tid & 63, a = tid >> 6;   // wave = transform row a = tile row of the epilogue
  const int i = lane & 31, h = lane >> 5;
  const int tyi = i >> 3, txi = i & 7;
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

  const __amdgpu_buffer_rsrc_t irsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.in), 0, (int)p.in_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t wrsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.wino), 0, 4 * 16 * 256 * 4, 0x00020000);
  const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, (int)p.out_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rrsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.res), 0, (int)p.res_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t srsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.sh), 0, (int)p.sh_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t arsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(AVG ? p.res2 : p.in), 0, (int)(AVG ? p.res2_bytes : 0u), 0x00020000);

  // ---- once per workgroup: final weights + bias into LDS, this wave's transformed conv2 weights into registers
#pragma unroll
  for (int k = 0; k < CT_WF_FLOATS / 4 / 256; ++k)
    reinterpret_cast<f32x4*>(Wf)[tid + 256 * k] = reinterpret_cast<const f32x4*>(p.wf)[tid + 256 * k];
  if (tid < 16) reinterpret_cast<f32x4*>(Bf)[tid] = reinterpret_cast<const f32x4*>(p.biasf)[tid];
  f32x4 Uw[16];   // step = g * 4 + b: U_{a,b}[co = lane & 31][ci = 8g + 4h + e]
  {
    const unsigned wbase = (unsigned)((a * 16 * 256 + lane * 4) * 4);
#pragma unroll
    for (int s = 0; s < 16; ++s)
      Uw[s] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, wbase, s * 1024, 0));
  }

  // ---- window DMA: piece j = a + 4k holds window pixels 8j .. 8j + 7; lane (pixel slot q, 16-byte slot sl) fetches channel
  // quad sl ^ ((q >> 1) & 7) of window pixel q.  The geometry is recomputed per block from an OPAQUE copy of the lane id:
  // hoisted out of the block loop it would hold 12 registers through the K loop (the kernel sits at the 256-register budget
  // of two waves per SIMD).
  auto dma_window = [&](unsigned b) {
    int ln = lane;
    asm volatile("" : "+v"(ln));
    const int bx = (int)(b % (unsigned)p.tbx);
    const unsigned r = b / (unsigned)p.tbx;
    const int by = (int)(r % (unsigned)p.tby), n = (int)(r / (unsigned)p.tby);
    const int wy0 = by * (2 * CT_TY) - 1, wx0 = bx * (2 * CT_TX) - 1;
    const int base = (((n * p.H + wy0) * p.W + wx0) * p.in_ld + p.in_off) * 4;
#pragma unroll
    for (int k = 0; k < CT_NPW; ++k) {
      const int j = a + 4 * k;
      const int q = 8 * j + (ln >> 3), sl = ln & 7;
      const int wr = q / CT_WW, wc = q - wr * CT_WW;
      const int quad = sl ^ ((q >> 1) & 7);
      const int y = wy0 + wr, x = wx0 + wc;
      const bool ok = q < CT_PIX && y >= 0 && y < p.H && x >= 0 && x < p.W;
      // zero fill = conv padding / ragged edge
      const unsigned off = ok ? (unsigned)(base + ((wr * p.W + wc) * p.in_ld + 4 * quad) * 4) : 0x80000000u;
      if (j < CT_PIECES)   // wave-uniform
        __builtin_amdgcn_raw_ptr_buffer_load_lds(irsrc, (__attribute__((address_space(3))) void*)(win + j * 256), 16, off,
                                                 0, 0, 0);
    }
  };

  // ---- this lane's 2 x 4 patch pixels of the input transform (as wino_conv3x3_body)
  //   a = 0: d0 - d2    a = 1: d1 + d2    a = 2: d2 - d1    a = 3: d1 - d3
  const int r0 = a == 0 ? 0 : (a == 2 ? 2 : 1);
  const int r1 = a == 0 ? 2 : (a == 1 ? 2 : (a == 2 ? 1 : 3));
  const f32x2 sgn = a == 1 ? f32x2{1.0f, 1.0f} : f32x2{-1.0f, -1.0f};
  // patch pixel (row r, column c) of tile (tyi, txi): window pixel q = (2 tyi + r) * 18 + 2 txi + c at float offset 32 q,
  // channel slot (2g + h) ^ ((q >> 1) & 7).  q(c = 0) is even, so columns (0, 1) and (2, 3) share a swizzle term and
  // (2g + h) ^ sw = 2g ^ (h ^ sw): two row bases + four (h ^ sw) << 2 terms describe all eight reads.
  const int qb0 = ((2 * tyi + r0) * CT_WW + 2 * txi), qb1 = ((2 * tyi + r1) * CT_WW + 2 * txi);
  const int hs00 = (h ^ ((qb0 >> 1) & 7)) << 2, hs01 = (h ^ (((qb0 >> 1) + 1) & 7)) << 2;
  const int hs10 = (h ^ ((qb1 >> 1) & 7)) << 2, hs11 = (h ^ (((qb1 >> 1) + 1) & 7)) << 2;
  const float* wrow0 = win + qb0 * 32;
  const float* wrow1 = win + qb1 * 32;
  // epilogue lane roles.  Column reduce: tile txo of tile row a, couts c4 .. c4 + 3.  1x1 GEMM (16x16x4 MFMA, two pixel
  // tiles of 16): pixels n = 16 pt + i16 of the wave's 32 (tile n >> 2, row (n >> 1) & 1, column n & 1), K quarter kq.
  const int txo = lane >> 3, c4 = (lane & 7) * 4;
  const f32x4 bias4 = *reinterpret_cast<const f32x4*>(p.bias2 + c4);
  const int i16 = lane & 15, kq = lane >> 4;
  // operand tile of wave a inside Rb: pixel n, 16-byte slot s -> float offset; rows = the planes k = n >> 3 (k < 4) of
  // tiles 8a .. 8a + 7, which only wave a reads in the column reduce
  auto yoff = [&](int n, int s) { return (((n >> 3) * 32 + 8 * a) * 32) + (n & 7) * 32 + ((s ^ (n & 7)) << 2); };

  unsigned blk = blockIdx.x;
  if (blk < p.nblocks) dma_window(blk);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // first window, weights: the loop's own wait below leaves 8 operations in flight

  for (; blk < p.nblocks; blk += gridDim.x) {
    // window of `blk` landed (this wave's pieces), previous block's stores retired; every wave is past its last LDS
    // read of the previous block (operand tiles alias Rb) and, the first time, Wf / Bf are written
    // (the previous block's 8 output stores are the youngest memory operations of the wave: vmcnt(8) leaves them in flight)
    asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
    __syncthreads();
    const int bx = (int)(blk % (unsigned)p.tbx);
    const unsigned rr = blk / (unsigned)p.tbx;
    const int by = (int)(rr % (unsigned)p.tby), n = (int)(rr / (unsigned)p.tby);

    // ---- early load: the identity of conv2's epilogue (needed right behind the barrier below)
    bool okp[2];
    int mp[2];
#pragma unroll
    for (int pt = 0; pt < 2; ++pt) {
      const int nn = 16 * pt + i16;
      const int oyp = by * (2 * CT_TY) + 2 * a + ((nn >> 1) & 1), oxp = bx * (2 * CT_TX) + 2 * (nn >> 2) + (nn & 1);
      okp[pt] = oyp < p.H && oxp < p.W;
      mp[pt] = (n * p.H + oyp) * p.W + oxp;
    }
    f32x4 rv[2][2];   // [j][ii]
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int ii = 0; ii < 2; ++ii) {
        const int oy = by * (2 * CT_TY) + 2 * a + ii, ox = bx * (2 * CT_TX) + 2 * txo + j;
        const bool ok = oy < p.H && ox < p.W;
        const unsigned off = ok ? (unsigned)((((n * p.H + oy) * p.W + ox) * p.res_ld + p.res_off + c4) * 4) : 0x80000000u;
        rv[j][ii] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rrsrc, off, 0, 0));
      }

    __builtin_amdgcn_s_setprio(0);
    // ---- conv2: one K-chunk of 32 channels, weights from registers
    f32x16 acc[4];
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[b][r] = 0.f;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      f32x4 d[8];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        d[c] = *reinterpret_cast<const f32x4*>(wrow0 + 32 * c + ((8 * g) ^ (c < 2 ? hs00 : hs01)));
        d[4 + c] = *reinterpret_cast<const f32x4*>(wrow1 + 32 * c + ((8 * g) ^ (c < 2 ? hs10 : hs11)));
      }
      f32x4 P[4], V[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) P[c] = wn_addsgn(d[c], sgn, d[4 + c]);
      V[0] = wn_sub_mfma(P[0], P[2]);
      V[1] = wn_add_mfma(P[1], P[2]);
      V[2] = wn_sub_mfma(P[2], P[1]);
      V[3] = wn_sub_mfma(P[1], P[3]);
#pragma unroll
      for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int s = 0; s < 4; ++s)
          acc[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(V[b][s], Uw[g * 4 + b][s], acc[b], 0, 0, 0);
    }

    // `short` in operand layout for the 1x1 GEMM: requested here (the patch registers are free again), used last
    f32x4 shv[2][2];   // [pixel tile][channel group of 16]: channels 32 + 16 gg + 4 kq .. + 3 of the concat
#pragma unroll
    for (int pt = 0; pt < 2; ++pt) {
      const unsigned off = okp[pt] ? (unsigned)((mp[pt] * p.sh_ld + p.sh_off + 4 * kq) * 4) : 0x80000000u;
#pragma unroll
      for (int gg = 0; gg < 2; ++gg)
        shv[pt][gg] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(srsrc, off, gg * 64, 0));
    }
    __builtin_amdgcn_s_setprio(2);
    // ---- output transform, rows: R[0] = M0 + M1 + M2, R[1] = M1 - M2 - M3 (registers -> Rb)
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_nop 15\n\ts_nop 3" ::: "memory");   // 32x32 MFMA write -> VALU read inside the asm adds below
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = (r & 3) + 8 * (r >> 2) + 4 * h;
      if (r & 1) continue;
      const f32x2 a0{acc[0][r], acc[0][r + 1]}, a1{acc[1][r], acc[1][r + 1]};
      const f32x2 a2{acc[2][r], acc[2][r + 1]}, a3{acc[3][r], acc[3][r + 1]};
      const f32x2 R0 = wn_pk_add(wn_pk_add(a0, a1), a2);
      const f32x2 R1 = wn_pk_sub(wn_pk_sub(a1, a2), a3);
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        Rb[((a * 2 + 0) * 32 + m + e) * 32 + i] = R0[e];
        Rb[((a * 2 + 1) * 32 + m + e) * 32 + i] = R1[e];
      }
    }
    __syncthreads();
    // the window is free: the next block's input travels while this block's epilogue and 1x1 GEMM run.  (Measured and
    // dropped: TWO window buffers - the next window requested a whole block ahead - with the 1x1 weights streamed from L2
    // instead of LDS to stay at two workgroups per CU: 254 against 234 us at N = 16, tools/tail_bench.py.)
    if (blk + gridDim.x < p.nblocks) dma_window(blk + gridDim.x);
    f32x4 r2v[2][4];   // [pixel tile][cout block of 16]
    if (AVG) {   // the other branch's output for this lane's pixels, in accumulator layout
#pragma unroll
      for (int pt = 0; pt < 2; ++pt)
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) {
          const unsigned off = okp[pt] ? (unsigned)((mp[pt] * p.res2_ld + p.res2_off + 16 * cb + 4 * kq) * 4) : 0x80000000u;
          r2v[pt][cb] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(arsrc, off, 0, 0));
        }
    }
    // ---- output transform, columns + conv2 epilogue.  ALL reads of Rb first: the operand tile written below aliases them
    const int t = a * 8 + txo;
    f32x4 q4[2][4];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int aa = 0; aa < 4; ++aa) q4[j][aa] = *reinterpret_cast<const f32x4*>(Rb + ((aa * 2 + j) * 32 + t) * 32 + c4);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const f32x4 y[2] = {wn_add(wn_add(q4[j][0], q4[j][1]), q4[j][2]), wn_sub(wn_sub(q4[j][1], q4[j][2]), q4[j][3])};
#pragma unroll
      for (int ii = 0; ii < 2; ++ii) {
        f32x4 v = wn_add(y[ii], bias4);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = wn_silu(v[e]);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = v[e] + rv[j][ii][e];
        if (p.post2 != 1.0f) {   // uniform; a bottleneck identity has scale 1 (x * 1.0f == x: the multiply is skipped, not changed)
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] *= p.post2;
        }
        *reinterpret_cast<f32x4*>(Rb + yoff(txo * 4 + ii * 2 + j, lane & 7)) = v;
      }
    }
    // ---- 1x1 GEMM, swapped operands: D[cout][pixel] = sum_ci Wf[cout][ci] * act[ci][pixel].  Instruction for instruction
    // the arithmetic of pw_resident_kernel<4, 4, 0, RES> (the launch this replaces): v_mfma_f32_16x16x4_f32, accumulators
    // started from the bias, K walked as g = 0..3 (16 channels), s = 0..3, one MFMA = channels 16 g + s + {0, 4, 8, 12} -
    // so the fused launch returns the unfused pair's values BIT FOR BIT.
    f32x4 yv[2][2];
#pragma unroll
    for (int pt = 0; pt < 2; ++pt)
#pragma unroll
      for (int gg = 0; gg < 2; ++gg) yv[pt][gg] = *reinterpret_cast<const f32x4*>(Rb + yoff(16 * pt + i16, 4 * gg + kq));
    __builtin_amdgcn_s_setprio(0);
    f32x4 acc2[2][4];
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) {
      const f32x4 bf = *reinterpret_cast<const f32x4*>(Bf + 16 * cb + 4 * kq);
      acc2[0][cb] = bf;
      acc2[1][cb] = bf;
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      f32x4 wfr[4];
#pragma unroll
      for (int cb = 0; cb < 4; ++cb) wfr[cb] = *reinterpret_cast<const f32x4*>(Wf + ((cb * 4 + g) * 64 + lane) * 4);
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
        for (int cb = 0; cb < 4; ++cb)
#pragma unroll
          for (int pt = 0; pt < 2; ++pt) {
            const f32x4 xf = g < 2 ? yv[pt][g & 1] : shv[pt][g & 1];
            acc2[pt][cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(wfr[cb][s4], xf[s4], acc2[pt][cb], 0, 0, 0);
          }
    }
    __builtin_amdgcn_s_setprio(2);
    // ---- final epilogue: lane = pixel 16 pt + i16, accumulator = couts 16 cb + 4 kq .. + 3
#pragma unroll
    for (int pt = 0; pt < 2; ++pt)
#pragma unroll
      for (int cb = 0; cb < 4; ++cb) {
        f32x4 v = acc2[pt][cb];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = wn_silu(v[e]);
        if (AVG) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = (v[e] + r2v[pt][cb][e]) * p.postf;
        }
        const unsigned off = okp[pt] ? (unsigned)((mp[pt] * p.out_ld + p.out_off + 16 * cb + 4 * kq) * 4) : 0x80000000u;
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), orsrc, off, 0, 0);
      }
  }
#endif
}

}  // namespace

size_t csp_tail_frag_floats() { return (size_t)CT_WF_FLOATS; }

// packed: the final conv's folded fp32 weights [64][Kpad = 64] (K index = ci).  out[((cb * 4 + g) * 64 + l) * 4 + e] =
// W[co = 16 cb + (l & 15)][ci = 16 g + 4 (l >> 4) + e]: pw_resident_kernel's weight image - one 16-byte LDS read per lane =
// the A operands of 4 MFMA steps.
int csp_tail_pack_frags(const float* packed, float* out) {
  ST_REQUIRE(packed && out, "csp_tail_pack_frags: null pointer");
  for (int cb = 0; cb < 4; ++cb)
    for (int g = 0; g < 4; ++g)
      for (int l = 0; l < 64; ++l)
        for (int e = 0; e < 4; ++e)
          out[((cb * 4 + g) * 64 + l) * 4 + e] = packed[(size_t)(16 * cb + (l & 15)) * 64 + 16 * g + 4 * (l >> 4) + e];
  return ST_OK;
}

// conv2: 3x3 / s1 / p1, 32 -> 32, SiLU, with identity, Winograd weights present; fin: 1x1, 64 -> 64, SiLU, reading the
// concat whose first 32 channels conv2 would have written; optional residual on fin = the two-branch average.
bool csp_tail_applicable(const StConvDesc& c2, const StConvDesc& f) {
  if (!c2.wgt_wino_dev || !c2.res_dev || !c2.in_dev || !c2.bias_dev || !f.in_dev || !f.bias_dev || !f.out1_dev) return false;
  if (c2.KH != 3 || c2.KW != 3 || c2.stride != 1 || c2.pad != 1 || c2.Cin != 32 || c2.Cout != 32 || c2.act != 1) return false;
  if (c2.up_dev || c2.out2_dev || f.up_dev || f.out2_dev) return false;
  if (f.KH != 1 || f.KW != 1 || f.stride != 1 || f.pad != 0 || f.Cin != 64 || f.Cout != 64 || f.act != 1) return false;
  if (f.N != c2.N || f.Hi != c2.Hi || f.Wi != c2.Wi) return false;
  if (f.in_dev != c2.out1_dev || f.in_ld != c2.out1_ld || f.in_off != c2.out1_off) return false;   // conv2 -> cat[0:32]
  if (f.in_off + 64 > f.in_ld || f.out1_off + 64 > f.out1_ld || c2.in_off + 32 > c2.in_ld || c2.res_off + 32 > c2.res_ld)
    return false;
  if ((c2.in_ld | c2.in_off | c2.res_ld | c2.res_off | f.in_ld | f.in_off | f.out1_ld | f.out1_off) & 3) return false;
  if (f.res_dev && (((f.res_ld | f.res_off) & 3) || f.res_off + 64 > f.res_ld)) return false;
  const uintptr_t al = reinterpret_cast<uintptr_t>(c2.in_dev) | reinterpret_cast<uintptr_t>(c2.res_dev) |
                       reinterpret_cast<uintptr_t>(f.in_dev) | reinterpret_cast<uintptr_t>(f.out1_dev) |
                       reinterpret_cast<uintptr_t>(f.res_dev) | reinterpret_cast<uintptr_t>(c2.bias_dev) |
                       reinterpret_cast<uintptr_t>(f.bias_dev);
  if (al & 15) return false;
  const long long M = (long long)c2.N * c2.Hi * c2.Wi, lim = 1ll << 31;
  if (M * c2.in_ld * 4 >= lim || M * c2.res_ld * 4 >= lim || M * f.in_ld * 4 >= lim || M * f.out1_ld * 4 >= lim) return false;
  if (f.res_dev && M * f.res_ld * 4 >= lim) return false;
  return true;
}

int csp_tail_launch(const StConvDesc& c2, const StConvDesc& f, const float* frag_fin_dev, hipStream_t stream) {
  ST_REQUIRE(frag_fin_dev && (reinterpret_cast<uintptr_t>(frag_fin_dev) & 15) == 0, "csp tail: fragment weights missing");
  ST_REQUIRE(csp_tail_applicable(c2, f),
             "csp tail: needs conv2 = 3x3 s1 p1 32 -> 32 SiLU with identity + Winograd weights and final = 1x1 64 -> 64 SiLU "
             "reading the concat conv2 writes into (same N, H, W)");
  TailArgs a;
  std::memset(&a, 0, sizeof(a));
  const long long M = (long long)c2.N * c2.Hi * c2.Wi;
  a.in = c2.in_dev; a.wino = c2.wgt_wino_dev; a.bias2 = c2.bias_dev; a.res = c2.res_dev;
  a.sh = f.in_dev; a.wf = frag_fin_dev; a.biasf = f.bias_dev; a.res2 = f.res_dev; a.out = f.out1_dev;
  a.N = c2.N; a.H = c2.Hi; a.W = c2.Wi;
  a.in_ld = c2.in_ld; a.in_off = c2.in_off; a.res_ld = c2.res_ld; a.res_off = c2.res_off;
  a.sh_ld = f.in_ld; a.sh_off = f.in_off + 32;
  a.res2_ld = f.res_ld; a.res2_off = f.res_off; a.out_ld = f.out1_ld; a.out_off = f.out1_off;
  a.post2 = c2.post_scale; a.postf = f.res_dev ? f.post_scale : 1.0f;
  a.tbx = ceil_div(c2.Wi, 2 * CT_TX); a.tby = ceil_div(c2.Hi, 2 * CT_TY);
  const long long blocks = (long long)c2.N * a.tbx * a.tby;
  ST_REQUIRE(blocks < (1ll << 31), "csp tail: grid too large");
  a.nblocks = (unsigned)blocks;
  a.in_bytes = (unsigned)(M * c2.in_ld * 4); a.res_bytes = (unsigned)(M * c2.res_ld * 4);
  a.sh_bytes = (unsigned)(M * f.in_ld * 4); a.res2_bytes = f.res_dev ? (unsigned)(M * f.res_ld * 4) : 0u;
  a.out_bytes = (unsigned)(M * f.out1_ld * 4);
  // persistent: two workgroups per CU (LDS 71 KB each), fewer for small inputs
  static int cus = 0;
  if (cus == 0) {
    int dev = 0, n = 0;
    ST_CHECK_HIP(hipGetDevice(&dev));
    ST_CHECK_HIP(hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev));
    cus = std::max(1, n);
  }
  const unsigned grid = (unsigned)std::min<long long>(blocks, 2ll * cus);
  constexpr int lds = CT_LDS_FLOATS * (int)sizeof(float);
  if (f.res_dev) {
    static int lds_set = 0;
    ST_ENSURE_DYNAMIC_LDS(wino_csp_tail_kernel<true>, lds, lds_set);
    hipLaunchKernelGGL(wino_csp_tail_kernel<true>, dim3(grid), dim3(256), lds, stream, a);
  } else {
    static int lds_set = 0;
    ST_ENSURE_DYNAMIC_LDS(wino_csp_tail_kernel<false>, lds, lds_set);
    hipLaunchKernelGGL(wino_csp_tail_kernel<false>, dim3(grid), dim3(256), lds, stream, a);
  }
  ST_CHECK_HIP(hipGetLastError());
  return ST_OK;
}

}  // namespace st

extern "C" size_t st_csp_tail_frag_floats(void) { return st::csp_tail_frag_floats(); }

extern "C" int st_csp_tail_pack_frags(const float* packed_wgt_host, float* out_host) {
  return st::csp_tail_pack_frags(packed_wgt_host, out_host);
}

extern "C" int st_conv3x3_csp_tail(const StConvDesc* conv2, const StConvDesc* fin, const float* frag_fin_dev,
                                   st_stream_t stream) {
  if (!conv2 || !fin) return st::set_error(ST_ERR_INVALID, "st_conv3x3_csp_tail: null descriptor");
  return st::csp_tail_launch(*conv2, *fin, frag_fin_dev, static_cast<hipStream_t>(stream));
}
