// Stereo cost volume -> soft-argmin disparity (the module BASELINE.json's north_star adds; the
// reference has no counterpart - it loads SGBM disparity PNGs, loading_disparity.py:71-134).
// Specification = oracle/st_oracle.c (oracle_costvolume / oracle_softargmin / oracle_disp_upsample):
//
//   cost[n,y,x,d] = ( sum_{c=0..C-1, in order, fmaf} L[n,y,x,c] * R[n,y,x-d,c] ) / C   for x-d >= 0
//                 = 0                                                                   otherwise
//   disp_lr[n,y,x] = sum_d d*e_d / sum_d e_d,  e_d = exp(T*cost_d - max_d T*cost_d)
//   disp[n,0..2,Y,X] = scale * bilinear_x`scale`(disp_lr) (align_corners=False) inside
//                      (valid_h, valid_w), 0 outside  -> the `disp_postp` tensor the detector's
//                      disparity branch and ocsort_disparity.py:115,132-134 consume.
//
// Mapping: HBM-lean VALU kernel.  One workgroup = one row segment of 64 pixels; the L tile and
// the R tile (64 + D - 1 pixels) are read once from HBM/L2 with coalesced 16 B loads and
// transposed into LDS as [c][x] (row length = 1 mod 32 -> conflict-free strided writes), so the
// c-loop reads are lane-contiguous.  Lane = (disparity group dg = lane/16, pixel = lane%16): each
// lane keeps DG = ceil(D/4) accumulators in registers; the per-pixel soft-argmin partials
// (max, sum e, sum d*e) of the 4 disparity groups are merged with wavefront shuffles
// (__shfl_xor 16, 32), so the D x H x W volume is only written when the caller asks for it.
#include <algorithm>
#include <cstdint>
#include <cstdlib>

#include "st_common.h"

namespace st {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float cv_expf(float x) {  // same polynomial as decode_nms.hip / oracle
  if (x > 88.72283f) return __builtin_inff();
  if (x < -103.0f) return 0.0f;
  const float n = rintf(x * 1.44269504088896341f);
  float r = fmaf(n, -0.693359375f, x);
  r = fmaf(n, 2.12194440e-4f, r);
  float p = 1.9875691500e-4f;
  p = fmaf(p, r, 1.3981999507e-3f);
  p = fmaf(p, r, 8.3334519073e-3f);
  p = fmaf(p, r, 4.1665795894e-2f);
  p = fmaf(p, r, 1.6666665459e-1f);
  p = fmaf(p, r, 5.0000001201e-1f);
  const float r2 = r * r;
  p = fmaf(p, r2, r);
  p = p + 1.0f;
  return ldexpf(p, (int)n);
}

// cv_expf on TWO values per instruction (packed fp32, default operand selection only - every constant is an explicit
// (c, c) scalar register pair): element for element the operations of cv_expf above, in its order, so the
// results are bit-identical.  For arguments <= 0 (soft-argmin: x = T c - max): the > 88.7 branch cannot be taken.
struct CvExpPk {
  f32x2 log2e, nln2hi, ln2lo, c0, c1, c2, c3, c4, c5, one;
};
__device__ __forceinline__ f32x2 cv_dup_s(float c) {
  f32x2 p = {c, c};
  asm volatile("" : "+s"(p));     // opaque: the broadcast cannot be folded into op_sel
  return p;
}
__device__ __forceinline__ CvExpPk cv_exp_pk_consts() {
  CvExpPk k;
  k.log2e = cv_dup_s(1.44269504088896341f); k.nln2hi = cv_dup_s(-0.693359375f); k.ln2lo = cv_dup_s(2.12194440e-4f);
  k.c0 = cv_dup_s(1.9875691500e-4f); k.c1 = cv_dup_s(1.3981999507e-3f); k.c2 = cv_dup_s(8.3334519073e-3f);
  k.c3 = cv_dup_s(4.1665795894e-2f); k.c4 = cv_dup_s(1.6666665459e-1f); k.c5 = cv_dup_s(5.0000001201e-1f);
  k.one = cv_dup_s(1.0f);
  return k;
}
// NB pairs advance stage by stage (a dependent packed FMA pays a wait state; NB independent chains fill it): the
// scheduling barriers keep the stages apart, the arithmetic per element is cv_expf's.
template <int NB>
__device__ __forceinline__ void cv_expf_pk_nonpos(f32x2 (&x)[NB], const CvExpPk& k) {
  f32x2 n[NB], r[NB], p[NB];
#define CV_STAGE(body)                       \
  _Pragma("unroll") for (int b = 0; b < NB; ++b) { body; } \
  __builtin_amdgcn_sched_barrier(0);
  CV_STAGE(r[b] = x[b] * k.log2e)
  CV_STAGE((n[b] = f32x2{rintf(r[b][0]), rintf(r[b][1])}))
  CV_STAGE(r[b] = __builtin_elementwise_fma(n[b], k.nln2hi, x[b]))
  CV_STAGE(r[b] = __builtin_elementwise_fma(n[b], k.ln2lo, r[b]))
  CV_STAGE(p[b] = __builtin_elementwise_fma(k.c0, r[b], k.c1))
  CV_STAGE(p[b] = __builtin_elementwise_fma(p[b], r[b], k.c2))
  CV_STAGE(p[b] = __builtin_elementwise_fma(p[b], r[b], k.c3))
  CV_STAGE(p[b] = __builtin_elementwise_fma(p[b], r[b], k.c4))
  CV_STAGE(p[b] = __builtin_elementwise_fma(p[b], r[b], k.c5))
  CV_STAGE((p[b] = __builtin_elementwise_fma(p[b], r[b] * r[b], r[b])))
  CV_STAGE(p[b] = p[b] + k.one)
#undef CV_STAGE
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    f32x2 e = {ldexpf(p[b][0], (int)n[b][0]), ldexpf(p[b][1], (int)n[b][1])};
    if (x[b][0] < -103.0f) e[0] = 0.0f;
    if (x[b][1] < -103.0f) e[1] = 0.0f;
    x[b] = e;
  }
}

constexpr int CV_TX = 64;     // pixels per workgroup
constexpr int CV_MAXDG = 64;  // accumulators per lane (D <= 256)

template <int DG>
__global__ __launch_bounds__(256) void costvolume_kernel(const float* __restrict__ featL,
                                                         const float* __restrict__ featR, int Hf, int Wf,
                                                         int C, int ld, int D, float temperature,
                                                         int rowL, int rowR, float* __restrict__ out_cost,
                                                         float* __restrict__ out_disp) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Ls = smem;             // [C][rowL]
  float* Rs = smem + C * rowL;  // [C][rowR], column j <-> x' = x0 - (D-1) + j
  const int x0 = blockIdx.x * CV_TX;
  const int y = blockIdx.y, n = blockIdx.z;
  const size_t rowbase = ((size_t)n * Hf + y) * Wf;
  const int tid = threadIdx.x;
  const int C4 = C >> 2;

  // ---- stage + transpose: coalesced float4 global reads (lanes along c), strided LDS writes
  for (int e = tid; e < CV_TX * C4; e += 256) {
    const int px = e / C4, c4 = e - px * C4;
    const int x = x0 + px;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (x < Wf) v = *reinterpret_cast<const f32x4*>(featL + (rowbase + x) * ld + 4 * c4);
#pragma unroll
    for (int k = 0; k < 4; ++k) Ls[(4 * c4 + k) * rowL + px] = v[k];
  }
  const int RW = CV_TX + D - 1;
  for (int e = tid; e < RW * C4; e += 256) {
    const int j = e / C4, c4 = e - j * C4;
    const int x = x0 - (D - 1) + j;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (x >= 0 && x < Wf) v = *reinterpret_cast<const f32x4*>(featR + (rowbase + x) * ld + 4 * c4);
#pragma unroll
    for (int k = 0; k < 4; ++k) Rs[(4 * c4 + k) * rowR + j] = v[k];
  }
  __syncthreads();

  const int lane = tid & 63, wave = tid >> 6;
  const int dg = lane >> 4;
  const int px = wave * 16 + (lane & 15);
  const int x = x0 + px;
  const int d0 = dg * DG;
  float acc[DG];
#pragma unroll
  for (int k = 0; k < DG; ++k) acc[k] = 0.f;
  // Rs column of (x, d) is px + (D-1) - d; for this lane d = d0 + k
  const float* lp = Ls + px;
  const float* rp = Rs + px + (D - 1) - d0;
  for (int c = 0; c < C; ++c) {
    const float a = lp[c * rowL];
    const float* rr = rp + c * rowR;
#pragma unroll
    for (int k = 0; k < DG; ++k) acc[k] = fmaf(a, rr[-k], acc[k]);  // rr[-k] is >= Rs row start while d0+k < D
  }
  const float fC = (float)C;
  float m = -__builtin_inff();
#pragma unroll
  for (int k = 0; k < DG; ++k) {
    const int d = d0 + k;
    float cst = acc[k] / fC;
    if (x - d < 0) cst = 0.f;
    acc[k] = cst;
    if (d < D) m = fmaxf(m, temperature * cst);
  }
  if (out_cost && x < Wf) {
    float* oc = out_cost + (rowbase + x) * (size_t)D + d0;
#pragma unroll
    for (int k = 0; k < DG; ++k)
      if (d0 + k < D) oc[k] = acc[k];
  }
  // soft-argmin: merge (max, sum e, sum d*e) over the 4 disparity groups with shuffles
  m = fmaxf(m, __shfl_xor(m, 16));
  m = fmaxf(m, __shfl_xor(m, 32));
  float s = 0.f, t = 0.f;
#pragma unroll
  for (int k = 0; k < DG; ++k) {
    const int d = d0 + k;
    if (d < D) {
      const float e = cv_expf(temperature * acc[k] - m);
      s += e;
      t = fmaf((float)d, e, t);
    }
  }
  s += __shfl_xor(s, 16); t += __shfl_xor(t, 16);
  s += __shfl_xor(s, 32); t += __shfl_xor(t, 32);
  if (dg == 0 && x < Wf && out_disp) out_disp[rowbase + x] = t / s;
}

// Register-tiled correlation: lane = 4 CONSECUTIVE pixels x DG disparities.  cost[x+p][d0+k] needs R[x+p-d0-k]:
// for the lane's 4 x DG outputs that is one window of DG+3 consecutive R columns per channel, so a channel step
// costs 1 + (DG+4)/4 ds_read_b128 for 4*DG FMAs (DG = 12: 5 reads per 48 FMAs; the one-pixel kernel above needs
// 13 ds_read_b32 per 12 FMAs and is LDS-bandwidth bound).  Same c-ascending fmaf chain per output => the volume
// stays bit-identical to the oracle.  Workgroup = 2 waves = 128 pixels of one feature row; channels are staged
// in chunks of CVT_CC (transposed [c][x], float4-aligned rows with an odd float4 stride: conflict-free b128
// reads AND conflict-free transposing writes with the lane = (16 pixels x 4 channel-quads) staging order).
constexpr int CVT_TX = 128;   // pixels per workgroup of the 2-wave form (NW waves: 64 * NW)
constexpr int CVT_CC = 16;    // channels per staging chunk (two LDS buffers)
static inline int cvt_row(int n) {  // >= n, multiple of 4, (row / 4) odd
  int r = (n + 3) & ~3;
  if (((r >> 2) & 1) == 0) r += 4;
  return r;
}

// FMA: how the diagonal accumulator pairs (below) are evaluated.  Every form runs the same fmaf chains, bit-identical:
//   0  two scalar v_fma_f32 per pair;
//   1  one v_pk_fma_f32 with the R operand broadcast by op_sel / op_sel_hi (what the compiler makes of f32x2{r, r});
//   2  one v_pk_fma_f32 WITHOUT op_sel on an explicit (r, r) register pair (one v_mov pair per window element and channel);
//   3  ROW pairs: (acc[p][k], acc[p][k-1]) += (L[p], L[p]) * (R[i], R[i+1]) with i = DG + p - k EVEN, so that the R pair is an
//      aligned 64-bit half of the ds_read_b128 result and only the four L values need an explicit (l, l) pair: one
//      v_pk_fma_f32 without op_sel, 22 packed + 4 scalar FMAs + 8 moves per channel (form 0: 48, form 1: 22 + 4 + 3).
// Round 4 found form 1 returning wrong sums while bf16 MFMAs of another kernel execute on the chip (tools/cv_stress.py,
// tools/micro/pkfma_corun.hip): the product library instantiates and launches form 3 only (see
// st_costvolume_softargmin; tests/test_cpu_oracle_and_abi.py checks the built code object for packed-fp32 op_sel).
// NWV: waves per workgroup (64 pixels each).  2 = round 2's form (40 KB of LDS per 128 pixels: three workgroups = 6 waves per
// CU, LDS-limited); 4 = 256 pixels share one R tile (9 % fewer staged floats, 73 KB per workgroup: two workgroups = 8
// waves per CU, register-limited).
template <int DG, int FMA, int NWV>
__global__ __launch_bounds__(64 * NWV) void costvolume_tiled_kernel(const float* __restrict__ featL,
                                                               const float* __restrict__ featR, int Hf, int Wf,
                                                               int C, int ld, int D, float temperature, int rowL,
                                                               int rowR, float* __restrict__ out_cost,
                                                               float* __restrict__ out_disp, int dbase, int gx,
                                                               int total, int per_xcd) {
  // dbase: first disparity of this launch's slab [dbase, dbase + 4*DG) - wide volumes (D > 128) are materialised
  // slab by slab (the fused soft-argmin needs all of D in one launch: out_disp must be null when dbase > 0)
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int bufsz = CVT_CC * (rowL + rowR);   // two buffers: [CVT_CC][rowL] + [CVT_CC][rowR] each
  // XCD-aware tile order: workgroups are dealt round-robin to the 8 XCDs (each with an L2 of its own), so the
  // linear id is re-read as (xcd, slot) and every XCD walks a CONTIGUOUS range of tiles - the 4*DG-pixel R halo two
  // neighbouring row segments share is then fetched from memory once, by the L2 both of them sit behind (it was
  // fetched twice: 1.22x the algorithmic bytes, profiles/r03_hbm_traffic.txt).
  const int w = (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3);
  if (w >= total) return;                    // uniform
  constexpr int CVT_TX = 64 * NWV, NT = 64 * NWV;   // (shadows the namespace constant: pixels / threads of THIS form)
  const int x0 = (w % gx) * CVT_TX;
  const int y = (w / gx) % Hf, n = w / (gx * Hf);
  const size_t rowbase = ((size_t)n * Hf + y) * Wf;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int dg = lane >> 4, pgi = lane & 15;
  const int px0 = wave * 64 + 4 * pgi;   // first of this lane's 4 pixels (block-relative)
  const int dl = dg * DG;                // this lane group's first disparity inside the slab
  const int d0 = dbase + dl;
  constexpr int NW = DG + 4;             // floats of the R window: x' = x - d0 - DG + i, i = 0 .. DG+3
  constexpr int RW = CVT_TX + 4 * DG;    // R columns of the block: column j <-> x' = x0 - dbase - 4*DG + j
  constexpr int RW16 = (RW + 15) & ~15;
  constexpr int CC4 = CVT_CC / 4;
  constexpr int NL = CVT_TX * CC4 / NT, NR = (RW16 * CC4 + NT - 1) / NT;   // float4 per thread per chunk

  // Accumulators in DIAGONAL pairs: cost[x+p][d0+k] and cost[x+p+1][d0+k+1] both multiply R window element
  // i = DG + p - k, so (acc[p][k], acc[p+1][k+1]) += (L[p], L[p+1]) * R[i] is ONE v_pk_fma_f32 with the R operand
  // broadcast by op_sel - no register shuffles (what SLP vectorisation of the scalar loop paid for its packing).
  // Per channel: 2 (DG - 1) packed + 4 scalar FMAs instead of 4 DG; each output's chain is still fmaf over c
  // ascending, so the volume is bit-identical.  accs[q][0] = acc[2q][DG-1], accs[q][1] = acc[2q+1][0].
  f32x2 accd[2][DG - 1];
  float accs[2][2];
#pragma unroll
  for (int q = 0; q < 2; ++q) {
#pragma unroll
    for (int k = 0; k < DG - 1; ++k) accd[q][k] = f32x2{0.f, 0.f};
    accs[q][0] = accs[q][1] = 0.f;
  }
  // Form 3, ROW pairs: accr[p][j] = (acc[p][k], acc[p][k-1]) with k = 2j + 2 for the even pixels (j < DG/2 - 1; k = 0 and
  // k = DG-1 stay scalar in acce[p/2][0..1]) and k = 2j + 1 for the odd pixels (j < DG/2): then the window index
  // i = DG + p - k of the pair's first element is even and (R[i], R[i+1]) is one aligned register pair.
  f32x2 accr[FMA == 3 ? 4 : 1][FMA == 3 ? DG / 2 : 1];
  float acce[2][2] = {{0.f, 0.f}, {0.f, 0.f}};
  if (FMA == 3) {
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
      for (int j = 0; j < DG / 2; ++j) accr[p][j] = f32x2{0.f, 0.f};
  }

  // Staging of one channel chunk, transposed to [c][x].  Lane order: 16 consecutive lanes = 16 consecutive
  // pixels of one channel quad (conflict-free transposing LDS writes with rows of 4 * odd floats).  The global
  // loads of chunk i+1 are issued BEFORE the FMAs of chunk i and stored to the other LDS buffer after them:
  // a load -> LDS-store loop serialises one HBM/L2 round trip per iteration, and that latency chain (not LDS
  // bandwidth) is what bounded the one-pixel kernel above.
  f32x4 stL[NL], stR[NR];
  auto stage_load = [&](int cb) {
    const int cc4 = min(CVT_CC, C - cb) >> 2;
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      const int e = tid + NT * i;
      const int px = (e & 15) + 16 * (e / (16 * CC4)), c4 = (e >> 4) % CC4;
      const int x = x0 + px;
      stL[i] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (c4 < cc4 && x < Wf) stL[i] = *reinterpret_cast<const f32x4*>(featL + (rowbase + x) * ld + cb + 4 * c4);
    }
#pragma unroll
    for (int i = 0; i < NR; ++i) {
      const int e = tid + NT * i;
      const int j = (e & 15) + 16 * (e / (16 * CC4)), c4 = (e >> 4) % CC4;
      const int x = x0 - dbase - 4 * DG + j;
      stR[i] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (c4 < cc4 && j < RW && x >= 0 && x < Wf)
        stR[i] = *reinterpret_cast<const f32x4*>(featR + (rowbase + x) * ld + cb + 4 * c4);
    }
  };
  auto stage_store = [&](int buf) {
    float* Ls = smem + buf * bufsz;
    float* Rs = Ls + CVT_CC * rowL;
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      const int e = tid + NT * i;
      const int px = (e & 15) + 16 * (e / (16 * CC4)), c4 = (e >> 4) % CC4;
#pragma unroll
      for (int k = 0; k < 4; ++k) Ls[(4 * c4 + k) * rowL + px] = stL[i][k];
    }
#pragma unroll
    for (int i = 0; i < NR; ++i) {
      const int e = tid + NT * i;
      const int j = (e & 15) + 16 * (e / (16 * CC4)), c4 = (e >> 4) % CC4;
      if (j < rowR) {
#pragma unroll
        for (int k = 0; k < 4; ++k) Rs[(4 * c4 + k) * rowR + j] = stR[i][k];
      }
    }
  };

  stage_load(0);
  stage_store(0);
  __syncthreads();
  int buf = 0;
  for (int cb = 0; cb < C; cb += CVT_CC, buf ^= 1) {
    const int cc = min(CVT_CC, C - cb);
    const bool more = cb + CVT_CC < C;
    if (more) stage_load(cb + CVT_CC);   // in flight during the FMAs below
    const float* lp = smem + buf * bufsz + px0;
    const float* rp = smem + buf * bufsz + CVT_CC * rowL + px0 + 4 * DG - dl - DG;   // window start (multiple of 4)
    // ---- accumulate this chunk, operands of channel c+1 prefetched while channel c multiplies
    f32x4 lv[2], rv[2][NW / 4];
    lv[0] = *reinterpret_cast<const f32x4*>(lp);
#pragma unroll
    for (int q = 0; q < NW / 4; ++q) rv[0][q] = *reinterpret_cast<const f32x4*>(rp + 4 * q);
    // A wave whose 64 pixels all lie right of the image (the second wave of a row's last segment when Wf % 128 is in
    // 1..64, e.g. Wf = 320 = 2.5 segments: one wave in six) stages and synchronises like the others but multiplies
    // nothing: its accumulators stay 0 and every store of it is masked anyway.  Wave-uniform.
    const int cend = (x0 + wave * 64 < Wf) ? cc : 0;
    for (int c = 0; c < cend; c += 2) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        // UNCONDITIONAL prefetch of channel c+h+1: after the chunk's last channel it reads one row past the
        // chunk (the next region of the LDS allocation, which carries one slack row for this) and the values
        // are never used.  With a `cn < cc` guard the loop body had branches, the compiler waited for every
        // read batch right after issuing it and copied one register set into the other per iteration:
        // 115 -> 93 us for the bench volume on random features (timing-only ablations).
        const int cn = c + h + 1;
        lv[h ^ 1] = *reinterpret_cast<const f32x4*>(lp + cn * rowL);
#pragma unroll
        for (int q = 0; q < NW / 4; ++q) rv[h ^ 1][q] = *reinterpret_cast<const f32x4*>(rp + cn * rowR + 4 * q);
        if (FMA == 3) {
#pragma unroll
          for (int p = 0; p < 4; ++p) {
            const float l = lv[h][p];
            f32x2 ld = {l, l};
            asm volatile("" : "+v"(ld));   // an explicit (l, l) register pair: the broadcast must not become op_sel
            const int odd = p & 1;
#pragma unroll
            for (int j = 0; j < DG / 2 - 1 + odd; ++j) {
              const int k = 2 * j + 2 - odd;          // the pair's first disparity; i = window index of (p, k), even
              const int i = DG + p - k;
              const f32x4 w = rv[h][i >> 2];
              const f32x2 rp = (i & 2) ? f32x2{w[2], w[3]} : f32x2{w[0], w[1]};
              accr[p][j] = __builtin_elementwise_fma(ld, rp, accr[p][j]);
            }
            if (!odd) {   // k = 0 (i = DG + p) and k = DG - 1 (i = p + 1) of the even pixels
              acce[p >> 1][0] = fmaf(l, rv[h][(DG + p) >> 2][(DG + p) & 3], acce[p >> 1][0]);
              acce[p >> 1][1] = fmaf(l, rv[h][(p + 1) >> 2][(p + 1) & 3], acce[p >> 1][1]);
            }
          }
        } else {
        // FMA == 2: window elements 2 .. DG+2 as explicit (r, r) pairs; the empty asm makes each pair an opaque value, so
        // the compiler has to build it in a register pair of its own instead of folding the broadcast into op_sel
        f32x2 rdup[FMA == 2 ? DG + 1 : 1];
        if (FMA == 2) {
#pragma unroll
          for (int i = 2; i <= DG + 2; ++i) {
            const float r = rv[h][i >> 2][i & 3];
            rdup[i - 2] = f32x2{r, r};
            asm volatile("" : "+v"(rdup[i - 2]));
          }
        }
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const f32x2 lpair = q ? f32x2{lv[h][2], lv[h][3]} : f32x2{lv[h][0], lv[h][1]};
#pragma unroll
          for (int k = 0; k < DG - 1; ++k) {
            const int i = DG + 2 * q - k;   // window index of x + 2q - d0 - k (= that of x + 2q + 1 - d0 - (k + 1))
            const float r = rv[h][i >> 2][i & 3];
            if (FMA == 1) accd[q][k] = __builtin_elementwise_fma(lpair, f32x2{r, r}, accd[q][k]);
            else if (FMA == 2) accd[q][k] = __builtin_elementwise_fma(lpair, rdup[i - 2], accd[q][k]);
            else accd[q][k] = f32x2{fmaf(lpair[0], r, accd[q][k][0]), fmaf(lpair[1], r, accd[q][k][1])};
          }
          constexpr int ilo = 1, ihi = DG + 1;   // + 2q: windows of (p = 2q, k = DG-1) and (p = 2q+1, k = 0)
          accs[q][0] = fmaf(lpair[0], rv[h][(ilo + 2 * q) >> 2][(ilo + 2 * q) & 3], accs[q][0]);
          accs[q][1] = fmaf(lpair[1], rv[h][(ihi + 2 * q) >> 2][(ihi + 2 * q) & 3], accs[q][1]);
        }
        }
      }
    }
    if (more) stage_store(buf ^ 1);
    __syncthreads();
  }

  float acc[4][DG];   // back to [pixel][disparity] (register renaming only)
#pragma unroll
  for (int q = 0; q < 2; ++q) {
#pragma unroll
    for (int k = 0; k < DG - 1; ++k) {
      acc[2 * q][k] = accd[q][k][0];
      acc[2 * q + 1][k + 1] = accd[q][k][1];
    }
    acc[2 * q][DG - 1] = accs[q][0];
    acc[2 * q + 1][0] = accs[q][1];
  }
  if (FMA == 3) {
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int odd = p & 1;
#pragma unroll
      for (int j = 0; j < DG / 2 - 1 + odd; ++j) {
        const int k = 2 * j + 2 - odd;
        acc[p][k] = accr[p][j][0];
        acc[p][k - 1] = accr[p][j][1];
      }
      if (!odd) {
        acc[p][0] = acce[p >> 1][0];
        acc[p][DG - 1] = acce[p >> 1][1];
      }
    }
  }
  // ---- cost = acc / C (0 where the match falls left of the image), optional volume store, fused soft-argmin
  const float fC = (float)C;
  const bool pow2 = (C & (C - 1)) == 0;   // then x * (1/C) == x / C exactly (no subnormal results here)
  const float rC = 1.0f / fC;
  float m[4], s[4], t[4];
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const int x = x0 + px0 + p;
    m[p] = -__builtin_inff();
#pragma unroll
    for (int k = 0; k < DG; ++k) {
      const int d = d0 + k;
      float cst = pow2 ? acc[p][k] * rC : acc[p][k] / fC;
      if (x - d < 0) cst = 0.f;
      acc[p][k] = cst;
      if (d < D) m[p] = fmaxf(m[p], temperature * cst);
    }
    if (out_cost && x < Wf) {
      float* oc = out_cost + (rowbase + x) * (size_t)D + d0;
      if ((D & 3) == 0 && d0 + DG <= D) {
#pragma unroll
        for (int q = 0; q < DG / 4; ++q) {
          const f32x4 v = {acc[p][4 * q], acc[p][4 * q + 1], acc[p][4 * q + 2], acc[p][4 * q + 3]};
          *reinterpret_cast<f32x4*>(oc + 4 * q) = v;
        }
      } else {
#pragma unroll
        for (int k = 0; k < DG; ++k)
          if (d0 + k < D) oc[k] = acc[p][k];
      }
    }
  }
  if (!out_disp) return;   // uniform
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    m[p] = fmaxf(m[p], __shfl_xor(m[p], 16));
    m[p] = fmaxf(m[p], __shfl_xor(m[p], 32));
    s[p] = 0.f; t[p] = 0.f;
#pragma unroll
    for (int k = 0; k < DG; ++k) {
      const int d = d0 + k;
      if (d < D) {
        const float e = cv_expf(temperature * acc[p][k] - m[p]);
        s[p] += e;
        t[p] = fmaf((float)d, e, t[p]);
      }
    }
    s[p] += __shfl_xor(s[p], 16); t[p] += __shfl_xor(t[p], 16);
    s[p] += __shfl_xor(s[p], 32); t[p] += __shfl_xor(t[p], 32);
    const int x = x0 + px0 + p;
    if (dg == 0 && x < Wf) out_disp[rowbase + x] = t[p] / s[p];
  }
}

__global__ __launch_bounds__(256) void softargmin_kernel(const float* __restrict__ cost, long long npix, int D,
                                                         float temperature, float* __restrict__ out_disp) {
  for (long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x; p < npix;
       p += (long long)gridDim.x * blockDim.x) {
    const float* c = cost + p * D;
    float m = -__builtin_inff();
    for (int d = 0; d < D; ++d) m = fmaxf(m, temperature * c[d]);
    float s = 0.f, t = 0.f;
    for (int d = 0; d < D; ++d) {
      const float e = cv_expf(temperature * c[d] - m);
      s += e;
      t = fmaf((float)d, e, t);
    }
    out_disp[p] = t / s;
  }
}

// Same arithmetic (same order: bit-identical results), but HBM-friendly: the block's pixels x D floats are one
// contiguous chunk, copied with coalesced float4 loads into LDS rows of `ldsw` floats (D + 4 or + 8, so that
// ldsw/4 is odd: 16 lanes x ds_read_b128 hit 64 distinct banks), then lane = pixel walks its row twice.
// The strided per-lane global reads of softargmin_kernel cost 484 us on the 8x184x320x48 volume; this one is
// bound by the 90 MB read.
__global__ __launch_bounds__(256) void softargmin_lds_kernel(const float* __restrict__ cost, long long npix, int D,
                                                             int ldsw, float temperature,
                                                             float* __restrict__ out_disp) {
  extern __shared__ float4 sa_smem4[];
  float* sm = reinterpret_cast<float*>(sa_smem4);
  const int PB = blockDim.x;
  const long long p0 = (long long)blockIdx.x * PB;
  const int np = (int)((npix - p0) < (long long)PB ? (npix - p0) : (long long)PB);
  const int D4 = D >> 2;
  const float4* src = reinterpret_cast<const float4*>(cost + p0 * D);
  const int nf4 = np * D4;
  for (int f0 = threadIdx.x; f0 < nf4; f0 += 4 * PB) {   // four loads in flight per thread, then their LDS writes
    float4 t[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int f = f0 + j * PB;
      if (f < nf4) t[j] = src[f];
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int f = f0 + j * PB;
      if (f < nf4) {
        const int pix = f / D4, k = f - pix * D4;
        *reinterpret_cast<float4*>(sm + pix * ldsw + 4 * k) = t[j];
      }
    }
  }
  __syncthreads();
  if ((int)threadIdx.x >= np) return;
  const float* c = sm + threadIdx.x * ldsw;
  float m = -__builtin_inff();
  for (int k = 0; k < D4; ++k) {
    const float4 v = *reinterpret_cast<const float4*>(c + 4 * k);
    m = fmaxf(m, temperature * v.x); m = fmaxf(m, temperature * v.y);
    m = fmaxf(m, temperature * v.z); m = fmaxf(m, temperature * v.w);
  }
  // exponentials two per instruction (cv_expf_pk_nonpos: element for element cv_expf), sums in the oracle's order
  const CvExpPk ek = cv_exp_pk_consts();
  const f32x2 t2 = cv_dup_s(temperature);
  f32x2 negm = {-m, -m};
  asm volatile("" : "+v"(negm));
  float s = 0.f, t = 0.f;
  int k = 0;
  for (; k + 2 <= D4; k += 2) {
    const f32x4 va = *reinterpret_cast<const f32x4*>(c + 4 * k), vb = *reinterpret_cast<const f32x4*>(c + 4 * k + 4);
    f32x2 x[4] = {f32x2{va[0], va[1]}, f32x2{va[2], va[3]}, f32x2{vb[0], vb[1]}, f32x2{vb[2], vb[3]}};
#pragma unroll
    for (int b = 0; b < 4; ++b) x[b] = t2 * x[b] + negm;
    cv_expf_pk_nonpos<4>(x, ek);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float e = x[j >> 1][j & 1];
      s += e;
      t = fmaf((float)(4 * k + j), e, t);
    }
  }
  if (k < D4) {
    const f32x4 va = *reinterpret_cast<const f32x4*>(c + 4 * k);
    f32x2 x[2] = {f32x2{va[0], va[1]}, f32x2{va[2], va[3]}};
#pragma unroll
    for (int b = 0; b < 2; ++b) x[b] = t2 * x[b] + negm;
    cv_expf_pk_nonpos<2>(x, ek);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float e = x[j >> 1][j & 1];
      s += e;
      t = fmaf((float)(4 * k + j), e, t);
    }
  }
  out_disp[p0 + threadIdx.x] = t / s;
}

// Volumes of 16 .. 192 levels in steps of 16 (the benched 48, the full-resolution mode's 192).  The LDS kernel above (other
// level counts) keeps a pixel's whole row in
// LDS (784 B at D = 192: 64 pixels = one wave per 50 KB, three waves per CU - measured 5.8 ms for the 5.8 GB volume of 8
// pairs, 0.12 of 8 TB/s).  Here the row lives in REGISTERS, split over K = 2 neighbouring lanes when D / 16 is even (96
// registers per lane at D = 192): a wave owns 64 / K pixels and brings their rows in through a private 5 KB LDS tile, 16
// disparities per lane at a time (coalesced 16-byte loads: four lanes cover one 64-byte chunk; tile rows of 20
// floats: conflict-free 16-byte reads).  The maximum is exact in any order (lanes combine by shuffle); every lane
// evaluates the exponentials of ITS values (exact per element, in parallel); only the two running sums are a serial
// chain in the oracle's order - lane 0 of a pixel runs d = 0 .. D/K-1, hands (s, t) to lane 1 by shuffle, which
// continues: bit-identical to the sequential evaluation.
template <int NCH, int K>
__global__ __launch_bounds__(256) void softargmin_reg_kernel(const float* __restrict__ cost, long long npix,
                                                             float temperature, float* __restrict__ out_disp) {
  constexpr int D = 16 * NCH, LCH = NCH / K, DL = 16 * LCH, PPW = 64 / K;   // chunks / values per lane, pixels per wave
  static_assert(NCH % K == 0, "the row must split evenly over the K lanes of a pixel");
  __shared__ __attribute__((aligned(16))) float tile[4][64 * 20];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int sub = lane % K;
  float* tw = tile[wave];
  const long long nwave = (npix + PPW - 1) / PPW;
  const CvExpPk ek = cv_exp_pk_consts();
  const f32x2 t2 = cv_dup_s(temperature);
  for (long long wv = (long long)blockIdx.x * 4 + wave; wv < nwave; wv += (long long)gridDim.x * 4) {
    const long long p0 = wv * PPW;
    f32x2 v2[DL / 2];   // the lane's values as register PAIRS: the scaling and the exponentials run two per instruction
    // chunk ch + 1 is requested from memory BEFORE chunk ch goes through the tile: one round trip in flight behind the
    // LDS transposition of the previous one (a wave has few neighbours here: two waves per SIMD)
    f32x4 xa[4];
    auto load_chunk = [&](int ch, f32x4 (&x)[4]) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int e = lane + 64 * i, row = e >> 2, q = e & 3;     // tile row = the lane that will read it
        const int px = row / K, sb = row % K;
        x[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (p0 + px < npix) x[i] = *reinterpret_cast<const f32x4*>(cost + (p0 + px) * D + sb * DL + ch * 16 + 4 * q);
      }
    };
    load_chunk(0, xa);
#pragma unroll
    for (int ch = 0; ch < LCH; ++ch) {
      f32x4 xb[4];
      if (ch + 1 < LCH) load_chunk(ch + 1, xb);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int e = lane + 64 * i, row = e >> 2, q = e & 3;
        *reinterpret_cast<f32x4*>(tw + row * 20 + 4 * q) = xa[i];
      }
      // the tile is private to this wave and LDS executes a wave's instructions in order: no workgroup barrier, the
      // fences keep the compiler from reordering and make it wait for the writes before the reads' data is used
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const f32x4 r = *reinterpret_cast<const f32x4*>(tw + lane * 20 + 4 * j);
        v2[ch * 8 + 2 * j] = f32x2{r[0], r[1]};
        v2[ch * 8 + 2 * j + 1] = f32x2{r[2], r[3]};
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();     // the next chunk overwrites the tile: every lane has read its row
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      if (ch + 1 < LCH) {
#pragma unroll
        for (int i = 0; i < 4; ++i) xa[i] = xb[i];
      }
    }
    // T c once (the oracle forms the same product for the maximum and for the exponent), maximum, exp(T c - max)
#pragma unroll
    for (int i = 0; i < DL / 2; ++i) v2[i] = t2 * v2[i];
    float m = -__builtin_inff();
#pragma unroll
    for (int i = 0; i < DL / 2; ++i) m = fmaxf(fmaxf(m, v2[i][0]), v2[i][1]);
    if (K == 2) m = fmaxf(m, __shfl_xor(m, 1));
    f32x2 negm = {-m, -m};
    asm volatile("" : "+v"(negm));
    constexpr int EB = 4;                    // pairs per batch of the packed exponential (DL / 2 is a multiple of 8)
#pragma unroll
    for (int i = 0; i < DL / 2; i += EB) {   // this lane's exponentials
      f32x2 xb[EB];
#pragma unroll
      for (int b = 0; b < EB; ++b) xb[b] = v2[i + b] + negm;
      cv_expf_pk_nonpos<EB>(xb, ek);
#pragma unroll
      for (int b = 0; b < EB; ++b) v2[i + b] = xb[b];
    }
    float s = 0.f, t = 0.f;
#pragma unroll
    for (int k = 0; k < K; ++k) {
      if (k > 0) {                        // the chain continues in the pixel's next lane
        const float su = __shfl_up(s, 1), tu = __shfl_up(t, 1);
        if (sub == k) { s = su; t = tu; }
      }
      if (sub == k) {
#pragma unroll
        for (int i = 0; i < DL; ++i) {
          s += v2[i >> 1][i & 1];
          t = fmaf((float)(k * DL + i), v2[i >> 1][i & 1], t);
        }
      }
    }
    const long long p = p0 + lane / K;
    if (sub == K - 1 && p < npix) out_disp[p] = t / s;
  }
}

// one output pixel of the bilinear x`scale` upsampling (align_corners=False, PyTorch area_pixel_compute_source_index),
// times scale, zero outside (valid_h, valid_w): the arithmetic of oracle_disp_upsample
__device__ __forceinline__ float disp_upsample_value(const float* __restrict__ b, int Hf, int Wf, int scale, float inv, int Y,
                                                     int X, int valid_h, int valid_w) {
  if (Y >= valid_h || X >= valid_w) return 0.f;
  float sy = ((float)Y + 0.5f) * inv - 0.5f;
  float sx = ((float)X + 0.5f) * inv - 0.5f;
  sy = sy < 0.f ? 0.f : sy;
  sx = sx < 0.f ? 0.f : sx;
  const int y0 = min((int)sy, Hf - 1), x0 = min((int)sx, Wf - 1);
  const int y1 = min(y0 + 1, Hf - 1), x1 = min(x0 + 1, Wf - 1);
  const float ly = sy - (float)y0, lx = sx - (float)x0;
  const float hy = 1.0f - ly, hx = 1.0f - lx;
  const float v00 = b[(size_t)y0 * Wf + x0], v01 = b[(size_t)y0 * Wf + x1];
  const float v10 = b[(size_t)y1 * Wf + x0], v11 = b[(size_t)y1 * Wf + x1];
  return (hy * (hx * v00 + lx * v01) + ly * (hx * v10 + lx * v11)) * (float)scale;
}

// four consecutive pixels of a row per thread, three 16-byte stores (W a multiple of 4, 16-byte aligned output): the
// kernel is write-bound (90 MB of disp_postp per 8 pairs against 2 MB read)
__global__ __launch_bounds__(256) void disp_upsample_pack4_kernel(const float* __restrict__ lr, int N, int Hf, int Wf,
                                                                  int scale, int H, int W, int valid_h, int valid_w,
                                                                  float* __restrict__ out) {
  const int W4 = W >> 2;
  const long long total = (long long)N * H * W4;
  const float inv = 1.0f / (float)scale;
  const size_t plane = (size_t)H * W;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    const int X = 4 * (int)(idx % W4);
    const long long t = idx / W4;
    const int Y = (int)(t % H);
    const int n = (int)(t / H);
    const float* b = lr + (size_t)n * Hf * Wf;
    f32x4 v;
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = disp_upsample_value(b, Hf, Wf, scale, inv, Y, X + e, valid_h, valid_w);
    float* o = out + (size_t)n * 3 * plane + (size_t)Y * W + X;
    *reinterpret_cast<f32x4*>(o) = v;
    *reinterpret_cast<f32x4*>(o + plane) = v;
    *reinterpret_cast<f32x4*>(o + 2 * plane) = v;
  }
}

// bilinear x`scale` (align_corners=False, PyTorch area_pixel_compute_source_index), times scale,
// zero outside (valid_h, valid_w), replicated to 3 channels NCHW.
__global__ __launch_bounds__(256) void disp_upsample_pack_kernel(const float* __restrict__ lr, int N, int Hf, int Wf,
                                                                 int scale, int H, int W, int valid_h, int valid_w,
                                                                 float* __restrict__ out) {
  const long long total = (long long)N * H * W;
  const float inv = 1.0f / (float)scale;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    const int X = (int)(idx % W);
    const long long t = idx / W;
    const int Y = (int)(t % H);
    const int n = (int)(t / H);
    float v = 0.f;
    if (Y < valid_h && X < valid_w) {
      float sy = ((float)Y + 0.5f) * inv - 0.5f;
      float sx = ((float)X + 0.5f) * inv - 0.5f;
      sy = sy < 0.f ? 0.f : sy;
      sx = sx < 0.f ? 0.f : sx;
      const int y0 = min((int)sy, Hf - 1), x0 = min((int)sx, Wf - 1);
      const int y1 = min(y0 + 1, Hf - 1), x1 = min(x0 + 1, Wf - 1);
      const float ly = sy - (float)y0, lx = sx - (float)x0;
      const float hy = 1.0f - ly, hx = 1.0f - lx;
      const float* b = lr + (size_t)n * Hf * Wf;
      const float v00 = b[(size_t)y0 * Wf + x0], v01 = b[(size_t)y0 * Wf + x1];
      const float v10 = b[(size_t)y1 * Wf + x0], v11 = b[(size_t)y1 * Wf + x1];
      v = (hy * (hx * v00 + lx * v01) + ly * (hx * v10 + lx * v11)) * (float)scale;
    }
    const size_t plane = (size_t)H * W;
    float* o = out + (size_t)n * 3 * plane + (size_t)Y * W + X;
    o[0] = v;
    o[plane] = v;
    o[2 * plane] = v;
  }
}

// Bilinear x`scale` upsampling of an NHWC feature map (align_corners=False; the arithmetic of oracle_feat_upsample, bit
// for bit): the feature side of the stereo module's full-resolution mode.  A thread produces one float4 (4 channels of
// one output pixel): 4 16-byte gathers from the small low-resolution map (L2-resident), one 16-byte store.
__global__ __launch_bounds__(256) void feat_upsample_kernel(const float* __restrict__ in, int N, int Hf, int Wf, int C4,
                                                            int in_ld, int scale, float* __restrict__ out) {
  const int H = Hf * scale, W = Wf * scale;
  const long long total = (long long)N * H * W * C4;
  const float inv = 1.0f / (float)scale;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    const int q = (int)(idx % C4);
    long long t = idx / C4;
    const int X = (int)(t % W);
    t /= W;
    const int Y = (int)(t % H);
    const int n = (int)(t / H);
    float sy = ((float)Y + 0.5f) * inv - 0.5f;
    float sx = ((float)X + 0.5f) * inv - 0.5f;
    sy = sy < 0.f ? 0.f : sy;
    sx = sx < 0.f ? 0.f : sx;
    const int y0 = min((int)sy, Hf - 1), x0 = min((int)sx, Wf - 1);
    const int y1 = min(y0 + 1, Hf - 1), x1 = min(x0 + 1, Wf - 1);
    const float ly = sy - (float)y0, lx = sx - (float)x0;
    const float hy = 1.0f - ly, hx = 1.0f - lx;
    const float* b = in + (size_t)n * Hf * Wf * in_ld + 4 * q;
    const f32x4 v00 = *reinterpret_cast<const f32x4*>(b + ((size_t)y0 * Wf + x0) * in_ld);
    const f32x4 v01 = *reinterpret_cast<const f32x4*>(b + ((size_t)y0 * Wf + x1) * in_ld);
    const f32x4 v10 = *reinterpret_cast<const f32x4*>(b + ((size_t)y1 * Wf + x0) * in_ld);
    const f32x4 v11 = *reinterpret_cast<const f32x4*>(b + ((size_t)y1 * Wf + x1) * in_ld);
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = hy * (hx * v00[e] + lx * v01[e]) + ly * (hx * v10[e] + lx * v11[e]);
    *reinterpret_cast<f32x4*>(out + (size_t)idx * 4) = o;
  }
}

static int row_len(int n) { return ((n + 30) / 32) * 32 + 1; }  // >= n, = 1 mod 32

}  // namespace st

extern "C" int st_costvolume_softargmin(const float* featL_dev, const float* featR_dev, int N, int Hf, int Wf,
                                        int C, int feat_ld, int D, float temperature, float* out_cost_dev,
                                        float* out_disp_dev, st_stream_t stream_) {
  using namespace st;
  ST_REQUIRE(featL_dev && featR_dev && (out_cost_dev || out_disp_dev), "st_costvolume_softargmin: null pointer");
  ST_REQUIRE(N > 0 && Hf > 0 && Wf > 0 && C > 0 && C % 4 == 0 && feat_ld % 4 == 0 && feat_ld >= C,
             "st_costvolume_softargmin: C and feat_ld must be positive multiples of 4");
  ST_REQUIRE(D > 0 && D <= 4 * CV_MAXDG, "st_costvolume_softargmin: D must be in [1, %d]", 4 * CV_MAXDG);
  ST_REQUIRE(Hf <= 65535 && N <= 65535, "st_costvolume_softargmin: grid too large");
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  // register-tiled kernel: DG (disparities per lane, 4 lanes share a pixel quad) a multiple of 4, up to 32
  int dgt = round_up((D + 3) / 4, 4);
  // volumes wider than the 128 disparities one launch of the tiled kernel covers are MATERIALISED in equal slabs of
  // <= 128 (e.g. the full-resolution sizing D = 192 = 2 x 96); with a fused soft-argmin (out_disp) they take the
  // generic kernel below, which sees all of D at once
  int slabs = 1;
  if (dgt > 32 && out_cost_dev && !out_disp_dev && D % 16 == 0) {
    slabs = ceil_div(D, 128);
    while (D % (16 * slabs)) ++slabs;
    dgt = D / slabs / 4;
  }
  if (dgt <= 32 && (out_cost_dev == nullptr || (reinterpret_cast<uintptr_t>(out_cost_dev) & 15) == 0)) {
    // waves per workgroup (see the kernel): 4 (256-pixel segments) when at most one wave of a row's last segment idles.
    // Measured (profiles/r05_cv_nwv_ab.txt): W = 1280 (the full-resolution sizing) 330 -> 295 us; Wf = 320 (the bench
    // volume: 1.25 segments of 256, three of eight waves idle while their workgroup holds 73 KB) 102 -> 122 us.
    int nwv = (Wf > 128 && round_up(Wf, 256) - Wf <= 64) ? 4 : 2;
#ifdef ST_ABLATION
    if (const char* e = getenv("ST_CV_NWV")) nwv = atoi(e) == 4 ? 4 : 2;   // A/B (tools/cv_bench.py)
#endif
    const int TX = 64 * nwv;
    const int rowLt = cvt_row(TX), rowRt = cvt_row(TX + 4 * dgt);
    const size_t ldst = ((size_t)2 * CVT_CC * (rowLt + rowRt) + rowRt) * sizeof(float);   // + the prefetch slack row
    const int gxt = (Wf + TX - 1) / TX;
    const long long totalt = (long long)gxt * Hf * N;
    ST_REQUIRE(totalt + 8 < (1ll << 31), "st_costvolume_softargmin: grid too large");
    const int per_xcd = (int)((totalt + 7) / 8);
    const dim3 gridt((unsigned)(8 * per_xcd)), blockt((unsigned)(64 * nwv));
    // The product launches form 3 ALWAYS (packed FMAs on aligned R pairs x explicit L pairs: no op_sel operand anywhere -
    // tests/test_cpu_oracle_and_abi.py checks the built code object).  Form 1 goes wrong whenever a bf16 MFMA of any
    // kernel on the chip executes beside it (the op_sel bit of a v_pk_fma_f32 source is dropped for single passes:
    // tools/micro/pkfma_corun.hip, profiles/r05_pkfma_corun.txt); form 2 costs more vector instructions than the scalar
    // form 0; form 3 is bit-identical to all of them, as fast as form 1 in the pipeline (92 us against 98.5 us for
    // form 0, profiles/r05_cv_form_inflight_ab.txt) and clean beside the aggressors that break form 1.  No process
    // state, no launch-order dependence; forms 0 - 2 exist in the tools build (2-wave workgroups) for the reproducer.
#ifdef ST_ABLATION
    int fma_mode = 3;
    if (const char* e = getenv("ST_CV_FMA")) fma_mode = atoi(e);   // tools/cv_stress.py
    if (fma_mode != 3 && nwv != 2) return set_error(ST_ERR_INVALID, "ST_CV_FMA forms 0-2 exist for ST_CV_NWV=2 only");
#define ST_CVT_LAUNCH(DGV)                                                                                    \
  do {                                                                                                         \
    if (fma_mode == 1) ST_CVT_LAUNCH_I(DGV, 1, 2);                                                             \
    else if (fma_mode == 2) ST_CVT_LAUNCH_I(DGV, 2, 2);                                                        \
    else if (fma_mode == 0) ST_CVT_LAUNCH_I(DGV, 0, 2);                                                        \
    else if (nwv == 4) ST_CVT_LAUNCH_I(DGV, 3, 4);                                                             \
    else ST_CVT_LAUNCH_I(DGV, 3, 2);                                                                           \
  } while (0)
#else
#define ST_CVT_LAUNCH(DGV)                                                                                    \
  do {                                                                                                         \
    if (nwv == 4) ST_CVT_LAUNCH_I(DGV, 3, 4); else ST_CVT_LAUNCH_I(DGV, 3, 2);                                 \
  } while (0)
#endif
#define ST_CVT_LAUNCH_I(DGV, PKV, NWVV)                                                                       \
  do {                                                                                                         \
    auto kern = costvolume_tiled_kernel<DGV, PKV, NWVV>;                                                       \
    static int lds_set = 0;                                                                                    \
    ST_ENSURE_DYNAMIC_LDS(kern, ldst, lds_set);                                                                \
    for (int sl = 0; sl < slabs; ++sl)                                                                         \
      hipLaunchKernelGGL(kern, gridt, blockt, ldst, stream, featL_dev, featR_dev, Hf, Wf, C, feat_ld, D,       \
                         temperature, rowLt, rowRt, out_cost_dev, out_disp_dev, sl * 4 * DGV, gxt,             \
                         (int)totalt, per_xcd);                                                                \
  } while (0)
    switch (dgt) {
      case 4: ST_CVT_LAUNCH(4); break;
      case 8: ST_CVT_LAUNCH(8); break;
      case 12: ST_CVT_LAUNCH(12); break;
      case 16: ST_CVT_LAUNCH(16); break;
      case 20: ST_CVT_LAUNCH(20); break;
      case 24: ST_CVT_LAUNCH(24); break;
      case 28: ST_CVT_LAUNCH(28); break;
      default: ST_CVT_LAUNCH(32); break;
    }
#undef ST_CVT_LAUNCH
#undef ST_CVT_LAUNCH_I
    ST_CHECK_HIP(hipGetLastError());
    return ST_OK;
  }
  const int rowL = row_len(CV_TX), rowR = row_len(CV_TX + D - 1);
  const size_t lds = (size_t)C * (rowL + rowR) * sizeof(float);
  ST_REQUIRE(lds <= 160 * 1024, "st_costvolume_softargmin: C=%d, D=%d needs %zu B of LDS (> 160 KiB)", C, D, lds);
  const dim3 grid((Wf + CV_TX - 1) / CV_TX, Hf, N), block(256);
  const int DG = (D + 3) / 4;
#define ST_CV_LAUNCH(DGV)                                                                                      \
  do {                                                                                                         \
    auto kern = costvolume_kernel<DGV>;                                                                        \
    static int lds_set = 0;                                                                                    \
    ST_ENSURE_DYNAMIC_LDS(kern, lds, lds_set);                                                                 \
    hipLaunchKernelGGL(kern, grid, block, lds, stream, featL_dev, featR_dev, Hf, Wf, C, feat_ld, D,            \
                       temperature, rowL, rowR, out_cost_dev, out_disp_dev);                                   \
  } while (0)
  if (DG <= 4) ST_CV_LAUNCH(4);
  else if (DG <= 8) ST_CV_LAUNCH(8);
  else if (DG <= 12) ST_CV_LAUNCH(12);
  else if (DG <= 16) ST_CV_LAUNCH(16);
  else if (DG <= 24) ST_CV_LAUNCH(24);
  else if (DG <= 32) ST_CV_LAUNCH(32);
  else if (DG <= 48) ST_CV_LAUNCH(48);
  else ST_CV_LAUNCH(64);
#undef ST_CV_LAUNCH
  ST_CHECK_HIP(hipGetLastError());
  return ST_OK;
}

extern "C" int st_softargmin(const float* cost_dev, int N, int Hf, int Wf, int D, float temperature,
                             float* out_disp_dev, st_stream_t stream_) {
  using namespace st;
  ST_REQUIRE(cost_dev && out_disp_dev && N > 0 && Hf > 0 && Wf > 0 && D > 0, "st_softargmin: bad argument");
  const long long npix = (long long)N * Hf * Wf;
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  if (D % 16 == 0 && D <= 192 && (reinterpret_cast<uintptr_t>(cost_dev) & 15) == 0) {
    // rows in registers (softargmin_reg_kernel): one lane per pixel up to 96 levels and for odd D / 16, two lanes beyond
    const int ppw = (D > 96 && (D / 16) % 2 == 0) ? 32 : 64;         // pixels per wave (K = 2 lanes per pixel, or 1)
    const long long nwave = (npix + ppw - 1) / ppw;
    const unsigned blocks = (unsigned)std::min<long long>((nwave + 3) / 4, 256 * 16);
#define ST_SA_LAUNCH(NCHV, KV)                                                                              \
  hipLaunchKernelGGL((softargmin_reg_kernel<NCHV, KV>), dim3(blocks), dim3(256), 0, stream, cost_dev, npix,  \
                     temperature, out_disp_dev)
    switch (D / 16) {
      case 1: ST_SA_LAUNCH(1, 1); break;
      case 2: ST_SA_LAUNCH(2, 1); break;
      case 3: ST_SA_LAUNCH(3, 1); break;
      case 4: ST_SA_LAUNCH(4, 1); break;
      case 5: ST_SA_LAUNCH(5, 1); break;
      case 6: ST_SA_LAUNCH(6, 1); break;
      case 7: ST_SA_LAUNCH(7, 1); break;
      case 8: ST_SA_LAUNCH(8, 2); break;
      case 9: ST_SA_LAUNCH(9, 1); break;
      case 10: ST_SA_LAUNCH(10, 2); break;
      case 11: ST_SA_LAUNCH(11, 1); break;
      default: ST_SA_LAUNCH(12, 2); break;
    }
#undef ST_SA_LAUNCH
    ST_CHECK_HIP(hipGetLastError());
    return ST_OK;
  }
  if (D % 4 == 0 && (reinterpret_cast<uintptr_t>(cost_dev) & 15) == 0) {
    const int ldsw = D + (((D >> 2) & 1) ? 8 : 4);
    int pb = 256;
    while (pb >= 64 && (size_t)pb * ldsw * 4 > 60 * 1024) pb >>= 1;
    const long long nblk = (npix + pb - 1) / pb;
    if (pb >= 64 && nblk < (1ll << 31)) {
      hipLaunchKernelGGL(softargmin_lds_kernel, dim3((unsigned)nblk), dim3(pb), (size_t)pb * ldsw * 4, stream,
                         cost_dev, npix, D, ldsw, temperature, out_disp_dev);
      ST_CHECK_HIP(hipGetLastError());
      return ST_OK;
    }
  }
  const int blocks = (int)std::min<long long>((npix + 255) / 256, 256 * 8);
  hipLaunchKernelGGL(softargmin_kernel, dim3(blocks), dim3(256), 0, stream, cost_dev,
                     npix, D, temperature, out_disp_dev);
  ST_CHECK_HIP(hipGetLastError());
  return ST_OK;
}

extern "C" int st_disp_upsample_pack(const float* disp_lr_dev, int N, int Hf, int Wf, int scale, int H, int W,
                                     int valid_h, int valid_w, float* disp_postp_dev, st_stream_t stream_) {
  using namespace st;
  ST_REQUIRE(disp_lr_dev && disp_postp_dev, "st_disp_upsample_pack: null pointer");
  ST_REQUIRE(N > 0 && Hf > 0 && Wf > 0 && scale > 0 && H == Hf * scale && W == Wf * scale,
             "st_disp_upsample_pack: output must be exactly scale x the low-res map");
  ST_REQUIRE(valid_h >= 0 && valid_h <= H && valid_w >= 0 && valid_w <= W, "st_disp_upsample_pack: bad valid region");
  if (W % 4 == 0 && (reinterpret_cast<uintptr_t>(disp_postp_dev) & 15) == 0) {
    const long long total4 = (long long)N * H * (W / 4);
    const int blocks4 = (int)std::min<long long>((total4 + 255) / 256, 256 * 16);
    hipLaunchKernelGGL(disp_upsample_pack4_kernel, dim3(blocks4), dim3(256), 0, static_cast<hipStream_t>(stream_),
                       disp_lr_dev, N, Hf, Wf, scale, H, W, valid_h, valid_w, disp_postp_dev);
    ST_CHECK_HIP(hipGetLastError());
    return ST_OK;
  }
  const long long total = (long long)N * H * W;
  const int blocks = (int)std::min<long long>((total + 255) / 256, 256 * 16);
  hipLaunchKernelGGL(disp_upsample_pack_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream_),
                     disp_lr_dev, N, Hf, Wf, scale, H, W, valid_h, valid_w, disp_postp_dev);
  ST_CHECK_HIP(hipGetLastError());
  return ST_OK;
}

extern "C" int st_feat_upsample(const float* feat_dev, int N, int Hf, int Wf, int C, int feat_ld, int scale,
                                float* out_dev, st_stream_t stream_) {
  using namespace st;
  ST_REQUIRE(feat_dev && out_dev, "st_feat_upsample: null pointer");
  ST_REQUIRE(N > 0 && Hf > 0 && Wf > 0 && scale > 0 && C > 0 && C % 4 == 0 && feat_ld % 4 == 0 && feat_ld >= C,
             "st_feat_upsample: C and feat_ld must be positive multiples of 4");
  ST_REQUIRE(((reinterpret_cast<uintptr_t>(feat_dev) | reinterpret_cast<uintptr_t>(out_dev)) & 15) == 0,
             "st_feat_upsample: pointers must be 16-byte aligned");
  const long long total = (long long)N * Hf * scale * Wf * scale * (C / 4);
  const int blocks = (int)std::min<long long>((total + 255) / 256, 256 * 32);
  hipLaunchKernelGGL(feat_upsample_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream_), feat_dev, N,
                     Hf, Wf, C / 4, feat_ld, scale, out_dev);
  ST_CHECK_HIP(hipGetLastError());
  return ST_OK;
}
