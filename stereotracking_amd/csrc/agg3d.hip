// 3-D aggregation of the stereo cost volume (north_star: "its 3D/2D aggregation"): single-channel 3x3x3 convolutions
// over (d, y, x) of the materialised volume [N][Hf][Wf][D], zero padded, optional SiLU.  The reference has no such
// function (its disparity is an offline product, reproducibility.md:166-194); the specification is
// oracle/st_oracle.c::oracle_agg3d and this kernel is BIT-EXACT against it (same fmaf order: rows j, columns k,
// disparity taps i; padded taps contribute fmaf(w, 0, acc); SiLU through the oracle's exp polynomial).
//
// Three kernels, one arithmetic:
//   vol_agg3d_kernel   volume -> volume layer (st_volume_agg3d): a workgroup owns a column strip of TW pixels x all D
//                      disparities and walks down a band of rows; the two rows above the current one live in REGISTERS
//                      (as level pairs), the current one passes through a single LDS row; the stencil's outer taps run on
//                      packed-fp32 FMAs with default operand selection.  A volume element is fetched (TW + 2) / TW x
//                      (RY + 2) / RY times per layer.
//   cv_agg3d_kernel    the same with the row computed from the feature maps instead of read (st_costvolume_agg3d: cost
//                      volume + first layer, the volume between them never reaches memory).
//   agg3d_kernel       the first streaming form (a ring of four rows in LDS, every tap an LDS read): tools build only
//                      (ST_A3_RING), the yardstick the register form was measured against (0.34 -> 0.47 of 8 TB/s at
//                      D = 192).
#include <algorithm>
#include <cstdlib>
#include <type_traits>

#include "st_common.h"

namespace st {
namespace {

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float a3_expf(float x) {  // same polynomial as decode_nms.hip / costvolume.hip / oracle
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

struct Agg3dArgs {
  const float* in;
  float* out;
  int N, Hf, Wf, D;
  int RY;        // rows per band
  float w[27];   // [i = kD][j = kH][k = kW]
  float bias;
  int act;
};

// (tools build) streaming stencil with a ring of four pixel rows ((TW + 2) x D floats each) in LDS - rows y-1, y, y+1 feed
// output row y while row y+2 travels from memory through registers into the fourth slot (one barrier per row); a thread
// produces 4 consecutive disparities of one pixel from 9 aligned 16-byte LDS reads + the two neighbours across the quad
// borders.  TW: pixels per strip; NST: float4 staging registers per thread = ceil((TW + 2) * (D / 4) / 256)
template <int TW, int NST>
__global__ __launch_bounds__(256) void agg3d_kernel(const Agg3dArgs a) {
  extern __shared__ float4 a3_smem4[];
  float* lds = reinterpret_cast<float*>(a3_smem4);
  const int D = a.D, DQ = D >> 2;
  constexpr int TC = TW + 2;
  const int x0 = blockIdx.x * TW, y0 = blockIdx.y * a.RY, n = blockIdx.z;
  const int y1 = min(y0 + a.RY, a.Hf);
  const int tid = threadIdx.x;
  const int rowf = TC * D;          // floats per staged pixel row
  const int nrow4 = TC * DQ;        // float4 per staged pixel row
  f32x4 st[NST];
  // one pixel row gy (columns x0-1 .. x0+TW, zero outside the image) from memory into registers / registers into slot
  auto load_row_into = [&](int gy, f32x4 (&r)[NST]) {
#pragma unroll
    for (int i = 0; i < NST; ++i) {
      const int e = tid + 256 * i;
      const int q = e % DQ, c = e / DQ;
      const int gx = x0 + c - 1;
      r[i] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (e < nrow4 && gy >= 0 && gy < a.Hf && gx >= 0 && gx < a.Wf)
        r[i] = *reinterpret_cast<const f32x4*>(a.in + (((size_t)n * a.Hf + gy) * a.Wf + gx) * D + 4 * q);
    }
  };
  auto store_row_from = [&](int gy, const f32x4 (&r)[NST]) {
    float* dst = lds + (size_t)((gy + 4) & 3) * rowf;
#pragma unroll
    for (int i = 0; i < NST; ++i) {
      const int e = tid + 256 * i;
      if (e < nrow4) *reinterpret_cast<f32x4*>(dst + 4 * e) = r[i];
    }
  };
  auto load_row = [&](int gy) { load_row_into(gy, st); };
  auto store_row = [&](int gy) { store_row_from(gy, st); };
  {   // prologue: the three rows of the first output row with all their loads in flight at once (one memory round trip)
    f32x4 p0[NST], p1[NST];
    load_row_into(y0 - 1, p0);
    load_row_into(y0, p1);
    load_row(y0 + 1);
    store_row_from(y0 - 1, p0);
    store_row_from(y0, p1);
    store_row(y0 + 1);
  }
  __syncthreads();
  const int nitem = TW * DQ;
  for (int y = y0; y < y1; ++y) {
    const bool more = y + 1 < y1;
    if (more) load_row(y + 2);      // in flight while this row is computed
    for (int it = tid; it < nitem; it += 256) {
      const int q = it % DQ, px = it / DQ;
      if (x0 + px >= a.Wf) continue;
      float acc[4] = {a.bias, a.bias, a.bias, a.bias};
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const float* row = lds + (size_t)((y + j - 1 + 4) & 3) * rowf;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          const float* p = row + (size_t)(px + k) * D + 4 * q;
          const f32x4 v = *reinterpret_cast<const f32x4*>(p);
          const float vm = q > 0 ? p[-1] : 0.0f;
          const float vp = q < DQ - 1 ? p[4] : 0.0f;
          const float vals[6] = {vm, v[0], v[1], v[2], v[3], vp};
#pragma unroll
          for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int i = 0; i < 3; ++i) acc[e] = fmaf(a.w[(i * 3 + j) * 3 + k], vals[e + i], acc[e]);
        }
      }
      f32x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = a.act ? acc[e] / (1.0f + a3_expf(-acc[e])) : acc[e];
      *reinterpret_cast<f32x4*>(a.out + (((size_t)n * a.Hf + y) * a.Wf + x0 + px) * D + 4 * q) = o;
    }
    // slot (y + 2) & 3 held row y - 2: last read while row y - 1 was computed, i.e. before the previous barrier
    if (more) store_row(y + 2);
    __syncthreads();
  }
}


// ---- fused first layer: cost volume from the feature maps -> 3x3x3 layer ------------------------------------------
// st_costvolume_agg3d: out = agg3d(costvolume(featL, featR)) without the volume in between ever reaching memory (at the
// full-resolution sizing the materialised form writes 6.3 GB and reads them back, per 8 pairs).  Arithmetic per cell
// is that of costvolume.hip followed by the kernel above, fmaf for fmaf (oracle_costvolume, then oracle_agg3d).
//
// A workgroup (192 threads) owns a strip of TW pixel columns x all D levels and walks down a band of rows.  Per row r:
//   loads     feature row r+1 is requested from memory first thing (buffer loads with a per-ROW descriptor: uniform base,
//             one 32-bit offset per thread, pixels outside the image by the range check);
//   produce   cost row r for columns x0-1 .. x0+TW from the staged feature row r (LDS, channel-major so that four
//             neighbouring pixels of one channel are one 16-byte read): a thread computes a 4 pixel x 4 level tile (7
//             right-image pixels feed its 16 cells; channel c+1's operands are read while channel c's FMAs run), 2 * D/4
//             threads add the two halo columns; the row goes to ONE LDS row, stored shifted by a level with zero pads;
//   (barrier) every thread takes the 6 columns x 6 levels it needs of the new cost row into registers, as level PAIRS;
//   stencil   output row r-1 from rows r-2, r-1 (kept in registers from the two iterations before) and r: 4 pixels x 4
//             levels per thread = eight accumulator pairs side by side; disparity taps 0 and 2 as packed FMAs on aligned
//             pairs (default operand selection, (w, w) in scalar register pairs), tap 1 scalar; 16-byte stores;
//   stage     feature row r+1 goes from registers into the stage (the wait in front allows the four stores to be in
//             flight: `vmcnt` counts loads and stores together);
//   (barrier)
// The three register rows rotate by name (the loop body is instantiated three times), not by copies; the first two
// iterations of a band (rows y0-1, y0) are instantiated without the output part.
struct CvAggArgs {
  const float* fl;
  const float* fr;
  float* out;
  int N, H, W, ld, D, RY;
  int strips, bands, order;   // 1-D grid of N * bands * strips workgroups; order 1: XCD-contiguous (below)
  float w[27];
  float bias;
  int act;
  float temperature;          // SA form (st_costvolume_agg3d_softargmin): the soft-argmin of every aggregated row is taken
  float* disp;                // in the kernel, [N][H][W] px; `out` is not written
};


// One output row of the 3x3x3 stencil from three register rows (6 columns x 3 level pairs each: levels 4q-1 .. 4q+4 of the
// columns px-1 .. px+4 of a thread's four pixels): outputs (4q, 4q+1) and (4q+2, 4q+3) of a pixel as two accumulator pairs;
// per (row j, column k): tap i = 0 and tap i = 2 as packed FMAs on the aligned level pairs (default operand selection; wp =
// the (w, w) pairs of those taps), tap i = 1 (odd pairs) as four scalar FMAs - per output the oracle's order j, k, i.  The
// four pixels advance together: eight independent chains for the issue slots.  16-byte stores through `orsrc` (pixels
// beyond the descriptor are dropped).  NOSTORE: tools only (the results stay alive, nothing is stored).
// LDSOUT: the four 16-byte results go to `ldsout + pi * ldspx` (an LDS row of the workgroup) instead of memory.
template <bool NOSTORE, bool LDSOUT = false>
__device__ __forceinline__ void a3_stencil_row(const f32x2 (&ra)[6][3], const f32x2 (&rb)[6][3], const f32x2 (&rc)[6][3],
                                               const f32x2 (&wp)[2][3][3], const float (&w)[27], float bias, int act,
                                               __amdgpu_buffer_rsrc_t orsrc, int voff0, int pxbytes,
                                               float* ldsout = nullptr, int ldspx = 0) {
  f32x2 p0[4], p1[4];
#pragma unroll
  for (int pi = 0; pi < 4; ++pi) p0[pi] = p1[pi] = f32x2{bias, bias};
#pragma unroll
  for (int j = 0; j < 3; ++j) {
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const float w1 = w[(1 * 3 + j) * 3 + k];
#pragma unroll
      for (int pi = 0; pi < 4; ++pi) {
        const f32x2 (&v)[3] = j == 0 ? ra[pi + k] : (j == 1 ? rb[pi + k] : rc[pi + k]);
        p0[pi] = __builtin_elementwise_fma(wp[0][j][k], v[0], p0[pi]);
        p1[pi] = __builtin_elementwise_fma(wp[0][j][k], v[1], p1[pi]);
      }
      __builtin_amdgcn_sched_barrier(0);     // keep the eight chains side by side (the scheduler would run them one by one)
#pragma unroll
      for (int pi = 0; pi < 4; ++pi) {
        const f32x2 (&v)[3] = j == 0 ? ra[pi + k] : (j == 1 ? rb[pi + k] : rc[pi + k]);
        p0[pi][0] = fmaf(w1, v[0][1], p0[pi][0]);
        p0[pi][1] = fmaf(w1, v[1][0], p0[pi][1]);
        p1[pi][0] = fmaf(w1, v[1][1], p1[pi][0]);
        p1[pi][1] = fmaf(w1, v[2][0], p1[pi][1]);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int pi = 0; pi < 4; ++pi) {
        const f32x2 (&v)[3] = j == 0 ? ra[pi + k] : (j == 1 ? rb[pi + k] : rc[pi + k]);
        p0[pi] = __builtin_elementwise_fma(wp[1][j][k], v[1], p0[pi]);
        p1[pi] = __builtin_elementwise_fma(wp[1][j][k], v[2], p1[pi]);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  // all eight chains end HERE: without this use the optimiser sinks the chains of pixels 1..3 behind the branches of
  // pixel 0's activation / store code and they run one after the other again
#pragma unroll
  for (int pi = 0; pi < 4; ++pi) asm volatile("" : "+v"(p0[pi]), "+v"(p1[pi]));
#pragma unroll
  for (int pi = 0; pi < 4; ++pi) {
    const float acc[4] = {p0[pi][0], p0[pi][1], p1[pi][0], p1[pi][1]};
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = act ? acc[e] / (1.0f + a3_expf(-acc[e])) : acc[e];
    if (NOSTORE) asm volatile("" ::"v"(o));
    else if (LDSOUT) *reinterpret_cast<f32x4*>(ldsout + pi * ldspx) = o;
    else __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), orsrc, voff0 + pi * pxbytes, 0, 0);
  }
}

// the (w, w) pairs of the taps i = 0 and i = 2 as SCALAR register pairs of their own: the empty asm makes each an opaque
// 64-bit value, so the broadcast cannot be folded into an op_sel modifier of the packed FMA
__device__ __forceinline__ void a3_weight_pairs(const float (&w)[27], f32x2 (&wp)[2][3][3]) {
#pragma unroll
  for (int j = 0; j < 3; ++j)
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      wp[0][j][k] = f32x2{w[(0 * 3 + j) * 3 + k], w[(0 * 3 + j) * 3 + k]};
      wp[1][j][k] = f32x2{w[(2 * 3 + j) * 3 + k], w[(2 * 3 + j) * 3 + k]};
      asm volatile("" : "+s"(wp[0][j][k]));
      asm volatile("" : "+s"(wp[1][j][k]));
    }
}

// DFIX: D == DMAX, known at compile time (every offset an immediate)
// SA (round 6, st_costvolume_agg3d_softargmin): the aggregated row does not leave the chip - the stencil's results go to an
// LDS row [TW][D + 4], and behind the iteration's barrier the workgroup takes its soft-argmin: thread (pixel tid % TW, part
// tid / TW) scales its 16 levels by the temperature and publishes their maximum, (barrier) takes the pixel's maximum and
// replaces the levels by exp(T c - m) (the oracle's polynomial), (barrier) and one lane per pixel runs the oracle's two
// running sums over the D exponentials IN ORDER (s += e; t = fmaf(d, e, t)) and stores t / s.  Bit-equal to
// st_costvolume_agg3d followed by st_softargmin; the 4 D bytes per pixel of volume are neither written nor read back.
template <int TW, int C, int DMAX, bool DFIX, int MODE = 0, bool SA = false>   // MODE: timing-only ablations of the tools build (0 in the product)
__global__ __launch_bounds__(192) __attribute__((amdgpu_waves_per_eu(3))) void cv_agg3d_kernel(const CvAggArgs a) {
  constexpr int NT = 192, CQ = C / 4, NG = TW / 4, FLW = TW + 8;
  extern __shared__ float4 a3_smem4[];
  float* lds = reinterpret_cast<float*>(a3_smem4);
  const int D = DFIX ? DMAX : a.D, DQ = D >> 2, FRW = D + TW + 4;
  // cost row: [TW + 2][DS], level d of a column at float d + 1, floats 0 and D + 1 stay zero (levels -1 and D of the zero
  // padded volume): the six levels 4q-1 .. 4q+4 a thread needs are one aligned 16-byte + one aligned 8-byte read, and
  // land in registers as the pairs (4q-1, 4q), (4q+1, 4q+2), (4q+3, 4q+4) the packed FMAs below take as they are
  const int DS = D + 4;
  float* costrow = lds;
  float* frs = costrow + (TW + 2) * DS;          // [C][FRW]: right-image pixels x0 - D .. x0 + TW + 3
  float* fls = frs + C * FRW;                    // [C][FLW]: left-image pixels x0 - 4 .. x0 + TW + 3
  const int DSA = D + 4;                         // SA: aggregated row [TW][DSA] (16-byte rows, pixel stride off the banks)
  float* aggrow = fls + C * FLW;
  float* pmax = aggrow + TW * DSA;               // SA: [D / 16][TW] partial maxima
  // Workgroup -> (pair, band, strip).  Neighbouring strips share 196 of the 212 right-image pixels they stage per row;
  // workgroups are dealt to the 8 XCDs (own L2 each) round-robin, so in launch order every XCD ends up fetching the whole
  // feature row.  Order 1 gives each XCD a CONTIGUOUS range of the (pair, band, strip) sequence: neighbours share an L2.
  int wg = blockIdx.x;
  if (a.order == 1) {
    const int total = gridDim.x, per = total >> 3;
    if (wg < (per << 3)) wg = (wg & 7) * per + (wg >> 3);
  }
  const int strip = wg % a.strips, band = (wg / a.strips) % a.bands, n = wg / (a.strips * a.bands);
  const int x0 = strip * TW, y0 = band * a.RY;
  const int y1 = min(y0 + a.RY, a.H);
  const int tid = threadIdx.x;
  const int q = tid % DQ, g = tid / DQ;
  const bool active = (DFIX && TW * DMAX / 16 == NT) ? true : g < NG;
  constexpr float invC = 1.0f / (float)C;        // C is a power of two: the product equals the oracle's quotient

  // staging plan of a feature row (per thread, fixed for the whole band): NFR float4 of the right image, one of the left
  constexpr int NFR = ((DMAX + TW + 4) * CQ + NT - 1) / NT;
  constexpr int NFL = (FLW * CQ + NT - 1) / NT;
  // float4 i of a thread: channel quad tid % CQ of stage pixel tid / CQ + i * (NT / CQ); a pixel left of the image gives a
  // negative (= huge unsigned) byte offset, one right of it an offset beyond the row: both outside the descriptor
  static_assert(NT % CQ == 0, "staging plan");
  constexpr int PSTEP = NT / CQ;
  const int scq = tid % CQ, spx = tid / CQ;
  const int fr_off0 = ((x0 - D + spx) * a.ld + 4 * scq) * 4, fl_off0 = ((x0 - 4 + spx) * a.ld + 4 * scq) * 4;   // bytes
  const int off_step = PSTEP * a.ld * 4;
  const int fr_dst0 = (4 * scq) * FRW + spx, fl_dst0 = (4 * scq) * FLW + spx;                                    // floats
  f32x4 stg[NFR], stl[NFL];
  // Feature rows travel through buffer loads: the descriptor is the ROW (uniform base, num_records = its bytes), a
  // thread's part is a fixed 32-bit byte offset; pixels left or right of the image carry an offset beyond the row and
  // read as zero by the descriptor's range check.
  const int rowbytes = a.W * a.ld * 4;
  auto load_feat = [&](int gy) {                 // feature row gy -> registers (zero outside the image)
    if (gy < 0 || gy >= a.H) {                   // uniform
#pragma unroll
      for (int i = 0; i < NFR; ++i) stg[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int i = 0; i < NFL; ++i) stl[i] = f32x4{0.f, 0.f, 0.f, 0.f};
      return;
    }
    const size_t rowbase = ((size_t)n * a.H + gy) * a.W * a.ld;
    const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.fr + rowbase), 0, rowbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rl = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.fl + rowbase), 0, rowbytes, 0x00020000);
#pragma unroll
    for (int i = 0; i < NFR; ++i) stg[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rr, fr_off0 + i * off_step, 0, 0));
#pragma unroll
    for (int i = 0; i < NFL; ++i) stl[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rl, fl_off0 + i * off_step, 0, 0));
  };
  auto store_feat = [&]() {                      // registers -> channel-major stage
#pragma unroll
    for (int i = 0; i < NFR; ++i)
      if (spx + i * PSTEP < FRW) {
#pragma unroll
        for (int t = 0; t < 4; ++t) frs[fr_dst0 + i * PSTEP + t * FRW] = stg[i][t];
      }
#pragma unroll
    for (int i = 0; i < NFL; ++i)
      if (spx + i * PSTEP < FLW) {
#pragma unroll
        for (int t = 0; t < 4; ++t) fls[fl_dst0 + i * PSTEP + t * FLW] = stl[i][t];
      }
  };
  auto produce = [&]() {                         // staged feature row -> cost row in LDS
    if (active) {
      float acc[4][4];
#pragma unroll
      for (int pi = 0; pi < 4; ++pi)
#pragma unroll
        for (int dj = 0; dj < 4; ++dj) acc[pi][dj] = 0.0f;
      // operands of channel c+1 are read while channel c's FMAs run (last iteration re-reads channel C-1: no branch)
      const float* lp = fls + 4 * g + 4;
      const float* rp = frs + (D + 4 * g - 4 * q - 4);
      f32x4 L = *reinterpret_cast<const f32x4*>(lp);
      f32x4 ra = *reinterpret_cast<const f32x4*>(rp), rb = *reinterpret_cast<const f32x4*>(rp + 4);
#pragma unroll 1
      for (int c = 0; c < ((MODE & 2) ? 0 : C); ++c) {
        const int cn = min(c + 1, C - 1);
        const f32x4 Ln = *reinterpret_cast<const f32x4*>(lp + cn * FLW);
        const f32x4 ran = *reinterpret_cast<const f32x4*>(rp + cn * FRW), rbn = *reinterpret_cast<const f32x4*>(rp + cn * FRW + 4);
        const float R[8] = {ra[0], ra[1], ra[2], ra[3], rb[0], rb[1], rb[2], rb[3]};
#pragma unroll
        for (int pi = 0; pi < 4; ++pi)
#pragma unroll
          for (int dj = 0; dj < 4; ++dj) acc[pi][dj] = fmaf(L[pi], R[4 + pi - dj], acc[pi][dj]);
        L = Ln; ra = ran; rb = rbn;
      }
#pragma unroll
      for (int pi = 0; pi < 4; ++pi) {
        float* dst = costrow + (1 + 4 * g + pi) * DS + 4 * q + 1;
        dst[0] = acc[pi][0] * invC;
        *reinterpret_cast<f32x2*>(dst + 1) = f32x2{acc[pi][1] * invC, acc[pi][2] * invC};
        dst[3] = acc[pi][3] * invC;
      }
    }
    if (tid < 2 * DQ) {                          // halo columns x0 - 1 (side 0) and x0 + TW (side 1)
      const int side = tid / DQ, hq = tid % DQ;
      const int pxo = side ? TW + 4 : 3;         // in fls
      // right-image pixels px - 4 hq - 3 .. px - 4 hq; relative to x0 - D: aligned (side 0) or one past a boundary
      const int roff = side ? D + TW - 4 * hq - 4 : D - 4 * hq - 4, sh = side ? 1 : 0;
      float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
      for (int c = 0; c < C; ++c) {
        const float l = fls[c * FLW + pxo];
        const float* rp = frs + c * FRW + roff;
        const f32x4 ra = *reinterpret_cast<const f32x4*>(rp), rb = *reinterpret_cast<const f32x4*>(rp + 4);
        const float R[4] = {sh ? ra[1] : ra[0], sh ? ra[2] : ra[1], sh ? ra[3] : ra[2], sh ? rb[0] : ra[3]};
#pragma unroll
        for (int dj = 0; dj < 4; ++dj) acc[dj] = fmaf(l, R[3 - dj], acc[dj]);
      }
      float* dst = costrow + (side ? TW + 1 : 0) * DS + 4 * hq + 1;
      dst[0] = acc[0] * invC;
      *reinterpret_cast<f32x2*>(dst + 1) = f32x2{acc[1] * invC, acc[2] * invC};
      dst[3] = acc[3] * invC;
    }
  };
  f32x2 wp[2][3][3];
  a3_weight_pairs(a.w, wp);
  // one iteration; (ra, rb) hold cost rows r-2, r-1, rc receives row r.  A register row: 6 columns x 3 level pairs
  auto step = [&](auto out_tag, int r, f32x2 (&ra)[6][3], f32x2 (&rb)[6][3], f32x2 (&rc)[6][3]) {
    constexpr bool OUT = decltype(out_tag)::value;
    // feature row r+1 is requested first thing (into registers; the stage keeps row r for `produce`), travels during the
    // whole iteration and is written to the stage at its end: the wait in front of that write allows the four output
    // stores issued after the loads to be still in flight (vmcnt counts loads and stores together)
    if (r + 1 <= y1 && !((MODE & 8) && OUT)) load_feat(r + 1);
    produce();
    __syncthreads();
    if (active) {
#pragma unroll
      for (int col = 0; col < 6; ++col) {
        const float* p = costrow + (4 * g + col) * DS + 4 * q;
        const f32x4 v = *reinterpret_cast<const f32x4*>(p);
        rc[col][0] = f32x2{v[0], v[1]};
        rc[col][1] = f32x2{v[2], v[3]};
        rc[col][2] = *reinterpret_cast<const f32x2*>(p + 4);
      }
    }
    const int y = r - 1;
    if (OUT && active) {
      // the strip's output row as a buffer: pixels beyond the image fall outside the descriptor and are dropped
      const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(
          a.out + (((size_t)n * a.H + y) * a.W + x0) * D, 0, min(TW, a.W - x0) * D * 4, 0x00020000);
      a3_stencil_row<(MODE & 1) != 0, SA>(ra, rb, rc, wp, a.w, a.bias, a.act, orsrc, ((4 * g) * D + 4 * q) * 4, D * 4,
                                          aggrow + (4 * g) * DSA + 4 * q, DSA);
    }
    if (r + 1 <= y1) store_feat();
    __syncthreads();
    if (SA && OUT) {                             // soft-argmin of output row y (uniform branch)
      const int spx = tid % TW, part = tid / TW;
      const bool has = part < (D >> 4);
      float* lv = aggrow + spx * DSA + 16 * part;
      if (has) {
        float m = -__builtin_inff();
#pragma unroll
        for (int i4 = 0; i4 < 4; ++i4) {
          const f32x4 v = *reinterpret_cast<const f32x4*>(lv + 4 * i4);
#pragma unroll
          for (int e = 0; e < 4; ++e) m = fmaxf(m, a.temperature * v[e]);
        }
        pmax[part * TW + spx] = m;
      }
      __syncthreads();
      if (has) {   // (the levels are read again rather than kept: the three register rows of the stencil stay live across this)
        float m = pmax[spx];
        for (int pp = 1; pp < (D >> 4); ++pp) m = fmaxf(m, pmax[pp * TW + spx]);
#pragma unroll
        for (int i4 = 0; i4 < 4; ++i4) {
          const f32x4 v = *reinterpret_cast<const f32x4*>(lv + 4 * i4);
          f32x4 ev;
#pragma unroll
          for (int e = 0; e < 4; ++e) ev[e] = a3_expf(a.temperature * v[e] - m);
          *reinterpret_cast<f32x4*>(lv + 4 * i4) = ev;
        }
      }
      __syncthreads();
      if (tid < TW && x0 + tid < a.W) {          // one lane per pixel: the oracle's sums, in its order
        // (the LDS reads run four quads ahead of the sums: a dependent chain of 2 D operations with a read latency in front of
        // every fourth one is what this lane would otherwise spend its time on)
        const f32x4* ep = reinterpret_cast<const f32x4*>(aggrow + tid * DSA);
        float ssum = 0.0f, tsum = 0.0f, df = 0.0f;
        f32x4 pre[4] = {ep[0], ep[1], ep[2], ep[3]};          // D >= 48: at least 12 quads
#pragma unroll 1
        for (int d4 = 0; d4 < DQ; d4 += 4) {
          f32x4 cur[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) cur[u] = pre[u];
#pragma unroll
          for (int u = 0; u < 4; ++u) pre[u] = ep[min(d4 + 4 + u, DQ - 1)];
#pragma unroll
          for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              ssum += cur[u][e];
              tsum = fmaf(df, cur[u][e], tsum);
              df += 1.0f;                                      // exact: level indices stay far below 2^24
            }
        }
        a.disp[((size_t)n * a.H + y) * a.W + x0 + tid] = tsum / ssum;
      }
    }
  };

  f32x2 r0[6][3], r1[6][3], r2[6][3];
#pragma unroll
  for (int c = 0; c < 6; ++c)
#pragma unroll
    for (int e = 0; e < 3; ++e) r0[c][e] = r1[c][e] = r2[c][e] = f32x2{0.f, 0.f};
  for (int e = tid; e < TW + 2; e += NT) {       // the two pad levels of every column
    costrow[e * DS] = 0.0f;
    costrow[e * DS + D + 1] = 0.0f;
  }
  load_feat(y0 - 1);
  store_feat();
  __syncthreads();
  const std::false_type fill{};
  const std::true_type emit{};
  step(fill, y0 - 1, r0, r1, r2);                // cost rows y0-1 and y0: no output row yet
  step(fill, y0, r1, r2, r0);
  for (int r = y0 + 1; r <= y1; r += 3) {        // row r completes output row r-1; y1 is the row under the last one
    step(emit, r, r2, r0, r1);
    if (r + 1 > y1) break;
    step(emit, r + 1, r0, r1, r2);
    if (r + 2 > y1) break;
    step(emit, r + 2, r1, r2, r0);
  }
}


// ---- volume -> volume layer on the same structure (st_volume_agg3d) --------------------------------------------------------
// The kernel above with the cost row read from the input volume instead of computed: 18 / 34 / 66 columns x D levels of
// row r+1 are requested (buffer loads, per-row descriptor: columns left / right of the image read as zero) right after the
// registers took row r, in front of the output stores, and go into the LDS row at the top of the next iteration.  One LDS
// row (14 KB) instead of the streaming kernel's ring of four (55 KB): three waves per SIMD instead of two; a volume element
// is fetched (TW + 2) / TW x (RY + 2) / RY times.
template <int TW, int DMAX, bool DFIX>
__global__ __launch_bounds__(192) __attribute__((amdgpu_waves_per_eu(3))) void vol_agg3d_kernel(const Agg3dArgs a) {
  constexpr int NT = 192, NG = TW / 4;
  constexpr int NV = ((TW + 2) * (DMAX / 4) + NT - 1) / NT;
  extern __shared__ float4 a3_smem4[];
  float* costrow = reinterpret_cast<float*>(a3_smem4);      // [TW + 2][DS], shifted by one level, zero pads (see above)
  const int D = DFIX ? DMAX : a.D, DQ = D >> 2, DS = D + 4;
  const int x0 = blockIdx.x * TW, y0 = blockIdx.y * a.RY, n = blockIdx.z;
  const int y1 = min(y0 + a.RY, a.Hf);
  const int tid = threadIdx.x;
  const int q = tid % DQ, g = tid / DQ;
  const bool active = (DFIX && TW * DMAX / 16 == NT) ? true : g < NG;
  const int nitem = (TW + 2) * DQ;
  // float4 i of a thread: quad e % DQ of column e / DQ (e = tid + NT i); byte offset in the image row: a column left of the
  // image is negative (= huge unsigned), one right of it beyond the row - outside the descriptor either way
  // (D fixed at compile time: NT is a multiple of D / 4, so float4 i sits NT / DQ columns right of float4 0 - one offset
  // register instead of NV)
  constexpr bool STEP = DFIX && NT % (DMAX / 4) == 0;
  constexpr int CSTEP = STEP ? NT / (DMAX / 4) : 0;
  int voff[STEP ? 1 : NV], ldst[STEP ? 1 : NV];
#pragma unroll
  for (int i = 0; i < (STEP ? 1 : NV); ++i) {
    const int e = tid + NT * i, col = e / DQ, qq = e - col * DQ;
    voff[i] = (STEP || e < nitem) ? ((x0 - 1 + col) * D + 4 * qq) * 4 : 0x7fffffff;
    ldst[i] = (STEP || e < nitem) ? col * DS + 4 * qq + 1 : -1;
  }
  auto voff_of = [&](int i) { return STEP ? (tid + NT * i < nitem ? voff[0] + i * CSTEP * D * 4 : 0x7fffffff) : voff[i]; };
  auto ldst_of = [&](int i) { return STEP ? (tid + NT * i < nitem ? ldst[0] + i * CSTEP * DS : -1) : ldst[i]; };
  f32x4 vst[NV];
  const int rowbytes = a.Wf * D * 4;
  auto load_vol = [&](int gy) {                  // row gy of the input volume -> registers (zero outside the image)
    if (gy < 0 || gy >= a.Hf) {                  // uniform
#pragma unroll
      for (int i = 0; i < NV; ++i) vst[i] = f32x4{0.f, 0.f, 0.f, 0.f};
      return;
    }
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(a.in + ((size_t)n * a.Hf + gy) * a.Wf * D), 0, rowbytes, 0x00020000);
#pragma unroll
    for (int i = 0; i < NV; ++i) vst[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff_of(i), 0, 0));
  };
  auto store_vol = [&]() {                       // registers -> the LDS row (level d at float d + 1)
#pragma unroll
    for (int i = 0; i < NV; ++i)
      if (ldst_of(i) >= 0) {
        float* dst = costrow + ldst_of(i);
        dst[0] = vst[i][0];
        *reinterpret_cast<f32x2*>(dst + 1) = f32x2{vst[i][1], vst[i][2]};
        dst[3] = vst[i][3];
      }
  };
  f32x2 wp[2][3][3];
  a3_weight_pairs(a.w, wp);
  const __amdgpu_buffer_rsrc_t nul = __builtin_amdgcn_make_buffer_rsrc(a.out, 0, 0, 0x00020000);
  auto step = [&](auto out_tag, int r, f32x2 (&ra)[6][3], f32x2 (&rb)[6][3], f32x2 (&rc)[6][3]) {
    constexpr bool OUT = decltype(out_tag)::value;
    store_vol();                                 // row r: requested one iteration ago, in front of that iteration's stores
    __syncthreads();
    if (active) {
#pragma unroll
      for (int col = 0; col < 6; ++col) {
        const float* p = costrow + (4 * g + col) * DS + 4 * q;
        const f32x4 v = *reinterpret_cast<const f32x4*>(p);
        rc[col][0] = f32x2{v[0], v[1]};
        rc[col][1] = f32x2{v[2], v[3]};
        rc[col][2] = *reinterpret_cast<const f32x2*>(p + 4);
      }
    }
    if (r + 1 <= y1) load_vol(r + 1);
    if (OUT) {
      if (active) {
        const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(
            a.out + (((size_t)n * a.Hf + (r - 1)) * a.Wf + x0) * D, 0, min(TW, a.Wf - x0) * D * 4, 0x00020000);
        a3_stencil_row<false>(ra, rb, rc, wp, a.w, a.bias, a.act, orsrc, ((4 * g) * D + 4 * q) * 4, D * 4);
      }
    } else {
      // the two fill iterations issue four stores as well (empty descriptor: dropped by the range check), so that every
      // path into the loop has the same loads-then-four-stores history and the wait for the loads stays vmcnt(4)
#pragma unroll
      for (int pi = 0; pi < 4; ++pi)
        __builtin_amdgcn_raw_buffer_store_b128(u32x4{0u, 0u, 0u, 0u}, nul, ((4 * g + pi) * D + 4 * q) * 4, 0, 0);
    }
    __syncthreads();
  };
  f32x2 r0[6][3], r1[6][3], r2[6][3];
#pragma unroll
  for (int c = 0; c < 6; ++c)
#pragma unroll
    for (int e = 0; e < 3; ++e) r0[c][e] = r1[c][e] = r2[c][e] = f32x2{0.f, 0.f};
  for (int e = tid; e < TW + 2; e += NT) {       // the two pad levels of every column
    costrow[e * DS] = 0.0f;
    costrow[e * DS + D + 1] = 0.0f;
  }
  load_vol(y0 - 1);
#pragma unroll
  for (int pi = 0; pi < 4; ++pi)
    __builtin_amdgcn_raw_buffer_store_b128(u32x4{0u, 0u, 0u, 0u}, nul, ((4 * g + pi) * D + 4 * q) * 4, 0, 0);
  const std::false_type fill{};
  const std::true_type emit{};
  step(fill, y0 - 1, r0, r1, r2);
  step(fill, y0, r1, r2, r0);
  for (int r = y0 + 1; r <= y1; r += 3) {
    step(emit, r, r2, r0, r1);
    if (r + 1 > y1) break;
    step(emit, r + 1, r0, r1, r2);
    if (r + 2 > y1) break;
    step(emit, r + 2, r1, r2, r0);
  }
}

}  // namespace
}  // namespace st

// Compute units of the current device (cached): the band-count searches below fill whole launch rounds of
// CUs x resident workgroups.  (Both kernels pin 3 waves per SIMD = 4 workgroups of 192 threads per CU; the count is a
// performance heuristic only - results are bit-exact for any band height.)
static int a3_cu_count(int* out) {
  static int cus = 0;
  if (cus == 0) {
    int dev = 0, n = 0;
    ST_CHECK_HIP(hipGetDevice(&dev));
    ST_CHECK_HIP(hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev));
    cus = std::max(1, n);
  }
  *out = cus;
  return ST_OK;
}

extern "C" int st_volume_agg3d(const float* vol_in_dev, float* vol_out_dev, int N, int Hf, int Wf, int D,
                               const float* weight27_host, float bias, int act, st_stream_t stream_) {
  using namespace st;
  ST_REQUIRE(vol_in_dev && vol_out_dev && weight27_host && vol_in_dev != vol_out_dev, "st_volume_agg3d: bad pointer");
  ST_REQUIRE(N > 0 && Hf > 0 && Wf > 0 && D >= 4 && D % 4 == 0 && D <= 192,
             "st_volume_agg3d: D must be a multiple of 4 in [4, 192] (got %d)", D);
  ST_REQUIRE(((reinterpret_cast<uintptr_t>(vol_in_dev) | reinterpret_cast<uintptr_t>(vol_out_dev)) & 15) == 0,
             "st_volume_agg3d: volumes must be 16-byte aligned");
  ST_REQUIRE(Hf < 65536 && N < 65536, "st_volume_agg3d: grid too large");
  // a volume row travels through 32-bit buffer offsets (per-row descriptors): Wf x D floats must stay below 2 GiB
  ST_REQUIRE((long long)Wf * D * 4 < (1ll << 31), "st_volume_agg3d: a volume row of Wf x D floats must be < 2 GiB");
  int cus = 0;
  ST_CHECK(a3_cu_count(&cus));
  Agg3dArgs a;
  a.in = vol_in_dev; a.out = vol_out_dev; a.N = N; a.Hf = Hf; a.Wf = Wf; a.D = D;
  for (int i = 0; i < 27; ++i) a.w[i] = weight27_host[i];
  a.bias = bias; a.act = act;
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  const int TW = D <= 48 ? 64 : (D <= 96 ? 32 : 16);
  const int strips = ceil_div(Wf, TW);
#ifdef ST_ABLATION
  const bool ring = getenv("ST_A3_RING") != nullptr;   // tools: the streaming kernel with the LDS ring of four rows
#else
  const bool ring = false;
#endif
  if (!ring) {
    // register-carried rows, one LDS row: TW x D / 16 threads (<= 192) own 4 pixel x 4 level tiles
    const int lds = (TW + 2) * (D + 4) * (int)sizeof(float);
    const long long slots = (long long)cus * 4;
    int best_b = 1;
    double best_e = -1.0;
    for (int b = 1; b <= 64 && ceil_div(Hf, b) >= 4; ++b) {
      const int ry = ceil_div(Hf, b), nb = ceil_div(Hf, ry);
      const long long wgs = (long long)N * strips * nb;
      const double fill = (double)wgs / (double)(ceil_div((int)std::min<long long>(wgs, 1 << 30), (int)slots) * slots);
      const double e = fill * ry / (ry + 2.0);
      if (e > best_e + 1e-9) { best_e = e; best_b = b; }
    }
    a.RY = std::min(Hf, ceil_div(Hf, best_b));
    const int bands = ceil_div(Hf, a.RY);
    ST_REQUIRE(bands < 65536, "st_volume_agg3d: grid too large");
    const dim3 grid((unsigned)strips, (unsigned)bands, (unsigned)N);
#define ST_VA3_LAUNCH(TWV, DMAXV)                                                                    \
  do {                                                                                               \
    auto kern = D == DMAXV ? vol_agg3d_kernel<TWV, DMAXV, true> : vol_agg3d_kernel<TWV, DMAXV, false>; \
    static int lds_set = 0;                                                                          \
    ST_ENSURE_DYNAMIC_LDS(kern, lds, lds_set);                                                       \
    hipLaunchKernelGGL(kern, grid, dim3(192), lds, stream, a);                                       \
  } while (0)
    if (TW == 64) ST_VA3_LAUNCH(64, 48);
    else if (TW == 32) ST_VA3_LAUNCH(32, 96);
    else ST_VA3_LAUNCH(16, 192);
#undef ST_VA3_LAUNCH
    ST_CHECK_HIP(hipGetLastError());
    return ST_OK;
  }
#ifdef ST_ABLATION
  // strip width by LDS budget (4 rows x (TW + 2) x D floats): 64 pixels up to 48 levels (51 KB), 32 up to 96 (52 KB),
  // 16 beyond (55 KB at D = 192) - two to three workgroups per CU in every case
  // Band height.  A workgroup's prologue (three rows) and its halo rows are overhead per band, a half-empty last launch
  // round is overhead per launch: pick the band count b (rows RY = ceil(Hf / b) >= 4) that maximises
  // (workgroups / slots rounded up to whole rounds) x RY / (RY + 3), slots = 256 CUs x workgroups per CU by LDS.
  const int lds = 4 * (TW + 2) * D * (int)sizeof(float);
  const long long slots = (long long)cus * std::max(1, (160 * 1024) / lds);
  int best_b = 1;
  double best_e = -1.0;
  for (int b = 1; b <= 64 && ceil_div(Hf, b) >= 4; ++b) {
    const int ry = ceil_div(Hf, b), nb = ceil_div(Hf, ry);
    const long long wgs = (long long)N * strips * nb;
    const double fill = (double)wgs / (double)(ceil_div((int)std::min<long long>(wgs, 1 << 30), (int)slots) * slots);
    const double e = fill * ry / (ry + 3.0);
    if (e > best_e + 1e-9) { best_e = e; best_b = b; }
  }
  int RY = std::min(Hf, ceil_div(Hf, best_b));
  a.RY = RY;
  const int bands = ceil_div(Hf, RY);
  ST_REQUIRE(bands < 65536, "st_volume_agg3d: grid too large");
  const dim3 grid((unsigned)strips, (unsigned)bands, (unsigned)N);
#define ST_A3_LAUNCH(TWV, NSTV)                                                                      \
  do {                                                                                               \
    auto kern = agg3d_kernel<TWV, NSTV>;                                                             \
    static int lds_set = 0;                                                                          \
    ST_ENSURE_DYNAMIC_LDS(kern, lds, lds_set);                                                       \
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, stream, a);                                       \
  } while (0)
  const int nst = ceil_div((TW + 2) * (D / 4), 256);
  if (TW == 64) {            // D <= 48: (66 * 12) / 256 -> up to 4
    if (nst <= 2) ST_A3_LAUNCH(64, 2); else ST_A3_LAUNCH(64, 4);
  } else if (TW == 32) {     // D <= 96: (34 * 24) / 256 -> 4
    ST_A3_LAUNCH(32, 4);
  } else {                   // D <= 192: (18 * 48) / 256 -> 4
    ST_A3_LAUNCH(16, 4);
  }
#undef ST_A3_LAUNCH
  ST_CHECK_HIP(hipGetLastError());
#endif
  return ST_OK;
}

extern "C" int st_costvolume_agg3d_supported(int C, int D) {
  return (C == 4 || C == 8 || C == 16) && D >= 4 && D % 4 == 0 && D <= 192;
}

static int cva_launch(const char* who, const float* featL_dev, const float* featR_dev, int N, int H, int W, int C, int ld,
                      int D, const float* weight27_host, float bias, int act, float* vol_out_dev, float temperature,
                      float* disp_out_dev, st_stream_t stream_) {
  using namespace st;
  const bool sa = disp_out_dev != nullptr;
  ST_REQUIRE(featL_dev && featR_dev && (vol_out_dev || sa) && weight27_host, "%s: bad pointer", who);
  ST_REQUIRE(N > 0 && H > 0 && W > 0, "%s: bad shape", who);
  ST_REQUIRE(st_costvolume_agg3d_supported(C, D),
             "%s: needs C in {4, 8, 16} and D a multiple of 4 in [4, 192] (got C = %d, D = %d)", who, C, D);
  ST_REQUIRE(!sa || D == 48 || D == 96 || D == 192,
             "%s: the fused soft-argmin is built for D = 48, 96 or 192 levels (got %d); other volumes take st_costvolume_agg3d + "
             "st_softargmin", who, D);
  ST_REQUIRE(ld >= C && ld % 4 == 0, "%s: ld must be a multiple of 4 and >= C", who);
  ST_REQUIRE(((reinterpret_cast<uintptr_t>(featL_dev) | reinterpret_cast<uintptr_t>(featR_dev) |
               reinterpret_cast<uintptr_t>(vol_out_dev)) & 15) == 0, "%s: buffers must be 16-byte aligned", who);
  ST_REQUIRE(H < 65536 && N < 65536, "%s: grid too large", who);
  // 32-bit buffer offsets inside one image row: feature rows (W x ld floats) and volume rows (W x D floats) below 2 GiB
  ST_REQUIRE((long long)W * ld * 4 < (1ll << 31) && (long long)W * D * 4 < (1ll << 31),
             "%s: an image row of W x ld (features) / W x D (volume) floats must be < 2 GiB", who);
  int cus = 0;
  ST_CHECK(a3_cu_count(&cus));
  CvAggArgs a;
  a.fl = featL_dev; a.fr = featR_dev; a.out = vol_out_dev;
  a.N = N; a.H = H; a.W = W; a.ld = ld; a.D = D;
  for (int i = 0; i < 27; ++i) a.w[i] = weight27_host[i];
  a.bias = bias; a.act = act;
  a.temperature = temperature; a.disp = disp_out_dev;
  // strip width: TW x D / 16 threads (<= 192) each own a 4 pixel x 4 level tile
  const int TW = D <= 48 ? 64 : (D <= 96 ? 32 : 16);
  const int strips = ceil_div(W, TW);
  // SA: + the aggregated row [TW][D + 4] and the partial maxima [D / 16][TW]
  const int lds = ((TW + 2) * (D + 4) + C * (D + TW + 4) + C * (TW + 8) + (sa ? TW * (D + 4) + (D / 16) * TW : 0)) * (int)sizeof(float);
  // band count: whole launch rounds over the device's CUs x 4 workgroups (register-bound), two produce-only rows per band
  const long long slots = (long long)cus * 4;
  int best_b = 1;
  double best_e = -1.0;
  for (int b = 1; b <= 64 && ceil_div(H, b) >= 4; ++b) {
    const int ry = ceil_div(H, b), nb = ceil_div(H, ry);
    const long long wgs = (long long)N * strips * nb;
    const double fill = (double)wgs / (double)(ceil_div((int)std::min<long long>(wgs, 1 << 30), (int)slots) * slots);
    const double e = fill * ry / (ry + 1.0);
    if (e > best_e + 1e-9) { best_e = e; best_b = b; }
  }
  a.RY = std::min(H, ceil_div(H, best_b));
  const int bands = ceil_div(H, a.RY);
  ST_REQUIRE(bands < 65536, "%s: grid too large", who);
  a.strips = strips; a.bands = bands;
  a.order = 1;
#ifdef ST_ABLATION
  if (const char* o = getenv("ST_CVA_ORDER")) a.order = atoi(o);     // tools: 0 = launch order
#endif
  ST_REQUIRE((long long)N * strips * bands < (1ll << 31), "%s: grid too large", who);
  const dim3 grid((unsigned)(N * strips * bands));
  hipStream_t stream = static_cast<hipStream_t>(stream_);
#define ST_CVA_LAUNCH(TWV, CV, DMAXV)                                                                \
  do {                                                                                               \
    if (sa) {                                                                                        \
      auto kern = cv_agg3d_kernel<TWV, CV, DMAXV, true, 0, true>;   /* D == DMAXV: required above */   \
      static int lds_set = 0;                                                                        \
      ST_ENSURE_DYNAMIC_LDS(kern, lds, lds_set);                                                     \
      hipLaunchKernelGGL(kern, grid, dim3(192), lds, stream, a);                                     \
    } else {                                                                                         \
      auto kern = D == DMAXV ? cv_agg3d_kernel<TWV, CV, DMAXV, true> : cv_agg3d_kernel<TWV, CV, DMAXV, false>; \
      static int lds_set = 0;                                                                        \
      ST_ENSURE_DYNAMIC_LDS(kern, lds, lds_set);                                                     \
      hipLaunchKernelGGL(kern, grid, dim3(192), lds, stream, a);                                     \
    }                                                                                                \
  } while (0)
#define ST_CVA_BY_C(TWV, DMAXV)                                                                      \
  do {                                                                                               \
    if (C == 4) ST_CVA_LAUNCH(TWV, 4, DMAXV);                                                        \
    else if (C == 8) ST_CVA_LAUNCH(TWV, 8, DMAXV);                                                   \
    else ST_CVA_LAUNCH(TWV, 16, DMAXV);                                                              \
  } while (0)
#ifdef ST_ABLATION
  if (const char* m = sa ? nullptr : getenv("ST_CVA_MODE")) {     // tools: timing-only ablations at the full-resolution shape
    const int mode = atoi(m);
    if (mode > 0 && D == 192 && C == 8) {
      auto launch = [&](auto kern) -> int {
        ST_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        hipLaunchKernelGGL(kern, grid, dim3(192), lds, stream, a);
        return ST_OK;
      };
      switch (mode) {
        case 1: return launch(cv_agg3d_kernel<16, 8, 192, true, 1>);
        case 9: return launch(cv_agg3d_kernel<16, 8, 192, true, 9>);
        case 2: return launch(cv_agg3d_kernel<16, 8, 192, true, 2>);
        case 8: return launch(cv_agg3d_kernel<16, 8, 192, true, 8>);
        default: break;
      }
    }
  }
#endif
  if (TW == 64) ST_CVA_BY_C(64, 48);
  else if (TW == 32) ST_CVA_BY_C(32, 96);
  else ST_CVA_BY_C(16, 192);
#undef ST_CVA_BY_C
#undef ST_CVA_LAUNCH
  ST_CHECK_HIP(hipGetLastError());
  return ST_OK;
}

extern "C" int st_costvolume_agg3d(const float* featL_dev, const float* featR_dev, int N, int H, int W, int C, int ld,
                                   int D, const float* weight27_host, float bias, int act, float* vol_out_dev,
                                   st_stream_t stream_) {
  ST_REQUIRE(vol_out_dev, "st_costvolume_agg3d: bad pointer");
  return cva_launch("st_costvolume_agg3d", featL_dev, featR_dev, N, H, W, C, ld, D, weight27_host, bias, act, vol_out_dev, 0.0f,
                    nullptr, stream_);
}

extern "C" int st_costvolume_agg3d_softargmin(const float* featL_dev, const float* featR_dev, int N, int H, int W, int C,
                                              int ld, int D, const float* weight27_host, float bias, int act,
                                              float temperature, float* disp_out_dev, st_stream_t stream_) {
  ST_REQUIRE(disp_out_dev, "st_costvolume_agg3d_softargmin: bad pointer");
  return cva_launch("st_costvolume_agg3d_softargmin", featL_dev, featR_dev, N, H, W, C, ld, D, weight27_host, bias, act, nullptr,
                    temperature, disp_out_dev, stream_);
}
