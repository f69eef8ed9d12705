// Batched GPU association: the depth-guided OC-SORT step of MANY independent sequences advanced in
// lockstep on the device - one workgroup (one wave) per sequence and frame.
//
// Behavioural spec = the host routine csrc/ocsort_tracker.cpp (st_tracker_track), i.e. reference
//   OCSORTTracker_Disparity.track            mmtrack/models/trackers/ocsort_tracker_disparity.py:345-618
//     ocm_assign_ids / ocr_assign_ids / online_smooth          :187-265, :273-317, :319-343
//   KalmanFilter initiate / predict / update                   mmtrack/models/motion/kalman_filter.py:60-189
//   lap.lapjv(extend_cost=True, cost_limit)                    -> the same dense Jonker-Volgenant as csrc/lapjv.cpp
// The ids and every row this kernel returns are EQUAL to the host routine's on the same detections
// (tests/test_batched_assoc_gpu.py): the association arithmetic is the same single IEEE operations in the same order
// (float32 boxes / IoU / direction term, float64 Kalman filter and assignment; the file is built with
// -ffp-contract=off), the assignment walks the same comparisons in the same order.  One documented last-bit
// difference: acosf (ocml here, glibc on the host: <= 1 ulp on the direction term).
//
// Why a wave per sequence.  One sequence is a SEQUENTIAL algorithm (the north_star keeps it on the CPU, and for one
// video the CPU wins: 0.05 ms per frame).  The GPU form pays when there are hundreds of short sequences per step
// (multi-camera serving): the embarrassingly parallel parts of a sequence (Kalman prediction per track, the
// tracks x detections cost matrix, the Kalman updates per matched track) run across the 64 lanes, the branchy
// bookkeeping and the Jonker-Volgenant path search run on lane 0 with their working set in LDS / L2, and the chip runs
// thousands of such waves at once.  State lives in device memory between steps (no per-frame host round trip).
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <memory>

#include "st_common.h"

namespace st {
namespace ba {

constexpr int WINCAP = 8;           // vel_delta_t + 1 <= 8 observations kept per track
constexpr double kBig = 1000000.0;
constexpr double kWPos = 1.0 / 20, kWVel = 1.0 / 160;

struct KState { double mean[8]; double cov[64]; };

struct DTrack {
  long long id;
  KState kf, saved;
  int tentative, tracked, last_frame, n_fed;
  float win[WINCAP][4];
  int win_valid[WINCAP];
  int win_len;
  long long n_obs;
  float last_valid[4];
  int trailing_none;
  float vel[2];
  int vel_placeholder;
};

// `poisoned`: sticky status of the sequence.  A capacity overflow is detected after the association has already
// mutated the tracks (Kalman predictions, observation windows, ids handed out), so the sequence is INVALID from then
// on: every later step reports the same status with an output count of 0 until frame_id 0 resets it.
struct SeqHeader { int n_tracks; int poisoned; long long num_tracks; };

struct Cfg {
  float obj_score_thr, init_track_thr, match_iou_thr, vel_consist_weight;
  int weight_iou_with_det_scores, num_tentatives, vel_delta_t, num_frames_retain;
  int max_tracks, max_dets;
};

// ---- Kalman filter (float64; the loops of ocsort_tracker.cpp) ---------------------------------------------------
__device__ void kf_initiate(const float meas[4], KState& s) {
  for (int i = 0; i < 4; ++i) { s.mean[i] = (double)meas[i]; s.mean[4 + i] = 0.0; }
  const double h = (double)meas[3];
  const double sd[8] = {2 * kWPos * h, 2 * kWPos * h, 1e-2, 2 * kWPos * h,
                        10 * kWVel * h, 10 * kWVel * h, 1e-5, 10 * kWVel * h};
  for (int i = 0; i < 64; ++i) s.cov[i] = 0.0;
  for (int i = 0; i < 8; ++i) s.cov[i * 8 + i] = sd[i] * sd[i];
}

__device__ void kf_predict(KState& s) {
  const double h = s.mean[3];
  const double sd[8] = {kWPos * h, kWPos * h, 1e-2, kWPos * h, kWVel * h, kWVel * h, 1e-5, kWVel * h};
  for (int i = 0; i < 4; ++i) s.mean[i] = s.mean[i] + s.mean[i + 4];
  double t[64];
  for (int i = 0; i < 8; ++i)
    for (int j = 0; j < 8; ++j) t[i * 8 + j] = j < 4 ? s.cov[i * 8 + j] + s.cov[i * 8 + j + 4] : s.cov[i * 8 + j];
  for (int i = 0; i < 8; ++i)
    for (int j = 0; j < 8; ++j) s.cov[i * 8 + j] = i < 4 ? t[i * 8 + j] + t[(i + 4) * 8 + j] : t[i * 8 + j];
  for (int i = 0; i < 8; ++i) s.cov[i * 8 + i] += sd[i] * sd[i];
}

__device__ void kf_update(KState& s, const float meas[4]) {
  const double h = s.mean[3];
  const double sd[4] = {kWPos * h, kWPos * h, 1e-1, kWPos * h};
  double S[16];
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) S[i * 4 + j] = s.cov[i * 8 + j] + (i == j ? sd[i] * sd[i] : 0.0);
  double L[16];
  for (int i = 0; i < 16; ++i) L[i] = 0.0;
  for (int j = 0; j < 4; ++j) {
    double d = S[j * 4 + j];
    for (int k = 0; k < j; ++k) d -= L[j * 4 + k] * L[j * 4 + k];
    d = sqrt(d);
    L[j * 4 + j] = d;
    for (int i = j + 1; i < 4; ++i) {
      double v = S[i * 4 + j];
      for (int k = 0; k < j; ++k) v -= L[i * 4 + k] * L[j * 4 + k];
      L[i * 4 + j] = v / d;
    }
  }
  double X[32];
  for (int c = 0; c < 8; ++c) {
    double y[4];
    for (int i = 0; i < 4; ++i) {
      double v = s.cov[c * 8 + i];
      for (int k = 0; k < i; ++k) v -= L[i * 4 + k] * y[k];
      y[i] = v / L[i * 4 + i];
    }
    for (int i = 3; i >= 0; --i) {
      double v = y[i];
      for (int k = i + 1; k < 4; ++k) v -= L[k * 4 + i] * X[k * 8 + c];
      X[i * 8 + c] = v / L[i * 4 + i];
    }
  }
  double innov[4];
  for (int i = 0; i < 4; ++i) innov[i] = (double)meas[i] - s.mean[i];
  for (int c = 0; c < 8; ++c) {
    double acc = 0.0;
    for (int r = 0; r < 4; ++r) acc += innov[r] * X[r * 8 + c];
    s.mean[c] += acc;
  }
  double SX[32];
  for (int r = 0; r < 4; ++r)
    for (int c = 0; c < 8; ++c) {
      double acc = 0.0;
      for (int k = 0; k < 4; ++k) acc += S[r * 4 + k] * X[k * 8 + c];
      SX[r * 8 + c] = acc;
    }
  for (int i = 0; i < 8; ++i)
    for (int j = 0; j < 8; ++j) {
      double acc = 0.0;
      for (int k = 0; k < 4; ++k) acc += X[k * 8 + i] * SX[k * 8 + j];
      s.cov[i * 8 + j] -= acc;
    }
}

// ---- float32 box helpers (single IEEE ops, reference order) -----------------------------------------------------
__device__ inline void xyxy_to_cxcyah(const float b[4], float o[4]) {
  o[0] = (b[2] + b[0]) / 2;
  o[1] = (b[3] + b[1]) / 2;
  const float w = b[2] - b[0], h = b[3] - b[1];
  o[2] = w / h;
  o[3] = h;
}
__device__ inline void cxcyah_to_xyxy(const float b[4], float o[4]) {
  const float w = b[2] * b[3];
  o[0] = b[0] - w / 2.0f;
  o[1] = b[1] - b[3] / 2.0f;
  o[2] = b[0] + w / 2.0f;
  o[3] = b[1] + b[3] / 2.0f;
}
__device__ inline float tmax(float a, float b) { return a != a ? a : (b != b ? b : (a > b ? a : b)); }
__device__ inline float tmin(float a, float b) { return a != a ? a : (b != b ? b : (a < b ? a : b)); }
__device__ inline float iou(const float a[4], const float b[4]) {
  const float area1 = (a[2] - a[0]) * (a[3] - a[1]);
  const float area2 = (b[2] - b[0]) * (b[3] - b[1]);
  const float w = tmax(tmin(a[2], b[2]) - tmax(a[0], b[0]), 0.f);
  const float h = tmax(tmin(a[3], b[3]) - tmax(a[1], b[1]), 0.f);
  const float overlap = w * h;
  const float uni = tmax(area1 + area2 - overlap, 1e-6f);
  return overlap / uni;
}

// ---- observation history ----------------------------------------------------------------------------------------
__device__ void push_obs(DTrack& t, const float* box, int vel_delta_t) {
  const int cap = vel_delta_t + 1;
  if (t.win_len == cap) {
    for (int k = 1; k < cap; ++k) {
      for (int e = 0; e < 4; ++e) t.win[k - 1][e] = t.win[k][e];
      t.win_valid[k - 1] = t.win_valid[k];
    }
    --t.win_len;
  }
  for (int e = 0; e < 4; ++e) t.win[t.win_len][e] = box ? box[e] : 0.f;
  t.win_valid[t.win_len] = box != nullptr;
  ++t.win_len;
  ++t.n_obs;
  if (box) {
    for (int e = 0; e < 4; ++e) t.last_valid[e] = box[e];
    t.trailing_none = 0;
  } else {
    ++t.trailing_none;
  }
}
__device__ const float* k_step_observation(const DTrack& t, int vel_delta_t) {
  if (t.n_obs > vel_delta_t && t.win_len == vel_delta_t + 1 && t.win_valid[0]) return t.win[0];
  return t.last_valid;
}
__device__ void set_velocity(DTrack& t, const float* b1, const float* b2) {
  const float s1 = ((b1[0] + b1[1]) + b1[2]) + b1[3], s2 = ((b2[0] + b2[1]) + b2[2]) + b2[3];
  if (s1 < 0 || s2 < 0) { t.vel[0] = t.vel[1] = -1.f; t.vel_placeholder = 1; return; }
  const float cx1 = (b1[0] + b1[2]) / 2.0f, cy1 = (b1[1] + b1[3]) / 2.0f;
  const float cx2 = (b2[0] + b2[2]) / 2.0f, cy2 = (b2[1] + b2[3]) / 2.0f;
  const float sy = cy2 - cy1, sx = cx2 - cx1;
  const float norm = sqrtf(sy * sy + sx * sx) + 1e-6f;
  t.vel[0] = sy / norm;
  t.vel[1] = sx / norm;
  t.vel_placeholder = (t.vel[0] + t.vel[1]) == -2.0f;
}
__device__ void init_track(DTrack& t, long long id, const float* row, int frame_id, int vel_delta_t) {
  t.id = id;
  t.n_fed = 1;
  t.last_frame = frame_id;
  t.tentative = frame_id != 0;
  t.win_len = 0;
  t.n_obs = 0;
  t.trailing_none = 0;
  for (int e = 0; e < 4; ++e) t.last_valid[e] = 0.f;
  t.vel[0] = t.vel[1] = -1.f;
  t.vel_placeholder = 1;
  float m[4];
  xyxy_to_cxcyah(row, m);
  kf_initiate(m, t.kf);
  t.saved = t.kf;
  push_obs(t, row, vel_delta_t);
  t.tracked = 1;
}
__device__ void update_track(DTrack& t, const float* row, int frame_id, const Cfg& cfg) {
  ++t.n_fed;
  t.last_frame = frame_id;
  if (t.tentative && t.n_fed >= cfg.num_tentatives) t.tentative = 0;
  float m[4];
  xyxy_to_cxcyah(row, m);
  kf_update(t.kf, m);
  t.tracked = 1;
  push_obs(t, row, cfg.vel_delta_t);
  set_velocity(t, k_step_observation(t, cfg.vel_delta_t), row);
}
__device__ void online_smooth(DTrack& t, const float* new_box) {
  float last[4];
  for (int i = 0; i < 4; ++i) last[i] = t.last_valid[i];
  const int gap = t.trailing_none;
  float step[4];
  for (int i = 0; i < 4; ++i) step[i] = (new_box[i] - last[i]) / (float)(gap + 1);
  t.kf = t.saved;
  for (int g = 0; g < gap; ++g) {
    float vb[4], m[4];
    for (int i = 0; i < 4; ++i) vb[i] = last[i] + (float)(g + 1) * step[i];
    xyxy_to_cxcyah(vb, m);
    kf_update(t.kf, m);
  }
}

// ---- dense Jonker-Volgenant on lap's (R + C)^2 extension, never materialised: at(i, j) is computed ----------------
struct Lap {
  const double* cost;   // R x C row major
  int R, C, n;
  double half;          // cost_limit / 2
  double* v;
  double* d;
  int *row_to_col, *col_to_row, *free_rows, *pred, *cols;
  char* once;
  int n_free;
  __device__ double at(int i, int j) const {
    if (i < R) {
      if (j < C) {
        const double c = cost[(size_t)i * C + j];
        return c != c ? 1e6 : c;
      }
      return half;
    }
    return j < C ? half : 0.0;
  }
  // ---- wave-cooperative form: every lane of the sequence's wave calls these; the O(n^2) scans run across the 64
  // lanes with EXACTLY the comparisons' outcomes of the serial solver (csrc/lapjv.cpp) - per-column work is
  // independent, value-only minima are order-free, and the (best, second) pair of a row scan is "lowest index of
  // the minimum, lowest index of the second smallest value among the rest", which is what the serial update rule
  // leaves behind (rows containing an unmatchable 1e6 sentinel fall back to the serial scan on lane 0).
  __device__ void reduce_columns(int lane, int* ctl) {
    for (int i = lane; i < n; i += 64) { row_to_col[i] = -1; col_to_row[i] = 0; v[i] = kBig; once[i] = 1; }
    __syncthreads();
    for (int j = lane; j < n; j += 64) {       // a lane owns its columns: rows in ascending order, strict <
      double vj = kBig;
      int cj = 0;
      for (int i = 0; i < n; ++i) {
        const double a = at(i, j);
        if (a < vj) { vj = a; cj = i; }
      }
      v[j] = vj;
      col_to_row[j] = cj;
    }
    __syncthreads();
    if (lane == 0) {
      for (int j = n - 1; j >= 0; --j) {
        const int i = col_to_row[j];
        if (row_to_col[i] < 0) {
          row_to_col[i] = j;
        } else {
          once[i] = 0;
          col_to_row[j] = -1;
        }
      }
      int nf = 0;
      for (int i = 0; i < n; ++i)
        if (row_to_col[i] < 0) free_rows[nf++] = i;
      n_free = nf;
      ctl[0] = nf;
    }
    __syncthreads();
    n_free = ctl[0];
    // reduction transfer, rows in ascending order (a row's slack uses the prices the earlier rows left)
    for (int i = 0; i < n; ++i) {
      const int j = row_to_col[i];
      if (j < 0 || !once[i]) continue;          // uniform: all lanes read the same values
      double slack = kBig;
      for (int k = lane; k < n; k += 64) {
        if (k == j) continue;
        const double r = at(i, k) - v[k];
        if (r < slack) slack = r;
      }
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) {
        const double o = __shfl_xor(slack, off);
        if (o < slack) slack = o;
      }
      __syncthreads();
      if (lane == 0) v[j] -= slack;
      __syncthreads();
    }
  }
  // serial row scan (the solver's own loop): used by lane 0 for rows that contain a sentinel
  __device__ void scan_row_serial(int i, int& best, int& second, double& u1, double& u2) const {
    best = 0; second = -1;
    u1 = at(i, 0) - v[0]; u2 = kBig;
    for (int j = 1; j < n; ++j) {
      const double r = at(i, j) - v[j];
      if (r < u2) {
        if (r >= u1) { u2 = r; second = j; }
        else { u2 = u1; u1 = r; second = best; best = j; }
      }
    }
  }
  __device__ void reduce_rows(int lane, int* ctl, double* ctld) {
    // ctl: 1 = cur, 2 = kept, 3 = todo, 4 = row i of this iteration (-1: done), 5.. scan results
    if (lane == 0) { ctl[1] = 0; ctl[2] = 0; ctl[3] = n_free; }
    long long sweeps = 0;
    __syncthreads();
    while (true) {
      if (lane == 0) ctl[4] = ctl[1] < ctl[3] ? free_rows[ctl[1]] : -1;
      __syncthreads();
      const int i = ctl[4];
      if (i < 0) break;
      ++sweeps;
      // parallel scan: lexicographic minimum of (r, j), then of the rest
      double b1 = kBig * 4; int j1 = 0x7fffffff; bool sentinel = false;
      for (int j = lane; j < n; j += 64) {
        const double r = at(i, j) - v[j];
        if (!(r < kBig)) sentinel = true;
        if (r < b1 || (r == b1 && j < j1)) { b1 = r; j1 = j; }
      }
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) {
        const double ob = __shfl_xor(b1, off);
        const int oj = __shfl_xor(j1, off);
        if (ob < b1 || (ob == b1 && oj < j1)) { b1 = ob; j1 = oj; }
      }
      const bool any_sentinel = __ballot(sentinel) != 0ull;
      double b2 = kBig * 4; int j2 = 0x7fffffff;
      for (int j = lane; j < n; j += 64) {
        if (j == j1) continue;
        const double r = at(i, j) - v[j];
        if (r < b2 || (r == b2 && j < j2)) { b2 = r; j2 = j; }
      }
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) {
        const double ob = __shfl_xor(b2, off);
        const int oj = __shfl_xor(j2, off);
        if (ob < b2 || (ob == b2 && oj < j2)) { b2 = ob; j2 = oj; }
      }
      __syncthreads();
      if (lane == 0) {
        int best, second;
        double u1, u2;
        if (any_sentinel || n < 2) scan_row_serial(i, best, second, u1, u2);
        else { best = j1; u1 = b1; second = j2; u2 = b2; }
        int cur = ctl[1] + 1, kept = ctl[2];
        int owner = col_to_row[best];
        const double lowered = v[best] - (u2 - u1);
        const bool lowers = lowered < v[best];
        if (sweeps < (long long)cur * n) {
          if (lowers) {
            v[best] = lowered;
          } else if (owner >= 0 && second >= 0) {
            best = second;
            owner = col_to_row[second];
          }
          if (owner >= 0) {
            if (lowers) free_rows[--cur] = owner;
            else free_rows[kept++] = owner;
          }
        } else if (owner >= 0) {
          free_rows[kept++] = owner;
        }
        row_to_col[i] = best;
        col_to_row[best] = i;
        ctl[1] = cur; ctl[2] = kept;
      }
      __syncthreads();
    }
    n_free = ctl[2];
    (void)ctld;
  }
  __device__ int shortest_path(int start) {   // lane 0 (the SCAN / TODO partition is order-dependent)
    int lo = 0, hi = 0, ready = 0, sink = -1;
    while (sink < 0) {
      if (lo == hi) {
        ready = lo;
        hi = lo + 1;
        double dmin = d[cols[lo]];
        for (int k = hi; k < n; ++k) {
          const int j = cols[k];
          if (d[j] <= dmin) {
            if (d[j] < dmin) { hi = lo; dmin = d[j]; }
            cols[k] = cols[hi];
            cols[hi++] = j;
          }
        }
        for (int k = lo; k < hi; ++k)
          if (col_to_row[cols[k]] < 0) sink = cols[k];
      }
      if (sink < 0) {
        int l = lo, h = hi;
        bool found = false;
        while (l != h && !found) {
          int j = cols[l++];
          const int i = col_to_row[j];
          const double dmin = d[j];
          const double base = at(i, j) - v[j] - dmin;
          for (int k = h; k < n; ++k) {
            j = cols[k];
            const double r = at(i, j) - v[j] - base;
            if (r < d[j]) {
              d[j] = r;
              pred[j] = i;
              if (r == dmin) {
                if (col_to_row[j] < 0) { sink = j; found = true; break; }
                cols[k] = cols[h];
                cols[h++] = j;
              }
            }
          }
        }
        if (!found) { lo = l; hi = h; }
      }
    }
    const double dmin = d[cols[lo]];
    for (int k = 0; k < ready; ++k) {
      const int j = cols[k];
      v[j] += d[j] - dmin;
    }
    return sink;
  }
  __device__ void solve(int lane, int* ctl, double* ctld) {
    reduce_columns(lane, ctl);
    for (int pass = 0; pass < 2; ++pass) {
      if (n_free <= 0) break;                    // uniform (n_free is set from ctl on every lane)
      reduce_rows(lane, ctl, ctld);
    }
    for (int f = 0; f < n_free; ++f) {
      const int start = free_rows[f];
      for (int j = lane; j < n; j += 64) {       // the path search's initial distances, across the lanes
        cols[j] = j;
        pred[j] = start;
        d[j] = at(start, j) - v[j];
      }
      __syncthreads();
      if (lane == 0) {
        int j = shortest_path(start);
        int i = -1;
        while (i != start) {
          i = pred[j];
          col_to_row[j] = i;
          const int prev = row_to_col[i];
          row_to_col[i] = j;
          j = prev;
        }
      }
      __syncthreads();
    }
  }
};

// per-sequence scratch (global memory, L2-resident): sizes in elements for n = max_tracks + max_dets
struct Scratch {
  double* cost;      // max_tracks x max_dets
  double *v, *d;     // n
  int *row_to_col, *col_to_row, *free_rows, *pred, *cols;   // n each
  char* once;        // n
  int *cand, *rest, *tidx, *confirmed, *tentative, *lost, *matched_det, *matched_trk, *d2r, *order;   // max_dets / max_tracks
  long long* ids;    // max_dets
  char* is_matched;  // max_tracks
};

__host__ __device__ inline size_t align8(size_t x) { return (x + 7) & ~(size_t)7; }

__host__ __device__ inline size_t scratch_bytes(int T, int M) {
  const size_t n = (size_t)T + M;
  size_t o = 0;
  o += align8(sizeof(double) * (size_t)T * M);
  o += 2 * align8(sizeof(double) * n);
  o += 5 * align8(sizeof(int) * n);
  o += align8(n);
  o += 10 * align8(sizeof(int) * n);
  o += align8(sizeof(long long) * (size_t)M);
  o += align8((size_t)T);
  return o;
}

__device__ inline Scratch carve(char* base, int T, int M) {
  const size_t n = (size_t)T + M;
  Scratch s;
  size_t o = 0;
  auto take = [&](size_t bytes) { char* p = base + o; o += align8(bytes); return p; };
  s.cost = (double*)take(sizeof(double) * (size_t)T * M);
  s.v = (double*)take(sizeof(double) * n);
  s.d = (double*)take(sizeof(double) * n);
  s.row_to_col = (int*)take(sizeof(int) * n);
  s.col_to_row = (int*)take(sizeof(int) * n);
  s.free_rows = (int*)take(sizeof(int) * n);
  s.pred = (int*)take(sizeof(int) * n);
  s.cols = (int*)take(sizeof(int) * n);
  s.once = take(n);
  s.cand = (int*)take(sizeof(int) * n);
  s.rest = (int*)take(sizeof(int) * n);
  s.tidx = (int*)take(sizeof(int) * n);
  s.confirmed = (int*)take(sizeof(int) * n);
  s.tentative = (int*)take(sizeof(int) * n);
  s.lost = (int*)take(sizeof(int) * n);
  s.matched_det = (int*)take(sizeof(int) * n);
  s.matched_trk = (int*)take(sizeof(int) * n);
  s.d2r = (int*)take(sizeof(int) * n);
  s.order = (int*)take(sizeof(int) * n);
  s.ids = (long long*)take(sizeof(long long) * (size_t)M);
  s.is_matched = take((size_t)T);
  return s;
}

// one association stage: rows = tracks tidx[0..R), cols = detections pool[0..Cn) -> d2r[c] = row matched to column c
// or -1.  The cost matrix is filled by all 64 lanes, the assignment runs on lane 0.
__device__ void assign_stage(const Cfg& cfg, DTrack* tracks, const int* tidx, int R, const int* pool, int Cn,
                             const float* dets, bool with_motion, Scratch& s, int lane) {
  for (int c = lane; c < Cn; c += 64) s.d2r[c] = -1;
  if (R == 0 || Cn == 0) {
    __syncthreads();
    return;
  }
  for (int e = lane; e < R * Cn; e += 64) {
    const int r = e / Cn, c = e - r * Cn;
    const DTrack& t = tracks[tidx[r]];
    float tb[4];
    if (with_motion) {
      const float m[4] = {(float)t.kf.mean[0], (float)t.kf.mean[1], (float)t.kf.mean[2], (float)t.kf.mean[3]};
      cxcyah_to_xyxy(m, tb);
    } else {
      for (int k = 0; k < 4; ++k) tb[k] = t.last_valid[k];
    }
    const float* ko = with_motion ? k_step_observation(t, cfg.vel_delta_t) : nullptr;
    const bool valid = with_motion && !t.vel_placeholder && (((ko[0] + ko[1]) + ko[2]) + ko[3]) != -4.0f;
    const float* dd = dets + (size_t)pool[c] * 8;
    float v = iou(tb, dd);
    if (cfg.weight_iou_with_det_scores) v = v * dd[4];
    float dist = 1.0f - v;
    if (with_motion) {
      const float cx1 = (ko[0] + ko[2]) / 2.0f, cy1 = (ko[1] + ko[3]) / 2.0f;
      const float cx2 = (dd[0] + dd[2]) / 2.0f, cy2 = (dd[1] + dd[3]) / 2.0f;
      const float sy = cy2 - cy1, sx = cx2 - cx1;
      const float norm = sqrtf(sy * sy + sx * sx) + 1e-6f;
      float cosv = (sy / norm) * t.vel[0] + (sx / norm) * t.vel[1];
      cosv = cosv < -1.f ? -1.f : (cosv > 1.f ? 1.f : cosv);
      const float ang = (acosf(cosv) - (float)(M_PI / 2.)) / (float)M_PI;
      const float term = ang * (valid ? 1.0f : 0.0f);
      dist = dist + term * cfg.vel_consist_weight;
    }
    s.cost[(size_t)r * Cn + c] = (double)dist;
  }
  __syncthreads();
  __shared__ int lap_ctl[8];
  __shared__ double lap_ctld[2];
  Lap lap;
  lap.cost = s.cost; lap.R = R; lap.C = Cn; lap.n = R + Cn;
  lap.half = (1.0 - (double)cfg.match_iou_thr) / 2.0;
  lap.v = s.v; lap.d = s.d; lap.row_to_col = s.row_to_col; lap.col_to_row = s.col_to_row;
  lap.free_rows = s.free_rows; lap.pred = s.pred; lap.cols = s.cols; lap.once = s.once; lap.n_free = 0;
  lap.solve(lane, lap_ctl, lap_ctld);
  for (int c = lane; c < Cn; c += 64) s.d2r[c] = lap.col_to_row[c] < R ? lap.col_to_row[c] : -1;
  __syncthreads();
}

// status codes written per sequence
constexpr int kOk = 0, kTrackOverflow = 1, kDetOverflow = 2;

__global__ __launch_bounds__(64) void assoc_step_kernel(Cfg cfg, const int* __restrict__ frame_ids,
                                                        const float* __restrict__ dets_all,
                                                        const int* __restrict__ counts, char* state_all,
                                                        size_t state_stride, char* scratch_all, size_t scratch_stride,
                                                        float* __restrict__ out_rows, long long* __restrict__ out_ids,
                                                        int* __restrict__ out_n, int* __restrict__ status,
                                                        int lap_in_lds) {
  const int b = blockIdx.x, lane = threadIdx.x;
  const int T = cfg.max_tracks, M = cfg.max_dets;
  SeqHeader* hdr = reinterpret_cast<SeqHeader*>(state_all + (size_t)b * state_stride);
  DTrack* tracks = reinterpret_cast<DTrack*>(reinterpret_cast<char*>(hdr) + sizeof(SeqHeader));
  Scratch s = carve(scratch_all + (size_t)b * scratch_stride, T, M);
  if (lap_in_lds) {   // the assignment's working arrays (37 B per row / column of the extended matrix) in LDS
    extern __shared__ double ba_dyn[];
    const size_t n = (size_t)T + M;
    char* base = reinterpret_cast<char*>(ba_dyn);
    size_t o = 0;
    auto take = [&](size_t bytes) { char* q = base + o; o += align8(bytes); return q; };
    s.v = (double*)take(sizeof(double) * n);
    s.d = (double*)take(sizeof(double) * n);
    s.row_to_col = (int*)take(sizeof(int) * n);
    s.col_to_row = (int*)take(sizeof(int) * n);
    s.free_rows = (int*)take(sizeof(int) * n);
    s.pred = (int*)take(sizeof(int) * n);
    s.cols = (int*)take(sizeof(int) * n);
    s.once = take(n);
  }
  const float* dets = dets_all + (size_t)b * M * 8;
  const int frame_id = frame_ids[b];
  const int n = counts[b];
  float* orow = out_rows + (size_t)b * M * 8;
  long long* oid = out_ids + (size_t)b * M;
  __shared__ int sh[12];   // n_tracks, n_order, n_cand, n_conf, n_tent, n_lost, n_matched, first_stage flag, status
  if (n < 0) {             // padding slot: this sequence has no frame in this step
    if (lane == 0) out_n[b] = -1;
    return;
  }
  if (frame_id != 0 && hdr->poisoned != kOk) {   // an earlier step overflowed: the state is not a tracker state any more
    if (lane == 0) { out_n[b] = 0; status[b] = hdr->poisoned; }
    return;
  }
  if (n > M) {
    if (lane == 0) { out_n[b] = 0; status[b] = kDetOverflow; hdr->poisoned = kDetOverflow; }
    return;
  }
  if (lane == 0) {
    if (frame_id == 0) { hdr->n_tracks = 0; hdr->num_tracks = 0; hdr->poisoned = kOk; }
    sh[0] = hdr->n_tracks;
    sh[8] = kOk;
    int n_order = 0;
    const bool first = hdr->n_tracks == 0 || n == 0;
    sh[7] = first;
    if (first) {
      for (int i = 0; i < n; ++i)
        if (dets[(size_t)i * 8 + 4] > cfg.init_track_thr) { s.order[n_order] = i; s.ids[n_order] = hdr->num_tracks++; ++n_order; }
      sh[1] = n_order;
    } else {
      int nc = 0;
      for (int i = 0; i < n; ++i) {
        const float* d = dets + (size_t)i * 8;
        const float area = (d[2] - d[0]) * (d[3] - d[1]);
        if (d[4] > cfg.obj_score_thr && area > 100.f) s.cand[nc++] = i;
      }
      int ncf = 0, nt = 0;
      for (int k = 0; k < hdr->n_tracks; ++k) {
        if (tracks[k].tentative) s.tentative[nt++] = k; else s.confirmed[ncf++] = k;
      }
      sh[2] = nc; sh[3] = ncf; sh[4] = nt; sh[6] = 0;
    }
  }
  __syncthreads();
  const bool first = sh[7];
  if (!first) {
    // 1. KF predict of the confirmed tracks (lanes over tracks)
    for (int q = lane; q < sh[3]; q += 64) {
      DTrack& t = tracks[s.confirmed[q]];
      if (t.last_frame != frame_id - 1) t.kf.mean[7] = 0;
      if (t.tracked) t.saved = t.kf;
      kf_predict(t.kf);
    }
    __syncthreads();
    // 2. confirmed, 3. tentative (OCM), 4. OCR on the still unmatched tracks
    for (int stage = 0; stage < 3; ++stage) {
      if (stage == 2) {
        if (lane == 0) {
          const int ntr = sh[0];
          for (int k = 0; k < ntr; ++k) s.is_matched[k] = 0;
          for (int i = 0; i < sh[6]; ++i) s.is_matched[s.matched_trk[i]] = 1;
          int nl = 0;
          for (int k = 0; k < ntr; ++k)
            if (!s.is_matched[k]) s.lost[nl++] = k;
          sh[5] = nl;
        }
        __syncthreads();
        if (sh[5] == 0) break;   // uniform
      }
      const int* tidx = stage == 0 ? s.confirmed : (stage == 1 ? s.tentative : s.lost);
      const int R = stage == 0 ? sh[3] : (stage == 1 ? sh[4] : sh[5]);
      assign_stage(cfg, tracks, tidx, R, s.cand, sh[2], dets, stage < 2, s, lane);
      if (lane == 0) {   // apply: matched pairs in column order, the rest stays in the pool
        int nm = sh[6], nr = 0;
        const int Cn = sh[2];
        for (int c = 0; c < Cn; ++c) {
          if (s.d2r[c] > -1) { s.matched_det[nm] = s.cand[c]; s.matched_trk[nm] = tidx[s.d2r[c]]; ++nm; }
          else s.rest[nr++] = s.cand[c];
        }
        for (int c = 0; c < nr; ++c) s.cand[c] = s.rest[c];
        sh[6] = nm; sh[2] = nr;
      }
      __syncthreads();
    }
    // 5. re-found tracks: smooth the KF over the gap (lanes over matches); unmatched tracks: mark lost
    if (lane == 0) {
      const int ntr = sh[0];
      for (int k = 0; k < ntr; ++k) s.is_matched[k] = 0;
      for (int i = 0; i < sh[6]; ++i) s.is_matched[s.matched_trk[i]] = 1;
    }
    __syncthreads();
    for (int i = lane; i < sh[6]; i += 64) {
      DTrack& t = tracks[s.matched_trk[i]];
      if (!t.tracked) online_smooth(t, dets + (size_t)s.matched_det[i] * 8);
    }
    for (int k = lane; k < sh[0]; k += 64)
      if (!s.is_matched[k]) { tracks[k].tracked = 0; push_obs(tracks[k], nullptr, cfg.vel_delta_t); }
    __syncthreads();
    if (lane == 0) {
      int n_order = 0;
      for (int i = 0; i < sh[6]; ++i) { s.order[n_order] = s.matched_det[i]; s.ids[n_order] = tracks[s.matched_trk[i]].id; ++n_order; }
      for (int c = 0; c < sh[2]; ++c) { s.order[n_order] = s.cand[c]; s.ids[n_order] = hdr->num_tracks++; ++n_order; }
      sh[1] = n_order;
    }
    __syncthreads();
  }
  // BaseTracker.update: feed every output row to its track.  Rows [0, n_matched) are the matched detections (their
  // track slot is known: no search by id), the rest start new tracks in fresh slots, in order.
  if (lane == 0) {
    int ntr = sh[0];
    const int n_order = sh[1];
    const int n_matched = first ? 0 : sh[6];
    for (int i = 0; i < n_order; ++i) {
      int k;
      if (i < n_matched) {
        k = s.matched_trk[i];
      } else if (ntr >= T) {
        sh[8] = kTrackOverflow;
        k = 0;
      } else {
        k = ntr++;
        tracks[k].id = -1;   // marks a slot to initialise below
      }
      s.tidx[i] = k;
    }
    if (sh[8] == kOk) hdr->n_tracks = ntr;
    sh[0] = ntr;
  }
  __syncthreads();
  if (sh[8] != kOk) {
    if (lane == 0) { status[b] = sh[8]; out_n[b] = 0; hdr->poisoned = sh[8]; }
    return;
  }
  for (int i = lane; i < sh[1]; i += 64) {
    const float* row = dets + (size_t)s.order[i] * 8;
    DTrack& t = tracks[s.tidx[i]];
    if (t.id < 0) init_track(t, s.ids[i], row, frame_id, cfg.vel_delta_t);
    else update_track(t, row, frame_id, cfg);
    for (int e = 0; e < 8; ++e) orow[(size_t)i * 8 + e] = row[e];
    oid[i] = s.ids[i];
  }
  __syncthreads();
  // pop the invalid tracks: order-preserving compaction; lane 0 decides the destinations, the wave moves the records
  if (lane == 0) {
    int w = 0;
    const int ntr = sh[0];
    for (int k = 0; k < ntr; ++k) {
      const DTrack& t = tracks[k];
      const bool drop = frame_id - t.last_frame >= cfg.num_frames_retain || (t.tentative && t.last_frame != frame_id);
      s.rest[k] = drop ? -1 : w;
      if (!drop) ++w;
    }
    hdr->n_tracks = w;
    out_n[b] = sh[1];
  }
  __syncthreads();
  {
    constexpr int WORDS = (int)(sizeof(DTrack) / 8);
    static_assert(sizeof(DTrack) % 8 == 0, "DTrack is moved as 8-byte words");
    const int ntr = sh[0];
    for (int k = 0; k < ntr; ++k) {          // ascending: a destination is never a slot that is still to be read
      const int w = s.rest[k];
      if (w < 0 || w == k) continue;         // uniform
      const unsigned long long* src = reinterpret_cast<const unsigned long long*>(&tracks[k]);
      unsigned long long* dst = reinterpret_cast<unsigned long long*>(&tracks[w]);
      for (int e = lane; e < WORDS; e += 64) dst[e] = src[e];
      __syncthreads();
    }
  }
}

}  // namespace ba
}  // namespace st

// ---- C ABI ------------------------------------------------------------------------------------------------------------
struct StBatchedTracker {
  st::ba::Cfg cfg;
  int batch;
  size_t state_stride, scratch_stride;
  size_t lap_lds_bytes;   // dynamic LDS for the assignment's working arrays; 0 = they stay in the L2-resident scratch
  int lds_set;
};

extern "C" int st_batched_tracker_create(const StTrackerConfig* cfg, int batch, int max_tracks, int max_dets,
                                         StBatchedTracker** out) {
  using namespace st;
  if (!cfg || !out) return set_error(ST_ERR_INVALID, "st_batched_tracker_create: null argument");
  ST_REQUIRE(cfg->struct_size == (int)sizeof(StTrackerConfig), "st_batched_tracker_create: struct_size mismatch");
  ST_REQUIRE(batch > 0 && max_tracks > 0 && max_dets > 0 && max_tracks + max_dets < (1 << 15),
             "st_batched_tracker_create: bad batch / capacities");
  ST_REQUIRE(cfg->vel_delta_t >= 0 && cfg->vel_delta_t + 1 <= ba::WINCAP && cfg->num_tentatives >= 1 &&
                 cfg->num_frames_retain >= 1,
             "st_batched_tracker_create: vel_delta_t must be in [0, %d]", ba::WINCAP - 1);
  auto t = std::make_unique<StBatchedTracker>();
  t->cfg.obj_score_thr = cfg->obj_score_thr; t->cfg.init_track_thr = cfg->init_track_thr;
  t->cfg.match_iou_thr = cfg->match_iou_thr; t->cfg.vel_consist_weight = cfg->vel_consist_weight;
  t->cfg.weight_iou_with_det_scores = cfg->weight_iou_with_det_scores; t->cfg.num_tentatives = cfg->num_tentatives;
  t->cfg.vel_delta_t = cfg->vel_delta_t; t->cfg.num_frames_retain = cfg->num_frames_retain;
  t->cfg.max_tracks = max_tracks; t->cfg.max_dets = max_dets;
  t->batch = batch;
  t->state_stride = ba::align8(sizeof(ba::SeqHeader) + sizeof(ba::DTrack) * (size_t)max_tracks);
  t->scratch_stride = ba::scratch_bytes(max_tracks, max_dets);
  {
    const size_t n = (size_t)max_tracks + max_dets;
    const size_t need = 2 * ba::align8(sizeof(double) * n) + 5 * ba::align8(sizeof(int) * n) + ba::align8(n);
    t->lap_lds_bytes = need <= 64 * 1024 ? need : 0;
    t->lds_set = 0;
  }
  *out = t.release();
  return ST_OK;
}

extern "C" int st_batched_tracker_destroy(StBatchedTracker* t) {
  delete t;
  return ST_OK;
}

extern "C" size_t st_batched_tracker_state_bytes(const StBatchedTracker* t) {
  return t ? t->state_stride * (size_t)t->batch : 0;
}
extern "C" size_t st_batched_tracker_scratch_bytes(const StBatchedTracker* t) {
  return t ? t->scratch_stride * (size_t)t->batch : 0;
}

extern "C" int st_batched_tracker_step(StBatchedTracker* t, const int32_t* frame_ids_dev, const float* dets_dev,
                                       const int32_t* counts_dev, void* state_dev, void* scratch_dev,
                                       float* out_rows_dev, int64_t* out_ids_dev, int32_t* out_counts_dev,
                                       int32_t* status_dev, st_stream_t stream) {
  using namespace st;
  if (!t) return set_error(ST_ERR_INVALID, "st_batched_tracker_step: null tracker");
  ST_REQUIRE(frame_ids_dev && dets_dev && counts_dev && state_dev && scratch_dev && out_rows_dev && out_ids_dev &&
                 out_counts_dev && status_dev,
             "st_batched_tracker_step: null pointer");
  ST_ENSURE_DYNAMIC_LDS(ba::assoc_step_kernel, t->lap_lds_bytes, t->lds_set);
  hipLaunchKernelGGL(ba::assoc_step_kernel, dim3(t->batch), dim3(64), t->lap_lds_bytes,
                     static_cast<hipStream_t>(stream), t->cfg, frame_ids_dev, dets_dev, counts_dev,
                     static_cast<char*>(state_dev), t->state_stride, static_cast<char*>(scratch_dev),
                     t->scratch_stride, out_rows_dev, reinterpret_cast<long long*>(out_ids_dev), out_counts_dev,
                     status_dev, t->lap_lds_bytes ? 1 : 0);
  ST_CHECK_HIP(hipGetLastError());
  return ST_OK;
}
