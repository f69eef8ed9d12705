// Host-side linear assignment for the CPU association step (north_star keeps the tracker on the CPU).
//
// Replaces the un-vendored `lap.lapjv(cost, extend_cost=True, cost_limit=c)` the reference calls at
// mmtrack/models/trackers/ocsort_tracker_disparity.py:260-261 and :312-313.  Track ids must be bit-exact, and when
// the optimum is not unique WHICH optimal assignment comes out is a property of the solver's visiting order, so
// this is the same dense Jonker-Volgenant procedure (column reduction with reduction transfer, two sweeps of
// augmenting row reduction, then shortest augmenting paths with the SCAN / TODO column partition) making the same
// comparisons in the same order on float64 as that solver; it is checked against the oracle's restatement
// (oracle/lapjv.py) on random, tie-heavy and enumerated <= 7x7 matrices (tests/test_cpu_tracker_oracle.py).
//
// The (R + C)^2 extension is lap's: the R x C costs top-left, cost_limit / 2 in both off-diagonal blocks, zeros
// bottom-right; a row matched into the padding is reported unmatched (-1).
#include <cmath>
#include <cstdint>
#include <vector>

#include "st_common.h"

namespace {

constexpr double kBig = 1000000.0;

struct Solver {
  int n;
  std::vector<double> c;   // n x n, row major
  std::vector<double> v;   // column prices
  std::vector<int> row_to_col, col_to_row, free_rows;
  int n_free = 0;

  double at(int i, int j) const { return c[(size_t)i * n + j]; }

  // Column reduction (every column takes its cheapest row, scanning columns from the last to the first so that a
  // row claimed twice keeps its highest-index column) + reduction transfer for rows claimed exactly once.
  void reduce_columns() {
    row_to_col.assign(n, -1);
    col_to_row.assign(n, 0);
    v.assign(n, kBig);
    for (int i = 0; i < n; ++i)
      for (int j = 0; j < n; ++j)
        if (at(i, j) < v[j]) { v[j] = at(i, j); col_to_row[j] = i; }
    std::vector<char> claimed_once(n, 1);
    for (int j = n - 1; j >= 0; --j) {
      const int i = col_to_row[j];
      if (row_to_col[i] < 0) {
        row_to_col[i] = j;
      } else {
        claimed_once[i] = 0;
        col_to_row[j] = -1;
      }
    }
    free_rows.assign(n, 0);
    n_free = 0;
    for (int i = 0; i < n; ++i) {
      if (row_to_col[i] < 0) {
        free_rows[n_free++] = i;
      } else if (claimed_once[i]) {
        const int j = row_to_col[i];
        double slack = kBig;
        for (int k = 0; k < n; ++k) {
          if (k == j) continue;
          const double r = at(i, k) - v[k];
          if (r < slack) slack = r;
        }
        v[j] -= slack;
      }
    }
  }

  // One sweep of augmenting row reduction over the current free rows.
  void reduce_rows() {
    const int todo = n_free;
    int cur = 0, kept = 0;
    long long sweeps = 0;
    while (cur < todo) {
      ++sweeps;
      const int i = free_rows[cur++];
      int best = 0, second = -1;
      double u1 = at(i, 0) - v[0], u2 = kBig;
      for (int j = 1; j < n; ++j) {
        const double r = at(i, j) - v[j];
        if (r < u2) {
          if (r >= u1) { u2 = r; second = j; }
          else { u2 = u1; u1 = r; second = best; best = j; }
        }
      }
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
          if (lowers) free_rows[--cur] = owner;   // revisit the displaced row at once
          else free_rows[kept++] = owner;
        }
      } else if (owner >= 0) {
        free_rows[kept++] = owner;
      }
      row_to_col[i] = best;
      col_to_row[best] = i;
    }
    n_free = kept;
  }

  // Shortest augmenting path from `start`; returns the free column it ends in and updates the prices.
  int shortest_path(int start, std::vector<int>& pred, std::vector<int>& cols, std::vector<double>& d) {
    int lo = 0, hi = 0, ready = 0, sink = -1;
    for (int j = 0; j < n; ++j) {
      cols[j] = j;
      pred[j] = start;
      d[j] = at(start, j) - v[j];
    }
    while (sink < 0) {
      if (lo == hi) {                      // SCAN list empty: move the columns of minimal d into it
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
      if (sink < 0) {                      // relax the TODO columns through the SCAN columns
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
        if (!found) { lo = l; hi = h; }   // (lo, hi) advance only when the SCAN list ran empty
      }
    }
    const double dmin = d[cols[lo]];
    for (int k = 0; k < ready; ++k) {
      const int j = cols[k];
      v[j] += d[j] - dmin;
    }
    return sink;
  }

  void augment() {
    std::vector<int> pred(n), cols(n);
    std::vector<double> d(n);
    for (int f = 0; f < n_free; ++f) {
      const int start = free_rows[f];
      int j = shortest_path(start, pred, cols, d);
      int i = -1;
      while (i != start) {
        i = pred[j];
        col_to_row[j] = i;
        const int prev = row_to_col[i];
        row_to_col[i] = j;
        j = prev;
      }
    }
  }

  void solve() {
    reduce_columns();
    for (int pass = 0; pass < 2 && n_free > 0; ++pass) reduce_rows();
    if (n_free > 0) augment();
  }
};

}  // namespace

// cost: row-major [n_rows][n_cols] float64.  x_out[n_rows]: column matched to row i or -1; y_out[n_cols]: row matched
// to column j or -1.  NaN costs (a NaN box out of extract_depth's empty-segment branch) are undefined behaviour in
// the upstream solver; here they are made unmatchable (1e6 > any cost_limit).
extern "C" int st_lapjv_extended(const double* cost, int n_rows, int n_cols, double cost_limit, int32_t* x_out,
                                 int32_t* y_out) {
  using namespace st;
  ST_REQUIRE(n_rows >= 0 && n_cols >= 0 && (long long)n_rows + n_cols < (1 << 15), "st_lapjv_extended: bad shape");
  ST_REQUIRE(std::isfinite(cost_limit), "st_lapjv_extended: cost_limit must be finite");
  if (n_rows == 0 || n_cols == 0) {
    for (int i = 0; i < n_rows; ++i) x_out[i] = -1;
    for (int j = 0; j < n_cols; ++j) y_out[j] = -1;
    return ST_OK;
  }
  ST_REQUIRE(cost && x_out && y_out, "st_lapjv_extended: null pointer");
  Solver s;
  s.n = n_rows + n_cols;
  s.c.assign((size_t)s.n * s.n, cost_limit / 2.0);
  for (int i = 0; i < n_rows; ++i)
    for (int j = 0; j < n_cols; ++j) {
      const double cij = cost[(size_t)i * n_cols + j];
      s.c[(size_t)i * s.n + j] = std::isnan(cij) ? 1e6 : cij;
    }
  for (int i = n_rows; i < s.n; ++i)
    for (int j = n_cols; j < s.n; ++j) s.c[(size_t)i * s.n + j] = 0.0;
  s.solve();
  for (int i = 0; i < n_rows; ++i) x_out[i] = s.row_to_col[i] < n_cols ? s.row_to_col[i] : -1;
  for (int j = 0; j < n_cols; ++j) y_out[j] = s.col_to_row[j] < n_rows ? s.col_to_row[j] : -1;
  return ST_OK;
}
