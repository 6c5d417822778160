// Host-side mesh topology: vertex->cell incidence and the P1 sparsity pattern,
// both emitted directly in the SELL-64 layouts the device kernels stream.
//
// Replaces dolfinx's sparsity-pattern / create_matrix step [ext] that the
// reference reaches through utils_dolfinx.py:385 (create_matrix) and every
// assemble_matrix call (utils_dolfinx.py:185,193).  Runs once per mesh, on the
// host cores, so it can be tested without a GPU (femo_topology_build_host).
#include <algorithm>
#include <atomic>
#include <climits>
#include <cstdarg>
#include <thread>

#include "femo_internal.h"

static thread_local std::string g_err;

void femo_set_error(const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_err = buf;
}

extern "C" const char* femo_last_error(void) { return g_err.c_str(); }

namespace {

template <class F>
void parallel_for(int64_t n, F&& body) {
  unsigned nt = std::thread::hardware_concurrency();
  if (nt == 0) nt = 1;
  if (nt > 16) nt = 16;     // containers usually grant a CPU quota well below the machine's core count; threads beyond it
                            // only get the whole process throttled (measured: 2.3 s of stalls in a 22 s bench run)
  if (n < 1 << 14) nt = 1;
  if (nt == 1) {
    body(0, n);
    return;
  }
  std::vector<std::thread> th;
  const int64_t chunk = (n + nt - 1) / nt;
  for (unsigned t = 0; t < nt; ++t) {
    const int64_t lo = t * chunk, hi = std::min<int64_t>(n, lo + chunk);
    if (lo >= hi) break;
    th.emplace_back([=, &body] { body(lo, hi); });
  }
  for (auto& t : th) t.join();
}

}  // namespace

int femo_build_topology(int tdim, int64_t n_vert, int64_t n_rows, int64_t n_cell,
                        const int32_t* conn, FemoTopology& T) {
  FEMO_REQUIRE(tdim == 2 || tdim == 3, "tdim must be 2 or 3 (got %d)", tdim);
  FEMO_REQUIRE(n_rows >= 0 && n_rows <= n_vert, "n_rows must be in [0, n_vert]");
  FEMO_REQUIRE(n_cell < (int64_t(1) << 29), "n_cell too large for packed visit words");
  FEMO_REQUIRE(n_vert < (int64_t(1) << 31), "n_vert exceeds int32 indices");
  const int nv = tdim + 1;
  T.tdim = tdim;
  T.n_vert = n_vert;
  T.n_rows = n_rows;
  T.n_cell = n_cell;
  T.n_slices = (n_rows + FEMO_WAVE - 1) / FEMO_WAVE;
  const int64_t n_pad = T.n_slices * FEMO_WAVE;

  // validate connectivity
  {
    std::atomic<int> bad{0};
    parallel_for(n_cell * nv, [&](int64_t lo, int64_t hi) {
      for (int64_t e = lo; e < hi; ++e)
        if (conn[e] < 0 || conn[e] >= n_vert) bad.store(1, std::memory_order_relaxed);
    });
    FEMO_REQUIRE(!bad.load(), "connectivity entry out of range [0, n_vert)");
  }

  // 1. valence of owned vertices
  std::vector<int32_t> deg(n_pad, 0);
  parallel_for(n_cell, [&](int64_t lo, int64_t hi) {
    for (int64_t c = lo; c < hi; ++c)
      for (int a = 0; a < nv; ++a) {
        const int32_t v = conn[c * nv + a];
        if (v < n_rows) __atomic_fetch_add(&deg[v], 1, __ATOMIC_RELAXED);
      }
  });
  // 2. CSR incidence, sorted by cell index per vertex
  std::vector<int64_t> off(n_pad + 1, 0);
  for (int64_t v = 0; v < n_pad; ++v) off[v + 1] = off[v] + deg[v];
  std::vector<int32_t> inc(off[n_pad]);
  {
    std::vector<int32_t> cur(n_pad, 0);
    parallel_for(n_cell, [&](int64_t lo, int64_t hi) {
      for (int64_t c = lo; c < hi; ++c)
        for (int a = 0; a < nv; ++a) {
          const int32_t v = conn[c * nv + a];
          if (v < n_rows) {
            const int32_t s = __atomic_fetch_add(&cur[v], 1, __ATOMIC_RELAXED);
            inc[off[v] + s] = (int32_t)((c << 2) | a);
          }
        }
    });
  }
  int max_val = 0;
  for (int64_t v = 0; v < n_rows; ++v) max_val = std::max(max_val, deg[v]);
  T.max_valence = max_val;
  parallel_for(n_rows, [&](int64_t lo, int64_t hi) {
    for (int64_t v = lo; v < hi; ++v) std::sort(inc.begin() + off[v], inc.begin() + off[v + 1]);
  });

  // 3. sorted unique neighbours per row
  T.rowlen.assign(n_pad, 0);
  std::vector<int64_t> noff(n_pad + 1, 0);
  {
    // upper bound tdim * valence per row, compacted afterwards
    std::vector<int64_t> uoff(n_pad + 1, 0);
    for (int64_t v = 0; v < n_pad; ++v) uoff[v + 1] = uoff[v] + (int64_t)deg[v] * tdim;
    std::vector<int32_t> tmp(uoff[n_pad]);
    parallel_for(n_rows, [&](int64_t lo, int64_t hi) {
      for (int64_t v = lo; v < hi; ++v) {
        int32_t* w = tmp.data() + uoff[v];
        int n = 0;
        for (int64_t e = off[v]; e < off[v + 1]; ++e) {
          const int64_t c = inc[e] >> 2;
          const int a = inc[e] & 3;
          for (int b = 0; b < nv; ++b)
            if (b != a) w[n++] = conn[c * nv + b];
        }
        std::sort(w, w + n);
        n = (int)(std::unique(w, w + n) - w);
        // a degenerate cell could repeat the row's own vertex
        int m = 0;
        for (int i = 0; i < n; ++i)
          if (w[i] != v) w[m++] = w[i];
        T.rowlen[v] = m;
      }
    });
    for (int64_t v = 0; v < n_pad; ++v) noff[v + 1] = noff[v] + T.rowlen[v];
    int max_len = 0;
    for (int64_t v = 0; v < n_rows; ++v) max_len = std::max(max_len, T.rowlen[v]);
    T.max_rowlen = max_len;
    FEMO_REQUIRE(max_len <= 254, "row with %d off-diagonal entries exceeds the 8-bit slot map",
                 max_len);
    T.nnz = noff[n_pad] + n_rows;

    // 3.5 regular slices, completed.  A slice is "regular" when every lane's k-th column is row + delta[k]: the SpMV
    // then fetches no column indices and reads x as 64 consecutive doubles per k.  Round 1-2 required the 64 rows to
    // have identical delta lists, which on the lexicographically numbered cube fails for every slice that contains a
    // vertex of the domain boundary (fewer neighbours): 35 % of the slices at n = 215.  Round 3: the slice's delta
    // set is the UNION of its rows' deltas; a row that lacks one of them gets a structural zero there (column
    // row + delta, value 0: no cell couples the two vertices, the assembly leaves the slot at zero).  Accepted when
    // the union is no larger than the longest row of the mesh (so no kernel capacity changes, and unstructured
    // numberings, whose union explodes, are rejected after a few rows) and every row + delta is a valid vertex.
    // T.real marks the true couplings (bit k of row v); the CSR exports skip the structural zeros.
    T.sdelta_stride = (T.max_rowlen + 1) & ~1;
    if (T.sdelta_stride < 2) T.sdelta_stride = 2;
    T.sdelta.assign((size_t)T.n_slices * T.sdelta_stride, INT32_MIN);
    std::vector<int32_t> slen(T.n_slices, -1);               // |delta set| of the regular slices, -1: irregular
    const int dcap = std::min(T.max_rowlen, 32);             // T.real is a 32-bit mask
    {
      std::atomic<int64_t> nreg{0};
      parallel_for(T.n_slices, [&](int64_t lo, int64_t hi) {
        int64_t local = 0;
        std::vector<int64_t> D;
        for (int64_t s = lo; s < hi; ++s) {
          if ((s + 1) * FEMO_WAVE > n_rows || dcap < 1) continue;       // partial last slice: general path
          D.clear();
          bool ok = true;
          for (int l = 0; l < FEMO_WAVE && ok; ++l) {
            const int64_t v = s * FEMO_WAVE + l;
            const int32_t* w = tmp.data() + uoff[v];
            for (int k = 0; k < T.rowlen[v]; ++k) {
              const int64_t d = (int64_t)w[k] - v;
              auto it = std::lower_bound(D.begin(), D.end(), d);
              if (it == D.end() || *it != d) {
                if ((int)D.size() >= dcap) { ok = false; break; }
                D.insert(it, d);
              }
            }
          }
          if (!ok || D.empty()) continue;
          const int64_t v0 = s * FEMO_WAVE, v1 = v0 + FEMO_WAVE - 1;
          if (v0 + D.front() < 0 || v1 + D.back() >= n_vert) continue;  // a completed column would leave [0, n_vert)
          for (size_t k = 0; k < D.size(); ++k) T.sdelta[s * T.sdelta_stride + k] = (int32_t)D[k];
          for (int k = (int)D.size(); k < T.sdelta_stride; ++k) T.sdelta[s * T.sdelta_stride + k] = 0;   // padding slot: own row, value 0
          slen[s] = (int)D.size();
          ++local;
        }
        nreg += local;
      });
      T.n_regular = nreg.load();
    }
    T.real.assign(n_pad, 0xFFFFFFFFu);

    // 4. SELL layouts
    T.vptr.assign(T.n_slices + 1, 0);
    T.mptr.assign(T.n_slices + 1, 0);
    for (int64_t s = 0; s < T.n_slices; ++s) {
      int wv = 0, wm = 0;
      for (int l = 0; l < FEMO_WAVE; ++l) {
        wv = std::max(wv, deg[s * FEMO_WAVE + l]);
        wm = std::max(wm, T.rowlen[s * FEMO_WAVE + l]);
      }
      if (slen[s] >= 0) wm = slen[s];
      wm = (wm + 1) & ~1;
      T.vptr[s + 1] = T.vptr[s] + (int64_t)wv * FEMO_WAVE;
      T.mptr[s + 1] = T.mptr[s] + (int64_t)wm * FEMO_WAVE;
    }
    T.visit_cell.assign(T.vptr[T.n_slices], -1);
    T.visit_slots.assign(T.vptr[T.n_slices], 0xFFFFFFFFu);
    T.cols.resize(T.mptr[T.n_slices]);
    parallel_for(T.n_slices, [&](int64_t lo, int64_t hi) {
      for (int64_t s = lo; s < hi; ++s) {
        const int wm = (int)((T.mptr[s + 1] - T.mptr[s]) / FEMO_WAVE);
        const bool reg = slen[s] >= 0;
        const int32_t* dl = T.sdelta.data() + s * T.sdelta_stride;
        for (int l = 0; l < FEMO_WAVE; ++l) {
          const int64_t v = s * FEMO_WAVE + l;
          const int32_t* w = tmp.data() + uoff[v];
          const int len_true = T.rowlen[v];
          const int len = reg ? slen[s] : len_true;
          const int32_t self = (int32_t)std::min<int64_t>(v, n_vert - 1);
          if (reg) {
            uint32_t real = 0;
            for (int k = 0; k < wm; ++k) {
              const int32_t c = k < len ? (int32_t)(v + dl[k]) : self;
              T.cols[femo_sell_index(T.mptr[s], k, l)] = c;
              if (k < len && std::binary_search(w, w + len_true, c)) real |= 1u << k;
            }
            T.real[v] = real;
          } else {
            for (int k = 0; k < wm; ++k)
              T.cols[femo_sell_index(T.mptr[s], k, l)] = k < len ? w[k] : self;
          }
          for (int64_t e = off[v]; e < off[v + 1]; ++e) {
            const int64_t c = inc[e] >> 2;
            const int a = inc[e] & 3;
            // byte j = slot (in this row) of the j-th vertex of the cell other than the visiting one
            uint32_t slots = 0xFFFFFFFFu;
            int j = 0;
            for (int b = 0; b < nv; ++b) {
              if (b == a) continue;
              const int32_t nb = conn[c * nv + b];
              uint32_t byte = 0xFFu;
              if (reg) {
                const int32_t d = (int32_t)((int64_t)nb - v);
                const int pos = (int)(std::lower_bound(dl, dl + len, d) - dl);
                if (pos < len && dl[pos] == d && nb != v) byte = (uint32_t)pos;
              } else {
                const int pos = (int)(std::lower_bound(w, w + len, nb) - w);
                if (pos < len && w[pos] == nb) byte = (uint32_t)pos;
              }
              slots = (slots & ~(0xFFu << (8 * j))) | (byte << (8 * j));
              ++j;
            }
            const int64_t idx = T.vptr[s] + (e - off[v]) * FEMO_WAVE + l;
            T.visit_cell[idx] = inc[e];
            T.visit_slots[idx] = slots;
          }
          if (reg) T.rowlen[v] = len;                         // stored entries of the row (structural zeros included)
        }
      }
    });
    // 6. short slices: irregular slices all of whose stored columns lie within +-32767 of their row keep a second copy of
    // the column indices as 16-bit deltas (pair-interleaved like `cols`): the SpMV then fetches 2 instead of 4 bytes per
    // entry for them.  A bandwidth-reducing numbering (Morton curve: ~2/3 of the slices at 10 M vertices; any banded
    // numbering of a smaller mesh: all of them) qualifies, a random one does not.  Flag: sdelta[s * stride + 1] = 1.
    T.cols16.assign(T.cols.size(), 0);
    std::atomic<int64_t> nshort{0};
    parallel_for(T.n_slices, [&](int64_t lo, int64_t hi) {
      int64_t local = 0;
      for (int64_t s = lo; s < hi; ++s) {
        if (slen[s] >= 0) continue;
        const int wm = (int)((T.mptr[s + 1] - T.mptr[s]) / FEMO_WAVE);
        if (wm == 0) continue;
        bool ok = true;
        for (int l = 0; l < FEMO_WAVE && ok; ++l) {
          const int64_t v = s * FEMO_WAVE + l;
          for (int k = 0; k < wm; ++k) {
            const int64_t d = (int64_t)T.cols[femo_sell_index(T.mptr[s], k, l)] - v;
            if (d < -32767 || d > 32767) { ok = false; break; }
          }
        }
        if (!ok) continue;
        for (int l = 0; l < FEMO_WAVE; ++l) {
          const int64_t v = s * FEMO_WAVE + l;
          for (int k = 0; k < wm; ++k) {
            const int64_t e = femo_sell_index(T.mptr[s], k, l);
            T.cols16[e] = (int16_t)((int64_t)T.cols[e] - v);
          }
        }
        T.sdelta[s * T.sdelta_stride + 1] = 1;
        ++local;
      }
      nshort += local;
    });
    T.n_short = nshort.load();
  }
  return 0;
}

void femo_topology_csr(const FemoTopology& T, int64_t* rowptr, int32_t* col) {
  rowptr[0] = 0;
  for (int64_t v = 0; v < T.n_rows; ++v) {
    int n = 0;
    for (int k = 0; k < T.rowlen[v]; ++k) n += (k >= 32 || ((T.real[v] >> k) & 1u)) ? 1 : 0;
    rowptr[v + 1] = rowptr[v] + n + 1;
  }
  if (!col) return;
  parallel_for(T.n_rows, [&](int64_t lo, int64_t hi) {
    for (int64_t v = lo; v < hi; ++v) {
      const int64_t s = v / FEMO_WAVE;
      const int l = (int)(v % FEMO_WAVE);
      int32_t* out = col + rowptr[v];
      bool placed = false;
      int n = 0;
      for (int k = 0; k < T.rowlen[v]; ++k) {
        if (k < 32 && !((T.real[v] >> k) & 1u)) continue;     // structural zero of a completed regular slice
        const int32_t c = T.cols[femo_sell_index(T.mptr[s], k, l)];
        if (!placed && c > v) {
          out[n++] = (int32_t)v;
          placed = true;
        }
        out[n++] = c;
      }
      if (!placed) out[n++] = (int32_t)v;
    }
  });
}

extern "C" int femo_topology_build_host(int tdim, int64_t n_vert, int64_t n_rows, int64_t n_cell,
                                        const int32_t* conn, int64_t info[FEMO_MESH_INFO_COUNT],
                                        int64_t* rowptr, int32_t* col) {
  FEMO_REQUIRE(conn != nullptr && info != nullptr, "null argument");
  FemoTopology T;
  FEMO_TRY(femo_build_topology(tdim, n_vert, n_rows, n_cell, conn, T));
  info[FEMO_MESH_TDIM] = tdim;
  info[FEMO_MESH_N_VERT] = n_vert;
  info[FEMO_MESH_N_ROWS] = n_rows;
  info[FEMO_MESH_N_CELL] = n_cell;
  info[FEMO_MESH_NNZ] = T.nnz;
  info[FEMO_MESH_SELL_ENTRIES] = T.mptr[T.n_slices];
  info[FEMO_MESH_MAX_ROWLEN] = T.max_rowlen;
  info[FEMO_MESH_MAX_VALENCE] = T.max_valence;
  info[FEMO_MESH_N_SLICES] = T.n_slices;
  info[FEMO_MESH_VISIT_ENTRIES] = T.vptr[T.n_slices];
  info[FEMO_MESH_REGULAR_SLICES] = T.n_regular;
  info[FEMO_MESH_SHORT_SLICES] = T.n_short;
  if (rowptr) femo_topology_csr(T, rowptr, col);
  return 0;
}
