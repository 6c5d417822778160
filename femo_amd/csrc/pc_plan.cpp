// Host-side plan of the auxiliary-lattice preconditioner (bpx.hip): lattice hierarchy over the
// bounding box, packed lattice coordinates of the owned vertices and their (brick, bin) sort.
// No GPU call in here; femo_pc_plan_host exposes it to the CPU test-suite, which compares it with
// the NumPy restatement (oracle/bpx_oracle.py).
#include <algorithm>
#include <cmath>
#include <cstdlib>

#include "femo_internal.h"

int femo_pc_make_plan(int dim, int64_t n_rows, const double* x, const double* lo, const double* hi,
                      int64_t n_vert_global, double spacing, FemoPcPlan& P) {
  FEMO_REQUIRE(dim == 2 || dim == 3, "preconditioner lattice: dim must be 2 or 3");
  FEMO_REQUIRE(n_rows >= 0 && (n_rows == 0 || x != nullptr) && lo && hi, "bad argument");
  P = FemoPcPlan();
  P.dim = dim;
  double ext[3] = {0, 0, 0}, ext_max = 0.0, vol = 1.0;
  for (int k = 0; k < dim; ++k) {
    P.lo[k] = lo[k]; P.hi[k] = hi[k];
    ext[k] = hi[k] - lo[k];
    FEMO_REQUIRE(ext[k] > 0.0, "degenerate bounding box along axis %d", k);
    ext_max = std::max(ext_max, ext[k]);
    vol *= ext[k];
  }
  // mesh size estimate and the finest lattice: spacing ~ 2 h, bins = m0 * 2^(levels-1) with m0 in {2, 3}
  const double n_glob = (double)std::max<int64_t>(n_vert_global, 1);
  const double h = std::pow(vol / n_glob, 1.0 / dim);
  const double target = std::max(2.0, ext_max / (spacing * h));
  int best_m0 = 2, best_lv = 1;
  double best = 1e300;
  // 3-D: the 8-byte packed coordinates hold 9 bits of bin per axis, so the finest lattice is capped at 384 = 3 * 2^7 bins
  // on its longest axis; a finer mesh (beyond ~800^3 vertices) gets a coarser mesh-to-lattice ratio instead of an error
  const int max_bins = dim == 3 ? (1 << (FEMO_PK3_FIELD - FEMO_PK3_BITS)) - 1 : (1 << (32 - FEMO_PK_BITS)) - 1;
  for (int m0 = 2; m0 <= 3; ++m0)
    for (int lv = 1; lv <= 12; ++lv) {
      if ((m0 << (lv - 1)) > max_bins) continue;
      const double score = std::fabs(std::log(m0 * std::ldexp(1.0, lv - 1) / target));
      if (score < best) { best = score; best_m0 = m0; best_lv = lv; }
    }
  P.n_levels = best_lv;
  for (int l = 0; l < best_lv; ++l) {
    P.H[l] = ext_max / (double)(best_m0 << l);
    P.nodes[l] = 1;
    for (int k = 0; k < 3; ++k) {
      // same number of halvings on every axis: bins on the coarsest level proportional to the extent
      P.n[l][k] = k < dim ? std::max(1, (int)std::lround(best_m0 * ext[k] / ext_max)) << l : 0;
      P.nodes[l] *= P.n[l][k] + 1;
    }
  }
  // owned vertices: packed lattice coordinates, then a counting sort by (brick, bin)
  const int D = dim, B = D == 3 ? 4 : 8, L = best_lv - 1;
  const int* nF = P.n[L];
  for (int k = 0; k < D; ++k)
    FEMO_REQUIRE(nF[k] < (D == 3 ? (1 << (FEMO_PK3_FIELD - FEMO_PK3_BITS)) : (1 << (32 - FEMO_PK_BITS))), "preconditioner lattice too fine for packed coordinates");
  int nbr[3] = {1, 1, 1};
  double inv_h[3] = {0, 0, 0};
  for (int k = 0; k < D; ++k) { nbr[k] = (nF[k] + B - 1) / B; inv_h[k] = nF[k] / ext[k]; }
  const int64_t n_all = (int64_t)nbr[0] * nbr[1] * nbr[2];
  FEMO_REQUIRE(n_all < (int64_t(1) << 24), "preconditioner lattice too fine");
  const int64_t nr = n_rows;
  constexpr int W = FEMO_PK_WORDS;
  P.pk.assign((size_t)std::max<int64_t>(nr * W, 1), 0u);
  std::vector<int32_t> key((size_t)std::max<int64_t>(nr, 1));     // brick id * 64 + bin inside the brick
  std::vector<int64_t> count((size_t)n_all * 64 + 1, 0);
  const int frac_bits = femo_pk_frac_bits(D);
  const uint32_t pk_mask = (1u << frac_bits) - 1u;
  for (int64_t v = 0; v < nr; ++v) {
    int64_t brick = 0, bstride = 1;
    int local = 0, lstride = 1;
    uint64_t word3 = 0;
    for (int k = 0; k < D; ++k) {
      const double gk = (x[v * D + k] - lo[k]) * inv_h[k];
      int b = (int)std::floor(gk);
      b = b < 0 ? 0 : (b > nF[k] - 1 ? nF[k] - 1 : b);
      double t = gk - b;
      t = t < 0.0 ? 0.0 : (t > 1.0 ? 1.0 : t);
      uint32_t tq = (uint32_t)(t * (double)(1u << frac_bits) + 0.5);
      if (tq > pk_mask) tq = pk_mask;
      const uint32_t field = ((uint32_t)b << frac_bits) | tq;
      if (D == 3) word3 |= (uint64_t)field << (FEMO_PK3_FIELD * k);
      else P.pk[v * W + k] = field;
      brick += (int64_t)(b / B) * bstride;
      bstride *= nbr[k];
      local += (b % B) * lstride;
      lstride *= B;
    }
    if (D == 3) { P.pk[v * W] = (uint32_t)word3; P.pk[v * W + 1] = (uint32_t)(word3 >> 32); }
    key[v] = (int32_t)(brick * 64 + local);
    ++count[(size_t)key[v] + 1];
  }
  for (size_t i = 1; i < count.size(); ++i) count[i] += count[i - 1];   // count[key] = first sorted position
  P.brick_ptr.assign(1, 0);
  for (int64_t id = 0; id < n_all; ++id) {
    const int64_t first = count[(size_t)id * 64], last = count[(size_t)id * 64 + 64];
    if (last == first) continue;
    FEMO_REQUIRE(last - first < (int64_t(1) << 32), "brick too large");
    P.brick_ptr.push_back(last);
    P.brick_base.push_back((int32_t)(id % nbr[0]) * B);
    P.brick_base.push_back((int32_t)((id / nbr[0]) % nbr[1]) * B);
    P.brick_base.push_back((int32_t)(id / ((int64_t)nbr[0] * nbr[1])) * B);
    for (int q = 0; q <= 64; ++q) P.bin_ptr.push_back((uint32_t)(count[(size_t)id * 64 + q] - first));
  }
  P.n_bricks = (int64_t)P.brick_ptr.size() - 1;
  P.perm.assign((size_t)std::max<int64_t>(nr, 1), 0);
  P.pk_sorted.assign((size_t)std::max<int64_t>(nr * W, 1), 0u);
  {
    std::vector<int64_t> fill(count.begin(), count.end() - 1);
    for (int64_t v = 0; v < nr; ++v) {      // stable: vertices of a bin stay in index order
      const int64_t at = fill[(size_t)key[v]]++;
      P.perm[at] = (int32_t)v;
      for (int k = 0; k < W; ++k) P.pk_sorted[at * W + k] = P.pk[v * W + k];
    }
  }
  if (P.brick_base.empty()) P.brick_base.assign(3, 0);
  if (P.bin_ptr.empty()) P.bin_ptr.assign(65, 0);
  return 0;
}

double femo_pc_spacing() {
  // FEMO_BPX_SPACING: finest lattice spacing in units of the mesh size (tuning knob, default 2)
  double spacing = 2.0;
  if (const char* e = getenv("FEMO_BPX_SPACING")) { const double v = atof(e); if (v >= 1.0 && v <= 8.0) spacing = v; }
  return spacing;
}

extern "C" int femo_pc_plan_host(int dim, int64_t n_rows, const double* x, const double* lo, const double* hi,
                                 int64_t n_vert_global, int32_t* n_levels, int32_t* bins, int64_t* n_bricks,
                                 uint32_t* pk, int32_t* perm, int64_t* brick_ptr, int32_t* brick_base, uint32_t* bin_ptr) {
  FEMO_REQUIRE(n_levels && n_bricks, "null argument");
  FemoPcPlan P;
  FEMO_TRY(femo_pc_make_plan(dim, n_rows, x, lo, hi, n_vert_global, femo_pc_spacing(), P));
  *n_levels = P.n_levels;
  *n_bricks = P.n_bricks;
  if (bins)
    for (int l = 0; l < P.n_levels; ++l)
      for (int k = 0; k < 3; ++k) bins[l * 3 + k] = P.n[l][k];
  if (pk) std::copy(P.pk.begin(), P.pk.begin() + n_rows * FEMO_PK_WORDS, pk);
  if (perm) std::copy(P.perm.begin(), P.perm.begin() + n_rows, perm);
  if (brick_ptr) std::copy(P.brick_ptr.begin(), P.brick_ptr.end(), brick_ptr);
  if (brick_base) std::copy(P.brick_base.begin(), P.brick_base.begin() + 3 * P.n_bricks, brick_base);
  if (bin_ptr) std::copy(P.bin_ptr.begin(), P.bin_ptr.begin() + 65 * P.n_bricks, bin_ptr);
  return 0;
}
