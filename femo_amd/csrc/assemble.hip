// Element-local P1 quadrature kernels for gfx950: residual, dR/du, dR/df,
// functional and its partials.
//
// Design: "owner computes".  One lane owns one matrix row /
// vector entry (= one vertex) and walks the cells incident to it through the
// SELL-64 vertex->cell incidence, so consecutive lanes stream consecutive
// words of the incidence (coalesced), element geometry is recomputed in
// registers from gathered coordinates (L2-resident), and every output word is
// written exactly once by its owner: no fp64 atomics, bitwise reproducible,
// and K[i][j] == K[j][i] bitwise because both rows evaluate vol * g_a.g_b from
// the same canonical cell ordering in the same (sorted) cell order.
//
// Replaces, per call site of the reference:
//   residual  -> utils_dolfinx.py:175-179 assembleVector   (state_model.py:85)
//   jacobian  -> utils_dolfinx.py:181-187 assembleMatrix   (state_model.py:132)
//                utils_dolfinx.py:189-202 assembleSystem   (state_model.py:149)
//   dRdf      -> state_model.py:141 assembleMatrix(computePartials(res, f))
//   functional value / partials -> output_model.py:69-87
#include "femo_internal.h"

namespace {

template <int D>
struct CellGeom {
  double vol;
  double g[D + 1][D];
};

template <int D>
__device__ __forceinline__ void load_conn(const int32_t* __restrict__ conn, int64_t c, int32_t v[D + 1]) {
  if constexpr (D == 3) {
    const int4 q = *reinterpret_cast<const int4*>(conn + c * 4);
    v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
  } else {
    v[0] = conn[c * 3 + 0]; v[1] = conn[c * 3 + 1]; v[2] = conn[c * 3 + 2];
  }
}

// The vertices of a visited cell without touching `conn`: local vertex a is the visiting row
// itself, the others sit in the row's own column list at the positions the slot bytes name
// (row + delta[pos] on regular slices).  `conn` rows of neighbouring lanes are 96 B apart (one
// cache line per lane), the row's columns are the lane's own 14 contiguous words.
template <int D>
__device__ __forceinline__ void row_cell_vertices(int64_t row, int lane, int a, uint32_t slots, bool regular,
                                                  const int32_t* __restrict__ dl, const int32_t* __restrict__ cols,
                                                  int64_t mb, int32_t v[D + 1]) {
  // branch-free per lane: the own slot byte is 0xFF, read position 0 instead and discard the value;
  // `regular` is uniform over the wave (a property of the slice)
  int pos[D + 1];
#pragma unroll
  for (int b = 0; b <= D; ++b) {
    const int p = (slots >> (8 * b)) & 0xFF;
    pos[b] = (b == a) ? 0 : p;
  }
  if (regular) {
#pragma unroll
    for (int b = 0; b <= D; ++b) v[b] = (int32_t)row + dl[pos[b]];
  } else {
#pragma unroll
    for (int b = 0; b <= D; ++b) v[b] = cols[mb + (int64_t)(pos[b] >> 1) * 128 + lane * 2 + (pos[b] & 1)];
  }
#pragma unroll
  for (int b = 0; b <= D; ++b) v[b] = (b == a) ? (int32_t)row : v[b];
}

// Geometry of a cell from its vertex coordinates p[a][k] (a = local vertex, canonical order)
template <int D>
__device__ __forceinline__ void cell_geom_p(const double (&p)[D + 1][D], CellGeom<D>& G) {
  if constexpr (D == 3) {
    double e1[3], e2[3], e3[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      e1[k] = p[1][k] - p[0][k];
      e2[k] = p[2][k] - p[0][k];
      e3[k] = p[3][k] - p[0][k];
    }
    double c1[3] = {e2[1] * e3[2] - e2[2] * e3[1], e2[2] * e3[0] - e2[0] * e3[2], e2[0] * e3[1] - e2[1] * e3[0]};
    double c2[3] = {e3[1] * e1[2] - e3[2] * e1[1], e3[2] * e1[0] - e3[0] * e1[2], e3[0] * e1[1] - e3[1] * e1[0]};
    double c3[3] = {e1[1] * e2[2] - e1[2] * e2[1], e1[2] * e2[0] - e1[0] * e2[2], e1[0] * e2[1] - e1[1] * e2[0]};
    const double det = e1[0] * c1[0] + e1[1] * c1[1] + e1[2] * c1[2];
    const double inv = 1.0 / det;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      G.g[1][k] = c1[k] * inv;
      G.g[2][k] = c2[k] * inv;
      G.g[3][k] = c3[k] * inv;
      G.g[0][k] = -(G.g[1][k] + G.g[2][k] + G.g[3][k]);
    }
    G.vol = fabs(det) * (1.0 / 6.0);
  } else {
    const double a = p[1][0] - p[0][0], b = p[1][1] - p[0][1];
    const double c = p[2][0] - p[0][0], d = p[2][1] - p[0][1];
    const double det = a * d - b * c;
    const double inv = 1.0 / det;
    G.g[1][0] = d * inv;  G.g[1][1] = -c * inv;
    G.g[2][0] = -b * inv; G.g[2][1] = a * inv;
    G.g[0][0] = -(G.g[1][0] + G.g[2][0]);
    G.g[0][1] = -(G.g[1][1] + G.g[2][1]);
    G.vol = fabs(det) * 0.5;
  }
}

template <int D>
__device__ __forceinline__ void cell_geom(const double* __restrict__ x, const int32_t v[D + 1], CellGeom<D>& G) {
  double p[D + 1][D];
#pragma unroll
  for (int a = 0; a <= D; ++a) {
    const double* q = x + (int64_t)v[a] * D;
#pragma unroll
    for (int k = 0; k < D; ++k) p[a][k] = q[k];
  }
  cell_geom_p<D>(p, G);
}

// cell_geom for the row walks: the visiting lane owns local vertex a and already holds its
// coordinates, so only the D other vertices are gathered (the walks are bound by the number of
// scattered loads per visited cell).  Values and vertex order are those of cell_geom: same bits.
template <int D>
__device__ __forceinline__ void load_other_vertices(const double* __restrict__ x, const int32_t v[D + 1], int a,
                                                    double (&o)[D][D]) {
#pragma unroll
  for (int j = 0; j < D; ++j) {
    const int32_t vj = (a <= j) ? v[j + 1] : v[j];       // the j-th vertex other than a
    const double* q = x + (int64_t)vj * D;
#pragma unroll
    for (int k = 0; k < D; ++k) o[j][k] = q[k];
  }
}

template <int D>
__device__ __forceinline__ void cell_geom_others(const double (&o)[D][D], int a, const double (&xo)[D], CellGeom<D>& G) {
  double p[D + 1][D];
#pragma unroll
  for (int b = 0; b <= D; ++b) {
#pragma unroll
    for (int k = 0; k < D; ++k) {
      const double lo = o[b > 0 ? b - 1 : 0][k], hi = o[b < D ? b : D - 1][k];
      p[b][k] = (a == b) ? xo[k] : ((a > b) ? hi : lo);
    }
  }
  cell_geom_p<D>(p, G);
}

template <int D>
__device__ __forceinline__ void cell_geom_owner(const double* __restrict__ x, const int32_t v[D + 1], int a,
                                                const double (&xo)[D], CellGeom<D>& G) {
  double o[D][D];
  load_other_vertices<D>(x, v, a, o);
  cell_geom_others<D>(o, a, xo, G);
}

// Row a of the gradient table with a lane-varying a.  Written as an exact blend
// (weights 1.0 / 0.0) rather than a select chain: LLVM folds select(load, load) on a
// private array into a dynamically indexed load, which forces the whole table into
// scratch memory (measured: 24 GB of HBM writes per launch on the 10 M-DOF mesh).
// 1.0*x + 0.0*y is exact for finite values, so K stays bitwise symmetric.
template <int D>
__device__ __forceinline__ void select_row(const CellGeom<D>& G, int a, double ga[D]) {
  double w[D + 1];
#pragma unroll
  for (int b = 0; b <= D; ++b) w[b] = (a == b) ? 1.0 : 0.0;
#pragma unroll
  for (int k = 0; k < D; ++k) {
    double s = w[0] * G.g[0][k];
#pragma unroll
    for (int b = 1; b <= D; ++b) s += w[b] * G.g[b][k];
    ga[k] = s;
  }
}

template <int D>
__device__ __forceinline__ double dotD(const double* p, const double* q) {
  // explicit roundings: the value must not depend on which operand the
  // compiler happens to fuse, so that K[i][j] == K[j][i] bitwise
  double s = __dmul_rn(p[0], q[0]);
#pragma unroll
  for (int k = 1; k < D; ++k) s = __fma_rn(p[k], q[k], s);
  return s;
}

// ------------------------------------------------ nonlinear Poisson + Nitsche --
// examples/nonlinear_poisson_opt/run_nonlinear_poisson_opt.py:88-125 (sym = True, beta):
//   + int u^3 v                       closed-form P1 monomial integrals (exact, = degree-4 rule)
//   - int_dO (grad u . n) v           nitsche_1
//   + int_dO (u_ex - u) (grad v . n)  nitsche_2
//   + beta/h_E int_dO (u - u_ex) v    penalty, h_E = largest vertex distance (UFL CellDiameter [ext])
// u_ex is the CG1 interpolant `aux`.  The facet opposite local vertex k has outward normal
// -g_k/|g_k| and measure D |T| |g_k|; bit k of bfacets[cell] marks it as a boundary facet.
template <int D>
__device__ __forceinline__ double cell_diameter(const double* __restrict__ x, const int32_t v[D + 1]) {
  double h2 = 0.0;
#pragma unroll
  for (int a = 0; a <= D; ++a)
#pragma unroll
    for (int b = a + 1; b <= D; ++b) {
      double s = 0.0;
#pragma unroll
      for (int k = 0; k < D; ++k) {
        const double t = x[(int64_t)v[a] * D + k] - x[(int64_t)v[b] * D + k];
        s += t * t;
      }
      h2 = fmax(h2, s);
    }
  return sqrt(h2);
}

// Adds the nonlinear/boundary parts of row a: Jacobian entries into krow[b] (if WANT_J) and
// the residual into *res (if WANT_R).  w[b] = (a == b) as 1.0/0.0; ue = u at the cell vertices.
template <int D, bool WANT_J, bool WANT_R>
__device__ __forceinline__ void nl_row(const CellGeom<D>& G, int a, const double w[D + 1], const double ue[D + 1],
                                       const double* __restrict__ x, const int32_t v[D + 1],
                                       const double* __restrict__ aux, unsigned bits, double beta, double sgn,
                                       double krow[D + 1], double* res) {
  constexpr double coef = (D == 2) ? 1.0 / 360.0 : 1.0 / 840.0;   // D!/(D+4)!
  double s1 = 0.0, s2 = 0.0, s3 = 0.0, ua = 0.0;
#pragma unroll
  for (int b = 0; b <= D; ++b) {
    s1 += ue[b]; s2 += ue[b] * ue[b]; s3 += ue[b] * ue[b] * ue[b];
    ua += w[b] * ue[b];
  }
  const double h2 = 0.5 * (s1 * s1 + s2);
  if (WANT_R) {
    const double h3 = (s1 * s1 * s1 + 3.0 * s1 * s2 + 2.0 * s3) * (1.0 / 6.0);
    *res += G.vol * coef * 6.0 * (h3 + ua * (h2 + ua * s1 + ua * ua));
  }
  if (WANT_J) {
#pragma unroll
    for (int b = 0; b <= D; ++b) {
      const double ub = ue[b];
      const double off = 2.0 * (h2 + ua * (s1 + ua) + ub * (s1 + ub) + ua * ub);
      const double dia = 2.0 * (4.0 * ua * s1 + 6.0 * ua * ua + 2.0 * h2);
      krow[b] += 3.0 * G.vol * coef * (w[b] * dia + (1.0 - w[b]) * off);
    }
  }
  if (bits == 0u) return;
  const double hE = cell_diameter<D>(x, v);
  double e[D + 1], ea = 0.0;
#pragma unroll
  for (int b = 0; b <= D; ++b) {
    e[b] = ue[b] - aux[v[b]];
    ea += w[b] * e[b];
  }
#pragma unroll
  for (int k = 0; k <= D; ++k) {
    if (!((bits >> k) & 1u)) continue;
    const double ng = sqrt(dotD<D>(G.g[k], G.g[k]));
    const double meas = D * G.vol * ng, inv = -1.0 / ng;
    double gn[D + 1], gna = 0.0, dun = 0.0, son = 0.0;
#pragma unroll
    for (int b = 0; b <= D; ++b) {
      gn[b] = dotD<D>(G.g[b], G.g[k]) * inv;
      gna += w[b] * gn[b];
      dun += gn[b] * ue[b];
      if (b != k) son += e[b];
    }
    const double a_on = 1.0 - w[k];                    // 1 if a is a vertex of the facet
    const double pen = beta / hE * meas * (1.0 / (D * (D + 1)));
    if (WANT_R) *res += sgn * gna * (-(son * (1.0 / D))) * meas + a_on * (-dun * meas * (1.0 / D) + pen * (ea + son));
    if (WANT_J) {
#pragma unroll
      for (int b = 0; b <= D; ++b) {
        double t = a_on * (-gn[b] * meas * (1.0 / D));
        if (b != k) t += -sgn * gna * meas * (1.0 / D) + a_on * pen * (1.0 + w[b]);
        krow[b] += t;
      }
    }
  }
}

// ---------------------------------------- Euler-Bernoulli beam, cubic Hermite ---
// examples/beam_thickness_opt/run_thickness_opt_cantilever_beam.py:71-79:
//   inner(div grad v, EI div grad u) dx - f v(L),  EI = E * width * t^3 / 12, t DG0 per element.
// A beam element couples its 4 DOFs (w_i, th_i, w_i+1, th_i+1) all-to-all, exactly like the
// vertices of a tetrahedron, so the beam is stored as a tdim = 3 "mesh" whose vertices are the
// DOFs (x[dof] = (node position, 0, 0), conn[e] = {2e, 2e+1, 2e+2, 2e+3}) and the incidence,
// the SELL pattern and the owner-computes walk are reused unchanged.
// Row a of the Hermite element matrix (12, 6h, -12, 6h; 6h, 4h^2, -6h, 2h^2; ...) / h^3.
__device__ __forceinline__ void beam_khat_row(double h, const double w[4], double k[4]) {
  const double i3 = 1.0 / (h * h * h);
  const double r0[4] = {12.0, 6.0 * h, -12.0, 6.0 * h};
  const double r1[4] = {6.0 * h, 4.0 * h * h, -6.0 * h, 2.0 * h * h};
  const double r2[4] = {-12.0, -6.0 * h, 12.0, -6.0 * h};
  const double r3[4] = {6.0 * h, 2.0 * h * h, -6.0 * h, 4.0 * h * h};
#pragma unroll
  for (int b = 0; b < 4; ++b) k[b] = (w[0] * r0[b] + w[1] * r1[b] + w[2] * r2[b] + w[3] * r3[b]) * i3;
}

// ---------------------------------------------------------------- residual --
template <int D, int PDE>
__global__ __launch_bounds__(FEMO_BLOCK) void k_residual(
    int64_t n_rows, int64_t n_blocks, const int64_t* __restrict__ vptr,
    const int32_t* __restrict__ visit_cell, const int32_t* __restrict__ conn,
    const double* __restrict__ x, const double* __restrict__ u, const double* __restrict__ f,
    const double* __restrict__ aux, const uint8_t* __restrict__ bfacets, double beta, double sgn,
    double* __restrict__ r) {
  const int64_t blk = femo_xcd_block(blockIdx.x, n_blocks);
  const int64_t row = blk * FEMO_BLOCK + threadIdx.x;
  const int64_t slice = row >> 6;
  const int lane = threadIdx.x & 63;
  if ((slice << 6) >= n_rows) return;
  const int64_t vb = vptr[slice];
  const int nvis = (int)((vptr[slice + 1] - vb) >> 6);
  double acc = 0.0;
  for (int s = 0; s < nvis; ++s) {
    const int32_t ca = visit_cell[vb + (int64_t)s * 64 + lane];
    if (ca < 0) continue;
    const int64_t c = ca >> 2;
    const int a = ca & 3;
    int32_t v[D + 1];
    load_conn<D>(conn, c, v);
    if constexpr (PDE == FEMO_PDE_EB_BEAM) {
      double w[4], k[4];
#pragma unroll
      for (int b = 0; b < 4; ++b) w[b] = (a == b) ? 1.0 : 0.0;
      const double h = x[(int64_t)v[2] * 3] - x[(int64_t)v[0] * 3];
      const double t = f[c];
      const double EI = beta * sgn * t * t * t * (1.0 / 12.0);         // params: E, width
      beam_khat_row(h, w, k);
#pragma unroll
      for (int b = 0; b < 4; ++b) acc += EI * k[b] * u[v[b]];
      continue;
    }
    CellGeom<D> G;
    cell_geom<D>(x, v, G);
    double gu[D], ue[D + 1];
#pragma unroll
    for (int k = 0; k < D; ++k) gu[k] = 0.0;
#pragma unroll
    for (int b = 0; b <= D; ++b) {
      ue[b] = u[v[b]];
#pragma unroll
      for (int k = 0; k < D; ++k) gu[k] += G.g[b][k] * ue[b];
    }
    double ga[D];
    select_row<D>(G, a, ga);
    acc += G.vol * dotD<D>(ga, gu) - f[c] * G.vol * (1.0 / (D + 1));
    if constexpr (PDE == FEMO_PDE_NL_POISSON) {
      double w[D + 1], dummy[D + 1];
#pragma unroll
      for (int b = 0; b <= D; ++b) w[b] = (a == b) ? 1.0 : 0.0;
      const unsigned bits = bfacets ? bfacets[c] : 0u;
      nl_row<D, false, true>(G, a, w, ue, x, v, aux, bits, beta, sgn, dummy, &acc);
    }
  }
  if constexpr (PDE == FEMO_PDE_EB_BEAM) {
    if (row < n_rows && aux != nullptr) acc -= aux[row];            // nodal load vector f v(L)
  }
  if (row < n_rows) r[row] = acc;
}

// ---------------------------------------------------------------- jacobian --
// One pass over the incidence produces any subset of
//   (diag0, vals0)  dR/du without BCs            state_model.py:132
//   (diag1, vals1)  dR/du with Dirichlet rows/cols eliminated, diagonal 1
//                                                 state_model.py:149 / dolfinx NonlinearProblem.J [ext]
//   rhs             Newton right-hand side  F + K[:,bc](g-u), rows bc = u-g
//                                                 dolfinx NonlinearProblem.F [ext], utils_dolfinx.py:431
// LDS strip acc[k][tid]: bank = (k*256 + tid)*2 mod 64 depends on tid only ->
// conflict-free for any per-lane k.
template <int D, int PDE>
__global__ __launch_bounds__(FEMO_BLOCK) void k_jacobian(
    int64_t n_rows, int64_t n_blocks, const int64_t* __restrict__ vptr,
    const int32_t* __restrict__ visit_cell, const uint32_t* __restrict__ visit_slots,
    const int64_t* __restrict__ mptr, const int32_t* __restrict__ cols,
    const int32_t* __restrict__ sdelta, int sdelta_stride,
    const int32_t* __restrict__ rowlen, const int32_t* __restrict__ conn,
    const double* __restrict__ x, const double* __restrict__ u, const double* __restrict__ f,
    const double* __restrict__ aux, const uint8_t* __restrict__ bfacets, double beta, double sgn,
    const uint8_t* __restrict__ bcmask, const double* __restrict__ bcval,
    double* __restrict__ diag0, double* __restrict__ vals0, double* __restrict__ diag1,
    double* __restrict__ vals1, double* __restrict__ rhs) {
  extern __shared__ double strip[];
  const int tid = threadIdx.x;
  const int64_t blk = femo_xcd_block(blockIdx.x, n_blocks);
  const int64_t row = blk * FEMO_BLOCK + tid;
  const int64_t slice = row >> 6;
  const int lane = tid & 63;
  if ((slice << 6) >= n_rows) return;
  const int len = rowlen[row];
  for (int k = 0; k < len; ++k) strip[k * FEMO_BLOCK + tid] = 0.0;
  const int64_t vb = vptr[slice];
  const int nvis = (int)((vptr[slice + 1] - vb) >> 6);
  const bool want_rhs = rhs != nullptr;
  const int64_t mb = mptr[slice];
  const int32_t* dl = sdelta + slice * sdelta_stride;
  const bool regular = dl[0] != INT32_MIN;
  double dsum = 0.0, racc = 0.0;
  double xo[D];
  {
    const int64_t r0 = row < n_rows ? row : 0;      // padded lanes of the last slice visit nothing
#pragma unroll
    for (int k = 0; k < D; ++k) xo[k] = x[r0 * D + k];
  }
  if constexpr (PDE != FEMO_PDE_EB_BEAM) {
    // Software-pipelined walk.  A visit is a chain of dependent loads (incidence words -> the
    // row's columns -> coordinates, state values); taken one visit at a time the wave sits in
    // s_waitcnt five times per cell.  Three stages in flight: incidence words of visit s+2, column
    // lookup and the gathers of visit s+1, arithmetic of visit s.  Same order of accumulation.
    constexpr bool NEED_U = PDE == FEMO_PDE_NL_POISSON;
    constexpr bool NEED_F = PDE != FEMO_PDE_MASS;
    auto fetch_ids = [&](int s, int32_t& ca, uint32_t& sl) {
      ca = -1; sl = 0u;
      if (s < nvis) {
        const int64_t vi = vb + (int64_t)s * 64 + lane;
        ca = visit_cell[vi];
        sl = visit_slots[vi];
      }
    };
    auto gather = [&](int32_t ca, uint32_t sl, int32_t (&v)[D + 1], double (&o)[D][D], double& fc, double (&ue)[D + 1]) {
      if (ca >= 0) {
        row_cell_vertices<D>(row, lane, ca & 3, sl, regular, dl, cols, mb, v);
        load_other_vertices<D>(x, v, ca & 3, o);
        if constexpr (NEED_F) {
          if (want_rhs) fc = f[ca >> 2];
        }
        if constexpr (NEED_U) {
#pragma unroll
          for (int b = 0; b <= D; ++b) ue[b] = u[v[b]];
        }
      }
    };
    int32_t ca0, ca1, ca2, v0[D + 1], v1[D + 1];
    uint32_t sl0, sl1, sl2;
    double o0[D][D], o1[D][D], f0 = 0.0, f1 = 0.0, ue0[D + 1], ue1[D + 1];
#pragma unroll
    for (int j = 0; j < D; ++j)
#pragma unroll
      for (int k = 0; k < D; ++k) { o0[j][k] = 0.0; o1[j][k] = 0.0; }
#pragma unroll
    for (int b = 0; b <= D; ++b) { v0[b] = 0; v1[b] = 0; ue0[b] = 0.0; ue1[b] = 0.0; }
    fetch_ids(0, ca0, sl0);
    fetch_ids(1, ca1, sl1);
    gather(ca0, sl0, v0, o0, f0, ue0);
    for (int s = 0; s < nvis; ++s) {
      fetch_ids(s + 2, ca2, sl2);
      gather(ca1, sl1, v1, o1, f1, ue1);
      if (ca0 >= 0) {
        const int a = ca0 & 3;
        CellGeom<D> G;
        double ga[D], krow[D + 1];
        cell_geom_others<D>(o0, a, xo, G);
        select_row<D>(G, a, ga);
        if constexpr (NEED_F) {
          if (want_rhs) racc -= f0 * G.vol * (1.0 / (D + 1));
        }
#pragma unroll
        for (int b = 0; b <= D; ++b) {
          if constexpr (PDE == FEMO_PDE_MASS) {
            // P1 mass matrix |T| (1 + delta_ab) / ((d+1)(d+2))  (utils_dolfinx.py:569 inner(Pv, w) dx)
            krow[b] = G.vol * (1.0 / ((D + 1) * (D + 2))) * ((a == b) ? 2.0 : 1.0);
          } else {
            krow[b] = __dmul_rn(G.vol, dotD<D>(ga, G.g[b]));
          }
          if constexpr (NEED_U) {
            if (want_rhs) racc += krow[b] * ue0[b];          // linear part of the residual: K u
          }
        }
        if constexpr (PDE == FEMO_PDE_NL_POISSON) {
          double w[D + 1];
#pragma unroll
          for (int b = 0; b <= D; ++b) w[b] = (a == b) ? 1.0 : 0.0;
          const unsigned bits = bfacets ? bfacets[ca0 >> 2] : 0u;
          if (want_rhs) nl_row<D, true, true>(G, a, w, ue0, x, v0, aux, bits, beta, sgn, krow, &racc);
          else nl_row<D, true, false>(G, a, w, ue0, x, v0, aux, bits, beta, sgn, krow, &racc);
        }
#pragma unroll
        for (int b = 0; b <= D; ++b) {
          if (b == a) {
            dsum += krow[b];
          } else {
            const int pos = (sl0 >> (8 * b)) & 0xFF;
            strip[pos * FEMO_BLOCK + tid] = __dadd_rn(strip[pos * FEMO_BLOCK + tid], krow[b]);
          }
        }
      }
      ca0 = ca1; sl0 = sl1; f0 = f1;
#pragma unroll
      for (int j = 0; j < D; ++j)
#pragma unroll
        for (int k = 0; k < D; ++k) o0[j][k] = o1[j][k];
#pragma unroll
      for (int b = 0; b <= D; ++b) { v0[b] = v1[b]; ue0[b] = ue1[b]; }
      ca1 = ca2; sl1 = sl2;
    }
  } else
  for (int s = 0; s < nvis; ++s) {
    const int64_t vi = vb + (int64_t)s * 64 + lane;
    const int32_t ca = visit_cell[vi];
    if (ca < 0) continue;
    const uint32_t slots = visit_slots[vi];
    const int64_t c = ca >> 2;
    const int a = ca & 3;
    int32_t v[D + 1];
    if constexpr (PDE == FEMO_PDE_EB_BEAM) load_conn<D>(conn, c, v);
    else row_cell_vertices<D>(row, lane, a, slots, regular, dl, cols, mb, v);
    double krow[D + 1];
    CellGeom<D> G;
    double ga[D];
    if constexpr (PDE == FEMO_PDE_EB_BEAM) {
      double w[4];
#pragma unroll
      for (int b = 0; b < 4; ++b) w[b] = (a == b) ? 1.0 : 0.0;
      const double h = x[(int64_t)v[2] * 3] - x[(int64_t)v[0] * 3];
      const double t = f[c];
      const double EI = beta * sgn * t * t * t * (1.0 / 12.0);         // params: E, width
      beam_khat_row(h, w, krow);
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        krow[b] *= EI;
        if (want_rhs) racc += krow[b] * u[v[b]];
      }
    } else {
    cell_geom_owner<D>(x, v, a, xo, G);
    select_row<D>(G, a, ga);
    if (want_rhs) racc -= f[c] * G.vol * (1.0 / (D + 1));
#pragma unroll
    for (int b = 0; b <= D; ++b) {
      if constexpr (PDE == FEMO_PDE_MASS) {
        // P1 mass matrix |T| (1 + delta_ab) / ((d+1)(d+2))  (utils_dolfinx.py:569 inner(Pv, w) dx)
        krow[b] = G.vol * (1.0 / ((D + 1) * (D + 2))) * ((a == b) ? 2.0 : 1.0);
      } else {
        krow[b] = __dmul_rn(G.vol, dotD<D>(ga, G.g[b]));
      }
      if constexpr (PDE != FEMO_PDE_POISSON) {
        if (want_rhs) racc += krow[b] * u[v[b]];         // linear part of the residual: K u
      }
    }
    }
    if constexpr (PDE == FEMO_PDE_NL_POISSON) {
      double w[D + 1], ue[D + 1];
#pragma unroll
      for (int b = 0; b <= D; ++b) {
        w[b] = (a == b) ? 1.0 : 0.0;
        ue[b] = u[v[b]];
      }
      const unsigned bits = bfacets ? bfacets[c] : 0u;
      if (want_rhs) nl_row<D, true, true>(G, a, w, ue, x, v, aux, bits, beta, sgn, krow, &racc);
      else nl_row<D, true, false>(G, a, w, ue, x, v, aux, bits, beta, sgn, krow, &racc);
    }
#pragma unroll
    for (int b = 0; b <= D; ++b) {
      if (b == a) {
        dsum += krow[b];
      } else {
        const int pos = (slots >> (8 * b)) & 0xFF;
        strip[pos * FEMO_BLOCK + tid] = __dadd_rn(strip[pos * FEMO_BLOCK + tid], krow[b]);
      }
    }
  }
  const bool valid = row < n_rows;
  if constexpr (PDE == FEMO_PDE_EB_BEAM) {
    if (want_rhs && valid && aux != nullptr) racc -= aux[row];      // nodal load vector f v(L)
  }
  const bool row_bc = bcmask != nullptr && valid && bcmask[row];
  if (diag0) diag0[row] = valid ? dsum : 1.0;  // padded rows of the last slice: harmless identity
  if (diag1) diag1[row] = (valid && !row_bc) ? dsum : 1.0;
  const int wm = (int)((mptr[slice + 1] - mb) >> 6);
  double lift = 0.0;
  // linear Poisson: (K u)_i from the finished row, one u gather per column instead of D+1 per
  // visited cell (padded entries hold 0 and point at the row itself)
  constexpr bool ROW_KU = PDE == FEMO_PDE_POISSON;
  if constexpr (ROW_KU) {
    if (want_rhs && valid) racc += dsum * u[row];
  }
  for (int k = 0; k < wm; k += 2) {
    const int64_t idx = mb + (int64_t)(k >> 1) * 128 + lane * 2;
    double2 o;
    o.x = k < len ? strip[k * FEMO_BLOCK + tid] : 0.0;
    o.y = (k + 1) < len ? strip[(k + 1) * FEMO_BLOCK + tid] : 0.0;
    if (vals0) *reinterpret_cast<double2*>(vals0 + idx) = o;
    if (bcmask != nullptr || (ROW_KU && want_rhs)) {
      const int2 cc = *reinterpret_cast<const int2*>(cols + idx);
      if constexpr (ROW_KU) {
        if (want_rhs && valid) racc += o.x * u[cc.x] + o.y * u[cc.y];
      }
    }
    if (bcmask != nullptr) {
      const int2 cc = *reinterpret_cast<const int2*>(cols + idx);
      const bool bx = bcmask[cc.x], by = bcmask[cc.y];
      if (want_rhs && !row_bc) {
        if (bx) lift += o.x * (bcval[cc.x] - u[cc.x]);
        if (by) lift += o.y * (bcval[cc.y] - u[cc.y]);
      }
      if (row_bc || bx) o.x = 0.0;
      if (row_bc || by) o.y = 0.0;
    }
    if (vals1) *reinterpret_cast<double2*>(vals1 + idx) = o;
  }
  if (want_rhs && valid) rhs[row] = row_bc ? (u[row] - bcval[row]) : (racc + lift);
}

// -------------------------------------------------------------------- dRdf --
// dR/dt of the beam: column e = (E width t_e^2 / 4) * Khat u_e  (4 entries aligned with conn[e])
__global__ __launch_bounds__(FEMO_BLOCK) void k_dRdt_beam(int64_t n_cell, double Ey, double width, const int32_t* __restrict__ conn,
                                                          const double* __restrict__ x, const double* __restrict__ u,
                                                          const double* __restrict__ t, double* __restrict__ vals) {
  for (int64_t c = (int64_t)blockIdx.x * FEMO_BLOCK + threadIdx.x; c < n_cell; c += (int64_t)gridDim.x * FEMO_BLOCK) {
    int32_t v[4];
    load_conn<3>(conn, c, v);
    const double h = x[(int64_t)v[2] * 3] - x[(int64_t)v[0] * 3];
    const double tc = t[c];
    const double dEI = Ey * width * tc * tc * 0.25;
    double ue[4];
#pragma unroll
    for (int b = 0; b < 4; ++b) ue[b] = u[v[b]];
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      double w[4] = {0.0, 0.0, 0.0, 0.0}, k[4];
      w[a] = 1.0;
      beam_khat_row(h, w, k);
      vals[c * 4 + a] = dEI * (k[0] * ue[0] + k[1] * ue[1] + k[2] * ue[2] + k[3] * ue[3]);
    }
  }
}

template <int D, int PDE>
__global__ __launch_bounds__(FEMO_BLOCK) void k_dRdf(int64_t n_cell, const int32_t* __restrict__ conn,
                                                     const double* __restrict__ x,
                                                     double* __restrict__ vals) {
  for (int64_t c = (int64_t)blockIdx.x * FEMO_BLOCK + threadIdx.x; c < n_cell;
       c += (int64_t)gridDim.x * FEMO_BLOCK) {
    int32_t v[D + 1];
    load_conn<D>(conn, c, v);
    CellGeom<D> G;
    cell_geom<D>(x, v, G);
    const double w = -G.vol * (1.0 / (D + 1));
    if constexpr (D == 3) {
      double2 o = {w, w};
      *reinterpret_cast<double2*>(vals + c * 4) = o;
      *reinterpret_cast<double2*>(vals + c * 4 + 2) = o;
    } else {
      vals[c * 3 + 0] = w; vals[c * 3 + 1] = w; vals[c * 3 + 2] = w;
    }
  }
}

template <int D>
__global__ __launch_bounds__(FEMO_BLOCK) void k_dRdf_apply_T(int64_t n_cell, const int32_t* __restrict__ conn,
                                                             const double* __restrict__ vals,
                                                             const double* __restrict__ xin,
                                                             double* __restrict__ y, int accumulate) {
  for (int64_t c = (int64_t)blockIdx.x * FEMO_BLOCK + threadIdx.x; c < n_cell;
       c += (int64_t)gridDim.x * FEMO_BLOCK) {
    int32_t v[D + 1];
    load_conn<D>(conn, c, v);
    double s = 0.0;
#pragma unroll
    for (int a = 0; a <= D; ++a) s += vals[c * (D + 1) + a] * xin[v[a]];
    y[c] = accumulate ? y[c] + s : s;
  }
}

template <int D>
__global__ __launch_bounds__(FEMO_BLOCK) void k_dRdf_apply_N(
    int64_t n_rows, int64_t n_blocks, const int64_t* __restrict__ vptr,
    const int32_t* __restrict__ visit_cell, const double* __restrict__ vals,
    const double* __restrict__ xin, double* __restrict__ y, int accumulate) {
  const int64_t blk = femo_xcd_block(blockIdx.x, n_blocks);
  const int64_t row = blk * FEMO_BLOCK + threadIdx.x;
  const int64_t slice = row >> 6;
  const int lane = threadIdx.x & 63;
  if ((slice << 6) >= n_rows) return;
  const int64_t vb = vptr[slice];
  const int nvis = (int)((vptr[slice + 1] - vb) >> 6);
  double acc = 0.0;
  for (int s = 0; s < nvis; ++s) {
    const int32_t ca = visit_cell[vb + (int64_t)s * 64 + lane];
    if (ca < 0) continue;
    const int64_t c = ca >> 2;
    acc += vals[c * (D + 1) + (ca & 3)] * xin[c];
  }
  if (row < n_rows) y[row] = accumulate ? y[row] + acc : acc;
}

// -------------------------------------------------------------- functional --
// J = 1/2 int (u-u_d)^2 + alpha/2 int f^2 ; P1 mass matrix in closed form:
// int_T e^2 = |T|/((d+1)(d+2)) (sum e_a^2 + (sum e_a)^2).
template <int D>
__global__ __launch_bounds__(FEMO_BLOCK) void k_functional_value(
    int64_t n_cell, int64_t n_rows, double alpha, const int32_t* __restrict__ conn,
    const double* __restrict__ x, const double* __restrict__ u, const double* __restrict__ f,
    const double* __restrict__ ud, double* __restrict__ partials) {
  __shared__ double lds[FEMO_BLOCK / 64];
  double acc = 0.0;
  for (int64_t c = (int64_t)blockIdx.x * FEMO_BLOCK + threadIdx.x; c < n_cell;
       c += (int64_t)gridDim.x * FEMO_BLOCK) {
    int32_t v[D + 1];
    load_conn<D>(conn, c, v);
    if (v[0] >= n_rows) continue;  // cell owned by the rank that owns its first vertex
    CellGeom<D> G;
    cell_geom<D>(x, v, G);
    double s1 = 0.0, s2 = 0.0;
#pragma unroll
    for (int a = 0; a <= D; ++a) {
      const double e = u[v[a]] - ud[v[a]];
      s1 += e;
      s2 += e * e;
    }
    const double fc = f[c];
    acc += 0.5 * G.vol * (1.0 / ((D + 1) * (D + 2))) * (s2 + s1 * s1) + 0.5 * alpha * fc * fc * G.vol;
  }
  const double s = femo_block_sum<FEMO_BLOCK>(acc, lds);
  if (threadIdx.x == 0) partials[blockIdx.x] = s;
}

// dJ/du_i = sum_cells |T|/((D+1)(D+2)) (e_i + sum_b e_b), e = u - u_d  (mass matrix times e).
// Same walk as the Jacobian: vertices from the row's own columns, own coordinates and own e in
// registers, so a visited cell costs D coordinate gathers and D gathers of e.
template <int D>
__global__ __launch_bounds__(FEMO_BLOCK) void k_functional_grad_u(
    int64_t n_rows, int64_t n_blocks, const int64_t* __restrict__ vptr,
    const int32_t* __restrict__ visit_cell, const uint32_t* __restrict__ visit_slots,
    const int64_t* __restrict__ mptr, const int32_t* __restrict__ cols, const int32_t* __restrict__ sdelta,
    int sdelta_stride, const double* __restrict__ x, const double* __restrict__ u, const double* __restrict__ ud,
    double* __restrict__ g) {
  const int64_t blk = femo_xcd_block(blockIdx.x, n_blocks);
  const int64_t row = blk * FEMO_BLOCK + threadIdx.x;
  const int64_t slice = row >> 6;
  const int lane = threadIdx.x & 63;
  if ((slice << 6) >= n_rows) return;
  const int64_t vb = vptr[slice];
  const int nvis = (int)((vptr[slice + 1] - vb) >> 6);
  const int64_t mb = mptr[slice];
  const int32_t* dl = sdelta + slice * sdelta_stride;
  const bool regular = dl[0] != INT32_MIN;
  const int64_t r0 = row < n_rows ? row : 0;
  double xo[D];
#pragma unroll
  for (int k = 0; k < D; ++k) xo[k] = x[r0 * D + k];
  const double eo = u[r0] - ud[r0];
  double acc = 0.0;
  // software-pipelined like the Jacobian walk: ids of visit s+2, gathers of s+1, arithmetic of s
  auto fetch_ids = [&](int s, int32_t& ca, uint32_t& sl) {
    ca = -1; sl = 0u;
    if (s < nvis) {
      const int64_t vi = vb + (int64_t)s * 64 + lane;
      ca = visit_cell[vi];
      sl = visit_slots[vi];
    }
  };
  auto gather = [&](int32_t ca, uint32_t sl, double (&o)[D][D], double& esum) {
    if (ca >= 0) {
      const int a = ca & 3;
      int32_t v[D + 1];
      row_cell_vertices<D>(row, lane, a, sl, regular, dl, cols, mb, v);
      load_other_vertices<D>(x, v, a, o);
      double e = 0.0;
#pragma unroll
      for (int j = 0; j < D; ++j) {
        const int32_t vj = (a <= j) ? v[j + 1] : v[j];
        e += u[vj] - ud[vj];
      }
      esum = e;
    }
  };
  int32_t ca0, ca1, ca2;
  uint32_t sl0, sl1, sl2;
  double o0[D][D], o1[D][D], e0 = 0.0, e1 = 0.0;
#pragma unroll
  for (int j = 0; j < D; ++j)
#pragma unroll
    for (int k = 0; k < D; ++k) { o0[j][k] = 0.0; o1[j][k] = 0.0; }
  fetch_ids(0, ca0, sl0);
  fetch_ids(1, ca1, sl1);
  gather(ca0, sl0, o0, e0);
  for (int s = 0; s < nvis; ++s) {
    fetch_ids(s + 2, ca2, sl2);
    gather(ca1, sl1, o1, e1);
    if (ca0 >= 0) {
      CellGeom<D> G;
      cell_geom_others<D>(o0, ca0 & 3, xo, G);
      acc += G.vol * (1.0 / ((D + 1) * (D + 2))) * (eo + (eo + e0));
    }
    ca0 = ca1; sl0 = sl1; e0 = e1;
#pragma unroll
    for (int j = 0; j < D; ++j)
#pragma unroll
      for (int k = 0; k < D; ++k) o0[j][k] = o1[j][k];
    ca1 = ca2; sl1 = sl2;
  }
  if (row < n_rows) g[row] = acc;
}

template <int D>
__global__ __launch_bounds__(FEMO_BLOCK) void k_functional_grad_f(int64_t n_cell, double alpha,
                                                                  const int32_t* __restrict__ conn,
                                                                  const double* __restrict__ x,
                                                                  const double* __restrict__ f,
                                                                  double* __restrict__ g) {
  for (int64_t c = (int64_t)blockIdx.x * FEMO_BLOCK + threadIdx.x; c < n_cell;
       c += (int64_t)gridDim.x * FEMO_BLOCK) {
    int32_t v[D + 1];
    load_conn<D>(conn, c, v);
    CellGeom<D> G;
    cell_geom<D>(x, v, G);
    g[c] = alpha * f[c] * G.vol;
  }
}

// DG0 field expressions for L2-projected outputs (fea_dolfinx.py:148-161, utils_dolfinx.py:549-583)
//   kind 0: |grad u| per cell (u CG1)      kind 1: w_c ** p (w DG0, p = params[0])
template <int D>
__global__ __launch_bounds__(FEMO_BLOCK) void k_cell_expr(int64_t n_cell, int kind, double p0, const int32_t* __restrict__ conn,
                                                          const double* __restrict__ x, const double* __restrict__ in,
                                                          double* __restrict__ out) {
  for (int64_t c = (int64_t)blockIdx.x * FEMO_BLOCK + threadIdx.x; c < n_cell; c += (int64_t)gridDim.x * FEMO_BLOCK) {
    if (kind == 1) {
      out[c] = pow(in[c], p0);
      continue;
    }
    int32_t v[D + 1];
    load_conn<D>(conn, c, v);
    CellGeom<D> G;
    cell_geom<D>(x, v, G);
    double gu[D];
#pragma unroll
    for (int k = 0; k < D; ++k) gu[k] = 0.0;
#pragma unroll
    for (int b = 0; b <= D; ++b) {
      const double ub = in[v[b]];
#pragma unroll
      for (int k = 0; k < D; ++k) gu[k] += G.g[b][k] * ub;
    }
    out[c] = sqrt(dotD<D>(gu, gu));
  }
}

__global__ void k_reduce_partials(int nblocks, int nsums, const double* __restrict__ partials,
                                  double* __restrict__ out) {
  __shared__ double lds[1024 / 64];
  for (int j = 0; j < nsums; ++j) {
    double acc = 0.0;
    for (int i = threadIdx.x; i < nblocks; i += 1024) acc += partials[(int64_t)j * FEMO_MAX_PARTIALS + i];
    const double s = femo_block_sum<1024>(acc, lds);
    if (threadIdx.x == 0) out[j] = s;
  }
}

inline int cell_grid(int64_t n_cell) {
  int64_t g = (n_cell + FEMO_BLOCK - 1) / FEMO_BLOCK;
  if (g > FEMO_MAX_PARTIALS) g = FEMO_MAX_PARTIALS;
  if (g < 1) g = 1;
  return (int)g;
}

inline int64_t row_blocks(const femo_mesh* m) { return (m->n_slices * FEMO_WAVE + FEMO_BLOCK - 1) / FEMO_BLOCK; }

}  // namespace

// tdim dispatch: KERNEL<3, extra...> or KERNEL<2, extra...>
#define FEMO_LAUNCH_D(m, KERNEL, grid, lds, st, ...)                                          \
  do {                                                                                        \
    if ((m)->tdim == 3) hipLaunchKernelGGL((KERNEL<3>), dim3(grid), dim3(FEMO_BLOCK), lds, st, __VA_ARGS__); \
    else hipLaunchKernelGGL((KERNEL<2>), dim3(grid), dim3(FEMO_BLOCK), lds, st, __VA_ARGS__);  \
  } while (0)
#define FEMO_LAUNCH_DP(m, KERNEL, P, grid, lds, st, ...)                                      \
  do {                                                                                        \
    if ((m)->tdim == 3) hipLaunchKernelGGL((KERNEL<3, P>), dim3(grid), dim3(FEMO_BLOCK), lds, st, __VA_ARGS__); \
    else hipLaunchKernelGGL((KERNEL<2, P>), dim3(grid), dim3(FEMO_BLOCK), lds, st, __VA_ARGS__); \
  } while (0)

static int check_nl(femo_mesh* m, int pde, const double* u, const double* aux) {
  FEMO_REQUIRE(pde == FEMO_PDE_POISSON || pde == FEMO_PDE_NL_POISSON || pde == FEMO_PDE_MASS || pde == FEMO_PDE_EB_BEAM,
               "pde kind %d not implemented", pde);
  if (pde == FEMO_PDE_EB_BEAM) {
    FEMO_REQUIRE(m->tdim == 3, "the Hermite beam lives on a 4-DOF-per-element (tdim = 3) mesh");
    FEMO_REQUIRE(u != nullptr, "the beam form needs the state u");
  }
  if (pde == FEMO_PDE_NL_POISSON) {
    FEMO_REQUIRE(u != nullptr, "the nonlinear Poisson form needs the state u");
    FEMO_REQUIRE(m->d_bfacets == nullptr || aux != nullptr, "Nitsche terms need the boundary data u_exact (aux)");
  }
  return 0;
}

int femo_launch_residual(femo_mesh* m, int pde, const double* params, const double* u,
                         const double* f, const double* aux, double* r) {
  FEMO_TRY(check_nl(m, pde, u, aux));
  FEMO_REQUIRE(pde != FEMO_PDE_MASS, "the mass form has no residual");
  const int64_t nb = row_blocks(m);
  if (nb == 0) return 0;
  hipStream_t st = m->ctx->stream;
  const double beta = params ? params[0] : 0.0;
  const double sgn = (params && params[1] != 0.0) ? params[1] : 1.0;
  if (pde == FEMO_PDE_EB_BEAM)
    hipLaunchKernelGGL((k_residual<3, FEMO_PDE_EB_BEAM>), dim3(nb), dim3(FEMO_BLOCK), 0, st, m->n_rows, nb, m->d_vptr, m->d_visit_cell, m->d_conn, m->d_x, u, f, aux, m->d_bfacets, params ? params[0] : 1.0, params ? params[1] : 1.0, r);
  else if (pde == FEMO_PDE_NL_POISSON)
    FEMO_LAUNCH_DP(m, k_residual, FEMO_PDE_NL_POISSON, nb, 0, st, m->n_rows, nb, m->d_vptr, m->d_visit_cell, m->d_conn, m->d_x, u, f, aux, m->d_bfacets, beta, sgn, r);
  else
    FEMO_LAUNCH_DP(m, k_residual, FEMO_PDE_POISSON, nb, 0, st, m->n_rows, nb, m->d_vptr, m->d_visit_cell, m->d_conn, m->d_x, u, f, aux, m->d_bfacets, beta, sgn, r);
  FEMO_HIP_CHECK(hipGetLastError());
  return 0;
}

template <int D, int PDE>
static int launch_system_t(femo_mesh* m, int64_t nb, size_t lds, const double* u, const double* f, const double* aux,
                           double beta, double sgn, const uint8_t* bcmask, const double* bcval, double* diag0, double* vals0,
                           double* diag1, double* vals1, double* rhs) {
  auto k = k_jacobian<D, PDE>;
  if (lds > 64 * 1024) FEMO_HIP_CHECK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(k, dim3(nb), dim3(FEMO_BLOCK), lds, m->ctx->stream, m->n_rows, nb, m->d_vptr, m->d_visit_cell, m->d_visit_slots, m->d_mptr, m->d_cols, m->d_sdelta, m->sdelta_stride, m->d_rowlen, m->d_conn, m->d_x, u, f, aux, m->d_bfacets, beta, sgn, bcmask, bcval, diag0, vals0, diag1, vals1, rhs);
  FEMO_HIP_CHECK(hipGetLastError());
  return 0;
}

int femo_launch_system(femo_mesh* m, int pde, const double* params, const double* u, const double* f,
                       const double* aux, const uint8_t* bcmask, const double* bcval, double* diag0,
                       double* vals0, double* diag1, double* vals1, double* rhs) {
  FEMO_TRY(check_nl(m, pde, u, aux));
  FEMO_REQUIRE(rhs == nullptr || (u != nullptr && f != nullptr), "the Newton right-hand side needs u and f");
  FEMO_REQUIRE((diag1 == nullptr && rhs == nullptr) || bcmask == nullptr || bcval != nullptr, "missing Dirichlet values");
  const int64_t nb = row_blocks(m);
  if (nb == 0) return 0;
  int cap = 16;
  while (cap < m->max_rowlen) cap *= 2;
  const size_t lds = (size_t)cap * FEMO_BLOCK * sizeof(double);
  FEMO_REQUIRE(lds <= 160 * 1024, "row length %d exceeds the LDS strip capacity", m->max_rowlen);
  const double beta = params ? params[0] : 0.0;
  const double sgn = (params && params[1] != 0.0) ? params[1] : 1.0;
  if (pde == FEMO_PDE_EB_BEAM) {
    FEMO_REQUIRE(f != nullptr, "the beam form needs the thickness field");
    return launch_system_t<3, FEMO_PDE_EB_BEAM>(m, nb, lds, u, f, aux, params ? params[0] : 1.0, params ? params[1] : 1.0,
                                                bcmask, bcval, diag0, vals0, diag1, vals1, rhs);
  }
  if (pde == FEMO_PDE_MASS) {
    FEMO_REQUIRE(rhs == nullptr, "the mass form has no residual");
    if (m->tdim == 3) return launch_system_t<3, FEMO_PDE_MASS>(m, nb, lds, u, f, aux, beta, sgn, bcmask, bcval, diag0, vals0, diag1, vals1, rhs);
    return launch_system_t<2, FEMO_PDE_MASS>(m, nb, lds, u, f, aux, beta, sgn, bcmask, bcval, diag0, vals0, diag1, vals1, rhs);
  }
  if (m->tdim == 3) {
    if (pde == FEMO_PDE_NL_POISSON) return launch_system_t<3, FEMO_PDE_NL_POISSON>(m, nb, lds, u, f, aux, beta, sgn, bcmask, bcval, diag0, vals0, diag1, vals1, rhs);
    return launch_system_t<3, FEMO_PDE_POISSON>(m, nb, lds, u, f, aux, beta, sgn, bcmask, bcval, diag0, vals0, diag1, vals1, rhs);
  }
  if (pde == FEMO_PDE_NL_POISSON) return launch_system_t<2, FEMO_PDE_NL_POISSON>(m, nb, lds, u, f, aux, beta, sgn, bcmask, bcval, diag0, vals0, diag1, vals1, rhs);
  return launch_system_t<2, FEMO_PDE_POISSON>(m, nb, lds, u, f, aux, beta, sgn, bcmask, bcval, diag0, vals0, diag1, vals1, rhs);
}

int femo_launch_dRdf(femo_mesh* m, int pde, const double* params, const double* u,
                     const double* f, double* vals) {
  FEMO_REQUIRE(pde == FEMO_PDE_POISSON || pde == FEMO_PDE_NL_POISSON || pde == FEMO_PDE_EB_BEAM, "pde kind %d not implemented", pde);
  if (m->n_cell == 0) return 0;
  const int g = cell_grid(m->n_cell);
  hipStream_t st = m->ctx->stream;
  if (pde == FEMO_PDE_EB_BEAM) {
    FEMO_REQUIRE(m->tdim == 3 && u != nullptr && f != nullptr, "dR/dt of the beam needs u, t and a tdim = 3 mesh");
    hipLaunchKernelGGL(k_dRdt_beam, dim3(g), dim3(FEMO_BLOCK), 0, st, m->n_cell, params ? params[0] : 1.0, params ? params[1] : 1.0, m->d_conn, m->d_x, u, f, vals);
    FEMO_HIP_CHECK(hipGetLastError());
    return 0;
  }
  FEMO_LAUNCH_DP(m, k_dRdf, 0, g, 0, st, m->n_cell, m->d_conn, m->d_x, vals);
  FEMO_HIP_CHECK(hipGetLastError());
  return 0;
}

int femo_launch_dRdf_apply(femo_mesh* m, const double* vals, int transpose, const double* x,
                           double* y, int accumulate) {
  hipStream_t st = m->ctx->stream;
  if (transpose) {
    if (m->n_cell == 0) return 0;
    const int g = cell_grid(m->n_cell);
    FEMO_LAUNCH_D(m, k_dRdf_apply_T, g, 0, st, m->n_cell, m->d_conn, vals, x, y, accumulate);
  } else {
    const int64_t nb = row_blocks(m);
    if (nb == 0) return 0;
    FEMO_LAUNCH_D(m, k_dRdf_apply_N, nb, 0, st, m->n_rows, nb, m->d_vptr, m->d_visit_cell, vals, x, y, accumulate);
  }
  FEMO_HIP_CHECK(hipGetLastError());
  return 0;
}

int femo_launch_cell_expr(femo_mesh* m, int kind, const double* params, const double* in, double* out) {
  FEMO_REQUIRE(kind == 0 || kind == 1, "cell expression kind %d not implemented", kind);
  if (m->n_cell == 0) return 0;
  const int g = cell_grid(m->n_cell);
  const double p0 = params ? params[0] : 1.0;
  hipStream_t st = m->ctx->stream;
  FEMO_LAUNCH_D(m, k_cell_expr, g, 0, st, m->n_cell, kind, p0, m->d_conn, m->d_x, in, out);
  FEMO_HIP_CHECK(hipGetLastError());
  return 0;
}

int femo_reduce_to_host(femo_ctx* ctx, int nblocks, int nsums, double* host_out) {
  FEMO_REQUIRE(nsums <= FEMO_NSCAL && nblocks <= FEMO_MAX_PARTIALS, "reduction too large");
  hipLaunchKernelGGL(k_reduce_partials, dim3(1), dim3(1024), 0, ctx->stream, nblocks, nsums, ctx->d_partials, ctx->d_scal);
  FEMO_HIP_CHECK(hipGetLastError());
  if (ctx->nranks > 1)
    FEMO_TRY(femo_coll_allreduce(ctx, ctx->d_scal, nsums, ctx->stream));
  FEMO_HIP_CHECK(hipMemcpyAsync(ctx->h_scal, ctx->d_scal, nsums * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  FEMO_HIP_CHECK(hipStreamSynchronize(ctx->stream));
  for (int j = 0; j < nsums; ++j) host_out[j] = ctx->h_scal[j];
  return 0;
}

int femo_launch_functional_value(femo_mesh* m, int kind, const double* params, const double* u,
                                 const double* f, const double* ud, double* host_value) {
  FEMO_REQUIRE(kind == FEMO_J_L2_TRACKING, "functional kind %d not implemented", kind);
  const double alpha = params ? params[0] : 0.0;
  const int g = cell_grid(m->n_cell);
  hipStream_t st = m->ctx->stream;
  FEMO_LAUNCH_D(m, k_functional_value, g, 0, st, m->n_cell, m->n_rows, alpha, m->d_conn, m->d_x, u, f, ud, m->ctx->d_partials);
  FEMO_HIP_CHECK(hipGetLastError());
  return femo_reduce_to_host(m->ctx, g, 1, host_value);
}

int femo_launch_functional_grad_u(femo_mesh* m, int kind, const double* params, const double* u,
                                  const double* f, const double* ud, double* gout) {
  FEMO_REQUIRE(kind == FEMO_J_L2_TRACKING, "functional kind %d not implemented", kind);
  const int64_t nb = row_blocks(m);
  if (nb == 0) return 0;
  hipStream_t st = m->ctx->stream;
  FEMO_LAUNCH_D(m, k_functional_grad_u, nb, 0, st, m->n_rows, nb, m->d_vptr, m->d_visit_cell, m->d_visit_slots, m->d_mptr, m->d_cols, m->d_sdelta, m->sdelta_stride, m->d_x, u, ud, gout);
  FEMO_HIP_CHECK(hipGetLastError());
  return 0;
}

int femo_launch_functional_grad_f(femo_mesh* m, int kind, const double* params, const double* u,
                                  const double* f, const double* ud, double* gout) {
  FEMO_REQUIRE(kind == FEMO_J_L2_TRACKING, "functional kind %d not implemented", kind);
  if (m->n_cell == 0) return 0;
  const double alpha = params ? params[0] : 0.0;
  const int g = cell_grid(m->n_cell);
  hipStream_t st = m->ctx->stream;
  FEMO_LAUNCH_D(m, k_functional_grad_f, g, 0, st, m->n_cell, alpha, m->d_conn, m->d_x, f, gout);
  FEMO_HIP_CHECK(hipGetLastError());
  return 0;
}
