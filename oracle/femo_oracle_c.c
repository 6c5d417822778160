/* C/OpenMP restatement of femo's Poisson hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * Same role and rules as oracle/femo_oracle.py (read its header): a checker and
 * the "port" CPU baseline of bench.py; never imported by femo_amd/.  PARITY
 * UNPINNED against FEniCSx (not installable here); pinned against the NumPy
 * oracle and its closed-form known answers by tests/test_oracle_c.py.
 *
 * It exists because the NumPy oracle cannot assemble the 10 M-DOF / 60 M-cell
 * benchmark mesh in reasonable time or memory.  Algorithms follow the
 * reference's call sequence (paths relative to /root/reference):
 *   cell loop + insertion into CSR   dolfinx assemble_matrix / MatSetValues [ext],
 *                                    utils_dolfinx.py:181-187, 189-202
 *   residual cell loop               utils_dolfinx.py:175-179 (run_poisson_opt.py:32-38)
 *   Dirichlet rows/cols              utils_dolfinx.py:189-202
 *   functional and partials          output_model.py:69-87 (run_poisson_opt.py:74-76)
 *   dR/df^T lambda                   state_model.py:196-200
 * The linear solver is CG with the stopping rule of the HIP engine (BASELINE.json
 * design; the reference factorises with MUMPS), Jacobi-preconditioned (oc_pcg_jacobi)
 * or with the auxiliary-lattice BPX preconditioner (oc_pcg_bpx: the algorithm of
 * oracle/bpx_oracle.py, which restates femo_amd/csrc/bpx.hip).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

static void geom(int d, const double* x, const int32_t* v, double* vol, double g[4][3]) {
  if (d == 3) {
    const double *p0 = x + 3 * (int64_t)v[0], *p1 = x + 3 * (int64_t)v[1], *p2 = x + 3 * (int64_t)v[2],
                 *p3 = x + 3 * (int64_t)v[3];
    double e1[3], e2[3], e3[3];
    for (int k = 0; k < 3; ++k) { e1[k] = p1[k] - p0[k]; e2[k] = p2[k] - p0[k]; e3[k] = p3[k] - p0[k]; }
    double c1[3] = {e2[1] * e3[2] - e2[2] * e3[1], e2[2] * e3[0] - e2[0] * e3[2], e2[0] * e3[1] - e2[1] * e3[0]};
    double c2[3] = {e3[1] * e1[2] - e3[2] * e1[1], e3[2] * e1[0] - e3[0] * e1[2], e3[0] * e1[1] - e3[1] * e1[0]};
    double c3[3] = {e1[1] * e2[2] - e1[2] * e2[1], e1[2] * e2[0] - e1[0] * e2[2], e1[0] * e2[1] - e1[1] * e2[0]};
    const double det = e1[0] * c1[0] + e1[1] * c1[1] + e1[2] * c1[2];
    for (int k = 0; k < 3; ++k) {
      g[1][k] = c1[k] / det; g[2][k] = c2[k] / det; g[3][k] = c3[k] / det;
      g[0][k] = -(g[1][k] + g[2][k] + g[3][k]);
    }
    *vol = fabs(det) / 6.0;
  } else {
    const double *p0 = x + 2 * (int64_t)v[0], *p1 = x + 2 * (int64_t)v[1], *p2 = x + 2 * (int64_t)v[2];
    const double a = p1[0] - p0[0], b = p1[1] - p0[1], c = p2[0] - p0[0], e = p2[1] - p0[1];
    const double det = a * e - b * c;
    g[1][0] = e / det; g[1][1] = -c / det;
    g[2][0] = -b / det; g[2][1] = a / det;
    g[0][0] = -(g[1][0] + g[2][0]); g[0][1] = -(g[1][1] + g[2][1]);
    *vol = fabs(det) / 2.0;
  }
}

static int cmp_i32(const void* a, const void* b) {
  const int32_t x = *(const int32_t*)a, y = *(const int32_t*)b;
  return (x > y) - (x < y);
}

/* CSR pattern (sorted columns, diagonal included).  Call with col == NULL to
 * get rowptr (and nnz = rowptr[n_vert]), then again with col allocated. */
int oc_pattern(int d, int64_t n_vert, int64_t n_cell, const int32_t* conn, int64_t* rowptr, int32_t* col) {
  const int nv = d + 1;
  int64_t* off = (int64_t*)calloc(n_vert + 1, sizeof(int64_t));
  if (!off) return 1;
  for (int64_t e = 0; e < n_cell * nv; ++e) off[conn[e] + 1]++;
  for (int64_t v = 0; v < n_vert; ++v) off[v + 1] += off[v];
  int32_t* inc = (int32_t*)malloc((size_t)off[n_vert] * sizeof(int32_t));
  int64_t* cur = (int64_t*)malloc((size_t)n_vert * sizeof(int64_t));
  if (!inc || !cur) return 1;
  memcpy(cur, off, n_vert * sizeof(int64_t));
  for (int64_t c = 0; c < n_cell; ++c)
    for (int a = 0; a < nv; ++a) inc[cur[conn[c * nv + a]]++] = (int32_t)c;
  free(cur);
  int fail = 0;
  if (!col) {
#pragma omp parallel for schedule(static)
    for (int64_t v = 0; v < n_vert; ++v) {
      const int64_t m = (off[v + 1] - off[v]) * nv;
      int32_t stackbuf[256];
      int32_t* w = m <= 256 ? stackbuf : (int32_t*)malloc(m * sizeof(int32_t));
      int n = 0;
      for (int64_t e = off[v]; e < off[v + 1]; ++e)
        for (int a = 0; a < nv; ++a) w[n++] = conn[(int64_t)inc[e] * nv + a];
      qsort(w, n, sizeof(int32_t), cmp_i32);
      int u = 0;
      for (int i = 0; i < n; ++i)
        if (i == 0 || w[i] != w[i - 1]) ++u;
      if (n == 0) u = 1; /* isolated vertex: diagonal only */
      rowptr[v + 1] = u;
      if (w != stackbuf) free(w);
    }
    rowptr[0] = 0;
    for (int64_t v = 0; v < n_vert; ++v) rowptr[v + 1] += rowptr[v];
  } else {
#pragma omp parallel for schedule(static)
    for (int64_t v = 0; v < n_vert; ++v) {
      const int64_t m = (off[v + 1] - off[v]) * nv;
      int32_t stackbuf[256];
      int32_t* w = m <= 256 ? stackbuf : (int32_t*)malloc(m * sizeof(int32_t));
      int n = 0;
      for (int64_t e = off[v]; e < off[v + 1]; ++e)
        for (int a = 0; a < nv; ++a) w[n++] = conn[(int64_t)inc[e] * nv + a];
      qsort(w, n, sizeof(int32_t), cmp_i32);
      int64_t o = rowptr[v];
      if (n == 0) col[o++] = (int32_t)v;
      for (int i = 0; i < n; ++i)
        if (i == 0 || w[i] != w[i - 1]) col[o++] = w[i];
      if (o != rowptr[v + 1]) {
#pragma omp atomic write
        fail = 1;
      }
      if (w != stackbuf) free(w);
    }
  }
  free(inc);
  free(off);
  return fail;
}

static inline int64_t find(const int64_t* rowptr, const int32_t* col, int32_t i, int32_t j) {
  int64_t lo = rowptr[i], hi = rowptr[i + 1] - 1;
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    if (col[mid] < j) lo = mid + 1; else hi = mid;
  }
  return lo;
}

/* dR/du of inner(grad u, grad v) dx: cell loop, insertion by binary search. */
void oc_assemble_stiffness(int d, int64_t n_vert, const double* x, int64_t n_cell, const int32_t* conn,
                           const int64_t* rowptr, const int32_t* col, double* val) {
  const int nv = d + 1;
  memset(val, 0, (size_t)rowptr[n_vert] * sizeof(double));
#pragma omp parallel for schedule(static)
  for (int64_t c = 0; c < n_cell; ++c) {
    const int32_t* v = conn + c * nv;
    double vol, g[4][3];
    geom(d, x, v, &vol, g);
    for (int a = 0; a < nv; ++a)
      for (int b = 0; b < nv; ++b) {
        double s = 0.0;
        for (int k = 0; k < d; ++k) s += g[a][k] * g[b][k];
        const int64_t p = find(rowptr, col, v[a], v[b]);
#pragma omp atomic
        val[p] += vol * s;
      }
  }
}

/* R_i = int grad u . grad phi_i - int f phi_i ; no BC treatment. */
void oc_residual(int d, int64_t n_vert, const double* x, int64_t n_cell, const int32_t* conn,
                 const double* u, const double* f, double* r) {
  const int nv = d + 1;
  memset(r, 0, (size_t)n_vert * sizeof(double));
#pragma omp parallel for schedule(static)
  for (int64_t c = 0; c < n_cell; ++c) {
    const int32_t* v = conn + c * nv;
    double vol, g[4][3], gu[3] = {0, 0, 0};
    geom(d, x, v, &vol, g);
    for (int b = 0; b < nv; ++b)
      for (int k = 0; k < d; ++k) gu[k] += g[b][k] * u[v[b]];
    for (int a = 0; a < nv; ++a) {
      double s = 0.0;
      for (int k = 0; k < d; ++k) s += g[a][k] * gu[k];
      const double t = vol * s - f[c] * vol / nv;
#pragma omp atomic
      r[v[a]] += t;
    }
  }
}

/* Rows and columns of the Dirichlet set zeroed, diagonal 1 (pattern kept). */
void oc_eliminate_bc(int64_t n, const int64_t* rowptr, const int32_t* col, double* val, const uint8_t* isbc) {
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n; ++i)
    for (int64_t p = rowptr[i]; p < rowptr[i + 1]; ++p)
      if (isbc[i] || isbc[col[p]]) val[p] = (col[p] == i) ? 1.0 : 0.0;
}

void oc_spmv(int64_t n, const int64_t* rowptr, const int32_t* col, const double* val, const double* x, double* y) {
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n; ++i) {
    double s = 0.0;
    for (int64_t p = rowptr[i]; p < rowptr[i + 1]; ++p) s += val[p] * x[col[p]];
    y[i] = s;
  }
}

/* Jacobi-PCG from x = 0; stops when sqrt(r^T D^-1 r) <= max(rtol sqrt(b^T D^-1 b), atol)
 * or after max_it iterations.  Returns the iteration count; *res = sqrt(r^T D^-1 r). */
int oc_pcg_jacobi(int64_t n, const int64_t* rowptr, const int32_t* col, const double* val, const double* b,
                  double* x, double rtol, double atol, int max_it, double* res) {
  double* r = (double*)malloc(n * sizeof(double));
  double* p = (double*)malloc(n * sizeof(double));
  double* q = (double*)malloc(n * sizeof(double));
  double* dinv = (double*)malloc(n * sizeof(double));
  double rz = 0.0, zz = 0.0, bb = 0.0;
#pragma omp parallel for schedule(static) reduction(+ : rz, zz, bb)
  for (int64_t i = 0; i < n; ++i) {
    double dg = 1.0;
    for (int64_t k = rowptr[i]; k < rowptr[i + 1]; ++k)
      if (col[k] == i) dg = val[k];
    dinv[i] = 1.0 / dg;
    x[i] = 0.0;
    r[i] = b[i];
    const double z = dinv[i] * r[i];
    p[i] = z;
    rz += r[i] * z; zz += r[i] * z; bb += r[i] * z;
  }
  double tol = rtol * sqrt(bb);
  if (atol > tol) tol = atol;
  int it = 0;
  if (sqrt(zz) > tol) {
    while (it < max_it) {
      double pq = 0.0;
#pragma omp parallel for schedule(static) reduction(+ : pq)
      for (int64_t i = 0; i < n; ++i) {
        double s = 0.0;
        for (int64_t k = rowptr[i]; k < rowptr[i + 1]; ++k) s += val[k] * p[col[k]];
        q[i] = s;
        pq += p[i] * s;
      }
      const double alpha = pq != 0.0 ? rz / pq : 0.0;
      double rz1 = 0.0;
      zz = 0.0;
#pragma omp parallel for schedule(static) reduction(+ : rz1, zz)
      for (int64_t i = 0; i < n; ++i) {
        x[i] += alpha * p[i];
        r[i] -= alpha * q[i];
        const double z = dinv[i] * r[i];
        rz1 += r[i] * z; zz += r[i] * z;
      }
      ++it;
      if (sqrt(zz) <= tol) break;
      const double beta = rz != 0.0 ? rz1 / rz : 0.0;
      rz = rz1;
#pragma omp parallel for schedule(static)
      for (int64_t i = 0; i < n; ++i) p[i] = dinv[i] * r[i] + beta * p[i];
    }
  }
  *res = sqrt(zz);
  free(r); free(p); free(q); free(dinv);
  return it;
}

/* ---- auxiliary-lattice BPX preconditioner (oracle/bpx_oracle.py in C/OpenMP) -------------
 * M^-1 = D^-1 + theta sum_l P_l C_l P_l^T; nested multilinear lattices over the bounding box,
 * quantised vertex fractions (20 bits in 2-D, 12 in 3-D), single-precision 1/s in the mesh transfers (tau below), pinned
 * vertices masked, lattice nodes on the pinned
 * boundary dropped (30 % rule on the finest level, injection to the coarser ones).          */
#define OC_MAX_LEVELS 14
typedef struct {
  int dim, n_levels;
  int n[OC_MAX_LEVELS][3];
  int64_t nodes[OC_MAX_LEVELS];
  double* g[OC_MAX_LEVELS];
  double* e[OC_MAX_LEVELS];
  double* coef[OC_MAX_LEVELS];
  int64_t n_vert;
  int32_t* bin;    /* n_vert * dim */
  double* t;       /* n_vert * dim, quantised */
  uint8_t* pinned;
} oc_bpx;

static inline int64_t node_id(const int* n, int i, int j, int k) { return ((int64_t)k * (n[1] + 1) + j) * (n[0] + 1) + i; }

static void bpx_restrict_mesh(const oc_bpx* B, const double* val, int only_pinned, int use_pinned_filter, double* g) {
  const int d = B->dim, L = B->n_levels - 1;
  memset(g, 0, B->nodes[L] * sizeof(double));
#pragma omp parallel for schedule(static)
  for (int64_t v = 0; v < B->n_vert; ++v) {
    const int pin = B->pinned[v] != 0;
    if (use_pinned_filter && (only_pinned ? !pin : pin)) continue;
    const double r = val ? val[v] : 1.0;
    for (int c = 0; c < (1 << d); ++c) {
      double w = r;
      int ijk[3] = {0, 0, 0};
      for (int k = 0; k < d; ++k) {
        const int bit = (c >> k) & 1;
        w *= bit ? B->t[v * d + k] : 1.0 - B->t[v * d + k];
        ijk[k] = B->bin[v * d + k] + bit;
      }
      if (w != 0.0) {
#pragma omp atomic
        g[node_id(B->n[L], ijk[0], ijk[1], ijk[2])] += w;
      }
    }
  }
}

static void bpx_lattice_restrict(const oc_bpx* B, int l) {   /* g_l from g_{l+1} */
  const int* nc = B->n[l];
  const int* nf = B->n[l + 1];
  const int d = B->dim;
#pragma omp parallel for schedule(static)
  for (int64_t idx = 0; idx < B->nodes[l]; ++idx) {
    const int i = (int)(idx % (nc[0] + 1)), j = (int)((idx / (nc[0] + 1)) % (nc[1] + 1));
    const int k = (int)(idx / ((int64_t)(nc[0] + 1) * (nc[1] + 1)));
    double acc = 0.0;
    for (int dz = (d == 3 ? -1 : 0); dz <= (d == 3 ? 1 : 0); ++dz) {
      const int fk = d == 3 ? 2 * k + dz : 0;
      if (fk < 0 || fk > nf[2]) continue;
      for (int dy = -1; dy <= 1; ++dy) {
        const int fj = 2 * j + dy;
        if (fj < 0 || fj > nf[1]) continue;
        for (int dx = -1; dx <= 1; ++dx) {
          const int fi = 2 * i + dx;
          if (fi < 0 || fi > nf[0]) continue;
          acc += (dx ? 0.5 : 1.0) * (dy ? 0.5 : 1.0) * (dz ? 0.5 : 1.0) * B->g[l + 1][node_id(nf, fi, fj, fk)];
        }
      }
    }
    B->g[l][idx] = acc;
  }
}

static void bpx_lattice_prolong(const oc_bpx* B, int l) {    /* e_l = I e_{l-1} + coef_l g_l */
  const int* nf = B->n[l];
  const int d = B->dim;
#pragma omp parallel for schedule(static)
  for (int64_t idx = 0; idx < B->nodes[l]; ++idx) {
    double v = B->coef[l][idx] * B->g[l][idx];
    if (l > 0) {
      const int* nc = B->n[l - 1];
      const int i = (int)(idx % (nf[0] + 1)), j = (int)((idx / (nf[0] + 1)) % (nf[1] + 1));
      const int k = (int)(idx / ((int64_t)(nf[0] + 1) * (nf[1] + 1)));
      const int ci[2] = {i >> 1, (i + 1) >> 1}, cj[2] = {j >> 1, (j + 1) >> 1};
      const int ck[2] = {d == 3 ? k >> 1 : 0, d == 3 ? (k + 1) >> 1 : 0};
      double acc = 0.0;
      for (int a = 0; a < 2; ++a)
        for (int b = 0; b < 2; ++b)
          for (int c = 0; c < 2; ++c) acc += B->e[l - 1][node_id(nc, ci[a], cj[b], ck[c])];
      v += 0.125 * acc;
    }
    B->e[l][idx] = v;
  }
}

void oc_bpx_destroy(oc_bpx* B) {
  if (!B) return;
  for (int l = 0; l < B->n_levels; ++l) { free(B->g[l]); free(B->e[l]); free(B->coef[l]); }
  free(B->bin); free(B->t); free(B->pinned); free(B);
}

/* lo/hi: bounding box; pinned: per-vertex flag (may be NULL).  Returns NULL on a degenerate box. */
oc_bpx* oc_bpx_create(int d, int64_t n_vert, const double* x, const uint8_t* pinned, const double* lo, const double* hi,
                      int64_t n_vert_global, double spacing) {
  oc_bpx* B = (oc_bpx*)calloc(1, sizeof(oc_bpx));
  B->dim = d; B->n_vert = n_vert;
  double ext[3] = {0, 0, 0}, ext_max = 0.0, vol = 1.0;
  for (int k = 0; k < d; ++k) { ext[k] = hi[k] - lo[k]; if (ext[k] > ext_max) ext_max = ext[k]; vol *= ext[k]; }
  if (!(vol > 0.0)) { free(B); return NULL; }
  const double h = pow(vol / (double)n_vert_global, 1.0 / d);
  double target = ext_max / (spacing * h);
  if (target < 2.0) target = 2.0;
  int best_m0 = 2, best_lv = 1;
  double best = 1e300;
  const int max_bins = d == 3 ? 511 : 4095;         /* the packed coordinates' bin fields (femo_amd/csrc/pc_plan.cpp) */
  for (int m0 = 2; m0 <= 3; ++m0)
    for (int lv = 1; lv <= 12; ++lv) {
      if ((m0 << (lv - 1)) > max_bins) continue;
      const double score = fabs(log(m0 * ldexp(1.0, lv - 1) / target));
      if (score < best) { best = score; best_m0 = m0; best_lv = lv; }
    }
  B->n_levels = best_lv;
  double H[OC_MAX_LEVELS];
  for (int l = 0; l < best_lv; ++l) {
    H[l] = ext_max / (double)(best_m0 << l);
    B->nodes[l] = 1;
    for (int k = 0; k < 3; ++k) {
      int base = 0;
      if (k < d) { base = (int)lround(best_m0 * ext[k] / ext_max); if (base < 1) base = 1; }
      B->n[l][k] = k < d ? base << l : 0;
      B->nodes[l] *= B->n[l][k] + 1;
    }
    B->g[l] = (double*)calloc(B->nodes[l], sizeof(double));
    B->e[l] = (double*)calloc(B->nodes[l], sizeof(double));
    B->coef[l] = (double*)calloc(B->nodes[l], sizeof(double));
  }
  const int L = best_lv - 1;
  B->bin = (int32_t*)malloc(n_vert * d * sizeof(int32_t));
  B->t = (double*)malloc(n_vert * d * sizeof(double));
  B->pinned = (uint8_t*)calloc(n_vert, 1);
  if (pinned) memcpy(B->pinned, pinned, n_vert);
  /* keep rule from the exact fractions, then quantise */
#pragma omp parallel for schedule(static)
  for (int64_t v = 0; v < n_vert; ++v)
    for (int k = 0; k < d; ++k) {
      const double g = (x[v * d + k] - lo[k]) * (B->n[L][k] / ext[k]);
      int b = (int)floor(g);
      if (b < 0) b = 0;
      if (b > B->n[L][k] - 1) b = B->n[L][k] - 1;
      double t = g - b;
      if (t < 0.0) t = 0.0;
      if (t > 1.0) t = 1.0;
      B->bin[v * d + k] = b;
      B->t[v * d + k] = t;
    }
  double* wf = B->g[L];
  double* wd = B->e[L];
  bpx_restrict_mesh(B, NULL, 0, 1, wf);
  bpx_restrict_mesh(B, NULL, 1, 1, wd);
  const double theta = 0.6;
  for (int64_t i = 0; i < B->nodes[L]; ++i) {
    const double c = theta * (d == 3 ? 3.0 / (8.0 * H[L]) : 3.0 / 8.0);
    B->coef[L][i] = (wf[i] > 0.0 && wd[i] <= 0.3 * (wf[i] + wd[i])) ? c : 0.0;
  }
  for (int l = L - 1; l >= 0; --l) {
    const double c = theta * (d == 3 ? 3.0 / (8.0 * H[l]) : 3.0 / 8.0);
    const int* nc = B->n[l];
    for (int64_t idx = 0; idx < B->nodes[l]; ++idx) {
      const int i = (int)(idx % (nc[0] + 1)), j = (int)((idx / (nc[0] + 1)) % (nc[1] + 1));
      const int k = (int)(idx / ((int64_t)(nc[0] + 1) * (nc[1] + 1)));
      B->coef[l][idx] = B->coef[l + 1][node_id(B->n[l + 1], 2 * i, 2 * j, d == 3 ? 2 * k : 0)] != 0.0 ? c : 0.0;
    }
  }
#pragma omp parallel for schedule(static)
  for (int64_t v = 0; v < n_vert * d; ++v) {
    /* packed fractions of the device: 20 bits in 2-D, 12 bits in 3-D (femo_internal.h, round 5) */
    const double q = d == 3 ? 4096.0 : 1048576.0;
    double tq = floor(B->t[v] * q + 0.5);
    if (tq > q - 1.0) tq = q - 1.0;
    B->t[v] = tq / q;
  }
  return B;
}

int oc_bpx_levels(const oc_bpx* B) { return B->n_levels; }

/* z = dinv r + P (sum_l ...) P^T r */
/* tau = s fl32(1/s), s = 1/sqrt(diag): the device carries 1/s of the two mesh transfers in single precision
 * (bpx_oracle.py::single_precision_scaling) */
static inline double bpx_tau(double dinv) {
  const double s = 1.0 / sqrt(1.0 / dinv);
  return s * (double)(float)(1.0 / s);
}

void oc_bpx_apply(oc_bpx* B, const double* dinv, const double* r, double* z) {
  const int L = B->n_levels - 1, d = B->dim;
  double* rt = (double*)malloc(B->n_vert * sizeof(double));
#pragma omp parallel for schedule(static)
  for (int64_t v = 0; v < B->n_vert; ++v) rt[v] = bpx_tau(dinv[v]) * r[v];
  bpx_restrict_mesh(B, rt, 0, 1, B->g[L]);
  free(rt);
  for (int l = L - 1; l >= 0; --l) bpx_lattice_restrict(B, l);
  for (int l = 0; l <= L; ++l) bpx_lattice_prolong(B, l);
  const double* e = B->e[L];
#pragma omp parallel for schedule(static)
  for (int64_t v = 0; v < B->n_vert; ++v) {
    double zz = dinv[v] * r[v];
    if (!B->pinned[v]) {
      double sum = 0.0;
      for (int c = 0; c < (1 << d); ++c) {
        double w = 1.0;
        int ijk[3] = {0, 0, 0};
        for (int k = 0; k < d; ++k) {
          const int bit = (c >> k) & 1;
          w *= bit ? B->t[v * d + k] : 1.0 - B->t[v * d + k];
          ijk[k] = B->bin[v * d + k] + bit;
        }
        sum += w * e[node_id(B->n[L], ijk[0], ijk[1], ijk[2])];
      }
      zz += bpx_tau(dinv[v]) * sum;
    }
    z[v] = zz;
  }
}

/* BPX-PCG from x = 0 with the engine's stopping rule: relative in the norm of the preconditioner,
 * sqrt(r^T M^-1 r) <= rtol sqrt(b^T M^-1 b); absolute in the Jacobi norm, sqrt(r^T D^-1 r) <= atol. */
int oc_pcg_bpx(oc_bpx* B, int64_t n, const int64_t* rowptr, const int32_t* col, const double* val, const double* b,
               double* x, double rtol, double atol, int max_it, double* res, double atol_pc) {
  double* r = (double*)malloc(n * sizeof(double));
  double* p = (double*)malloc(n * sizeof(double));
  double* q = (double*)malloc(n * sizeof(double));
  double* z = (double*)malloc(n * sizeof(double));
  double* dinv = (double*)malloc(n * sizeof(double));
  double bb = 0.0;
#pragma omp parallel for schedule(static) reduction(+ : bb)
  for (int64_t i = 0; i < n; ++i) {
    double dg = 1.0;
    for (int64_t k = rowptr[i]; k < rowptr[i + 1]; ++k)
      if (col[k] == i) dg = val[k];
    dinv[i] = 1.0 / dg;
    x[i] = 0.0;
    r[i] = b[i];
    bb += r[i] * dinv[i] * r[i];
  }
  double rho = bb;
  int it = 0;
  if (sqrt(rho) > atol && rho > 0.0) {
    oc_bpx_apply(B, dinv, r, z);
    double rz = 0.0;
#pragma omp parallel for schedule(static) reduction(+ : rz)
    for (int64_t i = 0; i < n; ++i) { p[i] = z[i]; rz += r[i] * z[i]; }
    double tol2 = rtol * rtol * rz;
    if (atol_pc * atol_pc > tol2) tol2 = atol_pc * atol_pc;   /* sqrt(r^T M^-1 r) <= max(rtol sqrt(b^T M^-1 b), atol_pc) */
    while (it < max_it && rz > tol2) {
      double pq = 0.0;
#pragma omp parallel for schedule(static) reduction(+ : pq)
      for (int64_t i = 0; i < n; ++i) {
        double s = 0.0;
        for (int64_t k = rowptr[i]; k < rowptr[i + 1]; ++k) s += val[k] * p[col[k]];
        q[i] = s;
        pq += p[i] * s;
      }
      const double alpha = pq != 0.0 ? rz / pq : 0.0;
      rho = 0.0;
#pragma omp parallel for schedule(static) reduction(+ : rho)
      for (int64_t i = 0; i < n; ++i) {
        x[i] += alpha * p[i];
        r[i] -= alpha * q[i];
        rho += r[i] * dinv[i] * r[i];
      }
      ++it;
      if (atol > 0.0 && sqrt(rho) <= atol) break;
      oc_bpx_apply(B, dinv, r, z);
      double rz1 = 0.0;
#pragma omp parallel for schedule(static) reduction(+ : rz1)
      for (int64_t i = 0; i < n; ++i) rz1 += r[i] * z[i];
      if (rz1 <= tol2) break;
      const double beta = rz != 0.0 ? rz1 / rz : 0.0;
      rz = rz1;
#pragma omp parallel for schedule(static)
      for (int64_t i = 0; i < n; ++i) p[i] = z[i] + beta * p[i];
    }
  }
  *res = sqrt(rho);
  free(r); free(p); free(q); free(z); free(dinv);
  return it;
}

/* J = 1/2 int (u-u_d)^2 + alpha/2 int f^2 */
double oc_functional(int d, const double* x, int64_t n_cell, const int32_t* conn, const double* u,
                     const double* f, const double* ud, double alpha) {
  const int nv = d + 1;
  double J = 0.0;
#pragma omp parallel for schedule(static) reduction(+ : J)
  for (int64_t c = 0; c < n_cell; ++c) {
    const int32_t* v = conn + c * nv;
    double vol, g[4][3], s1 = 0.0, s2 = 0.0;
    geom(d, x, v, &vol, g);
    for (int a = 0; a < nv; ++a) { const double e = u[v[a]] - ud[v[a]]; s1 += e; s2 += e * e; }
    J += 0.5 * vol / ((d + 1) * (d + 2)) * (s2 + s1 * s1) + 0.5 * alpha * f[c] * f[c] * vol;
  }
  return J;
}

void oc_functional_du(int d, int64_t n_vert, const double* x, int64_t n_cell, const int32_t* conn,
                      const double* u, const double* ud, double* gout) {
  const int nv = d + 1;
  memset(gout, 0, (size_t)n_vert * sizeof(double));
#pragma omp parallel for schedule(static)
  for (int64_t c = 0; c < n_cell; ++c) {
    const int32_t* v = conn + c * nv;
    double vol, g[4][3], e[4], s1 = 0.0;
    geom(d, x, v, &vol, g);
    for (int a = 0; a < nv; ++a) { e[a] = u[v[a]] - ud[v[a]]; s1 += e[a]; }
    for (int a = 0; a < nv; ++a) {
      const double t = vol / ((d + 1) * (d + 2)) * (e[a] + s1);
#pragma omp atomic
      gout[v[a]] += t;
    }
  }
}

void oc_functional_df(int d, const double* x, int64_t n_cell, const int32_t* conn, const double* f,
                      double alpha, double* gout) {
  const int nv = d + 1;
#pragma omp parallel for schedule(static)
  for (int64_t c = 0; c < n_cell; ++c) {
    double vol, g[4][3];
    geom(d, x, conn + c * nv, &vol, g);
    gout[c] = alpha * f[c] * vol;
  }
}

/* out[c] = sum_a dRdf[v_a, c] lam[v_a] = -|T_c|/(d+1) sum_a lam[v_a] */
void oc_dRdfT_apply(int d, const double* x, int64_t n_cell, const int32_t* conn, const double* lam, double* out) {
  const int nv = d + 1;
#pragma omp parallel for schedule(static)
  for (int64_t c = 0; c < n_cell; ++c) {
    const int32_t* v = conn + c * nv;
    double vol, g[4][3], s = 0.0;
    geom(d, x, v, &vol, g);
    for (int a = 0; a < nv; ++a) s += lam[v[a]];
    out[c] = -vol / nv * s;
  }
}

/* ---- nonlinear Poisson + Nitsche (BASELINE config 5): femo_oracle.py::nl_residual / nl_jacobian in C/OpenMP -------------
 * R(u; f) = int grad u . grad v + int u^3 v - int f v  + the symmetric (sgn = +1, penalty beta / h_E) or unsymmetric (sgn = -1)
 * Nitsche terms on the exterior facets, u_ex the CG1 interpolant of the Dirichlet data (run_nonlinear_poisson_opt.py:88-116).
 * The cubic term uses the closed-form P1 monomial integrals int prod phi^alpha = |T| d! prod(alpha!) / (|alpha| + d)!.
 * Round 5: so that bench.py's config-5 CPU baseline runs at the configuration's own size on all host cores. */
static double fact(int n) { double f = 1.0; for (int i = 2; i <= n; ++i) f *= i; return f; }

/* T3[a][b][c][e] = int phi_a phi_b phi_c phi_e / |T| */
static void cubic_table(int d, double T3[4][4][4][4]) {
  const int nv = d + 1;
  for (int a = 0; a < nv; ++a) for (int b = 0; b < nv; ++b) for (int c = 0; c < nv; ++c) for (int e = 0; e < nv; ++e) {
    int mult[4] = {0, 0, 0, 0};
    mult[a]++; mult[b]++; mult[c]++; mult[e]++;
    double p = fact(d);
    for (int k = 0; k < nv; ++k) p *= fact(mult[k]);
    T3[a][b][c][e] = p / fact(4 + d);
  }
}

typedef struct { double meas, hE, gn[4], nrm[3]; } oc_facet;

/* facet opposite local vertex k of the cell with gradients g and volume vol */
static void facet_piece(int d, const double* x, const int32_t* v, double vol, double g[4][3], int k, oc_facet* F) {
  const int nv = d + 1;
  double ng = 0.0;
  for (int j = 0; j < d; ++j) ng += g[k][j] * g[k][j];
  ng = sqrt(ng);
  for (int j = 0; j < d; ++j) F->nrm[j] = -g[k][j] / ng;          /* outward normal */
  F->meas = d * vol * ng;                                          /* |F_k| = d |T| |grad phi_k| */
  double hE = 0.0;
  for (int a = 0; a < nv; ++a)
    for (int b = a + 1; b < nv; ++b) {
      double s = 0.0;
      for (int j = 0; j < d; ++j) { const double t = x[d * (int64_t)v[a] + j] - x[d * (int64_t)v[b] + j]; s += t * t; }
      s = sqrt(s);
      if (s > hE) hE = s;
    }
  F->hE = hE;
  for (int a = 0; a < nv; ++a) {
    double s = 0.0;
    for (int j = 0; j < d; ++j) s += g[a][j] * F->nrm[j];
    F->gn[a] = s;
  }
}

void oc_nl_residual(int d, int64_t n_vert, const double* x, int64_t n_cell, const int32_t* conn, const double* u,
                    const double* f, const double* uex, const uint8_t* bmask, double beta, double sgn, double* r) {
  const int nv = d + 1;
  double T3[4][4][4][4];
  cubic_table(d, T3);
  memset(r, 0, (size_t)n_vert * sizeof(double));
#pragma omp parallel for schedule(static)
  for (int64_t c = 0; c < n_cell; ++c) {
    const int32_t* v = conn + c * nv;
    double vol, g[4][3], gu[3] = {0, 0, 0}, ue[4], Re[4];
    geom(d, x, v, &vol, g);
    for (int b = 0; b < nv; ++b) {
      ue[b] = u[v[b]];
      for (int k = 0; k < d; ++k) gu[k] += g[b][k] * ue[b];
    }
    for (int a = 0; a < nv; ++a) {
      double s = 0.0, cub = 0.0;
      for (int k = 0; k < d; ++k) s += g[a][k] * gu[k];
      for (int b = 0; b < nv; ++b) for (int cc = 0; cc < nv; ++cc) for (int e = 0; e < nv; ++e) cub += T3[a][b][cc][e] * ue[b] * ue[cc] * ue[e];
      Re[a] = vol * s - f[c] * vol / nv + vol * cub;
    }
    if (bmask && bmask[c]) {
      for (int k = 0; k < nv; ++k) {
        if (!((bmask[c] >> k) & 1)) continue;
        oc_facet F;
        facet_piece(d, x, v, vol, g, k, &F);
        double dun = 0.0, s_on = 0.0, er[4];
        for (int a = 0; a < nv; ++a) { er[a] = ue[a] - uex[v[a]]; dun += F.gn[a] * ue[a]; if (a != k) s_on += er[a]; }
        const double mean_e = s_on / d;
        for (int a = 0; a < nv; ++a) Re[a] += sgn * F.gn[a] * (-mean_e) * F.meas;               /* nitsche_2 */
        for (int a = 0; a < nv; ++a) {
          if (a == k) continue;
          Re[a] += -dun * F.meas / d;                                                           /* nitsche_1 */
          Re[a] += beta / F.hE * F.meas / (d * (d + 1)) * (er[a] + s_on);                       /* penalty (facet mass) */
        }
      }
    }
    for (int a = 0; a < nv; ++a) {
#pragma omp atomic
      r[v[a]] += Re[a];
    }
  }
}

void oc_nl_jacobian(int d, int64_t n_vert, const double* x, int64_t n_cell, const int32_t* conn, const double* u,
                    const uint8_t* bmask, double beta, double sgn, const int64_t* rowptr, const int32_t* col, double* val) {
  const int nv = d + 1;
  double T3[4][4][4][4];
  cubic_table(d, T3);
  memset(val, 0, (size_t)rowptr[n_vert] * sizeof(double));
#pragma omp parallel for schedule(static)
  for (int64_t c = 0; c < n_cell; ++c) {
    const int32_t* v = conn + c * nv;
    double vol, g[4][3], ue[4], Ke[4][4];
    geom(d, x, v, &vol, g);
    for (int b = 0; b < nv; ++b) ue[b] = u[v[b]];
    for (int a = 0; a < nv; ++a)
      for (int b = 0; b < nv; ++b) {
        double s = 0.0, q = 0.0;
        for (int k = 0; k < d; ++k) s += g[a][k] * g[b][k];
        for (int cc = 0; cc < nv; ++cc) for (int e = 0; e < nv; ++e) q += T3[a][b][cc][e] * ue[cc] * ue[e];
        Ke[a][b] = vol * (s + 3.0 * q);
      }
    if (bmask && bmask[c]) {
      for (int k = 0; k < nv; ++k) {
        if (!((bmask[c] >> k) & 1)) continue;
        oc_facet F;
        facet_piece(d, x, v, vol, g, k, &F);
        for (int a = 0; a < nv; ++a) {
          if (a == k) continue;
          for (int b = 0; b < nv; ++b) {
            Ke[a][b] += -F.gn[b] * F.meas / d;                 /* nitsche_1: -(g_b.n) int_F phi_a */
            Ke[b][a] += -sgn * F.gn[b] * F.meas / d;           /* nitsche_2: -sgn (g_b.n) int_F phi_a */
          }
          for (int b = 0; b < nv; ++b) {
            if (b == k) continue;
            Ke[a][b] += beta / F.hE * F.meas / (d * (d + 1)) * (a == b ? 2.0 : 1.0);
          }
        }
      }
    }
    for (int a = 0; a < nv; ++a)
      for (int b = 0; b < nv; ++b) {
        const int64_t p = find(rowptr, col, v[a], v[b]);
#pragma omp atomic
        val[p] += Ke[a][b];
      }
  }
}

int oc_num_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

void oc_set_num_threads(int n) {
#ifdef _OPENMP
  omp_set_num_threads(n);
#else
  (void)n;
#endif
}
