// Reissner-Mindlin shell, CG2^3 x CG1^3 on flat triangular facets (BASELINE config 3;
// replaces what examples/test_shell_m3l/shell_pde.py:219-332 obtains from shell_analysis_fenicsx + dolfinx + MUMPS).
// The formulation is restated and pinned in oracle/shell_oracle.py (Scordelis-Lo, Kirchhoff plate, rigid modes);
// the kernels here are checked against it entry by entry (tests/test_gpu_shell.py).
//
// What is here (DESIGN.md section 8 has the measurements and the versions that came before):
//   * degrees of freedom and the CSR pattern of the 27 x 27 element couplings are built on the host (Python,
//     femo_amd/fea/shell.py) and handed over as plain arrays, with the CSR position of every element entry;
//   * assembly: one thread per (cell, element column) forms the column from the facet frame and the quadrature
//     points in registers (B^T D B, nine strain rows) and adds its 27 entries with fp64 atomics;
//   * operator: the three dofs of a node share their columns, so the matrix is read as 3 x 3 blocks -- straight from
//     the CSR values (k_bcsr3_spmv) or, in the CG loop, from a block-SELL copy (k_bsell_spmv);
//   * solve: CG with device-side scalars (consumers fold the producers' per-block partials, the host polls a flag)
//     and a nested-lattice preconditioner: 3 x 3 point blocks of K as the smoother, 6 x 6 Galerkin node blocks on the
//     lattice levels, an exact dense solve (Galerkin operator formed on the device, blocked Cholesky on the fp64 matrix
//     cores) on the coarsest level kept -- 252 iterations at 1.97 M dofs where Jacobi needs ~1e5;
//   * partials and outputs: (dR/dh)^T lambda element by element from the strains of w and lambda, load and its
//     transpose, compliance, mass, elastic energy, the aggregated von Mises stress and its projection onto the vertices.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "femo_internal.h"

struct femo_shell {
  femo_ctx* ctx = nullptr;
  int64_t n_vert = 0, n_cell = 0, n_edge = 0, n_unode = 0, n_dof = 0, nnz = 0;
  double* d_x = nullptr;
  int32_t *d_conn = nullptr, *d_cedge = nullptr, *d_cols = nullptr, *d_epos = nullptr;
  int64_t* d_rowptr = nullptr;
  // node-block view of the pattern (dofs 3 b .. 3 b + 2 of a node share their columns, which come in runs of three):
  // block-row offsets and the first scalar column of every 3 x 3 block; nullptr if the pattern is not of that shape
  int64_t n_bnode = 0;
  int64_t* d_brow = nullptr;
  int32_t* d_bcols = nullptr;
  // block-SELL-16 copy of the matrix for the CG loop (k_bsell_spmv): slices of 16 consecutive block rows, per slice
  // and block slot the 16 column indices and the 9 x 16 values component by component (lane = block row)
  int64_t n_bslice = 0, bsell_blocks = 0;               // slices, 16-block groups (= sum of slots over slices)
  int64_t* d_bs_off = nullptr;                          // first 16-block group of every slice (n_bslice + 1)
  int32_t* d_bs_cols = nullptr;
  double* d_bs_vals = nullptr;
  uint64_t bs_vals_uid = 0, bs_vals_gen = 0;            // the stiffness the copy was made from
  // CG workspace
  double *d_r = nullptr, *d_p = nullptr, *d_q = nullptr, *d_dinv = nullptr, *d_scal = nullptr, *d_part = nullptr;
  int32_t* d_flag = nullptr;
  // lattice preconditioner (femo_shell_pc_create): P in ELL form (8 trilinear weights per level and dof), P^T as CSR
  int pc_width = 0, pc_levels = 0;
  int64_t n_lat = 0, pc_nodes = 0;
  std::vector<int64_t> level_off;                       // node offsets of the levels (pc_levels + 1 entries)
  int32_t *d_ell_idx = nullptr, *d_par_cols = nullptr, *d_chi_cols = nullptr;
  double *d_ell_w = nullptr, *d_par_vals = nullptr, *d_chi_vals = nullptr;
  double *d_coarse = nullptr, *d_t = nullptr, *d_e = nullptr, *d_z = nullptr;
  double* d_cblk = nullptr;                             // 6 x 6 inverse Galerkin blocks of the nodes above the coarse-solve level
  int32_t* d_lvl_node = nullptr;                        // levels above the coarse solve, per level and POINT: the eight lattice
  double* d_lvl_w = nullptr;                            // nodes and weights, contiguous ([level][point][8]; the ELL rows interleave
                                                        // all levels of a dof: 3 cache lines per access, 52 GB fetched by the
                                                        // node-block kernel at 1.97 M dofs)
  bool blk_ready = false;
  float* d_dinv3 = nullptr;                             // 3 x 3 inverse diagonal blocks of the points (finest-level smoother), single
                                                        // precision: a smoother rounded at 6e-8 is as good a smoother, products and sums stay fp64
                                                        // (round 3: 72 -> 36 bytes per point in both kernels that read it, every iteration)
  bool dinv3_ready = false;
  int32_t* d_fin_idx = nullptr;                         // the finest level's eight (unknown, weight) pairs per POINT (a P2 node's
  float* d_fin_w = nullptr;                             // three displacements / a vertex's three rotations share them); single
                                                        // precision like d_ptp_vals -- the same rounded numbers in both directions, so M^-1 stays symmetric
  int64_t* d_ptp_rowptr = nullptr;                      // P_L^T by (finest lattice node, field group): points and weights
  int32_t* d_ptp_cols = nullptr;
  float* d_ptp_vals = nullptr;
  int64_t *d_par_rowptr = nullptr, *d_chi_rowptr = nullptr;
  uint64_t pc_vals_uid = 0, pc_vals_gen = 0, pc_mask_hash = 0;     // what d_coarse was computed for
  // exact coarse solve (femo_shell_pc_coarse): on level cs_level the Galerkin operator P^T K P is formed as a dense
  // matrix, factorised (blocked Cholesky + triangular inverse, below) and A^-1 = L^-T L^-1 applied in place of the
  // diagonal levels 0 .. cs_level
  int cs_level = -1;
  int64_t cs_n = 0, cs_N = 0, cs_items = 0;            // unknowns of the level (6 x nodes), padded to 64s, items of the Galerkin kernel
  int64_t cs_max_item = 0;                             // points of the largest item
  bool cs_ready = false;                               // d_cs_A holds the factors of the inverse for the current stiffness and mask
  int64_t* d_cd_rowptr = nullptr;                      // composite restriction finest lattice -> levels cs_level .. L - 2
  int32_t* d_cd_cols = nullptr;
  double* d_cd_vals = nullptr;
  int32_t *d_cs_xyz = nullptr, *d_cs_pts = nullptr, *d_cs_nbr = nullptr, *d_cs_info = nullptr, *d_cs_pcell = nullptr;
  int64_t* d_cs_ptr = nullptr;
  double *d_cs_A = nullptr, *d_cs_tmp = nullptr;       // L^-T above / L^-1 below the diagonal (row-major, N x N); L^-1 g
  float* d_cs_Af = nullptr;                            // the same factors in single precision: what the iteration applies
  double* d_cs_dinv = nullptr;                         // inverses of the diagonal tiles of L
  double* d_cs_T = nullptr;                            // scratch of the level-wise triangular inversion (N x N)
  // Hermite-type lattice spaces (femo_shell_pc_hermite; used when the coarse solve and the node blocks are ready, else the
  // trilinear data above takes over): finest transfer weights per (point, corner), P_L^T rows per finest node (displacement
  // points, then rotation points), (a, b, c) of the node-level transfers, composed weights of the levels above the
  // coarse solve ([level][point][8], finest included) and of the coarse-solve level, composite restriction
  bool hermite = false, hermite_on = false;             // enabled (uploaded and not fallen back) / in use for the current stiffness
  bool hermite_loaded = false;                          // the device arrays exist (guards a second upload; survives a fallback)
  // Weight of the node-block levels in the additive sum (round 4).  The levels between the coarse solve and the finest
  // lattice overlap each other and the point-block smoother; summed with weight 1 they overshoot (the same reason the
  // Poisson BPX carries theta = 0.6).  Measured on the roof, iterations per solve for weights 1 / 0.5 / 0.3 / 0.25 / 0.12:
  // 362^2 (three block levels) 145 / 113 / 105 / 105 / 118, 256^2 118 / 107 / 104 / 102, 128^2 (two) 113 / 103 / 101 / 101,
  // 64^2 (one) 104 / 100 / 101 / 101; trilinear spaces at 362^2: 252 / 205 / 200.  A weight on the coarse solve (0.7, 2, 4)
  // or per-level weights change nothing beyond that.
  double w_levels = 0.3, w_coarse = 1.0;
  // items of k_pc_galerkin_blocks_w (femo_shell_pc_block_items): points grouped by (level above the coarse solve, cell)
  int64_t bi_items = 0;
  int64_t* d_bi_ptr = nullptr;
  int32_t *d_bi_lvl = nullptr, *d_bi_pts = nullptr, *d_bi_pcell = nullptr;
  uint8_t* d_fixbits = nullptr;
  // the Dirichlet mask of the last solve on the device, kept while the caller's array hashes the same (round 5: a solve used
  // to allocate, upload and free it, and to hash it byte by byte for the preconditioner's cache: 3 ms of idle device per solve)
  uint8_t* d_fixed_kept = nullptr;
  uint64_t fixed_kept_hash = 0;
  float4* d_fin_w4 = nullptr;
  int64_t* d_hp_rowptr = nullptr;
  int32_t* d_hp_cols = nullptr;
  float4* d_hp_w4 = nullptr;
  double *d_par_w5 = nullptr, *d_chi_w5 = nullptr;
  float4 *d_lvl_w4 = nullptr, *d_cs_w4 = nullptr;
  int64_t* d_hd_rowptr = nullptr;
  int32_t* d_hd_cols = nullptr;
  double* d_hd_w5 = nullptr;
  // penalty boundary terms (femo_shell_set_penalty): tagged edges, their coefficient and the CSR positions of their entries
  int64_t pen_n = 0;
  int32_t *d_pen_nodes = nullptr, *d_pen_pos = nullptr;
  double* d_pen_coef = nullptr;
  // partition (femo_shell_set_partition): this handle holds the cells that touch a point the rank
  // owns.  d_owned flags the owned points (dofs 3 p .. 3 p + 2); the rows of the others are zeroed after assembly, so
  // that K, right-hand sides and residuals are the rank's share and sums over the ranks are the global objects.  The
  // halo plan refreshes the entries of the points owned elsewhere.
  uint8_t* d_owned = nullptr;
  uint8_t* d_cell_owned = nullptr;                       // femo_shell_set_owned_cells: the cells whose scalar outputs this rank integrates
  int n_nbr = 0;
  std::vector<int32_t> nbr;
  std::vector<int64_t> send_ptr, recv_ptr;
  int32_t *d_send_idx = nullptr, *d_recv_idx = nullptr;
  double *d_send_buf = nullptr, *d_recv_buf = nullptr;
};

// relative weights of the additive parts of the preconditioner (femo_shell_pc_weights; applied at set-up time: the inverse
// node blocks of the levels above the coarse solve and the factor of the coarse inverse are scaled once per stiffness)
static double shell_level_weight(const femo_shell* s, int) { return s->w_levels; }
static double shell_coarse_weight(const femo_shell* s) { return s->w_coarse; }

// plain view of the device arrays for kernels
struct femo_shell_view {
  int64_t n_vert, n_cell, n_unode;
  const double* x;
  const int32_t *conn, *cedge;
  // partitioned shells (femo_shell_set_owned_cells): 1 for the cells this rank integrates in scalar outputs (each cell of the
  // whole mesh belongs to exactly one rank; the values are summed over the ranks), nullptr on one rank.  Gradients are
  // formed over ALL local cells: every cell around a point the rank owns is local, so their entries there are complete.
  const uint8_t* cell_owned;
};
__device__ __forceinline__ double shell_value_weight(const femo_shell_view& S, int64_t c) {
  return (S.cell_owned == nullptr || c >= S.n_cell || S.cell_owned[c]) ? 1.0 : 0.0;
}

namespace {

constexpr int SH_BLOCK = 256;
constexpr int SH_MAXPART = 4096;

// quadrature rules of oracle/shell_oracle.py: Dunavant degree 4 (in-plane terms), degree 2 (shear)
__constant__ double c_lam6[6][3] = {
    {0.108103018168070, 0.445948490915965, 0.445948490915965}, {0.445948490915965, 0.108103018168070, 0.445948490915965},
    {0.445948490915965, 0.445948490915965, 0.108103018168070}, {0.816847572980459, 0.091576213509771, 0.091576213509771},
    {0.091576213509771, 0.816847572980459, 0.091576213509771}, {0.091576213509771, 0.091576213509771, 0.816847572980459}};
__constant__ double c_w6[6] = {0.223381589678011, 0.223381589678011, 0.223381589678011,
                               0.109951743655322, 0.109951743655322, 0.109951743655322};
__constant__ double c_lam3[3][3] = {{2.0 / 3, 1.0 / 6, 1.0 / 6}, {1.0 / 6, 2.0 / 3, 1.0 / 6}, {1.0 / 6, 1.0 / 6, 2.0 / 3}};

struct Facet {
  double e1[3], e2[3], e3[3], area;
  double gl[3][2];            // tangent gradients of the barycentric coordinates
};

__device__ __forceinline__ void facet_frame(const double* __restrict__ x, const int32_t* __restrict__ conn, int64_t c, Facet& F) {
  double p[3][3];
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int k = 0; k < 3; ++k) p[a][k] = x[(int64_t)conn[c * 3 + a] * 3 + k];
  double t1[3], t2[3], n[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) { t1[k] = p[1][k] - p[0][k]; t2[k] = p[2][k] - p[0][k]; }
  n[0] = t1[1] * t2[2] - t1[2] * t2[1]; n[1] = t1[2] * t2[0] - t1[0] * t2[2]; n[2] = t1[0] * t2[1] - t1[1] * t2[0];
  const double dbl = sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
  const double l1 = sqrt(t1[0] * t1[0] + t1[1] * t1[1] + t1[2] * t1[2]);
#pragma unroll
  for (int k = 0; k < 3; ++k) { F.e3[k] = n[k] / dbl; F.e1[k] = t1[k] / l1; }
  F.e2[0] = F.e3[1] * F.e1[2] - F.e3[2] * F.e1[1];
  F.e2[1] = F.e3[2] * F.e1[0] - F.e3[0] * F.e1[2];
  F.e2[2] = F.e3[0] * F.e1[1] - F.e3[1] * F.e1[0];
  F.area = 0.5 * dbl;
  // tangent coordinates of the vertices: (0,0), (a,0), (b,c)
  const double a = t1[0] * F.e1[0] + t1[1] * F.e1[1] + t1[2] * F.e1[2];
  const double b = t2[0] * F.e1[0] + t2[1] * F.e1[1] + t2[2] * F.e1[2];
  const double cc = t2[0] * F.e2[0] + t2[1] * F.e2[1] + t2[2] * F.e2[2];
  const double X[3] = {0.0, a, b}, Y[3] = {0.0, 0.0, cc};
  const double det = 2.0 * F.area;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int j = (i + 1) % 3, k = (i + 2) % 3;
    F.gl[i][0] = (Y[j] - Y[k]) / det;
    F.gl[i][1] = (X[k] - X[j]) / det;
  }
}

// tangent gradient of P2 shape function a at barycentric point lam (vertices 0..2, then edges (0,1), (1,2), (2,0))
__device__ __forceinline__ void p2_grad(const Facet& F, const double* lam, int a, double& g1, double& g2) {
  if (a < 3) {
    const double d = 4.0 * lam[a] - 1.0;
    g1 = d * F.gl[a][0]; g2 = d * F.gl[a][1];
  } else {
    const int i = a - 3, j = (a - 2) % 3;
    g1 = 4.0 * (lam[j] * F.gl[i][0] + lam[i] * F.gl[j][0]);
    g2 = 4.0 * (lam[j] * F.gl[i][1] + lam[i] * F.gl[j][1]);
  }
}

__device__ __forceinline__ double p2_value(const double* lam, int a) {
  if (a < 3) return lam[a] * (2.0 * lam[a] - 1.0);
  const int i = a - 3, j = (a - 2) % 3;
  return 4.0 * lam[i] * lam[j];
}

// Column `col` (0..26) of the strain operator at one point: rows 0-2 membrane (Voigt, engineering shear), 3-5 bending,
// 6-7 transverse shear, 8 drilling (oracle/shell_oracle.py::_strain_operators)
__device__ __forceinline__ void strain_column(const Facet& F, const double* lam, int col, double (&b)[9]) {
#pragma unroll
  for (int r = 0; r < 9; ++r) b[r] = 0.0;
  if (col < 18) {
    const int a = col / 3, k = col % 3;
    double g1, g2;
    p2_grad(F, lam, a, g1, g2);
    b[0] = F.e1[k] * g1;
    b[1] = F.e2[k] * g2;
    b[2] = F.e1[k] * g2 + F.e2[k] * g1;
    b[6] = F.e3[k] * g1;
    b[7] = F.e3[k] * g2;
    b[8] = 0.5 * (F.e1[k] * g2 - F.e2[k] * g1);
  } else {
    const int v = (col - 18) / 3, k = (col - 18) % 3;
    const double g1 = F.gl[v][0], g2 = F.gl[v][1], M = lam[v];
    b[3] = -F.e2[k] * g1;
    b[4] = F.e1[k] * g2;
    b[5] = -F.e2[k] * g2 + F.e1[k] * g1;
    b[6] = F.e2[k] * M;
    b[7] = -F.e1[k] * M;
    b[8] = F.e3[k] * M;
  }
}

__device__ __forceinline__ int64_t shell_gdof(const femo_shell_view& S, int64_t c, int i) {
  if (i < 18) {
    const int a = i / 3, k = i % 3;
    const int64_t node = a < 3 ? (int64_t)S.conn[c * 3 + a] : S.n_vert + S.cedge[c * 3 + a - 3];
    return 3 * node + k;
  }
  const int v = (i - 18) / 3, k = (i - 18) % 3;
  return 3 * S.n_unode + 3 * (int64_t)S.conn[c * 3 + v] + k;
}

struct Material { double c11, c12, c33, mu_s, E; };     // plane stress, shear modulus x 5/6, Young's modulus

__device__ __forceinline__ Material material(double E, double nu) {
  Material m;
  const double f = E / (1.0 - nu * nu);
  m.c11 = f; m.c12 = f * nu; m.c33 = f * 0.5 * (1.0 - nu);
  m.mu_s = (5.0 / 6.0) * E / (2.0 * (1.0 + nu));
  m.E = E;
  return m;
}

// K_e[:, col] for every (cell, col): B^T D B over the two rules, added to the CSR values with atomics
__global__ __launch_bounds__(SH_BLOCK) void k_shell_assemble(femo_shell_view S, double E, double nu, const double* __restrict__ h,
                                                             const int32_t* __restrict__ epos, double* __restrict__ vals) {
  const int64_t t = (int64_t)blockIdx.x * SH_BLOCK + threadIdx.x;
  const int64_t c = t / 27;
  const int col = (int)(t % 27);
  if (c >= S.n_cell) return;
  Facet F;
  facet_frame(S.x, S.conn, c, F);
  const Material m = material(E, nu);
  const double hv[3] = {h[S.conn[c * 3]], h[S.conn[c * 3 + 1]], h[S.conn[c * 3 + 2]]};
  double acc[27];
#pragma unroll
  for (int i = 0; i < 27; ++i) acc[i] = 0.0;
  for (int q = 0; q < 6; ++q) {
    const double* lam = c_lam6[q];
    const double hq = hv[0] * lam[0] + hv[1] * lam[1] + hv[2] * lam[2];
    const double w = c_w6[q] * F.area;
    const double dm = w * hq, db = w * hq * hq * hq * (1.0 / 12.0), dd = w * m.E * hq * hq * hq;
    double bj[9];
    strain_column(F, lam, col, bj);
    // D B[:, col]: membrane, bending, drilling
    const double s0 = dm * (m.c11 * bj[0] + m.c12 * bj[1]), s1 = dm * (m.c12 * bj[0] + m.c11 * bj[1]), s2 = dm * m.c33 * bj[2];
    const double s3 = db * (m.c11 * bj[3] + m.c12 * bj[4]), s4 = db * (m.c12 * bj[3] + m.c11 * bj[4]), s5 = db * m.c33 * bj[5];
    const double s8 = dd * bj[8];
    for (int i = 0; i < 27; ++i) {
      double bi[9];
      strain_column(F, lam, i, bi);
      acc[i] += bi[0] * s0 + bi[1] * s1 + bi[2] * s2 + bi[3] * s3 + bi[4] * s4 + bi[5] * s5 + bi[8] * s8;
    }
  }
  for (int q = 0; q < 3; ++q) {
    const double* lam = c_lam3[q];
    const double hq = hv[0] * lam[0] + hv[1] * lam[1] + hv[2] * lam[2];
    const double ds = (1.0 / 3.0) * F.area * m.mu_s * hq;
    double bj[9];
    strain_column(F, lam, col, bj);
    const double s6 = ds * bj[6], s7 = ds * bj[7];
    for (int i = 0; i < 27; ++i) {
      double bi[9];
      strain_column(F, lam, i, bi);
      acc[i] += bi[6] * s6 + bi[7] * s7;
    }
  }
  const int32_t* ep = epos + c * 729;
  for (int i = 0; i < 27; ++i) atomicAdd(&vals[ep[i * 27 + col]], acc[i]);
}

// strains B w_e (9 rows) of an element vector at one point
__device__ __forceinline__ void element_strain(const Facet& F, const double* lam, const double (&we)[27], double (&s)[9]) {
#pragma unroll
  for (int r = 0; r < 9; ++r) s[r] = 0.0;
  for (int i = 0; i < 27; ++i) {
    double bi[9];
    strain_column(F, lam, i, bi);
#pragma unroll
    for (int r = 0; r < 9; ++r) s[r] += bi[r] * we[i];
  }
}

// out[b] += sum_e v_e^T (dK_e/dh_b) w_e  (one thread per cell): the thickness derivative of the bilinear form.
// v == w gives 2 dEnergy/dh.  energy != nullptr: per-block partials of 1/2 v^T K w as well.
__global__ __launch_bounds__(SH_BLOCK) void k_shell_dform_dh(femo_shell_view S, double E, double nu, const double* __restrict__ h,
                                                             const double* __restrict__ v, const double* __restrict__ w,
                                                             double* __restrict__ out, double* __restrict__ energy) {
  __shared__ double lds[SH_BLOCK / 64];
  const int64_t c = (int64_t)blockIdx.x * SH_BLOCK + threadIdx.x;
  double en = 0.0;
  if (c < S.n_cell) {
    Facet F;
    facet_frame(S.x, S.conn, c, F);
    const Material m = material(E, nu);
    const double hv[3] = {h[S.conn[c * 3]], h[S.conn[c * 3 + 1]], h[S.conn[c * 3 + 2]]};
    double ve[27], we[27];
    for (int i = 0; i < 27; ++i) {
      const int64_t g = shell_gdof(S, c, i);
      ve[i] = v[g]; we[i] = w[g];
    }
    double g[3] = {0.0, 0.0, 0.0};
    for (int q = 0; q < 6; ++q) {
      const double* lam = c_lam6[q];
      const double hq = hv[0] * lam[0] + hv[1] * lam[1] + hv[2] * lam[2];
      const double wq = c_w6[q] * F.area;
      double sv[9], sw[9];
      element_strain(F, lam, ve, sv);
      element_strain(F, lam, we, sw);
      const double mem = sv[0] * (m.c11 * sw[0] + m.c12 * sw[1]) + sv[1] * (m.c12 * sw[0] + m.c11 * sw[1]) + sv[2] * m.c33 * sw[2];
      const double ben = sv[3] * (m.c11 * sw[3] + m.c12 * sw[4]) + sv[4] * (m.c12 * sw[3] + m.c11 * sw[4]) + sv[5] * m.c33 * sw[5];
      const double dri = m.E * sv[8] * sw[8];
      en += wq * (hq * mem + hq * hq * hq * (ben * (1.0 / 12.0) + dri));
      const double d = wq * (mem + hq * hq * (0.25 * ben + 3.0 * dri));        // d/dh of h, h^3/12, h^3
#pragma unroll
      for (int b = 0; b < 3; ++b) g[b] += d * lam[b];
    }
    for (int q = 0; q < 3; ++q) {
      const double* lam = c_lam3[q];
      const double hq = hv[0] * lam[0] + hv[1] * lam[1] + hv[2] * lam[2];
      const double wq = (1.0 / 3.0) * F.area;
      double sv[9], sw[9];
      element_strain(F, lam, ve, sv);
      element_strain(F, lam, we, sw);
      const double sh = m.mu_s * (sv[6] * sw[6] + sv[7] * sw[7]);
      en += wq * hq * sh;
#pragma unroll
      for (int b = 0; b < 3; ++b) g[b] += wq * sh * lam[b];
    }
    if (out != nullptr) {
#pragma unroll
      for (int b = 0; b < 3; ++b) atomicAdd(&out[S.conn[c * 3 + b]], g[b]);
    }
  }
  if (energy != nullptr) {
    const double t = femo_block_sum<SH_BLOCK>(0.5 * en * shell_value_weight(S, c), lds);
    if (threadIdx.x == 0) energy[blockIdx.x] = t;
  }
}

// y += (dK/dh [dh]) w: the FORWARD product with the thickness partial of the elastic residual (state_model.py:176-188, fwd
// mode: d_residuals += dR/dh . d_h).  Element by element from the strains of w: sigma' = (d/dh of the section weights in the
// direction dh) D B w_e at every quadrature point, y_e = sum_q B^T sigma'.  One thread per cell, 27 atomics (the reverse
// product k_shell_dform_dh is its exact transpose: <v, y> = <dh, out> -- tests/test_gpu_shell_round3.py).
__global__ __launch_bounds__(SH_BLOCK) void k_shell_dform_dh_fwd(femo_shell_view S, double E, double nu, const double* __restrict__ h,
                                                                 const double* __restrict__ dh, const double* __restrict__ w,
                                                                 double* __restrict__ y) {
  const int64_t c = (int64_t)blockIdx.x * SH_BLOCK + threadIdx.x;
  if (c >= S.n_cell) return;
  Facet F;
  facet_frame(S.x, S.conn, c, F);
  const Material m = material(E, nu);
  const double hv[3] = {h[S.conn[c * 3]], h[S.conn[c * 3 + 1]], h[S.conn[c * 3 + 2]]};
  const double dv[3] = {dh[S.conn[c * 3]], dh[S.conn[c * 3 + 1]], dh[S.conn[c * 3 + 2]]};
  double we[27], acc[27];
  for (int i = 0; i < 27; ++i) { we[i] = w[shell_gdof(S, c, i)]; acc[i] = 0.0; }
  for (int q = 0; q < 6; ++q) {
    const double* lam = c_lam6[q];
    const double hq = hv[0] * lam[0] + hv[1] * lam[1] + hv[2] * lam[2];
    const double dq = dv[0] * lam[0] + dv[1] * lam[1] + dv[2] * lam[2];
    const double wq = c_w6[q] * F.area;
    const double dm = wq * dq, db = wq * 0.25 * hq * hq * dq, dd = wq * 3.0 * m.E * hq * hq * dq;   // d/dh of h, h^3/12, E h^3
    double sw[9];
    element_strain(F, lam, we, sw);
    const double s0 = dm * (m.c11 * sw[0] + m.c12 * sw[1]), s1 = dm * (m.c12 * sw[0] + m.c11 * sw[1]), s2 = dm * m.c33 * sw[2];
    const double s3 = db * (m.c11 * sw[3] + m.c12 * sw[4]), s4 = db * (m.c12 * sw[3] + m.c11 * sw[4]), s5 = db * m.c33 * sw[5];
    const double s8 = dd * sw[8];
    for (int i = 0; i < 27; ++i) {
      double bi[9];
      strain_column(F, lam, i, bi);
      acc[i] += bi[0] * s0 + bi[1] * s1 + bi[2] * s2 + bi[3] * s3 + bi[4] * s4 + bi[5] * s5 + bi[8] * s8;
    }
  }
  for (int q = 0; q < 3; ++q) {
    const double* lam = c_lam3[q];
    const double dq = dv[0] * lam[0] + dv[1] * lam[1] + dv[2] * lam[2];
    const double ds = (1.0 / 3.0) * F.area * m.mu_s * dq;
    double sw[9];
    element_strain(F, lam, we, sw);
    const double s6 = ds * sw[6], s7 = ds * sw[7];
    for (int i = 0; i < 27; ++i) {
      double bi[9];
      strain_column(F, lam, i, bi);
      acc[i] += bi[6] * s6 + bi[7] * s7;
    }
  }
  for (int i = 0; i < 27; ++i) atomicAdd(&y[shell_gdof(S, c, i)], acc[i]);
}

// F += int f . v  (f: CG1 vector field at the vertices, force per unit area), sign * that
__global__ __launch_bounds__(SH_BLOCK) void k_shell_load(femo_shell_view S, const double* __restrict__ f, double sign, double* __restrict__ Fv) {
  const int64_t c = (int64_t)blockIdx.x * SH_BLOCK + threadIdx.x;
  if (c >= S.n_cell) return;
  Facet F;
  facet_frame(S.x, S.conn, c, F);
  double fv[3][3];
#pragma unroll
  for (int b = 0; b < 3; ++b)
#pragma unroll
    for (int k = 0; k < 3; ++k) fv[b][k] = f[(int64_t)S.conn[c * 3 + b] * 3 + k];
  double acc[18];
#pragma unroll
  for (int i = 0; i < 18; ++i) acc[i] = 0.0;
  for (int q = 0; q < 6; ++q) {
    const double* lam = c_lam6[q];
    const double wq = c_w6[q] * F.area;
    double fq[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) fq[k] = lam[0] * fv[0][k] + lam[1] * fv[1][k] + lam[2] * fv[2][k];
    for (int a = 0; a < 6; ++a) {
      const double N = p2_value(lam, a) * wq;
#pragma unroll
      for (int k = 0; k < 3; ++k) acc[a * 3 + k] += N * fq[k];
    }
  }
  for (int i = 0; i < 18; ++i) atomicAdd(&Fv[shell_gdof(S, c, i)], sign * acc[i]);
}

// out[vertex b, k] += sign * int phi_b (lambda_u)_k : transpose of the load map applied to a state-sized vector
__global__ __launch_bounds__(SH_BLOCK) void k_shell_load_T(femo_shell_view S, const double* __restrict__ lam_state, double sign, double* __restrict__ out) {
  const int64_t c = (int64_t)blockIdx.x * SH_BLOCK + threadIdx.x;
  if (c >= S.n_cell) return;
  Facet F;
  facet_frame(S.x, S.conn, c, F);
  double le[18];
  for (int i = 0; i < 18; ++i) le[i] = lam_state[shell_gdof(S, c, i)];
  double acc[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
  for (int q = 0; q < 6; ++q) {
    const double* lam = c_lam6[q];
    const double wq = c_w6[q] * F.area;
    double uq[3] = {0, 0, 0};
    for (int a = 0; a < 6; ++a) {
      const double N = p2_value(lam, a);
#pragma unroll
      for (int k = 0; k < 3; ++k) uq[k] += N * le[a * 3 + k];
    }
#pragma unroll
    for (int b = 0; b < 3; ++b)
#pragma unroll
      for (int k = 0; k < 3; ++k) acc[b][k] += wq * lam[b] * uq[k];
  }
#pragma unroll
  for (int b = 0; b < 3; ++b)
#pragma unroll
    for (int k = 0; k < 3; ++k) atomicAdd(&out[(int64_t)S.conn[c * 3 + b] * 3 + k], sign * acc[b][k]);
}

// compliance 1/2 int u.u (partials per block) and, if grad != nullptr, its gradient M_u w added to grad
// cellw (optional, one weight per cell): the `dxx` measure of shell_pde.py:66,284 -- dx_2(10), a tagged subset of cells --
// as a DG0 indicator; cells of weight 0 are skipped
__global__ __launch_bounds__(SH_BLOCK) void k_shell_compliance(femo_shell_view S, const double* __restrict__ w, const double* __restrict__ cellw,
                                                               double* __restrict__ partials, double* __restrict__ grad) {
  __shared__ double lds[SH_BLOCK / 64];
  const int64_t c = (int64_t)blockIdx.x * SH_BLOCK + threadIdx.x;
  double J = 0.0;
  const double chi = (c < S.n_cell && cellw != nullptr) ? cellw[c] : 1.0;
  if (c < S.n_cell && chi != 0.0) {
    Facet F;
    facet_frame(S.x, S.conn, c, F);
    F.area *= chi;
    double ue[18], ge[18];
    for (int i = 0; i < 18; ++i) { ue[i] = w[shell_gdof(S, c, i)]; ge[i] = 0.0; }
    for (int q = 0; q < 6; ++q) {
      const double* lam = c_lam6[q];
      const double wq = c_w6[q] * F.area;
      double uq[3] = {0, 0, 0};
      for (int a = 0; a < 6; ++a) {
        const double N = p2_value(lam, a);
#pragma unroll
        for (int k = 0; k < 3; ++k) uq[k] += N * ue[a * 3 + k];
      }
      J += 0.5 * wq * (uq[0] * uq[0] + uq[1] * uq[1] + uq[2] * uq[2]);
      for (int a = 0; a < 6; ++a) {
        const double N = p2_value(lam, a) * wq;
#pragma unroll
        for (int k = 0; k < 3; ++k) ge[a * 3 + k] += N * uq[k];
      }
    }
    if (grad != nullptr)
      for (int i = 0; i < 18; ++i) atomicAdd(&grad[shell_gdof(S, c, i)], ge[i]);
  }
  if (partials != nullptr) {
    const double t = femo_block_sum<SH_BLOCK>(J, lds);
    if (threadIdx.x == 0) partials[blockIdx.x] = t;
  }
}

// int rho h (partials) and its gradient rho |T| / 3 per vertex
// J = 1 / alpha int (m sigma_vm)^rho dx, sigma_vm the von Mises stress of the in-plane stress C (eps + z kappa) at
// z = surface * h / 2 (oracle/shell_oracle.py::pnorm_stress; shell_pde.py:297-313), degree-4 rule; partials: per-block
// sums of the value, grad_w += dJ/dw (n_dof), grad_h += dJ/dh (n_vert).  One thread per cell.
__global__ __launch_bounds__(SH_BLOCK) void k_shell_pnorm_stress(femo_shell_view S, double E, double nu, const double* __restrict__ h,
                                                                 const double* __restrict__ w, double mscale, double rho, double inv_alpha,
                                                                 double surface, double* __restrict__ partials, double* __restrict__ grad_w,
                                                                 double* __restrict__ grad_h) {
  __shared__ double lds[SH_BLOCK / 64];
  const int64_t c = (int64_t)blockIdx.x * SH_BLOCK + threadIdx.x;
  double val = 0.0;
  if (c < S.n_cell) {
    Facet F;
    facet_frame(S.x, S.conn, c, F);
    const Material mt = material(E, nu);
    const double hv[3] = {h[S.conn[c * 3]], h[S.conn[c * 3 + 1]], h[S.conn[c * 3 + 2]]};
    double we[27], gw[27];
    for (int i = 0; i < 27; ++i) { we[i] = w[shell_gdof(S, c, i)]; gw[i] = 0.0; }
    double gh[3] = {0.0, 0.0, 0.0};
    for (int q = 0; q < 6; ++q) {
      const double* lam = c_lam6[q];
      const double z = 0.5 * surface * (hv[0] * lam[0] + hv[1] * lam[1] + hv[2] * lam[2]);
      const double wq = c_w6[q] * F.area;
      double sw[9];
      element_strain(F, lam, we, sw);
      const double e0 = sw[0] + z * sw[3], e1 = sw[1] + z * sw[4], e2 = sw[2] + z * sw[5];
      const double s0 = mt.c11 * e0 + mt.c12 * e1, s1 = mt.c12 * e0 + mt.c11 * e1, s2 = mt.c33 * e2;
      const double vm = sqrt(s0 * s0 - s0 * s1 + s1 * s1 + 3.0 * s2 * s2);
      if (!(vm > 0.0)) continue;
      const double pw = pow(mscale * vm, rho - 1.0);
      val += wq * pw * mscale * vm * inv_alpha;
      if (grad_w == nullptr && grad_h == nullptr) continue;
      const double fac = wq * rho * mscale * pw * inv_alpha / (2.0 * vm);             // dJ/dvm / (2 vm)
      const double d0 = fac * (2.0 * s0 - s1), d1 = fac * (2.0 * s1 - s0), d2 = fac * 6.0 * s2;   // dJ / d sigma
      const double t0 = mt.c11 * d0 + mt.c12 * d1, t1 = mt.c12 * d0 + mt.c11 * d1, t2 = mt.c33 * d2;   // dJ / d (eps + z kappa)
      const double dk = 0.5 * surface * (t0 * sw[3] + t1 * sw[4] + t2 * sw[5]);
#pragma unroll
      for (int b = 0; b < 3; ++b) gh[b] += dk * lam[b];
      if (grad_w != nullptr) {
        for (int col = 0; col < 27; ++col) {
          double bc[9];
          strain_column(F, lam, col, bc);
          gw[col] += t0 * (bc[0] + z * bc[3]) + t1 * (bc[1] + z * bc[4]) + t2 * (bc[2] + z * bc[5]);
        }
      }
    }
    if (grad_w != nullptr)
      for (int i = 0; i < 27; ++i)
        if (gw[i] != 0.0) atomicAdd(&grad_w[shell_gdof(S, c, i)], gw[i]);
    if (grad_h != nullptr) {
#pragma unroll
      for (int b = 0; b < 3; ++b) atomicAdd(&grad_h[S.conn[c * 3 + b]], gh[b]);
    }
  }
  if (partials != nullptr) {
    const double t = femo_block_sum<SH_BLOCK>(val * shell_value_weight(S, c), lds);
    if (threadIdx.x == 0) partials[blockIdx.x] = t;
  }
}

// right-hand side of the L2 projection of the von Mises stress onto CG1 (shell_pde.py:330-332): b_i += int sigma_vm phi_i,
// and the row sums of the P1 mass matrix, lumped_i += |T| / 3.  One thread per cell.
__global__ __launch_bounds__(SH_BLOCK) void k_shell_vm_rhs(femo_shell_view S, double E, double nu, const double* __restrict__ h,
                                                           const double* __restrict__ w, double surface, double* __restrict__ rhs,
                                                           double* __restrict__ lumped) {
  const int64_t c = (int64_t)blockIdx.x * SH_BLOCK + threadIdx.x;
  if (c >= S.n_cell) return;
  Facet F;
  facet_frame(S.x, S.conn, c, F);
  const Material mt = material(E, nu);
  const double hv[3] = {h[S.conn[c * 3]], h[S.conn[c * 3 + 1]], h[S.conn[c * 3 + 2]]};
  double we[27];
  for (int i = 0; i < 27; ++i) we[i] = w[shell_gdof(S, c, i)];
  double b[3] = {0.0, 0.0, 0.0};
  for (int q = 0; q < 6; ++q) {
    const double* lam = c_lam6[q];
    const double z = 0.5 * surface * (hv[0] * lam[0] + hv[1] * lam[1] + hv[2] * lam[2]);
    double sw[9];
    element_strain(F, lam, we, sw);
    const double e0 = sw[0] + z * sw[3], e1 = sw[1] + z * sw[4], e2 = sw[2] + z * sw[5];
    const double s0 = mt.c11 * e0 + mt.c12 * e1, s1 = mt.c12 * e0 + mt.c11 * e1, s2 = mt.c33 * e2;
    const double vm = sqrt(s0 * s0 - s0 * s1 + s1 * s1 + 3.0 * s2 * s2);
#pragma unroll
    for (int a = 0; a < 3; ++a) b[a] += c_w6[q] * F.area * vm * lam[a];
  }
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    atomicAdd(&rhs[S.conn[c * 3 + a]], b[a]);
    if (lumped != nullptr) atomicAdd(&lumped[S.conn[c * 3 + a]], F.area * (1.0 / 3.0));
  }
}

// y += M x with the P1 mass matrix of the surface, element by element: M_e = |T| / 12 (1 + delta)
__global__ __launch_bounds__(SH_BLOCK) void k_shell_p1_mass(femo_shell_view S, const double* __restrict__ x, double* __restrict__ y) {
  const int64_t c = (int64_t)blockIdx.x * SH_BLOCK + threadIdx.x;
  if (c >= S.n_cell) return;
  Facet F;
  facet_frame(S.x, S.conn, c, F);
  const int32_t v0 = S.conn[c * 3], v1 = S.conn[c * 3 + 1], v2 = S.conn[c * 3 + 2];
  const double x0 = x[v0], x1 = x[v1], x2 = x[v2], sum = x0 + x1 + x2, k = F.area * (1.0 / 12.0);
  atomicAdd(&y[v0], k * (sum + x0));
  atomicAdd(&y[v1], k * (sum + x1));
  atomicAdd(&y[v2], k * (sum + x2));
}

__global__ __launch_bounds__(SH_BLOCK) void k_shell_mass(femo_shell_view S, double rho, const double* __restrict__ h, double* __restrict__ partials,
                                                         double* __restrict__ grad) {
  __shared__ double lds[SH_BLOCK / 64];
  const int64_t c = (int64_t)blockIdx.x * SH_BLOCK + threadIdx.x;
  double M = 0.0;
  if (c < S.n_cell) {
    Facet F;
    facet_frame(S.x, S.conn, c, F);
    const double a3 = rho * F.area * (1.0 / 3.0);
#pragma unroll
    for (int b = 0; b < 3; ++b) {
      M += a3 * h[S.conn[c * 3 + b]];
      if (grad != nullptr) atomicAdd(&grad[S.conn[c * 3 + b]], a3);
    }
  }
  if (partials != nullptr) {
    const double t = femo_block_sum<SH_BLOCK>(M * shell_value_weight(S, c), lds);
    if (threadIdx.x == 0) partials[blockIdx.x] = t;
  }
}

// ------------------------------------------------- penalty boundary terms, inertia, regularisation (round 3) ----
// Edge mass matrices on [0, 1] x length: P2 (end vertices, midpoint) and P1
__constant__ double c_m2[3][3] = {{4.0 / 30, -1.0 / 30, 2.0 / 30}, {-1.0 / 30, 4.0 / 30, 2.0 / 30}, {2.0 / 30, 2.0 / 30, 16.0 / 30}};
__constant__ double c_m1[2][2] = {{2.0 / 6, 1.0 / 6}, {1.0 / 6, 2.0 / 6}};

// vals += K_pen: per tagged edge and component 9 + 4 entries at the CSR positions the host looked up (pos: 39 per edge,
// component-major: 9 displacement pairs row-major over (v0, v1, mid), then 4 rotation pairs over (v0, v1)); coef =
// beta (sum over adjacent cells of 1 / h_E) |edge|  (oracle/shell_oracle.py::penalty_matrix)
__global__ void k_shell_penalty_add(int64_t n_e, const int32_t* __restrict__ pos, const double* __restrict__ coef, double* __restrict__ vals) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n_e * 39) return;
  const int64_t e = t / 39;
  const int r = (int)(t % 39) % 13;
  const double m = r < 9 ? c_m2[r / 3][r % 3] : c_m1[(r - 9) / 2][(r - 9) % 2];
  atomicAdd(&vals[pos[t]], coef[e] * m);
}

// y += K_pen (x - g)   (g == nullptr: homogeneous data)
__global__ void k_shell_penalty_apply(int64_t n_e, const int32_t* __restrict__ nodes, const double* __restrict__ coef, int64_t n_unode,
                                      const double* __restrict__ x, const double* __restrict__ g, double* __restrict__ y) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n_e) return;
  const int64_t un[3] = {nodes[3 * e], nodes[3 * e + 1], nodes[3 * e + 2]};
  const double cf = coef[e];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    double d[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) { const int64_t dof = 3 * un[i] + k; d[i] = x[dof] - (g ? g[dof] : 0.0); }
#pragma unroll
    for (int i = 0; i < 3; ++i) atomicAdd(&y[3 * un[i] + k], cf * (c_m2[i][0] * d[0] + c_m2[i][1] * d[1] + c_m2[i][2] * d[2]));
    double t2[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) { const int64_t dof = 3 * n_unode + 3 * un[i] + k; t2[i] = x[dof] - (g ? g[dof] : 0.0); }
#pragma unroll
    for (int i = 0; i < 2; ++i) atomicAdd(&y[3 * n_unode + 3 * un[i] + k], cf * (c_m1[i][0] * t2[0] + c_m1[i][1] * t2[1]));
  }
}

// Inertial residual (shell_pde.py:255-256 kinetic_residual -> inertialResidual [ext]):
//   y += M(h) a,  M = int rho h  N_a N_b (displacements, P2) + int rho h^3 / 12  phi_a phi_b (rotations, P1), degree-4 rule;
//   out_h[b] += lam^T (dM/dh_b) a   when lam != nullptr (y is not written then).  One thread per cell.
//   dh != nullptr (with lam == nullptr): y += (dM/dh [dh]) a, the forward product -- the section weights h and h^3/12 replaced
//   by their derivatives in the direction dh.
__global__ __launch_bounds__(SH_BLOCK) void k_shell_inertia(femo_shell_view S, double rho, const double* __restrict__ h, const double* __restrict__ a,
                                                            const double* __restrict__ lam_state, double* __restrict__ y, double* __restrict__ out_h,
                                                            const double* __restrict__ dh = nullptr) {
  const int64_t c = (int64_t)blockIdx.x * SH_BLOCK + threadIdx.x;
  if (c >= S.n_cell) return;
  Facet F;
  facet_frame(S.x, S.conn, c, F);
  const double hv[3] = {h[S.conn[c * 3]], h[S.conn[c * 3 + 1]], h[S.conn[c * 3 + 2]]};
  double ae[27], le[27], acc[27];
  for (int i = 0; i < 27; ++i) {
    const int64_t gd = shell_gdof(S, c, i);
    ae[i] = a[gd];
    le[i] = lam_state ? lam_state[gd] : 0.0;
    acc[i] = 0.0;
  }
  double gh[3] = {0.0, 0.0, 0.0};
  for (int q = 0; q < 6; ++q) {
    const double* lam = c_lam6[q];
    const double hq = hv[0] * lam[0] + hv[1] * lam[1] + hv[2] * lam[2];
    const double wq = c_w6[q] * F.area * rho;
    double uq[3] = {0, 0, 0}, tq[3] = {0, 0, 0}, lu[3] = {0, 0, 0}, lt[3] = {0, 0, 0};
    for (int n = 0; n < 6; ++n) {
      const double N = p2_value(lam, n);
#pragma unroll
      for (int k = 0; k < 3; ++k) { uq[k] += N * ae[3 * n + k]; lu[k] += N * le[3 * n + k]; }
    }
#pragma unroll
    for (int b = 0; b < 3; ++b)
#pragma unroll
      for (int k = 0; k < 3; ++k) { tq[k] += lam[b] * ae[18 + 3 * b + k]; lt[k] += lam[b] * le[18 + 3 * b + k]; }
    if (lam_state == nullptr) {
      double cu = wq * hq, ct = wq * hq * hq * hq * (1.0 / 12.0);
      if (dh != nullptr) {
        const double dq = dh[S.conn[c * 3]] * lam[0] + dh[S.conn[c * 3 + 1]] * lam[1] + dh[S.conn[c * 3 + 2]] * lam[2];
        cu = wq * dq; ct = wq * 0.25 * hq * hq * dq;
      }
      for (int n = 0; n < 6; ++n) {
        const double N = p2_value(lam, n) * cu;
#pragma unroll
        for (int k = 0; k < 3; ++k) acc[3 * n + k] += N * uq[k];
      }
#pragma unroll
      for (int b = 0; b < 3; ++b)
#pragma unroll
        for (int k = 0; k < 3; ++k) acc[18 + 3 * b + k] += ct * lam[b] * tq[k];
    } else {
      const double d = wq * ((lu[0] * uq[0] + lu[1] * uq[1] + lu[2] * uq[2]) + 0.25 * hq * hq * (lt[0] * tq[0] + lt[1] * tq[1] + lt[2] * tq[2]));
#pragma unroll
      for (int b = 0; b < 3; ++b) gh[b] += d * lam[b];
    }
  }
  if (lam_state == nullptr) {
    for (int i = 0; i < 27; ++i) atomicAdd(&y[shell_gdof(S, c, i)], acc[i]);
  } else {
#pragma unroll
    for (int b = 0; b < 3; ++b) atomicAdd(&out_h[S.conn[c * 3 + b]], gh[b]);
  }
}

// `ShellPDE.regularization(h, type)` (shell_pde.py:262-282), alpha1 = 1e3, alpha2 = 1, CG1 thickness on flat facets:
//   kind 1 'H1':  1/2 alpha1 int |grad h|^2     kind 2 'L2H1': 1/2 alpha1 int h^2 + 1/2 alpha2 int h_mesh^2 |grad h|^2
//   kind 3 'L2':  1/2 alpha1 int h^2            h_mesh = CellDiameter = the largest vertex distance of the cell [ext]
// partials: per-block sums of the value; grad += d/dh.  One thread per cell.
__global__ __launch_bounds__(SH_BLOCK) void k_shell_regularization(femo_shell_view S, int kind, const double* __restrict__ h,
                                                                   double* __restrict__ partials, double* __restrict__ grad) {
  __shared__ double lds[SH_BLOCK / 64];
  const int64_t c = (int64_t)blockIdx.x * SH_BLOCK + threadIdx.x;
  double val = 0.0;
  if (c < S.n_cell) {
    Facet F;
    facet_frame(S.x, S.conn, c, F);
    const double a1 = 1e3, a2 = 1.0;
    const double hv[3] = {h[S.conn[c * 3]], h[S.conn[c * 3 + 1]], h[S.conn[c * 3 + 2]]};
    double g[3] = {0.0, 0.0, 0.0};
    if (kind == 2 || kind == 3) {
      const double k12 = F.area * (1.0 / 12.0), sum = hv[0] + hv[1] + hv[2];
#pragma unroll
      for (int b = 0; b < 3; ++b) {
        const double Mh = k12 * (sum + hv[b]);
        val += 0.5 * a1 * hv[b] * Mh;
        g[b] += a1 * Mh;
      }
    }
    if (kind == 1 || kind == 2) {
      double coef = a1 * F.area;
      if (kind == 2) {
        double d2 = 0.0;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
          const int j = (i + 1) % 3;
          double l2 = 0.0;
#pragma unroll
          for (int k = 0; k < 3; ++k) {
            const double dx = S.x[(int64_t)S.conn[c * 3 + i] * 3 + k] - S.x[(int64_t)S.conn[c * 3 + j] * 3 + k];
            l2 += dx * dx;
          }
          d2 = fmax(d2, l2);
        }
        coef = a2 * d2 * F.area;
      }
      const double g1 = F.gl[0][0] * hv[0] + F.gl[1][0] * hv[1] + F.gl[2][0] * hv[2];
      const double g2 = F.gl[0][1] * hv[0] + F.gl[1][1] * hv[1] + F.gl[2][1] * hv[2];
      val += 0.5 * coef * (g1 * g1 + g2 * g2);
#pragma unroll
      for (int b = 0; b < 3; ++b) g[b] += coef * (F.gl[b][0] * g1 + F.gl[b][1] * g2);
    }
    if (grad != nullptr) {
#pragma unroll
      for (int b = 0; b < 3; ++b) atomicAdd(&grad[S.conn[c * 3 + b]], g[b]);
    }
  }
  if (partials != nullptr) {
    const double t = femo_block_sum<SH_BLOCK>(val * shell_value_weight(S, c), lds);
    if (threadIdx.x == 0) partials[blockIdx.x] = t;
  }
}

// int coef h^p dx with the degree-4 rule (the thickness term of pnorm_stress(regularization=True), shell_pde.py:307-309:
// 0.5 * 1e3 * h**rho * dx) and its gradient.  One thread per cell.
__global__ __launch_bounds__(SH_BLOCK) void k_shell_hpower(femo_shell_view S, double coef, double p, const double* __restrict__ h,
                                                           double* __restrict__ partials, double* __restrict__ grad) {
  __shared__ double lds[SH_BLOCK / 64];
  const int64_t c = (int64_t)blockIdx.x * SH_BLOCK + threadIdx.x;
  double val = 0.0;
  if (c < S.n_cell) {
    Facet F;
    facet_frame(S.x, S.conn, c, F);
    const double hv[3] = {h[S.conn[c * 3]], h[S.conn[c * 3 + 1]], h[S.conn[c * 3 + 2]]};
    double g[3] = {0.0, 0.0, 0.0};
    for (int q = 0; q < 6; ++q) {
      const double* lam = c_lam6[q];
      const double hq = hv[0] * lam[0] + hv[1] * lam[1] + hv[2] * lam[2];
      const double wq = c_w6[q] * F.area * coef;
      const double pm1 = pow(hq, p - 1.0);
      val += wq * pm1 * hq;
#pragma unroll
      for (int b = 0; b < 3; ++b) g[b] += wq * p * pm1 * lam[b];
    }
    if (grad != nullptr) {
#pragma unroll
      for (int b = 0; b < 3; ++b) atomicAdd(&grad[S.conn[c * 3 + b]], g[b]);
    }
  }
  if (partials != nullptr) {
    const double t = femo_block_sum<SH_BLOCK>(val * shell_value_weight(S, c), lds);
    if (threadIdx.x == 0) partials[blockIdx.x] = t;
  }
}

// ---------------------------------------------------------------- CSR operator ----
// y = A x for rows [0, n), 16 lanes per row (the element-coupling pattern has ~50 entries per row: a whole wave per
// row left three quarters of the lanes idle and a quarter of the rows in flight).  `fixed` != nullptr: the masked
// operator M A M + (I - M) (strongly imposed dofs are identity rows and columns); mask_cols = 0 skips the column test
// for callers whose x is zero on the imposed dofs anyway (the CG directions) -- a byte gather per matrix entry.
// partials != nullptr: per-block partial of x.y.
__global__ __launch_bounds__(SH_BLOCK) void k_csr_spmv(int64_t n, const int64_t* __restrict__ rowptr, const int32_t* __restrict__ cols,
                                                       const double* __restrict__ vals, const uint8_t* __restrict__ fixed, int mask_cols,
                                                       const double* __restrict__ x, double* __restrict__ y, double* __restrict__ partials,
                                                       const int32_t* __restrict__ done, double* commit_dst = nullptr,
                                                       const double* commit_src = nullptr) {
  if (done != nullptr && *done) return;
  // the CG loop publishes gamma of the iteration here (every consumer of it runs after this launch, every block of
  // the kernel that produced it has finished): saves a launch of its own
  if (commit_dst != nullptr && blockIdx.x == 0 && threadIdx.x == 0) *commit_dst = *commit_src;
  __shared__ double lds[SH_BLOCK / 64];
  constexpr int SUB = 16;
  const int sl = threadIdx.x & (SUB - 1);
  const int64_t nsub = (int64_t)gridDim.x * (SH_BLOCK / SUB);
  double dot = 0.0;
  for (int64_t row = (int64_t)blockIdx.x * (SH_BLOCK / SUB) + (threadIdx.x / SUB); row < n; row += nsub) {
    double s = 0.0;
    const bool rf = fixed != nullptr && fixed[row];
    if (!rf) {
      const int64_t e1 = rowptr[row + 1];
      for (int64_t e = rowptr[row] + sl; e < e1; e += SUB) {
        const int32_t cidx = cols[e];
        const double v = vals[e] * x[cidx];
        s += (mask_cols && fixed != nullptr && fixed[cidx]) ? 0.0 : v;
      }
    }
#pragma unroll
    for (int off = SUB / 2; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
    if (sl == 0) {
      const double xr = x[row];
      const double yi = rf ? xr : s;
      y[row] = yi;
      dot += xr * yi;
    }
  }
  if (partials != nullptr) {
    const double t = femo_block_sum<SH_BLOCK>(dot, lds);
    if (threadIdx.x == 0) partials[blockIdx.x] = t;
  }
}

// The same product over the node-block view of the pattern: the three dofs of a node have the same columns and the
// columns come in runs of three, so one column index serves nine entries (8.4 instead of 12 bytes per entry) and the
// three x values of a block are loaded once for its three rows.  The value array is the scalar CSR one, untouched:
// row 3 b + i of block row b is the run vals[9 k0 + 3 i nb ..), block k at offset 3 (k - k0).  16 lanes per block row
// (13 blocks for an edge node, ~26 for a vertex node or a rotation).  Imposed dofs: identity rows; x must be zero on
// the imposed columns (the CG directions are).
// three consecutive doubles, 8-byte aligned: loaded as one 16-byte and one 8-byte access (global loads need no more
// than dword alignment on gfx9) -- 9 instead of 13 memory instructions per block
struct __attribute__((packed, aligned(8))) Triple { double a, b, c; };

template <int SUB>
__global__ __launch_bounds__(SH_BLOCK) void k_bcsr3_spmv(int64_t nb, const int64_t* __restrict__ brow, const int32_t* __restrict__ bcols,
                                                         const double* __restrict__ vals, const uint8_t* __restrict__ fixed,
                                                         const double* __restrict__ x, double* __restrict__ y, double* __restrict__ partials,
                                                         const int32_t* __restrict__ done, double* commit_dst = nullptr,
                                                         const double* commit_src = nullptr) {
  if (done != nullptr && *done) return;
  if (commit_dst != nullptr && blockIdx.x == 0 && threadIdx.x == 0) *commit_dst = *commit_src;
  __shared__ double lds[SH_BLOCK / 64];
  const int sl = threadIdx.x & (SUB - 1);
  const int64_t nsub = (int64_t)gridDim.x * (SH_BLOCK / SUB);
  double dot = 0.0;
  // Software pipeline over the group's block rows: the offsets of row b + 2 nsub and the first column index of row
  // b + nsub are requested before row b is computed, so a row costs one memory latency (values and x together) instead
  // of three in a chain (offsets -> column -> x).
  int64_t b = (int64_t)blockIdx.x * (SH_BLOCK / SUB) + (threadIdx.x / SUB);
  int64_t k0 = 0, k1 = 0, n0 = 0, n1 = 0;
  int32_t c = 0;
  if (b < nb) {
    k0 = brow[b]; k1 = brow[b + 1];
    if (k0 + sl < k1) c = bcols[k0 + sl];
  }
  if (b + nsub < nb) { n0 = brow[b + nsub]; n1 = brow[b + nsub + 1]; }
  for (; b < nb; b += nsub) {
    int64_t m0 = 0, m1 = 0;
    int32_t cn = 0;
    if (b + 2 * nsub < nb) { m0 = brow[b + 2 * nsub]; m1 = brow[b + 2 * nsub + 1]; }
    if (n0 + sl < n1) cn = bcols[n0 + sl];
    const int64_t len = 3 * (k1 - k0);
    const double* v0 = vals + 9 * k0;
    const double* v1 = v0 + len;
    const double* v2 = v1 + len;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0;
    for (int64_t k = k0 + sl; k < k1; k += SUB) {
      if (k >= k0 + SUB) c = bcols[k];
      const int64_t o = 3 * (k - k0);
      // plain loads: a lane reads 8 bytes at a stride of 24, so a cache line serves three instructions -- with
      // nontemporal loads the product took 135 us instead of 103 (988 k dofs).  Also slower: streaming the rows in
      // storage order (lane = entry: 109-121 us), 8 or 4 lanes per block row
      const Triple r0 = *reinterpret_cast<const Triple*>(v0 + o), r1 = *reinterpret_cast<const Triple*>(v1 + o),
                   r2 = *reinterpret_cast<const Triple*>(v2 + o), xc = *reinterpret_cast<const Triple*>(x + c);
      const double a00 = r0.a, a01 = r0.b, a02 = r0.c, a10 = r1.a, a11 = r1.b, a12 = r1.c, a20 = r2.a, a21 = r2.b, a22 = r2.c;
      const double x0 = xc.a, x1 = xc.b, x2 = xc.c;
      s0 += a00 * x0 + a01 * x1 + a02 * x2;
      s1 += a10 * x0 + a11 * x1 + a12 * x2;
      s2 += a20 * x0 + a21 * x1 + a22 * x2;
    }
#pragma unroll
    for (int off = SUB / 2; off > 0; off >>= 1) {
      s0 += __shfl_xor(s0, off, 64);
      s1 += __shfl_xor(s1, off, 64);
      s2 += __shfl_xor(s2, off, 64);
    }
    if (sl < 3) {
      const int64_t row = 3 * b + sl;
      const double s = sl == 0 ? s0 : (sl == 1 ? s1 : s2);
      const bool rf = fixed != nullptr && fixed[row];
      const double xr = x[row];
      const double yi = rf ? xr : s;
      y[row] = yi;
      dot += xr * yi;
    }
    k0 = n0; k1 = n1; c = cn;
    n0 = m0; n1 = m1;
  }
  if (partials != nullptr) {
    const double t = femo_block_sum<SH_BLOCK>(dot, lds);
    if (threadIdx.x == 0) partials[blockIdx.x] = t;
  }
}

// ---- block-SELL ----
constexpr int BSW = 32;                                  // block rows per slice (16 / 32 / 64: 0.355 / 0.347 / 0.355 ms per iteration at 1.97 M dofs)
// copy of the CSR values into the slice layout: group g = bs_off[slice] + slot holds block `slot` of the slice's BSW
// block rows: cols[BSW g + lane], vals[(9 g + comp) BSW + lane]; rows with fewer blocks are padded (column = own, 0)
__global__ __launch_bounds__(256) void k_bsell_fill(int64_t nb, const int64_t* __restrict__ brow, const int32_t* __restrict__ bcols,
                                                    const double* __restrict__ vals, const int64_t* __restrict__ bs_off,
                                                    int32_t* __restrict__ cols, double* __restrict__ out) {
  const int64_t slice = blockIdx.x;
  const int lane = threadIdx.x & (BSW - 1), sub = threadIdx.x / BSW;   // 256 / BSW slots in flight per pass
  const int64_t b = slice * BSW + lane;
  const int64_t g0 = bs_off[slice], nslot = bs_off[slice + 1] - g0;
  int64_t k0 = 0, k1 = 0;
  if (b < nb) { k0 = brow[b]; k1 = brow[b + 1]; }
  const int64_t len = 3 * (k1 - k0);
  for (int64_t sl = sub; sl < nslot; sl += 256 / BSW) {
    const int64_t g = g0 + sl;
    const bool have = k0 + sl < k1;
    cols[BSW * g + lane] = have ? bcols[k0 + sl] : (int32_t)(b < nb ? 3 * b : 0);
    const double* v = vals + 9 * k0 + 3 * sl;
#pragma unroll
    for (int fa = 0; fa < 3; ++fa)
#pragma unroll
      for (int fb = 0; fb < 3; ++fb) out[(9 * g + 3 * fa + fb) * BSW + lane] = have ? v[fa * len + fb] : 0.0;
  }
}

// y = A x from the block-SELL copy: lane = block row, no cross-lane reduction; every value load of a 16-lane group is
// one contiguous 128-byte piece.  Imposed dofs: identity rows; x must be zero on the imposed columns.
__global__ __launch_bounds__(SH_BLOCK) void k_bsell_spmv(int64_t nb, int64_t n_slice, const int64_t* __restrict__ bs_off,
                                                         const int32_t* __restrict__ cols, const double* __restrict__ vals,
                                                         const uint8_t* __restrict__ fixed, const double* __restrict__ x,
                                                         double* __restrict__ y, double* __restrict__ partials, const int32_t* __restrict__ done,
                                                         double* commit_dst = nullptr, const double* commit_src = nullptr) {
  if (done != nullptr && *done) return;
  if (commit_dst != nullptr && blockIdx.x == 0 && threadIdx.x == 0) *commit_dst = *commit_src;
  __shared__ double lds[SH_BLOCK / 64];
  const int lane = threadIdx.x & (BSW - 1);
  const int64_t nsub = (int64_t)gridDim.x * (SH_BLOCK / BSW);
  double dot = 0.0;
  for (int64_t slice = (int64_t)blockIdx.x * (SH_BLOCK / BSW) + (threadIdx.x / BSW); slice < n_slice; slice += nsub) {
    const int64_t g0 = bs_off[slice], g1 = bs_off[slice + 1];
    double s0 = 0.0, s1 = 0.0, s2 = 0.0;
    for (int64_t g = g0; g < g1; ++g) {
      const double* v = vals + (9 * g) * BSW + lane;
      // nontemporal: every line is used by exactly one instruction here (plain loads: 0.401 against 0.390 ms per iteration)
      const int32_t c = __builtin_nontemporal_load(cols + BSW * g + lane);
      const double a00 = __builtin_nontemporal_load(v), a01 = __builtin_nontemporal_load(v + 1 * BSW), a02 = __builtin_nontemporal_load(v + 2 * BSW);
      const double a10 = __builtin_nontemporal_load(v + 3 * BSW), a11 = __builtin_nontemporal_load(v + 4 * BSW), a12 = __builtin_nontemporal_load(v + 5 * BSW);
      const double a20 = __builtin_nontemporal_load(v + 6 * BSW), a21 = __builtin_nontemporal_load(v + 7 * BSW), a22 = __builtin_nontemporal_load(v + 8 * BSW);
      const Triple xc = *reinterpret_cast<const Triple*>(x + c);
      s0 += a00 * xc.a + a01 * xc.b + a02 * xc.c;
      s1 += a10 * xc.a + a11 * xc.b + a12 * xc.c;
      s2 += a20 * xc.a + a21 * xc.b + a22 * xc.c;
    }
    const int64_t b = slice * BSW + lane;
    if (b < nb) {
      const Triple xr = *reinterpret_cast<const Triple*>(x + 3 * b);
      const bool f0 = fixed != nullptr && fixed[3 * b], f1 = fixed != nullptr && fixed[3 * b + 1], f2 = fixed != nullptr && fixed[3 * b + 2];
      const double y0 = f0 ? xr.a : s0, y1 = f1 ? xr.b : s1, y2 = f2 ? xr.c : s2;
      y[3 * b] = y0; y[3 * b + 1] = y1; y[3 * b + 2] = y2;
      dot += xr.a * y0 + xr.b * y1 + xr.c * y2;
    }
  }
  if (partials != nullptr) {
    const double t = femo_block_sum<SH_BLOCK>(dot, lds);
    if (threadIdx.x == 0) partials[blockIdx.x] = t;
  }
}

__global__ void k_csr_diag_inv(int64_t n, const int64_t* __restrict__ rowptr, const int32_t* __restrict__ cols,
                               const double* __restrict__ vals, const uint8_t* __restrict__ fixed, double* __restrict__ dinv) {
  for (int64_t row = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; row < n; row += (int64_t)gridDim.x * blockDim.x) {
    double d = 1.0;
    if (fixed == nullptr || !fixed[row]) {
      d = 0.0;
      for (int64_t e = rowptr[row]; e < rowptr[row + 1]; ++e)
        if (cols[e] == row) d += vals[e];
    }
    dinv[row] = d != 0.0 ? 1.0 / d : 1.0;
  }
}

// sum of per-block partials, the same value in every thread of every block (fixed order: reproducible)
__device__ __forceinline__ double fold(const double* __restrict__ partials, int nb, double* lds) {
  double a = 0.0;
  for (int i = threadIdx.x; i < nb; i += SH_BLOCK) a += partials[i];
  a = femo_wave_sum(a);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) lds[w] = a;
  __syncthreads();
  double s = 0.0;
#pragma unroll
  for (int i = 0; i < SH_BLOCK / 64; ++i) s += lds[i];
  return s;
}

// scal: [0] gamma = r.z, [1] gamma0 (tolerance reference), [2] tol^2 factor
// r = b - A x0 is prepared by the host code; z = dinv r; p = z; partial r.z
__global__ __launch_bounds__(SH_BLOCK) void k_scg_init(int64_t n, const double* __restrict__ r, const double* __restrict__ dinv,
                                                       double* __restrict__ p, double* __restrict__ partials) {
  __shared__ double lds[SH_BLOCK / 64];
  double s = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * SH_BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * SH_BLOCK) {
    const double z = dinv[i] * r[i];
    p[i] = z;
    s += r[i] * z;
  }
  const double t = femo_block_sum<SH_BLOCK>(s, lds);
  if (threadIdx.x == 0) partials[blockIdx.x] = t;
}

__global__ __launch_bounds__(SH_BLOCK) void k_scg_gamma0(int nb, const double* __restrict__ partials, double rtol2, double atol2, double* __restrict__ scal,
                                                         int32_t* __restrict__ flag) {
  __shared__ double lds[SH_BLOCK / 64];
  const double g = fold(partials, nb, lds);
  if (threadIdx.x == 0) {
    scal[0] = g; scal[1] = g;
    scal[4] = g;                           // gamma as published by the first SpMV of the loop
    scal[2] = fmax(rtol2 * g, atol2);
    flag[0] = g <= scal[2] ? 1 : 0;
    flag[1] = 0;
  }
}

// x += alpha p; r -= alpha q; z = dinv r; partial r.z           alpha = gamma / (p.q), p.q folded here
__global__ __launch_bounds__(SH_BLOCK) void k_scg_xr(int64_t n, int nb_pq, const double* __restrict__ part_pq, const double* __restrict__ scal,
                                                     const double* __restrict__ p, const double* __restrict__ q, const double* __restrict__ dinv,
                                                     double* __restrict__ x, double* __restrict__ r, double* __restrict__ part_rz,
                                                     const int32_t* __restrict__ done) {
  if (*done) return;
  __shared__ double lds[SH_BLOCK / 64];
  const double pq = fold(part_pq, nb_pq, lds);
  const double alpha = pq != 0.0 ? scal[0] / pq : 0.0;
  double s = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * SH_BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * SH_BLOCK) {
    x[i] += alpha * p[i];
    const double ri = r[i] - alpha * q[i];
    r[i] = ri;
    s += ri * ri * dinv[i];
  }
  const double t = femo_block_sum<SH_BLOCK>(s, lds);
  if (threadIdx.x == 0) part_rz[blockIdx.x] = t;
}

// gamma' folded; beta = gamma'/gamma; p = dinv r + beta p; stopping test; one extra block-0 duty: publish gamma'
__global__ __launch_bounds__(SH_BLOCK) void k_scg_p(int64_t n, int it, int nb_rz, const double* __restrict__ part_rz, double* __restrict__ scal,
                                                    const double* __restrict__ r, const double* __restrict__ dinv, double* __restrict__ p,
                                                    int32_t* __restrict__ flag, double* __restrict__ gamma_out) {
  if (flag[0]) return;
  __shared__ double lds[SH_BLOCK / 64];
  const double g1 = fold(part_rz, nb_rz, lds);
  const double g0 = scal[0];
  const bool conv = g1 <= scal[2] || !(g1 == g1);
  const double beta = g0 != 0.0 ? g1 / g0 : 0.0;
  if (!conv) {
    for (int64_t i = (int64_t)blockIdx.x * SH_BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * SH_BLOCK)
      p[i] = dinv[i] * r[i] + beta * p[i];
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    gamma_out[0] = g1;                   // read by the next iteration only after this kernel has finished
    flag[1] = it + 1;
    if (conv) { flag[2] = (g1 == g1) ? 0 : 1; __threadfence(); flag[0] = it + 1; }
  }
}

// r = rhs on the free dofs, 0 on the strongly imposed ones (those are set exactly after the loop)
__global__ void k_rhs_free(int64_t n, const double* __restrict__ rhs, const uint8_t* __restrict__ fixed, double* __restrict__ r) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    r[i] = (fixed != nullptr && fixed[i]) ? 0.0 : rhs[i];
}

__global__ void k_set_fixed(int64_t n, const uint8_t* __restrict__ fixed, const double* __restrict__ xfix, double* __restrict__ x) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    if (fixed[i]) x[i] = xfix != nullptr ? xfix[i] : 0.0;
}

// lifting: x holds the prescribed values on fixed dofs and 0 elsewhere on entry of the caller's choice; b' = b - A_fc x_c on free rows
__global__ __launch_bounds__(SH_BLOCK) void k_csr_lift(int64_t n, const int64_t* __restrict__ rowptr, const int32_t* __restrict__ cols,
                                                       const double* __restrict__ vals, const uint8_t* __restrict__ fixed,
                                                       const double* __restrict__ xfix, const double* __restrict__ b, double* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int64_t nw = (int64_t)gridDim.x * (SH_BLOCK / 64);
  for (int64_t row = (int64_t)blockIdx.x * (SH_BLOCK / 64) + (threadIdx.x >> 6); row < n; row += nw) {
    double s = 0.0;
    if (!fixed[row]) {
      for (int64_t e = rowptr[row] + lane; e < rowptr[row + 1]; e += 64) {
        const int32_t cidx = cols[e];
        if (fixed[cidx]) s += vals[e] * xfix[cidx];
      }
    }
    s = femo_wave_sum(s);
    if (lane == 0) out[row] = fixed[row] ? xfix[row] : b[row] - s;
  }
}

// ------------------------------------------------------- lattice preconditioner ----
// M^-1 = D^-1 + sum_l P_l C_l P_l^T: P_l = trilinear interpolation from a lattice of spacing 2^-l x (bounding cube) to
// the dof nodes, component by component (3 displacement fields on the P2 nodes, 3 rotation fields on the vertices),
// C_l = 1 / diag(P_l^T K P_l) -- the Galerkin diagonal, which carries the h / h^3 scaling of the membrane and bending
// parts per component.  Measured with the oracle (Scordelis-Lo, rtol 1e-10): 979 / 1066 / 1149 iterations on
// 16^2 / 32^2 / 64^2 against 3171 / 7491 / 15139 for Jacobi -- nearly mesh independent where Jacobi grows like n.
// The lattices are nested (P_l = P_{l+1} T_l exactly), so only the finest one touches the dofs: g_L = P_L^T r, then
// g_l = T_l^T g_{l+1} down the hierarchy, e_0 = C_0 g_0, e_{l+1} = C_{l+1} g_{l+1} + T_l e_l up again, z = D^-1 r +
// P_L e_L.  (The first version applied every level's P_l directly: the rows of P_0^T have n_dof / 4 entries, 2.4 ms
// per iteration at 248 k dofs.)  P of every level is kept in ELL form for the Galerkin diagonals.

// diag[j] += sum_i sum_k P[i,j] K[i,k] P[k,j] over the free dofs: one thread per (row i, ELL slot a)
__global__ __launch_bounds__(SH_BLOCK) void k_pc_galerkin_diag(int64_t n, int width, const int64_t* __restrict__ rowptr, const int32_t* __restrict__ cols,
                                                               const double* __restrict__ vals, const uint8_t* __restrict__ fixed,
                                                               const int32_t* __restrict__ ell_idx, const double* __restrict__ ell_w,
                                                               double* __restrict__ diag, int first_slot) {
  // slots first_slot .. width - 1 of every row (the levels below belong to the coarse solve when there is one)
  const int wact = width - first_slot;
  const int64_t t = (int64_t)blockIdx.x * SH_BLOCK + threadIdx.x;
  const int64_t i = t / wact;
  const int a = first_slot + (int)(t % wact);
  if (i >= n || (fixed != nullptr && fixed[i])) return;
  const double wi = ell_w[i * width + a];
  if (wi == 0.0) return;
  const int32_t j = ell_idx[i * width + a];
  const int lev8 = (a >> 3) << 3;
  double acc = 0.0;
  for (int64_t e = rowptr[i]; e < rowptr[i + 1]; ++e) {
    const int32_t k = cols[e];
    if (fixed != nullptr && fixed[k]) continue;
    const int32_t* ik = ell_idx + (int64_t)k * width + lev8;
    const double* wk = ell_w + (int64_t)k * width + lev8;
    double pk = 0.0;
#pragma unroll
    for (int b = 0; b < 8; ++b) pk += ik[b] == j ? wk[b] : 0.0;
    acc += vals[e] * pk;
  }
  atomicAdd(&diag[j], wi * acc);
}

// Dense Galerkin operator of one coarse lattice level c: A[a, b] = sum_{i, j free} P[i, a] K[i, j] P[j, b], 6 n_c x 6 n_c.
// One workgroup per item = points (256 by default) of one coarse cell and one field group (displacements of P2 nodes /
// rotations of vertices): they share the cell's eight nodes a, and the eight nodes of any j they couple to lie in the
// 4 x 4 x 4 node neighbourhood of the cell (an element is smaller than a coarse cell; info[1] reports otherwise).  The
// sums over the item never leave the workgroup: table[3 x 6 components][64 neighbourhood nodes][8 a] in LDS (72 KB),
// flushed to the dense matrix once (item_nbr: the neighbourhood's level-local node numbers, -1 where the surface does
// not touch the lattice).  The item's 3 x 3 blocks are staged through LDS 256 at a time, one block per thread (values
// with the Dirichlet mask applied, the column's eight weights, its cell offset); then a wave takes one staged block
// per step, its 64 lanes being the 8 x 8 pairs (cell node a, corner k' of the column's cell), and adds the nine
// products with ds_add_f64.  History at 1.97 M dofs (12 M blocks): all threads following one block through global
// memory, sums in registers of the thread that owns (a, neighbourhood node): 27 ms; blocks staged through LDS, same
// ownership (one (thread, block) pair in eight has work): 20 ms whatever the item size; LDS table with a 16-way bank
// conflict: 16.4 ms; this layout: 8.1 ms.
constexpr int CG_MAXPTS = 256;
constexpr int CG_TABLE = 8 * 64 * 18;                     // doubles
constexpr size_t CG_LDS = (size_t)CG_TABLE * 8 + 256 * 9 * 8 + 256 * 8 * 8 + CG_MAXPTS * 8 * 8 + 256 * 4 + (CG_MAXPTS + 1) * 4 + CG_MAXPTS * 4 * 2;
__global__ __launch_bounds__(256) void k_pc_coarse_galerkin(int c, int width, int64_t off_c, int64_t lda, int64_t n_unode,
                                                            const int64_t* __restrict__ item_ptr, const int32_t* __restrict__ item_pts,
                                                            const int32_t* __restrict__ item_nbr, const int32_t* __restrict__ node_xyz,
                                                            const int32_t* __restrict__ pcell, const int64_t* __restrict__ brow,
                                                            const int32_t* __restrict__ bcols, const double* __restrict__ vals,
                                                            const uint8_t* __restrict__ fixed, const int32_t* __restrict__ ell_idx,
                                                            const double* __restrict__ ell_w, double* __restrict__ A, int32_t* __restrict__ info) {
  extern __shared__ double cg_lds[];
  double* table = cg_lds;                                                  // [3 fa][6][64 bl][8 a]
  double (*s_val)[9] = reinterpret_cast<double (*)[9]>(table + CG_TABLE);  // staged 3 x 3 blocks, Dirichlet mask applied
  double (*s_wb)[8] = reinterpret_cast<double (*)[8]>(s_val + 256);        // the column's weights on its cell's nodes
  double (*s_wpt)[8] = reinterpret_cast<double (*)[8]>(s_wb + 256);        // the item's points: weights on the cell's nodes
  int32_t* s_meta = reinterpret_cast<int32_t*>(s_wpt + CG_MAXPTS);         // ox | oy << 2 | oz << 4 | gj << 6 | q << 8, or -1
  int32_t* s_scan = s_meta + 256;                                          // blocks before point q of the chunk
  int32_t* s_k0 = s_scan + CG_MAXPTS + 1;
  int32_t* s_fi = s_k0 + CG_MAXPTS;
  const int64_t item = blockIdx.x;
  const int t = threadIdx.x;
  const int64_t pbeg = item_ptr[item], pend = item_ptr[item + 1];
  const int32_t pfirst = item_pts[pbeg];
  const int gi = pfirst >= n_unode ? 1 : 0;
  const int64_t e0 = (int64_t)(3 * pfirst) * width + 8 * c;
  const int32_t pk0 = pcell[pfirst];
  const int bx = pk0 & 1023, by = (pk0 >> 10) & 1023, bz = pk0 >> 20;
  for (int idx = t; idx < CG_TABLE; idx += 256) table[idx] = 0.0;
  int far = 0;
  const int wv = t >> 6, la = t & 7, lk = (t >> 3) & 7;                     // wave, cell node a, corner k' of the column's cell
  for (int64_t p0 = pbeg; p0 < pend; p0 += CG_MAXPTS) {
    const int npts = (int)min((int64_t)CG_MAXPTS, pend - p0);
    __syncthreads();
    if (t < npts) {
      const int32_t i = item_pts[p0 + t];
      const int64_t k0 = brow[i];
      s_k0[t] = (int32_t)k0;
      s_scan[t + 1] = (int32_t)(brow[i + 1] - k0);
      s_fi[t] = fixed == nullptr ? 0 : (fixed[3 * i] ? 1 : 0) | (fixed[3 * i + 1] ? 2 : 0) | (fixed[3 * i + 2] ? 4 : 0);
    }
    for (int idx = t; idx < npts * 8; idx += 256) {
      const int32_t i = item_pts[p0 + (idx >> 3)];
      s_wpt[idx >> 3][idx & 7] = ell_w[(int64_t)(3 * i) * width + 8 * c + (idx & 7)];
    }
    __syncthreads();
    if (t == 0) {
      int32_t run = 0;
      s_scan[0] = 0;
      for (int q = 0; q < npts; ++q) { run += s_scan[q + 1]; s_scan[q + 1] = run; }
    }
    __syncthreads();
    const int B = s_scan[npts];
    for (int base = 0; base < B; base += 256) {
      const int f = base + t;
      if (f < B) {
        int lo = 0, hi = npts - 1;                         // largest q with s_scan[q] <= f
        while (lo < hi) {
          const int mid = (lo + hi + 1) >> 1;
          if (s_scan[mid] <= f) lo = mid; else hi = mid - 1;
        }
        const int q = lo, lkk = f - s_scan[q];
        const int64_t k0 = s_k0[q];
        const int64_t len = 3 * (int64_t)(s_scan[q + 1] - s_scan[q]);
        const int32_t cj = bcols[k0 + lkk];
        const int32_t pk = pcell[cj / 3];
        const int ox = (pk & 1023) - bx + 1, oy = ((pk >> 10) & 1023) - by + 1, oz = (pk >> 20) - bz + 1;
        if ((unsigned)ox > 2u || (unsigned)oy > 2u || (unsigned)oz > 2u) {
          far = 1;
          s_meta[t] = -1;
        } else {
          s_meta[t] = ox | (oy << 2) | (oz << 4) | ((cj >= 3 * n_unode ? 1 : 0) << 6) | (q << 8);
          const double* v = vals + 9 * k0 + 3 * lkk;
          const int fi = s_fi[q];
          const int fj = fixed == nullptr ? 0 : (fixed[cj] ? 1 : 0) | (fixed[cj + 1] ? 2 : 0) | (fixed[cj + 2] ? 4 : 0);
#pragma unroll
          for (int fa = 0; fa < 3; ++fa)
#pragma unroll
            for (int fb = 0; fb < 3; ++fb) s_val[t][3 * fa + fb] = ((fi >> fa) & 1) || ((fj >> fb) & 1) ? 0.0 : v[fa * len + fb];
          const double* wj = ell_w + (int64_t)cj * width + 8 * c;
#pragma unroll
          for (int b = 0; b < 8; ++b) s_wb[t][b] = wj[b];
        }
      }
      __syncthreads();
      const int cnt = min(256, B - base);
      for (int e = wv; e < cnt; e += 4) {                  // one staged block per wave and step
        const int32_t m = s_meta[e];
        if (m < 0) continue;
        const int ox = m & 3, oy = (m >> 2) & 3, oz = (m >> 4) & 3, gj = (m >> 6) & 1, q = m >> 8;
        const double ww = s_wpt[q][la] * s_wb[e][lk];
        const int bl = (ox + (lk & 1)) + 4 * (oy + ((lk >> 1) & 1)) + 16 * (oz + (lk >> 2));
        // table[component][bl][a], a fastest: the 64 lanes of an update spread over all banks (4 lanes per 8-byte
        // bank pair, the minimum); with [a][bl][component] the eight a and the two z corners shared a bank, a 16-way
        // conflict on every ds_add_f64 (16.4 ms at 1.97 M dofs)
        double* dst = table + ((3 * gj) * 64 + bl) * 8 + la;
#pragma unroll
        for (int fa = 0; fa < 3; ++fa)
#pragma unroll
          for (int fb = 0; fb < 3; ++fb)
            __hip_atomic_fetch_add(dst + (6 * fa + fb) * 512, ww * s_val[e][3 * fa + fb], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
      __syncthreads();
    }
  }
  if (far && info != nullptr) atomicOr(&info[1], 1);
  __syncthreads();
  // flush: thread t owns node a = t & 7 against neighbourhood nodes t >> 3 and (t >> 3) + 32
  {
    const int a = t & 7, blq = t >> 3;
    const int64_t na = ell_idx[e0 + a] / 6 - off_c;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int bl = blq + 32 * s;
      const int32_t nb = item_nbr[item * 64 + bl];
      if (nb < 0) continue;
      const double* src = table + bl * 8 + a;
#pragma unroll
      for (int fa = 0; fa < 3; ++fa)
#pragma unroll
        for (int f = 0; f < 6; ++f) {
          const double v = src[(6 * fa + f) * 512];
          if (v != 0.0) atomicAdd(&A[(6 * na + 3 * gi + fa) * lda + 6 * (int64_t)nb + f], v);
        }
    }
  }
}


// ---- Galerkin set-up for the Hermite-type lattice spaces ------------------------------------------------------------------
// W_p,n = [alpha I, -[sigma]x] (3 x 6) for a displacement point, [0, w I] for a rotation point: blk[n] = sum over the pairs
// of points (p, q) that both touch node n of W_p,n^T K_pq W_q,n.  Same organisation as k_pc_galerkin_blocks (a thread per
// (point, level, component fa), the row's sums against the point's eight nodes in registers), but a row dof of a
// displacement point now feeds three rows of the block (U_fa with alpha, two Theta rows with -+sigma), so the LDS table is
// keyed by node and holds whole 6 x 6 blocks (upper triangle used).
__global__ __launch_bounds__(SH_BLOCK) void k_pc_galerkin_blocks_h(int64_t n_pts, int64_t n_unode, const int64_t* __restrict__ brow,
                                                                   const int32_t* __restrict__ bcols, const double* __restrict__ vals,
                                                                   const uint8_t* __restrict__ fixed, const int32_t* __restrict__ lvl_node,
                                                                   const float4* __restrict__ lvl_w4, double* __restrict__ blk) {
  constexpr int HS = 256;
  __shared__ int32_t h_key[HS];
  __shared__ double h_val[HS][36];
  for (int i = threadIdx.x; i < HS; i += SH_BLOCK) h_key[i] = -1;
  for (int i = threadIdx.x; i < HS * 36; i += SH_BLOCK) (&h_val[0][0])[i] = 0.0;
  __syncthreads();
  const int fa = (int)(blockIdx.y % 3);
  const int32_t* ln = lvl_node + (int64_t)(blockIdx.y / 3) * n_pts * 8;
  const float4* lw = lvl_w4 + (int64_t)(blockIdx.y / 3) * n_pts * 8;
  const int64_t p = (int64_t)blockIdx.x * SH_BLOCK + threadIdx.x;
  const bool active = p < n_pts && !(fixed != nullptr && fixed[3 * p + fa]);
  if (active) {
    int32_t nd[8];
    float4 wi[8];
    double acc[8][6];
#pragma unroll
    for (int a = 0; a < 8; ++a) {
      nd[a] = ln[p * 8 + a];
      wi[a] = lw[p * 8 + a];
      if (wi[a].x == 0.f && wi[a].y == 0.f && wi[a].z == 0.f && wi[a].w == 0.f) nd[a] = -1;
#pragma unroll
      for (int q = 0; q < 6; ++q) acc[a][q] = 0.0;
    }
    const bool pu = p < n_unode;
    const int64_t k0 = brow[p], k1 = brow[p + 1], len = 3 * (k1 - k0);
    const double* v = vals + 9 * k0 + fa * len;
    for (int64_t k = k0; k < k1; ++k) {
      const int32_t cj = bcols[k];
      const int32_t* ik = ln + (int64_t)(cj / 3) * 8;
      const float4* wk = lw + (int64_t)(cj / 3) * 8;
      const int64_t o = 3 * (k - k0);
      double m0 = v[o], m1 = v[o + 1], m2 = v[o + 2];
      if (fixed != nullptr) {
        if (fixed[cj]) m0 = 0.0;
        if (fixed[cj + 1]) m1 = 0.0;
        if (fixed[cj + 2]) m2 = 0.0;
      }
      const bool qu = cj < 3 * n_unode;
#pragma unroll
      for (int b = 0; b < 8; ++b) {
        const int32_t nb = ik[b];
        const float4 w = wk[b];
        // the row vector m (1 x 3) times W_q,b: [alpha m, sigma x m] for a displacement column, [0, w m] for a rotation column
        double c0, c1, c2, c3, c4, c5;
        if (qu) {
          c0 = (double)w.x * m0; c1 = (double)w.x * m1; c2 = (double)w.x * m2;
          c3 = (double)w.z * m2 - (double)w.w * m1; c4 = (double)w.w * m0 - (double)w.y * m2; c5 = (double)w.y * m1 - (double)w.z * m0;
        } else {
          c0 = c1 = c2 = 0.0;
          c3 = (double)w.x * m0; c4 = (double)w.x * m1; c5 = (double)w.x * m2;
        }
#pragma unroll
        for (int a = 0; a < 8; ++a) {
          const double on = nb == nd[a] ? 1.0 : 0.0;
          acc[a][0] += on * c0; acc[a][1] += on * c1; acc[a][2] += on * c2;
          acc[a][3] += on * c3; acc[a][4] += on * c4; acc[a][5] += on * c5;
        }
      }
    }
#pragma unroll
    for (int a = 0; a < 8; ++a) {
      if (nd[a] < 0) continue;
      const int32_t key = nd[a];
      int h = (int)(((uint32_t)key * 2654435761u) >> 24) & (HS - 1);
      int slot = -1;
      for (int probe = 0; probe < 16; ++probe) {
        const int32_t seen = atomicCAS(&h_key[h], -1, key);
        if (seen == -1 || seen == key) { slot = h; break; }
        h = (h + 1) & (HS - 1);
      }
      // rows of the block this dof feeds: (row, coefficient)
      int rw[3];
      double cf[3];
      int nrow;
      if (pu) {
        const double sg[3] = {(double)wi[a].y, (double)wi[a].z, (double)wi[a].w};
        const int k1i = (fa + 1) % 3, k2i = (fa + 2) % 3;          // (Theta x sigma)_fa = Theta_k1 sigma_k2 - Theta_k2 sigma_k1
        rw[0] = fa; cf[0] = (double)wi[a].x;
        rw[1] = 3 + k1i; cf[1] = sg[k2i];
        rw[2] = 3 + k2i; cf[2] = -sg[k1i];
        nrow = 3;
      } else {
        rw[0] = 3 + fa; cf[0] = (double)wi[a].x;
        nrow = 1;
      }
      for (int rI = 0; rI < nrow; ++rI) {
        if (cf[rI] == 0.0) continue;
#pragma unroll
        for (int q = 0; q < 6; ++q) {
          if (q < rw[rI] || acc[a][q] == 0.0) continue;
          const double val = cf[rI] * acc[a][q];
          if (slot >= 0) __hip_atomic_fetch_add(&h_val[slot][6 * rw[rI] + q], val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          else atomicAdd(&blk[36 * (int64_t)key + 6 * rw[rI] + q], val);
        }
      }
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < HS * 36; i += SH_BLOCK) {
    const int32_t key = h_key[i / 36];
    if (key < 0) continue;
    const double val = (&h_val[0][0])[i];
    if (val != 0.0) atomicAdd(&blk[36 * (int64_t)key + (i % 36)], val);
  }
}

// 6 x 6 Galerkin node blocks of the levels above the coarse solve, Hermite-type spaces, second version (round 4).  The
// version above walks every row three times per level from a thread per (point, level, component): K was fetched nine
// times (9 GB at 1.97 M dofs, 10.5 ms).  Here a WAVE takes an item -- at most 64 points of one cell of one level, so that
// all of them share the cell's eight nodes -- with lanes = (node a of the cell, column f' of the block):
//   per point p: the lanes first fetch the point's column indices and the level-l cells of its column points (one per lane:
//   two dependent loads per POINT, not per block); then, block by block, M[fa] += (K_pq)[fa][.] . u with u the column f' of
//   W_q,b and b the corner of q's cell that IS node a (b = a - cell offset; no match: nothing) -- the weight gathers of
//   the blocks are independent of each other; last acc[r] += (W_p,a^T M)[r] (r = 0..5);
//   per item: one flush of the upper triangles with atomics (the items of a cell and the neighbouring cells share nodes).
// The cell offset of a column point is at most one cell per axis when elements are smaller than the cells of the finest
// lattice; a larger one sets info[1] bit 1 and the caller falls back to the kernel above.
__global__ void k_point_fixbits(int64_t n_pts, const uint8_t* __restrict__ fixed, uint8_t* __restrict__ bits) {
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n_pts) return;
  bits[p] = fixed == nullptr ? 0 : (uint8_t)((fixed[3 * p] ? 1 : 0) | (fixed[3 * p + 1] ? 2 : 0) | (fixed[3 * p + 2] ? 4 : 0));
}

__global__ __launch_bounds__(256) void k_pc_galerkin_blocks_w(int64_t n_items, int64_t n_pts, int64_t n_unode, const int64_t* __restrict__ item_ptr,
                                                              const int32_t* __restrict__ item_lvl, const int32_t* __restrict__ item_pts,
                                                              const int32_t* __restrict__ pcell, const int64_t* __restrict__ brow,
                                                              const int32_t* __restrict__ bcols, const double* __restrict__ vals,
                                                              const uint8_t* __restrict__ fixbits, const int32_t* __restrict__ lvl_node,
                                                              const float4* __restrict__ lvl_w4, double* __restrict__ blk, int32_t* __restrict__ info) {
  constexpr int EC = 32;                                                   // blocks staged per round
  __shared__ double s_K[4][3][3 * EC];                                     // rows fa of the staged blocks (columns of fixed dofs zeroed)
  __shared__ float4 s_W[4][EC * 8];                                        // [block e][node a of THIS cell]: weight of the corner of q's cell that is node a, or 0
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int64_t item = (int64_t)blockIdx.x * 4 + wv;
  if (item >= n_items) return;
  const int a = lane & 7, fc = lane >> 3;                                  // node of the cell, column of the block (fc < 6)
  // column f' = fc of W_q,b as a 3-vector u = (sg[k] * w[ix[k]])_k with w = (alpha, sigma) of the corner: displacement column
  // fc < 3: alpha e_fc; fc = 3: (0, -s2, s1); 4: (s2, 0, -s0); 5: (-s1, s0, 0); rotation column: w e_(fc-3) for fc >= 3
  int ixu[3] = {0, 0, 0};
  double sgu[3] = {0.0, 0.0, 0.0}, sgr[3] = {0.0, 0.0, 0.0};
  if (fc < 3) sgu[fc] = 1.0;
  else if (fc == 3) { ixu[1] = 3; sgu[1] = -1.0; ixu[2] = 2; sgu[2] = 1.0; sgr[0] = 1.0; }
  else if (fc == 4) { ixu[0] = 3; sgu[0] = 1.0; ixu[2] = 1; sgu[2] = -1.0; sgr[1] = 1.0; }
  else if (fc == 5) { ixu[0] = 2; sgu[0] = -1.0; ixu[1] = 1; sgu[1] = 1.0; sgr[2] = 1.0; }
  const int64_t pbeg = item_ptr[item], pend = item_ptr[item + 1];
  const int lv = item_lvl[item];
  const int32_t* ln = lvl_node + (int64_t)lv * n_pts * 8;
  const float4* lw = lvl_w4 + (int64_t)lv * n_pts * 8;
  const int32_t* pc = pcell + (int64_t)lv * n_pts;
  const int32_t pfirst = item_pts[pbeg];
  const int32_t node_a = ln[(int64_t)pfirst * 8 + a];
  const int32_t ck = pc[pfirst];
  const int cx = ck & 1023, cy = (ck >> 10) & 1023, cz = ck >> 20;
  double acc[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
  int far = 0;
  double (*sK)[3 * EC] = s_K[wv];
  float4* sW = s_W[wv];
  const float* sWf = reinterpret_cast<const float*>(sW);
  // three-stage software pipeline over the item's points: every iteration issues ONE level of the chain point id -> row
  // extent (+ the point's own weights and Dirichlet bits) -> column indices for a later point, after the loads of the current
  // point, so that a point's round starts with its column indices in registers
  const int64_t npt = pend - pbeg;
  int32_t p_2 = npt > 2 ? item_pts[pbeg + 2] : 0;
  int32_t p_1 = npt > 1 ? item_pts[pbeg + 1] : 0;
  int64_t k0_1 = 0; int nb_1 = 0; float4 wp_1 = float4{0.f, 0.f, 0.f, 0.f}; int fi_1 = 0;
  if (npt > 1) {
    k0_1 = brow[p_1]; nb_1 = (int)(brow[p_1 + 1] - k0_1); wp_1 = lw[(int64_t)p_1 * 8 + a]; fi_1 = fixbits == nullptr ? 0 : fixbits[p_1];
  }
  int32_t p_0 = pfirst;
  int64_t k0_0 = brow[p_0];
  int nb_0 = (int)(brow[p_0 + 1] - k0_0);
  float4 wp_0 = lw[(int64_t)p_0 * 8 + a];
  int fi_0 = fixbits == nullptr ? 0 : fixbits[p_0];
  int32_t cj_0 = lane < min(EC, nb_0) ? bcols[k0_0 + lane] : 0;
  auto advance = [&](int64_t ip) {
    const int64_t left = pend - ip;
    cj_0 = left > 1 && lane < min(EC, nb_1) ? bcols[k0_1 + lane] : 0;
    p_0 = p_1; k0_0 = k0_1; nb_0 = nb_1; wp_0 = wp_1; fi_0 = fi_1;
    if (left > 2) {
      k0_1 = brow[p_2]; nb_1 = (int)(brow[p_2 + 1] - k0_1); wp_1 = lw[(int64_t)p_2 * 8 + a]; fi_1 = fixbits == nullptr ? 0 : fixbits[p_2];
    }
    p_1 = p_2;
    p_2 = left > 3 ? item_pts[ip + 3] : 0;
  };
  for (int64_t ip = pbeg; ip < pend; ++ip) {
    const int32_t p = p_0;
    const int64_t k0 = k0_0;
    const int nb = nb_0;
    const int64_t len = 3 * (int64_t)nb;
    const float4 wp = wp_0;
    const int fi = fi_0;
    const int32_t cj_first = cj_0;
    const bool pu = p < n_unode;
    double M0 = 0.0, M1 = 0.0, M2 = 0.0;
    if (nb == 0) advance(ip);
    for (int e0 = 0; e0 < nb; e0 += EC) {
      const int cnt = min(EC, nb - e0);
      // the round's loads, all issued before anything is used: the blocks' rows (3 x 3 cnt contiguous doubles), lane e < cnt the
      // column index of block e0 + e, then -- one dependent step -- its level-l cell and Dirichlet bits and the eight weights
      double kv[2][3];
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int fa = 0; fa < 3; ++fa) {
          const int idx = lane + 64 * h;
          kv[h][fa] = idx < 3 * cnt ? vals[9 * k0 + fa * len + 3 * e0 + idx] : 0.0;
        }
      int32_t my_cj = cj_first, my_meta = -1;
      if (e0 > 0) my_cj = lane < cnt ? bcols[k0 + e0 + lane] : 0;
      float4 wl[4];
#pragma unroll
      for (int h = 0; h < 4; ++h) {
        const int idx = lane + 64 * h, e = idx >> 3;
        const int32_t cje = __shfl(my_cj, e);
        wl[h] = e < cnt ? lw[(int64_t)(cje / 3) * 8 + (idx & 7)] : float4{0.f, 0.f, 0.f, 0.f};
      }
      if (lane < cnt) {
        const int32_t q = my_cj / 3;
        const int32_t pk = pc[q];
        const int ox = (pk & 1023) - cx + 1, oy = ((pk >> 10) & 1023) - cy + 1, oz = (pk >> 20) - cz + 1;
        if ((unsigned)ox > 2u || (unsigned)oy > 2u || (unsigned)oz > 2u) far = 1;
        else my_meta = ox | (oy << 2) | (oz << 4) | ((fixbits == nullptr ? 0 : fixbits[q]) << 6) | ((my_cj < 3 * n_unode ? 1 : 0) << 9);
      }
      if (e0 == 0) advance(ip);                            // the pipeline's loads for the points after this one
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int idx = lane + 64 * h;
        const int32_t me = __shfl(my_meta, idx / 3);
        const bool off = idx >= 3 * cnt || me < 0 || ((me >> (6 + idx % 3)) & 1);       // column of a fixed dof
#pragma unroll
        for (int fa = 0; fa < 3; ++fa)
          if (idx < 3 * EC) sK[fa][idx] = off ? 0.0 : kv[h][fa];
      }
      // weights: entry (e, corner b of q's cell) goes to the slot of the node a = b + cell offset of THIS cell, if it is one
#pragma unroll
      for (int h = 0; h < 4; ++h) sW[lane + 64 * h] = float4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int h = 0; h < 4; ++h) {
        const int idx = lane + 64 * h, e = idx >> 3, b = idx & 7;
        const int32_t me = __shfl(my_meta, e);
        const int tx = (b & 1) + (me & 3) - 1, ty = ((b >> 1) & 1) + ((me >> 2) & 3) - 1, tz = (b >> 2) + ((me >> 4) & 3) - 1;
        if (e < cnt && me >= 0 && (unsigned)tx < 2u && (unsigned)ty < 2u && (unsigned)tz < 2u) sW[8 * e + tx + 2 * ty + 4 * tz] = wl[h];
      }
      const uint64_t qmask = __ballot(lane < cnt && my_meta >= 0 && ((my_meta >> 9) & 1));      // blocks with a displacement column
      // LDS only (the wave's LDS operations execute in order): a fence would also wait for the loads of the pipeline
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      const float* wa = sWf + 4 * a;
#pragma unroll 2
      for (int e = 0; e < cnt; ++e) {
        const float* w = wa + 32 * e;
        double u0, u1, u2;
        if ((qmask >> e) & 1) { u0 = sgu[0] * (double)w[ixu[0]]; u1 = sgu[1] * (double)w[ixu[1]]; u2 = sgu[2] * (double)w[ixu[2]]; }
        else { const double al = (double)w[0]; u0 = sgr[0] * al; u1 = sgr[1] * al; u2 = sgr[2] * al; }
        const double* k = &sK[0][3 * e];
        M0 += k[0] * u0 + k[1] * u1 + k[2] * u2;
        M1 += k[3 * EC] * u0 + k[3 * EC + 1] * u1 + k[3 * EC + 2] * u2;
        M2 += k[6 * EC] * u0 + k[6 * EC + 1] * u1 + k[6 * EC + 2] * u2;
      }
      // LDS only (the wave's LDS operations execute in order): a fence would also wait for the loads of the pipeline
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    if (fi & 1) M0 = 0.0;
    if (fi & 2) M1 = 0.0;
    if (fi & 4) M2 = 0.0;
    if (pu) {
      const double al = (double)wp.x, s0 = (double)wp.y, s1 = (double)wp.z, s2 = (double)wp.w;
      acc[0] += al * M0; acc[1] += al * M1; acc[2] += al * M2;
      acc[3] += -s2 * M1 + s1 * M2;
      acc[4] += s2 * M0 - s0 * M2;
      acc[5] += -s1 * M0 + s0 * M1;
    } else {
      const double w = (double)wp.x;
      acc[3] += w * M0; acc[4] += w * M1; acc[5] += w * M2;
    }
  }
  if (far && info != nullptr) atomicOr(&info[1], 2);
  if (fc < 6 && node_a >= 0) {
#pragma unroll
    for (int r = 0; r < 6; ++r)
      if (r <= fc && acc[r] != 0.0) atomicAdd(&blk[36 * (int64_t)node_a + 6 * r + fc], acc[r]);
  }
}

// Dense Galerkin operator of the coarse-solve level for the Hermite-type spaces: A[6 a + f, 6 b + f'] with the composed
// (alpha, sigma) of that level.  Organisation of k_pc_coarse_galerkin (items = points of one coarse cell and one field
// group; the item's 3 x 3 blocks staged through LDS; a wave per staged block, lanes = (cell node a, corner k' of the
// column's cell); sums in an LDS table, flushed once).  pass 0: rows U of the displacement items against all six columns,
// T[fa][f] += alpha_p (K W_q)[fa][f]; pass 1: rows Theta against the Theta columns only -- [sigma_p]x K W_q^Theta for a
// displacement item, w_p K W_q^Theta for a rotation item; the (Theta, U) quadrant is the transpose of (U, Theta)
// (k_pc_coarse_mirror_tu): 27 LDS atomics per (block, a, k') in place of 36.
constexpr int CGH_CHUNK = 128;
constexpr size_t CGH_LDS = (size_t)CG_TABLE * 8 + CGH_CHUNK * 9 * 8 + CGH_CHUNK * 8 * 16 + CGH_CHUNK * 8 * 16 + CGH_CHUNK * 4 + (CGH_CHUNK + 1) * 4 + CGH_CHUNK * 4 * 2;
__global__ __launch_bounds__(256) void k_pc_coarse_galerkin_h(int pass, int c, int width, int64_t off_c, int64_t lda, int64_t n_unode,
                                                              const int64_t* __restrict__ item_ptr, const int32_t* __restrict__ item_pts,
                                                              const int32_t* __restrict__ item_nbr, const int32_t* __restrict__ pcell,
                                                              const int64_t* __restrict__ brow, const int32_t* __restrict__ bcols,
                                                              const double* __restrict__ vals, const uint8_t* __restrict__ fixed,
                                                              const int32_t* __restrict__ ell_idx, const float4* __restrict__ cs_w4,
                                                              double* __restrict__ A, int32_t* __restrict__ info) {
  extern __shared__ double cg_lds[];
  double* table = cg_lds;                                                  // [3 rows][6 cols][64 bl][8 a]
  double (*s_val)[9] = reinterpret_cast<double (*)[9]>(table + CG_TABLE);
  float4 (*s_wb)[8] = reinterpret_cast<float4 (*)[8]>(s_val + CGH_CHUNK);
  float4 (*s_wpt)[8] = reinterpret_cast<float4 (*)[8]>(s_wb + CGH_CHUNK);
  int32_t* s_meta = reinterpret_cast<int32_t*>(s_wpt + CGH_CHUNK);
  int32_t* s_scan = s_meta + CGH_CHUNK;
  int32_t* s_k0 = s_scan + CGH_CHUNK + 1;
  int32_t* s_fi = s_k0 + CGH_CHUNK;
  const int64_t item = blockIdx.x;
  const int t = threadIdx.x;
  const int64_t pbeg = item_ptr[item], pend = item_ptr[item + 1];
  const int32_t pfirst = item_pts[pbeg];
  const int gi = pfirst >= n_unode ? 1 : 0;
  if (pass == 0 && gi == 1) return;                                        // rotation points have no U rows
  const int64_t e0 = (int64_t)(3 * pfirst) * width + 8 * c;
  const int32_t pk0 = pcell[pfirst];
  const int bx = pk0 & 1023, by = (pk0 >> 10) & 1023, bz = pk0 >> 20;
  for (int idx = t; idx < CG_TABLE; idx += 256) table[idx] = 0.0;
  int far = 0;
  const int wv = t >> 6, la = t & 7, lk = (t >> 3) & 7;
  for (int64_t p0 = pbeg; p0 < pend; p0 += CGH_CHUNK) {
    const int npts = (int)min((int64_t)CGH_CHUNK, pend - p0);
    __syncthreads();
    if (t < npts) {
      const int32_t i = item_pts[p0 + t];
      const int64_t k0 = brow[i];
      s_k0[t] = (int32_t)k0;
      s_scan[t + 1] = (int32_t)(brow[i + 1] - k0);
      s_fi[t] = fixed == nullptr ? 0 : (fixed[3 * i] ? 1 : 0) | (fixed[3 * i + 1] ? 2 : 0) | (fixed[3 * i + 2] ? 4 : 0);
    }
    for (int idx = t; idx < npts * 8; idx += 256) {
      const int32_t i = item_pts[p0 + (idx >> 3)];
      s_wpt[idx >> 3][idx & 7] = cs_w4[(int64_t)i * 8 + (idx & 7)];
    }
    __syncthreads();
    if (t == 0) {
      int32_t run = 0;
      s_scan[0] = 0;
      for (int q = 0; q < npts; ++q) { run += s_scan[q + 1]; s_scan[q + 1] = run; }
    }
    __syncthreads();
    const int B = s_scan[npts];
    for (int base = 0; base < B; base += CGH_CHUNK) {
      const int f = base + t;
      if (t < CGH_CHUNK && f < B) {
        int lo = 0, hi = npts - 1;
        while (lo < hi) {
          const int mid = (lo + hi + 1) >> 1;
          if (s_scan[mid] <= f) lo = mid; else hi = mid - 1;
        }
        const int q = lo, lkk = f - s_scan[q];
        const int64_t k0 = s_k0[q];
        const int64_t len = 3 * (int64_t)(s_scan[q + 1] - s_scan[q]);
        const int32_t cj = bcols[k0 + lkk];
        const int32_t pk = pcell[cj / 3];
        const int ox = (pk & 1023) - bx + 1, oy = ((pk >> 10) & 1023) - by + 1, oz = (pk >> 20) - bz + 1;
        if ((unsigned)ox > 2u || (unsigned)oy > 2u || (unsigned)oz > 2u) {
          far = 1;
          s_meta[t] = -1;
        } else {
          s_meta[t] = ox | (oy << 2) | (oz << 4) | ((cj >= 3 * n_unode ? 1 : 0) << 6) | (q << 8);
          const double* v = vals + 9 * k0 + 3 * lkk;
          const int fi = s_fi[q];
          const int fj = fixed == nullptr ? 0 : (fixed[cj] ? 1 : 0) | (fixed[cj + 1] ? 2 : 0) | (fixed[cj + 2] ? 4 : 0);
#pragma unroll
          for (int fa = 0; fa < 3; ++fa)
#pragma unroll
            for (int fb = 0; fb < 3; ++fb) s_val[t][3 * fa + fb] = ((fi >> fa) & 1) || ((fj >> fb) & 1) ? 0.0 : v[fa * len + fb];
          const float4* wj = cs_w4 + (int64_t)(cj / 3) * 8;
#pragma unroll
          for (int b = 0; b < 8; ++b) s_wb[t][b] = wj[b];
        }
      }
      __syncthreads();
      const int cnt = min(CGH_CHUNK, B - base);
      for (int e = wv; e < cnt; e += 4) {
        const int32_t m = s_meta[e];
        if (m < 0) continue;
        const int ox = m & 3, oy = (m >> 2) & 3, oz = (m >> 4) & 3, gj = (m >> 6) & 1, q = m >> 8;
        const float4 wp = s_wpt[q][la], wq = s_wb[e][lk];
        const int bl = (ox + (lk & 1)) + 4 * (oy + ((lk >> 1) & 1)) + 16 * (oz + (lk >> 2));
        const double* K = s_val[e];
        // M = K W_q (3 x 6): U columns alpha_q K (displacement column), Theta columns -K [sigma_q]x or w_q K
        double MU[3][3], MT[3][3];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
          const double k0v = K[3 * i], k1v = K[3 * i + 1], k2v = K[3 * i + 2];
          if (gj == 0) {
            MU[i][0] = (double)wq.x * k0v; MU[i][1] = (double)wq.x * k1v; MU[i][2] = (double)wq.x * k2v;
            // -(K [s]x)[i][.]:  (K[s]x)[i][0] = k1 s2 - k2 s1, [i][1] = k2 s0 - k0 s2, [i][2] = k0 s1 - k1 s0
            MT[i][0] = -(k1v * (double)wq.w - k2v * (double)wq.z);
            MT[i][1] = -(k2v * (double)wq.y - k0v * (double)wq.w);
            MT[i][2] = -(k0v * (double)wq.z - k1v * (double)wq.y);
          } else {
            MU[i][0] = MU[i][1] = MU[i][2] = 0.0;
            MT[i][0] = (double)wq.x * k0v; MT[i][1] = (double)wq.x * k1v; MT[i][2] = (double)wq.x * k2v;
          }
        }
        double* dst = table + bl * 8 + la;
        if (pass == 0) {
          const double ap = (double)wp.x;
#pragma unroll
          for (int fa = 0; fa < 3; ++fa) {
#pragma unroll
            for (int fb = 0; fb < 3; ++fb) {
              if (gj == 0) __hip_atomic_fetch_add(dst + (6 * fa + fb) * 512, ap * MU[fa][fb], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
              __hip_atomic_fetch_add(dst + (6 * fa + 3 + fb) * 512, ap * MT[fa][fb], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
          }
        } else {
          // Theta rows against the Theta columns: R = [sigma_p]x MT (displacement item) or w_p MT (rotation item)
          double R[3][3];
          if (gi == 0) {
            const double s0 = (double)wp.y, s1 = (double)wp.z, s2 = (double)wp.w;
#pragma unroll
            for (int fb = 0; fb < 3; ++fb) {
              R[0][fb] = -s2 * MT[1][fb] + s1 * MT[2][fb];
              R[1][fb] = s2 * MT[0][fb] - s0 * MT[2][fb];
              R[2][fb] = -s1 * MT[0][fb] + s0 * MT[1][fb];
            }
          } else {
#pragma unroll
            for (int fa = 0; fa < 3; ++fa)
#pragma unroll
              for (int fb = 0; fb < 3; ++fb) R[fa][fb] = (double)wp.x * MT[fa][fb];
          }
#pragma unroll
          for (int fa = 0; fa < 3; ++fa)
#pragma unroll
            for (int fb = 0; fb < 3; ++fb)
              __hip_atomic_fetch_add(dst + (6 * fa + 3 + fb) * 512, R[fa][fb], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
      }
      __syncthreads();
    }
  }
  if (far && info != nullptr) atomicOr(&info[1], 1);
  __syncthreads();
  {
    const int a = t & 7, blq = t >> 3;
    const int64_t na = ell_idx[e0 + a] / 6 - off_c;
    const int row0 = pass == 0 ? 0 : 3;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int bl = blq + 32 * s;
      const int32_t nb = item_nbr[item * 64 + bl];
      if (nb < 0) continue;
      const double* src = table + bl * 8 + a;
#pragma unroll
      for (int fa = 0; fa < 3; ++fa)
#pragma unroll
        for (int f = (pass == 0 ? 0 : 3); f < 6; ++f) {
          const double v = src[(6 * fa + f) * 512];
          if (v != 0.0) atomicAdd(&A[(6 * na + row0 + fa) * lda + 6 * (int64_t)nb + f], v);
        }
    }
  }
}

// Dense Galerkin operator of the coarse-solve level on the fp64 matrix cores (round 4).  A = P_c^T (K P_c) is GEMM shaped:
// all points of an item (one coarse cell, one field group) share the cell's 8 nodes a -- the 48 rows (f, a) of their
// P_c^T -- and every column point of their blocks has its 8 nodes in the 4 x 4 x 4 node window of the cell -- the 384
// columns (f', bl) of Y = K P_c restricted to the item's rows.  Per chunk of MM_PTS points:
//   1. the chunk's 3 x 3 blocks are staged in LDS (one per thread: masked values, the column point's 8 weights, its cell
//      offset), as in k_pc_coarse_galerkin_h;
//   2. Y[(p, fa)][(f', bl)] += (K_pq W_q,k')[fa][f']: one wave per point walks its blocks, lanes = (f', k'), plain LDS
//      read-modify-writes (a wave's LDS operations execute in order, the rows of a point belong to one wave);
//   3. W^T[(f, a)][(p, fa)] = W_p,a[fa][f];
//   4. D[(f, a)][(f', bl)] += W^T Y with v_mfma_f64_16x16x4 (K = 3 MM_PTS = 24: six steps), accumulators in registers across
//      the chunks of the item: 3 x 24 tiles of 16 x 16, nine per wave.
// The tiles (Theta_1/2 rows, U columns) are not formed: k_pc_coarse_mirror_tu fills the (Theta, U) quadrant from (U, Theta).
// LDS atomics of the version before (k_pc_coarse_galerkin_h, 15 G of them at 1.97 M dofs): 18.7 ms; this kernel 6.9 ms as first
// written, 5.5 with the item's row extents loaded once and the next chunk's column indices and values prefetched under the
// MFMA phase, the operands of block e + 1 fetched before the read-modify-writes of block e, and the f' blocks of Y 66
// doubles apart.  Phase times at 248 k dofs (0.92 ms): step 2 0.49, MFMA 0.19, staging 0.12 -- step 2 is bound by the LDS
// instruction rate (per block and wave: 13 operand reads, 9 of them broadcasts of the block's values, and 6 for Y).
constexpr int MM_PTS = 8, MM_K = 3 * MM_PTS, MM_YP = 400, MM_YF = 66, MM_WP = 25, MM_STAGE = 256, MM_THREADS = 512, MM_ITEM = 256;
constexpr size_t MM_LDS = (size_t)MM_K * MM_YP * 8 + 48 * MM_WP * 8 + MM_STAGE * 9 * 8 + MM_STAGE * 8 * 16 + MM_STAGE * 4 + MM_PTS * 8 * 16 +
                          4 * (MM_ITEM + 1) + 4 * MM_ITEM * 2 + MM_ITEM + 64;
__global__ __launch_bounds__(MM_THREADS) void k_pc_coarse_galerkin_mm(int c, int width, int64_t off_c, int64_t lda, int64_t n_unode,
                                                                      const int64_t* __restrict__ item_ptr, const int32_t* __restrict__ item_pts,
                                                                      const int32_t* __restrict__ item_nbr, const int32_t* __restrict__ pcell,
                                                                      const int64_t* __restrict__ brow, const int32_t* __restrict__ bcols,
                                                                      const double* __restrict__ vals, const uint8_t* __restrict__ fixed,
                                                                      const int32_t* __restrict__ ell_idx, const float4* __restrict__ cs_w4,
                                                                      double* __restrict__ A, int32_t* __restrict__ info) {
  extern __shared__ double mm_lds[];
  double (*Y)[MM_YP] = reinterpret_cast<double (*)[MM_YP]>(mm_lds);                         // [3 pl + fa][MM_YF f' + bl]
  double (*Wt)[MM_WP] = reinterpret_cast<double (*)[MM_WP]>(mm_lds + MM_K * MM_YP);          // [8 f + a][3 pl + fa]
  double (*s_val)[9] = reinterpret_cast<double (*)[9]>(&Wt[48][0]);
  float4 (*s_wb)[8] = reinterpret_cast<float4 (*)[8]>(s_val + MM_STAGE);
  int32_t* s_meta = reinterpret_cast<int32_t*>(s_wb + MM_STAGE);
  float4 (*s_wpt)[8] = reinterpret_cast<float4 (*)[8]>(s_meta + MM_STAGE);
  int32_t* s_S = reinterpret_cast<int32_t*>(s_wpt + MM_PTS);               // blocks before point i of the item (MM_ITEM + 1 entries)
  int32_t* s_ik0 = s_S + MM_ITEM + 1;                                      // first block of point i
  int32_t* s_ipt = s_ik0 + MM_ITEM;                                        // point i of the item
  uint8_t* s_ifi = reinterpret_cast<uint8_t*>(s_ipt + MM_ITEM);            // its Dirichlet bits
  int32_t* s_wsum = reinterpret_cast<int32_t*>(s_ifi + MM_ITEM);           // scan scratch (4 waves)
  const int64_t item = blockIdx.x;
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
  const int64_t pbeg = item_ptr[item], pend = item_ptr[item + 1];
  const int np = (int)(pend - pbeg);                                       // <= MM_ITEM (femo_shell_pc_coarse checks)
  const int32_t pfirst = item_pts[pbeg];
  const int gi = pfirst >= n_unode ? 1 : 0;                                // 0: displacement points (rows U and Theta), 1: rotation points (Theta)
  const int64_t e0 = (int64_t)(3 * pfirst) * width + 8 * c;
  const int32_t pk0 = pcell[pfirst];
  const int bx = pk0 & 1023, by = (pk0 >> 10) & 1023, bz = pk0 >> 20;
  const int li = lane & 15, lk = lane >> 4;
  // the item's points, their block rows and Dirichlet bits, once: two dependent loads per ITEM instead of per chunk
  {
    int32_t len = 0;
    if (t < MM_ITEM) {
      int32_t i = 0, k0 = 0, fi = 0;
      if (t < np) {
        i = item_pts[pbeg + t];
        const int64_t b0 = brow[i];
        k0 = (int32_t)b0; len = (int32_t)(brow[i + 1] - b0);
        fi = fixed == nullptr ? 0 : (fixed[3 * i] ? 1 : 0) | (fixed[3 * i + 1] ? 2 : 0) | (fixed[3 * i + 2] ? 4 : 0);
      }
      s_ipt[t] = i; s_ik0[t] = k0; s_ifi[t] = (uint8_t)fi;
      int32_t v = len;                                                     // inclusive scan within the wave
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) { const int32_t u = __shfl_up(v, d); if (lane >= d) v += u; }
      if (lane == 63) s_wsum[wv] = v;
      len = v;
    }
    __syncthreads();
    if (t < MM_ITEM) {
      int32_t base = 0;
      for (int w = 0; w < wv; ++w) base += s_wsum[w];
      s_S[t + 1] = base + len;
      if (t == 0) s_S[0] = 0;
    }
    __syncthreads();
  }
  // accumulators: N-tiles nt = wv + 8 j (j = 0..2), M-tiles mt = 0..2
  typedef double d4 __attribute__((ext_vector_type(4)));
  d4 acc[3][3];
#pragma unroll
  for (int mt = 0; mt < 3; ++mt)
#pragma unroll
    for (int j = 0; j < 3; ++j) acc[mt][j] = d4{0.0, 0.0, 0.0, 0.0};
  int far = 0;
  // block t of a chunk: which of its points, which block of that point's row
  auto locate = [&](int i0, int npts, int f, int& q, int& lkk) {
    q = 0;
#pragma unroll
    for (int qq = 1; qq < MM_PTS; ++qq) q += (qq < npts && s_S[i0 + qq] <= f) ? 1 : 0;
    lkk = f - s_S[i0 + q];
  };
  // prefetch registers: column index and values of this thread's block of the NEXT chunk (issued before the MFMA phase)
  int32_t pf_cj = 0;
  double pf_v[9];
  bool pf_ok = false;
  auto prefetch = [&](int i0) {
    pf_ok = false;
    if (i0 >= np || t >= MM_STAGE) return;
    const int npts = min(MM_PTS, np - i0);
    const int f = s_S[i0] + t;
    if (f >= s_S[i0 + npts]) return;
    int q, lkk;
    locate(i0, npts, f, q, lkk);
    const int64_t k0 = s_ik0[i0 + q];
    const int64_t len = 3 * (int64_t)(s_S[i0 + q + 1] - s_S[i0 + q]);
    pf_cj = bcols[k0 + lkk];
    const double* v = vals + 9 * k0 + 3 * lkk;
#pragma unroll
    for (int fa = 0; fa < 3; ++fa)
#pragma unroll
      for (int fb = 0; fb < 3; ++fb) pf_v[3 * fa + fb] = v[fa * len + fb];
    pf_ok = true;
  };
  prefetch(0);
  for (int i0 = 0; i0 < np; i0 += MM_PTS) {
    const int npts = min(MM_PTS, np - i0);
    __syncthreads();                                                       // the MFMA phase of the chunk before has read Y and Wt
    for (int idx = t; idx < MM_K * MM_YP; idx += MM_THREADS) (&Y[0][0])[idx] = 0.0;
    if (t >= 64 && t < 64 + MM_PTS * 8) {
      const int q = (t - 64) >> 3, a = (t - 64) & 7;
      s_wpt[q][a] = q < npts ? cs_w4[(int64_t)s_ipt[i0 + q] * 8 + a] : float4{0.f, 0.f, 0.f, 0.f};
    }
    const int S0 = s_S[i0], B = s_S[i0 + npts] - S0;
    for (int base = 0; base < B; base += MM_STAGE) {
      const int fblk = base + t;
      if (t < MM_STAGE && fblk < B) {
        int q, lkk;
        locate(i0, npts, S0 + fblk, q, lkk);
        int32_t cj;
        double v9[9];
        if (base == 0 && pf_ok) {
          cj = pf_cj;
#pragma unroll
          for (int k = 0; k < 9; ++k) v9[k] = pf_v[k];
        } else {                                                           // rounds beyond the first (rare): not prefetched
          const int64_t k0 = s_ik0[i0 + q];
          const int64_t len = 3 * (int64_t)(s_S[i0 + q + 1] - s_S[i0 + q]);
          cj = bcols[k0 + lkk];
          const double* v = vals + 9 * k0 + 3 * lkk;
#pragma unroll
          for (int fa = 0; fa < 3; ++fa)
#pragma unroll
            for (int fb = 0; fb < 3; ++fb) v9[3 * fa + fb] = v[fa * len + fb];
        }
        const int32_t pk = pcell[cj / 3];
        const int ox = (pk & 1023) - bx + 1, oy = ((pk >> 10) & 1023) - by + 1, oz = (pk >> 20) - bz + 1;
        if ((unsigned)ox > 2u || (unsigned)oy > 2u || (unsigned)oz > 2u) {
          far = 1;                                                         // flagged; contributes nothing here
          s_meta[t] = 1 | (1 << 2) | (1 << 4);
#pragma unroll
          for (int k = 0; k < 9; ++k) s_val[t][k] = 0.0;
#pragma unroll
          for (int b = 0; b < 8; ++b) s_wb[t][b] = float4{0.f, 0.f, 0.f, 0.f};
        } else {
          s_meta[t] = ox | (oy << 2) | (oz << 4) | ((cj >= 3 * n_unode ? 1 : 0) << 6) | (q << 8);
          const int fi = s_ifi[i0 + q];
          const int fj = fixed == nullptr ? 0 : (fixed[cj] ? 1 : 0) | (fixed[cj + 1] ? 2 : 0) | (fixed[cj + 2] ? 4 : 0);
          const float4* wj = cs_w4 + (int64_t)(cj / 3) * 8;
          float4 w8[8];
#pragma unroll
          for (int b = 0; b < 8; ++b) w8[b] = wj[b];
#pragma unroll
          for (int fa = 0; fa < 3; ++fa)
#pragma unroll
            for (int fb = 0; fb < 3; ++fb) s_val[t][3 * fa + fb] = ((fi >> fa) & 1) || ((fj >> fb) & 1) ? 0.0 : v9[3 * fa + fb];
#pragma unroll
          for (int b = 0; b < 8; ++b) s_wb[t][b] = w8[b];
        }
      }
      __syncthreads();
      // Y of point pl = wv: its staged blocks are the slots [S[i0 + wv], S[i0 + wv + 1]) - S0 of this round.  Software
      // pipelined: the meta words of up to 64 blocks sit in the lanes (readlane, no LDS round trip), the weight and the nine
      // values of block e + 1 are loaded before the three read-modify-writes of block e -- per block one LDS latency, not three
      if (wv < npts) {
        const int pl = wv;
        const int e_lo = max(s_S[i0 + pl] - S0, base) - base, e_hi = min(s_S[i0 + pl + 1] - S0, base + MM_STAGE) - base;
        const int kq = lane & 7, fc = lane >> 3;
        // column f' = fc of W_q,k' as a 3-vector u = (sg[k] w[ix[k]])_k, w = (alpha, sigma): displacement column fc < 3: alpha e_fc;
        // 3: (0, -s2, s1); 4: (s2, 0, -s0); 5: (-s1, s0, 0); rotation column: w e_(fc - 3)
        int ix0 = 0, ix1 = 0, ix2 = 0;
        double sg0 = 0.0, sg1 = 0.0, sg2 = 0.0, sr0 = 0.0, sr1 = 0.0, sr2 = 0.0;
        if (fc == 0) sg0 = 1.0; else if (fc == 1) sg1 = 1.0; else if (fc == 2) sg2 = 1.0;
        else if (fc == 3) { ix1 = 3; sg1 = -1.0; ix2 = 2; sg2 = 1.0; sr0 = 1.0; }
        else if (fc == 4) { ix0 = 3; sg0 = 1.0; ix2 = 1; sg2 = -1.0; sr1 = 1.0; }
        else if (fc == 5) { ix0 = 2; sg0 = -1.0; ix1 = 1; sg1 = 1.0; sr2 = 1.0; }
        const int fcl = fc < 6 ? fc : 0;
        double* y0 = &Y[3 * pl][MM_YF * fcl];      // f' blocks 66 apart: with 64 the six f' lanes of a corner share their LDS banks
        for (int eb = e_lo; eb < e_hi; eb += 64) {
          const int cnt = min(64, e_hi - eb);
          const int32_t my_m = lane < cnt ? s_meta[eb + lane] : 0;
          const float* wf = reinterpret_cast<const float*>(&s_wb[eb][kq]);
          float w0 = wf[ix0], w1 = wf[ix1], w2 = wf[ix2], wa = wf[0];
          double K[9];
#pragma unroll
          for (int k = 0; k < 9; ++k) K[k] = s_val[eb][k];
          for (int e = 0; e < cnt; ++e) {
            const int32_t m = __builtin_amdgcn_readlane(my_m, e);
            const bool gj = (m >> 6) & 1;
            const double u0 = gj ? sr0 * (double)wa : sg0 * (double)w0, u1 = gj ? sr1 * (double)wa : sg1 * (double)w1,
                         u2 = gj ? sr2 * (double)wa : sg2 * (double)w2;
            const double c0 = K[0] * u0 + K[1] * u1 + K[2] * u2, c1 = K[3] * u0 + K[4] * u1 + K[5] * u2, c2 = K[6] * u0 + K[7] * u1 + K[8] * u2;
            if (e + 1 < cnt) {                                             // block e + 1's operands, before this block's read-modify-writes
              const float* wn = reinterpret_cast<const float*>(&s_wb[eb + e + 1][kq]);
              w0 = wn[ix0]; w1 = wn[ix1]; w2 = wn[ix2]; wa = wn[0];
#pragma unroll
              for (int k = 0; k < 9; ++k) K[k] = s_val[eb + e + 1][k];
            }
            const int ox = m & 3, oy = (m >> 2) & 3, oz = (m >> 4) & 3;
            const int bl = (ox + (kq & 1)) + 4 * (oy + ((kq >> 1) & 1)) + 16 * (oz + (kq >> 2));
            if (fc < 6) { y0[bl] += c0; y0[MM_YP + bl] += c1; y0[2 * MM_YP + bl] += c2; }
          }
        }
      }
      __syncthreads();
    }
    // W^T of the chunk: entry (m = 8 f + a, k = 3 pl + fa) = W_p,a[fa][f]
    for (int idx = t; idx < 48 * MM_K; idx += MM_THREADS) {
      const int m = idx / MM_K, k = idx - m * MM_K;
      const int f = m >> 3, a = m & 7, pl = k / 3, fa = k - 3 * pl;
      const float4 w = s_wpt[pl][a];
      double v = 0.0;
      if (gi == 0) {
        if (f < 3) v = f == fa ? (double)w.x : 0.0;
        else {
          // (-[sigma]x)[fa][j]: rows (0, s2, -s1), (-s2, 0, s0), (s1, -s0, 0)
          const int j = f - 3;
          const double s0 = (double)w.y, s1 = (double)w.z, s2 = (double)w.w;
          if (fa == 0) v = j == 1 ? s2 : (j == 2 ? -s1 : 0.0);
          else if (fa == 1) v = j == 0 ? -s2 : (j == 2 ? s0 : 0.0);
          else v = j == 0 ? s1 : (j == 1 ? -s0 : 0.0);
        }
      } else if (f >= 3) {
        v = (f - 3) == fa ? (double)w.x : 0.0;
      }
      Wt[m][k] = v;
    }
    prefetch(i0 + MM_PTS);                                                 // in flight during the MFMA phase
    __syncthreads();
    // D += W^T Y
#pragma unroll
    for (int ks = 0; ks < MM_K / 4; ++ks) {
      const int kk = 4 * ks + lk;
      double av[3];
#pragma unroll
      for (int mt = 0; mt < 3; ++mt) av[mt] = Wt[16 * mt + li][kk];
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const int nt = wv + 8 * j;
        const double bv = Y[kk][MM_YF * (nt >> 2) + 16 * (nt & 3) + li];
        if (gi == 0) acc[0][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[0], bv, acc[0][j], 0, 0, 0);
        acc[1][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[1], bv, acc[1][j], 0, 0, 0);
        if (nt >= 12) acc[2][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[2], bv, acc[2][j], 0, 0, 0);
      }
    }
  }
  if (far && info != nullptr) atomicOr(&info[1], 1);
  // flush: lane holds D[16 mt + lk + 4 i][16 nt + li]
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const int nt = wv + 8 * j;
    const int n = 16 * nt + li, fp = n >> 6, bl = n & 63;
    const int32_t nb = item_nbr[item * 64 + bl];
    if (nb < 0) continue;
#pragma unroll
    for (int mt = 0; mt < 3; ++mt) {
      if (mt == 2 && nt < 12) continue;
      if (mt == 0 && gi == 1) continue;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int m = 16 * mt + lk + 4 * i, f = m >> 3, a = m & 7;
        const double v = acc[mt][j][i];
        if (v == 0.0) continue;
        const int64_t na = ell_idx[e0 + a] / 6 - off_c;
        atomicAdd(&A[(6 * na + f) * lda + 6 * (int64_t)nb + fp], v);
      }
    }
  }
}

// the (Theta, U) quadrant of every node pair = transpose of (U, Theta): A[6 a + 3 + i, 6 b + j] = A[6 b + j, 6 a + 3 + i]
__global__ void k_pc_coarse_mirror_tu(int64_t n, int64_t lda, double* __restrict__ A) {
  const int64_t r = blockIdx.y;
  if (r >= n || (r % 6) < 3) return;
  for (int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; c < n; c += (int64_t)gridDim.x * blockDim.x)
    if ((c % 6) < 3) A[r * lda + c] = A[c * lda + r];
}

// packed lattice coordinates of the level-c cell of every point (its first ELL node is the cell's corner)
__global__ void k_pc_coarse_cells(int64_t n_pts, int c, int width, int64_t off_c, const int32_t* __restrict__ ell_idx,
                                  const int32_t* __restrict__ node_xyz, int32_t* __restrict__ pcell) {
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n_pts) return;
  const int32_t nd = ell_idx[(3 * p) * width + 8 * c] / 6 - (int32_t)off_c;
  pcell[p] = node_xyz[3 * nd] | (node_xyz[3 * nd + 1] << 10) | (node_xyz[3 * nd + 2] << 20);
}

// ---- dense factorisation of the coarse operator (N = n padded to a multiple of 64, row-major, leading dimension N) ----
// Written here rather than taken from rocSOLVER: loading that library costs 80-450 s on a fresh machine (a 0.9 GB
// shared object read from a cold disk), for a 3000 x 3000 matrix whose factorisation takes milliseconds.
// Blocked right-looking Cholesky A = L L^T with 64 x 64 tiles (lower triangle), then W = L^-1 row of tiles by row of
// tiles, stored transposed in the upper triangle (= L^-T, which the second half of the apply reads along rows) and
// mirrored into the lower one at the end.  A^-1 = L^-T L^-1 is applied in this product form, never formed: W^T W is
// positive definite whatever the rounding in W.
constexpr int DT = 64;                                   // tile edge
constexpr int DP = DT + 1;                               // LDS row pitch

// unknowns no free dof touches have an empty row and column, and so have the padding rows: unit diagonal (their
// restricted residual is zero); a relative 1e-13 on the others keeps the factorisation away from round-off rank deficiency
__global__ void k_pc_coarse_fix_diag(int64_t N, double* __restrict__ A, double ridge) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= N) return;
  const double d = A[r * N + r];
  A[r * N + r] = d > 0.0 ? d * (1.0 + ridge) : 1.0;
}

__device__ __forceinline__ void tile_load(double (*s)[DP], const double* __restrict__ g, int64_t ld, bool transpose) {
  for (int idx = threadIdx.x; idx < DT * DT; idx += 256) {
    const int r = idx / DT, c = idx % DT;
    const double v = g[(int64_t)r * ld + c];
    if (transpose) s[c][r] = v; else s[r][c] = v;
  }
}

// C += A B^T for 64 x 64 tiles with both operands in LDS as [row][k], on the fp64 matrix cores: wave w owns the
// 32 x 32 quadrant (w >> 1, w & 1) as 2 x 2 tiles of v_mfma_f64_16x16x4_f64; per k-step of 4 a lane supplies
// A[l & 15][l >> 4] and B[l >> 4][l & 15] and holds D[(l >> 4) + 4 i][l & 15] in register i (the f64 map, not the
// f32 one).  16 k-steps x 4 MFMAs per wave and tile product (the scalar version: 1024 FMAs per thread, ~4 us).
typedef double v4d __attribute__((ext_vector_type(4)));
struct TileAcc {
  v4d v[2][2];
  __device__ TileAcc() {
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) v[a][b] = v4d{0.0, 0.0, 0.0, 0.0};
  }
};

__device__ __forceinline__ void tile_mma(TileAcc& acc, const double (*sa)[DP], const double (*sb)[DP]) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int r0 = 32 * (w >> 1), c0 = 32 * (w & 1), li = lane & 15, lk = lane >> 4;
#pragma unroll 4
  for (int k0 = 0; k0 < DT; k0 += 4) {
    const double a0 = sa[r0 + li][k0 + lk], a1 = sa[r0 + 16 + li][k0 + lk];
    const double b0 = sb[c0 + li][k0 + lk], b1 = sb[c0 + 16 + li][k0 + lk];
    acc.v[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc.v[0][0], 0, 0, 0);
    acc.v[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, acc.v[0][1], 0, 0, 0);
    acc.v[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, acc.v[1][0], 0, 0, 0);
    acc.v[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc.v[1][1], 0, 0, 0);
  }
}

// f(row, col, value) for the 16 elements of the tile this lane holds
template <class F>
__device__ __forceinline__ void tile_foreach(const TileAcc& acc, F f) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int r0 = 32 * (w >> 1) + (lane >> 4), c0 = 32 * (w & 1) + (lane & 15);
#pragma unroll
  for (int sr = 0; sr < 2; ++sr)
#pragma unroll
    for (int sc = 0; sc < 2; ++sc)
#pragma unroll
      for (int i = 0; i < 4; ++i) f(r0 + 16 * sr + 4 * i, c0 + 16 * sc, acc.v[sr][sc][i]);
}

// diagonal tile kb: unblocked Cholesky in LDS, L_kk written back (lower part), its inverse (lower triangular, full
// 64 x 64 with zeros above) into dinv[kb]; info[2] = 1 if a pivot is not positive
// the diagonal tile in LDS `a` (all threads have passed a barrier after filling it): Cholesky factor in place, its inverse in
// `w`; L_kk goes to g (lower part), the inverse to dinv_k.  256 threads, ends without a barrier.
__device__ __forceinline__ void chol_diag_tile(double (*a)[DP], double (*w)[DP], double* __restrict__ g, int64_t N, double* __restrict__ dinv_k,
                                               int32_t* __restrict__ info) {
  // One wave factorises the tile column by column, lane r holding row r in registers (static indices: both loops
  // unrolled); the finished entries live in LDS as well, where the other lanes read row c as broadcasts:
  //   L[r][c] = (A[r][c] - sum_{k < c} L[r][k] L[c][k]) / L[c][c].
  // LDS operations of one wave execute in program order; the fences pin the compiler.  (History of this tile: three
  // workgroup barriers per column and the inverse through LDS 169 us, 8 of the factorisation's 10 ms; a wave working on
  // the LDS copy in place was slower (dependent read-modify-writes); a right-looking register version spilled 2 k
  // registers.)
#define FEMO_WAVE_SYNC() __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier()
  if (threadIdx.x < DT) {
    const int r = threadIdx.x;
    double row[DT];
#pragma unroll
    for (int c = 0; c < DT; ++c) row[c] = a[r][c];
    bool bad = false;
#pragma unroll
    for (int c = 0; c < DT; ++c) {
      double sum = 0.0;
#pragma unroll
      for (int k = 0; k < c; ++k) sum += row[k] * a[c][k];
      const double v = row[c] - sum;
      if (r == c) {
        if (!(v > 0.0)) bad = true;
        row[c] = sqrt(v > 0.0 ? v : 1.0);
        a[c][c] = row[c];
      }
      FEMO_WAVE_SYNC();
      if (r > c) {
        row[c] = v / a[c][c];
        a[r][c] = row[c];
      }
      FEMO_WAVE_SYNC();
    }
    if (bad) atomicOr(&info[2], 1);
  }
#undef FEMO_WAVE_SYNC
  __syncthreads();
  // inverse: thread c solves L x = e_c with its column in registers; the loops are uniform (entries above the
  // diagonal of the column are zeros), so every read of L is an LDS broadcast and the unrolled code has static
  // register indices
  if (threadIdx.x < DT) {
    const int c = threadIdx.x;
    double x[DT];
#pragma unroll
    for (int r = 0; r < DT; ++r) {
      double v = r == c ? 1.0 : 0.0;
#pragma unroll
      for (int k = 0; k < r; ++k) v -= a[r][k] * x[k];
      x[r] = r >= c ? v / a[r][r] : 0.0;
    }
#pragma unroll
    for (int r = 0; r < DT; ++r) w[r][c] = x[r];
  }
  __syncthreads();
  for (int idx = threadIdx.x; idx < DT * DT; idx += 256) {
    const int r = idx / DT, c = idx % DT;
    if (r >= c) g[(int64_t)r * N + c] = a[r][c];
    dinv_k[idx] = w[r][c];
  }
}

// diagonal tile kb: unblocked Cholesky in LDS, L_kk written back (lower part), its inverse (lower triangular, full
// 64 x 64 with zeros above) into dinv[kb]; info[2] = 1 if a pivot is not positive.  Launched for kb = 0 only: the later
// diagonal tiles are factorised by the workgroup of k_chol_update that completes them.
__global__ __launch_bounds__(256) void k_chol_diag(int64_t N, int kb, double* __restrict__ A, double* __restrict__ dinv, int32_t* __restrict__ info) {
  __shared__ double a[DT][DP];
  __shared__ double w[DT][DP];
  double* g = A + ((int64_t)kb * DT) * N + (int64_t)kb * DT;
  tile_load(a, g, N, false);
  __syncthreads();
  chol_diag_tile(a, w, g, N, dinv + (int64_t)kb * DT * DT, info);
}

// panel below the diagonal tile: L_ik = A_ik L_kk^-T = A_ik (dinv_k)^T, block rows i = kb + 1 + blockIdx.x
__global__ __launch_bounds__(256) void k_chol_panel(int64_t N, int kb, double* __restrict__ A, const double* __restrict__ dinv) {
  __shared__ double sa[DT][DP];
  __shared__ double sb[DT][DP];
  const int i = kb + 1 + blockIdx.x;
  double* g = A + ((int64_t)i * DT) * N + (int64_t)kb * DT;
  tile_load(sa, g, N, false);
  tile_load(sb, dinv + (int64_t)kb * DT * DT, DT, false);
  __syncthreads();
  TileAcc acc;
  tile_mma(acc, sa, sb);
  tile_foreach(acc, [&](int r, int c, double v) { g[(int64_t)r * N + c] = v; });
}

// trailing update: A_ij -= L_ik L_jk^T for kb < j <= i; blockIdx.x enumerates the pairs (i, j) of the trailing triangle
__global__ __launch_bounds__(256) void k_chol_update(int64_t N, int kb, double* __restrict__ A, double* __restrict__ dinv, int32_t* __restrict__ info) {
  __shared__ double sa[DT][DP];
  __shared__ double sb[DT][DP];
  // pair index -> (ii >= jj) in the triangle of edge m = nblk - kb - 1
  int ii = (int)((sqrt(8.0 * blockIdx.x + 1.0) - 1.0) * 0.5);
  while ((ii + 1) * (ii + 2) / 2 <= (int)blockIdx.x) ++ii;
  while (ii * (ii + 1) / 2 > (int)blockIdx.x) --ii;
  const int jj = blockIdx.x - ii * (ii + 1) / 2;
  const int i = kb + 1 + ii, j = kb + 1 + jj;
  tile_load(sa, A + ((int64_t)i * DT) * N + (int64_t)kb * DT, N, false);
  tile_load(sb, A + ((int64_t)j * DT) * N + (int64_t)kb * DT, N, false);
  __syncthreads();
  TileAcc acc;
  tile_mma(acc, sa, sb);
  double* g = A + ((int64_t)i * DT) * N + (int64_t)j * DT;
  if (blockIdx.x == 0) {
    // tile (kb + 1, kb + 1) is complete with this update: factorise it here, while the other workgroups update the rest of the
    // trailing matrix -- the next step then starts with its panel (the separate diagonal launch was 51 of a step's 92 us)
    __syncthreads();                                       // the products have read sa / sb
    tile_foreach(acc, [&](int r, int c, double v) { sa[r][c] = g[(int64_t)r * N + c] - v; });
    __syncthreads();
    chol_diag_tile(sa, sb, g, N, dinv + (int64_t)(kb + 1) * DT * DT, info);
    return;
  }
  tile_foreach(acc, [&](int r, int c, double v) { g[(int64_t)r * N + c] -= v; });
}

// row of tiles i of W = L^-1: W_ij = -dinv_i sum_{k = j}^{i - 1} L_ik W_kj for j = blockIdx.x < i, with W_jj = dinv_j and
// W_kj (k > j) read from where the earlier rows put it: transposed, in tile (j, k) of the upper triangle; W_ij goes to
// tile (j, i) the same way
__global__ __launch_bounds__(256) void k_trinv_row(int64_t N, int i, double* __restrict__ A, const double* __restrict__ dinv) {
  __shared__ double sa[DT][DP];
  __shared__ double sb[DT][DP];
  const int j = blockIdx.x;
  TileAcc acc;
  // the two tiles of step k + 1 travel from global memory to registers while step k multiplies
  double ra[16], rb[16];
  auto fetch = [&](int k) {
    const double* ga = A + ((int64_t)i * DT) * N + (int64_t)k * DT;                              // L_ik [r][m]
    const double* gb = k == j ? dinv + (int64_t)j * DT * DT : A + ((int64_t)j * DT) * N + (int64_t)k * DT;
    const int64_t ldb = k == j ? DT : N;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int idx = threadIdx.x + 256 * q, r = idx / DT, c = idx % DT;
      ra[q] = ga[(int64_t)r * N + c];
      rb[q] = gb[(int64_t)r * ldb + c];
    }
  };
  fetch(j);
  for (int k = j; k < i; ++k) {
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int idx = threadIdx.x + 256 * q, r = idx / DT, c = idx % DT;
      sa[r][c] = ra[q];
      if (k == j) sb[c][r] = rb[q];                        // sb[c][m] = W_jj[m][c]: the diagonal tile is stored untransposed
      else sb[r][c] = rb[q];                               // tile (j, k)[c][m] = W_kj[m][c]
    }
    __syncthreads();
    if (k + 1 < i) fetch(k + 1);
    tile_mma(acc, sa, sb);
  }
  __syncthreads();
  // S (in acc) transposed into sb: sb[c][m] = S[m][c]; then W_ij[r][c] = -sum_m dinv_i[r][m] S[m][c]
  tile_foreach(acc, [&](int r, int c, double v) { sb[c][r] = v; });
  tile_load(sa, dinv + (int64_t)i * DT * DT, DT, false);
  __syncthreads();
  TileAcc out;
  tile_mma(out, sa, sb);
  double* g = A + ((int64_t)j * DT) * N + (int64_t)i * DT;                                       // tile (j, i), transposed store
  tile_foreach(out, [&](int r, int c, double v) { g[(int64_t)c * N + r] = -v; });
}

// W = L^-1 by recursive doubling (round 4): with the diagonal tiles inverted (dinv), level s joins pairs of inverted diagonal
// blocks of 2^s tiles, [W_CC 0; W_RC W_RR] with W_RC = -W_RR (L_RC W_CC) -- two launches of independent tile products per
// level, 2 x 6 launches for 48 tiles, where the row-by-row version (k_trinv_row) is 47 dependent launches whose last ones
// loop over 47 tile products per workgroup (4.5 ms at n = 3060).  phase 0: T_ij = sum_{k = j .. c1 - 1} L_ik W_kj, stored
// TRANSPOSED in T (what phase 1 reads as its B operand); phase 1: W_ij = -sum_{k = r0 .. i} W_ik T_kj, stored transposed in
// the upper triangle of A like every finished tile of W.  Operands: L below the diagonal of A, finished W tiles (k, j), k > j,
// at tile (j, k) of A transposed, diagonal ones in dinv.
__global__ __launch_bounds__(256) void k_trinv_level(int64_t N, int nt, int B, int phase, double* __restrict__ A, const double* __restrict__ dinv,
                                                     double* __restrict__ T) {
  __shared__ double sa[DT][DP];
  __shared__ double sb[DT][DP];
  const int per = B * B;
  const int b = blockIdx.x / per, ij = blockIdx.x % per;
  const int c0 = 2 * b * B, r0 = c0 + B;
  const int i = r0 + ij / B, j = c0 + ij % B;
  if (i >= nt) return;
  const int k_lo = phase == 0 ? j : r0, k_hi = phase == 0 ? r0 : i + 1;          // [k_lo, k_hi)
  TileAcc acc;
  double ra[16], rb[16];
  // operand tiles of step k as [row][k] arrays: phase 0: a = L_ik (tile (i, k), direct), b^T = W_kj: tile (j, k) direct, or dinv_j transposed;
  // phase 1: a = W_ik: tile (k, i) transposed, or dinv_i direct; b^T = T_kj^T: tile (k, j) of T, direct
  auto fetch = [&](int k) {
    const double *ga, *gb;
    int64_t lda, ldb;
    if (phase == 0) {
      ga = A + ((int64_t)i * DT) * N + (int64_t)k * DT; lda = N;
      if (k == j) { gb = dinv + (int64_t)j * DT * DT; ldb = DT; } else { gb = A + ((int64_t)j * DT) * N + (int64_t)k * DT; ldb = N; }
    } else {
      if (k == i) { ga = dinv + (int64_t)i * DT * DT; lda = DT; } else { ga = A + ((int64_t)k * DT) * N + (int64_t)i * DT; lda = N; }
      gb = T + ((int64_t)k * DT) * N + (int64_t)j * DT; ldb = N;
    }
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int idx = threadIdx.x + 256 * q, r = idx / DT, c = idx % DT;
      ra[q] = ga[(int64_t)r * lda + c];
      rb[q] = gb[(int64_t)r * ldb + c];
    }
  };
  fetch(k_lo);
  for (int k = k_lo; k < k_hi; ++k) {
    __syncthreads();
    const bool ta = phase == 1 && k != i;                  // a arrives transposed
    const bool tb = phase == 0 && k == j;                  // b^T arrives transposed
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int idx = threadIdx.x + 256 * q, r = idx / DT, c = idx % DT;
      if (ta) sa[c][r] = ra[q]; else sa[r][c] = ra[q];
      if (tb) sb[c][r] = rb[q]; else sb[r][c] = rb[q];
    }
    __syncthreads();
    if (k + 1 < k_hi) fetch(k + 1);
    tile_mma(acc, sa, sb);
  }
  if (phase == 0) {
    double* g = T + ((int64_t)i * DT) * N + (int64_t)j * DT;                                     // T_ij^T: element (c, r)
    tile_foreach(acc, [&](int r, int c, double v) { g[(int64_t)c * N + r] = v; });
  } else {
    double* g = A + ((int64_t)j * DT) * N + (int64_t)i * DT;                                     // tile (j, i) = W_ij^T
    tile_foreach(acc, [&](int r, int c, double v) { g[(int64_t)c * N + r] = -v; });
  }
}

// diagonal tiles of the result: L_kk^-T above, L_kk^-1 below the diagonal
__global__ __launch_bounds__(256) void k_trinv_diag(int64_t N, double* __restrict__ A, const double* __restrict__ dinv) {
  const int kb = blockIdx.x;
  double* g = A + ((int64_t)kb * DT) * N + (int64_t)kb * DT;
  const double* d = dinv + (int64_t)kb * DT * DT;
  for (int idx = threadIdx.x; idx < DT * DT; idx += 256) {
    const int r = idx / DT, c = idx % DT;
    g[(int64_t)r * N + c] = r >= c ? d[r * DT + c] : d[c * DT + r];
  }
}

// off-diagonal part: the upper triangle holds L^-T; copy it transposed into the lower one (over L)
__global__ void k_pc_coarse_mirror(int64_t N, double* __restrict__ A) {
  const int64_t r = blockIdx.y, cidx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (cidx < N && cidx / DT > r / DT) A[cidx * N + r] = A[r * N + cidx];
}

// the factors of the coarse inverse as the iteration reads them: single precision (round 3).  M_c^-1 = B^T B with B = fl32(L^-1)
// is symmetric positive semi-definite whatever the rounding did, a relative 6e-8 away from the fp64 one -- nothing an
// iteration count sees --, and the two triangular products per iteration stream half the bytes (products and sums in fp64).
__global__ void k_pc_coarse_to_float(int64_t count, const double* __restrict__ A, float* __restrict__ Af, double scale) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (int64_t)gridDim.x * blockDim.x) Af[i] = (float)(scale * A[i]);
}

// one workgroup per row of the triangular factor: lower = 1: y = L^-1 x (entries 0 .. r of row r); 0: y = L^-T x (r .. n).
// (A wave per row took 25 us per pass at n = 3060: the long rows are 24 dependent rounds of loads for one wave.)
// The solver's x += alpha p carried by extra workgroups of the first coarse product (round 3, see FemoXUpdate in bpx.hip for
// the Poisson twin): the lattice kernels between the restriction and the prolongation are launch- and latency-bound, the
// update depends on nothing they compute, and it must only be done before the prolongation overwrites p.
struct ShellXCarry {
  double* x;
  const double* p;
  const double* alpha;
  int64_t n;
  int row_blocks;        // workgroups [0, row_blocks) do the product, the rest carry
};

template <class T>
__global__ __launch_bounds__(SH_BLOCK) void k_pc_coarse_apply(int64_t n, int64_t N, int lower, const T* __restrict__ W, const double* __restrict__ x,
                                                              double* __restrict__ y, const int32_t* __restrict__ done, ShellXCarry xc = {nullptr, nullptr, nullptr, 0, 0}) {
  if (done != nullptr && *done) return;
  if (xc.x != nullptr && (int)blockIdx.x >= xc.row_blocks) {
    const double alpha = *xc.alpha;
    const int64_t stride = (int64_t)(gridDim.x - xc.row_blocks) * SH_BLOCK;
    for (int64_t i = (int64_t)(blockIdx.x - xc.row_blocks) * SH_BLOCK + threadIdx.x; i < xc.n; i += stride) xc.x[i] += alpha * xc.p[i];
    return;
  }
  __shared__ double lds[SH_BLOCK / 64];
  // rows are paired long with short (r and n - 1 - r take n + 1 entries together): even work per workgroup
  const int64_t pair = blockIdx.x;
  for (int h = 0; h < 2; ++h) {
    const int64_t r = h == 0 ? pair : n - 1 - pair;
    if (h == 1 && r <= pair) break;
    const T* row = W + r * N;
    const int64_t k0 = lower ? 0 : r, k1 = lower ? r + 1 : n;
    double s = 0.0;
    for (int64_t k = k0 + threadIdx.x; k < k1; k += SH_BLOCK) s += (double)row[k] * x[k];
    __syncthreads();
    const double t = femo_block_sum<SH_BLOCK>(s, lds);
    if (threadIdx.x == 0) y[r] = t;
  }
}

__global__ void k_pc_invert(int64_t n, double* __restrict__ d) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    d[i] = d[i] > 0.0 ? 1.0 / d[i] : 0.0;
}

// The coarse end of the hierarchy in one single-workgroup launch: levels kc .. 0 down, then 0 .. kc up (a few
// thousand nodes; as separate launches each costs its ~5 us launch floor).  The levels communicate through global
// memory inside one workgroup: __syncthreads() drains the stores before the barrier.
struct LatLevels { int kc; int64_t off[18]; };
__global__ __launch_bounds__(1024) void k_lat_coarse_fused(LatLevels Lv, const int64_t* __restrict__ chi_rowptr, const int32_t* __restrict__ chi_cols,
                                                           const double* __restrict__ chi_vals, const int64_t* __restrict__ par_rowptr,
                                                           const int32_t* __restrict__ par_cols, const double* __restrict__ par_vals,
                                                           const double* __restrict__ coarse, double* g, double* e, const int32_t* __restrict__ done) {
  if (done != nullptr && *done) return;
  for (int l = Lv.kc; l >= 0; --l) {
    const int64_t n0 = Lv.off[l], cnt = (Lv.off[l + 1] - n0) * 6;
    for (int64_t t = threadIdx.x; t < cnt; t += 1024) {
      const int64_t node = n0 + t / 6;
      const int f = (int)(t % 6);
      double s = 0.0;
      for (int64_t k = chi_rowptr[node]; k < chi_rowptr[node + 1]; ++k) s += chi_vals[k] * g[6 * (int64_t)chi_cols[k] + f];
      g[6 * node + f] = s;
    }
    __syncthreads();
  }
  for (int l = 0; l <= Lv.kc; ++l) {
    const int64_t n0 = Lv.off[l], cnt = (Lv.off[l + 1] - n0) * 6;
    for (int64_t t = threadIdx.x; t < cnt; t += 1024) {
      const int64_t node = n0 + t / 6;
      const int f = (int)(t % 6);
      const int64_t u = 6 * node + f;
      double s = 0.0;
      for (int64_t k = par_rowptr[node]; k < par_rowptr[node + 1]; ++k) s += par_vals[k] * e[6 * (int64_t)par_cols[k] + f];
      e[u] = coarse[u] * g[u] + s;
    }
    __syncthreads();
  }
}

// g = P_L^T r on the finest lattice's nodes [m0, m1): one row per (node, field group) listing the points (first dof
// 3 p) that touch the node and their weights, SUB lanes per row, three components at a time
template <int SUB>
__global__ __launch_bounds__(SH_BLOCK) void k_pc_restrict(int64_t m0, int64_t m1, const int64_t* __restrict__ rowptr, const int32_t* __restrict__ cols,
                                                          const float* __restrict__ vals, const double* __restrict__ r, double* __restrict__ g,
                                                          const int32_t* __restrict__ done) {
  if (done != nullptr && *done) return;
  const int sl = threadIdx.x & (SUB - 1);
  const int64_t nsub = (int64_t)gridDim.x * (SH_BLOCK / SUB);
  const int64_t nrow = 2 * (m1 - m0);
  for (int64_t row = (int64_t)blockIdx.x * (SH_BLOCK / SUB) + (threadIdx.x / SUB); row < nrow; row += nsub) {
    double s0 = 0.0, s1 = 0.0, s2 = 0.0;
    const int64_t e1 = rowptr[row + 1];
    for (int64_t e = rowptr[row] + sl; e < e1; e += SUB) {
      const int32_t c = cols[e];
      const double w = (double)vals[e];
      const Triple rc = *reinterpret_cast<const Triple*>(r + c);
      s0 += w * rc.a;
      s1 += w * rc.b;
      s2 += w * rc.c;
    }
#pragma unroll
    for (int off = SUB / 2; off > 0; off >>= 1) {
      s0 += __shfl_xor(s0, off, 64);
      s1 += __shfl_xor(s1, off, 64);
      s2 += __shfl_xor(s2, off, 64);
    }
    if (sl < 3) g[6 * (m0 + (row >> 1)) + 3 * (row & 1) + sl] = sl == 0 ? s0 : (sl == 1 ? s1 : s2);
  }
}

// one lattice level, nodes [n0, n1), six fields per node (unknown 6 node + f), one thread per unknown:
//   down: g[u] = sum over the node's children c of w g[6 c + f]                       (T^T, <= 27 children)
//   up:   e[u] = C[u] g[u] + sum over the node's parents p of w e[6 p + f]           (T, <= 8 parents; none on level 0)
//   blocks != nullptr (up only): C is the node's 6 x 6 block (row f of blocks[36 node ..]) instead of the diagonal
__global__ __launch_bounds__(256) void k_lat_level(int64_t n0, int64_t n1, const int64_t* __restrict__ rowptr, const int32_t* __restrict__ cols,
                            const double* __restrict__ vals, const double* __restrict__ coarse, double* __restrict__ g,
                            double* __restrict__ e, int up, const int32_t* __restrict__ done, const double* __restrict__ blocks = nullptr,
                            double* __restrict__ dot_partials = nullptr) {
  if (done != nullptr && *done) return;
  // dot_partials (up, finest level only): per-block partial of e . g over the level -- with r . (B r) from the update
  // kernel it gives r . z before z exists (r . P e = (P^T r) . e), so the direction update is fused into the prolongation
  __shared__ double lds[256 / 64];
  double dot = 0.0;
  const int64_t total = (n1 - n0) * 6;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t node = n0 + t / 6;
    const int f = (int)(t % 6);
    const int64_t u = 6 * node + f;
    const double* src = up ? e : g;
    double s = 0.0;
    for (int64_t k = rowptr[node]; k < rowptr[node + 1]; ++k) s += vals[k] * src[6 * (int64_t)cols[k] + f];
    if (!up) { g[u] = s; continue; }
    if (blocks != nullptr) {
      const double* B = blocks + 36 * node + 6 * f;
      const double* gn = g + 6 * node;
#pragma unroll
      for (int q = 0; q < 6; ++q) s += B[q] * gn[q];
    } else {
      s += coarse[u] * g[u];
    }
    e[u] = s;
    dot += s * g[u];
  }
  if (dot_partials != nullptr) {
    const double tsum = femo_block_sum<256>(dot, lds);
    if (threadIdx.x == 0) dot_partials[blockIdx.x] = tsum;
  }
}

// per-point copies of the ELL entries of the levels first_slot / 8 and up (node = unknown / 6): [level][point][8]
__global__ void k_compact_levels(int64_t n_pts, int width, int first_slot, const int32_t* __restrict__ ell_idx,
                                 const double* __restrict__ ell_w, int32_t* __restrict__ node, double* __restrict__ w) {
  const int nlev = (width - first_slot) >> 3;
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n_pts * nlev * 8) return;
  const int a = (int)(t & 7);
  const int64_t p = (t >> 3) % n_pts;
  const int lv = (int)((t >> 3) / n_pts);
  const int64_t e = (3 * p) * width + first_slot + 8 * lv + a;
  node[t] = ell_idx[e] / 6;
  w[t] = ell_w[e];
}

// 6 x 6 Galerkin blocks of the lattice nodes, levels first_slot / 8 and up: blk[node][f][f'] = sum over free dofs i of
// field f and k of field f' that both touch the node of P[i, node] K[i, k] P[k, node].  One thread per (point, level,
// component fa of the point) over the node-block view of the matrix: it keeps the row's sums against the point's eight
// nodes of the level (8 x 6 registers), reads a column point's eight (node, weight) pairs of the level once per block
// and matches them against its own eight.  (A thread per (dof, slot) over the scalar rows took longer than the
// iterations the blocks save; a thread per (point, slot) re-read the column's pairs for every slot: 18 ms at 1.97 M
// dofs.)  The diagonal of a block is the scalar Galerkin diagonal.
__global__ __launch_bounds__(SH_BLOCK) void k_pc_galerkin_blocks(int64_t n_pts, int width, int64_t n_unode, const int64_t* __restrict__ brow,
                                                                 const int32_t* __restrict__ bcols, const double* __restrict__ vals,
                                                                 const uint8_t* __restrict__ fixed, const int32_t* __restrict__ lvl_node,
                                                                 const double* __restrict__ lvl_w, double* __restrict__ blk, int first_slot) {
  // A workgroup = 256 consecutive points, one level, one component (blockIdx.y = 3 level + fa): neighbouring points
  // touch the same few dozen lattice nodes, so their sums are combined in an LDS hash table (key = node and field
  // group) and leave the workgroup as one global atomic per entry -- the global atomics were three quarters of this
  // kernel's time (7.4 ms with, 1.75 without them at 988 k dofs).
  constexpr int HS = 512;
  __shared__ int32_t h_key[HS];
  __shared__ double h_val[HS][6];
  for (int i = threadIdx.x; i < HS; i += SH_BLOCK) {
    h_key[i] = -1;
#pragma unroll
    for (int q = 0; q < 6; ++q) h_val[i][q] = 0.0;
  }
  __syncthreads();
  const int fa = (int)(blockIdx.y % 3);
  const int32_t* ln = lvl_node + (int64_t)(blockIdx.y / 3) * n_pts * 8;          // this level's [point][8] tables
  const double* lw = lvl_w + (int64_t)(blockIdx.y / 3) * n_pts * 8;
  const int64_t p = (int64_t)blockIdx.x * SH_BLOCK + threadIdx.x;
  const bool active = p < n_pts && !(fixed != nullptr && fixed[3 * p + fa]);
  if (active) {
    int32_t nd[8];
    double wi[8], acc[8][6];
#pragma unroll
    for (int a = 0; a < 8; ++a) {
      nd[a] = ln[p * 8 + a];
      wi[a] = lw[p * 8 + a];
      if (wi[a] == 0.0) nd[a] = -1;
#pragma unroll
      for (int q = 0; q < 6; ++q) acc[a][q] = 0.0;
    }
    const int gi = p >= n_unode ? 1 : 0;
    const int64_t k0 = brow[p], k1 = brow[p + 1], len = 3 * (k1 - k0);
    const double* v = vals + 9 * k0 + fa * len;
    for (int64_t k = k0; k < k1; ++k) {
      const int32_t cj = bcols[k];
      const int32_t* ik = ln + (int64_t)(cj / 3) * 8;
      const double* wk = lw + (int64_t)(cj / 3) * 8;
      const int64_t o = 3 * (k - k0);
      double m0 = v[o], m1 = v[o + 1], m2 = v[o + 2];
      if (fixed != nullptr) {
        if (fixed[cj]) m0 = 0.0;
        if (fixed[cj + 1]) m1 = 0.0;
        if (fixed[cj + 2]) m2 = 0.0;
      }
      const bool gj = cj >= 3 * n_unode;
#pragma unroll
      for (int b = 0; b < 8; ++b) {
        const int32_t nb = ik[b];
        const double wb = wk[b];
#pragma unroll
        for (int a = 0; a < 8; ++a) {
          const double ww = nb == nd[a] ? wb : 0.0;
          if (gj) { acc[a][3] += ww * m0; acc[a][4] += ww * m1; acc[a][5] += ww * m2; }
          else { acc[a][0] += ww * m0; acc[a][1] += ww * m1; acc[a][2] += ww * m2; }
        }
      }
    }
    // upper triangle only (the block is symmetric; k_pc_invert_blocks mirrors it)
    const int fi = 3 * gi + fa;
#pragma unroll
    for (int a = 0; a < 8; ++a) {
      if (nd[a] < 0) continue;
      const int32_t key = 2 * nd[a] + gi;
      int h = (int)(((uint32_t)key * 2654435761u) >> 23) & (HS - 1);
      int slot = -1;
      for (int probe = 0; probe < 16; ++probe) {
        const int32_t seen = atomicCAS(&h_key[h], -1, key);
        if (seen == -1 || seen == key) { slot = h; break; }
        h = (h + 1) & (HS - 1);
      }
#pragma unroll
      for (int q = 0; q < 6; ++q) {
        if (q < fi || acc[a][q] == 0.0) continue;
        const double val = wi[a] * acc[a][q];
        if (slot >= 0) __hip_atomic_fetch_add(&h_val[slot][q], val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        else atomicAdd(&blk[36 * (int64_t)nd[a] + 6 * fi + q], val);          // table full around this key: straight to memory
      }
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < HS; i += SH_BLOCK) {
    const int32_t key = h_key[i];
    if (key < 0) continue;
    const int fi = 3 * (key & 1) + fa;
#pragma unroll
    for (int q = 0; q < 6; ++q) {
      const double val = h_val[i][q];
      if (q >= fi && val != 0.0) atomicAdd(&blk[36 * (int64_t)(key >> 1) + 6 * fi + q], val);
    }
  }
}

// inverse of the 3 x 3 diagonal block of every point (imposed dofs: unit row and column), the smoother of the finest
// level in place of 1 / diag: it sees the coupling of a node's three displacement (rotation) components
__global__ void k_pt_block_inv(int64_t n_pts, const int64_t* __restrict__ brow, const int32_t* __restrict__ bcols,
                               const double* __restrict__ vals, const uint8_t* __restrict__ fixed, float* __restrict__ dinv3,
                               double* __restrict__ dinv) {
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n_pts) return;
  const int64_t k0 = brow[p], k1 = brow[p + 1], len = 3 * (k1 - k0);
  double a[3][3] = {{1.0, 0.0, 0.0}, {0.0, 1.0, 0.0}, {0.0, 0.0, 1.0}};
  for (int64_t k = k0; k < k1; ++k)
    if (bcols[k] == 3 * p) {
      const double* v = vals + 9 * k0 + 3 * (k - k0);
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) a[i][j] = v[i * len + j];
      break;
    }
#pragma unroll
  for (int i = 0; i < 3; ++i)
    if ((fixed != nullptr && fixed[3 * p + i]) || !(a[i][i] > 0.0)) {
#pragma unroll
      for (int j = 0; j < 3; ++j) { a[i][j] = 0.0; a[j][i] = 0.0; }
      a[i][i] = 1.0;
    }
  // 1 / diag as well (what k_csr_diag_inv computes by scanning whole scalar rows: 5.8 GB fetched at 1.97 M dofs)
#pragma unroll
  for (int i = 0; i < 3; ++i) dinv[3 * p + i] = 1.0 / a[i][i];
  // symmetric 3 x 3 inverse by cofactors
  const double s01 = 0.5 * (a[0][1] + a[1][0]), s02 = 0.5 * (a[0][2] + a[2][0]), s12 = 0.5 * (a[1][2] + a[2][1]);
  const double c00 = a[1][1] * a[2][2] - s12 * s12, c01 = s02 * s12 - s01 * a[2][2], c02 = s01 * s12 - s02 * a[1][1];
  const double c11 = a[0][0] * a[2][2] - s02 * s02, c12 = s01 * s02 - a[0][0] * s12, c22 = a[0][0] * a[1][1] - s01 * s01;
  const double det = a[0][0] * c00 + s01 * c01 + s02 * c02;
  float* B = dinv3 + 9 * p;
  if (det > 0.0 && c00 > 0.0 && c22 > 0.0) {
    const double id = 1.0 / det;
    B[0] = (float)(c00 * id); B[1] = (float)(c01 * id); B[2] = (float)(c02 * id);
    B[3] = (float)(c01 * id); B[4] = (float)(c11 * id); B[5] = (float)(c12 * id);
    B[6] = (float)(c02 * id); B[7] = (float)(c12 * id); B[8] = (float)(c22 * id);
  } else {                                                 // not positive definite in floating point: plain Jacobi for this point
    B[0] = (float)(1.0 / a[0][0]); B[1] = 0.f; B[2] = 0.f; B[3] = 0.f; B[4] = (float)(1.0 / a[1][1]); B[5] = 0.f; B[6] = 0.f; B[7] = 0.f; B[8] = (float)(1.0 / a[2][2]);
  }
}

// in-place inverse of every node's 6 x 6 block (symmetric positive definite on the fields that have free dofs; a field
// without any gets a zero row and column): Gauss-Jordan without pivoting on the symmetrised block
__global__ void k_pc_invert_blocks(int64_t node0, int64_t node1, double* __restrict__ blk, double scale) {
  const int64_t node = node0 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (node >= node1) return;
  double* B = blk + 36 * node;
  double a[6][6], inv[6][6];
  bool dead[6];
#pragma unroll
  for (int r = 0; r < 6; ++r)
#pragma unroll
    for (int c = 0; c < 6; ++c) { a[r][c] = r <= c ? B[6 * r + c] : B[6 * c + r]; inv[r][c] = r == c ? 1.0 : 0.0; }    // upper triangle holds the sums
#pragma unroll
  for (int r = 0; r < 6; ++r) dead[r] = !(a[r][r] > 0.0);
#pragma unroll
  for (int r = 0; r < 6; ++r)
    if (dead[r]) {
#pragma unroll
      for (int c = 0; c < 6; ++c) { a[r][c] = 0.0; a[c][r] = 0.0; }
      a[r][r] = 1.0;
    }
#pragma unroll
  for (int r = 0; r < 6; ++r) a[r][r] *= 1.0 + 1e-12;
#pragma unroll
  for (int p = 0; p < 6; ++p) {
    const double d = 1.0 / a[p][p];
#pragma unroll
    for (int c = 0; c < 6; ++c) { a[p][c] *= d; inv[p][c] *= d; }
#pragma unroll
    for (int r = 0; r < 6; ++r) {
      if (r == p) continue;
      const double m = a[r][p];
#pragma unroll
      for (int c = 0; c < 6; ++c) { a[r][c] -= m * a[p][c]; inv[r][c] -= m * inv[p][c]; }
    }
  }
#pragma unroll
  for (int r = 0; r < 6; ++r)
#pragma unroll
    for (int c = 0; c < 6; ++c) B[6 * r + c] = (dead[r] || dead[c]) ? 0.0 : scale * 0.5 * (inv[r][c] + inv[c][r]);
}

// all levels between the coarse-solve level and the finest lattice at once: g[node] = sum over the finest lattice's
// nodes of the composite child transfer (T_l^T ... T_{F-1}^T, built on the host), one wave per node, six fields per lane
__global__ __launch_bounds__(SH_BLOCK) void k_lat_down_composite(int64_t row0, int64_t n_rows, const int64_t* __restrict__ rowptr,
                                                                 const int32_t* __restrict__ cols, const double* __restrict__ vals,
                                                                 double* __restrict__ g, const int32_t* __restrict__ done) {
  if (done != nullptr && *done) return;
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * (SH_BLOCK / 64) + (threadIdx.x >> 6);
  if (row >= n_rows) return;
  double s[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
  for (int64_t k = rowptr[row] + lane; k < rowptr[row + 1]; k += 64) {
    const double w = vals[k];
    const double* src = g + 6 * (int64_t)cols[k];
#pragma unroll
    for (int f = 0; f < 6; ++f) s[f] += w * src[f];
  }
#pragma unroll
  for (int f = 0; f < 6; ++f) s[f] = femo_wave_sum(s[f]);
  if (lane == 0) {
    double* dst = g + 6 * (row0 + row);
#pragma unroll
    for (int f = 0; f < 6; ++f) dst[f] = s[f];
  }
}


// ---- Hermite-type lattice spaces (round 4; fea/shell.py::hermite_lattice, oracle/shell_oracle.py::LatticePreconditioner) --
// The nodal rotations of a lattice are the slopes of its displacement interpolation:
//   displacement point:  u = sum_n [ alpha_n U_n + Theta_n x sigma_n ]        (w4 = (alpha, sigma), 8 nodes)
//   rotation point:      theta = sum_n w_n Theta_n                            (w4 = (w, 0, 0, 0))
//   lattice to lattice:  U_c = a U_p + Theta_p x b,   Theta_c = c Theta_p     (w5 = (a, b, c) per (child, parent))
// and the transposes  g_U[n] += alpha r,  g_Theta[n] += sigma x r  /  g_U[p] += a g_U[c],  g_Theta[p] += c g_Theta[c] + b x g_U[c].
// A bending mode (w quadratic, theta = grad w) is then reproduced by lattices far coarser than the shell is thick, where
// the trilinear spaces of rounds 2-3 lock: 253 -> ~130 iterations on the 362 x 362 roof at the same cost per iteration.
struct V3 { double x, y, z; };
__device__ __forceinline__ V3 cross3(const V3& a, const V3& b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }

// g = P_L^T r on the finest lattice's nodes [m0, m1): per node a row of displacement points (first dof 3 p, w4) and a row
// of rotation points; SUB lanes per node
template <int SUB>
__global__ __launch_bounds__(SH_BLOCK) void k_pc_restrict_h(int64_t m0, int64_t m1, const int64_t* __restrict__ rowptr, const int32_t* __restrict__ cols,
                                                            const float4* __restrict__ w4, const double* __restrict__ r, double* __restrict__ g,
                                                            const int32_t* __restrict__ done) {
  if (done != nullptr && *done) return;
  const int sl = threadIdx.x & (SUB - 1);
  const int64_t nsub = (int64_t)gridDim.x * (SH_BLOCK / SUB);
  for (int64_t k = (int64_t)blockIdx.x * (SH_BLOCK / SUB) + (threadIdx.x / SUB); k < m1 - m0; k += nsub) {
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, t0 = 0.0, t1 = 0.0, t2 = 0.0;
    const int64_t e0 = rowptr[2 * k], e1 = rowptr[2 * k + 1], e2 = rowptr[2 * k + 2];
    // U entries per lane and trip, all their loads issued before the first is used (entries beyond the row are clamped to
    // its last one and weighted 0): a lane walks ~3 entries of a row and each is a chain weight/column -> gathered residual;
    // one at a time the kernel ran at 2.3 TB/s with every SIMD full (round 5)
    constexpr int U = 4;
    for (int64_t e = e0 + sl; e < e1; e += U * SUB) {
      float4 w[U];
      int32_t c[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int64_t eu = e + u * SUB;
        const bool in = eu < e1;
        const int64_t ec = in ? eu : e1 - 1;
        w[u] = w4[ec];
        c[u] = cols[ec];
        if (!in) w[u] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
      Triple rc[U];
#pragma unroll
      for (int u = 0; u < U; ++u) rc[u] = *reinterpret_cast<const Triple*>(r + c[u]);
#pragma unroll
      for (int u = 0; u < U; ++u) {
        s0 += (double)w[u].x * rc[u].a; s1 += (double)w[u].x * rc[u].b; s2 += (double)w[u].x * rc[u].c;
        // sigma x r
        t0 += (double)w[u].z * rc[u].c - (double)w[u].w * rc[u].b;
        t1 += (double)w[u].w * rc[u].a - (double)w[u].y * rc[u].c;
        t2 += (double)w[u].y * rc[u].b - (double)w[u].z * rc[u].a;
      }
    }
    for (int64_t e = e1 + sl; e < e2; e += U * SUB) {
      double w[U];
      int32_t c[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int64_t eu = e + u * SUB;
        const bool in = eu < e2;
        const int64_t ec = in ? eu : e2 - 1;
        w[u] = in ? (double)w4[ec].x : 0.0;
        c[u] = cols[ec];
      }
      Triple rc[U];
#pragma unroll
      for (int u = 0; u < U; ++u) rc[u] = *reinterpret_cast<const Triple*>(r + c[u]);
#pragma unroll
      for (int u = 0; u < U; ++u) { t0 += w[u] * rc[u].a; t1 += w[u] * rc[u].b; t2 += w[u] * rc[u].c; }
    }
#pragma unroll
    for (int off = SUB / 2; off > 0; off >>= 1) {
      s0 += __shfl_xor(s0, off, 64); s1 += __shfl_xor(s1, off, 64); s2 += __shfl_xor(s2, off, 64);
      t0 += __shfl_xor(t0, off, 64); t1 += __shfl_xor(t1, off, 64); t2 += __shfl_xor(t2, off, 64);
    }
    if (sl < 6) {
      const double v = sl == 0 ? s0 : (sl == 1 ? s1 : (sl == 2 ? s2 : (sl == 3 ? t0 : (sl == 4 ? t1 : t2))));
      g[6 * (m0 + k) + sl] = v;
    }
  }
}

// one lattice level, nodes [n0, n1), a thread per (node, field group):
//   down: g[p] from the node's children (chi rows), up: e[c] = B_c g_c + (transfer of the parents' e) with the node's
//   6 x 6 block B; w5 = (a, bx, by, bz, c) per entry.  dot_partials (up, finest level): per-block partial of e . g.
constexpr int LAT_H_LANES = 4;      // lanes per (node, field group) in k_lat_level_h
__global__ __launch_bounds__(256) void k_lat_level_h(int64_t n0, int64_t n1, const int64_t* __restrict__ rowptr, const int32_t* __restrict__ cols,
                                                     const double* __restrict__ w5, double* __restrict__ g, double* __restrict__ e, int up,
                                                     const int32_t* __restrict__ done, const double* __restrict__ blocks,
                                                     double* __restrict__ dot_partials) {
  if (done != nullptr && *done) return;
  __shared__ double lds[256 / 64];
  double dot = 0.0;
  // Round 5: four lanes share a (node, group) and split its <= 8 entries -- a lane used to walk them one after the other, each a
  // dependent index -> gather round trip, on levels too small (<= 52 k nodes) to hide it with other waves.
  const int sub = threadIdx.x & (LAT_H_LANES - 1);
  const int64_t total = (n1 - n0) * 2;
  const int64_t stride = (int64_t)gridDim.x * (blockDim.x / LAT_H_LANES);
  // (whole quads take the same trips: the shuffles below are executed by all four lanes)
  for (int64_t t = (int64_t)blockIdx.x * (blockDim.x / LAT_H_LANES) + (threadIdx.x / LAT_H_LANES); t < total; t += stride) {
    const int64_t node = n0 + (t >> 1);
    const int grp = (int)(t & 1);
    const double* src = up ? e : g;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0;
    for (int64_t k = rowptr[node] + sub; k < rowptr[node + 1]; k += LAT_H_LANES) {
      const double* w = w5 + 5 * k;
      const double* sv = src + 6 * (int64_t)cols[k];
      if (up) {
        if (grp == 0) {                     // U_c = a U_p + Theta_p x b
          const V3 th = {sv[3], sv[4], sv[5]}, b = {w[1], w[2], w[3]};
          const V3 cb = cross3(th, b);
          s0 += w[0] * sv[0] + cb.x; s1 += w[0] * sv[1] + cb.y; s2 += w[0] * sv[2] + cb.z;
        } else {                            // Theta_c = c Theta_p
          s0 += w[4] * sv[3]; s1 += w[4] * sv[4]; s2 += w[4] * sv[5];
        }
      } else {
        if (grp == 0) {                     // g_U[p] += a g_U[c]
          s0 += w[0] * sv[0]; s1 += w[0] * sv[1]; s2 += w[0] * sv[2];
        } else {                            // g_Theta[p] += c g_Theta[c] + b x g_U[c]
          const V3 gu = {sv[0], sv[1], sv[2]}, b = {w[1], w[2], w[3]};
          const V3 cb = cross3(b, gu);
          s0 += w[4] * sv[3] + cb.x; s1 += w[4] * sv[4] + cb.y; s2 += w[4] * sv[5] + cb.z;
        }
      }
    }
#pragma unroll
    for (int off = LAT_H_LANES / 2; off > 0; off >>= 1) {
      s0 += __shfl_xor(s0, off, 64); s1 += __shfl_xor(s1, off, 64); s2 += __shfl_xor(s2, off, 64);
    }
    if (sub != 0) continue;
    double* out = (up ? e : g) + 6 * node + 3 * grp;
    if (!up) { out[0] = s0; out[1] = s1; out[2] = s2; continue; }
    const double* B = blocks + 36 * node + 18 * grp;
    const double* gn = g + 6 * node;
#pragma unroll
    for (int q = 0; q < 6; ++q) { s0 += B[q] * gn[q]; s1 += B[6 + q] * gn[q]; s2 += B[12 + q] * gn[q]; }
    out[0] = s0; out[1] = s1; out[2] = s2;
    dot += s0 * gn[3 * grp] + s1 * gn[3 * grp + 1] + s2 * gn[3 * grp + 2];
  }
  if (dot_partials != nullptr) {
    const double tsum = femo_block_sum<256>(dot, lds);
    if (threadIdx.x == 0) dot_partials[blockIdx.x] = tsum;
  }
}

// all levels between the coarse-solve level and the finest lattice at once (composite of the child transfers, built on
// the host in the same (A, B, C) form): one wave per node
__global__ __launch_bounds__(SH_BLOCK) void k_lat_down_composite_h(int64_t row0, int64_t n_rows, const int64_t* __restrict__ rowptr,
                                                                   const int32_t* __restrict__ cols, const double* __restrict__ w5,
                                                                   double* __restrict__ g, const int32_t* __restrict__ done) {
  if (done != nullptr && *done) return;
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * (SH_BLOCK / 64) + (threadIdx.x >> 6);
  if (row >= n_rows) return;
  double s[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
  for (int64_t k = rowptr[row] + lane; k < rowptr[row + 1]; k += 64) {
    const double* w = w5 + 5 * k;
    const double* sv = g + 6 * (int64_t)cols[k];
    const V3 gu = {sv[0], sv[1], sv[2]}, b = {w[1], w[2], w[3]};
    const V3 cb = cross3(b, gu);
    s[0] += w[0] * gu.x; s[1] += w[0] * gu.y; s[2] += w[0] * gu.z;
    s[3] += w[4] * sv[3] + cb.x; s[4] += w[4] * sv[4] + cb.y; s[5] += w[4] * sv[5] + cb.z;
  }
#pragma unroll
  for (int f = 0; f < 6; ++f) s[f] = femo_wave_sum(s[f]);
  if (lane == 0) {
    double* dst = g + 6 * (row0 + row);
#pragma unroll
    for (int f = 0; f < 6; ++f) dst[f] = s[f];
  }
}

// (P_L e_L) of one point, 8 lanes per point (lane sl = corner): the three components, valid in all 8 lanes
__device__ __forceinline__ void prolong_point_h(int64_t p, int sl, int64_t n_unode, const int32_t* __restrict__ fin_idx, const float4* __restrict__ fin_w4,
                                                const double* __restrict__ t, double& s0, double& s1, double& s2) {
  const float4 w = fin_w4[p * 8 + sl];
  const int32_t idx = fin_idx[p * 8 + sl];
  if (p < n_unode) {
    // idx = 6 node: the node's six values as three 16-byte loads (48 node bytes from a 256-byte aligned base)
    const double2* tn = reinterpret_cast<const double2*>(t + idx);
    const double2 q0 = tn[0], q1 = tn[1], q2 = tn[2];
    const V3 th = {q1.y, q2.x, q2.y}, sg = {(double)w.y, (double)w.z, (double)w.w};
    const V3 cb = cross3(th, sg);
    s0 = (double)w.x * q0.x + cb.x; s1 = (double)w.x * q0.y + cb.y; s2 = (double)w.x * q1.x + cb.z;
  } else {
    const Triple tp = *reinterpret_cast<const Triple*>(t + idx);     // idx = 6 node + 3
    s0 = (double)w.x * tp.a; s1 = (double)w.x * tp.b; s2 = (double)w.x * tp.c;
  }
#pragma unroll
  for (int off = 4; off > 0; off >>= 1) {
    s0 += __shfl_xor(s0, off, 64);
    s1 += __shfl_xor(s1, off, 64);
    s2 += __shfl_xor(s2, off, 64);
  }
}

// z = D^-1 r + P_L e_L (8 lanes per point, one finest-level entry each, three components) and the per-block partial of
// r.z; imposed dofs: z = 0
__global__ __launch_bounds__(SH_BLOCK) void k_pc_prolong(int64_t n_pts, const int32_t* __restrict__ fin_idx, const float* __restrict__ fin_w,
                                                         const uint8_t* __restrict__ fixed, const double* __restrict__ dinv,
                                                         const double* __restrict__ r, const double* __restrict__ t, double* __restrict__ z,
                                                         double* __restrict__ partials, const int32_t* __restrict__ done,
                                                         const float* __restrict__ dinv3 = nullptr, const float4* __restrict__ fin_w4 = nullptr,
                                                         int64_t n_unode = 0) {
  if (done != nullptr && *done) return;
  __shared__ double lds[SH_BLOCK / 64];
  constexpr int SUB = 8;
  const int sl = threadIdx.x & (SUB - 1);
  const int64_t nsub = (int64_t)gridDim.x * (SH_BLOCK / SUB);
  double dot = 0.0;
  for (int64_t p = (int64_t)blockIdx.x * (SH_BLOCK / SUB) + (threadIdx.x / SUB); p < n_pts; p += nsub) {
    double s0, s1, s2;
    if (fin_w4 != nullptr) {                                 // Hermite-type finest transfer
      prolong_point_h(p, sl, n_unode, fin_idx, fin_w4, t, s0, s1, s2);
    } else {
      const double w = (double)fin_w[p * 8 + sl];
      const Triple tp = *reinterpret_cast<const Triple*>(t + fin_idx[p * 8 + sl]);
      s0 = w * tp.a; s1 = w * tp.b; s2 = w * tp.c;
#pragma unroll
      for (int off = SUB / 2; off > 0; off >>= 1) {
        s0 += __shfl_xor(s0, off, 64);
        s1 += __shfl_xor(s1, off, 64);
        s2 += __shfl_xor(s2, off, 64);
      }
    }
    if (sl < 3) {
      const int64_t row = 3 * p + sl;
      const bool rf = fixed != nullptr && fixed[row];
      const double ri = r[row];
      double sm;                                           // the smoother: point-block (3 x 3) or plain Jacobi
      if (dinv3 != nullptr) {
        const float* B = dinv3 + 9 * p + 3 * sl;
        const Triple rp = *reinterpret_cast<const Triple*>(r + 3 * p);
        sm = (double)B[0] * rp.a + (double)B[1] * rp.b + (double)B[2] * rp.c;
      } else {
        sm = dinv[row] * ri;
      }
      const double zi = rf ? 0.0 : sm + (sl == 0 ? s0 : (sl == 1 ? s1 : s2));
      z[row] = zi;
      dot += ri * zi;
    }
  }
  const double tsum = femo_block_sum<SH_BLOCK>(dot, lds);
  if (threadIdx.x == 0) partials[blockIdx.x] = tsum;
}

__global__ void k_copy(int64_t n, const double* __restrict__ a, double* __restrict__ b) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) b[i] = a[i];
}

// lattice mode of the CG kernels: x += alpha p; r -= alpha q (no norm: r.z comes from k_pc_prolong)
__global__ __launch_bounds__(SH_BLOCK) void k_scg_xr_plain(int64_t n, int nb_pq, const double* __restrict__ part_pq, const double* __restrict__ scal,
                                                           const double* __restrict__ p, const double* __restrict__ q,
                                                           double* __restrict__ x, double* __restrict__ r, const int32_t* __restrict__ done) {
  if (*done) return;
  __shared__ double lds[SH_BLOCK / 64];
  const double pq = fold(part_pq, nb_pq, lds);
  const double alpha = pq != 0.0 ? scal[0] / pq : 0.0;
  for (int64_t i = (int64_t)blockIdx.x * SH_BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * SH_BLOCK) {
    x[i] += alpha * p[i];
    r[i] -= alpha * q[i];
  }
}

// p = z + beta p with z given (see k_scg_p)
__global__ __launch_bounds__(SH_BLOCK) void k_scg_p_z(int64_t n, int it, int nb_rz, const double* __restrict__ part_rz, double* __restrict__ scal,
                                                      const double* __restrict__ z, double* __restrict__ p, int32_t* __restrict__ flag,
                                                      double* __restrict__ gamma_out) {
  if (flag[0]) return;
  __shared__ double lds[SH_BLOCK / 64];
  const double g1 = fold(part_rz, nb_rz, lds);
  const double g0 = scal[0];
  const bool conv = g1 <= scal[2] || !(g1 == g1);
  const double beta = g0 != 0.0 ? g1 / g0 : 0.0;
  if (!conv) {
    for (int64_t i = (int64_t)blockIdx.x * SH_BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * SH_BLOCK) p[i] = z[i] + beta * p[i];
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    gamma_out[0] = g1;
    flag[1] = it + 1;
    if (conv) { flag[2] = (g1 == g1) ? 0 : 1; __threadfence(); flag[0] = it + 1; }
  }
}

// x += alpha p; r -= alpha q per POINT (three dofs), and the per-block partial of r . (B r), B = the point's 3 x 3
// smoother block (or 1 / diag): the first half of r . z = r . B r + (P^T r) . e  (see k_lat_level, k_pc_prolong_fused)
__global__ __launch_bounds__(SH_BLOCK) void k_scg_xr_pt(int64_t n_pts, int nb_pq, const double* __restrict__ part_pq, const double* __restrict__ scal,
                                                        const double* __restrict__ p, const double* __restrict__ q, const double* __restrict__ dinv,
                                                        const float* __restrict__ dinv3, double* __restrict__ x, double* __restrict__ r,
                                                        double* __restrict__ part_rB, const int32_t* __restrict__ done, double* __restrict__ alpha_out = nullptr) {
  if (*done) return;
  __shared__ double lds[SH_BLOCK / 64];
  const double pq = fold(part_pq, nb_pq, lds);
  const double alpha = pq != 0.0 ? scal[0] / pq : 0.0;
  // alpha_out != nullptr: x += alpha p is carried by the preconditioner's first coarse product (ShellXCarry); this kernel
  // then streams q, r and the smoother blocks only
  const bool carry = alpha_out != nullptr;
  if (carry && blockIdx.x == 0 && threadIdx.x == 0) *alpha_out = alpha;
  double s = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * SH_BLOCK + threadIdx.x; i < n_pts; i += (int64_t)gridDim.x * SH_BLOCK) {
    const Triple qq = *reinterpret_cast<const Triple*>(q + 3 * i);
    Triple rr = *reinterpret_cast<const Triple*>(r + 3 * i);
    if (!carry) {
      const Triple pp = *reinterpret_cast<const Triple*>(p + 3 * i);
      Triple xx = *reinterpret_cast<const Triple*>(x + 3 * i);
      xx.a += alpha * pp.a; xx.b += alpha * pp.b; xx.c += alpha * pp.c;
      *reinterpret_cast<Triple*>(x + 3 * i) = xx;
    }
    rr.a -= alpha * qq.a; rr.b -= alpha * qq.b; rr.c -= alpha * qq.c;
    *reinterpret_cast<Triple*>(r + 3 * i) = rr;
    if (dinv3 != nullptr) {
      const float* B = dinv3 + 9 * i;
      s += rr.a * ((double)B[0] * rr.a + (double)B[1] * rr.b + (double)B[2] * rr.c) + rr.b * ((double)B[3] * rr.a + (double)B[4] * rr.b + (double)B[5] * rr.c) +
           rr.c * ((double)B[6] * rr.a + (double)B[7] * rr.b + (double)B[8] * rr.c);
    } else {
      s += rr.a * rr.a * dinv[3 * i] + rr.b * rr.b * dinv[3 * i + 1] + rr.c * rr.c * dinv[3 * i + 2];
    }
  }
  const double t = femo_block_sum<SH_BLOCK>(s, lds);
  if (threadIdx.x == 0) part_rB[blockIdx.x] = t;
}

// The prolongation with the direction update fused in: gamma' = r . z is known before z is formed (partials of
// r . B r from k_scg_xr_pt, of (P^T r) . e from the finest k_lat_level), so beta and the stopping test are, and the pass
// writes p = z + beta p directly -- z = B r + P_L e_L is never stored, k_scg_p_z and its three vector streams are gone.
__global__ __launch_bounds__(SH_BLOCK) void k_pc_prolong_fused(int64_t n_pts, int it, int nb_rB, const double* __restrict__ part_rB, int nb_te,
                                                               const double* __restrict__ part_te, double* __restrict__ scal,
                                                               const int32_t* __restrict__ fin_idx, const float* __restrict__ fin_w,
                                                               const uint8_t* __restrict__ fixed, const double* __restrict__ dinv,
                                                               const float* __restrict__ dinv3, const double* __restrict__ r,
                                                               const double* __restrict__ t, double* __restrict__ p,
                                                               int32_t* __restrict__ flag, double* __restrict__ gamma_out,
                                                               const float4* __restrict__ fin_w4 = nullptr, int64_t n_unode = 0) {
  if (flag[0]) return;
  __shared__ double lds[SH_BLOCK / 64];
  const double g1 = fold(part_rB, nb_rB, lds) + fold(part_te, nb_te, lds);
  const double g0 = scal[0];
  const bool conv = g1 <= scal[2] || !(g1 == g1);
  const double beta = g0 != 0.0 ? g1 / g0 : 0.0;
  if (!conv) {
    constexpr int SUB = 8;
    const int sl = threadIdx.x & (SUB - 1);
    const int64_t nsub = (int64_t)gridDim.x * (SH_BLOCK / SUB);
    for (int64_t pt = (int64_t)blockIdx.x * (SH_BLOCK / SUB) + (threadIdx.x / SUB); pt < n_pts; pt += nsub) {
      double s0, s1, s2;
      if (fin_w4 != nullptr) {
        prolong_point_h(pt, sl, n_unode, fin_idx, fin_w4, t, s0, s1, s2);
      } else {
        const double w = (double)fin_w[pt * 8 + sl];
        const Triple tp = *reinterpret_cast<const Triple*>(t + fin_idx[pt * 8 + sl]);
        s0 = w * tp.a; s1 = w * tp.b; s2 = w * tp.c;
#pragma unroll
        for (int off = SUB / 2; off > 0; off >>= 1) {
          s0 += __shfl_xor(s0, off, 64);
          s1 += __shfl_xor(s1, off, 64);
          s2 += __shfl_xor(s2, off, 64);
        }
      }
      if (sl < 3) {
        const int64_t row = 3 * pt + sl;
        const bool rf = fixed != nullptr && fixed[row];
        double sm;
        if (dinv3 != nullptr) {
          const float* B = dinv3 + 9 * pt + 3 * sl;
          const Triple rp = *reinterpret_cast<const Triple*>(r + 3 * pt);
          sm = (double)B[0] * rp.a + (double)B[1] * rp.b + (double)B[2] * rp.c;
        } else {
          sm = dinv[row] * r[row];
        }
        const double zi = rf ? 0.0 : sm + (sl == 0 ? s0 : (sl == 1 ? s1 : s2));
        p[row] = zi + beta * p[row];
      }
    }
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    gamma_out[0] = g1;
    flag[1] = it + 1;
    if (conv) { flag[2] = (g1 == g1) ? 0 : 1; __threadfence(); flag[0] = it + 1; }
  }
}

inline unsigned sgrid(int64_t n, int per = SH_BLOCK) {
  int64_t g = (n + per - 1) / per;
  if (g < 1) g = 1;
  return (unsigned)std::min<int64_t>(g, 1 << 20);
}

femo_shell_view view(const femo_shell* s) {
  femo_shell_view v;
  v.n_vert = s->n_vert; v.n_cell = s->n_cell; v.n_unode = s->n_unode;
  v.x = s->d_x; v.conn = s->d_conn; v.cedge = s->d_cedge;
  v.cell_owned = s->d_cell_owned;
  return v;
}

// ---- partitioned shells (several ranks): rows of points owned elsewhere, halo, all-reduced scalars ----------------------
// the scalar rows of block row p are the 9 (brow[p+1] - brow[p]) values from 9 brow[p] on
__global__ void k_zero_unowned_rows(int64_t n_pts, const uint8_t* __restrict__ owned, const int64_t* __restrict__ brow, double* __restrict__ vals) {
  const int lane = threadIdx.x & 63;
  for (int64_t p = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); p < n_pts; p += (int64_t)gridDim.x * (blockDim.x >> 6)) {
    if (owned[p]) continue;
    for (int64_t k = 9 * brow[p] + lane; k < 9 * brow[p + 1]; k += 64) vals[k] = 0.0;
  }
}

__global__ void k_mask_unowned(int64_t n_pts, const uint8_t* __restrict__ owned, double* __restrict__ v) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < 3 * n_pts; i += (int64_t)gridDim.x * blockDim.x)
    if (!owned[i / 3]) v[i] = 0.0;
}

__global__ void k_halo_pack(int64_t n, const int32_t* __restrict__ idx, const double* __restrict__ v, double* __restrict__ buf) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) buf[i] = v[idx[i]];
}

__global__ void k_halo_unpack(int64_t n, const int32_t* __restrict__ idx, const double* __restrict__ buf, double* __restrict__ v) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) v[idx[i]] = buf[i];
}

// out[0] = sum of the partials (one workgroup): what the all-reduce between a producer and its consumer works on
__global__ __launch_bounds__(SH_BLOCK) void k_fold1(int nb, const double* __restrict__ partials, double* __restrict__ out, const int32_t* __restrict__ done) {
  if (done != nullptr && *done) return;
  __shared__ double lds[SH_BLOCK / 64];
  const double g = fold(partials, nb, lds);
  if (threadIdx.x == 0) out[0] = g;
}

template <class T>
int to_device(T** d, const T* h, int64_t n, hipStream_t st) {
  FEMO_HIP_CHECK(hipMalloc(d, std::max<int64_t>(n, 1) * sizeof(T)));
  if (n > 0) FEMO_HIP_CHECK(hipMemcpyAsync(*d, h, n * sizeof(T), hipMemcpyHostToDevice, st));
  return 0;
}

int reduce_partials(femo_ctx* ctx, const double* d_part, int nb, double* host) {
  std::vector<double> h((size_t)nb);
  FEMO_HIP_CHECK(hipMemcpyAsync(h.data(), d_part, nb * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  FEMO_HIP_CHECK(hipStreamSynchronize(ctx->stream));
  double s = 0.0;
  for (double v : h) s += v;
  if (ctx->nranks > 1) {
    // a shell on several ranks is partitioned: the caller integrates over the cells it owns (cell weights) and the
    // value is the sum over the ranks
    FEMO_HIP_CHECK(hipMemcpyAsync(ctx->d_scal, &s, sizeof s, hipMemcpyHostToDevice, ctx->stream));
    FEMO_TRY(femo_coll_allreduce(ctx, ctx->d_scal, 1, ctx->stream));
    FEMO_HIP_CHECK(hipMemcpyAsync(&s, ctx->d_scal, sizeof s, hipMemcpyDeviceToHost, ctx->stream));
    FEMO_HIP_CHECK(hipStreamSynchronize(ctx->stream));
  }
  *host = s;
  return 0;
}

// entries of the points owned by other ranks <- their owners' values (v: a state-sized device vector)
int shell_halo(femo_shell* s, double* v, hipStream_t st) {
  if (s->d_owned == nullptr || s->ctx->nranks == 1 || s->n_nbr == 0) return 0;
  const int64_t ns = s->send_ptr[(size_t)s->n_nbr], nr = s->recv_ptr[(size_t)s->n_nbr];
  if (ns > 0) hipLaunchKernelGGL(k_halo_pack, dim3(sgrid(ns, 256)), dim3(256), 0, st, ns, s->d_send_idx, v, s->d_send_buf);
  FEMO_HIP_CHECK(hipGetLastError());
  FEMO_TRY(femo_coll_neighbors(s->ctx, s->n_nbr, s->nbr.data(), s->send_ptr.data(), s->d_send_buf, s->recv_ptr.data(), s->d_recv_buf, st));
  if (nr > 0) hipLaunchKernelGGL(k_halo_unpack, dim3(sgrid(nr, 256)), dim3(256), 0, st, nr, s->d_recv_idx, s->d_recv_buf, v);
  FEMO_HIP_CHECK(hipGetLastError());
  return 0;
}

// sum over the ranks of a device array (no-op on one rank)
int shell_allreduce(femo_shell* s, double* d, int64_t n, hipStream_t st) {
  if (s->d_owned == nullptr || s->ctx->nranks == 1) return 0;
  return femo_coll_allreduce(s->ctx, d, n, st);
}

// the rank's share of an assembled matrix: rows of the points owned elsewhere are zero
int shell_zero_unowned_rows(femo_shell* s, double* vals, hipStream_t st) {
  if (s->d_owned == nullptr) return 0;
  FEMO_REQUIRE(s->d_brow != nullptr, "a partitioned shell needs the node-block view of the pattern");
  hipLaunchKernelGGL(k_zero_unowned_rows, dim3(sgrid(s->n_bnode, 4)), dim3(256), 0, st, s->n_bnode, s->d_owned, s->d_brow, vals);
  FEMO_HIP_CHECK(hipGetLastError());
  return 0;
}

}  // namespace

extern "C" {

int femo_shell_create(femo_ctx* ctx, int64_t n_vert, const double* x, int64_t n_cell, const int32_t* conn, int64_t n_edge,
                      const int32_t* cell_edges, const int64_t* rowptr, const int32_t* cols, const int32_t* elem_pos,
                      femo_shell** out) {
  FEMO_REQUIRE(ctx && x && conn && cell_edges && rowptr && cols && elem_pos && out, "null argument");
  FEMO_REQUIRE(n_vert > 0 && n_cell > 0 && n_edge > 0, "empty shell mesh");
  FEMO_HIP_CHECK(hipSetDevice(ctx->device));
  femo_shell* s = new femo_shell();
  s->ctx = ctx;
  s->n_vert = n_vert; s->n_cell = n_cell; s->n_edge = n_edge;
  s->n_unode = n_vert + n_edge;
  s->n_dof = 3 * s->n_unode + 3 * n_vert;
  s->nnz = rowptr[s->n_dof];
  FEMO_REQUIRE(s->nnz > 0 && s->nnz < (int64_t)1 << 31, "pattern too large for 32-bit element positions");
  hipStream_t st = ctx->stream;
  FEMO_TRY(to_device(&s->d_x, x, n_vert * 3, st));
  FEMO_TRY(to_device(&s->d_conn, conn, n_cell * 3, st));
  FEMO_TRY(to_device(&s->d_cedge, cell_edges, n_cell * 3, st));
  FEMO_TRY(to_device(&s->d_rowptr, rowptr, s->n_dof + 1, st));
  FEMO_TRY(to_device(&s->d_cols, cols, s->nnz, st));
  FEMO_TRY(to_device(&s->d_epos, elem_pos, n_cell * 729, st));
  {
    // node-block view: valid when every node's three rows have the same columns in runs of three (fea/shell.py numbers
    // the dofs 3 node + component, so the element-coupling pattern always is)
    const int64_t nbn = s->n_dof / 3;
    std::vector<int64_t> brow((size_t)nbn + 1, 0);
    std::vector<int32_t> bcols;
    bcols.reserve((size_t)(s->nnz / 9));
    bool ok = s->n_dof % 3 == 0;
    for (int64_t b = 0; ok && b < nbn; ++b) {
      const int64_t r0 = rowptr[3 * b], len = rowptr[3 * b + 1] - r0;
      ok = len % 3 == 0 && rowptr[3 * b + 2] - rowptr[3 * b + 1] == len && rowptr[3 * b + 3] - rowptr[3 * b + 2] == len &&
           r0 == 9 * brow[(size_t)b];
      for (int64_t j = 0; ok && j < len; j += 3) {
        const int32_t c = cols[r0 + j];
        ok = c % 3 == 0 && cols[r0 + j + 1] == c + 1 && cols[r0 + j + 2] == c + 2 && cols[r0 + len + j] == c &&
             cols[r0 + 2 * len + j] == c;
        bcols.push_back(c);
      }
      brow[(size_t)b + 1] = brow[(size_t)b] + len / 3;
    }
    if (ok) {
      s->n_bnode = nbn;
      FEMO_TRY(to_device(&s->d_brow, brow.data(), nbn + 1, st));
      FEMO_TRY(to_device(&s->d_bcols, bcols.data(), (int64_t)bcols.size(), st));
      // block-SELL: slots per slice = the longest of its BSW block rows
      const int64_t nsl = (nbn + BSW - 1) / BSW;
      std::vector<int64_t> off((size_t)nsl + 1, 0);
      for (int64_t sl = 0; sl < nsl; ++sl) {
        int64_t mx = 0;
        for (int64_t b = BSW * sl; b < std::min<int64_t>(BSW * sl + BSW, nbn); ++b) mx = std::max(mx, brow[(size_t)b + 1] - brow[(size_t)b]);
        off[(size_t)sl + 1] = off[(size_t)sl] + mx;
      }
      s->n_bslice = nsl; s->bsell_blocks = off[(size_t)nsl];
      FEMO_TRY(to_device(&s->d_bs_off, off.data(), nsl + 1, st));
      FEMO_HIP_CHECK(hipMalloc(&s->d_bs_cols, std::max<int64_t>(s->bsell_blocks, 1) * BSW * sizeof(int32_t)));
      FEMO_HIP_CHECK(hipMalloc(&s->d_bs_vals, std::max<int64_t>(s->bsell_blocks, 1) * 9 * BSW * sizeof(double)));
      FEMO_HIP_CHECK(hipStreamSynchronize(st));
    }
  }
  const int64_t n = s->n_dof;
  FEMO_HIP_CHECK(hipMalloc(&s->d_r, n * sizeof(double)));
  FEMO_HIP_CHECK(hipMalloc(&s->d_p, n * sizeof(double)));
  FEMO_HIP_CHECK(hipMalloc(&s->d_q, n * sizeof(double)));
  FEMO_HIP_CHECK(hipMalloc(&s->d_dinv, n * sizeof(double)));
  FEMO_HIP_CHECK(hipMalloc(&s->d_scal, 8 * sizeof(double)));
  FEMO_HIP_CHECK(hipMalloc(&s->d_part, 3 * SH_MAXPART * sizeof(double)));
  FEMO_HIP_CHECK(hipMalloc(&s->d_flag, 4 * sizeof(int32_t)));
  FEMO_HIP_CHECK(hipStreamSynchronize(st));
  *out = s;
  return 0;
}

int femo_shell_destroy(femo_shell* s) {
  if (!s) return 0;
  hipStreamSynchronize(s->ctx->stream);
  hipFree(s->d_x); hipFree(s->d_conn); hipFree(s->d_cedge); hipFree(s->d_rowptr); hipFree(s->d_cols); hipFree(s->d_epos); hipFree(s->d_brow); hipFree(s->d_bcols); hipFree(s->d_bs_off); hipFree(s->d_bs_cols); hipFree(s->d_bs_vals);
  hipFree(s->d_ptp_rowptr); hipFree(s->d_ptp_cols); hipFree(s->d_ptp_vals);
  hipFree(s->d_cs_xyz); hipFree(s->d_cs_ptr); hipFree(s->d_cs_pts); hipFree(s->d_cs_nbr); hipFree(s->d_cs_A); hipFree(s->d_cs_Af); hipFree(s->d_cs_tmp); hipFree(s->d_cs_dinv); hipFree(s->d_cs_T); hipFree(s->d_cs_info); hipFree(s->d_cs_pcell);
  hipFree(s->d_cd_rowptr); hipFree(s->d_cd_cols); hipFree(s->d_cd_vals);
  hipFree(s->d_cell_owned);
  hipFree(s->d_fin_w4); hipFree(s->d_hp_rowptr); hipFree(s->d_hp_cols); hipFree(s->d_hp_w4); hipFree(s->d_par_w5); hipFree(s->d_chi_w5);
  hipFree(s->d_lvl_w4); hipFree(s->d_cs_w4); hipFree(s->d_hd_rowptr); hipFree(s->d_hd_cols); hipFree(s->d_hd_w5);
  hipFree(s->d_pen_nodes); hipFree(s->d_pen_pos); hipFree(s->d_pen_coef);
  hipFree(s->d_owned); hipFree(s->d_send_idx); hipFree(s->d_recv_idx); hipFree(s->d_send_buf); hipFree(s->d_recv_buf);
  hipFree(s->d_r); hipFree(s->d_p); hipFree(s->d_q); hipFree(s->d_dinv); hipFree(s->d_scal); hipFree(s->d_part); hipFree(s->d_flag);
  hipFree(s->d_ell_idx); hipFree(s->d_ell_w);
  hipFree(s->d_par_rowptr); hipFree(s->d_par_cols); hipFree(s->d_par_vals); hipFree(s->d_chi_rowptr); hipFree(s->d_chi_cols); hipFree(s->d_chi_vals);
  hipFree(s->d_fixed_kept); hipFree(s->d_bi_ptr); hipFree(s->d_bi_lvl); hipFree(s->d_bi_pts); hipFree(s->d_bi_pcell); hipFree(s->d_fixbits);
  hipFree(s->d_coarse); hipFree(s->d_cblk); hipFree(s->d_lvl_node); hipFree(s->d_lvl_w); hipFree(s->d_dinv3); hipFree(s->d_t); hipFree(s->d_e); hipFree(s->d_z); hipFree(s->d_fin_idx); hipFree(s->d_fin_w);
  delete s;
  return 0;
}

// Lattice preconditioner data (built on the host: fea/shell.py::ShellSpace.lattice_pc): P as ELL, `width` = 8 x levels
// entries per dof (column, weight; weight 0 pads), and P^T as CSR over the n_lat lattice unknowns.
int femo_shell_pc_create(femo_shell* s, int width, int64_t n_nodes, int n_levels, const int64_t* level_offsets,
                         const int32_t* ell_idx, const double* ell_w,
                         const int64_t* pt_rowptr, const int32_t* pt_cols, const double* pt_vals,
                         const int64_t* par_rowptr, const int32_t* par_cols, const double* par_vals,
                         const int64_t* chi_rowptr, const int32_t* chi_cols, const double* chi_vals) {
  FEMO_REQUIRE(s && level_offsets && ell_idx && ell_w && pt_rowptr && pt_cols && pt_vals && par_rowptr && par_cols && par_vals &&
               chi_rowptr && chi_cols && chi_vals, "null argument");
  FEMO_REQUIRE(n_levels > 0 && width == 8 * n_levels && n_nodes > 0 && level_offsets[n_levels] == n_nodes, "bad preconditioner shape");
  FEMO_REQUIRE(s->pc_width == 0, "the shell already has a preconditioner");
  hipStream_t st = s->ctx->stream;
  FEMO_HIP_CHECK(hipSetDevice(s->ctx->device));
  const int64_t n_lat = 6 * n_nodes;
  FEMO_TRY(to_device(&s->d_ell_idx, ell_idx, s->n_dof * width, st));
  FEMO_TRY(to_device(&s->d_ell_w, ell_w, s->n_dof * width, st));
  FEMO_TRY(to_device(&s->d_par_rowptr, par_rowptr, n_nodes + 1, st));
  FEMO_TRY(to_device(&s->d_par_cols, par_cols, par_rowptr[n_nodes], st));
  FEMO_TRY(to_device(&s->d_par_vals, par_vals, par_rowptr[n_nodes], st));
  FEMO_TRY(to_device(&s->d_chi_rowptr, chi_rowptr, n_nodes + 1, st));
  FEMO_TRY(to_device(&s->d_chi_cols, chi_cols, chi_rowptr[n_nodes], st));
  FEMO_TRY(to_device(&s->d_chi_vals, chi_vals, chi_rowptr[n_nodes], st));
  {
    // The finest level runs every iteration, in arrays of its own and per POINT (a P2 node with its three displacements,
    // a vertex with its three rotations: dofs 3 p .. 3 p + 2): the three components share the eight lattice nodes and
    // weights, so the prolongation reads 8 (index, weight) pairs per point instead of 24, and P_L^T has one row per
    // (lattice node, field group) listing points instead of six rows listing dofs.
    const int64_t n_pts = s->n_dof / 3;
    FEMO_REQUIRE(s->n_dof % 3 == 0, "dofs are not numbered 3 point + component");
    std::vector<int32_t> fi((size_t)n_pts * 8);
    std::vector<float> fw((size_t)n_pts * 8);
    for (int64_t p = 0; p < n_pts; ++p)
      for (int a = 0; a < 8; ++a) {
        const int64_t e0 = (3 * p) * width + (width - 8) + a;
        fi[(size_t)p * 8 + a] = ell_idx[e0];                      // unknown of component 0; components 1, 2 follow it
        fw[(size_t)p * 8 + a] = (float)ell_w[e0];
        for (int k = 1; k < 3; ++k) {
          const int64_t ek = (3 * p + k) * width + (width - 8) + a;
          FEMO_REQUIRE(ell_idx[ek] == ell_idx[e0] + k && ell_w[ek] == ell_w[e0], "components of a point do not share their lattice weights");
        }
      }
    FEMO_TRY(to_device(&s->d_fin_idx, fi.data(), n_pts * 8, st));
    FEMO_TRY(to_device(&s->d_fin_w, fw.data(), n_pts * 8, st));
    // rows 6 m + 0 (displacement group) and 6 m + 3 (rotation group) of P^T for the finest level's nodes m
    const int64_t m0 = level_offsets[n_levels - 1], m1 = level_offsets[n_levels];
    std::vector<int64_t> rp((size_t)(2 * (m1 - m0) + 1), 0);
    std::vector<int32_t> pc;
    std::vector<float> pv;
    for (int64_t m = m0; m < m1; ++m)
      for (int g = 0; g < 2; ++g) {
        const int64_t row = 6 * m + 3 * g;
        for (int64_t e = pt_rowptr[row]; e < pt_rowptr[row + 1]; ++e) {
          FEMO_REQUIRE(pt_cols[e] % 3 == 0, "P^T row of component 0 lists another component");
          pc.push_back(pt_cols[e]);
          pv.push_back((float)pt_vals[e]);
        }
        for (int k = 1; k < 3; ++k)
          FEMO_REQUIRE(pt_rowptr[row + k + 1] - pt_rowptr[row + k] == pt_rowptr[row + 1] - pt_rowptr[row], "P^T rows of a field group differ");
        rp[(size_t)(2 * (m - m0) + g + 1)] = (int64_t)pc.size();
      }
    FEMO_TRY(to_device(&s->d_ptp_rowptr, rp.data(), (int64_t)rp.size(), st));
    FEMO_TRY(to_device(&s->d_ptp_cols, pc.data(), (int64_t)pc.size(), st));
    FEMO_TRY(to_device(&s->d_ptp_vals, pv.data(), (int64_t)pv.size(), st));
    FEMO_HIP_CHECK(hipStreamSynchronize(st));
  }
  FEMO_HIP_CHECK(hipMalloc(&s->d_coarse, n_lat * sizeof(double)));
  FEMO_HIP_CHECK(hipMalloc(&s->d_cblk, n_nodes * 36 * sizeof(double)));
  FEMO_HIP_CHECK(hipMalloc(&s->d_dinv3, (s->n_dof / 3) * 9 * sizeof(float)));
  FEMO_HIP_CHECK(hipMalloc(&s->d_t, n_lat * sizeof(double)));
  FEMO_HIP_CHECK(hipMalloc(&s->d_e, n_lat * sizeof(double)));
  FEMO_HIP_CHECK(hipMalloc(&s->d_z, s->n_dof * sizeof(double)));
  FEMO_HIP_CHECK(hipStreamSynchronize(st));
  s->level_off.assign(level_offsets, level_offsets + n_levels + 1);
  s->pc_width = width; s->pc_levels = n_levels; s->n_lat = n_lat; s->pc_nodes = n_nodes;
  return 0;
}

// Dense Galerkin operator of the coarse-solve level for the current stiffness and mask, and the factors of its inverse.
// On failure (a pivot not positive, an element larger than a coarse cell) the coarse solve is switched off and the
// diagonal levels take over: the preconditioner changes, the solution does not.
static int shell_pc_coarse_setup(femo_shell* s, const femo_vec* vals, const uint8_t* d_fixed) {
  s->cs_ready = false;
  s->hermite_on = false;
  if (s->cs_level < 0 || s->d_brow == nullptr) return 0;
  hipStream_t st = s->ctx->stream;
  const int64_t n = s->cs_n, N = s->cs_N;
  const int nblk = (int)(N / DT);
  static const bool dbg = FEMO_TUNE_ENV("FEMO_DEBUG_COARSE") != nullptr;
  auto now = [&] { if (dbg) (void)hipStreamSynchronize(st); return std::chrono::steady_clock::now(); };
  auto t0 = now();
  FEMO_HIP_CHECK(hipMemsetAsync(s->d_cs_A, 0, N * N * sizeof(double), st));
  FEMO_HIP_CHECK(hipMemsetAsync(s->d_cs_info, 0, 4 * sizeof(int32_t), st));
  if (s->hermite && s->d_lvl_node != nullptr && !femo_env_flag("FEMO_SHELL_TRILINEAR") && !femo_env_flag("FEMO_SHELL_NO_BLOCKS")) {
    if (femo_env_flag("FEMO_SHELL_CG_ATOMIC") || s->cs_max_item > MM_ITEM) {      // (items above 256 points: only with FEMO_SHELL_CG_CHUNK)
      for (int pass = 0; pass < 2; ++pass)
        hipLaunchKernelGGL(k_pc_coarse_galerkin_h, dim3((unsigned)s->cs_items), dim3(256), CGH_LDS, st, pass, s->cs_level, s->pc_width, s->level_off[s->cs_level], N,
                           s->n_unode, s->d_cs_ptr, s->d_cs_pts, s->d_cs_nbr, s->d_cs_pcell, s->d_brow, s->d_bcols, vals->d, d_fixed, s->d_ell_idx,
                           s->d_cs_w4, s->d_cs_A, s->d_cs_info);
    } else {
      hipLaunchKernelGGL(k_pc_coarse_galerkin_mm, dim3((unsigned)s->cs_items), dim3(MM_THREADS), MM_LDS, st, s->cs_level, s->pc_width, s->level_off[s->cs_level], N,
                         s->n_unode, s->d_cs_ptr, s->d_cs_pts, s->d_cs_nbr, s->d_cs_pcell, s->d_brow, s->d_bcols, vals->d, d_fixed, s->d_ell_idx,
                         s->d_cs_w4, s->d_cs_A, s->d_cs_info);
    }
    hipLaunchKernelGGL(k_pc_coarse_mirror_tu, dim3(sgrid(n, 256), (unsigned)n), dim3(256), 0, st, n, N, s->d_cs_A);
    s->hermite_on = true;
  } else {
    hipLaunchKernelGGL(k_pc_coarse_galerkin, dim3((unsigned)s->cs_items), dim3(256), CG_LDS, st, s->cs_level, s->pc_width, s->level_off[s->cs_level], N,
                       s->n_unode, s->d_cs_ptr, s->d_cs_pts, s->d_cs_nbr, s->d_cs_xyz, s->d_cs_pcell, s->d_brow, s->d_bcols, vals->d, d_fixed, s->d_ell_idx,
                       s->d_ell_w, s->d_cs_A, s->d_cs_info);
    s->hermite_on = false;
  }
  FEMO_HIP_CHECK(hipGetLastError());
  FEMO_TRY(shell_allreduce(s, s->d_cs_A, N * N, st));      // partitioned: every rank formed P^T (its rows of K) P
  // Hermite-type spaces: the six unknowns of a node the surface barely touches are nearly dependent (a rotation about the
  // line through the few points that see the node moves nothing), the operator is semi-definite there up to rounding and
  // the order of the atomic sums decided whether a pivot came out positive -- a relative 1e-9 on the diagonal settles it
  // (the prototype used 1e-8; the iteration counts do not move)
  hipLaunchKernelGGL(k_pc_coarse_fix_diag, dim3(sgrid(N, 256)), dim3(256), 0, st, N, s->d_cs_A, s->hermite_on ? 1e-9 : 1e-13);
  FEMO_HIP_CHECK(hipGetLastError());
  auto t1 = now();
  for (int kb = 0; kb < nblk; ++kb) {
    if (kb == 0) hipLaunchKernelGGL(k_chol_diag, dim3(1), dim3(256), 0, st, N, kb, s->d_cs_A, s->d_cs_dinv, s->d_cs_info);
    const int m = nblk - kb - 1;
    if (m > 0) {
      hipLaunchKernelGGL(k_chol_panel, dim3(m), dim3(256), 0, st, N, kb, s->d_cs_A, s->d_cs_dinv);
      hipLaunchKernelGGL(k_chol_update, dim3(m * (m + 1) / 2), dim3(256), 0, st, N, kb, s->d_cs_A, s->d_cs_dinv, s->d_cs_info);   // also factorises tile kb + 1
    }
  }
  auto t2 = now();
  if (s->d_cs_T != nullptr && !femo_env_flag("FEMO_SHELL_TRINV_ROWS")) {
    for (int B = 1; B < nblk; B *= 2) {
      const int nb = (nblk + 2 * B - 1) / (2 * B);
      for (int phase = 0; phase < 2; ++phase)
        hipLaunchKernelGGL(k_trinv_level, dim3((unsigned)(nb * B * B)), dim3(256), 0, st, N, nblk, B, phase, s->d_cs_A, s->d_cs_dinv, s->d_cs_T);
    }
  } else {
    for (int i = 1; i < nblk; ++i) hipLaunchKernelGGL(k_trinv_row, dim3(i), dim3(256), 0, st, N, i, s->d_cs_A, s->d_cs_dinv);
  }
  hipLaunchKernelGGL(k_trinv_diag, dim3(nblk), dim3(256), 0, st, N, s->d_cs_A, s->d_cs_dinv);
  hipLaunchKernelGGL(k_pc_coarse_mirror, dim3(sgrid(N, 256), (unsigned)N), dim3(256), 0, st, N, s->d_cs_A);
  if (s->d_cs_Af != nullptr) hipLaunchKernelGGL(k_pc_coarse_to_float, dim3(2048), dim3(256), 0, st, N * N, s->d_cs_A, s->d_cs_Af, std::sqrt(shell_coarse_weight(s)));
  FEMO_HIP_CHECK(hipGetLastError());
  int32_t info[4] = {0, 0, 0, 0};
  FEMO_HIP_CHECK(hipMemcpyAsync(info, s->d_cs_info, sizeof info, hipMemcpyDeviceToHost, st));
  FEMO_HIP_CHECK(hipStreamSynchronize(st));
  s->cs_ready = info[1] == 0 && info[2] == 0;
  if (s->d_owned != nullptr && s->ctx->nranks > 1) {
    // the ranks must take the same branch (the collectives of the iteration depend on it): the coarse solve is used only
    // if no rank saw a failure (the factorisation is replicated, `far` is a property of the rank's cells)
    const double bad = s->cs_ready ? 0.0 : 1.0;
    double all = 0.0;
    FEMO_HIP_CHECK(hipMemcpyAsync(s->d_cs_tmp, &bad, sizeof bad, hipMemcpyHostToDevice, st));
    FEMO_TRY(shell_allreduce(s, s->d_cs_tmp, 1, st));
    FEMO_HIP_CHECK(hipMemcpyAsync(&all, s->d_cs_tmp, sizeof all, hipMemcpyDeviceToHost, st));
    FEMO_HIP_CHECK(hipStreamSynchronize(st));
    s->cs_ready = all == 0.0;
  }
  if (s->hermite_on && !s->cs_ready) {
    // said once per shell: bench and tests read the state back through femo_shell_pc_info
    fprintf(stderr, "[femo] warning: the Hermite-type coarse operator could not be factorised (far %d, pivot %d); this shell falls back to the trilinear hierarchy\n", info[1], info[2]);
    // the Hermite-type transfers need their own (composed) coarse operator: without it the trilinear hierarchy takes over,
    // whose dense operator is formed now (the preconditioner changes, the solution does not)
    s->hermite = false;
    return shell_pc_coarse_setup(s, vals, d_fixed);
  }
  if (dbg) {
    auto t3 = now();
    auto ms = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    fprintf(stderr, "[femo] coarse solve: level %d, %lld unknowns (padded %lld), %lld items; Galerkin %.2f ms, Cholesky %.2f ms, inverse %.2f ms; far %d, pivot %d\n",
            s->cs_level, (long long)n, (long long)N, (long long)s->cs_items, ms(t0, t1), ms(t1, t2), ms(t2, t3), info[1], info[2]);
  }
  return 0;
}

// z = M^-1 r (lattice preconditioner) and the per-block partials of r.z; enqueues 2 L + 1 launches
// Pte != nullptr: the fused form -- the finest level's up kernel also emits the partials of e . g into Pte (*nb_te blocks)
// and the prolongation is left to the caller (k_pc_prolong_fused).
static int shell_pc_apply(femo_shell* s, const uint8_t* d_fixed, double* Prz, unsigned gz, const int32_t* done,
                          double* Pte = nullptr, int* nb_te = nullptr, const ShellXCarry* carry = nullptr) {
  hipStream_t st = s->ctx->stream;
  const int L = s->pc_levels;
  const bool herm = s->hermite_on && s->cs_ready && s->blk_ready;
  auto level_up = [&](int l, const double* blocks) {
    const int64_t n0 = s->level_off[l], n1 = s->level_off[l + 1];
    const bool last = l == L - 1 && Pte != nullptr;
    if (herm) {
      const unsigned g = last ? std::min<unsigned>(sgrid((n1 - n0) * 2 * LAT_H_LANES, 256), 1024u) : sgrid((n1 - n0) * 2 * LAT_H_LANES, 256);
      if (last && nb_te) *nb_te = (int)g;
      hipLaunchKernelGGL(k_lat_level_h, dim3(g), dim3(256), 0, st, n0, n1, s->d_par_rowptr, s->d_par_cols, s->d_par_w5, s->d_t, s->d_e, 1, done, blocks,
                         last ? Pte : (double*)nullptr);
      return;
    }
    const unsigned g = last ? std::min<unsigned>(sgrid((n1 - n0) * 6, 256), 1024u) : sgrid((n1 - n0) * 6, 256);   // few partials: every block of the prolongation folds them
    if (last && nb_te) *nb_te = (int)g;
    hipLaunchKernelGGL(k_lat_level, dim3(g), dim3(256), 0, st, n0, n1, s->d_par_rowptr, s->d_par_cols, s->d_par_vals,
                       s->d_coarse, s->d_t, s->d_e, 1, done, blocks, last ? Pte : (double*)nullptr);
  };
  const int64_t m0 = s->level_off[L - 1], m1 = s->level_off[L];
  if (herm)
    hipLaunchKernelGGL(k_pc_restrict_h<32>, dim3(std::min<unsigned>(sgrid(m1 - m0, SH_BLOCK / 32), 1 << 16)), dim3(SH_BLOCK), 0, st, m0, m1,
                       s->d_hp_rowptr, s->d_hp_cols, s->d_hp_w4, s->d_r, s->d_t, done);
  else
  // 32 lanes per row (rows hold ~100 points; 8 / 16 / 32 / 64 lanes: 0.362 / 0.355 / 0.351 / 0.351 ms per iteration)
  hipLaunchKernelGGL(k_pc_restrict<32>, dim3(std::min<unsigned>(sgrid(2 * (m1 - m0), SH_BLOCK / 32), 1 << 16)), dim3(SH_BLOCK), 0, st, m0, m1,
                     s->d_ptp_rowptr, s->d_ptp_cols, s->d_ptp_vals, s->d_r, s->d_t, done);
  if (s->d_owned != nullptr && s->ctx->nranks > 1) {
    // partitioned: r is zero on the points owned elsewhere, so the sum over the ranks is P^T r; the lattice levels below
    // are replicated (same arithmetic on every rank)
    FEMO_HIP_CHECK(hipGetLastError());
    FEMO_TRY(shell_allreduce(s, s->d_t + 6 * m0, 6 * (m1 - m0), st));
  }
  if (s->cs_ready) {
    // levels above the coarse-solve level as before; on it the dense inverse replaces the diagonal levels 0 .. cs
    const int cs = s->cs_level;
    if (herm && s->d_hd_rowptr != nullptr && L - 1 > cs) {
      const int64_t rows = s->level_off[L - 1] - s->level_off[cs];
      hipLaunchKernelGGL(k_lat_down_composite_h, dim3(sgrid(rows, SH_BLOCK / 64)), dim3(SH_BLOCK), 0, st, s->level_off[cs], rows, s->d_hd_rowptr,
                         s->d_hd_cols, s->d_hd_w5, s->d_t, done);
    } else if (herm) {
      for (int l = L - 2; l >= cs; --l) {
        const int64_t n0 = s->level_off[l], n1 = s->level_off[l + 1];
        hipLaunchKernelGGL(k_lat_level_h, dim3(sgrid((n1 - n0) * 2 * LAT_H_LANES, 256)), dim3(256), 0, st, n0, n1, s->d_chi_rowptr, s->d_chi_cols, s->d_chi_w5, s->d_t, s->d_e, 0,
                           done, (const double*)nullptr, (double*)nullptr);
      }
    } else if (s->d_cd_rowptr != nullptr && L - 1 > cs) {
      const int64_t rows = s->level_off[L - 1] - s->level_off[cs];
      hipLaunchKernelGGL(k_lat_down_composite, dim3(sgrid(rows, SH_BLOCK / 64)), dim3(SH_BLOCK), 0, st, s->level_off[cs], rows, s->d_cd_rowptr,
                         s->d_cd_cols, s->d_cd_vals, s->d_t, done);
    } else {
      for (int l = L - 2; l >= cs; --l) {
        const int64_t n0 = s->level_off[l], n1 = s->level_off[l + 1];
        hipLaunchKernelGGL(k_lat_level, dim3(sgrid((n1 - n0) * 6, 256)), dim3(256), 0, st, n0, n1, s->d_chi_rowptr, s->d_chi_cols, s->d_chi_vals,
                           s->d_coarse, s->d_t, s->d_e, 0, done);
      }
    }
    const unsigned gp = (unsigned)((s->cs_n + 1) / 2);
    FEMO_REQUIRE(carry == nullptr || s->d_cs_Af != nullptr, "the carried x update rides in the single-precision coarse product");
    if (s->d_cs_Af != nullptr) {
      ShellXCarry xc = {nullptr, nullptr, nullptr, 0, 0};
      unsigned g1 = gp;
      if (carry != nullptr) { xc = *carry; xc.row_blocks = (int)gp; g1 = gp + 1024u; }
      hipLaunchKernelGGL(k_pc_coarse_apply<float>, dim3(g1), dim3(SH_BLOCK), 0, st, s->cs_n, s->cs_N, 1, (const float*)s->d_cs_Af,
                         s->d_t + 6 * s->level_off[cs], s->d_cs_tmp, done, xc);
      hipLaunchKernelGGL(k_pc_coarse_apply<float>, dim3(gp), dim3(SH_BLOCK), 0, st, s->cs_n, s->cs_N, 0, (const float*)s->d_cs_Af, s->d_cs_tmp,
                         s->d_e + 6 * s->level_off[cs], done);
    } else {
      hipLaunchKernelGGL(k_pc_coarse_apply<double>, dim3(gp), dim3(SH_BLOCK), 0, st, s->cs_n, s->cs_N, 1, (const double*)s->d_cs_A,
                         s->d_t + 6 * s->level_off[cs], s->d_cs_tmp, done);
      hipLaunchKernelGGL(k_pc_coarse_apply<double>, dim3(gp), dim3(SH_BLOCK), 0, st, s->cs_n, s->cs_N, 0, (const double*)s->d_cs_A, s->d_cs_tmp,
                         s->d_e + 6 * s->level_off[cs], done);
    }
    for (int l = cs + 1; l < L; ++l) level_up(l, s->blk_ready ? s->d_cblk : (const double*)nullptr);
  } else {
  // levels 0 .. kc (at most 256 nodes each, never the finest: with 4096 the one workgroup took 244 us, with 768 still 71) go through the fused single-workgroup kernel
    int kc = -1;
    while (kc + 1 < L - 1 && kc + 1 < 16 && s->level_off[kc + 2] - s->level_off[kc + 1] <= 256) ++kc;
    for (int l = L - 2; l > kc; --l) {
      const int64_t n0 = s->level_off[l], n1 = s->level_off[l + 1];
      hipLaunchKernelGGL(k_lat_level, dim3(sgrid((n1 - n0) * 6, 256)), dim3(256), 0, st, n0, n1, s->d_chi_rowptr, s->d_chi_cols, s->d_chi_vals,
                         s->d_coarse, s->d_t, s->d_e, 0, done);
    }
    if (kc >= 0) {
      LatLevels Lv;
      Lv.kc = kc;
      for (int l = 0; l <= kc + 1; ++l) Lv.off[l] = s->level_off[l];
      hipLaunchKernelGGL(k_lat_coarse_fused, dim3(1), dim3(1024), 0, st, Lv, s->d_chi_rowptr, s->d_chi_cols, s->d_chi_vals, s->d_par_rowptr,
                         s->d_par_cols, s->d_par_vals, s->d_coarse, s->d_t, s->d_e, done);
    }
    for (int l = kc + 1; l < L; ++l) level_up(l, (const double*)nullptr);
  }
  if (Pte != nullptr) { FEMO_HIP_CHECK(hipGetLastError()); return 0; }
  hipLaunchKernelGGL(k_pc_prolong, dim3(gz), dim3(SH_BLOCK), 0, st, s->n_dof / 3, s->d_fin_idx, s->d_fin_w, d_fixed, s->d_dinv, s->d_r, s->d_e,
                     s->d_z, Prz, done, s->dinv3_ready ? s->d_dinv3 : (const float*)nullptr, herm ? s->d_fin_w4 : (const float4*)nullptr, s->n_unode);
  FEMO_HIP_CHECK(hipGetLastError());
  return 0;
}

int femo_shell_pc_coarse(femo_shell* s, int level, const int32_t* node_xyz, int64_t n_items, const int64_t* item_ptr,
                         const int32_t* item_pts, const int32_t* item_nbr, const int64_t* down_rowptr, const int32_t* down_cols,
                         const double* down_vals) {
  FEMO_REQUIRE(s && node_xyz && item_ptr && item_pts && item_nbr, "null argument");
  FEMO_REQUIRE(s->pc_width > 0, "femo_shell_pc_coarse needs femo_shell_pc_create first");
  FEMO_REQUIRE(level >= 0 && level < s->pc_levels - 1 && n_items > 0 && s->cs_level < 0, "bad coarse-solve level");
  FEMO_REQUIRE(s->d_brow != nullptr, "the coarse solve needs the node-block view of the pattern");
  hipStream_t st = s->ctx->stream;
  FEMO_HIP_CHECK(hipSetDevice(s->ctx->device));
  const int64_t nodes = s->level_off[level + 1] - s->level_off[level];
  const int64_t n = 6 * nodes;
  FEMO_REQUIRE(n <= 8192, "coarse-solve level too large for a dense inverse");
  s->cs_max_item = 0;
  for (int64_t it = 0; it < n_items; ++it) {
    FEMO_REQUIRE(item_ptr[it + 1] > item_ptr[it], "empty Galerkin item");
    s->cs_max_item = std::max<int64_t>(s->cs_max_item, item_ptr[it + 1] - item_ptr[it]);
  }
  FEMO_TRY(to_device(&s->d_cs_xyz, node_xyz, 3 * nodes, st));
  FEMO_TRY(to_device(&s->d_cs_ptr, item_ptr, n_items + 1, st));
  FEMO_TRY(to_device(&s->d_cs_pts, item_pts, item_ptr[n_items], st));
  FEMO_TRY(to_device(&s->d_cs_nbr, item_nbr, n_items * 64, st));
  const int64_t N = (n + DT - 1) / DT * DT;
  FEMO_HIP_CHECK(hipMalloc(&s->d_cs_A, N * N * sizeof(double)));
  if (!femo_env_flag("FEMO_SHELL_COARSE_FP64")) FEMO_HIP_CHECK(hipMalloc(&s->d_cs_Af, N * N * sizeof(float)));
  FEMO_HIP_CHECK(hipMalloc(&s->d_cs_tmp, N * sizeof(double)));
  FEMO_HIP_CHECK(hipMalloc(&s->d_cs_dinv, N * DT * sizeof(double)));
  FEMO_HIP_CHECK(hipMalloc(&s->d_cs_T, N * N * sizeof(double)));
  FEMO_HIP_CHECK(hipMalloc(&s->d_cs_info, 4 * sizeof(int32_t)));
  FEMO_REQUIRE(s->level_off[level + 1] - s->level_off[level] > 0 && (1 << 10) > (1 << (level + 1)), "lattice too fine for 10-bit coordinates");
  FEMO_HIP_CHECK(hipMalloc(&s->d_cs_pcell, (s->n_dof / 3) * sizeof(int32_t)));
  hipLaunchKernelGGL(k_pc_coarse_cells, dim3(sgrid(s->n_dof / 3, 256)), dim3(256), 0, st, s->n_dof / 3, level, s->pc_width, s->level_off[level],
                     s->d_ell_idx, s->d_cs_xyz, s->d_cs_pcell);
  FEMO_HIP_CHECK(hipGetLastError());
  FEMO_HIP_CHECK(hipStreamSynchronize(st));
  if (down_rowptr != nullptr && down_cols != nullptr && down_vals != nullptr) {
    // rows: the nodes of levels `level` .. L - 2 in their global order, columns: global node numbers of the finest lattice
    const int64_t rows = s->level_off[s->pc_levels - 1] - s->level_off[level];
    FEMO_TRY(to_device(&s->d_cd_rowptr, down_rowptr, rows + 1, st));
    FEMO_TRY(to_device(&s->d_cd_cols, down_cols, down_rowptr[rows], st));
    FEMO_TRY(to_device(&s->d_cd_vals, down_vals, down_rowptr[rows], st));
    FEMO_HIP_CHECK(hipStreamSynchronize(st));
  }
  {
    const int first_slot = 8 * (level + 1);
    const int64_t n_pts = s->n_dof / 3, cnt = n_pts * ((s->pc_width - first_slot) / 8) * 8;
    if (cnt > 0) {
      FEMO_HIP_CHECK(hipMalloc(&s->d_lvl_node, cnt * sizeof(int32_t)));
      FEMO_HIP_CHECK(hipMalloc(&s->d_lvl_w, cnt * sizeof(double)));
      hipLaunchKernelGGL(k_compact_levels, dim3(sgrid(cnt, 256)), dim3(256), 0, st, n_pts, s->pc_width, first_slot, s->d_ell_idx, s->d_ell_w,
                         s->d_lvl_node, s->d_lvl_w);
      FEMO_HIP_CHECK(hipGetLastError());
    }
  }
  FEMO_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_pc_coarse_galerkin), hipFuncAttributeMaxDynamicSharedMemorySize, (int)CG_LDS));
  s->cs_level = level; s->cs_n = n; s->cs_N = N; s->cs_items = n_items;
  s->pc_vals_uid = 0; s->pc_vals_gen = 0;                 // next solve recomputes the preconditioner's numbers
  return 0;
}

// Items of the node-block set-up kernel for the Hermite-type spaces (fea/shell.py::node_block_items): the points of every
// level above the coarse solve grouped by lattice cell, at most 64 per item; pcell[level][point] = packed cell coordinates.
int femo_shell_pc_block_items(femo_shell* s, int64_t n_items, const int64_t* item_ptr, const int32_t* item_lvl, const int32_t* item_pts,
                              const int32_t* pcell) {
  FEMO_REQUIRE(s && item_ptr && item_lvl && item_pts && pcell && n_items > 0, "null argument");
  FEMO_REQUIRE(s->hermite, "femo_shell_pc_block_items needs femo_shell_pc_hermite first");
  FEMO_REQUIRE(s->d_bi_ptr == nullptr, "the shell already has its node-block items");
  hipStream_t st = s->ctx->stream;
  FEMO_HIP_CHECK(hipSetDevice(s->ctx->device));
  const int64_t n_pts = s->n_dof / 3;
  const int n_above = s->pc_levels - 1 - s->cs_level;
  for (int64_t i = 0; i < n_items; ++i) {
    FEMO_REQUIRE(item_lvl[i] >= 0 && item_lvl[i] < n_above, "item level out of range");
    FEMO_REQUIRE(item_ptr[i + 1] > item_ptr[i] && item_ptr[i + 1] - item_ptr[i] <= 64, "an item holds 1 .. 64 points");
  }
  FEMO_REQUIRE(item_ptr[0] == 0 && item_ptr[n_items] == (int64_t)n_above * n_pts, "every point belongs to one item per level");
  FEMO_TRY(to_device(&s->d_bi_ptr, item_ptr, n_items + 1, st));
  FEMO_TRY(to_device(&s->d_bi_lvl, item_lvl, n_items, st));
  FEMO_TRY(to_device(&s->d_bi_pts, item_pts, item_ptr[n_items], st));
  FEMO_TRY(to_device(&s->d_bi_pcell, pcell, (int64_t)n_above * n_pts, st));
  FEMO_HIP_CHECK(hipMalloc(&s->d_fixbits, n_pts));
  FEMO_HIP_CHECK(hipStreamSynchronize(st));
  s->bi_items = n_items;
  s->pc_vals_uid = 0; s->pc_vals_gen = 0;
  return 0;
}

int femo_shell_pc_weights(femo_shell* s, double w_levels, double w_coarse) {
  FEMO_REQUIRE(s != nullptr, "null argument");
  FEMO_REQUIRE(w_levels > 0.0 && w_coarse > 0.0, "the weights of the preconditioner's parts must be positive (M^-1 stays positive definite)");
  s->w_levels = w_levels; s->w_coarse = w_coarse;
  s->pc_vals_uid = 0; s->pc_vals_gen = 0;                 // next solve recomputes the preconditioner's numbers
  return 0;
}

// Hermite-type lattice spaces on the hierarchy of femo_shell_pc_create / femo_shell_pc_coarse (fea/shell.py::hermite_lattice).
int femo_shell_pc_hermite(femo_shell* s, const float* fin_w4, const int64_t* hp_rowptr, const int32_t* hp_cols, const float* hp_w4,
                          const double* par_w5, const double* chi_w5, const float* lvl_w4, const float* cs_w4,
                          const int64_t* down_rowptr, const int32_t* down_cols, const double* down_w5) {
  FEMO_REQUIRE(s && fin_w4 && hp_rowptr && hp_cols && hp_w4 && par_w5 && chi_w5 && lvl_w4 && cs_w4, "null argument");
  FEMO_REQUIRE(s->pc_width > 0 && s->cs_level >= 0, "femo_shell_pc_hermite needs femo_shell_pc_create and femo_shell_pc_coarse first");
  FEMO_REQUIRE(!s->hermite_loaded, "the shell already has its Hermite-type lattice data");
  hipStream_t st = s->ctx->stream;
  FEMO_HIP_CHECK(hipSetDevice(s->ctx->device));
  const int64_t n_pts = s->n_dof / 3;
  const int L = s->pc_levels;
  const int64_t m0 = s->level_off[L - 1], m1 = s->level_off[L];
  std::vector<int64_t> h_par((size_t)s->pc_nodes + 1), h_chi((size_t)s->pc_nodes + 1);
  FEMO_HIP_CHECK(hipMemcpy(h_par.data(), s->d_par_rowptr, (s->pc_nodes + 1) * sizeof(int64_t), hipMemcpyDeviceToHost));
  FEMO_HIP_CHECK(hipMemcpy(h_chi.data(), s->d_chi_rowptr, (s->pc_nodes + 1) * sizeof(int64_t), hipMemcpyDeviceToHost));
  FEMO_TRY(to_device(&s->d_fin_w4, reinterpret_cast<const float4*>(fin_w4), n_pts * 8, st));
  FEMO_TRY(to_device(&s->d_hp_rowptr, hp_rowptr, 2 * (m1 - m0) + 1, st));
  FEMO_TRY(to_device(&s->d_hp_cols, hp_cols, hp_rowptr[2 * (m1 - m0)], st));
  FEMO_TRY(to_device(&s->d_hp_w4, reinterpret_cast<const float4*>(hp_w4), hp_rowptr[2 * (m1 - m0)], st));
  FEMO_TRY(to_device(&s->d_par_w5, par_w5, 5 * h_par[(size_t)s->pc_nodes], st));
  FEMO_TRY(to_device(&s->d_chi_w5, chi_w5, 5 * h_chi[(size_t)s->pc_nodes], st));
  const int n_above = L - 1 - s->cs_level;                      // levels cs + 1 .. L - 1
  FEMO_TRY(to_device(&s->d_lvl_w4, reinterpret_cast<const float4*>(lvl_w4), (int64_t)n_above * n_pts * 8, st));
  FEMO_TRY(to_device(&s->d_cs_w4, reinterpret_cast<const float4*>(cs_w4), n_pts * 8, st));
  if (down_rowptr != nullptr && down_cols != nullptr && down_w5 != nullptr) {
    const int64_t rows = s->level_off[L - 1] - s->level_off[s->cs_level];
    FEMO_TRY(to_device(&s->d_hd_rowptr, down_rowptr, rows + 1, st));
    FEMO_TRY(to_device(&s->d_hd_cols, down_cols, down_rowptr[rows], st));
    FEMO_TRY(to_device(&s->d_hd_w5, down_w5, 5 * down_rowptr[rows], st));
  }
  FEMO_HIP_CHECK(hipStreamSynchronize(st));
  FEMO_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_pc_coarse_galerkin_h), hipFuncAttributeMaxDynamicSharedMemorySize, (int)CGH_LDS));
  FEMO_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_pc_coarse_galerkin_mm), hipFuncAttributeMaxDynamicSharedMemorySize, (int)MM_LDS));
  s->hermite = true;
  s->hermite_loaded = true;
  s->pc_vals_uid = 0; s->pc_vals_gen = 0;
  return 0;
}

// State of the lattice preconditioner as the DEVICE side has it: out = {Hermite-type data uploaded, bit 0: Hermite-type spaces
// enabled (uploaded and not fallen back) | bit 1: in use for the stiffness last set up, coarse solve factorised, node blocks ready}.  After a failed factorisation of the Hermite-type
// coarse operator the library falls back to the trilinear hierarchy for good; callers that report or pin iteration counts
// read the state here instead of remembering what they asked for.
int femo_shell_pc_info(const femo_shell* s, int32_t out[4]) {
  FEMO_REQUIRE(s && out, "null argument");
  out[0] = s->hermite_loaded ? 1 : 0;
  out[1] = (s->hermite ? 1 : 0) | ((s->hermite && s->hermite_on && s->cs_ready && s->blk_ready) ? 2 : 0);
  out[2] = s->cs_ready ? 1 : 0;
  out[3] = s->blk_ready ? 1 : 0;
  return 0;
}

// For tests: the dense coarse operator (inverse = 0) or the factors of its inverse (1: L^-T above, L^-1 below the
// diagonal) for `vals` and the mask, row-major n x n on the host; the unknown count through *n_out.
int femo_shell_pc_coarse_matrix(femo_shell* s, const femo_vec* vals, const uint8_t* fixed_host, int inverse, double* out, int64_t* n_out) {
  FEMO_REQUIRE(s && vals && n_out, "null argument");
  FEMO_REQUIRE(s->cs_level >= 0, "no coarse solve on this shell");
  hipStream_t st = s->ctx->stream;
  *n_out = s->cs_n;
  if (out == nullptr) return 0;
  uint8_t* d_fixed = nullptr;
  if (fixed_host != nullptr) FEMO_TRY(to_device(&d_fixed, fixed_host, s->n_dof, st));
  const int64_t n = s->cs_n, N = s->cs_N;
  if (inverse) {
    FEMO_TRY(shell_pc_coarse_setup(s, vals, d_fixed));
    FEMO_REQUIRE(s->cs_ready, "the coarse operator could not be factorised");
  } else {
    FEMO_HIP_CHECK(hipMemsetAsync(s->d_cs_A, 0, N * N * sizeof(double), st));
    FEMO_HIP_CHECK(hipMemsetAsync(s->d_cs_info, 0, 4 * sizeof(int32_t), st));
    if (s->hermite && !femo_env_flag("FEMO_SHELL_TRILINEAR")) {
      for (int pass = 0; pass < 2; ++pass)
        hipLaunchKernelGGL(k_pc_coarse_galerkin_h, dim3((unsigned)s->cs_items), dim3(256), CGH_LDS, st, pass, s->cs_level, s->pc_width, s->level_off[s->cs_level], N,
                           s->n_unode, s->d_cs_ptr, s->d_cs_pts, s->d_cs_nbr, s->d_cs_pcell, s->d_brow, s->d_bcols, vals->d, d_fixed, s->d_ell_idx,
                           s->d_cs_w4, s->d_cs_A, s->d_cs_info);
      hipLaunchKernelGGL(k_pc_coarse_mirror_tu, dim3(sgrid(n, 256), (unsigned)n), dim3(256), 0, st, n, N, s->d_cs_A);
    } else {
      hipLaunchKernelGGL(k_pc_coarse_galerkin, dim3((unsigned)s->cs_items), dim3(256), CG_LDS, st, s->cs_level, s->pc_width, s->level_off[s->cs_level], N,
                         s->n_unode, s->d_cs_ptr, s->d_cs_pts, s->d_cs_nbr, s->d_cs_xyz, s->d_cs_pcell, s->d_brow, s->d_bcols, vals->d, d_fixed, s->d_ell_idx,
                         s->d_ell_w, s->d_cs_A, s->d_cs_info);
    }
    FEMO_HIP_CHECK(hipGetLastError());
    s->cs_ready = false;
  }
  s->pc_vals_uid = 0; s->pc_vals_gen = 0;
  FEMO_HIP_CHECK(hipMemcpy2DAsync(out, n * sizeof(double), s->d_cs_A, N * sizeof(double), n * sizeof(double), n, hipMemcpyDeviceToHost, st));
  FEMO_HIP_CHECK(hipStreamSynchronize(st));
  if (d_fixed) (void)hipFree(d_fixed);
  return 0;
}

int64_t femo_shell_ndof(const femo_shell* s) { return s ? s->n_dof : -1; }
int64_t femo_shell_nnz(const femo_shell* s) { return s ? s->nnz : -1; }

int femo_shell_assemble(femo_shell* s, double E, double nu, const femo_vec* h, femo_vec* vals) {
  FEMO_REQUIRE(s && h && vals, "null argument");
  FEMO_REQUIRE(h->n >= s->n_vert && vals->n >= s->nnz, "vector size mismatch in shell_assemble");
  FEMO_REQUIRE(E > 0.0 && nu > -1.0 && nu < 0.5, "bad material");
  hipStream_t st = s->ctx->stream;
  femo_vec_touch(vals);
  FEMO_HIP_CHECK(hipMemsetAsync(vals->d, 0, s->nnz * sizeof(double), st));
  hipLaunchKernelGGL(k_shell_assemble, dim3(sgrid(s->n_cell * 27)), dim3(SH_BLOCK), 0, st, view(s), E, nu, h->d, s->d_epos, vals->d);
  FEMO_HIP_CHECK(hipGetLastError());
  FEMO_TRY(shell_zero_unowned_rows(s, vals->d, st));       // partitioned: the rank's share of K (its points' rows are complete)
  return 0;
}

int femo_shell_matvec(femo_shell* s, const femo_vec* vals, const uint8_t* fixed_dev_or_null, const femo_vec* x, femo_vec* y) {
  FEMO_REQUIRE(s && vals && x && y, "null argument");
  FEMO_REQUIRE(vals->n >= s->nnz && x->n >= s->n_dof && y->n >= s->n_dof && x->d != y->d, "vector size mismatch in shell_matvec");
  femo_vec_touch(y);
  hipStream_t st = s->ctx->stream;
  if (s->d_brow != nullptr && fixed_dev_or_null == nullptr) {
    hipLaunchKernelGGL(k_bcsr3_spmv<16>, dim3(std::min<unsigned>(sgrid(s->n_bnode, SH_BLOCK / 16), SH_MAXPART)), dim3(SH_BLOCK), 0, st, s->n_bnode,
                       s->d_brow, s->d_bcols, vals->d, (const uint8_t*)nullptr, x->d, y->d, (double*)nullptr, (const int32_t*)nullptr,
                       (double*)nullptr, (const double*)nullptr);
  } else {
    hipLaunchKernelGGL(k_csr_spmv, dim3(sgrid(s->n_dof, SH_BLOCK / 16)), dim3(SH_BLOCK), 0, st, s->n_dof, s->d_rowptr, s->d_cols,
                       vals->d, fixed_dev_or_null, 1, x->d, y->d, (double*)nullptr, (const int32_t*)nullptr);
  }
  FEMO_HIP_CHECK(hipGetLastError());
  return 0;
}

int femo_shell_load(femo_shell* s, const femo_vec* f, double sign, int accumulate, femo_vec* F) {
  FEMO_REQUIRE(s && f && F, "null argument");
  FEMO_REQUIRE(f->n >= 3 * s->n_vert && F->n >= s->n_dof, "vector size mismatch in shell_load");
  hipStream_t st = s->ctx->stream;
  femo_vec_touch(F);
  if (!accumulate) FEMO_HIP_CHECK(hipMemsetAsync(F->d, 0, s->n_dof * sizeof(double), st));
  hipLaunchKernelGGL(k_shell_load, dim3(sgrid(s->n_cell)), dim3(SH_BLOCK), 0, st, view(s), f->d, sign, F->d);
  FEMO_HIP_CHECK(hipGetLastError());
  return 0;
}

int femo_shell_load_T(femo_shell* s, const femo_vec* lam, double sign, int accumulate, femo_vec* out) {
  FEMO_REQUIRE(s && lam && out, "null argument");
  FEMO_REQUIRE(lam->n >= s->n_dof && out->n >= 3 * s->n_vert, "vector size mismatch in shell_load_T");
  hipStream_t st = s->ctx->stream;
  femo_vec_touch(out);
  if (!accumulate) FEMO_HIP_CHECK(hipMemsetAsync(out->d, 0, 3 * s->n_vert * sizeof(double), st));
  hipLaunchKernelGGL(k_shell_load_T, dim3(sgrid(s->n_cell)), dim3(SH_BLOCK), 0, st, view(s), lam->d, sign, out->d);
  FEMO_HIP_CHECK(hipGetLastError());
  return 0;
}

// out_b (+)= sign * v^T (dK/dh_b) w;  energy (optional) = 1/2 v^T K(h) w
int femo_shell_dform_dh(femo_shell* s, double E, double nu, const femo_vec* h, const femo_vec* v, const femo_vec* w,
                        int accumulate, femo_vec* out, double* energy) {
  FEMO_REQUIRE(s && h && v && w, "null argument");
  FEMO_REQUIRE(h->n >= s->n_vert && v->n >= s->n_dof && w->n >= s->n_dof, "vector size mismatch in shell_dform_dh");
  FEMO_REQUIRE(out == nullptr || out->n >= s->n_vert, "output shorter than n_vert");
  hipStream_t st = s->ctx->stream;
  const unsigned g = sgrid(s->n_cell);
  FEMO_REQUIRE(energy == nullptr || g <= 3 * SH_MAXPART, "mesh too large for the energy reduction buffer");
  if (out) {
    femo_vec_touch(out);
    if (!accumulate) FEMO_HIP_CHECK(hipMemsetAsync(out->d, 0, s->n_vert * sizeof(double), st));
  }
  hipLaunchKernelGGL(k_shell_dform_dh, dim3(g), dim3(SH_BLOCK), 0, st, view(s), E, nu, h->d, v->d, w->d, out ? out->d : nullptr,
                     energy ? s->d_part : nullptr);
  FEMO_HIP_CHECK(hipGetLastError());
  if (energy) FEMO_TRY(reduce_partials(s->ctx, s->d_part, (int)g, energy));
  return 0;
}

// y (+)= (dK/dh [dh]) w: forward product with the thickness partial of the elastic residual
int femo_shell_dform_dh_fwd(femo_shell* s, double E, double nu, const femo_vec* h, const femo_vec* dh, const femo_vec* w, int accumulate, femo_vec* y) {
  FEMO_REQUIRE(s && h && dh && w && y, "null argument");
  FEMO_REQUIRE(h->n >= s->n_vert && dh->n >= s->n_vert && w->n >= s->n_dof && y->n >= s->n_dof && w->d != y->d, "vector size mismatch in shell_dform_dh_fwd");
  hipStream_t st = s->ctx->stream;
  femo_vec_touch(y);
  if (!accumulate) FEMO_HIP_CHECK(hipMemsetAsync(y->d, 0, s->n_dof * sizeof(double), st));
  hipLaunchKernelGGL(k_shell_dform_dh_fwd, dim3(sgrid(s->n_cell)), dim3(SH_BLOCK), 0, st, view(s), E, nu, h->d, dh->d, w->d, y->d);
  FEMO_HIP_CHECK(hipGetLastError());
  return 0;
}

int femo_shell_compliance(femo_shell* s, const femo_vec* w, double* value, int accumulate, femo_vec* grad) {
  return femo_shell_compliance_dx(s, w, nullptr, value, accumulate, grad);
}

int femo_shell_compliance_dx(femo_shell* s, const femo_vec* w, const femo_vec* cell_weight, double* value, int accumulate, femo_vec* grad) {
  FEMO_REQUIRE(s && w, "null argument");
  FEMO_REQUIRE(w->n >= s->n_dof && (grad == nullptr || grad->n >= s->n_dof), "vector size mismatch in shell_compliance");
  FEMO_REQUIRE(cell_weight == nullptr || cell_weight->n >= s->n_cell, "cell weights shorter than n_cell");
  hipStream_t st = s->ctx->stream;
  const unsigned g = sgrid(s->n_cell);
  FEMO_REQUIRE(value == nullptr || g <= 3 * SH_MAXPART, "mesh too large for the reduction buffer");
  if (grad) {
    femo_vec_touch(grad);
    if (!accumulate) FEMO_HIP_CHECK(hipMemsetAsync(grad->d, 0, s->n_dof * sizeof(double), st));
  }
  hipLaunchKernelGGL(k_shell_compliance, dim3(g), dim3(SH_BLOCK), 0, st, view(s), w->d, cell_weight ? cell_weight->d : (const double*)nullptr,
                     value ? s->d_part : nullptr, grad ? grad->d : nullptr);
  FEMO_HIP_CHECK(hipGetLastError());
  if (value) FEMO_TRY(reduce_partials(s->ctx, s->d_part, (int)g, value));
  return 0;
}

int femo_shell_pnorm_stress(femo_shell* s, double E, double nu, const femo_vec* h, const femo_vec* w, double m, double rho, double alpha,
                            double surface, double* value, int accumulate, femo_vec* grad_w, femo_vec* grad_h) {
  FEMO_REQUIRE(s && h && w, "null argument");
  FEMO_REQUIRE(h->n >= s->n_vert && w->n >= s->n_dof && (grad_w == nullptr || grad_w->n >= s->n_dof) &&
               (grad_h == nullptr || grad_h->n >= s->n_vert), "vector size mismatch in shell_pnorm_stress");
  FEMO_REQUIRE(E > 0.0 && nu > -1.0 && nu < 0.5 && m > 0.0 && rho >= 1.0 && alpha > 0.0, "bad parameters of the stress aggregate");
  hipStream_t st = s->ctx->stream;
  const unsigned g = sgrid(s->n_cell);
  FEMO_REQUIRE(value == nullptr || g <= 3 * SH_MAXPART, "mesh too large for the reduction buffer");
  if (grad_w) {
    femo_vec_touch(grad_w);
    if (!accumulate) FEMO_HIP_CHECK(hipMemsetAsync(grad_w->d, 0, s->n_dof * sizeof(double), st));
  }
  if (grad_h) {
    femo_vec_touch(grad_h);
    if (!accumulate) FEMO_HIP_CHECK(hipMemsetAsync(grad_h->d, 0, s->n_vert * sizeof(double), st));
  }
  hipLaunchKernelGGL(k_shell_pnorm_stress, dim3(g), dim3(SH_BLOCK), 0, st, view(s), E, nu, h->d, w->d, m, rho, 1.0 / alpha, surface,
                     value ? s->d_part : nullptr, grad_w ? grad_w->d : nullptr, grad_h ? grad_h->d : nullptr);
  FEMO_HIP_CHECK(hipGetLastError());
  if (value) FEMO_TRY(reduce_partials(s->ctx, s->d_part, (int)g, value));
  return 0;
}

int femo_shell_vm_rhs(femo_shell* s, double E, double nu, const femo_vec* h, const femo_vec* w, double surface, femo_vec* rhs,
                      femo_vec* lumped) {
  FEMO_REQUIRE(s && h && w && rhs, "null argument");
  FEMO_REQUIRE(h->n >= s->n_vert && w->n >= s->n_dof && rhs->n >= s->n_vert && (lumped == nullptr || lumped->n >= s->n_vert),
               "vector size mismatch in shell_vm_rhs");
  FEMO_REQUIRE(E > 0.0 && nu > -1.0 && nu < 0.5, "bad material");
  hipStream_t st = s->ctx->stream;
  femo_vec_touch(rhs);
  FEMO_HIP_CHECK(hipMemsetAsync(rhs->d, 0, s->n_vert * sizeof(double), st));
  if (lumped) {
    femo_vec_touch(lumped);
    FEMO_HIP_CHECK(hipMemsetAsync(lumped->d, 0, s->n_vert * sizeof(double), st));
  }
  hipLaunchKernelGGL(k_shell_vm_rhs, dim3(sgrid(s->n_cell)), dim3(SH_BLOCK), 0, st, view(s), E, nu, h->d, w->d, surface, rhs->d,
                     lumped ? lumped->d : nullptr);
  FEMO_HIP_CHECK(hipGetLastError());
  return 0;
}

int femo_shell_p1_mass(femo_shell* s, const femo_vec* x, femo_vec* y) {
  FEMO_REQUIRE(s && x && y, "null argument");
  FEMO_REQUIRE(x->n >= s->n_vert && y->n >= s->n_vert && x->d != y->d, "vector size mismatch in shell_p1_mass");
  hipStream_t st = s->ctx->stream;
  femo_vec_touch(y);
  FEMO_HIP_CHECK(hipMemsetAsync(y->d, 0, s->n_vert * sizeof(double), st));
  hipLaunchKernelGGL(k_shell_p1_mass, dim3(sgrid(s->n_cell)), dim3(SH_BLOCK), 0, st, view(s), x->d, y->d);
  FEMO_HIP_CHECK(hipGetLastError());
  return 0;
}

int femo_shell_mass(femo_shell* s, double rho, const femo_vec* h, double* value, int accumulate, femo_vec* grad) {
  FEMO_REQUIRE(s && h, "null argument");
  FEMO_REQUIRE(h->n >= s->n_vert && (grad == nullptr || grad->n >= s->n_vert), "vector size mismatch in shell_mass");
  hipStream_t st = s->ctx->stream;
  const unsigned g = sgrid(s->n_cell);
  FEMO_REQUIRE(value == nullptr || g <= 3 * SH_MAXPART, "mesh too large for the reduction buffer");
  if (grad) {
    femo_vec_touch(grad);
    if (!accumulate) FEMO_HIP_CHECK(hipMemsetAsync(grad->d, 0, s->n_vert * sizeof(double), st));
  }
  hipLaunchKernelGGL(k_shell_mass, dim3(g), dim3(SH_BLOCK), 0, st, view(s), rho, h->d, value ? s->d_part : nullptr, grad ? grad->d : nullptr);
  FEMO_HIP_CHECK(hipGetLastError());
  if (value) FEMO_TRY(reduce_partials(s->ctx, s->d_part, (int)g, value));
  return 0;
}

// ---- penalty boundary terms: K_pen = sum over tagged edges of coef_e x (edge mass matrices), all six fields ----
int femo_shell_set_penalty(femo_shell* s, int64_t n_edges, const int32_t* edge_nodes, const double* coef, const int32_t* pos) {
  FEMO_REQUIRE(s != nullptr && n_edges >= 0, "bad argument");
  FEMO_REQUIRE(n_edges == 0 || (edge_nodes && coef && pos), "null argument");
  hipStream_t st = s->ctx->stream;
  FEMO_HIP_CHECK(hipSetDevice(s->ctx->device));
  FEMO_HIP_CHECK(hipStreamSynchronize(st));
  hipFree(s->d_pen_nodes); hipFree(s->d_pen_pos); hipFree(s->d_pen_coef);
  s->d_pen_nodes = s->d_pen_pos = nullptr; s->d_pen_coef = nullptr; s->pen_n = 0;
  if (n_edges == 0) return 0;
  for (int64_t e = 0; e < n_edges; ++e) {
    FEMO_REQUIRE(edge_nodes[3 * e] >= 0 && edge_nodes[3 * e] < s->n_vert && edge_nodes[3 * e + 1] >= 0 && edge_nodes[3 * e + 1] < s->n_vert &&
                 edge_nodes[3 * e + 2] >= s->n_vert && edge_nodes[3 * e + 2] < s->n_unode, "penalty edge %lld: bad node numbers", (long long)e);
    for (int k = 0; k < 39; ++k) FEMO_REQUIRE(pos[39 * e + k] >= 0 && pos[39 * e + k] < s->nnz, "penalty edge %lld: position outside the pattern", (long long)e);
  }
  FEMO_TRY(to_device(&s->d_pen_nodes, edge_nodes, 3 * n_edges, st));
  FEMO_TRY(to_device(&s->d_pen_pos, pos, 39 * n_edges, st));
  FEMO_TRY(to_device(&s->d_pen_coef, coef, n_edges, st));
  FEMO_HIP_CHECK(hipStreamSynchronize(st));
  s->pen_n = n_edges;
  return 0;
}

int femo_shell_penalty_add(femo_shell* s, femo_vec* vals) {
  FEMO_REQUIRE(s && vals, "null argument");
  FEMO_REQUIRE(vals->n >= s->nnz, "vector size mismatch in shell_penalty_add");
  if (s->pen_n == 0) return 0;
  femo_vec_touch(vals);
  hipLaunchKernelGGL(k_shell_penalty_add, dim3(sgrid(s->pen_n * 39, 256)), dim3(256), 0, s->ctx->stream, s->pen_n, s->d_pen_pos, s->d_pen_coef, vals->d);
  FEMO_HIP_CHECK(hipGetLastError());
  FEMO_TRY(shell_zero_unowned_rows(s, vals->d, s->ctx->stream));
  return 0;
}

// Partition of a shell over the ranks of the context.  The handle was created on the rank's cells: all
// cells that touch a point it owns (points: P2 nodes and rotation vertices, dofs 3 p .. 3 p + 2).  owned_points flags
// them (n_dof / 3 bytes); segment k of send_dofs lists the dofs whose values rank nbr[k] needs, segment k of recv_dofs the
// dofs that receive rank nbr[k]'s values in the same order.
int femo_shell_set_partition(femo_shell* s, const uint8_t* owned_points, int n_nbr, const int32_t* nbr, const int64_t* send_ptr,
                             const int32_t* send_dofs, const int64_t* recv_ptr, const int32_t* recv_dofs) {
  FEMO_REQUIRE(s && owned_points, "null argument");
  FEMO_REQUIRE(n_nbr >= 0 && (n_nbr == 0 || (nbr && send_ptr && send_dofs && recv_ptr && recv_dofs)), "bad halo plan");
  FEMO_REQUIRE(s->d_owned == nullptr, "the shell already has a partition");
  FEMO_REQUIRE(s->d_brow != nullptr && s->n_dof % 3 == 0, "a partitioned shell needs the node-block view of the pattern");
  hipStream_t st = s->ctx->stream;
  FEMO_HIP_CHECK(hipSetDevice(s->ctx->device));
  const int64_t n_pts = s->n_dof / 3;
  for (int k = 0; k < n_nbr; ++k) {
    FEMO_REQUIRE(nbr[k] >= 0 && nbr[k] < s->ctx->nranks && nbr[k] != s->ctx->rank, "bad neighbour rank %d", nbr[k]);
    FEMO_REQUIRE(send_ptr[k + 1] >= send_ptr[k] && recv_ptr[k + 1] >= recv_ptr[k], "halo segments not ordered");
  }
  const int64_t ns = n_nbr ? send_ptr[n_nbr] : 0, nr = n_nbr ? recv_ptr[n_nbr] : 0;
  for (int64_t i = 0; i < ns; ++i) FEMO_REQUIRE(send_dofs[i] >= 0 && send_dofs[i] < s->n_dof && owned_points[send_dofs[i] / 3], "a rank sends a dof it does not own");
  for (int64_t i = 0; i < nr; ++i) FEMO_REQUIRE(recv_dofs[i] >= 0 && recv_dofs[i] < s->n_dof && !owned_points[recv_dofs[i] / 3], "a rank receives a dof it owns");
  FEMO_TRY(to_device(&s->d_owned, owned_points, n_pts, st));
  s->n_nbr = n_nbr;
  if (n_nbr > 0) {
    s->nbr.assign(nbr, nbr + n_nbr);
    s->send_ptr.assign(send_ptr, send_ptr + n_nbr + 1);
    s->recv_ptr.assign(recv_ptr, recv_ptr + n_nbr + 1);
    FEMO_TRY(to_device(&s->d_send_idx, send_dofs, ns, st));
    FEMO_TRY(to_device(&s->d_recv_idx, recv_dofs, nr, st));
    FEMO_HIP_CHECK(hipMalloc(&s->d_send_buf, std::max<int64_t>(ns, 1) * sizeof(double)));
    FEMO_HIP_CHECK(hipMalloc(&s->d_recv_buf, std::max<int64_t>(nr, 1) * sizeof(double)));
  }
  FEMO_HIP_CHECK(hipStreamSynchronize(st));
  s->pc_vals_uid = 0; s->pc_vals_gen = 0; s->bs_vals_uid = 0; s->bs_vals_gen = 0;
  return 0;
}

// x on the points owned by other ranks <- the owners' values (collective over the ranks of the partition)
// The cells whose scalar outputs (mass, stress aggregate, energy, regularisation terms) this rank integrates: a uint8 per local
// cell, exactly one rank per cell of the whole mesh; the library sums the values over the ranks.  NULL clears it.
int femo_shell_set_owned_cells(femo_shell* s, const uint8_t* owned_cells) {
  FEMO_REQUIRE(s != nullptr, "null argument");
  (void)hipFree(s->d_cell_owned);
  s->d_cell_owned = nullptr;
  if (owned_cells != nullptr) FEMO_TRY(to_device(&s->d_cell_owned, owned_cells, s->n_cell, s->ctx->stream));
  FEMO_HIP_CHECK(hipStreamSynchronize(s->ctx->stream));
  return 0;
}

int femo_shell_halo(femo_shell* s, femo_vec* x) {
  FEMO_REQUIRE(s && x, "null argument");
  FEMO_REQUIRE(x->n >= s->n_dof, "vector size mismatch in shell_halo");
  FEMO_REQUIRE(s->d_owned != nullptr, "femo_shell_halo needs femo_shell_set_partition");
  femo_vec_touch(x);
  return shell_halo(s, x->d, s->ctx->stream);
}

// x <- 0 on the points owned by other ranks: the rank's share of a vector assembled over its cells
int femo_shell_mask_unowned(femo_shell* s, femo_vec* x) {
  FEMO_REQUIRE(s && x, "null argument");
  FEMO_REQUIRE(x->n >= s->n_dof, "vector size mismatch in shell_mask_unowned");
  if (s->d_owned == nullptr) return 0;
  femo_vec_touch(x);
  hipLaunchKernelGGL(k_mask_unowned, dim3(sgrid(s->n_dof, 256)), dim3(256), 0, s->ctx->stream, s->n_dof / 3, s->d_owned, x->d);
  FEMO_HIP_CHECK(hipGetLastError());
  return 0;
}

int femo_shell_penalty_apply(femo_shell* s, const femo_vec* x, const femo_vec* g, int accumulate, femo_vec* y) {
  FEMO_REQUIRE(s && x && y, "null argument");
  FEMO_REQUIRE(x->n >= s->n_dof && y->n >= s->n_dof && (g == nullptr || g->n >= s->n_dof) && x->d != y->d, "vector size mismatch in shell_penalty_apply");
  hipStream_t st = s->ctx->stream;
  femo_vec_touch(y);
  if (!accumulate) FEMO_HIP_CHECK(hipMemsetAsync(y->d, 0, s->n_dof * sizeof(double), st));
  if (s->pen_n == 0) return 0;
  hipLaunchKernelGGL(k_shell_penalty_apply, dim3(sgrid(s->pen_n, 256)), dim3(256), 0, st, s->pen_n, s->d_pen_nodes, s->d_pen_coef, s->n_unode, x->d,
                     g ? g->d : (const double*)nullptr, y->d);
  FEMO_HIP_CHECK(hipGetLastError());
  return 0;
}

// y (+)= M(h) acc: the inertial residual for the accelerations `acc` in state layout
int femo_shell_inertia_apply(femo_shell* s, double rho, const femo_vec* h, const femo_vec* acc, int accumulate, femo_vec* y) {
  FEMO_REQUIRE(s && h && acc && y, "null argument");
  FEMO_REQUIRE(h->n >= s->n_vert && acc->n >= s->n_dof && y->n >= s->n_dof && acc->d != y->d, "vector size mismatch in shell_inertia_apply");
  hipStream_t st = s->ctx->stream;
  femo_vec_touch(y);
  if (!accumulate) FEMO_HIP_CHECK(hipMemsetAsync(y->d, 0, s->n_dof * sizeof(double), st));
  hipLaunchKernelGGL(k_shell_inertia, dim3(sgrid(s->n_cell)), dim3(SH_BLOCK), 0, st, view(s), rho, h->d, acc->d, (const double*)nullptr, y->d, (double*)nullptr);
  FEMO_HIP_CHECK(hipGetLastError());
  return 0;
}

// out_b (+)= lam^T (dM/dh_b) acc: the thickness partial of the inertial residual, transposed
int femo_shell_inertia_dh(femo_shell* s, double rho, const femo_vec* h, const femo_vec* lam, const femo_vec* acc, int accumulate, femo_vec* out) {
  FEMO_REQUIRE(s && h && lam && acc && out, "null argument");
  FEMO_REQUIRE(h->n >= s->n_vert && lam->n >= s->n_dof && acc->n >= s->n_dof && out->n >= s->n_vert, "vector size mismatch in shell_inertia_dh");
  hipStream_t st = s->ctx->stream;
  femo_vec_touch(out);
  if (!accumulate) FEMO_HIP_CHECK(hipMemsetAsync(out->d, 0, s->n_vert * sizeof(double), st));
  hipLaunchKernelGGL(k_shell_inertia, dim3(sgrid(s->n_cell)), dim3(SH_BLOCK), 0, st, view(s), rho, h->d, acc->d, lam->d, (double*)nullptr, out->d);
  FEMO_HIP_CHECK(hipGetLastError());
  return 0;
}

// y (+)= (dM/dh [dh]) acc: the thickness partial of the inertial residual, forward mode
int femo_shell_inertia_dh_fwd(femo_shell* s, double rho, const femo_vec* h, const femo_vec* dh, const femo_vec* acc, int accumulate, femo_vec* y) {
  FEMO_REQUIRE(s && h && dh && acc && y, "null argument");
  FEMO_REQUIRE(h->n >= s->n_vert && dh->n >= s->n_vert && acc->n >= s->n_dof && y->n >= s->n_dof && acc->d != y->d, "vector size mismatch in shell_inertia_dh_fwd");
  hipStream_t st = s->ctx->stream;
  femo_vec_touch(y);
  if (!accumulate) FEMO_HIP_CHECK(hipMemsetAsync(y->d, 0, s->n_dof * sizeof(double), st));
  hipLaunchKernelGGL(k_shell_inertia, dim3(sgrid(s->n_cell)), dim3(SH_BLOCK), 0, st, view(s), rho, h->d, acc->d, (const double*)nullptr, y->d, (double*)nullptr, dh->d);
  FEMO_HIP_CHECK(hipGetLastError());
  return 0;
}

// kind 1 'H1', 2 'L2H1', 3 'L2' (shell_pde.py:262-282); value and / or gradient w.r.t. the thickness
int femo_shell_regularization(femo_shell* s, int kind, const femo_vec* h, double* value, int accumulate, femo_vec* grad) {
  FEMO_REQUIRE(s && h, "null argument");
  FEMO_REQUIRE(kind >= 1 && kind <= 3, "unknown regularisation kind %d", kind);
  FEMO_REQUIRE(h->n >= s->n_vert && (grad == nullptr || grad->n >= s->n_vert), "vector size mismatch in shell_regularization");
  hipStream_t st = s->ctx->stream;
  const unsigned g = sgrid(s->n_cell);
  FEMO_REQUIRE(value == nullptr || g <= 3 * SH_MAXPART, "mesh too large for the reduction buffer");
  if (grad) {
    femo_vec_touch(grad);
    if (!accumulate) FEMO_HIP_CHECK(hipMemsetAsync(grad->d, 0, s->n_vert * sizeof(double), st));
  }
  hipLaunchKernelGGL(k_shell_regularization, dim3(g), dim3(SH_BLOCK), 0, st, view(s), kind, h->d, value ? s->d_part : nullptr, grad ? grad->d : nullptr);
  FEMO_HIP_CHECK(hipGetLastError());
  if (value) FEMO_TRY(reduce_partials(s->ctx, s->d_part, (int)g, value));
  return 0;
}

// int coef h^p dx and its thickness gradient
int femo_shell_hpower(femo_shell* s, double coef, double p, const femo_vec* h, double* value, int accumulate, femo_vec* grad) {
  FEMO_REQUIRE(s && h, "null argument");
  FEMO_REQUIRE(h->n >= s->n_vert && (grad == nullptr || grad->n >= s->n_vert), "vector size mismatch in shell_hpower");
  hipStream_t st = s->ctx->stream;
  const unsigned g = sgrid(s->n_cell);
  FEMO_REQUIRE(value == nullptr || g <= 3 * SH_MAXPART, "mesh too large for the reduction buffer");
  if (grad) {
    femo_vec_touch(grad);
    if (!accumulate) FEMO_HIP_CHECK(hipMemsetAsync(grad->d, 0, s->n_vert * sizeof(double), st));
  }
  hipLaunchKernelGGL(k_shell_hpower, dim3(g), dim3(SH_BLOCK), 0, st, view(s), coef, p, h->d, value ? s->d_part : nullptr, grad ? grad->d : nullptr);
  FEMO_HIP_CHECK(hipGetLastError());
  if (value) FEMO_TRY(reduce_partials(s->ctx, s->d_part, (int)g, value));
  return 0;
}

// K_ff x_f = b_f - K_fc g_c with x_c = g_c on the dofs flagged in `fixed` (host array of n_dof bytes, values in xfix);
// PCG, stops on sqrt(r.M^-1 r) <= max(rtol sqrt(r0.M^-1 r0), atol); opts->pc = 0: M = D (Jacobi), 1: the lattice
// preconditioner of femo_shell_pc_create.  K symmetric: the same call serves the adjoint (fea_dolfinx.py:208-222).
// The preconditioner's numbers for the current stiffness and Dirichlet set (kept while both stay the same): dense coarse
// operator and its factors, node blocks (or Galerkin diagonals), point blocks.
// 64-bit hash of the caller's Dirichlet mask, 32 bytes per step in four independent lanes (identifies the mask for the caches
// below: the device copy and the preconditioner's numbers)
static uint64_t shell_mask_hash(const uint8_t* p, int64_t n) {
  if (p == nullptr) return 1469598103934665603ull;
  uint64_t h[4] = {0x9E3779B97F4A7C15ull, 0xC2B2AE3D27D4EB4Full, 0x165667B19E3779F9ull, 0x27D4EB2F165667C5ull};
  int64_t i = 0;
  for (; i + 32 <= n; i += 32) {
    uint64_t w[4];
    memcpy(w, p + i, 32);
#pragma unroll
    for (int k = 0; k < 4; ++k) { h[k] = (h[k] ^ w[k]) * 0x9FB21C651E98DF25ull; h[k] ^= h[k] >> 32; }
  }
  uint64_t t = 1469598103934665603ull ^ (uint64_t)n;
  for (; i < n; ++i) t = (t ^ p[i]) * 1099511628211ull;
  uint64_t r = t;
  for (int k = 0; k < 4; ++k) { r = (r ^ h[k]) * 0xD6E8FEB86659FD93ull; r ^= r >> 29; }
  return r != 0 ? r : 1;
}

// The mask on the device (nullptr without one) and its hash; the copy belongs to the shell and is re-uploaded only when the
// caller's array changed.
static int shell_mask(femo_shell* s, const uint8_t* fixed_host, const uint8_t** d_fixed, uint64_t* hash) {
  *hash = shell_mask_hash(fixed_host, s->n_dof);
  *d_fixed = nullptr;
  if (fixed_host == nullptr) return 0;
  if (s->d_fixed_kept == nullptr) {
    FEMO_HIP_CHECK(hipMalloc(&s->d_fixed_kept, std::max<int64_t>(s->n_dof, 1)));
    s->fixed_kept_hash = 0;
  }
  if (s->fixed_kept_hash != *hash) {
    // (the stream may still run kernels of an earlier call that read the old mask: same stream, ordered)
    FEMO_HIP_CHECK(hipMemcpyAsync(s->d_fixed_kept, fixed_host, s->n_dof, hipMemcpyHostToDevice, s->ctx->stream));
    FEMO_HIP_CHECK(hipStreamSynchronize(s->ctx->stream));        // the caller's (pageable) array may change after the call returns
    s->fixed_kept_hash = *hash;
  }
  *d_fixed = s->d_fixed_kept;
  return 0;
}

static int shell_pc_setup(femo_shell* s, const femo_vec* vals, uint64_t mh, const uint8_t* d_fixed) {
  hipStream_t st = s->ctx->stream;
  const int64_t n = s->n_dof;
  // Galerkin diagonals of the current stiffness and Dirichlet set (kept while both stay the same; mh: shell_mask_hash)
  if (s->pc_vals_uid != vals->uid || s->pc_vals_gen != vals->gen || s->pc_mask_hash != mh || vals->uid == 0) {
    FEMO_TRY(shell_pc_coarse_setup(s, vals, d_fixed));
    // levels the coarse solve does not replace: 6 x 6 node blocks (they see the coupling of the displacement
    // components and rotations at a node: 238 -> 203 iterations on the 128 x 128 roof, 412 -> 376 on 362 x 362), or
    // the Galerkin diagonals when there is no coarse solve
    const int first_slot = s->cs_ready ? 8 * (s->cs_level + 1) : 0;
    s->blk_ready = false;
    if (s->cs_ready && s->d_lvl_node != nullptr && getenv("FEMO_SHELL_NO_BLOCKS") == nullptr) {
      const int64_t nd0 = s->level_off[s->cs_level + 1], nd1 = s->level_off[s->pc_levels];
      FEMO_HIP_CHECK(hipMemsetAsync(s->d_cblk + 36 * nd0, 0, (nd1 - nd0) * 36 * sizeof(double), st));
      bool by_items = s->hermite_on && s->bi_items > 0 && !femo_env_flag("FEMO_SHELL_BLOCKS_BY_ROWS");
      if (by_items) {
        hipLaunchKernelGGL(k_point_fixbits, dim3(sgrid(n / 3, 256)), dim3(256), 0, st, n / 3, d_fixed, s->d_fixbits);
        hipLaunchKernelGGL(k_pc_galerkin_blocks_w, dim3((unsigned)((s->bi_items + 3) / 4)), dim3(256), 0, st, s->bi_items, n / 3, s->n_unode, s->d_bi_ptr,
                           s->d_bi_lvl, s->d_bi_pts, s->d_bi_pcell, s->d_brow, s->d_bcols, vals->d, (const uint8_t*)s->d_fixbits, s->d_lvl_node,
                           s->d_lvl_w4, s->d_cblk, s->d_cs_info);
        int32_t binfo[4] = {0, 0, 0, 0};
        FEMO_HIP_CHECK(hipMemcpyAsync(binfo, s->d_cs_info, sizeof binfo, hipMemcpyDeviceToHost, st));
        FEMO_HIP_CHECK(hipStreamSynchronize(st));
        if (binfo[1] & 2) {                                 // an element spans more than a cell of some level: the row-wise kernel has no such limit
          by_items = false;
          s->bi_items = 0;
          FEMO_HIP_CHECK(hipMemsetAsync(s->d_cblk + 36 * nd0, 0, (nd1 - nd0) * 36 * sizeof(double), st));
        }
      }
      if (by_items) {
      } else if (s->hermite_on)
        hipLaunchKernelGGL(k_pc_galerkin_blocks_h, dim3(sgrid(n / 3), 3 * ((s->pc_width - first_slot) / 8)), dim3(SH_BLOCK), 0, st, n / 3, s->n_unode,
                           s->d_brow, s->d_bcols, vals->d, d_fixed, s->d_lvl_node, s->d_lvl_w4, s->d_cblk);
      else
        hipLaunchKernelGGL(k_pc_galerkin_blocks, dim3(sgrid(n / 3), 3 * ((s->pc_width - first_slot) / 8)), dim3(SH_BLOCK), 0, st, n / 3, s->pc_width,
                           s->n_unode, s->d_brow, s->d_bcols, vals->d, d_fixed, s->d_lvl_node, s->d_lvl_w, s->d_cblk, first_slot);
      FEMO_HIP_CHECK(hipGetLastError());
      FEMO_TRY(shell_allreduce(s, s->d_cblk + 36 * nd0, (nd1 - nd0) * 36, st));
      for (int l = s->cs_level + 1; l < s->pc_levels; ++l) {
        const int64_t a0 = s->level_off[l], a1 = s->level_off[l + 1];
        if (a1 > a0) hipLaunchKernelGGL(k_pc_invert_blocks, dim3(sgrid(a1 - a0, 256)), dim3(256), 0, st, a0, a1, s->d_cblk, shell_level_weight(s, l));
      }
      s->blk_ready = true;
    } else {
      FEMO_HIP_CHECK(hipMemsetAsync(s->d_coarse, 0, s->n_lat * sizeof(double), st));
      hipLaunchKernelGGL(k_pc_galerkin_diag, dim3(sgrid(n * (s->pc_width - first_slot))), dim3(SH_BLOCK), 0, st, n, s->pc_width, s->d_rowptr, s->d_cols,
                         vals->d, d_fixed, s->d_ell_idx, s->d_ell_w, s->d_coarse, first_slot);
      FEMO_HIP_CHECK(hipGetLastError());
      FEMO_TRY(shell_allreduce(s, s->d_coarse, s->n_lat, st));
      hipLaunchKernelGGL(k_pc_invert, dim3(sgrid(s->n_lat)), dim3(256), 0, st, s->n_lat, s->d_coarse);
    }
    s->dinv3_ready = false;
    if (s->d_brow != nullptr && getenv("FEMO_SHELL_NO_POINT_BLOCKS") == nullptr) {
      hipLaunchKernelGGL(k_pt_block_inv, dim3(sgrid(n / 3, 256)), dim3(256), 0, st, n / 3, s->d_brow, s->d_bcols, vals->d, d_fixed, s->d_dinv3, s->d_dinv);
      s->dinv3_ready = true;
    }
    s->pc_vals_uid = vals->uid; s->pc_vals_gen = vals->gen; s->pc_mask_hash = mh;
  }
  return 0;
}

// z = M^-1 r of the lattice preconditioner for `vals` and the mask (tests compare it with oracle/shell_oracle.py::
// LatticePreconditioner.apply); entries of r on imposed dofs are ignored, z is zero there.
int femo_shell_pc_apply(femo_shell* s, const femo_vec* vals, const uint8_t* fixed_host, const femo_vec* r, femo_vec* z) {
  FEMO_REQUIRE(s && vals && r && z, "null argument");
  const int64_t n = s->n_dof;
  FEMO_REQUIRE(s->pc_width > 0, "femo_shell_pc_apply needs femo_shell_pc_create");
  FEMO_REQUIRE(vals->n >= s->nnz && r->n >= n && z->n >= n && n % 3 == 0, "vector size mismatch in shell_pc_apply");
  FEMO_REQUIRE(s->d_owned == nullptr, "femo_shell_pc_apply: one rank only");
  hipStream_t st = s->ctx->stream;
  femo_vec_touch(z);
  const uint8_t* d_fixed = nullptr;
  uint64_t mask_hash = 0;
  FEMO_TRY(shell_mask(s, fixed_host, &d_fixed, &mask_hash));
  const unsigned gv = std::min<unsigned>(sgrid(n), SH_MAXPART);
  hipLaunchKernelGGL(k_rhs_free, dim3(gv), dim3(256), 0, st, n, r->d, d_fixed, s->d_r);
  FEMO_TRY(shell_pc_setup(s, vals, mask_hash, d_fixed));
  const unsigned gz = std::min<unsigned>(sgrid(n / 3, SH_BLOCK / 8), SH_MAXPART);
  FEMO_TRY(shell_pc_apply(s, d_fixed, s->d_part + SH_MAXPART, gz, nullptr));
  FEMO_HIP_CHECK(hipMemcpyAsync(z->d, s->d_z, n * sizeof(double), hipMemcpyDeviceToDevice, st));
  FEMO_HIP_CHECK(hipStreamSynchronize(st));
  return 0;
}

int femo_shell_solve(femo_shell* s, const femo_vec* vals, const uint8_t* fixed_host, const femo_vec* xfix, const femo_vec* b,
                     femo_vec* x, const femo_solver_opts* opts, femo_solve_info* info) {
  FEMO_REQUIRE(s && vals && b && x && opts && info, "null argument");
  const int64_t n = s->n_dof;
  FEMO_REQUIRE(vals->n >= s->nnz && b->n >= n && x->n >= n && b->d != x->d, "vector size mismatch in shell_solve");
  FEMO_REQUIRE(fixed_host == nullptr || xfix == nullptr || xfix->n >= n, "prescribed values shorter than n_dof");
  femo_ctx* ctx = s->ctx;
  hipStream_t st = ctx->stream;
  memset(info, 0, sizeof *info);
  femo_vec_touch(x);
  const uint8_t* d_fixed = nullptr;
  uint64_t mask_hash = 0;
  FEMO_TRY(shell_mask(s, fixed_host, &d_fixed, &mask_hash));
  const unsigned gv = std::min<unsigned>(sgrid(n), SH_MAXPART);
  // workgroups of the operator product (their per-block partials of p.q are folded by k_scg_xr*): 16 rows, or 16
  // node blocks of three rows, per workgroup pass
  const bool bsell = s->d_bs_vals != nullptr && getenv("FEMO_SHELL_NO_BSELL") == nullptr;
  // block-SELL: 16 slices per workgroup pass, at most 2048 workgroups (0.376 ms per iteration at 1.97 M dofs against
  // 0.390 with one pass per workgroup: fewer partial sums for the consumers to fold)
  const unsigned gs = bsell ? std::min<unsigned>(sgrid(s->n_bslice, SH_BLOCK / BSW), 2048u)
                            : std::min<unsigned>(s->d_brow != nullptr ? sgrid(s->n_bnode, SH_BLOCK / 16) : sgrid(n, SH_BLOCK / 16), SH_MAXPART);
  double *Ppq = s->d_part, *Prz = s->d_part + SH_MAXPART, *gam = s->d_scal + 4;
  FEMO_HIP_CHECK(hipEventRecord(ctx->ev0, st));
  // right-hand side with lifting (into q); zero initial guess
  FEMO_HIP_CHECK(hipMemsetAsync(x->d, 0, n * sizeof(double), st));
  const double* rhs = b->d;
  if (d_fixed != nullptr && xfix != nullptr) {
    hipLaunchKernelGGL(k_csr_lift, dim3(gs), dim3(SH_BLOCK), 0, st, n, s->d_rowptr, s->d_cols, vals->d, d_fixed, xfix->d, b->d, s->d_q);
    rhs = s->d_q;
  }
  hipLaunchKernelGGL(k_rhs_free, dim3(gv), dim3(256), 0, st, n, rhs, d_fixed, s->d_r);
  // Partitioned shell (femo_shell_set_partition): the rows of K and the entries of r on points owned elsewhere are zero, so
  // every dot product below is the rank's share and P^T r, P^T K P sum over the ranks to the global objects; the producers'
  // partials are folded into one number, all-reduced, and handed to the unfused consumers as a single "partial".  The
  // direction p is refreshed on the halo before every product (x follows: it is updated with the refreshed p).
  const bool multi = s->d_owned != nullptr;
  if (multi) {
    FEMO_REQUIRE(n % 3 == 0, "a partitioned shell numbers its dofs 3 point + component");
    hipLaunchKernelGGL(k_mask_unowned, dim3(gv), dim3(256), 0, st, n / 3, s->d_owned, s->d_r);
  }
  double *one_pq = s->d_scal + 6, *one_rz = s->d_scal + 7;   // the all-reduced p.q and r.z
  const bool lattice = opts->pc == 1;
  // 1 / diag: with the lattice preconditioner and its point blocks it comes out of k_pt_block_inv (below, and only when
  // the stiffness or the mask changed)
  if (!(lattice && s->d_brow != nullptr && getenv("FEMO_SHELL_NO_POINT_BLOCKS") == nullptr))
    hipLaunchKernelGGL(k_csr_diag_inv, dim3(gv), dim3(256), 0, st, n, s->d_rowptr, s->d_cols, vals->d, d_fixed, s->d_dinv);
  FEMO_REQUIRE(!lattice || s->pc_width > 0, "opts->pc = 1 needs femo_shell_pc_create");
  if (bsell && (s->bs_vals_uid != vals->uid || s->bs_vals_gen != vals->gen || vals->uid == 0)) {
    hipLaunchKernelGGL(k_bsell_fill, dim3((unsigned)s->n_bslice), dim3(256), 0, st, s->n_bnode, s->d_brow, s->d_bcols, vals->d, s->d_bs_off,
                       s->d_bs_cols, s->d_bs_vals);
    s->bs_vals_uid = vals->uid; s->bs_vals_gen = vals->gen;
  }
  const unsigned gz = std::min<unsigned>(sgrid(n / 3, SH_BLOCK / 8), SH_MAXPART);     // k_pc_prolong: 8 lanes per point
  const unsigned gx = std::min<unsigned>(sgrid(n / 3), 1024u);                        // k_scg_xr_pt: a thread per point, few partials
  double* Pte = s->d_part + 2 * SH_MAXPART;
  // direction update fused into the prolongation (dofs numbered 3 point + component: every shell pattern of fea/shell.py)
  const bool fused = opts->pc == 1 && n % 3 == 0 && !multi && !femo_env_flag("FEMO_SHELL_UNFUSED");
  if (lattice) {
    FEMO_TRY(shell_pc_setup(s, vals, mask_hash, d_fixed));
    FEMO_TRY(shell_pc_apply(s, d_fixed, Prz, gz, nullptr));
    hipLaunchKernelGGL(k_copy, dim3(gv), dim3(256), 0, st, n, s->d_z, s->d_p);
  } else {
    hipLaunchKernelGGL(k_scg_init, dim3(gv), dim3(SH_BLOCK), 0, st, n, s->d_r, s->d_dinv, s->d_p, Prz);
  }
  // x += alpha p inside the preconditioner's first coarse product (fused loop, one rank, coarse solve in single precision)
  const bool carry_x = fused && s->cs_ready && s->d_cs_Af != nullptr && !femo_env_flag("FEMO_SHELL_NO_XCARRY");
  const int nb_rz0 = lattice ? (int)gz : (int)gv;
  if (multi) {
    hipLaunchKernelGGL(k_fold1, dim3(1), dim3(SH_BLOCK), 0, st, nb_rz0, Prz, one_rz, (const int32_t*)nullptr);
    FEMO_HIP_CHECK(hipGetLastError());
    FEMO_TRY(shell_allreduce(s, one_rz, 1, st));
    hipLaunchKernelGGL(k_scg_gamma0, dim3(1), dim3(SH_BLOCK), 0, st, 1, one_rz, opts->rtol * opts->rtol, opts->atol * opts->atol, s->d_scal, s->d_flag);
  } else {
    hipLaunchKernelGGL(k_scg_gamma0, dim3(1), dim3(SH_BLOCK), 0, st, nb_rz0, Prz, opts->rtol * opts->rtol, opts->atol * opts->atol, s->d_scal, s->d_flag);
  }
  FEMO_HIP_CHECK(hipGetLastError());
  const int max_it = opts->max_it > 0 ? opts->max_it : 100000;
  const int batch = opts->check_every > 0 ? opts->check_every : 64;
  int32_t h_flag[4] = {0, 0, 0, 0};
  double h_scal[8];
  FEMO_HIP_CHECK(hipMemcpyAsync(h_flag, s->d_flag, sizeof h_flag, hipMemcpyDeviceToHost, st));
  FEMO_HIP_CHECK(hipMemcpyAsync(h_scal, s->d_scal, sizeof h_scal, hipMemcpyDeviceToHost, st));
  FEMO_HIP_CHECK(hipStreamSynchronize(st));
  info->rhs_norm = std::sqrt(h_scal[1]);
  int it = 0, since_mark = 0;
  const int n_sample = 4;
  int n_ev = 0;
  bool stalled = false;
  double best = HUGE_VAL, best_mark = HUGE_VAL;
  while (!h_flag[0] && it < max_it) {
    const int it_end = std::min(it + batch, max_it);
    for (; it < it_end; ++it) {
      // HIP events around four of the operator products (iterations 4 .. 7) for the roofline record of bench.py
      const bool sample = it >= 4 && it < 4 + n_sample;
      if (sample) FEMO_HIP_CHECK(hipEventRecord(ctx->ev_pool[2 * n_ev], st));
      // p is zero on the imposed dofs (r and the initial direction are): no column mask needed
      if (multi) FEMO_TRY(shell_halo(s, s->d_p, st));
      if (bsell)
        hipLaunchKernelGGL(k_bsell_spmv, dim3(gs), dim3(SH_BLOCK), 0, st, s->n_bnode, s->n_bslice, s->d_bs_off, s->d_bs_cols, s->d_bs_vals, d_fixed, s->d_p, s->d_q, Ppq, s->d_flag, s->d_scal, gam);
      else if (s->d_brow != nullptr)
        hipLaunchKernelGGL(k_bcsr3_spmv<16>, dim3(gs), dim3(SH_BLOCK), 0, st, s->n_bnode, s->d_brow, s->d_bcols, vals->d, d_fixed, s->d_p, s->d_q, Ppq, s->d_flag, s->d_scal, gam);
      else
        hipLaunchKernelGGL(k_csr_spmv, dim3(gs), dim3(SH_BLOCK), 0, st, n, s->d_rowptr, s->d_cols, vals->d, d_fixed, 0, s->d_p, s->d_q, Ppq, s->d_flag, s->d_scal, gam);
      if (sample) { FEMO_HIP_CHECK(hipEventRecord(ctx->ev_pool[2 * n_ev + 1], st)); ++n_ev; }
      if (multi) {
        hipLaunchKernelGGL(k_fold1, dim3(1), dim3(SH_BLOCK), 0, st, (int)gs, Ppq, one_pq, s->d_flag);
        FEMO_HIP_CHECK(hipGetLastError());
        FEMO_TRY(shell_allreduce(s, one_pq, 1, st));
        if (lattice) {
          hipLaunchKernelGGL(k_scg_xr_plain, dim3(gv), dim3(SH_BLOCK), 0, st, n, 1, one_pq, s->d_scal, s->d_p, s->d_q, x->d, s->d_r, s->d_flag);
          FEMO_TRY(shell_pc_apply(s, d_fixed, Prz, gz, s->d_flag));
          hipLaunchKernelGGL(k_fold1, dim3(1), dim3(SH_BLOCK), 0, st, (int)gz, Prz, one_rz, s->d_flag);
          FEMO_HIP_CHECK(hipGetLastError());
          FEMO_TRY(shell_allreduce(s, one_rz, 1, st));
          hipLaunchKernelGGL(k_scg_p_z, dim3(gv), dim3(SH_BLOCK), 0, st, n, it, 1, one_rz, s->d_scal, s->d_z, s->d_p, s->d_flag, gam);
        } else {
          hipLaunchKernelGGL(k_scg_xr, dim3(gv), dim3(SH_BLOCK), 0, st, n, 1, one_pq, s->d_scal, s->d_p, s->d_q, s->d_dinv, x->d, s->d_r, Prz, s->d_flag);
          hipLaunchKernelGGL(k_fold1, dim3(1), dim3(SH_BLOCK), 0, st, (int)gv, Prz, one_rz, s->d_flag);
          FEMO_HIP_CHECK(hipGetLastError());
          FEMO_TRY(shell_allreduce(s, one_rz, 1, st));
          hipLaunchKernelGGL(k_scg_p, dim3(gv), dim3(SH_BLOCK), 0, st, n, it, 1, one_rz, s->d_scal, s->d_r, s->d_dinv, s->d_p, s->d_flag, gam);
        }
      } else if (lattice && fused) {
        // r . z = r . B r + (P^T r) . e is known before z is: the update emits the first part, the finest lattice level
        // the second, and the prolongation writes p = z + beta p at once (9 launches and 3 vector streams fewer
        // per iteration than the unfused form below)
        int nb_te = 0;
        if (carry_x) {
          const ShellXCarry xc = {x->d, s->d_p, s->d_scal + 5, n, 0};
          hipLaunchKernelGGL(k_scg_xr_pt, dim3(gx), dim3(SH_BLOCK), 0, st, n / 3, (int)gs, Ppq, s->d_scal, s->d_p, s->d_q, s->d_dinv,
                             s->dinv3_ready ? s->d_dinv3 : (const float*)nullptr, x->d, s->d_r, Prz, s->d_flag, s->d_scal + 5);
          FEMO_TRY(shell_pc_apply(s, d_fixed, Prz, gz, s->d_flag, Pte, &nb_te, &xc));
        } else {
          hipLaunchKernelGGL(k_scg_xr_pt, dim3(gx), dim3(SH_BLOCK), 0, st, n / 3, (int)gs, Ppq, s->d_scal, s->d_p, s->d_q, s->d_dinv,
                             s->dinv3_ready ? s->d_dinv3 : (const float*)nullptr, x->d, s->d_r, Prz, s->d_flag);
          FEMO_TRY(shell_pc_apply(s, d_fixed, Prz, gz, s->d_flag, Pte, &nb_te));
        }
        // 2048 workgroups = one resident round of 8 waves per SIMD, ten trips each: 45.0 us at 1.97 M dofs against 49.3 with 4096,
        // 56.6 with 8192, 61.5 with 1024 (every workgroup starts with two folds and three dependent scalar reads; fewer
        // partials to fold change nothing: 44.4 - 46.3 us with 256 - 1024 of each)
        const unsigned gzf = std::min<unsigned>(gz, 2048u);
        hipLaunchKernelGGL(k_pc_prolong_fused, dim3(gzf), dim3(SH_BLOCK), 0, st, n / 3, it, (int)gx, Prz, nb_te, Pte, s->d_scal, s->d_fin_idx, s->d_fin_w,
                           d_fixed, s->d_dinv, s->dinv3_ready ? s->d_dinv3 : (const float*)nullptr, s->d_r, s->d_e, s->d_p, s->d_flag, gam,
                           (s->hermite_on && s->cs_ready && s->blk_ready) ? s->d_fin_w4 : (const float4*)nullptr, s->n_unode);
      } else if (lattice) {
        hipLaunchKernelGGL(k_scg_xr_plain, dim3(gv), dim3(SH_BLOCK), 0, st, n, (int)gs, Ppq, s->d_scal, s->d_p, s->d_q, x->d, s->d_r, s->d_flag);
        FEMO_TRY(shell_pc_apply(s, d_fixed, Prz, gz, s->d_flag));
        hipLaunchKernelGGL(k_scg_p_z, dim3(gv), dim3(SH_BLOCK), 0, st, n, it, (int)gz, Prz, s->d_scal, s->d_z, s->d_p, s->d_flag, gam);
      } else {
        hipLaunchKernelGGL(k_scg_xr, dim3(gv), dim3(SH_BLOCK), 0, st, n, (int)gs, Ppq, s->d_scal, s->d_p, s->d_q, s->d_dinv, x->d, s->d_r, Prz, s->d_flag);
        hipLaunchKernelGGL(k_scg_p, dim3(gv), dim3(SH_BLOCK), 0, st, n, it, (int)gv, Prz, s->d_scal, s->d_r, s->d_dinv, s->d_p, s->d_flag, gam);
      }
    }
    FEMO_HIP_CHECK(hipGetLastError());
    FEMO_HIP_CHECK(hipMemcpyAsync(h_flag, s->d_flag, sizeof h_flag, hipMemcpyDeviceToHost, st));
    FEMO_HIP_CHECK(hipMemcpyAsync(h_scal, s->d_scal, sizeof h_scal, hipMemcpyDeviceToHost, st));
    FEMO_HIP_CHECK(hipStreamSynchronize(st));
    // Attainable accuracy: sqrt(r.M^-1 r) measures the error in the energy norm, where fp64 delivers about
    // eps sqrt(cond K) of the solution (1e-11 for a thin shell); a tolerance below that is never met and CG wanders
    // (450 s on a 2 k-dof roof with rtol 1e-12 and the coarse solve, whose norm is honest about the smooth modes).
    // Once the residual is below 1e-9 of the initial one in that norm and the best value seen has not halved in 8
    // batches, the solve ends with converged = 2.
    if (!h_flag[0]) {
      const double g = h_scal[0];
      if (g == g && g < best) {
        if (g < 0.5 * best_mark) { best_mark = g; since_mark = 0; }
        best = g;
      }
      if (++since_mark > 8 && best <= 1e-18 * h_scal[1]) { stalled = true; break; }
    }
  }
  if (d_fixed != nullptr) hipLaunchKernelGGL(k_set_fixed, dim3(gv), dim3(256), 0, st, n, d_fixed, xfix ? xfix->d : nullptr, x->d);
  if (multi) FEMO_TRY(shell_halo(s, x->d, st));          // the caller reads a consistent state on all its points
  FEMO_HIP_CHECK(hipMemcpyAsync(h_scal, s->d_scal, sizeof h_scal, hipMemcpyDeviceToHost, st));
  FEMO_HIP_CHECK(hipEventRecord(ctx->ev1, st));
  FEMO_HIP_CHECK(hipStreamSynchronize(st));
  float ms = 0.f;
  FEMO_HIP_CHECK(hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
  info->solve_ms = ms;
  {
    const int iters_run = h_flag[0] ? h_flag[1] : it;
    double acc = 0.0;
    int used = 0;
    for (int i = 0; i < n_ev; ++i) {
      if (4 + i >= iters_run) break;                     // a launch behind the converged iteration returned at once
      float t = 0.f;
      FEMO_HIP_CHECK(hipEventElapsedTime(&t, ctx->ev_pool[2 * i], ctx->ev_pool[2 * i + 1]));
      acc += t; ++used;
    }
    info->spmv_ms = acc;
    info->spmv_samples = used;
  }
  info->iterations = h_flag[0] ? h_flag[1] : it;
  info->converged = h_flag[0] ? (h_flag[2] ? -1 : 1) : (stalled ? 2 : 0);
  info->residual_norm = std::sqrt(std::max(h_flag[0] && h_flag[1] > 0 ? h_scal[4] : h_scal[0], 0.0));
  return 0;
}

}  // extern "C"
