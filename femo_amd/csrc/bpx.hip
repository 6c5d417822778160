// Auxiliary-lattice BPX preconditioner for the CG solves (new design; the reference
// factorises with MUMPS, BASELINE.json asks for CG).  With Jacobi alone CG needs O(n)
// iterations on an n^d grid (970 per solve on the 10 M-DOF cube); this additive multilevel
// preconditioner makes the count mesh-independent (~40) at the price of a few gather /
// scatter passes per iteration:
//
//     M^-1 = D^-1 + sum_l  P_l C_l P_l^T
//
// P_l = multilinear interpolation from a regular lattice (2^l m_k bins per axis over the
// bounding box of the mesh) to the mesh vertices, computed on the fly from the vertex
// coordinates -- no coarse matrices, no coarse solves, nothing mesh-structure specific.
// C_l = 0.6 / (diagonal of the Q1 Laplacian on lattice l) = 0.6 * 3/(8 H_l) in 3-D, 0.6 * 3/8 in 2-D
// (classical BPX scaling; Galerkin diagonals measured the same iteration counts).  Dirichlet
// vertices are masked out of P (their rows are identity rows; vertices on Nitsche facets count
// as Dirichlet, the penalty pins them), and the lattice nodes sitting on that boundary are
// dropped on every level: a finest-lattice node is dropped when more than 30 % of its hat
// function's vertex mass is Dirichlet, coarser lattices inherit the flag by injection (their
// nodes are a subset of the finest ones).  Measured with this rule (rtol 1e-14): 40-44
// iterations on the jittered 3-D cube, the 2-D square and both Nitsche Jacobians, against
// 200-1200 with Jacobi; without the node rule 70-130.  Nested lattices: only the finest
// one touches the mesh (restriction brick by brick through LDS into a lattice that fits in
// L2 / Infinity Cache, prolongation by gathers); the coarser levels are reached lattice to
// lattice with 27-point (9-point) stencils.  Both mesh transfers read 12-byte packed lattice
// coordinates (bin, 20-bit fraction per axis) instead of the vertex coordinates.  CG runs on
// the symmetrically scaled system Ah = S A S, so in scaled variables
//     zh = rh + S^-1 (sum_l P_l C_l P_l^T) S^-1 rh.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <type_traits>

#include "femo_internal.h"

namespace {

struct LatticeLevel {
  int n[3];            // bins per axis (1 for unused axes in 2-D)
  int64_t nodes;       // (n0+1)(n1+1)(n2+1)
  double H;            // spacing
  double* g = nullptr;     // restricted residual
  double* e = nullptr;     // accumulated correction
  double* coef = nullptr;  // C_l * keep mask
};

}  // namespace

struct femo_pc {
  int dim = 3, n_levels = 0;
  double lo[3] = {0, 0, 0}, hi[3] = {1, 1, 1};
  std::vector<LatticeLevel> L;
  uint64_t built_key = 0;   // identity of the Dirichlet mask the coef arrays were built for
  double* g_all = nullptr;  // the g arrays of all levels, coarsest first, contiguous
  // The levels the brick kernel accumulates into (the finest n_fused + 1) exist twice; applies alternate between
  // the two copies.  The fused prolongation reads the second-finest level from several workgroups, so it cannot
  // clear it in place: it clears the OTHER copy, which the next restriction will accumulate into.
  double* g_alt = nullptr;
  int parity = 0;
  int n_fused = 0;          // coarser levels the brick kernel restricts to directly (besides the finest)
  int brick_pf = 4;         // staging depth of the brick kernel: smallest of 2..4 with 256 x brick_pf >= the fullest brick
  bool fused_cycle_seen = false;   // the last apply ran the fused lattice cycle (femo_pc_carries_xupdate)
  bool coarse_lds_set = false;
  // owned vertices sorted by brick (BRICK^dim bins of the finest lattice), for the restriction
  int64_t n_bricks = 0;
  int32_t* d_perm = nullptr;        // sorted position -> vertex
  uint32_t* d_pk = nullptr;         // packed lattice coordinates, 8 B per owned vertex (vertex order)
  uint32_t* d_pk_sorted = nullptr;  // the same in sorted order
  // 1/s ROUNDED TO SINGLE PRECISION, 0 on pinned vertices, refreshed per solve: in sorted order for the restriction and in mesh
  // order for the prolongation (the same numbers, so S^-1 P and P^T S^-1 stay transposes; the rounding perturbs the
  // preconditioner's scaling by 6e-8 relative, the solution not at all)
  float* d_w_sorted = nullptr;
  float* d_sinv = nullptr;
  double* d_dot_partials = nullptr; // per-block partials of g_L.e_L (4096)
  // partitioned meshes: only the finest-lattice nodes that several ranks touch are exchanged
  bool shared_ready = false;
  int64_t n_shared = 0;
  int32_t* d_shared_idx = nullptr;  // finest-lattice nodes touched by >= 2 ranks (same list on every rank)
  double* d_dot_weight = nullptr;   // 1/(ranks touching the node) where this rank touches it, else 0
  double* d_xbuf = nullptr;         // [shared nodes of level L | all of level L-1]: the one all-reduce per apply
  double* d_dot_scalar = nullptr;
  int64_t* d_brick_ptr = nullptr;   // n_bricks + 1
  uint32_t* d_bin_ptr = nullptr;    // 65 per brick: start of each of its 64 bins, relative to the brick
  int32_t* d_brick_base = nullptr;  // 3 per brick: first bin of the brick along each axis
  bool coef_valid = false;
  // merged BPX-PCG (femo_internal.h): the restricted residual of levels T-1 .. L as state, [T-1 | T | L-1 | L] like g_all
  double* gs = nullptr;
  int64_t gs_n = 0;
  double* d_lat_partials = nullptr;   // per carrier workgroup of k_lattice_coarse_m: sum C g^2 over its part of levels T, L-1
  double* d_rr_partials = nullptr;    // ... and r.r of its part of the updated residual
  // N > 1, merged loop (round 5): the three brick-filled levels T, L-1, L are ALL exchanged sparsely.  Indices below
  // address the contiguous range [T | L-1 | L] (the layout of g_all / g_alt / gs + nodes(T-1) / coef_all from level T on).
  //   d_mshared_idx: nodes that several ranks touch (same list on every rank): their partial sums travel, every rank keeps
  //                  their state and adds their part of the lattice dot itself;
  //   d_mint_idx:    nodes only this rank touches: updated locally, their part of the lattice dot travels as 3 scalars;
  //                  the first n_mint_coarse entries lie on levels T and L-1 (the carriers of k_lattice_coarse_m update
  //                  them; the finest level's are updated tile by tile in k_lattice_prolong3_m).
  // Level T-1 (a few thousand nodes, where the replicated single-workgroup hierarchy starts) travels dense, already
  // restricted: the pack kernel applies R to the rank's PARTIAL h_T (restriction is linear).  Round 4 sent levels T and L-1
  // whole (133 k of the 138 k doubles of an all-reduce at C4 on 8 ranks) and scattered the sums back with an unpack launch.
  int64_t n_mshared = 0, n_mint = 0, n_mint_coarse = 0;
  int32_t* d_mshared_idx = nullptr;
  int32_t* d_mint_idx = nullptr;
  double* d_mbuf = nullptr;           // [shared nodes of T, L-1, L | R h_T on level T-1 | MS_NRED scalars]: the one all-reduce per iteration
  bool merged_lds_set = false;        // k_lattice_coarse_m's LDS limit raised on this context's device (was a process-wide static)
  // ... and the part of the finest three levels this rank needs: the 8^3 (16^2) tiles of k_lattice_prolong3_m that hold a
  // node it touches (1/N of the lattice plus the interface layers)
  int64_t n_my_tiles = 0;
  int32_t* d_my_tiles = nullptr;
  std::vector<uint8_t> tile_mine;     // host copy of the tile mask (from the brick list: every node a local vertex touches lies in one of them)
  double* coef_all = nullptr;         // the coef arrays of all levels, coarsest first, contiguous (like g_all)
};

namespace {

__device__ __forceinline__ int64_t node_index(const int* n, int i, int j, int k) {
  return ((int64_t)k * (n[1] + 1) + j) * (n[0] + 1) + i;
}

struct Lat {           // by-value lattice descriptor for kernels
  int n[3];
  double lo[3], inv_h[3];
};

// bin and local coordinates of a point; weights of the 2^D corners
template <int D>
__device__ __forceinline__ void locate(const Lat& lat, const double* __restrict__ x, int64_t v, int i0[3], double t[3]) {
  i0[2] = 0; t[2] = 0.0;
#pragma unroll
  for (int k = 0; k < D; ++k) {
    const double g = (x[v * D + k] - lat.lo[k]) * lat.inv_h[k];
    int b = (int)floor(g);
    b = b < 0 ? 0 : (b > lat.n[k] - 1 ? lat.n[k] - 1 : b);
    i0[k] = b;
    double f = g - b;
    t[k] = f < 0.0 ? 0.0 : (f > 1.0 ? 1.0 : f);
  }
}

// g_L += P_L^T (rh / s)   (owned, unmasked vertices); `val` == nullptr restricts the constant 1
template <int D>
__global__ __launch_bounds__(FEMO_BLOCK) void k_restrict_mesh(int64_t n_rows, Lat lat, const double* __restrict__ x,
                                                              const double* __restrict__ val, const double* __restrict__ s,
                                                              const uint8_t* __restrict__ mask, int only_masked,
                                                              double* __restrict__ g, const int32_t* __restrict__ done) {
  if (done != nullptr && *done) return;
  for (int64_t v = (int64_t)blockIdx.x * FEMO_BLOCK + threadIdx.x; v < n_rows; v += (int64_t)gridDim.x * FEMO_BLOCK) {
    const bool m = mask != nullptr && mask[v];
    if (only_masked ? !m : m) continue;
    const double r = val ? val[v] / s[v] : 1.0;
    int i0[3];
    double t[3];
    locate<D>(lat, x, v, i0, t);
#pragma unroll
    for (int c = 0; c < (1 << D); ++c) {
      double w = r;
      int ijk[3] = {i0[0], i0[1], i0[2]};
#pragma unroll
      for (int k = 0; k < D; ++k) {
        const int bit = (c >> k) & 1;
        w *= bit ? t[k] : 1.0 - t[k];
        ijk[k] += bit;
      }
      if (w != 0.0) atomicAdd(&g[node_index(lat.n, ijk[0], ijk[1], ijk[2])], w);
    }
  }
}

// Lattice coordinates of a vertex on the finest lattice, 8 bytes per vertex (femo_internal.h): 2-D two words
// bin << 20 | 20-bit fraction, 3-D one 64-bit word of three fields bin << 12 | 12-bit fraction.  Both transfers decode the
// same words, so P and P^T stay exact transposes; the quantisation of the weights only perturbs the preconditioner.
// Rounds 1-4 spent 12 B per 3-D vertex (three 20-bit words); the 24 B of coordinates were never read by the transfers.
constexpr int PK_BITS = FEMO_PK_BITS;
constexpr uint32_t PK_MASK = (1u << PK_BITS) - 1u;
constexpr int PK3_BITS = FEMO_PK3_BITS, PK3_FIELD = FEMO_PK3_FIELD;
constexpr uint32_t PK3_MASK = (1u << PK3_BITS) - 1u, PK3_FMASK = (1u << PK3_FIELD) - 1u;
__device__ __forceinline__ uint2 load_pk(const uint32_t* __restrict__ pk, int64_t v) { return reinterpret_cast<const uint2*>(pk)[v]; }
template <int D>
__device__ __forceinline__ uint32_t pk_field(const uint2 w, int k) {
  if constexpr (D == 2) return k == 0 ? w.x : w.y;
  const uint64_t q = (uint64_t)w.x | ((uint64_t)w.y << 32);
  return (uint32_t)(q >> (PK3_FIELD * k)) & PK3_FMASK;
}
template <int D>
__device__ __forceinline__ void unpack_coord(const uint2 w, int k, int& bin, double& t) {
  const uint32_t f = pk_field<D>(w, k);
  if constexpr (D == 2) { bin = (int)(f >> PK_BITS); t = (double)(f & PK_MASK) * (1.0 / (double)(1u << PK_BITS)); }
  else { bin = (int)(f >> PK3_BITS); t = (double)(f & PK3_MASK) * (1.0 / (double)(1u << PK3_BITS)); }
}
// the fraction alone, exact in fp32 (20 / 12 bits)
template <int D>
__device__ __forceinline__ float pk_fraction(const uint2 w, int k) {
  const uint32_t f = pk_field<D>(w, k);
  if constexpr (D == 2) return (float)(f & PK_MASK) * (1.0f / (float)(1u << PK_BITS));
  return (float)(f & PK3_MASK) * (1.0f / (float)(1u << PK3_BITS));
}

// The restriction every iteration runs.  Vertices are sorted by brick (B^D bins of the finest
// lattice) and, inside a brick, by bin.  One workgroup per brick:
//   A. all lanes stage the brick's vertices in LDS (value r/s, t per axis) with coalesced loads;
//   B. lane = bin (a brick has 64 bins = one wave; the 4 waves take every 4th vertex of a bin):
//      the 2^D corner sums of the bin are accumulated in registers -- no atomics, no conflicts;
//   C. a lattice node adds up its <= 2^D adjacent bins' corner sums (fixed order: reproducible
//      per brick), keeps the result in LDS for the fused coarser levels and flushes it with one
//      global atomic per node and brick.
// The first version accumulated with ds_add_f64 per vertex and corner and was bound by same-address
// LDS atomics (~11 vertices per bin): 101-150 us at C4 against ~65 us for this one.
template <int D> struct Brick { static constexpr int B = D == 3 ? 4 : 8; static constexpr int N1 = B + 1; static constexpr int NLOC = D == 3 ? N1 * N1 * N1 : N1 * N1; static constexpr int NBIN = 64; static constexpr int NC = 1 << D; };
// Vertices staged per pass = PF x 256, PF = entries per thread (bricks with more loop, unpipelined).  A 3-D brick of 4^3
// bins holds ~700 vertices (exactly 729 on the benchmark cube), a 2-D brick of 8^2 bins ~310; rounds 1-2 always staged 1024,
// so a quarter (3-D) to three quarters (2-D) of the staging loads fetched the clamped last entry.  Round 3: the launcher
// picks the smallest PF in 2..4 that holds the mesh's fullest brick (femo_pc::brick_pf).

// w_sorted[i] = 1/s of the i-th sorted vertex, 0 for pinned vertices (once per solve: s changes with
// every assembly).  Leaves one gather (the residual) in the restriction's staging loop.
// sinv[v] = 1/s of vertex v in mesh order: the mesh prolongation multiplies with the same rounded reciprocal (round 3: it
// divided per vertex and iteration, ~12 of its ~100 instructions)
__global__ void k_pc_weights(int64_t n, const int32_t* __restrict__ perm, const double* __restrict__ s,
                             const uint8_t* __restrict__ mask, float* __restrict__ w, float* __restrict__ sinv) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int32_t v = perm[i];
    const float r = (mask != nullptr && mask[v]) ? 0.0f : (float)(1.0 / s[v]);
    w[i] = r;
    sinv[v] = r;
  }
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains the wave's outstanding GLOBAL
// operations (s_waitcnt vmcnt(0)): in the first version of this kernel every barrier after a batch of lattice
// atomics waited a full round trip to L2 for them, and the loads prefetched for the next brick would be waited
// for as well.  Nothing here communicates through global memory inside a workgroup.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// Round 2: persistent workgroups, software-pipelined over bricks.  A brick's inputs are a chain of three
// dependent loads (brick table -> permutation -> gathered residual); one brick at a time the workgroup
// idled through each of them and, at every barrier, through the acknowledgement of its atomics, and then
// through two 27-tap loops of dependent LDS reads on a handful of lanes (the fused coarser levels): 13 us per
// brick, 118 us per launch at C4 with six workgroups per CU.  Now
//  * the table of brick k+3, the permutation of brick k+2 and the values of brick k+1 are in flight (in
//    registers) while brick k is reduced in LDS.  The number of atomics is data dependent, so every wait of the
//    compiler for a vector-memory result is a full drain (vmcnt(0), returns are counted in order): all global
//    traffic of an iteration -- the atomics of the PREVIOUS brick, flushed from a second copy of its node sums,
//    and the three prefetch stages -- is issued in one burst right after the single drain;
//  * the fused levels are separable 3-tap passes (x, y, z) with all taps of a pass issued together, run by wave 0
//    while the other waves already stage the next brick.
template <int D> struct BrickNodes {
  // node sums of a brick on the finest lattice and the fused coarser ones: 5^3 + 3^3 + 2^3 (9^2 + 5^2 + 3^2 + 2^2)
  static constexpr int NLEV = D == 3 ? 3 : 4;
  static constexpr int TOTAL = D == 3 ? 125 + 27 + 8 : 81 + 25 + 9 + 4;
  __host__ __device__ static constexpr int offset(int lev) {
    return D == 3 ? (lev == 0 ? 0 : (lev == 1 ? 125 : 152)) : (lev == 0 ? 0 : (lev == 1 ? 81 : (lev == 2 ? 106 : 115)));
  }
};

// value of lane (l ^ 1) [quad_perm 1,0,3,2 = 0xB1] or (l ^ 2) [2,3,0,1 = 0x4E]: DPP moves, no LDS crossbar
template <int CTRL>
__device__ __forceinline__ double quad_xor(double v) {
  const int lo = __double2loint(v), hi = __double2hiint(v);
  return __hiloint2double(__builtin_amdgcn_update_dpp(0, hi, CTRL, 0xF, 0xF, false),
                          __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xF, 0xF, false));
}

__device__ __forceinline__ void wave_lds_sync() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// coarse = R fine for one brick, R = tensor product of [1/2 1 1/2]; fine has (NF+1)^D nodes, coarse (NF/2+1)^D.
// One wave; tmpA / tmpB hold the intermediate passes.
template <int D, int NF>
__device__ __forceinline__ void brick_restrict_level(const double* fine, double* coarse, double* tmpA, double* tmpB, int lane) {
  constexpr int F1 = NF + 1, C1 = NF / 2 + 1;
  constexpr int NYZ = D == 3 ? F1 * F1 : F1;
  for (int o = lane; o < NYZ * C1; o += 64) {                 // x
    const int J = o % C1, yz = o / C1, x = 2 * J;
    const double* row = fine + yz * F1;
    const double a = row[x];
    const double l = x > 0 ? row[x - 1] : 0.0, r = x < NF ? row[x + 1] : 0.0;
    tmpA[o] = a + 0.5 * (l + r);
  }
  wave_lds_sync();
  constexpr int NZ = D == 3 ? F1 : 1;
  double* dstY = D == 3 ? tmpB : coarse;
  for (int o = lane; o < NZ * C1 * C1; o += 64) {             // y
    const int Jx = o % C1, Jy = (o / C1) % C1, z = o / (C1 * C1), y = 2 * Jy;
    const int base = (z * F1 + y) * C1 + Jx;
    const double a = tmpA[base];
    const double l = y > 0 ? tmpA[base - C1] : 0.0, r = y < NF ? tmpA[base + C1] : 0.0;
    dstY[o] = a + 0.5 * (l + r);
  }
  wave_lds_sync();
  if constexpr (D == 3) {
    for (int o = lane; o < C1 * C1 * C1; o += 64) {           // z
      const int Jz = o / (C1 * C1), rem = o % (C1 * C1), z = 2 * Jz;
      const int base = z * C1 * C1 + rem;
      const double a = tmpB[base];
      const double l = z > 0 ? tmpB[base - C1 * C1] : 0.0, r = z < NF ? tmpB[base + C1 * C1] : 0.0;
      coarse[o] = a + 0.5 * (l + r);
    }
    wave_lds_sync();
  }
}

// one global atomic per non-zero node sum of a finished brick on level LEV (NF = bins per axis of the brick there)
template <int D, int NF>
__device__ __forceinline__ void brick_flush_level(const double* src, double* gl, const int (&ln)[3], int b0, int b1, int b2, int tid) {
  constexpr int C1 = NF + 1;
  constexpr int NCL = D == 3 ? C1 * C1 * C1 : C1 * C1;
  for (int j = tid; j < NCL; j += FEMO_BLOCK) {
    const double a = src[j];
    const int J0 = j % C1, J1 = (j / C1) % C1, J2 = D == 3 ? j / (C1 * C1) : 0;
    const int i0 = b0 + J0, i1 = b1 + J1, i2 = D == 3 ? b2 + J2 : 0;
    if (a != 0.0 && i0 <= ln[0] && i1 <= ln[1] && i2 <= ln[2]) atomicAdd(&gl[node_index(ln, i0, i1, i2)], a);
  }
}

#ifndef FEMO_BRICK_WAVES
#define FEMO_BRICK_WAVES 4      // <= 128 VGPRs: four workgroups per CU instead of three (89 -> 84 us at C4)
#endif
template <int D, int PF>
__global__ __launch_bounds__(FEMO_BLOCK, FEMO_BRICK_WAVES) void k_restrict_bricks(int64_t n_bricks, const int64_t* __restrict__ brick_ptr,
                                                                const int32_t* __restrict__ brick_base, const uint32_t* __restrict__ bin_ptr,
                                                                const int32_t* __restrict__ perm, const uint32_t* __restrict__ pk,
                                                                Lat lat, const double* __restrict__ val,
                                                                const float* __restrict__ w_sorted,
                                                                double* __restrict__ g, int n_fused, const int32_t* __restrict__ done) {
  if (done != nullptr && *done) return;
  constexpr int B = Brick<D>::B, N1 = Brick<D>::N1, NLOC = Brick<D>::NLOC, NC = Brick<D>::NC;
  constexpr int BRICK_CHUNK = PF * FEMO_BLOCK;         // staged entries per pass
  constexpr int NTOT = BrickNodes<D>::TOTAL;
  __shared__ double nodes[2][NTOT];
  __shared__ double tmpA[128], tmpB[64];
  __shared__ double sval[BRICK_CHUNK];
  __shared__ float st[D][BRICK_CHUNK];          // 20-bit fractions are exact in fp32
  __shared__ double binsum[NC][64];
  const int tid = threadIdx.x;
  const int bin = tid >> 2, sub = tid & 3;      // 4 adjacent lanes share a bin
  struct Meta { int64_t start, end; int b0, b1, b2; uint32_t lo, hi; };
  auto load_meta = [&](int64_t brick) -> Meta {
    Meta M;
    const bool ok = brick < n_bricks;
    const int64_t bk = ok ? brick : 0;
    M.start = brick_ptr[bk];
    M.end = ok ? brick_ptr[bk + 1] : M.start;   // beyond the last brick: empty
    M.b0 = brick_base[bk * 3]; M.b1 = brick_base[bk * 3 + 1]; M.b2 = brick_base[bk * 3 + 2];
    const uint32_t* bp = bin_ptr + bk * 65;
    M.lo = bp[bin]; M.hi = bp[bin + 1];
    return M;
  };
  // entries beyond the end of a brick are clamped to its last one (index 0 for the empty bricks past the end of
  // the list) and get weight 0: no divergent branches around the loads
  auto entry = [&](const Meta& M, int q, bool& live) -> int64_t {
    const int64_t i = M.start + tid + q * FEMO_BLOCK;
    live = i < M.end;
    const int64_t last = M.end > 0 ? M.end - 1 : 0;
    return live ? i : last;
  };
  auto load_perm = [&](const Meta& M, int32_t (&p)[PF]) {
#pragma unroll
    for (int q = 0; q < PF; ++q) {
      bool live;
      p[q] = perm[entry(M, q, live)];
    }
  };
  auto load_vals = [&](const Meta& M, const int32_t (&p)[PF], double (&v)[PF], float (&w)[PF], uint2 (&k)[PF]) {
#pragma unroll
    for (int q = 0; q < PF; ++q) {
      bool live;
      const int64_t i = entry(M, q, live);
      v[q] = val[(int64_t)p[q]];
      const float ws = w_sorted[i];
      w[q] = live ? ws : 0.0f;
      k[q] = load_pk(pk, i);
    }
  };
  auto flush = [&](const double* nd, int b0, int b1, int b2) {
    double* gl = g;
    int ln[3] = {lat.n[0], lat.n[1], lat.n[2]};
    auto coarser = [&]() {
      ln[0] >>= 1; ln[1] >>= 1; ln[2] >>= 1;
      gl -= (int64_t)(ln[0] + 1) * (ln[1] + 1) * (ln[2] + 1);       // levels are stored coarsest first, contiguously
    };
    brick_flush_level<D, B>(nd, gl, ln, b0, b1, b2, tid);
    if (n_fused >= 1) { coarser(); brick_flush_level<D, B / 2>(nd + BrickNodes<D>::offset(1), gl, ln, b0 >> 1, b1 >> 1, b2 >> 1, tid); }
    if (n_fused >= 2) { coarser(); brick_flush_level<D, B / 4>(nd + BrickNodes<D>::offset(2), gl, ln, b0 >> 2, b1 >> 2, b2 >> 2, tid); }
    if constexpr (D == 2) {
      if (n_fused >= 3) { coarser(); brick_flush_level<D, B / 8>(nd + BrickNodes<D>::offset(3), gl, ln, b0 >> 3, b1 >> 3, b2 >> 3, tid); }
    }
  };
  const int64_t G = gridDim.x;
  Meta M0 = load_meta(blockIdx.x), M1 = load_meta(blockIdx.x + G), M2 = load_meta(blockIdx.x + 2 * G);
  int32_t p0[PF], p1[PF];
  load_perm(M0, p0);
  load_perm(M1, p1);
  double v0[PF];
  float w0[PF];
  uint2 k0[PF];
  load_vals(M0, p0, v0, w0, k0);
  int cur = 0;
  bool have_prev = false;
  int pb0 = 0, pb1 = 0, pb2 = 0;
  for (int64_t brick = blockIdx.x; brick < n_bricks; brick += G) {
    const int base[3] = {M0.b0, M0.b1, M0.b2};
    const int64_t start = M0.start, end = M0.end;
    const int64_t bin_lo = start + M0.lo, bin_hi = start + M0.hi;
    double c[NC];
#pragma unroll
    for (int q = 0; q < NC; ++q) c[q] = 0.0;
    // (1) the one drain of the iteration: the values of this brick (everything older has landed with them)
#pragma unroll
    for (int q = 0; q < PF; ++q) {
      const int j = tid + q * FEMO_BLOCK;
      sval[j] = v0[q] * (double)w0[q];
#pragma unroll
      for (int d = 0; d < D; ++d) st[d][j] = pk_fraction<D>(k0[q], d);
    }
    lds_barrier();            // staging visible; wave 0 has finished the fused levels of the previous brick
    // (2) all global traffic of the iteration in one burst: the previous brick's atomics ...
    if (have_prev) flush(nodes[cur ^ 1], pb0, pb1, pb2);
    // ... and the three prefetch stages (consumed one, two and three bricks later)
    const Meta M3 = load_meta(brick + 3 * G);
    int32_t p2[PF];
    load_perm(M2, p2);
    double v1[PF];
    float w1[PF];
    uint2 k1[PF];
    load_vals(M1, p1, v1, w1, k1);
    // (3) LDS phases
    for (int64_t chunk = start; chunk < end; chunk += BRICK_CHUNK) {
      const int64_t chunk_end = chunk + BRICK_CHUNK < end ? chunk + BRICK_CHUNK : end;
      if (chunk > start) {                        // bricks above BRICK_CHUNK vertices (rare): unpipelined passes
        lds_barrier();
        for (int64_t i = chunk + tid; i < chunk_end; i += FEMO_BLOCK) {
          sval[i - chunk] = val[perm[i]] * (double)w_sorted[i];
          const uint2 wk = load_pk(pk, i);
#pragma unroll
          for (int k = 0; k < D; ++k) st[k][i - chunk] = pk_fraction<D>(wk, k);
        }
        lds_barrier();
      }
      for (int64_t j = bin_lo + sub; j < bin_hi; j += 4) {
        if (j < chunk || j >= chunk_end) continue;
        const double r = sval[j - chunk];
        double t[D];
#pragma unroll
        for (int k = 0; k < D; ++k) t[k] = (double)st[k][j - chunk];
        // corner weights r * prod_k (t_k or 1 - t_k), built axis by axis
        double w[NC];
        w[0] = r;
#pragma unroll
        for (int k = 0; k < D; ++k) {
#pragma unroll
          for (int q = (1 << k) - 1; q >= 0; --q) {
            const double hi = w[q] * t[k];
            w[q | (1 << k)] = hi;
            w[q] = w[q] - hi;
          }
        }
#pragma unroll
        for (int q = 0; q < NC; ++q) c[q] += w[q];
      }
    }
#pragma unroll
    for (int q = 0; q < NC; ++q) {
      double v = c[q];
      v += quad_xor<0xB1>(v);                     // lanes 4b .. 4b+3 hold the partial sums of bin b
      v += quad_xor<0x4E>(v);
      if (sub == 0) binsum[q][bin] = v;
    }
    lds_barrier();
    double* nd = nodes[cur];
    for (int j = tid; j < NLOC; j += FEMO_BLOCK) {
      const int J[3] = {j % N1, (j / N1) % N1, D == 3 ? j / (N1 * N1) : 0};
      double a = 0.0;
#pragma unroll
      for (int q = 0; q < NC; ++q) {
        // the bin for which this node is corner q
        const int b0 = J[0] - (q & 1), b1 = J[1] - ((q >> 1) & 1), b2 = D == 3 ? J[2] - ((q >> 2) & 1) : 0;
        const bool in = !(b0 < 0 || b0 >= B || b1 < 0 || b1 >= B || b2 < 0 || (D == 3 && b2 >= B));
        const double t = in ? binsum[q][D == 3 ? (b2 * B + b1) * B + b0 : b1 * B + b0] : 0.0;
        a += t;
      }
      nd[j] = a;
    }
    lds_barrier();
    // the next n_fused coarser lattices straight from the LDS copy (the brick starts on a multiple of B bins, so
    // its nodes' parents on those levels are its own corner/edge/face nodes): wave 0, the others move on
    if (tid < 64) {
      if (n_fused >= 1) brick_restrict_level<D, B>(nd, nd + BrickNodes<D>::offset(1), tmpA, tmpB, tid);
      if (n_fused >= 2) brick_restrict_level<D, B / 2>(nd + BrickNodes<D>::offset(1), nd + BrickNodes<D>::offset(2), tmpA, tmpB, tid);
      if constexpr (D == 2) {
        if (n_fused >= 3) brick_restrict_level<D, B / 4>(nd + BrickNodes<D>::offset(2), nd + BrickNodes<D>::offset(3), tmpA, tmpB, tid);
      }
    }
    // rotate the pipeline
    have_prev = true; pb0 = base[0]; pb1 = base[1]; pb2 = base[2];
    cur ^= 1;
    M0 = M1; M1 = M2; M2 = M3;
#pragma unroll
    for (int q = 0; q < PF; ++q) {
      p1[q] = p2[q];
      v0[q] = v1[q]; w0[q] = w1[q];
      k0[q] = k1[q];
    }
  }
  lds_barrier();
  if (have_prev) flush(nodes[cur ^ 1], pb0, pb1, pb2);
}

// zh = rh + (1/s) P_L e_L, delivered as
//   mode 0: out = zh
//   mode 1: out = zh + beta out   (the next search direction; zh is never stored)
//   mode 2: out = zh              (first direction)
// In modes 1/2 gamma' = rh.zh = rho + g_L.e_L comes from the lattice (rho = rh.rh is folded here from the
// per-block partials of k_pcg_xr, or read from `rho` when it was reduced before; the g.e partials of the finest
// k_lattice_prolong are folded here as well, by every block in the same order), beta = gamma'/gamma_cur, and
// gamma' is left in *gamma_nxt for the next alpha.
// Stopping test of the PCG loop (st != nullptr): gamma' = r^T M^-1 r is the squared residual in the norm of
// the preconditioner, sqrt(gamma'/gamma_0) tracks the relative energy-norm error (measured: equal to within
// 10 % from the third iteration on), independently of the mesh size -- unlike r^T D^-1 r, whose ratio to the
// error grows like cond(D^-1 A) ~ n^2.  mode 2 stores tol^2 = rtol^2 * factor * gamma_0, mode 1 compares; every
// block takes the same decision from the same numbers, block 0 publishes it.
struct PcgStop {
  double rtol2_factor;     // rtol^2 (x b.D^-1 b / r0.D^-1 r0 for a non-zero initial guess)
  double atol_pc2;         // absolute threshold on gamma
  double* tolg2;           // device scalar: rtol^2 * gamma_0
  int32_t* flags;          // [0] stamp (it + 1) once converged, [1] iterations, [2] breakdown
  int it;
};
// merged BPX-PCG: where gamma' = r.r + sum_l C g^2 comes from (S == nullptr: the classic apply)
struct MergedScal {
  const double* S;
  int multi, init;
  int nb_lat; const double* lat_partials;   // carriers of k_lattice_coarse_m: levels T, L-1
  int nb_rr; const double* rr_partials;     // ... and r.r of the updated residual (one rank)
  double atol2;                             // absolute threshold on r.r (0: none)
};

// Partitioned meshes (round 5): the launch is split in two.  The FIRST (hf.n_verts > 0, hf.skip == nullptr) computes the new
// direction only on the vertices of the halo send list and writes it to the vector AND to the send buffer -- the pack kernel
// of the ghost refresh is gone and the exchange starts before the bulk of the vector exists; it takes every decision the full
// launch takes but leaves the scalars and flags alone.  The SECOND (hf.skip != nullptr) does everything else and skips those
// vertices (their old p is gone).
struct HaloFirst {
  int64_t n_verts; const int32_t* verts; const int32_t* slot_ptr; const int32_t* slots; double* send_buf;   // first launch
  const uint8_t* skip;                                                                                        // second launch
  // round 6, device-initiated refresh: the first launch stores the new direction of a send vertex straight into the
  // neighbours' inboxes (no send buffer) and every workgroup of it bumps their counters on its way out -- also when
  // the solve has converged and it has nothing to store (femo_internal.h: FemoHaloDirect)
  const FemoHaloPeers* peers; unsigned long long epoch;
  // n_send_blocks > 0: ONE launch plays both parts -- workgroups [0, n_send_blocks) the first (list walk, stores, counter
  // bumps), the others the second (everything else, send vertices skipped): one launch and one scalar prologue less
  int n_send_blocks;
};
template <int D>
__global__ __launch_bounds__(FEMO_BLOCK) void k_prolong_mesh(int64_t n_rows, Lat lat, const uint32_t* __restrict__ pk,
                                                             const double* __restrict__ rh, const float* __restrict__ sinv,
                                                             const uint8_t* __restrict__ mask, const double* __restrict__ e,
                                                             double* __restrict__ out, int mode, int nb_dot,
                                                             const double* __restrict__ dot_partials, const double* __restrict__ dot_global,
                                                             int nb_rho, const double* __restrict__ rho_partials, double* __restrict__ rho,
                                                             const double* __restrict__ gamma_cur, double* __restrict__ gamma_nxt,
                                                             const int32_t* __restrict__ done, PcgStop st, MergedScal ms, HaloFirst hf) {
  const bool both = hf.n_send_blocks > 0;
  const bool role_send = both ? (int)blockIdx.x < hf.n_send_blocks : hf.n_verts > 0;   // this workgroup walks the send list
  const int bid = (both && !role_send) ? (int)blockIdx.x - hf.n_send_blocks : (int)blockIdx.x;
  const int nblk = both ? (role_send ? hf.n_send_blocks : (int)gridDim.x - hf.n_send_blocks) : (int)gridDim.x;
  const bool signals = role_send && hf.peers != nullptr;            // ... and is a producer of a device-initiated ghost refresh
  if (done != nullptr && *done) {
    if (signals) femo_halo_signal(hf.peers);
    return;
  }
  __shared__ double lds[2 * (FEMO_BLOCK / 64)];
  double beta = 0.0;
  if (mode != 0) {
    double a = 0.0, b = 0.0;
    for (int i = threadIdx.x; i < nb_dot; i += FEMO_BLOCK) a += dot_partials[i];
    for (int i = threadIdx.x; i < nb_rho; i += FEMO_BLOCK) b += rho_partials[i];
    if (ms.S != nullptr) {
      for (int i = threadIdx.x; i < ms.nb_lat; i += FEMO_BLOCK) a += ms.lat_partials[i];
      for (int i = threadIdx.x; i < ms.nb_rr; i += FEMO_BLOCK) b += ms.rr_partials[i];
    }
    a = femo_wave_sum(a);
    b = femo_wave_sum(b);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) { lds[w] = a; lds[FEMO_BLOCK / 64 + w] = b; }
    __syncthreads();
    double ge = 0.0, rr = 0.0;
#pragma unroll
    for (int i = 0; i < FEMO_BLOCK / 64; ++i) { ge += lds[i]; rr += lds[FEMO_BLOCK / 64 + i]; }
    if (dot_global != nullptr) ge = *dot_global;      // already folded and all-reduced (nb_dot == 0)
    if (nb_rho == 0 && ms.S == nullptr) rr = *rho;    // reduced before (stopping test on r.D^-1 r, or several ranks)
    if (ms.S != nullptr) {
      // merged loop: the lattice dot is sum_l C g^2 -- the finest level's partials and the carriers' partials folded above,
      // the LDS-resident coarse levels from workgroup 0 of k_lattice_coarse_m; on N ranks the finest nodes a single rank
      // touches and r.r of the updated residual follow from the reduced scalars of this iteration's all-reduce
      const double alpha = ms.init ? -1.0 : ms.S[MS_ALPHA];
      ge += ms.S[MS_DOTC];
      if (ms.multi) {
        const double* R = ms.S + MS_RED;
        ge += R[4] - 2.0 * alpha * R[5] + alpha * alpha * R[6];
        rr = ms.init ? ms.S[MS_RR] : R[3] - 2.0 * alpha * R[1] + alpha * alpha * R[2];
      } else if (ms.init) {
        rr = ms.S[MS_RR];
      }
      if (rr < 0.0) rr = 0.0;
    }
    const double g1 = rr + ge, g0 = *gamma_cur;
    if (mode == 1) beta = g0 != 0.0 ? g1 / g0 : 0.0;
    const bool first = bid == 0 && threadIdx.x == 0 && !role_send;      // (the halo-first part writes no scalar)
    if (first) {
      *gamma_nxt = g1;
      if (nb_rho > 0 || ms.S != nullptr) *rho = rr;
    }
    if (st.flags != nullptr) {
      if (mode == 2) {
        const double t2 = fmax((ms.S != nullptr ? ms.S[MS_FACTOR] : st.rtol2_factor) * g1, st.atol_pc2);
        if (first) {
          *st.tolg2 = t2;
          if (g1 <= t2) {                             // the initial residual is already below the absolute threshold
            st.flags[1] = 0;
            st.flags[2] = 0;
            __threadfence();
            st.flags[0] = 1;
          }
        }
      } else {
        const bool bad = !(g1 == g1);
        const bool below_atol = ms.S != nullptr && ms.atol2 > 0.0 && rr <= ms.atol2;      // absolute test in the Jacobi norm
        if (g1 <= *st.tolg2 || below_atol || bad) {   // converged after st.it + 1 iterations: the direction is not needed
          if (first) {
            st.flags[1] = st.it + 1;
            st.flags[2] = bad ? 1 : 0;
            __threadfence();
            st.flags[0] = st.it + 1;
          }
          if (signals) femo_halo_signal(hf.peers);
          return;
        }
        if (first) st.flags[1] = st.it + 1;
      }
    }
  }
  const int64_t n_walk = role_send ? hf.n_verts : n_rows;
  // 36 B per vertex: r 8, packed coordinates 8, 1/s 4 (single precision, 0 on pinned vertices: the mask byte is not read),
  // p 8 + 8.  Rounds 1-4: 45 B (12 B of coordinates, 1/s in double precision, the mask).
  auto direction = [&](int64_t v) -> double {
    const uint2 wk = load_pk(pk, v);
    int i0[3] = {0, 0, 0};
    double t[3] = {0.0, 0.0, 0.0};
#pragma unroll
    for (int k = 0; k < D; ++k) unpack_coord<D>(wk, k, i0[k], t[k]);
    double sum = 0.0;
#pragma unroll
    for (int c = 0; c < (1 << D); ++c) {
      double w = 1.0;
      int ijk[3] = {i0[0], i0[1], i0[2]};
#pragma unroll
      for (int k = 0; k < D; ++k) {
        const int bit = (c >> k) & 1;
        w *= bit ? t[k] : 1.0 - t[k];
        ijk[k] += bit;
      }
      sum += w * e[node_index(lat.n, ijk[0], ijk[1], ijk[2])];
    }
    const float si = sinv[v];                               // the same rounded 1/s the restriction multiplies with (k_pc_weights)
    const double z = si == 0.0f ? rh[v] : rh[v] + sum * (double)si;   // pinned rows stay r even if the correction is not finite
    return mode == 1 ? z + beta * out[v] : z;
  };
  const int64_t stride = (int64_t)nblk * FEMO_BLOCK;
  for (int64_t w_i = (int64_t)bid * FEMO_BLOCK + threadIdx.x; w_i < n_walk; w_i += stride) {
    const int64_t v = role_send ? (int64_t)hf.verts[w_i] : w_i;
    if (!role_send && hf.skip != nullptr && hf.skip[v]) continue;
    const double pv = direction(v);
    out[v] = pv;
    if (role_send) {
      if (hf.peers != nullptr)
        for (int32_t q = hf.slot_ptr[w_i]; q < hf.slot_ptr[w_i + 1]; ++q) femo_halo_store(hf.peers, hf.epoch, hf.slots[q], pv);
      else
        for (int32_t q = hf.slot_ptr[w_i]; q < hf.slot_ptr[w_i + 1]; ++q) hf.send_buf[hf.slots[q]] = pv;
    }
  }
  if (signals) femo_halo_signal(hf.peers);
}

// coarse[I] = sum over the fine nodes 2I-1, 2I, 2I+1 (per axis) with weights 1/2, 1, 1/2
__device__ __forceinline__ double lattice_restrict_node(int64_t idx, const int* nc, const int* nf, int dim,
                                                        const double* __restrict__ fine) {
  const int i = (int)(idx % (nc[0] + 1));
  const int j = (int)((idx / (nc[0] + 1)) % (nc[1] + 1));
  const int k = (int)(idx / ((int64_t)(nc[0] + 1) * (nc[1] + 1)));
  // branch-free: out-of-range taps read a clamped node with weight 0, so all 9 / 27 reads are issued together
  // (with `continue` around them they were a chain of dependent round trips)
  double acc = 0.0;
  if (dim == 3) {
    double v[27], w[27];
#pragma unroll
    for (int t = 0; t < 27; ++t) {
      const int dx = t % 3 - 1, dy = (t / 3) % 3 - 1, dz = t / 9 - 1;
      const int fi = 2 * i + dx, fj = 2 * j + dy, fk = 2 * k + dz;
      const bool in = fi >= 0 && fi <= nf[0] && fj >= 0 && fj <= nf[1] && fk >= 0 && fk <= nf[2];
      const int ci = min(max(fi, 0), nf[0]), cj = min(max(fj, 0), nf[1]), ck = min(max(fk, 0), nf[2]);
      w[t] = in ? (dx ? 0.5 : 1.0) * (dy ? 0.5 : 1.0) * (dz ? 0.5 : 1.0) : 0.0;
      v[t] = fine[node_index(nf, ci, cj, ck)];
    }
#pragma unroll
    for (int t = 0; t < 27; ++t) acc += w[t] * v[t];
  } else {
    double v[9], w[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const int dx = t % 3 - 1, dy = t / 3 - 1;
      const int fi = 2 * i + dx, fj = 2 * j + dy;
      const bool in = fi >= 0 && fi <= nf[0] && fj >= 0 && fj <= nf[1];
      const int ci = min(max(fi, 0), nf[0]), cj = min(max(fj, 0), nf[1]);
      w[t] = in ? (dx ? 0.5 : 1.0) * (dy ? 0.5 : 1.0) : 0.0;
      v[t] = fine[node_index(nf, ci, cj, 0)];
    }
#pragma unroll
    for (int t = 0; t < 9; ++t) acc += w[t] * v[t];
  }
  return acc;
}

// (interpolation of e_c)(fine node idx)
__device__ __forceinline__ double lattice_interp_node(int64_t idx, const int* nf, const int* nc, int dim,
                                                      const double* __restrict__ ec) {
  const int i = (int)(idx % (nf[0] + 1));
  const int j = (int)((idx / (nf[0] + 1)) % (nf[1] + 1));
  const int k = (int)(idx / ((int64_t)(nf[0] + 1) * (nf[1] + 1)));
  const int ci[2] = {i >> 1, (i + 1) >> 1}, cj[2] = {j >> 1, (j + 1) >> 1};
  const int ck[2] = {dim == 3 ? k >> 1 : 0, dim == 3 ? (k + 1) >> 1 : 0};
  double acc = 0.0;
  for (int a = 0; a < 2; ++a)
    for (int b = 0; b < 2; ++b)
      for (int c = 0; c < 2; ++c) acc += ec[node_index(nc, ci[a], cj[b], ck[c])];
  return 0.125 * acc;   // even index: both parents coincide (2 x 1/2); odd: the two neighbours at 1/2 each
}

// 32-bit index arithmetic for the LDS-resident coarse levels (a few thousand nodes): the 64-bit divisions and products of
// the general helpers above are ~10x the instructions, and workgroup 0 of k_lattice_coarse_m is the critical path of an
// iteration on a partitioned mesh
__device__ __forceinline__ int node_index32(const int* n, int i, int j, int k) { return (k * (n[1] + 1) + j) * (n[0] + 1) + i; }
__device__ __forceinline__ double lattice_restrict_node32(int idx, const int* nc, const int* nf, int dim, const double* __restrict__ fine) {
  const int n0 = nc[0] + 1, n1 = nc[1] + 1;
  const int i = idx % n0, jk = idx / n0;
  const int j = jk % n1, k = jk / n1;
  double acc = 0.0;
  if (dim == 3) {
    double v[27], w[27];
#pragma unroll
    for (int t = 0; t < 27; ++t) {
      const int dx = t % 3 - 1, dy = (t / 3) % 3 - 1, dz = t / 9 - 1;
      const int fi = 2 * i + dx, fj = 2 * j + dy, fk = 2 * k + dz;
      const bool in = fi >= 0 && fi <= nf[0] && fj >= 0 && fj <= nf[1] && fk >= 0 && fk <= nf[2];
      const int ci = min(max(fi, 0), nf[0]), cj = min(max(fj, 0), nf[1]), ck = min(max(fk, 0), nf[2]);
      w[t] = in ? (dx ? 0.5 : 1.0) * (dy ? 0.5 : 1.0) * (dz ? 0.5 : 1.0) : 0.0;
      v[t] = fine[node_index32(nf, ci, cj, ck)];
    }
#pragma unroll
    for (int t = 0; t < 27; ++t) acc += w[t] * v[t];
  } else {
    double v[9], w[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const int dx = t % 3 - 1, dy = t / 3 - 1;
      const int fi = 2 * i + dx, fj = 2 * j + dy;
      const bool in = fi >= 0 && fi <= nf[0] && fj >= 0 && fj <= nf[1];
      const int ci = min(max(fi, 0), nf[0]), cj = min(max(fj, 0), nf[1]);
      w[t] = in ? (dx ? 0.5 : 1.0) * (dy ? 0.5 : 1.0) : 0.0;
      v[t] = fine[node_index32(nf, ci, cj, 0)];
    }
#pragma unroll
    for (int t = 0; t < 9; ++t) acc += w[t] * v[t];
  }
  return acc;
}
__device__ __forceinline__ double lattice_interp_node32(int idx, const int* nf, const int* nc, int dim, const double* __restrict__ ec) {
  const int n0 = nf[0] + 1, n1 = nf[1] + 1;
  const int i = idx % n0, jk = idx / n0;
  const int j = jk % n1, k = jk / n1;
  const int ci[2] = {i >> 1, (i + 1) >> 1}, cj[2] = {j >> 1, (j + 1) >> 1};
  const int ck[2] = {dim == 3 ? k >> 1 : 0, dim == 3 ? (k + 1) >> 1 : 0};
  double acc = 0.0;
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int c = 0; c < 2; ++c) acc += ec[node_index32(nc, ci[a], cj[b], ck[c])];
  return 0.125 * acc;
}

__global__ void k_lattice_restrict(int nc0, int nc1, int nc2, int nf0, int nf1, int nf2, int dim,
                                   const double* __restrict__ fine, double* __restrict__ coarse, const int32_t* __restrict__ done) {
  if (done != nullptr && *done) return;
  const int64_t total = (int64_t)(nc0 + 1) * (nc1 + 1) * (nc2 + 1);
  const int nc[3] = {nc0, nc1, nc2}, nf[3] = {nf0, nf1, nf2};
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x)
    coarse[idx] = lattice_restrict_node(idx, nc, nf, dim, fine);
}

// e_f[i] = (interpolation of e_c)(i) + coef_f[i] * g_f[i]     (e_c == nullptr: coarsest level)
// dot_partials != nullptr (finest level): per-block partial of g.e -- with it the caller has
// rh.zh = rh.rh + g_L.e_L without a pass over the mesh (see femo_pc_apply).  dot_weight (partitioned
// meshes with the sparse exchange): 1/(number of ranks touching the node) on the nodes this rank
// touches, 0 elsewhere, so that the per-rank dots add up to the global one.
__global__ __launch_bounds__(256) void k_lattice_prolong(int nf0, int nf1, int nf2, int nc0, int nc1, int nc2, int dim,
                                                         const double* __restrict__ ec, const double* __restrict__ coef, double* __restrict__ g,
                                                         int zero_g, double* __restrict__ ef, double* __restrict__ dot_partials,
                                                         const double* __restrict__ dot_weight, const int32_t* __restrict__ done) {
  if (done != nullptr && *done) return;
  __shared__ double lds[256 / 64];
  const int64_t total = (int64_t)(nf0 + 1) * (nf1 + 1) * (nf2 + 1);
  const int nc[3] = {nc0, nc1, nc2}, nf[3] = {nf0, nf1, nf2};
  double dot = 0.0;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const double gi = g[idx];
    double v = coef[idx] * gi;
    if (zero_g) g[idx] = 0.0;      // accumulated by atomics: left clean for the next restriction
    if (ec != nullptr) v += lattice_interp_node(idx, nf, nc, dim, ec);
    ef[idx] = v;
    dot += dot_weight != nullptr ? gi * v * dot_weight[idx] : gi * v;
  }
  if (dot_partials != nullptr) {
    const double t = femo_block_sum<256>(dot, lds);
    if (threadIdx.x == 0) dot_partials[blockIdx.x] = t;
  }
}

// The coarse end of the hierarchy in ONE single-workgroup launch: restrict g down from level `top`
// (filled by the brick kernel or the last multi-block restriction) to level 0, then build the
// corrections e_0 .. e_{top-1} on the way up.  These levels have a few thousand nodes; as separate
// launches they cost ~4.7 us each in launch latency alone (2 x top launches per iteration).
// Their g and e stay in LDS: with global memory between the phases the kernel took 36 us at C4
// (six dependent round trips), more than the launches it replaced.
// The top level of the single-workgroup coarse kernels lives in registers, FEMO_COARSE_TOPR entries per thread of 1024: up to
// 5120 nodes (round 4: was 4096 -- a 2-D lattice of 64 bins has 65^2 = 4225 nodes and a 3-D one of 16 bins 17^3 = 4913, and
// with those as top level the whole cycle fell back to one launch per lattice level: config 5 ran 14 launches per iteration).
constexpr int FEMO_COARSE_TOPR = 5;
constexpr int64_t FEMO_COARSE_TOP_MAX = 1024 * FEMO_COARSE_TOPR;

struct CoarseLevels {
  int n_levels;                 // levels 0 .. n_levels-1 are handled here; level n_levels is `top`
  int n[FEMO_PC_MAX_LEVELS][3];
  double* g[FEMO_PC_MAX_LEVELS];
  double* e[FEMO_PC_MAX_LEVELS];
  const double* coef[FEMO_PC_MAX_LEVELS];
  int64_t nodes[FEMO_PC_MAX_LEVELS];
  int64_t off[FEMO_PC_MAX_LEVELS];     // LDS offset (doubles) of level l: g at off, e at off + nodes
  int emit_top;                        // also e_top = coef_top g_top + I e_{top-1} (global), g_top cleared: the level the
                                       // brick kernel filled, so that no multi-block restriction / prolongation touches it
  int top_in_lds;                      // g_top is copied to LDS at off[top] first (<= FEMO_COARSE_TOP_MAX nodes and room for it)
  // restrict_top: g_top is not read but formed here, g_top = R g_finer (27-point restriction from the level the
  // brick kernel filled): the launch that did it (k_lattice_restrict, 5.6 us = one launch floor) is folded in
  const double* finer_g;
  int finer_n[3];
  int restrict_top;
};

__global__ __launch_bounds__(1024) void k_lattice_coarse(CoarseLevels L, int dim, const int32_t* __restrict__ done, FemoXUpdate xu) {
  if (done != nullptr && *done) return;
  if (blockIdx.x > 0) {
    // workgroups 1 .. gridDim.x - 1 exist only to carry the solver's x += alpha p (FemoXUpdate) on the compute units the
    // lattice work of workgroup 0 leaves idle
    const double alpha = *xu.alpha;
    const int64_t n2 = xu.n >> 1;
    double2* x2 = reinterpret_cast<double2*>(xu.x);
    const double2* p2 = reinterpret_cast<const double2*>(xu.p);
    const int64_t stride = (int64_t)(gridDim.x - 1) * 1024;
    for (int64_t i = (int64_t)(blockIdx.x - 1) * 1024 + threadIdx.x; i < n2; i += stride) {
      double2 xi = x2[i];
      const double2 pi = p2[i];
      xi.x += alpha * pi.x; xi.y += alpha * pi.y;
      x2[i] = xi;
    }
    if ((xu.n & 1) && blockIdx.x == 1 && threadIdx.x == 0) xu.x[xu.n - 1] += alpha * xu.p[xu.n - 1];
    return;
  }
  // g and e of the levels below `top` live in LDS for the whole launch (a few thousand nodes);
  // only e_{top-1}, which the next prolongation reads, goes back to global memory.
  // Round 2: every global read of the launch is issued before the first LDS phase -- g_top goes to LDS with
  // coalesced loads (the 27-point restriction read it from global memory: 27 dependent L2 round trips per
  // thread, most of the 15.7 us the launch took), the coefficients of all levels wait in registers -- and the
  // barriers order LDS only, so no phase waits for the stores of the one before.
  extern __shared__ double coarse_lds[];
  const int top = L.n_levels;
  const int tid = threadIdx.x;
  constexpr int TOPR = FEMO_COARSE_TOPR;                           // g_top / coef_top entries per thread kept in registers
  const int64_t n_top = L.nodes[top];
  double* g_top_lds = coarse_lds + L.off[top];      // the host reserves nodes[top] doubles there when they fit
  const bool top_in_lds = L.top_in_lds != 0;
  double gt[TOPR], ct[TOPR];
  if (top_in_lds) {
#pragma unroll
    for (int q = 0; q < TOPR; ++q) {
      const int64_t idx = tid + q * 1024;
      if (L.restrict_top) gt[q] = idx < n_top ? lattice_restrict_node(idx, L.n[top], L.finer_n, dim, L.finer_g) : 0.0;
      else gt[q] = idx < n_top ? L.g[top][idx] : 0.0;
      ct[q] = (L.emit_top && idx < n_top) ? L.coef[top][idx] : 0.0;
    }
  }
  // coefficient of level l at node tid (levels with <= 1024 nodes), fetched one phase ahead
  auto coef_of = [&](int l) -> double { return (l < top && L.nodes[l] <= 1024 && tid < L.nodes[l]) ? L.coef[l][tid] : 0.0; };
  double c_cur = coef_of(0);
  if (top_in_lds) {
#pragma unroll
    for (int q = 0; q < TOPR; ++q) {
      const int64_t idx = tid + q * 1024;
      if (idx < n_top) g_top_lds[idx] = gt[q];
    }
    lds_barrier();
  }
  for (int l = top - 1; l >= 0; --l) {
    const int64_t total = L.nodes[l];
    const double* fine = l + 1 == top ? (top_in_lds ? g_top_lds : L.g[top]) : coarse_lds + L.off[l + 1];
    double* gl = coarse_lds + L.off[l];
    for (int64_t idx = tid; idx < total; idx += 1024) gl[idx] = lattice_restrict_node(idx, L.n[l], L.n[l + 1], dim, fine);
    lds_barrier();
  }
  for (int l = 0; l < top; ++l) {
    const int64_t total = L.nodes[l];
    const double* gl = coarse_lds + L.off[l];
    double* el = coarse_lds + L.off[l] + L.nodes[l];
    const double* ec = l > 0 ? coarse_lds + L.off[l - 1] + L.nodes[l - 1] : nullptr;
    const double c_nxt = coef_of(l + 1);
    for (int64_t idx = tid; idx < total; idx += 1024) {
      double v = (total <= 1024 ? c_cur : L.coef[l][idx]) * gl[idx];
      if (l > 0) v += lattice_interp_node(idx, L.n[l], L.n[l - 1], dim, ec);
      el[idx] = v;
      if (l == top - 1 && !L.emit_top) L.e[l][idx] = v;
    }
    c_cur = c_nxt;
    lds_barrier();
  }
  if (L.emit_top) {
    const double* ec = coarse_lds + L.off[top - 1] + L.nodes[top - 1];
    if (top_in_lds) {
#pragma unroll
      for (int q = 0; q < TOPR; ++q) {
        const int64_t idx = tid + q * 1024;
        if (idx < n_top) {
          if (!L.restrict_top) L.g[top][idx] = 0.0;  // every restriction read of it came from the LDS copy
          L.e[top][idx] = ct[q] * gt[q] + lattice_interp_node(idx, L.n[top], L.n[top - 1], dim, ec);
        }
      }
    } else {
      for (int64_t idx = tid; idx < n_top; idx += 1024) {
        const double gi = L.g[top][idx];
        L.g[top][idx] = 0.0;                        // every restriction read of it happened before the barriers above
        L.e[top][idx] = L.coef[top][idx] * gi + lattice_interp_node(idx, L.n[top], L.n[top - 1], dim, ec);
      }
    }
  }
}

// The three finest levels in one launch: a workgroup takes a tile of the finest lattice (8^3 or 16^2 nodes) and
// builds in LDS the corrections of the two coarser levels on the tile's ancestors -- 3^3 / 5^2 nodes of level
// L-2, then 5^3 / 9^2 nodes of level L-1 (recomputed by neighbouring tiles: 2.4x resp. 3.4x those levels' work,
// which is 1/8 resp. 1/64 of the finest level's) -- and prolongs from there:
//   e_c = coef_c g_c + I e_cc (global)     e_m = coef_m g_m + I e_c     e_f = coef_f g_f + I e_m     dot += g_f e_f (w)
// g_f is cleared in place (one tile owns a node); g_m and g_c are read by several workgroups, so their OTHER
// copies are cleared (see femo_pc::g_alt): the restriction after the next accumulates into those.
struct FineLevels {
  int ncc[3], nc[3], nm[3], nf[3];
  const double *e_cc, *coef_c, *g_c, *coef_m, *g_m, *coef_f, *dot_weight;
  double *g_c_other, *g_m_other, *g_f, *e_f, *dot_partials;
};

template <int D>
__global__ __launch_bounds__(256) void k_lattice_prolong3(FineLevels P, const int32_t* __restrict__ done) {
  if (done != nullptr && *done) return;
  constexpr int TF = D == 3 ? 8 : 16, TM = TF / 2 + 1, TC = TF / 4 + 1;
  constexpr int NC = D == 3 ? TC * TC * TC : TC * TC, NM = D == 3 ? TM * TM * TM : TM * TM, NT = D == 3 ? TF * TF * TF : TF * TF;
  __shared__ double ec[NC];
  __shared__ double em[NM];
  __shared__ double red[256 / 64];
  int tn[3] = {1, 1, 1};
#pragma unroll
  for (int k = 0; k < D; ++k) tn[k] = (P.nf[k] + TF) / TF;             // nodes 0 .. nf[k]
  const int64_t n_tiles = (int64_t)tn[0] * tn[1] * tn[2];
  // value of the patch `src` (side TS, origin slo) interpolated to node fi of the next finer level
  auto from_patch = [](const double* src, int TS, const int (&fi)[3], const int (&slo)[3]) -> double {
    int a[3], b[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) { a[k] = (fi[k] >> 1) - slo[k]; b[k] = ((fi[k] + 1) >> 1) - slo[k]; }
    if constexpr (D == 3) {
      return 0.125 * (src[(a[2] * TS + a[1]) * TS + a[0]] + src[(a[2] * TS + a[1]) * TS + b[0]] + src[(a[2] * TS + b[1]) * TS + a[0]] +
                      src[(a[2] * TS + b[1]) * TS + b[0]] + src[(b[2] * TS + a[1]) * TS + a[0]] + src[(b[2] * TS + a[1]) * TS + b[0]] +
                      src[(b[2] * TS + b[1]) * TS + a[0]] + src[(b[2] * TS + b[1]) * TS + b[0]]);
    } else {
      return 0.25 * (src[a[1] * TS + a[0]] + src[a[1] * TS + b[0]] + src[b[1] * TS + a[0]] + src[b[1] * TS + b[0]]);
    }
  };
  double dot = 0.0;
  for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    int lo[3] = {0, 0, 0}, mlo[3] = {0, 0, 0}, clo[3] = {0, 0, 0};
    {
      int64_t t = tile;
#pragma unroll
      for (int k = 0; k < D; ++k) { lo[k] = (int)(t % tn[k]) * TF; t /= tn[k]; mlo[k] = lo[k] >> 1; clo[k] = lo[k] >> 2; }
    }
    // Round 2: all global reads of the tile are issued before the first LDS phase and the barriers order LDS only
    // (three dependent round trips and three store drains per tile before: 19 us per launch at C4).
    static_assert(NC <= 256 && NM <= 256, "one patch node per thread");
    constexpr int NQ = NT / 256;
    const int p = threadIdx.x;
    // level c (coarsest of the three): coef, g and the eight parents on level cc
    bool c_in = false, c_own = true;
    int64_t c_idx = 0;
    double c_coef = 0.0, c_g = 0.0, c_par = 0.0;
    if (p < NC) {
      int ci[3] = {0, 0, 0};
      int q = p;
      c_in = true;
#pragma unroll
      for (int k = 0; k < D; ++k) {
        const int pk = q % TC; q /= TC;
        ci[k] = clo[k] + pk;
        c_in = c_in && ci[k] <= P.nc[k];
        c_own = c_own && pk < TF / 4;                                   // fine node 4 ci lies in this tile
      }
      if (c_in) {
        c_idx = node_index(P.nc, ci[0], ci[1], ci[2]);
        c_coef = P.coef_c[c_idx];
        c_g = P.g_c[c_idx];
        c_par = lattice_interp_node(c_idx, P.nc, P.ncc, D, P.e_cc);
      }
    }
    // level m
    bool m_in = false, m_own = true;
    int64_t m_idx = 0;
    int mi[3] = {0, 0, 0};
    double m_coef = 0.0, m_g = 0.0;
    if (p < NM) {
      int q = p;
      m_in = true;
#pragma unroll
      for (int k = 0; k < D; ++k) {
        const int pk = q % TM; q /= TM;
        mi[k] = mlo[k] + pk;
        m_in = m_in && mi[k] <= P.nm[k];
        m_own = m_own && pk < TF / 2;                                   // fine node 2 mi lies in this tile
      }
      if (m_in) {
        m_idx = node_index(P.nm, mi[0], mi[1], mi[2]);
        m_coef = P.coef_m[m_idx];
        m_g = P.g_m[m_idx];
      }
    }
    // finest level
    bool f_in[NQ];
    int64_t f_idx[NQ];
    int fi[NQ][3];
    double f_g[NQ], f_coef[NQ], f_w[NQ];
#pragma unroll
    for (int r = 0; r < NQ; ++r) {
      int q = p + r * 256;
      f_in[r] = true;
      fi[r][0] = fi[r][1] = fi[r][2] = 0;
#pragma unroll
      for (int k = 0; k < D; ++k) {
        const int qk = q % TF; q /= TF;
        fi[r][k] = lo[k] + qk;
        f_in[r] = f_in[r] && fi[r][k] <= P.nf[k];
      }
      f_idx[r] = 0; f_g[r] = 0.0; f_coef[r] = 0.0; f_w[r] = 1.0;
      if (f_in[r]) {
        f_idx[r] = node_index(P.nf, fi[r][0], fi[r][1], fi[r][2]);
        f_g[r] = P.g_f[f_idx[r]];
        f_coef[r] = P.coef_f[f_idx[r]];
        if (P.dot_weight != nullptr) f_w[r] = P.dot_weight[f_idx[r]];
      }
    }
    // LDS phases
    if (p < NC) {
      ec[p] = c_in ? c_coef * c_g + c_par : 0.0;
      if (c_in && c_own) P.g_c_other[c_idx] = 0.0;
    }
    lds_barrier();
    if (p < NM) {
      em[p] = m_in ? m_coef * m_g + from_patch(ec, TC, mi, clo) : 0.0;
      if (m_in && m_own) P.g_m_other[m_idx] = 0.0;
    }
    lds_barrier();
#pragma unroll
    for (int r = 0; r < NQ; ++r) {
      if (!f_in[r]) continue;
      const double gi = f_g[r];
      P.g_f[f_idx[r]] = 0.0;
      const double v = f_coef[r] * gi + from_patch(em, TM, fi[r], mlo);
      P.e_f[f_idx[r]] = v;
      dot += P.dot_weight != nullptr ? gi * v * f_w[r] : gi * v;
    }
    lds_barrier();
  }
  if (P.dot_partials != nullptr) {
    const double t = femo_block_sum<256>(dot, red);
    if (threadIdx.x == 0) P.dot_partials[blockIdx.x] = t;
  }
}

// ---- merged BPX-PCG: the lattice part of one iteration with the restricted residual as state ---------------------------
// (femo_internal.h "merged BPX-PCG").  h = this apply's brick accumulators = P^T(q/s) (P^T(r0/s) in
// the first apply), gs = the state.  Three launches:
//   k_lattice_coarse_m   workgroup 0: alpha; gs_{T-1} -= alpha R h_T; levels 0 .. T-1 in LDS, e_{T-1} out, sum C g^2 -> S[MS_DOTC]
//                        the other workgroups: x += alpha p, r -= alpha q (+ partial r.r), gs -= alpha h on levels T and L-1
//   k_lattice_prolong3_m gs_L -= alpha h_L in place (one tile owns a node), e of the three finest levels, partial sum C g^2
//   k_prolong_mesh       gamma' = r.r + sum_l C g^2, beta, p = z + beta p, stopping test
// 27-point restriction of a 3-D lattice level as three 1-D passes (x, y, z: out = 0.5 f[2i-1] + f[2i] + 0.5 f[2i+1] per axis) through
// two LDS scratch arrays.  The 27-tap form costs one thread ~500 instructions of index arithmetic per node and only the coarse
// level's nodes have a thread (343 of 1024 for 13^3 -> 7^3): 3.7 us per level in workgroup 0 of k_lattice_coarse_m, the critical
// path of an iteration on small and partitioned meshes; the passes take ~40 instructions per output on up to all 1024 threads.
// Ends with a barrier (out is visible).  fine may be global or LDS.
__device__ __forceinline__ void lattice_restrict3_sep(const double* __restrict__ fine, const int* nf, const int* nc, double* out, double* t1, double* t2, int tid) {
  const int fx = nf[0] + 1, fy = nf[1] + 1, fz = nf[2] + 1, cx = nc[0] + 1, cy = nc[1] + 1, cz = nc[2] + 1;
  for (int idx = tid; idx < cx * fy * fz; idx += 1024) {
    const int i = idx % cx, r = idx / cx;
    const double* row = fine + r * fx;
    const int f = 2 * i;
    double v = row[f];
    if (f > 0) v += 0.5 * row[f - 1];
    if (f + 1 < fx) v += 0.5 * row[f + 1];
    t1[idx] = v;
  }
  lds_barrier();
  for (int idx = tid; idx < cx * cy * fz; idx += 1024) {
    const int i = idx % cx, r = idx / cx, j = r % cy, k = r / cy;
    const int f = 2 * j;
    const double* col = t1 + (k * fy) * cx + i;
    double v = col[f * cx];
    if (f > 0) v += 0.5 * col[(f - 1) * cx];
    if (f + 1 < fy) v += 0.5 * col[(f + 1) * cx];
    t2[idx] = v;
  }
  lds_barrier();
  const int cxy = cx * cy;
  for (int idx = tid; idx < cxy * cz; idx += 1024) {
    const int ij = idx % cxy, k = idx / cxy;
    const int f = 2 * k;
    double v = t2[f * cxy + ij];
    if (f > 0) v += 0.5 * t2[(f - 1) * cxy + ij];
    if (f + 1 < fz) v += 0.5 * t2[(f + 1) * cxy + ij];
    out[idx] = v;
  }
  lds_barrier();
}

// The kernel arguments of the lattice kernels are structs of 0.3-1.1 KB; the compiler fetches each field with a scalar load
// where it is first needed, and every first touch of a 64-byte line of the argument segment is a miss of the scalar cache
// (~0.5-1 us, one after the other along the critical path: 9 us between the launch and alpha in workgroup 0 of
// k_lattice_coarse_m, measured with early exits).  One dword of every line, loaded back to back at the kernel's entry, turns
// the chain into a single round trip.
template <int BYTES>
__device__ __forceinline__ void kernarg_warm() {
  typedef const __attribute__((address_space(4))) uint32_t* karg_ptr;
  karg_ptr ka = (karg_ptr)__builtin_amdgcn_kernarg_segment_ptr();
  static_assert(BYTES >= 64 && BYTES <= 20 * 64, "kernarg_warm covers 20 lines");
  constexpr int LAST = (BYTES - 4) & ~3;                      // never past the segment: lines beyond it repeat its last dword
#define FEMO_KA(i) ((i) * 64 < LAST ? (i) * 64 : LAST)
  uint32_t t0, t1, t2, t3, t4, t5, t6, t7, t8, t9;
  // (inline assembly: written as ordinary loads the compiler folds them into the argument values and drops them)
  asm volatile("s_load_dword %0, %10, %11\n\ts_load_dword %1, %10, %12\n\ts_load_dword %2, %10, %13\n\ts_load_dword %3, %10, %14\n\t"
               "s_load_dword %4, %10, %15\n\ts_load_dword %5, %10, %16\n\ts_load_dword %6, %10, %17\n\ts_load_dword %7, %10, %18\n\t"
               "s_load_dword %8, %10, %19\n\ts_load_dword %9, %10, %20\n\ts_waitcnt lgkmcnt(0)"
               : "=&s"(t0), "=&s"(t1), "=&s"(t2), "=&s"(t3), "=&s"(t4), "=&s"(t5), "=&s"(t6), "=&s"(t7), "=&s"(t8), "=&s"(t9)
               : "s"(ka), "n"(FEMO_KA(0)), "n"(FEMO_KA(1)), "n"(FEMO_KA(2)), "n"(FEMO_KA(3)), "n"(FEMO_KA(4)), "n"(FEMO_KA(5)), "n"(FEMO_KA(6)),
                 "n"(FEMO_KA(7)), "n"(FEMO_KA(8)), "n"(FEMO_KA(9)));
  if (BYTES > 10 * 64) {
    asm volatile("s_load_dword %0, %10, %11\n\ts_load_dword %1, %10, %12\n\ts_load_dword %2, %10, %13\n\ts_load_dword %3, %10, %14\n\t"
                 "s_load_dword %4, %10, %15\n\ts_load_dword %5, %10, %16\n\ts_load_dword %6, %10, %17\n\ts_load_dword %7, %10, %18\n\t"
                 "s_load_dword %8, %10, %19\n\ts_load_dword %9, %10, %20\n\ts_waitcnt lgkmcnt(0)"
                 : "=&s"(t0), "=&s"(t1), "=&s"(t2), "=&s"(t3), "=&s"(t4), "=&s"(t5), "=&s"(t6), "=&s"(t7), "=&s"(t8), "=&s"(t9)
                 : "s"(ka), "n"(FEMO_KA(10)), "n"(FEMO_KA(11)), "n"(FEMO_KA(12)), "n"(FEMO_KA(13)), "n"(FEMO_KA(14)), "n"(FEMO_KA(15)),
                   "n"(FEMO_KA(16)), "n"(FEMO_KA(17)), "n"(FEMO_KA(18)), "n"(FEMO_KA(19)));
  }
#undef FEMO_KA
}

struct MergedCarry {
  double* S;
  int cur, multi, init;
  int nb_pq, nb_pq2;
  const double *pq_partials, *pq_partials2;
  double *x, *r;
  const double *p, *q;
  int64_t n;
  double* rr_partials;
  // one rank: the brick-filled levels below the finest as ONE contiguous range of n_dense nodes (state gs_c, this apply's
  // accumulators h_c, coefficients coef_c -- below); h_c_other: the other parity's accumulators, cleared here for the
  // restriction after the next
  double* h_c_other;
  int64_t n_dense;
  double* lat_partials;
  double* gs_top;             // state of level T-1
  // N > 1 (round 5): the reduced buffer of this iteration [sums on the shared nodes of levels T, L-1, L | R h_T on level T-1 |
  // MS_NRED scalars] is read in place -- no unpack launch.  Indices address the contiguous range [T | L-1 | L] (gs_c, h_c,
  // coef_c: its state, this apply's accumulators, its coefficients).  Every rank keeps the state of ALL shared nodes and adds
  // their part of the lattice dot itself; of the nodes only this rank touches, those on levels T and L-1 are updated here
  // (int_idx), the finest level's tile by tile in k_lattice_prolong3_m.  Accumulators are cleared in place: workgroup 0 reads
  // R h_T from the buffer, nothing else reads them in this launch.
  const double* buf;
  int64_t n_shared; const int32_t* shared_idx;
  int64_t n_intc; const int32_t* int_idx;
  double* gs_c; double* h_c; const double* coef_c;
  int64_t n_top_buf;                // nodes of level T-1 (the dense part of the buffer)
  int dbg;                          // timing experiments (FEMO_TUNING builds): 1 = workgroup 0 returns early, 2 = the carriers do
  int64_t sep_off;                  // LDS offset (doubles) of the scratch of lattice_restrict3_sep, -1: the 27-tap restrictions
  int flat;                         // 3-D, 1 <= levels below T-1 <= 4, scratch fits: the chain below level T-1 in ONE down and ONE up step (below)
};

__global__ __launch_bounds__(1024) void k_lattice_coarse_m(CoarseLevels L, int dim, const int32_t* __restrict__ done, MergedCarry mc) {
  // every load the scalars need is issued before the first one is used: the exit flag, gamma and the partials were three
  // dependent round trips to L2 (~3 us at the head of a 14-30 us kernel that is the critical path of small iterations)
  kernarg_warm<sizeof(CoarseLevels) + 16 + sizeof(MergedCarry)>();
  __shared__ double red[1024 / 64];
  const int32_t stop = done != nullptr ? *done : 0;
  const double gamma = mc.init ? 0.0 : mc.S[MS_GAMMA + mc.cur];
  double a_pq = 0.0;
  if (!mc.init) {
    if (mc.multi) {
      a_pq = mc.buf[mc.n_shared + mc.n_top_buf];          // p.q, reduced
    } else {
      for (int i = threadIdx.x; i < mc.nb_pq; i += 1024) a_pq += mc.pq_partials[i];
      for (int i = threadIdx.x; i < mc.nb_pq2; i += 1024) a_pq += mc.pq_partials2[i];
    }
  }
  // workgroup 0: the operands of its top level as well (they do not depend on alpha)
  constexpr int TOPR = FEMO_COARSE_TOPR;
  double pre_h[TOPR], pre_g[TOPR], pre_c[TOPR];
  if (blockIdx.x == 0 && mc.dbg != 1) {
    const int top = L.n_levels;
    const int64_t n_top = L.nodes[top];
#pragma unroll
    for (int q = 0; q < TOPR; ++q) {
      const int64_t idx = threadIdx.x + q * 1024;
      const bool in = idx < n_top;
      pre_h[q] = in && mc.multi ? mc.buf[mc.n_shared + idx] : 0.0;     // N ranks: R h_T, restricted by the pack kernels and summed
      pre_g[q] = in ? mc.gs_top[idx] : 0.0;
      pre_c[q] = in ? L.coef[top][idx] : 0.0;
    }
  }
  if (stop) return;
  extern __shared__ double coarse_lds[];
  // one rank: R h_T for the top level by separable passes into LDS, before alpha is needed (it does not depend on alpha)
  double* const sep = mc.sep_off >= 0 ? coarse_lds + mc.sep_off : nullptr;
  if (blockIdx.x == 0 && mc.dbg != 1 && sep != nullptr && !mc.multi) {
    const int top = L.n_levels;
    const int64_t n_top = L.nodes[top];
    const int64_t s1 = (int64_t)(L.n[top][0] + 1) * (L.finer_n[1] + 1) * (L.finer_n[2] + 1);
    const int64_t s2 = (int64_t)(L.n[top][0] + 1) * (L.n[top][1] + 1) * (L.finer_n[2] + 1);
    lattice_restrict3_sep(L.finer_g, L.finer_n, L.n[top], sep + s1 + s2, sep, sep + s1, threadIdx.x);
#pragma unroll
    for (int q = 0; q < TOPR; ++q) {
      const int64_t idx = threadIdx.x + q * 1024;
      pre_h[q] = idx < n_top ? sep[s1 + s2 + idx] : 0.0;
    }
    lds_barrier();                                      // the scratch is reused by the levels below
  }
  double alpha = -1.0, pq = 0.0;
  if (!mc.init) {
    pq = mc.multi ? a_pq : femo_block_sum_all<1024>(a_pq, red);
    alpha = pq != 0.0 ? gamma / pq : 0.0;
  }
  if (blockIdx.x > 0) {
    if (mc.dbg == 2) return;
    const int nc = gridDim.x - 1, b = blockIdx.x - 1;
    double rr = 0.0, lat = 0.0;
    if (mc.init) {
      // first apply: nothing to update, but r.r of the initial residual is what the next iteration's all-reduce carries
      for (int64_t i = (int64_t)b * 1024 + threadIdx.x; i < mc.n; i += (int64_t)nc * 1024) rr += mc.r[i] * mc.r[i];
    } else {
      const int64_t n2 = mc.n >> 1;
      double2* x2 = reinterpret_cast<double2*>(mc.x);
      double2* r2 = reinterpret_cast<double2*>(mc.r);
      const double2* p2 = reinterpret_cast<const double2*>(mc.p);
      const double2* q2 = reinterpret_cast<const double2*>(mc.q);
      for (int64_t i = (int64_t)b * 1024 + threadIdx.x; i < n2; i += (int64_t)nc * 1024) {
        double2 xi = x2[i], ri = r2[i];
        const double2 pi = p2[i], qi = q2[i];
        xi.x += alpha * pi.x; xi.y += alpha * pi.y;
        ri.x -= alpha * qi.x; ri.y -= alpha * qi.y;
        x2[i] = xi; r2[i] = ri;
        rr += ri.x * ri.x + ri.y * ri.y;
      }
      if ((mc.n & 1) && b == 0 && threadIdx.x == 0) {
        const int64_t i = mc.n - 1;
        mc.x[i] += alpha * mc.p[i];
        const double ri = mc.r[i] - alpha * mc.q[i];
        mc.r[i] = ri;
        rr += ri * ri;
      }
    }
    if (!mc.multi) {
      // one rank: every brick-filled level but the finest, one contiguous range (two levels in 3-D, three in 2-D)
      for (int64_t i = (int64_t)b * 1024 + threadIdx.x; i < mc.n_dense; i += (int64_t)nc * 1024) {
        const double g = mc.gs_c[i] - alpha * mc.h_c[i];
        mc.gs_c[i] = g;
        mc.h_c_other[i] = 0.0;
        lat += mc.coef_c[i] * g * g;
      }
    } else {
      for (int64_t i = (int64_t)b * 1024 + threadIdx.x; i < mc.n_shared; i += (int64_t)nc * 1024) {
        const int32_t j = mc.shared_idx[i];
        const double g = mc.gs_c[j] - alpha * mc.buf[i];
        mc.gs_c[j] = g;
        mc.h_c[j] = 0.0;                     // k_lattice_prolong3_m then finds nothing to subtract on the finest level's shared nodes
        lat += mc.coef_c[j] * g * g;         // replicated: every rank adds the shared nodes' terms from the same numbers
      }
      for (int64_t i = (int64_t)b * 1024 + threadIdx.x; i < mc.n_intc; i += (int64_t)nc * 1024) {
        const int32_t j = mc.int_idx[i];
        mc.gs_c[j] -= alpha * mc.h_c[j];     // (their part of the lattice dot travelled as three scalars)
        mc.h_c[j] = 0.0;
      }
    }
    const double t0 = femo_block_sum<1024>(rr, red);
    if (threadIdx.x == 0) mc.rr_partials[b] = t0;
    const double t1 = femo_block_sum<1024>(lat, red);
    if (threadIdx.x == 0) mc.lat_partials[b] = t1;
    return;
  }
  if (threadIdx.x == 0) { mc.S[MS_ALPHA] = alpha; mc.S[MS_PQ] = pq; }
  // N ranks: the reduced scalars where k_prolong_mesh looks for them
  // (and the three sums the next pack launch accumulates with atomics are left zeroed)
  if (mc.multi && threadIdx.x < MS_NRED) {
    double* tail = const_cast<double*>(mc.buf) + mc.n_shared + mc.n_top_buf;
    mc.S[MS_RED + threadIdx.x] = tail[threadIdx.x];
    if (threadIdx.x >= 4) tail[threadIdx.x] = 0.0;
  }
  if (mc.dbg == 1 || mc.dbg == 3) return;
  // workgroup 0: the LDS-resident coarse end, as k_lattice_coarse with restrict_top and emit_top, on the updated state
  const int top = L.n_levels;
  const int tid = threadIdx.x;
  const int64_t n_top = L.nodes[top];
  double* g_top_lds = coarse_lds + L.off[top];
  double gt[TOPR], ct[TOPR];
  double dot = 0.0;
  // The 27-point restriction of h_T to level T-1 is ~14 us of index arithmetic when ONE compute unit does all 2197 nodes
  // of the 96^3 lattice's level (measured with the carriers switched off: 32 us for this workgroup, 21 without the
  // restriction; staging h_T in LDS first changed nothing -- it is instruction issue, not the gathers).  On N ranks the
  // pack kernels form it before the all-reduce (pre_h, from the reduced buffer); on one rank the carriers' mesh-sized
  // streams hide this workgroup at the sizes where the lattice is that large.
#pragma unroll
  for (int q = 0; q < TOPR; ++q) {
    const int64_t idx = tid + q * 1024;
    const bool in = idx < n_top;
    const double hres = !in ? 0.0 : ((mc.multi || sep != nullptr) ? pre_h[q] : lattice_restrict_node32((int)idx, L.n[top], L.finer_n, dim, L.finer_g));
    const double g0 = pre_g[q];
    ct[q] = pre_c[q];
    gt[q] = g0 - alpha * hres;
    dot += ct[q] * gt[q] * gt[q];
  }
  auto coef_of = [&](int l) -> double { return (l < top && L.nodes[l] <= 1024 && tid < L.nodes[l]) ? L.coef[l][tid] : 0.0; };
  double c_cur = coef_of(0);
#pragma unroll
  for (int q = 0; q < TOPR; ++q) {
    const int64_t idx = tid + q * 1024;
    if (idx < n_top) { g_top_lds[idx] = gt[q]; mc.gs_top[idx] = gt[q]; }
  }
  lds_barrier();
  if (mc.dbg == 4) return;
  if (mc.flat) {
    // Round 5: the levels below T-1 are linear images of g_{T-1}: level T-1-k is its restriction with the hat of half-width 2^k
    // per axis (R^k; clipping at the range ends commutes with the composition), and the correction on level T-1 is
    // C g_{T-1} + sum_k (trilinear interpolation of C_k g_k from level T-1-k) (I^k).  So instead of a chain of 3 barriers per
    // level down and one per level up (9 + 2 at the 13^3 -> 7^3 -> 4^3 lattices of C4: workgroup 0 is the critical path of an
    // iteration at <= 1.3 M rows), ALL levels go through the three separable passes together and one pass builds e_{T-1}:
    // 4 barriers.  Same operator, different association of the sums (the classic apply, which the oracle parity tests use,
    // keeps the level-by-level form).
    const int fx = L.n[top][0] + 1, fy = L.n[top][1] + 1, fz = L.n[top][2] + 1;
    double cpre[4];
#pragma unroll
    for (int k = 1; k <= 4; ++k) cpre[k - 1] = (k <= top && tid < L.nodes[top - k]) ? L.coef[top - k][tid] : 0.0;
    int64_t o = 0;
    for (int k = 1; k <= top; ++k) {                                   // x
      const int cx = L.n[top - k][0] + 1, w = 1 << k, cnt = cx * fy * fz;
      const double inv = 1.0 / w;
      double* t1 = sep + o;
      for (int idx = tid; idx < cnt; idx += 1024) {
        const int i = idx % cx, r = idx / cx, f0 = i << k;
        const double* row = g_top_lds + r * fx;
        double v = row[f0];
        for (int j = 1; j < w; ++j) {
          const double wt = 1.0 - j * inv;
          if (f0 - j >= 0) v += wt * row[f0 - j];
          if (f0 + j < fx) v += wt * row[f0 + j];
        }
        t1[idx] = v;
      }
      o += cnt;
    }
    lds_barrier();
    int64_t o1 = 0, o2 = o;
    for (int k = 1; k <= top; ++k) {                                   // y
      const int cx = L.n[top - k][0] + 1, cy = L.n[top - k][1] + 1, w = 1 << k, cnt = cx * cy * fz;
      const double inv = 1.0 / w;
      const double* t1 = sep + o1;
      double* t2 = sep + o2;
      for (int idx = tid; idx < cnt; idx += 1024) {
        const int i = idx % cx, r = idx / cx, j0 = r % cy, z = r / cy, f0 = j0 << k;
        const double* col = t1 + (z * fy) * cx + i;
        double v = col[f0 * cx];
        for (int j = 1; j < w; ++j) {
          const double wt = 1.0 - j * inv;
          if (f0 - j >= 0) v += wt * col[(f0 - j) * cx];
          if (f0 + j < fy) v += wt * col[(f0 + j) * cx];
        }
        t2[idx] = v;
      }
      o1 += cx * fy * fz; o2 += cnt;
    }
    lds_barrier();
    o2 = o;
    for (int k = 1; k <= top; ++k) {                                   // z; stores C g and adds C g^2
      const int l = top - k, cx = L.n[l][0] + 1, cy = L.n[l][1] + 1, cz = L.n[l][2] + 1, w = 1 << k, cxy = cx * cy;
      const double inv = 1.0 / w;
      const double* t2 = sep + o2;
      double* cg = coarse_lds + L.off[l];
      for (int idx = tid; idx < cxy * cz; idx += 1024) {               // (<= 1024 nodes per level: at most one trip, idx == tid)
        const int ij = idx % cxy, k0 = idx / cxy, f0 = k0 << k;
        double v = t2[f0 * cxy + ij];
        for (int j = 1; j < w; ++j) {
          const double wt = 1.0 - j * inv;
          if (f0 - j >= 0) v += wt * t2[(f0 - j) * cxy + ij];
          if (f0 + j < fz) v += wt * t2[(f0 + j) * cxy + ij];
        }
        const double c = cpre[k - 1] * v;
        dot += c * v;
        cg[idx] = c;
      }
      o2 += cxy * fz;
    }
    lds_barrier();
    if (mc.dbg == 6) return;
#pragma unroll
    for (int q = 0; q < TOPR; ++q) {
      const int idx = tid + q * 1024;
      if (idx < n_top) {
        const int i = idx % fx, jk = idx / fx, j = jk % fy, kk = jk / fy;
        double v = ct[q] * gt[q];
        for (int k = 1; k <= top; ++k) {
          const int l = top - k, cx = L.n[l][0] + 1, cy = L.n[l][1] + 1, w = 1 << k;
          const double inv = 1.0 / w;
          const double* cg = coarse_lds + L.off[l];
          const int a0 = i >> k, a1 = j >> k, a2 = kk >> k;
          const double t0 = (i & (w - 1)) * inv, t1 = (j & (w - 1)) * inv, t2 = (kk & (w - 1)) * inv;
          const int b0 = min(a0 + 1, L.n[l][0]), b1 = min(a1 + 1, L.n[l][1]), b2 = min(a2 + 1, L.n[l][2]);
          const double x00 = cg[(a2 * cy + a1) * cx + a0] * (1.0 - t0) + cg[(a2 * cy + a1) * cx + b0] * t0;
          const double x10 = cg[(a2 * cy + b1) * cx + a0] * (1.0 - t0) + cg[(a2 * cy + b1) * cx + b0] * t0;
          const double x01 = cg[(b2 * cy + a1) * cx + a0] * (1.0 - t0) + cg[(b2 * cy + a1) * cx + b0] * t0;
          const double x11 = cg[(b2 * cy + b1) * cx + a0] * (1.0 - t0) + cg[(b2 * cy + b1) * cx + b0] * t0;
          v += (x00 * (1.0 - t1) + x10 * t1) * (1.0 - t2) + (x01 * (1.0 - t1) + x11 * t1) * t2;
        }
        L.e[top][idx] = v;
      }
    }
    const double tf = femo_block_sum<1024>(dot, red);
    if (tid == 0) mc.S[MS_DOTC] = tf;
    return;
  }
  for (int l = top - 1; l >= 0; --l) {
    const int64_t total = L.nodes[l];
    const double* fine = l + 1 == top ? g_top_lds : coarse_lds + L.off[l + 1];
    double* gl = coarse_lds + L.off[l];
    if (sep != nullptr) {
      const int64_t s1 = (int64_t)(L.n[l][0] + 1) * (L.n[l + 1][1] + 1) * (L.n[l + 1][2] + 1);
      lattice_restrict3_sep(fine, L.n[l + 1], L.n[l], gl, sep, sep + s1, tid);
    } else {
      for (int64_t idx = tid; idx < total; idx += 1024) gl[idx] = lattice_restrict_node32((int)idx, L.n[l], L.n[l + 1], dim, fine);
      lds_barrier();
    }
  }
  if (mc.dbg == 5) return;
  for (int l = 0; l < top; ++l) {
    const int64_t total = L.nodes[l];
    const double* gl = coarse_lds + L.off[l];
    double* el = coarse_lds + L.off[l] + L.nodes[l];
    const double* ec = l > 0 ? coarse_lds + L.off[l - 1] + L.nodes[l - 1] : nullptr;
    const double c_nxt = coef_of(l + 1);
    for (int64_t idx = tid; idx < total; idx += 1024) {
      const double c = total <= 1024 ? c_cur : L.coef[l][idx];
      const double g = gl[idx];
      double v = c * g;
      dot += v * g;
      if (l > 0) v += lattice_interp_node32((int)idx, L.n[l], L.n[l - 1], dim, ec);
      el[idx] = v;
    }
    c_cur = c_nxt;
    lds_barrier();
  }
  if (mc.dbg == 6) return;
  if (top > 0) {
    const double* ec = coarse_lds + L.off[top - 1] + L.nodes[top - 1];
#pragma unroll
    for (int q = 0; q < TOPR; ++q) {
      const int64_t idx = tid + q * 1024;
      if (idx < n_top) L.e[top][idx] = ct[q] * gt[q] + lattice_interp_node32((int)idx, L.n[top], L.n[top - 1], dim, ec);
    }
  } else {                                          // level T-1 is the coarsest lattice: nothing below it
#pragma unroll
    for (int q = 0; q < TOPR; ++q) {
      const int64_t idx = tid + q * 1024;
      if (idx < n_top) L.e[top][idx] = ct[q] * gt[q];
    }
  }
  const double t = femo_block_sum<1024>(dot, red);
  if (tid == 0) mc.S[MS_DOTC] = t;
}

// FineLevels of the merged variant: g_c / g_m are the STATE of levels T and L-1 (already updated by the carriers of
// k_lattice_coarse_m), g_f the state of the finest level (updated here), h_f its accumulator (cleared here).  The other
// parity's accumulators of levels T and L-1 are cleared by the carriers (g_c_other / g_m_other are not used here).
// BT threads per tile (round 5): 128 where the tiles of a lattice outnumber the 2048 workgroups of 256 threads a device holds at once
// (96^3 bins = 13^3 = 2197 tiles: 149 workgroups took a second tile and the launch lasted two tile latencies, 27 us; with 128
// threads 2560 workgroups fit and every tile has its own: 25.1 -> 22.1 us; 2-D at n = 2236, 4225 tiles: 24.0 -> 20.8)
template <int D, int BT>
__global__ __launch_bounds__(BT) void k_lattice_prolong3_m(FineLevels P, double* __restrict__ h_f, const double* __restrict__ S, int init,
                                                            const int32_t* __restrict__ tile_list, int64_t n_list,
                                                            const int32_t* __restrict__ done) {
  if (done != nullptr && *done) return;
  constexpr int TF = D == 3 ? 8 : 16, TM = TF / 2 + 1, TC = TF / 4 + 1;
  constexpr int NC = D == 3 ? TC * TC * TC : TC * TC, NM = D == 3 ? TM * TM * TM : TM * TM, NT = D == 3 ? TF * TF * TF : TF * TF;
  __shared__ double ec[NC];
  __shared__ double em[NM];
  __shared__ double red[BT / 64];
  const double alpha = init ? -1.0 : S[MS_ALPHA];
  int tn[3] = {1, 1, 1};
#pragma unroll
  for (int k = 0; k < D; ++k) tn[k] = (P.nf[k] + TF) / TF;
  const int64_t n_tiles = (int64_t)tn[0] * tn[1] * tn[2];
  auto from_patch = [](const double* src, int TS, const int (&fi)[3], const int (&slo)[3]) -> double {
    int a[3], b[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) { a[k] = (fi[k] >> 1) - slo[k]; b[k] = ((fi[k] + 1) >> 1) - slo[k]; }
    if constexpr (D == 3) {
      return 0.125 * (src[(a[2] * TS + a[1]) * TS + a[0]] + src[(a[2] * TS + a[1]) * TS + b[0]] + src[(a[2] * TS + b[1]) * TS + a[0]] +
                      src[(a[2] * TS + b[1]) * TS + b[0]] + src[(b[2] * TS + a[1]) * TS + a[0]] + src[(b[2] * TS + a[1]) * TS + b[0]] +
                      src[(b[2] * TS + b[1]) * TS + a[0]] + src[(b[2] * TS + b[1]) * TS + b[0]]);
    } else {
      return 0.25 * (src[a[1] * TS + a[0]] + src[a[1] * TS + b[0]] + src[b[1] * TS + a[0]] + src[b[1] * TS + b[0]]);
    }
  };
  double dot = 0.0;
  // tile_list (N > 1): only the tiles that hold a node this rank touches
  const int64_t n_walk = tile_list != nullptr ? n_list : n_tiles;
  for (int64_t ti = blockIdx.x; ti < n_walk; ti += gridDim.x) {
    const int64_t tile = tile_list != nullptr ? (int64_t)tile_list[ti] : ti;
    int lo[3] = {0, 0, 0}, mlo[3] = {0, 0, 0}, clo[3] = {0, 0, 0};
    {
      int64_t t = tile;
#pragma unroll
      for (int k = 0; k < D; ++k) { lo[k] = (int)(t % tn[k]) * TF; t /= tn[k]; mlo[k] = lo[k] >> 1; clo[k] = lo[k] >> 2; }
    }
    static_assert(NC <= BT && NM <= BT && NT % BT == 0, "one patch node per thread");
    constexpr int NQ = NT / BT;
    const int p = threadIdx.x;
    bool c_in = false, c_own = true;
    int64_t c_idx = 0;
    double c_coef = 0.0, c_g = 0.0, c_par = 0.0;
    if (p < NC) {
      int ci[3] = {0, 0, 0};
      int q = p;
      c_in = true;
#pragma unroll
      for (int k = 0; k < D; ++k) {
        const int pk = q % TC; q /= TC;
        ci[k] = clo[k] + pk;
        c_in = c_in && ci[k] <= P.nc[k];
        c_own = c_own && pk < TF / 4;
      }
      if (c_in) {
        c_idx = node_index(P.nc, ci[0], ci[1], ci[2]);
        c_coef = P.coef_c[c_idx];
        c_g = P.g_c[c_idx];
        c_par = lattice_interp_node(c_idx, P.nc, P.ncc, D, P.e_cc);
      }
    }
    bool m_in = false, m_own = true;
    int64_t m_idx = 0;
    int mi[3] = {0, 0, 0};
    double m_coef = 0.0, m_g = 0.0;
    if (p < NM) {
      int q = p;
      m_in = true;
#pragma unroll
      for (int k = 0; k < D; ++k) {
        const int pk = q % TM; q /= TM;
        mi[k] = mlo[k] + pk;
        m_in = m_in && mi[k] <= P.nm[k];
        m_own = m_own && pk < TF / 2;
      }
      if (m_in) {
        m_idx = node_index(P.nm, mi[0], mi[1], mi[2]);
        m_coef = P.coef_m[m_idx];
        m_g = P.g_m[m_idx];
      }
    }
    bool f_in[NQ];
    int64_t f_idx[NQ];
    int fi[NQ][3];
    double f_g[NQ], f_coef[NQ], f_w[NQ];
#pragma unroll
    for (int r = 0; r < NQ; ++r) {
      int q = p + r * BT;
      f_in[r] = true;
      fi[r][0] = fi[r][1] = fi[r][2] = 0;
#pragma unroll
      for (int k = 0; k < D; ++k) {
        const int qk = q % TF; q /= TF;
        fi[r][k] = lo[k] + qk;
        f_in[r] = f_in[r] && fi[r][k] <= P.nf[k];
      }
      f_idx[r] = 0; f_g[r] = 0.0; f_coef[r] = 0.0; f_w[r] = 1.0;
      if (f_in[r]) {
        f_idx[r] = node_index(P.nf, fi[r][0], fi[r][1], fi[r][2]);
        f_g[r] = P.g_f[f_idx[r]] - alpha * h_f[f_idx[r]];
        f_coef[r] = P.coef_f[f_idx[r]];
        if (P.dot_weight != nullptr) f_w[r] = P.dot_weight[f_idx[r]];
      }
    }
    if (p < NC) {
      ec[p] = c_in ? c_coef * c_g + c_par : 0.0;
    }
    lds_barrier();
    if (p < NM) {
      em[p] = m_in ? m_coef * m_g + from_patch(ec, TC, mi, clo) : 0.0;
    }
    lds_barrier();
#pragma unroll
    for (int r = 0; r < NQ; ++r) {
      if (!f_in[r]) continue;
      const double gi = f_g[r];
      P.g_f[f_idx[r]] = gi;
      h_f[f_idx[r]] = 0.0;
      const double cg = f_coef[r] * gi;
      P.e_f[f_idx[r]] = cg + from_patch(em, TM, fi[r], mlo);
      dot += cg * gi * f_w[r];
    }
    lds_barrier();
  }
  if (P.dot_partials != nullptr) {
    const double t = femo_block_sum<BT>(dot, red);
    if (threadIdx.x == 0) P.dot_partials[blockIdx.x] = t;
  }
}

// N > 1: what the single all-reduce of an iteration carries -- [h on the shared nodes of levels T, L-1, L | R h_T on level T-1
// (dense; formed here from the rank's PARTIAL h_T: restriction is linear, so the sum over the ranks is R of the summed h_T) |
// 7 scalars].  The scalars: p.q, r.q, q.q (partials of the SpMV launches), r.r (partials of the previous iteration's
// carriers) -- folded by workgroup 0 from partials that earlier LAUNCHES wrote -- and, over the nodes of the three levels
// that only this rank touches, sum C g g, C g h, C h h: one fp64 atomic per workgroup and sum into the tail (which
// k_lattice_coarse_m leaves zeroed).  No workgroup waits for another: round 4's last-block ticket cost two device-scope
// fences on the critical path (21 us for this launch, 33 us with every thread fencing; measured round 5).  No unpack launch
// follows the all-reduce: k_lattice_coarse_m reads the sums straight from this buffer.
constexpr int PACK_GRID = 256;
struct PackArgs {
  int64_t n_shared; const int32_t* shared_idx;        // into the contiguous [T | L-1 | L] range
  const double* h;                                     // this apply's accumulators of that range
  int64_t n_top; int nc[3], nf[3], dim;                // level T-1 <- level T (which leads the range)
  int64_t n_int; const int32_t* int_idx; const double* gs; const double* coef;
  int nb_q[2]; const double* Pq[2];
  int nb_rr; const double* Prr;
  double* buf;
  int dbg;
};
__global__ __launch_bounds__(256) void k_pack_merged(PackArgs a, const int32_t* __restrict__ done) {
  if (done != nullptr && *done) return;
  __shared__ double red[256 / 64];
  // workgroup 0 folds, the others share the lists (one workgroup with both would be the launch's critical path)
  const int64_t stride = (int64_t)(gridDim.x - 1) * 256;
  const int64_t t0 = (int64_t)(blockIdx.x - 1) * 256 + threadIdx.x;
  double* tail = a.buf + a.n_shared + a.n_top;
  double t;
  if (blockIdx.x == 0) {
    // p.q, q.q, r.q: slots 0, 1, 2 of each SpMV launch's triple; tail order p.q, r.q, q.q, r.r, gg, gh, hh
    const int slot_of[3] = {0, 2, 1};
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int k = 0; k < 3; ++k)
      for (int l = 0; l < 2; ++l)
        for (int i = threadIdx.x; i < a.nb_q[l]; i += 256) acc[k] += a.Pq[l][(int64_t)slot_of[k] * FEMO_MAX_PARTIALS + i];
    for (int i = threadIdx.x; i < a.nb_rr; i += 256) acc[3] += a.Prr[i];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      t = femo_block_sum<256>(acc[k], red);
      if (threadIdx.x == 0) tail[k] = t;
    }
    return;
  }
  for (int64_t i = t0; i < a.n_shared; i += stride) a.buf[i] = a.h[a.shared_idx[i]];
  if (!(a.dbg & 2))
    for (int64_t i = t0; i < a.n_top; i += stride) a.buf[a.n_shared + i] = lattice_restrict_node32((int)i, a.nc, a.nf, a.dim, a.h);
  double gg = 0.0, gh = 0.0, hh = 0.0;
  if (!(a.dbg & 1))
    for (int64_t i = t0; i < a.n_int; i += stride) {
      const int32_t j = a.int_idx[i];
      const double c = a.coef[j], g = a.gs[j], h = a.h[j];
      gg += c * g * g; gh += c * g * h; hh += c * h * h;
    }
  if (t0 - threadIdx.x < a.n_int) {                  // workgroups that saw entries of the list (wave-uniform)
    t = femo_block_sum<256>(gg, red); if (threadIdx.x == 0) atomicAdd(&tail[4], t);
    t = femo_block_sum<256>(gh, red); if (threadIdx.x == 0) atomicAdd(&tail[5], t);
    t = femo_block_sum<256>(hh, red); if (threadIdx.x == 0) atomicAdd(&tail[6], t);
  }
}

// several regions cleared by one launch (a solve's set-up issued a dozen 5 us fills one after the other)
struct ZeroRegions { double* p[8]; int64_t n[8]; int count; };
__global__ void k_zero_regions(ZeroRegions z) {
  for (int r = 0; r < z.count; ++r)
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < z.n[r]; i += (int64_t)gridDim.x * blockDim.x) z.p[r][i] = 0.0;
}

// finest level: coef = C where the hat function is not dominated by Dirichlet vertices, else 0
__global__ void k_lattice_coef(int64_t nodes, double c, const double* __restrict__ w_free, const double* __restrict__ w_dir,
                               double* __restrict__ coef) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nodes; i += (int64_t)gridDim.x * blockDim.x) {
    const double f = w_free[i], d = w_dir[i];
    coef[i] = (f > 0.0 && d <= 0.3 * (f + d)) ? c : 0.0;
  }
}

// coarser levels: node I sits on fine node 2I and inherits its keep flag
__global__ void k_lattice_coef_inject(int nc0, int nc1, int nc2, int nf0, int nf1, int nf2, int dim, double c,
                                      const double* __restrict__ coef_f, double* __restrict__ coef_c) {
  const int64_t total = (int64_t)(nc0 + 1) * (nc1 + 1) * (nc2 + 1);
  const int nf[3] = {nf0, nf1, nf2};
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int i = (int)(idx % (nc0 + 1));
    const int j = (int)((idx / (nc0 + 1)) % (nc1 + 1));
    const int k = (int)(idx / ((int64_t)(nc0 + 1) * (nc1 + 1)));
    coef_c[idx] = coef_f[node_index(nf, 2 * i, 2 * j, dim == 3 ? 2 * k : 0)] != 0.0 ? c : 0.0;
  }
}

inline unsigned lat_grid(int64_t n) {
  int64_t g = (n + 255) / 256;
  return (unsigned)std::max<int64_t>(1, std::min<int64_t>(g, 2048));
}

Lat make_lat(const femo_pc* pc, const LatticeLevel& l) {
  Lat a;
  for (int k = 0; k < 3; ++k) {
    a.n[k] = l.n[k];
    a.lo[k] = pc->lo[k];
    a.inv_h[k] = k < pc->dim ? l.n[k] / (pc->hi[k] - pc->lo[k]) : 0.0;
  }
  return a;
}

}  // namespace

// -------------------------------------------------------------------------------------
int femo_pc_build(femo_mesh* m) {
  if (m->pc) return 0;
  femo_ctx* ctx = m->ctx;
  FEMO_REQUIRE(m->n_vert > 0, "empty mesh");
  // the plan is host work (pc_plan.cpp): lattice choice, packed coordinates, (brick, bin) sort -- 0.5 s at C4
  const int D = m->tdim;
  const int64_t nr = m->n_rows;
  std::vector<double> hx((size_t)std::max<int64_t>(nr * D, 1));
  FEMO_HIP_CHECK(hipMemcpyAsync(hx.data(), m->d_x, nr * D * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  FEMO_HIP_CHECK(hipStreamSynchronize(ctx->stream));
  FemoPcPlan P;
  FEMO_TRY(femo_pc_make_plan(D, nr, hx.data(), m->bbox_lo, m->bbox_hi, m->n_vert_global > 0 ? m->n_vert_global : m->n_vert,
                             femo_pc_spacing(), P));
  femo_pc* pc = new femo_pc();
  pc->dim = D;
  pc->n_levels = P.n_levels;
  pc->L.resize(P.n_levels);
  int64_t total = 0;
  for (int k = 0; k < 3; ++k) { pc->lo[k] = P.lo[k]; pc->hi[k] = P.hi[k]; }
  for (int l = 0; l < P.n_levels; ++l) {
    LatticeLevel& L = pc->L[l];
    for (int k = 0; k < 3; ++k) L.n[k] = P.n[l][k];
    L.nodes = P.nodes[l];
    L.H = P.H[l];
    total += L.nodes;
    FEMO_HIP_CHECK(hipMalloc(&L.e, L.nodes * sizeof(double)));
  }
  FEMO_HIP_CHECK(hipMalloc(&pc->coef_all, total * sizeof(double)));
  {
    int64_t off = 0;
    for (auto& L : pc->L) { L.coef = pc->coef_all + off; off += L.nodes; }
  }
  // every level once, then a second copy of the levels the brick kernel accumulates into (sized for the largest
  // n_fused: the three / four finest levels), see femo_pc::g_alt
  int64_t alt = 0;
  for (int l = std::max(0, P.n_levels - 1 - (D == 3 ? 2 : 3)); l < P.n_levels; ++l) alt += pc->L[l].nodes;
  FEMO_HIP_CHECK(hipMalloc(&pc->g_all, (total + alt) * sizeof(double)));
  FEMO_HIP_CHECK(hipMemset(pc->g_all, 0, (total + alt) * sizeof(double)));
  {
    int64_t off = 0;
    for (auto& L : pc->L) { L.g = pc->g_all + off; off += L.nodes; }
  }
  // Coarser levels the brick kernel restricts to by itself.  On one GPU fusing two (3-D) or three
  // (2-D) levels and running the lattice restrictions separately measure the same (72-75 ms per bench
  // cycle either way).  On partitioned meshes the bricks produce the finest level (exchanged sparsely,
  // pc_setup_shared) and the next one (summed densely); FEMO_BPX_FUSED overrides on one rank (tests, tuning).
  // (round 5: partitioned 2-D meshes fuse three levels like one rank does -- with two, level T-1 of the 1024^2 lattice of the
  // 5 M-DOF square has 129^2 nodes, beyond what the single-workgroup coarse kernel keeps, and the merged loop was refused)
  pc->n_fused = std::min(D == 3 ? 2 : 3, pc->n_levels - 1);
  if (const char* e = getenv("FEMO_BPX_FUSED"); e != nullptr && ctx->nranks == 1) pc->n_fused = std::max(0, std::min(std::min(D == 3 ? 2 : 3, pc->n_levels - 1), atoi(e)));
  {
    // second copy of the fused levels: the same layout as the first, right behind all levels
    pc->g_alt = pc->g_all + total;
  }
  pc->n_bricks = P.n_bricks;
  auto upload = [&](auto** dst, const auto& src) -> int {
    using T = typename std::remove_reference<decltype(src)>::type::value_type;
    FEMO_HIP_CHECK(hipMalloc(dst, std::max<size_t>(src.size(), 1) * sizeof(T)));
    if (!src.empty()) FEMO_HIP_CHECK(hipMemcpy(*dst, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice));
    return 0;
  };
  FEMO_TRY(upload(&pc->d_perm, P.perm));
  FEMO_TRY(upload(&pc->d_pk, P.pk));
  FEMO_TRY(upload(&pc->d_pk_sorted, P.pk_sorted));
  FEMO_TRY(upload(&pc->d_brick_ptr, P.brick_ptr));
  {
    int64_t fullest = 0;
    for (size_t b = 0; b + 1 < P.brick_ptr.size(); ++b) fullest = std::max<int64_t>(fullest, P.brick_ptr[b + 1] - P.brick_ptr[b]);
    pc->brick_pf = fullest <= 2 * FEMO_BLOCK ? 2 : (fullest <= 3 * FEMO_BLOCK ? 3 : 4);
    if (const char* env = FEMO_TUNE_ENV("FEMO_BRICK_PF")) pc->brick_pf = std::min(4, std::max(2, atoi(env)));
  }
  FEMO_TRY(upload(&pc->d_bin_ptr, P.bin_ptr));
  FEMO_TRY(upload(&pc->d_brick_base, P.brick_base));
  {
    // tiles of the fine-lattice kernel (8^3 / 16^2 finest nodes) that hold a node of one of this mesh's bricks: all of
    // them on one GPU with the mesh's own lattice, the rank's corner of the lattice on a partitioned mesh
    const LatticeLevel& Fl = pc->L[P.n_levels - 1];
    const int TF = D == 3 ? 8 : 16, BB = D == 3 ? 4 : 8;
    int tn[3] = {1, 1, 1};
    for (int k = 0; k < D; ++k) tn[k] = (Fl.n[k] + TF) / TF;
    pc->tile_mine.assign((size_t)tn[0] * tn[1] * tn[2], 0);
    for (int64_t b = 0; b < P.n_bricks; ++b) {
      int lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0};
      for (int k = 0; k < D; ++k) {
        const int b0 = P.brick_base[(size_t)b * 3 + k];
        lo[k] = b0 / TF;
        hi[k] = std::min(b0 + BB, Fl.n[k]) / TF;
      }
      for (int z = lo[2]; z <= hi[2]; ++z)
        for (int y = lo[1]; y <= hi[1]; ++y)
          for (int x = lo[0]; x <= hi[0]; ++x) pc->tile_mine[((size_t)z * tn[1] + y) * tn[0] + x] = 1;
    }
    std::vector<int32_t> tiles;
    for (size_t t = 0; t < pc->tile_mine.size(); ++t)
      if (pc->tile_mine[t]) tiles.push_back((int32_t)t);
    pc->n_my_tiles = (int64_t)tiles.size();
    FEMO_TRY(upload(&pc->d_my_tiles, tiles));
  }
  FEMO_HIP_CHECK(hipMalloc(&pc->d_w_sorted, std::max<size_t>(P.perm.size(), 1) * sizeof(float)));
  FEMO_HIP_CHECK(hipMalloc(&pc->d_sinv, std::max<size_t>(P.perm.size(), 1) * sizeof(float)));
  FEMO_HIP_CHECK(hipMalloc(&pc->d_dot_partials, 4096 * sizeof(double)));
  m->pc = pc;
  return 0;
}

void femo_pc_destroy(femo_mesh* m) {
  if (!m->pc) return;
  for (auto& L : m->pc->L) (void)hipFree(L.e);
  (void)hipFree(m->pc->coef_all);
  (void)hipFree(m->pc->g_all);
  (void)hipFree(m->pc->d_perm); (void)hipFree(m->pc->d_pk); (void)hipFree(m->pc->d_pk_sorted); (void)hipFree(m->pc->d_w_sorted); (void)hipFree(m->pc->d_sinv); (void)hipFree(m->pc->d_dot_partials);
  (void)hipFree(m->pc->d_shared_idx); (void)hipFree(m->pc->d_dot_weight); (void)hipFree(m->pc->d_xbuf); (void)hipFree(m->pc->d_dot_scalar);
  (void)hipFree(m->pc->d_brick_ptr); (void)hipFree(m->pc->d_bin_ptr); (void)hipFree(m->pc->d_brick_base);
  (void)hipFree(m->pc->gs); (void)hipFree(m->pc->d_lat_partials); (void)hipFree(m->pc->d_rr_partials);
  (void)hipFree(m->pc->d_my_tiles);
  (void)hipFree(m->pc->d_mshared_idx); (void)hipFree(m->pc->d_mint_idx); (void)hipFree(m->pc->d_mbuf);
  delete m->pc;
  m->pc = nullptr;
}

// ---- sparse exchange of the finest lattice on partitioned meshes ---------------------------
__global__ void k_mark_touched(int64_t n, const double* __restrict__ g, double* __restrict__ t) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) t[i] = g[i] != 0.0 ? 1.0 : 0.0;
}
template <class T>
__global__ void k_fill_ones(int64_t n, T* __restrict__ a) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) a[i] = 1.0;
}
// buf = [g_fine[shared_idx] | g_coarse | *piggy]: a scalar of the caller rides along (r.r of the PCG loop)
__global__ void k_pack_shared(int64_t n_shared, const int32_t* __restrict__ idx, const double* __restrict__ g_fine,
                              int64_t n_coarse, const double* __restrict__ g_coarse, const double* __restrict__ piggy,
                              double* __restrict__ buf, const int32_t* __restrict__ done) {
  if (done != nullptr && *done) return;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_shared + n_coarse; i += (int64_t)gridDim.x * blockDim.x)
    buf[i] = i < n_shared ? g_fine[idx[i]] : g_coarse[i - n_shared];
  if (piggy != nullptr && blockIdx.x == 0 && threadIdx.x == 0) buf[n_shared + n_coarse] = *piggy;
}
__global__ void k_unpack_shared(int64_t n_shared, const int32_t* __restrict__ idx, double* __restrict__ g_fine,
                                int64_t n_coarse, double* __restrict__ g_coarse, double* __restrict__ piggy,
                                const double* __restrict__ buf, const int32_t* __restrict__ done) {
  if (done != nullptr && *done) return;
  if (piggy != nullptr && blockIdx.x == 0 && threadIdx.x == 0) *piggy = buf[n_shared + n_coarse];
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_shared + n_coarse; i += (int64_t)gridDim.x * blockDim.x) {
    if (i < n_shared) g_fine[idx[i]] = buf[i];
    else g_coarse[i - n_shared] = buf[i];
  }
}
__global__ __launch_bounds__(1024) void k_fold_partials(int nb, const double* __restrict__ partials, double* __restrict__ out,
                                                        const int32_t* __restrict__ done) {
  if (done != nullptr && *done) return;
  __shared__ double lds[1024 / 64];
  double acc = 0.0;
  for (int i = threadIdx.x; i < nb; i += 1024) acc += partials[i];
  const double t = femo_block_sum<1024>(acc, lds);
  if (threadIdx.x == 0) out[0] = t;
}

// resident workgroups of the persistent brick kernel per CU (registers and LDS decide; asked once)
template <int D, int PF>
static int bricks_per_cu_of() {
  int nb = 0;
  const hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_restrict_bricks<D, PF>, FEMO_BLOCK, 0);
  return (e == hipSuccess && nb > 0) ? nb : 3;
}

static int bricks_per_cu(int dim, int pf) {
  static int cached[2][5] = {{0, 0, 0, 0, 0}, {0, 0, 0, 0, 0}};
  int& c = cached[dim == 3 ? 1 : 0][pf];
  if (c == 0) {
    if (dim == 3) c = pf == 2 ? bricks_per_cu_of<3, 2>() : (pf == 3 ? bricks_per_cu_of<3, 3>() : bricks_per_cu_of<3, 4>());
    else c = pf == 2 ? bricks_per_cu_of<2, 2>() : (pf == 3 ? bricks_per_cu_of<2, 3>() : bricks_per_cu_of<2, 4>());
    if (const char* env = FEMO_TUNE_ENV("FEMO_BRICKS_PER_CU")) c = std::max(1, atoi(env));
  }
  return c;
}

// the brick restriction for the mesh's dimension and staging depth
#define FEMO_LAUNCH_BRICKS(pc, gb, st, ...)                                                                                        \
  do {                                                                                                                             \
    if ((pc)->dim == 3) {                                                                                                          \
      if ((pc)->brick_pf == 2) hipLaunchKernelGGL((k_restrict_bricks<3, 2>), dim3(gb), dim3(FEMO_BLOCK), 0, st, __VA_ARGS__);      \
      else if ((pc)->brick_pf == 3) hipLaunchKernelGGL((k_restrict_bricks<3, 3>), dim3(gb), dim3(FEMO_BLOCK), 0, st, __VA_ARGS__); \
      else hipLaunchKernelGGL((k_restrict_bricks<3, 4>), dim3(gb), dim3(FEMO_BLOCK), 0, st, __VA_ARGS__);                          \
    } else {                                                                                                                       \
      if ((pc)->brick_pf == 2) hipLaunchKernelGGL((k_restrict_bricks<2, 2>), dim3(gb), dim3(FEMO_BLOCK), 0, st, __VA_ARGS__);      \
      else if ((pc)->brick_pf == 3) hipLaunchKernelGGL((k_restrict_bricks<2, 3>), dim3(gb), dim3(FEMO_BLOCK), 0, st, __VA_ARGS__); \
      else hipLaunchKernelGGL((k_restrict_bricks<2, 4>), dim3(gb), dim3(FEMO_BLOCK), 0, st, __VA_ARGS__);                          \
    }                                                                                                                              \
  } while (0)

// Which lattice nodes do several ranks touch?  A rank's restriction only reaches the nodes around its own vertices (on
// the finest lattice and, through the fused restrictions of the brick kernel, on the next coarser ones) and its
// prolongation only reads those, so between ranks it is enough to complete the sums on the nodes that more than one
// rank touches (the layers along the partition interfaces).  Collective, once per mesh.  Two sets of lists:
//   classic loop (femo_pc_apply): the finest level's shared nodes; the coarser fused levels travel whole;
//   merged loop (femo_pc_merged_apply, round 5): shared / single-rank nodes of ALL brick-filled levels (femo_pc::d_mshared_idx).
static bool merged_shape_ok(const femo_pc* pc, bool multi);
static int pc_setup_shared(femo_mesh* m) {
  femo_pc* pc = m->pc;
  if (pc->shared_ready) return 0;
  femo_ctx* ctx = m->ctx;
  hipStream_t st = ctx->stream;
  const int nl = pc->n_levels, nf = pc->n_fused;
  const int T = nl - 1 - nf;
  LatticeLevel& F = pc->L[nl - 1];
  const Lat lat = make_lat(pc, F);
  double* first = pc->L[T].g;                                // accumulators of levels T .. L (first copy), contiguous
  const int64_t n_all = (F.g + F.nodes) - first;
  const int64_t off_F = F.g - first;
  // touched pattern: restrict the constant 1 with unit weights (all interpolation weights are >= 0)
  double *ones = nullptr, *tmp = nullptr;
  FEMO_HIP_CHECK(hipMalloc(&ones, std::max<int64_t>(m->n_vert, 1) * sizeof(double)));
  FEMO_HIP_CHECK(hipMalloc(&tmp, n_all * sizeof(double)));
  hipLaunchKernelGGL(k_fill_ones<double>, dim3(lat_grid(m->n_vert)), dim3(256), 0, st, m->n_vert, ones);
  hipLaunchKernelGGL(k_fill_ones<float>, dim3(lat_grid(m->n_rows)), dim3(256), 0, st, std::max<int64_t>(m->n_rows, 0), pc->d_w_sorted);
  FEMO_HIP_CHECK(hipMemsetAsync(first, 0, n_all * sizeof(double), st));
  if (pc->n_bricks > 0) {
    const unsigned gb = (unsigned)std::min<int64_t>(pc->n_bricks, (int64_t)ctx->n_cu * bricks_per_cu(pc->dim, pc->brick_pf));
    FEMO_LAUNCH_BRICKS(pc, gb, st, pc->n_bricks, pc->d_brick_ptr, pc->d_brick_base, pc->d_bin_ptr, pc->d_perm, pc->d_pk_sorted, lat, ones, pc->d_w_sorted, F.g, nf, (const int32_t*)nullptr);
  }
  hipLaunchKernelGGL(k_mark_touched, dim3(lat_grid(n_all)), dim3(256), 0, st, n_all, first, tmp);
  FEMO_HIP_CHECK(hipGetLastError());
  std::vector<double> mine((size_t)n_all), cnt((size_t)n_all);
  FEMO_HIP_CHECK(hipMemcpyAsync(mine.data(), tmp, n_all * sizeof(double), hipMemcpyDeviceToHost, st));
  FEMO_TRY(femo_coll_allreduce(ctx, tmp, n_all, st));
  FEMO_HIP_CHECK(hipMemcpyAsync(cnt.data(), tmp, n_all * sizeof(double), hipMemcpyDeviceToHost, st));
  FEMO_HIP_CHECK(hipStreamSynchronize(st));
  (void)hipFree(ones);
  (void)hipFree(tmp);
  if (ctx->model) {
    // femo_comm_model (one context standing in for a rank of an N-GPU job): nobody else contributes touch counts.  The
    // nodes a real neighbour would share are the outer layers of the touched box that face the inside of the lattice --
    // two layers per cut face on every level (what the emulated 8-rank runs show for block partitions).
    for (int l = T; l < nl; ++l) {
      const LatticeLevel& Lv = pc->L[l];
      const int64_t off = Lv.g - first;
      const int64_t n0 = Lv.n[0] + 1, n1 = Lv.n[1] + 1;
      int lo[3] = {1 << 30, 1 << 30, 1 << 30}, hi[3] = {-1, -1, -1};
      for (int64_t i = 0; i < Lv.nodes; ++i) {
        if (mine[(size_t)(off + i)] == 0.0) continue;
        const int c[3] = {(int)(i % n0), (int)((i / n0) % n1), (int)(i / (n0 * n1))};
        for (int k = 0; k < 3; ++k) { lo[k] = std::min(lo[k], c[k]); hi[k] = std::max(hi[k], c[k]); }
      }
      for (int64_t i = 0; i < Lv.nodes; ++i) {
        if (mine[(size_t)(off + i)] == 0.0) continue;
        const int c[3] = {(int)(i % n0), (int)((i / n0) % n1), (int)(i / (n0 * n1))};
        bool sh = false;
        for (int k = 0; k < pc->dim; ++k) sh = sh || (hi[k] < Lv.n[k] && c[k] >= hi[k] - 1) || (lo[k] > 0 && c[k] <= lo[k] + 1);
        if (sh) cnt[(size_t)(off + i)] = 2.0;
      }
    }
  }
  // classic loop: the finest level
  std::vector<int32_t> shared;
  std::vector<double> weight((size_t)F.nodes, 0.0);
  for (int64_t i = 0; i < F.nodes; ++i) {
    if (cnt[(size_t)(off_F + i)] >= 1.5) shared.push_back((int32_t)i);
    // (model communicator: the all-reduce of the classic loop's weighted dot is the identity, so the rank's share is the whole)
    if (mine[(size_t)(off_F + i)] != 0.0) weight[(size_t)i] = ctx->model ? 1.0 : 1.0 / cnt[(size_t)(off_F + i)];
  }
  pc->n_shared = (int64_t)shared.size();
  auto upload_i32 = [&](int32_t** dst, const std::vector<int32_t>& v) -> int {
    FEMO_HIP_CHECK(hipMalloc(dst, std::max<size_t>(v.size(), 1) * sizeof(int32_t)));
    if (!v.empty()) FEMO_HIP_CHECK(hipMemcpy(*dst, v.data(), v.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    return 0;
  };
  if (merged_shape_ok(pc, true)) {
    FEMO_REQUIRE(n_all < (int64_t(1) << 31), "lattice too large for 32-bit node lists");
    std::vector<int32_t> msh, mint;
    const int64_t n_coarse_nodes = off_F;                     // levels T and L-1 lead the range
    int64_t n_int_coarse = 0;
    for (int64_t i = 0; i < n_all; ++i) {
      if (cnt[(size_t)i] >= 1.5) msh.push_back((int32_t)i);
      else if (mine[(size_t)i] != 0.0) { mint.push_back((int32_t)i); if (i < n_coarse_nodes) ++n_int_coarse; }
    }
    pc->n_mshared = (int64_t)msh.size(); pc->n_mint = (int64_t)mint.size(); pc->n_mint_coarse = n_int_coarse;
    FEMO_TRY(upload_i32(&pc->d_mshared_idx, msh));
    FEMO_TRY(upload_i32(&pc->d_mint_idx, mint));
    FEMO_HIP_CHECK(hipMalloc(&pc->d_mbuf, (pc->n_mshared + pc->L[T - 1].nodes + MS_NRED) * sizeof(double)));
    FEMO_HIP_CHECK(hipMemsetAsync(pc->d_mbuf, 0, (pc->n_mshared + pc->L[T - 1].nodes + MS_NRED) * sizeof(double), st));
  }
  const int64_t n_coarse = off_F;                             // classic loop: the coarser fused levels travel whole
  FEMO_TRY(upload_i32(&pc->d_shared_idx, shared));
  FEMO_HIP_CHECK(hipMalloc(&pc->d_dot_weight, F.nodes * sizeof(double)));
  FEMO_HIP_CHECK(hipMalloc(&pc->d_xbuf, (pc->n_shared + n_coarse + 1) * sizeof(double)));
  FEMO_HIP_CHECK(hipMalloc(&pc->d_dot_scalar, sizeof(double)));
  FEMO_HIP_CHECK(hipMemcpy(pc->d_dot_weight, weight.data(), F.nodes * sizeof(double), hipMemcpyHostToDevice));
  FEMO_HIP_CHECK(hipMemsetAsync(first, 0, n_all * sizeof(double), st));
  pc->shared_ready = true;
  return 0;
}

// coef arrays for the Dirichlet mask identified by `key` (mask == nullptr: no Dirichlet vertices)
static int pc_prepare(femo_mesh* m, const uint8_t* mask, uint64_t key) {
  femo_pc* pc = m->pc;
  if (pc->coef_valid && pc->built_key == key) return 0;
  femo_ctx* ctx = m->ctx;
  hipStream_t st = ctx->stream;
  const int gv = (int)lat_grid(m->n_rows);
  const int nl = pc->n_levels;
  LatticeLevel& F = pc->L[nl - 1];
  // hat-function mass on free (F.g) and Dirichlet (F.e) vertices of the finest lattice
  FEMO_HIP_CHECK(hipMemsetAsync(F.g, 0, F.nodes * sizeof(double), st));
  FEMO_HIP_CHECK(hipMemsetAsync(F.e, 0, F.nodes * sizeof(double), st));
  const Lat lat = make_lat(pc, F);
  const double* none = nullptr;
  if (pc->dim == 3) {
    hipLaunchKernelGGL(k_restrict_mesh<3>, dim3(gv), dim3(FEMO_BLOCK), 0, st, m->n_rows, lat, m->d_x, none, none, mask, 0, F.g, (const int32_t*)nullptr);
    if (mask) hipLaunchKernelGGL(k_restrict_mesh<3>, dim3(gv), dim3(FEMO_BLOCK), 0, st, m->n_rows, lat, m->d_x, none, none, mask, 1, F.e, (const int32_t*)nullptr);
  } else {
    hipLaunchKernelGGL(k_restrict_mesh<2>, dim3(gv), dim3(FEMO_BLOCK), 0, st, m->n_rows, lat, m->d_x, none, none, mask, 0, F.g, (const int32_t*)nullptr);
    if (mask) hipLaunchKernelGGL(k_restrict_mesh<2>, dim3(gv), dim3(FEMO_BLOCK), 0, st, m->n_rows, lat, m->d_x, none, none, mask, 1, F.e, (const int32_t*)nullptr);
  }
  if (ctx->nranks > 1) {
    FEMO_TRY(femo_coll_allreduce(ctx, F.g, F.nodes, st));
    FEMO_TRY(femo_coll_allreduce(ctx, F.e, F.nodes, st));
  }
  // 1 / diag of the Q1 Laplacian, damped: weighting the lattice terms by 0.6 against the Jacobi
  // term measured 12-15 % fewer iterations than 1.0 on every case tried (0.35 and 1.0 are both worse)
  constexpr double THETA = 0.6;
  auto level_c = [&](const LatticeLevel& L) { return THETA * (pc->dim == 3 ? 3.0 / (8.0 * L.H) : 3.0 / 8.0); };
  hipLaunchKernelGGL(k_lattice_coef, dim3(lat_grid(F.nodes)), dim3(256), 0, st, F.nodes, level_c(F), F.g, F.e, F.coef);
  for (int l = nl - 2; l >= 0; --l) {
    LatticeLevel& C = pc->L[l];
    const LatticeLevel& Fi = pc->L[l + 1];
    hipLaunchKernelGGL(k_lattice_coef_inject, dim3(lat_grid(C.nodes)), dim3(256), 0, st, C.n[0], C.n[1], C.n[2], Fi.n[0], Fi.n[1], Fi.n[2], pc->dim, level_c(C), Fi.coef, C.coef);
  }
  FEMO_HIP_CHECK(hipGetLastError());
  FEMO_HIP_CHECK(hipMemsetAsync(F.g, 0, F.nodes * sizeof(double), st));   // the restriction expects clean accumulators
  pc->built_key = key;
  pc->coef_valid = true;
  return 0;
}

// zh = M^-1 rh in scaled variables; partials[block] = rh.zh per block (gv blocks)
int femo_pc_apply(femo_mesh* m, const uint8_t* mask, uint64_t mask_key, const double* s, const double* rh, double* out,
                  int mode, double* rho, const double* gamma_cur, double* gamma_nxt, const int32_t* done, int gv,
                  bool rho_is_partial, const FemoPcgStop* stop, int nb_rho, const double* rho_partials, const FemoXUpdate* xupdate) {
  femo_pc* pc = m->pc;
  femo_ctx* ctx = m->ctx;
  FEMO_TRY(pc_prepare(m, mask, mask_key));
  const hipStream_t st = ctx->stream;
  const double* rsrc = rh;
  const int nl = pc->n_levels, nf = pc->n_fused;
  const int T = nl - 1 - nf;                                   // coarsest level the brick kernel fills
  LatticeLevel& F = pc->L[nl - 1];
  const Lat lat = make_lat(pc, F);
  // accumulators of this apply / of the next one (levels >= T exist twice, see femo_pc::g_alt)
  const int par = pc->parity;
  auto G = [&](int l, int which) -> double* {
    if (l < T || which == 0) return pc->L[l].g;
    return pc->g_alt + (pc->L[l].g - pc->L[T].g);
  };
  double* gF = G(nl - 1, par);
  // g of the finest nf+1 levels: zero on entry (femo_pc_begin, then the prolongation kernels clean up)
  if (pc->n_bricks > 0) {
    const unsigned gb = (unsigned)std::min<int64_t>(pc->n_bricks, (int64_t)ctx->n_cu * bricks_per_cu(pc->dim, pc->brick_pf));
    FEMO_LAUNCH_BRICKS(pc, gb, st, pc->n_bricks, pc->d_brick_ptr, pc->d_brick_base, pc->d_bin_ptr, pc->d_perm, pc->d_pk_sorted, lat, rsrc, pc->d_w_sorted, gF, nf, done);
  }
  const bool sparse = ctx->nranks > 1 && pc->shared_ready;
  if (sparse) {
    // one all-reduce per apply: the finest-level nodes several ranks touch + the whole coarser fused levels
    double* gc = nf >= 1 ? G(T, par) : nullptr;
    const int64_t n_coarse = nf >= 1 ? gF - gc : 0;
    const int64_t count = pc->n_shared + n_coarse;
    // rho_is_partial: *rho holds this rank's part of rh.rh; it rides in the same all-reduce
    double* piggy = rho_is_partial ? rho : nullptr;
    if (count > 0 || piggy != nullptr) {
      hipLaunchKernelGGL(k_pack_shared, dim3(lat_grid(std::max<int64_t>(count, 1))), dim3(256), 0, st, pc->n_shared, pc->d_shared_idx, gF, n_coarse, gc, piggy, pc->d_xbuf, done);
      FEMO_TRY(femo_coll_allreduce(ctx, pc->d_xbuf, count + (piggy ? 1 : 0), st));
      hipLaunchKernelGGL(k_unpack_shared, dim3(lat_grid(std::max<int64_t>(count, 1))), dim3(256), 0, st, pc->n_shared, pc->d_shared_idx, gF, n_coarse, gc, piggy, pc->d_xbuf, done);
    }
  } else if (rho_is_partial) {
    FEMO_REQUIRE(false, "femo_pc_apply: a partial rho needs the sparse exchange (femo_pc_can_piggyback)");
  } else if (ctx->nranks > 1) {   // dense: the contiguous accumulators of the finest nf+1 levels
    double* first = G(T, par);
    const int64_t count = (gF + F.nodes) - first;
    FEMO_TRY(femo_coll_allreduce(ctx, first, count, st));
  }
  const double* dotw = sparse ? pc->d_dot_weight : nullptr;
  int nb_dot = (int)lat_grid(F.nodes);
  // Fused lattice cycle, 3 launches instead of 6 (3-D): the restriction from the coarsest level the bricks filled
  // (multi-block: a single workgroup gathering 27 values per node of a 15 k-node level took 20 us), one workgroup
  // for everything below it, and one launch for the three finest levels.
  int64_t below = 0;
  for (int l = 0; l + 1 < T; ++l) below += pc->L[l].nodes;
  const bool fused_cycle = nf >= 2 && T >= 2 && T < FEMO_PC_MAX_LEVELS - 1 && below * 2 * (int64_t)sizeof(double) <= 144 * 1024 &&
                           pc->L[T - 1].nodes <= FEMO_COARSE_TOP_MAX && FEMO_TUNE_ENV("FEMO_BPX_UNFUSED_LATTICE") == nullptr;
  pc->fused_cycle_seen = fused_cycle;
  FEMO_REQUIRE(xupdate == nullptr || fused_cycle, "femo_pc_apply: the x update rides in the fused lattice cycle only");
  if (fused_cycle) {
    CoarseLevels CL;
    CL.n_levels = T - 1;                                           // levels 0 .. T-2 in LDS, e_{T-1} emitted
    CL.emit_top = 1;
    for (int l = 0; l <= T - 1; ++l) {
      for (int k = 0; k < 3; ++k) CL.n[l][k] = pc->L[l].n[k];
      CL.g[l] = pc->L[l].g; CL.e[l] = pc->L[l].e; CL.coef[l] = pc->L[l].coef;
      CL.nodes[l] = pc->L[l].nodes;
      CL.off[l] = l == 0 ? 0 : CL.off[l - 1] + 2 * CL.nodes[l - 1];
    }
    size_t lds = (size_t)below * 2 * sizeof(double);
    CL.top_in_lds = (CL.nodes[T - 1] <= FEMO_COARSE_TOP_MAX && lds + (size_t)CL.nodes[T - 1] * sizeof(double) <= 150 * 1024) ? 1 : 0;
    if (CL.top_in_lds) lds += (size_t)CL.nodes[T - 1] * sizeof(double);
    CL.restrict_top = CL.top_in_lds;
    CL.finer_g = G(T, par);
    for (int k = 0; k < 3; ++k) CL.finer_n[k] = pc->L[T].n[k];
    if (!CL.restrict_top) {                                     // no room for g_top in LDS: restrict it with a launch of its own
      LatticeLevel& C = pc->L[T - 1];
      const LatticeLevel& Fi = pc->L[T];
      hipLaunchKernelGGL(k_lattice_restrict, dim3(lat_grid(C.nodes)), dim3(256), 0, st, C.n[0], C.n[1], C.n[2], Fi.n[0], Fi.n[1], Fi.n[2], pc->dim, G(T, par), C.g, done);
    }
    if (lds > 64 * 1024 && !pc->coarse_lds_set) {
      FEMO_HIP_CHECK(hipFuncSetAttribute((const void*)k_lattice_coarse, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      pc->coarse_lds_set = true;
    }
    {
      // one workgroup does the lattice work; with an x update to carry, one more per remaining compute unit
      FemoXUpdate xu = {nullptr, nullptr, nullptr, 0};
      unsigned grid = 1;
      if (xupdate != nullptr && xupdate->n > 0) { xu = *xupdate; grid = (unsigned)std::max(2, ctx->n_cu); }
      hipLaunchKernelGGL(k_lattice_coarse, dim3(grid), dim3(1024), lds, st, CL, pc->dim, done, xu);
    }
    for (int l = T; l <= nl - 4; ++l) {                         // 2-D only (three fused levels): the level in between
      LatticeLevel& Fi = pc->L[l];
      hipLaunchKernelGGL(k_lattice_prolong, dim3(lat_grid(Fi.nodes)), dim3(256), 0, st, Fi.n[0], Fi.n[1], Fi.n[2], pc->L[l - 1].n[0], pc->L[l - 1].n[1], pc->L[l - 1].n[2], pc->dim, pc->L[l - 1].e, Fi.coef, G(l, par), 1, Fi.e, (double*)nullptr, (const double*)nullptr, done);
    }
    FineLevels FL;
    const LatticeLevel &Lcc = pc->L[nl - 4], &Lc = pc->L[nl - 3], &Lm = pc->L[nl - 2];
    for (int k = 0; k < 3; ++k) { FL.ncc[k] = Lcc.n[k]; FL.nc[k] = Lc.n[k]; FL.nm[k] = Lm.n[k]; FL.nf[k] = F.n[k]; }
    FL.e_cc = Lcc.e;
    FL.coef_c = Lc.coef; FL.g_c = G(nl - 3, par); FL.g_c_other = G(nl - 3, par ^ 1);
    FL.coef_m = Lm.coef; FL.g_m = G(nl - 2, par); FL.g_m_other = G(nl - 2, par ^ 1);
    FL.coef_f = F.coef; FL.g_f = gF; FL.e_f = F.e;
    FL.dot_partials = mode != 0 ? pc->d_dot_partials : nullptr;
    FL.dot_weight = dotw;
    const int TF = pc->dim == 3 ? 8 : 16;
    int64_t tiles = 1;
    for (int k = 0; k < pc->dim; ++k) tiles *= (F.n[k] + TF) / TF;
    nb_dot = (int)std::min<int64_t>(tiles, 2048);
    if (pc->dim == 3) hipLaunchKernelGGL(k_lattice_prolong3<3>, dim3(nb_dot), dim3(256), 0, st, FL, done);
    else hipLaunchKernelGGL(k_lattice_prolong3<2>, dim3(nb_dot), dim3(256), 0, st, FL, done);
    pc->parity ^= 1;
  } else {
    // levels with at most COARSE_NODES nodes (and below the brick-fused ones) go through the
    // single-workgroup kernel; `cut` = first level handled by multi-block launches
    // levels 0 .. cut-1 go through the single-workgroup kernel, which reads g of level `cut` from
    // global memory: one CU gathers 27 values per coarse node, so that level must stay small (with
    // 15.6 k nodes the phase alone took 25 us at C4)
    constexpr int64_t COARSE_TOP_NODES = FEMO_COARSE_TOP_MAX;
    int cut = 0;
    int64_t coarse_total = 0;
    while (cut < nl - 1 - nf && cut < FEMO_PC_MAX_LEVELS - 1 && pc->L[cut + 1].nodes <= COARSE_TOP_NODES) coarse_total += pc->L[cut++].nodes;
    for (int l = nl - 2 - nf; l >= cut; --l) {
      LatticeLevel& C = pc->L[l];
      const LatticeLevel& Fi = pc->L[l + 1];
      hipLaunchKernelGGL(k_lattice_restrict, dim3(lat_grid(C.nodes)), dim3(256), 0, st, C.n[0], C.n[1], C.n[2], Fi.n[0], Fi.n[1], Fi.n[2], pc->dim, G(l + 1, par), C.g, done);
    }
    if (cut > 0) {
      CoarseLevels CL;
      CL.n_levels = cut;
      CL.emit_top = 0;
      for (int l = 0; l <= cut; ++l) {
        for (int k = 0; k < 3; ++k) CL.n[l][k] = pc->L[l].n[k];
        CL.g[l] = G(l, par); CL.e[l] = pc->L[l].e; CL.coef[l] = pc->L[l].coef;
        CL.nodes[l] = pc->L[l].nodes;
        CL.off[l] = l == 0 ? 0 : CL.off[l - 1] + 2 * CL.nodes[l - 1];
      }
      size_t lds = (size_t)coarse_total * 2 * sizeof(double);
      CL.top_in_lds = (CL.nodes[cut] <= FEMO_COARSE_TOP_MAX && lds + (size_t)CL.nodes[cut] * sizeof(double) <= 150 * 1024) ? 1 : 0;
      if (CL.top_in_lds) lds += (size_t)CL.nodes[cut] * sizeof(double);
      CL.restrict_top = 0; CL.finer_g = nullptr; CL.finer_n[0] = CL.finer_n[1] = CL.finer_n[2] = 0;
      if (lds > 64 * 1024 && !pc->coarse_lds_set) {
        FEMO_HIP_CHECK(hipFuncSetAttribute((const void*)k_lattice_coarse, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        pc->coarse_lds_set = true;
      }
      hipLaunchKernelGGL(k_lattice_coarse, dim3(1), dim3(1024), lds, st, CL, pc->dim, done, FemoXUpdate{nullptr, nullptr, nullptr, 0});
    }
    for (int l = cut; l < nl; ++l) {
      LatticeLevel& Fi = pc->L[l];
      const double* ec = l > 0 ? pc->L[l - 1].e : nullptr;
      const int* nc = l > 0 ? pc->L[l - 1].n : Fi.n;
      double* dots = (l == nl - 1 && mode != 0) ? pc->d_dot_partials : nullptr;
      hipLaunchKernelGGL(k_lattice_prolong, dim3(lat_grid(Fi.nodes)), dim3(256), 0, st, Fi.n[0], Fi.n[1], Fi.n[2], nc[0], nc[1], nc[2], pc->dim, ec, Fi.coef, G(l, par), l >= T ? 1 : 0, Fi.e, dots, l == nl - 1 ? dotw : (const double*)nullptr, done);
    }
  }
  const double* dot_global = nullptr;
  if (sparse && mode != 0) {
    // each rank only holds the finest level on the nodes it touches: its weighted dot is a partial sum
    hipLaunchKernelGGL(k_fold_partials, dim3(1), dim3(1024), 0, st, nb_dot, pc->d_dot_partials, pc->d_dot_scalar, done);
    FEMO_TRY(femo_coll_allreduce(ctx, pc->d_dot_scalar, 1, st));
    dot_global = pc->d_dot_scalar;
    nb_dot = 0;
  }
  PcgStop ps;
  ps.rtol2_factor = stop ? stop->rtol2_factor : 0.0;
  ps.atol_pc2 = stop ? stop->atol_pc2 : 0.0;
  ps.tolg2 = stop ? stop->tolg2 : nullptr;
  ps.flags = stop ? stop->flags : nullptr;
  ps.it = stop ? stop->it : 0;
  if (pc->dim == 3)
    hipLaunchKernelGGL(k_prolong_mesh<3>, dim3(gv), dim3(FEMO_BLOCK), 0, st, m->n_rows, lat, pc->d_pk, rh, pc->d_sinv, mask, F.e, out, mode, nb_dot, pc->d_dot_partials, dot_global, nb_rho, rho_partials, rho, gamma_cur, gamma_nxt, done, ps, MergedScal{}, HaloFirst{});
  else
    hipLaunchKernelGGL(k_prolong_mesh<2>, dim3(gv), dim3(FEMO_BLOCK), 0, st, m->n_rows, lat, pc->d_pk, rh, pc->d_sinv, mask, F.e, out, mode, nb_dot, pc->d_dot_partials, dot_global, nb_rho, rho_partials, rho, gamma_cur, gamma_nxt, done, ps, MergedScal{}, HaloFirst{});
  FEMO_HIP_CHECK(hipGetLastError());
  return 0;
}

// start of a solve: per-vertex weights of the current operator; the atomically accumulated g arrays must be zero
int femo_pc_begin(femo_mesh* m, const double* s, const uint8_t* mask) {
  femo_pc* pc = m->pc;
  if (m->ctx->nranks > 1 && getenv("FEMO_BPX_DENSE_ALLREDUCE") == nullptr) FEMO_TRY(pc_setup_shared(m));   // once; uses d_w_sorted as scratch
  if (m->n_rows > 0) {
    hipLaunchKernelGGL(k_pc_weights, dim3(lat_grid(m->n_rows)), dim3(256), 0, m->ctx->stream, m->n_rows, pc->d_perm, s, mask, pc->d_w_sorted, pc->d_sinv);
    FEMO_HIP_CHECK(hipGetLastError());
  }
  const int nl = pc->n_levels;
  double* first = pc->L[nl - 1 - pc->n_fused].g;
  const int64_t count = (pc->L[nl - 1].g + pc->L[nl - 1].nodes) - first;
  FEMO_HIP_CHECK(hipMemsetAsync(first, 0, count * sizeof(double), m->ctx->stream));
  FEMO_HIP_CHECK(hipMemsetAsync(pc->g_alt, 0, count * sizeof(double), m->ctx->stream));
  pc->parity = 0;
  return 0;
}


// ---- merged BPX-PCG: host side ---------------------------------------------------------------------------------------
// the shape the merged kernels are written for: two brick-fused levels, everything below level T-1 (and g_{T-1}) in LDS
// (two fused levels in 3-D, three in 2-D: the carriers' loops and the node lists are generic over the brick-filled levels; in
// 2-D the level between T-1 and the tile kernel's three gets its correction from a k_lattice_prolong launch on the state)
static bool merged_shape_ok(const femo_pc* pc, bool multi) {
  const int nl = pc->n_levels, nf = pc->n_fused;
  const int T = nl - 1 - nf;
  (void)multi;
  if (!(nf == 2 || (nf == 3 && pc->dim == 2)) || T < 1 || T >= FEMO_PC_MAX_LEVELS - 1) return false;
  int64_t below = 0;
  for (int l = 0; l + 1 < T; ++l) below += pc->L[l].nodes;
  const int64_t lds = below * 2 * (int64_t)sizeof(double) + pc->L[T - 1].nodes * (int64_t)sizeof(double);
  return pc->L[T - 1].nodes <= FEMO_COARSE_TOP_MAX && below * 2 * (int64_t)sizeof(double) <= 144 * 1024 && lds <= 150 * 1024;
}

bool femo_pc_merged_ok(femo_mesh* m) {
  // FEMO_PCG_CLASSIC: A/B switch and the tests of the classic loop (read per solve, never per launch)
  if (femo_env_flag("FEMO_PCG_CLASSIC") || femo_env_flag("FEMO_BPX_DENSE_ALLREDUCE") || femo_env_flag("FEMO_BPX_UNFUSED_LATTICE")) return false;
  if (femo_pc_build(m) != 0) return false;
  return merged_shape_ok(m->pc, m->ctx->nranks > 1);
}

int femo_pc_merged_collectives(const femo_mesh* m) { return m->ctx->nranks > 1 ? 1 : 0; }

// start of a solve with the merged loop: weights of the current operator, clean accumulators, zero state
int femo_pc_merged_begin(femo_mesh* m, const double* s, const uint8_t* mask, const FemoZeroExtra* extra) {
  femo_pc* pc = m->pc;
  femo_ctx* ctx = m->ctx;
  const hipStream_t st = ctx->stream;
  if (ctx->nranks > 1) FEMO_TRY(pc_setup_shared(m));   // once; collective
  const int nl = pc->n_levels, T = nl - 1 - pc->n_fused;
  if (!pc->gs) {
    pc->gs_n = (pc->L[nl - 1].g + pc->L[nl - 1].nodes) - pc->L[T - 1].g;
    FEMO_HIP_CHECK(hipMalloc(&pc->gs, pc->gs_n * sizeof(double)));
    FEMO_HIP_CHECK(hipMalloc(&pc->d_lat_partials, FEMO_MAX_PARTIALS * sizeof(double)));
    FEMO_HIP_CHECK(hipMalloc(&pc->d_rr_partials, FEMO_MAX_PARTIALS * sizeof(double)));
  }
  if (m->n_rows > 0) hipLaunchKernelGGL(k_pc_weights, dim3(lat_grid(m->n_rows)), dim3(256), 0, st, m->n_rows, pc->d_perm, s, mask, pc->d_w_sorted, pc->d_sinv);
  double* first = pc->L[T].g;
  const int64_t count = (pc->L[nl - 1].g + pc->L[nl - 1].nodes) - first;
  ZeroRegions z;
  z.count = 3;
  z.p[0] = first; z.n[0] = count;
  z.p[1] = pc->g_alt; z.n[1] = count;
  z.p[2] = pc->gs; z.n[2] = pc->gs_n;
  if (ctx->nranks > 1 && pc->d_mbuf != nullptr) {
    // the scalar tail of the all-reduce buffer: k_pack_merged ADDS the three single-rank lattice sums into it and relies on
    // the coarse kernel of the same application to clear them again -- a solve that ended between the two (failed
    // collective, an early return) must not leak its sums into the next one
    z.p[3] = pc->d_mbuf + pc->n_mshared + pc->L[T - 1].nodes; z.n[3] = MS_NRED;
    z.count = 4;
  }
  // the caller's small fills (ghost tails of the Krylov vectors on a partitioned mesh) ride along: round 5's set-up issued
  // them as three memsets of 5 us each, one launch + one gap apiece
  if (extra != nullptr)
    for (int k = 0; k < extra->count && z.count < 8; ++k)
      if (extra->p[k] != nullptr && extra->n[k] > 0) { z.p[z.count] = extra->p[k]; z.n[z.count] = extra->n[k]; ++z.count; }
  hipLaunchKernelGGL(k_zero_regions, dim3(lat_grid(std::max(count, pc->gs_n))), dim3(256), 0, st, z);
  FEMO_HIP_CHECK(hipGetLastError());
  pc->parity = 0;
  return 0;
}

// One preconditioner application of the merged loop (see the kernels above).  V.q == nullptr: the first one.
int femo_pc_merged_apply(femo_mesh* m, const uint8_t* mask, uint64_t mask_key, const FemoMergedVecs& V, double* S,
                         const int32_t* done, const FemoPcgStop* stop) {
  femo_pc* pc = m->pc;
  femo_ctx* ctx = m->ctx;
  FEMO_TRY(pc_prepare(m, mask, mask_key));
  const hipStream_t st = ctx->stream;
  const int nl = pc->n_levels, nf = pc->n_fused;
  const int T = nl - 1 - nf;
  FEMO_REQUIRE(merged_shape_ok(pc, ctx->nranks > 1) && pc->gs != nullptr, "femo_pc_merged_apply: lattice shape not supported / begin not called");
  LatticeLevel& F = pc->L[nl - 1];
  const Lat lat = make_lat(pc, F);
  const int par = pc->parity;
  auto H = [&](int l, int which) -> double* { return which == 0 ? pc->L[l].g : pc->g_alt + (pc->L[l].g - pc->L[T].g); };
  auto GS = [&](int l) -> double* { return pc->gs + (pc->L[l].g - pc->L[T - 1].g); };
  const bool init = V.q == nullptr;
  const bool multi = ctx->nranks > 1;
  double* hF = H(nl - 1, par);
  if (pc->n_bricks > 0) {
    const unsigned gb = (unsigned)std::min<int64_t>(pc->n_bricks, (int64_t)ctx->n_cu * bricks_per_cu(pc->dim, pc->brick_pf));
    FEMO_LAUNCH_BRICKS(pc, gb, st, pc->n_bricks, pc->d_brick_ptr, pc->d_brick_base, pc->d_bin_ptr, pc->d_perm, pc->d_pk_sorted, lat, init ? (const double*)V.r : V.q, pc->d_w_sorted, hF, nf, done);
  }
  const int64_t n_top = pc->L[T - 1].nodes;
  if (multi) {
    FEMO_REQUIRE(pc->shared_ready && pc->d_mbuf != nullptr, "femo_pc_merged_apply: the sparse lattice exchange is not set up");
    PackArgs a;
    a.n_shared = pc->n_mshared; a.shared_idx = pc->d_mshared_idx; a.h = H(T, par);
    a.n_top = n_top; a.dim = pc->dim;
    for (int k = 0; k < 3; ++k) { a.nc[k] = pc->L[T - 1].n[k]; a.nf[k] = pc->L[T].n[k]; }
    a.n_int = pc->n_mint; a.int_idx = pc->d_mint_idx; a.gs = GS(T); a.coef = pc->L[T].coef;
    for (int l = 0; l < 2; ++l) { a.nb_q[l] = init ? 0 : V.nb_q[l]; a.Pq[l] = V.Pq[l]; }
    a.nb_rr = init ? 0 : std::max(1, ctx->n_cu - 1); a.Prr = pc->d_rr_partials;
    a.buf = pc->d_mbuf;
    static const int dbg_pack = FEMO_TUNE_ENV("FEMO_DEBUG_PACK") ? atoi(FEMO_TUNE_ENV("FEMO_DEBUG_PACK")) : 0;
    a.dbg = dbg_pack;
    const int64_t work = std::max(std::max(a.n_shared, a.n_int), n_top);
    const unsigned gp = 1u + (unsigned)std::min<int64_t>(PACK_GRID, std::max<int64_t>(1, (work + 255) / 256));
    hipLaunchKernelGGL(k_pack_merged, dim3(gp), dim3(256), 0, st, a, done);
    FEMO_TRY(femo_coll_allreduce(ctx, pc->d_mbuf, a.n_shared + n_top + MS_NRED, st));
  }
  // workgroup 0: the LDS-resident levels; the others: vector and lattice updates
  CoarseLevels CL;
  CL.n_levels = T - 1;
  CL.emit_top = 1;
  int64_t below = 0;
  for (int l = 0; l <= T - 1; ++l) {
    for (int k = 0; k < 3; ++k) CL.n[l][k] = pc->L[l].n[k];
    CL.g[l] = pc->L[l].g; CL.e[l] = pc->L[l].e; CL.coef[l] = pc->L[l].coef;
    CL.nodes[l] = pc->L[l].nodes;
    CL.off[l] = l == 0 ? 0 : CL.off[l - 1] + 2 * CL.nodes[l - 1];
    if (l + 1 < T) below += pc->L[l].nodes;
  }
  const size_t lds = (size_t)below * 2 * sizeof(double) + (size_t)CL.nodes[T - 1] * sizeof(double);
  CL.top_in_lds = 1; CL.restrict_top = 1;
  CL.finer_g = H(T, par);
  for (int k = 0; k < 3; ++k) CL.finer_n[k] = pc->L[T].n[k];
  if (lds > 64 * 1024 && !pc->coarse_lds_set) {
    FEMO_HIP_CHECK(hipFuncSetAttribute((const void*)k_lattice_coarse, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    pc->coarse_lds_set = true;
  }
  const int n_carry = std::max(1, ctx->n_cu - 1);
  MergedCarry mc;
  mc.S = S; mc.cur = V.cur; mc.multi = multi ? 1 : 0; mc.init = init ? 1 : 0;
  mc.nb_pq = V.nb_q[0]; mc.nb_pq2 = V.nb_q[1]; mc.pq_partials = V.Pq[0]; mc.pq_partials2 = V.Pq[1];
  mc.x = V.x; mc.r = V.r; mc.p = V.p; mc.q = V.q; mc.n = V.n;
  mc.rr_partials = pc->d_rr_partials; mc.lat_partials = pc->d_lat_partials;
  mc.h_c_other = H(T, par ^ 1);
  mc.n_dense = hF - H(T, par);
  mc.gs_top = GS(T - 1);
  mc.buf = multi ? pc->d_mbuf : nullptr;
  mc.n_shared = multi ? pc->n_mshared : 0; mc.shared_idx = pc->d_mshared_idx;
  mc.n_intc = multi ? pc->n_mint_coarse : 0; mc.int_idx = pc->d_mint_idx;
  mc.gs_c = GS(T); mc.h_c = H(T, par); mc.coef_c = pc->L[T].coef;
  mc.n_top_buf = n_top;
  static const int dbg_coarse = FEMO_TUNE_ENV("FEMO_DEBUG_COARSE") ? atoi(FEMO_TUNE_ENV("FEMO_DEBUG_COARSE")) : 0;
  mc.dbg = dbg_coarse;
  // scratch of the separable restrictions (3-D): the largest of the top restriction (one rank: level T -> T-1, + its output)
  // and the LDS-resident ones; behind the levels when it fits
  size_t lds_all = lds;
  mc.sep_off = -1;
  mc.flat = 0;
  if (pc->dim == 3 && !femo_env_flag("FEMO_BPX_TAPS27")) {
    auto need = [&](const int* nc, const int* nfn, bool with_out) -> int64_t {
      const int64_t s1 = (int64_t)(nc[0] + 1) * (nfn[1] + 1) * (nfn[2] + 1), s2 = (int64_t)(nc[0] + 1) * (nc[1] + 1) * (nfn[2] + 1);
      return s1 + s2 + (with_out ? (int64_t)(nc[0] + 1) * (nc[1] + 1) * (nc[2] + 1) : 0);
    };
    int64_t scratch = multi ? 0 : need(pc->L[T - 1].n, pc->L[T].n, true);
    for (int l = 0; l + 1 <= T - 1; ++l) scratch = std::max(scratch, need(pc->L[l].n, pc->L[l + 1].n, false));
    // the flattened chain (k_lattice_coarse_m: mc.flat): x- and y-pass outputs of ALL levels below T-1 at once
    int64_t flat_need = 0;
    const int top = T - 1;
    // (the z-pass of the flattened chain reads the prefetched coefficient of node `tid`: one trip per level, i.e. every level
    // below T-1 must have at most 1024 nodes -- true for the m0 in {2, 3} doubling lattices with TOP_MAX = 5120; checked here)
    const bool one_trip = top >= 1 && pc->L[top - 1].nodes <= 1024;
    if (top >= 1 && top <= 4 && one_trip && !femo_env_flag("FEMO_BPX_CHAIN")) {
      const int* nt = pc->L[top].n;
      for (int k = 1; k <= top; ++k) {
        const int* nc = pc->L[top - k].n;
        flat_need += (int64_t)(nc[0] + 1) * (nt[1] + 1) * (nt[2] + 1) + (int64_t)(nc[0] + 1) * (nc[1] + 1) * (nt[2] + 1);
      }
    }
    const int64_t off = (int64_t)(lds / sizeof(double));
    mc.flat = 0;
    if (flat_need > 0 && (off + std::max(scratch, flat_need)) * (int64_t)sizeof(double) <= 158 * 1024) {
      mc.flat = 1;
      scratch = std::max(scratch, flat_need);
    }
    if ((off + scratch) * (int64_t)sizeof(double) <= 158 * 1024) { mc.sep_off = off; lds_all = (size_t)(off + scratch) * sizeof(double); }
    else mc.flat = 0;
  }
  if (lds_all > 64 * 1024 && !pc->merged_lds_set) {
    FEMO_HIP_CHECK(hipFuncSetAttribute((const void*)k_lattice_coarse_m, hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024));
    pc->merged_lds_set = true;
  }
  hipLaunchKernelGGL(k_lattice_coarse_m, dim3(1 + n_carry), dim3(1024), lds_all, st, CL, pc->dim, done, mc);
  for (int l = T; l <= nl - 4; ++l) {            // N ranks, 2-D (three fused levels): the level between T-1 and the tile kernel's three
    LatticeLevel& Fi = pc->L[l];
    hipLaunchKernelGGL(k_lattice_prolong, dim3(lat_grid(Fi.nodes)), dim3(256), 0, st, Fi.n[0], Fi.n[1], Fi.n[2], pc->L[l - 1].n[0], pc->L[l - 1].n[1], pc->L[l - 1].n[2], pc->dim, pc->L[l - 1].e, Fi.coef, GS(l), 0, Fi.e, (double*)nullptr, (const double*)nullptr, done);
  }
  FineLevels FL;
  const LatticeLevel &Lcc = pc->L[nl - 4], &Lc = pc->L[nl - 3], &Lm = pc->L[nl - 2];
  for (int k = 0; k < 3; ++k) { FL.ncc[k] = Lcc.n[k]; FL.nc[k] = Lc.n[k]; FL.nm[k] = Lm.n[k]; FL.nf[k] = F.n[k]; }
  FL.e_cc = Lcc.e;
  FL.coef_c = Lc.coef; FL.g_c = GS(nl - 3); FL.g_c_other = H(nl - 3, par ^ 1);
  FL.coef_m = Lm.coef; FL.g_m = GS(nl - 2); FL.g_m_other = H(nl - 2, par ^ 1);
  FL.coef_f = F.coef; FL.g_f = GS(nl - 1); FL.e_f = F.e;
  // N > 1: the shared nodes' part of the dot comes from the carriers (replicated), the single-rank nodes' part from the
  // reduced scalars; this launch adds nothing and walks the rank's own tiles only
  FL.dot_partials = multi ? nullptr : pc->d_dot_partials;
  FL.dot_weight = nullptr;
  const int TF = pc->dim == 3 ? 8 : 16;
  int64_t tiles = 1;
  for (int k = 0; k < pc->dim; ++k) tiles *= (F.n[k] + TF) / TF;
  const int32_t* tile_list = pc->n_my_tiles < tiles ? pc->d_my_tiles : nullptr;      // all of them: walk the plain range
  if (tile_list != nullptr) tiles = pc->n_my_tiles;
  // 128 threads per tile when 256-thread workgroups would not all be resident at once (see the kernel)
  static const int bt_env = FEMO_TUNE_ENV("FEMO_PROLONG3_BT") ? atoi(FEMO_TUNE_ENV("FEMO_PROLONG3_BT")) : 0;
  const bool small_blocks = bt_env ? bt_env == 128 : tiles > 2048;      // (1.03 M rows, 1000 tiles: 7.7 us with 256 threads, 8.5 with 128)
  const int grid3 = (int)std::max<int64_t>(1, std::min<int64_t>(tiles, small_blocks ? 4096 : 2048));
  const int nb_dot = multi ? 0 : grid3;
  if (pc->dim == 3) {
    if (small_blocks) hipLaunchKernelGGL((k_lattice_prolong3_m<3, 128>), dim3(grid3), dim3(128), 0, st, FL, hF, (const double*)S, init ? 1 : 0, tile_list, tiles, done);
    else hipLaunchKernelGGL((k_lattice_prolong3_m<3, 256>), dim3(grid3), dim3(256), 0, st, FL, hF, (const double*)S, init ? 1 : 0, tile_list, tiles, done);
  } else {
    if (small_blocks) hipLaunchKernelGGL((k_lattice_prolong3_m<2, 128>), dim3(grid3), dim3(128), 0, st, FL, hF, (const double*)S, init ? 1 : 0, tile_list, tiles, done);
    else hipLaunchKernelGGL((k_lattice_prolong3_m<2, 256>), dim3(grid3), dim3(256), 0, st, FL, hF, (const double*)S, init ? 1 : 0, tile_list, tiles, done);
  }
  pc->parity ^= 1;
  PcgStop ps;
  ps.rtol2_factor = stop ? stop->rtol2_factor : 0.0;
  ps.atol_pc2 = stop ? stop->atol_pc2 : 0.0;
  ps.tolg2 = stop ? stop->tolg2 : nullptr;
  ps.flags = stop ? stop->flags : nullptr;
  ps.it = stop ? stop->it : 0;
  MergedScal ms;
  ms.S = S; ms.multi = multi ? 1 : 0; ms.init = init ? 1 : 0;
  ms.nb_lat = n_carry; ms.lat_partials = pc->d_lat_partials;
  ms.nb_rr = (multi || init) ? 0 : n_carry; ms.rr_partials = pc->d_rr_partials;
  ms.atol2 = V.atol2;
  const int mode = init ? 2 : 1;
  double* gamma_cur = S + MS_GAMMA + V.cur;
  double* gamma_nxt = init ? S + MS_GAMMA + V.cur : S + MS_GAMMA + (V.cur ^ 1);
  auto prolong = [&](unsigned grid, const HaloFirst& hf) {
    if (pc->dim == 3)
      hipLaunchKernelGGL(k_prolong_mesh<3>, dim3(grid), dim3(FEMO_BLOCK), 0, st, m->n_rows, lat, pc->d_pk, (const double*)V.r, pc->d_sinv, mask, F.e, V.p, mode, nb_dot, pc->d_dot_partials, (const double*)nullptr, 0, (const double*)nullptr, S + MS_RR, (const double*)gamma_cur, gamma_nxt, done, ps, ms, hf);
    else
      hipLaunchKernelGGL(k_prolong_mesh<2>, dim3(grid), dim3(FEMO_BLOCK), 0, st, m->n_rows, lat, pc->d_pk, (const double*)V.r, pc->d_sinv, mask, F.e, V.p, mode, nb_dot, pc->d_dot_partials, (const double*)nullptr, 0, (const double*)nullptr, S + MS_RR, (const double*)gamma_cur, gamma_nxt, done, ps, ms, hf);
  };
  if (femo_pc_merged_sends_halo(m)) {
    // the interface vertices first, straight into the send buffer; the exchange of the new direction then travels on the
    // communication stream under the bulk of the prolongation and the interior slices of the next SpMV (solver.hip:
    // femo_halo_spmv_inflight waits for it before the boundary slices)
    if (femo_halo_direct_ready(m)) {
      // device-initiated (round 6): the stores go to the neighbours' inboxes, the workgroups bump their counters; the
      // consumer is the small pull launch in front of the next product's boundary slices (solver.hip: halo_spmv_inflight)
      ++ctx->n_neighbor; ctx->neighbor_doubles += m->send_ptr[m->n_nbr];
      const unsigned long long epoch = femo_halo_direct_begin(m);
      m->hd->loop_epoch = epoch;
      // ONE launch: the first n_blocks workgroups walk the send list, the others everything else
      HaloFirst h1 = {m->n_send_verts, m->d_send_uvert, m->d_send_uptr, m->d_send_uslot, nullptr, m->d_send_flag, m->hd->d_peers, epoch, m->hd->n_blocks};
      prolong((unsigned)(m->hd->n_blocks + V.gv), h1);
      FEMO_HIP_CHECK(hipGetLastError());
      return 0;
    } else {
      HaloFirst h1 = {m->n_send_verts, m->d_send_uvert, m->d_send_uptr, m->d_send_uslot, m->d_send_buf, nullptr, nullptr, 0ull, 0};
      prolong((unsigned)std::max<int64_t>(1, std::min<int64_t>((m->n_send_verts + FEMO_BLOCK - 1) / FEMO_BLOCK, V.gv)), h1);
      FEMO_HIP_CHECK(hipEventRecord(ctx->ev_main, st));
      FEMO_HIP_CHECK(hipStreamWaitEvent(ctx->comm_stream, ctx->ev_main, 0));
      FEMO_TRY(femo_coll_neighbors(ctx, m->n_nbr, m->nbr.data(), m->send_ptr.data(), m->d_send_buf, m->recv_ptr.data(), V.p + m->n_rows, ctx->comm_stream));
      FEMO_HIP_CHECK(hipEventRecord(ctx->ev_comm, ctx->comm_stream));
    }
    HaloFirst h2 = {0, nullptr, nullptr, nullptr, nullptr, m->d_send_flag, nullptr, 0ull, 0};
    prolong((unsigned)V.gv, h2);
  } else {
    prolong((unsigned)V.gv, HaloFirst{});
  }
  FEMO_HIP_CHECK(hipGetLastError());
  return 0;
}

bool femo_pc_merged_sends_halo(const femo_mesh* m) {
  return m->ctx->nranks > 1 && m->n_nbr > 0 && m->d_slices_int != nullptr && m->n_send_verts > 0 && m->d_send_flag != nullptr &&
         m->ctx->comm_stream != nullptr;
}

// can the PCG loop hand its partial rh.rh to femo_pc_apply instead of all-reducing it itself?

bool femo_pc_carries_xupdate(const femo_mesh* m) {
  const bool off = femo_env_flag("FEMO_PCG_NO_XCARRY");        // read per solve (a test switches it), never per launch
  return !off && m->pc != nullptr && m->pc->fused_cycle_seen && m->ctx->nranks == 1;
}

bool femo_pc_can_piggyback(const femo_mesh* m) { return m->pc != nullptr && m->ctx->nranks > 1 && m->pc->shared_ready; }

int femo_pc_levels(const femo_mesh* m, int* n_levels, int64_t* finest_nodes) {
  if (!m->pc) { *n_levels = 0; *finest_nodes = 0; return 0; }
  *n_levels = m->pc->n_levels;
  *finest_nodes = m->pc->L.back().nodes;
  return 0;
}
