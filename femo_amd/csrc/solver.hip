// SELL-64 SpMV and Jacobi-preconditioned CG for gfx950.
//
// Replaces the reference's PETSc calls:
//   Mat*Vec / multTranspose  utils_dolfinx.py:256-264, 275-287   (state_model.py:176-200)
//   KSP preonly + LU(MUMPS)  utils_dolfinx.py:476-512            (fea_dolfinx.py:192-222)
// CG + Jacobi instead of LU is BASELINE.json's design: a sparse
// LU of the 10 M-DOF 3-D operator does not fit, and CSR/SELL SpMV streams at the
// HBM roofline.
//
// SpMV layout: the diagonal is a dense vector; off-diagonals live in SELL-64
// slices with pair interleave, so lane l of a wave owns row 64*slice+l and each
// wave-instruction reads 1 KiB of values (16 B/lane) and 512 B of columns
// (8 B/lane), fully coalesced.  Consecutive lanes are consecutive rows, so for
// any bandwidth-reducing vertex order the x-gathers of one instruction fall in
// a few cache lines.  Row sums are sequential per lane: deterministic.
//
// CG keeps every scalar of the recurrence on the device: kernels reduce into
// per-block partials, a one-block kernel folds them (plus an RCCL all-reduce
// when nranks > 1) and the consumers read the folded scalars.  The host only
// polls a "done" flag every `check_every` iterations, two batches deep.
#include <algorithm>
#include <climits>
#include <cmath>
#include <cstdlib>

#include "femo_internal.h"

namespace {

// ------------------------------------------------------------------- SpMV ---
// NP pair-steps of one row, fully unrolled: all value/column loads are issued
// before the first gather, all gathers before the first FMA, so one slice costs
// two memory round trips whatever its width.  Summation order is k = 0, 1, 2, ...
// 16-B / 8-B native vectors so that the nontemporal builtin emits ONE global_load_dwordx4 / x2
// per lane (on HIP's struct double2 it splits into two 8-B loads: half the access width)
typedef double femo_v2d __attribute__((ext_vector_type(2)));
typedef int femo_v2i __attribute__((ext_vector_type(2)));

__device__ __forceinline__ double2 nt_load2(const double2* p) {
  const femo_v2d v = __builtin_nontemporal_load(reinterpret_cast<const femo_v2d*>(p));
  return make_double2(v.x, v.y);
}
__device__ __forceinline__ int2 nt_load2(const int2* p) {
  const femo_v2i v = __builtin_nontemporal_load(reinterpret_cast<const femo_v2i*>(p));
  return make_int2(v.x, v.y);
}
// NT = false (round 5): ordinary loads for operators whose stored values fit the 256 MB Infinity Cache -- they are read again
// 28 times per solve, and the streaming hint kept them from staying there (1.03 M rows: 27.6 -> 21.0 us per product; at
// 10 M rows, 1.4 GB of values, the hint is worth 2 %: launch_spmv picks by size)
template <bool NT> __device__ __forceinline__ double2 mat_load2(const double2* p) {
  if constexpr (NT) return nt_load2(p);
  else { const femo_v2d v = *reinterpret_cast<const femo_v2d*>(p); return make_double2(v.x, v.y); }
}
template <bool NT> __device__ __forceinline__ int mat_load(const int* p) {
  if constexpr (NT) return __builtin_nontemporal_load(p);
  else return *p;
}

template <int NP, bool NT>
__device__ __forceinline__ double row_pairs(const double2* __restrict__ v2, const int2* __restrict__ c2,
                                            const double* __restrict__ x, double acc) {
  double2 a[NP];
  int2 j[NP];
#pragma unroll
  for (int m = 0; m < NP; ++m) {
    if (NT) {
      a[m] = nt_load2(&v2[m * 64]);
      j[m] = nt_load2(&c2[m * 64]);
    } else {
      a[m] = v2[m * 64];
      j[m] = c2[m * 64];
    }
  }
  double xv[2 * NP];
#pragma unroll
  for (int m = 0; m < NP; ++m) {
    xv[2 * m] = x[j[m].x];
    xv[2 * m + 1] = x[j[m].y];
  }
#pragma unroll
  for (int m = 0; m < NP; ++m) {
    acc += a[m].x * xv[2 * m];
    acc += a[m].y * xv[2 * m + 1];
  }
  return acc;
}

template <bool NT>
__device__ __forceinline__ double row_sum(int npair, const double2* __restrict__ v2, const int2* __restrict__ c2,
                                          const double* __restrict__ x, double acc) {
  while (npair > 8) {
    acc = row_pairs<8, NT>(v2, c2, x, acc);
    v2 += 8 * 64; c2 += 8 * 64; npair -= 8;
  }
  switch (npair) {  // wave-uniform
    case 8: return row_pairs<8, NT>(v2, c2, x, acc);
    case 7: return row_pairs<7, NT>(v2, c2, x, acc);
    case 6: return row_pairs<6, NT>(v2, c2, x, acc);
    case 5: return row_pairs<5, NT>(v2, c2, x, acc);
    case 4: return row_pairs<4, NT>(v2, c2, x, acc);
    case 3: return row_pairs<3, NT>(v2, c2, x, acc);
    case 2: return row_pairs<2, NT>(v2, c2, x, acc);
    case 1: return row_pairs<1, NT>(v2, c2, x, acc);
    default: return acc;
  }
}

// Short slice: the columns as 16-bit deltas from the row, two per 4-byte word (one global_load_dword per pair and lane
// instead of a dwordx2)
template <int NP, bool NT>
__device__ __forceinline__ double row_pairs_short(const double2* __restrict__ v2, const int* __restrict__ c16,
                                                  const double* __restrict__ xrow, double acc) {
  double2 a[NP];
  int j[NP];
#pragma unroll
  for (int m = 0; m < NP; ++m) {
    a[m] = mat_load2<NT>(&v2[m * 64]);
    j[m] = mat_load<NT>(&c16[m * 64]);
  }
  double xv[2 * NP];
#pragma unroll
  for (int m = 0; m < NP; ++m) {
    xv[2 * m] = xrow[(int)(short)(j[m] & 0xFFFF)];
    xv[2 * m + 1] = xrow[j[m] >> 16];
  }
#pragma unroll
  for (int m = 0; m < NP; ++m) {
    acc += a[m].x * xv[2 * m];
    acc += a[m].y * xv[2 * m + 1];
  }
  return acc;
}

template <bool NT>
__device__ __forceinline__ double row_sum_short(int npair, const double2* __restrict__ v2, const int* __restrict__ c16,
                                                const double* __restrict__ xrow, double acc) {
  while (npair > 8) {
    acc = row_pairs_short<8, NT>(v2, c16, xrow, acc);
    v2 += 8 * 64; c16 += 8 * 64; npair -= 8;
  }
  switch (npair) {  // wave-uniform
    case 8: return row_pairs_short<8, NT>(v2, c16, xrow, acc);
    case 7: return row_pairs_short<7, NT>(v2, c16, xrow, acc);
    case 6: return row_pairs_short<6, NT>(v2, c16, xrow, acc);
    case 5: return row_pairs_short<5, NT>(v2, c16, xrow, acc);
    case 4: return row_pairs_short<4, NT>(v2, c16, xrow, acc);
    case 3: return row_pairs_short<3, NT>(v2, c16, xrow, acc);
    case 2: return row_pairs_short<2, NT>(v2, c16, xrow, acc);
    case 1: return row_pairs_short<1, NT>(v2, c16, xrow, acc);
    default: return acc;
  }
}

// Regular slice: column k of lane l is row + delta[k]; x is read as 64 consecutive
// doubles per k (one coalesced 512-B load), no column indices are fetched.
template <int NP, bool NT>
__device__ __forceinline__ double row_pairs_regular(const double2* __restrict__ v2, const int32_t* __restrict__ delta,
                                                    const double* __restrict__ xrow, double acc) {
  double2 a[NP];
#pragma unroll
  for (int m = 0; m < NP; ++m) {
    a[m] = mat_load2<NT>(&v2[m * 64]);
  }
  double xv[2 * NP];
#pragma unroll
  for (int m = 0; m < NP; ++m) {
    xv[2 * m] = xrow[delta[2 * m]];
    xv[2 * m + 1] = xrow[delta[2 * m + 1]];
  }
#pragma unroll
  for (int m = 0; m < NP; ++m) {
    acc += a[m].x * xv[2 * m];
    acc += a[m].y * xv[2 * m + 1];
  }
  return acc;
}

template <bool NT>
__device__ __forceinline__ double row_sum_regular(int npair, const double2* __restrict__ v2, const int32_t* __restrict__ delta,
                                                  const double* __restrict__ xrow, double acc) {
  while (npair > 8) {
    acc = row_pairs_regular<8, NT>(v2, delta, xrow, acc);
    v2 += 8 * 64; delta += 16; npair -= 8;
  }
  switch (npair) {  // wave-uniform
    case 8: return row_pairs_regular<8, NT>(v2, delta, xrow, acc);
    case 7: return row_pairs_regular<7, NT>(v2, delta, xrow, acc);
    case 6: return row_pairs_regular<6, NT>(v2, delta, xrow, acc);
    case 5: return row_pairs_regular<5, NT>(v2, delta, xrow, acc);
    case 4: return row_pairs_regular<4, NT>(v2, delta, xrow, acc);
    case 3: return row_pairs_regular<3, NT>(v2, delta, xrow, acc);
    case 2: return row_pairs_regular<2, NT>(v2, delta, xrow, acc);
    case 1: return row_pairs_regular<1, NT>(v2, delta, xrow, acc);
    default: return acc;
  }
}

// ---- boundary slices of a partitioned mesh inside the ONE launch of the merged loop (round 6) -------------------------
// With the device-initiated ghost refresh the product needs no second launch and no copy of the inbox: the slices with
// ghost columns sit at the END of every XCD's range of the slice list (flagged by the sign bit), the wave that reaches one
// first waits for the neighbours' counters (they have had the whole interior to arrive) and then gathers ghost columns
// straight from its own inbox generation.  femo_internal.h: FemoHaloDirect.
struct SpmvGhost {
  const unsigned long long* cnt; const int32_t* blocks; int n_nbr; unsigned long long epoch; int32_t* err;
  const double* ghost;          // inbox generation of this exchange, ghost k at ghost[k]
  int64_t n_own;
};
__device__ __forceinline__ void spmv_halo_wait_wave(const SpmvGhost& g, int lane) {
  if (lane < g.n_nbr) {
    const unsigned long long want = g.epoch * (unsigned long long)g.blocks[lane];
    const unsigned long long* c = g.cnt + (size_t)lane * FEMO_HALO_CNT_STRIDE;
    const long long t0 = wall_clock64();
    while (__hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < want) {
      __builtin_amdgcn_s_sleep(2);
      if (wall_clock64() - t0 > 400000000ll) { atomicExch(g.err, 1); break; }
    }
  }
  // the wave reconverges here; nothing below may be issued before the counters were seen (the inbox is uncached memory:
  // no fence beyond ordering is needed, see femo_halo_signal)
  __builtin_amdgcn_wave_barrier();
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
}
template <int NP, bool NT>
__device__ __forceinline__ double row_pairs_gh(const double2* __restrict__ v2, const int2* __restrict__ c2,
                                               const double* __restrict__ x, const double* __restrict__ gbase, int n_own, double acc) {
  double2 a[NP];
  int2 j[NP];
#pragma unroll
  for (int m = 0; m < NP; ++m) { a[m] = mat_load2<NT>(&v2[m * 64]); j[m] = c2[m * 64]; }
  double xv[2 * NP];
#pragma unroll
  for (int m = 0; m < NP; ++m) {
    // (one `sc0 sc1` load either way: the ghost entries must come from the memory side, femo_halo_store; the owned entries of
    // these few slices can afford to)
    xv[2 * m] = femo_halo_load((j[m].x < n_own ? x : gbase) + j[m].x);
    xv[2 * m + 1] = femo_halo_load((j[m].y < n_own ? x : gbase) + j[m].y);
  }
#pragma unroll
  for (int m = 0; m < NP; ++m) {
    acc += a[m].x * xv[2 * m];
    acc += a[m].y * xv[2 * m + 1];
  }
  return acc;
}
template <bool NT>
__device__ __forceinline__ double row_sum_gh(int npair, const double2* __restrict__ v2, const int2* __restrict__ c2,
                                             const double* __restrict__ x, const double* __restrict__ gbase, int n_own, double acc) {
  while (npair > 4) {
    acc = row_pairs_gh<4, NT>(v2, c2, x, gbase, n_own, acc);
    v2 += 4 * 64; c2 += 4 * 64; npair -= 4;
  }
  switch (npair) {  // wave-uniform
    case 4: return row_pairs_gh<4, NT>(v2, c2, x, gbase, n_own, acc);
    case 3: return row_pairs_gh<3, NT>(v2, c2, x, gbase, n_own, acc);
    case 2: return row_pairs_gh<2, NT>(v2, c2, x, gbase, n_own, acc);
    case 1: return row_pairs_gh<1, NT>(v2, c2, x, gbase, n_own, acc);
    default: return acc;
  }
}

// DOT: 0 none; 1 partial d.Ax into partials[block] (d = dvec or x); 2 additionally partial x.x,
// 3 additionally partial Ax.Ax, into the next slot; 4 (merged BPX-PCG): x.Ax, Ax.Ax and dvec.Ax into three
// consecutive slots (p.q, q.q, r.q: everything the single all-reduce of an iteration carries besides the lattice)
template <int DOT, bool UNIT, bool NT = true, bool GH = false>
__global__ __launch_bounds__(FEMO_BLOCK) void k_spmv_sell(
    int64_t n_rows, int64_t n_slices, const int64_t* __restrict__ mptr,
    const int32_t* __restrict__ cols, const int16_t* __restrict__ cols16, const int32_t* __restrict__ sdelta, int sdelta_stride,
    const double* __restrict__ vals, const double* __restrict__ diag, const double* __restrict__ x,
    double* __restrict__ y, double* __restrict__ partials, const int32_t* __restrict__ done,
    const int32_t* __restrict__ slice_list, int64_t n_list, const double* __restrict__ dvec, SpmvGhost gh = SpmvGhost{}) {
  if (done != nullptr && *done) return;
  bool waited = false;
  __shared__ double lds[FEMO_BLOCK / 64];
  const int lane = threadIdx.x & 63;
  // wave-uniform by construction: tell the compiler, so that slice metadata (offsets, the
  // per-slice delta table) comes through scalar loads instead of 64 identical vector loads
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // XCD-aware: blockIdx % 8 labels the XCD group; each group walks its own
  // contiguous eighth of the slices, its waves interleaved slice by slice, so the
  // x window the group gathers from stays in that XCD's L2.
  const int xcd = blockIdx.x & 7;
  const int64_t blk_in_xcd = blockIdx.x >> 3;
  const int64_t waves_per_xcd = (int64_t)(gridDim.x >> 3) * (FEMO_BLOCK / 64);
  // slice_list != null: walk that subset (interior or boundary slices of a partitioned mesh)
  const int64_t n_walk = slice_list ? n_list : n_slices;
  const int64_t s_lo = n_walk * xcd / 8, s_hi = n_walk * (xcd + 1) / 8;
  double dot = 0.0, dot2 = 0.0, dot3 = 0.0;
  for (int64_t si = s_lo + blk_in_xcd * (FEMO_BLOCK / 64) + wave; si < s_hi; si += waves_per_xcd) {
    const int32_t entry = slice_list ? slice_list[si] : (int32_t)si;
    const bool bnd = GH && entry < 0;                      // sign bit: the slice has ghost columns (wave-uniform)
    const int64_t slice = GH ? (int64_t)(entry & 0x7FFFFFFF) : (slice_list ? (int64_t)entry : si);
    const int64_t base = mptr[slice];
    const int npair = (int)((mptr[slice + 1] - base) >> 7);
    const int64_t row = (slice << 6) + lane;
    if (GH && bnd && !waited) { spmv_halo_wait_wave(gh, lane); waited = true; }
    const double xr = x[row < n_rows ? row : 0];
    double acc = UNIT ? xr : diag[row] * xr;   // UNIT: symmetrically scaled operator, diagonal == 1
    const double2* __restrict__ v2 = reinterpret_cast<const double2*>(vals + base) + lane;
    const int32_t* __restrict__ dl = sdelta + slice * sdelta_stride;
    if (GH && bnd) {           // 32-bit columns whatever the slice's class; ghost columns come from the inbox
      const int2* __restrict__ c2 = reinterpret_cast<const int2*>(cols + base) + lane;
      acc = row_sum_gh<NT>(npair, v2, c2, x, gh.ghost - gh.n_own, (int)gh.n_own, acc);
    } else if (dl[0] != INT32_MIN) {  // wave-uniform (scalar load)
      acc = row_sum_regular<NT>(npair, v2, dl, x + row, acc);
    } else if (dl[1] == 1) {   // 16-bit column deltas (the clamped row of a lane beyond n_rows still addresses valid entries)
      const int* __restrict__ c16 = reinterpret_cast<const int*>(cols16) + (base >> 1) + lane;
      acc = row_sum_short<NT>(npair, v2, c16, x + row, acc);
    } else {
      const int2* __restrict__ c2 = reinterpret_cast<const int2*>(cols + base) + lane;
      acc = row_sum<NT>(npair, v2, c2, x, acc);
    }
    if (row < n_rows) {
      y[row] = acc;
      if (DOT == 4) {
        dot += acc * xr;
        dot2 += acc * acc;
        dot3 += acc * dvec[row];
      } else {
        if (DOT) dot += acc * (dvec ? dvec[row] : xr);
        if (DOT == 2) dot2 += xr * xr;
        if (DOT == 3) dot2 += acc * acc;
      }
    }
  }
  if (DOT) {
    const double s = femo_block_sum<FEMO_BLOCK>(dot, lds);
    if (threadIdx.x == 0) partials[blockIdx.x] = s;
  }
  if (DOT >= 2) {
    const double s = femo_block_sum<FEMO_BLOCK>(dot2, lds);
    if (threadIdx.x == 0) partials[FEMO_MAX_PARTIALS + blockIdx.x] = s;
  }
  if (DOT == 4) {
    const double s = femo_block_sum<FEMO_BLOCK>(dot3, lds);
    if (threadIdx.x == 0) partials[2 * FEMO_MAX_PARTIALS + blockIdx.x] = s;
  }
}

// y += A^T x by scatter (partitioned, structurally non-symmetric-valued operators): the transposed entry of a
// ghost column lives in a row of another rank, so the explicit transposed values cannot be formed locally.
// Row i adds a_ij x_i to y_j for all its entries, ghost columns included; the ghost tail of y then travels back
// to the owners, who add it (femo_halo_reverse_add: SURVEY.md section 8(e) "reverse halo scatter-add").  Scattered
// fp64 atomics are slow; this is the correctness path of the unsymmetric-Nitsche adjoint
// on N > 1, not a benchmark path.  y[0:n_vert] must be zero on entry.
template <bool UNIT>
__global__ __launch_bounds__(FEMO_BLOCK) void k_spmv_sell_T_scatter(int64_t n_rows, int64_t n_slices, const int64_t* __restrict__ mptr,
                                                                    const int32_t* __restrict__ cols, const int32_t* __restrict__ rowlen,
                                                                    const double* __restrict__ vals, const double* __restrict__ diag,
                                                                    const double* __restrict__ x, double* __restrict__ y,
                                                                    const int32_t* __restrict__ done) {
  if (done != nullptr && *done) return;
  const int64_t row = (int64_t)blockIdx.x * FEMO_BLOCK + threadIdx.x;
  if (row >= n_rows) return;
  const int64_t base = mptr[row >> 6];
  const int lane = (int)(row & 63), len = rowlen[row];
  const double xi = x[row];
  unsafeAtomicAdd(&y[row], UNIT ? xi : diag[row] * xi);
  for (int k = 0; k < len; ++k) {
    const int64_t e = femo_sell_index(base, k, lane);
    unsafeAtomicAdd(&y[cols[e]], vals[e] * xi);
  }
}

// y[send_idx[i]] += buf[i]
__global__ void k_unpack_add(int64_t n, const int32_t* __restrict__ idx, const double* __restrict__ buf, double* __restrict__ y) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    unsafeAtomicAdd(&y[idx[i]], buf[i]);       // a vertex can be sent to several neighbours
}

// ------------------------------------------------------------ transposition --
__global__ void k_build_tperm(int64_t n_rows, int64_t n_slices, const int64_t* __restrict__ mptr,
                              const int32_t* __restrict__ cols, const int32_t* __restrict__ rowlen, const uint32_t* __restrict__ rowreal,
                              int32_t* __restrict__ tperm, int32_t* __restrict__ err) {
  const int64_t row = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= n_slices * 64) return;
  const int64_t slice = row >> 6;
  const int lane = (int)(row & 63);
  const int64_t base = mptr[slice];
  const int wm = (int)((mptr[slice + 1] - base) >> 6);
  const int len = rowlen[row];
  for (int k = 0; k < wm; ++k) {
    const int64_t e = femo_sell_index(base, k, lane);
    int32_t t = (int32_t)e;  // padding and structural zeros (completed regular slices) map to themselves (value 0)
    if (k < len && (k >= 32 || ((rowreal[row] >> k) & 1u))) {
      const int64_t j = cols[e];
      if (j >= n_rows) {
        atomicExch(err, 1);  // transposed entry lives on another rank
      } else {
        const int64_t sj = j >> 6;
        const int lj = (int)(j & 63);
        const int64_t bj = mptr[sj];
        const int lenj = rowlen[j];
        int found = -1;
        for (int kk = 0; kk < lenj; ++kk) {
          const int64_t ej = femo_sell_index(bj, kk, lj);
          if (cols[ej] == row) { found = (int)ej; break; }
        }
        if (found < 0) atomicExch(err, 2);  // structurally unsymmetric pattern
        else t = found;
      }
    }
    tperm[e] = t;
  }
}

__global__ void k_gather(int64_t n, const int32_t* __restrict__ perm, const double* __restrict__ in,
                         double* __restrict__ out) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    out[i] = in[perm[i]];
}

// ------------------------------------------------------------- CG kernels ---
// Jacobi-preconditioned CG is run as plain CG on the symmetrically scaled system
//     (S A S) xh = S b,   x = S xh,   S = diag(A)^-1/2
// which generates the same iterates in exact arithmetic, has a unit diagonal (no
// diagonal or D^-1 stream per iteration) and makes gamma = rh.rh = r^T D^-1 r the
// only norm needed.  Convergence: sqrt(gamma) <= max(rtol sqrt(bh.bh), atol), the
// "natural" norm of preconditioned CG (PETSc KSP_NORM_NATURAL [ext]).
//
// Per iteration (single GPU, 3 launches, 10 N-vectors of traffic besides the matrix):
//   k_spmv_sell<true,true>   qh = Ah ph, partial ph.qh                  (x gather N, y write N)
//   k_cg_update_r            alpha = gamma/delta; rh -= alpha qh; partial rh.rh     (3 N)
//   k_cg_update_xp           xh += alpha ph; ph = rh + beta ph; convergence flag    (5 N)
// Scalars never visit the host: every block folds the producers' per-block partials
// itself in a fixed order, so all blocks see bitwise identical values.  With nranks > 1
// the single-reduction recurrence further down is used instead (one fold + one RCCL
// all-reduce per iteration).  Not captured in a hipGraph on purpose: every launch runs
// >= 40 us at the sizes of interest against ~3.5 us of host cost per launch, and the
// device-side boundary costs the same eagerly or replayed (MI355X_MICROARCH.md price list).
// partial slots: 0 = delta, 1/2 = gamma of even/odd iterations, 3 = scratch.
// scal: [0],[1] = gamma even/odd, [2] = delta, [3] = tol^2.   flags: [0] done, [1] iterations, [2] breakdown.
constexpr int S_GAMMA = 0, S_DELTA = 2, S_TOL2 = 3;
constexpr int P_DELTA = 0, P_GAMMA = 1;

template <int NT>
__device__ __forceinline__ double femo_block_sum_bcast(double v, double* lds /* NT/64 */) {
  v = femo_wave_sum(v);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) lds[w] = v;
  __syncthreads();
  double s = 0.0;
#pragma unroll
  for (int i = 0; i < NT / 64; ++i) s += lds[i];
  return s;
}

__device__ __forceinline__ double cg_scalar(const double* __restrict__ partials, int nb, double* lds) {
  double a = 0.0;
  for (int i = threadIdx.x; i < nb; i += FEMO_BLOCK) a += partials[i];
  return femo_block_sum_bcast<FEMO_BLOCK>(a, lds);
}

__global__ void k_invsqrt_diag(int64_t n, const double* __restrict__ diag, double* __restrict__ s) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    s[i] = 1.0 / sqrt(diag[i]);
}

// valsS[row][k] = s[row] * vals[row][k] * s[col]   (one wave per slice)
__global__ __launch_bounds__(FEMO_BLOCK) void k_scale_sell(int64_t n_slices, const int64_t* __restrict__ mptr,
                                                           const int32_t* __restrict__ cols, const int32_t* __restrict__ sdelta,
                                                           int sdelta_stride, const double* __restrict__ vals,
                                                           const double* __restrict__ s, int64_t n_vert, double* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int64_t nw = (int64_t)gridDim.x * (FEMO_BLOCK / 64);
  for (int64_t slice = (int64_t)blockIdx.x * (FEMO_BLOCK / 64) + (threadIdx.x >> 6); slice < n_slices; slice += nw) {
    const int64_t base = mptr[slice];
    const int wm = (int)((mptr[slice + 1] - base) >> 6);
    const int64_t row = (slice << 6) + lane;
    const double si = s[row < n_vert ? row : 0];
    const int32_t* dl = sdelta + slice * sdelta_stride;
    const bool regular = dl[0] != INT32_MIN;
    for (int k = 0; k < wm; k += 2) {
      const int64_t idx = base + (int64_t)(k >> 1) * 128 + lane * 2;
      const double2 v = *reinterpret_cast<const double2*>(vals + idx);
      int64_t c0, c1;
      if (regular) { c0 = row + dl[k]; c1 = row + dl[k + 1]; }
      else { const int2 cc = *reinterpret_cast<const int2*>(cols + idx); c0 = cc.x; c1 = cc.y; }
      double2 o;
      o.x = si * v.x * s[c0];
      o.y = si * v.y * s[c1];
      *reinterpret_cast<double2*>(out + idx) = o;
    }
  }
}

// rh = S (b - q) (q = A x0 or null); ph = rh; xh = 0; partials: slot 1 = rh.rh, slot 2 = (S b).(S b).
// Identity rows (idrow: the Dirichlet rows of the last assembly; their columns are eliminated too) are solved
// here, xh = rh exactly, and take no part in the iteration or in either norm: with u = 1 as initial state the
// lifted values on those rows are O(1) while the interior right-hand side is O(h^3), and a tolerance relative to
// the whole vector would be no tolerance for the interior unknowns.
__global__ __launch_bounds__(FEMO_BLOCK) void k_cg_init(int64_t n, const double* __restrict__ b, const double* __restrict__ q,
                                                        const double* __restrict__ s, double* __restrict__ r,
                                                        double* __restrict__ p, double* __restrict__ xh,
                                                        double* __restrict__ partials, const uint8_t* __restrict__ idrow) {
  __shared__ double lds[FEMO_BLOCK / 64];
  double s0 = 0.0, s1 = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * FEMO_BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * FEMO_BLOCK) {
    const double si = s[i], bi = si * b[i];
    const double ri = q ? si * (b[i] - q[i]) : bi;
    if (idrow != nullptr && idrow[i]) {
      r[i] = 0.0; p[i] = 0.0; xh[i] = ri;
      continue;
    }
    r[i] = ri; p[i] = ri; xh[i] = 0.0;
    s0 += ri * ri; s1 += bi * bi;
  }
  double t = femo_block_sum<FEMO_BLOCK>(s0, lds);
  if (threadIdx.x == 0) partials[1 * FEMO_MAX_PARTIALS + blockIdx.x] = t;
  t = femo_block_sum<FEMO_BLOCK>(s1, lds);
  if (threadIdx.x == 0) partials[2 * FEMO_MAX_PARTIALS + blockIdx.x] = t;
}

// rh -= alpha qh ; partial rh.rh into the gamma slot of the next parity
__global__ __launch_bounds__(FEMO_BLOCK) void k_cg_update_r(int64_t n, int cur, int nb_d, int nb_g,
                                                            double* __restrict__ partials, const double* __restrict__ scal,
                                                            const double* __restrict__ q, double* __restrict__ r,
                                                            const int32_t* __restrict__ done) {
  if (*done) return;
  __shared__ double lds[FEMO_BLOCK / 64];
  const double gamma = cg_scalar(partials + (P_GAMMA + cur) * FEMO_MAX_PARTIALS, nb_g, lds);
  const double delta = cg_scalar(partials + P_DELTA * FEMO_MAX_PARTIALS, nb_d, lds);
  const double alpha = delta != 0.0 ? gamma / delta : 0.0;
  double s0 = 0.0;
  const int64_t n2 = n >> 1;
  const double2* q2 = reinterpret_cast<const double2*>(q);
  double2* r2 = reinterpret_cast<double2*>(r);
  for (int64_t i = (int64_t)blockIdx.x * FEMO_BLOCK + threadIdx.x; i < n2; i += (int64_t)gridDim.x * FEMO_BLOCK) {
    const double2 qi = q2[i];
    double2 ri = r2[i];
    ri.x -= alpha * qi.x; ri.y -= alpha * qi.y;
    r2[i] = ri;
    s0 += ri.x * ri.x + ri.y * ri.y;
  }
  if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
    const double ri = r[n - 1] - alpha * q[n - 1];
    r[n - 1] = ri;
    s0 += ri * ri;
  }
  const double t = femo_block_sum<FEMO_BLOCK>(s0, lds);
  if (threadIdx.x == 0) partials[(P_GAMMA + (cur ^ 1)) * FEMO_MAX_PARTIALS + blockIdx.x] = t;
}

// xh += alpha ph ; then (unless converged) ph = rh + beta ph
__global__ __launch_bounds__(FEMO_BLOCK) void k_cg_update_xp(int64_t n, int cur, int it, int nb_d, int nb_g,
                                                             const double* __restrict__ partials, const double* __restrict__ scal,
                                                             const double* __restrict__ r, double* __restrict__ p,
                                                             double* __restrict__ xh, int32_t* __restrict__ flags) {
  // flags[0] = (iteration index + 1) at which convergence was detected, 0 while running.
  // A block of THIS launch may already have set it; only an earlier iteration's stamp
  // means "done" here, otherwise late blocks would skip their part of the x update.
  const int32_t stamp = flags[0];
  if (stamp != 0 && stamp != it + 1) return;
  __shared__ double lds[FEMO_BLOCK / 64];
  const int nxt = cur ^ 1;
  const double gamma = cg_scalar(partials + (P_GAMMA + cur) * FEMO_MAX_PARTIALS, nb_g, lds);
  const double gamma1 = cg_scalar(partials + (P_GAMMA + nxt) * FEMO_MAX_PARTIALS, nb_g, lds);
  const double delta = cg_scalar(partials + P_DELTA * FEMO_MAX_PARTIALS, nb_d, lds);
  const double alpha = delta != 0.0 ? gamma / delta : 0.0;
  const bool bad = !(gamma1 == gamma1) || !(alpha == alpha);  // NaN: not SPD or diverged
  const bool converged = gamma1 <= scal[S_TOL2] || bad;
  const double beta = gamma != 0.0 ? gamma1 / gamma : 0.0;
  const int64_t n2 = n >> 1;
  const double2* r2 = reinterpret_cast<const double2*>(r);
  double2* p2 = reinterpret_cast<double2*>(p);
  double2* x2 = reinterpret_cast<double2*>(xh);
  if (converged) {
    if (!bad) {
      for (int64_t i = (int64_t)blockIdx.x * FEMO_BLOCK + threadIdx.x; i < n2; i += (int64_t)gridDim.x * FEMO_BLOCK) {
        const double2 pi = p2[i];
        double2 xi = x2[i];
        xi.x += alpha * pi.x; xi.y += alpha * pi.y;
        x2[i] = xi;
      }
      if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) xh[n - 1] += alpha * p[n - 1];
    }
  } else {
    for (int64_t i = (int64_t)blockIdx.x * FEMO_BLOCK + threadIdx.x; i < n2; i += (int64_t)gridDim.x * FEMO_BLOCK) {
      const double2 ri = r2[i];
      double2 pi = p2[i], xi = x2[i];
      xi.x += alpha * pi.x; xi.y += alpha * pi.y;
      pi.x = ri.x + beta * pi.x; pi.y = ri.y + beta * pi.y;
      x2[i] = xi; p2[i] = pi;
    }
    if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
      const int64_t i = n - 1;
      const double pi = p[i];
      xh[i] += alpha * pi;
      p[i] = r[i] + beta * pi;
    }
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    flags[1] = it + 1;
    if (converged) {
      flags[2] = bad ? 1 : 0;
      __threadfence();
      flags[0] = it + 1;
    }
  }
}

// ---- BPX-preconditioned CG (bpx.hip) ------------------------------------------------
// scal: [0],[1] = gamma = rh.zh of even/odd iterations, [2] = delta = ph.Ah ph, [3] = tol^2,
// [4] = rho = rh.rh (the natural-norm residual the stopping test uses, same as Jacobi-CG).
constexpr int S_RHO = 4;
constexpr int S_ALPHA = 8;  // alpha of the current iteration (k_pcg_xr with carry_x: the x update runs inside the preconditioner)
constexpr int S_TOLG = 5;   // rtol^2 * gamma_0: the stopping threshold on gamma = rh.zh (set on the device by the first apply)

// out[0] = sum of `nb` partials (+ `nb2` partials of the next slot pair, overlapped SpMV)
__global__ __launch_bounds__(1024) void k_pcg_fold(int nb, const double* __restrict__ partials, int nb2,
                                                   const double* __restrict__ partials2, double* __restrict__ out,
                                                   const int32_t* __restrict__ done) {
  if (*done) return;
  __shared__ double lds[1024 / 64];
  double acc = 0.0;
  for (int i = threadIdx.x; i < nb; i += 1024) acc += partials[i];
  for (int i = threadIdx.x; i < nb2; i += 1024) acc += partials2[i];
  const double t = femo_block_sum<1024>(acc, lds);
  if (threadIdx.x == 0) out[0] = t;
}

// xh += alpha ph ; rh -= alpha qh ; partial rh.rh.  nb_d > 0: delta is folded here from the
// SpMV's per-block partials (single GPU: no fold kernel, no all-reduce), else read from scal.
__global__ __launch_bounds__(FEMO_BLOCK) void k_pcg_xr(int64_t n, int cur, int nb_d, const double* __restrict__ partials_d,
                                                       double* __restrict__ scal,
                                                       const double* __restrict__ q, const double* __restrict__ p,
                                                       const double* r_in, double* r, double* __restrict__ xh,
                                                       double* __restrict__ partials, const int32_t* __restrict__ done, int carry_x = 0) {
  // r = r_in - alpha q: in place (r_in == r) or into the other residual buffer when the restriction of the
  // caller keeps the old residual
  if (*done) return;
  __shared__ double lds[FEMO_BLOCK / 64];
  const double gamma = scal[S_GAMMA + cur];
  const double delta = nb_d > 0 ? cg_scalar(partials_d, nb_d, lds) : scal[S_DELTA];
  if (nb_d > 0 && blockIdx.x == 0 && threadIdx.x == 0) scal[S_DELTA] = delta;   // for the breakdown test
  const double alpha = delta != 0.0 ? gamma / delta : 0.0;
  double s0 = 0.0;
  const int64_t n2 = n >> 1;
  const double2* q2 = reinterpret_cast<const double2*>(q);
  const double2* p2 = reinterpret_cast<const double2*>(p);
  double2* r2 = reinterpret_cast<double2*>(r);
  const double2* r2_in = reinterpret_cast<const double2*>(r_in);
  double2* x2 = reinterpret_cast<double2*>(xh);
  if (carry_x) {
    // the residual only (3 of the 6 vector streams); x += alpha p is carried by the preconditioner's coarse-lattice launch
    if (blockIdx.x == 0 && threadIdx.x == 0) scal[S_ALPHA] = alpha;
    for (int64_t i = (int64_t)blockIdx.x * FEMO_BLOCK + threadIdx.x; i < n2; i += (int64_t)gridDim.x * FEMO_BLOCK) {
      const double2 qi = q2[i];
      double2 ri = r2_in[i];
      ri.x -= alpha * qi.x; ri.y -= alpha * qi.y;
      r2[i] = ri;
      s0 += ri.x * ri.x + ri.y * ri.y;
    }
    if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
      const double ri = r_in[n - 1] - alpha * q[n - 1];
      r[n - 1] = ri;
      s0 += ri * ri;
    }
    const double t = femo_block_sum<FEMO_BLOCK>(s0, lds);
    if (threadIdx.x == 0) partials[blockIdx.x] = t;
    return;
  }
  for (int64_t i = (int64_t)blockIdx.x * FEMO_BLOCK + threadIdx.x; i < n2; i += (int64_t)gridDim.x * FEMO_BLOCK) {
    const double2 qi = q2[i], pi = p2[i];
    double2 ri = r2_in[i], xi = x2[i];
    xi.x += alpha * pi.x; xi.y += alpha * pi.y;
    ri.x -= alpha * qi.x; ri.y -= alpha * qi.y;
    x2[i] = xi; r2[i] = ri;
    s0 += ri.x * ri.x + ri.y * ri.y;
  }
  if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
    const int64_t i = n - 1;
    xh[i] += alpha * p[i];
    const double ri = r_in[i] - alpha * q[i];
    r[i] = ri;
    s0 += ri * ri;
  }
  const double t = femo_block_sum<FEMO_BLOCK>(s0, lds);
  if (threadIdx.x == 0) partials[blockIdx.x] = t;
}

// stopping test on the reduced rho; runs alone on the stream, so later kernels see a settled flag.
// nb > 0: folds the rho partials first (single GPU); else rho was folded and all-reduced before.
__global__ __launch_bounds__(1024) void k_pcg_check(int it, int nb, const double* __restrict__ partials,
                                                    double* __restrict__ scal, int32_t* __restrict__ flags) {
  if (flags[0]) return;
  __shared__ double lds[1024 / 64];
  double rho = 0.0;
  if (nb > 0) {
    double acc = 0.0;
    for (int i = threadIdx.x; i < nb; i += 1024) acc += partials[i];
    rho = femo_block_sum<1024>(acc, lds);
  }
  if (threadIdx.x != 0) return;
  if (nb > 0) scal[S_RHO] = rho; else rho = scal[S_RHO];
  const double delta = scal[S_DELTA];
  const bool bad = !(rho == rho) || !(delta == delta);
  flags[1] = it + 1;
  if (rho <= scal[S_TOL2] || bad) {
    flags[2] = bad ? 1 : 0;
    flags[0] = it + 1;
  }
}

// ---- single-reduction CG (Chronopoulos & Gear) for nranks > 1 ---------------------
// One all-reduce of (gamma = r.r, delta = r.Ar) and one halo exchange per iteration:
//   beta = gamma/gamma_old; alpha = gamma / (delta - beta*gamma/alpha_old)
//   p = r + beta p; s = w + beta s; x += alpha p; r -= alpha s;   then w = A r with both dots
// fused into the SpMV.  scal: (delta, gamma) of parity c at [2c], [2c+1] (contiguous for one
// all-reduce); alpha of parity c at [4+c]; tol^2 at [6].
constexpr int M_A = 4, M_TOL2 = 6;

__global__ __launch_bounds__(FEMO_BLOCK) void k_cgm_update(int64_t n, int it, double* __restrict__ scal,
                                                           const double* __restrict__ w, double* __restrict__ r,
                                                           double* __restrict__ p, double* __restrict__ sv,
                                                           double* __restrict__ xh, int32_t* __restrict__ flags) {
  const int32_t stamp = flags[0];
  if (stamp != 0 && stamp != it + 1) return;
  const int cur = it & 1, prv = cur ^ 1;
  const double delta = scal[2 * cur], gamma = scal[2 * cur + 1];
  const bool bad = !(gamma == gamma) || !(delta == delta);
  if (gamma <= scal[M_TOL2] || bad) {              // converged before this update: `it` iterations done
    if (blockIdx.x == 0 && threadIdx.x == 0) {
      flags[1] = it;
      flags[2] = bad ? 1 : 0;
      __threadfence();
      flags[0] = it + 1;
    }
    return;
  }
  double beta = 0.0, alpha;
  if (it == 0) {
    alpha = delta != 0.0 ? gamma / delta : 0.0;
  } else {
    const double g_old = scal[2 * prv + 1], a_old = scal[M_A + prv];
    beta = g_old != 0.0 ? gamma / g_old : 0.0;
    const double den = delta - beta * gamma / a_old;
    alpha = den != 0.0 ? gamma / den : 0.0;
  }
  const int64_t n2 = n >> 1;
  const double2* w2 = reinterpret_cast<const double2*>(w);
  double2* r2 = reinterpret_cast<double2*>(r);
  double2* p2 = reinterpret_cast<double2*>(p);
  double2* s2 = reinterpret_cast<double2*>(sv);
  double2* x2 = reinterpret_cast<double2*>(xh);
  for (int64_t i = (int64_t)blockIdx.x * FEMO_BLOCK + threadIdx.x; i < n2; i += (int64_t)gridDim.x * FEMO_BLOCK) {
    const double2 wi = w2[i];
    double2 ri = r2[i], pi = p2[i], si = s2[i], xi = x2[i];
    pi.x = ri.x + beta * pi.x; pi.y = ri.y + beta * pi.y;
    si.x = wi.x + beta * si.x; si.y = wi.y + beta * si.y;
    xi.x += alpha * pi.x; xi.y += alpha * pi.y;
    ri.x -= alpha * si.x; ri.y -= alpha * si.y;
    p2[i] = pi; s2[i] = si; x2[i] = xi; r2[i] = ri;
  }
  if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
    const int64_t i = n - 1;
    const double pi = r[i] + beta * p[i], si = w[i] + beta * sv[i];
    p[i] = pi; sv[i] = si; xh[i] += alpha * pi; r[i] -= alpha * si;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    scal[M_A + cur] = alpha;                         // read by the next iteration (other parity slot)
    flags[1] = it + 1;
  }
}

// fold partial slots 0 (delta) and 1 (gamma) into buf[0], buf[1] = scal + 2*parity
__global__ __launch_bounds__(1024) void k_cgm_fold(int nblocks, int nblocks2, const double* __restrict__ partials,
                                                   double* __restrict__ buf, const int32_t* __restrict__ done) {
  if (*done) return;
  __shared__ double lds[1024 / 64];
  for (int j = 0; j < 2; ++j) {
    double acc = 0.0;
    for (int i = threadIdx.x; i < nblocks; i += 1024) acc += partials[(int64_t)j * FEMO_MAX_PARTIALS + i];
    // second launch of an overlapped SpMV (boundary slices): slots 2, 3
    for (int i = threadIdx.x; i < nblocks2; i += 1024) acc += partials[(int64_t)(j + 2) * FEMO_MAX_PARTIALS + i];
    const double t = femo_block_sum<1024>(acc, lds);
    if (threadIdx.x == 0) buf[j] = t;                // buf = {delta, gamma}, contiguous for one all-reduce
  }
}

// ---- BiCGSTAB on the scaled system (non-symmetric operators) --------------------------
// van der Vorst's recurrence; S A S keeps a unit diagonal, i.e. Jacobi preconditioning.
// Scalars live in scal[]; producers write partials, a one-block fold (plus all-reduce when
// nranks > 1) sits between producer and consumer.  Not the headline path: kept simple.
constexpr int B_RHO = 0, B_RHO_OLD = 1, B_ALPHA = 2, B_OMEGA = 3, B_R0V = 4, B_TS = 5, B_TT = 6, B_RR = 7, B_TOL2 = 8;

__global__ __launch_bounds__(FEMO_BLOCK) void k_bi_init(int64_t n, const double* __restrict__ b, const double* __restrict__ q,
                                                        const double* __restrict__ s, double* __restrict__ r, double* __restrict__ r0,
                                                        double* __restrict__ p, double* __restrict__ v, double* __restrict__ xh,
                                                        double* __restrict__ partials) {
  __shared__ double lds[FEMO_BLOCK / 64];
  double s0 = 0.0, s1 = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * FEMO_BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * FEMO_BLOCK) {
    const double si = s[i], bi = si * b[i];
    const double ri = q ? si * (b[i] - q[i]) : bi;
    r[i] = ri; r0[i] = ri; p[i] = 0.0; v[i] = 0.0; xh[i] = 0.0;
    s0 += ri * ri; s1 += bi * bi;
  }
  double t = femo_block_sum<FEMO_BLOCK>(s0, lds);
  if (threadIdx.x == 0) partials[blockIdx.x] = t;
  t = femo_block_sum<FEMO_BLOCK>(s1, lds);
  if (threadIdx.x == 0) partials[FEMO_MAX_PARTIALS + blockIdx.x] = t;
}

// fold `nsums` consecutive partial slots into scal[dst .. dst+nsums)
__global__ __launch_bounds__(1024) void k_bi_fold(int nblocks, int nsums, const double* __restrict__ partials, double* __restrict__ dst,
                                                  const int32_t* __restrict__ done) {
  if (*done) return;
  __shared__ double lds[1024 / 64];
  for (int j = 0; j < nsums; ++j) {
    double acc = 0.0;
    for (int i = threadIdx.x; i < nblocks; i += 1024) acc += partials[(int64_t)j * FEMO_MAX_PARTIALS + i];
    const double t = femo_block_sum<1024>(acc, lds);
    if (threadIdx.x == 0) dst[j] = t;
  }
}

// convergence test on (rho, rr) after `it` finished iterations (one thread; every later launch sees the flag)
__global__ void k_bi_check(int it, const double* __restrict__ scal, int32_t* __restrict__ flags) {
  if (flags[0]) return;
  const double rho = scal[B_RHO], rr = scal[B_RR];
  const bool bad = !(rr == rr) || !(rho == rho);
  if (rr <= scal[B_TOL2] || bad) {
    flags[1] = it;
    flags[2] = bad ? 1 : 0;
    flags[0] = 1;
  }
}

// p = r + beta (p - omega v)
__global__ __launch_bounds__(FEMO_BLOCK) void k_bi_p(int64_t n, int it, const double* __restrict__ scal, const double* __restrict__ r,
                                                     const double* __restrict__ v, double* __restrict__ p, const int32_t* __restrict__ flags) {
  if (flags[0]) return;
  double beta = 0.0, omega = 0.0;
  if (it > 0) {
    omega = scal[B_OMEGA];
    beta = (scal[B_RHO] / scal[B_RHO_OLD]) * (scal[B_ALPHA] / omega);
  }
  for (int64_t i = (int64_t)blockIdx.x * FEMO_BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * FEMO_BLOCK)
    p[i] = r[i] + beta * (p[i] - omega * v[i]);
}

// alpha = rho / (r0, v); s = r - alpha v
__global__ __launch_bounds__(FEMO_BLOCK) void k_bi_s(int64_t n, double* __restrict__ scal, const double* __restrict__ r,
                                                     const double* __restrict__ v, double* __restrict__ sv, const int32_t* __restrict__ done) {
  if (*done) return;
  const double alpha = scal[B_RHO] / scal[B_R0V];
  for (int64_t i = (int64_t)blockIdx.x * FEMO_BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * FEMO_BLOCK)
    sv[i] = r[i] - alpha * v[i];
  if (blockIdx.x == 0 && threadIdx.x == 0) scal[B_ALPHA] = alpha;   // not read by this launch
}

// omega = (t,s)/(t,t); x += alpha p + omega s; r = s - omega t; partials (r0, r), (r, r)
__global__ __launch_bounds__(FEMO_BLOCK) void k_bi_xr(int64_t n, int it, double* __restrict__ scal, const double* __restrict__ p,
                                                      const double* __restrict__ sv, const double* __restrict__ t,
                                                      const double* __restrict__ r0, double* __restrict__ xh, double* __restrict__ r,
                                                      double* __restrict__ partials, int32_t* __restrict__ flags) {
  if (flags[0]) return;
  __shared__ double lds[FEMO_BLOCK / 64];
  const double tt = scal[B_TT];
  const double omega = tt != 0.0 ? scal[B_TS] / tt : 0.0;
  const double alpha = scal[B_ALPHA];
  double s0 = 0.0, s1 = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * FEMO_BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * FEMO_BLOCK) {
    const double si = sv[i];
    xh[i] += alpha * p[i] + omega * si;
    const double ri = si - omega * t[i];
    r[i] = ri;
    s0 += r0[i] * ri; s1 += ri * ri;
  }
  double f = femo_block_sum<FEMO_BLOCK>(s0, lds);
  if (threadIdx.x == 0) partials[blockIdx.x] = f;
  f = femo_block_sum<FEMO_BLOCK>(s1, lds);
  if (threadIdx.x == 0) partials[FEMO_MAX_PARTIALS + blockIdx.x] = f;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    scal[B_OMEGA] = omega;               // read by the next launches only
    scal[B_RHO_OLD] = scal[B_RHO];       // rho of this iteration; B_RHO is overwritten by the next fold
    flags[1] = it + 1;
  }
}

// (rho, rr) of one fold land in buf[0], buf[1]; move them to their scal slots
__global__ void k_bi_set_rho(const double* __restrict__ buf, double* __restrict__ scal, const int32_t* __restrict__ done) {
  if (*done) return;
  scal[B_RHO] = buf[0];
  scal[B_RR] = buf[1];
}

// x = (add ? x : 0) + s .* xh
__global__ void k_unscale(int64_t n, int add, const double* __restrict__ s, const double* __restrict__ xh, double* __restrict__ x) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    x[i] = (add ? x[i] : 0.0) + s[i] * xh[i];
}

// the same, but only once the loop has converged (flags[0] != 0): enqueued behind every batch of the merged loop so that
// the poll that finds the solve converged finds the solution too -- one host round trip less per solve (round 6)
__global__ void k_unscale_done(int64_t n, int add, const double* __restrict__ s, const double* __restrict__ xh, double* __restrict__ x,
                               const int32_t* __restrict__ flags) {
  if (flags[0] == 0) return;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    x[i] = (add ? x[i] : 0.0) + s[i] * xh[i];
}

__global__ __launch_bounds__(FEMO_BLOCK) void k_dot(int64_t n, const double* __restrict__ a, const double* __restrict__ b,
                                                    double* __restrict__ partials) {
  __shared__ double lds[FEMO_BLOCK / 64];
  double s = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * FEMO_BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * FEMO_BLOCK) s += a[i] * b[i];
  const double t = femo_block_sum<FEMO_BLOCK>(s, lds);
  if (threadIdx.x == 0) partials[blockIdx.x] = t;
}

// out[j] = sum of partial slot j (consecutive slots starting at `partials`)
__global__ __launch_bounds__(1024) void k_reduce_partials_at(int nblocks, int nsums, const double* __restrict__ partials,
                                                             double* __restrict__ out) {
  __shared__ double lds[1024 / 64];
  for (int j = 0; j < nsums; ++j) {
    double acc = 0.0;
    for (int i = threadIdx.x; i < nblocks; i += 1024) acc += partials[(int64_t)j * FEMO_MAX_PARTIALS + i];
    const double t = femo_block_sum<1024>(acc, lds);
    if (threadIdx.x == 0) out[j] = t;
  }
}

__global__ void k_pack(int64_t n, const int32_t* __restrict__ idx, const double* __restrict__ x, double* __restrict__ buf) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) buf[i] = x[idx[i]];
}

inline int vec_grid(const femo_ctx* ctx, int64_t n) {
  int64_t g = (n / 2 + FEMO_BLOCK - 1) / FEMO_BLOCK;
  const int64_t cap = (int64_t)ctx->n_cu * 8;
  if (g > cap) g = cap;
  if (g > FEMO_MAX_PARTIALS) g = FEMO_MAX_PARTIALS;
  if (g < 1) g = 1;
  return (int)g;
}

}  // namespace

int femo_spmv_grid(const femo_mesh* m) {
  int64_t g = (m->n_slices + 3) / 4;           // one slice per wave if the mesh is small
  const int64_t cap = (int64_t)m->ctx->n_cu * 4;  // 4 x 256 threads resident per CU (~100 VGPRs)
  if (g > cap) g = cap;
  if (g > FEMO_MAX_PARTIALS) g = FEMO_MAX_PARTIALS;
  g = (g + 7) & ~int64_t(7);                   // whole XCD groups
  if (g < 8) g = 8;
  return (int)g;
}

static int launch_spmv(const femo_mat* A, const double* vals, const double* x, double* y,
                       double* partials, const int32_t* done, bool unit = false, bool dot2 = false,
                       const int32_t* slice_list = nullptr, int64_t n_list = 0, hipStream_t stream = nullptr,
                       const double* dvec = nullptr, bool dot_yy = false, bool dot3 = false, const SpmvGhost* gh = nullptr) {
  const femo_mesh* m = A->mesh;
  const int64_t n_walk = slice_list ? n_list : m->n_slices;
  if (n_walk == 0 && !partials) return 0;
  int64_t g = femo_spmv_grid(m);
  if (slice_list) {                                  // size the grid for the subset (same rules)
    g = std::min<int64_t>(g, std::max<int64_t>(8, ((n_walk + 3) / 4 + 7) & ~int64_t(7)));
  }
  hipStream_t st = stream ? stream : m->ctx->stream;
#define FEMO_SPMV_ARGS m->n_rows, m->n_slices, m->d_mptr, m->d_cols, m->d_cols16, m->d_sdelta, m->sdelta_stride, vals, A->d_diag, x, y, partials, done, slice_list, n_list, dvec
  // streaming hint on the matrix loads only where the stored values cannot stay in the Infinity Cache between two products
  const bool nt = (int64_t)m->sell_entries * (int64_t)sizeof(double) > FEMO_LLC_MATRIX_BYTES;
#define FEMO_SPMV_LAUNCH(DOT, UNIT)                                                                                     \
  do {                                                                                                                  \
    if (nt) hipLaunchKernelGGL((k_spmv_sell<DOT, UNIT, true>), dim3(g), dim3(FEMO_BLOCK), 0, st, FEMO_SPMV_ARGS);        \
    else hipLaunchKernelGGL((k_spmv_sell<DOT, UNIT, false>), dim3(g), dim3(FEMO_BLOCK), 0, st, FEMO_SPMV_ARGS);          \
  } while (0)
  if (gh != nullptr) {                                    // the merged loop's one launch over [interior | boundary] slices
    FEMO_REQUIRE(partials && unit && dot3 && slice_list, "ghost-aware product: merged-loop variant only");
    if (nt) hipLaunchKernelGGL((k_spmv_sell<4, true, true, true>), dim3(g), dim3(FEMO_BLOCK), 0, st, FEMO_SPMV_ARGS, *gh);
    else hipLaunchKernelGGL((k_spmv_sell<4, true, false, true>), dim3(g), dim3(FEMO_BLOCK), 0, st, FEMO_SPMV_ARGS, *gh);
  }
  else if (partials && unit && dot3) FEMO_SPMV_LAUNCH(4, true);
  else if (partials && unit && dot_yy) FEMO_SPMV_LAUNCH(3, true);
  else if (partials && unit && dot2) FEMO_SPMV_LAUNCH(2, true);
  else if (partials && unit) FEMO_SPMV_LAUNCH(1, true);
  else if (partials) FEMO_SPMV_LAUNCH(1, false);
  else if (unit) FEMO_SPMV_LAUNCH(0, true);
  else FEMO_SPMV_LAUNCH(0, false);
#undef FEMO_SPMV_LAUNCH
#undef FEMO_SPMV_ARGS
  FEMO_HIP_CHECK(hipGetLastError());
  return 0;
}

// Ghost refresh of x overlapped with the interior rows of y = A x (nranks > 1):
//   comm stream : pack + grouped ncclSend/ncclRecv into the ghost tail of x
//   main stream : SpMV over the slices without ghost columns, then (after the halo
//                 event) over the slices that read ghosts.
// Partials (if requested) land in slot pairs: interior at `partials`, boundary at
// `partials + 2*FEMO_MAX_PARTIALS`; *g_int / *g_bnd return the block counts to fold.
static int halo_spmv_overlapped(const femo_mat* A, const double* vals, double* x, double* y, double* partials,
                                const int32_t* done, bool unit, bool dot2, int* g_int, int* g_bnd,
                                const double* dvec = nullptr, int n_slots = 2) {
  femo_mesh* m = A->mesh;
  femo_ctx* ctx = m->ctx;
  hipStream_t st = ctx->stream, cs = ctx->comm_stream;
  FEMO_HIP_CHECK(hipEventRecord(ctx->ev_main, st));
  FEMO_HIP_CHECK(hipStreamWaitEvent(cs, ctx->ev_main, 0));
  {
    femo_vec v; v.ctx = ctx; v.d = x; v.n = m->n_vert; v.owned = false;
    FEMO_TRY(femo_halo_exchange_on(m, &v, cs));
  }
  FEMO_HIP_CHECK(hipEventRecord(ctx->ev_comm, cs));
  auto grid_of = [&](int64_t n_walk) {
    int64_t g = femo_spmv_grid(m);
    return (int)std::min<int64_t>(g, std::max<int64_t>(8, ((n_walk + 3) / 4 + 7) & ~int64_t(7)));
  };
  if (g_int) *g_int = grid_of(m->n_int);
  if (g_bnd) *g_bnd = grid_of(m->n_bnd);
  // n_slots == 3: the merged PCG's triple [x.Ax | Ax.Ax | dvec.Ax], boundary launch three slots further on
  const bool dot3 = n_slots == 3;
  FEMO_TRY(launch_spmv(A, vals, x, y, partials, done, unit, dot2, m->d_slices_int, m->n_int, st, dvec, false, dot3));
  FEMO_HIP_CHECK(hipStreamWaitEvent(st, ctx->ev_comm, 0));
  FEMO_TRY(launch_spmv(A, vals, x, y, partials ? partials + n_slots * FEMO_MAX_PARTIALS : nullptr, done, unit, dot2,
                       m->d_slices_bnd, m->n_bnd, st, dvec, false, dot3));
  return 0;
}

// The same product when the ghost refresh of x is ALREADY in flight on the communication stream (the merged BPX-PCG's
// prolongation started it, femo_pc_merged_apply): interior slices now, boundary slices behind the halo event.
static int halo_spmv_inflight(const femo_mat* A, const double* vals, double* x, double* y, double* partials,
                              const int32_t* done, bool unit, int* g_int, int* g_bnd, const double* dvec, int n_slots) {
  femo_mesh* m = A->mesh;
  femo_ctx* ctx = m->ctx;
  hipStream_t st = ctx->stream;
  auto grid_of = [&](int64_t n_walk) {
    int64_t g = femo_spmv_grid(m);
    return (int)std::min<int64_t>(g, std::max<int64_t>(8, ((n_walk + 3) / 4 + 7) & ~int64_t(7)));
  };
  if (g_int) *g_int = grid_of(m->n_int);
  if (g_bnd) *g_bnd = grid_of(m->n_bnd);
  const bool dot3 = n_slots == 3;
  if (femo_halo_direct_ready(m) && m->d_slices_all != nullptr && dot3 && unit && partials != nullptr) {
    // round 6: ONE launch.  Interior slices first in every XCD's range, the slices with ghost columns behind them; their
    // waves wait for the neighbours' counters and read the ghosts from this rank's inbox (k_spmv_sell<.., GH = true>).
    FemoHaloDirect* h = m->hd;
    if (ctx->emu != nullptr) FEMO_TRY(femo_emu_rendezvous(ctx, st));       // (emulated ranks meet on the host, halo_direct.hip)
    SpmvGhost gh;
    gh.cnt = h->counters; gh.blocks = h->d_blocks; gh.n_nbr = m->n_nbr; gh.epoch = h->loop_epoch; gh.err = h->d_err;
    gh.ghost = h->inbox + (int64_t)(h->loop_epoch & 1ull) * h->n_ghost; gh.n_own = m->n_rows;
    if (g_int) *g_int = femo_spmv_grid(m);
    if (g_bnd) *g_bnd = 0;
    return launch_spmv(A, vals, x, y, partials, done, unit, false, m->d_slices_all, m->n_slices, st, dvec, false, dot3, &gh);
  }
  FEMO_TRY(launch_spmv(A, vals, x, y, partials, done, unit, false, m->d_slices_int, m->n_int, st, dvec, false, dot3));
  if (femo_halo_direct_ready(m)) {
    // device-initiated refresh (round 6): the neighbours' prolongations stored the new direction into this rank's inbox
    // while the interior slices were multiplied; one small launch of the SAME stream waits for their counters and moves
    // the generation into the ghost tail -- no second stream, no event
    FEMO_TRY(femo_halo_direct_pull(m, m->hd->loop_epoch, x + m->n_rows, st));
  } else {
    FEMO_HIP_CHECK(hipStreamWaitEvent(st, ctx->ev_comm, 0));
  }
  FEMO_TRY(launch_spmv(A, vals, x, y, partials ? partials + n_slots * FEMO_MAX_PARTIALS : nullptr, done, unit, false,
                       m->d_slices_bnd, m->n_bnd, st, dvec, false, dot3));
  return 0;
}

int femo_launch_spmv(const femo_mat* A, const double* vals, const double* x, double* y, double* partials) {
  return launch_spmv(A, vals, x, y, partials, nullptr);
}

int femo_mat_ensure_transpose(femo_mat* A) {
  femo_mesh* m = A->mesh;
  femo_ctx* ctx = m->ctx;
  if (A->valsT_valid) return 0;
  if (m->sell_entries == 0) { A->valsT_valid = true; return 0; }
  if (!m->d_tperm) {
    FEMO_REQUIRE(m->sell_entries < (int64_t(1) << 31), "pattern too large for int32 transpose map");
    FEMO_HIP_CHECK(hipMalloc(&m->d_tperm, m->sell_entries * sizeof(int32_t)));
    FEMO_HIP_CHECK(hipMemsetAsync(ctx->d_flags + 3, 0, sizeof(int32_t), ctx->stream));
    const int64_t nr = m->n_slices * 64;
    hipLaunchKernelGGL(k_build_tperm, dim3((nr + 255) / 256), dim3(256), 0, ctx->stream, m->n_rows, m->n_slices, m->d_mptr, m->d_cols, m->d_rowlen, m->d_rowreal, m->d_tperm, ctx->d_flags + 3);
    FEMO_HIP_CHECK(hipGetLastError());
    int32_t err = 0;
    FEMO_HIP_CHECK(hipMemcpyAsync(&err, ctx->d_flags + 3, sizeof err, hipMemcpyDeviceToHost, ctx->stream));
    FEMO_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    if (err != 0) {
      hipFree(m->d_tperm);
      m->d_tperm = nullptr;
      femo_set_error(err == 1 ? "transposed operator needs entries owned by another rank"
                              : "sparsity pattern is not structurally symmetric");
      return 3;
    }
  }
  if (!A->d_valsT) FEMO_HIP_CHECK(hipMalloc(&A->d_valsT, m->sell_entries * sizeof(double)));
  hipLaunchKernelGGL(k_gather, dim3(2048), dim3(256), 0, ctx->stream, m->sell_entries, m->d_tperm, A->d_vals, A->d_valsT);
  FEMO_HIP_CHECK(hipGetLastError());
  A->valsT_valid = true;
  return 0;
}

// flag[slice] = 1 if any stored column of the slice is a ghost (>= n_rows)
__global__ void k_flag_ghost_slices(int64_t n_rows, int64_t n_slices, const int64_t* __restrict__ mptr,
                                    const int32_t* __restrict__ cols, int32_t* __restrict__ flag) {
  const int lane = threadIdx.x & 63;
  const int64_t slice = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (slice >= n_slices) return;
  const int64_t base = mptr[slice];
  const int wm = (int)((mptr[slice + 1] - base) >> 6);
  int any = 0;
  for (int k = 0; k < wm; ++k) any |= cols[femo_sell_index(base, k, lane)] >= n_rows;   // padding = own row < n_rows
  const unsigned long long b = __ballot(any);
  if (lane == 0) flag[slice] = b != 0ull;
}

int femo_mesh_classify_slices(femo_mesh* m) {
  femo_ctx* ctx = m->ctx;
  hipFree(m->d_slices_int); hipFree(m->d_slices_bnd);
  m->d_slices_int = m->d_slices_bnd = nullptr;
  m->n_int = m->n_bnd = 0;
  if (m->n_slices == 0) return 0;
  int32_t* d_flag = nullptr;
  FEMO_HIP_CHECK(hipMalloc(&d_flag, m->n_slices * sizeof(int32_t)));
  hipLaunchKernelGGL(k_flag_ghost_slices, dim3((unsigned)((m->n_slices + 3) / 4)), dim3(256), 0, ctx->stream, m->n_rows, m->n_slices, m->d_mptr, m->d_cols, d_flag);
  FEMO_HIP_CHECK(hipGetLastError());
  std::vector<int32_t> flag(m->n_slices), li, lb;
  FEMO_HIP_CHECK(hipMemcpyAsync(flag.data(), d_flag, m->n_slices * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
  FEMO_HIP_CHECK(hipStreamSynchronize(ctx->stream));
  hipFree(d_flag);
  for (int64_t s = 0; s < m->n_slices; ++s) (flag[s] ? lb : li).push_back((int32_t)s);
  m->n_int = (int64_t)li.size(); m->n_bnd = (int64_t)lb.size();
  FEMO_HIP_CHECK(hipMalloc(&m->d_slices_int, std::max<size_t>(li.size(), 1) * sizeof(int32_t)));
  FEMO_HIP_CHECK(hipMalloc(&m->d_slices_bnd, std::max<size_t>(lb.size(), 1) * sizeof(int32_t)));
  // one list for the single-launch product of the merged loop (round 6): the kernel gives XCD k the k-th eighth of the list;
  // inside every eighth the interior slices come first, the slices with ghost columns (sign bit set) last
  hipFree(m->d_slices_all); m->d_slices_all = nullptr;
  if (!lb.empty() && femo_env_flag("FEMO_SPMV_TWO_LAUNCHES") == false) {
    std::vector<int32_t> all((size_t)m->n_slices);
    size_t w = 0;
    for (int xcd = 0; xcd < 8; ++xcd) {
      const int64_t lo = m->n_slices * xcd / 8, hi = m->n_slices * (xcd + 1) / 8;
      for (int64_t sl = lo; sl < hi; ++sl) if (!flag[(size_t)sl]) all[w++] = (int32_t)sl;
      for (int64_t sl = lo; sl < hi; ++sl) if (flag[(size_t)sl]) all[w++] = (int32_t)sl | (int32_t)0x80000000;
    }
    FEMO_HIP_CHECK(hipMalloc(&m->d_slices_all, all.size() * sizeof(int32_t)));
    FEMO_HIP_CHECK(hipMemcpyAsync(m->d_slices_all, all.data(), all.size() * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
    FEMO_HIP_CHECK(hipStreamSynchronize(ctx->stream));
  }
  if (!li.empty()) FEMO_HIP_CHECK(hipMemcpyAsync(m->d_slices_int, li.data(), li.size() * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
  if (!lb.empty()) FEMO_HIP_CHECK(hipMemcpyAsync(m->d_slices_bnd, lb.data(), lb.size() * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
  FEMO_HIP_CHECK(hipStreamSynchronize(ctx->stream));
  return 0;
}

// ---------------------------------------------------------------- halo ------
int femo_halo_exchange_on(femo_mesh* m, femo_vec* x, hipStream_t st) {
  FEMO_REQUIRE(m && x, "null argument");
  if (m->n_nbr == 0) return 0;
  // Nobody has written x since its ghosts were refreshed: they are still the owners' values.  Skipping is SPMD-consistent
  // (every rank runs the same sequence of writes and refreshes) and it keeps host mirrors of x valid: round 5 refreshed u
  // for J, dJ/du, ... after its copy-out, each refresh a new generation, each new generation a real re-upload of the host
  // copy by the next operator call (two blocking PCIe uploads + their waits per cycle and rank, round 6).
  static const bool always = femo_env_flag("FEMO_HALO_ALWAYS");       // (comparison runs; read once)
  if (!always && x->ghost_gen == x->gen && x->uid != 0 && x->n >= m->n_vert) return 0;
  femo_vec_touch(x);                                  // ghost entries change
  x->ghost_gen = x->gen;
  femo_ctx* ctx = m->ctx;
  FEMO_REQUIRE(ctx->comm != nullptr || ctx->emu != nullptr || ctx->model, "halo exchange before femo_comm_init");
  FEMO_REQUIRE(x->n >= m->n_vert, "vector shorter than n_vert");
  if (femo_halo_direct_ready(m)) return femo_halo_direct_exchange(m, x->d, x->d + m->n_rows, st);   // device-initiated (round 6)
  const int64_t ns = m->send_ptr[m->n_nbr];
  if (ns > 0) {
    hipLaunchKernelGGL(k_pack, dim3((unsigned)((ns + 255) / 256)), dim3(256), 0, st, ns, m->d_send_idx, x->d, m->d_send_buf);
    FEMO_HIP_CHECK(hipGetLastError());
  }
  return femo_coll_neighbors(ctx, m->n_nbr, m->nbr.data(), m->send_ptr.data(), m->d_send_buf, m->recv_ptr.data(),
                             x->d + m->n_rows, st);
}

extern "C" int femo_halo_exchange(femo_mesh* m, femo_vec* x) {
  FEMO_REQUIRE(m && x, "null argument");
  return femo_halo_exchange_on(m, x, m->ctx->stream);
}

static int halo_raw(femo_mesh* m, double* x) {
  femo_vec v;
  v.ctx = m->ctx; v.d = x; v.n = m->n_vert; v.owned = false;
  return femo_halo_exchange(m, &v);
}

// ghost tail of y -> owners, who add it to their entries (the transpose of the ghost refresh)
static int halo_reverse_add(femo_mesh* m, double* y, hipStream_t st) {
  if (m->n_nbr == 0) return 0;
  femo_ctx* ctx = m->ctx;
  FEMO_REQUIRE(ctx->comm != nullptr || ctx->emu != nullptr || ctx->model, "halo exchange before femo_comm_init");
  const int64_t ns = m->send_ptr[m->n_nbr];
  // roles swapped: what this rank receives in a forward exchange (its ghost tail) is what it sends back
  FEMO_TRY(femo_coll_neighbors(ctx, m->n_nbr, m->nbr.data(), m->recv_ptr.data(), y + m->n_rows, m->send_ptr.data(), m->d_send_buf, st));
  if (ns > 0) {
    hipLaunchKernelGGL(k_unpack_add, dim3((unsigned)std::min<int64_t>((ns + 255) / 256, 2048)), dim3(256), 0, st, ns, m->d_send_idx, m->d_send_buf, y);
    FEMO_HIP_CHECK(hipGetLastError());
  }
  return 0;
}

// y = A^T x on a partitioned mesh, A given by its own (untransposed) values
static int spmv_transposed_scatter(const femo_mat* A, const double* vals, bool unit, const double* x, double* y, const int32_t* done) {
  femo_mesh* m = A->mesh;
  hipStream_t st = m->ctx->stream;
  FEMO_HIP_CHECK(hipMemsetAsync(y, 0, (size_t)m->n_vert * sizeof(double), st));
  if (m->n_rows > 0) {
    const unsigned g = (unsigned)((m->n_rows + FEMO_BLOCK - 1) / FEMO_BLOCK);
    if (unit) hipLaunchKernelGGL((k_spmv_sell_T_scatter<true>), dim3(g), dim3(FEMO_BLOCK), 0, st, m->n_rows, m->n_slices, m->d_mptr, m->d_cols, m->d_rowlen, vals, A->d_diag, x, y, done);
    else hipLaunchKernelGGL((k_spmv_sell_T_scatter<false>), dim3(g), dim3(FEMO_BLOCK), 0, st, m->n_rows, m->n_slices, m->d_mptr, m->d_cols, m->d_rowlen, vals, A->d_diag, x, y, done);
    FEMO_HIP_CHECK(hipGetLastError());
  }
  return halo_reverse_add(m, y, st);
}

// can the explicit transposed values be formed on this rank?  (no on partitioned meshes: see k_build_tperm)
static bool transpose_is_local(const femo_mesh* m) { return m->n_nbr == 0; }

// ----------------------------------------------------------------- API ------
extern "C" int femo_mat_spmv(const femo_mat* A, int transpose, const femo_vec* x, femo_vec* y) {
  FEMO_REQUIRE(A && x && y, "null argument");
  femo_mesh* m = A->mesh;
  FEMO_REQUIRE(x->n >= m->n_vert && y->n >= m->n_rows, "vector size mismatch in spmv");
  FEMO_REQUIRE(x->d != y->d, "spmv cannot run in place");
  FEMO_TRY(femo_vec_await(x));             // a deferred upload of x (every reader awaits)
  femo_vec_touch(y);
  const double* vals = A->d_vals;
  if (transpose && !transpose_is_local(m)) {
    FEMO_REQUIRE(y->n >= m->n_vert, "transposed product on a partitioned mesh: y needs room for the ghost contributions");
    return spmv_transposed_scatter(A, A->d_vals, false, x->d, y->d, nullptr);
  }
  if (transpose) {
    FEMO_TRY(femo_mat_ensure_transpose(const_cast<femo_mat*>(A)));
    vals = A->d_valsT;
  }
  if (m->n_nbr > 0) femo_vec_touch(const_cast<femo_vec*>(x));     // its ghost entries are refreshed
  if (m->n_nbr > 0 && m->d_slices_int != nullptr)
    return halo_spmv_overlapped(A, vals, x->d, y->d, nullptr, nullptr, false, false, nullptr, nullptr);
  if (m->n_nbr > 0) FEMO_TRY(halo_raw(m, x->d));
  return launch_spmv(A, vals, x->d, y->d, nullptr, nullptr);
}

extern "C" int femo_vec_dot(const femo_vec* x, const femo_vec* y, int64_t n, double* out) {
  FEMO_REQUIRE(x && y && out, "null argument");
  FEMO_REQUIRE(n <= x->n && n <= y->n, "dot length exceeds vector size");
  FEMO_TRY(femo_vec_await(x));
  FEMO_TRY(femo_vec_await(y));
  femo_ctx* ctx = x->ctx;
  const int g = vec_grid(ctx, n);
  hipLaunchKernelGGL(k_dot, dim3(g), dim3(FEMO_BLOCK), 0, ctx->stream, n, x->d, y->d, ctx->d_partials);
  FEMO_HIP_CHECK(hipGetLastError());
  return femo_reduce_to_host(ctx, g, 1, out);
}

namespace {
// rb != nullptr: one more sum behind the k pairs (slot k) -- rho_0 of the Krylov loops for the right-hand side rb and the
// operator with diagonal `diag`, sum over the rows that are not identity rows of (rb_i / sqrt(diag_i))^2, formed exactly as
// k_invsqrt_diag + k_cg_init form it: Newton takes the solver's own "nothing to iterate on" decision from it (round 6)
struct DotPairs { const double* a[4]; const double* b[4]; int k; const double* rb; const double* diag; const uint8_t* idrow; };
__global__ __launch_bounds__(FEMO_BLOCK) void k_dots(int64_t n, DotPairs d, double* __restrict__ partials) {
  __shared__ double lds[FEMO_BLOCK / 64];
  double s[4] = {0.0, 0.0, 0.0, 0.0};
  double rho = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * FEMO_BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * FEMO_BLOCK) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (j < d.k) s[j] += d.a[j][i] * d.b[j][i];
    if (d.rb != nullptr && !(d.idrow != nullptr && d.idrow[i])) {
      const double si = 1.0 / sqrt(d.diag[i]);
      const double ri = si * d.rb[i];
      rho += ri * ri;
    }
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    if (j < d.k) {
      const double t = femo_block_sum<FEMO_BLOCK>(s[j], lds);
      if (threadIdx.x == 0) partials[(int64_t)j * FEMO_MAX_PARTIALS + blockIdx.x] = t;
    }
  }
  if (d.rb != nullptr) {
    const double t = femo_block_sum<FEMO_BLOCK>(rho, lds);
    if (threadIdx.x == 0) partials[(int64_t)d.k * FEMO_MAX_PARTIALS + blockIdx.x] = t;
  }
}
}  // namespace

// x = what a Krylov solve of A x = b returns when it decides not to iterate from the zero guess: b_i / diag_i on the identity
// rows of the last assembly (the loops solve them up front, k_cg_init), 0 elsewhere
__global__ void k_identity_solve(int64_t n, int64_t n_all, const double* __restrict__ b, const double* __restrict__ diag,
                                 const uint8_t* __restrict__ idrow, double* __restrict__ x) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_all; i += (int64_t)gridDim.x * blockDim.x) {
    double v = 0.0;
    if (i < n && idrow != nullptr && idrow[i]) { const double si = 1.0 / sqrt(diag[i]); v = si * (si * b[i]); }
    x[i] = v;
  }
}
extern "C" int femo_mat_identity_solve(const femo_mat* A, const femo_vec* b, femo_vec* x) {
  FEMO_REQUIRE(A && b && x, "null argument");
  const femo_mesh* m = A->mesh;
  FEMO_REQUIRE(b->n >= m->n_rows && x->n >= m->n_rows, "vector size mismatch");
  FEMO_TRY(femo_vec_await(b));
  femo_vec_touch(x);
  if (x->n == 0) return 0;
  hipLaunchKernelGGL(k_identity_solve, dim3(2048), dim3(256), 0, m->ctx->stream, m->n_rows, x->n, b->d, A->d_diag, A->has_idrows ? A->d_idrows : nullptr, x->d);
  FEMO_HIP_CHECK(hipGetLastError());
  return 0;
}

static int vec_dots(int k, const femo_vec* const* x, const femo_vec* const* y, int64_t n, double* out, const femo_mat* A, const femo_vec* rb);
extern "C" int femo_vec_dots(int k, const femo_vec* const* x, const femo_vec* const* y, int64_t n, double* out) {
  return vec_dots(k, x, y, n, out, nullptr, nullptr);
}
// ... and out[k] = rho_0 a Krylov solve of A with right-hand side rb would start from (see k_dots): the same launch, the same
// reduction, the same host synchronisation
extern "C" int femo_vec_dots_rhs(int k, const femo_vec* const* x, const femo_vec* const* y, int64_t n, double* out,
                                 const femo_mat* A, const femo_vec* rb) {
  FEMO_REQUIRE(A && rb && A->d_diag, "femo_vec_dots_rhs: null matrix / right-hand side");
  FEMO_REQUIRE(n <= rb->n && n <= A->mesh->n_rows, "femo_vec_dots_rhs: length exceeds the operator's rows");
  return vec_dots(k, x, y, n, out, A, rb);
}
static int vec_dots(int k, const femo_vec* const* x, const femo_vec* const* y, int64_t n, double* out, const femo_mat* A, const femo_vec* rb) {
  FEMO_REQUIRE(x && y && out && k >= 1 && k <= 4, "femo_vec_dots: 1..4 pairs");
  DotPairs d;
  d.k = k;
  d.rb = nullptr; d.diag = nullptr; d.idrow = nullptr;
  if (A != nullptr) {
    FEMO_TRY(femo_vec_await(rb));
    d.rb = rb->d; d.diag = A->d_diag; d.idrow = A->has_idrows ? A->d_idrows : nullptr;
  }
  for (int j = 0; j < 4; ++j) {
    const int jj = j < k ? j : 0;
    FEMO_REQUIRE(x[jj] && y[jj] && n <= x[jj]->n && n <= y[jj]->n, "dot length exceeds vector size");
    d.a[j] = x[jj]->d; d.b[j] = y[jj]->d;
    FEMO_TRY(femo_vec_await(x[jj]));
    FEMO_TRY(femo_vec_await(y[jj]));
  }
  femo_ctx* ctx = x[0]->ctx;
  const int g = vec_grid(ctx, n);
  hipLaunchKernelGGL(k_dots, dim3(g), dim3(FEMO_BLOCK), 0, ctx->stream, n, d, ctx->d_partials);
  FEMO_HIP_CHECK(hipGetLastError());
  return femo_reduce_to_host(ctx, g, k + (A != nullptr ? 1 : 0), out);
}

static int ensure_scaled(femo_mat* A, bool transpose);

extern "C" int femo_bench_spmv(const femo_mat* A, const femo_vec* x, femo_vec* y, int reps, double* ms_per_launch) {
  FEMO_REQUIRE(A && x && y && ms_per_launch && reps > 0, "bad argument");
  femo_ctx* ctx = A->mesh->ctx;
  FEMO_REQUIRE(x->n >= A->mesh->n_vert && y->n >= A->mesh->n_rows, "vector size mismatch");
  femo_vec_touch(y);
  // the launch the CG loop issues: scaled operator, unit diagonal, fused p.Ap partials
  FEMO_TRY(ensure_scaled(const_cast<femo_mat*>(A), false));
  for (int i = 0; i < 3; ++i) FEMO_TRY(launch_spmv(A, A->d_valsS, x->d, y->d, ctx->d_partials + 3 * FEMO_MAX_PARTIALS, nullptr, true));
  FEMO_HIP_CHECK(hipEventRecord(ctx->ev0, ctx->stream));
  for (int i = 0; i < reps; ++i) FEMO_TRY(launch_spmv(A, A->d_valsS, x->d, y->d, ctx->d_partials + 3 * FEMO_MAX_PARTIALS, nullptr, true));
  FEMO_HIP_CHECK(hipEventRecord(ctx->ev1, ctx->stream));
  FEMO_HIP_CHECK(hipEventSynchronize(ctx->ev1));
  float ms = 0.f;
  FEMO_HIP_CHECK(hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
  *ms_per_launch = (double)ms / reps;
  return 0;
}

namespace {
struct CgWork {
  double *r, *p, *q, *xh, *sv, *t, *r0;
};

// need: 0 = standard CG, 1 = single-reduction CG (+sv), 2 = BiCGSTAB (+sv, t, r0)
int ensure_work(femo_ctx* ctx, int64_t n_rows, int64_t n_vert, CgWork& w, int need, bool zero = true) {
  const int64_t len = std::max(n_rows, n_vert) + 2;
  if (ctx->cg_n < len) {
    hipFree(ctx->cg_r); hipFree(ctx->cg_p); hipFree(ctx->cg_q); hipFree(ctx->cg_dinv); hipFree(ctx->cg_s);
    hipFree(ctx->cg_t); hipFree(ctx->cg_r0);
    ctx->cg_r = ctx->cg_p = ctx->cg_q = ctx->cg_dinv = ctx->cg_s = ctx->cg_t = ctx->cg_r0 = nullptr;
    ctx->cg_n = 0;
    FEMO_HIP_CHECK(hipMalloc(&ctx->cg_r, len * sizeof(double)));
    FEMO_HIP_CHECK(hipMalloc(&ctx->cg_p, len * sizeof(double)));
    FEMO_HIP_CHECK(hipMalloc(&ctx->cg_q, len * sizeof(double)));
    FEMO_HIP_CHECK(hipMalloc(&ctx->cg_dinv, len * sizeof(double)));   // holds xh
    ctx->cg_n = len;
  }
  if (need >= 1 && !ctx->cg_s) FEMO_HIP_CHECK(hipMalloc(&ctx->cg_s, ctx->cg_n * sizeof(double)));
  if (need >= 2 && !ctx->cg_t) FEMO_HIP_CHECK(hipMalloc(&ctx->cg_t, ctx->cg_n * sizeof(double)));
  if (need >= 2 && !ctx->cg_r0) FEMO_HIP_CHECK(hipMalloc(&ctx->cg_r0, ctx->cg_n * sizeof(double)));
  // (zero == false: the caller's init kernel writes every entry it will read -- no ghost tail)
  if (zero) {
    FEMO_HIP_CHECK(hipMemsetAsync(ctx->cg_p, 0, ctx->cg_n * sizeof(double), ctx->stream));
    FEMO_HIP_CHECK(hipMemsetAsync(ctx->cg_r, 0, ctx->cg_n * sizeof(double), ctx->stream));
    if (need >= 1) FEMO_HIP_CHECK(hipMemsetAsync(ctx->cg_s, 0, ctx->cg_n * sizeof(double), ctx->stream));
  }
  w.r = ctx->cg_r; w.p = ctx->cg_p; w.q = ctx->cg_q; w.xh = ctx->cg_dinv; w.sv = ctx->cg_s;
  w.t = ctx->cg_t; w.r0 = ctx->cg_r0;
  return 0;
}
}  // namespace

// S = diag^-1/2 of the current assembly (ghost entries from their owners)
static int ensure_s(femo_mat* A) {
  femo_mesh* m = A->mesh;
  femo_ctx* ctx = m->ctx;
  if (A->s_valid) return 0;
  const int64_t nd = std::max<int64_t>(m->n_vert, m->n_slices * FEMO_WAVE) + 2;
  if (!A->d_s) FEMO_HIP_CHECK(hipMalloc(&A->d_s, nd * sizeof(double)));
  if (m->n_rows > 0) {
    hipLaunchKernelGGL(k_invsqrt_diag, dim3(2048), dim3(256), 0, ctx->stream, m->n_rows, A->d_diag, A->d_s);
    FEMO_HIP_CHECK(hipGetLastError());
  }
  if (m->n_nbr > 0) {   // ghost columns need the owner's scale factor
    femo_vec v; v.ctx = ctx; v.d = A->d_s; v.n = m->n_vert; v.owned = false;
    FEMO_TRY(femo_halo_exchange(m, &v));
  }
  A->s_valid = true;
  return 0;
}

// ... and the scaled values S A S (or S A^T S): what the Krylov loops iterate on.  Kept apart
// from ensure_s because a solve that stops on its initial residual never needs them.
static int ensure_scaled(femo_mat* A, bool transpose) {
  femo_mesh* m = A->mesh;
  femo_ctx* ctx = m->ctx;
  if (A->scaled_valid && A->scaled_transposed == transpose) return 0;
  const double* src = A->d_vals;
  if (transpose) {
    FEMO_TRY(femo_mat_ensure_transpose(A));
    src = A->d_valsT;
  }
  FEMO_TRY(ensure_s(A));
  if (!A->d_valsS) FEMO_HIP_CHECK(hipMalloc(&A->d_valsS, std::max<int64_t>(m->sell_entries, 1) * sizeof(double) + 64));
  if (m->n_slices > 0) {
    hipLaunchKernelGGL(k_scale_sell, dim3(2048), dim3(FEMO_BLOCK), 0, ctx->stream, m->n_slices, m->d_mptr, m->d_cols, m->d_sdelta, m->sdelta_stride, src, A->d_s, m->n_vert, A->d_valsS);
    FEMO_HIP_CHECK(hipGetLastError());
  }
  A->scaled_valid = true;
  A->scaled_transposed = transpose;
  return 0;
}

// ---- merged BPX-PCG (round 4) ---------------------------------------------------------------------------------------
// One all-reduce + one halo exchange per iteration on N ranks, five launches per iteration on one (SpMV, brick
// restriction of q = A p, coarse lattice + vector updates, fine lattice, mesh prolongation + direction update), one
// host synchronisation before the loop and one after it.  See femo_internal.h "merged BPX-PCG" and DESIGN.md section 4.
// Set-up on the device: rho0 = r0.r0 and bb = b.D^-1 b (folded here on one rank, reduced before on several), the
// thresholds, and the decision not to iterate at all when the initial residual is below the absolute tolerance.
__global__ __launch_bounds__(1024) void k_pcg_setup(int nb, const double* __restrict__ partials, double* __restrict__ S,
                                                    double rtol2, double atol2, int32_t* __restrict__ flags) {
  __shared__ double lds[1024 / 64];
  double rho0, bb;
  if (nb > 0) {
    double a = 0.0, b = 0.0;
    for (int i = threadIdx.x; i < nb; i += 1024) { a += partials[i]; b += partials[FEMO_MAX_PARTIALS + i]; }
    rho0 = femo_block_sum_all<1024>(a, lds);
    bb = femo_block_sum_all<1024>(b, lds);
  } else {
    rho0 = S[MS_RED]; bb = S[MS_RED + 1];
  }
  if (threadIdx.x != 0) return;
  S[MS_GAMMA] = 0.0; S[MS_GAMMA + 1] = 0.0; S[MS_PQ] = 0.0; S[MS_ALPHA] = 0.0; S[MS_DOTC] = 0.0; S[MS_TOLG] = 0.0;
  S[MS_TOL2] = atol2;
  S[MS_RR] = rho0;
  S[MS_BB] = bb;
  S[MS_FACTOR] = rtol2 * (rho0 > 0.0 ? bb / rho0 : 1.0);
  flags[1] = 0; flags[2] = 0; flags[3] = 0;
  const bool bad = !(rho0 == rho0);
  if (!(rho0 > atol2) || rho0 == 0.0) {          // nothing to iterate on (or NaN): every later launch returns at once
    flags[2] = bad ? 1 : 0;
    __threadfence();
    flags[0] = 1;
  } else {
    flags[0] = 0;
  }
}

static int solve_pcg_bpx_merged(femo_mat* A, const femo_vec* b, femo_vec* x, const femo_solver_opts* opts, femo_solve_info* info) {
  femo_mesh* m = A->mesh;
  femo_ctx* ctx = m->ctx;
  const int64_t n = m->n_rows;
  hipStream_t st = ctx->stream;
  FEMO_HIP_CHECK(hipEventRecord(ctx->ev0, st));
  FEMO_TRY(ensure_s(A));
  CgWork w;
  const bool multi = ctx->nranks > 1;
  // k_cg_init writes every owned entry of r, p and x^; only the ghost tails have to be defined (zero) -- they and the ghost
  // tail of the solution are cleared by the preconditioner's own clearing launch (round 6; round 5: three memsets)
  FEMO_TRY(ensure_work(ctx, n, m->n_vert, w, 0, /*zero=*/false));
  const int gv = vec_grid(ctx, n);
  const int gs = femo_spmv_grid(m);
  int32_t* h_flags = reinterpret_cast<int32_t*>(ctx->h_scal + FEMO_NSCAL);
  double* P = ctx->d_partials;
  double* S = ctx->d_scal;
  const uint8_t* mask = A->pc_has_mask ? A->d_pcmask : nullptr;
  FemoZeroExtra ze;
  ze.count = 0;
  if (m->n_vert > n) {
    const int64_t tail = ctx->cg_n - n;                          // (ensure_work: vectors of max(n_rows, n_vert) + 2 entries)
    ze.p[ze.count] = w.p + n; ze.n[ze.count++] = tail;
    ze.p[ze.count] = w.r + n; ze.n[ze.count++] = tail;
  }
  if (opts->zero_guess && x->n > n) { ze.p[ze.count] = x->d + n; ze.n[ze.count++] = x->n - n; }   // ghost tail of the solution, as the classic loop leaves it
  FEMO_TRY(femo_pc_merged_begin(m, A->d_s, mask, &ze));
  const double* q0 = nullptr;
  if (!opts->zero_guess) {
    if (m->n_nbr > 0) FEMO_TRY(halo_raw(m, x->d));
    FEMO_TRY(launch_spmv(A, A->d_vals, x->d, w.q, nullptr, nullptr));
    q0 = w.q;
  }
  hipLaunchKernelGGL(k_cg_init, dim3(gv), dim3(FEMO_BLOCK), 0, st, n, b->d, q0, A->d_s, w.r, w.p, w.xh, P, A->has_idrows ? A->d_idrows : nullptr);
  const double atol2 = opts->atol * opts->atol;
  if (multi) {
    hipLaunchKernelGGL(k_reduce_partials_at, dim3(1), dim3(1024), 0, st, gv, 2, P + FEMO_MAX_PARTIALS, S + MS_RED);
    FEMO_TRY(femo_coll_allreduce(ctx, S + MS_RED, 2, st));
    hipLaunchKernelGGL(k_pcg_setup, dim3(1), dim3(1024), 0, st, 0, P + FEMO_MAX_PARTIALS, S, opts->rtol * opts->rtol, atol2, ctx->d_flags);
  } else {
    hipLaunchKernelGGL(k_pcg_setup, dim3(1), dim3(1024), 0, st, gv, P + FEMO_MAX_PARTIALS, S, opts->rtol * opts->rtol, atol2, ctx->d_flags);
  }
  FemoPcgStop stop;
  stop.rtol2_factor = 0.0;                       // read from S[MS_FACTOR] on the device
  stop.atol_pc2 = opts->atol_pc * opts->atol_pc;
  stop.tolg2 = S + MS_TOLG;
  stop.flags = ctx->d_flags;
  stop.it = -1;
  FemoMergedVecs V;
  V.x = w.xh; V.r = w.r; V.p = w.p; V.q = nullptr; V.n = n; V.cur = 0; V.gv = gv; V.atol2 = 0.0;
  V.nb_q[0] = V.nb_q[1] = 0; V.Pq[0] = V.Pq[1] = nullptr;
  FEMO_TRY(femo_pc_merged_apply(m, mask, A->pc_key, V, S, ctx->d_flags, &stop));
  FEMO_HIP_CHECK(hipGetLastError());
  // The one synchronisation before the loop: rho0, bb, gamma0 and the flag.  x of a solve that does not iterate (the
  // identity rows, solved by k_cg_init) is written speculatively in front of it -- 6 us that save such a solve (Newton's
  // later passes) a second stream drain; an iterating solve overwrites it at the end.
  const bool spec_x = n > 0 && opts->zero_guess;
  if (spec_x) hipLaunchKernelGGL(k_unscale_done, dim3(2048), dim3(256), 0, st, n, 0, A->d_s, w.xh, x->d, ctx->d_flags);   // (only if the flag says so: an iterating solve skips the 44 us at C4)
  FEMO_HIP_CHECK(hipMemcpyAsync(h_flags, ctx->d_flags, 4 * sizeof(int32_t), hipMemcpyDeviceToHost, st));
  FEMO_HIP_CHECK(hipMemcpyAsync(ctx->h_scal, S, FEMO_NSCAL * sizeof(double), hipMemcpyDeviceToHost, st));
  FEMO_HIP_CHECK(hipEventRecord(ctx->ev1, st));
  FEMO_HIP_CHECK(hipEventSynchronize(ctx->ev1));
  const double rho0 = ctx->h_scal[MS_RR], bb = ctx->h_scal[MS_BB], gamma0 = ctx->h_scal[MS_GAMMA];
  info->rhs_norm = std::sqrt(bb);
  const int max_it = opts->max_it > 0 ? opts->max_it : 10000;
  info->spmv_ms = 0.0; info->spmv_samples = 0;
  info->pc_rhs_norm = std::sqrt(gamma0 * (rho0 > 0.0 ? bb / rho0 : 1.0));
  if (h_flags[0]) {                               // below the absolute tolerance (k_pcg_setup) or below atol_pc (first apply)
    info->pc_residual_norm = std::sqrt(gamma0);
    if (!spec_x && n > 0) {
      hipLaunchKernelGGL(k_unscale, dim3(2048), dim3(256), 0, st, n, 1, A->d_s, w.xh, x->d);
      FEMO_HIP_CHECK(hipEventRecord(ctx->ev1, st));
      FEMO_HIP_CHECK(hipEventSynchronize(ctx->ev1));
    }
    float ms0 = 0.f;
    FEMO_HIP_CHECK(hipEventElapsedTime(&ms0, ctx->ev0, ctx->ev1));
    info->iterations = 0;
    info->loop_allreduces = 0;
    info->converged = (h_flags[2] || !(rho0 == rho0)) ? -1 : 1;
    info->residual_norm = std::sqrt(rho0);
    info->solve_ms = ms0;
    return 0;
  }
  FEMO_TRY(ensure_scaled(A, false));

  const int n_sample = 4, sample_from = 2;
  int n_ev = 0;
  // Batches: with a prediction (the smaller of the last two counts on this mesh) exactly that many iterations first, then
  // 2, 3, 4, 6, 8, 8 ... -- each batch polled synchronously (a solve that follows a short one paid a
  // blocking poll every second iteration).  Without a prediction: 8 at a time.
  const int batch = opts->check_every > 0 ? std::min(opts->check_every, 8) : 8;
  // Round 5: the prediction follows the REDUCTION this solve needs, ln(gamma_0 / threshold), at the rate the last solve on
  // this mesh converged with (ln of its reduction per iteration).  The smaller of the last two counts mispredicted whenever a
  // short solve (Newton's second pass: one iteration) sat between two long ones: the long solve after it was polled after
  // 1, 3, 6, 10, 16 ... iterations, ~35 us of idle device per poll.
  int predicted = 0;
  {
    const double tolg = ctx->h_scal[MS_TOLG];
    if (m->pcg_rate > 0.0 && gamma0 > 0.0 && tolg > 0.0 && gamma0 > tolg) {
      const double need = std::log(gamma0 / tolg);
      predicted = (int)std::ceil(need / m->pcg_rate) + 1;
      predicted = std::max(1, std::min(predicted, 4 * std::max(m->pcg_last_iters, 8)));
    } else if (m->pcg_rate <= 0.0) {
      const int last2 = std::min(m->pcg_last_iters, m->pcg_prev_iters);
      predicted = last2 > 0 ? last2 : 0;
    }
  }
  int it = 0, grow = 2;
  bool done = false;
  const int64_t ar0 = ctx->n_allreduce;
  double* Pq_int = P + 4 * FEMO_MAX_PARTIALS;     // triples [p.q | q.q | r.q]: interior / only launch, boundary launch
  double* Pq_bnd = P + 7 * FEMO_MAX_PARTIALS;
  V.q = w.q; V.atol2 = atol2;
  while (!done) {
    int this_batch = batch;
    if (predicted > 0) {
      if (it == 0) this_batch = predicted;
      else { this_batch = grow; grow = std::min(8, grow + (grow + 1) / 2); }
    }
    const int it_end = it + this_batch < max_it ? it + this_batch : max_it;
    for (; it < it_end; ++it) {
      const bool sample = it >= sample_from && it < sample_from + n_sample;
      if (sample) FEMO_HIP_CHECK(hipEventRecord(ctx->ev_pool[2 * n_ev], st));
      int g1 = gs, g2 = 0;
      if (multi) {
        if (femo_pc_merged_sends_halo(m)) {
          FEMO_TRY(halo_spmv_inflight(A, A->d_valsS, w.p, w.q, Pq_int, ctx->d_flags, true, &g1, &g2, w.r, 3));
        } else if (m->n_nbr > 0 && m->d_slices_int != nullptr) {
          FEMO_TRY(halo_spmv_overlapped(A, A->d_valsS, w.p, w.q, Pq_int, ctx->d_flags, true, false, &g1, &g2, w.r, 3));
        } else {
          if (m->n_nbr > 0) FEMO_TRY(halo_raw(m, w.p));
          FEMO_TRY(launch_spmv(A, A->d_valsS, w.p, w.q, Pq_int, ctx->d_flags, true, false, nullptr, 0, nullptr, w.r, false, true));
        }
      } else {
        FEMO_TRY(launch_spmv(A, A->d_valsS, w.p, w.q, Pq_int, ctx->d_flags, true));
      }
      if (sample) { FEMO_HIP_CHECK(hipEventRecord(ctx->ev_pool[2 * n_ev + 1], st)); ++n_ev; }
      stop.it = it;
      V.cur = it & 1;
      V.nb_q[0] = g1; V.nb_q[1] = g2; V.Pq[0] = Pq_int; V.Pq[1] = Pq_bnd;
      FEMO_TRY(femo_pc_merged_apply(m, mask, A->pc_key, V, S, ctx->d_flags, &stop));
    }
    FEMO_HIP_CHECK(hipGetLastError());
    // the poll carries everything the end of a converged solve needs: x (unscaled on the device only if the flag is set),
    // the flags and the scalars -- the batch that converges costs ONE host round trip, not two (round 6)
    if (n > 0) hipLaunchKernelGGL(k_unscale_done, dim3(2048), dim3(256), 0, st, n, opts->zero_guess ? 0 : 1, A->d_s, w.xh, x->d, ctx->d_flags);
    // a consumer of the device-initiated ghost refresh that gave up waiting (4 s) must not pass for a result: its flag
    // travels in the spare word of the poll
    if (femo_halo_direct_ready(m)) FEMO_HIP_CHECK(hipMemcpyAsync(ctx->d_flags + 3, m->hd->d_err, sizeof(int32_t), hipMemcpyDeviceToDevice, st));
    FEMO_HIP_CHECK(hipMemcpyAsync(h_flags, ctx->d_flags, 4 * sizeof(int32_t), hipMemcpyDeviceToHost, st));
    FEMO_HIP_CHECK(hipMemcpyAsync(ctx->h_scal, S, FEMO_NSCAL * sizeof(double), hipMemcpyDeviceToHost, st));
    FEMO_HIP_CHECK(hipEventRecord(ctx->ev_pool[2 * n_sample], st));
    FEMO_HIP_CHECK(hipEventSynchronize(ctx->ev_pool[2 * n_sample]));
    FEMO_REQUIRE(!(femo_halo_direct_ready(m) && h_flags[3] != 0),
                 "ghost refresh: a consumer waited 4 s for a neighbour's stores and gave up (rank %d of %d) -- the solve is void", ctx->rank, ctx->nranks);
    if (h_flags[0] || it >= max_it) done = true;
  }
  info->loop_allreduces = (int32_t)(ctx->n_allreduce - ar0);
  float ms = 0.f;
  if (h_flags[0]) {
    FEMO_HIP_CHECK(hipEventElapsedTime(&ms, ctx->ev0, ctx->ev_pool[2 * n_sample]));
  } else {
    // not converged within max_it: x of the last iterate, the final scalars and the time with one more synchronisation
    if (n > 0) hipLaunchKernelGGL(k_unscale, dim3(2048), dim3(256), 0, st, n, opts->zero_guess ? 0 : 1, A->d_s, w.xh, x->d);
    FEMO_HIP_CHECK(hipMemcpyAsync(h_flags, ctx->d_flags, 4 * sizeof(int32_t), hipMemcpyDeviceToHost, st));
    FEMO_HIP_CHECK(hipMemcpyAsync(ctx->h_scal, S, FEMO_NSCAL * sizeof(double), hipMemcpyDeviceToHost, st));
    FEMO_HIP_CHECK(hipEventRecord(ctx->ev1, st));
    FEMO_HIP_CHECK(hipEventSynchronize(ctx->ev1));
    FEMO_HIP_CHECK(hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
  }
  const int iters = h_flags[1];
  const int conv = h_flags[0] ? (h_flags[2] ? -1 : 1) : 0;
  if (conv == 1 && iters > 0) { m->pcg_prev_iters = m->pcg_last_iters; m->pcg_last_iters = iters; }
  if (conv == 1 && iters >= 4) {
    const double g_end = ctx->h_scal[MS_GAMMA + (iters & 1)];
    if (g_end > 0.0 && gamma0 > g_end) m->pcg_rate = std::log(gamma0 / g_end) / iters;
  }
  double acc = 0.0;
  for (int i = 0; i < n_ev; ++i) {
    float t = 0.f;
    FEMO_HIP_CHECK(hipEventElapsedTime(&t, ctx->ev_pool[2 * i], ctx->ev_pool[2 * i + 1]));
    acc += t;
  }
  info->spmv_ms = acc;
  info->spmv_samples = n_ev;
  info->pc_residual_norm = std::sqrt(ctx->h_scal[MS_GAMMA + (iters & 1)]);
  info->iterations = iters;
  info->converged = conv;
  info->residual_norm = std::sqrt(ctx->h_scal[MS_RR]);
  info->solve_ms = ms;
  return 0;
}

extern "C" int femo_mat_prescale(femo_mat* A) {
  FEMO_REQUIRE(A != nullptr, "null argument");
  return ensure_scaled(A, false);
}

// CG with the auxiliary-lattice BPX preconditioner.  Same scaled system, same stopping norm
// (sqrt(rh.rh) = sqrt(r^T D^-1 r)) as the Jacobi path; scalars live on the device and are
// all-reduced there when the mesh is partitioned.
static int solve_pcg_bpx(femo_mat* A, int transpose, const femo_vec* b, femo_vec* x,
                         const femo_solver_opts* opts, femo_solve_info* info) {
  femo_mesh* m = A->mesh;
  femo_ctx* ctx = m->ctx;
  const int64_t n = m->n_rows;
  FEMO_REQUIRE(A->bpx_ok, "the BPX preconditioner needs an operator assembled from a Poisson-type form");
  FEMO_REQUIRE(!transpose, "the BPX preconditioner is for symmetric operators");
  hipStream_t st = ctx->stream;
  if (!(ctx->comm != nullptr && ctx->nranks == 1 && femo_env_flag("FEMO_FORCE_MULTI")) && femo_pc_merged_ok(m))
    return solve_pcg_bpx_merged(A, b, x, opts, info);
  FEMO_HIP_CHECK(hipEventRecord(ctx->ev0, st));
  FEMO_TRY(ensure_s(A));
  FEMO_TRY(femo_pc_build(m));
  CgWork w;
  const bool multi = ctx->nranks > 1 || (ctx->comm != nullptr && femo_env_flag("FEMO_FORCE_MULTI"));
  FEMO_TRY(ensure_work(ctx, n, m->n_vert, w, 1));
  const int gv = vec_grid(ctx, n);
  const int gs = femo_spmv_grid(m);
  int32_t* h_flags = reinterpret_cast<int32_t*>(ctx->h_scal + FEMO_NSCAL);
  double* P = ctx->d_partials;
  double* S = ctx->d_scal;
  const uint8_t* mask = A->pc_has_mask ? A->d_pcmask : nullptr;
  FEMO_TRY(femo_pc_begin(m, A->d_s, mask));
  auto allreduce1 = [&](double* d) -> int {
    if (multi) FEMO_TRY(femo_coll_allreduce(ctx, d, 1, st));
    return 0;
  };

  FEMO_HIP_CHECK(hipMemsetAsync(ctx->d_flags, 0, 4 * sizeof(int32_t), st));
  const double* q0 = nullptr;
  if (opts->zero_guess) {
    FEMO_HIP_CHECK(hipMemsetAsync(x->d, 0, x->n * sizeof(double), st));
  } else {
    if (m->n_nbr > 0) FEMO_TRY(halo_raw(m, x->d));
    FEMO_TRY(launch_spmv(A, A->d_vals, x->d, w.q, nullptr, nullptr));
    q0 = w.q;
  }
  hipLaunchKernelGGL(k_cg_init, dim3(gv), dim3(FEMO_BLOCK), 0, st, n, b->d, q0, A->d_s, w.r, w.p, w.xh, P, A->has_idrows ? A->d_idrows : nullptr);
  hipLaunchKernelGGL(k_reduce_partials_at, dim3(1), dim3(1024), 0, st, gv, 2, P + FEMO_MAX_PARTIALS, S);
  FEMO_HIP_CHECK(hipGetLastError());
  if (multi) FEMO_TRY(femo_coll_allreduce(ctx, S, 2, st));
  FEMO_HIP_CHECK(hipMemcpyAsync(ctx->h_scal, S, 2 * sizeof(double), hipMemcpyDeviceToHost, st));
  FEMO_HIP_CHECK(hipStreamSynchronize(st));
  const double rho0 = ctx->h_scal[0], bb = ctx->h_scal[1];
  // Stopping rule: relative in the norm of the preconditioner, sqrt(r^T M^-1 r) <= rtol sqrt(b^T M^-1 b)
  // (tested inside femo_pc_apply, where gamma = r^T M^-1 r becomes known), absolute in the Jacobi norm,
  // sqrt(r^T D^-1 r) <= atol (Newton's rounding-floor rule is stated in that norm; tested by k_pcg_check).
  const double bnorm = std::sqrt(bb);
  const double tol = opts->atol;
  info->rhs_norm = bnorm;
  const int max_it = opts->max_it > 0 ? opts->max_it : 10000;
  auto finish = [&](int iters, int conv, double rho) -> int {
    if (n > 0) {
      hipLaunchKernelGGL(k_unscale, dim3(2048), dim3(256), 0, st, n, opts->zero_guess ? 0 : 1, A->d_s, w.xh, x->d);
      FEMO_HIP_CHECK(hipGetLastError());
    }
    FEMO_HIP_CHECK(hipEventRecord(ctx->ev1, st));
    FEMO_HIP_CHECK(hipEventSynchronize(ctx->ev1));
    float ms = 0.f;
    FEMO_HIP_CHECK(hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
    info->iterations = iters;
    info->converged = conv;
    info->residual_norm = std::sqrt(rho);
    info->solve_ms = ms;
    return 0;
  };
  if (!(std::sqrt(rho0) > tol) || rho0 == 0.0) return finish(0, rho0 == rho0 ? 1 : -1, rho0);

  double hs[FEMO_NSCAL] = {0};
  hs[S_TOL2] = tol * tol;
  hs[S_RHO] = rho0;
  memcpy(ctx->h_scal, hs, sizeof hs);
  FEMO_HIP_CHECK(hipMemcpyAsync(S, ctx->h_scal, sizeof hs, hipMemcpyHostToDevice, st));
  FEMO_HIP_CHECK(hipStreamSynchronize(st));
  // ph0 = zh0 = M^-1 rh0, gamma0 = rh0.zh0 = rho0 + g_L.e_L (lattice dot: global on every rank, no all-reduce);
  // the same launch stores the threshold max(rtol^2 gamma_0 bb/rho0, atol_pc^2) (relative to the right-hand
  // side, not to r0) and tests gamma_0 against it
  FemoPcgStop stop;
  stop.rtol2_factor = opts->rtol * opts->rtol * (rho0 > 0.0 ? bb / rho0 : 1.0);
  stop.atol_pc2 = opts->atol_pc * opts->atol_pc;
  stop.tolg2 = S + S_TOLG;
  stop.flags = ctx->d_flags;
  stop.it = -1;
  FEMO_TRY(femo_pc_apply(m, mask, A->pc_key, A->d_s, w.r, w.p, 2, S + S_RHO, S + S_GAMMA + 1, S + S_GAMMA, ctx->d_flags, gv, false, &stop));
  FEMO_HIP_CHECK(hipMemcpyAsync(h_flags, ctx->d_flags, 4 * sizeof(int32_t), hipMemcpyDeviceToHost, st));
  FEMO_HIP_CHECK(hipMemcpyAsync(ctx->h_scal, S, FEMO_NSCAL * sizeof(double), hipMemcpyDeviceToHost, st));
  FEMO_HIP_CHECK(hipStreamSynchronize(st));
  const double gamma0 = ctx->h_scal[S_GAMMA];
  info->pc_rhs_norm = std::sqrt(gamma0 * (rho0 > 0.0 ? bb / rho0 : 1.0));
  info->pc_residual_norm = std::sqrt(gamma0);
  if (h_flags[0]) return finish(0, gamma0 == gamma0 ? 1 : -1, rho0);      // below atol_pc before the first iteration
  FEMO_TRY(ensure_scaled(A, false));   // the iteration's operator S A S (the apply above only needed S)

  const int n_sample = 4, sample_from = 2;
  int n_ev = 0;
  const bool local_scalars = !multi && m->n_nbr == 0;
  const bool piggyback = multi && femo_pc_can_piggyback(m);
  // x += alpha p inside the preconditioner's single-workgroup launch (the first apply, above, has settled whether the
  // fused lattice cycle runs on this mesh)
  const bool carry_x = local_scalars && femo_pc_carries_xupdate(m);
  const bool use_atol = opts->atol > 0.0;
  // Iterations are enqueued in batches and the host polls the "done" stamp; launches behind the converged iteration
  // return at once but still cost ~5 us each (6 per iteration).  Round 1: fixed batches of 8, polled two deep -- a
  // solve that converges at iteration 28 enqueued 40 (12 dead iterations).  Round 2: first batch = the count of
  // earlier solves on this mesh + 1, then batches of 3, still two deep: 4 dead iterations per solve (12 % of the
  // SpMV launches in the kernel trace were early exits).  Round 3: with a prediction the first batch is exactly the
  // earlier count (BPX counts repeat from solve to solve: 28, 28, 28 ...) and the host waits for its stamp before it
  // enqueues anything else; beyond the prediction two iterations at a time (a host round trip of ~30 us per pair,
  // at most one dead iteration).  Without a prediction: batches of 8, two deep, as before.
  const int batch = opts->check_every > 0 ? std::min(opts->check_every, 8) : 8;
  double* r_cur = w.r;
  // (the smaller of the last two counts: Newton's later solves need far fewer iterations than its first)
  const int last2 = std::min(m->pcg_last_iters, m->pcg_prev_iters);
  const int predicted = last2 > 0 ? last2 : 0;
  int it = 0, polled = 0;
  bool done = false;
  int pending[2] = {-1, -1};
  const int64_t ar0 = ctx->n_allreduce;
  while (!done) {
    const int this_batch = predicted > 0 ? (it == 0 ? predicted : 2) : batch;
    const int it_end = it + this_batch < max_it ? it + this_batch : max_it;
    for (; it < it_end; ++it) {
      const int cur = it & 1, nxt = cur ^ 1;
      const bool sample = it >= sample_from && it < sample_from + n_sample;
      if (sample) FEMO_HIP_CHECK(hipEventRecord(ctx->ev_pool[2 * n_ev], st));
      int g1 = gs, g2 = 0;
      if (m->n_nbr > 0 && m->d_slices_int != nullptr) {
        FEMO_TRY(halo_spmv_overlapped(A, A->d_valsS, w.p, w.q, P, ctx->d_flags, true, false, &g1, &g2));
      } else {
        if (m->n_nbr > 0) FEMO_TRY(halo_raw(m, w.p));
        FEMO_TRY(launch_spmv(A, A->d_valsS, w.p, w.q, P + P_DELTA * FEMO_MAX_PARTIALS, ctx->d_flags, true));
      }
      if (sample) { FEMO_HIP_CHECK(hipEventRecord(ctx->ev_pool[2 * n_ev + 1], st)); ++n_ev; }
      double* Pd = P + P_DELTA * FEMO_MAX_PARTIALS;
      double* Pr = P + 1 * FEMO_MAX_PARTIALS;
      double* Pg = P + 2 * FEMO_MAX_PARTIALS;   // boundary-slice partials of the overlapped SpMV
      stop.it = it;
      if (local_scalars) {
        // single GPU: the consumers fold the per-block partials themselves
        // ph = M^-1 rh + beta ph in one pass: rh.zh = rho + g_L.e_L is known before the mesh prolongation, and
        // with it the stopping test.  The Jacobi-norm test runs only when an absolute tolerance is set.
        if (use_atol) {
          hipLaunchKernelGGL(k_pcg_xr, dim3(gv), dim3(FEMO_BLOCK), 0, st, n, cur, g1, Pd, S, w.q, w.p, (const double*)r_cur, r_cur, w.xh, Pr, ctx->d_flags);
          hipLaunchKernelGGL(k_pcg_check, dim3(1), dim3(1024), 0, st, it, gv, Pr, S, ctx->d_flags);
          FEMO_TRY(femo_pc_apply(m, mask, A->pc_key, A->d_s, r_cur, w.p, 1, S + S_RHO, S + S_GAMMA + cur, S + S_GAMMA + nxt, ctx->d_flags, gv, false, &stop));
        } else if (carry_x) {
          hipLaunchKernelGGL(k_pcg_xr, dim3(gv), dim3(FEMO_BLOCK), 0, st, n, cur, g1, Pd, S, w.q, w.p, (const double*)r_cur, r_cur, w.xh, Pr, ctx->d_flags, 1);
          const FemoXUpdate xu = {w.xh, w.p, S + S_ALPHA, n};
          FEMO_TRY(femo_pc_apply(m, mask, A->pc_key, A->d_s, r_cur, w.p, 1, S + S_RHO, S + S_GAMMA + cur, S + S_GAMMA + nxt, ctx->d_flags, gv, false, &stop, gv, Pr, &xu));
        } else {
          hipLaunchKernelGGL(k_pcg_xr, dim3(gv), dim3(FEMO_BLOCK), 0, st, n, cur, g1, Pd, S, w.q, w.p, (const double*)r_cur, r_cur, w.xh, Pr, ctx->d_flags);
          FEMO_TRY(femo_pc_apply(m, mask, A->pc_key, A->d_s, r_cur, w.p, 1, S + S_RHO, S + S_GAMMA + cur, S + S_GAMMA + nxt, ctx->d_flags, gv, false, &stop, gv, Pr));
        }
      } else {
        hipLaunchKernelGGL(k_pcg_fold, dim3(1), dim3(1024), 0, st, g1, Pd, g2, Pg, S + S_DELTA, ctx->d_flags);
        FEMO_TRY(allreduce1(S + S_DELTA));
        hipLaunchKernelGGL(k_pcg_xr, dim3(gv), dim3(FEMO_BLOCK), 0, st, n, cur, 0, Pd, S, w.q, w.p, (const double*)w.r, w.r, w.xh, Pr, ctx->d_flags);
        hipLaunchKernelGGL(k_pcg_fold, dim3(1), dim3(1024), 0, st, gv, Pr, 0, (const double*)nullptr, S + S_RHO, ctx->d_flags);
        if (piggyback) {
          // the rank's part of rh.rh rides in the lattice all-reduce of the preconditioner; the stopping
          // test then runs after the apply (one wasted apply in the last iteration, one collective less in all)
          FEMO_TRY(femo_pc_apply(m, mask, A->pc_key, A->d_s, w.r, w.p, 1, S + S_RHO, S + S_GAMMA + cur, S + S_GAMMA + nxt, ctx->d_flags, gv, true, &stop));
          if (use_atol) hipLaunchKernelGGL(k_pcg_check, dim3(1), dim3(1024), 0, st, it, 0, Pr, S, ctx->d_flags);
        } else {
          FEMO_TRY(allreduce1(S + S_RHO));
          if (use_atol) hipLaunchKernelGGL(k_pcg_check, dim3(1), dim3(1024), 0, st, it, 0, Pr, S, ctx->d_flags);
          FEMO_TRY(femo_pc_apply(m, mask, A->pc_key, A->d_s, w.r, w.p, 1, S + S_RHO, S + S_GAMMA + cur, S + S_GAMMA + nxt, ctx->d_flags, gv, false, &stop));
        }
      }
    }
    FEMO_HIP_CHECK(hipGetLastError());
    const int slot = polled & 1;
    FEMO_HIP_CHECK(hipMemcpyAsync(h_flags + 4 * slot, ctx->d_flags, 4 * sizeof(int32_t), hipMemcpyDeviceToHost, st));
    FEMO_HIP_CHECK(hipEventRecord(ctx->ev_pool[2 * n_sample + slot], st));
    pending[slot] = it;
    ++polled;
    const int prev = polled & 1;
    const bool last = it >= max_it;
    if (predicted > 0) {                                  // synchronous poll of the batch just enqueued
      FEMO_HIP_CHECK(hipEventSynchronize(ctx->ev_pool[2 * n_sample + slot]));
      if (h_flags[4 * slot] || last) done = true;
      pending[slot] = -1;
      continue;
    }
    if (pending[prev] >= 0) {
      FEMO_HIP_CHECK(hipEventSynchronize(ctx->ev_pool[2 * n_sample + prev]));
      if (h_flags[4 * prev]) done = true;
      pending[prev] = -1;
    }
    if (!done && last) {
      FEMO_HIP_CHECK(hipEventSynchronize(ctx->ev_pool[2 * n_sample + slot]));
      done = true;
    }
  }
  info->loop_allreduces = (int32_t)(ctx->n_allreduce - ar0);
  FEMO_HIP_CHECK(hipMemcpyAsync(h_flags, ctx->d_flags, 4 * sizeof(int32_t), hipMemcpyDeviceToHost, st));
  FEMO_HIP_CHECK(hipMemcpyAsync(ctx->h_scal, S, FEMO_NSCAL * sizeof(double), hipMemcpyDeviceToHost, st));
  FEMO_HIP_CHECK(hipStreamSynchronize(st));
  const int iters = h_flags[1];
  const int conv = h_flags[0] ? (h_flags[2] ? -1 : 1) : 0;
  if (conv == 1 && iters > 0) { m->pcg_prev_iters = m->pcg_last_iters; m->pcg_last_iters = iters; }
  double acc = 0.0;
  for (int i = 0; i < n_ev; ++i) {
    float t = 0.f;
    FEMO_HIP_CHECK(hipEventElapsedTime(&t, ctx->ev_pool[2 * i], ctx->ev_pool[2 * i + 1]));
    acc += t;
  }
  info->spmv_ms = acc;
  info->spmv_samples = n_ev;
  info->pc_residual_norm = std::sqrt(ctx->h_scal[S_GAMMA + (iters & 1)]);
  return finish(iters, conv, ctx->h_scal[S_RHO]);
}

// z = M^-1 r with the BPX preconditioner of A, in unscaled variables:
//   M^-1 = D^-1 + theta sum_l P_l C_l P_l^T   (oracle/bpx_oracle.py restates it; tests compare)
extern "C" int femo_mat_pc_apply(const femo_mat* A_, const femo_vec* r, femo_vec* z) {
  FEMO_REQUIRE(A_ && r && z, "null argument");
  femo_mat* A = const_cast<femo_mat*>(A_);
  femo_mesh* m = A->mesh;
  femo_ctx* ctx = m->ctx;
  const int64_t n = m->n_rows;
  FEMO_REQUIRE(A->bpx_ok, "the BPX preconditioner needs an operator assembled from a Poisson-type form");
  FEMO_REQUIRE(r->n >= n && z->n >= n && r->d != z->d, "vector size mismatch in pc_apply");
  femo_vec_touch(z);
  hipStream_t st = ctx->stream;
  FEMO_TRY(ensure_s(A));
  FEMO_TRY(femo_pc_build(m));
  CgWork w;
  FEMO_TRY(ensure_work(ctx, n, m->n_vert, w, 1));
  const uint8_t* mask = A->pc_has_mask ? A->d_pcmask : nullptr;
  FEMO_TRY(femo_pc_begin(m, A->d_s, mask));
  const int gv = vec_grid(ctx, n);
  FEMO_HIP_CHECK(hipMemsetAsync(ctx->d_flags, 0, 4 * sizeof(int32_t), st));
  if (n == 0) return 0;
  // rh = S r, zh = Mh^-1 rh, z = S zh
  hipLaunchKernelGGL(k_cg_init, dim3(gv), dim3(FEMO_BLOCK), 0, st, n, r->d, (const double*)nullptr, A->d_s, w.r, w.p, w.xh, ctx->d_partials, (const uint8_t*)nullptr);
  FEMO_TRY(femo_pc_apply(m, mask, A->pc_key, A->d_s, w.r, w.sv, 0, nullptr, nullptr, nullptr, ctx->d_flags, gv));
  hipLaunchKernelGGL(k_unscale, dim3(2048), dim3(256), 0, st, n, 0, A->d_s, w.sv, z->d);
  FEMO_HIP_CHECK(hipGetLastError());
  return 0;
}

extern "C" int femo_solve_cg(const femo_mat* A_, int transpose, const femo_vec* b, femo_vec* x,
                             const femo_solver_opts* opts, femo_solve_info* info) {
  FEMO_REQUIRE(A_ && b && x && opts && info, "null argument");
  femo_mat* A = const_cast<femo_mat*>(A_);
  femo_mesh* m = A->mesh;
  femo_ctx* ctx = m->ctx;
  const int64_t n = m->n_rows;
  FEMO_REQUIRE(b->n >= n && x->n >= m->n_vert, "vector size mismatch in solve_cg");
  FEMO_REQUIRE(b->d != x->d, "solve_cg cannot run in place");
  femo_vec_touch(x);
  memset(info, 0, sizeof *info);
  FEMO_REQUIRE(opts->pc == FEMO_PC_JACOBI || opts->pc == FEMO_PC_BPX, "unknown preconditioner %d", opts->pc);
  if (opts->pc == FEMO_PC_BPX) return solve_pcg_bpx(A, transpose, b, x, opts, info);
  hipStream_t st = ctx->stream;
  FEMO_HIP_CHECK(hipEventRecord(ctx->ev0, st));
  FEMO_TRY(ensure_s(A));
  if (transpose && !opts->zero_guess) FEMO_TRY(femo_mat_ensure_transpose(A));
  CgWork w;
  // FEMO_FORCE_MULTI=1 runs the all-reduce code path on a 1-rank communicator (tests)
  const bool multi = ctx->nranks > 1 || (ctx->comm != nullptr && femo_env_flag("FEMO_FORCE_MULTI"));
  FEMO_TRY(ensure_work(ctx, n, m->n_vert, w, multi ? 1 : 0));
  const int gv = vec_grid(ctx, n);
  const int gs = femo_spmv_grid(m);
  int32_t* h_flags = reinterpret_cast<int32_t*>(ctx->h_scal + FEMO_NSCAL);  // 2 slots x 4 ints
  double* P = ctx->d_partials;

  FEMO_HIP_CHECK(hipMemsetAsync(ctx->d_flags, 0, 4 * sizeof(int32_t), st));
  // r0 = b - A x0 with the unscaled operator; the iteration then solves for the correction
  const double* q0 = nullptr;
  if (opts->zero_guess) {
    FEMO_HIP_CHECK(hipMemsetAsync(x->d, 0, x->n * sizeof(double), st));
  } else {
    if (m->n_nbr > 0) FEMO_TRY(halo_raw(m, x->d));
    const double* vals = A->d_vals;
    if (transpose) vals = A->d_valsT;
    FEMO_TRY(launch_spmv(A, vals, x->d, w.q, nullptr, nullptr));
    q0 = w.q;
  }
  hipLaunchKernelGGL(k_cg_init, dim3(gv), dim3(FEMO_BLOCK), 0, st, n, b->d, q0, A->d_s, w.r, w.p, w.xh, P, A->has_idrows ? A->d_idrows : nullptr);
  FEMO_HIP_CHECK(hipGetLastError());
  // gamma0 and ||S b||^2 (all-reduced when multi): partial slots 1 and 2
  hipLaunchKernelGGL(k_reduce_partials_at, dim3(1), dim3(1024), 0, st, gv, 2, P + FEMO_MAX_PARTIALS, ctx->d_scal);
  FEMO_HIP_CHECK(hipGetLastError());
  if (multi) FEMO_TRY(femo_coll_allreduce(ctx, ctx->d_scal, 2, st));
  FEMO_HIP_CHECK(hipMemcpyAsync(ctx->h_scal, ctx->d_scal, 2 * sizeof(double), hipMemcpyDeviceToHost, st));
  FEMO_HIP_CHECK(hipStreamSynchronize(st));
  const double gamma0 = ctx->h_scal[0], bb = ctx->h_scal[1];
  const double bnorm = std::sqrt(bb);
  double tol = opts->rtol * bnorm;
  if (opts->atol > tol) tol = opts->atol;
  info->rhs_norm = bnorm;
  const int max_it = opts->max_it > 0 ? opts->max_it : 10000;
  auto finish = [&](int iters, int conv, double gamma) -> int {
    if (n > 0) {
      hipLaunchKernelGGL(k_unscale, dim3(2048), dim3(256), 0, st, n, opts->zero_guess ? 0 : 1, A->d_s, w.xh, x->d);
      FEMO_HIP_CHECK(hipGetLastError());
    }
    FEMO_HIP_CHECK(hipEventRecord(ctx->ev1, st));
    FEMO_HIP_CHECK(hipEventSynchronize(ctx->ev1));
    float ms = 0.f;
    FEMO_HIP_CHECK(hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
    info->iterations = iters;
    info->converged = conv;
    info->residual_norm = std::sqrt(gamma);
    info->solve_ms = ms;
    return 0;
  };
  if (!(std::sqrt(gamma0) > tol))   // converged at once, or NaN (reported as breakdown)
    return finish(0, gamma0 == gamma0 ? 1 : -1, gamma0);
  FEMO_TRY(ensure_scaled(A, transpose != 0));

  double hs[FEMO_NSCAL] = {0};
  if (multi) hs[M_TOL2] = tol * tol;
  else { hs[S_GAMMA + 0] = gamma0; hs[S_TOL2] = tol * tol; }
  memcpy(ctx->h_scal, hs, sizeof hs);
  FEMO_HIP_CHECK(hipMemcpyAsync(ctx->d_scal, ctx->h_scal, sizeof hs, hipMemcpyHostToDevice, st));
  FEMO_HIP_CHECK(hipStreamSynchronize(st));  // h_scal is reused below

  const int n_sample = 4, sample_from = 4;
  int n_ev = 0;
  // w = Ah r with both dots, folded and all-reduced into the (delta, gamma) pair of `parity`
  auto merged_spmv = [&](int parity, bool sample) -> int {
    int g1 = gs, g2 = 0;
    if (sample) FEMO_HIP_CHECK(hipEventRecord(ctx->ev_pool[2 * n_ev], st));
    if (m->n_nbr > 0 && m->d_slices_int != nullptr) {
      FEMO_TRY(halo_spmv_overlapped(A, A->d_valsS, w.r, w.q, P, ctx->d_flags, true, true, &g1, &g2));
    } else {
      if (m->n_nbr > 0) FEMO_TRY(halo_raw(m, w.r));
      FEMO_TRY(launch_spmv(A, A->d_valsS, w.r, w.q, P, ctx->d_flags, true, true));
    }
    if (sample) { FEMO_HIP_CHECK(hipEventRecord(ctx->ev_pool[2 * n_ev + 1], st)); ++n_ev; }
    hipLaunchKernelGGL(k_cgm_fold, dim3(1), dim3(1024), 0, st, g1, g2, P, ctx->d_scal + 2 * parity, ctx->d_flags);
    FEMO_TRY(femo_coll_allreduce(ctx, ctx->d_scal + 2 * parity, 2, st));
    return 0;
  };
  if (multi) FEMO_TRY(merged_spmv(0, false));

  const int batch = opts->check_every > 0 ? opts->check_every : 32;
  int it = 0, polled = 0;
  bool done = false;
  int pending[2] = {-1, -1};
  while (!done) {
    const int it_end = it + batch < max_it ? it + batch : max_it;
    for (; it < it_end; ++it) {
      const int cur = it & 1, nxt = cur ^ 1;
      const bool sample = it >= sample_from && it < sample_from + n_sample;
      if (multi) {
        // single-reduction CG: update from the reduced (delta, gamma) of parity cur, then the next SpMV
        hipLaunchKernelGGL(k_cgm_update, dim3(gv), dim3(FEMO_BLOCK), 0, st, n, it, ctx->d_scal, w.q, w.r, w.p, w.sv, w.xh, ctx->d_flags);
        FEMO_TRY(merged_spmv(nxt, sample));
      } else {
        if (sample) FEMO_HIP_CHECK(hipEventRecord(ctx->ev_pool[2 * n_ev], st));
        FEMO_TRY(launch_spmv(A, A->d_valsS, w.p, w.q, P + P_DELTA * FEMO_MAX_PARTIALS, ctx->d_flags, true));
        if (sample) { FEMO_HIP_CHECK(hipEventRecord(ctx->ev_pool[2 * n_ev + 1], st)); ++n_ev; }
        hipLaunchKernelGGL(k_cg_update_r, dim3(gv), dim3(FEMO_BLOCK), 0, st, n, cur, gs, gv, P, ctx->d_scal, w.q, w.r, ctx->d_flags);
        hipLaunchKernelGGL(k_cg_update_xp, dim3(gv), dim3(FEMO_BLOCK), 0, st, n, cur, it, gs, gv, P, ctx->d_scal, w.r, w.p, w.xh, ctx->d_flags);
      }
    }
    FEMO_HIP_CHECK(hipGetLastError());
    const int slot = polled & 1;
    FEMO_HIP_CHECK(hipMemcpyAsync(h_flags + 4 * slot, ctx->d_flags, 4 * sizeof(int32_t), hipMemcpyDeviceToHost, st));
    FEMO_HIP_CHECK(hipEventRecord(ctx->ev_pool[2 * n_sample + slot], st));
    pending[slot] = it;
    ++polled;
    const int prev = polled & 1;
    const bool last = it >= max_it;
    if (pending[prev] >= 0) {
      FEMO_HIP_CHECK(hipEventSynchronize(ctx->ev_pool[2 * n_sample + prev]));
      if (h_flags[4 * prev]) done = true;
      pending[prev] = -1;
    }
    if (!done && last) {
      FEMO_HIP_CHECK(hipEventSynchronize(ctx->ev_pool[2 * n_sample + slot]));
      done = true;
    }
  }
  // final scalars: gamma of the last completed iteration sits in the (iters & 1) slot
  FEMO_HIP_CHECK(hipMemcpyAsync(h_flags, ctx->d_flags, 4 * sizeof(int32_t), hipMemcpyDeviceToHost, st));
  FEMO_HIP_CHECK(hipStreamSynchronize(st));
  const int iters = h_flags[1];
  const int conv = h_flags[0] ? (h_flags[2] ? -1 : 1) : 0;
  double gamma_f = 0.0;
  if (multi) {
    FEMO_HIP_CHECK(hipMemcpyAsync(ctx->h_scal, ctx->d_scal, FEMO_NSCAL * sizeof(double), hipMemcpyDeviceToHost, st));
    FEMO_HIP_CHECK(hipStreamSynchronize(st));
    gamma_f = ctx->h_scal[2 * (iters & 1) + 1];
  } else {
    hipLaunchKernelGGL(k_reduce_partials_at, dim3(1), dim3(1024), 0, st, gv, 1, P + (P_GAMMA + (iters & 1)) * FEMO_MAX_PARTIALS, ctx->d_scal + 8);
    FEMO_HIP_CHECK(hipMemcpyAsync(ctx->h_scal, ctx->d_scal + 8, sizeof(double), hipMemcpyDeviceToHost, st));
    FEMO_HIP_CHECK(hipStreamSynchronize(st));
    gamma_f = ctx->h_scal[0];
  }
  double acc = 0.0;
  for (int i = 0; i < n_ev; ++i) {
    float t = 0.f;
    FEMO_HIP_CHECK(hipEventElapsedTime(&t, ctx->ev_pool[2 * i], ctx->ev_pool[2 * i + 1]));
    acc += t;
  }
  info->spmv_ms = acc;
  info->spmv_samples = n_ev;
  return finish(iters, conv, gamma_f);
}


// ------------------------------------------------------------- BiCGSTAB ------
extern "C" int femo_solve_bicgstab(const femo_mat* A_, int transpose, const femo_vec* b, femo_vec* x,
                                   const femo_solver_opts* opts, femo_solve_info* info) {
  FEMO_REQUIRE(A_ && b && x && opts && info, "null argument");
  femo_mat* A = const_cast<femo_mat*>(A_);
  femo_mesh* m = A->mesh;
  femo_ctx* ctx = m->ctx;
  const int64_t n = m->n_rows;
  FEMO_REQUIRE(b->n >= n && x->n >= m->n_vert, "vector size mismatch in solve_bicgstab");
  FEMO_REQUIRE(b->d != x->d, "solve_bicgstab cannot run in place");
  femo_vec_touch(x);
  memset(info, 0, sizeof *info);
  hipStream_t st = ctx->stream;
  FEMO_HIP_CHECK(hipEventRecord(ctx->ev0, st));
  // transposed operator of a partitioned mesh: applied by scatter + reverse halo add from the untransposed values
  const bool scatter_T = transpose != 0 && !transpose_is_local(m);
  FEMO_TRY(ensure_scaled(A, transpose != 0 && !scatter_T));
  CgWork w;
  FEMO_TRY(ensure_work(ctx, n, m->n_vert, w, 2));
  const bool multi = ctx->nranks > 1 || (ctx->comm != nullptr && femo_env_flag("FEMO_FORCE_MULTI"));
  const int gv = vec_grid(ctx, n);
  const int gs = femo_spmv_grid(m);
  double* P = ctx->d_partials;
  double* S = ctx->d_scal;
  int32_t* h_flags = reinterpret_cast<int32_t*>(ctx->h_scal + FEMO_NSCAL);
  FEMO_HIP_CHECK(hipMemsetAsync(ctx->d_flags, 0, 4 * sizeof(int32_t), st));
  FEMO_HIP_CHECK(hipMemsetAsync(w.sv, 0, ctx->cg_n * sizeof(double), st));   // sv, p: gathered incl. ghosts
  const double* q0 = nullptr;
  if (opts->zero_guess) {
    FEMO_HIP_CHECK(hipMemsetAsync(x->d, 0, x->n * sizeof(double), st));
  } else {
    if (scatter_T) {
      FEMO_TRY(spmv_transposed_scatter(A, A->d_vals, false, x->d, w.q, nullptr));
    } else {
      if (m->n_nbr > 0) FEMO_TRY(halo_raw(m, x->d));
      FEMO_TRY(launch_spmv(A, transpose ? A->d_valsT : A->d_vals, x->d, w.q, nullptr, nullptr));
    }
    q0 = w.q;
  }
  hipLaunchKernelGGL(k_bi_init, dim3(gv), dim3(FEMO_BLOCK), 0, st, n, b->d, q0, A->d_s, w.r, w.r0, w.p, w.q, w.xh, P);
  FEMO_HIP_CHECK(hipGetLastError());
  hipLaunchKernelGGL(k_reduce_partials_at, dim3(1), dim3(1024), 0, st, gv, 2, P, S + 12);
  if (multi) FEMO_TRY(femo_coll_allreduce(ctx, S + 12, 2, st));
  FEMO_HIP_CHECK(hipMemcpyAsync(ctx->h_scal, S + 12, 2 * sizeof(double), hipMemcpyDeviceToHost, st));
  FEMO_HIP_CHECK(hipStreamSynchronize(st));
  const double rr0 = ctx->h_scal[0], bb = ctx->h_scal[1];
  const double bnorm = std::sqrt(bb);
  double tol = opts->rtol * bnorm;
  if (opts->atol > tol) tol = opts->atol;
  info->rhs_norm = bnorm;
  const int max_it = opts->max_it > 0 ? opts->max_it : 10000;
  auto finish = [&](int iters, int conv, double rr) -> int {
    if (n > 0) {
      hipLaunchKernelGGL(k_unscale, dim3(2048), dim3(256), 0, st, n, opts->zero_guess ? 0 : 1, A->d_s, w.xh, x->d);
      FEMO_HIP_CHECK(hipGetLastError());
    }
    FEMO_HIP_CHECK(hipEventRecord(ctx->ev1, st));
    FEMO_HIP_CHECK(hipEventSynchronize(ctx->ev1));
    float ms = 0.f;
    FEMO_HIP_CHECK(hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
    info->iterations = iters;
    info->converged = conv;
    info->residual_norm = std::sqrt(rr);
    info->solve_ms = ms;
    return 0;
  };
  if (!(std::sqrt(rr0) > tol)) return finish(0, rr0 == rr0 ? 1 : -1, rr0);
  double hs[FEMO_NSCAL] = {0};
  hs[B_RHO] = rr0; hs[B_RHO_OLD] = 1.0; hs[B_ALPHA] = 1.0; hs[B_OMEGA] = 1.0; hs[B_RR] = rr0; hs[B_TOL2] = tol * tol;
  memcpy(ctx->h_scal, hs, sizeof hs);
  FEMO_HIP_CHECK(hipMemcpyAsync(S, ctx->h_scal, sizeof hs, hipMemcpyHostToDevice, st));
  FEMO_HIP_CHECK(hipStreamSynchronize(st));

  const int batch = opts->check_every > 0 ? opts->check_every : 32;
  int it = 0;
  bool done = false;
  while (!done) {
    const int it_end = it + batch < max_it ? it + batch : max_it;
    for (; it < it_end; ++it) {
      hipLaunchKernelGGL(k_bi_check, dim3(1), dim3(1), 0, st, it, S, ctx->d_flags);
      hipLaunchKernelGGL(k_bi_p, dim3(gv), dim3(FEMO_BLOCK), 0, st, n, it, S, w.r, w.q, w.p, ctx->d_flags);
      if (scatter_T) {
        FEMO_TRY(spmv_transposed_scatter(A, A->d_valsS, true, w.p, w.q, ctx->d_flags));                       // v
        hipLaunchKernelGGL(k_dot, dim3(gv), dim3(FEMO_BLOCK), 0, st, n, w.r0, w.q, P);                            // (r0, v)
      } else {
        if (m->n_nbr > 0) FEMO_TRY(halo_raw(m, w.p));
        FEMO_TRY(launch_spmv(A, A->d_valsS, w.p, w.q, P, ctx->d_flags, true, false, nullptr, 0, nullptr, w.r0));   // v, (r0, v)
      }
      hipLaunchKernelGGL(k_bi_fold, dim3(1), dim3(1024), 0, st, scatter_T ? gv : gs, 1, P, S + B_R0V, ctx->d_flags);
      if (multi) FEMO_TRY(femo_coll_allreduce(ctx, S + B_R0V, 1, st));
      hipLaunchKernelGGL(k_bi_s, dim3(gv), dim3(FEMO_BLOCK), 0, st, n, S, w.r, w.q, w.sv, ctx->d_flags);
      if (scatter_T) {
        FEMO_TRY(spmv_transposed_scatter(A, A->d_valsS, true, w.sv, w.t, ctx->d_flags));                      // t
        hipLaunchKernelGGL(k_dot, dim3(gv), dim3(FEMO_BLOCK), 0, st, n, w.sv, w.t, P);                            // (s, t)
        hipLaunchKernelGGL(k_dot, dim3(gv), dim3(FEMO_BLOCK), 0, st, n, w.t, w.t, P + FEMO_MAX_PARTIALS);         // (t, t)
      } else {
        if (m->n_nbr > 0) FEMO_TRY(halo_raw(m, w.sv));
        FEMO_TRY(launch_spmv(A, A->d_valsS, w.sv, w.t, P, ctx->d_flags, true, false, nullptr, 0, nullptr, nullptr, true));  // t, (s,t), (t,t)
      }
      hipLaunchKernelGGL(k_bi_fold, dim3(1), dim3(1024), 0, st, scatter_T ? gv : gs, 2, P, S + B_TS, ctx->d_flags);
      if (multi) FEMO_TRY(femo_coll_allreduce(ctx, S + B_TS, 2, st));
      hipLaunchKernelGGL(k_bi_xr, dim3(gv), dim3(FEMO_BLOCK), 0, st, n, it, S, w.p, w.sv, w.t, w.r0, w.xh, w.r, P, ctx->d_flags);
      hipLaunchKernelGGL(k_bi_fold, dim3(1), dim3(1024), 0, st, gv, 2, P, S + 12, ctx->d_flags);
      if (multi) FEMO_TRY(femo_coll_allreduce(ctx, S + 12, 2, st));
      hipLaunchKernelGGL(k_bi_set_rho, dim3(1), dim3(1), 0, st, S + 12, S, ctx->d_flags);
    }
    hipLaunchKernelGGL(k_bi_check, dim3(1), dim3(1), 0, st, it, S, ctx->d_flags);
    FEMO_HIP_CHECK(hipGetLastError());
    FEMO_HIP_CHECK(hipMemcpyAsync(h_flags, ctx->d_flags, 4 * sizeof(int32_t), hipMemcpyDeviceToHost, st));
    FEMO_HIP_CHECK(hipStreamSynchronize(st));
    if (h_flags[0] || it >= max_it) done = true;
  }
  FEMO_HIP_CHECK(hipMemcpyAsync(ctx->h_scal, S, FEMO_NSCAL * sizeof(double), hipMemcpyDeviceToHost, st));
  FEMO_HIP_CHECK(hipStreamSynchronize(st));
  const int conv = h_flags[0] ? (h_flags[2] ? -1 : 1) : 0;
  const int iters = h_flags[0] ? h_flags[1] : it;
  return finish(iters, conv, ctx->h_scal[B_RR]);
}
