// Internal declarations shared by the translation units of libfemo_hip.so.
// Not part of the ABI (see include/femo_hip.h).
#pragma once

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdint>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "femo_hip.h"
#include "femo_hip_test.h"

// ---------------------------------------------------------------- errors ----
void femo_set_error(const char* fmt, ...);

// ------------------------------------------------- host synchronisations ----
// Every blocking wait of the host on the device inside the library is counted (femo_host_sync_stats, include/femo_hip.h;
// bench.py reports host_syncs_per_step): each one is an idle device for the round trip -- 20-50 us -- which is what a
// cycle on a small mesh or on one rank of eight is made of.  The three HIP entry points are
// shadowed for the translation units of the library, which all include this header after hip_runtime.h.
#include <atomic>
extern std::atomic<long long> femo_host_sync_count;
inline hipError_t femo_counted_stream_sync(hipStream_t s) { femo_host_sync_count.fetch_add(1, std::memory_order_relaxed); return hipStreamSynchronize(s); }
inline hipError_t femo_counted_event_sync(hipEvent_t e) { femo_host_sync_count.fetch_add(1, std::memory_order_relaxed); return hipEventSynchronize(e); }
inline hipError_t femo_counted_device_sync() { femo_host_sync_count.fetch_add(1, std::memory_order_relaxed); return hipDeviceSynchronize(); }
#define hipStreamSynchronize(s) femo_counted_stream_sync(s)
#define hipEventSynchronize(e) femo_counted_event_sync(e)
#define hipDeviceSynchronize() femo_counted_device_sync()

#define FEMO_HIP_CHECK(expr)                                                        \
  do {                                                                              \
    hipError_t e__ = (expr);                                                        \
    if (e__ != hipSuccess) {                                                        \
      femo_set_error("%s:%d: %s failed: %s", __FILE__, __LINE__, #expr,             \
                     hipGetErrorString(e__));                                       \
      return 1;                                                                     \
    }                                                                               \
  } while (0)

#define FEMO_NCCL_CHECK(expr)                                                       \
  do {                                                                              \
    ncclResult_t r__ = (expr);                                                      \
    if (r__ != ncclSuccess) {                                                       \
      femo_set_error("%s:%d: %s failed: %s", __FILE__, __LINE__, #expr,             \
                     ncclGetErrorString(r__));                                      \
      return 1;                                                                     \
    }                                                                               \
  } while (0)

#define FEMO_REQUIRE(cond, ...)                                                     \
  do {                                                                              \
    if (!(cond)) {                                                                  \
      femo_set_error(__VA_ARGS__);                                                  \
      return 2;                                                                     \
    }                                                                               \
  } while (0)

#define FEMO_TRY(expr)                                                              \
  do {                                                                              \
    int rc__ = (expr);                                                              \
    if (rc__ != 0) return rc__;                                                     \
  } while (0)

// Tuning / timing switches read from the environment exist only in builds with -DFEMO_TUNING (make TUNING=1): the
// product library never calls getenv on a launch path.  Functional options that tests select
// (FEMO_FORCE_MULTI, FEMO_HOST_VERIFY, FEMO_BPX_DENSE_ALLREDUCE, FEMO_BPX_FUSED, FEMO_SHELL_NO_*) are read once.
#if defined(FEMO_TUNING)
#define FEMO_TUNE_ENV(name) getenv(name)
#else
#define FEMO_TUNE_ENV(name) (static_cast<const char*>(nullptr))
#endif
inline bool femo_env_flag(const char* name) { return getenv(name) != nullptr; }

// ------------------------------------------------------------- constants ----
constexpr int FEMO_WAVE = 64;            // gfx950 wavefront
constexpr int FEMO_BLOCK = 256;          // 4 waves = 4 SELL slices per workgroup
constexpr int FEMO_MAX_PARTIALS = 2048;  // persistent reduction grids: 256 CUs x 8
constexpr int64_t FEMO_LLC_MATRIX_BYTES = 200ll << 20;   // SELL values up to this size are read with ordinary loads (they stay in the 256 MB Infinity Cache)
constexpr int FEMO_NSCAL = 32;           // device scalars of the CG recurrence
constexpr int FEMO_PARTIAL_SLOTS = 12;   // slots of FEMO_MAX_PARTIALS per-block partials each (femo_ctx::d_partials)
constexpr int FEMO_STAGE_SLOTS = 4;      // pinned staging slots per context (8 MiB each)

// SELL-64 with pair interleave: entry k of lane l in a slice starting at P:
//   P + (k >> 1) * 128 + l * 2 + (k & 1)
__host__ __device__ inline int64_t femo_sell_index(int64_t base, int k, int lane) {
  return base + (int64_t)(k >> 1) * (2 * FEMO_WAVE) + lane * 2 + (k & 1);
}

// ------------------------------------------------------------- topology ----
struct FemoTopology {
  int tdim = 0;
  int64_t n_vert = 0, n_rows = 0, n_cell = 0, n_slices = 0;
  // vertex -> cell incidence, SELL-64 (entry s of lane l: vptr[slice] + s*64 + l)
  std::vector<int64_t> vptr;
  std::vector<int32_t> visit_cell;    // (cell << 2) | local index, -1 = padding
  std::vector<uint32_t> visit_slots;  // byte j = off-diagonal slot of the j-th cell vertex other than the visiting one
  // off-diagonal sparsity pattern, SELL-64 pair-interleaved
  std::vector<int64_t> mptr;
  std::vector<int32_t> cols;          // padding entries carry the row's own index
  std::vector<int32_t> rowlen;        // off-diagonal entries STORED per row (n_slices*64): true couplings + structural zeros
  std::vector<uint32_t> real;         // bit k: entry k of the row is a true coupling (all ones outside completed regular slices)
  // Slices whose 64 rows all have cols[k] = row + delta[k] (locally regular numbering; rows that lack a delta of the
  // slice's set carry a structural zero there, topology.cpp) carry their deltas here; SpMV then needs no column
  // indices for them.
  std::vector<int32_t> sdelta;        // n_slices * sdelta_stride, sdelta[s*stride] = INT32_MIN if irregular
  int sdelta_stride = 0;              // = padded max row length
  int64_t n_regular = 0;
  // irregular slices whose columns all lie within +-32767 of their row: 16-bit deltas, same layout as `cols`
  // (flag: sdelta[s*stride + 1] == 1); the SpMV reads these instead of the 32-bit indices
  std::vector<int16_t> cols16;
  int64_t n_short = 0;
  int64_t nnz = 0;                    // true nonzeros incl. diagonal
  int max_rowlen = 0, max_valence = 0;
};

// Returns 0 or sets the error string.
int femo_build_topology(int tdim, int64_t n_vert, int64_t n_rows, int64_t n_cell,
                        const int32_t* conn, FemoTopology& T);
void femo_topology_csr(const FemoTopology& T, int64_t* rowptr, int32_t* col);

// -------------------------------------------------------------- handles ----
struct femo_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  double* d_partials = nullptr;  // [FEMO_PARTIAL_SLOTS][FEMO_MAX_PARTIALS]
  double* d_scal = nullptr;      // [FEMO_NSCAL]
  int32_t* d_flags = nullptr;    // [4]: done, iterations, breakdown, spare
  double* h_scal = nullptr;      // pinned mirror: FEMO_NSCAL doubles + 4 int32
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  std::vector<hipEvent_t> ev_pool;
  ncclComm_t comm = nullptr;
  // The neighbour exchanges (ncclSend/ncclRecv on the comm stream, overlapped with the interior SpMV) have a communicator
  // of their own (ncclCommSplit of `comm`, round 5): RCCL serialises the operations of ONE communicator, so on a shared
  // one the halo exchange of iteration k+1 and the all-reduce of iteration k could not be in flight together.  Opt-in
  // (FEMO_SPLIT_COMM=1, round 6) and agreed on collectively in femo_comm_init: null on EVERY rank unless the split worked on
  // every rank; the default is the single communicator.
  ncclComm_t comm_halo = nullptr;
  struct femo_emu_group* emu = nullptr;      // in-process rank emulation (tests; comm.cpp)
  bool model = false;                        // femo_comm_model: N-rank code paths, collectives complete without moving data
  int rank = 0, nranks = 1;
  // collectives issued through femo_coll_allreduce / femo_coll_neighbors since the last femo_comm_stats(reset):
  // calls and doubles moved (per rank) -- the bench's scaling record and the tests count them per CG iteration
  int64_t n_allreduce = 0, allreduce_doubles = 0, n_neighbor = 0, neighbor_doubles = 0;
  hipStream_t comm_stream = nullptr;          // halo exchange overlapped with interior rows
  hipEvent_t ev_main = nullptr, ev_comm = nullptr;
  int n_cu = 256;
  // pinned staging ring for pageable host memory (hostmem.cpp), allocated on first use
  double* stage[FEMO_STAGE_SLOTS] = {};
  hipEvent_t stage_ev[FEMO_STAGE_SLOTS] = {};
  // copy-out stream of the asynchronous result path (femo_vec_get_host_async) and the device scratch of
  // accumulate-on-copy-out (hostmem.cpp), both created on first use
  hipStream_t copy_stream = nullptr;
  hipEvent_t ev_copy = nullptr;
  double* d_accum = nullptr;
  int64_t accum_n = 0;
  // CG workspace, grown on demand and reused across solves
  double *cg_r = nullptr, *cg_p = nullptr, *cg_q = nullptr, *cg_dinv = nullptr, *cg_s = nullptr;
  double *cg_t = nullptr, *cg_r0 = nullptr;
  int64_t cg_n = 0;
};

struct femo_vec {
  femo_ctx* ctx = nullptr;
  double* d = nullptr;
  int64_t n = 0;
  bool owned = true;
  // provenance of host copies (hostmem.cpp): process-unique id (0 = wrapped memory, never trusted)
  // and a generation that every entry point writing the vector bumps
  uint64_t uid = 0, gen = 0;
  // generation at which the ghost tail was last refreshed from the owners (partitioned meshes): a refresh of a vector nobody
  // has written since is skipped -- no exchange, no new generation (round 6).  Every writer bumps `gen` (femo_vec_touch), so
  // the comparison is exact; vectors wrapped on the fly (solver work vectors) never match.
  uint64_t ghost_gen = UINT64_MAX;
  // an asynchronous copy-out of this vector is (or was) in flight on the context's copy stream: the next writer
  // makes the compute stream wait for it (every writing entry point calls femo_vec_touch BEFORE it launches)
  hipEvent_t d2h_ev = nullptr;
  bool d2h_pending = false;
  // a deferred upload of this vector (femo_vec_set_host_deferred) is (or was) in flight on the copy stream: the first
  // reader or writer on the compute stream waits for it (femo_vec_await; round 4: the f-independent half of the first
  // assembly pass, S A S and the preconditioner weights run under the upload of f)
  hipEvent_t h2d_ev = nullptr;
  bool h2d_pending = false;
};
void femo_vec_wait_readers(femo_vec* v);   // hostmem.cpp
extern "C" int femo_vec_await(const femo_vec* v);     // hostmem.cpp: compute stream waits for a deferred upload of v (no-op otherwise)
inline void femo_vec_touch(femo_vec* v) {
  if (!v) return;
  ++v->gen;
  if (v->d2h_pending) femo_vec_wait_readers(v);
  if (v->h2d_pending) femo_vec_await(v);
}
// out = a + b on the stream (api.hip; the device half of femo_vec_add_to_host)
int femo_launch_sum(double* out, const double* a, const double* b, int64_t n, hipStream_t st);
// out = a * x on the stream (api.hip; a host array known to be a * a device vector is "uploaded" this way)
int femo_launch_scale(double* out, double a, const double* x, int64_t n, hipStream_t st);
int femo_launch_fill(double* out, double value, int64_t n, hipStream_t st);
void femo_vec_register(femo_vec* v);     // after creation: assigns uid, enters the live table
void femo_vec_unregister(femo_vec* v);   // before destruction

struct femo_pc;   // auxiliary-lattice BPX hierarchy (bpx.hip)

// host-side plan of that hierarchy (pc_plan.cpp)
constexpr int FEMO_PC_MAX_LEVELS = 14;
// Packed lattice coordinates of a vertex on the finest lattice: 8 bytes per vertex in both dimensions.
//   2-D: two 32-bit words (x, y), each bin << 20 | 20-bit fraction;
//   3-D (round 5): ONE 64-bit word of three 21-bit fields (x in bits 0..20, y in 21..41, z in 42..62), each
//        bin << 12 | 12-bit fraction (bins <= 511 per axis; the fraction's 2.4e-4 of a bin only perturbs the preconditioner --
//        both transfers decode the same words, so P and P^T stay exact transposes).  Round 1-4 kept three 32-bit words.
constexpr int FEMO_PK_BITS = 20;
constexpr int FEMO_PK3_BITS = 12;
constexpr int FEMO_PK3_FIELD = 21;
constexpr int FEMO_PK_WORDS = 2;          // 32-bit words per vertex
inline int femo_pk_frac_bits(int dim) { return dim == 3 ? FEMO_PK3_BITS : FEMO_PK_BITS; }
struct FemoPcPlan {
  int dim = 0, n_levels = 0;
  int n[FEMO_PC_MAX_LEVELS][3] = {};      // bins per axis, coarsest level first
  int64_t nodes[FEMO_PC_MAX_LEVELS] = {};
  double H[FEMO_PC_MAX_LEVELS] = {};
  double lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0};
  int64_t n_bricks = 0;
  std::vector<uint32_t> pk, pk_sorted;    // FEMO_PK_WORDS words per owned vertex, vertex order / sorted order
  std::vector<int32_t> perm;              // sorted position -> vertex
  std::vector<int64_t> brick_ptr;         // n_bricks + 1
  std::vector<int32_t> brick_base;        // 3 per brick
  std::vector<uint32_t> bin_ptr;          // 65 per brick
};
int femo_pc_make_plan(int dim, int64_t n_rows, const double* x, const double* lo, const double* hi,
                      int64_t n_vert_global, double spacing, FemoPcPlan& P);
double femo_pc_spacing();

struct femo_mesh {
  femo_ctx* ctx = nullptr;
  int tdim = 0;
  int64_t n_vert = 0, n_rows = 0, n_cell = 0, n_slices = 0;
  int64_t nnz = 0, sell_entries = 0, visit_entries = 0;
  int max_rowlen = 0, max_valence = 0;
  double* d_x = nullptr;
  int32_t* d_conn = nullptr;
  int64_t* d_vptr = nullptr;
  int32_t* d_visit_cell = nullptr;
  uint32_t* d_visit_slots = nullptr;
  int64_t* d_mptr = nullptr;
  int32_t* d_cols = nullptr;
  int32_t* d_rowlen = nullptr;
  uint32_t* d_rowreal = nullptr;  // FemoTopology::real
  int32_t* d_sdelta = nullptr;   // per-slice column deltas (see FemoTopology::sdelta)
  int sdelta_stride = 0;
  int64_t n_regular = 0, n_short = 0;
  int16_t* d_cols16 = nullptr;   // FemoTopology::cols16
  void* d_visit_rec = nullptr;   // per incidence entry, 12 B: (slots, 1/(36|T|)) for the Poisson walks, built on first use
  double* d_load = nullptr;      // load vector of the Poisson residual for the f identified by (load_uid, load_gen)
  double* d_zero_load = nullptr; // zeros of the same length (deferred-upload path of femo_launch_system)
  uint64_t load_uid = 0, load_gen = 0;
  int pcg_last_iters = 0, pcg_prev_iters = 0;   // iterations of the last two converged BPX-PCG solves on this mesh (size the first batch)
  double pcg_rate = 0.0;                        // ln(gamma_0 / gamma_end) / iterations of the last converged merged BPX-PCG solve with >= 4 iterations
  struct femo_mat* mass = nullptr;  // P1 mass matrix on the operator pattern (geometry only; built on first use): dJ/du = M (u - u_d)
  double* d_mass_e = nullptr;       // n_vert: the difference the mass product is applied to
  double* d_mass_g = nullptr;       // n_rows: M (u - u_d) of the vectors identified by mass_key (functional value -> grad_u)
  uint64_t mass_key[4] = {0, 0, 0, 0};   // (u uid, u gen, u_d uid, u_d gen); uid 0 = unknown: never matches
  double* d_cellvol = nullptr;      // |T_c| per cell, and the same with 0 for cells owned by another rank (first vertex
  double* d_cellvol_own = nullptr;  // a ghost): geometry + partition only, built on first use
  double* d_cell_t = nullptr;       // n_cell scratch: f_c |T_c| / (D+1) for the load-vector walk
  double* d_pipe_dummy = nullptr; // k_poisson_system_pipe: a line that absorbs the stores of padded entries / lanes
  double* d_ubc = nullptr;        // ... and u with the prescribed values imposed
  uint8_t* d_bfacets = nullptr;  // per cell: bit k = facet opposite local vertex k is on the boundary (optional)
  int32_t* d_tperm = nullptr;  // lazily built: SELL entry -> SELL entry of the transposed nonzero
  std::vector<int64_t> h_mptr;
  // halo plan (n_nbr == 0 on a single GPU)
  int n_nbr = 0;
  std::vector<int32_t> nbr;
  std::vector<int64_t> send_ptr, recv_ptr;
  int32_t* d_send_idx = nullptr;
  double* d_send_buf = nullptr;
  // the send list by VERTEX (round 5; a vertex sent to several neighbours occupies several slots): the merged BPX-PCG loop
  // computes the new search direction on these vertices first, straight into the send buffer (k_prolong_mesh), so that the
  // exchange travels under the rest of the prolongation and the interior SpMV and no pack kernel competes with the SpMV
  int64_t n_send_verts = 0;
  int32_t* d_send_uvert = nullptr;   // distinct owned vertices that are sent, ascending
  int32_t* d_send_uptr = nullptr;    // n_send_verts + 1: their slots in the send buffer ...
  int32_t* d_send_uslot = nullptr;   // ... listed here
  uint8_t* d_send_flag = nullptr;    // n_rows bytes: 1 = the vertex is on the send list
  double* d_scratch = nullptr;  // n_vert doubles, lazily allocated (Dirichlet lifting)
  // slices without / with ghost columns (set with the halo plan)
  int32_t* d_slices_int = nullptr; int32_t* d_slices_bnd = nullptr;
  int64_t n_int = 0, n_bnd = 0;
  int32_t* d_slices_all = nullptr;       // both, XCD eighth by XCD eighth: interior first, ghost-column slices (sign bit) last
  struct FemoHaloDirect* hd = nullptr;   // device-initiated ghost refresh (round 6), see FemoHaloDirect
  // geometry of the whole (global) mesh for the BPX lattice; local values until femo_mesh_set_global
  double bbox_lo[3] = {0, 0, 0}, bbox_hi[3] = {0, 0, 0};
  int64_t n_vert_global = 0;
  uint8_t* d_bvmask = nullptr;   // vertices on marked boundary facets (built with d_bfacets)
  uint64_t bfacets_version = 0;
  femo_pc* pc = nullptr;         // built on first use
};

// ---- device-initiated ghost refresh (round 6; halo_direct.hip) ------------------------------------------------------
// Every rank owns an INBOX per mesh -- uncached device memory: two generations of one double per ghost vertex and one
// 64-byte counter line per neighbour -- that its neighbours map (hipIpcOpenMemHandle between processes, the plain
// address between the emulated ranks of one process).  A producer kernel stores the owned values a neighbour needs
// straight into that neighbour's inbox and, workgroup by workgroup, bumps the neighbour's counter for this rank
// (system-scope release); a consumer spins on its own counters until they reach epoch x (workgroups per exchange of
// that neighbour) and reads its ghosts from its own inbox.  No send buffer, no pack launch, no second stream, no event
// hop, no RCCL launch: the exchange is two kernels of the compute stream, or none where the producer is the kernel
// that computes the values anyway (k_prolong_mesh of the merged BPX-PCG) .
//   generation = epoch & 1: a rank may run ONE exchange ahead of a neighbour (it cannot finish exchange e + 1 before
//   the neighbour has produced e + 1, which the neighbour's stream orders behind its own consumption of e), so two
//   generations make overwriting unread ghosts impossible without any acknowledgement traffic.
//   every producer of a mesh uses the same number of workgroups (n_blocks) and bumps even when it has nothing to do
//   (a solve that converged before the launch ran): the counters stay epoch x n_blocks whatever the kernels decided.
constexpr int FEMO_MAX_NBR = 32;
constexpr int FEMO_HALO_CNT_STRIDE = 8;              // counters sit 64 bytes apart
struct FemoHaloPeers {                               // lives in device memory (femo_mesh::hd->d_peers), read with scalar loads
  int32_t n_nbr;
  int32_t n_send;                                    // slots of the send list
  int32_t send_ptr[FEMO_MAX_NBR + 1];                // slot ranges by neighbour
  double* seg[FEMO_MAX_NBR];                         // neighbour k: start of this rank's segment in ITS inbox (generation 0)
  int64_t stride[FEMO_MAX_NBR];                      // ... and the distance to generation 1 (its ghost count)
  unsigned long long* cnt[FEMO_MAX_NBR];             // ... and its counter for this rank
  const uint8_t* slot_nbr;                           // neighbour index of every send slot
  double* const* dst[2];                             // per send slot: its address in the owner's inbox, generation 0 / 1 (one
                                                     // coalesced pointer load per store instead of four dependent table lookups)
};
struct FemoHaloDirect {
  bool ready = false;
  bool same_process = false, loopback = false;       // emulated ranks / the model communicator (own scratch as "peer")
  void* inbox_raw = nullptr;                         // [2 x n_ghost doubles | pad | n_nbr counter lines]
  double* inbox = nullptr;
  unsigned long long* counters = nullptr;
  int64_t n_ghost = 0;
  int n_blocks = 0;                                  // workgroups (= counter bumps per neighbour) of this rank's producers
  FemoHaloPeers* d_peers = nullptr;
  int32_t* d_blocks = nullptr;                       // [n_nbr]: bumps per exchange of neighbour k's producers
  uint8_t* d_slot_nbr = nullptr;
  double** d_dst = nullptr;                          // [2][n_send]: FemoHaloPeers::dst
  double* d_loop = nullptr;                          // loopback: absorbs the producer's stores
  int32_t* d_err = nullptr;                          // set by a consumer that gave up waiting
  unsigned long long epoch = 0;                      // exchanges issued on this mesh (host side; SPMD: the same on every rank)
  unsigned long long loop_epoch = 0;                 // the exchange the merged BPX-PCG's last prolongation produced (consumed by the next product)
  std::vector<void*> opened;                         // IPC mappings (closed with the mesh)
  int64_t exchanges = 0;
};
struct femo_mesh;
void femo_halo_direct_free(femo_mesh* m);
int femo_emu_rendezvous(femo_ctx* ctx, hipStream_t st);   // comm.cpp: emulated ranks meet on the host (no-op for real ranks)
bool femo_halo_direct_ready(const femo_mesh* m);
// y[n_rows + i] <- the owners' x, through the inboxes, on `st`
int femo_halo_direct_exchange(femo_mesh* m, const double* x_owned, double* ghost_tail, hipStream_t st);
// consumer half only (the producer was another kernel: k_prolong_mesh), and the epoch bookkeeping of such a producer
unsigned long long femo_halo_direct_begin(femo_mesh* m);                 // returns the epoch of the exchange now being produced
int femo_halo_direct_pull(femo_mesh* m, unsigned long long epoch, double* ghost_tail, hipStream_t st);
#if defined(__HIPCC__)
// producer side, at the end of a workgroup that stored into the peers' inboxes (or had nothing to store): uniform per block
// The inboxes are UNCACHED memory for every mapper (hipDeviceMallocUncached: the stores above left for their owner's memory
// and no cache on the way keeps them), so what the bump needs is ORDER, not a cache flush: every wave waits until its own
// stores have been acknowledged (vmcnt), the waves of the workgroup meet, then the counters are bumped with relaxed
// system-scope atomics (executed at the owner's memory).  The by-the-book alternative -- __threadfence_system() + a release
// atomic -- writes this XCD's whole L2 back on gfx950 (buffer_wbl2 sc1) once per producer workgroup: measured +11 us per
// PCG iteration on the 1.26 M-row block, more than the two event hops the design removes.  The self-test of the plan
// (femo_mesh_halo_direct_selftest: three rounds of epoch-dependent patterns) is what checks the assumption on a given box.
__device__ __forceinline__ void femo_halo_signal(const FemoHaloPeers* P) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if ((int)threadIdx.x < P->n_nbr)      // (global address space spelled out: pointers loaded from memory would make these flat_ instructions)
    (void)__hip_atomic_fetch_add((__attribute__((address_space(1))) unsigned long long*)P->cnt[threadIdx.x], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
// Payload stores and loads are `sc0 sc1` instructions (relaxed system-scope atomics) on top of the uncached mapping: every
// byte leaves for / comes from the memory side whatever the page attributes a mapper ended up with (MI355X_MICROARCH.md,
// "Valid forms": {sc0 sc1 stores and loads both sides} + every storing wave's vmcnt wait + barrier before the counter add).
__device__ __forceinline__ void femo_halo_store(const FemoHaloPeers* P, unsigned long long epoch, int32_t slot, double v) {
  __hip_atomic_store((__attribute__((address_space(1))) double*)P->dst[epoch & 1ull][slot], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ double femo_halo_load(const double* p) {
  return __hip_atomic_load((const __attribute__((address_space(1))) double*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
#endif

struct femo_bc {
  femo_mesh* mesh = nullptr;
  int64_t n = 0;
  int32_t* d_dofs = nullptr;
  double* d_vals = nullptr;
  uint8_t* d_mask = nullptr;  // n_vert
  double* d_dense = nullptr;  // n_vert: prescribed value on the set, 0 elsewhere
  uint64_t* d_rowmask = nullptr;  // per row: bit k = its k-th off-diagonal column is in the set, bit 63 = the row itself
                                  // (rows with at most 62 columns; saves the assembly one byte gather per matrix entry)
  uint64_t uid = 0;           // process-unique, never reused (keys cached preconditioner data)
};

struct femo_mat {
  femo_mesh* mesh = nullptr;
  double* d_diag = nullptr;   // n_slices*64
  double* d_vals = nullptr;   // sell_entries
  double* d_valsT = nullptr;  // lazily built transposed values
  bool valsT_valid = false;
  double* d_valsS = nullptr;  // S A S (or S A^T S), S = diag^-1/2: what the CG iterates on
  double* d_s = nullptr;      // S, n_vert entries (ghosts filled by halo exchange)
  bool scaled_valid = false, scaled_transposed = false;
  bool s_valid = false;       // d_s matches the current diagonal (set before the values are scaled)
  // identity rows of the last assembly (strong Dirichlet set): the Krylov loops solve them up front
  uint8_t* d_idrows = nullptr;  // n_vert bytes (own copy: the femo_bc may be destroyed before the matrix)
  bool has_idrows = false;
  uint64_t idrows_uid = 0;
  // BPX: which vertices the last assembly pinned (strong Dirichlet set and/or Nitsche facets)
  bool bpx_ok = false;          // assembled from a second-order scalar PDE on a geometric mesh
  uint8_t* d_pcmask = nullptr;  // n_vert bytes, valid when pc_key != 0 and pc_has_mask
  bool pc_has_mask = false;
  uint64_t pc_key = 0;
};

// ------------------------------------------------------ device utilities ----
#if defined(__HIPCC__)
__device__ __forceinline__ double femo_wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;  // valid in lane 0
}

// Deterministic block sum (fixed tree), result valid in thread 0.
template <int NT>
__device__ __forceinline__ double femo_block_sum(double v, double* lds /* NT/64 */) {
  v = femo_wave_sum(v);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) lds[w] = v;
  __syncthreads();
  double s = 0.0;
  if (threadIdx.x == 0) {
#pragma unroll
    for (int i = 0; i < NT / 64; ++i) s += lds[i];
  }
  return s;
}

// the same sum delivered to every thread of the block
template <int NT>
__device__ __forceinline__ double femo_block_sum_all(double v, double* lds /* NT/64 */) {
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

// Blocks are dealt round-robin over the 8 XCDs (blockIdx % 8 labels the XCD
// group); give each XCD one contiguous range of logical blocks so that its L2
// sees one contiguous slab of rows.  Speed only, never correctness.
__device__ __forceinline__ int64_t femo_xcd_block(int64_t b, int64_t nb) {
  const int64_t per = nb >> 3;          // blocks per XCD in the divisible part
  const int64_t main = per << 3;
  if (b >= main) return b;              // tail blocks keep their index
  return (b & 7) * per + (b >> 3);
}
#endif

// kernel launchers implemented in the .hip files -------------------------------
int femo_launch_residual(femo_mesh* m, int pde, const double* params, const double* u,
                         const double* f, const double* aux, double* r, uint64_t f_uid = 0, uint64_t f_gen = 0);
// f_vec / A_solve (optional): the input vector behind `f` and the matrix behind (diag1, vals1).  When an upload of f is
// still in flight the linear-Poisson pass assembles the matrix and K u' first, scales A_solve for the Krylov loop, and
// only then waits for f, builds the load vector and completes the right-hand side.
int femo_launch_system(femo_mesh* m, int pde, const double* params, const double* u, const double* f,
                       const double* aux, const uint8_t* bcmask, const double* bcval, double* diag0, double* vals0,
                       double* diag1, double* vals1, double* rhs, uint64_t f_uid = 0, uint64_t f_gen = 0,
                       const uint64_t* bc_rowmask = nullptr, const femo_vec* f_vec = nullptr, femo_mat* A_solve = nullptr);
// (femo_mat_prescale: public since ABI 8, include/femo_hip.h)
int femo_launch_cell_expr(femo_mesh* m, int kind, const double* params, const double* in, double* out);
int femo_launch_dRdf(femo_mesh* m, int pde, const double* params, const double* u,
                     const double* f, double* vals);
int femo_launch_dRdf_apply(femo_mesh* m, const double* vals, int transpose, const double* x,
                           double* y, int accumulate);
int femo_launch_dRdf_cell(femo_mesh* m, double* cvals);
int femo_launch_dRdf_cell_apply(femo_mesh* m, const double* cvals, int transpose, const double* x, double* y, int accumulate);
// key (optional): uid / generation of u and u_d -- M (u - u_d) formed for the value is kept for the gradient
int femo_launch_functional_value(femo_mesh* m, int kind, const double* params, const double* u,
                                 const double* f, const double* ud, double* host_value, const uint64_t* key = nullptr);
int femo_launch_functional_grad_u(femo_mesh* m, int kind, const double* params, const double* u,
                                  const double* f, const double* ud, double* g, const uint64_t* key = nullptr);
int femo_launch_functional_grad_f(femo_mesh* m, int kind, const double* params, const double* u,
                                  const double* f, const double* ud, double* g);
int femo_launch_spmv(const femo_mat* A, const double* vals, const double* x, double* y,
                     double* partials /* or null: fused dot(x,y) partials */);
int femo_spmv_grid(const femo_mesh* m);
int femo_halo_exchange_on(femo_mesh* m, femo_vec* x, hipStream_t st);
int femo_mesh_classify_slices(femo_mesh* m);
int femo_mat_ensure_transpose(femo_mat* A);
int femo_coll_allreduce(femo_ctx* ctx, double* d, int64_t count, hipStream_t st);
int femo_coll_neighbors(femo_ctx* ctx, int n_nbr, const int32_t* nbr, const int64_t* send_ptr, const double* d_send,
                        const int64_t* recv_ptr, double* d_recv, hipStream_t st);
int femo_pc_build(femo_mesh* m);
void femo_pc_destroy(femo_mesh* m);
// mode 0: out = M^-1 rh (scaled variables).  mode 1: out = M^-1 rh + beta out with beta = gamma'/(*gamma_cur),
// gamma' = rh.M^-1 rh = *rho + g_L.e_L, written to *gamma_nxt.  mode 2: like 1 with beta = 0.
// stopping test of the BPX-PCG loop, evaluated inside the preconditioner apply (see k_prolong_mesh)
struct FemoPcgStop {
  double rtol2_factor;
  double atol_pc2;
  double* tolg2;
  int32_t* flags;
  int it;
};
// The solver's x += alpha p, carried by the idle compute units of the single-workgroup coarse-lattice kernel (round 3): the
// update depends on nothing the preconditioner computes and nothing in the iteration waits for it, while k_lattice_coarse
// keeps one CU busy for ~24 us at C4.  alpha: device scalar written by k_pcg_xr.  Taken only when femo_pc_carries_xupdate().
struct FemoXUpdate {
  double* x;
  const double* p;
  const double* alpha;
  int64_t n;
};
// ---- merged BPX-PCG (round 4): ONE all-reduce and one halo exchange per iteration on N ranks ------------------
// The restricted residual of the finest levels is kept as STATE and updated by linearity, g <- g - alpha P^T(q/s)
// with q = A p: the restriction no longer waits for alpha, so p.q travels in the same all-reduce as the lattice
// sums; r.r of the updated residual and the lattice dot g_L.e_L = sum_l sum_i C_l,i g_l,i^2 (e_l = C_l g_l + I e_l-1,
// g_l-1 = I^T g_l) follow from reduced scalars, r'.r' = r.r - 2 alpha r.q + alpha^2 q.q and likewise for the nodes a
// single rank touches.  Device scalars (femo_ctx::d_scal):
constexpr int MS_GAMMA = 0;    // [0], [1]: gamma = r.M^-1 r of even / odd iterations
constexpr int MS_PQ = 2;       // p.Ap of the current iteration (breakdown test)
constexpr int MS_TOL2 = 3;     // atol^2 (Jacobi norm)
constexpr int MS_RR = 4;       // r.r (global)
constexpr int MS_TOLG = 5;     // stopping threshold on gamma
constexpr int MS_FACTOR = 6;   // rtol^2 * b.b / r0.r0
constexpr int MS_BB = 7;       // b.D^-1 b
constexpr int MS_ALPHA = 8;
constexpr int MS_DOTC = 9;     // sum over the LDS-resident coarse levels of C g^2
constexpr int MS_RED = 10;     // N > 1: the seven reduced scalars p.q, r.q, q.q, r.r, and over single-rank finest nodes C g g, C g h, C h h
constexpr int MS_NRED = 7;
struct FemoMergedVecs {
  double* x; double* r; double* p; const double* q;   // q == nullptr: the first apply (g = P^T(r/s), alpha = -1, no updates)
  int64_t n;
  int cur;                                             // parity of the current gamma
  int nb_q[2]; const double* Pq[2];                    // SpMV partial triples [p.q | q.q | r.q] (one or two launches), FEMO_MAX_PARTIALS apart
  int gv;                                              // grid of the mesh-sized kernels
  double atol2;                                        // absolute threshold on r.r (0: none)
};
bool femo_pc_merged_ok(femo_mesh* m);                 // the fused lattice cycle with two brick-fused levels runs on this mesh
struct FemoZeroExtra { double* p[3]; int64_t n[3]; int count; };     // small regions the caller wants cleared by the same launch
int femo_pc_merged_begin(femo_mesh* m, const double* s, const uint8_t* mask, const FemoZeroExtra* extra = nullptr);
int femo_pc_merged_apply(femo_mesh* m, const uint8_t* mask, uint64_t mask_key, const FemoMergedVecs& V, double* S,
                         const int32_t* done, const struct FemoPcgStop* stop);
// N > 1: does femo_pc_merged_apply start the halo exchange of the direction it produces (interface vertices first, send buffer
// filled by the prolongation itself)?  Then the loop's SpMV must not exchange again: femo_halo_spmv_inflight.
bool femo_pc_merged_sends_halo(const femo_mesh* m);
int femo_pc_merged_collectives(const femo_mesh* m);   // all-reduces per iteration of the merged loop on this mesh (0 on one rank)
// nb_rho > 0: rho = rh.rh is folded from rho_partials[0:nb_rho] inside the apply (and stored to *rho)
int femo_pc_apply(femo_mesh* m, const uint8_t* mask, uint64_t mask_key, const double* s, const double* rh, double* out,
                  int mode, double* rho, const double* gamma_cur, double* gamma_nxt, const int32_t* done, int gv,
                  bool rho_is_partial = false, const FemoPcgStop* stop = nullptr, int nb_rho = 0,
                  const double* rho_partials = nullptr, const FemoXUpdate* xupdate = nullptr);
bool femo_pc_carries_xupdate(const femo_mesh* m);   // after the first apply on the mesh: the fused lattice cycle runs
bool femo_pc_can_piggyback(const femo_mesh* m);
int femo_pc_begin(femo_mesh* m, const double* s, const uint8_t* mask);
int femo_pc_levels(const femo_mesh* m, int* n_levels, int64_t* finest_nodes);
int femo_reduce_to_host(femo_ctx* ctx, int nblocks, int nsums, double* host_out);
