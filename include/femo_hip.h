/* femo_hip.h -- C-ABI of libfemo_hip.so: MI355X (gfx950) engine for femo's
 * PDE-residual hot path (assemble residual / dR/du / dR/df, forward and
 * transposed linear solves, functional and its partials).
 *
 * The reference (RuruX/femo @ 2024_08_07) has NO FFI: its hot path is Python
 * calling dolfinx/PETSc through pybind/petsc4py.  Each entry point below
 * therefore cites the *Python call site* it replaces (paths relative to the
 * reference tree).  INTEGRATION.md shows the ctypes stub a maintainer would add
 * to femo/fea/utils_dolfinx.py to bind them.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes, no C++/torch types.
 *   - every function returns 0 on success, non-zero on error; the message is
 *     available from femo_last_error() (thread-local).  No exception crosses
 *     the ABI.
 *   - the caller owns every host buffer; the library owns device memory behind
 *     opaque handles.  A femo_ctx owns one HIP stream; all work of the handles
 *     created from it is enqueued on that stream.  Handles are not thread-safe;
 *     distinct contexts may be used from distinct threads.
 *   - all arithmetic is fp64; indices are int32 (dolfinx/PETSc default), sizes
 *     and offsets int64.
 *   - host<->device copies are synchronous w.r.t. the host (they synchronise the
 *     context's stream); compute entry points are asynchronous unless they
 *     return a scalar to the host.  See "host memory" below for pinned blocks and
 *     the provenance rule that elides repeated uploads.
 */
#ifndef FEMO_HIP_H
#define FEMO_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FEMO_ABI_VERSION 10

typedef struct femo_ctx  femo_ctx;   /* device + stream + reduction workspace (+ RCCL communicator) */
typedef struct femo_vec  femo_vec;   /* fp64 device vector  (dolfinx Function.vector / PETSc Vec)     */
typedef struct femo_mesh femo_mesh;  /* P1 simplex mesh + vertex->cell incidence + sparsity pattern   */
typedef struct femo_bc   femo_bc;    /* strong Dirichlet set (fea_dolfinx.py:169-176 add_strong_bc)   */
typedef struct femo_mat  femo_mat;   /* N x N sparse matrix on the mesh pattern (PETSc Mat)           */
typedef struct femo_shell femo_shell; /* Reissner-Mindlin shell space CG2^3 x CG1^3 on a triangulated surface (below) */

/* closed catalogue of residual forms (UFL is not available) */
enum femo_pde_kind {
  FEMO_PDE_POISSON = 0,       /* examples/poisson_opt/run_poisson_opt.py:32-38            */
  FEMO_PDE_NL_POISSON = 1,    /* examples/nonlinear_poisson_opt/...py:88-125              */
  FEMO_PDE_MASS = 2,          /* inner(Pv, w) dx, matrix only (utils_dolfinx.py:567-572)  */
  FEMO_PDE_EB_BEAM = 3        /* examples/beam_thickness_opt/run_thickness...py:71-79: cubic Hermite
                                 Euler-Bernoulli beam; mesh vertices = DOFs (w_i, th_i), 4 per element,
                                 params = {E, width}, f = thickness per element, aux = nodal load */
};

/* closed catalogue of scalar output forms */
enum femo_functional_kind {
  FEMO_J_L2_TRACKING = 0      /* 1/2 int (u-u_d)^2 + alpha/2 int f^2 ; run_poisson_opt.py:74-76 */
};

enum femo_mesh_info_key {
  FEMO_MESH_TDIM = 0, FEMO_MESH_N_VERT = 1, FEMO_MESH_N_ROWS = 2, FEMO_MESH_N_CELL = 3,
  FEMO_MESH_NNZ = 4,            /* true nonzeros of the N x N pattern incl. diagonal   */
  FEMO_MESH_SELL_ENTRIES = 5,   /* padded off-diagonal entries stored                  */
  FEMO_MESH_MAX_ROWLEN = 6, FEMO_MESH_MAX_VALENCE = 7, FEMO_MESH_N_SLICES = 8,
  FEMO_MESH_VISIT_ENTRIES = 9,
  FEMO_MESH_REGULAR_SLICES = 10, /* slices whose column indices are row + per-slice deltas */
  FEMO_MESH_SHORT_SLICES = 11,   /* irregular slices with 16-bit column deltas (SpMV reads 2 B per entry) */
  FEMO_MESH_INFO_COUNT = 12
};

/* preconditioner of femo_solve_cg (femo_solver_opts.pc) */
enum {
  FEMO_PC_JACOBI = 0,   /* diagonal scaling only                                                    */
  FEMO_PC_BPX    = 1    /* + additive multilevel correction on an auxiliary lattice (csrc/bpx.hip);  */
                        /*   operators assembled from FEMO_PDE_POISSON / FEMO_PDE_NL_POISSON only    */
};

/* Stopping rules.  D = diag(A), M = the preconditioner (rows / columns of the Dirichlet set the matrix was
 * eliminated with are identity rows: they are solved exactly up front and take no part in any norm).
 *   FEMO_PC_JACOBI:  sqrt(r^T D^-1 r) <= max(rtol sqrt(b^T D^-1 b), atol)
 *   FEMO_PC_BPX:     sqrt(r^T M^-1 r) <= max(rtol sqrt(b^T M^-1 b), atol_pc)   or   sqrt(r^T D^-1 r) <= atol (if > 0)
 *                    r^T M^-1 r is PCG's own gamma; sqrt(gamma / gamma_0) tracks the relative energy-norm error
 *                    independently of the mesh size, which the Jacobi-norm ratio does not (its constant is cond(D^-1 A)). */
typedef struct femo_solver_opts {
  double rtol;
  double atol;          /* absolute, Jacobi norm (both preconditioners)                 */
  int32_t max_it;
  int32_t zero_guess;   /* 1: x is taken as 0 on entry (skips the initial SpMV)         */
  int32_t check_every;  /* iterations enqueued between host convergence polls (0 = 32)  */
  int32_t pc;           /* FEMO_PC_*                                                     */
  double atol_pc;       /* absolute, norm of the preconditioner (FEMO_PC_BPX only)      */
} femo_solver_opts;

typedef struct femo_solve_info {
  int32_t iterations;
  int32_t converged;    /* 1 converged, 0 hit max_it, -1 breakdown (NaN); femo_shell_solve only: 2 = stalled at the
                           attainable accuracy (below 1e-9 of the initial residual in the preconditioner's norm, no
                           progress over 8 polls)                                          */
  double  residual_norm;/* sqrt(r^T D^-1 r) at exit (recurrence residual)               */
  double  pc_residual_norm; /* sqrt(r^T M^-1 r) at exit and for the right-hand side     */
  double  pc_rhs_norm;      /* (FEMO_PC_BPX; 0 otherwise)                                */
  double  rhs_norm;     /* sqrt(b^T D^-1 b)                                             */
  double  solve_ms;     /* device time of the solve (HIP events on the ctx stream)      */
  double  spmv_ms;      /* accumulated device time of sampled SpMV launches             */
  int32_t spmv_samples; /* number of SpMV launches that were individually timed         */
  int32_t loop_allreduces;   /* all-reduce calls issued inside the iteration loop (0 on one rank): the merged BPX-PCG issues one per iteration */
} femo_solve_info;

/* ---- errors / probing ---------------------------------------------------- */
const char* femo_last_error(void);
int  femo_abi_version(void);
int  femo_device_count(int* n);

/* ---- context --------------------------------------------------------------
 * stream: a hipStream_t created by the caller (e.g. torch.cuda.Stream().cuda_stream)
 * or NULL to let the library create one.  Replaces the implicit PETSc/MPI
 * global state (utils_dolfinx.py:32 comm = MPI.COMM_WORLD).                  */
int   femo_ctx_create(int device_id, void* stream, femo_ctx** out);
int   femo_ctx_destroy(femo_ctx* ctx);
int   femo_ctx_sync(femo_ctx* ctx);
void* femo_ctx_stream(femo_ctx* ctx);

/* ---- vectors --------------------------------------------------------------
 * utils_dolfinx.py:155-167 getFuncArray / setFuncArray, :300-311 update.     */
int     femo_vec_create(femo_ctx* ctx, int64_t n, femo_vec** out);          /* zero-filled */
int     femo_vec_wrap(femo_ctx* ctx, void* device_ptr, int64_t n, femo_vec** out); /* borrow */
int     femo_vec_destroy(femo_vec* v);
int64_t femo_vec_size(const femo_vec* v);
void*   femo_vec_device_ptr(femo_vec* v);         /* mutable: counts as a write to v (provenance records and per-content
                                                     caches of v are dropped; a pending copy-out of v is waited for
                                                     on the device before the next kernel on the context's stream) */
const void* femo_vec_device_ptr_const(const femo_vec* v);   /* read-only access, no side effect */
int     femo_vec_set_host(femo_vec* v, const double* host, int64_t n);      /* setFuncArray */
int     femo_vec_get_host(const femo_vec* v, double* host, int64_t n);      /* getFuncArray */
/* host[0:n] += v[0:n]: the `d_inputs[name] += dRdf^T dR` of state_model.py:190-200 without a second
 * pass over the host array (the add runs in the host threads that drain the staging slots).       */
int     femo_vec_add_to_host(const femo_vec* v, double* host, int64_t n);
int     femo_vec_get_host_async(const femo_vec* v, double* host, int64_t n);   /* see "asynchronous results" below */
/* Deferred upload (round 4): like femo_vec_set_host, but from a whole block of femo_host_alloc the copy runs on the
 * context's copy stream and the call returns at once; the matrix half of the next assembly pass, its scaling and the
 * preconditioner weights -- none of which depend on the input -- then run under the transfer (state_model.py:94-103
 * sends the inputs before it solves).  Writers of v and the assembly entry points wait by themselves; before any
 * OTHER entry point reads v the caller says femo_vec_await_upload (engine.deferred_uploads does it on leaving the
 * scope).  Everything else (caller-owned or pageable memory, elided uploads) is femo_vec_set_host.              */
int     femo_vec_set_host_deferred(femo_vec* v, const double* host, int64_t n);
int     femo_vec_await_upload(const femo_vec* v);
int     femo_vec_fill(femo_vec* v, double value);                           /* Vec.set      */
int     femo_vec_copy(femo_vec* dst, const femo_vec* src);
int     femo_vec_axpy(femo_vec* y, double a, const femo_vec* x);            /* y += a x     */
int     femo_vec_dot(const femo_vec* x, const femo_vec* y, int64_t n, double* out);
/* out[j] = x[j][0:n] . y[j][0:n] for j < k <= 4: one pass, one reduction (all-reduced over the ranks) and ONE
 * host synchronisation for all of them -- the norms a Newton step of utils_dolfinx.py:419-449 looks at
 * (||F||, ||u||, u.Au) used to cost a stream drain each.                                                    */
int     femo_vec_dots(int k, const femo_vec* const* x, const femo_vec* const* y, int64_t n, double* out);
/* ABI 10: the same, plus out[k] = rho_0 = sum over the non-identity rows of (rb_i / sqrt(diag(A)_i))^2 -- the squared
 * Jacobi norm of the right-hand side `rb` the Krylov loops of A start from.  Newton (utils_dolfinx.py:419-449: always three
 * passes) reads it in the one host synchronisation it has after an assembly pass and takes the solver's own "nothing to
 * iterate on" decision (rho_0 <= atol^2) without launching the solve.                                                  */
int     femo_vec_dots_rhs(int k, const femo_vec* const* x, const femo_vec* const* y, int64_t n, double* out,
                          const femo_mat* A, const femo_vec* rb);
/* ... and what such a solve returns when it does not iterate from the zero guess: x_i = b_i / diag_i on the identity rows of A's
 * last assembly, 0 elsewhere (one small launch, no synchronisation).                                                    */
int     femo_mat_identity_solve(const femo_mat* A, const femo_vec* b, femo_vec* x);

/* ---- host memory of the array boundary (csrc/hostmem.cpp) -------------------------------------
 * The CSDL operators exchange NumPy arrays with the FE layer in every method (state_model.py:81-84,
 * 94-96, 123-124, 171-172; output_model.py:70-72, 78-80); here each exchange is a PCIe transfer.
 *   - femo_host_alloc / femo_host_free: pinned blocks (recycled, kept pinned).  Transfers from / to
 *     them are one DMA; from / to other (pageable) memory the library pipelines 8 MiB chunks through
 *     pinned staging slots with a pool of host threads.
 *   - femo_host_register / _unregister: pin a caller-owned range in place (e.g. a backend's state vector).
 *   - provenance: a pinned block filled by femo_vec_get_host, or uploaded by femo_vec_set_host,
 *     from its base address is remembered as an exact copy of that device vector.  Sending it again
 *     (to the same or another vector) moves no PCIe bytes while the source vector is unchanged.
 *     Whoever writes to such a block on the host MUST call femo_host_touch first/afterwards; the
 *     library's own host writers do.  FEMO_HOST_VERIFY=1 (environment) checks every elided upload.
 *   - femo_host_copy / femo_host_axpby: y = x, y = a x + b y on the library's host threads, for
 *     drivers that keep their variables in pinned blocks (touch the destination themselves).
 *     femo_host_axpby(a, x, 0, y) with x a block that mirrors a device vector records y as a times that
 *     vector (femo_vec_set_host from y is then a device-side scale), and queues the host pass behind a
 *     copy-out of x that is still in flight instead of waiting for it (y is then in flight itself).   */
typedef struct femo_host_stats {
  int64_t h2d_pinned, h2d_pinned_bytes;     /* one DMA from a pinned block              */
  int64_t h2d_staged, h2d_staged_bytes;     /* pageable source through the staging ring */
  int64_t h2d_skipped, h2d_skipped_bytes;   /* elided: the vector already held the data */
  int64_t h2d_as_d2d, h2d_as_d2d_bytes;     /* elided: copied from another device vector */
  int64_t d2h_pinned, d2h_pinned_bytes;
  int64_t d2h_staged, d2h_staged_bytes;     /* pageable destination, or accumulate      */
  int64_t d2h_async, d2h_async_bytes;       /* femo_vec_get_host_async                   */
  int64_t d2h_device_sum, d2h_device_sum_bytes; /* accumulate formed on the device      */
  int64_t h2d_deferred, h2d_deferred_bytes; /* of the h2d_pinned ones: femo_vec_set_host_deferred, on the copy stream */
} femo_host_stats;
int femo_host_alloc(int64_t bytes, void** out);
int femo_host_free(void* p);
int femo_host_trim(void);                    /* release the recycled blocks            */
int femo_host_register(void* p, int64_t bytes);
int femo_host_unregister(void* p);
int femo_host_touch(void* p);                /* the block containing p was written on the host */
int femo_host_is_pinned(const void* p, int64_t bytes);
int femo_host_threads(void);
int femo_host_fill(double* p, int64_t n, double value);   /* p[:] = value; at the base of a pinned block the library remembers
                                                            * it as that constant: sending it to a vector is a device-side fill */
int femo_host_copy(double* dst, const double* src, int64_t n);
int femo_host_axpby(int64_t n, double a, const double* x, double b, double* y);
/* Asynchronous results: femo_vec_get_host_async returns before the bytes have landed when `host` lies in a
 * pinned block (otherwise it is femo_vec_get_host).  Library entry points wait by themselves; code that reads the
 * block on its own calls femo_host_wait(p) (one block) or femo_host_sync() (all) first.                       */
int femo_host_wait(const void* p);
int femo_host_sync(void);
int femo_host_get_stats(femo_host_stats* out);
int femo_host_reset_stats(void);

/* ---- mesh -----------------------------------------------------------------
 * x: (n_vert, tdim) row-major; conn: (n_cell, tdim+1).  n_rows <= n_vert is
 * the number of *owned* vertices (rows of every operator); vertices
 * [n_rows, n_vert) are ghosts (single-GPU: n_rows == n_vert).  Builds the
 * vertex->cell incidence and the sparsity pattern once (dolfinx create_matrix /
 * sparsity pattern [ext], utils_dolfinx.py:385).                              */
int femo_mesh_create(femo_ctx* ctx, int tdim, int64_t n_vert, int64_t n_rows,
                     const double* x, int64_t n_cell, const int32_t* conn,
                     femo_mesh** out);
int femo_mesh_destroy(femo_mesh* mesh);
int femo_mesh_info(const femo_mesh* mesh, int64_t info[FEMO_MESH_INFO_COUNT]);
/* mask[c] bit k set <=> the facet of cell c opposite its local vertex k is an exterior facet
 * (the `ds` measure of the reference's boundaryResidual, run_nonlinear_poisson_opt.py:98-117).
 * NULL clears it.                                                              */
int femo_mesh_set_boundary_facets(femo_mesh* mesh, const uint8_t* mask);
/* Partitioned meshes: bounding box (tdim doubles each) and vertex count of the WHOLE mesh, so
 * that every rank builds the same preconditioner lattice.  Call before the first BPX solve.  */
int femo_mesh_set_global(femo_mesh* mesh, const double* lo, const double* hi, int64_t n_vert_global);
/* Preconditioner lattice of the mesh (built on first use): levels and nodes of the finest one. */
int femo_mesh_pc_info(const femo_mesh* mesh, int32_t* n_levels, int64_t* finest_nodes);
/* CSR view of the pattern (rowptr: n_rows+1, col: NNZ; sorted columns).       */
int femo_mesh_pattern_csr(const femo_mesh* mesh, int64_t* rowptr, int32_t* col);

/* Host-only topology build (no GPU touched): fills the same arrays that
 * femo_mesh_create uploads.  Used by the CPU test-suite.  Buffers may be NULL
 * to query sizes via info[].                                                  */
int femo_topology_build_host(int tdim, int64_t n_vert, int64_t n_rows, int64_t n_cell,
                             const int32_t* conn, int64_t info[FEMO_MESH_INFO_COUNT],
                             int64_t* rowptr, int32_t* col);

/* Host-only plan of the BPX preconditioner lattice for `n_rows` owned vertices (no GPU touched; CPU
 * test-suite): levels and bins per axis (bins[3*l + k], coarsest level first), packed lattice
 * coordinates pk[n_rows*2] (8 B per vertex; 2-D: two words bin << 20 | 20-bit fraction, 3-D: one 64-bit word of three
 * 21-bit fields bin << 12 | 12-bit fraction), the (brick, bin) sort perm[n_rows] with
 * brick_ptr[n_bricks+1], brick_base[3*n_bricks], bin_ptr[65*n_bricks].  Array arguments may be NULL
 * (first call: sizes).                                                                           */
int femo_pc_plan_host(int dim, int64_t n_rows, const double* x, const double* lo, const double* hi,
                      int64_t n_vert_global, int32_t* n_levels, int32_t* bins, int64_t* n_bricks,
                      uint32_t* pk, int32_t* perm, int64_t* brick_ptr, int32_t* brick_base, uint32_t* bin_ptr);

/* ---- Dirichlet set (fea_dolfinx.py:169-176; dolfinx dirichletbc [ext]) ---- */
int femo_bc_create(femo_mesh* mesh, int64_t n, const int32_t* dofs, const double* vals,
                   femo_bc** out);
int femo_bc_destroy(femo_bc* bc);

/* ---- assembly -------------------------------------------------------------
 * params: up to 8 doubles of form constants (unused for POISSON; NL_POISSON: params[0] = Nitsche
 *         penalty beta (0 = none), params[1] = sgn of the nitsche_2 term: +1 symmetric (default), -1 unsymmetric).  aux: extra CG1 field of the form or NULL (NL_POISSON: the boundary data u_exact).
 *         FEMO_PDE_NL_POISSON = POISSON + int u^3 v (run_nonlinear_poisson_opt.py:88-96) and, when
 *         femo_mesh_set_boundary_facets was called, the symmetric Nitsche terms of :98-117.
 * residual: state_model.py:85 assembleVector(residual_form) -> utils:175-179; NO BCs.
 * jacobian: state_model.py:132 assembleMatrix(dR_du) (bc == NULL) and
 *           state_model.py:149 assembleSystem(dR_du, res, bcs) -> utils:189-202
 *           (bc != NULL: rows and columns of the set zeroed, diagonal 1).
 * dRdf:     state_model.py:141 assembleMatrix(derivative(res, f)); stored
 *           cell-major as (n_cell, tdim+1) values aligned with conn, i.e. the
 *           CSC of the N x n_cell matrix (column c has rows conn[c,:]).        */
int femo_assemble_residual(femo_mesh* mesh, int pde, const double* params,
                           const femo_vec* u, const femo_vec* f, const femo_vec* aux, femo_vec* r);
int femo_mat_create(femo_mesh* mesh, femo_mat** out);
int femo_mat_destroy(femo_mat* A);
int femo_assemble_jacobian(femo_mesh* mesh, int pde, const double* params,
                           const femo_vec* u, const femo_vec* f, const femo_vec* aux,
                           const femo_bc* bc, femo_mat* J);
int femo_assemble_dRdf(femo_mesh* mesh, int pde, const double* params,
                       const femo_vec* u, const femo_vec* f, femo_vec* vals);
/* The same matrix for the forms whose column c has ONE value on all its rows (Poisson-type residuals with a DG0 source:
 * dR_i/df_c = -int_c phi_i = -|T_c|/(d+1), state_model.py:136-146): cvals holds n_cell values instead of
 * (d+1) n_cell, and femo_dRdf_cell_apply is femo_dRdf_apply on that compact form (a quarter of the bytes).          */
int femo_assemble_dRdf_cell(femo_mesh* mesh, int pde, const double* params, femo_vec* cvals);
int femo_dRdf_cell_apply(femo_mesh* mesh, const femo_vec* cvals, int transpose,
                         const femo_vec* x, femo_vec* y, int accumulate);
/* One pass over the mesh for any subset of: J_nobc (state_model.py:132), A_bc
 * (state_model.py:149) and the Newton right-hand side  rhs = F + K[:,bc](g-u),
 * rhs[bc] = u-g  (dolfinx NonlinearProblem.F/J [ext], utils_dolfinx.py:431).
 * Unused outputs are NULL.  Results are identical to the separate calls.       */
int femo_assemble_system(femo_mesh* mesh, int pde, const double* params,
                         const femo_vec* u, const femo_vec* f, const femo_vec* aux,
                         const femo_bc* bc, femo_mat* J_nobc, femo_mat* A_bc, femo_vec* rhs);
/* b[bc] = u[bc] - g  (dolfinx set_bc(b, bcs, x, -1.0) [ext]).                    */
int femo_bc_apply_rhs(const femo_bc* bc, const femo_vec* u, femo_vec* b);
/* Newton right-hand side with Dirichlet lifting, dolfinx NonlinearProblem.F
 * [ext] as driven by utils_dolfinx.py:431:  b = F + K[:,bc](g-u); b[bc] = u-g. */
int femo_newton_rhs(const femo_mat* K_nobc, const femo_vec* F, const femo_vec* u,
                    const femo_bc* bc, femo_vec* b);

/* ---- operator application (state_model.py:161-200) -------------------------
 * mat_spmv:   utils_dolfinx.py:256-264 (A*x) / :275-287 (A^T*R).
 * dRdf_apply: transpose=1: y (n_cell) = dRdf^T x (n_vert); transpose=0:
 *             y (n_rows) = dRdf x (n_cell).  accumulate=1 adds into y.         */
int femo_mat_spmv(const femo_mat* A, int transpose, const femo_vec* x, femo_vec* y);
int femo_dRdf_apply(femo_mesh* mesh, const femo_vec* vals, int transpose,
                    const femo_vec* x, femo_vec* y, int accumulate);
int femo_mat_export_csr(const femo_mat* A, int64_t* rowptr, int32_t* col, double* val);
int femo_mat_diagonal(const femo_mat* A, femo_vec* d);

/* S = diag(A)^-1/2 and the scaled copy S A S the Krylov loops iterate on, formed NOW instead of at the start of the first
 * solve with A (collective on a partitioned mesh: the ghost entries of S come from their owners).  For callers that have
 * idle time before the solve -- the operator layer assembles and scales the adjoint system of a linear form while the
 * input of the forward solve is still on its way to the device (state_model.py:117-158 reordered, same work).           */
int femo_mat_prescale(femo_mat* A);
/* ---- linear solve (fea_dolfinx.py:192-222; utils_dolfinx.py:476-512) --------
 * Jacobi-preconditioned CG on A (transpose=0) or A^T (transpose=1).  The
 * reference factorises with MUMPS; CG+Jacobi is the BASELINE.json design.
 * A must be symmetric positive definite (Poisson with Dirichlet elimination).  */
int femo_solve_cg(const femo_mat* A, int transpose, const femo_vec* b, femo_vec* x,
                  const femo_solver_opts* opts, femo_solve_info* info);

/* z = M^-1 r with the BPX preconditioner femo_solve_cg uses for A (pc = FEMO_PC_BPX), in unscaled
 * variables: M^-1 = D^-1 + 0.6 sum_l P_l C_l P_l^T (csrc/bpx.hip; restated in oracle/bpx_oracle.py).
 * For callers that drive their own Krylov loop, and for the parity tests.                         */
int femo_mat_pc_apply(const femo_mat* A, const femo_vec* r, femo_vec* z);

/* BiCGSTAB on A or A^T for non-symmetric operators (e.g. unsymmetric Nitsche terms), Jacobi
 * preconditioning by symmetric diagonal scaling; stops on ||S r||_2 <= max(rtol ||S b||_2, atol),
 * S = diag(A)^-1/2.  Same options / info as femo_solve_cg.                                      */
int femo_solve_bicgstab(const femo_mat* A, int transpose, const femo_vec* b, femo_vec* x,
                        const femo_solver_opts* opts, femo_solve_info* info);

/* ---- scalar output and its partials (output_model.py:69-87) ---------------- */
int femo_functional_value(femo_mesh* mesh, int kind, const double* params,
                          const femo_vec* u, const femo_vec* f, const femo_vec* u_d,
                          double* value);
int femo_functional_grad_u(femo_mesh* mesh, int kind, const double* params,
                           const femo_vec* u, const femo_vec* f, const femo_vec* u_d,
                           femo_vec* g);
int femo_functional_grad_f(femo_mesh* mesh, int kind, const double* params,
                           const femo_vec* u, const femo_vec* f, const femo_vec* u_d,
                           femo_vec* g);

/* ---- DG0 field expressions for projected outputs (fea_dolfinx.py:148-161) --------
 * kind 0: out_c = |grad in| (in: CG1, n_vert);  kind 1: out_c = in_c ** params[0] (in: DG0). */
int femo_cell_expression(femo_mesh* mesh, int kind, const double* params,
                         const femo_vec* in, femo_vec* out);
/* y_i = x_i / d_i  (Vec.pointwiseDivide, utils_dolfinx.py:566)                       */
int femo_vec_pointwise_divide(femo_vec* y, const femo_vec* x, const femo_vec* d, int64_t n);

/* ---- measurement helper ------------------------------------------------------
 * Launches `reps` SpMVs of A on x bracketed by HIP events on the ctx stream and
 * returns the mean device milliseconds per launch (bench.py roofline leg).     */
int femo_bench_spmv(const femo_mat* A, const femo_vec* x, femo_vec* y, int reps,
                    double* ms_per_launch);

/* ---- multi-GPU (new design; the reference is single-rank) ----
 * One process per GPU.  unique id = ncclUniqueId (128 bytes) created on rank 0
 * and broadcast by the host (torch.distributed).  After comm_init every dot
 * product inside femo_solve_cg / femo_vec_dot / femo_functional_value is
 * all-reduced over xGMI; femo_mesh_set_halo installs the ghost exchange plan:
 * for neighbour k, send_idx[send_ptr[k]:send_ptr[k+1]] are owned local indices
 * whose values go to rank nbr[k]; values received from nbr[k] land in local
 * ghost slots [n_rows + recv_ptr[k], n_rows + recv_ptr[k+1]).                  */
int femo_comm_unique_id(char id[128]);
int femo_comm_init(femo_ctx* ctx, const char id[128], int rank, int nranks);
int femo_comm_rank(const femo_ctx* ctx, int* rank, int* nranks);
int femo_mesh_set_halo(femo_mesh* mesh, int n_nbr, const int32_t* nbr,
                       const int64_t* send_ptr, const int32_t* send_idx,
                       const int64_t* recv_ptr);
int femo_halo_exchange(femo_mesh* mesh, femo_vec* x);
int femo_allreduce_sum(femo_ctx* ctx, double* host_inout, int n);

/* ---- device-initiated ghost refresh (ABI 9; new design -- the reference's ghost updates are implicit PETSc scatters,
 * utils_dolfinx.py:167,200 [ext]; SURVEY.md section 5 "distributed comm backend": IPC-mapped peer buffers over xGMI) ----
 * Instead of ncclSend/ncclRecv, a rank's producer kernels store the values a neighbour needs straight into that
 * neighbour's INBOX (uncached device memory the neighbour exported) and bump a counter there; consumers wait on their own
 * counters.  Two kernels of the compute stream per refresh -- none where the producer is the kernel that computes the
 * values anyway (the merged BPX-PCG's prolongation).  Set-up, collective over the ranks that share `mesh`'s halo plan
 * (femo_amd/dist does it through the control plane):
 *   1. every rank: femo_mesh_halo_direct_export -> 64-byte hipIpcMemHandle (other processes) / device address (same
 *      process), workgroups per producer launch;
 *   2. ranks exchange {handle, address, workgroups, their neighbour list and recv_ptr};
 *   3. every rank: femo_mesh_halo_direct_connect, per neighbour k of its plan: the neighbour's handle / address
 *      (mode 0 = handles of other processes, 1 = addresses in this process), remote_offset[k] = the neighbour's recv_ptr
 *      entry for THIS rank, remote_n_ghost[k] = its ghost count, remote_slot[k] = this rank's index in ITS neighbour
 *      list, remote_blocks[k] = its workgroups per producer launch;
 *   4. every rank: femo_mesh_halo_direct_selftest (one exchange of a known pattern through the solver's device code);
 *   5. the ranks reduce the results; femo_mesh_halo_direct_enable(mesh, all passed) -- all or none.
 * Without an enabled plan femo_halo_exchange and the solvers use ncclSend/ncclRecv as before.                        */
int femo_mesh_halo_direct_export(femo_mesh* mesh, char ipc_handle[64], uint64_t* address, int32_t* n_blocks);
int femo_mesh_halo_direct_connect(femo_mesh* mesh, int mode, const char* handles, const uint64_t* addresses,
                                  const int64_t* remote_offset, const int64_t* remote_n_ghost,
                                  const int32_t* remote_slot, const int32_t* remote_blocks);
int femo_mesh_halo_direct_selftest(femo_mesh* mesh, int* ok);
int femo_mesh_halo_direct_enable(femo_mesh* mesh, int on);
/* out = {enabled, exchanges issued, consumer time-outs seen, workgroups per producer launch} */
int femo_mesh_halo_direct_info(femo_mesh* mesh, int64_t out[4]);

/* Blocking waits of the host on the device (hipStreamSynchronize / hipEventSynchronize / hipDeviceSynchronize) the library
 * has executed in this process since the last reset -- each costs an idle device for one host round trip.  bench.py
 * divides by the cycle count (`host_syncs_per_step`).  (The reference synchronises implicitly in every PETSc / dolfinx
 * call: it has no asynchronous path, utils_dolfinx.py:155-212.)                                                      */
int femo_host_sync_stats(int64_t* count, int reset);

/* Collectives issued on this context since the last reset: out = {all-reduce calls, doubles all-reduced, neighbour
 * exchanges, doubles sent}.  What `bench.py`'s scaling record and the tests divide by the CG iteration count (the
 * reference's only collective is the ghost update of utils_dolfinx.py:32,236).                                    */
int femo_comm_stats(femo_ctx* ctx, int64_t out[4], int reset);

/* ---- Reissner-Mindlin shell (examples/test_shell_m3l/shell_pde.py:219-332) ----------
 * State w = (u_mid in CG2^3, theta in CG1^3) on flat triangular facets: 3 dofs per P2 node (vertices [0, n_vert), then
 * edge midpoints), then 3 per vertex; thickness h and surface load f are CG1 fields at the vertices (shell_pde.py:
 * 228-231).  Formulation: oracle/shell_oracle.py (membrane + bending + shear + drilling energies, ElasticModel of the
 * un-vendored shell_analysis_fenicsx restated from its published source, pinned by the Scordelis-Lo roof).
 * The host side (femo_amd/fea/shell.py) numbers the edges and builds the CSR pattern of the element couplings:
 *   cell_edges[c][k] = edge of local edge k = (0,1), (1,2), (2,0);   rowptr / cols: pattern over the n_dof dofs;
 *   elem_pos[c][27 i + j] = CSR position of the coupling of local dofs i, j (local order: 6 displacement nodes x 3
 *   components, then 3 rotation nodes x 3).
 * Matrices are plain value arrays (femo_vec of nnz entries) on that pattern.                                        */
int femo_shell_create(femo_ctx* ctx, int64_t n_vert, const double* x, int64_t n_cell, const int32_t* conn, int64_t n_edge,
                      const int32_t* cell_edges, const int64_t* rowptr, const int32_t* cols, const int32_t* elem_pos,
                      femo_shell** out);
int femo_shell_destroy(femo_shell* s);
int64_t femo_shell_ndof(const femo_shell* s);
int64_t femo_shell_nnz(const femo_shell* s);
/* K(h): stiffness of the elastic energy (pdeRes / elasticEnergy, shell_pde.py:246-253); dR/dw of the linear residual */
int femo_shell_assemble(femo_shell* s, double E, double nu, const femo_vec* h, femo_vec* vals);
/* y = K x; fixed_dev != NULL (device array of n_dof bytes): the operator with strongly imposed dofs as identity rows/columns */
int femo_shell_matvec(femo_shell* s, const femo_vec* vals, const uint8_t* fixed_dev_or_null, const femo_vec* x, femo_vec* y);
/* F (+)= sign * int f . v (weakFormResidual's load term) and its transpose: out (+)= sign * (dF/df)^T lambda */
int femo_shell_load(femo_shell* s, const femo_vec* f, double sign, int accumulate, femo_vec* F);
int femo_shell_load_T(femo_shell* s, const femo_vec* lam, double sign, int accumulate, femo_vec* out);
/* out_b (+)= v^T (dK/dh_b) w -- (dR/dh)^T lambda for R = K(h) w - F with v = lambda; *energy = 1/2 v^T K w (shell_pde.py:299-302) */
int femo_shell_dform_dh(femo_shell* s, double E, double nu, const femo_vec* h, const femo_vec* v, const femo_vec* w,
                        int accumulate, femo_vec* out, double* energy);
/* y (+)= (dK/dh [dh]) w: the FORWARD product with the thickness partial of the elastic residual (the fwd branch of
 * compute_jacvec_product, state_model.py:176-188); the exact transpose of femo_shell_dform_dh: <v, y> = <dh, out>.         */
int femo_shell_dform_dh_fwd(femo_shell* s, double E, double nu, const femo_vec* h, const femo_vec* dh, const femo_vec* w,
                            int accumulate, femo_vec* y);
/* 1/2 int u_mid . u_mid (shell_pde.py:287-288) and its gradient; int rho h (shell_pde.py:293-294) and its gradient */
int femo_shell_compliance(femo_shell* s, const femo_vec* w, double* value, int accumulate, femo_vec* grad);
int femo_shell_mass(femo_shell* s, double rho, const femo_vec* h, double* value, int accumulate, femo_vec* grad);
/* The compliance over a tagged subset of cells -- the `dxx` measure the shell drivers pass (shell_pde.py:66,284-285:
 * dx_2(10)): cell_weight is a DG0 indicator (n_cell doubles; NULL: the whole surface).                                 */
int femo_shell_compliance_dx(femo_shell* s, const femo_vec* w, const femo_vec* cell_weight, double* value, int accumulate, femo_vec* grad);
/* Thickness regularisation of `compliance` (shell_pde.py:262-282, ShellPDE.regularization): kind 1 'H1', 2 'L2H1',
 * 3 'L2' (alpha1 = 1e3, alpha2 = 1, h_mesh = CellDiameter); value and/or grad (+)= d/dh.  femo_shell_hpower: int coef h^p dx
 * with the degree-4 rule -- the thickness term of pnorm_stress(regularization=True), shell_pde.py:307-309.              */
int femo_shell_regularization(femo_shell* s, int kind, const femo_vec* h, double* value, int accumulate, femo_vec* grad);
int femo_shell_hpower(femo_shell* s, double coef, double p, const femo_vec* h, double* value, int accumulate, femo_vec* grad);
/* Penalty form of the boundary conditions, `pdeRes(..., penalty=True, dss, dSS, g)` (shell_pde.py:34,59-61,246-253 ->
 * ElasticModel.weakFormResidual of the un-vendored shell_analysis_fenicsx; restated in oracle/shell_oracle.py::
 * penalty_matrix in the idiom of the tree's other penalty terms, run_poisson_opt.py:60, motor_pde.py:177-178):
 *     R_pen(dw) = sum over tagged edges of coef_e int_e (w - g) . dw,    coef_e = beta (1/h_E('+') + 1/h_E('-')),
 * all six fields of w, exterior (ds) and interior (dS) facets alike.  The host passes per tagged edge its three
 * displacement nodes (end vertices, midpoint node), coef_e |e| and the CSR positions of its 3 x (9 + 4) entries
 * (component-major; 9 displacement pairs row-major over (v0, v1, mid), then 4 rotation pairs over (v0, v1)).
 * femo_shell_penalty_add: vals += K_pen (after femo_shell_assemble);  femo_shell_penalty_apply: y (+)= K_pen (x - g).   */
int femo_shell_set_penalty(femo_shell* s, int64_t n_edges, const int32_t* edge_nodes, const double* coef, const int32_t* pos);
int femo_shell_penalty_add(femo_shell* s, femo_vec* vals);
int femo_shell_penalty_apply(femo_shell* s, const femo_vec* x, const femo_vec* g, int accumulate, femo_vec* y);
/* Inertial residual, `kinetic_residual(rho, h)` (shell_pde.py:255-256 -> ElasticModel.inertialResidual [absent package];
 * its use with accelerations: run_aeroelasticity_dynamic.py:93): y (+)= M(h) acc with
 * M = int rho h N_a N_b (displacements) + int rho h^3/12 phi_a phi_b (rotations), consistent, degree-4 rule;
 * femo_shell_inertia_dh: out_b (+)= lam^T (dM/dh_b) acc, its thickness partial transposed.                             */
int femo_shell_inertia_apply(femo_shell* s, double rho, const femo_vec* h, const femo_vec* acc, int accumulate, femo_vec* y);
int femo_shell_inertia_dh(femo_shell* s, double rho, const femo_vec* h, const femo_vec* lam, const femo_vec* acc, int accumulate, femo_vec* out);
/* y (+)= (dM/dh [dh]) acc: the same for the inertial residual.                                                           */
int femo_shell_inertia_dh_fwd(femo_shell* s, double rho, const femo_vec* h, const femo_vec* dh, const femo_vec* acc, int accumulate, femo_vec* y);
/* L2 projection of the von Mises stress onto CG1 (shell_pde.py:315-332; the field output of the shell drivers,
 * shell_dynamic_pde.py:82-83,129): rhs_i = int sigma_vm phi_i, lumped_i = row sum of the P1 mass matrix (may be NULL);
 * femo_shell_p1_mass: y = M x with that mass matrix (the host side runs Jacobi-CG with it, utils_dolfinx.py:549-583). */
int femo_shell_vm_rhs(femo_shell* s, double E, double nu, const femo_vec* h, const femo_vec* w, double surface, femo_vec* rhs,
                      femo_vec* lumped);
int femo_shell_p1_mass(femo_shell* s, const femo_vec* x, femo_vec* y);
/* J = 1 / alpha int (m sigma_vm)^rho dx: the aggregated von Mises stress of the shell drivers (shell_pde.py:297-313,
 * `pnorm_stress`; sigma(z) = C (eps + z kappa) at z = surface * h / 2, surface = +1 top, 0 mid, -1 bottom as in
 * shell_pde.py:315-328).  value and/or partials: grad_w (+)= dJ/dw (n_dof), grad_h (+)= dJ/dh (n_vert).             */
int femo_shell_pnorm_stress(femo_shell* s, double E, double nu, const femo_vec* h, const femo_vec* w, double m, double rho, double alpha,
                            double surface, double* value, int accumulate, femo_vec* grad_w, femo_vec* grad_h);
/* Lattice preconditioner for femo_shell_solve (opts->pc = 1): M^-1 = D^-1 + sum_l P_l C_l P_l^T, P_l = trilinear
 * interpolation from nested lattices over the bounding cube (2, 4, ... cells per axis) to the dof nodes, per component,
 * C_l = 1 / diag(P_l^T K P_l) (recomputed on the device when K or the Dirichlet set change).  The host passes
 *   ell_idx / ell_w   P of every level: 8 (lattice unknown, weight) pairs per level and dof (width = 8 n_levels);
 *                     lattice unknown = 6 * node + field, nodes numbered level by level (level_offsets, n_levels + 1)
 *   pt_*              the finest level's P^T as CSR over all 6 n_nodes unknowns
 *   par_* / chi_*     node-level transfers between consecutive lattices: parents (<= 8) and children (<= 27) of a node. */
int femo_shell_pc_create(femo_shell* s, int width, int64_t n_nodes, int n_levels, const int64_t* level_offsets,
                         const int32_t* ell_idx, const double* ell_w,
                         const int64_t* pt_rowptr, const int32_t* pt_cols, const double* pt_vals,
                         const int64_t* par_rowptr, const int32_t* par_cols, const double* par_vals,
                         const int64_t* chi_rowptr, const int32_t* chi_cols, const double* chi_vals);
/* Exact coarse solve for the lattice preconditioner (optional, after femo_shell_pc_create): on lattice level `level`
 * (not the finest; 6 x nodes <= 8192 unknowns) the Galerkin operator P^T K P is formed as a dense matrix on the device,
 * factorised by the library's own blocked Cholesky + triangular inverse on the fp64 matrix cores (shell.hip) whenever
 * the stiffness or the Dirichlet set change, and A^-1 = L^-T L^-1 applied in place
 * of the diagonal levels 0 .. level: M^-1 = D^-1 + sum_{l > level} P_l C_l P_l^T + P_c (P_c^T K P_c)^-1 P_c^T.  What
 * the reference's direct solver (MUMPS, utils_dolfinx.py:476-512) does for the whole matrix is done here for the
 * ~3000 unknowns that carry the smooth, nearly inextensional modes a diagonal cannot see.
 *   node_xyz   int32[3 x nodes]: lattice coordinates of the level's nodes (level-local numbering)
 *   item_*     work items of the Galerkin kernel: points (first dof / 3) of one coarse cell and one field group, any
 *              number each, 256 by default (item_ptr n_items + 1, item_pts), and the level-local numbers of the 4 x 4 x 4 nodes around
 *              the cell (item_nbr 64 per item, x fastest, -1 where the surface does not touch the lattice)
 *   down_*     optional (NULL: level by level): composite restriction from the finest lattice's nodes to the nodes of
 *              levels `level` .. n_levels - 2 as CSR (rows in node order, columns = node numbers on the finest lattice):
 *              one launch per iteration in place of n_levels - 1 - level                                             */
int femo_shell_pc_coarse(femo_shell* s, int level, const int32_t* node_xyz, int64_t n_items, const int64_t* item_ptr,
                         const int32_t* item_pts, const int32_t* item_nbr, const int64_t* down_rowptr, const int32_t* down_cols,
                         const double* down_vals);
/* For tests: the dense coarse operator (inverse = 0) or the factors of its inverse (1: L^-T above, L^-1 below the
 * diagonal) for `vals` and the mask, row-major n x n into `out` (host; NULL: only *n_out).                           */
/* Hermite-type lattice spaces on the hierarchy of femo_shell_pc_create / femo_shell_pc_coarse (round 4): the nodal rotations
 * of a lattice act as the slopes of its displacement interpolation, u = sum_n [alpha_n U_n + Theta_n x sigma_n], so coarse
 * lattices reproduce bending modes instead of locking on them (the solve that stands where the reference factorises with
 * MUMPS, shell_pde.py:246-253, needs about half the iterations).  fin_w4: (alpha, sigma) per (point, corner) of the finest
 * lattice ((w, 0, 0, 0) for rotation points); hp_*: P_L^T by finest node -- row 2 k lists the displacement points of node k
 * (first dof, w4), row 2 k + 1 its rotation points; par_w5 / chi_w5: (a, b, c) per entry of the parent / child CSR;
 * lvl_w4: composed weights of the levels above the coarse solve, [level][point][8][4]; cs_w4: of the coarse-solve level;
 * down_*: composite restriction finest lattice -> levels cs .. L - 2 in (A, B, C) form (optional).  On a partitioned shell: the
 * rows of the rank's local points on the global lattice (the Galerkin sums are all-reduced like the trilinear ones).       */
int femo_shell_pc_hermite(femo_shell* s, const float* fin_w4, const int64_t* hp_rowptr, const int32_t* hp_cols, const float* hp_w4,
                          const double* par_w5, const double* chi_w5, const float* lvl_w4, const float* cs_w4,
                          const int64_t* down_rowptr, const int32_t* down_cols, const double* down_w5);
/* What the device side runs with: out = {Hermite-type data uploaded, bit 0: Hermite-type spaces enabled (not fallen back) |
 * bit 1: in use for the last stiffness set up, coarse solve factorised, node blocks ready}.  The library falls back to the trilinear hierarchy (once, with a warning
 * on stderr) when the Hermite-type coarse operator cannot be factorised; reports and pinned iteration counts read this.  */
int femo_shell_pc_info(const femo_shell* s, int32_t out[4]);
/* Items of the node-block set-up for the Hermite-type spaces (optional; without it the blocks are formed row by row, 10.5
 * instead of ~2 ms at 1.97 M dofs): for every level above the coarse solve (item_lvl = level - cs_level - 1) the points grouped by
 * the lattice cell that contains them, at most 64 per item (item_ptr into item_pts; every point once per level);
 * pcell[level][point] = the cell's coordinates packed as x | y << 10 | z << 20.                                        */
int femo_shell_pc_block_items(femo_shell* s, int64_t n_items, const int64_t* item_ptr, const int32_t* item_lvl, const int32_t* item_pts,
                              const int32_t* pcell);
/* Weights of the additive parts of the lattice preconditioner: w_levels multiplies the node-block corrections of the levels
 * between the coarse solve and the finest lattice (default 0.3: the overlapping levels overshoot when summed with weight 1,
 * like the Poisson BPX's theta; 145 -> 105 iterations on the 1.97 M-dof roof), w_coarse the exact coarse solve (default 1).
 * Both must be positive.  Takes effect at the next set-up (the next solve).                                               */
int femo_shell_pc_weights(femo_shell* s, double w_levels, double w_coarse);
/* z = M^-1 r of the lattice preconditioner for the stiffness `vals` and the mask (set up if needed): what the parity tests
 * compare with the oracle's operator.  One rank.                                                                      */
int femo_shell_pc_apply(femo_shell* s, const femo_vec* vals, const uint8_t* fixed_host, const femo_vec* r, femo_vec* z);
int femo_shell_pc_coarse_matrix(femo_shell* s, const femo_vec* vals, const uint8_t* fixed_host, int inverse, double* out,
                                int64_t* n_out);
/* K x = b with x = xfix on the dofs flagged in fixed_host (n_dof bytes; NULL: none; xfix NULL: zero values).
 * PCG, sqrt(r.M^-1 r) <= max(rtol sqrt(r0.M^-1 r0), atol), opts->pc: 0 Jacobi, 1 lattice; K is symmetric, so the
 * adjoint solve (fea_dolfinx.py:208-222) is the same call.  The reference uses MUMPS (utils_dolfinx.py:476-512).
 * info->converged: 1 tolerance met, 2 stalled at the attainable accuracy (residual below 1e-9 of the initial one in the
 * preconditioner's norm and no progress over 8 batches of check_every iterations), 0 max_it, -1 breakdown (NaN).      */
int femo_shell_solve(femo_shell* s, const femo_vec* vals, const uint8_t* fixed_host, const femo_vec* xfix, const femo_vec* b,
                     femo_vec* x, const femo_solver_opts* opts, femo_solve_info* info);
/* The shell on the N ranks of a context (femo_comm_init; the reference is single-rank, SURVEY.md section 0.3 -- the
 * partitioning of section 8(e) applied to the shell).  Each rank creates its handle on the cells that touch a point it
 * owns (a point: a P2 node with its three displacements or a vertex with its three rotations, dofs 3 p .. 3 p + 2),
 * builds the lattice arrays of femo_shell_pc_create / _pc_coarse on the GLOBAL lattice (same nodes on every rank, the
 * rows of its own points), then declares the partition:
 *   owned_points  uint8[n_dof / 3]: 1 for the points this rank owns
 *   nbr           the ranks it exchanges with; segment k of send_dofs (send_ptr) lists the owned dofs rank nbr[k] holds
 *                 copies of, segment k of recv_dofs (recv_ptr) the local dofs of the points rank nbr[k] owns, in the order
 *                 that rank sends them.
 * From then on femo_shell_assemble / _penalty_add leave the rows of points owned elsewhere zero (the rank's share of K:
 * owned rows are complete because every cell around an owned point is local), femo_shell_solve masks the right-hand
 * side the same way, refreshes the search direction on the halo before every product, all-reduces its two scalars, the
 * restricted residual on the finest lattice and -- per stiffness -- the Galerkin blocks and the dense coarse operator,
 * and returns x consistent on all local points.  Values of outputs are all-reduced: integrate over owned cells only
 * (femo_shell_compliance_dx with the ownership indicator as cell weight).
 * femo_shell_halo: x on the points owned elsewhere <- the owners' values.  femo_shell_mask_unowned: x <- 0 there.        */
int femo_shell_set_partition(femo_shell* s, const uint8_t* owned_points, int n_nbr, const int32_t* nbr, const int64_t* send_ptr,
                             const int32_t* send_dofs, const int64_t* recv_ptr, const int32_t* recv_dofs);
/* Partitioned shells: the local cells whose SCALAR outputs (mass, volume, p-norm stress, elastic energy, regularisation and
 * h-power terms; shell_pde.py:262-313) this rank integrates -- one rank per cell of the whole mesh, the values are summed
 * over the ranks.  Gradients are integrated over all local cells (complete on the points the rank owns).                */
int femo_shell_set_owned_cells(femo_shell* s, const uint8_t* owned_cells);
int femo_shell_halo(femo_shell* s, femo_vec* x);
int femo_shell_mask_unowned(femo_shell* s, femo_vec* x);

#ifdef __cplusplus
}
#endif
#endif /* FEMO_HIP_H */
