// C-ABI entry points: handles, host<->device movement, and the small
// elementwise kernels (fill / axpy / Dirichlet lifting).  See include/femo_hip.h
// for the reference call site each entry replaces.
#include <algorithm>

#include <atomic>

#include "femo_internal.h"

std::atomic<long long> femo_host_sync_count{0};

namespace {

__global__ void k_fill(int64_t n, double v, double* __restrict__ x) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) x[i] = v;
}

__global__ void k_axpy(int64_t n, double a, const double* __restrict__ x, double* __restrict__ y) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) y[i] += a * x[i];
}

__global__ void k_sum(int64_t n, const double* __restrict__ a, const double* __restrict__ b, double* __restrict__ out) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) out[i] = a[i] + b[i];
}

__global__ void k_scale(int64_t n, double a, const double* __restrict__ x, double* __restrict__ out) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) out[i] = a * x[i];
}

__global__ void k_pdiv(int64_t n, const double* __restrict__ x, const double* __restrict__ d, double* __restrict__ y) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) y[i] = x[i] / d[i];
}

// vertices of the marked boundary facets (facet k = the one opposite local vertex k)
__global__ void k_boundary_vertices(int64_t n_cell, int tdim, const int32_t* __restrict__ conn,
                                    const uint8_t* __restrict__ bfacets, uint8_t* __restrict__ bv) {
  for (int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; c < n_cell; c += (int64_t)gridDim.x * blockDim.x) {
    const unsigned bits = bfacets[c];
    if (!bits) continue;
    for (int k = 0; k <= tdim; ++k)
      if (bits & (1u << k))
        for (int j = 0; j <= tdim; ++j)
          if (j != k) bv[conn[c * (tdim + 1) + j]] = 1;
  }
}

__global__ void k_or_masks(int64_t n, const uint8_t* __restrict__ a, const uint8_t* __restrict__ b, uint8_t* __restrict__ out) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    out[i] = (a ? a[i] : 0) | (b ? b[i] : 0);
}

__global__ void k_bc_mask(int64_t n, const int32_t* __restrict__ dofs, const double* __restrict__ vals,
                          uint8_t* __restrict__ mask, double* __restrict__ dense) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    mask[dofs[i]] = 1;
    dense[dofs[i]] = vals[i];
  }
}

// rowmask[row]: bit k = the row's k-th off-diagonal column is in the Dirichlet set, bit 63 = the row itself
__global__ void k_bc_rowmask(int64_t n_rows, int64_t n_slices, const int64_t* __restrict__ mptr, const int32_t* __restrict__ cols,
                             const int32_t* __restrict__ rowlen, const uint8_t* __restrict__ mask, uint64_t* __restrict__ out) {
  const int64_t row = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= n_slices * 64) return;
  uint64_t bits = 0;
  if (row < n_rows) {
    const int64_t base = mptr[row >> 6];
    const int lane = (int)(row & 63), len = rowlen[row];
    for (int k = 0; k < len; ++k)
      if (mask[cols[femo_sell_index(base, k, lane)]]) bits |= uint64_t(1) << k;
    if (mask[row]) bits |= uint64_t(1) << 63;
  }
  out[row] = bits;
}

// w = (g - u) on the Dirichlet set, 0 elsewhere (w pre-zeroed)
__global__ void k_bc_lift_vec(int64_t n, const int32_t* __restrict__ dofs, const double* __restrict__ g,
                              const double* __restrict__ u, double* __restrict__ w) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int32_t d = dofs[i];
    w[d] = g[i] - u[d];
  }
}

// b = F + Kw ; then b[bc] = u - g
__global__ void k_add_into(int64_t n, const double* __restrict__ a, double* __restrict__ b) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) b[i] += a[i];
}

__global__ void k_bc_set_rhs(int64_t n, int64_t n_rows, const int32_t* __restrict__ dofs, const double* __restrict__ g,
                             const double* __restrict__ u, double* __restrict__ b) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int32_t d = dofs[i];
    if (d < n_rows) b[d] = u[d] - g[i];
  }
}

// structural zeros of completed regular slices (FemoTopology::real) are not part of the exported pattern
__global__ void k_export_rows(int64_t n_rows, const int64_t* __restrict__ mptr, const int32_t* __restrict__ cols,
                              const int32_t* __restrict__ rowlen, const uint32_t* __restrict__ rowreal, const double* __restrict__ diag,
                              const double* __restrict__ vals, const int64_t* __restrict__ rowptr,
                              int32_t* __restrict__ ocol, double* __restrict__ oval) {
  const int64_t row = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= n_rows) return;
  const int64_t base = mptr[row >> 6];
  const int lane = (int)(row & 63);
  const int len = rowlen[row];
  int64_t o = rowptr[row];
  bool placed = false;
  const uint32_t real = rowreal[row];
  for (int k = 0; k < len; ++k) {
    if (k < 32 && !((real >> k) & 1u)) continue;
    const int64_t e = femo_sell_index(base, k, lane);
    const int32_t c = cols[e];
    if (!placed && c > row) {
      ocol[o] = (int32_t)row; oval[o] = diag[row]; ++o; placed = true;
    }
    ocol[o] = c; oval[o] = vals[e]; ++o;
  }
  if (!placed) { ocol[o] = (int32_t)row; oval[o] = diag[row]; }
}

inline unsigned grid_for(int64_t n) {
  int64_t g = (n + 255) / 256;
  if (g > 4096) g = 4096;
  if (g < 1) g = 1;
  return (unsigned)g;
}

template <class T>
int upload(T** dptr, const std::vector<T>& h, hipStream_t st) {
  const size_t bytes = std::max<size_t>(h.size(), 1) * sizeof(T);
  FEMO_HIP_CHECK(hipMalloc(dptr, bytes + 64));
  if (!h.empty()) FEMO_HIP_CHECK(hipMemcpyAsync(*dptr, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice, st));
  return 0;
}

}  // namespace

int femo_launch_sum(double* out, const double* a, const double* b, int64_t n, hipStream_t st) {
  if (n == 0) return 0;
  hipLaunchKernelGGL(k_sum, dim3(grid_for(n)), dim3(256), 0, st, n, a, b, out);
  FEMO_HIP_CHECK(hipGetLastError());
  return 0;
}

int femo_launch_fill(double* out, double value, int64_t n, hipStream_t st) {
  if (n == 0) return 0;
  hipLaunchKernelGGL(k_fill, dim3(grid_for(n)), dim3(256), 0, st, n, value, out);
  FEMO_HIP_CHECK(hipGetLastError());
  return 0;
}

int femo_launch_scale(double* out, double a, const double* x, int64_t n, hipStream_t st) {
  if (n == 0) return 0;
  hipLaunchKernelGGL(k_scale, dim3(grid_for(n)), dim3(256), 0, st, n, a, x, out);
  FEMO_HIP_CHECK(hipGetLastError());
  return 0;
}

extern "C" {

int femo_abi_version(void) { return FEMO_ABI_VERSION; }

int femo_host_sync_stats(int64_t* count, int reset) {
  FEMO_REQUIRE(count != nullptr, "null argument");
  *count = (int64_t)femo_host_sync_count.load(std::memory_order_relaxed);
  if (reset) femo_host_sync_count.store(0, std::memory_order_relaxed);
  return 0;
}

int femo_device_count(int* n) {
  FEMO_REQUIRE(n != nullptr, "null argument");
  hipError_t e = hipGetDeviceCount(n);
  if (e != hipSuccess) {
    *n = 0;
    femo_set_error("hipGetDeviceCount: %s", hipGetErrorString(e));
    return 1;
  }
  return 0;
}

// ------------------------------------------------------------------ ctx -----
int femo_ctx_create(int device_id, void* stream, femo_ctx** out) {
  FEMO_REQUIRE(out != nullptr, "null argument");
  *out = nullptr;
  int ndev = 0;
  FEMO_HIP_CHECK(hipGetDeviceCount(&ndev));
  FEMO_REQUIRE(ndev > 0, "no HIP device visible: libfemo_hip has no CPU fallback");
  FEMO_REQUIRE(device_id >= 0 && device_id < ndev, "device %d out of range (%d visible)", device_id, ndev);
  FEMO_HIP_CHECK(hipSetDevice(device_id));
  femo_ctx* c = new femo_ctx();
  c->device = device_id;
  if (stream) {
    c->stream = reinterpret_cast<hipStream_t>(stream);
  } else {
    FEMO_HIP_CHECK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    c->own_stream = true;
  }
  hipDeviceProp_t prop;
  FEMO_HIP_CHECK(hipGetDeviceProperties(&prop, device_id));
  c->n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  FEMO_HIP_CHECK(hipMalloc(&c->d_partials, FEMO_PARTIAL_SLOTS * FEMO_MAX_PARTIALS * sizeof(double)));
  FEMO_HIP_CHECK(hipMalloc(&c->d_scal, FEMO_NSCAL * sizeof(double)));
  FEMO_HIP_CHECK(hipMalloc(&c->d_flags, 8 * sizeof(int32_t)));
  FEMO_HIP_CHECK(hipMemsetAsync(c->d_flags, 0, 8 * sizeof(int32_t), c->stream));
  FEMO_HIP_CHECK(hipHostMalloc(&c->h_scal, (FEMO_NSCAL + 8) * sizeof(double), hipHostMallocDefault));
  FEMO_HIP_CHECK(hipEventCreate(&c->ev0));
  FEMO_HIP_CHECK(hipEventCreate(&c->ev1));
  FEMO_HIP_CHECK(hipStreamCreateWithFlags(&c->comm_stream, hipStreamNonBlocking));
  FEMO_HIP_CHECK(hipEventCreateWithFlags(&c->ev_main, hipEventDisableTiming));
  FEMO_HIP_CHECK(hipEventCreateWithFlags(&c->ev_comm, hipEventDisableTiming));
  c->ev_pool.resize(16);
  for (auto& e : c->ev_pool) FEMO_HIP_CHECK(hipEventCreate(&e));
  *out = c;
  return 0;
}

int femo_ctx_destroy(femo_ctx* c) {
  if (!c) return 0;
  hipSetDevice(c->device);
  hipStreamSynchronize(c->stream);
  if (c->comm_stream) hipStreamSynchronize(c->comm_stream);
  if (c->comm_halo) ncclCommDestroy(c->comm_halo);
  if (c->comm) ncclCommDestroy(c->comm);
  if (c->comm_stream) hipStreamDestroy(c->comm_stream);
  if (c->ev_main) hipEventDestroy(c->ev_main);
  if (c->ev_comm) hipEventDestroy(c->ev_comm);
  hipFree(c->d_partials); hipFree(c->d_scal); hipFree(c->d_flags);
  hipFree(c->cg_r); hipFree(c->cg_p); hipFree(c->cg_q); hipFree(c->cg_dinv); hipFree(c->cg_s); hipFree(c->cg_t); hipFree(c->cg_r0);
  hipHostFree(c->h_scal);
  for (int k = 0; k < FEMO_STAGE_SLOTS; ++k) {
    if (c->stage[k]) hipHostFree(c->stage[k]);
    if (c->stage_ev[k]) hipEventDestroy(c->stage_ev[k]);
  }
  if (c->copy_stream) { hipStreamSynchronize(c->copy_stream); hipStreamDestroy(c->copy_stream); }
  if (c->ev_copy) hipEventDestroy(c->ev_copy);
  hipFree(c->d_accum);
  hipEventDestroy(c->ev0); hipEventDestroy(c->ev1);
  for (auto& e : c->ev_pool) hipEventDestroy(e);
  if (c->own_stream) hipStreamDestroy(c->stream);
  delete c;
  return 0;
}

int femo_ctx_sync(femo_ctx* c) {
  FEMO_REQUIRE(c != nullptr, "null context");
  FEMO_HIP_CHECK(hipStreamSynchronize(c->stream));
  return 0;
}

void* femo_ctx_stream(femo_ctx* c) { return c ? (void*)c->stream : nullptr; }

// ------------------------------------------------------------------ vec -----
int femo_vec_create(femo_ctx* ctx, int64_t n, femo_vec** out) {
  FEMO_REQUIRE(ctx && out && n >= 0, "bad argument");
  FEMO_HIP_CHECK(hipSetDevice(ctx->device));
  femo_vec* v = new femo_vec();
  v->ctx = ctx; v->n = n; v->owned = true;
  FEMO_HIP_CHECK(hipMalloc(&v->d, (n + 2) * sizeof(double)));
  FEMO_HIP_CHECK(hipMemsetAsync(v->d, 0, (n + 2) * sizeof(double), ctx->stream));
  femo_vec_register(v);
  *out = v;
  return 0;
}

int femo_vec_wrap(femo_ctx* ctx, void* device_ptr, int64_t n, femo_vec** out) {
  FEMO_REQUIRE(ctx && out && device_ptr && n >= 0, "bad argument");
  FEMO_REQUIRE((reinterpret_cast<uintptr_t>(device_ptr) & 15) == 0, "wrapped pointer must be 16-byte aligned");
  femo_vec* v = new femo_vec();
  v->ctx = ctx; v->n = n; v->owned = false; v->d = static_cast<double*>(device_ptr);
  femo_vec_register(v);
  *out = v;
  return 0;
}

int femo_vec_destroy(femo_vec* v) {
  if (!v) return 0;
  femo_vec_unregister(v);
  if (v->d2h_ev) {
    hipEventSynchronize(v->d2h_ev);                    // a copy-out may still be reading the vector
    hipEventDestroy(v->d2h_ev);
  }
  if (v->h2d_ev) {
    hipEventSynchronize(v->h2d_ev);                    // ... or a deferred upload still writing it
    hipEventDestroy(v->h2d_ev);
  }
  if (v->owned && v->d) {
    hipStreamSynchronize(v->ctx->stream);
    hipFree(v->d);
  }
  delete v;
  return 0;
}

int64_t femo_vec_size(const femo_vec* v) { return v ? v->n : -1; }
// A mutable pointer leaves the library: assume the caller writes through it (the generation moves, so no host block
// keeps counting as a mirror of v and per-content caches keyed by (uid, generation) are rebuilt).
void* femo_vec_device_ptr(femo_vec* v) {
  if (!v) return nullptr;
  femo_vec_touch(v);
  return v->d;
}
const void* femo_vec_device_ptr_const(const femo_vec* v) { return v ? v->d : nullptr; }

// femo_vec_set_host / femo_vec_get_host / femo_vec_add_to_host: hostmem.cpp

int femo_vec_fill(femo_vec* v, double value) {
  FEMO_REQUIRE(v != nullptr, "null argument");
  if (v->n == 0) return 0;
  femo_vec_touch(v);
  hipLaunchKernelGGL(k_fill, dim3(grid_for(v->n)), dim3(256), 0, v->ctx->stream, v->n, value, v->d);
  FEMO_HIP_CHECK(hipGetLastError());
  return 0;
}

int femo_vec_copy(femo_vec* dst, const femo_vec* src) {
  FEMO_REQUIRE(dst && src, "null argument");
  FEMO_REQUIRE(dst->n == src->n, "size mismatch in vec_copy");
  FEMO_TRY(femo_vec_await(src));
  femo_vec_touch(dst);
  FEMO_HIP_CHECK(hipMemcpyAsync(dst->d, src->d, src->n * sizeof(double), hipMemcpyDeviceToDevice, dst->ctx->stream));
  return 0;
}

int femo_vec_axpy(femo_vec* y, double a, const femo_vec* x) {
  FEMO_REQUIRE(y && x, "null argument");
  FEMO_REQUIRE(y->n == x->n, "size mismatch in vec_axpy");
  if (y->n == 0) return 0;
  FEMO_TRY(femo_vec_await(x));
  femo_vec_touch(y);
  hipLaunchKernelGGL(k_axpy, dim3(grid_for(y->n)), dim3(256), 0, y->ctx->stream, y->n, a, x->d, y->d);
  FEMO_HIP_CHECK(hipGetLastError());
  return 0;
}

// ----------------------------------------------------------------- mesh -----
int femo_mesh_create(femo_ctx* ctx, int tdim, int64_t n_vert, int64_t n_rows, const double* x,
                     int64_t n_cell, const int32_t* conn, femo_mesh** out) {
  FEMO_REQUIRE(ctx && x && conn && out, "null argument");
  FEMO_HIP_CHECK(hipSetDevice(ctx->device));
  FemoTopology T;
  FEMO_TRY(femo_build_topology(tdim, n_vert, n_rows, n_cell, conn, T));
  femo_mesh* m = new femo_mesh();
  m->ctx = ctx; m->tdim = tdim; m->n_vert = n_vert; m->n_rows = n_rows; m->n_cell = n_cell;
  m->n_slices = T.n_slices; m->nnz = T.nnz; m->sell_entries = T.mptr[T.n_slices];
  m->visit_entries = T.vptr[T.n_slices]; m->max_rowlen = T.max_rowlen; m->max_valence = T.max_valence;
  m->h_mptr = T.mptr;
  m->n_vert_global = n_vert;
  for (int k = 0; k < tdim && k < 3; ++k) {
    double lo = 0.0, hi = 0.0;
    if (n_vert > 0) { lo = hi = x[k]; }
    for (int64_t v = 1; v < n_vert; ++v) {
      const double c = x[v * tdim + k];
      lo = c < lo ? c : lo; hi = c > hi ? c : hi;
    }
    m->bbox_lo[k] = lo; m->bbox_hi[k] = hi;
  }
  hipStream_t st = ctx->stream;
  FEMO_HIP_CHECK(hipMalloc(&m->d_x, std::max<int64_t>(n_vert * tdim, 1) * sizeof(double) + 64));
  FEMO_HIP_CHECK(hipMemcpyAsync(m->d_x, x, n_vert * tdim * sizeof(double), hipMemcpyHostToDevice, st));
  FEMO_HIP_CHECK(hipMalloc(&m->d_conn, std::max<int64_t>(n_cell * (tdim + 1), 1) * sizeof(int32_t) + 64));
  FEMO_HIP_CHECK(hipMemcpyAsync(m->d_conn, conn, n_cell * (tdim + 1) * sizeof(int32_t), hipMemcpyHostToDevice, st));
  FEMO_TRY(upload(&m->d_vptr, T.vptr, st));
  FEMO_TRY(upload(&m->d_visit_cell, T.visit_cell, st));
  FEMO_TRY(upload(&m->d_visit_slots, T.visit_slots, st));
  FEMO_TRY(upload(&m->d_mptr, T.mptr, st));
  FEMO_TRY(upload(&m->d_cols, T.cols, st));
  FEMO_TRY(upload(&m->d_rowlen, T.rowlen, st));
  FEMO_TRY(upload(&m->d_rowreal, T.real, st));
  FEMO_TRY(upload(&m->d_sdelta, T.sdelta, st));
  m->sdelta_stride = T.sdelta_stride; m->n_regular = T.n_regular; m->n_short = T.n_short;
  FEMO_TRY(upload(&m->d_cols16, T.cols16, st));
  FEMO_HIP_CHECK(hipStreamSynchronize(st));  // T's host buffers die with this scope
  *out = m;
  return 0;
}

int femo_mesh_destroy(femo_mesh* m) {
  if (!m) return 0;
  hipStreamSynchronize(m->ctx->stream);
  hipFree(m->d_x); hipFree(m->d_conn); hipFree(m->d_vptr); hipFree(m->d_visit_cell);
  hipFree(m->d_visit_slots); hipFree(m->d_mptr); hipFree(m->d_cols); hipFree(m->d_rowlen); hipFree(m->d_rowreal);
  femo_pc_destroy(m);
  femo_halo_direct_free(m);
  hipFree(m->d_bvmask); hipFree(m->d_visit_rec); hipFree(m->d_load); hipFree(m->d_zero_load); hipFree(m->d_pipe_dummy); hipFree(m->d_ubc);
  if (m->mass) { femo_mat_destroy(m->mass); m->mass = nullptr; }
  hipFree(m->d_mass_e); hipFree(m->d_mass_g); hipFree(m->d_cellvol); hipFree(m->d_cellvol_own); hipFree(m->d_cell_t);
  hipFree(m->d_sdelta); hipFree(m->d_cols16); hipFree(m->d_bfacets); hipFree(m->d_tperm); hipFree(m->d_send_idx); hipFree(m->d_send_uvert); hipFree(m->d_send_uptr); hipFree(m->d_send_uslot); hipFree(m->d_send_flag); hipFree(m->d_send_buf); hipFree(m->d_scratch); hipFree(m->d_slices_int); hipFree(m->d_slices_bnd); hipFree(m->d_slices_all);
  delete m;
  return 0;
}

int femo_mesh_info(const femo_mesh* m, int64_t info[FEMO_MESH_INFO_COUNT]) {
  FEMO_REQUIRE(m && info, "null argument");
  info[FEMO_MESH_TDIM] = m->tdim;
  info[FEMO_MESH_N_VERT] = m->n_vert;
  info[FEMO_MESH_N_ROWS] = m->n_rows;
  info[FEMO_MESH_N_CELL] = m->n_cell;
  info[FEMO_MESH_NNZ] = m->nnz;
  info[FEMO_MESH_SELL_ENTRIES] = m->sell_entries;
  info[FEMO_MESH_MAX_ROWLEN] = m->max_rowlen;
  info[FEMO_MESH_MAX_VALENCE] = m->max_valence;
  info[FEMO_MESH_N_SLICES] = m->n_slices;
  info[FEMO_MESH_VISIT_ENTRIES] = m->visit_entries;
  info[FEMO_MESH_REGULAR_SLICES] = m->n_regular;
  info[FEMO_MESH_SHORT_SLICES] = m->n_short;
  return 0;
}

int femo_mesh_set_boundary_facets(femo_mesh* m, const uint8_t* mask) {
  FEMO_REQUIRE(m != nullptr, "null argument");
  hipFree(m->d_bfacets); hipFree(m->d_bvmask);
  m->d_bfacets = nullptr; m->d_bvmask = nullptr;
  ++m->bfacets_version;
  if (!mask || m->n_cell == 0) return 0;
  hipStream_t st = m->ctx->stream;
  FEMO_HIP_CHECK(hipMalloc(&m->d_bfacets, m->n_cell + 64));
  FEMO_HIP_CHECK(hipMemcpyAsync(m->d_bfacets, mask, m->n_cell, hipMemcpyHostToDevice, st));
  FEMO_HIP_CHECK(hipMalloc(&m->d_bvmask, std::max<int64_t>(m->n_vert, 1) + 64));
  FEMO_HIP_CHECK(hipMemsetAsync(m->d_bvmask, 0, std::max<int64_t>(m->n_vert, 1) + 64, st));
  hipLaunchKernelGGL(k_boundary_vertices, dim3(grid_for(m->n_cell)), dim3(256), 0, st, m->n_cell, m->tdim, m->d_conn, m->d_bfacets, m->d_bvmask);
  FEMO_HIP_CHECK(hipGetLastError());
  FEMO_HIP_CHECK(hipStreamSynchronize(st));
  return 0;
}

int femo_mesh_pc_info(const femo_mesh* m, int32_t* n_levels, int64_t* finest_nodes) {
  FEMO_REQUIRE(m && n_levels && finest_nodes, "null argument");
  int nl = 0;
  FEMO_TRY(femo_pc_levels(m, &nl, finest_nodes));
  *n_levels = nl;
  return 0;
}

int femo_mesh_set_global(femo_mesh* m, const double* lo, const double* hi, int64_t n_vert_global) {
  FEMO_REQUIRE(m && lo && hi && n_vert_global >= m->n_rows, "bad argument");
  FEMO_REQUIRE(m->pc == nullptr, "set the global geometry before the first preconditioned solve");
  for (int k = 0; k < m->tdim && k < 3; ++k) {
    FEMO_REQUIRE(lo[k] <= m->bbox_lo[k] && hi[k] >= m->bbox_hi[k], "global bounding box does not contain the local mesh");
    m->bbox_lo[k] = lo[k]; m->bbox_hi[k] = hi[k];
  }
  m->n_vert_global = n_vert_global;
  return 0;
}

static int pattern_rowptr(const femo_mesh* m, std::vector<int64_t>& rowptr) {
  std::vector<int32_t> rl(m->n_slices * FEMO_WAVE);
  std::vector<uint32_t> real(m->n_slices * FEMO_WAVE);
  if (!rl.empty()) {
    FEMO_HIP_CHECK(hipMemcpyAsync(rl.data(), m->d_rowlen, rl.size() * sizeof(int32_t), hipMemcpyDeviceToHost, m->ctx->stream));
    FEMO_HIP_CHECK(hipMemcpyAsync(real.data(), m->d_rowreal, real.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, m->ctx->stream));
    FEMO_HIP_CHECK(hipStreamSynchronize(m->ctx->stream));
  }
  rowptr.assign(m->n_rows + 1, 0);
  for (int64_t v = 0; v < m->n_rows; ++v) {
    int n = 0;                                             // true couplings only (structural zeros are not exported)
    for (int k = 0; k < rl[v]; ++k) n += (k >= 32 || ((real[v] >> k) & 1u)) ? 1 : 0;
    rowptr[v + 1] = rowptr[v] + n + 1;
  }
  return 0;
}

// ------------------------------------------------------------------- bc -----
int femo_bc_create(femo_mesh* m, int64_t n, const int32_t* dofs, const double* vals, femo_bc** out) {
  FEMO_REQUIRE(m && out && n >= 0 && (n == 0 || (dofs && vals)), "bad argument");
  for (int64_t i = 0; i < n; ++i)
    FEMO_REQUIRE(dofs[i] >= 0 && dofs[i] < m->n_vert, "Dirichlet dof %d out of range", dofs[i]);
  femo_bc* b = new femo_bc();
  static std::atomic<uint64_t> next_uid{0};      // contexts may live on different host threads
  b->mesh = m; b->n = n; b->uid = ++next_uid;
  hipStream_t st = m->ctx->stream;
  FEMO_HIP_CHECK(hipMalloc(&b->d_dofs, std::max<int64_t>(n, 1) * sizeof(int32_t)));
  FEMO_HIP_CHECK(hipMalloc(&b->d_vals, std::max<int64_t>(n, 1) * sizeof(double)));
  FEMO_HIP_CHECK(hipMalloc(&b->d_mask, std::max<int64_t>(m->n_vert, 1) + 64));
  FEMO_HIP_CHECK(hipMemsetAsync(b->d_mask, 0, std::max<int64_t>(m->n_vert, 1) + 64, st));
  FEMO_HIP_CHECK(hipMalloc(&b->d_dense, (std::max<int64_t>(m->n_vert, 1) + 2) * sizeof(double)));
  FEMO_HIP_CHECK(hipMemsetAsync(b->d_dense, 0, (std::max<int64_t>(m->n_vert, 1) + 2) * sizeof(double), st));
  if (n > 0) {
    FEMO_HIP_CHECK(hipMemcpyAsync(b->d_dofs, dofs, n * sizeof(int32_t), hipMemcpyHostToDevice, st));
    FEMO_HIP_CHECK(hipMemcpyAsync(b->d_vals, vals, n * sizeof(double), hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(k_bc_mask, dim3(grid_for(n)), dim3(256), 0, st, n, b->d_dofs, b->d_vals, b->d_mask, b->d_dense);
    FEMO_HIP_CHECK(hipGetLastError());
  }
  if (m->max_rowlen <= 62 && m->n_slices > 0) {
    const int64_t nr = m->n_slices * FEMO_WAVE;
    FEMO_HIP_CHECK(hipMalloc(&b->d_rowmask, nr * sizeof(uint64_t)));
    hipLaunchKernelGGL(k_bc_rowmask, dim3((unsigned)((nr + 255) / 256)), dim3(256), 0, st, m->n_rows, m->n_slices, m->d_mptr, m->d_cols, m->d_rowlen, b->d_mask, b->d_rowmask);
    FEMO_HIP_CHECK(hipGetLastError());
  }
  FEMO_HIP_CHECK(hipStreamSynchronize(st));
  *out = b;
  return 0;
}

int femo_bc_destroy(femo_bc* b) {
  if (!b) return 0;
  hipStreamSynchronize(b->mesh->ctx->stream);
  hipFree(b->d_dofs); hipFree(b->d_vals); hipFree(b->d_mask); hipFree(b->d_dense); hipFree(b->d_rowmask);
  delete b;
  return 0;
}

// ------------------------------------------------------------------ mat -----
int femo_mat_create(femo_mesh* m, femo_mat** out) {
  FEMO_REQUIRE(m && out, "null argument");
  femo_mat* A = new femo_mat();
  A->mesh = m;
  const int64_t nd = std::max<int64_t>(m->n_slices * FEMO_WAVE, 1);
  FEMO_HIP_CHECK(hipMalloc(&A->d_diag, nd * sizeof(double)));
  FEMO_HIP_CHECK(hipMalloc(&A->d_vals, std::max<int64_t>(m->sell_entries, 1) * sizeof(double) + 64));
  FEMO_HIP_CHECK(hipMemsetAsync(A->d_diag, 0, nd * sizeof(double), m->ctx->stream));
  FEMO_HIP_CHECK(hipMemsetAsync(A->d_vals, 0, std::max<int64_t>(m->sell_entries, 1) * sizeof(double), m->ctx->stream));
  *out = A;
  return 0;
}

int femo_mat_destroy(femo_mat* A) {
  if (!A) return 0;
  hipStreamSynchronize(A->mesh->ctx->stream);
  hipFree(A->d_diag); hipFree(A->d_vals); hipFree(A->d_valsT); hipFree(A->d_valsS); hipFree(A->d_s);
  hipFree(A->d_pcmask); hipFree(A->d_idrows);
  delete A;
  return 0;
}

// Records which vertices this assembly pins (strong Dirichlet set, Nitsche facets) for the BPX
// preconditioner; the key identifies the combination so cached lattice data can be reused.  Also keeps the
// identity rows (the strong Dirichlet set the matrix was eliminated with) for the Krylov loops.
static int note_pinned_vertices(femo_mat* A, int pde, const double* params, const femo_bc* bc) {
  femo_mesh* m = A->mesh;
  if (bc != nullptr && bc->n > 0) {
    if (A->idrows_uid != bc->uid) {
      const int64_t nb = std::max<int64_t>(m->n_vert, 1) + 64;
      if (!A->d_idrows) FEMO_HIP_CHECK(hipMalloc(&A->d_idrows, nb));
      FEMO_HIP_CHECK(hipMemcpyAsync(A->d_idrows, bc->d_mask, nb, hipMemcpyDeviceToDevice, m->ctx->stream));
      A->idrows_uid = bc->uid;
    }
    A->has_idrows = true;
  } else {
    A->has_idrows = false;
  }
  A->bpx_ok = (pde == FEMO_PDE_POISSON || pde == FEMO_PDE_NL_POISSON);
  if (!A->bpx_ok) return 0;
  const bool nitsche = m->d_bvmask != nullptr && params != nullptr && params[0] != 0.0;
  const bool strong = bc != nullptr && bc->n > 0;
  // The key must change on every rank when the Dirichlet set is replaced, also on ranks whose block holds none
  // of its vertices (bc->n == 0): the coefficient rebuild behind it runs all-reduces, and a rank that kept its old
  // key would not enter them.  Ranks create their sets in the same order, so the process-local uid agrees.
  const uint64_t key = 1 + (bc != nullptr ? bc->uid << 20 : 0) + (nitsche ? (m->bfacets_version << 1) | 1 : 0);
  if (A->pc_key == key) return 0;
  A->pc_has_mask = strong || nitsche;
  if (A->pc_has_mask) {
    const int64_t nb = std::max<int64_t>(m->n_vert, 1);
    if (!A->d_pcmask) FEMO_HIP_CHECK(hipMalloc(&A->d_pcmask, nb + 64));
    hipLaunchKernelGGL(k_or_masks, dim3(grid_for(nb)), dim3(256), 0, m->ctx->stream, m->n_vert,
                       strong ? bc->d_mask : (const uint8_t*)nullptr, nitsche ? m->d_bvmask : (const uint8_t*)nullptr, A->d_pcmask);
    FEMO_HIP_CHECK(hipGetLastError());
  }
  A->pc_key = key;
  return 0;
}

int femo_assemble_residual(femo_mesh* m, int pde, const double* params, const femo_vec* u,
                           const femo_vec* f, const femo_vec* aux, femo_vec* r) {
  FEMO_REQUIRE(m && u && f && r, "null argument");
  FEMO_REQUIRE(u->n >= m->n_vert && f->n >= m->n_cell && r->n >= m->n_rows, "vector size mismatch in assemble_residual");
  FEMO_REQUIRE(aux == nullptr || aux->n >= m->n_vert, "aux field shorter than n_vert");
  if (m->n_nbr > 0) FEMO_TRY(femo_halo_exchange(m, const_cast<femo_vec*>(u)));
  femo_vec_touch(r);
  return femo_launch_residual(m, pde, params, u->d, f->d, aux ? aux->d : nullptr, r->d, f->uid, f->gen);
}

int femo_assemble_jacobian(femo_mesh* m, int pde, const double* params, const femo_vec* u,
                           const femo_vec* f, const femo_vec* aux, const femo_bc* bc, femo_mat* J) {
  FEMO_REQUIRE(m && J, "null argument");
  FEMO_REQUIRE(J->mesh == m, "matrix belongs to another mesh");
  FEMO_REQUIRE(bc == nullptr || bc->mesh == m, "Dirichlet set belongs to another mesh");
  J->valsT_valid = false; J->scaled_valid = false; J->s_valid = false;
  FEMO_REQUIRE(aux == nullptr || aux->n >= m->n_vert, "aux field shorter than n_vert");
  if (u && m->n_nbr > 0 && pde != FEMO_PDE_POISSON) FEMO_TRY(femo_halo_exchange(m, const_cast<femo_vec*>(u)));
  const double* ax = aux ? aux->d : nullptr;
  FEMO_TRY(note_pinned_vertices(J, pde, params, bc));
  if (bc)
    return femo_launch_system(m, pde, params, u ? u->d : nullptr, f ? f->d : nullptr, ax, bc->d_mask, bc->d_dense,
                              nullptr, nullptr, J->d_diag, J->d_vals, nullptr, f ? f->uid : 0, f ? f->gen : 0, bc->d_rowmask);
  return femo_launch_system(m, pde, params, u ? u->d : nullptr, f ? f->d : nullptr, ax, nullptr, nullptr,
                            J->d_diag, J->d_vals, nullptr, nullptr, nullptr);
}

int femo_assemble_system(femo_mesh* m, int pde, const double* params, const femo_vec* u,
                         const femo_vec* f, const femo_vec* aux, const femo_bc* bc, femo_mat* J_nobc,
                         femo_mat* A_bc, femo_vec* rhs) {
  FEMO_REQUIRE(m != nullptr, "null argument");
  FEMO_REQUIRE(J_nobc || A_bc || rhs, "nothing to assemble");
  FEMO_REQUIRE((!J_nobc || J_nobc->mesh == m) && (!A_bc || A_bc->mesh == m), "matrix belongs to another mesh");
  FEMO_REQUIRE(bc == nullptr || bc->mesh == m, "Dirichlet set belongs to another mesh");
  FEMO_REQUIRE(J_nobc != A_bc || J_nobc == nullptr, "the two matrices must be distinct");
  if (rhs) {
    FEMO_REQUIRE(u && f, "the Newton right-hand side needs u and f");
    FEMO_REQUIRE(u->n >= m->n_vert && f->n >= m->n_cell && rhs->n >= m->n_rows, "vector size mismatch in assemble_system");
  }
  FEMO_REQUIRE(aux == nullptr || aux->n >= m->n_vert, "aux field shorter than n_vert");
  FEMO_TRY(femo_vec_await(u));          // before the halo exchange packs it
  FEMO_TRY(femo_vec_await(aux));
  if (u && m->n_nbr > 0 && (rhs || pde != FEMO_PDE_POISSON)) FEMO_TRY(femo_halo_exchange(m, const_cast<femo_vec*>(u)));
  if (J_nobc) { J_nobc->valsT_valid = false; J_nobc->scaled_valid = false; J_nobc->s_valid = false; FEMO_TRY(note_pinned_vertices(J_nobc, pde, params, nullptr)); }
  femo_vec_touch(rhs);
  if (A_bc) { A_bc->valsT_valid = false; A_bc->scaled_valid = false; A_bc->s_valid = false; FEMO_TRY(note_pinned_vertices(A_bc, pde, params, bc)); }
  return femo_launch_system(m, pde, params, u ? u->d : nullptr, f ? f->d : nullptr, aux ? aux->d : nullptr,
                            bc ? bc->d_mask : nullptr, bc ? bc->d_dense : nullptr,
                            J_nobc ? J_nobc->d_diag : nullptr, J_nobc ? J_nobc->d_vals : nullptr,
                            A_bc ? A_bc->d_diag : nullptr, A_bc ? A_bc->d_vals : nullptr,
                            rhs ? rhs->d : nullptr, f ? f->uid : 0, f ? f->gen : 0, bc ? bc->d_rowmask : nullptr, f,
                            (rhs && A_bc && !J_nobc) ? A_bc : nullptr);
}

int femo_bc_apply_rhs(const femo_bc* bc, const femo_vec* u, femo_vec* b) {
  FEMO_REQUIRE(bc && u && b, "null argument");
  femo_mesh* m = bc->mesh;
  FEMO_REQUIRE(u->n >= m->n_vert && b->n >= m->n_rows, "vector size mismatch in bc_apply_rhs");
  femo_vec_touch(b);
  if (bc->n > 0) {
    hipLaunchKernelGGL(k_bc_set_rhs, dim3(grid_for(bc->n)), dim3(256), 0, m->ctx->stream, bc->n, m->n_rows, bc->d_dofs, bc->d_vals, u->d, b->d);
    FEMO_HIP_CHECK(hipGetLastError());
  }
  return 0;
}

int femo_assemble_dRdf(femo_mesh* m, int pde, const double* params, const femo_vec* u,
                       const femo_vec* f, femo_vec* vals) {
  FEMO_REQUIRE(m && vals, "null argument");
  FEMO_REQUIRE(vals->n >= m->n_cell * (m->tdim + 1), "dRdf value buffer too small");
  femo_vec_touch(vals);
  return femo_launch_dRdf(m, pde, params, u ? u->d : nullptr, f ? f->d : nullptr, vals->d);
}

int femo_assemble_dRdf_cell(femo_mesh* m, int pde, const double* params, femo_vec* cvals) {
  FEMO_REQUIRE(m && cvals, "null argument");
  FEMO_REQUIRE(pde == FEMO_PDE_POISSON || pde == FEMO_PDE_NL_POISSON, "dR/df has one value per cell only for the Poisson-type forms (pde kind %d)", pde);
  FEMO_REQUIRE(cvals->n >= m->n_cell, "dRdf value buffer too small");
  femo_vec_touch(cvals);
  return femo_launch_dRdf_cell(m, cvals->d);
}

int femo_dRdf_cell_apply(femo_mesh* m, const femo_vec* cvals, int transpose, const femo_vec* x, femo_vec* y, int accumulate) {
  FEMO_REQUIRE(m && cvals && x && y, "null argument");
  FEMO_REQUIRE(cvals->n >= m->n_cell, "dRdf value buffer too small");
  if (transpose) {
    FEMO_REQUIRE(x->n >= m->n_vert && y->n >= m->n_cell, "vector size mismatch in dRdf^T apply");
    if (m->n_nbr > 0) FEMO_TRY(femo_halo_exchange(m, const_cast<femo_vec*>(x)));
  } else {
    FEMO_REQUIRE(x->n >= m->n_cell && y->n >= m->n_rows, "vector size mismatch in dRdf apply");
  }
  femo_vec_touch(y);
  return femo_launch_dRdf_cell_apply(m, cvals->d, transpose, x->d, y->d, accumulate);
}

int femo_newton_rhs(const femo_mat* K, const femo_vec* F, const femo_vec* u, const femo_bc* bc, femo_vec* b) {
  FEMO_REQUIRE(K && F && u && bc && b, "null argument");
  femo_mesh* m = K->mesh;
  femo_ctx* ctx = m->ctx;
  FEMO_REQUIRE(u->n >= m->n_vert && F->n >= m->n_rows && b->n >= m->n_rows, "vector size mismatch in newton_rhs");
  FEMO_REQUIRE(b->d != F->d, "newton_rhs cannot run in place");
  femo_vec_touch(b);
  if (!m->d_scratch) FEMO_HIP_CHECK(hipMalloc(&m->d_scratch, (std::max<int64_t>(m->n_vert, 1) + 2) * sizeof(double)));
  double* w = m->d_scratch;  // w = (g - u) on the set, 0 elsewhere
  FEMO_HIP_CHECK(hipMemsetAsync(w, 0, (m->n_vert + 2) * sizeof(double), ctx->stream));
  if (bc->n > 0)
    hipLaunchKernelGGL(k_bc_lift_vec, dim3(grid_for(bc->n)), dim3(256), 0, ctx->stream, bc->n, bc->d_dofs, bc->d_vals, u->d, w);
  FEMO_TRY(femo_launch_spmv(K, K->d_vals, w, b->d, nullptr));
  if (m->n_rows > 0) {
    hipLaunchKernelGGL(k_add_into, dim3(grid_for(m->n_rows)), dim3(256), 0, ctx->stream, m->n_rows, F->d, b->d);
    if (bc->n > 0)
      hipLaunchKernelGGL(k_bc_set_rhs, dim3(grid_for(bc->n)), dim3(256), 0, ctx->stream, bc->n, m->n_rows, bc->d_dofs, bc->d_vals, u->d, b->d);
  }
  FEMO_HIP_CHECK(hipGetLastError());
  return 0;
}

int femo_dRdf_apply(femo_mesh* m, const femo_vec* vals, int transpose, const femo_vec* x, femo_vec* y, int accumulate) {
  FEMO_REQUIRE(m && vals && x && y, "null argument");
  FEMO_REQUIRE(vals->n >= m->n_cell * (m->tdim + 1), "dRdf value buffer too small");
  if (transpose) {
    FEMO_REQUIRE(x->n >= m->n_vert && y->n >= m->n_cell, "vector size mismatch in dRdf^T apply");
    if (m->n_nbr > 0) FEMO_TRY(femo_halo_exchange(m, const_cast<femo_vec*>(x)));
  } else {
    FEMO_REQUIRE(x->n >= m->n_cell && y->n >= m->n_rows, "vector size mismatch in dRdf apply");
  }
  femo_vec_touch(y);
  return femo_launch_dRdf_apply(m, vals->d, transpose, x->d, y->d, accumulate);
}

int femo_mesh_pattern_csr(const femo_mesh* m, int64_t* rowptr, int32_t* col) {
  FEMO_REQUIRE(m && rowptr, "null argument");
  std::vector<int64_t> rp;
  FEMO_TRY(pattern_rowptr(m, rp));
  std::copy(rp.begin(), rp.end(), rowptr);
  if (!col) return 0;
  // reuse the matrix exporter with throw-away values
  femo_mat* A = nullptr;
  FEMO_TRY(femo_mat_create(const_cast<femo_mesh*>(m), &A));
  std::vector<double> val(m->nnz);
  int rc = femo_mat_export_csr(A, rowptr, col, val.data());
  femo_mat_destroy(A);
  return rc;
}

int femo_mat_export_csr(const femo_mat* A, int64_t* rowptr, int32_t* col, double* val) {
  FEMO_REQUIRE(A && rowptr && col && val, "null argument");
  const femo_mesh* m = A->mesh;
  hipStream_t st = m->ctx->stream;
  std::vector<int64_t> rp;
  FEMO_TRY(pattern_rowptr(m, rp));
  std::copy(rp.begin(), rp.end(), rowptr);
  if (m->n_rows == 0) return 0;
  int64_t* d_rp = nullptr; int32_t* d_col = nullptr; double* d_val = nullptr;
  FEMO_HIP_CHECK(hipMalloc(&d_rp, rp.size() * sizeof(int64_t)));
  FEMO_HIP_CHECK(hipMalloc(&d_col, m->nnz * sizeof(int32_t)));
  FEMO_HIP_CHECK(hipMalloc(&d_val, m->nnz * sizeof(double)));
  FEMO_HIP_CHECK(hipMemcpyAsync(d_rp, rp.data(), rp.size() * sizeof(int64_t), hipMemcpyHostToDevice, st));
  hipLaunchKernelGGL(k_export_rows, dim3((unsigned)((m->n_rows + 255) / 256)), dim3(256), 0, st, m->n_rows, m->d_mptr, m->d_cols, m->d_rowlen, m->d_rowreal, A->d_diag, A->d_vals, d_rp, d_col, d_val);
  FEMO_HIP_CHECK(hipGetLastError());
  FEMO_HIP_CHECK(hipMemcpyAsync(col, d_col, m->nnz * sizeof(int32_t), hipMemcpyDeviceToHost, st));
  FEMO_HIP_CHECK(hipMemcpyAsync(val, d_val, m->nnz * sizeof(double), hipMemcpyDeviceToHost, st));
  FEMO_HIP_CHECK(hipStreamSynchronize(st));
  hipFree(d_rp); hipFree(d_col); hipFree(d_val);
  return 0;
}

int femo_mat_diagonal(const femo_mat* A, femo_vec* d) {
  FEMO_REQUIRE(A && d, "null argument");
  FEMO_REQUIRE(d->n >= A->mesh->n_rows, "vector too small");
  femo_vec_touch(d);
  FEMO_HIP_CHECK(hipMemcpyAsync(d->d, A->d_diag, A->mesh->n_rows * sizeof(double), hipMemcpyDeviceToDevice, A->mesh->ctx->stream));
  return 0;
}

// ----------------------------------------------------------- functional -----
int femo_functional_value(femo_mesh* m, int kind, const double* params, const femo_vec* u,
                          const femo_vec* f, const femo_vec* u_d, double* value) {
  FEMO_REQUIRE(m && u && f && u_d && value, "null argument");
  FEMO_REQUIRE(u->n >= m->n_vert && u_d->n >= m->n_vert && f->n >= m->n_cell, "vector size mismatch in functional");
  if (m->n_nbr > 0) FEMO_TRY(femo_halo_exchange(m, const_cast<femo_vec*>(u)));
  const uint64_t key[4] = {u->uid, u->gen, u_d->uid, u_d->gen};
  return femo_launch_functional_value(m, kind, params, u->d, f->d, u_d->d, value, key);
}

int femo_functional_grad_u(femo_mesh* m, int kind, const double* params, const femo_vec* u,
                           const femo_vec* f, const femo_vec* u_d, femo_vec* g) {
  FEMO_REQUIRE(m && u && u_d && g, "null argument");
  FEMO_REQUIRE(u->n >= m->n_vert && u_d->n >= m->n_vert && g->n >= m->n_rows, "vector size mismatch in functional grad_u");
  if (m->n_nbr > 0) FEMO_TRY(femo_halo_exchange(m, const_cast<femo_vec*>(u)));
  const uint64_t key[4] = {u->uid, u->gen, u_d->uid, u_d->gen};      // before the touch: g may alias neither
  femo_vec_touch(g);
  return femo_launch_functional_grad_u(m, kind, params, u->d, f ? f->d : nullptr, u_d->d, g->d, key);
}

int femo_functional_grad_f(femo_mesh* m, int kind, const double* params, const femo_vec* u,
                           const femo_vec* f, const femo_vec* u_d, femo_vec* g) {
  FEMO_REQUIRE(m && f && g, "null argument");
  FEMO_REQUIRE(f->n >= m->n_cell && g->n >= m->n_cell, "vector size mismatch in functional grad_f");
  femo_vec_touch(g);
  return femo_launch_functional_grad_f(m, kind, params, u ? u->d : nullptr, f->d, u_d ? u_d->d : nullptr, g->d);
}

int femo_cell_expression(femo_mesh* m, int kind, const double* params, const femo_vec* in, femo_vec* out) {
  FEMO_REQUIRE(m && in && out, "null argument");
  FEMO_REQUIRE(out->n >= m->n_cell && in->n >= (kind == 0 ? m->n_vert : m->n_cell), "vector size mismatch in cell_expression");
  if (kind == 0 && m->n_nbr > 0) FEMO_TRY(femo_halo_exchange(m, const_cast<femo_vec*>(in)));
  femo_vec_touch(out);
  return femo_launch_cell_expr(m, kind, params, in->d, out->d);
}

int femo_vec_pointwise_divide(femo_vec* y, const femo_vec* x, const femo_vec* d, int64_t n) {
  FEMO_REQUIRE(y && x && d, "null argument");
  FEMO_REQUIRE(n <= y->n && n <= x->n && n <= d->n, "length exceeds vector size");
  if (n == 0) return 0;
  femo_vec_touch(y);
  hipLaunchKernelGGL(k_pdiv, dim3(grid_for(n)), dim3(256), 0, y->ctx->stream, n, x->d, d->d, y->d);
  FEMO_HIP_CHECK(hipGetLastError());
  return 0;
}

// ----------------------------------------------------------------- comm -----
int femo_comm_unique_id(char id[128]) {
  FEMO_REQUIRE(id != nullptr, "null argument");
  static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
  ncclUniqueId u;
  FEMO_NCCL_CHECK(ncclGetUniqueId(&u));
  memcpy(id, &u, 128);
  return 0;
}

int femo_comm_init(femo_ctx* ctx, const char id[128], int rank, int nranks) {
  FEMO_REQUIRE(ctx && id, "null argument");
  FEMO_REQUIRE(nranks >= 1 && rank >= 0 && rank < nranks, "bad rank %d of %d", rank, nranks);
  FEMO_REQUIRE(ctx->comm == nullptr, "communicator already initialised");
  FEMO_HIP_CHECK(hipSetDevice(ctx->device));
  ncclUniqueId u;
  memcpy(&u, id, 128);
  FEMO_NCCL_CHECK(ncclCommInitRank(&ctx->comm, nranks, u, rank));
  ctx->rank = rank;
  ctx->nranks = nranks;
  // Second communicator for the neighbour exchanges (femo_internal.h: comm_halo).  OPT-IN (FEMO_SPLIT_COMM=1) until a
  // multi-GPU run has exercised halo traffic on one communicator concurrently with the all-reduce on another (ADVICE
  // round 5): the default is the single communicator, on which RCCL orders the two streams' operations itself.  When
  // asked for, the decision is COLLECTIVE: every rank calls the split, the ranks all-reduce "my split worked" over
  // `comm`, and comm_halo is used only if it worked everywhere -- a rank that fell back on its own while its peers sent
  // on comm_halo would hang the job.
  ctx->comm_halo = nullptr;
  if (femo_env_flag("FEMO_SPLIT_COMM") && !femo_env_flag("FEMO_SINGLE_COMM")) {
    ncclComm_t split = nullptr;
    const bool ok = ncclCommSplit(ctx->comm, 0, rank, &split, nullptr) == ncclSuccess && split != nullptr;
    ctx->h_scal[0] = ok ? 0.0 : 1.0;                          // number of ranks whose split failed
    FEMO_HIP_CHECK(hipMemcpyAsync(ctx->d_scal, ctx->h_scal, sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    FEMO_NCCL_CHECK(ncclAllReduce(ctx->d_scal, ctx->d_scal, 1, ncclDouble, ncclSum, ctx->comm, ctx->stream));
    FEMO_HIP_CHECK(hipMemcpyAsync(ctx->h_scal, ctx->d_scal, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    FEMO_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    if (ctx->h_scal[0] == 0.0) ctx->comm_halo = split;
    else if (ok) ncclCommDestroy(split);
  }
  return 0;
}

int femo_comm_rank(const femo_ctx* ctx, int* rank, int* nranks) {
  FEMO_REQUIRE(ctx && rank && nranks, "null argument");
  *rank = ctx->rank;
  *nranks = ctx->nranks;
  return 0;
}

int femo_mesh_set_halo(femo_mesh* m, int n_nbr, const int32_t* nbr, const int64_t* send_ptr,
                       const int32_t* send_idx, const int64_t* recv_ptr) {
  FEMO_REQUIRE(m && n_nbr >= 0, "bad argument");
  FEMO_REQUIRE(n_nbr == 0 || (nbr && send_ptr && recv_ptr), "null halo plan");
  hipFree(m->d_send_idx); hipFree(m->d_send_buf); hipFree(m->d_slices_int); hipFree(m->d_slices_bnd); hipFree(m->d_slices_all);
  m->d_slices_all = nullptr;
  hipFree(m->d_send_uvert); hipFree(m->d_send_uptr); hipFree(m->d_send_uslot); hipFree(m->d_send_flag);
  m->d_send_uvert = m->d_send_uptr = m->d_send_uslot = nullptr; m->d_send_flag = nullptr; m->n_send_verts = 0;
  m->d_send_idx = nullptr; m->d_send_buf = nullptr; m->d_slices_int = m->d_slices_bnd = nullptr;
  m->n_int = m->n_bnd = 0;
  femo_halo_direct_free(m);                 // a direct plan belongs to the halo plan it was connected for
  m->n_nbr = n_nbr;
  m->nbr.assign(nbr, nbr + n_nbr);
  m->send_ptr.assign(send_ptr, send_ptr + n_nbr + (n_nbr ? 1 : 0));
  m->recv_ptr.assign(recv_ptr, recv_ptr + n_nbr + (n_nbr ? 1 : 0));
  if (n_nbr == 0) return 0;
  const int64_t ns = send_ptr[n_nbr], nr = recv_ptr[n_nbr];
  FEMO_REQUIRE(nr == m->n_vert - m->n_rows, "halo plan receives %lld values but the mesh has %lld ghosts",
               (long long)nr, (long long)(m->n_vert - m->n_rows));
  for (int64_t i = 0; i < ns; ++i)
    FEMO_REQUIRE(send_idx[i] >= 0 && send_idx[i] < m->n_rows, "halo send index out of the owned range");
  FEMO_HIP_CHECK(hipMalloc(&m->d_send_idx, std::max<int64_t>(ns, 1) * sizeof(int32_t)));
  FEMO_HIP_CHECK(hipMalloc(&m->d_send_buf, std::max<int64_t>(ns, 1) * sizeof(double)));
  if (ns > 0) {
    FEMO_HIP_CHECK(hipMemcpyAsync(m->d_send_idx, send_idx, ns * sizeof(int32_t), hipMemcpyHostToDevice, m->ctx->stream));
    FEMO_HIP_CHECK(hipStreamSynchronize(m->ctx->stream));
  }
  {
    // the send list by vertex (femo_internal.h: d_send_uvert)
    FEMO_REQUIRE(ns < (int64_t(1) << 31), "halo plan too large for 32-bit slots");
    std::vector<int32_t> order((size_t)ns);
    for (int64_t i = 0; i < ns; ++i) order[(size_t)i] = (int32_t)i;
    std::stable_sort(order.begin(), order.end(), [&](int32_t a, int32_t b) { return send_idx[a] < send_idx[b]; });
    std::vector<int32_t> uvert, uptr, uslot((size_t)ns);
    std::vector<uint8_t> flag((size_t)std::max<int64_t>(m->n_rows, 1), 0);
    for (int64_t k = 0; k < ns; ++k) {
      const int32_t v = send_idx[order[(size_t)k]];
      if (uvert.empty() || uvert.back() != v) { uvert.push_back(v); uptr.push_back((int32_t)k); flag[(size_t)v] = 1; }
      uslot[(size_t)k] = order[(size_t)k];
    }
    uptr.push_back((int32_t)ns);
    m->n_send_verts = (int64_t)uvert.size();
    FEMO_HIP_CHECK(hipMalloc(&m->d_send_uvert, std::max<size_t>(uvert.size(), 1) * sizeof(int32_t)));
    FEMO_HIP_CHECK(hipMalloc(&m->d_send_uptr, uptr.size() * sizeof(int32_t)));
    FEMO_HIP_CHECK(hipMalloc(&m->d_send_uslot, std::max<size_t>(uslot.size(), 1) * sizeof(int32_t)));
    FEMO_HIP_CHECK(hipMalloc(&m->d_send_flag, flag.size()));
    if (!uvert.empty()) FEMO_HIP_CHECK(hipMemcpy(m->d_send_uvert, uvert.data(), uvert.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    FEMO_HIP_CHECK(hipMemcpy(m->d_send_uptr, uptr.data(), uptr.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    if (!uslot.empty()) FEMO_HIP_CHECK(hipMemcpy(m->d_send_uslot, uslot.data(), uslot.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    FEMO_HIP_CHECK(hipMemcpy(m->d_send_flag, flag.data(), flag.size(), hipMemcpyHostToDevice));
  }
  // slices without ghost columns can be multiplied while the halo is in flight
  FEMO_TRY(femo_mesh_classify_slices(m));
  // The model communicator (femo_comm_model) has no peers to connect to: its device-initiated ghost refresh runs against
  // the rank's own scratch (loopback) -- the same producer stores, counter bumps and consumer waits as a real rank, no
  // wire.  FEMO_HALO_RCCL=1 keeps the model on the ncclSend/Recv-shaped path (comparison runs).
  if (m->ctx->model && n_nbr <= FEMO_MAX_NBR && !femo_env_flag("FEMO_HALO_RCCL")) {
    char handle[64]; uint64_t addr = 0; int32_t nb = 0; int ok = 0;
    FEMO_TRY(femo_mesh_halo_direct_export(m, handle, &addr, &nb));
    FEMO_TRY(femo_mesh_halo_direct_connect(m, 2, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr));
    FEMO_TRY(femo_mesh_halo_direct_selftest(m, &ok));
    FEMO_TRY(femo_mesh_halo_direct_enable(m, ok));
  }
  return 0;
}

int femo_allreduce_sum(femo_ctx* ctx, double* host_inout, int n) {
  FEMO_REQUIRE(ctx && host_inout && n >= 0 && n <= FEMO_NSCAL, "bad argument");
  if (ctx->nranks == 1 || n == 0) return 0;
  memcpy(ctx->h_scal, host_inout, n * sizeof(double));
  FEMO_HIP_CHECK(hipMemcpyAsync(ctx->d_scal, ctx->h_scal, n * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  FEMO_TRY(femo_coll_allreduce(ctx, ctx->d_scal, n, ctx->stream));
  FEMO_HIP_CHECK(hipMemcpyAsync(ctx->h_scal, ctx->d_scal, n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  FEMO_HIP_CHECK(hipStreamSynchronize(ctx->stream));
  memcpy(host_inout, ctx->h_scal, n * sizeof(double));
  return 0;
}

}  // extern "C"
