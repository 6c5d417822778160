// Device-initiated ghost refresh (round 6): see FemoHaloDirect in femo_internal.h.
//
// New design -- the reference has no halo code at all (dolfinx / PETSc ghost updates are
// implicit, femo/fea/utils_dolfinx.py:167,200); SURVEY.md section 5 names "IPC-mapped peer buffers" next to
// ncclSend/ncclRecv as the MI355X way.  xGMI is point to point: a store into a mapped peer buffer IS the message.
//
// Set-up is collective and driven by the control plane (femo_amd/dist): every rank exports its inbox
// (femo_mesh_halo_direct_export), the ranks exchange the handles / addresses and their halo plans, every rank connects
// to its neighbours (femo_mesh_halo_direct_connect), runs one test exchange with a known pattern through the same
// device code the solver uses (femo_mesh_halo_direct_selftest) and the ranks agree -- all or none -- on using it
// (femo_mesh_halo_direct_enable).  A mesh without an enabled direct plan exchanges through ncclSend/ncclRecv as before.
#include "femo_internal.h"

namespace {

constexpr long long HALO_TIMEOUT_TICKS = 400000000ll;      // 4 s of the 100 MHz wall clock: a peer that never arrives

// consumer side: thread k < n_nbr spins on counter k of this rank's inbox; every thread of the block leaves with the
// neighbours' stores of exchange `epoch` visible
__device__ __forceinline__ void halo_wait(const unsigned long long* cnt, const int32_t* blocks, int n_nbr,
                                          unsigned long long epoch, int32_t* err) {
  if ((int)threadIdx.x < n_nbr) {
    const unsigned long long want = epoch * (unsigned long long)blocks[threadIdx.x];
    const unsigned long long* c = cnt + (size_t)threadIdx.x * FEMO_HALO_CNT_STRIDE;
    const long long t0 = wall_clock64();
    // relaxed system-scope loads (they bypass the caches); no acquire fence afterwards: the inbox is uncached memory, its
    // loads are issued behind the barrier below and are served by the memory the neighbours' stores went to (an acquire at
    // system scope would invalidate this XCD's L2 for every kernel that follows -- see femo_halo_signal)
    while (__hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < want) {
      __builtin_amdgcn_s_sleep(4);
      if (wall_clock64() - t0 > HALO_TIMEOUT_TICKS) {
        atomicExch(err, 1);
        break;
      }
    }
  }
  __syncthreads();
}

// generic producer: the owned values of the send list, straight into the neighbours' inboxes
__global__ __launch_bounds__(FEMO_BLOCK) void k_halo_push(const FemoHaloPeers* __restrict__ P, const int32_t* __restrict__ send_idx,
                                                         const double* __restrict__ x, unsigned long long epoch) {
  const int32_t ns = P->n_send;
  for (int32_t i = (int32_t)(blockIdx.x * FEMO_BLOCK + threadIdx.x); i < ns; i += (int32_t)(gridDim.x * FEMO_BLOCK))
    femo_halo_store(P, epoch, i, x[send_idx[i]]);
  femo_halo_signal(P);
}

// generic consumer: wait, then inbox generation -> ghost tail of the vector
__global__ __launch_bounds__(FEMO_BLOCK) void k_halo_pull(const unsigned long long* __restrict__ cnt, const int32_t* __restrict__ blocks,
                                                         int n_nbr, unsigned long long epoch, int32_t* __restrict__ err,
                                                         const double* __restrict__ inbox, int64_t n_ghost, double* __restrict__ tail) {
  halo_wait(cnt, blocks, n_nbr, epoch, err);
  const double* src = inbox + (int64_t)(epoch & 1ull) * n_ghost;
  for (int64_t i = (int64_t)blockIdx.x * FEMO_BLOCK + threadIdx.x; i < n_ghost; i += (int64_t)gridDim.x * FEMO_BLOCK)
    tail[i] = femo_halo_load(src + i);
}

// self-test: every slot carries a number both sides can compute
__device__ __forceinline__ double test_value(int src_rank, int dst_rank, int64_t i, unsigned long long epoch) {
  return 1.0 + (double)src_rank * 4096.0 + (double)dst_rank + (double)(i % 1021) * (1.0 / 1024.0) + (double)epoch * 65536.0;
}
__global__ __launch_bounds__(FEMO_BLOCK) void k_halo_test_push(const FemoHaloPeers* __restrict__ P, const int32_t* __restrict__ nbr_rank,
                                                              int my_rank, unsigned long long epoch) {
  const int32_t ns = P->n_send;
  for (int32_t i = (int32_t)(blockIdx.x * FEMO_BLOCK + threadIdx.x); i < ns; i += (int32_t)(gridDim.x * FEMO_BLOCK)) {
    const int k = P->slot_nbr[i];
    femo_halo_store(P, epoch, i, test_value(my_rank, nbr_rank[k], i - P->send_ptr[k], epoch));
  }
  femo_halo_signal(P);
}
__global__ __launch_bounds__(FEMO_BLOCK) void k_halo_test_check(const unsigned long long* __restrict__ cnt, const int32_t* __restrict__ blocks,
                                                               int n_nbr, unsigned long long epoch, int32_t* __restrict__ err,
                                                               const double* __restrict__ inbox, int64_t n_ghost,
                                                               const int64_t* __restrict__ recv_ptr, const int32_t* __restrict__ nbr_rank,
                                                               int my_rank, int32_t* __restrict__ bad) {
  halo_wait(cnt, blocks, n_nbr, epoch, err);
  const double* src = inbox + (int64_t)(epoch & 1ull) * n_ghost;
  for (int64_t i = (int64_t)blockIdx.x * FEMO_BLOCK + threadIdx.x; i < n_ghost; i += (int64_t)gridDim.x * FEMO_BLOCK) {
    int k = 0;
    while (k + 1 < n_nbr && i >= recv_ptr[k + 1]) ++k;
    const double want = test_value(nbr_rank[k], my_rank, i - recv_ptr[k], epoch);
    if (femo_halo_load(src + i) != want) atomicAdd(bad, 1);
  }
}

int pull_grid(int64_t n_ghost) { return (int)std::max<int64_t>(1, std::min<int64_t>(64, (n_ghost + FEMO_BLOCK - 1) / FEMO_BLOCK)); }

}  // namespace

void femo_halo_direct_free(femo_mesh* m) {
  FemoHaloDirect* h = m->hd;
  if (!h) return;
  for (void* p : h->opened) hipIpcCloseMemHandle(p);
  hipFree(h->inbox_raw); hipFree(h->d_peers); hipFree(h->d_blocks); hipFree(h->d_slot_nbr); hipFree(h->d_dst); hipFree(h->d_loop); hipFree(h->d_err);
  delete h;
  m->hd = nullptr;
}

bool femo_halo_direct_ready(const femo_mesh* m) { return m->hd != nullptr && m->hd->ready; }

unsigned long long femo_halo_direct_begin(femo_mesh* m) {
  FemoHaloDirect* h = m->hd;
  ++h->exchanges;
  return ++h->epoch;
}

// Emulated ranks are host threads of one process on ONE GPU: a consumer grid spinning for a producer another thread has
// not launched yet could keep that producer off the device.  The emulation therefore meets on the host between the two
// halves (like its other collectives, comm.cpp); real ranks never do.

int femo_halo_direct_pull(femo_mesh* m, unsigned long long epoch, double* ghost_tail, hipStream_t st) {
  FemoHaloDirect* h = m->hd;
  FEMO_REQUIRE(h && h->ready, "direct halo: not connected");
  if (m->ctx->emu != nullptr) FEMO_TRY(femo_emu_rendezvous(m->ctx, st));
  if (h->n_ghost == 0) return 0;
  hipLaunchKernelGGL(k_halo_pull, dim3(pull_grid(h->n_ghost)), dim3(FEMO_BLOCK), 0, st, h->counters, h->d_blocks, m->n_nbr, epoch, h->d_err, h->inbox, h->n_ghost, ghost_tail);
  FEMO_HIP_CHECK(hipGetLastError());
  return 0;
}

int femo_halo_direct_exchange(femo_mesh* m, const double* x_owned, double* ghost_tail, hipStream_t st) {
  FemoHaloDirect* h = m->hd;
  FEMO_REQUIRE(h && h->ready, "direct halo: not connected");
  femo_ctx* ctx = m->ctx;
  ++ctx->n_neighbor; ctx->neighbor_doubles += m->send_ptr[m->n_nbr];
  const unsigned long long epoch = femo_halo_direct_begin(m);
  hipLaunchKernelGGL(k_halo_push, dim3(h->n_blocks), dim3(FEMO_BLOCK), 0, st, h->d_peers, m->d_send_idx, x_owned, epoch);
  FEMO_HIP_CHECK(hipGetLastError());
  return femo_halo_direct_pull(m, epoch, ghost_tail, st);
}

extern "C" {

// Allocates this rank's inbox for `mesh` (the halo plan must be set) and returns what the neighbours need: the IPC handle
// (other processes), the address (same process), the number of workgroups of this rank's producers.
int femo_mesh_halo_direct_export(femo_mesh* m, char ipc_handle[64], uint64_t* address, int32_t* n_blocks) {
  FEMO_REQUIRE(m && ipc_handle && address && n_blocks, "null argument");
  FEMO_REQUIRE(m->n_nbr > 0 && m->n_nbr <= FEMO_MAX_NBR, "direct halo: %d neighbours (1 .. %d supported)", m->n_nbr, FEMO_MAX_NBR);
  static_assert(sizeof(hipIpcMemHandle_t) == 64, "hipIpcMemHandle_t is 64 bytes");
  femo_halo_direct_free(m);
  FemoHaloDirect* h = new FemoHaloDirect();
  m->hd = h;
  h->n_ghost = m->n_vert - m->n_rows;
  const int64_t ns = m->send_ptr[m->n_nbr];
  FEMO_REQUIRE(ns < (int64_t(1) << 31), "direct halo: send list too long");
  // one workgroup per 256 send vertices, at most one per CU (64 was measured to double the prolongation's halo-first part)
  h->n_blocks = (int)std::max<int64_t>(1, std::min<int64_t>(256, (std::max<int64_t>(m->n_send_verts, 1) + FEMO_BLOCK - 1) / FEMO_BLOCK));
  const size_t data_bytes = (((size_t)2 * (size_t)std::max<int64_t>(h->n_ghost, 1) * sizeof(double)) + 63) & ~size_t(63);
  const size_t cnt_bytes = (size_t)m->n_nbr * FEMO_HALO_CNT_STRIDE * sizeof(unsigned long long);
  // uncached: the neighbours' stores arrive behind this GPU's L2, so its own reads must not be served from there
  hipError_t e = hipExtMallocWithFlags(&h->inbox_raw, data_bytes + cnt_bytes, hipDeviceMallocUncached);
  if (e != hipSuccess) {
    femo_set_error("direct halo: hipExtMallocWithFlags(uncached) failed: %s", hipGetErrorString(e));
    femo_halo_direct_free(m);
    return 2;
  }
  FEMO_HIP_CHECK(hipMemset(h->inbox_raw, 0, data_bytes + cnt_bytes));
  h->inbox = static_cast<double*>(h->inbox_raw);
  h->counters = reinterpret_cast<unsigned long long*>(static_cast<char*>(h->inbox_raw) + data_bytes);
  FEMO_HIP_CHECK(hipMalloc(&h->d_err, 2 * sizeof(int32_t)));
  FEMO_HIP_CHECK(hipMemset(h->d_err, 0, 2 * sizeof(int32_t)));
  memset(ipc_handle, 0, 64);
  hipIpcMemHandle_t ih;
  if (hipIpcGetMemHandle(&ih, h->inbox_raw) == hipSuccess) memcpy(ipc_handle, &ih, 64);
  else (void)hipGetLastError();                         // same-process use does not need it; connect() reports a missing handle
  *address = (uint64_t)(uintptr_t)h->inbox_raw;
  *n_blocks = h->n_blocks;
  return 0;
}

// For neighbour k (order of the halo plan): where ITS inbox is (handle of another process, or address in this one;
// mode 2 = loopback: this rank's own scratch stands in for every peer -- the model communicator), the offset (doubles)
// of THIS rank's segment in it and ITS ghost count, the index of its counter for this rank, its workgroups per exchange.
int femo_mesh_halo_direct_connect(femo_mesh* m, int mode, const char* handles /* n_nbr x 64 */, const uint64_t* addresses,
                                  const int64_t* remote_offset, const int64_t* remote_n_ghost, const int32_t* remote_slot,
                                  const int32_t* remote_blocks) {
  FEMO_REQUIRE(m && m->hd && m->hd->inbox_raw, "direct halo: export first");
  FEMO_REQUIRE(mode == 2 || (remote_offset && remote_n_ghost && remote_slot && remote_blocks), "null argument");
  FemoHaloDirect* h = m->hd;
  const int nn = m->n_nbr;
  const int64_t ns = m->send_ptr[nn];
  FemoHaloPeers P;
  memset(&P, 0, sizeof P);
  P.n_nbr = nn; P.n_send = (int32_t)ns;
  for (int k = 0; k <= nn; ++k) P.send_ptr[k] = (int32_t)m->send_ptr[k];
  std::vector<int32_t> blocks((size_t)nn);
  h->same_process = mode == 1; h->loopback = mode == 2;
  if (mode == 2) {
    FEMO_HIP_CHECK(hipMalloc(&h->d_loop, 2 * std::max<int64_t>(ns, 1) * sizeof(double)));
    for (int k = 0; k < nn; ++k) {
      P.seg[k] = h->d_loop + m->send_ptr[k];
      P.stride[k] = ns;
      P.cnt[k] = h->counters + (size_t)k * FEMO_HALO_CNT_STRIDE;
      blocks[(size_t)k] = h->n_blocks;
    }
  } else {
    for (int k = 0; k < nn; ++k) {
      char* base = nullptr;
      if (mode == 1) {
        FEMO_REQUIRE(addresses && addresses[k], "direct halo: no address for neighbour %d", k);
        base = reinterpret_cast<char*>((uintptr_t)addresses[k]);
      } else {
        FEMO_REQUIRE(handles != nullptr, "direct halo: no IPC handles");
        hipIpcMemHandle_t ih;
        memcpy(&ih, handles + (size_t)k * 64, 64);
        void* p = nullptr;
        hipError_t e = hipIpcOpenMemHandle(&p, ih, hipIpcMemLazyEnablePeerAccess);
        if (e != hipSuccess) {
          femo_set_error("direct halo: hipIpcOpenMemHandle for neighbour rank %d failed: %s", m->nbr[(size_t)k], hipGetErrorString(e));
          return 2;
        }
        h->opened.push_back(p);
        base = static_cast<char*>(p);
      }
      const size_t rdata = (((size_t)2 * (size_t)std::max<int64_t>(remote_n_ghost[k], 1) * sizeof(double)) + 63) & ~size_t(63);
      P.seg[k] = reinterpret_cast<double*>(base) + remote_offset[k];
      P.stride[k] = remote_n_ghost[k];
      P.cnt[k] = reinterpret_cast<unsigned long long*>(base + rdata) + (size_t)remote_slot[k] * FEMO_HALO_CNT_STRIDE;
      blocks[(size_t)k] = remote_blocks[k];
    }
  }
  std::vector<uint8_t> slot_nbr((size_t)std::max<int64_t>(ns, 1), 0);
  for (int k = 0; k < nn; ++k)
    for (int64_t i = m->send_ptr[k]; i < m->send_ptr[k + 1]; ++i) slot_nbr[(size_t)i] = (uint8_t)k;
  FEMO_HIP_CHECK(hipMalloc(&h->d_slot_nbr, slot_nbr.size()));
  FEMO_HIP_CHECK(hipMemcpy(h->d_slot_nbr, slot_nbr.data(), slot_nbr.size(), hipMemcpyHostToDevice));
  P.slot_nbr = h->d_slot_nbr;
  {
    const size_t n1 = (size_t)std::max<int64_t>(ns, 1);
    std::vector<double*> dst(2 * n1, nullptr);
    for (int g = 0; g < 2; ++g)
      for (int k = 0; k < nn; ++k)
        for (int64_t i = m->send_ptr[k]; i < m->send_ptr[k + 1]; ++i)
          dst[(size_t)g * n1 + (size_t)i] = P.seg[k] + (int64_t)g * P.stride[k] + (i - m->send_ptr[k]);
    FEMO_HIP_CHECK(hipMalloc(&h->d_dst, dst.size() * sizeof(double*)));
    FEMO_HIP_CHECK(hipMemcpy(h->d_dst, dst.data(), dst.size() * sizeof(double*), hipMemcpyHostToDevice));
    P.dst[0] = h->d_dst; P.dst[1] = h->d_dst + n1;
  }
  FEMO_HIP_CHECK(hipMalloc(&h->d_peers, sizeof P));
  FEMO_HIP_CHECK(hipMemcpy(h->d_peers, &P, sizeof P, hipMemcpyHostToDevice));
  FEMO_HIP_CHECK(hipMalloc(&h->d_blocks, (size_t)nn * sizeof(int32_t)));
  FEMO_HIP_CHECK(hipMemcpy(h->d_blocks, blocks.data(), (size_t)nn * sizeof(int32_t), hipMemcpyHostToDevice));
  return 0;
}

// One exchange of a known pattern through the producer / consumer device code (collective: every rank calls it after
// every rank has connected).  *ok = 1 when every ghost slot received exactly what its owner must have sent and no wait
// timed out.  The three test exchanges count as epochs 1 to 3.
int femo_mesh_halo_direct_selftest(femo_mesh* m, int* ok) {
  FEMO_REQUIRE(m && m->hd && m->hd->d_peers && ok, "direct halo: connect first");
  FemoHaloDirect* h = m->hd;
  femo_ctx* ctx = m->ctx;
  hipStream_t st = ctx->stream;
  const int nn = m->n_nbr;
  struct Scratch {                                    // freed on every way out (the macros below return early)
    int32_t* nbr = nullptr; int64_t* recv = nullptr;
    ~Scratch() { hipFree(nbr); hipFree(recv); }
  } scratch;
  FEMO_HIP_CHECK(hipMalloc(&scratch.nbr, (size_t)nn * sizeof(int32_t)));
  FEMO_HIP_CHECK(hipMalloc(&scratch.recv, (size_t)(nn + 1) * sizeof(int64_t)));
  int32_t* const d_nbr = scratch.nbr; int64_t* const d_recv = scratch.recv;
  FEMO_HIP_CHECK(hipMemcpy(d_nbr, m->nbr.data(), (size_t)nn * sizeof(int32_t), hipMemcpyHostToDevice));
  FEMO_HIP_CHECK(hipMemcpy(d_recv, m->recv_ptr.data(), (size_t)(nn + 1) * sizeof(int64_t), hipMemcpyHostToDevice));
  FEMO_HIP_CHECK(hipMemsetAsync(h->d_err, 0, 2 * sizeof(int32_t), st));
  int32_t res[2] = {0, 0};
  // three rounds, back to back on the stream, values that depend on the epoch: both generations of the inbox, and a round
  // whose stores were still sitting in a cache somewhere would show the previous round's numbers
  for (int round = 0; round < 3; ++round) {
    const unsigned long long epoch = femo_halo_direct_begin(m);
    hipLaunchKernelGGL(k_halo_test_push, dim3(h->n_blocks), dim3(FEMO_BLOCK), 0, st, h->d_peers, d_nbr, ctx->rank, epoch);
    if (ctx->emu != nullptr) FEMO_TRY(femo_emu_rendezvous(ctx, st));
    if (h->loopback) {
      // the loopback's stores land in the scratch, not in the inbox: only the wait is exercised
      hipLaunchKernelGGL(k_halo_pull, dim3(1), dim3(FEMO_BLOCK), 0, st, h->counters, h->d_blocks, nn, epoch, h->d_err, h->inbox, (int64_t)0, (double*)nullptr);
    } else {
      hipLaunchKernelGGL(k_halo_test_check, dim3(pull_grid(h->n_ghost)), dim3(FEMO_BLOCK), 0, st, h->counters, h->d_blocks, nn, epoch, h->d_err, h->inbox, h->n_ghost, d_recv, d_nbr, ctx->rank, h->d_err + 1);
    }
  }
  FEMO_HIP_CHECK(hipGetLastError());
  FEMO_HIP_CHECK(hipMemcpyAsync(res, h->d_err, sizeof res, hipMemcpyDeviceToHost, st));
  FEMO_HIP_CHECK(hipStreamSynchronize(st));
  FEMO_HIP_CHECK(hipMemsetAsync(h->d_err, 0, 2 * sizeof(int32_t), st));
  *ok = (res[0] == 0 && res[1] == 0) ? 1 : 0;
  return 0;
}

// The collective decision: on = 1 only when EVERY rank's self-test passed (the control plane reduces the flags).
int femo_mesh_halo_direct_enable(femo_mesh* m, int on) {
  FEMO_REQUIRE(m != nullptr, "null argument");
  if (!m->hd) return 0;
  if (on) { FEMO_REQUIRE(m->hd->d_peers != nullptr, "direct halo: connect first"); m->hd->ready = true; }
  else femo_halo_direct_free(m);
  return 0;
}

// out = {enabled, exchanges issued, consumer timeouts seen, workgroups per producer}
int femo_mesh_halo_direct_info(femo_mesh* m, int64_t out[4]) {
  FEMO_REQUIRE(m && out, "null argument");
  out[0] = out[1] = out[2] = out[3] = 0;
  if (!m->hd) return 0;
  FemoHaloDirect* h = m->hd;
  int32_t err = 0;
  if (h->d_err) {
    FEMO_HIP_CHECK(hipStreamSynchronize(m->ctx->stream));
    FEMO_HIP_CHECK(hipMemcpy(&err, h->d_err, sizeof err, hipMemcpyDeviceToHost));
  }
  out[0] = h->ready ? 1 : 0; out[1] = h->exchanges; out[2] = err; out[3] = h->n_blocks;
  return 0;
}

}  // extern "C"
