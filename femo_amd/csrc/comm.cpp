// Collectives of the library behind two entry points, femo_coll_allreduce and femo_coll_neighbors:
// RCCL when the context has a communicator, or -- for tests -- an in-process emulation in which
// several contexts ON THE SAME GPU, driven by one host thread each, act as the ranks of a group.
// RCCL cannot put two ranks on one device; the emulation runs the *same* library code a
// multi-GPU job runs (owned rows, halo plans, all-reduced scalars and lattice accumulators) with
// every collective staged through host memory and a barrier.  Slow by design, never used by bench.py.
#include <chrono>
#include <condition_variable>
#include <mutex>

#include "femo_internal.h"

struct femo_emu_group {
  int nranks = 0;
  std::mutex mu;
  std::condition_variable cv;
  int arrived = 0;
  uint64_t generation = 0;
  int contributed = 0;
  bool failed = false;
  std::vector<double> sum;
  std::vector<std::vector<std::vector<double>>> box;    // [src][dst]
};

namespace {

constexpr int EMU_TIMEOUT_S = 60;   // a rank that never arrives (exception in its thread) must not hang the others

// all ranks meet; the last one to arrive runs `on_last` before anybody leaves
template <class F>
int emu_barrier(femo_emu_group* g, F on_last) {
  std::unique_lock<std::mutex> lk(g->mu);
  if (g->failed) return -1;
  const uint64_t gen = g->generation;
  if (++g->arrived == g->nranks) {
    g->arrived = 0;
    on_last();
    ++g->generation;
    g->cv.notify_all();
    return 0;
  }
  if (!g->cv.wait_for(lk, std::chrono::seconds(EMU_TIMEOUT_S), [&] { return g->generation != gen || g->failed; })) {
    g->failed = true;
    g->cv.notify_all();
    return -1;
  }
  return g->failed ? -1 : 0;
}

int emu_allreduce(femo_ctx* ctx, double* d, int64_t count, hipStream_t st) {
  femo_emu_group* g = ctx->emu;
  std::vector<double> local((size_t)count);
  FEMO_HIP_CHECK(hipMemcpyAsync(local.data(), d, count * sizeof(double), hipMemcpyDeviceToHost, st));
  FEMO_HIP_CHECK(hipStreamSynchronize(st));
  {
    std::lock_guard<std::mutex> lk(g->mu);
    if (g->contributed == 0) g->sum.assign((size_t)count, 0.0);
    FEMO_REQUIRE((int64_t)g->sum.size() == count, "emulated all-reduce: ranks disagree on the count (%lld vs %lld)",
                 (long long)g->sum.size(), (long long)count);
    for (int64_t i = 0; i < count; ++i) g->sum[(size_t)i] += local[(size_t)i];
    ++g->contributed;
  }
  FEMO_REQUIRE(emu_barrier(g, [] {}) == 0, "emulated all-reduce: a rank did not arrive");
  {
    std::lock_guard<std::mutex> lk(g->mu);
    local = g->sum;
  }
  FEMO_REQUIRE(emu_barrier(g, [g] { g->contributed = 0; }) == 0, "emulated all-reduce: a rank did not arrive");
  FEMO_HIP_CHECK(hipMemcpyAsync(d, local.data(), count * sizeof(double), hipMemcpyHostToDevice, st));
  FEMO_HIP_CHECK(hipStreamSynchronize(st));
  return 0;
}

int emu_neighbors(femo_ctx* ctx, int n_nbr, const int32_t* nbr, const int64_t* send_ptr, const double* d_send,
                  const int64_t* recv_ptr, double* d_recv, hipStream_t st) {
  femo_emu_group* g = ctx->emu;
  const int64_t ns = send_ptr[n_nbr], nr = recv_ptr[n_nbr];
  std::vector<double> sendbuf((size_t)std::max<int64_t>(ns, 1)), recvbuf((size_t)std::max<int64_t>(nr, 1));
  if (ns > 0) FEMO_HIP_CHECK(hipMemcpyAsync(sendbuf.data(), d_send, ns * sizeof(double), hipMemcpyDeviceToHost, st));
  FEMO_HIP_CHECK(hipStreamSynchronize(st));
  {
    std::lock_guard<std::mutex> lk(g->mu);
    for (int k = 0; k < n_nbr; ++k)
      g->box[(size_t)ctx->rank][(size_t)nbr[k]].assign(sendbuf.begin() + send_ptr[k], sendbuf.begin() + send_ptr[k + 1]);
  }
  FEMO_REQUIRE(emu_barrier(g, [] {}) == 0, "emulated halo exchange: a rank did not arrive");
  {
    std::lock_guard<std::mutex> lk(g->mu);
    for (int k = 0; k < n_nbr; ++k) {
      const std::vector<double>& in = g->box[(size_t)nbr[k]][(size_t)ctx->rank];
      FEMO_REQUIRE((int64_t)in.size() == recv_ptr[k + 1] - recv_ptr[k], "emulated halo exchange: rank %d sends %lld values, rank %d expects %lld",
                   nbr[k], (long long)in.size(), ctx->rank, (long long)(recv_ptr[k + 1] - recv_ptr[k]));
      std::copy(in.begin(), in.end(), recvbuf.begin() + recv_ptr[k]);
    }
  }
  FEMO_REQUIRE(emu_barrier(g, [] {}) == 0, "emulated halo exchange: a rank did not arrive");
  if (nr > 0) FEMO_HIP_CHECK(hipMemcpyAsync(d_recv, recvbuf.data(), nr * sizeof(double), hipMemcpyHostToDevice, st));
  FEMO_HIP_CHECK(hipStreamSynchronize(st));
  return 0;
}

}  // namespace

// Emulated ranks meet on the host: everything this rank enqueued on `st` has completed when the barrier opens
// (halo_direct.hip: between the producer and the consumer half of a device-initiated ghost refresh).
int femo_emu_rendezvous(femo_ctx* ctx, hipStream_t st) {
  if (ctx->emu == nullptr) return 0;
  FEMO_HIP_CHECK(hipStreamSynchronize(st));
  FEMO_REQUIRE(emu_barrier(ctx->emu, [] {}) == 0, "emulated ghost refresh: a rank did not arrive");
  return 0;
}

// d[0..count) <- sum over the ranks, in place, ordered on `st`
int femo_coll_allreduce(femo_ctx* ctx, double* d, int64_t count, hipStream_t st) {
  if (count <= 0) return 0;
  ++ctx->n_allreduce; ctx->allreduce_doubles += count;
  if (ctx->model) return 0;                                   // femo_comm_model: the rank's own contribution is the sum
  if (ctx->emu != nullptr) return emu_allreduce(ctx, d, count, st);
  FEMO_REQUIRE(ctx->comm != nullptr, "collective before femo_comm_init");
  FEMO_NCCL_CHECK(ncclAllReduce(d, d, (size_t)count, ncclDouble, ncclSum, ctx->comm, st));
  return 0;
}

// neighbour-wise exchange: segment k of d_send goes to rank nbr[k], segment k of d_recv comes from it
int femo_coll_neighbors(femo_ctx* ctx, int n_nbr, const int32_t* nbr, const int64_t* send_ptr, const double* d_send,
                        const int64_t* recv_ptr, double* d_recv, hipStream_t st) {
  if (n_nbr == 0) return 0;
  ++ctx->n_neighbor; ctx->neighbor_doubles += send_ptr[n_nbr];
  if (ctx->model) {                                           // femo_comm_model: nobody answers -- the ghost entries are ZERO
    const int64_t nr = recv_ptr[n_nbr];                       // (defined values: the receive area may be recycled memory)
    if (nr > 0) FEMO_HIP_CHECK(hipMemsetAsync(d_recv, 0, (size_t)nr * sizeof(double), st));
    return 0;
  }
  if (ctx->emu != nullptr) return emu_neighbors(ctx, n_nbr, nbr, send_ptr, d_send, recv_ptr, d_recv, st);
  FEMO_REQUIRE(ctx->comm != nullptr, "halo exchange before femo_comm_init");
  const ncclComm_t comm = ctx->comm_halo != nullptr ? ctx->comm_halo : ctx->comm;
  FEMO_NCCL_CHECK(ncclGroupStart());
  for (int k = 0; k < n_nbr; ++k) {
    const int64_t sc = send_ptr[k + 1] - send_ptr[k];
    const int64_t rc = recv_ptr[k + 1] - recv_ptr[k];
    if (sc > 0) FEMO_NCCL_CHECK(ncclSend(d_send + send_ptr[k], (size_t)sc, ncclDouble, nbr[k], comm, st));
    if (rc > 0) FEMO_NCCL_CHECK(ncclRecv(d_recv + recv_ptr[k], (size_t)rc, ncclDouble, nbr[k], comm, st));
  }
  FEMO_NCCL_CHECK(ncclGroupEnd());
  return 0;
}

extern "C" {

int femo_comm_stats(femo_ctx* ctx, int64_t out[4], int reset) {
  FEMO_REQUIRE(ctx && out, "null argument");
  out[0] = ctx->n_allreduce; out[1] = ctx->allreduce_doubles; out[2] = ctx->n_neighbor; out[3] = ctx->neighbor_doubles;
  if (reset) ctx->n_allreduce = ctx->allreduce_doubles = ctx->n_neighbor = ctx->neighbor_doubles = 0;
  return 0;
}

int femo_emu_group_create(int nranks, femo_emu_group** out) {
  FEMO_REQUIRE(out != nullptr && nranks >= 1 && nranks <= 64, "bad argument");
  femo_emu_group* g = new femo_emu_group();
  g->nranks = nranks;
  g->box.assign((size_t)nranks, std::vector<std::vector<double>>((size_t)nranks));
  *out = g;
  return 0;
}

int femo_emu_group_destroy(femo_emu_group* g) {
  delete g;
  return 0;
}

int femo_comm_model(femo_ctx* ctx, int rank, int nranks) {
  FEMO_REQUIRE(ctx != nullptr, "null argument");
  FEMO_REQUIRE(nranks >= 1 && rank >= 0 && rank < nranks, "bad rank %d of %d", rank, nranks);
  FEMO_REQUIRE(ctx->comm == nullptr && ctx->emu == nullptr, "communicator already initialised");
  ctx->model = true;
  ctx->rank = rank;
  ctx->nranks = nranks;
  return 0;
}

int femo_comm_emulate(femo_ctx* ctx, femo_emu_group* g, int rank) {
  FEMO_REQUIRE(ctx && g, "null argument");
  FEMO_REQUIRE(rank >= 0 && rank < g->nranks, "bad rank %d of %d", rank, g->nranks);
  FEMO_REQUIRE(ctx->comm == nullptr && ctx->emu == nullptr, "communicator already initialised");
  ctx->emu = g;
  ctx->rank = rank;
  ctx->nranks = g->nranks;
  return 0;
}

}  // extern "C"
