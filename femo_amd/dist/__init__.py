"""Multi-GPU execution: one process per GPU (SPMD), RCCL over xGMI inside the C library.

Each rank runs the *same* operator stack (FEA / FEAModel / StateOperation / ...) on
its local mesh (``DistMesh``): vectors hold owned entries first, then ghosts; the
library refreshes ghosts by neighbour-wise ``ncclSend/ncclRecv`` wherever an
operator reads across the partition boundary (residual, SpMV inside CG, dR/df^T,
functional) and all-reduces every dot product.  ``torch.distributed`` (gloo) is
only the control plane: rendezvous, broadcast of the ncclUniqueId, barriers and
the max-over-ranks timing.  New design; the reference is single-rank.
"""
from __future__ import annotations

import json
import os
import time
from typing import Optional

import numpy as np

from ..fea.mesh import Mesh
from .partition import LocalMesh, build_local_mesh, rcb_partition


class DistMesh(Mesh):
    """A rank's local piece of a partitioned mesh; drop-in for ``Mesh`` in the FEA stack."""

    def __init__(self, local: LocalMesh, n_vert_global: int, n_cell_global: int, bbox=None, box_domain: bool = False):
        super().__init__(local.x, local.conn)
        self.local = local
        self.n_vert_global = int(n_vert_global)
        self.n_cell_global = int(n_cell_global)
        self.bbox = bbox        # (lo, hi) of the whole mesh: every rank builds the same BPX lattice
        self.box_domain = box_domain   # the whole mesh fills its bounding box (generated squares / cubes)

    @property
    def n_owned(self) -> int:
        return self.local.n_owned

    def boundary_facet_mask(self) -> np.ndarray:
        """Exterior facets are those of the WHOLE mesh: the cut faces of the partition are not a
        boundary (a facet seen once among the local cells may have its other cell on another rank).
        Box-filling meshes decide from the coordinates; otherwise ``partition_mesh(..., facets=True)``
        slices the whole mesh's mask at partition time (the whole mesh is not kept)."""
        if getattr(self, "_bfacets", None) is None:
            if self.box_domain:
                from .structured import box_boundary_facets
                lo, hi = self.bbox
                self._bfacets = box_boundary_facets(self.x, self.conn, float(np.min(lo)), float(np.max(hi)))
            elif self.local.nranks > 1:
                raise RuntimeError("DistMesh: exterior facets were not prepared; partition the mesh with facets=True")
            else:
                return super().boundary_facet_mask()
        return self._bfacets

    def device(self, ctx):
        if self._device is None or self._ctx is not ctx:
            from ..engine import DeviceMesh
            L = self.local
            dm = DeviceMesh(ctx, self.x, self.conn, n_rows=L.n_owned)
            if self.bbox is not None:
                dm.set_global(self.bbox[0], self.bbox[1], self.n_vert_global)
            if L.nranks > 1:
                dm.set_halo(L.nbr, L.send_ptr, L.send_idx, L.recv_ptr)
            self._device, self._ctx = dm, ctx
        return self._device


def partition_mesh(mesh: Mesh, rank: int, nranks: int, facets: bool = False) -> DistMesh:
    """Rank ``rank``'s piece of a mesh that every rank holds whole (imported meshes).  Nothing of the whole
    mesh is kept afterwards: ``facets=True`` slices its exterior-facet mask now (forms with facet terms need
    it), and its lattice occupancy -- every rank must pick the same preconditioner -- is evaluated here."""
    part = rcb_partition(mesh.x, nranks)
    local = build_local_mesh(mesh.x, mesh.conn, part, rank, nranks)
    dm = DistMesh(local, mesh.n_vert, mesh.n_cell, bbox=(mesh.x.min(axis=0), mesh.x.max(axis=0)))
    dm._occupancy = mesh.lattice_occupancy()
    if facets or getattr(mesh, "_bfacets", None) is not None:
        dm._bfacets = np.ascontiguousarray(mesh.boundary_facet_mask()[local.cell_global])
    return dm


def local_unit_mesh(n: int, dim: int, rank: int, nranks: int, jitter: float = 0.0, seed: int = 20240807) -> DistMesh:
    """Rank ``rank``'s block of the n^dim unit square / cube, generated locally (dist/structured.py): host
    memory and set-up time per rank scale like 1/nranks.  ``_occupancy`` is the LOCAL value; callers that run
    several ranks take the maximum over ranks before the first solve (bench_distributed does)."""
    from .structured import local_structured
    local = local_structured(n, dim, rank, nranks, jitter, seed)
    n1 = n + 1
    dm = DistMesh(local, n1 ** dim, (2 if dim == 2 else 6) * n ** dim, bbox=(np.zeros(dim), np.ones(dim)), box_domain=True)
    dm.n = n
    return dm


def init_process_group(rank: int, world: int):
    """Control plane over gloo (127.0.0.1 rendezvous from MASTER_ADDR/MASTER_PORT)."""
    import torch.distributed as dist
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group("gloo", rank=rank, world_size=world)
    return dist


def init_comm(ctx, rank: int, world: int) -> None:
    """Create the RCCL communicator of ``ctx``: rank 0 makes the ncclUniqueId, gloo broadcasts it."""
    from ..engine import Context
    dist = init_process_group(rank, world)
    box = [Context.comm_unique_id() if rank == 0 else None]
    dist.broadcast_object_list(box, src=0)
    ctx.comm_init(box[0], rank, world)


class _quiet_stdout:
    """gloo and RCCL print banners on the C-level stdout; bench.py's contract is ONE JSON line there.
    Route fd 1 to stderr while the run is set up and timed, restore it for the result."""

    def __enter__(self):
        import sys
        sys.stdout.flush()
        self._saved = os.dup(1)
        os.dup2(2, 1)
        return self

    def __exit__(self, *exc):
        import ctypes
        import sys
        sys.stdout.flush()
        ctypes.CDLL(None).fflush(None)          # RCCL's banner sits in the C stdio buffer until exit otherwise
        os.dup2(self._saved, 1)
        os.close(self._saved)
        return False


class TorchControl:
    """Control plane of a multi-process run: torch.distributed over gloo (rendezvous, barriers, small host
    reductions, the ncclUniqueId broadcast).  Never carries field data."""

    def __init__(self, rank: int, world: int):
        self.rank, self.world = rank, world
        self.dist = init_process_group(rank, world)

    def init_comm(self, ctx) -> None:
        init_comm(ctx, self.rank, self.world)

    def barrier(self) -> None:
        self.dist.barrier()

    def allreduce(self, values, op: str = "sum") -> np.ndarray:
        import torch
        t = torch.tensor(np.asarray(values, dtype=np.float64).ravel())
        self.dist.all_reduce(t, op={"sum": self.dist.ReduceOp.SUM, "max": self.dist.ReduceOp.MAX}[op])
        return t.numpy()

    def gather(self, values) -> np.ndarray:
        import torch
        t = torch.tensor(np.asarray(values, dtype=np.float64).ravel())
        out = [torch.zeros_like(t) for _ in range(self.world)]
        self.dist.all_gather(out, t)
        return np.stack([o.numpy() for o in out])

    def broadcast(self, values, n: int, src: int = 0) -> np.ndarray:
        """``values`` (length n, rank ``src`` only; None elsewhere) to every rank: checker data, outside any timed region."""
        import torch
        t = torch.from_numpy(np.ascontiguousarray(values, dtype=np.float64).ravel().copy()) if self.rank == src else torch.empty(int(n), dtype=torch.float64)
        self.dist.broadcast(t, src=src)
        return t.numpy()


class ThreadControl:
    """The same interface for ranks that are host threads of one process (rank emulation on one GPU, tests)."""

    class Shared:
        def __init__(self, world: int):
            import threading
            self.world = world
            self.barrier = threading.Barrier(world, timeout=300)
            self.slots = [None] * world

    def __init__(self, rank: int, shared: "ThreadControl.Shared", group):
        self.rank, self.world, self._s, self._group = rank, shared.world, shared, group

    def init_comm(self, ctx) -> None:
        ctx.comm_emulate(self._group, self.rank)

    def barrier(self) -> None:
        self._s.barrier.wait()

    def gather(self, values) -> np.ndarray:
        self._s.slots[self.rank] = np.asarray(values, dtype=np.float64).ravel().copy()
        self._s.barrier.wait()
        out = np.stack(self._s.slots)
        self._s.barrier.wait()
        return out

    def allreduce(self, values, op: str = "sum") -> np.ndarray:
        g = self.gather(values)
        return g.sum(axis=0) if op == "sum" else g.max(axis=0)

    def broadcast(self, values, n: int, src: int = 0) -> np.ndarray:
        if self.rank == src:
            self._s.slots[src] = np.ascontiguousarray(values, dtype=np.float64).ravel()
        self._s.barrier.wait()
        out = self._s.slots[src]
        self._s.barrier.wait()
        return out


def bench_distributed(args, rank: int, world: int, local_rank: int):
    """bench.py for N > 1: the 10 M-DOF mesh is partitioned over the ranks (strong scaling).
    Returns the result line on rank 0 (None elsewhere); bench.py prints it.  A rank that fails takes the job
    down instead of leaving the others in a barrier."""
    from ..engine import Context
    from ..fea import utils_hip
    control = TorchControl(rank, world)
    try:
        ctx = Context(local_rank)
        utils_hip.set_context(ctx)
        control.init_comm(ctx)
        result = run_distributed_bench(args, ctx, control)
        control.barrier()
        return result
    except BaseException:                        # noqa: BLE001
        import sys
        import traceback
        traceback.print_exc()
        sys.stderr.flush()
        os._exit(1)                              # torchrun tears the other ranks down


def run_distributed_bench(args, ctx, control, cpu_baseline: bool = True):
    """One rank's part of the N > 1 benchmark: rank-local mesh, the operator stack on it, the timed cycles,
    the JSON line on rank 0.  ``ctx`` already has its communicator (RCCL or emulated)."""
    import bench as B
    from ..engine import Vec, host_wait as E_host_wait, pinned_array
    from ..fea import utils_hip

    rank, world = control.rank, control.world
    utils_hip.KSP_OPTIONS["pc"] = B.PC = getattr(args, "pc", "bpx")
    t0 = time.perf_counter()
    rss0 = _rss_mb()
    mesh = local_unit_mesh(args.n, 3, rank, world, jitter=getattr(args, "jitter", 0.0))
    mesh._occupancy = float(control.allreduce([mesh.lattice_occupancy()], "max")[0])    # one choice of preconditioner for all
    n_dof, n_cell_g = mesh.n_vert_global, mesh.n_cell_global
    # like N = 1: NumPy arrays at the operator boundary, every rank holding its own share of f, u and dJ/df in
    # pinned host memory (a distributed driver; each GPU has its own PCIe link)
    sim, fea = B.build_problem(mesh, device=False)
    dm = mesh.device(ctx)
    K, W = args.steps, args.warmup
    f_host = [pinned_array(f) for f in B.source_fields(mesh, min(K + W, 4))]
    from ..engine import pinned_full
    u0 = pinned_full(mesh.n_vert, 0.0)
    setup_s = time.perf_counter() - t0
    rss_setup = _rss_mb() - rss0

    # pinned pool sized during set-up (see bench.py: two generations of result blocks are alive at a time)
    from ..engine import pinned_empty
    n_f, n_u = int(np.size(sim['f'])), int(np.size(sim['u']))          # sizes of the arrays that cross the boundary
    prime = [pinned_empty(n_f) for _ in range(4)] + [pinned_empty(n_u) for _ in range(8)]
    del prime
    g = None
    for w in range(W):
        g = B.one_cycle(sim, fea, f_host[w % len(f_host)], u0)
    ctx.sync()
    control.barrier()
    if rank == 0:
        del utils_hip.LAST_KSP_INFO[:]
    control.barrier()
    ctx.comm_stats(reset=True)
    t0 = time.perf_counter()
    for k in range(K):
        g = B.one_cycle(sim, fea, f_host[(W + k) % len(f_host)], u0)   # the gradient is the caller's: held until replaced
    ctx.sync()
    control.barrier()
    elapsed = float(control.allreduce([time.perf_counter() - t0], "max")[0])
    comm = ctx.comm_stats()
    ms_per_step = elapsed / max(K, 1) * 1e3

    import threading
    me = threading.get_ident()                               # emulated ranks are threads sharing the module's log
    infos = [i for i in utils_hip.LAST_KSP_INFO if i.get("thread") == me]
    per = len(infos) // K if K else 0
    its_per_step = [i["iterations"] for i in infos[:per]]
    cg_ms = sum(i["solve_ms"] for i in infos) / max(K, 1)
    A_mat = [w[1] for k, w in utils_hip._WORK.items() if k[1] == "newton_A" and w[0] is mesh][0].mat
    xv = Vec(ctx, mesh.n_vert).set(np.random.default_rng(rank).standard_normal(mesh.n_vert))
    yv = Vec(ctx, mesh.n_vert)
    control.barrier()
    spmv_ms = min(A_mat.bench_spmv(xv, yv, 50) for _ in range(3))      # local rows only, no halo: the kernel's own rate
    local_nnz = dm.info["nnz"]
    B_A = B.spmv_algorithmic_bytes(local_nnz, mesh.n_owned)
    achieved = B_A / (spmv_ms * 1e-3) / 1e9
    L = mesh.local
    lat = dm.pc_info()
    # bytes a rank moves per CG iteration besides its HBM traffic: ghost values in and out, the lattice all-reduce
    halo_out, halo_in = int(L.send_ptr[-1]) * 8, int(L.recv_ptr[-1]) * 8
    stats = control.gather([mesh.n_owned, mesh.n_vert - mesh.n_owned, len(L.nbr), halo_out, halo_in, achieved, spmv_ms,
                            B.stored_bytes(dm.info, mesh.n_owned), rss_setup, setup_s])
    # ---- self-check of the partitioned run (outside the timed region): one more cycle; rank 0 computes the DST-exact
    # cycle of the WHOLE mesh (oracle/c_port.py::poisson_cycle_dst, no iterative solve), every rank compares the entries
    # it owns, the largest error over the ranks goes into the record.  Structured cube only.
    check = None
    if not getattr(args, "no_check", False) and not getattr(args, "jitter", 0.0):
        kc = (W + K) % len(f_host)
        g_chk = np.array(E_host_wait(B.one_cycle(sim, fea, f_host[kc], u0)), copy=True)
        u_chk = np.array(sim['u'], copy=True)
        n_cell_g = mesh.n_cell_global
        ref_u = ref_g = None
        if rank == 0:
            ref_u, ref_g = B.dst_reference_cycle(args.n, kc, min(K + W, 4))     # the checker lives with the harness, not in the package
        ref_u = control.broadcast(ref_u, n_dof)
        ref_g = control.broadcast(ref_g, n_cell_g)
        own_v, own_c = L.vert_global[:L.n_owned], L.cell_global[L.cell_owned]
        eu = float(np.abs(u_chk[:L.n_owned] - ref_u[own_v]).max()) if L.n_owned else 0.0
        eg = float(np.abs(g_chk[L.cell_owned] - ref_g[own_c]).max()) if own_c.size else 0.0
        errs = control.allreduce([eu, eg], "max")
        check = {"u_rel_err": float(errs[0] / np.abs(ref_u).max()), "grad_rel_err": float(errs[1] / np.abs(ref_g).max()),
                 "tolerance": 1e-10, "norm": "max over all ranks' owned entries, relative to the largest entry",
                 "against": "DST-exact cycle of the whole mesh on rank 0 (oracle/c_port.py::poisson_cycle_dst)"}
        del ref_u, ref_g
    if rank != 0:
        return None
    result = {
        "metric": B.METRIC, "value": n_dof / (ms_per_step * 1e-3), "unit": "DOFs/s",
        "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": ms_per_step,
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {
            "workload": (f"3-D linear Poisson, P1 tets, unit cube n={args.n}: {n_dof} DOFs, {n_cell_g} cells, "
                         f"{world}-way block partition {'x'.join(str(g) for g in _grid(world))} (rank-local mesh generation), "
                         f"ghost-DOF halo (ncclSend/Recv) + RCCL all-reduce {B.PC.upper()}-CG; same cycle as N=1, every rank with "
                         f"NumPy arrays of its share of f, u and dJ/df at the operator boundary"),
            "boundary": "host (per rank: NumPy in pinned blocks; H2D + D2H inside the timed region)",
            "preconditioner": B.PC, "pc_lattice": lat,
            "n": args.n, "n_dof": n_dof, "n_cell": n_cell_g, "parallelism": f"block{world}",
            "owned_per_rank": [int(s[0]) for s in stats], "ghosts_per_rank": [int(s[1]) for s in stats],
            "neighbours_per_rank": [int(s[2]) for s in stats],
            "halo_bytes_sent_per_exchange_per_rank": [int(s[3]) for s in stats],
            "halo_bytes_received_per_exchange_per_rank": [int(s[4]) for s in stats],
            # counted on rank 0 over the timed cycles (femo_comm_stats); the merged loop issues one all-reduce and one
            # halo exchange per ENQUEUED iteration (batches: a few iterations behind the converged one are enqueued too)
            "allreduce_per_cg_iteration": (sum(i.get("loop_allreduces", 0) for i in infos) / max(sum(i["iterations"] for i in infos), 1)),
            "collectives_per_step_rank0": {k: v / max(K, 1) for k, v in comm.items()},
            "allreduce_payload": "shared finest-lattice nodes + levels L-1, L-2 whole + 7 scalars (p.q, r.q, q.q, r.r, 3 lattice sums) in ONE ncclAllReduce per iteration",
            "linear_solves_per_step": per, "cg_iterations_per_step": its_per_step, "cg_ms_per_step": cg_ms,
            "non_cg_ms_per_step": ms_per_step - cg_ms,
            "setup_s_per_rank": [float(s[9]) for s in stats], "setup_rss_mb_per_rank": [float(s[8]) for s in stats],
        },
        "roofline": {
            "bound": "hbm", "achieved": float(stats[0][5]), "peak": B.HBM_PEAK_GBS, "unit": "GB/s",
            "frac": float(stats[0][5]) / B.HBM_PEAK_GBS, "traffic": None,
            "frac_physical": float(stats[0][7]) / (float(stats[0][6]) * 1e-3) / 1e9 / B.HBM_PEAK_GBS,
            "physical_bytes_source": "stored bytes of the SELL format (no PMC pass at N > 1)",
            "kernel": "k_spmv_sell<1,true> on rank 0's local rows (per GPU), 150 back-to-back launches",
            "algorithmic_bytes_per_launch": B_A, "avg_launch_ms": float(stats[0][6]), "launches_timed": 150,
            "achieved_per_rank": [float(s[5]) for s in stats],
        },
    }
    if check is not None:
        result["check"] = check
    if cpu_baseline and not getattr(args, "no_cpu_baseline", False):
        counts = its_per_step if its_per_step else [0]
        n = args.n
        nnz = n_dof + 2 * (3 * n * (n + 1) ** 2 + 3 * n * n * (n + 1) + n ** 3)
        result["cpu_baseline"] = B.cpu_baseline(args, counts, n_dof, n_cell_g, nnz)
    return result


def _grid(world: int):
    from .structured import process_grid
    return process_grid(world, 3)


def _rss_mb() -> float:
    try:
        with open("/proc/self/statm") as fh:
            return int(fh.read().split()[1]) * os.sysconf("SC_PAGE_SIZE") / 1e6
    except Exception:
        return 0.0
