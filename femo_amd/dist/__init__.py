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

    def __init__(self, local: LocalMesh, n_vert_global: int, n_cell_global: int, bbox=None):
        super().__init__(local.x, local.conn)
        self.local = local
        self.n_vert_global = int(n_vert_global)
        self.n_cell_global = int(n_cell_global)
        self.bbox = bbox        # (lo, hi) of the whole mesh: every rank builds the same BPX lattice
        self._global_mesh = None

    @property
    def n_owned(self) -> int:
        return self.local.n_owned

    def boundary_facet_mask(self) -> np.ndarray:
        """Exterior facets are those of the WHOLE mesh: the cut faces of the partition are not a
        boundary (a facet seen once among the local cells may have its other cell on another rank).
        Computed on first use from the global mesh -- only forms with facet terms (Nitsche) ask."""
        if getattr(self, "_bfacets", None) is None:
            g = getattr(self, "_global_mesh", None)
            if g is None:
                if self.local.nranks > 1:
                    raise RuntimeError("DistMesh without its global mesh cannot tell exterior facets from partition cuts")
                return super().boundary_facet_mask()
            self._bfacets = np.ascontiguousarray(g.boundary_facet_mask()[self.local.cell_global])
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


def partition_mesh(mesh: Mesh, rank: int, nranks: int) -> DistMesh:
    part = rcb_partition(mesh.x, nranks)
    local = build_local_mesh(mesh.x, mesh.conn, part, rank, nranks)
    dm = DistMesh(local, mesh.n_vert, mesh.n_cell, bbox=(mesh.x.min(axis=0), mesh.x.max(axis=0)))
    dm._occupancy = mesh.lattice_occupancy()      # of the WHOLE mesh: every rank must pick the same preconditioner
    dm._global_mesh = mesh                        # for boundary_facet_mask(), computed only if a form needs facets
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


def bench_distributed(args, rank: int, world: int, local_rank: int):
    """bench.py for N > 1: the 10 M-DOF mesh is partitioned over the ranks (strong scaling).
    Returns the result line on rank 0 (None elsewhere); bench.py prints it."""
    result = _bench_distributed(args, rank, world, local_rank)
    import torch.distributed as dist
    dist.barrier()
    return result


def _bench_distributed(args, rank: int, world: int, local_rank: int):
    import bench as B
    from ..engine import Context, DeviceArray, Vec
    from ..fea import utils_hip
    from ..fea.mesh import createUnitCubeMesh

    dist = init_process_group(rank, world)
    import torch
    ctx = Context(local_rank)
    utils_hip.set_context(ctx)
    utils_hip.KSP_OPTIONS["pc"] = B.PC = getattr(args, "pc", "bpx")
    init_comm(ctx, rank, world)
    t0 = time.perf_counter()
    gmesh = createUnitCubeMesh(args.n, jitter=getattr(args, 'jitter', 0.0))
    n_dof, n_cell_g = gmesh.n_vert, gmesh.n_cell
    mesh = partition_mesh(gmesh, rank, world)
    del gmesh
    sim, fea = B.build_problem(mesh, device=True)
    dm = mesh.device(ctx)
    K, W = args.steps, args.warmup
    f_host = B.source_fields(mesh, min(K + W, 4))
    f_dev = [DeviceArray(Vec(ctx, mesh.n_cell).set(f)) for f in f_host]
    setup_s = time.perf_counter() - t0

    for w in range(W):
        B.one_cycle(sim, fea, f_dev[w % len(f_dev)])
    ctx.sync()
    dist.barrier()
    del utils_hip.LAST_KSP_INFO[:]
    t0 = time.perf_counter()
    for k in range(K):
        B.one_cycle(sim, fea, f_dev[(W + k) % len(f_dev)])
    ctx.sync()
    dist.barrier()
    elapsed = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    dist.all_reduce(elapsed, op=dist.ReduceOp.MAX)
    ms_per_step = float(elapsed[0]) / K * 1e3

    infos = list(utils_hip.LAST_KSP_INFO)
    its_per_step = [i["iterations"] for i in infos[:len(infos) // K]] if K else []
    cg_ms = sum(i["solve_ms"] for i in infos) / max(K, 1)
    A_mat = [w[1] for k, w in utils_hip._WORK.items() if k[1] == "newton_A"][0].mat
    xv = Vec(ctx, mesh.n_vert).set(np.random.default_rng(rank).standard_normal(mesh.n_vert))
    yv = Vec(ctx, mesh.n_vert)
    spmv_ms = min(A_mat.bench_spmv(xv, yv, 50) for _ in range(3))
    local_nnz = dm.info["nnz"]
    B_A = B.spmv_algorithmic_bytes(local_nnz, mesh.n_owned)
    achieved = B_A / (spmv_ms * 1e-3) / 1e9
    stats = torch.tensor([mesh.n_owned, mesh.n_vert - mesh.n_owned, len(mesh.local.nbr)], dtype=torch.float64)
    gathered = [torch.zeros_like(stats) for _ in range(world)]
    dist.all_gather(gathered, stats)
    if rank == 0:
        result = {
            "metric": B.METRIC, "value": n_dof / (ms_per_step * 1e-3), "unit": "DOFs/s",
            "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {
                "workload": (f"3-D linear Poisson, P1 tets, unit cube n={args.n}: {n_dof} DOFs, {n_cell_g} cells, "
                             f"RCB {world}-way vertex partition, ghost-DOF halo (ncclSend/Recv) + RCCL all-reduce "
                             f"{B.PC.upper()}-CG; same cycle as N=1"),
                "preconditioner": B.PC, "pc_lattice": dm.pc_info(),
                "n": args.n, "n_dof": n_dof, "n_cell": n_cell_g, "parallelism": f"rcb{world}",
                "owned_per_rank": [int(g[0]) for g in gathered], "ghosts_per_rank": [int(g[1]) for g in gathered],
                "neighbours_per_rank": [int(g[2]) for g in gathered],
                "cg_iterations_per_step": its_per_step, "cg_ms_per_step": cg_ms,
                "non_cg_ms_per_step": ms_per_step - cg_ms, "setup_s": setup_s,
            },
            "roofline": {
                "bound": "hbm", "achieved": achieved, "peak": B.HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / B.HBM_PEAK_GBS, "traffic": None,
                "kernel": "k_spmv_sell<true,true> on rank 0's local rows (per GPU)",
                "algorithmic_bytes_per_launch": B_A, "avg_launch_ms": spmv_ms, "launches_timed": 150,
            },
        }
        return result
    return None
