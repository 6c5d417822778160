"""Rank-local generation of the benchmark meshes: every rank builds only its own block of the unit square /
cube (owned vertices, one layer of ghost cells, ghost vertices, halo plan), never the whole mesh.

Round 1 built the full 10 M-DOF mesh on every rank and partitioned it there (8 x the host memory and ~17 s per
rank); here host memory and set-up time per rank scale like 1/N.  Vertex and cell numbering, coordinates and
the seeded jitter are those of ``fea.mesh.createUnitCubeMesh`` / ``createUnitSquareMesh``, so a rank's local
mesh is exactly what ``build_local_mesh`` extracts from the whole mesh for the same owner map (tested).

The owner map is a block decomposition -- what recursive coordinate bisection gives on these grids when the
cuts fall between grid planes (METIS, which BASELINE.json names, is not installed): ``nranks`` is factorised
into a process grid over the slower axes -- the fastest axis stays whole on one node (``process_grid``): 2 ranks
cut z, 4 ranks z and y, 8 ranks are 1 x 2 x 4 pencils with at most 5 neighbours each.
"""
from __future__ import annotations

from typing import List, Sequence, Tuple

import numpy as np

from .partition import LocalMesh

_KUHN = ((0, 1, 2), (0, 2, 1), (1, 0, 2), (1, 2, 0), (2, 0, 1), (2, 1, 0))      # fea/mesh.py


def process_grid(nranks: int, dim: int) -> Tuple[int, ...]:
    """Factorisation of nranks into ``dim`` factors (x, y[, z]), as even as possible, larger factors on the slower axes
    (slabs of the slowest axis are contiguous in the lexicographic numbering).

    Round 5: **the fastest axis is never cut on one node** (up to 8 ranks in 2-D, 64 in 3-D): 8 ranks are 1 x 2 x 4 pencils,
    not 2 x 2 x 2 blocks.  A SELL slice is 64 consecutive vertices of an x-line; with a cut x-face every x-line ends in a
    vertex that couples to ghosts, so ~60 % of a block's slices read ghost columns -- they lose the regular (index-free)
    SpMV path and all count as "boundary" slices that must wait for the halo exchange (measured on rank 0's 108^3 block of
    the 215^3 cube: interior launch 16 us, boundary launch 29 us, 45 us together against 30 us for the same rows without a
    cut x-face).  With x whole the ghost-coupled rows are whole x-lines on the y- and z-faces: ~5 % of the slices.  The
    price is surface: 58 k ghosts instead of 35 k on a rank of the 215^3 cube (latency-bound messages either way) and at
    most 5 neighbours instead of 7."""
    grid = [1] * dim
    rest = nranks
    p = 2
    factors: List[int] = []
    while rest > 1:
        while rest % p == 0:
            factors.append(p)
            rest //= p
        p += 1
    keep_x = dim >= 2 and nranks <= (8 if dim == 2 else 64)
    axes = range(1, dim) if keep_x else range(dim)
    for f in sorted(factors, reverse=True):
        k = min(axes, key=lambda a: (grid[a], -a))          # the smallest factor so far, slowest axis first
        grid[k] *= f
    return tuple(grid)


def axis_ranges(n1: int, p: int) -> np.ndarray:
    """Boundaries of p balanced contiguous pieces of range(n1): piece k is [b[k], b[k+1])."""
    return np.array([(n1 * k) // p for k in range(p + 1)], dtype=np.int64)


class BlockOwner:
    """Owner rank of the vertices of the (n+1)^dim grid, by global vertex id (vectorised)."""

    def __init__(self, n: int, dim: int, nranks: int):
        self.n, self.dim, self.nranks = n, dim, nranks
        self.grid = process_grid(nranks, dim)
        self.bounds = [axis_ranges(n + 1, p) for p in self.grid]

    def coords(self, gid: np.ndarray) -> List[np.ndarray]:
        n1 = self.n + 1
        out, rest = [], np.asarray(gid, dtype=np.int64)
        for _ in range(self.dim):
            out.append(rest % n1)
            rest = rest // n1
        return out

    def __call__(self, gid: np.ndarray) -> np.ndarray:
        ijk = self.coords(gid)
        rank = np.zeros(np.shape(gid), dtype=np.int64)
        stride = 1
        for a in range(self.dim):
            rank += (np.searchsorted(self.bounds[a], ijk[a], side="right") - 1) * stride
            stride *= self.grid[a]
        return rank.astype(np.int32)

    def block(self, rank: int) -> List[Tuple[int, int]]:
        """[lo, hi) of the owned vertex indices per axis."""
        out, rest = [], rank
        for a in range(self.dim):
            k = rest % self.grid[a]
            rest //= self.grid[a]
            out.append((int(self.bounds[a][k]), int(self.bounds[a][k + 1])))
        return out


def _grid_coordinates(n: int, dim: int, gid: np.ndarray, jitter: float, seed: int) -> np.ndarray:
    """Coordinates of the vertices ``gid`` exactly as createUnitSquareMesh / createUnitCubeMesh place them."""
    n1 = n + 1
    g = np.arange(n1) / n
    g[-1] = 1.0
    x = np.empty((gid.size, dim))
    rest = gid.astype(np.int64)
    for a in range(dim):
        x[:, a] = g[rest % n1]
        rest = rest // n1
    if jitter != 0.0:
        # fea.mesh._apply_jitter draws one (n_vert, dim) block from the seeded generator; the same stream is
        # drawn here and indexed (the only array of global size this module touches: 8 dim bytes per vertex)
        rng = np.random.default_rng(seed)
        d = rng.uniform(-1.0, 1.0, size=(n1 ** dim, dim))[gid] * (jitter / n)
        interior = np.all((x > 1e-9) & (x < 1.0 - 1e-9), axis=1)
        x[interior] += d[interior]
    return x


def _candidate_cells(n: int, dim: int, box: Sequence[Tuple[int, int]]):
    """Global ids and connectivity (global vertex ids) of the simplices of every grid cell that touches the
    vertex box: grid cells [lo-1, hi) per axis, clipped to the grid."""
    n1 = n + 1
    ranges = [np.arange(max(lo - 1, 0), min(hi, n), dtype=np.int64) for lo, hi in box]
    if dim == 2:
        ii, jj = np.meshgrid(ranges[0], ranges[1], indexing="ij")
        ii, jj = ii.ravel(), jj.ravel()
        cube = jj * n + ii
        v0 = jj * n1 + ii
        conn = np.empty((cube.size, 2, 3), dtype=np.int64)
        conn[:, 0, 0] = v0; conn[:, 0, 1] = v0 + 1; conn[:, 0, 2] = v0 + n1 + 1
        conn[:, 1, 0] = v0; conn[:, 1, 1] = v0 + n1 + 1; conn[:, 1, 2] = v0 + n1
        cell = cube[:, None] * 2 + np.arange(2)[None, :]
        return cell.ravel(), conn.reshape(-1, 3)
    ii, jj, kk = np.meshgrid(ranges[0], ranges[1], ranges[2], indexing="ij")
    ii, jj, kk = ii.ravel(), jj.ravel(), kk.ravel()
    cube = kk * n * n + jj * n + ii
    base = kk * n1 * n1 + jj * n1 + ii
    stride = (1, n1, n1 * n1)
    conn = np.empty((cube.size, 6, 4), dtype=np.int64)
    for t, perm in enumerate(_KUHN):
        v = base.copy()
        conn[:, t, 0] = v
        for s, ax in enumerate(perm):
            v = v + stride[ax]
            conn[:, t, s + 1] = v
    cell = cube[:, None] * 6 + np.arange(6)[None, :]
    return cell.ravel(), conn.reshape(-1, 4)


def local_structured(n: int, dim: int, rank: int, nranks: int, jitter: float = 0.0, seed: int = 20240807) -> LocalMesh:
    """Rank ``rank``'s piece of the n^dim unit mesh under the block owner map, built from local data only.
    Same local numbering and halo-plan conventions as ``partition.build_local_mesh``."""
    owner = BlockOwner(n, dim, nranks)
    box = owner.block(rank)
    cell_g, conn_g = _candidate_cells(n, dim, box)
    own_conn = owner(conn_g)                                        # (n_cand, dim+1)
    keep = (own_conn == rank).any(axis=1)
    cell_g, conn_g, own_conn = cell_g[keep], conn_g[keep], own_conn[keep]
    order = np.argsort(cell_g, kind="stable")                       # ascending global cell id, like build_local_mesh
    cell_g, conn_g, own_conn = cell_g[order], conn_g[order], own_conn[order]
    touched = np.unique(conn_g)
    t_owner = owner(touched)
    owned_g = touched[t_owner == rank]
    # every vertex of the box is touched by a local cell except on degenerate grids; take the box itself
    n1 = n + 1
    axes = [np.arange(lo, hi, dtype=np.int64) for lo, hi in box]
    if dim == 2:
        jj, ii = np.meshgrid(axes[1], axes[0], indexing="ij")
        box_g = (jj * n1 + ii).ravel()
    else:
        kk, jj, ii = np.meshgrid(axes[2], axes[1], axes[0], indexing="ij")
        box_g = (kk * n1 * n1 + jj * n1 + ii).ravel()
    owned_g = np.union1d(owned_g, box_g)
    ghost_g = touched[t_owner != rank]
    ghost_owner = t_owner[t_owner != rank]
    go = np.lexsort((ghost_g, ghost_owner))
    ghost_g, ghost_owner = ghost_g[go], ghost_owner[go]
    vert_global = np.concatenate([owned_g, ghost_g])
    sorter = np.argsort(vert_global, kind="stable")

    def g2l(g):
        pos = np.searchsorted(vert_global, g, sorter=sorter)
        return sorter[pos]

    send = {}
    for q in np.unique(own_conn):
        if q == rank:
            continue
        cells_q = (own_conn == q).any(axis=1)
        v = np.unique(conn_g[cells_q][own_conn[cells_q] == rank])
        if v.size:
            send[int(q)] = v
    nbr = np.array(sorted(set(send) | set(int(q) for q in np.unique(ghost_owner))), dtype=np.int32)
    send_ptr = np.zeros(len(nbr) + 1, dtype=np.int64)
    recv_ptr = np.zeros(len(nbr) + 1, dtype=np.int64)
    send_idx = []
    for k, q in enumerate(nbr):
        sv = send.get(int(q), np.zeros(0, dtype=np.int64))
        send_idx.append(g2l(sv))
        send_ptr[k + 1] = send_ptr[k] + sv.size
        recv_ptr[k + 1] = recv_ptr[k] + int(np.count_nonzero(ghost_owner == q))
    x = _grid_coordinates(n, dim, vert_global, jitter, seed)
    return LocalMesh(
        rank=rank, nranks=nranks, x=np.ascontiguousarray(x), conn=np.ascontiguousarray(g2l(conn_g).astype(np.int32)),
        n_owned=int(owned_g.size), vert_global=vert_global.astype(np.int64), cell_global=cell_g.astype(np.int64),
        cell_owned=(own_conn[:, 0] == rank), nbr=nbr, send_ptr=send_ptr,
        send_idx=(np.concatenate(send_idx) if send_idx else np.zeros(0, np.int64)).astype(np.int32), recv_ptr=recv_ptr)


def box_boundary_facets(x: np.ndarray, conn: np.ndarray, lo=0.0, hi=1.0, atol: float = 1e-9) -> np.ndarray:
    """Exterior-facet mask of cells of a mesh that fills the box [lo, hi]^dim: a facet is exterior iff all its
    vertices lie on one face of the box (valid for the generated squares and cubes, cut faces of a partition
    never qualify).  uint8 per cell, bit k = facet opposite local vertex k (Mesh.boundary_facet_mask)."""
    dim = x.shape[1]
    on = []                                                      # per (axis, side): vertex lies on that face
    for a in range(dim):
        on.append(np.abs(x[:, a] - lo) <= atol)
        on.append(np.abs(x[:, a] - hi) <= atol)
    mask = np.zeros(conn.shape[0], dtype=np.uint8)
    d1 = conn.shape[1]
    for k in range(d1):
        others = np.delete(conn, k, axis=1)
        ext = np.zeros(conn.shape[0], dtype=bool)
        for face in on:
            ext |= face[others].all(axis=1)
        mask |= ext.astype(np.uint8) << np.uint8(k)
    return mask
