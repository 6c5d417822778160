"""Mesh partitioning and ghost-DOF halo plans (host side, NumPy only).

New design -- the reference is single-rank (SURVEY.md section 0 finding 3, section 8(e)).
Rows (vertices) are owned by exactly one rank; a rank keeps every cell that
touches one of its vertices (one layer of ghost cells, computed redundantly, so
no matrix entries are exchanged) and ghost copies of the other vertices of those
cells.  Local numbering: owned vertices first (ascending global id), then
ghosts grouped by owner rank (ascending), ascending global id inside a group,
which is exactly the order the owner packs its send buffer in.

METIS is not installed (BASELINE.json asks for it); the partitioner is recursive
coordinate bisection (RCB), which yields the 2x2x2 block decomposition on the
benchmark cube.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List

import numpy as np


def rcb_partition(x: np.ndarray, nparts: int) -> np.ndarray:
    """Recursive coordinate bisection of points ``x`` (n, d) into ``nparts`` balanced
    parts (any nparts >= 1).  Returns the owner of every point (int32)."""
    n = x.shape[0]
    part = np.zeros(n, dtype=np.int32)
    stack = [(np.arange(n), 0, nparts)]
    while stack:
        idx, first, k = stack.pop()
        if k == 1 or idx.size == 0:
            part[idx] = first
            continue
        kl = k // 2
        pts = x[idx]
        ext = pts.max(axis=0) - pts.min(axis=0)
        ax = int(np.argmax(ext))
        n_left = int(round(idx.size * kl / k))
        # stable order: coordinate, then index -> deterministic on structured grids
        order = np.lexsort((idx, pts[:, ax]))
        stack.append((idx[order[:n_left]], first, kl))
        stack.append((idx[order[n_left:]], first + kl, k - kl))
    return part


@dataclass
class LocalMesh:
    """What one rank needs: its piece of the mesh in local numbering + the halo plan."""
    rank: int
    nranks: int
    x: np.ndarray               # (n_local_vert, d)
    conn: np.ndarray            # (n_local_cell, d+1) local vertex ids
    n_owned: int
    vert_global: np.ndarray     # local vertex -> global vertex
    cell_global: np.ndarray     # local cell -> global cell
    cell_owned: np.ndarray      # bool: cell's first vertex is owned (DG0 values owned by this rank)
    nbr: np.ndarray             # neighbour ranks (ascending)
    send_ptr: np.ndarray        # (n_nbr+1,) int64
    send_idx: np.ndarray        # owned local ids, packed per neighbour
    recv_ptr: np.ndarray        # (n_nbr+1,) int64; ghost slot = n_owned + recv_ptr[k] + j


def build_local_mesh(x: np.ndarray, conn: np.ndarray, part: np.ndarray, rank: int, nranks: int) -> LocalMesh:
    """Extract rank ``rank``'s local mesh from the global mesh and the vertex owner map.
    Every rank can call this independently: the plans are consistent by construction."""
    conn = np.asarray(conn)
    owner_of_conn = part[conn]                                    # (n_cell, d+1)
    mine = owner_of_conn == rank
    cell_mask = mine.any(axis=1)
    cell_global = np.nonzero(cell_mask)[0]
    lconn_g = conn[cell_global]                                   # global ids of local cells
    owned_g = np.nonzero(part == rank)[0]
    touched = np.unique(lconn_g)
    ghost_g = touched[part[touched] != rank]
    # ghosts grouped by owner, ascending global id inside a group
    order = np.lexsort((ghost_g, part[ghost_g]))
    ghost_g = ghost_g[order]
    ghost_owner = part[ghost_g]
    nbr_recv = np.unique(ghost_owner)
    # send lists: my owned vertices that appear in a cell together with a vertex owned by q
    # (those cells are local on q, so q holds these vertices as ghosts)
    local_owner = owner_of_conn[cell_global]                      # (n_lc, d+1)
    send = {}
    for q in np.unique(local_owner):
        if q == rank:
            continue
        cells_q = (local_owner == q).any(axis=1)
        sub = lconn_g[cells_q]
        v = np.unique(sub[local_owner[cells_q] == rank])
        if v.size:
            send[int(q)] = v
    nbr = np.array(sorted(set(send) | set(int(q) for q in nbr_recv)), dtype=np.int32)
    vert_global = np.concatenate([owned_g, ghost_g]).astype(np.int64)
    g2l = np.full(x.shape[0], -1, dtype=np.int64)
    g2l[vert_global] = np.arange(vert_global.size)
    send_ptr = np.zeros(len(nbr) + 1, dtype=np.int64)
    recv_ptr = np.zeros(len(nbr) + 1, dtype=np.int64)
    send_idx: List[np.ndarray] = []
    for k, q in enumerate(nbr):
        sv = send.get(int(q), np.zeros(0, dtype=np.int64))
        send_idx.append(g2l[sv])
        send_ptr[k + 1] = send_ptr[k] + sv.size
        recv_ptr[k + 1] = recv_ptr[k] + int(np.count_nonzero(ghost_owner == q))
    # ghosts are grouped by ascending owner and nbr is ascending: offsets line up
    assert recv_ptr[-1] == ghost_g.size
    lconn = g2l[lconn_g].astype(np.int32)
    return LocalMesh(
        rank=rank, nranks=nranks, x=np.ascontiguousarray(x[vert_global]), conn=np.ascontiguousarray(lconn),
        n_owned=int(owned_g.size), vert_global=vert_global, cell_global=cell_global.astype(np.int64),
        cell_owned=(local_owner[:, 0] == rank), nbr=nbr, send_ptr=send_ptr,
        send_idx=(np.concatenate(send_idx) if send_idx else np.zeros(0, np.int64)).astype(np.int32),
        recv_ptr=recv_ptr)
