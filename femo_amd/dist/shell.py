"""The shell on N ranks: partition of a `ShellSpace` and the rank-local arrays of `femo_shell_set_partition`.

New design -- the reference is single-rank (SURVEY.md section 0 finding 3); this is section 8(e)'s partitioning applied to
the CG2^3 x CG1^3 shell of BASELINE config 3 (missing #4).  The unit of ownership is the POINT: a P2 node
with its three displacements (vertices, then edge midpoints) or a vertex with its three rotations -- the node blocks of the
stiffness.  A vertex's two points go with the vertex, an edge midpoint with the edge's lower-numbered vertex, so every
point of a cell is owned by the owner of one of the cell's vertices, and a rank that keeps all cells touching one of its
vertices holds every cell around every point it owns: the rows of its points come out of the local assembly complete
and no matrix entry is ever exchanged (one layer of ghost cells computed redundantly, as in `dist/partition.py`).

Every rank holds the global mesh on the host (a surface mesh: 131 k vertices at BASELINE config 3) and cuts its part out
of it; plans are consistent by construction, nothing is negotiated.  The lattice of the preconditioner is the GLOBAL one
(`fea/shell.py::lattice_pc` on the global space): same nodes and numbering on every rank, each rank keeps the rows of its
local points, and sums over the ranks of P^T r and P^T K P are the serial objects.
"""
from __future__ import annotations

from typing import Optional

import numpy as np

from ..fea.shell import ShellSpace, lattice_pc
from .partition import rcb_partition


class ShellPartition:
    """Rank ``rank``'s part of ``space`` for the vertex owner map ``part`` (default: recursive coordinate bisection)."""

    def __init__(self, space: ShellSpace, rank: int, nranks: int, part: Optional[np.ndarray] = None):
        self.global_space, self.rank, self.nranks = space, int(rank), int(nranks)
        part = rcb_partition(space.x, nranks) if part is None else np.asarray(part, dtype=np.int32)
        if part.shape != (space.n_vert,) or part.min() < 0 or part.max() >= nranks:
            raise ValueError("ShellPartition: the owner map must give a rank in [0, nranks) for every vertex")
        self.part = part
        nv, nu = space.n_vert, space.n_unode
        # owner of every global point: displacement nodes (vertices, edges), then rotation vertices
        self.point_owner = np.concatenate([part, part[space.edge_vertices[:, 0]], part])
        conn = space.conn.astype(np.int64)
        cell_ranks = part[conn]                                                       # (nc, 3)
        self.cell_global = np.nonzero((cell_ranks == rank).any(axis=1))[0]
        lconn_g = conn[self.cell_global]
        self.vert_global = np.unique(lconn_g)                                          # ascending: local order = global order
        lconn = np.searchsorted(self.vert_global, lconn_g).astype(np.int32)
        self.space = ShellSpace(space.x[self.vert_global], lconn)
        L = self.space
        # local edge -> global edge through the (lower, upper) global vertex pair (both numberings sort by it)
        gkey = space.edge_vertices[:, 0] * nv + space.edge_vertices[:, 1]
        lkey = self.vert_global[L.edge_vertices[:, 0]] * nv + self.vert_global[L.edge_vertices[:, 1]]
        self.edge_global = np.searchsorted(gkey, lkey)
        if not np.array_equal(gkey[self.edge_global], lkey):
            raise RuntimeError("ShellPartition: a local edge is not an edge of the global mesh")
        # local point -> global point, local dof -> global dof (state layout of fea/shell.py on both sides)
        self.point_global = np.concatenate([self.vert_global, nv + self.edge_global, nu + self.vert_global])
        self.dof_global = (3 * self.point_global[:, None] + np.arange(3)[None, :]).ravel()
        self.owned_points = np.ascontiguousarray(self.point_owner[self.point_global] == rank, dtype=np.uint8)
        self.owned_dofs = np.repeat(self.owned_points.astype(bool), 3)
        # a cell is integrated by the owner of its first vertex (outputs: sum over owned cells, all-reduced)
        self.cell_owned = cell_ranks[self.cell_global, 0] == rank
        self._halo()

    def _halo(self) -> None:
        """Neighbour ranks and the dofs exchanged with each, both sides ordered by global point number."""
        sp, L, rank = self.global_space, self.space, self.rank
        cell_pts = np.concatenate([L.conn.astype(np.int64), L.n_vert + L.cell_edges.astype(np.int64),
                                   L.n_unode + L.conn.astype(np.int64)], axis=1)       # (n_lc, 9) local points of a cell
        cell_ranks = self.part[sp.conn[self.cell_global].astype(np.int64)]             # (n_lc, 3)
        owned = self.owned_points.astype(bool)
        ghost = np.nonzero(~owned)[0]
        ghost_owner = self.point_owner[self.point_global[ghost]]
        send = {}
        for q in np.unique(cell_ranks):
            if q == rank:
                continue
            # cells with a vertex of q are local on q: q holds copies of their points, mine among them
            pts = np.unique(cell_pts[(cell_ranks == q).any(axis=1)])
            pts = pts[owned[pts]]
            if pts.size:
                send[int(q)] = pts[np.argsort(self.point_global[pts], kind="stable")]
        nbr = sorted(set(send) | set(int(q) for q in np.unique(ghost_owner)))
        self.nbr = np.asarray(nbr, dtype=np.int32)
        sp_, rp_, sd, rd = [0], [0], [], []
        three = np.arange(3)[None, :]
        for q in nbr:
            s = send.get(q, np.zeros(0, dtype=np.int64))
            g = ghost[ghost_owner == q]
            g = g[np.argsort(self.point_global[g], kind="stable")]
            sd.append((3 * s[:, None] + three).ravel())
            rd.append((3 * g[:, None] + three).ravel())
            sp_.append(sp_[-1] + 3 * s.size)
            rp_.append(rp_[-1] + 3 * g.size)
        self.send_ptr, self.recv_ptr = np.asarray(sp_, dtype=np.int64), np.asarray(rp_, dtype=np.int64)
        cat = lambda a: np.ascontiguousarray(np.concatenate(a) if a else np.zeros(0), dtype=np.int32)
        self.send_dofs, self.recv_dofs = cat(sd), cat(rd)
        if self.recv_ptr[-1] != 3 * ghost.size:
            raise RuntimeError("ShellPartition: a ghost point has no owner among the neighbours")

    # vectors ------------------------------------------------------------------------------------------
    def local_state(self, w_global: np.ndarray) -> np.ndarray:
        return np.ascontiguousarray(np.asarray(w_global, dtype=np.float64)[self.dof_global])

    def local_vertex_field(self, h_global: np.ndarray) -> np.ndarray:
        h = np.asarray(h_global, dtype=np.float64)
        return np.ascontiguousarray(h[self.vert_global])

    def owned_vertices(self) -> np.ndarray:
        """Local ids of the vertices this rank owns (thickness gradients are complete there)."""
        return np.nonzero(self.part[self.vert_global] == self.rank)[0]

    def scatter_owned(self, w_local: np.ndarray, out_global: np.ndarray) -> np.ndarray:
        """out_global[owned dofs] = w_local[owned dofs]; the ranks' calls together fill the global state."""
        m = self.owned_dofs
        out_global[self.dof_global[m]] = np.asarray(w_local)[m]
        return out_global

    # preconditioner -----------------------------------------------------------------------------------
    def lattice(self, finest: Optional[int] = None, global_lattice: Optional[dict] = None) -> dict:
        """`lattice_pc` of the global space with the rows of the local dofs (what `femo_shell_pc_create` takes on this rank)."""
        import scipy.sparse as sp
        G = global_lattice if global_lattice is not None else lattice_pc(self.global_space, finest)
        L = dict(G)
        d = self.dof_global
        L["ell_idx"] = np.ascontiguousarray(G["ell_idx"][d])
        L["ell_w"] = np.ascontiguousarray(G["ell_w"][d])
        L["n_unode"] = int(self.space.n_unode)
        width, n = G["width"], d.size
        fin = slice(width - 8, width)
        P = sp.csr_matrix((L["ell_w"][:, fin].ravel(), (np.repeat(np.arange(n), 8), L["ell_idx"][:, fin].ravel())), shape=(n, G["n_lat"]))
        Pt = P.T.tocsr()
        Pt.sort_indices()
        L["pt_rowptr"], L["pt_cols"], L["pt_vals"] = Pt.indptr.astype(np.int64), Pt.indices.astype(np.int32), np.ascontiguousarray(Pt.data)
        return L
