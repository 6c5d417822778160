"""Partition of the shell over N ranks (femo_amd/dist/shell.py), host side only: ownership, halo plans, the rank-local
lattice rows -- and the algebra the partitioned solver relies on, restated with the oracle's matrices: the ranks' shares
(local assembly with the rows of points owned elsewhere zeroed) sum to the global stiffness, load and Galerkin operator."""
import numpy as np
import pytest
import scipy.sparse as sp

from femo_amd.dist.shell import ShellPartition
from femo_amd.fea.shell import ShellSpace, lattice_pc
from oracle import shell_oracle as so


def _mesh(kind):
    if kind == "roof":
        return so.scordelis_lo_mesh(10, 7)
    pts, conn = so.plate_mesh(9)
    rng = np.random.default_rng(4)
    pts = pts + np.c_[0.02 * rng.standard_normal((len(pts), 2)), 0.05 * np.sin(3 * pts[:, 0])]       # warped, irregular
    return pts, conn


@pytest.mark.parametrize("kind", ["roof", "plate"])
@pytest.mark.parametrize("world", [2, 3, 5])
def test_ownership_and_halo_plans(kind, world):
    pts, conn = _mesh(kind)
    G = ShellSpace(pts, conn)
    parts = [ShellPartition(G, r, world) for r in range(world)]
    n_pts = G.n_dof // 3
    count = np.zeros(n_pts, dtype=int)
    for P in parts:
        own = P.owned_points.astype(bool)
        count[P.point_global[own]] += 1
        # local numbering keeps the state layout: dofs 3 p + k, displacement nodes (vertices, edges), rotations
        L = P.space
        assert P.point_global.size == L.n_dof // 3 and np.unique(P.point_global).size == P.point_global.size
        assert np.allclose(L.x, G.x[P.vert_global])
        assert np.allclose(L.unode_x, G.unode_x[P.point_global[:L.n_unode]])
        # every cell around an owned point is local
        cells_of = {}
        gcell_pts = np.concatenate([G.conn, G.n_vert + G.cell_edges, G.n_unode + G.conn], axis=1)
        local_cells = set(P.cell_global.tolist())
        owned_global = set(P.point_global[own].tolist())
        for c in range(G.n_cell):
            if owned_global.intersection(gcell_pts[c].tolist()):
                assert c in local_cells
    assert np.all(count == 1)                                         # every point has exactly one owner
    cell_count = np.zeros(G.n_cell, dtype=int)
    for P in parts:
        cell_count[P.cell_global[P.cell_owned]] += 1
    assert np.all(cell_count == 1)                                    # and every cell one integrating rank
    # what r sends to q is what q expects from r, dof by dof in global numbering
    for P in parts:
        for k, q in enumerate(P.nbr):
            Q = parts[int(q)]
            kk = int(np.nonzero(Q.nbr == P.rank)[0][0])
            sent = P.dof_global[P.send_dofs[P.send_ptr[k]:P.send_ptr[k + 1]]]
            expected = Q.dof_global[Q.recv_dofs[Q.recv_ptr[kk]:Q.recv_ptr[kk + 1]]]
            assert np.array_equal(sent, expected)
        ghosts = np.nonzero(~P.owned_dofs)[0]
        assert np.array_equal(np.sort(P.recv_dofs), ghosts)           # every ghost dof is refreshed exactly once


def test_shares_sum_to_the_global_operators():
    pts, conn = so.scordelis_lo_mesh(8, 6)
    G = ShellSpace(pts, conn)
    V = so.ShellSpace(pts, conn)
    rng = np.random.default_rng(0)
    h = 0.25 * (1.0 + 0.3 * rng.random(V.n_vert))
    f = rng.standard_normal((V.n_vert, 3))
    K = so.assemble(V, so.element_stiffness(V, h, 4.32e8, 0.3))
    F = so.load_vector(V, f)
    Lg = lattice_pc(G)
    n_lat = Lg["n_lat"]

    def prolongation(L, n):
        rows = np.repeat(np.arange(n), L["width"])
        return sp.csr_matrix((L["ell_w"].ravel(), (rows, L["ell_idx"].ravel())), shape=(n, n_lat))

    Pg = prolongation(Lg, G.n_dof)
    world = 3
    Ksum = sp.csr_matrix(K.shape)
    Fsum = np.zeros(V.n_dof)
    Asum = sp.csr_matrix((n_lat, n_lat))
    w = rng.standard_normal(V.n_dof)
    Kw = np.zeros(V.n_dof)
    for r in range(world):
        P = ShellPartition(G, r, world)
        Vl = so.ShellSpace(P.space.x, P.space.conn)
        assert np.array_equal(Vl.cell_dofs, P.space.cell_dofs)
        Kl = so.assemble(Vl, so.element_stiffness(Vl, h[P.vert_global], 4.32e8, 0.3)).tolil()
        Fl = so.load_vector(Vl, f[P.vert_global])
        ghost = np.nonzero(~P.owned_dofs)[0]
        Kl[ghost, :] = 0.0                                            # k_zero_unowned_rows
        Fl[ghost] = 0.0                                               # k_mask_unowned
        Kl = Kl.tocsr()
        S = sp.csr_matrix((np.ones(P.dof_global.size), (P.dof_global, np.arange(P.dof_global.size))), shape=(V.n_dof, P.dof_global.size))
        Ksum = Ksum + S @ Kl @ S.T
        Fsum += S @ Fl
        # the product of the solver: local rows times the locally held (halo-refreshed) vector
        Kw += S @ (Kl @ w[P.dof_global])
        Ll = P.lattice(global_lattice=Lg)
        Pl = prolongation(Ll, P.dof_global.size)
        assert abs(Pl - Pg[P.dof_global]).max() == 0.0
        Asum = Asum + Pl.T @ Kl @ Pl
    scale = abs(K).max()
    assert abs(Ksum - K).max() <= 1e-12 * scale
    assert np.abs(Fsum - F).max() <= 1e-12 * np.abs(F).max()
    assert np.abs(Kw - K @ w).max() <= 1e-12 * np.abs(K @ w).max()
    A = Pg.T @ K @ Pg
    assert abs(Asum - A).max() <= 1e-11 * abs(A).max()


def test_one_rank_partition_is_the_whole_mesh():
    pts, conn = so.scordelis_lo_mesh(4, 4)
    G = ShellSpace(pts, conn)
    P = ShellPartition(G, 0, 1)
    assert P.owned_points.all() and P.nbr.size == 0 and P.send_dofs.size == 0
    assert np.array_equal(P.dof_global, np.arange(G.n_dof)) and np.array_equal(P.space.conn, G.conn)


# ---- two real processes over gloo: the partitioned solver's algorithm in NumPy on the partition's plans ----------------

def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _shell_worker(rank, world, port, out_dir):
    """What femo_shell_solve does on a partitioned handle, restated with SciPy: zeroed foreign rows, masked right-hand side,
    halo refresh of the direction (isend / irecv on the plan's dof lists), all-reduced scalars, and the lattice
    preconditioner with the restricted residual and the Galerkin operators summed over the ranks."""
    import os
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    pts, conn = so.scordelis_lo_mesh(8, 6)
    G = ShellSpace(pts, conn)
    P = ShellPartition(G, rank, world)
    Vl = so.ShellSpace(P.space.x, P.space.conn)
    Vg = so.ShellSpace(pts, conn)
    rng = np.random.default_rng(0)
    h = 0.25 * (1.0 + 0.3 * rng.random(Vg.n_vert))
    f = np.tile([0.0, 0.0, -90.0], (Vg.n_vert, 1))
    nd = P.dof_global.size
    own = P.owned_dofs
    fixed_g = np.zeros(Vg.n_dof, dtype=bool)
    fixed_g[_roof_fixed(Vg)] = True
    fixed = fixed_g[P.dof_global]
    free = sp.diags((~fixed).astype(float))
    K = so.assemble(Vl, so.element_stiffness(Vl, h[P.vert_global], 4.32e8, 0.3)).tolil()
    K[np.nonzero(~own)[0], :] = 0.0
    K = (free @ K.tocsr() @ free).tocsr()                           # the rank's share, imposed rows / columns eliminated
    b = so.load_vector(Vl, f[P.vert_global]) * own * (~fixed)

    def allsum(a):
        t = torch.from_numpy(np.ascontiguousarray(np.atleast_1d(np.asarray(a, dtype=np.float64))).copy())
        dist.all_reduce(t)
        return t.numpy()

    def halo(v):
        reqs, recv = [], []
        for k, q in enumerate(P.nbr):
            sb = torch.from_numpy(np.ascontiguousarray(v[P.send_dofs[P.send_ptr[k]:P.send_ptr[k + 1]]]))
            rb = torch.zeros(int(P.recv_ptr[k + 1] - P.recv_ptr[k]), dtype=torch.float64)
            reqs += [dist.isend(sb, int(q)), dist.irecv(rb, int(q))]
            recv.append((k, rb, sb))
        for r_ in reqs:
            r_.wait()
        for k, rb, _ in recv:
            v[P.recv_dofs[P.recv_ptr[k]:P.recv_ptr[k + 1]]] = rb.numpy()

    # preconditioner: Jacobi on the own rows + the global lattice levels with all-reduced Galerkin diagonals
    L = P.lattice()
    n_lat = L["n_lat"]
    rows = np.repeat(np.arange(nd), L["width"])
    Pl = (free @ sp.csr_matrix((L["ell_w"].ravel(), (rows, L["ell_idx"].ravel())), shape=(nd, n_lat))).tocsr()
    gdiag = allsum((Pl.T @ K @ Pl).diagonal())
    cinv = np.where(gdiag > 0.0, 1.0 / np.where(gdiag > 0.0, gdiag, 1.0), 0.0)
    kd = K.diagonal()
    dinv = np.where(kd > 0.0, 1.0 / np.where(kd > 0.0, kd, 1.0), 0.0)

    def apply_pc(r):
        return dinv * r + Pl @ (cinv * allsum(Pl.T @ r))            # the second term is consistent on every local point

    x = np.zeros(nd)
    r = b.copy()
    z = apply_pc(r)
    p = z.copy()
    g = float(allsum(r @ z)[0])
    g0, its = g, 0
    while g > 1e-22 * g0 and its < 20000:
        halo(p)
        q = K @ p
        a = g / float(allsum(p @ q)[0])
        x += a * p
        r -= a * q
        z = apply_pc(r)
        g1 = float(allsum(r @ z)[0])
        p = z + (g1 / g) * p
        g = g1
        its += 1
    halo(x)
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), x=x, dofs=P.dof_global, own=own, its=its)
    dist.barrier()
    dist.destroy_process_group()


def _roof_fixed(V):
    ux, vx, L = V.unode_x, V.x, 25.0
    on = lambda arr, val: np.nonzero(np.isclose(arr, val, atol=1e-6))[0]
    return np.unique(np.concatenate([
        V.u_dof(on(ux[:, 0], L), 1), V.u_dof(on(ux[:, 0], L), 2),
        V.u_dof(on(ux[:, 1], 0.0), 1), V.theta_dof(on(vx[:, 1], 0.0), 0), V.theta_dof(on(vx[:, 1], 0.0), 2),
        V.u_dof(on(ux[:, 0], 0.0), 0), V.theta_dof(on(vx[:, 0], 0.0), 1), V.theta_dof(on(vx[:, 0], 0.0), 2)]))


def test_two_process_gloo_shell_solve(tmp_path):
    import torch.multiprocessing as mp
    world = 2
    mp.spawn(_shell_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    pts, conn = so.scordelis_lo_mesh(8, 6)
    V = so.ShellSpace(pts, conn)
    rng = np.random.default_rng(0)
    h = 0.25 * (1.0 + 0.3 * rng.random(V.n_vert))
    K = so.assemble(V, so.element_stiffness(V, h, 4.32e8, 0.3))
    wref = so.solve(K, so.load_vector(V, np.tile([0.0, 0.0, -90.0], (V.n_vert, 1))), _roof_fixed(V))
    w = np.full(V.n_dof, np.nan)
    its = set()
    for r in range(world):
        d = np.load(tmp_path / f"rank{r}.npz")
        w[d["dofs"][d["own"]]] = d["x"][d["own"]]
        its.add(int(d["its"]))
        # ghost copies were refreshed from their owners
    assert not np.isnan(w).any() and len(its) == 1 and 10 < its.pop() < 20000
    for r in range(world):
        d = np.load(tmp_path / f"rank{r}.npz")
        assert np.abs(d["x"] - w[d["dofs"]]).max() <= 1e-13 * np.abs(w).max()
    assert np.abs(w - wref).max() <= 1e-8 * np.abs(wref).max()
