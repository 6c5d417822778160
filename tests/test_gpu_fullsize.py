"""Parity at BASELINE.json's full sizes (C2: n=100, 1.03 M DOFs; C4: n=215, 10.08 M DOFs) through
size-independent properties, since the LU oracle cannot run there:
  * the forward solve against the DST-exact solution of the *discrete* problem (oracle.dst_solve),
  * linearity of the solution operator and the adjoint identity <A^-1 b, c> = <b, A^-T c>,
  * the adjoint total against a directional finite difference of the functional,
  * fused vs separate assembly and SpMV vs residual consistency (R(u; 0) = K u)."""
import numpy as np
import pytest

from oracle import c_port
from oracle import femo_oracle as fo

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n", [100, 215])
def test_full_size_properties(ctx, n):
    from femo_amd import engine as E
    from femo_amd.fea.mesh import createUnitCubeMesh
    mesh = createUnitCubeMesh(n)
    N, NC = mesh.n_vert, mesh.n_cell
    dm = E.DeviceMesh(ctx, mesh.x, mesh.conn)
    assert dm.info["nnz"] == N + 2 * (3 * n * (n + 1) ** 2 + 3 * n * n * (n + 1) + n ** 3)
    bd = fo.boundary_vertices_box(mesh.x)
    bc = E.DirichletSet(dm, bd, 0.0)
    rng = np.random.default_rng(n)
    xc = mesh.centroids()
    f = np.prod(np.sin(np.pi * xc), axis=1) * (1.0 + 0.5 * np.cos(3 * np.pi * xc[:, 0]) * xc[:, 2]) + 0.1
    F, U0 = E.Vec(ctx, NC).set(f), E.Vec(ctx, N)
    A, K, B = E.Mat(dm), E.Mat(dm), E.Vec(ctx, N)
    E.assemble_system(dm, 0, None, U0, F, bc, K, A, B)                 # u = 0: B = -load on interior rows, 0 on bc
    # (1) forward solve vs the DST-exact discrete solution
    load = -c_port.residual(3, mesh.x, mesh.conn, np.zeros(N), f)      # independent C/OpenMP load vector
    load[bd] = 0.0
    assert np.abs(B.get() + load).max() < 1e-13 * np.abs(load).max()
    om = fo.OMesh(3, mesh.x, mesh.conn, n)
    u_exact = fo.dst_solve(om, load)
    X = E.Vec(ctx, N)
    Bpos = E.Vec(ctx, N).fill(0.0).axpy(-1.0, B)
    info = A.solve_cg(Bpos, X, rtol=1e-14)
    u = X.get()
    assert info.converged == 1
    assert np.abs(u - u_exact).max() < 1e-10 * np.abs(u_exact).max()
    # (2) adjoint identity and linearity of the solve
    c = rng.standard_normal(N)
    c[bd] = 0.0
    C, Y = E.Vec(ctx, N).set(c), E.Vec(ctx, N)
    A.solve_cg(C, Y, transpose=True, rtol=1e-14)
    lhs, rhs = float(u @ c), float(load @ Y.get())
    assert abs(lhs - rhs) < 1e-10 * max(abs(lhs), abs(rhs))
    S, Z = E.Vec(ctx, N).set(load + 0.5 * c), E.Vec(ctx, N)
    A.solve_cg(S, Z, rtol=1e-14)
    expect = u + 0.5 * Y.get()
    assert np.abs(Z.get() - expect).max() < 1e-10 * np.abs(expect).max()
    # (3) SpMV vs residual: R(u; 0) = K u, and A = K off the Dirichlet set
    Zero, R, KU = E.Vec(ctx, NC), E.Vec(ctx, N), E.Vec(ctx, N)
    E.assemble_residual(dm, 0, None, X, Zero, R)
    K.mult(X, KU)
    assert np.abs(R.get() - KU.get()).max() < 1e-12 * np.abs(KU.get()).max()
    interior = np.ones(N, bool)
    interior[bd] = False
    AU = E.Vec(ctx, N)
    A.mult(X, AU)                                                       # u vanishes on the boundary
    assert np.abs(AU.get()[interior] - KU.get()[interior]).max() < 1e-13 * np.abs(KU.get()).max()
    # (4) adjoint total vs directional finite difference of J (exact reduced gradient: lam = 0 on bc)
    alpha = 1e-3
    ud = fo.u_target(mesh.x)
    UD, G, GF, LAM = E.Vec(ctx, N).set(ud), E.Vec(ctx, N), E.Vec(ctx, NC), E.Vec(ctx, N)
    E.functional_grad_u(dm, 0, [alpha], X, F, UD, G)
    g = np.array(G.get())                      # results are read-only (they mirror the device vector): copy to edit
    g[bd] = 0.0
    A.solve_cg(E.Vec(ctx, N).set(g), LAM, transpose=True, rtol=1e-14)
    DV, DT = E.Vec(ctx, NC * 4), E.Vec(ctx, NC)
    E.assemble_dRdf(dm, 0, None, X, F, DV)
    E.dRdf_apply(dm, DV, LAM, DT, transpose=True)
    E.functional_grad_f(dm, 0, [alpha], X, F, UD, GF)
    grad = GF.get() - DT.get()
    dirn = rng.standard_normal(NC)
    eps = 1e-3

    def J_of(ff):
        Fv = E.Vec(ctx, NC).set(ff)
        Bv, Xv = E.Vec(ctx, N), E.Vec(ctx, N)
        E.assemble_system(dm, 0, None, U0, Fv, bc, None, None, Bv)
        A.solve_cg(E.Vec(ctx, N).fill(0.0).axpy(-1.0, Bv), Xv, rtol=1e-14)
        return E.functional_value(dm, 0, [alpha], Xv, Fv, UD)

    fd = (J_of(f + eps * dirn) - J_of(f - eps * dirn)) / (2 * eps)        # J is quadratic in f: exact up to round-off
    an = float(grad @ dirn)
    assert abs(fd - an) < 1e-7 * abs(an)
