"""Generates tests/golden/*.npz from closed-form known answers and from the NumPy
oracle (there is no reference run to record: FEniCSx is not installable here and
femo ships no fixtures -- SURVEY.md section 8(c); parity vs FEniCSx stays UNPINNED).

    python tests/make_golden.py

The element KATs are closed-form (independent of the oracle); the assembled
fixtures freeze the oracle's output so that later edits to it are detected.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import femo_oracle as fo  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def element_kats():
    """Closed-form P1 element matrices on the reference simplices (textbook values)."""
    k_tri = 0.5 * np.array([[2, -1, -1], [-1, 1, 0], [-1, 0, 1]], dtype=float)
    k_tet = (1.0 / 6.0) * np.array([[3, -1, -1, -1], [-1, 1, 0, 0], [-1, 0, 1, 0], [-1, 0, 0, 1]], dtype=float)
    m_tri = (0.5 / 12.0) * (np.ones((3, 3)) + np.eye(3))
    m_tet = ((1.0 / 6.0) / 20.0) * (np.ones((4, 4)) + np.eye(4))
    return dict(k_tri=k_tri, k_tet=k_tet, m_tri=m_tri, m_tet=m_tet,
                load_tri=np.full(3, 0.5 / 3.0), load_tet=np.full(4, (1.0 / 6.0) / 4.0))


def assembled(d, n, jitter):
    m = fo.unit_square_mesh(n, jitter) if d == 2 else fo.unit_cube_mesh(n, jitter)
    bd = fo.boundary_vertices_box(m.x)
    rng = np.random.default_rng(100 * d + n)
    f = 0.086 * (1.0 + 0.3 * rng.uniform(-1, 1, m.n_cell))
    u_d = fo.u_target(m.x)
    lin = fo.linearize(m, bd)
    ref = fo.reference_cycle(m, f, u_d, bd, np.zeros(len(bd)))
    u_rand = rng.standard_normal(m.n_vert)
    return dict(x=m.x, conn=m.conn, bc_dofs=bd, f=f, u_d=u_d,
                dRdu_indptr=lin.dRdu.indptr, dRdu_indices=lin.dRdu.indices, dRdu_data=lin.dRdu.data,
                A_data=lin.A.data, dRdf_indptr=lin.dRdf.indptr, dRdf_indices=lin.dRdf.indices,
                dRdf_data=lin.dRdf.data, u=ref["u"], J=ref["J"], grad=ref["grad"], lam=ref["lam"],
                u_rand=u_rand, residual_u_rand=fo.residual(m, u_rand, f))


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    np.savez(os.path.join(OUT, "element_kats.npz"), **element_kats())
    for d, n, jit in [(2, 4, 0.0), (2, 8, 0.2), (3, 4, 0.0), (3, 6, 0.2)]:
        np.savez_compressed(os.path.join(OUT, f"poisson_d{d}_n{n}_j{int(jit * 10)}.npz"), **assembled(d, n, jit))
    # nonlinear Poisson + symmetric Nitsche (examples/nonlinear_poisson_opt): f = 0.1, u from 1
    m = fo.unit_square_mesh(8, 0.2)
    bm = fo.boundary_facets(m)
    f = 0.1 * np.ones(m.n_cell)
    uex = fo.u_exact_nl(m.x)
    ref = fo.nl_reference_cycle(m, f, uex, bm)
    np.savez_compressed(os.path.join(OUT, "nl_poisson_d2_n8.npz"), x=m.x, conn=m.conn, bmask=bm, f=f, u_ex=uex,
                        u=ref["u"], J=ref["J"], grad=ref["grad"], lam=ref["lam"], newton_its=ref["newton_its"])
    print("wrote", sorted(os.listdir(OUT)))
