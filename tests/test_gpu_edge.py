"""Edge cases of the C-ABI on the GPU: tiny and ragged meshes, borrowed device pointers, an
external stream, and the error contract (status + femo_last_error -> FemoError, no crash)."""
import numpy as np
import pytest

from oracle import femo_oracle as fo

pytestmark = pytest.mark.gpu


def test_single_cell_and_sub_wave_meshes(ctx):
    from femo_amd import engine as E
    for d in (2, 3):
        x = np.vstack([np.zeros(d), np.eye(d)]) * 0.7 + 0.1
        conn = np.arange(d + 1, dtype=np.int32)[None, :]
        m = fo.OMesh(d, x, conn)
        dm = E.DeviceMesh(ctx, x, conn)
        assert dm.info["n_slices"] == 1 and dm.info["nnz"] == (d + 1) ** 2
        J = E.Mat(dm)
        E.assemble_jacobian(dm, 0, None, None, None, None, J)
        assert np.abs(J.to_scipy().toarray() - fo.stiffness(m).toarray()).max() < 1e-14
        u, f = np.arange(1.0, d + 2), np.array([2.5])
        R = E.Vec(ctx, d + 1)
        E.assemble_residual(dm, 0, None, E.Vec(ctx, d + 1).set(u), E.Vec(ctx, 1).set(f), R)
        assert np.abs(R.get() - fo.residual(m, u, f)).max() < 1e-14
    # 65 rows: one full slice + one row in a second slice; unstructured renumbering
    m = fo.unit_square_mesh(6, 0.2)
    perm = np.random.default_rng(2).permutation(m.n_vert)
    x, conn = m.x[np.argsort(perm)], perm[m.conn].astype(np.int32)
    mm = fo.OMesh(2, x, conn)
    dm = E.DeviceMesh(ctx, x, conn)
    bd = fo.boundary_vertices_box(x)
    A = E.Mat(dm)
    E.assemble_jacobian(dm, 0, None, None, None, E.DirichletSet(dm, bd, 0.0), A)
    Ao = fo.eliminate_bc(fo.stiffness(mm), bd)
    assert np.array_equal(A.to_scipy().indices, Ao.indices) and np.abs(A.to_scipy().data - Ao.data).max() < 1e-13
    b = np.random.default_rng(3).standard_normal(mm.n_vert)
    X = E.Vec(ctx, mm.n_vert)
    info = A.solve_cg(E.Vec(ctx, mm.n_vert).set(b), X, rtol=1e-13)
    import scipy.sparse.linalg as spla
    assert info.converged == 1 and np.abs(X.get() - spla.splu(Ao.tocsc()).solve(b)).max() < 1e-10


def test_wrapped_pointer_and_vector_ops(ctx):
    from femo_amd import engine as E
    a = E.Vec(ctx, 1001).set(np.arange(1001.0))
    alias = E.Vec(ctx, 1001, device_ptr=a.device_ptr)          # femo_vec_wrap: borrows, never frees
    alias.axpy(2.0, a)                                         # a += 2a through the alias
    assert np.array_equal(a.get(), 3.0 * np.arange(1001.0))
    assert a.dot(alias) == pytest.approx(9.0 * np.sum(np.arange(1001.0) ** 2))
    del alias
    assert a.get()[1000] == 3000.0
    b = E.Vec(ctx, 1001).fill(2.0)
    E.pointwise_divide(b, a, b, 1001)
    assert np.array_equal(b.get(), 1.5 * np.arange(1001.0))
    assert E.Vec(ctx, 0).get().shape == (0,)                   # empty vectors are legal


def test_error_contract(ctx):
    from femo_amd import engine as E
    from femo_amd._lib import FemoError
    m = fo.unit_square_mesh(4)
    dm = E.DeviceMesh(ctx, m.x, m.conn)
    u, f, r = E.Vec(ctx, m.n_vert), E.Vec(ctx, m.n_cell), E.Vec(ctx, m.n_vert)
    with pytest.raises(FemoError, match="size mismatch"):
        E.assemble_residual(dm, 0, None, E.Vec(ctx, 3), f, r)
    with pytest.raises(FemoError, match="not implemented"):
        E.assemble_residual(dm, 7, None, u, f, r)
    with pytest.raises(FemoError, match="size mismatch"):
        u.set(np.zeros(5))
    with pytest.raises(FemoError, match="out of range"):
        E.DirichletSet(dm, [m.n_vert + 3], 0.0)
    with pytest.raises(FemoError, match="out of range"):
        E.DeviceMesh(ctx, m.x, m.conn + 1000)
    A = E.Mat(dm)
    with pytest.raises(FemoError, match="in place"):
        A.mult(u, u)
    other = E.DeviceMesh(ctx, m.x, m.conn)
    with pytest.raises(FemoError, match="another mesh"):
        E.assemble_jacobian(other, 0, None, None, None, None, A)
    with pytest.raises(FemoError, match="state u"):
        E.assemble_jacobian(dm, 1, [10.0], None, f, None, A)       # the nonlinear form needs u
    # non-finite data is reported as a breakdown (-1), iteration caps as 0; neither crashes or hangs
    bd = fo.boundary_vertices_box(m.x)
    E.assemble_jacobian(dm, 0, None, None, None, E.DirichletSet(dm, bd, 0.0), A)
    bad = np.ones(m.n_vert)
    bad[7] = np.nan
    info = A.solve_cg(E.Vec(ctx, m.n_vert).set(bad), E.Vec(ctx, m.n_vert), rtol=1e-14, max_it=200)
    assert info.converged == -1
    rhs = np.random.default_rng(0).standard_normal(m.n_vert)
    info = A.solve_cg(E.Vec(ctx, m.n_vert).set(rhs), E.Vec(ctx, m.n_vert), rtol=1e-30, max_it=3)
    assert info.converged == 0 and info.iterations == 3


def test_external_stream(ctx):
    """A context can run on a caller-provided HIP stream (here: another context's stream)."""
    from femo_amd import engine as E
    c2 = E.Context(0, stream=ctx.stream)
    assert c2.stream == ctx.stream
    v = E.Vec(c2, 10).fill(4.0)
    assert np.all(v.get() == 4.0)
    del v
    c2.close()


def test_vector_set_is_scalar_only(ctx):
    """Function.vector.set mirrors PETSc's Vec.set(alpha): a scalar (or the length-1 array `update`
    passes, utils_dolfinx.py:308-309) broadcasts; a longer array fails loudly instead of using a[0]."""
    from femo_amd.fea import utils_hip
    from femo_amd.fea.fea_hip import Function, FunctionSpace, getFuncArray, setFuncArray, update
    from femo_amd.fea.mesh import createUnitSquareMesh
    utils_hip.set_context(ctx)
    V = FunctionSpace(createUnitSquareMesh(3), ('CG', 1))
    fn = Function(V)
    fn.vector.set(2.5)
    assert np.all(getFuncArray(fn) == 2.5)
    update(fn, np.array([0.75]))
    assert np.all(getFuncArray(fn) == 0.75)
    with pytest.raises(ValueError, match="scalar"):
        fn.vector.set(np.arange(16.0))
    setFuncArray(fn, np.arange(16.0))
    assert np.array_equal(getFuncArray(fn), np.arange(16.0))
