"""The host boundary (csrc/hostmem.cpp): pinned blocks, the staged pageable path, accumulate-on-copy-out and
the provenance rule that elides repeated uploads -- every elision checked against the bytes (FEMO_HOST_VERIFY)."""
import gc
import os

import numpy as np
import pytest

from oracle import femo_oracle as fo

pytestmark = pytest.mark.gpu

SIZES = [0, 1, 7, 1000, (1 << 20) + 3, 5 * (1 << 20) + 11]      # below / across / beyond the 4 x 8 MiB staging ring


@pytest.fixture(autouse=True)
def _verify_elisions():
    os.environ["FEMO_HOST_VERIFY"] = "1"
    yield
    os.environ.pop("FEMO_HOST_VERIFY", None)


@pytest.mark.parametrize("n", SIZES)
def test_round_trips_pageable_and_pinned(ctx, n):
    from femo_amd import engine as E
    rng = np.random.default_rng(n)
    a = rng.standard_normal(n)
    v = E.Vec(ctx, n)
    v.set(a)                                            # pageable source: staged
    back = v.get()                                      # pinned, read-only result
    assert not back.flags.writeable and (n == 0 or E.is_pinned(back))
    assert np.array_equal(back, a)
    out = np.full(n, np.nan)
    v.get(out=out)                                      # pageable destination: staged
    assert np.array_equal(out, a)
    p = E.pinned_array(a)                               # pinned source: one DMA
    w = E.Vec(ctx, n)
    w.set(p)
    assert np.array_equal(w.get(), a)
    acc = rng.standard_normal(n)
    expect = acc + a
    got = acc.copy()
    v.add_to_host(got)                                  # pageable accumulate
    assert np.array_equal(got, expect)
    got_p = E.pinned_empty(n)
    got_p[:] = acc
    E.host_touch(got_p)
    v.add_to_host(got_p)                                # pinned accumulate
    assert np.array_equal(got_p, expect)


def test_uploads_are_elided_only_while_exact(ctx):
    from femo_amd import engine as E
    n = 300_001
    a = np.random.default_rng(1).standard_normal(n)
    v, w = E.Vec(ctx, n), E.Vec(ctx, n)
    v.set(a)
    E.host_stats(reset=True)
    h = v.get()                                         # h mirrors v
    v.set(h)                                            # same vector, unchanged: skipped
    st = E.host_stats()
    assert st["h2d_skipped"] == 1 and st["h2d_pinned"] == 0
    w.set(h)                                            # another vector: device-to-device copy
    st = E.host_stats()
    assert st["h2d_as_d2d"] == 1 and st["h2d_pinned"] == 0
    assert np.array_equal(w.get(), a)
    v.fill(2.0)                                         # v changed: h no longer mirrors it, but it mirrors w now
    v.set(h)
    st = E.host_stats()
    assert st["h2d_as_d2d"] == 2 and st["h2d_pinned"] == 0
    assert np.array_equal(v.get(), a)
    w.fill(3.0); v.fill(2.0)                            # no device vector holds the content any more: real upload
    v.set(h)
    assert E.host_stats()["h2d_pinned"] == 1
    assert np.array_equal(v.get(), a)
    # after the upload the block mirrors v again
    v.set(h)
    assert E.host_stats()["h2d_skipped"] == 2
    # a host-side writer announces itself: the elision stops
    hw = E.writable(h)
    hw[0] = 123.0
    v.set(hw)
    assert E.host_stats()["h2d_pinned"] == 2
    assert v.get()[0] == 123.0
    w.set(a)
    # a destroyed source vector cannot serve as the source of a device-to-device copy
    h2 = w.get()
    del w
    gc.collect()
    v.set(h2)
    assert np.array_equal(v.get(), a)


def test_every_writing_entry_point_bumps_the_generation(ctx):
    """If an entry point wrote a vector without announcing it, a later upload of an older host copy would be
    skipped and the stale device content would survive: upload after each kind of device-side write."""
    from femo_amd import engine as E
    from femo_amd import _lib
    from femo_amd.fea.mesh import createUnitSquareMesh
    mesh = createUnitSquareMesh(6, jitter=0.1)
    dm = mesh.device(ctx)
    n, nc = mesh.n_vert, mesh.n_cell
    rng = np.random.default_rng(5)
    u, f, r, g = E.Vec(ctx, n), E.Vec(ctx, nc), E.Vec(ctx, n), E.Vec(ctx, nc)
    u.set(rng.standard_normal(n)); f.set(rng.standard_normal(nc))
    bd = fo.boundary_vertices_box(mesh.x)
    ds = E.DirichletSet(dm, bd, np.zeros(len(bd)))
    A, K = E.Mat(dm), E.Mat(dm)
    vals = E.Vec(ctx, nc * 3)
    ones = E.Vec(ctx, n).fill(1.0)

    def writes(vec, fn):
        marker = np.full(vec.n, 7.25)
        vec.set(marker)
        h = vec.get()                                   # mirrors vec
        fn()                                            # device-side write
        after = np.array(vec.get())
        vec.set(h)                                      # must really upload (FEMO_HOST_VERIFY would also catch a wrong skip)
        assert np.array_equal(vec.get(), marker), fn
        return after

    writes(r, lambda: E.assemble_residual(dm, _lib.PDE_POISSON, None, u, f, r))
    writes(r, lambda: E.assemble_system(dm, _lib.PDE_POISSON, None, u, f, ds, None, A, r))
    writes(r, lambda: r.fill(1.0))
    writes(r, lambda: r.axpy(2.0, u))
    writes(r, lambda: r.copy_from(u))
    writes(r, lambda: E.bc_apply_rhs(ds, u, r))
    E.assemble_system(dm, _lib.PDE_POISSON, None, u, f, ds, K, A, None)
    writes(r, lambda: E.newton_rhs(K, ones, u, ds, r))
    writes(r, lambda: A.mult(u, r))
    writes(r, lambda: A.diagonal(r))
    writes(r, lambda: A.solve_cg(ones, r, rtol=1e-10))
    writes(r, lambda: A.solve_cg(ones, r, rtol=1e-10, pc="bpx"))
    writes(r, lambda: A.solve_bicgstab(ones, r, rtol=1e-10))
    writes(r, lambda: A.pc_apply(ones, r))
    writes(vals, lambda: E.assemble_dRdf(dm, _lib.PDE_POISSON, None, None, None, vals))
    writes(g, lambda: E.dRdf_apply(dm, vals, u, g, transpose=True))
    writes(r, lambda: E.dRdf_apply(dm, vals, f, r, transpose=False))
    writes(r, lambda: E.functional_grad_u(dm, 0, [1e-6], u, f, ones, r))
    writes(g, lambda: E.functional_grad_f(dm, 0, [1e-6], u, f, ones, g))
    writes(g, lambda: E.cell_expression(dm, 0, None, u, g))
    writes(r, lambda: E.pointwise_divide(r, u, ones, n))
    writes(r, lambda: A.bench_spmv(u, r, 1))


def test_blocks_are_recycled(ctx):
    from femo_amd import engine as E
    a = E.pinned_empty(12345)
    p = a.ctypes.data
    del a
    gc.collect()
    b = E.pinned_empty(12345)
    assert b.ctypes.data == p                            # same pinned block, no new hipHostMalloc
    c = E.pinned_empty(12345)
    assert c.ctypes.data != p


def test_host_helpers(ctx):
    from femo_amd import engine as E
    n = 700_003
    rng = np.random.default_rng(3)
    x, y0 = rng.standard_normal(n), rng.standard_normal(n)
    y = y0.copy()
    E.host_axpby(-1.0, x, 0.0, y)
    assert np.array_equal(y, -x)
    y = y0.copy()
    E.host_axpby(2.0, x, 1.0, y)
    assert np.array_equal(y, 2.0 * x + y0)
    z = np.full(n, np.nan)
    E.host_axpby(0.0, z, 0.0, z)
    assert not z.any()
    E.host_copy(z, x)
    assert np.array_equal(z, x)


def test_operator_cycle_moves_each_array_once(ctx):
    """One cycle through FEAModel / StateOperation / OutputOperation with NumPy arrays at the boundary:
    unchanged inputs and the state the solve just returned are not sent again (every elision verified), the
    result equals the oracle's cycle and equals the cycle of a driver that owns pageable arrays."""
    from femo_amd import engine as E
    from femo_amd.fea.mesh import createUnitCubeMesh
    from tests.test_gpu_operators import make_sim
    n = 10
    mesh = createUnitCubeMesh(n, jitter=0.2)
    om = fo.unit_cube_mesh(n, jitter=0.2)
    f = fo.f_star(fo.centroids(om)) * 0.7 + 0.1
    bd = fo.boundary_vertices_box(om.x)
    ref = fo.reference_cycle(om, f, fo.u_target(om.x), bd, np.zeros(len(bd)))
    sim, fea, _, _ = make_sim(mesh, device=False)
    sim['f'] = f
    sim.run()
    E.host_stats(reset=True)
    sim['f'] = E.pinned_array(f)                        # new source in pinned memory (adopted by reference)
    sim.run()
    g = sim.compute_totals('l2_functional', 'f')
    st = E.host_stats()
    nf, nu = mesh.n_cell * 8, mesh.n_vert * 8
    assert isinstance(g, np.ndarray) and g.flags.writeable
    # PCIe: f once; the initial guess is the state the previous run returned (still on the device), f / u / psi come
    # back to the later operator methods as the arrays already sent, and the adjoint seed -dJ/du is known to be
    # -1 x a device vector (femo_host_axpby from a block that mirrors it): formed on the device
    assert st["h2d_pinned_bytes"] + st["h2d_staged_bytes"] == nf, st
    assert st["h2d_skipped"] + st["h2d_as_d2d"] == 9, st
    # down: u, dJ/df, dJ/du, psi, dR/df^T psi
    down = st["d2h_pinned_bytes"] + st["d2h_staged_bytes"] + st["d2h_async_bytes"] + st["d2h_device_sum_bytes"]
    assert down == 2 * nf + 3 * nu, st
    # u, dJ/du, dJ/df and psi leave asynchronously; dR/df^T psi is added to dJ/df on the device and leaves as the sum
    assert st["d2h_async"] == 4 and st["d2h_device_sum"] == 1 and st["d2h_staged"] == 0, st

    def rel(a, b):
        return np.abs(np.asarray(a) - b).max() / np.abs(b).max()

    assert rel(sim['u'], ref['u']) < 1e-10 and rel(g, ref['grad']) < 1e-10
    assert abs(sim['l2_functional'][0] - ref['J'][0]) < 1e-10 * abs(ref['J'][0])
    # a driver with pageable storage gets the same numbers through the staged path
    sim_p, _, _, _ = make_sim(mesh, device=False, pinned=False)
    sim_p['f'] = f
    sim_p.run()
    gp = sim_p.compute_totals('l2_functional', 'f')
    assert rel(sim_p['u'], ref['u']) < 1e-10 and rel(gp, ref['grad']) < 1e-10


def test_asynchronous_results(ctx):
    """Vec.get under lazy_results returns before the bytes have landed: a later device-side write of the vector
    must not leak into the copy, library calls on the array wait by themselves, a block is not recycled under a
    running DMA."""
    from femo_amd import engine as E
    n = 6_000_011
    a = np.random.default_rng(11).standard_normal(n)
    v, w = E.Vec(ctx, n), E.Vec(ctx, n)
    v.set(a)
    E.host_stats(reset=True)
    for trial in range(3):
        with E.lazy_results():
            h = v.get()
        v.fill(-1.0)                                    # enqueued behind the copy (femo_vec_touch waits on the device)
        v.axpy(2.0, v)
        assert np.array_equal(E.host_wait(h), a), trial
        v.set(h)                                        # h mirrors the OLD generation: a real upload
        assert np.array_equal(v.get(), a)
    st = E.host_stats()
    assert st["d2h_async"] == 3 and st["h2d_pinned"] == 3, st
    with E.lazy_results():
        h = v.get()
    v.set(h)                                            # skipped, without waiting for the bytes
    w.set(h)                                            # elided (device to device); the block mirrors w from now on
    st = E.host_stats()
    assert st["h2d_as_d2d"] == 1 and st["h2d_skipped"] == 1, st
    assert np.array_equal(w.get(), a)
    out = np.empty(n)
    E.host_copy(out, h)                                 # library call on a block in flight: waits
    assert np.array_equal(out, a)
    with E.lazy_results():
        h2 = v.get()
        p = h2.ctypes.data
        del h2                                          # freed while the DMA may still run: the pool waits
        gc.collect()
        h3 = w.get()
    assert np.array_equal(E.host_wait(h3), a)
    E.host_sync()
    with E.lazy_results(False):
        assert np.array_equal(v.get(), a)               # not lazy: complete on return


def test_scaled_copies_are_formed_on_the_device(ctx):
    """y = a x on the host (femo_host_axpby) with x a block that mirrors a device vector: y is recorded as a x that
    vector -- also while x is still on its way down, the host pass is then queued behind the copy -- and sending y
    to a vector is a device-side scale.  Every elision checked byte for byte (FEMO_HOST_VERIFY)."""
    from femo_amd import engine as E
    n = 4_000_003
    a = np.random.default_rng(21).standard_normal(n)
    v, w = E.Vec(ctx, n).set(a), E.Vec(ctx, n)
    for lazy in (False, True):
        E.host_stats(reset=True)
        with E.lazy_results(lazy):
            h = v.get()
        y = E.pinned_empty(n)
        E.host_axpby(-2.5, h, 0.0, y)                   # returns at once in the lazy case
        w.fill(0.0)
        w.set(y)
        st = E.host_stats()
        assert st["h2d_as_d2d"] == 1 and st["h2d_pinned"] == 0, (lazy, st)
        assert np.array_equal(w.get(), -2.5 * a)
        assert np.array_equal(E.host_wait(y), -2.5 * a)
        # chained: z = 2 y is -5 x the vector
        z = E.pinned_empty(n)
        E.host_axpby(2.0, y, 0.0, z)
        w.set(z)
        assert np.array_equal(w.get(), -5.0 * a) and np.array_equal(E.host_wait(z), -5.0 * a)
        # the vector changes: y no longer describes it, a real upload follows
        v.axpy(1.0, v)
        E.host_axpby(3.0, h, 0.0, y)
        w.set(y)
        assert E.host_stats()["h2d_pinned"] >= 1
        assert np.array_equal(w.get(), 3.0 * a)
        v.set(a)
        del h, y, z
        gc.collect()


def test_constant_blocks_are_filled_on_the_device(ctx):
    from femo_amd import engine as E
    n = 2_000_003
    v = E.Vec(ctx, n).fill(7.0)
    z = E.pinned_full(n, 0.25)
    assert not z.flags.writeable and np.all(z == 0.25)
    E.host_stats(reset=True)
    v.set(z)
    st = E.host_stats()
    assert st["h2d_pinned"] == 0 and st["h2d_as_d2d"] == 1, st
    assert np.all(v.get() == 0.25)
    v.fill(1.0)
    v.set(z)                                            # still the constant, whatever the vector holds
    assert np.all(v.get() == 0.25) and E.host_stats()["h2d_pinned"] == 0
    zw = E.writable(z)                                  # announced write: an ordinary block from now on
    zw[5] = 3.0
    v.set(zw)
    assert E.host_stats()["h2d_pinned"] == 1
    got = v.get()
    assert got[5] == 3.0 and got[4] == 0.25


def test_accumulate_on_the_device(ctx):
    """add_to_host into a block that still mirrors a live vector: the sum is formed on the device and is the
    host sum bit for bit; the block mirrors nothing afterwards; any doubt falls back to the host path."""
    from femo_amd import engine as E
    n = 3_000_017
    rng = np.random.default_rng(12)
    a, b = rng.standard_normal(n), rng.standard_normal(n)
    g, c = E.Vec(ctx, n).set(a), E.Vec(ctx, n).set(b)
    E.host_stats(reset=True)
    with E.lazy_results():
        h = g.get()
    hw = E.writable(h, announce=False)
    c.add_to_host(hw)
    st = E.host_stats()
    assert st["d2h_device_sum"] == 1 and st["d2h_staged"] == 0, st
    assert np.array_equal(hw, a + b)
    g.set(hw)                                           # must be a real upload now
    assert E.host_stats()["h2d_pinned"] == 1
    assert np.array_equal(g.get(), a + b)
    # announced write: host path
    h = g.get()
    hw = E.writable(h)
    c.add_to_host(hw)
    assert E.host_stats()["d2h_staged"] == 1
    assert np.array_equal(hw, (a + b) + b)
    # the mirrored vector changed in between: host path
    g.set(a)
    h = g.get()
    hw = E.writable(h, announce=False)
    g.fill(0.0)
    c.add_to_host(hw)
    assert E.host_stats()["d2h_staged"] == 2 and E.host_stats()["d2h_device_sum"] == 1
    assert np.array_equal(hw, a + b)


def test_accumulate_into_a_scaled_block(ctx):
    """A block filled by host_axpby(a != 1, x, 0, y) is recorded as a * vector; adding into it must give a * vector + v
    (the device-sum path used the unscaled mirror)."""
    from femo_amd import engine as E
    n = 2_000_011
    rng = np.random.default_rng(33)
    a, b = rng.standard_normal(n), rng.standard_normal(n)
    g, c = E.Vec(ctx, n).set(a), E.Vec(ctx, n).set(b)
    os.environ.pop("FEMO_HOST_VERIFY", None)            # the defect was silent only without the verifier
    try:
        for lazy in (False, True):
            with E.lazy_results(lazy):
                h = g.get()
            y = E.pinned_empty(n)
            E.host_axpby(-2.5, h, 0.0, y)
            E.host_stats(reset=True)
            c.add_to_host(y)
            assert E.host_stats()["d2h_device_sum"] == 0
            assert np.array_equal(y, -2.5 * a + b)
            del h, y
    finally:
        os.environ["FEMO_HOST_VERIFY"] = "1"


def test_caller_arrays_are_pinned_on_second_sight(ctx):
    """A pageable NumPy array handed over repeatedly (a backend's own variable storage, utils_dolfinx.py:155-167,
    300-311): first transfer staged, pinned in place at the second, DMA from then on; never trusted as a mirror (its
    owner may write it); unpinned when NumPy frees it; temporaries are never pinned."""
    from femo_amd import engine as E
    n = 1_500_003
    rng = np.random.default_rng(5)
    a = rng.standard_normal(n)                      # pageable, owns its memory
    v = E.Vec(ctx, n)
    E.host_stats(reset=True)
    v.set(a)
    st = E.host_stats()
    assert st["h2d_staged"] == 1 and st["h2d_pinned"] == 0 and not E.is_pinned(a)
    v.set(a)                                        # second sight: registered, this transfer already is a DMA
    st = E.host_stats()
    assert E.is_pinned(a) and st["h2d_pinned"] == 1 and st["h2d_staged"] == 1
    a[7] = 42.0                                     # the owner writes without telling anyone ...
    v.set(a)                                        # ... and the bytes still go over: no elision for caller memory
    st = E.host_stats()
    assert st["h2d_pinned"] == 2 and st["h2d_skipped"] == 0 and st["h2d_as_d2d"] == 0
    assert v.get()[7] == 42.0 and np.array_equal(v.get(), a)
    # results into the caller's array and accumulation take the pinned path as well
    out = np.empty(n)
    v.get(out=out); v.get(out=out)
    assert E.is_pinned(out) and np.array_equal(out, a)
    acc = np.ones(n)
    v.add_to_host(acc); v.add_to_host(acc)
    assert E.is_pinned(acc) and np.array_equal(acc, (1.0 + a) + a)
    # views of the same owner count as the owner; small arrays and one-off temporaries are left alone
    E.host_stats(reset=True)
    v2 = E.Vec(ctx, n - 3)
    v2.set(a[3:]); v2.set(a[:-3])
    assert E.host_stats()["h2d_pinned"] == 2
    small = np.zeros(1000)
    s = E.Vec(ctx, 1000)
    s.set(small); s.set(small)
    assert not E.is_pinned(small)
    for _ in range(3):
        v.set(rng.standard_normal(n))
    # freeing the array unpins it (the address may be handed out again by the allocator)
    addr, nb = a.ctypes.data, a.nbytes
    del a, out, acc
    gc.collect()
    lib = E._lib.load()
    assert not lib.femo_host_is_pinned(addr, nb)
    E.auto_register(False)
    try:
        b = rng.standard_normal(n)
        v.set(b); v.set(b)
        assert not E.is_pinned(b)
    finally:
        E.auto_register(True)


def test_two_outputs_do_not_alias_in_device_mode(ctx):
    """compute_totals(of=[a, b]) on the device returns distinct buffers, and a result kept from an earlier
    call survives the next one (pooled work arrays were handed out)."""
    from femo_amd.fea.mesh import createUnitSquareMesh
    from tests.test_gpu_operators import make_sim
    mesh = createUnitSquareMesh(12)
    res = {}
    for device in (False, True):
        sim, fea, f_ex, _ = make_sim(mesh, device=device)
        sim['f'] = np.asarray(f_ex.vector.getArray())
        sim.run()
        g1 = sim.compute_totals('l2_functional', 'f')
        keep = np.array(g1, copy=True)
        sim['f'] = 0.5 * np.asarray(f_ex.vector.getArray())
        sim.run()
        g2 = sim.compute_totals('l2_functional', 'f')
        assert g2 is not g1
        assert np.array_equal(np.asarray(g1), keep)      # the earlier result was not overwritten
        res[device] = (keep, np.array(g2, copy=True))
    assert np.allclose(res[False][0], res[True][0], rtol=1e-9, atol=1e-18)
    assert np.allclose(res[False][1], res[True][1], rtol=1e-9, atol=1e-18)


def test_pageable_driver_results_are_not_overwritten(ctx):
    """Simulator(pinned=False) keeps persistent pageable adjoint storage; a total handed out by one compute_totals
    must survive the next one while its caller holds it (the storage array itself was returned and
    rewritten), and the storage must come back into rotation once released (no fresh 0.5 GB array per cycle)."""
    from femo_amd.fea.mesh import createUnitSquareMesh
    from tests.test_gpu_operators import make_sim
    mesh = createUnitSquareMesh(12)
    sim, fea, f_ex, _ = make_sim(mesh, device=False, pinned=False)
    f0 = np.asarray(f_ex.vector.getArray())
    held, copies = [], []
    for k in range(3):
        sim['f'] = (1.0 - 0.25 * k) * f0
        sim.run()
        g = sim.compute_totals('l2_functional', 'f')
        held.append(g)
        copies.append(np.array(g, copy=True))
        for a, c in zip(held, copies):
            assert np.array_equal(np.asarray(a), c)
    assert len({id(a) for a in held}) == 3
    assert not np.array_equal(copies[0], copies[1])
    first = id(held[0])
    del held[:], g, a
    sim['f'] = 0.3 * f0
    sim.run()
    g = sim.compute_totals('l2_functional', 'f')
    rings = [r for r in sim._adj_storage.values() if any(g is a for a in r)]
    assert len(rings) == 1 and len(rings[0]) <= 4 and first in {id(a) for a in rings[0]}


def test_deferred_upload_of_f_matches_the_synchronous_cycle(ctx):
    """The inputs of solve_residual_equations travel as deferred uploads (femo_vec_set_host_deferred): the first Newton
    pass assembles A and K u' against a zero load vector, scales A, and only then waits for f.  Same state, functional
    and gradient as the synchronous path (rounding of one subtraction), the upload is counted as deferred, and a host
    write announced right after the cycle (femo_host_touch waits for the copy) cannot race it."""
    from bench import build_problem, one_cycle, source_fields
    from femo_amd import engine as E
    from femo_amd.fea import utils_hip
    from femo_amd.fea.mesh import createUnitCubeMesh
    utils_hip.set_context(ctx)
    mesh = createUnitCubeMesh(56)                       # 1.05 M cells: f is 8.4 MB, above the deferral threshold
    fs = source_fields(mesh, 2)
    out = {}
    os.environ.pop("FEMO_HOST_VERIFY", None)            # the verifier makes every upload synchronous (it compares bytes)
    for mode in ("deferred", "sync"):
        sim, fea = build_problem(mesh, device=False)
        if mode == "sync":
            fea.async_results = False
            sim.async_results = False
        one_cycle(sim, fea, E.pinned_array(fs[0]), E.pinned_full(mesh.n_vert, 0.0))      # builds the load vector storage
        E.host_stats(reset=True)
        fpin = E.pinned_array(fs[1])
        g = one_cycle(sim, fea, fpin, E.pinned_full(mesh.n_vert, 0.0))
        st = E.host_stats()
        out[mode] = (np.array(E.host_wait(sim['u']), copy=True), np.array(E.host_wait(g), copy=True),
                     float(np.asarray(sim['l2_functional']).ravel()[0]), st["h2d_deferred"])
        w = E.writable(fpin)                             # announces the write: waits for anything in flight on the block
        w[:] = 0.0
        utils_hip.clear_workspaces()
    os.environ["FEMO_HOST_VERIFY"] = "1"
    assert out["deferred"][3] >= 1 and out["sync"][3] == 0
    rel = lambda a, b: np.abs(a - b).max() / np.abs(b).max()
    assert rel(out["deferred"][0], out["sync"][0]) < 1e-11
    assert rel(out["deferred"][1], out["sync"][1]) < 1e-11
    assert abs(out["deferred"][2] - out["sync"][2]) < 1e-12 * abs(out["sync"][2])


def test_early_linearisation_is_the_same_cycle(ctx):
    """Round 5: for a form with constant partials (linear Poisson) StateOperation assembles dR/du, A, dR/df and S A S of the
    adjoint system while f is still on its way to the device (solve_residual_equations), and compute_derivatives of that
    cycle finds it done.  The same kernels on the same data, earlier: state, functional and gradient agree with the cycle that
    linearises in compute_derivatives to the solver tolerance (not bitwise: the BPX brick restriction accumulates with fp64
    atomics, so two runs of the SAME cycle already differ in the last bits); the assembled matrix is bit-identical, and a
    compute_derivatives without a solve before it still assembles."""
    from bench import build_problem, one_cycle, source_fields
    from femo_amd import engine as E
    from femo_amd.fea import utils_hip
    from femo_amd.fea.mesh import createUnitCubeMesh
    utils_hip.set_context(ctx)
    mesh = createUnitCubeMesh(56, jitter=0.2)
    fs = source_fields(mesh, 2)
    os.environ.pop("FEMO_HOST_VERIFY", None)
    out = {}
    try:
        for early in (True, False):
            sim, fea = build_problem(mesh, device=False)
            fea.early_linearisation = early
            u0 = E.pinned_full(mesh.n_vert, 0.0)
            op = [o for _, o in sim.ops if hasattr(o, 'apply_inverse_jacobian')][0]
            if early:
                # round 6: a forward-only evaluation linearises nothing -- the early path starts with the
                # first solve AFTER derivatives have been asked for once
                sim['f'] = E.pinned_array(fs[0]); sim['u'] = u0
                sim.run()
                assert getattr(op, 'A', None) is None and not getattr(op, '_early_done', False)
            one_cycle(sim, fea, E.pinned_array(fs[0]), u0)
            assert not getattr(op, '_early_done', False)
            g = one_cycle(sim, fea, E.pinned_array(fs[1]), u0)
            out[early] = (np.array(E.host_wait(sim['u']), copy=True), np.array(E.host_wait(g), copy=True),
                          float(np.asarray(sim['l2_functional']).ravel()[0]), op)
            if early:
                assert op._early_done is False                      # consumed by this cycle's compute_derivatives
                # a second compute_derivatives without a solve in between takes the ordinary path (and gives the same matrix)
                v1 = np.array(op.A.mat.export_csr()[2], copy=True)
                op.compute_derivatives({'f': sim.values['f']}, {'u': sim.values['u']}, {})
                assert np.array_equal(v1, op.A.mat.export_csr()[2])
            utils_hip.clear_workspaces()
    finally:
        os.environ["FEMO_HOST_VERIFY"] = "1"
    def rel(a, b):
        return float(np.linalg.norm(a - b) / np.linalg.norm(b))
    ru, rg = rel(out[True][0], out[False][0]), rel(out[True][1], out[False][1])
    assert ru < 1e-10 and rg < 1e-10, (ru, rg)
    assert abs(out[True][2] - out[False][2]) <= 1e-10 * abs(out[False][2])
