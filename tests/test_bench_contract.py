"""bench.py's stdout contract, checked without a GPU: the option parser, the JSON-only stdout
wrapper and the CPU-baseline leg."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_quiet_stdout_routes_c_level_prints_to_stderr():
    code = (
        "import os, sys, json\n"
        f"sys.path.insert(0, {ROOT!r})\n"
        "from femo_amd.dist import _quiet_stdout\n"
        "import ctypes\n"
        "with _quiet_stdout():\n"
        "    os.write(1, b'banner from a C library\\n')\n"
        "    ctypes.CDLL(None).printf(b'buffered C stdio banner\\n')\n"
        "    print('python noise')\n"
        "print(json.dumps({'ok': 1}))\n"
    )
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, check=True)
    assert p.stdout.strip().splitlines() == ['{"ok": 1}']
    assert "banner from a C library" in p.stderr and "python noise" in p.stderr
    assert "buffered C stdio banner" in p.stderr


def test_cpu_baseline_leg_runs_the_same_preconditioner():
    sys.path.insert(0, ROOT)
    import bench

    class Args:
        cpu_n, n, jitter = 20, 215, 0.0

    for pc in ("bpx", "jacobi"):
        bench.PC = pc
        out = bench.cpu_baseline(Args, [42, 0, 0, 42], 10077696, 59630250, 150048286)
        assert out["kind"] == "port" and out["unit"] == "DOFs/s" and out["value"] > 0 and out["cores"] >= 1
        assert pc.upper() in out["sample"] and "scaled to n=215" in out["sample"]
    bench.PC = "bpx"

    class Same:                       # the default: the benchmark's own size, nothing extrapolated
        cpu_n, n, jitter = 0, 16, 0.0

    out = bench.cpu_baseline(Same, [30, 0, 0, 30], 17 ** 3, 6 * 16 ** 3, 1)
    assert "nothing scaled" in out["sample"] and out["value"] > 0
    # the port mirrors the engine's Newton noise-floor rule: solves 2 and 3 stop at once
    assert "split_s" in out and out["sample"].count("[") >= 1


def test_cpu_port_newton_noise_rule():
    sys.path.insert(0, ROOT)
    import numpy as np
    from oracle import c_port
    from oracle import femo_oracle as fo
    m = fo.unit_cube_mesh(12)
    f = np.ones(m.n_cell)
    bd = fo.boundary_vertices_box(m.x)
    out = c_port.poisson_cycle(3, m.x, m.conn, f, fo.u_target(m.x), bd, 1e-6, rtol=1e-14, pc="bpx")
    assert out["it_fwd"][0] > 5 and out["it_fwd"][1] <= 2 and out["it_fwd"][2] <= 2
    ref = fo.reference_cycle(m, f, fo.u_target(m.x), bd, np.zeros(len(bd)))
    assert np.abs(out["u"] - ref["u"]).max() <= 1e-10 * np.abs(ref["u"]).max()
    assert np.abs(out["grad"] - ref["grad"]).max() <= 1e-10 * np.abs(ref["grad"]).max()


def test_multi_gpu_flag_without_launcher_starts_a_torchrun_job(monkeypatch):
    """`python bench.py --gpus 2` must not quietly benchmark one GPU (it relaunches itself under torchrun)."""
    sys.path.insert(0, ROOT)
    import bench
    calls = []
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2", "--steps", "1"])
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    import subprocess as sp

    class R:
        returncode = 7

    monkeypatch.setattr(sp, "run", lambda cmd, **k: (calls.append(cmd), R())[1])
    try:
        bench.main()
    except SystemExit as e:
        assert e.code == 7
    assert calls and "torch.distributed.run" in calls[0] and "--nproc-per-node=2" in calls[0]


def test_dst_exact_cycle_matches_the_lu_reference():
    """The full-size checker (oracle/c_port.py::poisson_cycle_dst) = the LU reference cycle to round-off."""
    sys.path.insert(0, ROOT)
    import numpy as np
    from oracle import c_port
    from oracle import femo_oracle as fo
    for d, n in ((3, 10), (2, 24)):
        m = fo.unit_cube_mesh(n) if d == 3 else fo.unit_square_mesh(n)
        xc = fo.centroids(m)
        f = np.prod(np.sin(np.pi * xc), axis=1) * (1.0 + 0.3 * xc[:, 0]) + 0.05
        bd = fo.boundary_vertices_box(m.x)
        ref = fo.reference_cycle(m, f, fo.u_target(m.x), bd, np.zeros(len(bd)))
        out = c_port.poisson_cycle_dst(n, d, m.x, m.conn, f, fo.u_target(m.x), bd, fo.ALPHA_POISSON)
        assert np.abs(out["u"] - ref["u"]).max() <= 1e-13 * np.abs(ref["u"]).max()
        assert np.abs(out["grad"] - ref["grad"]).max() <= 1e-13 * np.abs(ref["grad"]).max()
        assert abs(out["J"] - ref["J"][0]) <= 1e-13 * abs(ref["J"][0])


def test_self_check_maps_renumbered_meshes_back():
    """bench.self_check on a permuted + Morton-reordered cube: the canonical numbering is recovered from the
    coordinates, so exact values in the benchmark's numbering give round-off errors and a wrong entry shows."""
    sys.path.insert(0, ROOT)
    import numpy as np
    import bench
    from femo_amd.fea.mesh import createUnitCubeMesh
    from oracle import femo_oracle as fo
    n = 6
    canon = createUnitCubeMesh(n)
    mesh = canon.permuted(seed=3, cells=True).reordered()
    assert canon.reordered() is not canon and np.array_equal(canon.reordered().conn, canon.conn)      # structured numbering: kept
    assert not np.array_equal(canon.reordered(force=True).conn, canon.conn)
    f = bench.source_fields(mesh, 1)[0]
    om = fo.OMesh(3, mesh.x, mesh.conn, n)
    bd = fo.boundary_vertices_box(mesh.x)
    ref = fo.reference_cycle(om, f, fo.u_target(mesh.x), bd, np.zeros(len(bd)))       # LU in the benchmark's own numbering

    class A:
        pass
    a = A()
    a.n, a.jitter, a.permute, a.reorder = n, 0.0, True, True
    chk = bench.self_check(a, mesh, f, ref["u"], ref["J"][0], ref["grad"])
    assert chk["u_rel_err"] < 1e-12 and chk["grad_rel_err"] < 1e-12 and chk["J_rel_err"] < 1e-12
    bad = ref["grad"].copy()
    bad[5] *= 1.0 + 1e-6
    assert bench.self_check(a, mesh, f, ref["u"], ref["J"][0], bad)["grad_rel_err"] > 1e-9
    a.permute = a.reorder = False
    f0 = bench.source_fields(canon, 1)[0]
    ref0 = fo.reference_cycle(fo.OMesh(3, canon.x, canon.conn, n), f0, fo.u_target(canon.x), fo.boundary_vertices_box(canon.x), np.zeros(len(bd)))
    chk0 = bench.self_check(a, canon, f0, ref0["u"], ref0["J"][0], ref0["grad"])
    assert max(chk0["u_rel_err"], chk0["grad_rel_err"], chk0["J_rel_err"]) < 1e-12
