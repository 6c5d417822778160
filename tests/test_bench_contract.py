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
        cpu_n, n = 20, 215

    for pc in ("bpx", "jacobi"):
        bench.PC = pc
        out = bench.cpu_baseline(Args, [42, 0, 0, 42], 10077696, 59630250, 150048286)
        assert out["kind"] == "port" and out["unit"] == "DOFs/s" and out["value"] > 0 and out["cores"] >= 1
        assert pc.upper() in out["sample"]
    bench.PC = "bpx"
