"""bench.py end to end on a small mesh: the record the driver stores must be internally consistent
(round 1 shipped iteration counts and a forward/adjoint split corrupted by a 64-entry log cap)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(*flags):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *flags], capture_output=True, text=True,
                       timeout=900, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-4000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, p.stdout
    return json.loads(lines[0])


def test_bench_record_with_twenty_steps():
    r = _bench("--mesh-n", "12", "--steps", "20", "--warmup", "1", "--cpu-n", "12")
    c = r["config"]
    assert r["n_gpus"] == 1 and r["steps"] == 20 and r["unit"] == "DOFs/s" and r["dtype"] == "f64"
    assert abs(r["value"] - c["n_dof"] / (r["ms_per_step"] * 1e-3)) < 1e-6 * r["value"]
    assert c["linear_solves_per_step"] == 4                 # Newton's three solves + the adjoint solve, every step
    its = c["cg_iterations_per_step"]
    assert len(its) == 4 and its[0] > 5 and its[3] > 5 and its[1] <= 2 and its[2] <= 2
    s = c["split_ms_per_step"]
    assert s["forward_solves"] > 0 and s["adjoint_solve"] > 0
    assert abs(s["forward_solves"] + s["adjoint_solve"] - c["cg_ms_per_step"]) < 1e-9
    assert abs(sum(s.values()) - r["ms_per_step"]) < 1e-6 * r["ms_per_step"]
    assert "host" in c["boundary"]
    # the array boundary: f goes up once per step, repeated uploads are elided
    # (f only: the cold-start state is a constant block, filled on the device; the adjoint seed -dJ/du is formed on
    # the device from the vector it negates)
    assert c["pcie"]["h2d_bytes_per_step"] == 8 * c["n_cell"]
    assert c["pcie"]["d2h_bytes_per_step"] == 8 * (2 * c["n_cell"] + 3 * c["n_dof"])
    assert c["pcie"]["uploads_elided_per_step"] >= 7
    rf = r["roofline"]
    assert rf["bound"] == "hbm" and rf["peak"] == 8000.0 and rf["launches_timed"] > 0
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12 and rf["frac_physical"] > 0
    assert r["device_resident"]["ms_per_step"] > 0 and r["pageable_boundary"]["ms_per_step"] > 0
    cb = r["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and "nothing scaled" in cb["sample"]


def test_bench_permuted_numbering():
    r = _bench("--mesh-n", "10", "--steps", "2", "--warmup", "1", "--permute", "--no-cpu-baseline", "--no-pcie")
    assert r["config"]["permuted"] is True and r["config"]["regular_slices"] == 0
    assert r["config"]["linear_solves_per_step"] == 4
