"""Where does the non-CG time of one cycle go?  Wraps the utils_hip entry points with
synchronising timers (diagnostic only)."""
import sys, time, os, functools, collections
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import numpy as np
from femo_amd.engine import Context, DeviceArray, Vec
from femo_amd.fea import utils_hip
from femo_amd.fea.mesh import createUnitCubeMesh
import femo_amd.engine as E
import bench

n = int(sys.argv[1]) if len(sys.argv) > 1 else 215
ctx = Context(0); utils_hip.set_context(ctx)
mesh = createUnitCubeMesh(n)
sim, fea = bench.build_problem(mesh, device=True)
fs = [DeviceArray(Vec(ctx, mesh.n_cell).set(f)) for f in bench.source_fields(mesh, 2)]
bench.one_cycle(sim, fea, fs[0]); ctx.sync()
acc = collections.OrderedDict()
def wrap(mod, name):
    fn = getattr(mod, name)
    @functools.wraps(fn)
    def w(*a, **k):
        ctx.sync(); t = time.perf_counter()
        r = fn(*a, **k)
        ctx.sync(); d = time.perf_counter() - t
        e = acc.setdefault(name, [0, 0.0]); e[0] += 1; e[1] += d
        return r
    setattr(mod, name, w)
for name in ["assemble_residual", "assemble_jacobian", "assemble_system", "assemble_dRdf", "newton_rhs", "dRdf_apply",
             "functional_value", "functional_grad_u", "functional_grad_f"]:
    wrap(E, name)
for cls, name in [(E.Mat, "solve_cg"), (E.Mat, "mult"), (E.Vec, "__init__"), (E.Vec, "dot"), (E.Vec, "axpy"),
                  (E.Vec, "copy_from"), (E.Vec, "fill"), (E.Mat, "__init__")]:
    wrap(cls, name)
ctx.sync(); t0 = time.perf_counter()
bench.one_cycle(sim, fea, fs[1]); ctx.sync()
tot = time.perf_counter() - t0
print(f"cycle {tot*1e3:.1f} ms (with sync-instrumentation)")
s = 0
for k, (c, t) in acc.items():
    print(f"  {k:22s} x{c:3d} {t*1e3:9.2f} ms"); s += t
print(f"  accounted {s*1e3:.1f} ms; unaccounted {1e3*(tot-s):.1f} ms")
for i in utils_hip.LAST_KSP_INFO[-4:]:
    print("  ksp", i)
print("cpu probe:", os.cpu_count(), len(os.sched_getaffinity(0)))
os.system("cat /sys/fs/cgroup/cpu.max 2>/dev/null; nproc; free -g | head -2; lscpu | grep -E 'Model name|Socket|^CPU\\(s\\)' ")
