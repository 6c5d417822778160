"""Where the host-boundary cycle spends its time: every operator method of one bench cycle timed (stream
synchronised before and after) with NumPy arrays at the boundary and with DeviceArrays, side by side."""
import os
import sys
import time
from collections import OrderedDict

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import bench as B
from femo_amd import engine as E
from femo_amd.csdl_opt import output_model, state_model
from femo_amd.engine import Context, DeviceArray, Vec
from femo_amd.fea import utils_hip
from femo_amd.fea.mesh import createUnitCubeMesh

n = int(sys.argv[1]) if len(sys.argv) > 1 else 215
ctx = Context(0)
utils_hip.set_context(ctx)
mesh = createUnitCubeMesh(n)
T = OrderedDict()


def timed(cls, name):
    fn = getattr(cls, name)

    def wrapper(self, *a, **k):
        ctx.sync()
        t0 = time.perf_counter()
        out = fn(self, *a, **k)
        ctx.sync()
        key = f"{cls.__name__}.{name}" + (f"[{a[-1]}]" if a and isinstance(a[-1], str) else "")
        T.setdefault(key, []).append((time.perf_counter() - t0) * 1e3)
        return out
    setattr(cls, name, wrapper)


for cls, names in ((state_model.StateOperation, ("solve_residual_equations", "compute_derivatives", "compute_jacvec_product", "apply_inverse_jacobian")),
                   (output_model.OutputOperation, ("compute", "compute_derivatives"))):
    for nm in names:
        timed(cls, nm)

f_host = B.source_fields(mesh, 3)
res = {}
for mode in ("host", "device"):
    sim, fea = B.build_problem(mesh, device=(mode == "device"))
    if mode == "host":
        fs = [E.pinned_array(f) for f in f_host]
        u0 = E.pinned_array(np.zeros(mesh.n_vert))
    else:
        fs = [DeviceArray(Vec(ctx, mesh.n_cell).set(f)) for f in f_host]
        u0 = None
    for k in range(2):
        B.one_cycle(sim, fea, fs[k], u0)
    T.clear()
    ctx.sync()
    t0 = time.perf_counter()
    B.one_cycle(sim, fea, fs[2], u0)
    ctx.sync()
    total = (time.perf_counter() - t0) * 1e3
    res[mode] = (total, {k: sum(v) for k, v in T.items()})
    del sim, fea, fs
print(f"{'method':55s} {'host ms':>9s} {'device ms':>10s}")
for k in res["host"][1]:
    print(f"{k:55s} {res['host'][1][k]:9.2f} {res['device'][1].get(k, float('nan')):10.2f}")
print(f"{'sum of methods':55s} {sum(res['host'][1].values()):9.2f} {sum(res['device'][1].values()):10.2f}")
print(f"{'whole cycle (incl. driver)':55s} {res['host'][0]:9.2f} {res['device'][0]:10.2f}")
