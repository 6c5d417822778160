"""Host-side (Python) profile of the config-2 cycle: where the interpreter spends the time between the launches.
usage: python scripts/profile_host_c2.py [n] [cycles]"""
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

import bench as B  # noqa: E402
from femo_amd import engine as E  # noqa: E402
from femo_amd.engine import Context  # noqa: E402
from femo_amd.fea import utils_hip  # noqa: E402
from femo_amd.fea.mesh import createUnitCubeMesh  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
cycles = int(sys.argv[2]) if len(sys.argv) > 2 else 60
ctx = Context(0)
utils_hip.set_context(ctx)
mesh = createUnitCubeMesh(n)
sim, fea = B.build_problem(mesh, device=False)
mesh.device(ctx)
fs = [E.pinned_array(f) for f in B.source_fields(mesh, 3)]
u0 = E.pinned_full(mesh.n_vert, 0.0)
B._prime_pool(sim)
for k in range(10):
    B.one_cycle(sim, fea, fs[k % 3], u0)
ctx.sync()
t0 = time.perf_counter()
for k in range(cycles):
    B.one_cycle(sim, fea, fs[k % 3], u0)
ctx.sync()
print(f"plain: {(time.perf_counter() - t0) / cycles * 1e3:.3f} ms per cycle")
pr = cProfile.Profile()
pr.enable()
for k in range(cycles):
    B.one_cycle(sim, fea, fs[k % 3], u0)
ctx.sync()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
st.sort_stats("cumulative").print_stats(45)
