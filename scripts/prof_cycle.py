import os, sys, time, cProfile, pstats
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import bench as B
from femo_amd import engine as E
from femo_amd.engine import Context, DeviceArray, Vec
from femo_amd.fea import utils_hip
from femo_amd.fea.mesh import createUnitCubeMesh
n = int(sys.argv[1]) if len(sys.argv) > 1 else 215
ctx = Context(0); utils_hip.set_context(ctx)
mesh = createUnitCubeMesh(n)
f_host = B.source_fields(mesh, 3)
for mode in ("device", "host"):
    sim, fea = B.build_problem(mesh, device=(mode == "device"))
    if mode == "host":
        fs = [E.pinned_array(f) for f in f_host]; u0 = E.pinned_array(np.zeros(mesh.n_vert))
    else:
        fs = [DeviceArray(Vec(ctx, mesh.n_cell).set(f)) for f in f_host]; u0 = None
    for k in range(2): B.one_cycle(sim, fea, fs[k], u0)
    ctx.sync()
    pr = cProfile.Profile(); pr.enable()
    t0 = time.perf_counter()
    g = B.one_cycle(sim, fea, fs[2], u0)
    ctx.sync()
    dt = time.perf_counter() - t0
    pr.disable()
    print("=====", mode, "cycle %.1f ms" % (dt * 1e3))
    pstats.Stats(pr).sort_stats("tottime").print_stats(14)
    del sim, fea, fs
