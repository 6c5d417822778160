import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench as B
from femo_amd import engine as E
from femo_amd.engine import Context
from femo_amd.fea import utils_hip
ctx = Context(0)
utils_hip.set_context(ctx)
which = sys.argv[1:]
if "sm" in which:
    t0 = time.time(); r = B.bench_scaling_model(ctx, 215, 5, 60.0, {}); print("sm", round(r["ms_per_cycle_block"], 2), round(time.time() - t0, 1), flush=True)
if "c2" in which:
    r = B.bench_config2(ctx, 10); print("c2", round(r["ms_per_cycle"], 2), flush=True)
r = B.bench_config5(ctx, 5); print("c5", round(r["ms_per_cycle"], 2), flush=True)
r = B.bench_config5(ctx, 5); print("c5 again", round(r["ms_per_cycle"], 2), flush=True)
