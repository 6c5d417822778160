import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as B
from femo_amd.engine import Context
from femo_amd.fea import utils_hip
from femo_amd.fea.fea_hip import FEA
ctx = Context(0)
utils_hip.set_context(ctx)
for mode in sys.argv[1:]:
    FEA.deferred_uploads = mode != "nodefer"
    if mode == "classic": os.environ["FEMO_PCG_CLASSIC"] = "1"
    r = B.bench_config5(ctx, 5)
    print(mode, round(r["ms_per_cycle"], 2), r["cg_iterations_per_cycle"], flush=True)
