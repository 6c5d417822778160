"""The 8-rank block of bench.py's "scaling_model" leg on its own (for kernel traces): one rank's (n/2)^3 block of the
headline cube with the whole mesh's preconditioner lattice.  usage: run_scaling_model.py [n_global] [steps] [--model-only]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as B  # noqa: E402
from femo_amd.engine import Context  # noqa: E402
from femo_amd.fea import utils_hip  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 215
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
ctx = Context(0)
utils_hip.set_context(ctx)
print(json.dumps(B.bench_scaling_model(ctx, n, steps, float("nan"), {}, one_rank_leg="--model-only" not in sys.argv, global_its=[28, 0, 0, 28] if n == 215 else None)))
