"""`bench.py --gpus W` with the W ranks EMULATED on one GPU (one host thread + one context per rank, host-staged collectives):
the record of rank 0 as JSON.  Timings mean nothing (all ranks share the GPU, collectives go through the host); what it shows
at the benchmark's size is the partition, the merged BPX-PCG loop's iteration counts and collective counts, the sizes of the
exchanged buffers and the self-check against the DST-exact cycle.  usage: run_emulated_ranks_bench.py [n=215] [world=8]"""
import json
import os
import sys
import threading
from argparse import Namespace

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from femo_amd.dist import ThreadControl
from bench import run_distributed_bench
from femo_amd.engine import Context, EmuGroup
from femo_amd.fea import utils_hip

n = int(sys.argv[1]) if len(sys.argv) > 1 else 215
world = int(sys.argv[2]) if len(sys.argv) > 2 else 8
group = EmuGroup(world)
shared = ThreadControl.Shared(world)
args = Namespace(n=n, steps=2, warmup=1, pc="bpx", jitter=0.0, cpu_n=0, no_cpu_baseline=True)
out, err = [None] * world, [None] * world


def body(rank):
    try:
        ctx = Context(0)
        control = ThreadControl(rank, shared, group)
        control.init_comm(ctx)
        utils_hip.set_context(ctx, thread_local=True)
        try:
            out[rank] = run_distributed_bench(args, ctx, control, cpu_baseline=False)
            ctx.sync()
        finally:
            utils_hip.set_context(None, thread_local=True)
    except BaseException as e:          # noqa: BLE001
        err[rank] = e
        shared.barrier.abort()


threads = [threading.Thread(target=body, args=(r,)) for r in range(world)]
for t in threads:
    t.start()
for t in threads:
    t.join(timeout=1500)
for e in err:
    if e is not None:
        raise e
r = out[0]
r["note"] = (f"{world} ranks emulated on ONE GPU (threads, host-staged collectives): value / ms_per_step are NOT multi-GPU timings; "
             "the record documents the partition, iteration and collective counts and the self-check at this size")
print(json.dumps(r))
