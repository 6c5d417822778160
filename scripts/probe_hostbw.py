import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from femo_amd import engine as E
from femo_amd.engine import Context
ctx = Context(0)
n = 59_630_250
print("threads", E._lib.load().femo_host_threads(), "cpus", len(os.sched_getaffinity(0)), open("/sys/fs/cgroup/cpu.max").read().strip())
a = E.pinned_empty(n); b = E.pinned_empty(n); c = np.empty(n); d = np.empty(n)
a[:] = 1.0; c[:] = 2.0; b[:] = 0; d[:] = 0
def t(label, fn, reps=3):
    fn()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    dt = (time.perf_counter() - t0) / reps
    print(f"{label:40s} {dt*1e3:8.2f} ms  {n*8/dt/1e9:7.1f} GB/s (one-way bytes)")
t("host_copy pinned->pinned", lambda: E.host_copy(b, a))
t("host_copy pageable->pinned", lambda: E.host_copy(b, c))
t("host_copy pageable->pageable", lambda: E.host_copy(d, c))
t("host_axpby pinned", lambda: E.host_axpby(-1.0, a, 0.0, b))
t("numpy copyto pageable", lambda: np.copyto(d, c))
for k in range(3):
    x = E.pinned_empty(n)          # fresh / recycled block
    t0 = time.perf_counter(); E.host_copy(x, a); print("copy into new block %.2f ms" % ((time.perf_counter()-t0)*1e3))
    del x
