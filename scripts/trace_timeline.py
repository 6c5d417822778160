"""Timeline of ONE cycle out of a rocprofv3 --kernel-trace [--memory-copy-trace] CSV run: every kernel and copy with
its duration and the idle gap before it, from the K-th launch of a marker kernel to the next one.
usage: trace_timeline.py DIR MARKER_KERNEL K [collapse]
`collapse`: runs of the PCG iteration's kernels are summarised per solve instead of listed."""
import csv, glob, os, sys
d, marker, k = sys.argv[1], sys.argv[2], int(sys.argv[3])
collapse = len(sys.argv) > 4
ev = []
for path in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(path)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]))
for path in glob.glob(os.path.join(d, "**", "*memory_copy_trace.csv"), recursive=True):
    for r in csv.DictReader(open(path)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", "") + " " + r.get("Bytes", r.get("Size", ""))))
ev.sort()
marks = [i for i, e in enumerate(ev) if marker in e[2]]
i0, i1 = marks[k], marks[k + 1]
t0 = ev[i0][0]
print(f"cycle window {(ev[i1][0] - t0) / 1e3:.1f} us, {i1 - i0} events")
LOOP = ("k_spmv_sell<1, true", "k_pcg_xr", "k_restrict_bricks", "k_lattice_coarse", "k_lattice_prolong3", "k_prolong_mesh", "k_pcg_")
busy = 0.0
prev_end = ev[i0][0]
run = None
def flush():
    global run
    if run:
        print(f"  [{run['n']} loop kernels: busy {run['busy']:.1f} us, gaps {run['gap']:.1f} us, span {run['busy'] + run['gap']:.1f} us]")
    run = None
for s, e, n in ev[i0:i1]:
    gap = (s - prev_end) / 1e3
    dur = (e - s) / 1e3
    busy += dur
    if collapse and any(n.startswith(x) for x in LOOP):
        if run is None:
            run = {"n": 0, "busy": 0.0, "gap": 0.0}
            print(f"t={(s - t0) / 1e3:9.1f}  gap {gap:7.1f}  (loop starts)")
            gap = 0.0
        run["n"] += 1; run["busy"] += dur; run["gap"] += max(gap, 0.0)
    else:
        flush()
        print(f"t={(s - t0) / 1e3:9.1f}  gap {gap:7.1f}  {dur:8.1f} us  {n[:70]}")
    prev_end = max(prev_end, e)
flush()
print(f"busy {busy:.1f} us of {(ev[i1][0] - t0) / 1e3:.1f}")
