"""Idle gaps of the GPU between consecutive kernels of a rocprofv3 --kernel-trace CSV run (host-side stalls).
usage: trace_gaps.py DIR [MIN_GAP_US]"""
import csv, glob, os, sys
d = sys.argv[1]; min_gap = float(sys.argv[2]) if len(sys.argv) > 2 else 300.0
rows = []
for path in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(path)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]))
rows.sort()
t0 = rows[0][0]
busy = sum(e - s for s, e, _ in rows)
print(f"span {(rows[-1][1]-t0)/1e6:.1f} ms, busy {busy/1e6:.1f} ms")
for (s0, e0, n0), (s1, e1, n1) in zip(rows, rows[1:]):
    gap = (s1 - e0) / 1e3
    if gap >= min_gap:
        print(f"t={(e0-t0)/1e6:9.2f} ms  gap {gap/1e3:8.2f} ms  after {n0[:40]:40s} before {n1[:40]}")
