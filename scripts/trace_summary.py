"""Per-kernel summary of a rocprofv3 --kernel-trace --output-format csv run.
usage: trace_summary.py DIR [MIN_US] [TIMELINE_KERNEL N]
Launches shorter than MIN_US (default 5) are listed in their own column and left out of the averages: the
Krylov loops enqueue iterations ahead of the convergence poll, and those launches return at `if (*done) return;`
(round 1's profile averaged them in and understated the SpMV's duration)."""
import csv
import glob
import os
import sys
from collections import defaultdict

d = sys.argv[1]
min_us = float(sys.argv[2]) if len(sys.argv) > 2 else 5.0
rows = []
for path in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    with open(path) as fh:
        for r in csv.DictReader(fh):
            name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name))
rows.sort()
acc = defaultdict(list)
for s, e, n in rows:
    acc[n].append((e - s) / 1e3)
tot = sum(sum(v) for v in acc.values())
print(f"total kernel time {tot / 1e3:.2f} ms, {len(rows)} launches; averages over launches >= {min_us:g} us")
print("name,calls,early_exit_calls,total_ms,avg_us,min_us,max_us,percent")
for n, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
    real = [t for t in v if t >= min_us] or v
    print(f"{n},{len(real)},{len(v) - len(real) if real is not v else 0},{sum(v) / 1e3:.3f},{sum(real) / len(real):.1f},{min(real):.1f},{max(real):.1f},{100 * sum(v) / tot:.1f}")
if len(sys.argv) > 3:
    key, cnt = sys.argv[3], int(sys.argv[4]) if len(sys.argv) > 4 else 24
    idx = [i for i, r in enumerate(rows) if key in r[2] and (r[1] - r[0]) / 1e3 >= min_us]
    i0 = idx[len(idx) // 2]
    print("--- timeline from the middle:", key)
    for s, e, n in rows[i0:i0 + cnt]:
        print(f"{n[:48]:48s} {(e - s) / 1e3:8.1f} us")
