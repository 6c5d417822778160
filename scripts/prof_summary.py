"""Per-kernel summary (calls, total, average) and one-iteration timeline from a rocprofv3 results .db."""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
rows = list(cur.execute("select name, count(*), sum(end-start)/1e6, avg(end-start)/1e3, min(end-start)/1e3, max(end-start)/1e3 "
                        "from kernels group by name order by 3 desc"))
tot = sum(r[2] for r in rows)
print(f"total kernel time {tot:.2f} ms")
print("name,calls,total_ms,avg_us,min_us,max_us,percent")
for r in rows[: int(sys.argv[2]) if len(sys.argv) > 2 else 40]:
    name = r[0].replace("(anonymous namespace)::", "").replace("void ", "")
    name = name.split("(")[0]
    print(f"{name},{r[1]},{r[2]:.3f},{r[3]:.1f},{r[4]:.1f},{r[5]:.1f},{100 * r[2] / tot:.1f}")
if len(sys.argv) > 3:
    ks = list(cur.execute("select name,start,end from kernels order by start"))
    idx = [i for i, r in enumerate(ks) if sys.argv[3] in r[0]]
    i0 = idx[len(idx) // 2]
    print("--- timeline from the middle", sys.argv[3])
    for r in ks[i0:i0 + int(sys.argv[4]) if len(sys.argv) > 4 else i0 + 24]:
        name = r[0].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        print(f"{name[:48]:48s} {(r[2] - r[1]) / 1e3:8.1f} us")
