"""Mean counter value per kernel and dispatch from rocprofv3 --pmc CSV output directories.
usage: pmc_table.py OUT.csv DIR [DIR ...]   (every DIR is one --pmc pass written with --output-format csv)"""
import csv
import glob
import os
import sys
from collections import defaultdict

out, dirs = sys.argv[1], sys.argv[2:]
acc = defaultdict(lambda: [0.0, 0])
for d in dirs:
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(path) as fh:
            for row in csv.DictReader(fh):
                k = row.get("Kernel_Name", "")
                k = k.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
                a = acc[(k, row["Counter_Name"])]
                a[0] += float(row["Counter_Value"])
                a[1] += 1
with open(out, "w") as fh:
    fh.write("kernel,counter,mean_per_dispatch,dispatches\n")
    for (k, c), (s, n) in sorted(acc.items()):
        fh.write(f'"{k}",{c},{s / n:.6g},{n}\n')
print(open(out).read())
