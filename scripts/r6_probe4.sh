#!/bin/bash
# round 6: HIP API calls between the deferred upload of f and the first early kernel on the model rank
set -u
R=$(pwd); O=$R/gpurun_out/r6_p4; mkdir -p $O
export TMPDIR=/tmp
(cd /tmp && timeout 300 rocprofv3 --hip-runtime-trace --kernel-trace --output-format csv -d $O/trace -- python3 $R/scripts/run_scaling_model.py 215 3 --model-only > $O/model.json 2> $O/err.txt)
ls -R $O/trace | head -20
python3 - <<PY
import csv, glob
api = []
for p in glob.glob("$O/trace/**/*hip_api_trace.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        api.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Function"]))
api.sort()
print(len(api), "api calls")
# last occurrence of a long-ish window: find hipMemcpyAsync calls followed by hipStreamWaitEvent before a launch
idx = [i for i, a in enumerate(api) if a[2] == "hipMemcpyAsync"]
# print the API sequence around the last few deferred uploads: look for the pattern EventRecord, StreamWaitEvent, MemcpyAsync, EventRecord, EventRecord
shown = 0
for i in reversed(idx):
    if i >= 3 and api[i-1][2] == "hipStreamWaitEvent" and api[i-2][2] == "hipEventRecord" and api[i+1][2] == "hipEventRecord" and api[i+2][2] == "hipEventRecord":
        t0 = api[i][0]
        print("---- deferred upload at", t0)
        for a in api[i-6:i+40]:
            print(f"  {(a[0]-t0)/1e3:9.1f} us  {(a[1]-a[0])/1e3:7.1f}  {a[2]}")
        shown += 1
        if shown == 2: break
PY
rm -rf $O/trace
