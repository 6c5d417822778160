"""profiles/rNN_pmc_traffic.json from the hardware-counter tables of scripts/collect_profiles.sh.
usage: pmc_traffic.py OUT.json TABLE_n215.csv [TABLE_n215_permute.csv] [--also KEY=TABLE.csv:KERNEL ...]
(--also: bytes per launch of KERNEL in TABLE under KEY, e.g. spmv_n100=pmc_kernels_n100.csv:k_spmv_sell<1, true>)
bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: FETCH_SIZE / WRITE_SIZE are in KiB; the factor 2 is the gfx950
correction for wide coalesced streaming reads (MI355X_MICROARCH.md, HBM section; checked in round 1 on k_cg_update_xp:
counter / bytes = 0.501).  Kernels that gather (assembly, permuted SpMV) use access widths the guide calls
uncalibrated: their figures are upper estimates."""
import csv
import json
import sys


def table(path):
    t = {}
    with open(path) as fh:
        for r in csv.DictReader(fh):
            t.setdefault(r["kernel"], {})[r["counter"]] = float(r["mean_per_dispatch"])
    return t


def entry(c):
    f, w = c.get("FETCH_SIZE", 0.0), c.get("WRITE_SIZE", 0.0)
    return {"fetch_KiB": f, "write_KiB": w, "bytes_2F_plus_W": (2 * f + w) * 1024}


import os
import subprocess


def _commit():
    # the build the counters were collected from: FEMO_COLLECT_COMMIT (set by the collection script: the GPU box has no .git) or git
    c = os.environ.get("FEMO_COLLECT_COMMIT")
    if c:
        return c
    try:
        return subprocess.run(["git", "rev-parse", "--short=12", "HEAD"], capture_output=True, text=True, timeout=10).stdout.strip() or "unknown"
    except Exception:
        return "unknown"


out = {"note": __doc__.split("\n", 3)[3].strip(), "collected_at_commit": _commit()}
args = sys.argv[2:]
also = []
if "--also" in args:
    k = args.index("--also")
    also, args = args[k + 1:], args[:k]
for tag, path in zip(("n215", "n215_permute"), args):
    t = table(path)
    spmv_name = next((k for k in t if k.startswith("k_spmv_sell<1, true")), "k_spmv_sell<1, true>")   # (<1, true, NT> since round 5)
    spmv = t.get(spmv_name, {})
    key = "spmv_" + tag.replace("_permute", "_permuted")
    e = entry(spmv)
    out[key] = e["bytes_2F_plus_W"]
    out[key + "_fetch_KiB"], out[key + "_write_KiB"] = e["fetch_KiB"], e["write_KiB"]
    out["other_kernels_" + tag] = {k: entry(c) for k, c in t.items()
                                   if k != spmv_name and ("FETCH_SIZE" in c) and
                                   any(s in k for s in ("poisson_system", "p1_row_walk", "restrict_bricks", "prolong_mesh", "pcg_xr", "lattice_prolong3", "lattice_coarse_m", "k_spmv_sell<0", "k_vec_diff", "k_scale_sell"))}
for spec in also:
    key, rest = spec.split("=", 1)
    path, kern = rest.split(":", 1)
    tb = table(path)
    e = entry(tb.get(kern, tb.get(next((k for k in tb if k.startswith(kern.rstrip(">"))), kern), {})))
    if e["fetch_KiB"] > 0:
        out[key] = e["bytes_2F_plus_W"]
        out[key + "_fetch_KiB"], out[key + "_write_KiB"] = e["fetch_KiB"], e["write_KiB"]
json.dump(out, open(sys.argv[1], "w"), indent=1)
print(json.dumps(out, indent=1)[:1500])
