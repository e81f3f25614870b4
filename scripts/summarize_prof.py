"""Distils a scripts/profile.sh output directory into small committed files:
  profiles/<tag>_kernel_stats_<label>.csv   rocprofv3 --kernel-trace --stats summary (our kernels)
  profiles/<tag>_pmc_<label>.json           HBM bytes per kernel from FETCH_SIZE / WRITE_SIZE
<label> names the ONE workload the profiled command ran (headline, config3, config4, skewed, 2b, 5): traffic / algorithmic
bytes can be recomputed per config from the file alone.  A kernel name is launched at several sizes within one call (the
1 GiB launch next to 64-tile samples and second looks over a handful of tiles), so every kernel carries
  launches, fetch_mean / fetch_max, write_mean / write_max, hbm_max = fetch_max + write_max
and the per-config ratio uses the *_max figures (the full-size launch; fetch and write come from separate passes, their
largest launches are the same dispatch because both scale with the launch's tile count).
PMC handling follows guides/MI355X_MICROARCH.md "HBM": FETCH_SIZE and WRITE_SIZE are in KiB, collected in separate
passes; on gfx950 FETCH_SIZE reports HALF of a wide coalesced streaming read, so it is doubled.
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def find(d, pat):
    r = glob.glob(os.path.join(d, "**", pat), recursive=True)
    return r[0] if r else None


def short(name):
    n = name.replace("(anonymous namespace)::", "").split("(")[0]
    n = n.replace("void ", "").replace("rsn::", "")
    return n.strip()


def main():
    out, tag = sys.argv[1], sys.argv[2]
    label = sys.argv[3] if len(sys.argv) > 3 else "headline"
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    prof = os.path.join(root, "profiles")
    os.makedirs(prof, exist_ok=True)
    stats = find(os.path.join(out, "trace"), "*kernel_stats.csv")
    if stats:
        rows = list(csv.DictReader(open(stats)))
        keep = [r for r in rows if "rsn::" in r.get("Name", "")]
        with open(os.path.join(prof, "%s_kernel_stats_%s.csv" % (tag, label)), "w", newline="") as f:
            w = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
            w.writeheader()
            for r in keep:
                w.writerow(r)
        for r in keep:
            print("%-60s calls %5s avg %10.1f us  max %10.1f us  %5s%%" % (short(r["Name"])[:60], r["Calls"], float(r["AverageNs"]) / 1e3,
                                                                           float(r.get("MaxNs", 0) or 0) / 1e3, r["Percentage"]))
    res = defaultdict(dict)
    for kind, key, scale in (("pmc_fetch", "fetch", 2.0), ("pmc_write", "write", 1.0)):
        p = find(os.path.join(out, kind), "*counter_collection.csv")
        if not p:
            continue
        acc = defaultdict(list)
        per_dispatch = defaultdict(float)
        for r in csv.DictReader(open(p)):                # one row per (dispatch, counter instance): sum the instances of a dispatch
            if "rsn::" not in r.get("Kernel_Name", ""):
                continue
            per_dispatch[(short(r["Kernel_Name"]), r.get("Dispatch_Id", r.get("Correlation_Id", "")))] += float(r["Counter_Value"]) * 1024.0 * scale
        for (k, _), v in per_dispatch.items():
            acc[k].append(v)
        for k, vals in acc.items():
            res[k]["launches"] = len(vals)
            res[k][key + "_mean"] = sum(vals) / len(vals)
            res[k][key + "_max"] = max(vals)
    final = {}
    for k, v in sorted(res.items()):
        v["hbm_max"] = v.get("fetch_max", 0) + v.get("write_max", 0)
        final[k] = {a: (round(b) if isinstance(b, float) else b) for a, b in v.items()}
        print("%-40s launches %4d  fetch max %.3e (mean %.3e)  write max %.3e (mean %.3e) B" % (
            k, v.get("launches", 0), v.get("fetch_max", 0), v.get("fetch_mean", 0), v.get("write_max", 0), v.get("write_mean", 0)))
    json.dump(final, open(os.path.join(prof, "%s_pmc_%s.json" % (tag, label)), "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
