"""Distils a scripts/profile.sh output directory into small committed files:
  profiles/<tag>_kernel_stats.csv   rocprofv3 --kernel-trace --stats summary (our kernels)
  profiles/<tag>_pmc.json           HBM bytes per launch per kernel from FETCH_SIZE / WRITE_SIZE
PMC handling follows guides/MI355X_MICROARCH.md "HBM": FETCH_SIZE and WRITE_SIZE are in KiB,
collected in separate passes; on gfx950 FETCH_SIZE reports HALF of a wide coalesced
streaming read, so it is doubled.
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
    n = name.split("(")[0]
    n = n.replace("void ", "").replace("rsn::", "")
    return n.strip()


def main():
    out, tag = sys.argv[1], sys.argv[2]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    prof = os.path.join(root, "profiles")
    os.makedirs(prof, exist_ok=True)
    stats = find(os.path.join(out, "trace"), "*kernel_stats.csv")
    if stats:
        rows = list(csv.DictReader(open(stats)))
        keep = [r for r in rows if "rsn::" in r.get("Name", "")]
        with open(os.path.join(prof, tag + "_kernel_stats.csv"), "w", newline="") as f:
            w = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
            w.writeheader()
            for r in keep:
                w.writerow(r)
        for r in keep:
            print("%-60s calls %5s avg %10.1f us  %5s%%" % (short(r["Name"])[:60], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
    res = defaultdict(dict)
    for kind, key, scale in (("pmc_fetch", "fetch_bytes", 2.0), ("pmc_write", "write_bytes", 1.0)):
        p = find(os.path.join(out, kind), "*counter_collection.csv")
        if not p:
            continue
        acc = defaultdict(lambda: [0.0, 0])
        for r in csv.DictReader(open(p)):
            if "rsn::" not in r.get("Kernel_Name", ""):
                continue
            a = acc[short(r["Kernel_Name"])]
            a[0] += float(r["Counter_Value"]) * 1024.0 * scale
            a[1] += 1
        for k, (tot, cnt) in acc.items():
            res[k][key] = tot / cnt
    final = {}
    for k, v in res.items():
        v["hbm_bytes"] = v.get("fetch_bytes", 0) + v.get("write_bytes", 0)
        final[k] = v                                  # keyed by the kernel's own (template) name: bench.py picks by prefix
        print("%-40s fetch %.3e  write %.3e B per launch" % (k, v.get("fetch_bytes", 0), v.get("write_bytes", 0)))
    json.dump({k: round(v["hbm_bytes"]) for k, v in final.items()}, open(os.path.join(prof, tag + "_pmc.json"), "w"), indent=1, sort_keys=True)
    json.dump(final, open(os.path.join(prof, tag + "_pmc_detail.json"), "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
