#!/bin/bash
# usage: scripts/pmc.sh <tag> <counters...> -- <python script args>   (runs on the GPU box)
TAG=$1; shift
CTRS=""
while [ "$1" != "--" ]; do CTRS="$CTRS $1"; shift; done
shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --pmc $CTRS --kernel-trace --output-format csv -d $OUT -- python3 "$@" > $OUT/log.txt 2>&1
cd $R
python3 - "$OUT" <<'PY'
import csv, glob, os, sys
from collections import defaultdict
p = glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True)
acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
for r in csv.DictReader(open(p[0])):
    k = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "").replace("rsn::", "")
    a = acc[k][r["Counter_Name"]]
    a[0] += float(r["Counter_Value"]); a[1] += 1
for k, d in acc.items():
    if not k.startswith("k_"): continue
    print(k)
    for c, (t, n) in sorted(d.items()):
        print("    %-28s %.4g per launch (%d launches)" % (c, t / n, n))
PY
