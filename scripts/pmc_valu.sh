#!/bin/bash
# VALU / SALU / LDS instruction counts and busy cycles of every kernel of one bench workload (default: config 4); runs on the GPU box.
# A kernel whose VALU count x ~3.5 cycles / 1024 SIMDs comes to its duration is bound by vector issue (scripts/valu_probe.cpp).
R=${GRAFT_REPO_ROOT:-$(pwd)}
W=${1:-4}
OUT=$R/gpurun_out/pmc_valu_$W
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES --kernel-trace --output-format csv -d $OUT -- python3 $R/bench.py $( [ "$W" = headline ] && echo "--steps 3 --warmup 2 --no-cpu --no-others" || echo "--profile-only $W" ) > $OUT/log.txt 2>&1
cd $R
python3 - "$OUT" <<'PY'
import csv, glob, os, sys
from collections import defaultdict
p = glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True)
best = {}
for r in csv.DictReader(open(p[0])):
    k = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "").replace("rsn::", "")
    key = (k, r["Dispatch_Id"])
    best.setdefault(key, {})[r["Counter_Name"]] = float(r["Counter_Value"])
big = {}
for (k, d), c in best.items():
    if k not in big or c.get("GRBM_GUI_ACTIVE", 0) > big[k].get("GRBM_GUI_ACTIVE", 0): big[k] = c
print("%-46s %9s %10s %10s %10s  %s" % ("kernel (its largest launch)", "ms", "VALU", "SALU", "LDS", "VALU x 3.5 cyc / 1024 SIMDs, as a share of the kernel"))
for k, c in sorted(big.items(), key=lambda kv: -kv[1].get("GRBM_GUI_ACTIVE", 0))[:24]:
    cyc = c.get("GRBM_GUI_ACTIVE", 0) / 8
    if cyc < 2e4: continue
    print("%-46s %9.3f %10.3g %10.3g %10.3g  %.2f" % (k[:46], cyc / 2.4e6, c.get("SQ_INSTS_VALU", 0), c.get("SQ_INSTS_SALU", 0), c.get("SQ_INSTS_LDS", 0), c.get("SQ_INSTS_VALU", 0) * 3.5 / 1024 / cyc))
PY
