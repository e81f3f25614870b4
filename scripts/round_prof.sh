#!/bin/bash
# GPU box: the round's per-workload profiles (kernel stats + HBM bytes), VALU counts of every kernel of `skewed`, config 3 and config 4
cd "${GRAFT_REPO_ROOT:-$(pwd)}" || exit 1
TAG=${1:-r06a}
mkdir -p gpurun_out/profiles_new
for L in headline config3 config4 skewed 2b; do bash scripts/profile.sh $TAG $L > gpurun_out/prof_${TAG}_${L}.txt 2>&1; tail -2 gpurun_out/prof_${TAG}_${L}.txt; done
python3 scripts/traffic_report.py $TAG > profiles/${TAG}_traffic.txt 2>&1
for W in skewed 3 4; do bash scripts/pmc_valu.sh $W > profiles/${TAG}_valu_$([ $W = 4 ] && echo config4 || ([ $W = 3 ] && echo config3 || echo $W)).txt 2>&1; done
cp profiles/${TAG}_* gpurun_out/profiles_new/
ls gpurun_out/profiles_new | grep $TAG
