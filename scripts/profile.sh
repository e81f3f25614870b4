#!/bin/bash
# Runs on the GPU box (via gpurun): kernel-trace stats and PMC passes for bench.py.
# usage: scripts/profile.sh <tag> [bench args...]
# Writes gpurun_out/prof_<tag>/ ; scripts/summarize_prof.py then distils it into profiles/.
TAG=${1:-r01}; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 3 --warmup 2 --no-cpu $@"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py $ARGS > $OUT/trace.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- python3 $R/bench.py $ARGS > $OUT/pmc_fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- python3 $R/bench.py $ARGS > $OUT/pmc_write.log 2>&1
cd $R && python3 scripts/summarize_prof.py $OUT $TAG > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
