#!/bin/bash
# Runs on the GPU box (via gpurun): kernel-trace stats and PMC passes for bench.py, ONE workload per invocation so that the
# committed summaries can be recomputed from per config.
# usage: scripts/profile.sh <tag> <label> [bench args...]
#   label = headline -> bench.py --no-others            (config 2a: the timed region only)
#   label = config3 | config4 | skewed | 2b | 5 -> bench.py --profile-only <that config>   (that workload alone: no headline step)
# Writes gpurun_out/prof_<tag>_<label>/ ; scripts/summarize_prof.py distils it into profiles/<tag>_*_<label>.*
TAG=${1:-r03}; LABEL=${2:-headline}; shift; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_${TAG}_${LABEL}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
case $LABEL in
  headline) ARGS="--steps 3 --warmup 2 --no-cpu --no-others $@" ;;
  config3) ARGS="--profile-only 3 $@" ;;
  config4) ARGS="--profile-only 4 $@" ;;
  *) ARGS="--profile-only $LABEL $@" ;;
esac
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py $ARGS > $OUT/trace.log 2>&1
timeout 900 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- python3 $R/bench.py $ARGS > $OUT/pmc_fetch.log 2>&1
timeout 900 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- python3 $R/bench.py $ARGS > $OUT/pmc_write.log 2>&1
cd $R && python3 scripts/summarize_prof.py $OUT $TAG $LABEL > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
