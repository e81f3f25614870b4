#!/bin/bash
# runs on the GPU box (gpurun): from the snapshot's root, or from the current directory when started by hand
cd "${GRAFT_REPO_ROOT:-$(pwd)}" || exit 1
TAG=${1:-r04a}
for L in headline config3 config4 skewed; do bash scripts/profile.sh $TAG $L > gpurun_out/prof_${TAG}_${L}.txt 2>&1; tail -3 gpurun_out/prof_${TAG}_${L}.txt; done
mkdir -p gpurun_out/profiles_new && cp profiles/${TAG}_* gpurun_out/profiles_new/
ls gpurun_out/profiles_new
