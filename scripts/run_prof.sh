cd $GRAFT_REPO_ROOT
TAG=${1:-r03a}
for L in headline config3 config4 skewed; do bash scripts/profile.sh $TAG $L > gpurun_out/prof_${TAG}_${L}.txt 2>&1; tail -3 gpurun_out/prof_${TAG}_${L}.txt; done
mkdir -p gpurun_out/profiles_new && cp profiles/${TAG}_* gpurun_out/profiles_new/
ls gpurun_out/profiles_new
