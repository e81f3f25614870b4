#!/bin/bash
# Runs on the GPU box: the non-headline configurations of BASELINE.json, per-kernel event timings.
# usage: scripts/other_configs.sh <tag>   -> gpurun_out/<tag>_other_configs.txt (+ rocprofv3 stats of the LZSS path)
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${TAG}_other_configs.txt
mkdir -p $R/gpurun_out
{
  timeout 300 python3 $R/scripts/quick_bench2.py 1024
  timeout 300 python3 $R/scripts/quick_2b.py 1024
  for k in text text1 period random; do timeout 300 python3 $R/scripts/quick_lzss.py $k 1024; done
  timeout 300 python3 $R/scripts/quick_layered.py 1024
} 2>&1 | grep -v amdgpu.ids > $OUT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${TAG}_lzss -- python3 $R/scripts/quick_lzss.py text 1024 > $R/gpurun_out/prof_${TAG}_lzss.log 2>&1
cp $(ls $R/gpurun_out/prof_${TAG}_lzss/*/*kernel_stats.csv | head -1) $R/gpurun_out/${TAG}_lzss_text_kernel_stats.csv 2>/dev/null
cat $OUT
