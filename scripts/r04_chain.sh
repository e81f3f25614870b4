#!/bin/bash
# GPU box: k_match_chain variants -- correctness (LZSS suite under the variant), then config 4's LZSS layer timed
cd "${GRAFT_REPO_ROOT:-$(pwd)}" || exit 1
mkdir -p gpurun_out
for env in "$@"; do
  echo "== $env"
  [ -n "$SKIPTESTS" ] || (env $env timeout 1500 python -m pytest tests/test_gpu_lzss.py tests/test_gpu_fuzz.py -x -q 2>&1 | tail -3)
  env $env timeout 300 python scripts/quick_cfg4_lzss.py 2>&1 | grep -E "config 4|match_chain|chain_tail |tok_emit|chain stats"
done
