#!/bin/bash
# GPU box: k_match_chain variants -- correctness (LZSS suite), then config 4's LZSS layer timed, with the walk's own counters
cd "${GRAFT_REPO_ROOT:-$(pwd)}" || exit 1
mkdir -p gpurun_out
[ -n "$SKIPTESTS" ] || (timeout 1500 python -m pytest tests/test_gpu_lzss.py -x -q 2>&1 | tail -5)
echo "== default build"; timeout 300 python scripts/quick_cfg4_lzss.py 2>&1 | grep -E "config 4|match_chain|chain_tail |tok_emit"
for L in $@; do
  echo "== $L"; RSN_LIB_PATH=scripts/ab/librsn_$L.so timeout 300 python scripts/quick_cfg4_lzss.py 2>&1 | grep -E "config 4|match_chain|chain stats" | tail -5
done
