#!/bin/bash
# GPU box (r05): the block-synchronous chain walk (k_match_walk) -- a small case first under a short timeout (a barrier that
# does not meet would hang the box), then the LZSS suites, then config 4's LZSS layer timed against the r04 walk
cd "${GRAFT_REPO_ROOT:-$(pwd)}" || exit 1
mkdir -p gpurun_out
echo "== smoke"
timeout 180 python -m pytest tests/test_gpu_lzss.py -x -q -k "fixtures or small_alphabet or text_and_period" 2>&1 | tail -5
[ "${PIPESTATUS[0]}" = "0" ] || { echo "smoke failed: stopping"; exit 1; }
if [ -z "$SKIPTESTS" ]; then
  echo "== suites"
  timeout 1500 python -m pytest tests/test_gpu_lzss.py tests/test_gpu_fuzz.py -x -q 2>&1 | tail -8
fi
echo "== config 4, new walk"
timeout 300 python scripts/quick_cfg4_lzss.py 2>&1 | grep -E "config 4|match_chain|chain_tail|tok_emit|esc_write|lzss_"
echo "== config 4, r04 walk"
RSN_LZSS_OLD_WALK=1 timeout 300 python scripts/quick_cfg4_lzss.py 2>&1 | grep -E "config 4|match_chain"
