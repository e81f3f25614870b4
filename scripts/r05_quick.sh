#!/bin/bash
# GPU box (r05): config 4's LZSS layer with the walk's counters (a -DRSN_WALK_STATS build), then timed with the product build
cd "${GRAFT_REPO_ROOT:-$(pwd)}" || exit 1
[ -f scripts/ab/librsn_wstats.so ] && RSN_LIB_PATH=scripts/ab/librsn_wstats.so timeout 300 python scripts/quick_cfg4_lzss.py 2>&1 | grep -E "walk stats" | tail -2
timeout 300 python scripts/quick_cfg4_lzss.py 2>&1 | grep -E "config 4|match_chain|chain_tail |tok_emit"
if [ -n "$TESTS" ]; then timeout 1500 python -m pytest tests/test_gpu_lzss.py tests/test_gpu_fuzz.py -q 2>&1 | tail -8; fi
