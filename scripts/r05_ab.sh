#!/bin/bash
# GPU box (r05): config 4's LZSS layer under A/B builds in scripts/ab/ (walk counters from the librsn_ws* builds, time from the others)
cd "${GRAFT_REPO_ROOT:-$(pwd)}" || exit 1
for lib in "$@"; do
  echo "== $lib"
  RSN_LIB_PATH=scripts/ab/librsn_$lib.so timeout 300 python scripts/quick_cfg4_lzss.py 2>&1 | grep -E "walk stats|config 4|match_chain" | tail -4
done
