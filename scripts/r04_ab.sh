#!/bin/bash
# GPU box: A/B of compile-time variants of librsn (scripts/ab/librsn_<tag>.so, built with `make BUILD=build_<tag> OUT=../../scripts/ab/librsn_<tag>.so EXTRA=-D...`):
# config 4's LZSS layer timed under each, the product build first.  usage: r04_ab.sh <tag>...
cd "${GRAFT_REPO_ROOT:-$(pwd)}" || exit 1
mkdir -p gpurun_out
for rep in 1 2; do
  echo "== product"; timeout 300 python scripts/quick_cfg4_lzss.py 2>&1 | grep -E "config 4|match_chain|chain stats"
  for tag in "$@"; do
    echo "== $tag"; RSN_LIB_PATH=$PWD/scripts/ab/librsn_$tag.so timeout 300 python scripts/quick_cfg4_lzss.py 2>&1 | grep -E "config 4|match_chain|chain stats|rror"
  done
done
