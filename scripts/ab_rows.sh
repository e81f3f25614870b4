#!/bin/bash
# k_match_chain: lanes per chain x build variant, Zipf text 1 GiB (A/B libraries built into scripts/ab/, see raisin_amd/csrc/Makefile)
#   usage: ab_rows.sh <variant> ...     LANES="64 16 8" selects the lane counts (64 only with the default library)
R=${GRAFT_REPO_ROOT:-$(pwd)}
run() { echo "== $1 lanes=$2"; RSN_LIB_PATH=$3 RSN_LZSS_CHAIN_LANES=$2 python3 $R/scripts/quick_lzss.py text 1024 2>&1 | grep "encode\|match_chain\|chain stats" | grep -v "^decode"; }
for L in ${LANES:-64 16 8}; do run default $L ""; done
for v in "$@"; do for L in ${LANES:-16 8}; do [ $L = 64 ] || run $v $L $R/scripts/ab/librsn_$v.so; done; done
