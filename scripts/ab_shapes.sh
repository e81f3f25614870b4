#!/bin/bash
# LZSS encode over the data shapes, a wavefront per chain against eight chains per wavefront (RSN_LZSS_CHAIN_LANES); runs on the GPU box
R=${GRAFT_REPO_ROOT:-$(pwd)}
for L in ${LANES:-64 8}; do
  echo "== lanes $L"
  export RSN_LZSS_CHAIN_LANES=$L
  python3 $R/scripts/lzss_shapes.py 64 2>&1 | grep -v amdgpu.ids
  python3 $R/scripts/lzss_mixed.py 2>&1 | grep -v amdgpu.ids | head -4
  for k in text text1; do python3 $R/scripts/quick_lzss.py $k 1024 2>&1 | grep "encode\|match_chain" | grep -v "^decode"; done
  python3 $R/scripts/quick_lzss.py period 1024 2>&1 | grep "encode" | grep -v "^decode"
  python3 $R/scripts/lzss_random.py 2>&1 | grep -v amdgpu.ids | head -3
done
