#!/bin/bash
# GPU box: Huffman decode tuning sweep (skewed / raw text)
cd "${GRAFT_REPO_ROOT:-$(pwd)}" || exit 1
for k in skewed 4; do
  for env in "X=1" "RSN_DEC_WARM=64" "RSN_DEC_WARM=96" "RSN_DEC_K=10" "RSN_DEC_K=10 RSN_DEC_WARM=96" "RSN_DEC_K=9"; do
    echo "== $k $env"; env $env timeout 300 python scripts/quick_huff.py $k 1024 2>&1 | grep -A5 " decode " | grep -v scan
  done
done
