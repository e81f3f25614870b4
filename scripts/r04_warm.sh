#!/bin/bash
# GPU box: the Huffman decoder's warm-up length (RSN_DEC_WARM) on `skewed` and config 4
cd "${GRAFT_REPO_ROOT:-$(pwd)}" || exit 1
run() {
  python bench.py --profile-only skewed,4 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1])['profile_only']
for k in ('skewed','4'):
    d=j[k]['kernels_decode_ms']; print(' ', k, 'decode', j[k]['decode_ms'], {x:d[x] for x in d if x.startswith('huff_dec')})"
}
for w in 128 96 64 32; do echo "== warm $w"; RSN_DEC_WARM=$w run; done
