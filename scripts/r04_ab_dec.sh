#!/bin/bash
# GPU box: A/B of compile-time variants of librsn on the general Huffman decoder (`skewed`) and config 4's decode.  usage: r04_ab_dec.sh <tag>...
cd "${GRAFT_REPO_ROOT:-$(pwd)}" || exit 1
run() {
  python bench.py --profile-only skewed,4 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1])['profile_only']
for k in ('skewed','4'):
    d=j[k]['kernels_decode_ms']; print(' ', k, 'decode', j[k]['decode_ms'], {x:d[x] for x in d if x.startswith('huff_dec')})"
}
for rep in 1 2; do
  echo "== product"; run
  for tag in "$@"; do echo "== $tag"; RSN_LIB_PATH=$PWD/scripts/ab/librsn_$tag.so run; done
done
