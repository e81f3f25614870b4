#!/bin/bash
# GPU box: A/B of compile-time variants of librsn on the LZSS decoder (config 4's text, config 3).  usage: r04_ab_lzd.sh <tag>...
cd "${GRAFT_REPO_ROOT:-$(pwd)}" || exit 1
run() {
  python bench.py --profile-only 4,3 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1])['profile_only']
for k in ('4','3'):
    d=j[k]['kernels_decode_ms']; print(' ', k, 'decode', j[k]['decode_ms'], {x:d[x] for x in d if x.startswith('lzss_')})"
}
for rep in 1 2; do
  echo "== product"; run
  for tag in "$@"; do echo "== $tag"; RSN_LIB_PATH=$PWD/scripts/ab/librsn_$tag.so run; done
done
