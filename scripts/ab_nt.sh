#!/bin/bash
# A/B of the nontemporal-access masks (scripts/ab/librsn_nt<mask>.so, built by `make BUILD=build_nt<m> OUT=... EXTRA=-DRSN_NT_MASK=<m>`)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
for m in ${MASKS:-base 1 6 24 31}; do
  if [ $m = base ]; then unset RSN_LIB_PATH; else export RSN_LIB_PATH=$R/scripts/ab/librsn_nt$m.so; fi
  for rep in 1 2; do
    python3 bench.py --steps 30 --warmup 5 --no-cpu --no-others | python3 -c "
import json,sys
j=json.loads(sys.stdin.readline())
k=j['kernels']
print('mask $m: step %.4f ms  hist %.4f  emit %.4f  dec %.4f' % (j['ms_per_step'], k['huff_byte_hist']['ms'], k['huff_emit']['ms'], k['huff_dec_flat']['ms']))
"
  done
done
