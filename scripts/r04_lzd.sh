#!/bin/bash
# GPU box: LZSS decode of config 4's text layer -- per-kernel times and k_lzd_resolve's per-phase cycles
cd "${GRAFT_REPO_ROOT:-$(pwd)}" || exit 1
cat > /tmp/lzd.py <<'PY'
import sys, os, time
sys.path.insert(0, os.getcwd())
import torch, workloads as W
from raisin_amd import _lib, lz
n = 1024 << 20
src = W.config_input("4", n, "cuda")
c = lz.compress_tensor(src)
out = torch.empty(n + (1 << 20), dtype=torch.uint8, device="cuda")
d = lz.decompress_tensor(c, out=out)
torch.cuda.synchronize()
_lib.prof_enable(True); _lib.prof_reset()
t0 = time.perf_counter(); d = lz.decompress_tensor(c, out=out); torch.cuda.synchronize(); t1 = time.perf_counter()
print("config 4 lzss decode: %.2f ms ok=%s" % ((t1 - t0) * 1e3, bool(torch.equal(d, src))))
for k, (cnt, ms) in sorted(_lib.prof_get().items()):
    print("  %-22s %3d launches  %.3f ms" % (k, cnt, ms))
PY
for env in "X=1" "RSN_LZD_STATS=1" $@; do echo "== $env"; env $env timeout 300 python /tmp/lzd.py 2>&1 | grep -v "^$" | tail -14; done
