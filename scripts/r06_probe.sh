cd $GRAFT_REPO_ROOT
echo '--- tests'; timeout 900 python -m pytest tests/test_gpu_lzss_small.py tests/test_gpu_fuzz.py tests/test_gpu_abi_shim.py tests/test_gpu_engine.py -m gpu -x -q 2>&1 | tail -5
timeout 600 python - <<'PY'
import sys, time
sys.path.insert(0, '.')
from raisin_amd import _lib, lz
sam = open('tests/golden/samiam.txt','rb').read()
for n in (25, 256, 1024, 2048, 3461):
    d = sam[:n]
    c = lz.CompressAsync(d)
    te, td = [], []
    for _ in range(30):
        t0 = time.perf_counter(); lz.CompressAsync(d); te.append((time.perf_counter() - t0) * 1e3)
        t0 = time.perf_counter(); lz.Decompress(c); td.append((time.perf_counter() - t0) * 1e3)
    print(n, "encode ms", round(sorted(te)[15], 4), "decode ms", round(sorted(td)[15], 4))
PY
