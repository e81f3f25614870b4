import sys; sys.path.insert(0, ".")
import torch, time
import workloads as W
from raisin_amd import lz, _lib
n = 1 << 30
for cfg in ("3", "4"):
    src = W.config_input(cfg, n, "cuda:0")
    out = torch.empty(n + n // 8 + (1 << 20), dtype=torch.uint8, device="cuda:0")
    ts = []
    for rep in range(6):
        torch.cuda.synchronize(); t = time.perf_counter()
        c = lz.compress_tensor(src, out=out)
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t) * 1e3)
    _lib.prof_enable(True); _lib.prof_reset()
    c = lz.compress_tensor(src, out=out)
    p = _lib.prof_get(); _lib.prof_enable(False)
    print(cfg, "encode ms:", [round(x, 3) for x in ts], "kernels:", {k: round(v[1], 3) for k, v in p.items() if v[0] and v[1] > 0.01})
