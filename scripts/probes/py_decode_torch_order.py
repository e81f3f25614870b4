"""which part of torch's presence costs the pipelined decode its full duplex: the import, the initialised runtime, the order?"""
import sys, time, ctypes, os; sys.path.insert(0, ".")
import numpy as np
mode = sys.argv[1]
if mode in ("import", "init", "init_use"):
    import torch
    if mode in ("init", "init_use"): torch.cuda.init()
    if mode == "init_use": x = torch.ones(1 << 20, device="cuda:0"); torch.cuda.synchronize()
L = ctypes.CDLL("raisin_amd/librsn.so", mode=(os.RTLD_DEEPBIND | os.RTLD_NOW) if os.environ.get("DEEPBIND") else ctypes.DEFAULT_MODE)
for f in (L.rsn_huffman_compress, L.rsn_huffman_decompress): f.restype = ctypes.c_int
L.rsn_free.argtypes = [ctypes.c_void_p]
n = 1 << 30
src = np.random.default_rng(1).integers(0, 128, size=n, dtype=np.uint8)
def raw(fn, ptr, size):
    out = ctypes.POINTER(ctypes.c_uint8)(); got = ctypes.c_size_t(0)
    t0 = time.perf_counter()
    rc = fn(ptr, ctypes.c_size_t(size), ctypes.byref(out), ctypes.byref(got)); assert rc == 0, rc
    return out, got.value, (time.perf_counter() - t0) * 1e3
c_blk, c_n, t = raw(L.rsn_huffman_compress, src.ctypes.data_as(ctypes.c_char_p), src.size)
def run(label):
    ts = []
    for _ in range(4):
        d, dn, t = raw(L.rsn_huffman_decompress, ctypes.cast(c_blk, ctypes.c_char_p), c_n); ts.append(round(t, 1)); L.rsn_free(d)
    print("%-44s decode ms %s" % (label, ts), flush=True)
run("mode %s%s" % (mode, ", librsn loaded RTLD_DEEPBIND" if os.environ.get("DEEPBIND") else ""))
if mode == "init_use":
    import torch
    y = torch.arange(1 << 20, device="cuda:0", dtype=torch.float32).sum().item(); print("torch still computes:", y)
    # a torch tensor through librsn's device entry point
    L.rsn_huffman_compress_dev.restype = ctypes.c_int
    t_in = (torch.arange(1 << 24, device="cuda:0") % 97).to(torch.uint8); t_out = torch.empty((1 << 24) + (1 << 22), dtype=torch.uint8, device="cuda:0"); torch.cuda.synchronize()
    got = ctypes.c_size_t(0)
    rc = L.rsn_huffman_compress_dev(ctypes.c_void_p(t_in.data_ptr()), ctypes.c_size_t(t_in.numel()), ctypes.c_void_p(t_out.data_ptr()), ctypes.c_size_t(t_out.numel()), ctypes.byref(got), None)
    print("a torch tensor through rsn_huffman_compress_dev: rc", rc, "bytes", got.value, "first bytes", bytes(t_out[:12].cpu().numpy()))
if mode == "late":
    import torch; torch.cuda.init(); x = torch.ones(1 << 20, device="cuda:0"); torch.cuda.synchronize()
    run("... after torch initialised SECOND")
