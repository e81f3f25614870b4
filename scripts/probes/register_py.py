"""How long does hipHostRegister of 64 MiB take inside a Python process (the pipelined host decode pins its pieces in place)?"""
import ctypes, time, sys
import numpy as np
hip = ctypes.CDLL("/opt/rocm/lib/libamdhip64.so")
hip.hipHostRegister.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_uint]
hip.hipHostUnregister.argtypes = [ctypes.c_void_p]
n = 1 << 30
a = np.ones(n, dtype=np.uint8)
hip.hipSetDevice(0)
piece = 64 << 20
for rep in range(2):
    ts, tu = [], []
    for k in range(4):
        p = a.ctypes.data + k * piece
        p = (p + 4095) & ~4095
        t0 = time.perf_counter(); rc = hip.hipHostRegister(p, piece - 4096, 0); t1 = time.perf_counter()
        rc2 = hip.hipHostUnregister(p); t2 = time.perf_counter()
        ts.append((t1 - t0) * 1e3); tu.append((t2 - t1) * 1e3)
    print("register 64 MiB: %s ms (rc %d); unregister: %s ms (rc %d)" % (" ".join("%.2f" % t for t in ts), rc, " ".join("%.2f" % t for t in tu), rc2))
