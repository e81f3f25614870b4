import os
print("before numpy:", len(os.sched_getaffinity(0)), "cpus")
import numpy as np
print("after numpy :", len(os.sched_getaffinity(0)), "cpus")
import threading
out = []
t = threading.Thread(target=lambda: out.append(len(os.sched_getaffinity(0))))
t.start(); t.join()
print("a new thread:", out[0], "cpus")
print({k: v for k, v in os.environ.items() if "OMP" in k or "BLAS" in k or "MKL" in k or "GOMP" in k or "KMP" in k})
