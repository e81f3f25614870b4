"""Per-call latency of both codecs, device-resident, 4 KiB ... 8 MiB of the test suite's UTF-8 word text."""
import sys, os, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import torch
from raisin_amd import lz, huffman
from test_gpu_lzss import text
for n in (4096, 65536, 1 << 20, 8 << 20):
    d = torch.frombuffer(bytearray(text(5, n)), dtype=torch.uint8).cuda()
    for mod, name in ((lz, "lzss"), (huffman, "huffman")):
        c = mod.compress_tensor(d); o = mod.decompress_tensor(c)
        torch.cuda.synchronize()
        reps = 50
        t0 = time.perf_counter()
        for _ in range(reps): c = mod.compress_tensor(d)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        for _ in range(reps): o = mod.decompress_tensor(c)
        torch.cuda.synchronize(); t2 = time.perf_counter()
        print("%-8s %8d B: encode %8.1f us (%7.2f GB/s)  decode %8.1f us (%7.2f GB/s)" % (name, n, (t1 - t0) / reps * 1e6, n / ((t1 - t0) / reps) / 1e9, (t2 - t1) / reps * 1e6, n / ((t2 - t1) / reps) / 1e9))
