"""the four host-buffer calls on data shaped like what people compress: tables, logs, records of a fixed size, bitmaps, DNA, base64, JSON,
word lists, sparse buffers, runs.  16 MiB each (argv[1] = MiB), median of 3 warm calls, GB/s of input; a kind whose LZSS encode falls
below a tenth of text's is what to look at.  The round trip is checked; the first 96 KiB also against the oracle's bytes."""
import base64
import random
import sys
import time

sys.path.insert(0, ".")
import numpy as np

from oracle import oracle
from raisin_amd import huffman, lz

N = (int(sys.argv[1]) if len(sys.argv) > 1 else 16) << 20
only = sys.argv[2] if len(sys.argv) > 2 else None
dev = len(sys.argv) > 3 and sys.argv[3] == "dev"          # device tensors in and out (no PCIe): what the kernels make of the shape
trace = len(sys.argv) > 3 and sys.argv[3] == "trace"      # with RSN_DEBUG=1 in the environment: the encoder's stages and per-kernel times of one LZSS call
rng = random.Random(11)
nrng = np.random.default_rng(11)
words = ["".join(rng.choice("etaoinshrdlucmfwypvbgkqjxz") for _ in range(rng.randint(1, 9))) for _ in range(3000)]


def text(n):
    out, size = [], 0
    while size < n:
        w = rng.choice(words) + " "; out.append(w); size += len(w)
    return "".join(out).encode()[:n]


def lines(make, n):
    out, size, i = [], 0, 0
    while size < n:
        s = make(i); out.append(s); size += len(s); i += 1
    return "".join(out).encode()[:n]


def csv(n):
    return lines(lambda i: "%d,%0.2f,%s,2026-10-%02dT%02d:%02d:%02d,%d\n" % (100000 + i, rng.random() * 1000, rng.choice(words), 1 + i // 86400 % 28, i // 3600 % 24, i // 60 % 60, i % 60, rng.randint(0, 9)), n)


def log(n):
    tmpl = ["GET /api/v1/items/%d HTTP/1.1 200 %d", "POST /api/v1/login HTTP/1.1 401 %d %d", "connection from 10.0.%d.%d closed", "worker %d finished job %d in 12 ms"]
    return lines(lambda i: "2026-10-04T10:%02d:%02d.%03d INFO " % (i // 60000 % 60, i // 1000 % 60, i % 1000) + rng.choice(tmpl) % (rng.randint(0, 255), rng.randint(0, 99999)) + "\n", n)


def records(stride, n_random, counter=True):
    def gen(n):
        k = n // stride + 1
        a = np.zeros((k, stride), dtype=np.uint8)
        a[:] = np.frombuffer(bytes(rng.randrange(32, 127) for _ in range(stride)), dtype=np.uint8)
        if counter:
            a[:, :4] = np.arange(k, dtype="<u4").view(np.uint8).reshape(k, 4)
        if n_random:
            a[:, 8:8 + n_random] = nrng.integers(0, 256, size=(k, n_random), dtype=np.uint8)
        return a.tobytes()[:n]
    return gen


def bitmap(n, row=1024, change=0.01):
    k = n // row + 1
    base = nrng.integers(0, 128, size=row, dtype=np.uint8)
    out = np.empty((k, row), dtype=np.uint8)
    cur = base.copy()
    for r in range(k):
        idx = nrng.integers(0, row, size=max(1, int(row * change)))
        cur[idx] = nrng.integers(0, 128, size=idx.size, dtype=np.uint8)
        out[r] = cur
    return out.tobytes()[:n]


def json_like(n):
    return lines(lambda i: '{"id": %d, "name": "%s", "tags": ["%s", "%s"], "score": %0.3f, "active": %s},\n' % (i, rng.choice(words), rng.choice(words), rng.choice(words), rng.random(), rng.choice(["true", "false"])), n)


def sparse(n, every=100):
    a = np.zeros(n, dtype=np.uint8)
    idx = nrng.integers(0, n, size=n // every)
    a[idx] = nrng.integers(1, 128, size=idx.size, dtype=np.uint8)
    return a.tobytes()


def runs(n, longest=1000):
    out, size = [], 0
    while size < n:
        k = rng.randint(1, longest); out.append(bytes([rng.randrange(32, 127)]) * k); size += k
    return b"".join(out)[:n]


def edited_document(n, doc=100000, edits=20):
    d = bytearray(text(doc))
    out, size = [], 0
    while size < n:
        for _ in range(edits):
            at = rng.randrange(len(d) - 10); d[at:at + rng.randint(1, 8)] = rng.choice(words).encode()
        out.append(bytes(d)); size += len(d)
    return b"".join(out)[:n]


kinds = {
    "text": text,
    "csv": csv,
    "log lines": log,
    "json": json_like,
    "records 64 B, counter + 8 random": records(64, 8),
    "records 16 B, counter + 4 random": records(16, 4),
    "records 256 B, counter only": records(256, 0),
    "records 100 B, counter + 20 random": records(100, 20),
    "records 4096 B, counter only": records(4096, 0),
    "records 4100 B, counter only": records(4100, 0),
    "bitmap rows 1024, 1% change": bitmap,
    "bitmap rows 4096, 1% change": lambda n: bitmap(n, 4096),
    "bitmap rows 3000, 0.1% change": lambda n: bitmap(n, 3000, 0.001),
    "DNA": lambda n: nrng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=n).tobytes(),
    "base64": lambda n: base64.b64encode(nrng.integers(0, 256, size=n, dtype=np.uint8).tobytes())[:n],
    "hex": lambda n: nrng.integers(0, 256, size=n // 2 + 1, dtype=np.uint8).tobytes().hex().encode()[:n],
    "bits as 0 and 1": lambda n: nrng.choice(np.frombuffer(b"01", dtype=np.uint8), size=n).tobytes(),
    "word list, sorted": lambda n: "\n".join(sorted(rng.choice(words) + rng.choice(words) for _ in range(n // 8))).encode()[:n],
    "sparse: a byte in 100": sparse,
    "sparse: a byte in 3000": lambda n: sparse(n, 3000),
    "runs up to 1000": runs,
    "runs up to 20": lambda n: runs(n, 20),
    "a document edited and repeated": edited_document,
    "csv, nine rows in ten the same": lambda n: lines(lambda i: "0,0,0,0.0,\n" if rng.random() < 0.9 else "%d,%d,%d,%0.1f,x\n" % (rng.randint(0, 99), rng.randint(0, 9), i % 7, rng.random()), n),
    "csv, 99 rows in 100 the same": lambda n: lines(lambda i: "0,0,0,0.0,\n" if rng.random() < 0.99 else "%d,%d,%d,%0.1f,x\n" % (rng.randint(0, 99), rng.randint(0, 9), i % 7, rng.random()), n),
    "a log line repeated 1-300 times": lambda n: lines(lambda i: (rng.choice(["worker idle\n", "heartbeat ok 200\n", "retrying connection to 10.0.0.%d\n" % rng.randint(1, 9)])) * rng.randint(1, 300), n),
    "utf-16 text": lambda n: text(n // 2).decode().encode("utf-16-le")[:n],
    "32-bit integers, small": lambda n: nrng.integers(0, 1000, size=n // 4 + 1, dtype="<u4").tobytes()[:n],
    "32-bit floats near 1": lambda n: (1.0 + nrng.random(n // 4 + 1, dtype=np.float32) * 1e-3).astype("<f4").tobytes()[:n],
    "html-like, many <": lambda n: lines(lambda i: "<tr><td>%d</td><td>%s</td></tr>\n" % (i, rng.choice(words)), n),
}


def med(fn, *a):
    fn(*a)
    ts = []
    for _ in range(3):
        t0 = time.perf_counter(); r = fn(*a); ts.append(time.perf_counter() - t0)
    return r, sorted(ts)[1]


print("%-36s | %8s %8s %6s | %8s %8s %6s   (GB/s of input, %% of input)" % ("kind, %d MiB" % (N >> 20), "huff enc", "huff dec", "ratio", "lzss enc", "lzss dec", "ratio"), flush=True)
for name, gen in kinds.items():
    if only and only not in name:
        continue
    data = gen(N)
    n = len(data)
    if dev:
        import torch
        src = torch.frombuffer(bytearray(data), dtype=torch.uint8).cuda()

        def tmed(fn, *a):
            fn(*a); torch.cuda.synchronize()
            ts = []
            for _ in range(3):
                t0 = time.perf_counter(); r = fn(*a); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
            return r, sorted(ts)[1]
        c, he = tmed(huffman.compress_tensor, src)
        d, hd = tmed(huffman.decompress_tensor, c)
        z, le = tmed(lz.compress_tensor, src)
        u, ld = tmed(lz.decompress_tensor, z)
        note = "" if torch.equal(u, src) else " LZSS MISMATCH"
        print("%-36s | %8.2f %8.2f %5.1f%% | %8.2f %8.2f %5.1f%%%s" % (name, n / he / 1e9, n / hd / 1e9, 100.0 * c.numel() / n, n / le / 1e9, n / ld / 1e9, 100.0 * z.numel() / n, note), flush=True)
        continue
    if trace:
        from raisin_amd import _lib
        lz.CompressAsync(data)
        sys.stderr.write("=== %s, %d bytes\n" % (name, n)); sys.stderr.flush()
        _lib.prof_enable(True); _lib.prof_reset()
        t0 = time.perf_counter(); z = lz.CompressAsync(data); t = time.perf_counter() - t0
        pe = _lib.prof_get(); _lib.prof_enable(False)
        top = sorted(((v[1], k, v[0]) for k, v in pe.items()), reverse=True)[:8]
        sys.stderr.write("%.2f ms, %d -> %d B; %s\n" % (t * 1e3, n, len(z), ", ".join("%s x%d %.2f" % (k, m, ms) for ms, k, m in top))); sys.stderr.flush()
        continue
    c, he = med(huffman.Compress, data)
    d, hd = med(huffman.Decompress, c)
    note = "" if d == oracle.huffman_decompress(c) else " HUFFMAN MISMATCH"   # (lossy where the reference is: bytes that are not UTF-8)
    z, le = med(lz.CompressAsync, data)
    d, ld = med(lz.Decompress, z)
    if d != data:
        note += " LZSS MISMATCH"
    head = data[:96 << 10]
    if lz.CompressAsync(head) != oracle.lzss_compress(head) or huffman.Compress(head) != oracle.huffman_compress(head):
        note += " NOT THE ORACLE'S BYTES"
    print("%-36s | %8.2f %8.2f %5.1f%% | %8.2f %8.2f %5.1f%%%s" % (name, n / he / 1e9, n / hd / 1e9, 100.0 * len(c) / n, n / le / 1e9, n / ld / 1e9, 100.0 * len(z) / n, note), flush=True)
