"""Mirror of /root/reference/engine (engine.go, util.go) for the codecs on the
accelerated path.  The engine is the CALLER of the hot path: layering, file
I/O, ratio / lossless / entropy reporting stay on the host exactly as the
reference does them; every byte of codec work goes through librsn."""
import io
import math
import os
import time
from dataclasses import dataclass

from . import huffman, lz

# engine.go:32 lists 11 engines; the MI355X build carries the two this path names
# (the others -- arithmetic, dmc, mcc, stdlib bindings -- are out of scope).
Engines = ["lzss", "huffman"]
Suites = {"all": list(Engines), "suite": list(Engines)}

# engine.go:48-58 / :101-111
Readers = {"lzss": lz.NewReader, "huffman": huffman.NewReader}
Writers = {"lzss": lz.NewWriter, "huffman": huffman.NewWriter}


class CompressedFile:
    """engine.go:39-45,60-139"""

    def __init__(self, CompressionEngine="", Compressed=b"", MaxSearchBufferLength=4096):
        self.CompressionEngine = CompressionEngine
        self.Compressed = bytes(Compressed)
        self.Decompressed = None
        self.pos = 0
        self.MaxSearchBufferLength = MaxSearchBufferLength  # never read by the reference either (engine.go:445)

    def Write(self, content):
        newWriter = Writers[self.CompressionEngine]        # unknown engine: KeyError (reference: nil type assertion panics)
        b = io.BytesIO()
        w = newWriter(b)
        w.Write(content)                                   # ONE write of the whole buffer (engine.go:133)
        w.Close()
        compressed = b.getvalue()
        self.Compressed += compressed
        return len(compressed)

    def Read(self, size):
        """engine.go:60-98: first call decompresses everything, then serves `size` bytes; returns (chunk, eof)."""
        if self.Decompressed is None:
            r = Readers[self.CompressionEngine](io.BytesIO(self.Compressed))
            self.Decompressed = r.Read(-1)
        chunk = self.Decompressed[self.pos:self.pos + size]
        eof = len(self.Decompressed) - self.pos <= size
        if not eof:
            self.pos += size
        return chunk, eof


def compress(content, algorithms):
    """engine.go:443-452: layers applied in order."""
    for algorithm in algorithms:
        f = CompressedFile(MaxSearchBufferLength=4096)
        f.CompressionEngine = algorithm
        f.Write(content)
        content = f.Compressed
    return content


def decompress(content, algorithms):
    """engine.go:454-479: layers undone in reverse order, read through a 512-byte buffer."""
    for algorithm in reversed(algorithms):
        f = CompressedFile(CompressionEngine=algorithm, Compressed=content)
        while True:
            _, eof = f.Read(512)
            if eof:
                break
        content = f.Decompressed
    return content


def CompressFile(algorithms, path, output):
    """engine.go:157-172"""
    data = open(path, "rb").read()
    print("Compressing...")
    out = compress(data, algorithms)
    open(output, "wb").write(out)
    print("Original bytes: %d" % len(data))
    print("Compressed bytes: %d" % len(out))
    print("Compression ratio: %.2f%%" % (len(out) / len(data) * 100 if data else float("nan")))
    return out


BATCH_BYTES = 4 << 30        # files are read and batched in groups of at most this many bytes: N x 1 GiB inputs never all sit in memory


def CompressFiles(algorithms, files, extension):
    """engine.go:150-154: one .rsn per input, file after file.  Several files under one Huffman layer go through
    rsn_huffman_compress_batch -- upload / encode / download overlapped on the device -- in GROUPS of at most BATCH_BYTES, in the loop's
    order and with the loop's semantics: a file's lines are printed and its .rsn is written before the next group is read, and a file
    the batch cannot take (empty: the reference panics in heap.Pop, huffman.go:102) or a group that fails falls back to the per-file
    loop, which stops at the failing file with every earlier .rsn already on disk -- exactly where the reference's loop would stop."""
    files = list(files)
    if len(files) > 1 and list(algorithms) == ["huffman"]:
        i = 0
        while i < len(files):
            group, datas, size = [], [], 0
            bad = False                                          # files[i] is empty, missing or unreadable: the per-file call below meets it in order
            while i < len(files):
                try:
                    fsize = os.path.getsize(files[i])
                    if fsize == 0 or (group and size + fsize > BATCH_BYTES):
                        bad = fsize == 0
                        break
                    d = open(files[i], "rb").read()
                except OSError:                                  # (a file that cannot be read ends the group -- the ones before it are
                    bad = True                                   #  compressed and written first, as the reference's loop would have, engine.go:150-154)
                    break
                group.append(files[i]); datas.append(d); size += len(d); i += 1
            outs = None
            if len(group) > 1:
                try:
                    outs = huffman.CompressBatch(datas)
                except Exception:                               # noqa: BLE001 -- the loop finds the file that fails, after writing the ones before it
                    outs = None
            if outs is None:
                for f in group:
                    CompressFile(algorithms, f, f + extension)
            else:
                for f, data, out in zip(group, datas, outs):
                    print("Compressing...")
                    open(f + extension, "wb").write(out)
                    print("Original bytes: %d" % len(data))
                    print("Compressed bytes: %d" % len(out))
                    print("Compression ratio: %.2f%%" % (len(out) / len(data) * 100 if data else float("nan")))
            if bad:
                CompressFile(algorithms, files[i], files[i] + extension)   # raises like the reference panics; the earlier files are written
                i += 1
        return
    for f in files:
        CompressFile(algorithms, f, f + extension)


def DecompressFile(algorithms, path, output):
    """engine.go:187-199"""
    data = open(path, "rb").read()
    print("Decompressing...")
    out = decompress(data, algorithms)
    with open(output, "wb") as fh:                                 # check(err) after WriteFile (engine.go:196-197): raises before any caller deletes the input
        fh.write(out)
    return out


def DecompressFiles(algorithms, files, extension):
    for f in files:
        path = f + extension
        if extension.strip() in ("", "."):
            path = os.path.splitext(f)[0]
        DecompressFile(algorithms, f, path)


@dataclass
class Result:
    """engine.go:201-209"""
    CompressionEngine: str
    TimeTaken: str
    Ratio: float
    ActualEntropy: float
    Entropy: float
    Lossless: bool
    Failed: bool


def _entropy(counts, total):
    """goent discrete.Entropy(p, math.Log): -sum p ln p (natural log, engine.go:410)."""
    h = 0.0
    for c in counts:
        if c:
            p = c / total
            h -= p * math.log(p)
    return h


def _byte_counts(b):
    import numpy as np
    return np.bincount(np.frombuffer(b, dtype=np.uint8), minlength=256).tolist() if b else []


def BenchmarkFile(algorithms, fileString, PrintStats=False):
    """engine.go:357-441: ONE timer around compress+decompress, ratio%, lossless by
    byte equality, and the entropy columns -- including the quirk at :412-423 that
    the 'compressed' entropy is computed over the DECOMPRESSED bytes divided by the
    compressed length."""
    data = open(fileString, "rb").read()
    name = ",".join(algorithms)
    entropy = _entropy(_byte_counts(data), len(data)) if data else 0.0
    start = time.perf_counter()
    compressed = compress(data, algorithms)
    decompressed = decompress(compressed, algorithms)
    dur = time.perf_counter() - start
    lossless = decompressed == data
    ratio = len(compressed) / len(data) * 100 if data else float("nan")
    actual = _entropy(_byte_counts(decompressed), len(compressed)) if compressed else 0.0
    res = Result(name, _time_taken(dur), ratio, actual, entropy, lossless, False)
    if PrintStats:
        print("Lossless: %s" % str(lossless).lower())
        print("Original bytes: %d" % len(data))
        print("Compressed bytes: %d" % len(compressed))
        if not lossless:
            print("Decompressed bytes: %d" % len(decompressed))
        print("Compression ratio: %.2f%%" % ratio)
        print("Time taken: %s" % res.TimeTaken)
    return res


def AsyncBenchmarkFile(algorithms, fileString):
    """engine.go:310-355: a panic in a codec becomes a 'failed' row."""
    try:
        return BenchmarkFile(algorithms, fileString)
    except Exception:  # noqa: BLE001 -- mirrors recover()
        return Result(",".join(algorithms), "failed", 0.0, 0.0, 0.0, False, True)


def _suite_order(results):
    """engine.go:266-276: lossless rows before lossy ones, each group by ascending ratio."""
    return sorted(results, key=lambda r: (not r.Lossless, r.Ratio))


BenchmarkTimeout = 60.0   # engine.go:216 `timeout := 1 * time.Minute`


def _go_duration_ns(ns):
    """time.Duration.String() (Go's published format: the largest unit that leaves a non-zero integer part below a
    second -- "190µs", "1.23ms" --, "XhYmZ.ZZZs" from a second up, fractions without trailing zeros, "0s" for zero)."""
    if ns == 0:
        return "0s"
    sign, u = ("-", -ns) if ns < 0 else ("", ns)

    def frac(v, prec):                       # fmtFrac: v / 10^prec, and the fraction's digits without trailing zeros
        q, r = divmod(v, 10 ** prec)
        digits = ("%0*d" % (prec, r)).rstrip("0") if prec else ""
        return q, ("." + digits) if digits else ""

    if u < 1000000000:
        if u < 1000:
            return "%s%dns" % (sign, u)
        prec, unit = (3, "\u00b5s") if u < 1000000 else (6, "ms")
        q, f = frac(u, prec)
        return "%s%d%s%s" % (sign, q, f, unit)
    sec, f = frac(u, 9)
    out = "%d%ss" % (sec % 60, f)
    mins = sec // 60
    if mins > 0:
        out = "%dm" % (mins % 60) + out
        if mins // 60 > 0:
            out = "%dh" % (mins // 60) + out
    return sign + out


def _go_round_ns(ns, m):
    """time.Duration.Round(m): to the nearest multiple of m, halfway values away from zero."""
    if m <= 0:
        return ns
    r = abs(ns) % m
    v = abs(ns) - r if r + r < m else abs(ns) + m - r
    return -v if ns < 0 else v


def _go_duration(seconds):
    """time.Duration.String() of a duration given in seconds (the timeout row, engine.go:258: ">1m0s")."""
    return _go_duration_ns(int(round(seconds * 1e9)))


def _time_taken(seconds):
    """engine.go:425: duration.Round(10*time.Microsecond).String() -- "190µs", "4.61s"."""
    return _go_duration_ns(_go_round_ns(int(round(seconds * 1e9)), 10000))


def BenchmarkSuite(files, algorithms, out=None, timeout=None):
    """engine.go:213-309 without the HTML report.  As in the reference, every algorithm entry of a
    file runs CONCURRENTLY on its own thread (one goroutine each, :235-244; librsn keeps one device
    context per calling thread), the suite waits at most `timeout` (1 minute, :216,246, util.go:15)
    and every entry that has not delivered by then gets a ">1m0s" DNF row (:256-263) while its
    thread is left running.  Per file one table: header, finished rows in the reference's order,
    failed rows as DNF, the File/Size footer.  Returns the flat list of results.
    An entry that misses the deadline keeps running on its daemon thread -- like the reference's goroutine -- so it goes on
    using the device (and holds its thread's scratch arena) while the next file is timed."""
    import sys
    import threading
    out = out or sys.stdout
    timeout = BenchmarkTimeout if timeout is None else timeout
    all_results = []
    for i, f in enumerate(files):
        print("Compressing file %d/%d - %s" % (i + 1, len(files), f), file=out)
        file_size = os.path.getsize(f)                          # ioutil.ReadFile + check(err) before anything starts (:223-225)
        slots, threads = {}, []
        for layer in algorithms:
            name = ",".join(layer)
            print("Benchmarking", name, file=out)
            box = []
            slots[name] = box                                   # resultChans[algorithmsString] (:240): a repeated entry shares its slot
            t = threading.Thread(target=lambda layer=layer, box=box, f=f: box.append(AsyncBenchmarkFile(layer, f)), daemon=True)
            t.start()
            threads.append(t)
        deadline = time.monotonic() + timeout                   # waitTimeout(&wg, timeout)
        for t in threads:
            t.join(max(0.0, deadline - time.monotonic()))
        done, failed = [], []
        for name, box in slots.items():
            if box:
                r = box[0]
                (failed if r.Failed else done).append(r)
            else:                                               # select default: nothing delivered in time
                failed.append(Result(name, ">" + _go_duration(timeout), 0.0, 0.0, 0.0, False, True))
        rows = [("engine", "time taken", "compression ratio", "actual entropy", "theoretical entropy", "lossless")]
        for r in _suite_order(done):
            rows.append((r.CompressionEngine, r.TimeTaken, "%.2f%%" % r.Ratio, "%.2f" % r.ActualEntropy, "%.2f" % r.Entropy, str(r.Lossless).lower()))
            all_results.append(r)
        for r in failed:
            rows.append((r.CompressionEngine, r.TimeTaken, "DNF", "DNF", "DNF", str(r.Lossless).lower()))
            all_results.append(r)
        rows.append(("File", f, "Size", ByteCountSI(file_size), "", ""))
        width = [max(len(row[c]) for row in rows) for c in range(6)]
        for row in rows:
            print(" | ".join(cell.ljust(w) for cell, w in zip(row, width)).rstrip(), file=out)
    return all_results


def parseAlgorithms(s):
    """cmd/cli.go:203-231: "a,[b,c]" -> [[a],[b,c]]"""
    algorithms, buf, layer, in_layer = [], "", [], False
    for ch in s:
        if ch == ",":
            if in_layer and buf:
                layer.append(buf)
            elif buf:
                algorithms.append([buf])
            buf = ""
        elif ch == "[":
            in_layer = True
        elif ch == "]":
            layer.append(buf)
            buf = ""
            in_layer = False
            algorithms.append(layer)
            layer = []
        else:
            buf += ch
    if buf:
        algorithms.append([buf])
    return algorithms


def ByteCountSI(b):
    """engine/util.go:30-42"""
    unit = 1000
    if b < unit:
        return "%d B" % b
    div, exp = unit, 0
    n = b // unit
    while n >= unit:
        div *= unit
        exp += 1
        n //= unit
    return "%.1f %cB" % (b / div, "kMGTPE"[exp])
