#!/usr/bin/env python3
"""bench.py -- headline benchmark: BASELINE.json configs[1] (config 5's per-GPU chunk when --gpus > 1).

  python bench.py --gpus N --steps K --warmup W          (N > 1 without a launcher: bench.py starts its N ranks itself)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Timed workload (per GPU): `-algorithm=huffman` on a 1 GiB uniform-random buffer (config "2a": bytes uniform
over 0x00..0x7F so that the reference's rune-level Huffman is lossless; splitmix64 seed 0x5EED0002, rank r
uses 0x5EED0050+r when N > 1 -- workloads.py).  A step is what the reference's BenchmarkFile times
(engine/engine.go:379-406): compress, then decompress, of one buffer, input already resident in HBM.
value = uncompressed MB (1e6 B) through encode+decode per second, whole job.

After the timed region (N = 1 only) every other BASELINE config runs at the same size and is reported under
`other_configs`: 2b (uniform 0x00..0xFF: the rune path, lossy exactly like the reference), `skewed` (unequal code
lengths: the general Huffman kernels), 3 (`lzss`, 4096-periodic), 4 (`lzss,huffman` on Zipf text) -- each with
encode/decode ms, ratio, lossless, a bit-exact check against the oracle on a sample, the dominant kernel, the
algorithmic-byte fractions of the HBM peak and its own threaded CPU baseline.

Kernel timings come from HIP events recorded by librsn on its own launch stream (rsn_prof_*), live inside the
timed region.  The CPU baseline is oracle/cpu_baseline.c -- the C restatement of the reference (the reference is
Go and cannot run here: kind "port") on all host cores, on a bounded sample, on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (guides/MI355X_MICROARCH.md)


# librsn's profiling names of the headline (flat-code) kernels -> the kernel's name in a rocprofv3 trace
TRACE_NAME = {"huff_emit": "k_emit_flat", "huff_dec_flat": "k_dec_flat", "huff_byte_hist": "k_byte_hist"}


def load_traffic(prof_name):
    """HBM bytes per launch of one headline kernel from the newest committed PMC summary of the HEADLINE workload
    (profiles/<tag>_pmc_headline.json: scripts/profile.sh <tag> headline + summarize_prof.py -- FETCH_SIZE doubled per the gfx950
    note + WRITE_SIZE, separate passes, the kernel's largest launch): a constant of that profile, not a measurement of this run;
    the file's name is reported next to it."""
    import glob
    want = TRACE_NAME.get(prof_name)
    best = None
    for p in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_headline.json"))):
        try:
            d = json.load(open(p))
        except Exception:
            continue
        for k, v in d.items():
            if want and k.startswith(want) and isinstance(v, dict) and v.get("hbm_max"):
                best = (v["hbm_max"], os.path.basename(p))
    if best is None:                                     # the older summaries (r01/r02: one mean per kernel name)
        for p in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc.json"))):
            try:
                d = json.load(open(p))
            except Exception:
                continue
            for k, v in d.items():
                if want and k.startswith(want):
                    best = (v, os.path.basename(p))
    return best


# ---------------------------------------------------------------------------------------------- CPU baselines
def _timed(fn):
    t0 = time.perf_counter()
    r = fn()
    return r, time.perf_counter() - t0


def _grow(run, sizes_mib, target_s):
    """Runs `run(mib)` on growing samples until one takes long enough to be a measurement (or the list ends)."""
    res = None
    for mib in sizes_mib:
        res = run(mib)
        if res["seconds"] >= target_s / 4:
            break
    return res


def cpu_huffman(make_sample, cores, target_s=10.0):
    """Threaded oracle: encode + decode of one buffer (engine.go:379-406 times both together)."""
    from oracle import oracle as O

    def run(mib):
        s = make_sample(mib << 20)
        c, te = _timed(lambda: O.huffman_compress_mt(s, cores))
        d, td = _timed(lambda: O.huffman_decompress_mt(c, cores))
        return {"value": round(len(s) / 1e6 / (te + td), 2), "unit": "MB/s", "cores": cores, "kind": "port",
                "sample": "%d MiB prefix of the same buffer: oracle/cpu_baseline.c on %d threads, encode %.2f s + decode %.2f s"
                          % (mib, cores, te, td),
                "encode_MBps": round(len(s) / 1e6 / te, 2), "decode_MBps": round(len(s) / 1e6 / td, 2),
                "seconds": te + td, "_c": c, "_s": s}
    return _grow(run, (32, 128, 512), target_s)


def cpu_lzss(make_sample, cores, window=4096, target_s=10.0, sizes=(2, 8, 32, 128)):
    """Threaded oracle in the reference's own shape (a Reference for every position in parallel, serial compaction,
    lzss.go:117-151), serial decode (lzss.go:323-364); plus the lazy single-thread form for context."""
    from oracle import oracle as O

    def run(mib):
        s = make_sample(mib << 20)
        c, te = _timed(lambda: O.lzss_compress_mt(s, window, cores, 4096))
        d, td = _timed(lambda: O.lzss_decompress(c))
        return {"value": round(len(s) / 1e6 / (te + td), 2), "unit": "MB/s", "cores": cores, "kind": "port",
                "sample": "%d MiB prefix of the same buffer: every-position match table on %d threads + serial compaction "
                          "%.2f s, serial decode %.2f s" % (mib, cores, te, td),
                "encode_MBps": round(len(s) / 1e6 / te, 2), "decode_MBps": round(len(s) / 1e6 / td, 2),
                "seconds": te + td, "_c": c, "_s": s}
    return _grow(run, sizes, target_s)


def cpu_lzss_per_position(make_sample, cores, mib=1):
    """The reference's goroutine-per-byte shape taken literally: one task per position (lzss.go:117-130)."""
    from oracle import oracle as O
    s = make_sample(mib << 20)
    _, t = _timed(lambda: O.lzss_compress_mt(s, 4096, cores, 1))
    _, tl = _timed(lambda: O.lzss_compress(s, 4096))
    return {"one_task_per_position_MBps": round(len(s) / 1e6 / t, 2), "lazy_single_thread_MBps": round(len(s) / 1e6 / tl, 2),
            "sample_MiB": mib, "cores": cores}


def _public(d):
    return {k: v for k, v in d.items() if not k.startswith("_") and k != "seconds"}


# ---------------------------------------------------------------------------------------------- other configs
ALG_NOTE = "algorithmic HBM bytes (SURVEY.md 8d): huffman encode 2N+C, decode C+N_out; lzss encode N+C, decode C+N"


def run_other_configs(torch, device, n, cores, with_cpu, names):
    import workloads as W
    from oracle import oracle as O
    from raisin_amd import _lib, huffman, lz
    out = {}

    def gpu_bytes(t):
        return bytes(t.cpu().numpy())

    def one(name):
        src = W.config_input("4" if name == "4" else name, n, device)
        layers = {"2b": ["huffman"], "skewed": ["huffman"], "3": ["lzss"], "4": ["lzss", "huffman"]}[name]
        enc = {"huffman": huffman.compress_tensor, "lzss": lz.compress_tensor}
        dec = {"huffman": huffman.decompress_tensor, "lzss": lz.decompress_tensor}

        ebuf, dbuf = {}, {}                                 # output buffers, sized by the warm-up pass and reused by the timed passes

        def compress(x, bufs=None):                         # engine.go:443-452: layers in order
            sizes = []
            for i, a in enumerate(layers):
                x = enc[a](x, out=bufs.get(i) if bufs else None)
                sizes.append(int(x.numel()))
            return x, sizes

        def decompress(x, bufs=None):                       # engine.go:454-479: layers in reverse
            outs = []
            for i, a in enumerate(reversed(layers)):
                x = dec[a](x, out=bufs.get(i) if bufs else None)
                outs.append(int(x.numel()))
            return x, outs

        c, wsz = compress(src)                              # warm-up: scratch arenas grow here, not in the timed pass
        d, dsz = decompress(c)
        del c, d
        for i, m in enumerate(wsz):
            ebuf[i] = torch.empty(m + (1 << 16), dtype=torch.uint8, device=device)
        for i, m in enumerate(dsz):
            dbuf[i] = torch.empty(m + (1 << 16), dtype=torch.uint8, device=device)
        torch.cuda.synchronize(device)
        _lib.prof_enable(True)
        reps = 2
        te = td = 0.0
        prof_e, prof_d = {}, {}
        for _ in range(reps):
            _lib.prof_reset()
            (c, sizes), t = _timed(lambda: compress(src, ebuf))   # the C ABI calls return after their stream has been synchronised
            te += t
            prof_e = _lib.prof_get()
            _lib.prof_reset()
            (d, _), t = _timed(lambda: decompress(c, dbuf))
            td += t
            prof_d = _lib.prof_get()
        _lib.prof_enable(False)
        te, td = te / reps * 1e3, td / reps * 1e3
        C, n_out = int(c.numel()), int(d.numel())
        lossless = bool(n_out == n and torch.equal(d, src))
        # algorithmic bytes of the whole call(s)
        if name == "4":
            l1 = sizes[0]
            alg_e, alg_d = (n + l1) + (2 * l1 + C), (C + l1) + (l1 + n)
        elif layers == ["lzss"]:
            alg_e, alg_d = n + C, C + n
        else:
            alg_e, alg_d = 2 * n + C, C + n_out
        ent = {
            "algorithm": ",".join(layers), "bytes": n, "encode_ms": round(te, 3), "decode_ms": round(td, 3),
            "round_trip_MBps": round(n / 1e6 / ((te + td) / 1e3), 1), "ratio_pct": round(100.0 * C / n, 3), "lossless": lossless,
            "decoded_bytes": n_out,
            "encode_frac_of_hbm_peak": round(alg_e / (te / 1e3) / 1e9 / HBM_PEAK_GBPS, 5),
            "decode_frac_of_hbm_peak": round(alg_d / (td / 1e3) / 1e9 / HBM_PEAK_GBPS, 5),
            "kernels_encode_ms": {k: round(v[1], 3) for k, v in sorted(prof_e.items())},
            "kernels_decode_ms": {k: round(v[1], 3) for k, v in sorted(prof_d.items())},
        }
        if name == "4":
            ent["layer_sizes"] = sizes
        allk = {**{k: v[1] for k, v in prof_e.items()}, **{k: v[1] for k, v in prof_d.items()}}
        dom = max(allk, key=allk.get)
        kalg = {"huff_byte_hist": n, "huff_rune_hist": n, "huff_tile_bits_rune": n, "huff_emit": n + C, "huff_emit_rune": n + C,
                "huff_dec_sync": C, "huff_dec_emit": C + n_out, "huff_dec_flat": C + n_out}
        if name in ("3", "4"):
            kalg.update({"lzss_match_chain": n, "lzss_match_hash": n, "lzss_match": n})   # reads the (escaped) stream once
        ent["dominant_kernel"] = {"name": dom, "ms": round(allk[dom], 3)}
        if dom in kalg and name != "4":
            ent["dominant_kernel"]["algorithmic_bytes"] = kalg[dom]
            ent["dominant_kernel"]["frac_of_hbm_peak"] = round(kalg[dom] / (allk[dom] / 1e3) / 1e9 / HBM_PEAK_GBPS, 5)
        # bit-exact against the oracle on a sample (prefix of the same buffer), through the same layers
        smp = {"2b": 32 << 20, "skewed": 32 << 20, "3": 2 << 20, "4": 8 << 20}[name]
        smp = min(smp, n)
        pre = src[:smp].contiguous()
        got, _ = compress(pre)
        ref = gpu_bytes(pre)
        for a in layers:
            ref = O.huffman_compress(ref) if a == "huffman" else O.lzss_compress(ref, 4096)
        ent["bit_exact_vs_oracle_on_sample"] = bool(gpu_bytes(got) == ref)
        ent["oracle_sample_MiB"] = smp >> 20
        if with_cpu:
            def make(k, src=src):
                return gpu_bytes(src[:min(k, n)])
            if layers == ["huffman"]:
                ent["cpu_baseline"] = _public(cpu_huffman(make, cores, target_s=6.0))
            elif layers == ["lzss"]:
                ent["cpu_baseline"] = _public(cpu_lzss(make, cores, sizes=(1, 4, 16), target_s=6.0))
            else:                                          # layered: lzss on the sample, huffman on ITS output, both timed
                r1 = cpu_lzss(make, cores, target_s=6.0)
                l1c = r1["_c"]
                c2, t2e = _timed(lambda: O.huffman_compress_mt(l1c, cores))
                _, t2d = _timed(lambda: O.huffman_decompress_mt(c2, cores))
                tot = r1["seconds"] + t2e + t2d
                ent["cpu_baseline"] = {"value": round(len(r1["_s"]) / 1e6 / tot, 2), "unit": "MB/s", "cores": cores, "kind": "port",
                                       "sample": r1["sample"] + "; huffman layer on its output: encode %.2f s + decode %.2f s" % (t2e, t2d),
                                       "lzss_layer_MBps": r1["value"]}
                ent["cpu_lzss_shape"] = cpu_lzss_per_position(make, cores)
        out[name] = ent
        del src, c, d, ebuf, dbuf
        torch.cuda.empty_cache()

    def chunks_on_one_gpu(k=8):
        """Config 5 as one GPU sees it: k independent chunks (seeds 0x5EED0050 + i), one complete .rsn segment each, encoded one
        after the other (engine.CompressFiles' loop, engine.go:150-154); the multi-GPU form is `bench.py --gpus N`.
        Inputs and outputs are ONE allocation each, touched by two warm-up passes (r02: 16 separate 1 GiB blocks from torch's
        caching allocator gave 5.6 ms on one box and 7.9 on the driver's); every chunk's time is reported."""
        cap = n + n // 8 + (1 << 20)
        cap = (cap + 255) & ~255
        src_all = torch.empty(k * n, dtype=torch.uint8, device=device)
        out_all = torch.empty(k * cap, dtype=torch.uint8, device=device)
        srcs = [src_all[i * n:(i + 1) * n] for i in range(k)]
        outs = [out_all[i * cap:(i + 1) * cap] for i in range(k)]
        for i in range(k):
            srcs[i].copy_(W.config_input("5", n, device, chunk=i))
        for _ in range(2):
            for i in range(k):
                huffman.compress_tensor(srcs[i], out=outs[i])
        torch.cuda.synchronize(device)
        reps, te = 3, 0.0
        per_chunk = [0.0] * k
        passes = []
        _lib.prof_enable(True)
        _lib.prof_reset()
        for _ in range(reps):
            t_pass = 0.0
            segs = []
            for i in range(k):
                seg, t = _timed(lambda i=i: huffman.compress_tensor(srcs[i], out=outs[i]))
                segs.append(seg)
                per_chunk[i] += t
                t_pass += t
            passes.append(t_pass * 1e3)
            te += t_pass
        prof5 = _lib.prof_get()
        _lib.prof_enable(False)
        te = te / reps * 1e3
        dec = torch.empty(n + (1 << 20), dtype=torch.uint8, device=device)
        ok = all(bool(torch.equal(huffman.decompress_tensor(segs[i], out=dec), srcs[i])) for i in range(k))   # every segment decodes on its own
        C = sum(int(x.numel()) for x in segs)
        out["5"] = {"algorithm": "huffman", "chunks": k, "bytes": k * n, "encode_ms": round(te, 3), "encode_MBps": round(k * n / 1e6 / (te / 1e3), 1),
                    "ratio_pct": round(100.0 * C / (k * n), 3), "lossless": ok,
                    "encode_frac_of_hbm_peak": round((2 * k * n + C) / (te / 1e3) / 1e9 / HBM_PEAK_GBPS, 5),
                    "per_chunk_ms": [round(x / reps * 1e3, 4) for x in per_chunk], "pass_ms": [round(x, 3) for x in passes],
                    "kernels_ms_per_chunk": {kk: round(v[1] / (reps * k), 4) for kk, v in sorted(prof5.items())},
                    "note": "one GPU, chunks one after the other, wall time per call; one allocation for the 8 inputs and one for the 8 outputs, "
                            "two warm-up passes; the sharded form with its gather is the --gpus N run"}
        del srcs, outs, segs, dec, src_all, out_all
        torch.cuda.empty_cache()

    for name in names:
        try:
            if name == "5":
                chunks_on_one_gpu()
            else:
                one(name)
        except Exception as e:          # noqa: BLE001 -- reported in the line, the other configs still run
            out[name] = {"error": "%s: %s" % (type(e).__name__, e)}
            torch.cuda.empty_cache()
    return out


def self_launch(n_ranks):
    """`python bench.py --gpus N` without a launcher: this process becomes the launcher's parent -- it starts
    `python -m torch.distributed.run --nproc-per-node N bench.py <same flags>` as a CHILD (one rank per GPU over RCCL), lets the
    child's stdout (rank 0's JSON line) through and returns its exit code.  Nothing here imports torch or touches the GPU: a process
    that has initialised the GPU must never be replaced or re-exec'ed on this pool."""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: what RCCL needs on this host driver
    env.setdefault("OMP_NUM_THREADS", "8")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_ranks),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--mib", type=int, default=1024, help="buffer size per GPU (default: the 1 GiB of BASELINE.json)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baselines")
    ap.add_argument("--no-others", action="store_true", help="skip the other BASELINE configs after the timed region")
    ap.add_argument("--others", default="2b,skewed,3,4,5")
    ap.add_argument("--profile-only", default="", help="run ONLY these other configs (no headline step, no CPU baselines) and print their entries: "
                    "what scripts/profile.sh puts under rocprofv3, so that a workload's counters hold that workload's launches and nothing else")
    ap.add_argument("--dist-backend", default="nccl", help="nccl (= RCCL, one GPU per rank) or gloo (control-flow test: all ranks share GPU 0)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args.gpus))            # plain `python bench.py --gpus N`: start the N ranks ourselves (nothing has touched the GPU yet)

    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        print("bench.py: --gpus %d but the launcher started %d rank(s) (WORLD_SIZE)" % (args.gpus, world), file=sys.stderr)
        sys.exit(2)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if args.dist_backend == "gloo":
            local_rank = 0                       # test mode: every rank drives GPU 0, collectives run on host tensors
            dist.init_process_group("gloo")
        else:
            if local_rank >= torch.cuda.device_count():
                print("bench.py: rank %d has no GPU of its own (%d visible): one rank per GPU; `--dist-backend gloo` shares GPU 0 "
                      "for a control-flow check" % (local_rank, torch.cuda.device_count()), file=sys.stderr)
                sys.exit(2)
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    device = torch.device("cuda", local_rank)
    torch.cuda.set_device(device)

    import workloads as W
    from raisin_amd import _lib, huffman
    from raisin_amd import shard as _shard
    _lib.check(_lib.lib().rsn_device_set(local_rank))

    n = args.mib << 20
    if args.profile_only:
        from oracle import oracle as O
        O.build()
        names = [x for x in args.profile_only.split(",") if x]
        print(json.dumps({"profile_only": run_other_configs(torch, device, n, O.host_cores(), False, names)}), flush=True)
        return
    seed = _shard.chunk_seed(rank, world)
    src = W.uniform_bytes(n, seed, 128, device)
    comp_buf = torch.empty(n + n // 8 + (1 << 20), dtype=torch.uint8, device=device)
    dec_buf = torch.empty(n + (1 << 20), dtype=torch.uint8, device=device)

    def step():
        c = huffman.compress_tensor(src, out=comp_buf)
        d = huffman.decompress_tensor(c, out=dec_buf)
        return c, d

    def fence():
        torch.cuda.synchronize(device)
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize(device)

    for _ in range(args.warmup):
        step()
    _lib.prof_enable(True)
    _lib.prof_reset()
    fence()
    t0 = time.perf_counter()
    t_enc = 0.0
    for _ in range(args.steps):
        a = time.perf_counter()
        c = huffman.compress_tensor(src, out=comp_buf)
        b = time.perf_counter()
        d = huffman.decompress_tensor(c, out=dec_buf)
        t_enc += b - a
    fence()
    elapsed = time.perf_counter() - t0
    prof = _lib.prof_get()
    _lib.prof_enable(False)

    t_max = elapsed
    if dist is not None:
        t_max = _shard.max_over_ranks(dist, elapsed, device if args.dist_backend == "nccl" else torch.device("cpu"))

    lossless = bool(d.numel() == n and torch.equal(d, src))   # checked BEFORE anything else touches dec_buf
    comp_n = int(c.numel())

    # context for the roofline: what a plain device-to-device copy and a read-only pass reach on THIS GPU
    # (torch kernels, timed with torch events on torch's stream; not part of the timed region above)
    def _rate(fn, nbytes, reps=10):
        for _ in range(3):
            fn()
        torch.cuda.synchronize(device)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize(device)
        return nbytes / (e0.elapsed_time(e1) / reps) / 1e6
    copy_gbps = _rate(lambda: dec_buf[:n].copy_(src), 2 * n)
    read_gbps = _rate(lambda: src.view(torch.int64).sum(), n)

    gather_ms = None
    gather_stuck = False
    if dist is not None:
        # config 5: compressed segments to rank 0 over RCCL, timed on its own (not part of `value`)
        seg = c.clone() if args.dist_backend == "nccl" else c.cpu()
        fence()
        # The gather is extra information: it runs under a watchdog so that a stuck collective can
        # never cost the job its throughput line (value does not depend on it).
        import threading
        box = {}

        def _gather():
            torch.cuda.set_device(device)
            g0 = time.perf_counter()
            got = _shard.gather_segments(dist, seg, 0)
            torch.cuda.synchronize(device)
            box["ms"] = (time.perf_counter() - g0) * 1e3
            box["ok"] = rank != 0 or (len(got) == world and got[0].numel() == comp_n)

        th = threading.Thread(target=_gather, daemon=True)
        th.start()
        th.join(120.0)
        gather_stuck = th.is_alive()
        if not gather_stuck and box.get("ok"):
            gather_ms = box["ms"]

    if rank == 0:
        K = args.steps
        per = {k: v[1] / v[0] for k, v in prof.items()}      # avg ms per launch
        cnt = {k: v[0] / K for k, v in prof.items()}         # launches per step
        C = comp_n
        alg = {  # ALGORITHMIC HBM bytes per launch (DESIGN.md "kernels")
            "huff_byte_hist": n, "huff_emit": n + C, "huff_dec_sync": C, "huff_dec_emit": C + n, "huff_dec_flat": C + n,
        }
        kernels = {}
        for k, ms in per.items():
            ent = {"ms": round(ms, 4), "launches_per_step": round(cnt[k], 2)}
            if k in alg:
                ent["algorithmic_bytes"] = alg[k]
                ent["GBps"] = round(alg[k] / ms / 1e6, 1)
                ent["frac_of_hbm_peak"] = round(alg[k] / ms / 1e6 / HBM_PEAK_GBPS, 4)
            kernels[k] = ent
        dom = max((k for k in per if k in alg), key=lambda k: per[k] * cnt[k])
        traffic = load_traffic(dom)
        roofline = {
            "kernel": dom, "bound": "hbm", "achieved": round(alg[dom] / per[dom] / 1e6, 1), "peak": HBM_PEAK_GBPS,
            "unit": "GB/s", "frac": round(alg[dom] / per[dom] / 1e6 / HBM_PEAK_GBPS, 4), "traffic": traffic[0] if traffic else None,
            "traffic_source": ("profiles/" + traffic[1]) if traffic else None, "algorithmic_bytes": alg[dom],
        }
        enc_ms = t_enc / K * 1e3
        dec_ms = (elapsed - t_enc) / K * 1e3
        gpu_enc_ms = sum(per[k] * cnt[k] for k in per if k.startswith("huff_") and "dec" not in k)
        out = {
            "metric": "encode+decode MB/s", "value": round(world * K * n / 1e6 / t_max, 1), "unit": "MB/s",
            "n_gpus": world, "steps": K, "warmup": args.warmup, "ms_per_step": round(t_max / K * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": "huffman encode+decode, %d MiB uniform-random bytes 0x00-0x7F per GPU (BASELINE configs[%d]), "
                                   "splitmix64 seed 0x%X" % (args.mib, 1 if world == 1 else 4, seed),
                       "algorithm": "huffman", "bytes_per_gpu": n, "chunks": world},
            "encode_MBps": round(n / 1e6 / (enc_ms / 1e3), 1), "decode_MBps": round(n / 1e6 / (dec_ms / 1e3), 1),
            "encode_ms": round(enc_ms, 4), "decode_ms": round(dec_ms, 4),
            "encode_kernel_ms": round(gpu_enc_ms, 4),
            "encode_frac_of_hbm_peak_2N_plus_C": round((2 * n + C) / (enc_ms / 1e3) / 1e9 / HBM_PEAK_GBPS, 4),
            "encode_input_read_frac_of_hbm_peak": round(n / (enc_ms / 1e3) / 1e9 / HBM_PEAK_GBPS, 4),
            "ratio_pct": round(100.0 * C / n, 3), "lossless": lossless,
            "hbm_calibration_GBps": {"torch_copy_read_plus_write": round(copy_gbps, 1), "torch_read_only_sum": round(read_gbps, 1)},
            "roofline": roofline, "kernels": kernels,
        }
        if gather_ms is not None:
            out["gather_ms"] = round(gather_ms, 3)
        if dist is not None and gather_ms is None:
            out["gather_ms"] = None
            out["gather_note"] = "segment gather did not complete within 120 s" if gather_stuck else "segment gather failed its size check"
        if world == 1:
            from oracle import oracle as O
            O.build()
            cores = O.host_cores()
            # (everything below is extra information: whatever goes wrong there is reported in the line, never instead of it)
            if not args.no_cpu:
                try:
                    def make(k):
                        return bytes(src[:min(k, n)].cpu().numpy())
                    cb = cpu_huffman(make, cores)
                    smp = cb["_s"]
                    pre = min(len(smp), 32 << 20)          # the threaded baseline's bytes on the whole sample, the plain oracle's on a prefix
                    ok = bytes(huffman.compress_tensor(src[:len(smp)].contiguous()).cpu().numpy()) == cb["_c"]
                    ok = ok and bytes(huffman.compress_tensor(src[:pre].contiguous()).cpu().numpy()) == O.huffman_compress(smp[:pre])
                    out["bit_exact_vs_oracle_on_sample"] = bool(ok)
                    out["cpu_baseline"] = _public(cb)
                except Exception as e:                      # noqa: BLE001
                    out["cpu_baseline_error"] = "%s: %s" % (type(e).__name__, e)
            del comp_buf, dec_buf
            torch.cuda.empty_cache()
            if not args.no_others:
                names = [x for x in args.others.split(",") if x]
                try:
                    out["other_configs"] = run_other_configs(torch, device, n, cores, not args.no_cpu, names)
                    out["other_configs_note"] = ALG_NOTE + "; one warm-up pass, then the mean of 2 timed passes per config"
                except Exception as e:                      # noqa: BLE001
                    out["other_configs_error"] = "%s: %s" % (type(e).__name__, e)
        print(json.dumps(out), flush=True)
    if dist is not None:
        if gather_stuck:
            # a collective is still pending: leave without waiting for it.  The throughput line is out and says so (gather_ms null +
            # gather_note); the exit code stays 0 -- the gather is extra information, and a failing code could cost the job its line.
            sys.stdout.flush()
            os._exit(0)
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
