#!/usr/bin/env python3
"""bench.py -- headline benchmark: BASELINE.json configs[1] (config 5 when --gpus > 1).

  python bench.py --gpus N --steps K --warmup W

Workload (per GPU): `-algorithm=huffman` on a 1 GiB uniform-random buffer (bytes
uniform over 0x00..0x7F so that the reference's rune-level Huffman is lossless,
SURVEY.md 8d "2a"; seed 0x5EED0002, rank r uses 0x5EED0050+r when N > 1).
A step is what the reference's BenchmarkFile times (engine/engine.go:379-406):
compress, then decompress, of one buffer, input already resident in HBM.
value = uncompressed MB (1e6 B) through encode+decode per second, whole job.

Timing of the kernels comes from HIP events recorded by librsn on its own launch
stream (rsn_prof_*), live inside the timed region.  The CPU baseline is the
oracle (oracle/, a C restatement of the reference: the reference is Go and cannot
run here) timed on a bounded sample on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (guides/MI355X_MICROARCH.md)


def make_input(torch, n, seed, device, hi=128):
    g = torch.Generator(device=device).manual_seed(seed)
    return torch.randint(0, hi, (n,), dtype=torch.uint8, device=device, generator=g)


def cpu_baseline(sample):
    """Oracle encode+decode on a bounded sample, single thread."""
    from oracle import oracle as O
    O.build()
    t0 = time.perf_counter()
    c = O.huffman_compress(sample)
    t1 = time.perf_counter()
    d = O.huffman_decompress(c)
    t2 = time.perf_counter()
    assert d == sample
    return c, {
        "value": round(len(sample) / 1e6 / (t2 - t0), 3), "unit": "MB/s", "cores": 1, "kind": "port",
        "sample": "%d MiB prefix of the same buffer, encode %.2f s + decode %.2f s, oracle/ (C restatement) on 1 host core of %d"
                  % (len(sample) >> 20, t1 - t0, t2 - t1, os.cpu_count()),
        "encode_MBps": round(len(sample) / 1e6 / (t1 - t0), 3), "decode_MBps": round(len(sample) / 1e6 / (t2 - t1), 3),
    }


def load_traffic():
    """HBM bytes per launch from the committed PMC profile, if present (scripts/profile.sh)."""
    import glob
    best = None
    for p in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc.json"))):
        try:
            best = json.load(open(p))
        except Exception:
            pass
    return best or {}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--mib", type=int, default=1024, help="buffer size per GPU (default: the 1 GiB of BASELINE.json)")
    ap.add_argument("--cpu-sample-mib", type=int, default=128)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--dist-backend", default="nccl", help="nccl (= RCCL, one GPU per rank) or gloo (control-flow test: all ranks share GPU 0)")
    args = ap.parse_args()

    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    if world > 1:
        import torch.distributed as dist
        if args.dist_backend == "gloo":
            local_rank = 0                       # test mode: every rank drives GPU 0, collectives run on host tensors
            dist.init_process_group("gloo")
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    elif args.gpus > 1:
        print("bench.py: --gpus %d needs the torch.distributed launcher (WORLD_SIZE unset)" % args.gpus, file=sys.stderr)
        sys.exit(2)
    device = torch.device("cuda", local_rank)
    torch.cuda.set_device(device)

    from raisin_amd import _lib, huffman
    _lib.check(_lib.lib().rsn_device_set(local_rank))

    n = args.mib << 20
    from raisin_amd import shard as _shard
    seed = _shard.chunk_seed(rank, world)
    src = make_input(torch, n, seed, device)
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
        from raisin_amd import shard
        t_max = shard.max_over_ranks(dist, elapsed, device if args.dist_backend == "nccl" else torch.device("cpu"))

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
        from raisin_amd import shard
        seg = c.clone() if args.dist_backend == "nccl" else c.cpu()
        fence()
        # The gather is extra information: it runs under a watchdog so that a stuck collective can
        # never cost the job its throughput line (value does not depend on it).
        import threading
        box = {}

        def _gather():
            torch.cuda.set_device(device)
            g0 = time.perf_counter()
            got = shard.gather_segments(dist, seg, 0)
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
        traffic = load_traffic().get(dom)
        roofline = {
            "kernel": dom, "bound": "hbm", "achieved": round(alg[dom] / per[dom] / 1e6, 1), "peak": HBM_PEAK_GBPS,
            "unit": "GB/s", "frac": round(alg[dom] / per[dom] / 1e6 / HBM_PEAK_GBPS, 4), "traffic": traffic,
        }
        enc_ms = t_enc / K * 1e3
        dec_ms = (elapsed - t_enc) / K * 1e3
        gpu_enc_ms = sum(per[k] * cnt[k] for k in per if k.startswith("huff_") and "dec" not in k)
        out = {
            "metric": "encode+decode MB/s", "value": round(world * K * n / 1e6 / t_max, 1), "unit": "MB/s",
            "n_gpus": world, "steps": K, "warmup": args.warmup, "ms_per_step": round(t_max / K * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": "huffman encode+decode, %d MiB uniform-random bytes 0x00-0x7F per GPU (BASELINE configs[%d])"
                                   % (args.mib, 1 if world == 1 else 4),
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
        if world == 1 and not args.no_cpu:
            sample = bytes(src[: min(n, args.cpu_sample_mib << 20)].cpu().numpy())
            ref_c, cb = cpu_baseline(sample)
            gpu_c = bytes(huffman.compress_tensor(src[: len(sample)].contiguous()).cpu().numpy())
            out["bit_exact_vs_oracle_on_sample"] = bool(gpu_c == ref_c)
            out["cpu_baseline"] = cb
        if dist is not None and gather_ms is None:
            out["gather_ms"] = None
            out["gather_note"] = "segment gather did not complete within 120 s" if gather_stuck else "segment gather failed its size check"
        print(json.dumps(out), flush=True)
    if dist is not None:
        if gather_stuck:
            os._exit(0)                 # a collective is still pending: leave without waiting for it
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
