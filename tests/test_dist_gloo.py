"""N>1 path on CPU: two processes over gloo run the chunk sharding and the
variable-length gather of finished segments exactly as bench.py does on RCCL."""
import os
import socket
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist

    from raisin_amd import shard
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    try:
        mine = shard.chunks_for_rank(5, rank, world)
        # one variable-length "segment" per rank, content derived from its chunk seeds
        g = torch.Generator().manual_seed(shard.chunk_seed(rank, world))
        seg = torch.randint(0, 256, (1000 + 777 * rank,), dtype=torch.uint8, generator=g)
        got = shard.gather_segments(dist, seg, 0)
        t = shard.max_over_ranks(dist, 1.0 + rank, torch.device("cpu"))
        out = {"rank": rank, "chunks": mine, "tmax": t, "sizes": [int(x.numel()) for x in got] if got else None}
        # the gather as bench.py times it: a channel's first use is slow (RCCL sets peer-to-peer channels up lazily) --
        # simulated here by a first gather_segments call that takes 0.4 s -- and must land in gather_warmup_ms, not in gather_ms
        import time
        real, calls = shard.gather_segments, []

        def slow_first(d, seg_, dst=0):
            calls.append(int(seg_.numel()))
            if len(calls) == 1:
                time.sleep(0.4)
            return real(d, seg_, dst)
        shard.gather_segments = slow_first
        try:
            rounds = [seg, torch.empty(0, dtype=torch.uint8) if rank == 1 else seg[:10]]     # a rank with fewer chunks sends an empty segment
            g = shard.timed_gather(dist, rounds, torch.device("cpu"), 0)
        finally:
            shard.gather_segments = real
        out["timed"] = {"calls": calls, "gather_ms": g["gather_ms"], "gather_warmup_ms": g["gather_warmup_ms"], "gathered_bytes": g["gathered_bytes"],
                        "segments": [[int(x.numel()) for x in r] for r in g["segments"]] if g["segments"] is not None else None}
        out["per_rank"] = shard.per_rank(dist, [10.0 + rank, 0.5 * (rank + 1)], torch.device("cpu"))
        if rank == 0:
            ok = True
            for r in range(world):
                gr = torch.Generator().manual_seed(shard.chunk_seed(r, world))
                ok = ok and torch.equal(got[r], torch.randint(0, 256, (1000 + 777 * r,), dtype=torch.uint8, generator=gr))
            out["ok"] = ok
        q.put(out)
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_two_rank_gather_and_timing():
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=90) for _ in procs), key=lambda d: d["rank"])
    for p in procs:
        p.join(30)
    assert res[0]["chunks"] == [0, 2, 4] and res[1]["chunks"] == [1, 3]
    assert res[0]["sizes"] == [1000, 1777] and res[0]["ok"] is True and res[1]["sizes"] is None
    assert res[0]["tmax"] == res[1]["tmax"] == 2.0
    for r in (0, 1):
        t = res[r]["timed"]
        assert t["calls"][0] == 1 and len(t["calls"]) == 3                      # the untimed 1-byte round, then the two timed ones
        assert t["gather_warmup_ms"] >= 400 and t["gather_ms"] < 300, t         # the slow first use is NOT inside gather_ms
        assert res[r]["per_rank"] == [[10.0, 0.5], [11.0, 1.0]]                 # every rank's figures, in rank order, on every rank
    assert res[0]["timed"]["segments"] == [[1000, 1777], [10, 0]] and res[0]["timed"]["gathered_bytes"] == 2787
    assert res[1]["timed"]["segments"] is None


@pytest.mark.timeout(300)
def test_bench_without_a_launcher_starts_its_own_ranks(monkeypatch):
    """`python bench.py --gpus 2` as the driver calls it (no WORLD_SIZE): bench.py must become the launcher's parent and forward the
    child's exit code -- on this GPU-less box the two ranks fail when they reach for cuda:0, which is the point of the check: they were
    started (two rank tracebacks / a torchrun failure summary), nothing printed the old "started under a launcher" refusal, and the
    exit code is the child's, not 2.  The self-launch itself must not import torch (it runs before anything may touch a GPU)."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--mib", "1",
                          "--dist-backend", "gloo"], capture_output=True, text=True, env=env, timeout=280)
    import torch
    if not torch.cuda.is_available():
        assert out.returncode not in (0, 2), (out.returncode, out.stderr[-500:])
        assert "launcher started" not in out.stderr and "torch.distributed.run" not in out.stdout
        assert "ChildFailedError" in out.stderr or "exitcode" in out.stderr       # torchrun's summary of the two failed ranks
    else:
        assert out.returncode == 0 and out.stdout.count('"n_gpus": 2') == 1
    src = open(os.path.join(ROOT, "bench.py")).read()
    body = src[src.index("def self_launch"):src.index("def main")]
    assert "import torch" not in body
