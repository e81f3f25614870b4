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
