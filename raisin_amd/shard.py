"""Multi-GPU sharding of independent chunks (BASELINE config 5).

The reference's unit of work is the file: engine.CompressFiles loops over files
and writes one .rsn each (engine/engine.go:150-154).  Chunks are therefore
independent -- every rank runs the whole single-GPU pipeline on its own chunks
and there is NO collective on the data path.  The only exchange is the optional
gather of the finished, variable-length segments to rank 0 (RCCL has no gatherv:
one all_gather of the sizes, then grouped send/recv straight into rank 0).
"""


def chunks_for_rank(n_chunks, rank, world):
    """chunk k -> rank k mod world."""
    return [k for k in range(n_chunks) if k % world == rank]


def chunk_seed(k, world):
    """Seeds of BASELINE.md: 0x5EED0002 for the single-chunk run, 0x5EED0050+k for config 5."""
    return 0x5EED0002 if world == 1 else 0x5EED0050 + k


def gather_segments(dist, segment, dst=0):
    """Gathers one variable-length uint8 tensor per rank to `dst`.
    Returns the list of segments in rank order on `dst`, None elsewhere."""
    import torch
    world, rank = dist.get_world_size(), dist.get_rank()
    dev = segment.device
    sizes = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(sizes, torch.tensor([segment.numel()], dtype=torch.int64, device=dev))
    sizes = [int(s.item()) for s in sizes]
    if rank == dst:
        bufs = [segment if r == dst else torch.empty(sizes[r], dtype=torch.uint8, device=dev) for r in range(world)]
        ops = [dist.P2POp(dist.irecv, bufs[r], r) for r in range(world) if r != dst and sizes[r]]
    else:
        bufs = None
        ops = [dist.P2POp(dist.isend, segment, dst)] if segment.numel() else []
    if ops:
        for w in dist.batch_isend_irecv(ops):
            w.wait()
    return bufs


def max_over_ranks(dist, seconds, device):
    """The job's time is the slowest rank's time."""
    import torch
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
