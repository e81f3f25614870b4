"""GPU: BASELINE.json's full-size configurations, checked through size-independent
properties (the oracle would need minutes at 1 GiB), plus bench.py's N>1 control flow."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GIB = 1 << 30


def test_config2_huffman_1GiB_uniform_7bit(oracle):
    import torch
    from raisin_amd import huffman
    g = torch.Generator(device="cuda").manual_seed(0x5EED0002)
    src = torch.randint(0, 128, (GIB,), dtype=torch.uint8, device="cuda", generator=g)
    c = huffman.compress_tensor(src)
    head = bytes(c[:4096].cpu().numpy())
    sep = head.index(b"\\\n")
    ents, _ = oracle.header_entries(head[:sep + 2] + b"\x00")
    counts = torch.bincount(src.view(-1).to(torch.int32), minlength=128).cpu().numpy()
    assert sorted(int(f) for f, _ in ents) == sorted(int(x) for x in counts)          # header == true histogram
    assert sum(int(f) for f, _ in ents) == GIB
    assert c.numel() == sep + 3 + GIB * 7 // 8 and head[sep + 2] == 0                   # 128 near-equal counts -> all codes 7 bits
    d = huffman.decompress_tensor(c)
    assert d.numel() == GIB and torch.equal(d, src)                                     # encode -> decode round trip
    # flat code: symbol i occupies payload bits [7i, 7i+7) and its code is its rank in the tree walk;
    # check the first 1 Mi symbols of the 1 GiB payload against the library's own code table
    table = {r: code for r, f, code, l in huffman.table(bytes(src[:1 << 26].cpu().numpy()))}
    pay = np.frombuffer(bytes(c[sep + 3: sep + 3 + (1 << 20) * 7 // 8].cpu().numpy()), dtype=np.uint8)
    codes = np.unpackbits(pay)[: (1 << 20) * 7].reshape(-1, 7).dot(1 << np.arange(6, -1, -1))
    sample = np.frombuffer(bytes(src[:1 << 20].cpu().numpy()), dtype=np.uint8)
    # codes are a bijection of the 128 symbols and identical symbols always map to identical codes
    pairs = set(zip(sample.tolist(), codes.tolist()))
    assert len(pairs) == 128 and len({a for a, _ in pairs}) == 128 and len({b for _, b in pairs}) == 128
    assert len(ents) == 128 and len(table) == 128


def test_config3_lzss_1GiB_period_4096(oracle):
    import torch
    from raisin_amd import lz
    rng = np.random.default_rng(0x5EED0003)
    vals = np.array([v for v in range(256) if v not in (0x5C, 0xFF)], dtype=np.uint8)
    blk = vals[rng.integers(0, len(vals), size=4096)]
    src = torch.from_numpy(np.tile(blk, GIB // 4096)).cuda()
    c = lz.compress_tensor(src)
    host = bytes(c.cpu().numpy())
    head = oracle.lzss_compress(bytes(blk) * 3)                    # first block + two periods from the oracle
    first = head[: len(head) - 2 * len(b"<4096,4096>")]
    assert host.startswith(first)
    assert host[len(first):] == b"<4096,4096>" * (GIB // 4096 - 1)  # every later period is one token
    d = lz.decompress_tensor(c)
    assert d.numel() == GIB and torch.equal(d, src)


def test_config4_layered_256MiB_text(oracle):
    import torch
    from raisin_amd import huffman, lz
    rng = np.random.default_rng(0x5EED0004)
    vocab = [bytes(rng.integers(97, 123, size=int(rng.integers(2, 10)), dtype=np.uint8)) for _ in range(4096)]
    ranks = rng.zipf(1.3, size=1 << 26) % 4096
    text = b" ".join(vocab[r] for r in ranks)[: 1 << 28]
    src = torch.frombuffer(bytearray(text), dtype=torch.uint8).cuda()
    l1 = lz.compress_tensor(src)
    l2 = huffman.compress_tensor(l1)
    back = lz.decompress_tensor(huffman.decompress_tensor(l2))
    assert torch.equal(back, src)                                   # lossless through both layers
    # a 4 MiB prefix is bit-exact against the oracle (both layers); LZSS output is prefix-stable only
    # up to the last token, so compare the layered result of the prefix itself
    pre = text[: 1 << 22]
    got = bytes(huffman.compress_tensor(lz.compress_tensor(src[: 1 << 22].contiguous())).cpu().numpy())
    assert got == oracle.huffman_compress(oracle.lzss_compress(pre))


def test_bench_two_ranks_control_flow():
    """bench.py's world-size-2 path on one GPU (gloo collectives, both ranks on GPU 0)."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", "29731", os.path.join(ROOT, "bench.py"),
                          "--gpus", "2", "--steps", "2", "--warmup", "1", "--mib", "64", "--dist-backend", "gloo"],
                         capture_output=True, text=True, env=env, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    j = json.loads(line)
    assert j["n_gpus"] == 2 and j["lossless"] is True and j["scaling"] == "weak"
    assert j["config"]["chunks"] == 2 and "gather_ms" in j and j["value"] > 0


def test_beyond_4GiB_offsets():
    """5 GiB per call: every byte/bit offset above 2^32 (flat and general Huffman paths)."""
    import torch
    from raisin_amd import huffman
    n = 5 * GIB
    g = torch.Generator(device="cuda").manual_seed(7)
    base = torch.randint(0, 128, (GIB,), dtype=torch.uint8, device="cuda", generator=g)
    src = base.repeat(5)
    del base
    c = huffman.compress_tensor(src)
    assert c.numel() > n * 7 // 8                       # 128 equiprobable symbols: flat 7-bit code
    d = huffman.decompress_tensor(c)
    assert d.numel() == n and torch.equal(d, src)
    del c, d
    # general path: a skewed 64 MiB block tiled to 5 GiB
    w = torch.tensor([2.0 ** (-i / 6) for i in range(96)], device="cuda")
    blk = (torch.multinomial(w, 1 << 26, replacement=True).to(torch.uint8) + 32)
    src = blk.repeat(80)
    c = huffman.compress_tensor(src)
    d = huffman.decompress_tensor(c)
    assert d.numel() == n and torch.equal(d, src)
