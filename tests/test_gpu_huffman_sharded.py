"""GPU parity: ONE Huffman stream from G slices of one input (rsn_huffman_compress_sharded, SURVEY 8e "intra-file sharding") is byte
for byte the single call's stream -- and therefore the oracle's: summed histograms, one tree, bit-offset stitching around the single
front pad (huffman.go:245-255).  G workers share the one GPU of this box (how RSN_BATCH_WORKERS exercises the batch split)."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _inputs(samiam):
    import workloads as W
    rng = np.random.default_rng(41)
    flat = bytes(W.config_input("2a", 3 << 20).numpy())                          # 128 equiprobable symbols: the flat 7-bit path
    skew = bytes(W.config_input("skewed", (2 << 20) + 12345).numpy())            # general codes up to 19 bits
    runes = bytes(W.config_input("2b", (1 << 20) + 77).numpy())                  # every byte value: the rune path, invalid sequences
    text = "héllo wörld — ünïcode ✓ 𝄞 ".encode() * 30000                          # valid multi-byte sequences: a cut must not split one
    mixed = skew[:700000] + text[:600001] + flat[:500000]                        # an all-ASCII slice next to slices with runes
    return [samiam * 40, flat, skew, runes, text, mixed, b"a" * 300000, b"ab" * 100000 + b"c", bytes(rng.integers(0, 4, 200001, dtype=np.uint8))]


@pytest.mark.parametrize("G", [1, 2, 3, 8])
def test_sharded_stream_equals_the_single_call(oracle, samiam, G):
    from raisin_amd import huffman
    for k, data in enumerate(_inputs(samiam)):
        ref = huffman.Compress(data)
        got = huffman.CompressSharded(data, G)
        assert got == ref, (G, k, len(data), len(got), len(ref))
        if len(data) <= (3 << 20):
            assert ref == oracle.huffman_compress(data), (k,)


def test_sharded_slices_of_uneven_bits_and_many_workers(oracle):
    """Slices whose bit totals are not multiples of 8 (every joint shares a byte), more slices than the input has kilobytes, and a
    64-slice split of 40 MiB; the decode of the stitched stream is the input."""
    from raisin_amd import huffman
    rng = np.random.default_rng(5)
    p = np.array([2.0 ** (-i / 3) for i in range(40)]); p /= p.sum()
    data = rng.choice(np.arange(40, 80, dtype=np.uint8), size=40 << 20, p=p).tobytes()
    ref = huffman.Compress(data)
    for G in (5, 64):
        assert huffman.CompressSharded(data, G) == ref
    assert huffman.Decompress(ref) == data
    small = data[:5000]
    assert huffman.CompressSharded(small, 200) == huffman.Compress(small) == oracle.huffman_compress(small)
    assert huffman.CompressSharded(b"xy", 8) == huffman.Compress(b"xy")


def test_env_switch_routes_the_plain_entry_point():
    """RSN_HUFF_SHARDS=4: rsn_huffman_compress itself (what the cgo shim binds) produces the stream from four slices -- same bytes."""
    code = ("import sys, hashlib; sys.path.insert(0, %r)\n"
            "import workloads as W\nfrom raisin_amd import huffman\n"
            "d = bytes(W.config_input('skewed', 6 << 20).numpy())\n"
            "print(hashlib.sha256(huffman.Compress(d)).hexdigest())\n" % ROOT)
    outs = []
    for env in ({}, {"RSN_HUFF_SHARDS": "4"}, {"RSN_HUFF_SHARDS": "3", "RSN_HOST_TIMING": "1"}):
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=dict(os.environ, **env))
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(r.stdout.split()[-1])
    assert outs[0] == outs[1] == outs[2]
