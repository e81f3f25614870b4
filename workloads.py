"""Synthetic inputs of BASELINE.json's configs, one definition for
bench.py, the tests and the scripts.  Everything is a pure function of (seed, n): byte i never
depends on n, so a prefix of a buffer is the buffer of that length, and the same bytes come out on
the CPU and on the GPU (torch ops only; numpy cross-check in tests/test_workloads.py).

    splitmix64 draw i of seed s:  z = s + (i+1)*0x9E3779B97F4A7C15 (mod 2^64)
                                  z = (z ^ z>>30) * 0xBF58476D1CE4E5B9
                                  z = (z ^ z>>27) * 0x94D049BB133111EB ;  z ^= z>>31

  config 2a  uniform_bytes(seed 0x5EED0002, hi=128)   byte i = low byte of draw i, & 0x7F
  config 2b  uniform_bytes(seed 0x5EED0002, hi=256)   byte i = low byte of draw i
  config 3   periodic(seed 0x5EED0003)                one 4096-byte block, byte j = V[(draw j >> 11) % 254],
                                                      V = the 254 byte values other than 0x5C and 0xFF, repeated
  config 4   zipf_text(seed 0x5EED0004)               4096 lower-case words of 2..9 letters, word k drawn with
                                                      p ~ k^-1.3, separated by ' ' ('\\n' after every 16th word)
  config 5   chunk k = uniform_bytes(seed 0x5EED0050 + k, hi=128)
  skewed     skewed_bytes(seed 0x5EED0012)            96 printable symbols, p(k) ~ 2^(-k/6): unequal code lengths,
                                                      the general (not the flat) Huffman kernels
"""
import math

SEED_2 = 0x5EED0002
SEED_3 = 0x5EED0003
SEED_4 = 0x5EED0004
SEED_5 = 0x5EED0050
SEED_SKEW = 0x5EED0012

_GAMMA = 0x9E3779B97F4A7C15
_M1 = 0xBF58476D1CE4E5B9
_M2 = 0x94D049BB133111EB
_CHUNK = 1 << 25


def _s64(v):
    """The signed 64-bit integer with the same bits as v mod 2^64 (torch has no uint64 arithmetic)."""
    v &= (1 << 64) - 1
    return v - (1 << 64) if v >= (1 << 63) else v


def _lsr(torch, z, s):
    """Logical right shift of an int64 tensor."""
    return (z >> s) & ((1 << (64 - s)) - 1)


def splitmix64(torch, seed, start, count, device):
    """Draws start .. start+count-1 of the stream, as int64 bit patterns (wrap-around arithmetic)."""
    idx = torch.arange(start + 1, start + count + 1, dtype=torch.int64, device=device)
    z = idx * _s64(_GAMMA) + _s64(seed)
    z = (z ^ _lsr(torch, z, 30)) * _s64(_M1)
    z = (z ^ _lsr(torch, z, 27)) * _s64(_M2)
    return z ^ _lsr(torch, z, 31)


def uniform_bytes(n, seed=SEED_2, hi=128, device="cpu"):
    """Configs 2a (hi=128), 2b (hi=256) and 5."""
    import torch
    out = torch.empty(n, dtype=torch.uint8, device=device)
    for s in range(0, n, _CHUNK):
        c = min(_CHUNK, n - s)
        out[s:s + c] = (splitmix64(torch, seed, s, c, device) & (hi - 1)).to(torch.uint8)
    return out


def periodic(n, seed=SEED_3, period=4096, device="cpu"):
    """Config 3: the escaped period stays `period` because the block holds neither 0x5C nor 0xFF."""
    import torch
    vals = torch.tensor([v for v in range(256) if v not in (0x5C, 0xFF)], dtype=torch.uint8, device=device)
    draw = _lsr(torch, splitmix64(torch, seed, 0, period, device), 11) % 254
    blk = vals[draw]
    reps = (n + period - 1) // period
    return blk.repeat(reps)[:n].contiguous()


def skewed_bytes(n, seed=SEED_SKEW, device="cpu"):
    """96 printable symbols (0x20..0x7F) with p(k) ~ 2^(-k/6): code lengths from 3 to ~19 bits."""
    import torch
    w = [2.0 ** (-k / 6.0) for k in range(96)]
    tot = sum(w)
    cdf, acc = [], 0.0
    for x in w:
        acc += x / tot
        cdf.append(acc)
    cdf[-1] = 1.0
    cdf_t = torch.tensor(cdf, dtype=torch.float64, device=device)
    out = torch.empty(n, dtype=torch.uint8, device=device)
    for s in range(0, n, _CHUNK):
        c = min(_CHUNK, n - s)
        u = _lsr(torch, splitmix64(torch, seed, s, c, device), 11).to(torch.float64) * (1.0 / (1 << 53))
        out[s:s + c] = (torch.searchsorted(cdf_t, u, right=True).clamp_(max=95) + 32).to(torch.uint8)
    return out


_VOCAB = 4096
_ZIPF_S = 1.3


def _vocabulary(torch, seed, device):
    """Word k (0-based): length 2 + (draw (2k) >> 11) % 8 letters, letter j = 'a' + (draw (V*2 + 16k + j) >> 11) % 26.
    Returns (table [V, 10] uint8 -- letters then the separator slot, lens [V] int64 including the separator)."""
    d = _lsr(torch, splitmix64(torch, seed ^ 0x766F636162, 0, _VOCAB * 2 + _VOCAB * 16, device), 11)   # its own stream ("vocab")
    lens = 2 + d[0:2 * _VOCAB:2] % 8
    letters = (97 + d[2 * _VOCAB:].view(_VOCAB, 16)[:, :9] % 26).to(torch.uint8)
    table = torch.zeros(_VOCAB, 10, dtype=torch.uint8, device=device)
    table[:, :9] = letters
    return table, lens + 1


def zipf_text(n, seed=SEED_4, device="cpu"):
    """Config 4: ASCII, no '<', no byte >= 0x80, so both layers are reference-lossless."""
    import torch
    table, lens = _vocabulary(torch, seed, device)
    w = [(k + 1) ** -_ZIPF_S for k in range(_VOCAB)]
    tot = math.fsum(w)
    cdf, acc = [], 0.0
    for x in w:
        acc += x / tot
        cdf.append(acc)
    cdf[-1] = 1.0
    cdf_t = torch.tensor(cdf, dtype=torch.float64, device=device)
    out = torch.empty(n, dtype=torch.uint8, device=device)
    done, word0 = 0, 0
    while done < n:
        wchunk = min(1 << 23, (n - done) // 3 + 16)                # whole words per round: the bytes do not depend on the rounds
        u = _lsr(torch, splitmix64(torch, seed, word0, wchunk, device), 11).to(torch.float64) * (1.0 / (1 << 53))
        rank = torch.searchsorted(cdf_t, u, right=True).clamp_(max=_VOCAB - 1)
        wl = lens[rank]
        ends = torch.cumsum(wl, 0)
        take = min(int(ends[-1].item()), n - done)
        starts = ends - wl
        j = torch.arange(take, dtype=torch.int64, device=device)
        wi = torch.searchsorted(ends, j, right=True)                # word of output byte j
        ci = j - starts[wi]                                         # character inside it (== len-1: the separator)
        ch = table[rank[wi], ci.clamp(max=9)]
        sep = torch.where((word0 + wi) % 16 == 15, 10, 32).to(torch.uint8)
        out[done:done + take] = torch.where(ci == wl[wi] - 1, sep, ch)
        done += take
        word0 += wchunk
    return out


def config_input(name, n, device="cpu", chunk=0):
    if name == "2a":
        return uniform_bytes(n, SEED_2, 128, device)
    if name == "2b":
        return uniform_bytes(n, SEED_2, 256, device)
    if name == "3":
        return periodic(n, SEED_3, 4096, device)
    if name == "4":
        return zipf_text(n, SEED_4, device)
    if name == "5":
        return uniform_bytes(n, SEED_5 + chunk, 128, device)
    if name == "skewed":
        return skewed_bytes(n, SEED_SKEW, device)
    raise ValueError(name)
