"""literal.py -- a SECOND, independent restatement of the reference's two codecs, written the way the Go reads: strings of '0' / '1',
a list-backed container/heap, bytes.Index, the per-position recursion.  TEST INFRASTRUCTURE ONLY (like everything under oracle/): it
exists so that the C restatement (huffman_oracle.c, lzss_oracle.c) is checked against something that was not derived from it --
tests/test_oracle.py runs both on a few hundred small inputs and demands the same bytes.  It is not reference output either: the
reference is pure Go and no Go toolchain exists in the image ("parity partially pinned").  Pure Python, small inputs only.

Follows /root/reference/compressor/huffman/huffman.go and /root/reference/compressor/lz/lzss.go, line cites inline.  Where Go's behaviour
is not defined by the source alone it is stated here:
  * map iteration order (huffman.go:312: the header's entry order) is random in Go; `header_order` below takes the order as an
    argument -- the default is this repository's canonical one (ascending rune, a '\\' that would be last moved to the front);
  * container/heap is go1.15's src/container/heap/heap.go (Init / Push / Pop / up / down), restated from its published algorithm;
  * `range` over a string decodes UTF-8 with Go's accept ranges; an invalid byte is U+FFFD and advances one byte.
"""

RUNE_ERROR = 0xFFFD


# ------------------------------------------------------------------ Go's utf8 (unicode/utf8: DecodeRuneInString as `range` uses it)
def _decode_rune(b, i):
    n = len(b)
    b0 = b[i]
    if b0 < 0x80:
        return b0, 1
    if b0 < 0xC2 or b0 > 0xF4:
        return RUNE_ERROR, 1
    need = 2 if b0 < 0xE0 else 3 if b0 < 0xF0 else 4
    if i + need > n:
        return RUNE_ERROR, 1
    lo, hi = 0x80, 0xBF
    if b0 == 0xE0:
        lo = 0xA0
    elif b0 == 0xED:
        hi = 0x9F
    elif b0 == 0xF0:
        lo = 0x90
    elif b0 == 0xF4:
        hi = 0x8F
    if not (lo <= b[i + 1] <= hi):
        return RUNE_ERROR, 1
    r = b0 & (0x1F if need == 2 else 0x0F if need == 3 else 0x07)
    r = (r << 6) | (b[i + 1] & 0x3F)
    for k in range(2, need):
        if (b[i + k] & 0xC0) != 0x80:
            return RUNE_ERROR, 1
        r = (r << 6) | (b[i + k] & 0x3F)
    return r, need


def go_runes(b):
    """[(byte offset, rune)] as `for j, c := range string(b)` yields them."""
    out, i = [], 0
    while i < len(b):
        r, k = _decode_rune(b, i)
        out.append((i, r))
        i += k
    return out


def go_string_of_rune(r):
    """string(rune): the UTF-8 encoding; surrogates and values beyond U+10FFFF become U+FFFD."""
    if r < 0 or r > 0x10FFFF or 0xD800 <= r <= 0xDFFF:
        r = RUNE_ERROR
    return chr(r).encode("utf-8")


# ------------------------------------------------------------------ container/heap over a list, Less = freq only (huffman.go:43-45)
class _Heap:
    def __init__(self, items):
        self.h = list(items)

    def less(self, i, j):
        return self.h[i][0] < self.h[j][0]

    def swap(self, i, j):
        self.h[i], self.h[j] = self.h[j], self.h[i]

    def up(self, j):
        while True:
            i = (j - 1) // 2 if j > 0 else 0              # Go: (j - 1) / 2 truncates towards zero: j = 0 gives 0
            if i == j or not self.less(j, i):
                break
            self.swap(i, j)
            j = i

    def down(self, i0, n):
        i = i0
        while True:
            j1 = 2 * i + 1
            if j1 >= n or j1 < 0:
                break
            j = j1
            j2 = j1 + 1
            if j2 < n and self.less(j2, j1):
                j = j2
            if not self.less(j, i):
                break
            self.swap(i, j)
            i = j
        return i > i0

    def init(self):
        n = len(self.h)
        for i in range(n // 2 - 1, -1, -1):
            self.down(i, n)

    def push(self, x):
        self.h.append(x)
        self.up(len(self.h) - 1)

    def pop(self):
        n = len(self.h) - 1
        self.swap(0, n)
        self.down(0, n)
        return self.h.pop()


# ------------------------------------------------------------------ huffman.go
def build_tree(sym_freqs):
    """huffman.go:58-103.  Trees are (freq, rune) leaves or (freq, left, right) nodes."""
    keys = sorted(sym_freqs)                                  # sort.Ints(keys)
    values = sorted(sym_freqs.values())                       # sort.Ints(values)
    temp1, temp2 = [], []
    for value in values:                                      # :76-87
        for i, key in enumerate(keys):
            if sym_freqs[key] == value:
                temp1.append(key)
                temp2.append(value)
                keys[i] = keys[-1]                            # remove(): s[i] = s[len(s)-1]; s[:len(s)-1]
                keys.pop()
                keys.sort()
                break
    trees = _Heap([(temp2[i], temp1[i]) for i in range(len(sym_freqs))])
    trees.init()
    while len(trees.h) > 1:
        a = trees.pop()
        b = trees.pop()
        trees.push((a[0] + b[0], a, b))
    return trees.pop()                                        # (an empty table: IndexError here, a panic there -- :102)


def print_codes(tree, prefix="", vals=None, bins=None):
    """huffman.go:110-127: depth first, '0' = left first."""
    vals = [] if vals is None else vals
    bins = [] if bins is None else bins
    if len(tree) == 2:
        vals.append(tree[1])
        bins.append(prefix)
    else:
        print_codes(tree[1], prefix + "0", vals, bins)
        print_codes(tree[2], prefix + "1", vals, bins)
    return vals, bins


def as_byte_slice(bits):
    """bitString.AsByteSlice, huffman.go:174-191: groups of eight from the END; the leading group may be short."""
    out = []
    i = len(bits)
    while i > 0:
        s = bits[0:i] if i - 8 < 0 else bits[i - 8:i]
        out.insert(0, int(s, 2))
        i -= 8
    return bytes(out)


def header_order(sym_freqs):
    """This repository's choice among Go's random map orders: ascending rune, a '\\' that would be last goes first
    (the reference's own decoder indexes past the header's end on that order, huffman.go:210)."""
    keys = sorted(sym_freqs)
    if len(keys) > 1 and keys[-1] == 0x5C:
        keys = [0x5C] + keys[:-1]
    return keys


def huffman_compress(data, order=None):
    """huffman.go:299-325 + encode :229-256."""
    data = bytes(data)
    sym_freqs = {}
    for _, c in go_runes(data):                               # :309-311
        sym_freqs[c] = sym_freqs.get(c, 0) + 1
    estring = b""
    for key in (order or header_order(sym_freqs)):            # :312-318 (Go: random order)
        if key != 10:
            estring += str(sym_freqs[key]).encode() + b"|" + go_string_of_rune(key)
        else:
            estring += str(sym_freqs[key]).encode() + b"|\\n"
    tree = build_tree(sym_freqs)
    vals, bins = print_codes(tree)
    answer = []
    for _, c in go_runes(data):                               # :235-241
        answer.append(bins[vals.index(c)] if c in vals else bins[0])
    answer = "".join(answer)
    diff = format(8 - len(answer) % 8, "b")                   # :245
    if diff == "1000":
        diff = "0"
    first = as_byte_slice(diff)
    final = as_byte_slice(answer)
    return estring + b"\\\n" + first + final


def decode_tree(tree):
    """huffman.go:196-227, byte for byte (the scan indexes bytes; a symbol is the rune that BEGINS at byte i + 1)."""
    sym_freqs = {}
    temp = ""
    runes = dict(go_runes(tree))                              # byte offset -> rune (`for j, c := range tree`)
    i = 0
    while i < len(tree):
        ch = tree[i:i + 1]
        if ch != b"|":
            if ch.isdigit() and ch.isascii():                 # strconv.Atoi(string(tree[i])) succeeds on one ASCII digit (a byte >= 0x80 becomes a two-byte string: no)
                temp += ch.decode()
        else:
            freq = int(temp) if temp.strip().isdigit() else 0
            temp = ""
            if tree[i + 1] == 0x5C and tree[i + 2] == ord("n"):   # IndexError where Go panics (:210)
                sym_freqs[10] = freq
                i += 1
            elif i + 1 in runes:
                sym_freqs[runes[i + 1]] = freq
            if i + 1 >= len(tree):
                raise IndexError("header ends in '|'")
            i += 1
        i += 1
    return build_tree(sym_freqs)


def huffman_decompress(blob):
    """huffman.go:258-297 + findCodes :131-153 (as a loop: the recursion is tail calls)."""
    blob = bytes(blob)
    k = blob.find(b"\\\n")
    if k < 0:
        raise IndexError("no separator (sections[1] out of range)")
    tree = decode_tree(blob[:k])
    byte_arr = blob[k + 2:]
    if not byte_arr:
        raise IndexError("no pad byte")
    diff = byte_arr[0]
    content = "".join(format(b, "08b") for b in byte_arr[1:])
    if diff > len(content):
        raise IndexError("slice bounds out of range")
    data = content[diff:]
    answer = b""
    i, mx, node = 0, len(data), tree
    while True:                                               # findCodes
        if i > mx:
            break
        if len(node) == 2:
            answer += go_string_of_rune(node[1])
            if i < mx:
                node = tree
                if len(tree) == 2:
                    raise RecursionError("a bare leaf with data left: the reference recurses without end")
                continue
            break
        if i == mx:
            raise IndexError("data[i] out of range: the payload ends inside a codeword")
        node = node[1] if data[i] == "0" else node[2]
        i += 1
    return answer


# ------------------------------------------------------------------ lzss.go
def encode_opening_symbols(data):
    """lzss.go:369-389 (foundEscape is never set: its branches are dead)."""
    out = bytearray()
    for val in data:
        if val == 0x3C:
            val = 0xFF
        elif val == 0xFF or val == 0x5C:
            out.append(0x5C)
        out.append(val)
    return bytes(out)


def decode_opening_symbols(data):
    """lzss.go:391-406."""
    out = bytearray()
    found = False
    for val in data:
        if val == 0xFF and not found:
            out.append(0x3C)
        elif val == 0x5C and not found:
            found = True
        else:
            found = False
            out.append(val)
    return bytes(out)


def _worker(search, scan, nxt):
    """compressorWorker, lzss.go:166-184: (is_reference, negative offset, size) for the longest scan + prefix of nxt that occurs in search."""
    index = search.find(scan)                                 # bytes.Index: the leftmost occurrence
    if index == -1:
        return (False, 0, len(scan))
    negative = len(search) - index
    if nxt:
        deeper = _worker(search, scan + nxt[:1], nxt[1:])
        if deeper[0]:
            return deeper
    return (True, negative, len(scan))


def lzss_compress(data, window=4096):
    """CompressAsync, lzss.go:109-151 (the goroutine per position as a loop; maxSearchBufferLength <= 0: the whole prefix)."""
    import sys
    fc = encode_opening_symbols(bytes(data))
    refs = []
    sys.setrecursionlimit(max(sys.getrecursionlimit(), len(fc) + 1000))
    for i in range(len(fc)):
        search = fc[:i]
        if window > 0 and len(search) > window:
            search = search[len(search) - window:]
        refs.append(_worker(search, fc[i:i + 1], fc[i + 1:]))
    out = bytearray()
    ignore = 0
    for i, (is_ref, neg, size) in enumerate(refs):            # :134-151
        if ignore > 0:
            ignore -= 1
        elif is_ref:
            ignore = size - 1
            enc = b"<" + str(neg).encode() + b"," + str(size).encode() + b">"      # getEncoding :318-320
            out += enc if len(enc) < size else fc[i:i + size]
        else:
            out += fc[i:i + 1]
    return bytes(out)


def lzss_decompress(blob):
    """Decompress, lzss.go:323-364.  Atoi's errors are dropped there (a field that is not a number reads as 0); a slice that leaves the
    data raises here (Go: panics, or -- within the slice's capacity -- reads stale bytes: not reproduced)."""
    def atoi(b):
        try:
            s = b.decode("ascii")
        except UnicodeDecodeError:
            return 0
        if not s or not (s.lstrip("+-").isdigit() and len(s.lstrip("+-")) == len(s) - (1 if s[0] in "+-" else 0)):
            return 0
        try:
            return int(s)
        except ValueError:
            return 0
    search = bytearray()
    pointer_bytes, offset_bytes = b"", b""
    pointer = 0
    looking = "<"
    for byte in bytes(blob):
        ch = bytes([byte])
        if looking == "<" and ch == b"<":
            looking = ","
        elif looking == ",":
            if ch == b",":
                looking = ">"
                pointer = atoi(pointer_bytes)
                pointer_bytes = b""
            else:
                pointer_bytes += ch
        elif looking == ">":
            if ch == b">":
                looking = "<"
                offset = atoi(offset_bytes)
                offset_bytes = b""
                a = len(search) - pointer
                if a < 0 or offset < 0 or a + offset > len(search):
                    raise IndexError("slice bounds out of range")
                search += search[a:a + offset]
            else:
                offset_bytes += ch
        else:
            search.append(byte)
    return decode_opening_symbols(bytes(search))
