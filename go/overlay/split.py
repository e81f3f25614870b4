#!/usr/bin/env python3
"""Step 1 of INTEGRATION.md, done by a program instead of by prose: split the functions librsn replaces out of
compressor/lz/lzss.go and compressor/huffman/huffman.go of a go-compression/raisin checkout into `*_purego.go`
files (build tag `!rsn`), drop the cgo overlay files (`*_rsn.go`, tag `rsn`) next to them, and CHECK the result:

  * under either tag, every package-level identifier a file uses is defined exactly once among the files of that build
    (moving compressorWorker with CompressAsync left CompressRecursive, lzss.go:203, calling nothing);
  * no file keeps an import it no longer uses ("imported and not used" is a compile error in Go) and every
    selector `pkg.Name` a file uses has its import (the `sync` of lzss.go:11 goes with CompressAsync);
  * nothing is lost: every line of the original outside its import block is in exactly one of the two files.

What moves (and nothing else):
  package lz       CompressAsync (lzss.go:109-154), compressorWorkerAsync (:156-164), Compress (:224-316),
                   Decompress (:323-364).  compressorWorker (:166-184) STAYS: CompressRecursive (:189-220) calls it.
  package huffman  Compress (huffman.go:299-325), Decompress (:327-330).

No Go toolchain exists in the build image, so the check is an identifier scan, not a compile; it is exact for
the two files it is written for (no dot-imports, no shadowed package names at the use sites scanned).

    python go/overlay/split.py <raisin checkout> [--librsn <dir with librsn.so>] [--dry-run]
"""
import argparse
import os
import re
import shutil
import sys

HERE = os.path.dirname(os.path.abspath(__file__))

PLAN = {
    "lz": {"file": "compressor/lz/lzss.go", "purego": "lzss_purego.go", "overlay": "compressor/lz/lzss_rsn.go",
           "move": ["CompressAsync", "compressorWorkerAsync", "Compress", "Decompress"]},
    "huffman": {"file": "compressor/huffman/huffman.go", "purego": "huffman_purego.go", "overlay": "compressor/huffman/huffman_rsn.go",
                "move": ["Compress", "Decompress"]},
}


# ---------------------------------------------------------------------------------------------------- a little Go lexing
def blank_noncode(src):
    """src with comments, string, raw-string and rune literals replaced by spaces (newlines kept): what is left is code."""
    out, i, n = [], 0, len(src)
    while i < n:
        c = src[i]
        if src.startswith("//", i):
            j = src.find("\n", i)
            j = n if j < 0 else j
            out.append(" " * (j - i)); i = j
        elif src.startswith("/*", i):
            j = src.find("*/", i + 2)
            j = n if j < 0 else j + 2
            out.append(re.sub(r"[^\n]", " ", src[i:j])); i = j
        elif c == "`":
            j = src.find("`", i + 1)
            j = n if j < 0 else j + 1
            out.append(re.sub(r"[^\n]", " ", src[i:j])); i = j
        elif c in "\"'":
            j = i + 1
            while j < n and src[j] != c and src[j] != "\n":
                j += 2 if src[j] == "\\" else 1
            j = min(j + 1, n)
            out.append(" " * (j - i)); i = j
        else:
            out.append(c); i += 1
    return "".join(out)


def top_level_funcs(src):
    """{name: (start, end)} for package-level functions (no receiver); start includes the doc comment lines directly
    above, end is one past the closing brace's line."""
    code = blank_noncode(src)
    funcs = {}
    for m in re.finditer(r"^func\s+([A-Za-z_]\w*)\s*\(", code, flags=re.M):
        depth, j = 0, code.index("{", m.end())
        # the body's opening brace is the first one after the signature's parentheses close
        par, k = 0, m.end() - 1
        while True:
            if code[k] == "(":
                par += 1
            elif code[k] == ")":
                par -= 1
            elif code[k] == "{" and par == 0:
                j = k
                break
            k += 1
        k = j
        while True:
            if code[k] == "{":
                depth += 1
            elif code[k] == "}":
                depth -= 1
                if depth == 0:
                    break
            k += 1
        end = src.find("\n", k)
        end = len(src) if end < 0 else end + 1
        start = m.start()
        while True:                                          # pull in the doc comment
            prev_end = start - 1
            if prev_end <= 0:
                break
            prev_start = src.rfind("\n", 0, prev_end) + 1
            if src[prev_start:prev_end].lstrip().startswith("//"):
                start = prev_start
            else:
                break
        funcs[m.group(1)] = (start, end)
    return funcs


def import_block(src):
    m = re.search(r"^import\s*\((.*?)^\)[ \t]*\n", src, flags=re.M | re.S)
    if not m:
        raise SystemExit("no grouped import block found")
    entries = []
    for ln in m.group(1).splitlines():
        mm = re.match(r'\s*(?:([A-Za-z_.]\w*)\s+)?"([^"]+)"', ln)
        if mm:
            path = mm.group(2)
            name = mm.group(1) or path.rsplit("/", 1)[-1]
            entries.append((name, ln.strip()))
    return m.start(), m.end(), entries


def used_selectors(code_blanked):
    return set(re.findall(r"(?<![\w.])([A-Za-z_]\w*)\s*\.\s*[A-Za-z_]\w*", code_blanked))


def package_level_defs(src):
    """identifiers defined at package level: funcs without receiver, types, and names in var / const declarations."""
    code = blank_noncode(src)
    defs = []
    defs += re.findall(r"^func\s+([A-Za-z_]\w*)\s*\(", code, flags=re.M)
    defs += re.findall(r"^type\s+([A-Za-z_]\w*)", code, flags=re.M)
    defs += re.findall(r"^(?:var|const)\s+([A-Za-z_]\w*)", code, flags=re.M)
    for m in re.finditer(r"^(?:var|const)\s*\((.*?)^\)", code, flags=re.M | re.S):
        defs += re.findall(r"^\s*([A-Za-z_]\w*)", m.group(1), flags=re.M)
    return defs


def identifiers(code_blanked):
    return set(re.findall(r"(?<![\w.])([A-Za-z_]\w*)", code_blanked))


# ---------------------------------------------------------------------------------------------------- the split
def render_imports(entries):
    return "import (\n" + "".join("\t%s\n" % e for _, e in entries) + ")\n" if entries else ""


def split_source(src, move, pkg):
    funcs = top_level_funcs(src)
    missing = [f for f in move if f not in funcs]
    if missing:
        raise SystemExit("package %s: functions not found (already split?): %s" % (pkg, ", ".join(missing)))
    spans = sorted(funcs[f] for f in move)
    ib_start, ib_end, imports = import_block(src)
    moved, kept, pos = [], [], 0
    for a, b in spans:
        kept.append(src[pos:a]); moved.append(src[a:b]); pos = b
        if src[pos:pos + 1] == "\n" and kept[-1].endswith("\n\n"):   # the blank line after a moved function goes with it
            pos += 1
    kept.append(src[pos:])
    kept_src, moved_body = "".join(kept), "\n".join(moved)
    # imports: each file keeps exactly what its code selects from
    kb_start, kb_end, _ = import_block(kept_src)
    kept_code = blank_noncode(kept_src[:kb_start] + kept_src[kb_end:])
    kept_imports = [e for e in imports if e[0] in used_selectors(kept_code)]
    moved_imports = [e for e in imports if e[0] in used_selectors(blank_noncode(moved_body))]
    kept_out = kept_src[:kb_start] + render_imports(kept_imports) + kept_src[kb_end:]
    moved_out = ("//go:build !rsn\n// +build !rsn\n\n"
                 "// The pure-Go bodies that librsn replaces under `-tags rsn` (moved here unchanged by go/overlay/split.py).\n"
                 "package %s\n\n%s\n%s" % (pkg, render_imports(moved_imports), moved_body))
    return kept_out, moved_out


def check_build(files, orig_defs, orig_imports, label):
    """files: {name: source} of one build configuration of one package."""
    problems, defined = [], {}
    for name, src in files.items():
        for d in package_level_defs(src):
            if d in defined and d not in ("_",):
                problems.append("%s: %s defined in both %s and %s" % (label, d, defined[d], name))
            defined[d] = name
    for name, src in files.items():
        s, e, imps = import_block(src) if re.search(r"^import\s*\(", src, flags=re.M) else (0, 0, [])
        code = blank_noncode(src[:s] + src[e:])
        code = re.sub(r"^import\s+\"C\"\s*$", "", code, flags=re.M)
        sel = used_selectors(code)
        for iname, _ in imps:
            if iname not in sel:
                problems.append("%s: %s imports %s and does not use it" % (label, name, iname))
        have = {i for i, _ in imps} | ({"C"} if re.search(r'^import\s+"C"', src, flags=re.M) else set())
        for iname in orig_imports | {"runtime", "unsafe", "C"}:
            if iname in sel and iname not in have and not re.search(r"\b%s\s+\[\]byte|\b%s\s*:?=" % (iname, iname), code):
                problems.append("%s: %s uses %s. without importing it" % (label, name, iname))
        for ident in identifiers(code) & orig_defs:
            if ident not in defined:
                problems.append("%s: %s uses %s, which no file of this build defines" % (label, name, ident))
    return problems


def process(root, pkg, plan, overlay_dir, dry_run):
    path = os.path.join(root, plan["file"])
    src = open(path).read()
    kept, moved = split_source(src, plan["move"], pkg)
    overlay = open(os.path.join(overlay_dir, plan["overlay"])).read()
    _, _, imps = import_block(src)
    orig_defs, orig_imports = set(package_level_defs(src)), {i for i, _ in imps}
    base = os.path.basename(plan["file"])
    problems = []
    problems += check_build({base: kept, plan["purego"]: moved}, orig_defs, orig_imports, "%s, no tag" % pkg)
    problems += check_build({base: kept, os.path.basename(plan["overlay"]): overlay}, orig_defs, orig_imports, "%s, -tags rsn" % pkg)
    # the replaced entry points must exist, exported, under the tag as well
    odefs = set(package_level_defs(overlay))
    for f in plan["move"]:
        if f[0].isupper() and f not in odefs:
            problems.append("%s: the overlay does not define %s" % (pkg, f))
    # conservation: nothing but the import block changed
    ib_s, ib_e, _ = import_block(src)
    def body_lines(text):
        if re.search(r"^import\s*\(", text, flags=re.M):
            s, e, _ = import_block(text)
            text = text[:s] + text[e:]
        return [ln for ln in text.splitlines() if ln.strip() and not ln.startswith(("//go:build", "// +build", "package ", "// The pure-Go bodies"))]
    want = sorted(body_lines(src))
    got = sorted(body_lines(kept) + body_lines(moved))
    if want != got:
        problems.append("%s: lines were lost or duplicated by the split" % pkg)
    for text, name in ((kept, base), (moved, plan["purego"])):
        code = blank_noncode(text)
        if code.count("{") != code.count("}") or code.count("(") != code.count(")"):
            problems.append("%s: unbalanced braces in %s" % (pkg, name))
    if problems:
        return problems
    if not dry_run:
        d = os.path.dirname(path)
        open(path, "w").write(kept)
        open(os.path.join(d, plan["purego"]), "w").write(moved)
        shutil.copy(os.path.join(overlay_dir, plan["overlay"]), os.path.join(d, os.path.basename(plan["overlay"])))
    return []


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("checkout", help="root of a go-compression/raisin working tree")
    ap.add_argument("--librsn", help="directory holding librsn.so: copied with include/rsn.h to <checkout>/third_party/librsn/")
    ap.add_argument("--dry-run", action="store_true", help="check only, write nothing")
    a = ap.parse_args(argv)
    problems = []
    for pkg, plan in PLAN.items():
        problems += process(a.checkout, pkg, plan, HERE, a.dry_run)
    if problems:
        print("\n".join(problems), file=sys.stderr)
        return 1
    if a.librsn and not a.dry_run:
        tp = os.path.join(a.checkout, "third_party", "librsn")
        os.makedirs(os.path.join(tp, "include"), exist_ok=True)
        shutil.copy(os.path.join(a.librsn, "librsn.so"), tp)
        shutil.copy(os.path.join(HERE, "..", "..", "include", "rsn.h"), os.path.join(tp, "include"))
    print("split ok: build with `go build -tags rsn ./...` (librsn) or without the tag (pure Go, unchanged behaviour)")
    return 0


if __name__ == "__main__":
    sys.exit(main())
