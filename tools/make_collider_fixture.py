#!/usr/bin/env python3
"""Builds tests/golden/collider_subset.json from the reference's hscollider regression corpus
(tools/hscollider/test_cases/{pcre,corpora}/*.txt: `id:/regex/flags{ext}` pattern lines;
corpus lines either `id="corpus": to, to, ...` -- the corpus WITH the end offsets the reference
must report, tools/hscollider/ColliderCorporaParser.rl:106-147 -- or the older `id:corpus`
without them; C-style escapes in both).

Kept: every pattern the hs_* facade accepts. For the lines that carry their expected end
offsets those ARE the golden values (kind "reference"). For the older lines hscollider would ask
libpcre, which this image does not have, so the expectation comes from Python's `re` (kind
"model") and only for patterns that mean the same in Python's dialect as in PCRE: for each end
offset whether some start matches, with the whole corpus visible (a lookahead pins the end), only
the first match under HS_FLAG_SINGLEMATCH. The same model is run over the "reference" lines too
and every disagreement is printed: it has to be empty for the model to be trusted.

Run here (needs /root/reference); the fixture travels with the repository. Nothing at test time
reads the reference."""
import glob
import json
import os
import re
import signal
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hyperscan_amd import hs  # noqa: E402

REF = "/root/reference/tools/hscollider/test_cases"
FLAGS = {"i": hs.HS_FLAG_CASELESS, "s": hs.HS_FLAG_DOTALL, "m": hs.HS_FLAG_MULTILINE, "H": hs.HS_FLAG_SINGLEMATCH,
         "L": hs.HS_FLAG_SOM_LEFTMOST, "O": 0, "V": hs.HS_FLAG_ALLOWEMPTY, "8": hs.HS_FLAG_UTF8, "P": hs.HS_FLAG_PREFILTER}
MAX_CORPUS = 600  # bytes; longer corpora are left out (the model is quadratic-ish)


def unescape(s):
    """corpus escapes (ColliderCorporaParser.rl:109-124): \\xHH, \\0 \\a \\e \\f \\n \\v \\r \\t, and
    backslash + non-alphanumeric = that character"""
    out, i = bytearray(), 0
    b = s.encode("latin-1")
    while i < len(b):
        c = b[i]
        if c == 0x5C and i + 1 < len(b):
            n = chr(b[i + 1])
            if n == "x" and i + 3 < len(b) and re.fullmatch(rb"[0-9a-fA-F]{2}", b[i + 2:i + 4]):
                out.append(int(b[i + 2:i + 4], 16)); i += 4; continue
            m = {"n": 10, "t": 9, "r": 13, "f": 12, "v": 11, "a": 7, "e": 27, "0": 0}.get(n)
            if m is not None:
                out.append(m); i += 2; continue
            if not n.isalnum():
                out.append(b[i + 1]); i += 2; continue
        out.append(c); i += 1
    return bytes(out)


def split_new_format(rest):
    """`corpus": 3, 4` after the opening quote -> (corpus bytes, [ends]); the corpus ends at the first
    unescaped quote"""
    i = 0
    while i < len(rest):
        if rest[i] == "\\":
            i += 2
            continue
        if rest[i] == '"':
            break
        i += 1
    tail = rest[i + 1:]
    if not tail.startswith(":"):
        return None
    ends = [int(x) for x in re.findall(r"\d+", tail[1:])]
    return unescape(rest[:i]), sorted(set(ends))


def python_dialect(pat):
    """the same language in Python's re, or None when the dialects differ"""
    if re.search(r"\{,|\\0\d|\\[1-9]|\\[eQEhHvVRNKGXCpPcgko]|\[:|\(\?[^:<#imsP-]|\(\?<[=!]|\\x\{|\(\*|\(\?[ims-]*x", pat):
        return None
    if re.search(r"(?<!^)\(\?[ims-]+\)", pat):  # options after the start
        return None
    py = pat.replace("\\Z", "(?=\\n?\\Z)").replace("\\z", "\\Z")
    py = re.sub(r"\(\?<([A-Za-z_]\w*)>", r"(?P<\1>", py)
    py = re.sub(r"\(\?'([A-Za-z_]\w*)'", r"(?P<\1>", py)
    try:
        re.compile(py.encode("latin-1"))
    except (re.error, OverflowError, RecursionError):
        return None
    return py


class Timeout(Exception):
    pass


def expected(py, flags, ext, data):
    rf = (re.I if flags & hs.HS_FLAG_CASELESS else 0) | (re.S if flags & hs.HS_FLAG_DOTALL else 0) | \
         (re.M if flags & hs.HS_FLAG_MULTILINE else 0)
    inline = ""
    m = re.match(r"\(\?[ims-]+\)", py)
    if m:  # keep leading options in front of the wrapper group
        inline, py = m.group(0), py[m.end():]
    out = []
    for to in range(len(data) + 1):
        rx = re.compile(("%s(?:%s)(?=[\\s\\S]{%d}\\Z)" % (inline, py, len(data) - to)).encode("latin-1"), rf)
        # leftmost start whose match can end at `to`; empty matches are never reported
        frm = None
        for mm in rx.finditer(data):
            if mm.end() == to and mm.start() < to:
                frm = mm.start()
                break
        if frm is None:  # finditer's own choice of end may hide a start: ask every start
            for f in range(to):
                mm = rx.match(data, f)
                if mm and mm.end() == to:
                    frm = f
                    break
        if frm is None:
            continue
        if "min_offset" in ext and to < ext["min_offset"]:
            continue
        if "max_offset" in ext and to > ext["max_offset"]:
            continue
        if "min_length" in ext and to - frm < ext["min_length"]:
            continue
        out.append([frm if flags & hs.HS_FLAG_SOM_LEFTMOST else 0, to])
        if flags & hs.HS_FLAG_SINGLEMATCH:
            break
    return out


def main():
    corpora = {}
    for f in sorted(glob.glob(REF + "/corpora/*.txt")):
        for line in open(f, encoding="latin-1"):
            line = line.rstrip("\n")
            line = line.strip()
            m = re.match(r'(\d+)="(.*)$', line)
            if m:
                sp = split_new_format(m.group(2))
                if sp:
                    corpora.setdefault(int(m.group(1)), []).append((sp[0], sp[1]))
                continue
            m = re.match(r"(\d+):(.*)$", line)
            if m:
                corpora.setdefault(int(m.group(1)), []).append((unescape(m.group(2)), None))
    cases, seen, dropped = [], 0, {"flags": 0, "facade": 0, "no_corpus": 0, "slow": 0}
    disagree = []
    signal.signal(signal.SIGALRM, lambda *_: (_ for _ in ()).throw(Timeout()))
    for f in sorted(glob.glob(REF + "/pcre/*.txt")):
        for line in open(f, encoding="latin-1"):
            line = line.rstrip("\n")
            m = re.match(r"(\d+):/(.*)/([A-Za-z0-9]*)(?:\{(.*)\})?$", line)
            if not m:
                continue
            seen += 1
            pid, pat, fl, exts = int(m.group(1)), m.group(2), m.group(3), m.group(4)
            if any(c not in FLAGS for c in fl):
                dropped["flags"] += 1
                continue
            flags = 0
            for c in fl:
                flags |= FLAGS[c]
            ext = {}
            if exts:
                try:
                    ext = {k: int(v) for k, v in (kv.split("=") for kv in exts.split(","))}
                except ValueError:
                    ext = {"bad": 1}
                if set(ext) - {"min_offset", "max_offset", "min_length"}:
                    dropped["flags"] += 1
                    continue
            try:
                hs.Database.compile_ext([pat], [flags], [pid], [hs.ExprExt.make(**ext) if ext else None])
            except hs.HsError:
                dropped["facade"] += 1
                continue
            py = None if flags & hs.HS_FLAG_UTF8 else python_dialect(pat)  # (the model is bytes-only)
            kept_c, kept_e, kinds = [], [], []
            try:
                signal.alarm(60)
                for data, ends in corpora.get(pid, []):
                    if len(data) > MAX_CORPUS:
                        continue
                    model = sorted({to for _f, to in expected(py, flags, ext, data)}) if py is not None else None
                    if ends is not None:
                        # (under SINGLEMATCH the file still lists every match: hscollider accepts any ONE
                        # of them, tools/hscollider/main.cpp:522-534; the model reports the first)
                        want = ends[:1] if flags & hs.HS_FLAG_SINGLEMATCH else ends
                        if model is not None and model != want:
                            disagree.append((pid, pat, fl, data, ends, model))
                        kept_c.append(data); kept_e.append(ends); kinds.append("reference")
                    elif model is not None:
                        kept_c.append(data); kept_e.append(model); kinds.append("model")
                signal.alarm(0)
            except Timeout:
                dropped["slow"] += 1
                continue
            if not kept_c:
                dropped["no_corpus"] += 1
                continue
            cases.append({"file": os.path.basename(f), "id": pid, "pattern": pat, "flags": "".join(c for c in fl if c != "O"),
                          "ext": ext, "corpora": [c.hex() for c in kept_c], "ends": kept_e, "kind": kinds})
    out = os.path.join(ROOT, "tests", "golden", "collider_subset.json")
    with open(out, "w") as fh:
        json.dump({"source": "tools/hscollider/test_cases/{pcre,corpora}/*.txt", "model": "python re, see tools/make_collider_fixture.py",
                   "cases": cases}, fh, separators=(",", ":"))
    n_corp = sum(len(c["corpora"]) for c in cases)
    n_ref = sum(k == "reference" for c in cases for k in c["kind"])
    n_match = sum(len(e) for c in cases for e in c["ends"])
    print(f"{seen} pattern lines, {len(cases)} kept ({n_corp} corpora, {n_ref} with the reference's own expectations, "
          f"{n_match} expected matches), dropped {dropped}")
    print(f"python model vs reference expectations: {len(disagree)} disagreements")
    for d in disagree[:40]:
        print("  ", d)
    print("wrote", out, os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main()
