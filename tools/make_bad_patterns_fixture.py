#!/usr/bin/env python3
"""tests/golden/bad_patterns.json from the reference's unit/hyperscan/bad_patterns.txt
(`id:/pattern/flags{ext} #expected compile error`, read by unit/hyperscan/bad_patterns.cpp):
every line is a pattern hs_compile must refuse. Run here (needs /root/reference)."""
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rows = []
for line in open("/root/reference/unit/hyperscan/bad_patterns.txt", encoding="latin-1"):
    m = re.match(r"(\d+):/(.*)/([A-Za-z0-9]*)(\{[^}]*\})?\s*#(.*)$", line.rstrip("\n"))
    if not m:
        continue
    pid, pat, fl, ext, msg = m.groups()
    e = {}
    if ext:
        try:
            e = {k.strip(): int(v) for k, v in (kv.split("=") for kv in ext.strip("{}").split(","))}
        except ValueError:
            e = {"unparsed": ext}
    rows.append({"id": int(pid), "pattern_hex": pat.encode("latin-1").hex(), "flags": fl, "ext": e, "message": msg.strip()})
out = os.path.join(ROOT, "tests", "golden", "bad_patterns.json")
json.dump({"source": "unit/hyperscan/bad_patterns.txt", "rows": rows}, open(out, "w"), indent=0)
print(len(rows), "rows ->", out)
