"""tools/hsbench.py: the reference benchmarker's input formats (pattern file,
SQLite corpus) and output lines. Parsing and corpus I/O run anywhere; the scan needs a GPU."""
import os
import re
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import hsbench  # noqa: E402


def test_expression_lines_parse_like_the_reference():
    # util/ExpressionParser.rl: ID:/regex/flags{ext}
    p = hsbench.parse_expression_line
    assert p("1:/abc/") == (1, b"abc", 0, {})
    assert p("7:/a\\/b/is") == (7, b"a\\/b", 3, {})
    assert p("9:/foo.*bar/HL8") == (9, b"foo.*bar", 8 | 256 | 32, {})
    assert p("3:/x/i{min_offset=4,max_offset=10}") == (3, b"x", 1, {"min_offset": 4, "max_offset": 10})
    assert p("4:/y/{min_length=2}") == (4, b"y", 0, {"min_length": 2})
    assert p("# comment") is None and p("   ") is None
    for bad in ("abc", "1:abc", "1:/abc", "1:/abc/z", "1:/abc/{bogus=1}"):
        with pytest.raises(hsbench.ParseError):
            p(bad)


def test_corpus_roundtrip_uses_the_reference_schema(tmp_path):
    import sqlite3

    rng = np.random.default_rng(4)
    lens = rng.integers(1, 300, 200)
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    corpus = rng.integers(0, 256, int(off[-1]), dtype=np.uint8)
    path = str(tmp_path / "c.db")
    hsbench.write_corpus(path, corpus, off, stream_of_block=np.arange(200) // 4)
    db = sqlite3.connect(path)
    cols = [r[1] for r in db.execute("pragma table_info(chunk)")]
    assert cols == ["id", "stream_id", "data"]  # scripts/CorpusBuilder.py:19-25
    assert db.execute("select count(*) from chunk").fetchone()[0] == 200
    db.close()
    c2, o2, n_streams = hsbench.read_corpus(path)
    assert np.array_equal(c2, corpus) and np.array_equal(o2, off) and n_streams == 50
    with pytest.raises(SystemExit):
        hsbench.write_corpus(path, corpus, off)  # CorpusBuilder refuses to overwrite


@pytest.mark.gpu
def test_hsbench_end_to_end(tmp_path):
    pats = tmp_path / "pats.txt"
    pats.write_text("# literals\n1:/needle/\n2:/Hay/i\n3:/stack/\n")
    corpus_db = str(tmp_path / "corpus.db")
    text = (b"needle in a haystack " * 50 + b"HAY hay Needle ") * 40
    lens = np.full(len(text) // 97, 97)
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    data = np.frombuffer(text[: int(off[-1])], dtype=np.uint8)
    hsbench.write_corpus(corpus_db, data, off)
    want = 0
    for b in range(off.size - 1):
        blk = bytes(data[int(off[b]):int(off[b + 1])])
        want += len(re.findall(b"(?=needle)", blk)) + len(re.findall(b"(?i)(?=hay)", blk)) + len(re.findall(b"(?=stack)", blk))
    env = dict(os.environ, PYTHONPATH=ROOT)
    # (the last two: the reference's own loop -- one hs_scan per block -- with a launch per call and through the small-batch server)
    for extra in ([], ["--literal-on"], ["--literal-on", "--resident"], ["--one-scan-per-block"], ["--one-scan-per-block", "--server", "1"],
                  ["--literal-on", "--one-scan-per-block", "--server", "2"]):
        out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "hsbench.py"), "-e", str(pats), "-c", corpus_db,
                              "-N", "-n", "3"] + extra, capture_output=True, text=True, env=env, timeout=300)
        assert out.returncode == 0, out.stdout + out.stderr
        lines = out.stdout.splitlines()
        for key in ("Signatures:", "Hyperscan info:", "Expression count:  3", "Bytecode size:", "Database CRC:",
                    "Scratch size:", "Compile time:", "Time spent scanning:", "Corpus size:", "Matches per iteration:",
                    "Overall block rate:", "Mean throughput (overall):", "Max throughput (per core):"):
            assert any(l.startswith(key) for l in lines), (key, out.stdout)
        m = re.search(r"Matches per iteration:\s+(\d+)", out.stdout)
        assert int(m.group(1)) == want, (extra, out.stdout)
        assert f"({off.size - 1} blocks)" in out.stdout
        assert "INCONSISTENT" not in out.stdout
        if "--server" in extra:
            m = re.search(r"Small-batch server:\s+(\d+) calls served, (\d+) server launches", out.stdout)
            assert m and int(m.group(1)) >= 3 * (off.size - 1) and int(m.group(2)) <= 3, out.stdout
