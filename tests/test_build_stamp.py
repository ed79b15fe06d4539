"""The library that is loaded was built from the sources of the tree the tests run in: libhsgpu.so is git-ignored and
travels to the GPU box beside the sources (tests/conftest.py builds it only where it is missing), so a stale one would
otherwise go unnoticed. csrc/Makefile stamps the sha256 of its sources into the library (hsgpu_source_hash)."""
import hashlib
import os
import re

import pytest

from hyperscan_amd import _native

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "hyperscan_amd", "csrc")


def tree_hash():
    mk = open(os.path.join(CSRC, "Makefile")).read()
    names = ["Makefile"]
    for var in ("SRCS", "HDRS"):
        names += re.search(r"^%s\s*:=\s*(.*)$" % var, mk, re.M).group(1).split()
    h = hashlib.sha256()
    for n in sorted(set(names)):  # make's $(sort ...): lexical, duplicates removed
        h.update(open(os.path.join(CSRC, n), "rb").read())
    return h.hexdigest()[:32]


def check():
    lib = _native.load_library()
    assert lib.hsgpu_source_hash().decode() == tree_hash(), "libhsgpu.so was not built from this tree: run __graft_entry__.build()"


def test_library_was_built_from_this_tree():
    check()


@pytest.mark.gpu
def test_library_was_built_from_this_tree_on_the_gpu_box():
    check()
