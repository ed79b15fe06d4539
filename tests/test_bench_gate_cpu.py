"""bench.py's parity gate on the CPU: it must pass on the true records and refuse a bench line for every kind of
wrong output -- a missing record, an extra one, a wrong id, records out of delivery order. (The gate is the only
thing between a wrong kernel and a published number.)"""
import numpy as np
import pytest

import bench
from hyperscan_amd import corpus as cp
from tests import oracle_binding as ob


def _records(lits, corpus, off):
    want = ob.Oracle(lits).collect_blocks(corpus, off)
    # literal index of each record: ids are unique in the generated sets
    lit_of = {l.id: i for i, l in enumerate(lits)}
    recs = np.zeros((want.size, 4), dtype=np.uint32)
    recs[:, 0], recs[:, 1], recs[:, 2] = want["block"], want["end"], want["id"]
    recs[:, 3] = [lit_of[int(i)] for i in want["id"]]
    order = np.lexsort((recs[:, 3], recs[:, 1], recs[:, 0]))
    return np.ascontiguousarray(recs[order]), want


def test_parity_gate_accepts_the_truth_and_refuses_everything_else():
    lits = cp.teddy_literals(64, seed=2)
    corpus, off = cp.packet_corpus(1 << 20, lits, seed=3, match_every=2048)
    recs, want = _records(lits, corpus, off)
    kg = int(off.size - 1) // 2
    gate = (kg, want[want["block"] < kg])
    assert len(gate[1]) > 50
    assert "identical" in bench.parity_gate(recs, gate, lits)
    inside = np.flatnonzero(recs[:, 0] < kg)
    # a record missing
    with pytest.raises(AssertionError):
        bench.parity_gate(np.delete(recs, inside[len(inside) // 2], axis=0), gate, lits)
    # one too many (a duplicate: also breaks strict delivery order)
    with pytest.raises(AssertionError):
        bench.parity_gate(np.insert(recs, inside[3], recs[inside[3]], axis=0), gate, lits)
    # a wrong id
    bad = recs.copy()
    bad[inside[5], 2] ^= 1
    with pytest.raises(AssertionError):
        bench.parity_gate(bad, gate, lits)
    # a wrong end offset
    bad = recs.copy()
    bad[inside[7], 1] += 1
    with pytest.raises(AssertionError):
        bench.parity_gate(bad, gate, lits)
    # the right multiset in the wrong order
    bad = recs.copy()
    a, b = inside[10], inside[11]
    bad[[a, b]] = bad[[b, a]]
    with pytest.raises(AssertionError, match="delivery order"):
        bench.parity_gate(bad, gate, lits)
    # records beyond the gate's blocks are not its business
    tail = recs.copy()
    tail[recs[:, 0] >= kg, 2] ^= 7
    assert "identical" in bench.parity_gate(tail, gate, lits)


def test_whole_corpus_reference_gate_accepts_the_truth_and_refuses_a_single_wrong_record():
    """reference_gate_full (round 5: every block of the corpus against the compiled reference, on all host threads, range by
    range) -- the same refusals as the bounded gate, wherever in the corpus the damage sits"""
    lits = cp.teddy_literals(64, seed=2)
    corpus, off = cp.packet_corpus(4 << 20, lits, seed=5, match_every=2048)
    recs, want = _records(lits, corpus, off)
    msg = bench.reference_gate_full(lits, corpus, off, recs, "test")
    assert f"ALL {off.size - 1} blocks / {len(want)} matches" in msg
    for where in (0, len(recs) // 2, len(recs) - 1):
        bad = recs.copy()
        bad[where, 2] ^= 1
        with pytest.raises(AssertionError, match="PARITY FAILURE"):
            bench.reference_gate_full(lits, corpus, off, bad, "test")
        with pytest.raises(AssertionError):
            bench.reference_gate_full(lits, corpus, off, np.delete(recs, where, axis=0), "test")
    bad = recs.copy()
    a = len(recs) // 3
    bad[[a, a + 1]] = bad[[a + 1, a]]
    with pytest.raises(AssertionError):
        bench.reference_gate_full(lits, corpus, off, bad, "test")
