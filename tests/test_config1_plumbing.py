"""BASELINE.json configs[0] — "single AMT receipt-inclusion proof at index 0, depth 3, CPU reference path
(plumbing)", SURVEY.md §8(d) config 1: an Amtv0<Receipt> with 100 receipts (height 2 ⇒ 3 node levels), the proof
for the receipt at index 0, its verdict and the set of witness blocks it touches.  CPU: the oracle.  GPU: the
engine through the C ABI, same verdicts, same touched set."""
import numpy as np
import pytest

import claims
from tools.synth import Tipset


@pytest.fixture(scope="module")
def tip():
    # every receipt has events, so exec index 0 has a claim; one planted match puts a known event under the filter
    return Tipset(n_receipts=100, n_parents=1, dup_permille=0, n_planted=1, variety=0, max_events=2, no_events_permille=0)


def receipts_path_blocks(tip, index):
    """Walk the receipts AMT by hand (pure Python, bit width 3): the block ids on the path to `index`."""
    import pyamt  # noqa: F401  (same wire format; only the constants are needed here)

    def block_of(cid):
        return tip.find_block(cid)

    path = []
    b = block_of(tip.receipts_root)
    path.append(b)
    raw = tip.block(b)
    assert raw[0] == 0x83  # v0 root: [height, count, node]
    height, count = raw[1], raw[2] if raw[2] < 24 else raw[3]
    assert count == 100 and height == 2
    pos = 3 if raw[2] < 24 else 4
    node = raw[pos:]
    for h in range(height, 0, -1):
        assert node[0] == 0x83 and node[1] == 0x41  # [bitmap(1 byte), links, values]
        bitmap = node[2]
        sub = (index // (8 ** h)) % 8
        assert bitmap >> sub & 1
        rank = bin(bitmap & ((1 << sub) - 1)).count("1")
        links = node[4:]  # after the array header of the links (≤ 8 entries: one byte)
        cid = links[rank * 43 + 5: rank * 43 + 43]
        b = block_of(bytes(cid))
        path.append(b)
        node = tip.block(b)
    return path


def test_oracle_single_receipt_proof(tip, oracle):
    st = oracle.store(tip.data, tip.off, tip.lens, tip.cids)
    i0 = int(np.nonzero(tip.claim_exec == 0)[0][0])
    ec = claims.EventClaims(tip, indices=[i0])
    assert st.verify_event_proofs(ec, mode=0).tolist() == [1]
    ec.arr[0].exec_index = 1
    assert st.verify_event_proofs(ec, mode=0).tolist() != [1]
    # depth 3: root + two levels of nodes on the path to index 0, all distinct blocks of the witness
    path = receipts_path_blocks(tip, 0)
    assert len(path) == 3 and len(set(path)) == 3
    s, vals = st.amt_get(tip.receipts_root, 0, "receipt", [0, 99, 100])
    assert s.tolist() == [1, 1, 32] and vals[0][0] == 0x84
    st.close()


@pytest.mark.gpu
def test_engine_single_receipt_proof(tip, engine, oracle):
    w = engine.witness(tip.data, tip.off, tip.lens, tip.cids)
    st = oracle.store(tip.data, tip.off, tip.lens, tip.cids)
    i0 = int(np.nonzero(tip.claim_exec == 0)[0][0])
    ec = claims.EventClaims(tip, indices=[i0])
    assert w.verify_event_proofs(ec.arr, ec.n).tolist() == [1]
    ec.arr[0].exec_index = 1
    assert np.array_equal(w.verify_event_proofs(ec.arr, ec.n), st.verify_event_proofs(ec, mode=0))
    gs, gl = w.amt_get(tip.receipts_root, 0, "receipt", [0, 99, 100])
    assert gs.tolist() == [1, 1, 32]
    # the generator over the same tipset records the receipt path of every matching receipt: for the planted
    # one that is exactly the hand-walked path
    s, m, msg, ids = w.generate_event_proofs(tip.parent_cids, tip.child_cid, tip.topic0, tip.topic1, actor=tip.filter_actor)
    os_, otrip, omsg, owit = st.generate_event_proof(tip.parent_cids, tip.child_cid, tip.topic0, tip.topic1, actor=tip.filter_actor)
    assert s == os_ == 1 and len(m) == len(otrip) >= 1
    assert np.array_equal(tip.cids[ids], owit)
    for e in sorted(set(m["exec_index"].tolist())):
        assert set(receipts_path_blocks(tip, int(e))) <= set(ids.tolist())
    w.close()
    st.close()
