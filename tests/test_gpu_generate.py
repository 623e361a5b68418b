"""GPU: generator-side witness materialisation vs the CPU oracle's restatement of
`generate_event_proof` (src/proofs/events/generator.rs:75-178) and `generate_storage_proof`
(src/proofs/storage/generator.rs:29-69): same proofs, same message CIDs, same witness blocks in the same
`Cid: Ord` order — and the generated bundle must verify against its OWN pruned witness."""
import numpy as np
import pytest

import claims
from tools.synth import Tipset

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def tip():
    return Tipset(n_receipts=3000, n_parents=3, dup_permille=60, n_planted=7, variety=1, max_events=5,
                  no_events_permille=100, n_actors=3000, n_contracts=8, slots_per_contract=12, storage_layout_mix=1,
                  n_actor_queries=12)


@pytest.fixture(scope="module")
def both(tip, engine, oracle):
    w = engine.witness(tip.data, tip.off, tip.lens, tip.cids)
    st = oracle.store(tip.data, tip.off, tip.lens, tip.cids)
    yield w, st
    w.close()
    st.close()


def triples(m):
    return np.stack([m["exec_index"], m["event_index"], m["emitter"]], axis=1) if len(m) else np.zeros((0, 3), np.uint64)


@pytest.mark.parametrize("actor", ["filter", None])
def test_generate_event_proofs(tip, both, engine, oracle, actor):
    w, st = both
    a = tip.filter_actor if actor == "filter" else None
    gs, gm, gmsg, gids = w.generate_event_proofs(tip.parent_cids, tip.child_cid, tip.topic0, tip.topic1, actor=a)
    os_, otrip, omsg, owit = st.generate_event_proof(tip.parent_cids, tip.child_cid, tip.topic0, tip.topic1, actor=a)
    assert gs == os_ == 1
    assert len(gm) == len(otrip) > 0
    assert np.array_equal(triples(gm), otrip)
    assert np.array_equal(gmsg, omsg)
    # message CID of proof i is execution_order[exec_index]
    assert np.array_equal(gmsg, tip.exec_order[gm["exec_index"].astype(np.int64)])
    # the materialised witness: same blocks, same order
    assert np.array_equal(tip.cids[gids], owit)
    assert len(gids) < tip.n_blocks // 2  # it IS pruned
    # round trip: the generated proofs verify against the witness the generator materialised — and only
    # just: removing any single block of it breaks at least one proof
    sub = gids.astype(np.int64)
    pw = engine.witness(tip.data, tip.off[sub], tip.lens[sub], tip.cids[sub])
    ec = claims.EventClaims(tip, generated=(gm, gmsg))
    got = pw.verify_event_proofs(ec.arr, ec.n)
    assert (got == 1).all(), got.tolist()
    pw.close()
    pst = oracle.store(tip.data, tip.off[sub], tip.lens[sub], tip.cids[sub])
    assert (pst.verify_event_proofs(ec, mode=0) == 1).all()
    pst.close()


def test_generate_event_proofs_no_match(tip, both):
    w, st = both
    t0 = bytes(32)
    gs, gm, gmsg, gids = w.generate_event_proofs(tip.parent_cids, tip.child_cid, t0, tip.topic1)
    os_, otrip, omsg, owit = st.generate_event_proof(tip.parent_cids, tip.child_cid, t0, tip.topic1)
    assert gs == os_ == 1 and len(gm) == len(otrip) == 0
    # base witness + the message AMTs are still recorded
    assert np.array_equal(tip.cids[gids], owit) and len(gids) > 3 + len(tip.parent_cids)


def test_generate_event_proofs_errors(tip, engine, oracle):
    """Err paths: missing child header, missing receipts root, a parent that is not a header, a broken
    message AMT — the same status on both sides."""
    bogus = b"\x01\x71\xa0\xe4\x02\x20" + bytes(range(32))
    cases = [
        ("missing child", None, tip.parent_cids, bogus),
        ("child is not a header", None, tip.parent_cids, tip.receipts_root),
        ("parent is not a header", None, [tip.parent_cids[0], tip.receipts_root], tip.child_cid),
        ("missing parent", None, [bogus], tip.child_cid),
        ("no receipts root", tip.find_block(tip.receipts_root), tip.parent_cids, tip.child_cid),
    ]
    for name, drop, parents, child in cases:
        keep = np.ones(tip.n_blocks, dtype=bool)
        if drop is not None:
            keep[drop] = False
        idx = np.nonzero(keep)[0]
        w = engine.witness(tip.data, tip.off[idx], tip.lens[idx], tip.cids[idx])
        st = oracle.store(tip.data, tip.off[idx], tip.lens[idx], tip.cids[idx])
        gs, gm, _, gids = w.generate_event_proofs(parents, child, tip.topic0, tip.topic1)
        os_, otrip, _, owit = st.generate_event_proof(parents, child, tip.topic0, tip.topic1)
        assert gs == os_ and gs >= 64, (name, gs, os_)
        assert len(gm) == 0 and len(gids) == 0
        w.close()
        st.close()


def test_generate_storage_proofs(tip, both, engine, oracle):
    w, st = both
    ids = tip.sc_actor
    slots = tip.sc_slot
    assert len(ids) > 0
    out, wid = w.generate_storage_proofs(tip.child_cid, ids, slots)
    union = set()
    for i in range(len(ids)):
        os_, o3, oval, owit = st.generate_storage_proof(tip.child_cid, int(ids[i]), slots[i].tobytes())
        assert int(out["status"][i]) == os_, (i, int(out["status"][i]), os_)
        if os_ != 1:
            continue
        assert np.array_equal(out["parent_state_root"][i], o3[0])
        assert np.array_equal(out["actor_state_cid"][i], o3[1])
        assert np.array_equal(out["storage_root"][i], o3[2])
        assert np.array_equal(out["value"][i], oval)
        assert np.array_equal(out["value"][i], tip.sc_value[i])
        union |= {bytes(c) for c in owit}
        # the single-spec call materialises exactly the oracle's witness, in order
        _, wid1 = w.generate_storage_proofs(tip.child_cid, ids[i:i + 1], slots[i:i + 1])
        assert np.array_equal(tip.cids[wid1], owit)
    assert (out["status"] == 1).sum() > 0
    # absent actors → Err(actor not found), same as the oracle; their partial walks still record nothing extra
    got_union = [bytes(c) for c in tip.cids[wid]]
    assert got_union == sorted(union, key=lambda b: b[6:38])
    # round trip through the verifier with the pruned witness
    sub = wid.astype(np.int64)
    pw = engine.witness(tip.data, tip.off[sub], tip.lens[sub], tip.cids[sub])
    sc = claims.StorageClaims(tip)
    got = pw.verify_storage_proofs(sc.arr, sc.n)
    want = w.verify_storage_proofs(sc.arr, sc.n)
    assert np.array_equal(got, want)
    pw.close()


def test_generate_storage_absent_actor_and_slot(tip, both):
    w, st = both
    absent = [int(q) for q, p in zip(tip.query_ids, tip.query_present) if not p][:3]
    slot = bytes(31) + b"\x07"
    for aid in absent:
        out, wid = w.generate_storage_proofs(tip.child_cid, [aid], [np.frombuffer(slot, np.uint8)])
        os_, _, _, owit = st.generate_storage_proof(tip.child_cid, aid, slot)
        assert int(out["status"][0]) == os_ == 68  # "Actor not found"
    # a present contract, a slot nobody wrote: Ok with an all-zero value
    aid = int(tip.sc_actor[0])
    slot = bytes([0xEE] * 32)
    out, wid = w.generate_storage_proofs(tip.child_cid, [aid], [np.frombuffer(slot, np.uint8)])
    os_, o3, oval, owit = st.generate_storage_proof(tip.child_cid, aid, slot)
    assert int(out["status"][0]) == os_
    if os_ == 1:
        assert not out["value"][0].any() and not oval.any()
        assert np.array_equal(tip.cids[wid], owit)
