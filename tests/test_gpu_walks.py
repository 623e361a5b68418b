"""GPU: path-walk primitives (K5 amt_get, K7 hamt_get) and the storage-proof verifier vs the CPU
oracle on seeded synthetic tipsets, bit-exact, through the C ABI."""
import numpy as np
import pytest

import claims
from tools.synth import Tipset

pytestmark = pytest.mark.gpu


def idaddr(i: int) -> bytes:
    b = bytearray([0])
    while True:
        c = i & 0x7F
        i >>= 7
        if i:
            b.append(c | 0x80)
        else:
            b.append(c)
            return bytes(b)


def loc_bytes(T, loc):
    out = []
    for l in loc:
        if l["block"] == 0xFFFFFFFF:
            out.append(b"")
        else:
            o = int(T.off[l["block"]]) + int(l["off"])
            out.append(T.data[o: o + int(l["len"])].tobytes())
    return out


@pytest.fixture(scope="module")
def tip():
    return Tipset(n_receipts=3000, n_planted=7, variety=1, n_actors=20000, n_contracts=18, slots_per_contract=40,
                  storage_layout_mix=1, n_actor_queries=500, keep_full_state=0)


@pytest.fixture(scope="module")
def both(tip, engine, oracle):
    w = engine.witness(tip.data, tip.off, tip.lens, tip.cids)
    st = oracle.store(tip.data, tip.off, tip.lens, tip.cids)
    yield w, st
    w.close()
    st.close()


def test_witness_cids_all_verify(tip, both):
    w, _ = both
    st, nbad = w.verify_cids()
    assert nbad == 0 and (st == 1).all()


def test_amt_get_receipts(tip, both):
    w, st = both
    n = tip.params["n_receipts"]
    idx = np.array(list(range(0, n, 7)) + [n - 1, n, n + 5, 8 ** 7, 2 ** 63, 2 ** 64 - 1], dtype=np.uint64)
    gs, gl = w.amt_get(tip.receipts_root, 0, "receipt", idx)
    os_, ov = st.amt_get(tip.receipts_root, 0, "receipt", idx)
    assert np.array_equal(gs, os_)
    assert loc_bytes(tip, gl) == ov
    assert (gs[: len(idx) - 6] == 1).all() and gs[-1] == 64


def test_amt_get_events_and_wrong_types(tip, both, oracle):
    w, st = both
    # find an events root through the oracle, then read events from both sides
    os_, ov = st.amt_get(tip.receipts_root, 0, "receipt", tip.claim_exec[:50])
    for k in range(0, 50, 5):
        rc = ov[k]
        assert rc[0] == 0x84
        link = rc[rc.index(b"\xd8\x2a\x58\x27\x00") + 5:][:38]
        idx = np.arange(0, 40, dtype=np.uint64)
        gs, gl = w.amt_get(link, 3, "stamped_event", idx)
        es, ev = st.amt_get(link, 3, "stamped_event", idx)
        assert np.array_equal(gs, es) and loc_bytes(tip, gl) == ev
        # opening with the wrong version / value type must fail the same way on both sides
        for ver, kind in ((0, "stamped_event"), (3, "receipt"), (3, "cid")):
            gs, _ = w.amt_get(link, ver, kind, idx[:3])
            es, _ = st.amt_get(link, ver, kind, idx[:3])
            assert np.array_equal(gs, es)
    # a CID that is not in the witness
    missing = oracle.cid_for_block(b"not there")
    gs, _ = w.amt_get(missing, 0, "receipt", [0, 1])
    assert gs.tolist() == [65, 65]


def test_hamt_get_actors(tip, both):
    w, st = both
    keys = [idaddr(int(i)) for i in tip.query_ids] + [b"", b"\x00", bytes(40)]
    gs, gl = w.hamt_get(tip.actors_root, 5, "actor_state", keys)
    os_, ov = st.hamt_get(tip.actors_root, 5, "actor_state", keys)
    assert np.array_equal(gs, os_)
    assert loc_bytes(tip, gl) == ov
    present = tip.query_present.astype(bool)
    assert (gs[: len(present)][present] == 1).all() and (gs[: len(present)][~present] == 32).all()
    # wrong value type / bit width: identical failure on both sides
    for bw, kind in ((5, "vec_u8"), (4, "actor_state"), (8, "actor_state"), (9, "actor_state"), (0, "actor_state")):
        gs, _ = w.hamt_get(tip.actors_root, bw, kind, keys[:20])
        os_, _ = st.hamt_get(tip.actors_root, bw, kind, keys[:20])
        assert np.array_equal(gs, os_), (bw, kind)


def test_storage_proofs_all_layouts(tip, both):
    w, st = both
    sc = claims.StorageClaims(tip)
    got = w.verify_storage_proofs(sc.arr, sc.n)
    want = st.verify_storage_proofs(sc, mode=1)
    assert np.array_equal(got, want)
    assert (got == 1).all()


def test_storage_proofs_adversarial(tip, both, oracle):
    w, st = both
    n = 40
    sc = claims.StorageClaims(tip, indices=np.arange(n))
    upper = lambda s: s.decode().upper().encode()  # noqa: E731
    sc.set_str(0, "value", "0x" + "00" * 32)                       # wrong value → FALSE_VALUE (unless truly zero)
    sc.set_str(1, "value", sc.arr[1].value.decode().upper().replace("0X", "0x"))  # hex case-insensitive → TRUE
    sc.set_str(2, "value", sc.arr[2].value.decode()[:-2])          # short value string → FALSE
    sc.set_str(3, "slot", sc.arr[3].slot.decode()[2:])             # slot without 0x is fine
    sc.set_str(4, "slot", "0x0x" + sc.arr[4].slot.decode()[2:])    # trim_start_matches strips repeats
    sc.set_str(5, "slot", "0x1234")                                # not 32 bytes → Err
    sc.set_str(6, "child_block_cid", "garbage")                    # Err
    sc.set_str(7, "parent_state_root", upper(sc.arr[7].parent_state_root))  # parses, but not canonical → FALSE
    sc.set_str(8, "actor_state_cid", "f" + tip.sc_actor_state[8][:38].tobytes().hex())  # base16 form → FALSE
    sc.set_str(9, "storage_root", claims.cid_str(oracle.cid_for_block(b"other")))       # wrong root → FALSE
    absent = [int(i) for i, pr in zip(tip.query_ids, tip.query_present) if not pr]
    sc.arr[10].actor_id = absent[0]                                # actor not found (its path is in the witness) → Err
    sc.arr[11].actor_id = int(tip.sc_actor[11]) + 1                # another actor's state → FALSE_ACTOR_STATE
    sc.set_str(12, "child_block_cid", claims.cid_str(oracle.cid_for_block(b"nohdr")))   # missing header → Err
    sc.set_str(13, "child_block_cid", claims.cid_str(tip.receipts_root))                # not a header → Err decode
    sc.arr[14].child_epoch = 5                                     # epoch is not checked by the storage verifier
    sc.set_str(15, "value", "0X" + sc.arr[15].value.decode()[2:])  # "0X" prefix compares equal ignoring case
    got = w.verify_storage_proofs(sc.arr, sc.n)
    want = st.verify_storage_proofs(sc, mode=0)
    assert np.array_equal(got, want), (got.tolist(), want.tolist())
    assert got[5] == 69 and got[6] == 69 and got[10] == 68 and got[12] == 65 and got[13] == 66
    assert got[7] == 18 and got[8] == 19 and got[9] == 20 and got[11] == 19
    assert got[1] == 1 and got[3] == 1 and got[4] == 1 and got[14] == 1 and got[15] == 1
    # trust policy: F3 epoch range
    tp = claims.TrustPolicy(kind=1, ec_chain_empty=0, min_epoch=tip.child_epoch + 1, max_epoch=tip.child_epoch + 9)
    got = w.verify_storage_proofs(sc.arr, sc.n, trust=tp)
    want = st.verify_storage_proofs(sc, trust=tp, mode=0)
    assert np.array_equal(got, want)
    assert got[0] == 3 and got[6] == 69 and got[14] == 3


def test_create_event_filter(engine, oracle):
    t0, t1 = engine.create_event_filter("NewTopDownMessage(bytes32,uint256)", "calib-subnet-1")
    assert t0 == oracle.keccak256(b"NewTopDownMessage(bytes32,uint256)")
    assert t1 == b"calib-subnet-1" + b"\0" * 18
    _, t1 = engine.create_event_filter("x", "a" * 40)
    assert t1 == b"a" * 32
