"""GPU: the pieces of the drop-in boundary beyond the verifiers: the witness as a
`Blockstore` (get / has / put_keyed, src/proofs/common/blockstore.rs:26-39), event locations for a host `check_event`
closure (src/proofs/events/verifier.rs:51-56,247-251), and `generate_proof_bundle` as one call
(src/proofs/generator.rs:25-95) — each against the oracle / the writer's records."""
import numpy as np
import pytest

from conftest import fuzz_seed

import claims
import ipc_filecoin_proofs_amd as ipcfp
from tools.synth import Tipset

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def tip():
    return Tipset(n_receipts=2500, n_parents=3, dup_permille=40, n_planted=9, variety=1, max_events=4, n_actors=2500,
                  n_contracts=6, slots_per_contract=10, storage_layout_mix=1, n_actor_queries=8, seed=fuzz_seed(515))


def test_blockstore_get_has(engine, tip):
    with engine.witness(tip.data, tip.off, tip.lens, tip.cids) as w:
        pick = [0, 1, tip.n_blocks // 2, tip.n_blocks - 1]
        cids = [tip.cids[i, :38].tobytes() for i in pick] + [bytes.fromhex("0171a0e40220") + bytes(32)]
        has, ids = w.has(cids)
        assert has.tolist() == [1, 1, 1, 1, 0] and ids[:4].tolist() == pick and ids[4] == 0xFFFFFFFF
        for i in pick:
            assert w.get(tip.cids[i, :38].tobytes()) == tip.block(i)
        assert w.get(cids[-1]) is None


def test_blockstore_put_keyed_adds_and_replaces(engine, oracle, tip):
    with engine.witness(tip.data, tip.off, tip.lens, tip.cids) as w:
        n0 = w.block_count
        fresh = [b"\x82\x01\x02", b"\x80", bytes(range(200)) * 3]
        fresh_cids = [oracle.cid_for_block(b) for b in fresh]
        w.put_keyed(fresh_cids, fresh)
        assert w.block_count == n0 + 3
        for c, b in zip(fresh_cids, fresh):
            assert w.get(c) == b
        st, nbad = w.verify_cids()
        assert nbad == 0 and len(st) == n0 + 3
        # everything that was there still resolves: the whole tipset verifies as before
        s, has, m, _ = w.scan_events(tip.receipts_root, tip.topic0, tip.topic1, actor=tip.filter_actor, want_touched=False)
        assert s == 1 and len(m) >= 9
        # put_keyed on an existing CID replaces the block, unhashed (MemoryBlockstore; K1 then reports the mismatch)
        victim = tip.cids[5, :38].tobytes()
        w.put_keyed([victim], [b"\x81\x07"])
        assert w.get(victim) == b"\x81\x07" and w.block_count == n0 + 4
        st, nbad = w.verify_cids()
        assert nbad == 1 and st[-1] == 0


def test_event_locations_feed_a_host_closure(engine, oracle, tip):
    ec = claims.EventClaims(tip)
    ec.arr[2].exec_index += 1
    with engine.witness(tip.data, tip.off, tip.lens, tip.cids) as w:
        plain = w.verify_event_proofs(ec.arr, ec.n)
        st, loc = w.verify_event_proofs_located(ec.arr, ec.n)
        assert np.array_equal(st, plain) and st[2] == 8
        ok = st == 1
        assert (loc["block"][ok] != 0xFFFFFFFF).all() and loc["block"][2] == 0xFFFFFFFF
        events = w.read_values(loc, stride=2048)
    # the closure: "emitter is even" — evaluated on the host over the located StampedEvent bytes
    want = st.copy()
    checked = 0
    for i in np.nonzero(ok)[0]:
        log = claims.extract_evm_log(events[i])
        assert log is not None and log[0] == int(tip.claim_emitter[i])
        assert [t for t in log[1]] == [tip.claim_topics[i, t].tobytes() for t in range(int(tip.claim_ntopics[i]))]
        if log[0] % 2:
            want[i] = ipcfp.ST.FALSE_FILTER
        checked += 1
    assert checked > 100 and (want == ipcfp.ST.FALSE_FILTER).sum() > 10
    # ... and the same fold executed END TO END behind the C ABI (ipcfp_verify_event_proofs_with): the library calls the
    # predicate for exactly the proofs that are otherwise TRUE, in proof order, with the located StampedEvent bytes
    seen = []

    def emitter_is_even(idx, raw):
        seen.append(idx)
        assert raw == events[idx]
        return claims.extract_evm_log(raw)[0] % 2 == 0

    with engine.witness(tip.data, tip.off, tip.lens, tip.cids) as w:
        got = w.verify_event_proofs_with(ec.arr, ec.n, emitter_is_even)
        none = w.verify_event_proofs_with(ec.arr, ec.n, lambda i, raw: False)
    assert np.array_equal(got, want)
    assert seen == np.nonzero(ok)[0].tolist()
    assert np.array_equal(none, np.where(ok, ipcfp.ST.FALSE_FILTER, st))


def test_generate_proof_bundle_is_the_union_in_cid_order(engine, oracle, tip):
    sig, subnet = "NewTopDownMessage(bytes32,uint256)", "calib-subnet-1"
    sspecs = [(int(tip.sc_actor[k]), tip.sc_slot[k].tobytes()) for k in (0, 3, 7)]
    with engine.witness(tip.data, tip.off, tip.lens, tip.cids) as w:
        t0, t1 = engine.create_event_filter(sig, subnet)
        assert (t0, t1) == (tip.topic0, tip.topic1)
        out = w.generate_proof_bundle(tip.parent_cids, tip.child_cid, sspecs, [(sig, subnet, tip.filter_actor), (sig, subnet, None)])
        bad = w.generate_proof_bundle(tip.parent_cids, tip.child_cid, sspecs[:1] + [(10 ** 9, bytes(32))], [(sig, subnet, None)])
    assert out["first_error"] is None and (out["storage"]["status"] == 1).all() and out["event_status"].tolist() == [1, 1]
    ost = oracle.store(tip.data, tip.off, tip.lens, tip.cids)
    union = set()
    n_proofs = 0
    for k, (a, slot) in enumerate(sspecs):
        s, o3, val, wit = ost.generate_storage_proof(tip.child_cid, a, slot)
        assert s == 1 and np.array_equal(o3[0], out["storage"]["parent_state_root"][k]) and np.array_equal(val, out["storage"]["value"][k])
        union |= {bytes(c) for c in wit}
    for j, actor in enumerate((tip.filter_actor, None)):
        s, trip, msg, wit = ost.generate_event_proof(tip.parent_cids, tip.child_cid, tip.topic0, tip.topic1, actor=actor)
        assert s == 1
        sel = out["match_spec"] == j
        assert np.array_equal(out["matches"]["exec_index"][sel], trip[:, 0]) and np.array_equal(out["message_cids"][sel], msg)
        n_proofs += len(trip)
        union |= {bytes(c) for c in wit}
    ost.close()
    assert len(out["matches"]) == n_proofs
    got = [bytes(tip.cids[i]) for i in out["block_ids"]]
    assert got == sorted(union) and len(set(got)) == len(got)  # chain CIDs of one type: `Cid: Ord` = byte order
    # a failing spec aborts the bundle where the reference's `?` would (storage specs come first)
    assert bad["first_error"] == 1 and bad["storage"]["status"][1] == ipcfp.ST.ERR_ACTOR_NOT_FOUND
