"""CPU: the restated wire format (tests/bundle_ref.py) is self-consistent — write → parse is the identity,
the malformed-case table splits the way the serde rules say — and a bundle assembled from the oracle's
generator output verifies through the oracle after the JSON round trip."""
import numpy as np
import pytest

import bundle_cases
import bundle_ref
from tools.synth import Tipset

EXPECT_OK = {"good", "good pretty", "empty lists", "whitespace", "unknown top field", "unknown field first",
             "deep ignored 126", "deep ignored in block 124", "storage ok", "event ok", "storage twice",
             "storage actor u64 max", "storage epoch negative", "storage epoch i64 min", "storage extra field",
             "storage escaped strings", "event topics empty", "event parents empty", "event extra nested", "block ok",
             "block empty data", "block 1 pad", "block 2 pads", "block escaped ok", "block escaped slash", "block plus slash",
             "block extra", "block cid v0", "block cid sha256", "block cid identity empty", "block long"}


def test_case_table_split():
    got_ok = set()
    for name, text in bundle_cases.cases():
        try:
            bundle_ref.parse_bundle(text)
            got_ok.add(name)
        except bundle_ref.BundleError:
            pass
    assert got_ok == EXPECT_OK, (sorted(got_ok - EXPECT_OK), sorted(EXPECT_OK - got_ok))


def test_base64_strictness():
    for n in range(0, 70):
        raw = bytes((i * 37 + n) & 0xFF for i in range(n))
        import base64
        s = base64.b64encode(raw).decode()
        assert bundle_ref.b64decode_strict(s) == raw
        if s.endswith("="):
            with pytest.raises(bundle_ref.BundleError):
                bundle_ref.b64decode_strict(s.rstrip("="))


def test_round_trip_through_oracle(oracle):
    tip = Tipset(n_receipts=400, n_parents=2, n_planted=3, variety=1, max_events=4, n_actors=1200, n_contracts=4,
                 slots_per_contract=5, storage_layout_mix=1, n_actor_queries=6)
    st = oracle.store(tip.data, tip.off, tip.lens, tip.cids)
    s, trip, msg, wit = st.generate_event_proof(tip.parent_cids, tip.child_cid, tip.topic0, tip.topic1, actor=tip.filter_actor)
    assert s == 1
    want = {(int(e), int(v)) for e, v, _ in trip}
    sel = [i for i in range(len(tip.claim_exec)) if (int(tip.claim_exec[i]), int(tip.claim_event[i])) in want]
    ids = {tip.find_block(bytes(c[:38])) for c in wit}
    sidx = list(range(min(8, len(tip.sc_actor))))
    for i in sidx:
        s2, _, _, w2 = st.generate_storage_proof(tip.child_cid, int(tip.sc_actor[i]), tip.sc_slot[i].tobytes())
        assert s2 == 1
        ids |= {tip.find_block(bytes(c[:38])) for c in w2}
    st.close()
    blocks = sorted(((tip.cids[i, :38].tobytes(), tip.block(i)) for i in ids), key=lambda cb: cb[0][6:])
    text = bundle_ref.bundle_json(bundle_ref.storage_dicts(tip, sidx), bundle_ref.event_dicts(tip, sel), blocks)
    parsed = bundle_ref.parse_bundle(text)
    assert parsed["blocks"] == blocks
    assert parsed["event_proofs"] == bundle_ref.event_dicts(tip, sel)
    assert parsed["storage_proofs"] == bundle_ref.storage_dicts(tip, sidx)
    ev, sg = bundle_ref.claims_from_parsed(parsed)
    pst = oracle.store(*bundle_ref.tables_from_blocks(parsed["blocks"]))
    assert (pst.verify_event_proofs(ev, mode=0) == 1).all() and ev.n == len(sel) > 0
    assert (pst.verify_storage_proofs(sg, mode=0) == 1).all() and sg.n == len(sidx)
    pst.close()
