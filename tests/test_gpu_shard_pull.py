"""GPU: the SELF-PLANNED shard — ipcfp_witness_create_shard_pull.  Rank r of G gets nothing but the bundle in host memory
(transport form: blocks back to back, lengths, digests) and the tipset key; the device follows the links level by level
and reads its blocks out of the host buffer itself.  The shard must be exactly what the planner that sees the whole
witness makes (ipcfp_shard_plan_tipset), and G such shards must give the unsharded engine's and the oracle's results.
Reference loops being cut: src/proofs/verifier.rs:19-28,49-54, src/proofs/events/verifier.rs:62-71."""
import numpy as np
import pytest

import ipc_filecoin_proofs_amd as ipcfp
from ipc_filecoin_proofs_amd import shard
from tools.synth import Tipset

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def tip():
    return Tipset(n_receipts=40_000, n_parents=4, dup_permille=30, n_planted=40, max_events=4, no_events_permille=50,
                  variety=1, seed=77)


@pytest.fixture(scope="module")
def bundle(tip):
    pk = ipcfp.PackedWitnessTables(tip.data, tip.off, tip.lens, tip.cids, ingest=True)
    ipcfp.host_register(pk.data)   # an ingest buffer is registered once, when it is made
    yield pk
    ipcfp.host_unregister(pk.data)


@pytest.fixture(scope="module")
def claims_packed(tip):
    ts, cl, blob, blob_len = ipcfp.pack_event_claims(
        tip.parent_cids, tip.child_cid, tip.parent_epoch, tip.child_epoch, tip.claim_exec, tip.claim_event,
        tip.claim_emitter, tip.exec_order[tip.claim_exec.astype(np.int64)], tip.claim_ntopics, tip.claim_topics,
        tip.claim_datalen, tip.claim_data)
    n = len(cl)
    liars = np.arange(5, n, 17)
    cl["exec_index"][liars[0::3]] += 1
    cl["emitter"][liars[1::3]] ^= 1
    with_data = liars[2::3][cl["data_len"][liars[2::3]] > 0]
    blob[cl["data_off"][with_data]] ^= 0x40
    cl["exec_index"][n - 2] = 40_000 + 3          # names no receipt: the last shard's
    return ts, cl, blob, blob_len


@pytest.mark.parametrize("G", [1, 2, 3, 8])
def test_pulled_shard_holds_exactly_the_planned_blocks(engine, tip, bundle, G):
    with engine.witness(tip.data, tip.off, tip.lens, tip.cids) as full:
        for r in range(G):
            st, lo, hi, n_receipts, ids = full.shard_plan_tipset(tip.parent_cids, tip.child_cid, G, r)
            pst, w, plo, phi, pn, stats = engine.witness_shard_pull(bundle, tip.parent_cids, tip.child_cid, G, r)
            assert st == pst == 1 and (lo, hi, n_receipts) == (plo, phi, pn) and w.receipt_range == (lo, hi if r + 1 < G else 2 ** 64 - 1)  # (the last shard: everything from lo on)
            assert w.block_count == len(ids) == stats["blocks"]
            has, _ = w.has([bytes(tip.cids[i]) for i in ids[:: max(1, len(ids) // 4000)]])
            assert has.all()
            # what crossed PCIe: the tables of the bundle + the shard's own blocks (each on lines of its own)
            assert stats["table_bytes"] == 36 * tip.n_blocks
            want_bytes = int(((tip.lens[ids].astype(np.int64) + 127) // 128 * 128).sum())
            assert stats["block_bytes"] == want_bytes and stats["rounds"] >= 5
            assert stats["payload_bytes"] == int(tip.lens[ids].astype(np.int64).sum())
            st_cid, n_bad = w.verify_cids()
            assert n_bad == 0
            w.close()


@pytest.mark.parametrize("G", [1, 2, 3, 8])
def test_pulled_shards_equal_the_unsharded_result(engine, oracle, tip, bundle, claims_packed, G):
    ts, cl, blob, blob_len = claims_packed
    with engine.witness(tip.data, tip.off, tip.lens, tip.cids) as w:
        want = w.verify_event_claims(ts, cl, blob, blob_len)
        ws, whas, wm, _ = w.scan_events(tip.receipts_root, tip.topic0, tip.topic1, actor=tip.filter_actor, want_touched=False)
    status = np.full(len(cl), 255, dtype=np.uint8)
    has = np.zeros(40_000, dtype=np.uint8)
    n_matches = 0
    for r in range(G):
        s = shard.TipsetShard.from_pull(engine, bundle, tip.parent_cids, tip.child_cid, tip.receipts_root, G, r)
        assert s.n_receipts_total == 40_000
        s.route(ts, cl, blob, blob_len)
        status[s.positions.astype(np.int64)] = s.witness.verify_event_claims(ts, s.claims, s.blob, s.blob_len)
        sst, shas, sm, _ = s.witness.scan_events(tip.receipts_root, tip.topic0, tip.topic1, actor=tip.filter_actor,
                                                 want_touched=False)
        assert sst == 1 and len(shas) == s.hi - s.lo
        has[s.lo: s.hi] = shas
        n_matches += len(sm)
        s.close()
    assert np.array_equal(status, want) and (want != 1).sum() > 100
    assert np.array_equal(has, whas) and n_matches == len(wm)
    ost = oracle.store(tip.data, tip.off, tip.lens, tip.cids, threads=0)
    assert np.array_equal(ost.verify_event_claims_packed(ts, cl, blob, threads=0), want)
    ost.close()


@pytest.mark.parametrize("G", [2, 8])
def test_claim_slices_of_a_sorted_batch_equal_the_routed_claims(engine, tip, bundle, claims_packed, G):
    """A batch in exec_index order (what generate_event_proof emits) is cut by two binary searches, no host pass; the
    offsets are rebased on the device: same statuses as ipcfp_route_event_claims + ipcfp_verify_event_claims."""
    ts, cl, blob, blob_len = claims_packed
    cs = np.ascontiguousarray(cl[np.argsort(cl["exec_index"], kind="stable")])  # (the liars moved exec_index by one: sort again)
    with engine.witness(tip.data, tip.off, tip.lens, tip.cids) as w:
        want = w.verify_event_claims(ts, cs, blob, blob_len)
    status = np.full(len(cs), 255, dtype=np.uint8)
    for r in range(G):
        s = shard.TipsetShard.from_pull(engine, bundle, tip.parent_cids, tip.child_cid, tip.receipts_root, G, r)
        a, st = s.witness.verify_event_claims_range(ts, cs, blob, blob_len, s.lo, s.hi, r == G - 1)
        status[a: a + len(st)] = st
        s.close()
    assert np.array_equal(status, want) and (want != 1).sum() > 100
    with engine.witness(tip.data, tip.off, tip.lens, tip.cids) as w, pytest.raises(ipcfp.EngineError, match="exec_index order"):
        w.verify_event_claims_range(ts, np.ascontiguousarray(cs[::-1]), blob, blob_len, 100, 200, True)
    # a slice whose blob window does not hold a record's bytes: ERR_BAD_CLAIM (66) for that record, the rest unharmed
    with engine.witness(tip.data, tip.off, tip.lens, tip.cids) as w:
        bad = cs.copy()
        k = int(np.nonzero(bad["data_len"][len(bad) // 2:] > 0)[0][0]) + len(bad) // 2
        lo, hi = int(bad["exec_index"][k - 5]), int(bad["exec_index"][k + 5]) + 1
        a, st = w.verify_event_claims_range(ts, bad, blob, blob_len, lo, hi, False)
        assert np.array_equal(st, want[a: a + len(st)])
        # offsets beyond what the blob holds: the window is clamped to the blob, the record is refused
        bad["data_off"][k] = blob_len + 1000
        a, st = w.verify_event_claims_range(ts, bad, blob, blob_len, lo, hi, False)
        i = k - a
        assert st[i] == 69 and np.array_equal(np.delete(st, i), np.delete(want[a: a + len(st)], i))


def test_claim_slices_over_a_blob_in_another_order(engine, tip, claims_packed):
    """A large slice's blob window is GUESSED from the slice's two ends so that records and window can cross PCIe beside
    the walks; a blob that is not in claim order makes the guess miss — the rebase kernel says so and the call repeats
    with the exact window (one reduction on the device): the statuses are the unsharded call's either way."""
    ts, cl, blob, blob_len = claims_packed
    cs = np.ascontiguousarray(cl[np.argsort(cl["exec_index"], kind="stable")])
    rng = np.random.default_rng(5)
    blob2 = np.zeros(len(blob) + 64, dtype=np.uint8)
    pos = 7  # (offsets need no alignment)
    for i in rng.permutation(len(cs)):
        nt, dl = int(cs["n_topics"][i]), int(cs["data_len"][i])
        if nt:
            o = int(cs["topics_off"][i])
            blob2[pos: pos + 33 * nt] = blob[o: o + 33 * nt]
            cs["topics_off"][i] = pos
            pos += 33 * nt
        if dl:
            o = int(cs["data_off"][i])
            blob2[pos: pos + dl] = blob[o: o + dl]
            cs["data_off"][i] = pos
            pos += dl
    assert pos <= len(blob2)
    with engine.witness(tip.data, tip.off, tip.lens, tip.cids) as w:
        want = w.verify_event_claims(ts, cs, blob2, pos)
        assert (want == 1).sum() > 20_000 and (want != 1).sum() > 100
        n = len(cs)
        assert n * cs.dtype.itemsize // 2 >= 1 << 20  # (half the batch is still a slice the guess is made for)
        for lo, hi, last in ((0, int(cs["exec_index"][n // 2]), False), (int(cs["exec_index"][n // 2]), 1 << 40, True),
                             (int(cs["exec_index"][n // 3]), int(cs["exec_index"][n // 3 + 50]), False)):
            a, st = w.verify_event_claims_range(ts, cs, blob2, pos, lo, hi, last)
            assert len(st) > 40 and np.array_equal(st, want[a: a + len(st)])


def test_pull_of_tall_event_amts(engine):
    """Events AMTs of bit width 1 and height >= 8: every node of every tree of the rank's receipts is a level of its own."""
    tall = Tipset(n_receipts=60, n_planted=6, variety=1, max_events=700, events_bit_width=1, seed=79)
    pk = ipcfp.PackedWitnessTables(tall.data, tall.off, tall.lens, tall.cids, ingest=True)
    ipcfp.host_register(pk.data)
    try:
        with engine.witness(tall.data, tall.off, tall.lens, tall.cids) as full:
            for G in (1, 3):
                for r in range(G):
                    st, lo, hi, nr, ids = full.shard_plan_tipset(tall.parent_cids, tall.child_cid, G, r)
                    pst, w, plo, phi, pn, stats = engine.witness_shard_pull(pk, tall.parent_cids, tall.child_cid, G, r)
                    assert pst == st == 1 and (plo, phi) == (lo, hi) and w.block_count == len(ids) and stats["rounds"] >= 12
                    has, _ = w.has([bytes(tall.cids[i]) for i in ids])
                    assert has.all()
                    w.close()
    finally:
        ipcfp.host_unregister(pk.data)


def test_pull_without_a_receipts_root_has_no_range_to_cut_by(engine, tip):
    keep = np.ones(tip.n_blocks, dtype=bool)
    with engine.witness(tip.data, tip.off, tip.lens, tip.cids) as w:
        _, ids = w.has([tip.child_cid])
        keep[int(ids[0])] = False
    sub = ipcfp.witness_cut_host(tip.data, tip.off, tip.lens, tip.cids, np.nonzero(keep)[0].astype(np.uint32))
    pk = ipcfp.PackedWitnessTables(*sub, ingest=True)
    ipcfp.host_register(pk.data)
    try:
        st, w, lo, hi, nr, stats = engine.witness_shard_pull(pk, tip.parent_cids, tip.child_cid, 4, 2)
        assert st == 65 and w is None and (lo, hi, nr) == (0, 0, 0)
    finally:
        ipcfp.host_unregister(pk.data)


def test_pull_refuses_a_buffer_the_device_cannot_read(engine, tip):
    pk = ipcfp.PackedWitnessTables(tip.data.copy(), tip.off, tip.lens, tip.cids)   # pageable, not registered
    with pytest.raises(ipcfp.EngineError, match="device-readable"):
        engine.witness_shard_pull(pk, tip.parent_cids, tip.child_cid, 2, 0)
    with pytest.raises(ipcfp.EngineError, match="page boundary"):          # an ingest buffer owns its pages (include/ipcfp.h)
        ipcfp.host_register(pk.data[1:])
    pk = ipcfp.PackedWitnessTables(tip.data, tip.off, tip.lens, tip.cids, ingest=True)
    pk.lens = pk.lens.copy()
    pk.lens[7] += 1                                                         # tables that do not add up
    ipcfp.host_register(pk.data)
    try:
        with pytest.raises(ipcfp.EngineError, match="add up"):
            engine.witness_shard_pull(pk, tip.parent_cids, tip.child_cid, 2, 0)
    finally:
        ipcfp.host_unregister(pk.data)


# ---- the frontier's capacity is the documented defence against hostile DAGs, and the frontier counts tree
# POSITIONS while its capacity is sized by the bundle's BLOCKS ----
def _receipt_event_links(data: np.ndarray):
    """offsets of the 38 CID bytes of every `Receipt [0, h'', gas, link]` in the witness bytes (the synthetic writer's form)"""
    import re

    ret = b"|".join(re.escape(bytes([0x40 + k])) + (b".{%d}" % k if k else b"") for k in range(24))   # return_data: bytes(0..23)
    pat = re.compile(rb"\x84\x00(?:" + ret + rb")(?:\x1a.{4}|\x19.{2}|\x18.|[\x00-\x17])\xd8\x2a\x58\x27\x00", re.S)
    return [m.end() for m in pat.finditer(data.tobytes())]


def test_receipts_that_share_one_events_root(engine, oracle):
    """Identical events ⇒ identical events root; a bundle keeps ONE copy of it.  The events-root level of the walk then has
    one position per receipt (20 000) in a bundle of far fewer blocks: emitted once, expanded once, and the shard is what
    the planner over the whole witness makes — not `E_UNSUPPORTED: not a tree of this bundle`."""
    t = Tipset(n_receipts=20_000, n_parents=2, n_planted=10, max_events=3, no_events_permille=0, variety=0, seed=91)
    data = t.data.copy()
    at = _receipt_event_links(data)
    assert len(at) > 19_000          # (a few receipts carry longer return data or no events: they keep their own root)
    shared = data[at[0]: at[0] + 38].copy()
    for a in at:
        data[a: a + 38] = shared
    with engine.witness(data, t.off, t.lens, t.cids) as full:
        st, lo, hi, nr, ids = full.shard_plan_tipset(t.parent_cids, t.child_cid, 1, 0)
        assert st == 1 and nr == 20_000
    sub = ipcfp.witness_cut_host(data, t.off, t.lens, t.cids, ids)
    assert len(sub[2]) + 1024 < len(at)         # fewer blocks (+ the frontier's slack) than receipts that share the root
    pk = ipcfp.PackedWitnessTables(*sub, ingest=True)
    ipcfp.host_register(pk.data)
    try:
        with engine.witness(*sub) as full:
            want_scan = full.scan_events(t.receipts_root, t.topic0, t.topic1, actor=t.filter_actor, want_touched=False)
            has_all = np.zeros(20_000, dtype=np.uint8)
            for G in (1, 2, 5):
                for r in range(G):
                    st, lo, hi, nr, ids = full.shard_plan_tipset(t.parent_cids, t.child_cid, G, r)
                    pst, w, plo, phi, pn, stats = engine.witness_shard_pull(pk, t.parent_cids, t.child_cid, G, r)
                    assert pst == st == 1 and (plo, phi, pn) == (lo, hi, nr) and w.block_count == len(ids)
                    has, _ = w.has([bytes(sub[3][i]) for i in ids[:: max(1, len(ids) // 2000)]])
                    assert has.all()
                    sst, shas, sm, _ = w.scan_events(t.receipts_root, t.topic0, t.topic1, actor=t.filter_actor, want_touched=False)
                    assert sst == 1
                    if G == 5:
                        has_all[lo:hi] = shas
                    w.close()
            assert want_scan[0] == 1 and np.array_equal(has_all, want_scan[1])
        ost = oracle.store(*sub)
        os_, ohas, _, _ = ost.scan_events(t.receipts_root, t.topic0, t.topic1, actor=t.filter_actor)
        ost.close()
        assert os_ == 1 and np.array_equal(ohas, has_all)
    finally:
        ipcfp.host_unregister(pk.data)


def test_a_dag_that_outgrows_the_frontier_is_refused_not_followed(engine, tip):
    """Three receipts-AMT nodes whose eight links all name the next one (A → B → C → A: nothing re-hashes a witness block, so
    a bundle may say so) under a root of height 7: 8^k tree positions at level k from a handful of blocks.  The walk must stop
    at the frontier's capacity — `IPCFP_E_UNSUPPORTED`, no read or write past the buffers, rounds queued ahead of the host
    included — and the engine must be whole afterwards."""
    from pyamt import array, bstr, link, uint

    small = Tipset(n_receipts=300, n_parents=2, n_planted=3, max_events=2, variety=0, seed=92)
    at = _receipt_event_links(small.data)
    victims = [small.find_block(small.data[a: a + 38].tobytes()) for a in (at[0], at[100], at[200])]
    assert len(set(victims)) == 3 and min(victims) >= 0
    cid = [small.cids[v, :38].tobytes() for v in victims]

    def node(child):
        return array([bstr(b"\xff"), array([link(child)] * 8), array([])])

    new = {victims[0]: node(cid[1]), victims[1]: node(cid[2]), victims[2]: node(cid[0]),
           small.find_block(small.receipts_root): array([uint(7), uint(8 ** 8), node(cid[0])])}
    chunks, off, lens = [small.data], small.off.copy(), small.lens.copy()
    end = int(small.data.size)
    for i, b in new.items():
        chunks.append(np.frombuffer(b, dtype=np.uint8))
        off[i], lens[i] = end, len(b)
        end += len(b)
    data = np.concatenate(chunks)
    # (transport form wants the blocks back to back: cut "all of them" in table order)
    sub = ipcfp.witness_cut_host(data, off, lens, small.cids, np.arange(small.n_blocks, dtype=np.uint32))
    pk = ipcfp.PackedWitnessTables(*sub, ingest=True)
    ipcfp.host_register(pk.data)
    try:
        for G, r in ((1, 0), (2, 1)):
            with pytest.raises(ipcfp.EngineError, match="outgrew"):
                engine.witness_shard_pull(pk, small.parent_cids, small.child_cid, G, r)
    finally:
        ipcfp.host_unregister(pk.data)
    # the context is unharmed: an honest pull right after
    pk2 = ipcfp.PackedWitnessTables(tip.data, tip.off, tip.lens, tip.cids, ingest=True)
    ipcfp.host_register(pk2.data)
    try:
        pst, w, lo, hi, nr, stats = engine.witness_shard_pull(pk2, tip.parent_cids, tip.child_cid, 2, 1)
        assert pst == 1 and nr == 40_000 and (lo, hi) == (20_000, 40_000)
        _, n_bad = w.verify_cids()
        assert n_bad == 0
        w.close()
    finally:
        ipcfp.host_unregister(pk2.data)


def test_claim_range_refuses_a_batch_that_is_out_of_order_in_the_middle(engine, tip, bundle, claims_packed):
    """ipcfp_verify_event_claims_range finds its slice by binary search; a batch whose order breaks INSIDE the slice (the
    ends still look sorted) would route claims to the wrong shard silently.  The order is checked on the device, over the
    whole slice, where the records are anyway: IPCFP_E_INVALID."""
    ts, cl, blob, blob_len = claims_packed
    order = np.argsort(cl["exec_index"], kind="stable")
    cs = cl[order].copy()
    pst, w, lo, hi, nr, stats = engine.witness_shard_pull(bundle, tip.parent_cids, tip.child_cid, 2, 0)
    assert pst == 1
    try:
        a, st = w.verify_event_claims_range(ts, cs, blob, blob_len, lo, hi, False)
        assert len(st) > 1000
        bad = cs.copy()
        mid = a + len(st) // 2
        bad["exec_index"][mid] = bad["exec_index"][mid + 40]      # one record ahead of its place: ends untouched
        with pytest.raises(ipcfp.EngineError, match="exec_index order"):
            w.verify_event_claims_range(ts, bad, blob, blob_len, lo, hi, False)
        bad = cs.copy()
        bad["exec_index"][mid] = hi + 5                            # a record of the OTHER shard inside this slice
        with pytest.raises(ipcfp.EngineError, match="exec_index order"):
            w.verify_event_claims_range(ts, bad, blob, blob_len, lo, hi, False)
        a2, st2 = w.verify_event_claims_range(ts, cs, blob, blob_len, lo, hi, False)   # and the honest batch again
        assert a2 == a and np.array_equal(st, st2)
    finally:
        w.close()


def test_a_receipts_root_whose_count_lies_the_last_shard_owns_what_lies_beyond_it(engine):
    """`Amtv0::load` checks the root's count against nothing and `get(i)` answers for any i the height allows, so a witness
    whose receipts root says 572 over 700 receipts verifies every claim unsharded.  The ranges are cut on the count; whatever
    lies beyond it is the LAST shard's (as a claim beyond the count is): its pull, its plan and its enumeration run to the
    end of the tree.  Found by tools/gpu_fuzz_seeds.sh, seed 1010 (one flipped bit of the count)."""
    tip = Tipset(n_receipts=700, n_parents=3, dup_permille=80, n_planted=9, variety=1, max_events=5, no_events_permille=60, seed=5151)
    n_rcpt = tip.params["n_receipts"]
    # the root block: [height, count, node] — find it by its CID
    cids = [bytes(c) for c in tip.cids]
    root = bytes(tip.receipts_root)
    b = [c[: len(root)] for c in cids].index(root)
    o = int(tip.off[b])
    head = bytes(tip.data[o: o + 8])
    assert head[0] == 0x83, head.hex()
    data = tip.data.copy()
    if head[2] == 0x19:      # a two-byte count: clear its second-highest bit (n → n - 128 for 700: 572)
        assert int.from_bytes(head[3:5], "big") == n_rcpt
        lie = n_rcpt - (1 << (n_rcpt.bit_length() - 3))
        data[o + 3: o + 5] = np.frombuffer(lie.to_bytes(2, "big"), dtype=np.uint8)
    else:
        pytest.skip("the tipset's receipt count is not a two-byte integer")
    ts, cl, blob, blob_len = ipcfp.pack_event_claims(
        tip.parent_cids, tip.child_cid, tip.parent_epoch, tip.child_epoch, tip.claim_exec, tip.claim_event,
        tip.claim_emitter, tip.exec_order[tip.claim_exec.astype(np.int64)], tip.claim_ntopics, tip.claim_topics,
        tip.claim_datalen, tip.claim_data)
    with engine.witness(data, tip.off, tip.lens, tip.cids) as w:
        want = w.verify_event_claims(ts, cl, blob, blob_len)
        ws, whas, wm, _ = w.scan_events(tip.receipts_root, tip.topic0, tip.topic1, actor=tip.filter_actor, want_touched=False)
        assert (want[cl["exec_index"] >= lie] == 1).any() and ws == 1
        for G in (2, 3):
            plans = [w.shard_plan_tipset(tip.parent_cids, tip.child_cid, G, r) for r in range(G)]
            pk = ipcfp.PackedWitnessTables(data, tip.off, tip.lens, tip.cids, ingest=True)
            ipcfp.host_register(pk.data)
            try:
                status = np.full(len(cl), 255, dtype=np.uint8)
                has = np.zeros(n_rcpt, dtype=np.uint8)
                n_matches = 0
                for r in range(G):
                    st, sw, lo, hi, nr, _ = engine.witness_shard_pull(pk, tip.parent_cids, tip.child_cid, G, r)
                    assert st == 1 and nr == lie and (lo, hi) == ipcfp.shard_range(lie, G, r)
                    pst, plo, phi, pnr, pids = plans[r]
                    assert pst == 1 and (plo, phi, pnr) == (lo, hi, nr)
                    present, _ = sw.has([cids[i] for i in pids])
                    assert present.all() and sw.block_count == len(pids), (G, r, sw.block_count, len(pids))
                    # the planned shard (plan + device-side subset of the resident witness) answers as the pulled one does
                    ps = shard.TipsetShard(engine, w, tip.parent_cids, tip.child_cid, tip.receipts_root, G, r)
                    pc_r, pb_r, pbl_r = ps.route(ts, cl, blob, blob_len)
                    planned_status = ps.witness.verify_event_claims(ts, pc_r, pb_r, pbl_r)
                    ps.close()
                    pos, c_r, b_r, bl_r = ipcfp.route_event_claims(cl, blob, blob_len, lo, hi, r == G - 1)
                    status[pos.astype(np.int64)] = sw.verify_event_claims(ts, c_r, b_r, bl_r)
                    assert np.array_equal(planned_status, status[pos.astype(np.int64)]), (G, r)
                    sst, shas, sm, _ = sw.scan_events(tip.receipts_root, tip.topic0, tip.topic1, actor=tip.filter_actor, want_touched=False)
                    assert sst == 1
                    has[lo: lo + len(shas)] = shas
                    n_matches += len(sm)
                    sw.close()
            finally:
                ipcfp.host_unregister(pk.data)
            assert np.array_equal(status, want), (G, np.nonzero(status != want)[0][:6])
            assert np.array_equal(has[: len(whas)], whas) and n_matches == len(wm), G
