"""GPU: the reference's answer at what used to be the engine's limits (`IPCFP_E_UNSUPPORTED`).

* A witness-key CID of MORE than 40 bytes.  `cid` 0.11 takes multihashes of up to 64 bytes, `Cid::try_from` parses them
  (src/proofs/common/witness.rs:60-72) and `load_witness_store` keys blocks by them (src/proofs/events/verifier.rs:79-89).
  Such a CID crosses the ABI folded (include/ipcfp.h "CIDs": ff | len | blake2b-256(cid)) and the device folds the long
  links it reads out of blocks the same way.  Here a blake2b-512 CID (70 bytes) stands as child header CID, as the receipts
  root the header links to, as message CIDs inside the message AMTs (one of them in two parent blocks: first-seen dedupe),
  and as a link inside the state tree's HAMT — every status byte against the oracle, which keys its store by the true bytes.
* A tipset key of more than IPCFP_MAX_PARENTS (32) parent blocks: `verify_event_proof` compares and walks whatever key the
  claim names (src/proofs/events/verifier.rs:147-181, src/proofs/events/utils.rs:16-30).  33 and 100 parents on verify
  (strings, packed; both routes), exec_order, scan, generate, the shard planner and the self-planned shard.
"""
import hashlib

import numpy as np
import pytest

import claims
import ipc_filecoin_proofs_amd as ipcfp
from tools.synth import Tipset

pytestmark = pytest.mark.gpu

LINK38 = bytes.fromhex("d82a582700")          # tag 42, bytes(39), identity multibase — then the 38 CID bytes


def long_cid(seed: bytes) -> bytes:
    """CIDv1, dag-cbor, blake2b-512 (multicodec 0xb240), 64-byte digest: 70 bytes."""
    return bytes.fromhex("0171c0e40240") + hashlib.blake2b(seed, digest_size=64).digest()


def link_of(cid: bytes) -> bytes:
    assert len(cid) + 1 < 256
    return bytes([0xD8, 0x2A, 0x58, len(cid) + 1, 0x00]) + cid


class Rewritten:
    """A synthetic tipset's witness with some CIDs replaced by long ones: in the tables' keys and in every link to them."""

    def __init__(self, tip):
        self.tip = tip
        self.blocks = [tip.block(i) for i in range(tip.n_blocks)]
        self.cids = [tip.cids[i, :38].tobytes() for i in range(tip.n_blocks)]
        self.renamed = {}

    def rename(self, old38: bytes, new: bytes, rekey=True):
        """every link `old38` in every block becomes a link to `new`; the block stored under old38 (if any) is re-keyed"""
        old38 = bytes(old38[:38])
        pat, rep, n_links = LINK38 + old38, link_of(new), 0
        for i, b in enumerate(self.blocks):
            if pat in b:
                n_links += b.count(pat)
                self.blocks[i] = b.replace(pat, rep)
        if rekey:
            for i, c in enumerate(self.cids):
                if c == old38:
                    self.cids[i] = new
        self.renamed[old38] = new
        return n_links

    def tables(self):
        lens = np.array([len(b) for b in self.blocks], dtype=np.uint32)
        off = np.zeros(len(lens), dtype=np.uint64)
        off[1:] = np.cumsum(lens[:-1], dtype=np.uint64)
        data = np.frombuffer(b"".join(self.blocks), dtype=np.uint8).copy()
        return data, off, lens

    def s(self, cid38: bytes) -> str:
        """the CID string a claim names: the long one where the CID was renamed"""
        c = bytes(cid38[:38])
        return claims.cid_str(self.renamed.get(c, c))


def test_fold_is_the_documented_one_and_matches_strings(engine):
    # (`engine` first: torch's bundled HIP runtime must come up before libipcfp.so is loaded — conftest.py)
    c = long_cid(b"x")
    slot = ipcfp.cid_slot(c)
    assert len(c) == 70 and slot[0] == 0xFF and slot[1] == 70 and bytes(slot[34:]) == bytes(6)
    assert bytes(slot[2:34]) == hashlib.blake2b(c, digest_size=32).digest()
    # `Cid::try_from(&str)` of the engine folds the same way; a short CID stays itself
    s = claims.cid_str(c)
    assert ipcfp.cid_to_string(c) == s
    out = np.zeros(40, dtype=np.uint8)
    assert ipcfp.load_library().ipcfp_cid_from_string(s.encode(), out.ctypes.data) == 70 and np.array_equal(out, slot)
    short = bytes.fromhex("0171a0e40220") + bytes(range(32))
    assert bytes(ipcfp.cid_slot(short)) == short + b"\0\0"
    with pytest.raises(ipcfp.EngineError):
        ipcfp.cid_slot(c[:-1])   # not a CID: the multihash says 64 bytes


def make_long_tip():
    tip = Tipset(n_receipts=900, n_parents=3, n_planted=6, variety=1, max_events=4, dup_permille=60, n_actors=1500,
                 n_contracts=4, slots_per_contract=8, storage_layout_mix=1, n_actor_queries=6, seed=0x10C1D)
    rw = Rewritten(tip)
    # (1) the child header under a long CID (nothing links to it: the claims name it)
    rw.rename(tip.child_cid, long_cid(b"child"))
    # (2) the receipts root: a long link inside the child header, the root block re-keyed
    assert rw.rename(tip.receipts_root, long_cid(b"receipts")) == 1
    # (3) message CIDs: values of the message AMTs' leaves (not witness keys).  One that stands in two parent blocks
    # (a duplicate: first-seen dedupe must see the two long links as ONE message) and a handful of others
    raw = b"".join(rw.blocks)
    msgs = [tip.exec_order[e, :38].tobytes() for e in range(len(tip.exec_order))]
    dup = [m for m in msgs if raw.count(LINK38 + m) >= 2]
    assert dup, "the synthetic tipset holds no message in two parent blocks"
    chosen = dup[:2] + msgs[5:9] + [tip.exec_order[int(tip.claim_exec[0]), :38].tobytes()]
    for k, m in enumerate(dict.fromkeys(chosen)):
        assert rw.rename(m, long_cid(b"msg%d" % k), rekey=False) >= 1
    # (4) a link inside the state tree: the first link of the actors HAMT's root node, its child re-keyed
    root = rw.blocks[tip.find_block(tip.actors_root)]
    at = root.find(LINK38)
    assert at > 0
    child38 = root[at + 5: at + 43]
    assert rw.rename(child38, long_cid(b"hamt")) >= 1
    return tip, rw


@pytest.fixture(scope="module")
def long_tip():
    return make_long_tip()


def test_long_cids_event_proofs_every_status_vs_the_oracle(engine, oracle, long_tip):
    tip, rw = long_tip
    data, off, lens = rw.tables()
    slots = ipcfp.cid_slots(rw.cids)
    assert (slots[:, 0] == 0xFF).sum() == 3   # child header, receipts root, the HAMT node
    ec = claims.EventClaims(tip)
    for k in range(ec.n):
        ec.set_str(k, "child_block_cid", rw.s(tip.child_cid))
        ec.set_str(k, "message_cid", rw.s(tip.exec_order[int(tip.claim_exec[k])]))
    n_long_msg = sum(1 for k in range(ec.n) if tip.exec_order[int(tip.claim_exec[k]), :38].tobytes() in rw.renamed)
    assert n_long_msg >= 1
    # liars: another long CID of the same shape where the honest one stands
    ec.set_str(3, "child_block_cid", claims.cid_str(long_cid(b"no such header")))        # Err: missing child header
    ec.set_str(4, "message_cid", claims.cid_str(long_cid(b"no such message")))           # Ok(false): not in the order
    ec.set_str(5, "message_cid", claims.cid_str(tip.exec_order[int(tip.claim_exec[5]) + 1, :38].tobytes())
               if int(tip.claim_exec[5]) + 1 < len(tip.exec_order) else "bafy")
    ost = oracle.store_var(data, off, lens, rw.cids)
    want = ost.verify_event_proofs(ec, mode=1)
    assert (want == 1).sum() > ec.n // 2 and want[3] >= 64 and want[4] not in (1,) and want[4] < 64
    with engine.witness(data, off, lens, slots) as w:
        for fast in (1, 0):
            engine.set_tuning("fast_verify", fast)
            w.rebuild_index()
            got = w.verify_event_proofs(ec.arr, ec.n)
            assert np.array_equal(got, want), (fast, np.nonzero(got != want)[0][:8], got[got != want][:8], want[got != want][:8])
        engine.set_tuning("fast_verify", -1)
        # the packed lowering carries the folds; its route must agree with the strings'
        ts, cl, blob = ipcfp.pack_event_proofs(ec.arr, ec.n)
        assert np.array_equal(w.verify_event_claims(ts, cl, blob, len(blob)), want)
        # the execution order: long message CIDs come back folded, first occurrence only
        gs, gc = w.exec_order(tip.parent_cids)
        assert gs == 1 and len(gc) == len(tip.exec_order)
        want_order = ipcfp.cid_slots([rw.renamed.get(tip.exec_order[e, :38].tobytes(), tip.exec_order[e, :38].tobytes())
                                      for e in range(len(tip.exec_order))])
        assert np.array_equal(gc, want_order)
        # the scan walks the receipts AMT behind the long root
        root_slot = bytes(ipcfp.cid_slot(rw.renamed[tip.receipts_root[:38]]))
        st, has, m, _ = w.scan_events(root_slot, tip.topic0, tip.topic1, actor=tip.filter_actor, want_touched=False)
        with engine.witness(tip.data, tip.off, tip.lens, tip.cids) as w0:
            st0, has0, m0, _ = w0.scan_events(tip.receipts_root, tip.topic0, tip.topic1, actor=tip.filter_actor, want_touched=False)
        assert st == st0 == 1 and np.array_equal(has, has0) and len(m) == len(m0) > 0
        # K1 does not hash under a blake2b-512 CID: unchecked, never a mismatch
        cs, n_bad = w.verify_cids()
        assert n_bad == (cs == 0).sum() and (cs[slots[:, 0] == 0xFF] == 2).all()
    ost.close()


def test_long_cids_storage_proofs_every_status_vs_the_oracle(engine, oracle, long_tip):
    tip, rw = long_tip
    data, off, lens = rw.tables()
    slots = ipcfp.cid_slots(rw.cids)
    sc = claims.StorageClaims(tip)
    for k in range(sc.n):
        sc.set_str(k, "child_block_cid", rw.s(tip.child_cid))
    sc.set_str(1, "value", "0x" + "ee" * 32)
    sc.set_str(2, "child_block_cid", claims.cid_str(long_cid(b"not the header")))
    ost = oracle.store_var(data, off, lens, rw.cids)
    want = ost.verify_storage_proofs(sc, mode=1)
    assert (want == 1).sum() >= sc.n // 2 and want[1] != 1 and want[2] >= 64
    with engine.witness(data, off, lens, slots) as w:
        for table in (1, 0):
            engine.set_tuning("hamt_table", table)
            got = w.verify_storage_proofs(sc.arr, sc.n)
            assert np.array_equal(got, want), (table, got, want)
        engine.set_tuning("hamt_table", -1)
        # actor gets through the long link: status and value of every query, present and absent ids
        keys = [b"\x00" + bytes(_uvarint(int(i))) for i in tip.query_ids]
        st, loc = w.hamt_get(tip.actors_root, 5, "actor_state", keys)
        ost_st, ost_vals = ost.hamt_get(tip.actors_root, 5, "actor_state", keys)
        assert np.array_equal(st, ost_st) and (st == 1).sum() >= 1
        vals = w.read_values(loc)
        assert all(vals[i] == ost_vals[i] for i in range(len(keys)) if st[i] == 1)
    ost.close()


def _uvarint(n):
    out = bytearray()
    while True:
        b = n & 0x7F
        n >>= 7
        out.append(b | (0x80 if n else 0))
        if not n:
            return out


def test_bundle_json_with_a_long_block_cid(engine, oracle, long_tip):
    """f1: `ProofBlock.cid` as a JSON byte array of 70 numbers (src/proofs/common/bundle.rs:10-15) used to be refused."""
    import base64
    import json

    tip, rw = long_tip
    keep = sorted({tip.find_block(tip.child_cid), tip.find_block(tip.receipts_root), 0, 1, 2})
    blocks = [{"cid": list(rw.cids[i]), "data": base64.b64encode(rw.blocks[i]).decode()} for i in keep]
    text = json.dumps({"storage_proofs": [], "event_proofs": [], "blocks": blocks})
    b = engine.bundle(text.encode())
    try:
        w = b.witness
        assert w.block_count == len(keep)
        has, _ = w.has([bytes(ipcfp.cid_slot(rw.cids[i])) for i in keep])
        assert has.all()
        no, _ = w.has([bytes(ipcfp.cid_slot(long_cid(b"some other long cid")))])
        assert not no.any()
    finally:
        b.close()


# ---- tipset keys wider than the inline form ----
@pytest.mark.parametrize("P", [33, 100])
def test_tipsets_with_more_than_thirty_two_parent_blocks(engine, oracle, P):
    tp = Tipset(n_receipts=700, n_parents=P, n_planted=5, variety=1, max_events=4, dup_permille=40, seed=0x9A0 + P)
    assert len(tp.parent_cids) == P
    w = engine.witness(tp.data, tp.off, tp.lens, tp.cids)
    st = oracle.store(tp.data, tp.off, tp.lens, tp.cids)
    # exec_order
    gs, gc = w.exec_order(tp.parent_cids)
    os_, oc = st.exec_order(tp.parent_cids)
    assert gs == os_ == 1 and np.array_equal(gc, oc) and np.array_equal(gc, tp.exec_order)
    # verify, strings: honest claims and the liars a wide key invites
    ec = claims.EventClaims(tp)
    pstr = [claims.cid_str(c) for c in tp.parent_cids]
    ec.set_parents(2, pstr[:-1])                                   # one parent short: child_hdr.parents != parent_cids
    ec.set_parents(3, pstr[:40 % P] + [pstr[0]] + pstr[40 % P + 1:])   # one parent replaced (beyond / inside the inline 32)
    ec.set_parents(4, pstr[:32])                                   # exactly the inline count: another tipset key
    ec.set_parents(5, pstr + [pstr[0]])                            # one too many
    ec.set_parents(6, pstr[:P - 1] + ["not a cid"])                # the last string does not parse: Err
    ec.arr[7].exec_index += 1
    want = st.verify_event_proofs(ec, mode=1)
    assert (want == 1).sum() > ec.n // 2 and want[2] != 1 and want[6] >= 64
    assert np.array_equal(w.verify_event_proofs(ec.arr, ec.n), want)
    assert np.array_equal(st.verify_event_proofs(ec, mode=0), want)          # (as written agrees with the fair baseline)
    # verify, packed: the tipset ref carries its tail behind more_parents; both routes
    ts, cl, blob, blob_len = ipcfp.pack_event_claims(
        tp.parent_cids, tp.child_cid, tp.parent_epoch, tp.child_epoch, tp.claim_exec, tp.claim_event,
        tp.claim_emitter, tp.exec_order[tp.claim_exec.astype(np.int64)], tp.claim_ntopics, tp.claim_topics,
        tp.claim_datalen, tp.claim_data)
    assert int(ts["n_parents"][0]) == P and int(ts["more_parents"][0]) != 0
    want_p = st.verify_event_claims_packed(ts, cl, blob, threads=1)
    assert (want_p == 1).sum() > len(cl) // 2 and (want_p < 64).all()
    for fast in (1, 0):
        engine.set_tuning("fast_verify", fast)
        w.rebuild_index()
        assert np.array_equal(w.verify_event_claims(ts, cl, blob, blob_len), want_p)
    engine.set_tuning("fast_verify", -1)
    # the host lowering of the string claims gives the same wide refs
    ts2, cl2, blob2 = ipcfp.pack_event_proofs(ec.arr, ec.n)
    assert int(ts2["n_parents"].max()) == P + 1 and np.array_equal(w.verify_event_claims(ts2, cl2, blob2, len(blob2)), want)
    # scan + generate
    gsg, gm, gmsg, gids = w.generate_event_proofs(tp.parent_cids, tp.child_cid, tp.topic0, tp.topic1, actor=tp.filter_actor)
    osg, otrip, omsg, owit = st.generate_event_proof(tp.parent_cids, tp.child_cid, tp.topic0, tp.topic1, actor=tp.filter_actor)
    assert gsg == osg == 1 and len(gm) >= 5 and np.array_equal(gmsg, omsg) and np.array_equal(tp.cids[gids], owit)
    # a parent header missing from the witness: the reference's Err, from the general route
    keep = np.ones(tp.n_blocks, dtype=bool)
    keep[tp.find_block(tp.parent_cids[P - 1])] = False
    sub = ipcfp.witness_cut_host(tp.data, tp.off, tp.lens, tp.cids, np.nonzero(keep)[0].astype(np.uint32))
    with engine.witness(*sub) as w1:
        st1 = oracle.store(*sub)
        ec1 = claims.EventClaims(tp, indices=np.arange(0, 50))
        want1 = st1.verify_event_proofs(ec1, mode=1)
        assert (want1 >= 64).all() and np.array_equal(w1.verify_event_proofs(ec1.arr, ec1.n), want1)
        g1, _ = w1.exec_order(tp.parent_cids)
        o1, _ = st1.exec_order(tp.parent_cids)
        assert g1 == o1 >= 64
        st1.close()
    w.close()
    st.close()


@pytest.mark.parametrize("P", [33, 100])
def test_wide_tipset_shards_planned_and_pulled(engine, oracle, P):
    """The multi-GPU cut of a tipset whose key is wider than the inline form: the planner that sees the whole witness and
    the self-planned shard (the seeds of the pull come out of HBM) agree, and the shards' verdicts merge to the unsharded ones."""
    from ipc_filecoin_proofs_amd import shard

    tp = Tipset(n_receipts=1200, n_parents=P, n_planted=8, variety=1, max_events=3, dup_permille=30, seed=0x5A0 + P)
    ts, cl, blob, blob_len = ipcfp.pack_event_claims(
        tp.parent_cids, tp.child_cid, tp.parent_epoch, tp.child_epoch, tp.claim_exec, tp.claim_event,
        tp.claim_emitter, tp.exec_order[tp.claim_exec.astype(np.int64)], tp.claim_ntopics, tp.claim_topics,
        tp.claim_datalen, tp.claim_data)
    cl["emitter"][7] ^= 1
    pk = ipcfp.PackedWitnessTables(tp.data, tp.off, tp.lens, tp.cids, ingest=True)
    ipcfp.host_register(pk.data)
    try:
        with engine.witness(tp.data, tp.off, tp.lens, tp.cids) as full:
            want = full.verify_event_claims(ts, cl, blob, blob_len)
            assert (want == 1).sum() > len(cl) // 2 and want[7] != 1
            G = 3
            status = np.full(len(cl), 255, dtype=np.uint8)
            for r in range(G):
                st, lo, hi, nr, ids = full.shard_plan_tipset(tp.parent_cids, tp.child_cid, G, r)
                pst, w, plo, phi, pn, stats = engine.witness_shard_pull(pk, tp.parent_cids, tp.child_cid, G, r)
                assert st == pst == 1 and (lo, hi, nr) == (plo, phi, pn) and w.block_count == len(ids)
                w.close()
                s = shard.TipsetShard.from_pull(engine, pk, tp.parent_cids, tp.child_cid, tp.receipts_root, G, r)
                s.route(ts, cl, blob, blob_len)
                status[s.positions.astype(np.int64)] = s.witness.verify_event_claims(ts, s.claims, s.blob, s.blob_len)
                s.close()
            assert np.array_equal(status, want)
        ost = oracle.store(tp.data, tp.off, tp.lens, tp.cids)
        assert np.array_equal(ost.verify_event_claims_packed(ts, cl, blob, threads=1), want)
        ost.close()
    finally:
        ipcfp.host_unregister(pk.data)


# ---- differential fuzz at the former limits: random corruptions of a witness that HOLDS long CIDs and of one whose
# tipset key is wide — the fold must agree with the true bytes on every error path too (a corrupted long link is another
# long link, or no link at all) ----
def _mutate(blocks, rng, n_flips):
    out = list(blocks)
    touched = []
    for _ in range(n_flips):
        b = int(rng.integers(0, len(out)))
        if not out[b]:
            continue
        buf = bytearray(out[b])
        pos = int(rng.integers(0, len(buf)))
        mode = int(rng.integers(0, 4))
        if mode == 0:
            buf[pos] ^= 1 << int(rng.integers(0, 8))
        elif mode == 1:
            buf[pos] = int(rng.integers(0, 256))
        elif mode == 2:
            buf[pos] = [0x80, 0x9F, 0xFF, 0xD8, 0x5F, 0xF6, 0x00, 0x1B][int(rng.integers(0, 8))]
        else:
            buf[pos] = (buf[pos] + 1) & 0xFF
        out[b] = bytes(buf)
        touched.append(b)
    return out, touched


def test_fuzz_of_a_witness_with_long_cids(engine, oracle, long_tip):
    from conftest import fuzz_seed

    tip, rw = long_tip
    rng = np.random.default_rng(fuzz_seed(20260930))
    slots = ipcfp.cid_slots(rw.cids)
    ec = claims.EventClaims(tip, indices=np.arange(0, 300))
    for k in range(ec.n):
        ec.set_str(k, "child_block_cid", rw.s(tip.child_cid))
        ec.set_str(k, "message_cid", rw.s(tip.exec_order[int(tip.claim_exec[k])]))
    sc = claims.StorageClaims(tip)
    for k in range(sc.n):
        sc.set_str(k, "child_block_cid", rw.s(tip.child_cid))
    # the blocks whose bytes hold long links are hit on purpose half of the time
    hot = [i for i, b in enumerate(rw.blocks) if bytes.fromhex("d82a584700") in b]
    assert len(hot) >= 4
    n_err = 0
    for it in range(120):
        blocks, touched = _mutate(rw.blocks, rng, 1 + it % 3)
        if it % 2:
            b = hot[int(rng.integers(0, len(hot)))]
            buf = bytearray(blocks[b])
            at = buf.find(bytes.fromhex("d82a584700")) + int(rng.integers(0, 76))   # inside the long link
            buf[at] ^= 1 << int(rng.integers(0, 8))
            blocks[b] = bytes(buf)
            touched.append(b)
        lens = np.array([len(b) for b in blocks], dtype=np.uint32)
        off = np.zeros(len(lens), dtype=np.uint64)
        off[1:] = np.cumsum(lens[:-1], dtype=np.uint64)
        data = np.frombuffer(b"".join(blocks), dtype=np.uint8).copy()
        ost = oracle.store_var(data, off, lens, rw.cids)
        want_e = ost.verify_event_proofs(ec, mode=1)
        want_s = ost.verify_storage_proofs(sc, mode=1)
        with engine.witness(data, off, lens, slots) as w:
            got_e = w.verify_event_proofs(ec.arr, ec.n)
            got_s = w.verify_storage_proofs(sc.arr, sc.n)
        ost.close()
        ctx = f"round {it}, blocks {touched}"
        assert np.array_equal(got_e, want_e), ("events", ctx, np.nonzero(got_e != want_e)[0][:5], got_e[got_e != want_e][:5], want_e[got_e != want_e][:5])
        assert np.array_equal(got_s, want_s), ("storage", ctx, np.nonzero(got_s != want_s)[0][:5], got_s[got_s != want_s][:5], want_s[got_s != want_s][:5])
        n_err += int((want_e >= 64).any()) + int((want_s >= 64).any())
    assert n_err > 5


def test_fuzz_of_a_wide_tipset(engine, oracle):
    from conftest import fuzz_seed

    tp = Tipset(n_receipts=300, n_parents=40, n_planted=4, variety=1, max_events=3, dup_permille=80, seed=fuzz_seed(0x71DE))
    rng = np.random.default_rng(fuzz_seed(20260931))
    blocks = [tp.block(i) for i in range(tp.n_blocks)]
    headers = [tp.find_block(c) for c in tp.parent_cids] + [tp.find_block(tp.child_cid)]
    ec = claims.EventClaims(tp, indices=np.arange(0, 200))
    n_err = 0
    for it in range(80):
        mut, touched = _mutate(blocks, rng, 1 + it % 3)
        if it % 2:   # a parent header, the child header or a TxMeta on purpose
            b = headers[int(rng.integers(0, len(headers)))]
            buf = bytearray(mut[b])
            buf[int(rng.integers(0, len(buf)))] ^= 1 << int(rng.integers(0, 8))
            mut[b] = bytes(buf)
            touched.append(b)
        lens = np.array([len(b) for b in mut], dtype=np.uint32)
        off = np.zeros(len(lens), dtype=np.uint64)
        off[1:] = np.cumsum(lens[:-1], dtype=np.uint64)
        data = np.frombuffer(b"".join(mut), dtype=np.uint8).copy()
        ost = oracle.store(data, off, lens, tp.cids)
        want = ost.verify_event_proofs(ec, mode=1)
        o, oc = ost.exec_order(tp.parent_cids)
        with engine.witness(data, off, lens, tp.cids) as w:
            got = w.verify_event_proofs(ec.arr, ec.n)
            g, gc = w.exec_order(tp.parent_cids)
        ost.close()
        ctx = f"round {it}, blocks {touched}"
        assert np.array_equal(got, want), ("events", ctx, np.nonzero(got != want)[0][:5], got[got != want][:5], want[got != want][:5])
        assert g == o and np.array_equal(gc, oc), ("exec_order", ctx, g, o)
        n_err += int((want >= 64).any())
    assert n_err > 5
