"""GPU: the TRANSPORT forms of a bundle — what crosses PCIe in window T2 — against the plain forms they stand for.
  * ipcfp_witness_create_packed (no offset table, 32-byte digests + a prefix, escapes for other CIDs) must build the same
    witness as ipcfp_witness_create: CID verdicts, scan, execution order and every claim's status equal, and the oracle's;
  * ipcfp_expand_event_claims_device must rebuild ipcfp_event_claim_t[n] + blob byte for byte;
  * ipcfp_verify_event_claims_compact must give the statuses of ipcfp_verify_event_claims, on both verify routes.
Reference: src/proofs/common/bundle.rs:10-45 (blocks and event_proofs of a bundle), src/proofs/events/bundle.rs:5-23."""
import numpy as np
import pytest

from conftest import fuzz_seed
import torch

import claims
import ipc_filecoin_proofs_amd as ipcfp
import pyamt
from tools.synth import Tipset

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def tip():
    return Tipset(n_receipts=6000, n_parents=4, dup_permille=30, n_planted=12, max_events=4, no_events_permille=50,
                  variety=1, seed=fuzz_seed(404))


@pytest.fixture(scope="module")
def packed_claims(tip):
    ts, cl, blob, blob_len = ipcfp.pack_event_claims(
        tip.parent_cids, tip.child_cid, tip.parent_epoch, tip.child_epoch, tip.claim_exec, tip.claim_event,
        tip.claim_emitter, tip.exec_order[tip.claim_exec.astype(np.int64)], tip.claim_ntopics, tip.claim_topics,
        tip.claim_datalen, tip.claim_data)
    n = len(cl)
    liars = np.arange(3, n, 13)
    cl["exec_index"][liars[0::4]] += 1
    cl["emitter"][liars[1::4]] ^= 1
    cl["flags"][liars[2::4]] &= ~np.uint32(1)          # message_cid did not parse
    cl["message_cid"][liars[2::4]] = 0
    has_topic = liars[3::4][cl["n_topics"][liars[3::4]] > 0]
    blob[cl["topics_off"][has_topic]] = 0               # first topic not "0x" + 64 hex
    return ts, cl, blob, blob_len


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1).copy()).cuda()


def test_packed_witness_equals_the_plain_one(engine, oracle, tip, packed_claims):
    ts, cl, blob, blob_len = packed_claims
    data = tip.data.copy()
    bad = [5, 777, tip.n_blocks - 1]
    for b in bad:
        data[int(tip.off[b])] ^= 0x10                   # three blocks whose bytes no longer hash to their CID
    pk = ipcfp.PackedWitnessTables(data, tip.off, tip.lens, tip.cids)
    assert len(pk.esc_index) == 0 and pk.h2d_bytes < data.size + tip.off.nbytes + tip.lens.nbytes + tip.cids.nbytes - 16 * tip.n_blocks + 64
    with engine.witness(data, tip.off, tip.lens, tip.cids) as w, engine.witness_packed(pk) as wp:
        a, na = w.verify_cids()
        b, nb = wp.verify_cids()
        assert na == nb == len(bad) and np.array_equal(a, b) and np.nonzero(a != 1)[0].tolist() == bad
        sa = w.scan_events(tip.receipts_root, tip.topic0, tip.topic1, actor=tip.filter_actor)
        sb = wp.scan_events(tip.receipts_root, tip.topic0, tip.topic1, actor=tip.filter_actor)
        assert sa[0] == sb[0] and np.array_equal(sa[1], sb[1]) and np.array_equal(sa[2], sb[2]) and np.array_equal(sa[3], sb[3])
        ea, eb = w.exec_order(tip.parent_cids), wp.exec_order(tip.parent_cids)
        assert ea[0] == eb[0] and np.array_equal(ea[1], eb[1])
        va = w.verify_event_claims(ts, cl, blob, blob_len)
        vb = wp.verify_event_claims(ts, cl, blob, blob_len)
        assert np.array_equal(va, vb)
    ost = oracle.store(data, tip.off, tip.lens, tip.cids)
    vo = ost.verify_event_claims_packed(ts, cl, blob, threads=1)
    ost.close()
    said = vo != 255  # (the oracle's packed entry has no verdict for claims whose strings did not parse: 255)
    assert np.array_equal(vo[said], va[said]) and (va[~said] != 1).all() and said.sum() > len(cl) // 2


def test_packed_witness_with_other_cid_forms_and_duplicates(engine, oracle):
    """sha2-256 CIDs (escapes), an empty block, the same CID twice (last block wins): get() of every CID equal on both."""
    store = pyamt.Store()
    items = {i: pyamt.receipt(gas=1000 + i) for i in range(200)}
    root = pyamt.build_amt(store, items, version=0)
    extra = [store.put(bytes([0x80 | 3, 1, 2, i]), sha256_cid=True) for i in range(5)]
    data, off, lens, cids = store.tables()
    # append: an empty block under a standard CID, and a SECOND block under the CID of block 0
    more = bytes([0x82, 0x01, 0x02])
    data = np.concatenate([data, np.frombuffer(more, dtype=np.uint8)])
    off = np.concatenate([off, [data.size - 3, data.size - 3]]).astype(np.uint64)
    lens = np.concatenate([lens, [0, 3]]).astype(np.uint32)
    empty_cid = np.zeros(40, dtype=np.uint8)
    empty_cid[:38] = np.frombuffer(pyamt.cid_of(b""), dtype=np.uint8)
    cids = np.concatenate([cids, empty_cid[None, :], cids[0:1]])
    pk = ipcfp.PackedWitnessTables(data, off, lens, cids)
    assert len(pk.esc_index) == 5
    with engine.witness(data, off, lens, cids) as w, engine.witness_packed(pk) as wp:
        keys = [bytes(c) for c in cids]
        ha, ia = w.has(keys)
        hb, ib = wp.has(keys)
        assert np.array_equal(ha, hb) and np.array_equal(ia, ib) and ia[0] == len(cids) - 1
        probe = list(range(0, 210, 7))
        ga, la = w.amt_get(root, 0, "receipt", probe)
        gb, lb = wp.amt_get(root, 0, "receipt", probe)
        assert np.array_equal(ga, gb) and np.array_equal(la, lb)
        a, na = w.verify_cids()
        b, nb = wp.verify_cids()
        assert np.array_equal(a, b) and na == nb


def test_packed_witness_rejects_tables_that_do_not_add_up(engine, tip):
    pk = ipcfp.PackedWitnessTables(tip.data, tip.off, tip.lens, tip.cids)
    pk.lens = pk.lens.copy()
    pk.lens[3] += 1
    with pytest.raises(ipcfp.EngineError):
        engine.witness_packed(pk)
    pk.lens[3] -= 1
    pk.esc_index = np.array([9, 9], dtype=np.uint32)
    pk.esc_cids = np.zeros((2, 40), dtype=np.uint8)
    with pytest.raises(ipcfp.EngineError):
        engine.witness_packed(pk)


def test_device_expansion_is_byte_exact(engine, packed_claims):
    ts, cl, blob, blob_len = packed_claims
    groups, cc, cblob, cblob_len = ipcfp.compact_event_claims(cl, blob, blob_len)
    n = len(cl)
    d_cc, d_cb = dev(cc), dev(cblob)
    cap = cblob_len + 8 * n + 64
    d_out = torch.zeros(n * ipcfp.CLAIM_DTYPE.itemsize, dtype=torch.uint8, device="cuda")
    d_blob = torch.zeros(cap, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    got_len = engine.expand_event_claims_device(groups, d_cc.data_ptr(), n, d_cb.data_ptr(), cblob_len, d_out.data_ptr(),
                                                d_blob.data_ptr(), cap)
    assert got_len == blob_len
    assert d_out.cpu().numpy().tobytes() == cl.tobytes()
    assert d_blob.cpu().numpy()[:blob_len].tobytes() == blob[:blob_len].tobytes()
    # records that are out of range: tipset 0xffffffff, nothing written for them, the others unharmed
    bad = cc.copy()
    bad["group"][5] = 200
    bad["n_topics"][9] = 9
    d_bad = dev(bad)
    d_out.zero_()
    engine.expand_event_claims_device(groups, d_bad.data_ptr(), n, d_cb.data_ptr(), cblob_len, d_out.data_ptr(), d_blob.data_ptr(), cap)
    out = d_out.cpu().numpy().view(ipcfp.CLAIM_DTYPE)
    assert out["tipset"][5] == out["tipset"][9] == 0xFFFFFFFF and out["n_topics"][9] == 0 and (out["tipset"][:5] == 0).all()


def test_device_expansion_with_sizes_that_wrap_32_bits(engine, packed_claims):
    """70 000 records that each DECLARE 65 535 data bytes add up to 4.6 GB — the 32-bit offsets wrap, and a
    wrapped offset would pass the range checks against a 1 MB blob.  Every record must come out ERR_BAD_CLAIM
    (tipset 0xffffffff, nothing moved), and nothing may be written outside the output blob."""
    ts, cl, blob, blob_len = packed_claims
    groups, cc, cblob, cblob_len = ipcfp.compact_event_claims(cl, blob, blob_len)
    n = 70_000
    liar = np.zeros(n, dtype=ipcfp.COMPACT_DTYPE)
    liar[:] = cc[0]
    liar["n_topics"] = 0
    liar["data_len"] = 65535
    assert n * 65535 > 2**32
    small = np.full(1 << 20, 0xAB, dtype=np.uint8)
    d_cc, d_cb = dev(liar), dev(small)
    cap = 1 << 20
    guard = 4096
    d_out = torch.zeros(n * ipcfp.CLAIM_DTYPE.itemsize, dtype=torch.uint8, device="cuda")
    d_blob = torch.full((cap + guard,), 0x5A, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    engine.expand_event_claims_device(groups, d_cc.data_ptr(), n, d_cb.data_ptr(), len(small), d_out.data_ptr(), d_blob.data_ptr(), cap)
    out = d_out.cpu().numpy().view(ipcfp.CLAIM_DTYPE)
    assert (out["tipset"] == 0xFFFFFFFF).all() and (out["data_len"] == 0).all()
    assert (d_blob.cpu().numpy() == 0x5A).all()
    # just under the wrap (65 000 such records = 4.26e9 < 2^32 … no: still above 4 GiB? 65 000 * 65 535 = 4 259 775 000 < 4 294 967 296):
    # nothing wrapped, the per-record range check decides — the first 16 records fit the 1 MB blob, the rest do not
    m = 65_000
    assert m * 65535 < 2**32
    d_out.zero_()
    engine.expand_event_claims_device(groups, d_cc.data_ptr(), m, d_cb.data_ptr(), len(small), d_out.data_ptr(), d_blob.data_ptr(), cap)
    out = d_out.cpu().numpy().view(ipcfp.CLAIM_DTYPE)[:m]
    fit = (1 << 20) // 65535
    assert (out["tipset"][:fit] == 0).all() and (out["tipset"][fit:] == 0xFFFFFFFF).all()
    got = d_blob.cpu().numpy()
    assert (got[: fit * 65535] == 0xAB).all() and (got[cap:] == 0x5A).all()


@pytest.mark.parametrize("fast", [1, 0])
def test_compact_claims_verify_like_the_plain_ones(engine, oracle, tip, packed_claims, fast):
    ts, cl, blob, blob_len = packed_claims
    groups, cc, cblob, cblob_len = ipcfp.compact_event_claims(cl, blob, blob_len)
    engine.set_tuning("fast_verify", fast)
    try:
        with engine.witness(tip.data, tip.off, tip.lens, tip.cids) as w:
            want = w.verify_event_claims(ts, cl, blob, blob_len)
            w.rebuild_index()
            got = w.verify_event_claims_compact(ts, groups, cc, cblob, cblob_len)
            assert np.array_equal(got, want) and (want != 1).sum() > 100
            # a record naming no group: ERR_BAD_CLAIM for it, the rest as before
            bad = cc.copy()
            bad["group"][11] = 77
            w.rebuild_index()
            got2 = w.verify_event_claims_compact(ts, groups, bad, cblob, cblob_len)
            w.rebuild_index()
            plain_bad = cl.copy()
            plain_bad["tipset"][11] = 0xFFFFFFFF
            want2 = w.verify_event_claims(ts, plain_bad, blob, blob_len)
            assert np.array_equal(got2, want2) and got2[11] != want[11]
    finally:
        engine.set_tuning("fast_verify", -1)
    ost = oracle.store(tip.data, tip.off, tip.lens, tip.cids)
    vo = ost.verify_event_claims_packed(ts, cl, blob, threads=1)
    ost.close()
    said = vo != 255  # (no verdict from the oracle's packed entry for claims whose strings did not parse)
    assert np.array_equal(vo[said], want[said]) and (want[~said] != 1).all() and (want[said] == 1).sum() > len(cl) // 2
