"""GPU: the event table (csrc/kernels/event_table.h) never changes an outcome.  Scan and verify_event_proof are run
with the table (built by the scan, or by the verify call when no scan preceded it; a second filter counted from the
records) and without it (IPCFP_EVENT_TABLE=0: every receipt / claim walks its blocks), over events AMTs the table
covers (leaf roots up to 64 slots) and ones it does not (wider nodes, taller trees), with honest and lying claims
and with a damaged events block — all against the oracle."""
import os

import numpy as np
import pytest

from conftest import fuzz_seed

import ipc_filecoin_proofs_amd as ipcfp
from tools.synth import Tipset

pytestmark = pytest.mark.gpu


def packed(tip, lie=True):
    ts, cl, blob, blob_len = ipcfp.pack_event_claims(
        tip.parent_cids, tip.child_cid, tip.parent_epoch, tip.child_epoch, tip.claim_exec, tip.claim_event,
        tip.claim_emitter, tip.exec_order[tip.claim_exec.astype(np.int64)], tip.claim_ntopics, tip.claim_topics,
        tip.claim_datalen, tip.claim_data)
    if lie:
        n = len(cl)
        k = np.arange(2, n, 7)
        cl["event_index"][k[0::5]] += 1
        cl["event_index"][k[1::5]] = 63
        cl["event_index"][k[2::5]] = 64
        cl["event_index"][k[3::5]] = np.uint64(0xFFFFFFFFFFFFFFFF)
        cl["emitter"][k[4::5]] += 1
        t = np.arange(5, n, 11)
        t = t[cl["n_topics"][t] > 0]
        blob[cl["topics_off"][t] + 9] ^= 0x10
        u = np.arange(6, n, 13)
        cl["n_topics"][u[cl["n_topics"][u] > 0]] -= 1  # one topic fewer than the event has
    return ts, cl, blob, blob_len


def engine_results(engine, tip, ts, cl, blob, blob_len, table: bool, order: str, data=None):
    os.environ["IPCFP_EVENT_TABLE"] = "1" if table else "0"
    try:
        with engine.witness(tip.data if data is None else data, tip.off, tip.lens, tip.cids) as w:
            if order == "scan-first":
                scan = w.scan_events(tip.receipts_root, tip.topic0, tip.topic1, actor=tip.filter_actor)
                st = w.verify_event_claims(ts, cl, blob, blob_len)
            else:
                st = w.verify_event_claims(ts, cl, blob, blob_len)
                scan = w.scan_events(tip.receipts_root, tip.topic0, tip.topic1, actor=tip.filter_actor)
            scan_any = w.scan_events(tip.receipts_root, tip.topic0, tip.topic1, actor=None, want_touched=False)
    finally:
        os.environ.pop("IPCFP_EVENT_TABLE", None)
    return st, scan, scan_any


def same_scan(a, b):
    return (a[0] == b[0] and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2]) and
            (a[3] is None or b[3] is None or np.array_equal(a[3], b[3])))


@pytest.mark.parametrize("bit_width", [5, 3, 6, 7])
def test_table_and_walk_agree_with_the_oracle(engine, oracle, bit_width):
    tip = Tipset(n_receipts=6000, n_parents=3, n_planted=25, variety=1, max_events=9 if bit_width == 3 else 5,
                 no_events_permille=80, events_bit_width=bit_width, seed=fuzz_seed(900 + bit_width))
    ts, cl, blob, blob_len = packed(tip)
    ost = oracle.store(tip.data, tip.off, tip.lens, tip.cids, threads=0)
    want = ost.verify_event_claims_packed(ts, cl, blob, threads=0)
    o_scan = ost.scan_events(tip.receipts_root, tip.topic0, tip.topic1, actor=tip.filter_actor, threads=0)
    o_any = ost.scan_events(tip.receipts_root, tip.topic0, tip.topic1, actor=None, want_touched=False, threads=0)
    ost.close()
    assert (want != 255).all() and len(set(want.tolist())) >= 5
    ref = None
    for table, order in ((True, "scan-first"), (True, "verify-first"), (False, "scan-first")):
        st, scan, scan_any = engine_results(engine, tip, ts, cl, blob, blob_len, table, order)
        assert np.array_equal(st, want), (bit_width, table, order, np.nonzero(st != want)[0][:10], st[st != want][:10], want[st != want][:10])
        assert scan[0] == o_scan[0] == 1 and np.array_equal(scan[1], o_scan[1]) and len(scan[2]) == len(o_scan[2])
        assert np.array_equal(scan[2]["exec_index"], o_scan[2][:, 0]) and np.array_equal(scan[2]["event_index"], o_scan[2][:, 1])
        assert {bytes(c) for c in tip.cids[scan[3]]} == {bytes(c) for c in o_scan[3]}
        assert scan_any[0] == 1 and np.array_equal(scan_any[1], o_any[1]) and len(scan_any[2]) == len(o_any[2])
        assert np.array_equal(scan_any[2]["event_index"], o_any[2][:, 1])
        if ref is None:
            ref = (st, scan, scan_any)
        else:  # match records (block, off, len of the located events) are the same with and without the table
            assert same_scan(scan, ref[1]) and same_scan(scan_any, ref[2])


def test_damaged_events_blocks(engine, oracle):
    """An events AMT root that no longer decodes (byte flipped inside an event; block cut short) is an Err for every
    claim on that receipt and stops the scan at that receipt — identically with and without the table."""
    tip = Tipset(n_receipts=3000, n_parents=3, n_planted=12, variety=0, max_events=4, no_events_permille=0, seed=fuzz_seed(31))
    ts, cl, blob, blob_len = packed(tip, lie=False)
    with engine.witness(tip.data, tip.off, tip.lens, tip.cids) as w:
        _, _, m, _ = w.scan_events(tip.receipts_root, tip.topic0, tip.topic1, actor=None, want_touched=False)
    assert len(m) >= 6
    for victim_row, damage in ((1, "key"), (len(m) // 2, "type"), (len(m) - 2, "truncate")):
        rec = m[victim_row]
        b, off = int(rec["block"]), int(rec["off"])
        data = tip.data.copy()
        base = int(tip.off[b])
        lens = tip.lens.copy()
        if damage == "key":
            data[base + off + 1] = 0x5F        # the emitter becomes an indefinite-length byte string
        elif damage == "type":
            data[base + off] = 0xA2            # the StampedEvent tuple becomes a map
        else:
            lens[b] -= 1                       # the block loses its last byte
        ost = oracle.store(data, tip.off, lens, tip.cids, threads=0)
        want = ost.verify_event_claims_packed(ts, cl, blob, threads=0)
        o_scan = ost.scan_events(tip.receipts_root, tip.topic0, tip.topic1, actor=tip.filter_actor, want_touched=False, threads=0)
        ost.close()
        assert (want >= 64).sum() >= 1 and o_scan[0] >= 64
        for table in (True, False):
            os.environ["IPCFP_EVENT_TABLE"] = "1" if table else "0"
            try:
                with engine.witness(data, tip.off, lens, tip.cids) as w:
                    st = w.verify_event_claims(ts, cl, blob, blob_len)
                    gs = w.scan_events(tip.receipts_root, tip.topic0, tip.topic1, actor=tip.filter_actor, want_touched=False)
            finally:
                os.environ.pop("IPCFP_EVENT_TABLE", None)
            assert np.array_equal(st, want), (damage, table)
            assert gs[0] == o_scan[0], (damage, table)
