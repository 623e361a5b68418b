"""GPU: differential fuzzing of the SELF-PLANNED shard (ipcfp_witness_create_shard_pull).  Random byte flips inside witness
blocks (CIDs left alone) hit headers, TxMeta, AMT nodes of all three kinds and events: whatever the witness now says, G
pulled shards — each holding only what the device found by following the links — must give, merged, exactly the unsharded
engine's statuses and scan result, and where the whole witness is resident the pulled shard must hold what the planner
(ipcfp_shard_plan_tipset) lists for the same rank.  Reference loops being cut: src/proofs/verifier.rs:19-28,49-54,
src/proofs/events/verifier.rs:62-71."""
import numpy as np
import pytest

from conftest import fuzz_seed

import ipc_filecoin_proofs_amd as ipcfp
from ipc_filecoin_proofs_amd import shard
from test_gpu_fuzz import mutate
from tools.synth import Tipset

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def tip():
    return Tipset(n_receipts=700, n_parents=3, dup_permille=80, n_planted=9, variety=1, max_events=5, no_events_permille=60, seed=fuzz_seed(5151))


def test_pulled_shards_of_corrupted_witnesses_equal_the_unsharded_engine(tip, engine):
    rng = np.random.default_rng(fuzz_seed(50505))
    ts, cl, blob, blob_len = ipcfp.pack_event_claims(
        tip.parent_cids, tip.child_cid, tip.parent_epoch, tip.child_epoch, tip.claim_exec, tip.claim_event,
        tip.claim_emitter, tip.exec_order[tip.claim_exec.astype(np.int64)], tip.claim_ntopics, tip.claim_topics,
        tip.claim_datalen, tip.claim_data)
    n_rcpt = tip.params["n_receipts"]
    seen_err = seen_no_range = seen_plan_equal = 0
    for it in range(120):
        data, touched = mutate(tip, rng, n_flips=1 + it % 3)
        ctx = f"round {it}, blocks {touched}"
        with engine.witness(data, tip.off, tip.lens, tip.cids) as w:
            want = w.verify_event_claims(ts, cl, blob, blob_len)
            ws, whas, wm, _ = w.scan_events(tip.receipts_root, tip.topic0, tip.topic1, actor=tip.filter_actor, want_touched=False)
            G = 2 + it % 2
            plans = [w.shard_plan_tipset(tip.parent_cids, tip.child_cid, G, r) for r in range(G)]
        pk = ipcfp.PackedWitnessTables(data, tip.off, tip.lens, tip.cids, ingest=True)
        ipcfp.host_register(pk.data)
        try:
            status = np.full(len(cl), 255, dtype=np.uint8)
            has = np.zeros(n_rcpt, dtype=np.uint8)
            scans, n_matches, no_range = [], 0, False
            for r in range(G):
                st, sw, lo, hi, nr, stats = engine.witness_shard_pull(pk, tip.parent_cids, tip.child_cid, G, r)
                if st != 1:  # no receipts root to cut by: every claim of the unsharded run is an Err (the child header is needed by all)
                    assert sw is None and st == 65, ctx
                    no_range = True
                    break
                assert (lo, hi) == ipcfp.shard_range(nr, G, r), ctx
                pst, plo, phi, pnr, pids = plans[r]
                if pst == 1:
                    # The planner walks with the verifier's strict decoders and stops where they fail; the pull reads links
                    # leniently (a node that is no valid AMT node may still be expanded) — so on a corrupted witness the
                    # pulled shard is a SUPERSET of the planned one (equal on honest ones: tests/test_gpu_shard_pull.py),
                    # and what decides is that the verdicts below are the unsharded engine's.
                    assert (plo, phi, pnr) == (lo, hi, nr) and sw.block_count >= len(pids), (ctx, r, sw.block_count, len(pids))
                    present, _ = sw.has([bytes(tip.cids[i]) for i in pids])
                    assert present.all(), (ctx, r)
                    seen_plan_equal += int(sw.block_count == len(pids))
                pos, c_r, b_r, bl_r = ipcfp.route_event_claims(cl, blob, blob_len, lo, hi, r == G - 1)
                if len(pos):
                    status[pos.astype(np.int64)] = sw.verify_event_claims(ts, c_r, b_r, bl_r)
                sst, shas, sm, _ = sw.scan_events(tip.receipts_root, tip.topic0, tip.topic1, actor=tip.filter_actor, want_touched=False)
                scans.append((sst, sw.last_scan_phase()))
                assert (sst == 1) == (scans[-1][1] == 0), ctx
                if sst == 1:
                    has[lo: lo + len(shas)] = shas
                    n_matches += len(sm)
                sw.close()
        finally:
            ipcfp.host_unregister(pk.data)
        if no_range:
            # (every claim needs the child header and is an Err without it; the SCAN is handed its receipts root by the caller
            # and may well succeed on the whole witness — IPCFP_FUZZ_SEED=15, round 30 — but there is no range to cut it by)
            assert (want >= 64).all(), ctx
            seen_no_range += 1
            continue
        assert np.array_equal(status, want), (ctx, np.nonzero(status != want)[0][:6], status[status != want][:6], want[status != want][:6])
        # (an Err of the receipts enumeration in ANY shard precedes an Err of the events passes in a lower one: the unsharded
        # scan enumerates the whole tipset before it opens an events AMT — found by IPCFP_FUZZ_SEED=4, round 1)
        scan_status = ipcfp.merge_scan_status(scans)
        assert scan_status == ws, (ctx, scans, ws)
        if ws == 1:
            assert np.array_equal(has[: len(whas)], whas) and n_matches == len(wm), ctx
        seen_err += int((want >= 64).any())
    assert seen_err > 5 and seen_plan_equal > 50


def test_a_receipts_error_in_a_high_shard_precedes_an_events_error_in_a_low_one(tip, engine):
    """The scan enumerates the receipts of the WHOLE tipset before it opens an events AMT (the enumeration stands in for the
    reference's ChainGetParentReceipts call, src/proofs/events/generator.rs:199-204).  Two corruptions: a receipt of the first
    range names an events root nobody has (an Err of the events pass, ERR_MISSING_BLOCK), a receipts-AMT leaf of the last
    range does not decode (an Err of the enumeration, ERR_DECODE).  Unsharded, the decode error wins; the shards report
    (status, phase) and merge_scan_status names the same one — the lowest range's Err alone would be the other."""
    idx = np.array([100, 650], dtype=np.uint64)
    with engine.witness(tip.data, tip.off, tip.lens, tip.cids) as w:
        st, loc = w.amt_get(tip.receipts_root, 0, "receipt", idx)
        # a receipt of the first range that HAS an events root: [exit, ret, gas, link] ends in the link's digest
        k = 100
        while True:
            st, loc = w.amt_get(tip.receipts_root, 0, "receipt", np.array([k, 650], dtype=np.uint64))
            o = int(tip.off[loc[0]["block"]]) + int(loc[0]["off"])
            if tip.data[o + int(loc[0]["len"]) - 38 - 5] == 0xD8:
                break
            k += 1
    assert (st == 1).all() and k < 200
    data = tip.data.copy()
    data[o + int(loc[0]["len"]) - 1] ^= 0x5A                      # the events root of receipt k: a CID nobody has
    data[int(tip.off[loc[1]["block"]])] = 0x84                     # the leaf that holds receipt 650: `[bmap, links, values]` of 4
    with engine.witness(data, tip.off, tip.lens, tip.cids) as w:
        ws, _, _, _ = w.scan_events(tip.receipts_root, tip.topic0, tip.topic1, actor=tip.filter_actor, want_touched=False)
        assert ws == 66 and w.last_scan_phase() == ipcfp.SCAN_PHASE_RECEIPTS
    pk = ipcfp.PackedWitnessTables(data, tip.off, tip.lens, tip.cids, ingest=True)
    ipcfp.host_register(pk.data)
    try:
        scans = []
        for r in range(3):
            st, sw, lo, hi, nr, _ = engine.witness_shard_pull(pk, tip.parent_cids, tip.child_cid, 3, r)
            assert st == 1
            sst, _, _, _ = sw.scan_events(tip.receipts_root, tip.topic0, tip.topic1, actor=tip.filter_actor, want_touched=False)
            scans.append((sst, sw.last_scan_phase()))
            sw.close()
    finally:
        ipcfp.host_unregister(pk.data)
    assert scans == [(65, ipcfp.SCAN_PHASE_EVENTS), (1, 0), (66, ipcfp.SCAN_PHASE_RECEIPTS)], scans
    assert ipcfp.merge_scan_status(scans) == ws
