"""GPU: one tipset cut into G receipt-range shards — plan, sub-witness, routed claims, range-restricted
scan, verify, packed step message — run as G LOGICAL shards on the one GPU here; the merged result must equal the
unsharded engine's, bit for bit, and the oracle's.  The RCCL path is exercised with a 1-rank communicator (the
run-time binding of librccl, ncclCommInitRank, ncclAllGather on the engine's stream)."""
import numpy as np
import pytest

from conftest import fuzz_seed
import torch

import ipc_filecoin_proofs_amd as ipcfp
from ipc_filecoin_proofs_amd import shard
from tools.synth import Tipset

pytestmark = pytest.mark.gpu


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1)).cuda()


def run_shards(engine, tip, G, ts, cl, blob, comm=None, tamper=None):
    """Every shard of G in turn on this GPU; returns (merged result, list of TipsetShard counts)."""
    data = tip.data if tamper is None else tamper
    full = engine.witness(data, tip.off, tip.lens, tip.cids)
    shards = [shard.TipsetShard(engine, full, tip.parent_cids, tip.child_cid, tip.receipts_root, G, r) for r in range(G)]
    for s in shards:
        s.route(ts, cl, blob)
    layout = shard.Layout(max(s.n_claims for s in shards), max(s.hi - s.lo for s in shards),
                          max(s.witness.n for s in shards))
    msgs, ids = [], []
    for s in shards:
        d_cl, d_blob = dev(s.claims), dev(s.blob)
        d_status = torch.zeros(layout.w_status, dtype=torch.uint8, device="cuda")
        d_has = torch.zeros(layout.w_has, dtype=torch.uint8, device="cuda")
        d_hdr = dev(s.header())
        d_stage = torch.zeros(layout.bytes_per_rank, dtype=torch.uint8, device="cuda")
        d_recv = torch.zeros(layout.bytes_per_rank, dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()
        s.step(layout, comm, (tip.topic0, tip.topic1, tip.filter_actor), d_cl.data_ptr(), d_blob.data_ptr(),
               d_status.data_ptr(), d_has.data_ptr(), d_hdr.data_ptr(), d_stage.data_ptr(), d_recv.data_ptr())
        engine.sync()
        msgs.append(d_recv.cpu().numpy())
        ids.append(s.block_ids)
    merged = shard.merge(np.concatenate(msgs), layout, G, [s.positions for s in shards], len(cl), shards[0].n_receipts_total)
    counts = [(s.n_claims, s.hi - s.lo, s.witness.n) for s in shards]
    for s in shards:
        s.close()
    full.close()
    return merged, counts, ids


@pytest.fixture(scope="module")
def tip():
    return Tipset(n_receipts=40_000, n_parents=4, dup_permille=30, n_planted=40, max_events=4, no_events_permille=50,
                  variety=1, seed=fuzz_seed(77))


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
    with_data = liars[2::3][cl["data_len"][liars[2::3]] > 0]  # (a claim without data bytes has nothing to flip)
    blob[cl["data_off"][with_data]] ^= 0x40
    # claims that name no receipt at all: they belong to the LAST shard (shard.route_claims) and get the unsharded verdict
    cl["exec_index"][n - 2] = 40_000 + 3
    cl["exec_index"][n // 2] = (1 << 63) + 11
    return ts, cl, blob, blob_len


def test_shard_range_is_a_partition():
    for n in (0, 1, 7, 1000, 1_000_000, (1 << 64) - 1):
        for G in (1, 2, 3, 8, 64):
            prev = 0
            for r in range(G):
                lo, hi = ipcfp.shard_range(n, G, r)
                assert lo == prev and hi >= lo and hi - lo in (n // G, n // G + 1)
                prev = hi
            assert prev == n


@pytest.mark.parametrize("G", [1, 2, 3, 8])
def test_logical_shards_equal_the_unsharded_result(engine, oracle, tip, claims_packed, G):
    ts, cl, blob, blob_len = claims_packed
    with engine.witness(tip.data, tip.off, tip.lens, tip.cids) as w:
        want = w.verify_event_claims(ts, cl, blob, blob_len)
        ws, whas, wm, _ = w.scan_events(tip.receipts_root, tip.topic0, tip.topic1, actor=tip.filter_actor, want_touched=False)
    merged, counts, ids = run_shards(engine, tip, G, ts, cl, blob)
    assert np.array_equal(merged["status"], want) and (want != 1).sum() > 100 and (merged["status"] != 255).all()
    assert merged["scan_status"] == ws == 1 and np.array_equal(merged["has"], whas) and merged["n_matches"] == len(wm)
    assert merged["n_bad_cids"] == 0
    # the oracle agrees with both
    ost = oracle.store(tip.data, tip.off, tip.lens, tip.cids, threads=0)
    assert np.array_equal(ost.verify_event_claims_packed(ts, cl, blob, threads=0), want)
    ost.close()
    if G > 1:
        # placement: every rank holds the replicated part, and the shards together are the blocks a full pass touches
        common = set(ids[0].tolist())
        for x in ids[1:]:
            common &= set(x.tolist())
        assert len(common) > 0 and max(c[2] for c in counts) < tip.n_blocks
        per_rank_receipts = [c[1] for c in counts]
        assert sum(per_rank_receipts) == 40_000 and max(per_rank_receipts) - min(per_rank_receipts) <= 1


def test_a_tampered_block_is_reported_by_the_shard_that_holds_it(engine, tip, claims_packed):
    ts, cl, blob, _ = claims_packed
    data = tip.data.copy()
    # an events-AMT block of the last receipt range: flip a byte of its payload
    with engine.witness(tip.data, tip.off, tip.lens, tip.cids) as w:
        _, lo, hi, _, ids_last = w.shard_plan_tipset(tip.parent_cids, tip.child_cid, 4, 3)
        _, _, _, _, ids_first = w.shard_plan_tipset(tip.parent_cids, tip.child_cid, 4, 0)
    only_last = np.setdiff1d(ids_last, ids_first)
    victim = int(only_last[len(only_last) // 2])
    data[int(tip.off[victim]) + int(tip.lens[victim]) - 1] ^= 0x01
    merged, _, _ = run_shards(engine, tip, 4, ts, cl, blob, tamper=data)
    bad = [r["bad_cids"] for r in merged["per_rank"]]
    assert bad == [0, 0, 0, 1] and merged["n_bad_cids"] == 1


def test_misrouted_claim_is_an_error_not_a_verdict(engine, tip, claims_packed):
    """A claim about a receipt outside the shard's range cannot be judged there: its receipt path is not part of the
    shard's witness → ERR_MISSING_BLOCK (the router's mistake surfaces, it is never a silent Ok(false))."""
    ts, cl, blob, blob_len = claims_packed
    with engine.witness(tip.data, tip.off, tip.lens, tip.cids) as full:
        s = shard.TipsetShard(engine, full, tip.parent_cids, tip.child_cid, tip.receipts_root, 4, 1)
        honest = np.nonzero(np.isin(np.arange(len(cl)), np.arange(5, len(cl), 17), invert=True))[0]
        far = honest[honest > 35_000][:8]          # receipts of shard 3
        own = shard.route_claims(cl["exec_index"], s.lo, s.hi)[:8]
        c2, b2, bl2 = shard.subset_packed_claims(cl, blob, np.concatenate([own, far]))
        got = s.witness.verify_event_claims(ts, c2, b2, bl2)
        s.close()
    assert (got[8:] == 65).all() and set(got[:8].tolist()) <= {1, 8, 12, 13, 16}


def test_rccl_single_rank_allgather(engine):
    uid = ipcfp.comm_unique_id()
    assert len(uid) == 128
    comm = ipcfp.Comm(engine, uid, 1, 0)
    a = torch.arange(0, 200, dtype=torch.uint8, device="cuda")
    b = torch.full((56,), 7, dtype=torch.uint8, device="cuda")
    stage = torch.zeros(512, dtype=torch.uint8, device="cuda")
    recv = torch.zeros(512, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    comm.allgather_device(a.data_ptr(), recv.data_ptr(), 200)
    engine.sync()
    assert torch.equal(recv[:200], a)
    comm.allgather_segments([a.data_ptr(), b.data_ptr()], [200, 56], stage.data_ptr(), recv.data_ptr(), 512)
    engine.sync()
    want = torch.zeros(512, dtype=torch.uint8, device="cuda")
    want[:200] = a
    want[200:256] = 7
    assert torch.equal(recv, want) and torch.equal(stage, want)
    comm.close()


# ---- plan once, scatter: ipcfp_shard_plan_tipset_all + host-side cut / route (the shard is all a rank ever uploads) ----
@pytest.mark.parametrize("G", [1, 2, 3, 8])
def test_plan_all_equals_the_per_shard_plans(engine, tip, G):
    with engine.witness(tip.data, tip.off, tip.lens, tip.cids) as w:
        st, n_receipts, bounds, lists = w.shard_plan_tipset_all(tip.parent_cids, tip.child_cid, G)
        assert st == 1 and n_receipts == 40_000 and bounds[0] == 0 and bounds[G] == 40_000
        for r in range(G):
            st1, lo, hi, nr, ids = w.shard_plan_tipset(tip.parent_cids, tip.child_cid, G, r)
            assert st1 == 1 and (lo, hi) == (int(bounds[r]), int(bounds[r + 1])) and nr == n_receipts
            assert np.array_equal(lists[r], ids)


def test_plan_all_reports_the_first_error_like_the_single_plan(engine, tip):
    # the child header is not in the witness: the traversal stops where the reference's `?` would
    keep = np.ones(tip.n_blocks, dtype=bool)
    with engine.witness(tip.data, tip.off, tip.lens, tip.cids) as w:
        has, ids = w.has([tip.child_cid])
        keep[int(ids[0])] = False
    sub = ipcfp.witness_cut_host(tip.data, tip.off, tip.lens, tip.cids, np.nonzero(keep)[0].astype(np.uint32))
    with engine.witness(*sub) as w:
        st, n_receipts, bounds, lists = w.shard_plan_tipset_all(tip.parent_cids, tip.child_cid, 4)
        st1 = w.shard_plan_tipset(tip.parent_cids, tip.child_cid, 4, 2)[0]
    assert st == st1 == 65 and n_receipts == 0 and all(len(x) == 0 for x in lists)


@pytest.mark.parametrize("G", [1, 2, 3, 8])
def test_scattered_shards_uploaded_from_host_equal_the_unsharded_result(engine, oracle, tip, claims_packed, G):
    """The whole witness is resident ONCE (the planner); every shard then exists only as its own host buffers, is
    uploaded by itself, verified, and the merged verdicts equal the unsharded engine's and the oracle's."""
    ts, cl, blob, blob_len = claims_packed
    with engine.witness(tip.data, tip.off, tip.lens, tip.cids) as w:
        want = w.verify_event_claims(ts, cl, blob, blob_len)
        ws, whas, wm, _ = w.scan_events(tip.receipts_root, tip.topic0, tip.topic1, actor=tip.filter_actor, want_touched=False)
        plan = shard.TipsetPlan(w, tip.parent_cids, tip.child_cid, G)
    status = np.full(len(cl), 255, dtype=np.uint8)
    has = np.zeros(plan.n_receipts, dtype=np.uint8)
    n_matches, uploaded = 0, []
    for r in range(G):
        sub = plan.cut(r, tip.data, tip.off, tip.lens, tip.cids)
        pos, c_r, b_r, bl_r = plan.route(r, cl, blob, blob_len)
        s = shard.TipsetShard.from_plan(engine, plan, r, sub, tip.receipts_root)
        uploaded.append(sub[0].size)
        st_cid, n_bad = s.witness.verify_cids()
        assert n_bad == 0
        status[pos.astype(np.int64)] = s.witness.verify_event_claims(ts, c_r, b_r, bl_r)
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
    if G == 8:  # what a rank uploads shrinks with G (the message AMTs and headers are the replicated floor)
        assert max(uploaded) < 0.45 * tip.data.size


def test_shards_of_a_tipset_with_tall_event_amts(engine, oracle):
    """Events AMTs of bit width 1 and height >= 8 (taller than the planner's explicit stack): every shard's witness must
    still hold the whole tree of each of its receipts — merged statuses == unsharded == oracle."""
    tall = Tipset(n_receipts=60, n_planted=6, variety=1, max_events=700, events_bit_width=1, seed=79)
    ts, cl, blob, blob_len = ipcfp.pack_event_claims(
        tall.parent_cids, tall.child_cid, tall.parent_epoch, tall.child_epoch, tall.claim_exec, tall.claim_event,
        tall.claim_emitter, tall.exec_order[tall.claim_exec.astype(np.int64)], tall.claim_ntopics, tall.claim_topics,
        tall.claim_datalen, tall.claim_data)
    assert int(tall.claim_event.max()) >= 512
    with engine.witness(tall.data, tall.off, tall.lens, tall.cids) as w:
        want = w.verify_event_claims(ts, cl, blob, blob_len)
        ws, whas, wm, _ = w.scan_events(tall.receipts_root, tall.topic0, tall.topic1, actor=tall.filter_actor, want_touched=False)
    ost = oracle.store(tall.data, tall.off, tall.lens, tall.cids)
    assert np.array_equal(ost.verify_event_claims_packed(ts, cl, blob, threads=1), want) and (want == 1).sum() >= 20
    ost.close()
    for G in (2, 3):
        merged, counts, ids = run_shards(engine, tall, G, ts, cl, blob)
        assert np.array_equal(merged["status"], want)
        assert merged["scan_status"] == ws == 1 and np.array_equal(merged["has"], whas) and merged["n_matches"] == len(wm)
