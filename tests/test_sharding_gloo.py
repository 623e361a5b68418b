"""CPU, world_size 2, gloo: the host half of the N > 1 path (ipc_filecoin_proofs_amd.shard) — receipt ranges from
the C ABI's ipcfp_shard_range, claim routing, the step message's byte layout agreed over the host channel, ONE
all-gather, and the merge by claim position / receipt index.  There is no GPU here, so each rank fills its message
with what the engine would write there, computed by the CPU oracle for that rank's range (the checker); the merged
result must equal the unsharded oracle's.  The engine's own shard plan, sub-witness and range-restricted kernels
are tested on the GPU (tests/test_gpu_sharding.py, G logical shards on one device)."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_receipts, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("OMP_WAIT_POLICY", "passive")
    os.environ.setdefault("IPCFP_ORACLE_MAX_THREADS", "2")
    import torch
    import torch.distributed as dist

    import ipc_filecoin_proofs_amd as ipcfp
    import oracle_lib
    from ipc_filecoin_proofs_amd import shard
    from tools.synth import Tipset

    dist.init_process_group("gloo", rank=rank, world_size=world)
    tip = Tipset(n_receipts=n_receipts, n_planted=5, variety=1, seed=4242)
    orc = oracle_lib.load()
    st = orc.store(tip.data, tip.off, tip.lens, tip.cids)
    ts, cl, blob, blob_len = ipcfp.pack_event_claims(
        tip.parent_cids, tip.child_cid, tip.parent_epoch, tip.child_epoch, tip.claim_exec, tip.claim_event,
        tip.claim_emitter, tip.exec_order[tip.claim_exec.astype(np.int64)], tip.claim_ntopics, tip.claim_topics,
        tip.claim_datalen, tip.claim_data)
    cl["exec_index"][::5] += 1  # a fifth of the claims lie: positions must survive routing and merging
    cl["exec_index"][-3:] = [n_receipts, n_receipts + 7, (1 << 63) + 5]  # ... and some name no receipt at all
    want = st.verify_event_claims_packed(ts, cl, blob, threads=1)
    ws, whas, wtrip, _ = st.scan_events(tip.receipts_root, tip.topic0, tip.topic1, actor=tip.filter_actor, want_touched=False)
    # ---- this rank's share ----
    lo, hi = ipcfp.shard_range(n_receipts, world, rank)
    pos = shard.route_claims(cl["exec_index"], lo, hi, last=(rank == world - 1))
    c_r, b_r, bl_r = shard.subset_packed_claims(cl, blob, pos)
    local_status = st.verify_event_claims_packed(ts, c_r, b_r, threads=1)  # what the engine writes for these claims
    n_blocks_r = 100 + 13 * rank                                           # block counts differ between ranks
    ok_bits = np.packbits(np.ones(n_blocks_r, dtype=np.uint8), bitorder="little")
    n_matches_r = int(((wtrip[:, 0] >= lo) & (wtrip[:, 0] < hi)).sum())

    def allreduce_max(v):
        t = torch.from_numpy(v.copy())
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return t.numpy()

    layout = shard.Layout.agree((len(pos), hi - lo, n_blocks_r), allreduce_max)
    msg = np.zeros(layout.bytes_per_rank, dtype=np.uint8)
    msg[:shard.HEADER_BYTES] = np.array([len(pos), hi - lo, n_blocks_r, ws, n_matches_r, 0, lo, hi], dtype=np.uint64).view(np.uint8)
    msg[layout.off_status: layout.off_status + len(pos)] = local_status
    msg[layout.off_has: layout.off_has + (hi - lo)] = whas[lo:hi]
    msg[layout.off_bits: layout.off_bits + len(ok_bits)] = ok_bits
    gathered = torch.empty(world * layout.bytes_per_rank, dtype=torch.uint8)
    dist.all_gather_into_tensor(gathered, torch.from_numpy(msg))  # the one collective
    all_pos = shard.route_all(cl["exec_index"], n_receipts, world)
    # every claim has exactly one owner — the ones that point past the last receipt go to the last rank — so the merged
    # status bytes are the unsharded verifier's for EVERY claim
    merged = shard.merge(gathered.numpy(), layout, world, all_pos, len(cl), n_receipts)
    owned = np.zeros(len(cl), dtype=np.int64)
    for p in all_pos:
        owned[p] += 1
    ok = ((owned == 1).all() and np.array_equal(merged["status"], want) and (merged["status"] != 255).all() and
          np.array_equal(merged["has"], whas) and merged["n_matches"] == len(wtrip) and merged["scan_status"] == ws and
          merged["n_bad_cids"] == 0 and [r["blocks"] for r in merged["per_rank"]] == [100 + 13 * r for r in range(world)])
    np.save(os.path.join(out_dir, f"r{rank}.npy"), np.array([ok, int((owned != 1).sum()), int((merged["status"] != 1).sum()), len(cl)]))
    st.close()
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_receipts", [301, 64])
def test_two_rank_route_gather_merge(tmp_path, n_receipts):
    import torch.multiprocessing as mp

    port = _free_port()
    mp.spawn(_worker, args=(2, port, n_receipts, str(tmp_path)), nprocs=2, join=True)
    for r in range(2):
        ok, unowned, bad, n_claims = np.load(tmp_path / f"r{r}.npy")
        assert ok and unowned == 0 and bad >= n_claims // 5 and n_claims > n_receipts // 2


def test_shard_range_partitions():
    import ipc_filecoin_proofs_amd as ipcfp

    assert [ipcfp.shard_range(10, 3, r) for r in range(3)] == [(0, 3), (3, 6), (6, 10)]
    assert [ipcfp.shard_range(2, 4, r) for r in range(4)] == [(0, 0), (0, 1), (1, 1), (1, 2)]
    assert [ipcfp.shard_range(0, 2, r) for r in range(2)] == [(0, 0), (0, 0)]
    for n in (1_000_003, (1 << 64) - 1):
        b = [ipcfp.shard_range(n, 8, r) for r in range(8)]
        assert b[0][0] == 0 and b[-1][1] == n and all(x[1] == y[0] for x, y in zip(b, b[1:]))
        assert max(h - l for l, h in b) - min(h - l for l, h in b) <= 1


def test_scan_status_of_merged_shards_prefers_the_enumeration_phase():
    """The unsharded scan enumerates every receipt before it opens an events AMT: an Err of the enumeration in ANY shard
    precedes an Err of the events passes in a lower one; inside a phase the lowest receipt range decides; a header word is
    status | phase << 8 (ipcfp_scan_events_device summary_d)."""
    from ipc_filecoin_proofs_amd import shard
    import ipc_filecoin_proofs_amd as ipcfp

    R, E = ipcfp.SCAN_PHASE_RECEIPTS, ipcfp.SCAN_PHASE_EVENTS
    assert ipcfp.merge_scan_status([(1, 0), (1, 0)]) == 1
    assert ipcfp.merge_scan_status([(65, E), (1, 0), (66, R)]) == 66
    assert ipcfp.merge_scan_status([(65, E), (66, E), (1, 0)]) == 65
    assert ipcfp.merge_scan_status([(1, 0), (70, R), (66, R)]) == 70
    assert ipcfp.merge_scan_status([(1, 0), (67, 0)]) == 67  # (a message written without a phase: any Err, in range order)
    lay = shard.Layout(4, 8, 40)
    msg = np.zeros((3, lay.bytes_per_rank), dtype=np.uint8)
    for r, (st, ph) in enumerate([(65, E), (1, 0), (66, R)]):
        hdr = msg[r, : shard.HEADER_BYTES].view(np.uint64)
        hdr[:] = [0, 2, 40, st | (ph << 8), 0, 0, 2 * r, 2 * r + 2]
    merged = shard.merge(msg.reshape(-1), lay, 3, [np.zeros(0, dtype=np.int64)] * 3, 0, 6)
    assert merged["scan_status"] == 66 and [p["scan_phase"] for p in merged["per_rank"]] == [E, 0, R]


def test_layout_and_claim_subsetting():
    from ipc_filecoin_proofs_amd import shard
    import ipc_filecoin_proofs_amd as ipcfp

    lay = shard.Layout(1001, 125_000, 160_001)
    assert lay.off_status == 64 and lay.w_status % 16 == 0 and lay.w_status >= 1001
    assert lay.off_has == 64 + lay.w_status and lay.off_bits == lay.off_has + lay.w_has
    assert lay.w_bits >= (160_001 + 31) // 32 * 4 and lay.bytes_per_rank == lay.off_bits + lay.w_bits
    rng = np.random.default_rng(5)
    n = 50
    cl = np.zeros(n, dtype=ipcfp.CLAIM_DTYPE)
    cl["n_topics"] = rng.integers(0, 5, n)
    cl["data_len"] = rng.integers(0, 40, n)
    sizes = cl["n_topics"].astype(np.int64) * 33 + cl["data_len"]
    starts = np.cumsum(sizes) - sizes
    cl["topics_off"] = starts
    cl["data_off"] = starts + cl["n_topics"].astype(np.int64) * 33
    blob = rng.integers(0, 256, int(sizes.sum()) + 64, dtype=np.uint8)
    pos = np.array([3, 4, 17, 49, 0])
    c2, b2, bl2 = shard.subset_packed_claims(cl, blob, pos)
    assert bl2 == int(sizes[pos].sum()) and len(b2) == bl2 + 64
    for k, p in enumerate(pos):
        t0, t1 = int(cl["topics_off"][p]), int(cl["topics_off"][p]) + 33 * int(cl["n_topics"][p])
        assert b2[int(c2["topics_off"][k]): int(c2["topics_off"][k]) + (t1 - t0)].tobytes() == blob[t0:t1].tobytes()
        d0 = int(cl["data_off"][p])
        dl = int(cl["data_len"][p])
        assert b2[int(c2["data_off"][k]): int(c2["data_off"][k]) + dl].tobytes() == blob[d0:d0 + dl].tobytes()
