"""One tipset over the GPUs of a node: host-side orchestration above the C ABI.

The reference verifies a bundle proof by proof (src/proofs/verifier.rs:19-28,49-54,
src/proofs/events/verifier.rs:62-71).  Here rank r of G

  * plans   — `ipcfp_shard_plan_tipset`: the blocks it needs for the receipts [lo, hi) of the tipset (their events
              AMTs + the receipts-AMT paths; headers, TxMeta and message AMTs replicated),
  * places  — `ipcfp_witness_create_subset`: its own witness, tagged with the range,
              (or both in one call that needs nothing but the bundle in the rank's own host memory:
              `ipcfp_witness_create_shard_pull`, `TipsetShard.from_pull`)
  * routes  — the claims whose exec_index falls in [lo, hi),
  * steps   — CID index, K1, range-restricted scan, verify_event_proof of its claims — all through the same
              entry points a single-GPU host uses,
  * gathers — ONE `ncclAllGather` (RCCL, called directly by libipcfp.so) of
              [header | status bytes | has-match map | CID bitmap] per rank,

and `merge` lays the gathered messages out by claim position / receipt index.  Everything on the data path is
behind the C ABI; this module only sequences the calls (a Rust host would do the same over ffi.rs).  The widths of
the message parts must be agreed once at setup over the host's own channel (`Layout.agree`).
"""
from __future__ import annotations

import numpy as np

from . import binding as B

HEADER_BYTES = 64  # u64 x 8: n_claims, n_receipts, n_blocks, scan_status | scan_phase << 8, n_matches, n_bad_cids(filled by merge), lo, hi


class Layout:
    """Byte layout of one rank's message: widths are the maxima over ranks, so every rank sends the same size."""

    def __init__(self, max_claims: int, max_receipts: int, max_blocks: int):
        self.w_status = (int(max_claims) + 15) & ~15
        self.w_has = (int(max_receipts) + 15) & ~15
        self.w_bits = ((int(max_blocks) + 31) // 32 * 4 + 15) & ~15
        self.off_status = HEADER_BYTES
        self.off_has = self.off_status + self.w_status
        self.off_bits = self.off_has + self.w_has
        self.bytes_per_rank = self.off_bits + self.w_bits

    @staticmethod
    def agree(local_counts, allreduce_max):
        """local_counts = (n_claims, n_receipts, n_blocks) of this rank; allreduce_max: the host's channel
        (e.g. torch.distributed all_reduce MAX over a 3-vector, or the identity for one rank)."""
        return Layout(*[int(x) for x in allreduce_max(np.asarray(local_counts, dtype=np.int64))])


def route_claims(exec_index: np.ndarray, lo: int, hi: int, last: bool = False) -> np.ndarray:
    """Positions of the claims this rank verifies: exec_index in [lo, hi).  The LAST rank (`last`) also takes the claims
    whose exec_index names no receipt at all (>= hi = the receipt count): every claim has exactly one owner.  Such a
    claim never reaches a receipt — steps 1-3 of verify_single_proof (events/verifier.rs:92-204: trust anchors, header
    consistency, the execution order) run on data every rank holds and settle it with the same status the unsharded
    verifier gives (FALSE_EXEC_INDEX / FALSE_MSG_NOT_IN_EXEC or an earlier one)."""
    e = np.asarray(exec_index, dtype=np.uint64)
    sel = e >= np.uint64(lo)
    if not last:
        sel &= e < np.uint64(hi)
    return np.nonzero(sel)[0]


def route_all(exec_index: np.ndarray, n_receipts: int, n_shards: int):
    """route_claims for every rank of n_shards (what a host needs to merge the gathered status bytes)."""
    return [route_claims(exec_index, *B.shard_range(n_receipts, n_shards, r), last=(r == n_shards - 1)) for r in range(n_shards)]


def merge(gathered: np.ndarray, layout: Layout, n_ranks: int, claim_positions, n_claims_total: int, n_receipts_total: int):
    """gathered: u8[n_ranks * bytes_per_rank] (host copy of the all-gather result).
    claim_positions[r]: positions (in the caller's claim order) of the claims rank r verified.
    → dict(status u8[n_claims_total], has u8[n_receipts_total], scan_status, n_matches, n_bad_cids, per_rank)."""
    g = np.asarray(gathered, dtype=np.uint8).reshape(n_ranks, layout.bytes_per_rank)
    status = np.full(n_claims_total, 255, dtype=np.uint8)  # (every claim has an owner: nothing stays 255 — route_claims)
    has = np.zeros(n_receipts_total, dtype=np.uint8)
    n_matches, n_bad, per_rank, scans = 0, 0, [], []
    for r in range(n_ranks):
        hdr = g[r, :HEADER_BYTES].view(np.uint64)
        nc, nr, nb, sword, nm, _, lo, hi = [int(x) for x in hdr]
        sst, phase = sword & 0xFF, (sword >> 8) & 0xFF  # (ipcfp_scan_events_device summary_d: status | phase << 8)
        scans.append((sst, phase))
        pos = np.asarray(claim_positions[r])
        assert len(pos) == nc, "claim routing differs from what the rank reported"
        status[pos] = g[r, layout.off_status: layout.off_status + nc]
        has[lo: lo + nr] = g[r, layout.off_has: layout.off_has + nr]
        bits = np.unpackbits(g[r, layout.off_bits: layout.off_bits + (nb + 31) // 32 * 4], bitorder="little")[:nb]
        bad = int(nb - bits.sum())
        n_matches += nm
        n_bad += bad
        per_rank.append({"claims": nc, "receipts": nr, "blocks": nb, "scan_status": sst, "scan_phase": phase, "matches": nm,
                         "bad_cids": bad, "lo": lo, "hi": hi})
    # an Err of the receipts enumeration (any shard: the unsharded scan enumerates the whole tipset first) before an Err of
    # the events passes; inside a phase the lowest receipt range's
    scan_status = B.merge_scan_status(scans)
    return {"status": status, "has": has, "scan_status": scan_status, "n_matches": n_matches, "n_bad_cids": n_bad,
            "per_rank": per_rank}


def gather_ranges(blob: np.ndarray, starts: np.ndarray, lens: np.ndarray) -> np.ndarray:
    """Concatenation of blob[starts[i] : starts[i] + lens[i]] (vectorised)."""
    lens = np.asarray(lens, dtype=np.int64)
    starts = np.asarray(starts, dtype=np.int64)
    total = int(lens.sum())
    if total == 0:
        return np.zeros(0, dtype=np.uint8)
    first = np.cumsum(lens) - lens
    idx = np.repeat(starts - first, lens) + np.arange(total, dtype=np.int64)
    return np.asarray(blob, dtype=np.uint8)[idx]


def subset_packed_claims(claims: np.ndarray, blob: np.ndarray, positions: np.ndarray):
    """The packed claims at `positions` with a blob of their own (ipcfp_event_claim_t[], include/ipcfp.h):
    → (claims copy with topics_off / data_off rewritten, blob u8[] + 64 B slack, blob_len)."""
    c = np.ascontiguousarray(claims[positions]).copy()
    tl = c["n_topics"].astype(np.int64) * 33
    dl = c["data_len"].astype(np.int64)
    t_bytes = gather_ranges(blob, c["topics_off"], tl)
    d_bytes = gather_ranges(blob, c["data_off"], dl)
    t_first = np.cumsum(tl) - tl
    d_first = np.cumsum(dl) - dl + int(tl.sum())
    c["topics_off"] = t_first
    c["data_off"] = d_first
    out = np.zeros(len(t_bytes) + len(d_bytes) + 64, dtype=np.uint8)
    out[: len(t_bytes)] = t_bytes
    out[len(t_bytes): len(t_bytes) + len(d_bytes)] = d_bytes
    return c, out, len(t_bytes) + len(d_bytes)


class TipsetPlan:
    """PLAN ONCE, SCATTER: every shard's block list and receipt range of one tipset from ONE call
    (`ipcfp_shard_plan_tipset_all`), made where the whole witness is resident — the bundle's producer, or one rank.
    `cut` / `route` are host-only: what rank r uploads is its own blocks and claims, nothing else."""

    def __init__(self, full: B.Witness, parent_cids, child_cid: bytes, n_shards: int):
        st, n_receipts, bounds, lists = full.shard_plan_tipset_all(parent_cids, child_cid, n_shards)
        if st != 1:
            raise B.EngineError(f"shard plan failed with status {st}")
        self.n_shards, self.n_receipts, self.bounds, self.block_ids = int(n_shards), n_receipts, bounds, lists
        self.parent_cids, self.child_cid = parent_cids, child_cid

    @classmethod
    def from_parts(cls, n_shards: int, n_receipts: int, bounds, block_ids, parent_cids, child_cid: bytes):
        """A plan received over the host's own channel (rank 0 planned, the others got the lists)."""
        self = cls.__new__(cls)
        self.n_shards, self.n_receipts = int(n_shards), int(n_receipts)
        self.bounds = np.asarray(bounds, dtype=np.uint64)
        self.block_ids = [np.asarray(x, dtype=np.uint32) for x in block_ids]
        self.parent_cids, self.child_cid = parent_cids, child_cid
        return self

    def range(self, r: int):
        return int(self.bounds[r]), int(self.bounds[r + 1])

    def cut(self, r: int, data, off, lens, cids40):
        """Shard r's witness as host arrays (data, off, lens, cids) — `ipcfp_witness_cut_host`."""
        return B.witness_cut_host(data, off, lens, cids40, self.block_ids[r])

    def route(self, r: int, claims: np.ndarray, blob: np.ndarray, blob_len: int):
        """Shard r's claims → (positions, claims, blob, blob_len) — `ipcfp_route_event_claims`."""
        lo, hi = self.range(r)
        return B.route_event_claims(claims, blob, blob_len, lo, hi, r == self.n_shards - 1)


class TipsetShard:
    """Rank `shard` of `n_shards` for one tipset.  `full` is a Witness holding the whole tipset on this rank's GPU
    (setup only: it may be closed once the shard exists) — or see `from_plan`, which never holds more than the shard."""

    @classmethod
    def from_plan(cls, eng: B.Engine, plan: "TipsetPlan", shard: int, sub, receipts_root: bytes, witness=None):
        """Rank `shard` from a plan made elsewhere: `sub` = plan.cut(shard, …) in host memory is uploaded (or `witness`,
        an already created Witness of exactly those blocks, is adopted) and tagged with the receipt range."""
        self = cls.__new__(cls)
        self.eng, self.n_shards, self.shard = eng, plan.n_shards, int(shard)
        self.lo, self.hi = plan.range(shard)
        self.n_receipts_total, self.block_ids = plan.n_receipts, plan.block_ids[shard]
        self.witness = witness if witness is not None else eng.witness(*sub)
        # (the last shard enumerates whatever the tree holds from its lo on: the root's count is not checked by Amt::load, and
        # a claim beyond it is the last rank's)
        self.witness.set_receipt_range(self.lo, self.hi if self.shard + 1 < self.n_shards else (1 << 64) - 1)
        self.receipts_root = bytes(receipts_root)
        self.parent_cids, self.child_cid = plan.parent_cids, plan.child_cid
        return self

    @classmethod
    def from_pull(cls, eng: B.Engine, packed: "B.PackedWitnessTables", parent_cids, child_cid: bytes, receipts_root: bytes,
                  n_shards: int, shard: int):
        """Rank `shard` plans and fetches ITS shard by itself out of the bundle `packed` in (device-readable) host memory —
        `ipcfp_witness_create_shard_pull`: nobody holds the whole witness, no plan comes from elsewhere."""
        st, w, lo, hi, n_receipts, stats = eng.witness_shard_pull(packed, parent_cids, child_cid, n_shards, shard)
        if st != 1:
            raise B.EngineError(f"shard pull failed with status {st}")
        self = cls.__new__(cls)
        self.eng, self.n_shards, self.shard = eng, int(n_shards), int(shard)
        self.lo, self.hi, self.n_receipts_total, self.block_ids = lo, hi, n_receipts, None
        self.witness, self.pull_stats = w, stats
        self.receipts_root = bytes(receipts_root)
        self.parent_cids, self.child_cid = parent_cids, child_cid
        return self

    def __init__(self, eng: B.Engine, full: B.Witness, parent_cids, child_cid: bytes, receipts_root: bytes,
                 n_shards: int, shard: int):
        self.eng, self.n_shards, self.shard = eng, int(n_shards), int(shard)
        st, lo, hi, n_receipts, ids = full.shard_plan_tipset(parent_cids, child_cid, n_shards, shard)
        if st != 1:
            raise B.EngineError(f"shard plan failed with status {st}")
        self.lo, self.hi, self.n_receipts_total, self.block_ids = lo, hi, n_receipts, ids
        self.witness = full.subset(ids, lo, hi if self.shard + 1 < self.n_shards else (1 << 64) - 1)  # (the last shard: from_plan)
        self.receipts_root = bytes(receipts_root)
        self.parent_cids, self.child_cid = parent_cids, child_cid

    def route(self, tipsets: np.ndarray, claims: np.ndarray, blob: np.ndarray, blob_len=None):
        """This rank's share of a packed claim batch: claims whose exec_index is one of its receipts."""
        self.tipsets = np.ascontiguousarray(tipsets)
        self.positions, self.claims, self.blob, self.blob_len = B.route_event_claims(
            claims, blob, max(len(blob) - 64, 0) if blob_len is None else blob_len, self.lo, self.hi,
            self.shard == self.n_shards - 1)
        self.n_claims = len(self.positions)
        return self.claims, self.blob, self.blob_len

    @property
    def counts(self):
        return (getattr(self, "n_claims", 0), self.hi - self.lo, self.witness.n)

    def header(self) -> np.ndarray:
        """The static part of the step message's header (scan status / match count are written by the scan)."""
        return np.array([self.n_claims, self.hi - self.lo, self.witness.n, 0, 0, 0, self.lo, self.hi], dtype=np.uint64)

    def step(self, layout: Layout, comm, filt, claims_ptr: int, blob_ptr: int, status_ptr: int, has_ptr: int,
             header_ptr: int, staging_ptr: int, recv_ptr: int):
        """One verification pass of this shard + the one collective.  All pointers are HBM buffers of the caller:
        claims/blob (the routed claims), status (layout.w_status bytes), has (layout.w_has bytes), header (64 B,
        initialised from header()), staging (bytes_per_rank), recv (n_ranks * bytes_per_rank).  Asynchronous
        after the verify call returns: the gathered result is complete after eng.sync()."""
        topic0, topic1, actor = filt
        w = self.witness
        w.rebuild_index()                                                                    # K4
        w.verify_cids_async()                                                                # K1 (second stream)
        # verify first: it walks the receipts AMT along with the message AMTs and tabulates the events (counting the
        # matches of the last scan's filter); the scan that follows finds its enumeration and its PASS 1 done
        if self.n_claims:
            w.verify_event_claims_device(self.tipsets, claims_ptr, self.n_claims, blob_ptr, self.blob_len, status_ptr)
        st, nr, nm = w.scan_events_device(self.receipts_root, topic0, topic1, actor, has_ptr, layout.w_has,
                                          summary_ptr=header_ptr + 24)                       # K6, range-restricted
        bits_bytes = (w.n + 31) // 32 * 4
        B.allgather_segments(self.eng, comm, [header_ptr, status_ptr, has_ptr, w.cid_bitmap_ptr],
                             [HEADER_BYTES, layout.w_status, layout.w_has, bits_bytes], staging_ptr, recv_ptr,
                             layout.bytes_per_rank)                                          # the ONE collective
        return st, nr, nm

    def close(self):
        self.witness.close()
