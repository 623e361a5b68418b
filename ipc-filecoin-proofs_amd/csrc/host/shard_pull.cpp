// csrc/host/shard_pull.cpp — ipcfp_witness_create_shard_pull: rank r of G builds ITS shard of one tipset straight out of
// the bundle in host memory.  No rank ever holds the whole witness in HBM, nothing is planned elsewhere, the host cuts
// no block lists: the bundle's CID table is uploaded (36 bytes per block), the device finds the shard's blocks level by
// level and reads exactly those out of the host buffer (kernels/shard_pull.hip).
//
// The loops being cut: src/proofs/verifier.rs:19-28,49-54, src/proofs/events/verifier.rs:62-71; what a shard holds:
// src/proofs/events/generator.rs:122-177,195-301 (the receipts [lo, hi), their events AMTs), src/proofs/events/utils.rs:48-94
// (headers, TxMeta, message AMTs: replicated, the execution order is global).
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <new>
#include <vector>

#include "../common.h"
#include "../kernels/launch.h"
#include "../kernels/shard_pull.h"

using namespace ipcfp;

namespace ipcfp {
int witness_finish_create(ipcfp_ctx* ctx, ipcfp_witness* w, const uint8_t* raw_bytes_d, const uint64_t* raw_off_d,
                          const uint32_t* len_d_src, const uint8_t* cids_d_src);
CidKey key_from_slot(const uint8_t* slot40);
}  // namespace ipcfp

extern "C" {

int ipcfp_host_register(void* p, uint64_t bytes) {
    if (!p || !bytes || (reinterpret_cast<uintptr_t>(p) & 4095u)) return IPCFP_E_INVALID;  // (a buffer that owns its pages: ipcfp.h)
    return hipHostRegister(p, size_t(bytes), hipHostRegisterDefault) == hipSuccess ? IPCFP_OK : IPCFP_E_HIP;
}

int ipcfp_host_unregister(void* p) {
    if (!p) return IPCFP_E_INVALID;
    return hipHostUnregister(p) == hipSuccess ? IPCFP_OK : IPCFP_E_HIP;
}

int ipcfp_witness_create_shard_pull(ipcfp_ctx_t* ctx, const uint8_t* bytes, uint64_t nbytes, const uint32_t* len,
                                    const uint8_t* digests32, uint64_t n, const uint8_t* cid_prefix, uint32_t prefix_len,
                                    const uint32_t* esc_index, const uint8_t* esc_cids40, uint64_t n_esc,
                                    const uint8_t* parent_cids40, uint32_t n_parents, const uint8_t* child_cid40,
                                    uint32_t n_shards, uint32_t shard, ipcfp_status_t* status_out, uint64_t* receipt_lo,
                                    uint64_t* receipt_hi, uint64_t* n_receipts, ipcfp_shard_pull_stats_t* stats,
                                    ipcfp_witness_t** out) {
    if (!ctx || !out || !status_out || !receipt_lo || !receipt_hi || !child_cid40 || (n_parents && !parent_cids40) || n_shards == 0 ||
        shard >= n_shards)
        return IPCFP_E_INVALID;
    *out = nullptr;
    *status_out = IPCFP_ST_ERR;
    *receipt_lo = *receipt_hi = 0;
    if (n_receipts) *n_receipts = 0;
    if (stats) std::memset(stats, 0, sizeof *stats);
    if (n == 0 || !len || !digests32 || !bytes) return set_error(ctx, IPCFP_E_INVALID, "empty bundle or null table pointer");
    if (prefix_len > 8 || (prefix_len && !cid_prefix)) return set_error(ctx, IPCFP_E_INVALID, "CID prefix longer than 8 bytes");
    if (n_esc && (!esc_index || !esc_cids40)) return set_error(ctx, IPCFP_E_INVALID, "null escape table");
    if (n >= 0xffffffffull || n_esc > n) return set_error(ctx, IPCFP_E_UNSUPPORTED, "more than 2^32-2 blocks");
    if (n_parents > IPCFP_MAX_PARENTS_WIDE) return set_error(ctx, IPCFP_E_UNSUPPORTED, "more than %u parent blocks", unsigned(IPCFP_MAX_PARENTS_WIDE));
    if (!ctx->mailbox) return set_error(ctx, IPCFP_E_UNSUPPORTED, "the context has no mailbox page (IPCFP_MAILBOX=0)");
    IPCFP_ENTER(ctx);
    // the device reads the blocks itself: the buffer must be mapped for it (hipHostMalloc / hipHostRegister / ipcfp_host_register)
    uint8_t* bytes_dev = nullptr;
    if (hipHostGetDevicePointer(reinterpret_cast<void**>(&bytes_dev), const_cast<uint8_t*>(bytes), 0) != hipSuccess || !bytes_dev) {
        (void)hipGetLastError();
        return set_error(ctx, IPCFP_E_INVALID, "the bundle's bytes are not device-readable host memory: register the buffer first (ipcfp_host_register)");
    }
    const auto t_start = std::chrono::steady_clock::now();
    const uint32_t N = uint32_t(n);
    // ---- the bundle's tables: lengths, digests → offsets in the host buffer, 40-byte CID slots, the index over them ----
    DevBuf<uint32_t> glen, esc_i, slots, resident, pulled, copy_len, role;
    DevBuf<uint64_t> goff, scan_scratch, stage_off, copy_src, copy_dst;
    DevBuf<uint8_t> dig, esc_c, gcids, stage;
    DevBuf<PullItem> fa, fb;
    DevBuf<PullCtl> ctl;
    IPCFP_HIP(ctx, glen.alloc(N));
    IPCFP_HIP(ctx, goff.alloc(N));
    IPCFP_HIP(ctx, gcids.alloc(size_t(N) * IPCFP_CID_SLOT));
    IPCFP_HIP(ctx, scan_scratch.alloc(size_t(div_up(N, 1024)) + 2));
    int rc = upload(ctx, glen.p, len, size_t(N) * 4, ctx->stream);
    // The digest table is read ONCE, by the kernel that widens it to 40-byte slots: when it lies in device-readable host
    // memory too (the same ingest buffer) that kernel reads it where it is — 41 MB for a 1M-receipt tipset at the link's
    // rate instead of a staged copy in front of the kernel.
    const uint8_t* dig_src = nullptr;
    {
        uint8_t* dp = nullptr;
        if (hipHostGetDevicePointer(reinterpret_cast<void**>(&dp), const_cast<uint8_t*>(digests32), 0) == hipSuccess && dp) dig_src = dp;
        else (void)hipGetLastError();
    }
    if (!rc && !dig_src) {
        IPCFP_HIP(ctx, dig.alloc(size_t(N) * 32));
        rc = upload(ctx, dig.p, digests32, size_t(N) * 32, ctx->stream);
        dig_src = dig.p;
    }
    if (!rc && n_esc) {
        IPCFP_HIP(ctx, esc_i.alloc(n_esc));
        IPCFP_HIP(ctx, esc_c.alloc(n_esc * IPCFP_CID_SLOT));
        rc = upload(ctx, esc_i.p, esc_index, n_esc * 4, ctx->stream);
        if (!rc) rc = upload(ctx, esc_c.p, esc_cids40, n_esc * IPCFP_CID_SLOT, ctx->stream);
    }
    if (rc) return rc;
    uint64_t* total_d = scan_scratch.p + div_up(N, 1024) + 1;
    rc = launch_tight_offsets(ctx, glen.p, N, goff.p, total_d, scan_scratch.p);
    if (!rc) rc = launch_expand_cids(ctx, dig_src, N, cid_prefix, prefix_len, esc_i.p, esc_c.p, uint32_t(n_esc), gcids.p);
    if (rc) return rc;
    uint32_t size = 64;
    while (size < 2ull * N) size <<= 1;
    IPCFP_HIP(ctx, slots.alloc(size));
    IPCFP_HIP(ctx, hipMemsetAsync(slots.p, 0xff, size_t(size) * 4, ctx->stream));
    rc = launch_index_insert(ctx, gcids.p, N, slots.p, size - 1);
    if (rc) return rc;
    // ---- the pull's own state ----
    const uint32_t words = div_up(N, 32);
    const uint32_t fcap = N + 1024u;                          // items of one frontier (tree positions of one level)
    const uint64_t stage_cap = nbytes + 128ull * N + 256ull;  // every block of the bundle on lines of its own: the upper bound
    IPCFP_HIP(ctx, resident.alloc(words));
    IPCFP_HIP(ctx, role.alloc(N));
    IPCFP_HIP(ctx, hipMemsetAsync(role.p, 0, size_t(N) * 4, ctx->stream));
    IPCFP_HIP(ctx, stage_off.alloc(N));
    IPCFP_HIP(ctx, pulled.alloc(N));
    IPCFP_HIP(ctx, copy_src.alloc(fcap));
    IPCFP_HIP(ctx, copy_dst.alloc(fcap));
    IPCFP_HIP(ctx, copy_len.alloc(fcap));
    IPCFP_HIP(ctx, fa.alloc(fcap));
    IPCFP_HIP(ctx, fb.alloc(fcap));
    IPCFP_HIP(ctx, ctl.alloc(1));
    IPCFP_HIP(ctx, stage.alloc(stage_cap));
    IPCFP_HIP(ctx, hipMemsetAsync(resident.p, 0, size_t(words) * 4, ctx->stream));
    IPCFP_HIP(ctx, hipMemsetAsync(stage_off.p, 0, size_t(N) * 8, ctx->stream));
    IPCFP_HIP(ctx, hipMemsetAsync(ctl.p, 0, sizeof(PullCtl), ctx->stream));
    // the payload must add up to the buffer (the device trusts goff + len inside [0, nbytes))
    uint64_t total = 0;
    IPCFP_HIP(ctx, d2h_small(ctx, &total, total_d, sizeof total, ctx->stream));
    IPCFP_HIP(ctx, sync_stream(ctx, ctx->stream));
    if (total != nbytes)
        return set_error(ctx, IPCFP_E_INVALID, "the block lengths add up to %llu bytes, the buffer holds %llu", (unsigned long long)total,
                         (unsigned long long)nbytes);
    const auto t_tables = std::chrono::steady_clock::now();
    WitnessView view{};
    view.arena = stage.p;
    view.off = stage_off.p;
    view.len = glen.p;
    view.cids = gcids.p;
    view.slots = slots.p;
    view.mask = size - 1;
    view.n = N;
    view.touched = nullptr;
    PullTables t{};
    t.len = glen.p;
    t.goff = goff.p;
    t.resident = resident.p;
    t.stage_off = stage_off.p;
    t.pulled = pulled.p;
    t.pulled_cap = N;
    t.stage = stage.p;
    t.stage_cap = stage_cap;
    t.copy_src = copy_src.p;
    t.copy_dst = copy_dst.p;
    t.copy_len = copy_len.p;
    PullSeeds seeds;
    std::memset(&seeds, 0, sizeof seeds);
    seeds.child = key_from_slot(child_cid40);
    seeds.n_parents = n_parents;
    for (uint32_t k = 0; k < n_parents && k < IPCFP_MAX_PARENTS; ++k) seeds.parents[k] = key_from_slot(parent_cids40 + size_t(k) * IPCFP_CID_SLOT);
    DevBuf<CidKey> seeds_wide;  // a tipset key wider than the inline form: every key in HBM
    std::vector<CidKey> seeds_wide_h;
    if (n_parents > IPCFP_MAX_PARENTS) {
        seeds_wide_h.resize(n_parents);
        for (uint32_t k = 0; k < n_parents; ++k) seeds_wide_h[k] = key_from_slot(parent_cids40 + size_t(k) * IPCFP_CID_SLOT);
        IPCFP_HIP(ctx, seeds_wide.alloc(n_parents));
        IPCFP_HIP(ctx, hipMemcpyAsync(seeds_wide.p, seeds_wide_h.data(), size_t(n_parents) * sizeof(CidKey), hipMemcpyHostToDevice, ctx->stream));
        seeds.parents_wide = seeds_wide.p;
    }
    // rounds are queued ahead of the host's reading: whatever way this function is left from here on, the stream is drained
    // before the buffers above go back to the pool (the ordinary path has synchronised by then and pays nothing)
    StreamDrainGuard drain(ctx->stream);
    drain.armed = true;
    rc = launch_pull_seed(ctx, view, seeds, PullFrontier{fa.p, fcap, role.p}, ctl.p);
    if (rc) return rc;
    // ---- rounds.  Every kernel takes the frontier's true size from the device, so rounds are queued AHEAD of the host's
    // knowledge (two deep): the queue never runs dry between two rounds — a submission to an idle queue is picked up
    // ≈ 60 µs later, nine rounds of a 1M-receipt tipset were 0.6 ms of that.  The host reads each round's count from the
    // mailbox only to size the next grids and to see the end. ----
    static const bool trace = [] { const char* e = std::getenv("IPCFP_TRACE_PULL"); return e && e[0] == '1'; }();
    auto t_round = std::chrono::steady_clock::now();
    uint32_t queued = 0;  // rounds queued so far: round 0 publishes the seeds' count, round k >= 1 reads frontier (k odd ? a : b)
    auto queue_round = [&](uint32_t hint) -> int {
        const unsigned long long seq = ++ctx->mailbox_seq;
        const uint32_t k = queued++;
        const PullFrontier cur{(k & 1u) ? fa.p : fb.p, fcap, role.p}, next{(k & 1u) ? fb.p : fa.p, fcap, role.p};
        return launch_pull_round(ctx, view, bytes_dev, t, cur, k ? hint : 0u, next, ctl.p, n_shards, shard, ctx->mailbox_dev, seq);
    };
    auto wait_round = [&](unsigned long long seq, uint32_t& n_next, uint32_t& overflow) -> int {
        const unsigned long long* slot = ctx->mailbox + 8u * (seq & 3ull);
        const auto deadline = std::chrono::steady_clock::now() + std::chrono::seconds(20);
        for (uint32_t spins = 0; __atomic_load_n(slot, __ATOMIC_ACQUIRE) != seq; ++spins) {
            if ((spins & 1023u) == 1023u) {
                if (hipStreamQuery(ctx->stream) != hipErrorNotReady && __atomic_load_n(slot, __ATOMIC_ACQUIRE) != seq)
                    return set_error(ctx, IPCFP_E_HIP, "a pull round never reached the mailbox");
                if (std::chrono::steady_clock::now() > deadline) return set_error(ctx, IPCFP_E_HIP, "a pull round timed out");
            }
        }
        n_next = uint32_t(__atomic_load_n(slot + 1, __ATOMIC_RELAXED));
        overflow = uint32_t(__atomic_load_n(slot + 2, __ATOMIC_RELAXED));
        return IPCFP_OK;
    };
    const unsigned long long first_seq = ctx->mailbox_seq + 1;
    rc = queue_round(0);
    if (rc) return rc;
    uint32_t n_items = 0, overflow = 0, rounds = 0;
    rc = wait_round(first_seq, n_items, overflow);
    if (rc) return rc;
    constexpr uint32_t kMaxRounds = 160;  // headers, TxMeta, roots + the tallest AMT anything here loads (64 / bit width levels)
    constexpr uint32_t kAhead = 2;
    auto hint_of = [&](uint32_t n) { return uint32_t(std::min<uint64_t>(std::max<uint64_t>(8ull * n, 4096ull), fcap)); };
    uint32_t read = 0;  // rounds whose count the host has seen
    while (n_items) {
        while (queued < read + 1u + kAhead) {
            rc = queue_round(hint_of(n_items));
            if (rc) return rc;
        }
        ++read;
        if (++rounds > kMaxRounds) {
            (void)hipStreamSynchronize(ctx->stream);
            return set_error(ctx, IPCFP_E_UNSUPPORTED, "the shard's blocks are more than %u links deep", kMaxRounds);
        }
        const uint32_t n_was = n_items;
        rc = wait_round(first_seq + read, n_items, overflow);
        if (rc) return rc;
        if (trace) {
            const auto now = std::chrono::steady_clock::now();
            std::fprintf(stderr, "[pull] round %2u: %8u items  %8.1f us\n", rounds, n_was, std::chrono::duration<double, std::micro>(now - t_round).count());
            t_round = now;
        }
        if (overflow) {
            (void)hipStreamSynchronize(ctx->stream);
            return set_error(ctx, IPCFP_E_UNSUPPORTED, "the shard's walk outgrew its buffers (%s): not a tree of this bundle",
                             (overflow & 1u) ? "frontier" : "staging arena");
        }
    }
    PullCtl h{};
    IPCFP_HIP(ctx, d2h_small(ctx, &h, ctl.p, sizeof h, ctx->stream));
    IPCFP_HIP(ctx, sync_stream(ctx, ctx->stream));
    drain.armed = false;  // every queued round has run
    const auto t_pulled = std::chrono::steady_clock::now();
    if (stats) {
        stats->rounds = rounds;
        stats->blocks = h.n_pulled;
        stats->table_bytes = uint64_t(N) * 36 + n_esc * 44;
        stats->block_bytes = h.stage_used;
        stats->payload_bytes = h.payload;
        stats->tables_ms = std::chrono::duration<double, std::milli>(t_tables - t_start).count();
        stats->pull_ms = std::chrono::duration<double, std::milli>(t_pulled - t_tables).count();
    }
    if (!h.have_range) {  // no receipts root: there is no receipt range to cut by (the caller verifies on the whole bundle)
        *status_out = IPCFP_ST_ERR_MISSING_BLOCK;
        return IPCFP_OK;
    }
    // ---- the shard as a witness of its own: its blocks' tables, then the ordinary constructor over the staging arena ----
    const uint32_t np = h.n_pulled;
    std::unique_ptr<ipcfp_witness> w(new (std::nothrow) ipcfp_witness());
    if (!w) return IPCFP_E_NOMEM;
    w->ctx = ctx;
    w->n = np;
    w->nbytes = h.stage_used;  // (an upper bound: the lengths are not summed on the host)
    w->receipt_lo = h.lo;
    w->receipt_hi = shard + 1u == n_shards ? ~0ULL : h.hi;  // (the last shard enumerates what it holds: PullCtl::hi_walk)
    DevBuf<uint32_t> plen, bad;
    DevBuf<uint64_t> poff;
    DevBuf<uint8_t> pcids;
    IPCFP_HIP(ctx, plen.alloc(np));
    IPCFP_HIP(ctx, poff.alloc(np));
    IPCFP_HIP(ctx, pcids.alloc(size_t(np) * IPCFP_CID_SLOT));
    IPCFP_HIP(ctx, bad.alloc(1));
    IPCFP_HIP(ctx, hipMemsetAsync(bad.p, 0, 4, ctx->stream));
    rc = launch_subset_tables(ctx, pulled.p, np, N, stage_off.p, glen.p, gcids.p, poff.p, plen.p, pcids.p, bad.p);
    if (rc) return rc;
    rc = witness_finish_create(ctx, w.get(), stage.p, poff.p, plen.p, pcids.p);
    if (rc) return rc;
    if (stats) stats->create_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_pulled).count();
    *receipt_lo = h.lo;
    *receipt_hi = h.hi;
    if (n_receipts) *n_receipts = h.n_receipts;
    *status_out = IPCFP_ST_TRUE;
    *out = w.release();
    return IPCFP_OK;
}

}  // extern "C"
