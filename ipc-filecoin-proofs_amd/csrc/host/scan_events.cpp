// csrc/host/scan_events.cpp — `find_matching_events` over the HBM-resident tipset
// (src/proofs/events/generator.rs:180-307): enumerate the receipts AMT, PASS 1, prefix-sum, PASS 2.
#include <algorithm>
#include <chrono>
#include <cstring>
#include <memory>
#include <vector>

#include "../common.h"
#include "../kernels/amt_enum.h"
#include "../kernels/launch.h"
#include "exec_state.h"

using namespace ipcfp;

namespace ipcfp {
CidKey key_from_slot(const uint8_t* slot40);
}

namespace ipcfp {

// The host's side of the mailbox (common.h): spin until the device has published sequence number `seq`.  A launch that
// failed never publishes: after 5 s the stream is drained, which surfaces the error.
// The stream's last kernel has told the host (mailbox) that it is done but may not have retired yet: poll until the queue
// is empty — normally the first or second query — and fall back to the ordinary wait after 50 µs.
hipError_t settle_stream(ipcfp_ctx* ctx, hipStream_t s) {
    const auto deadline = std::chrono::steady_clock::now() + std::chrono::microseconds(50);
    for (uint32_t spins = 0;; ++spins) {
        const hipError_t e = hipStreamQuery(s);
        if (e != hipErrorNotReady) return e;
        if ((spins & 7u) == 7u && std::chrono::steady_clock::now() > deadline) return wait_stream(ctx, s);
    }
}

int mailbox_wait(ipcfp_ctx* ctx, unsigned long long seq, const char* what) {
    const auto deadline = std::chrono::steady_clock::now() + std::chrono::seconds(5);
    uint32_t spins = 0;
    while (__atomic_load_n(ctx->mailbox, __ATOMIC_ACQUIRE) != seq) {
        if ((++spins & 1023u) == 0 && std::chrono::steady_clock::now() > deadline) {
            IPCFP_HIP(ctx, sync_stream(ctx, ctx->stream));
            if (__atomic_load_n(ctx->mailbox, __ATOMIC_ACQUIRE) != seq) return set_error(ctx, IPCFP_E_HIP, "%s never reached the mailbox", what);
        }
    }
    return IPCFP_OK;
}

int scan_tail_scratch(ipcfp_ctx* ctx, uint32_t n_tiles, unsigned long long** out) {
    const size_t need = (size_t(n_tiles) + 3) * 8;
    if (need > ctx->scan_scratch_bytes) {
        if (ctx->scan_scratch) (void)hipFree(ctx->scan_scratch);  // (waits for the device: nothing still reads it)
        ctx->scan_scratch = nullptr;
        ctx->scan_scratch_bytes = 0;
        IPCFP_HIP(ctx, hipMalloc(&ctx->scan_scratch, need * 2));
        ctx->scan_scratch_bytes = need * 2;
        IPCFP_HIP(ctx, hipMemsetAsync(ctx->scan_scratch, 0, need * 2, ctx->stream));  // ticket / counters start at zero
    }
    *out = static_cast<unsigned long long*>(ctx->scan_scratch);
    return IPCFP_OK;
}

ScanParams scan_params_of(const ipcfp_event_filter_t& filter, int has_actor, uint64_t actor) {
    ScanParams p{};  // (zeroed, padding included: compared with memcmp)
    p.filter = filter;
    p.actor = has_actor ? actor : 0;
    p.has_actor = has_actor ? 1u : 0u;
    return p;
}

// Queue k_block_events for the witness (aux stream) unless its block table is already there.  `filter` (nullable):
// the scan filter whose matches the pass counts per block.  Called at the START of a scan / verify call, before the
// main stream's own work is queued, so that the parse runs beside the receipts enumeration (and K1).
int block_table_prefetch(ipcfp_ctx* ctx, ipcfp_witness* w, const ipcfp_event_filter_t* filter, int has_actor, uint64_t actor) {
    if (!w->use_event_table || w->n == 0 || w->bt_valid) return IPCFP_OK;
    const uint64_t n = w->n;
    if (!w->bt_blocks.p) {
        // record pool: generous for real tipsets (an event is > 64 encoded bytes), bounded by what the witness could
        // hold; a block that finds the pool exhausted is simply left to the walkers (RK_WALK)
        uint64_t cap = std::max<uint64_t>(4 * n + 1024, w->nbytes / 64);
        cap = std::min<uint64_t>(cap, 0xfffffff0ull);
        IPCFP_HIP(ctx, w->bt_blocks.alloc_unpooled(n));
        IPCFP_HIP(ctx, w->bt_events.alloc_unpooled(cap));
        IPCFP_HIP(ctx, w->bt_used.alloc_unpooled(size_t(kPoolParts) * kPoolCounterStride));
    }
    hipStream_t s = ctx->stream_aux;
    IPCFP_HIP(ctx, hipMemsetAsync(w->bt_used.p, 0, size_t(kPoolParts) * kPoolCounterStride * 4, s));
    w->bt_has_filter = filter != nullptr;
    w->bt_filter = ScanParams{};
    if (filter) w->bt_filter = scan_params_of(*filter, has_actor, actor);
    int rc = launch_block_events(ctx, s, w->arena.p, w->k1_meta.p, uint32_t(n), filter, has_actor, actor, w->bt_blocks.p,
                                 w->bt_events.p, uint32_t(w->bt_events.count), w->bt_used.p);
    if (rc) return rc;
    w->bt_valid = true;
    w->bt_joined = s == ctx->stream;
    if (!w->bt_joined) IPCFP_HIP(ctx, hipEventRecord(ctx->aux_event, s));
    return IPCFP_OK;
}

const uint32_t* event_table_counts(const EventTableCached* t, const ipcfp_event_filter_t& filter, int has_actor, uint64_t actor) {
    if (!t || !t->has_counts) return nullptr;
    const ScanParams want = scan_params_of(filter, has_actor, actor);
    return std::memcmp(&want, &t->counts_filter, sizeof want) == 0 ? t->counts.p : nullptr;
}

int event_table_join(ipcfp_ctx* ctx, ipcfp_witness* w) {
    if (w->bt_valid && !w->bt_joined) {
        IPCFP_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->aux_event, 0));
        w->bt_joined = true;
    }
    return IPCFP_OK;
}

int event_table_get(ipcfp_ctx* ctx, ipcfp_witness* w, const CidKey& root, const EnumCached* en, const EventTableCached** out,
                    bool on_aux) {
    for (auto& t : w->table_cache)
        if (t->lo == w->receipt_lo && t->hi == w->receipt_hi && std::memcmp(t->root, root.w, 40) == 0) {
            *out = t.get();
            return on_aux ? IPCFP_OK : event_table_join(ctx, w);  // (its kernels may still be on the aux stream)
        }
    int rc = block_table_prefetch(ctx, w, nullptr, 0, 0);  // (queued long ago by the callers that care)
    if (rc) return rc;
    on_aux = on_aux && ctx->stream_aux != ctx->stream;
    hipStream_t ks = on_aux ? ctx->stream_aux : ctx->stream;  // behind the block-order parse, or joined with it
    if (!on_aux) {
        rc = event_table_join(ctx, w);
        if (rc) return rc;
    }
    std::unique_ptr<EventTableCached> t(new EventTableCached());
    std::memcpy(t->root, root.w, 40);
    t->lo = w->receipt_lo;
    t->hi = w->receipt_hi;
    t->n = en->n;
    t->events = w->bt_events.p;
    const uint64_t n = en->n;
    IPCFP_HIP(ctx, t->receipts.alloc(n));
    IPCFP_HIP(ctx, t->err_word.alloc(1));
    IPCFP_HIP(ctx, hipMemsetAsync(t->err_word.p, 0xff, 8, ks));  // kNoEnumError
    // the per-block match counts belong to the filter the block pass ran with: the receipts inherit them
    t->has_counts = w->bt_has_filter;
    if (t->has_counts) {
        t->counts_filter = w->bt_filter;
        IPCFP_HIP(ctx, t->counts.alloc(n));
    }
    const WitnessView view = witness_view(w);
    rc = launch_receipt_events(ctx, view, reinterpret_cast<const LeafRef*>(en->leaves.p), uint32_t(n),
                               t->has_counts ? &w->bt_filter.filter : nullptr, int(w->bt_filter.has_actor), w->bt_filter.actor,
                               w->bt_blocks.p, t->receipts.p, t->has_counts ? t->counts.p : nullptr, t->err_word.p, ks);
    if (rc) return rc;
    if (on_aux) {  // the next reader on the main stream joins (event_table_join)
        IPCFP_HIP(ctx, hipEventRecord(ctx->aux_event, ks));
        w->bt_joined = false;
    }
    *out = t.get();
    w->table_cache.push_back(std::move(t));
    return IPCFP_OK;
}

// PASS 1 + prefix sum + PASS 2 on the device.  `touched_d` (nullable, device, words = ceil(n/32)) is
// OR-ed into, so a caller can accumulate one recorder across several steps (generate_event_proof).
int scan_events_device(ipcfp_ctx* ctx, ipcfp_witness* w, const CidKey& root, const ipcfp_event_filter_t& filter,
                       int has_actor, uint64_t actor, uint32_t* touched_d, ScanResult& out, uint64_t cap_matches) {
    const WitnessView view = witness_view(w);
    // the block-order event parse starts now, beside everything below up to the table lookup
    int rc0 = block_table_prefetch(ctx, w, &filter, has_actor, actor);
    if (rc0) return rc0;
    ctx->scan_hint = scan_params_of(filter, has_actor, actor);
    ctx->has_scan_hint = true;
    DevBuf<unsigned long long> err_own;
    unsigned long long* err_p = nullptr;  // kNoEnumError
    IPCFP_HIP(ctx, ctl_words(ctx, err_own, err_p, 1, true));
    struct { unsigned long long* p; } err{err_p};
    const EnumCached* en = nullptr;
    // receipts in index order; a shard witness (host/shard.cpp) enumerates its own index range only
    const uint64_t lo = w->receipt_lo, hi = w->receipt_hi;
    int rc = amt_enumerate_cached(ctx, w, root, 0, VK_RECEIPT, &en, lo, hi);
    if (rc) return rc;
    if (en->error != kNoEnumError) {
        out.status = enum_error_code(en->error);
        out.phase = IPCFP_SCAN_PHASE_RECEIPTS;
        return IPCFP_OK;
    }
    const LeafRef* leaves = reinterpret_cast<const LeafRef*>(en->leaves.p);
    const uint32_t n = uint32_t(en->n);
    // receipt indices are ascending: the last leaf gives the size of the per-index byte map
    uint64_t n_idx = 0;
    if (n && en->dense) {
        n_idx = n;  // leaf i has index lo + i
    } else if (n) {
        LeafRef last;
        IPCFP_HIP(ctx, d2h_small(ctx, &last, leaves + (n - 1), sizeof last, ctx->stream));
        IPCFP_HIP(ctx, sync_stream(ctx, ctx->stream));
        n_idx = last.index + 1 - lo;  // the per-index byte map starts at the shard's first index
    }
    DevBuf<uint32_t> counts, offsets;
    DevBuf<uint64_t> scratch, total_own;
    uint64_t* total_p = nullptr;
    IPCFP_HIP(ctx, counts.alloc(n));
    IPCFP_HIP(ctx, offsets.alloc(n));
    IPCFP_HIP(ctx, scratch.alloc(size_t(div_up(n, 1024)) + 2));
    IPCFP_HIP(ctx, ctl_words(ctx, total_own, total_p, 1, false));
    struct { uint64_t* p; } total{total_p};
    if (out.ext_has && out.ext_has_cap >= n_idx) {
        out.has_p = out.ext_has;
    } else {
        IPCFP_HIP(ctx, out.has.alloc(n_idx));
        out.has_p = out.has.p;
    }
    // (a dense enumeration has a leaf for every index of the map and PASS 2 writes the byte of every leaf: nothing to clear)
    if (n_idx && !en->dense) IPCFP_HIP(ctx, hipMemsetAsync(out.has_p, 0, n_idx, ctx->stream));
    // PASS 1: with the events tabulated once per witness (kernels/event_table.h) — the first scan builds the table
    // and counts in one kernel, a later one (another filter) counts from the records
    EventTableView tview{nullptr, nullptr};
    const uint32_t* cnt = counts.p;
    unsigned long long e_table = kNoEnumError;
    const unsigned long long* table_err_d = nullptr;  // the table's own error word, when its counts are this scan's PASS 1
    if (w->use_event_table && n) {
        const EventTableCached* table = nullptr;
        rc = event_table_get(ctx, w, root, en, &table);
        if (rc) return rc;
        tview = table->view();
        if (const uint32_t* c = event_table_counts(table, filter, has_actor, actor)) {
            cnt = c;  // counted while the table was built: PASS 1 is done
            table_err_d = table->err_word.p;  // ... and so is its error report
        } else {
            rc = launch_count_from_table(ctx, view, leaves, n, filter, has_actor, actor, tview, counts.p, err.p);
            if (rc) return rc;
        }
    } else {
        rc = launch_scan_pass1(ctx, view, leaves, n, filter, has_actor, actor, counts.p, err.p);
        if (rc) return rc;
    }
    // ---- the tail in ONE launch, results through the mailbox: no read-back copy, no stream synchronisation ----
    if (ctx->scan_fused != 0 && ctx->mailbox && tview.receipts && !touched_d && n && cap_matches <= (1ull << 26)) {
        unsigned long long* tail_scratch = nullptr;
        rc = scan_tail_scratch(ctx, div_up(n, 1024), &tail_scratch);
        if (rc) return rc;
        if (out.ext_matches && cap_matches) {
            out.matches_p = out.ext_matches;
        } else {
            IPCFP_HIP(ctx, out.matches.alloc(cap_matches));
            out.matches_p = out.matches.p;
        }
        const unsigned long long seq = ++ctx->mailbox_seq;
        rc = launch_scan_tail_fused(ctx, view, leaves, n, en->dense ? lo : ~0ull, filter, has_actor, actor, tview, cnt, offsets.p,
                                    cap_matches ? out.matches_p : nullptr, cap_matches, out.has_p, n_idx, lo,
                                    tail_scratch, ++ctx->scan_epoch, table_err_d, err.p,
                                    ctx->mailbox_dev, seq);
        if (rc) return rc;
        rc = mailbox_wait(ctx, seq, "the scan's results");
        if (rc) return rc;
        const uint64_t nm = __atomic_load_n(ctx->mailbox + 1, __ATOMIC_RELAXED);
        unsigned long long e1 = __atomic_load_n(ctx->mailbox + 3, __ATOMIC_RELAXED);
        const unsigned long long et = __atomic_load_n(ctx->mailbox + 2, __ATOMIC_RELAXED);
        const uint64_t walk = __atomic_load_n(ctx->mailbox + 4, __ATOMIC_RELAXED);
        if (et < e1) e1 = et;
        if (e1 != kNoEnumError) {
            out.status = enum_error_code(e1);
            out.phase = IPCFP_SCAN_PHASE_EVENTS;
            return IPCFP_OK;
        }
        if (walk) {  // matching receipts the table does not cover: the general walk writes their matches (offsets are in place)
            rc = launch_scan_pass2(ctx, view, root, leaves, n, filter, has_actor, actor, cnt, offsets.p,
                                   cap_matches ? out.matches_p : nullptr, cap_matches, out.has_p, n_idx, lo, &tview);
            if (rc) return rc;
            IPCFP_HIP(ctx, sync_stream(ctx, ctx->stream));
        }
        out.n_idx = n_idx;
        out.n_matches = nm;
        out.status = IPCFP_ST_TRUE;
        return IPCFP_OK;
    }
    if (table_err_d) IPCFP_HIP(ctx, d2h_small(ctx, &e_table, table_err_d, 8, ctx->stream));
    rc = launch_scan_u32(ctx, cnt, n, offsets.p, total.p, scratch.p);
    if (rc) return rc;
    uint64_t nm = 0;
    unsigned long long e1 = kNoEnumError;
    IPCFP_HIP(ctx, ctl_read(ctx, &nm, total.p, 8));
    IPCFP_HIP(ctx, ctl_read(ctx, &e1, err.p, 8));
    uint64_t cap = cap_matches;
    if (cap_matches > (1ull << 26)) {  // unknown (kAllMatches) or too big to reserve blindly:
        // size the match list to the count — one more synchronisation
        IPCFP_HIP(ctx, sync_stream(ctx, ctx->stream));
        if (e_table < e1) e1 = e_table;
        if (e1 != kNoEnumError) {
            out.status = enum_error_code(e1);
            out.phase = IPCFP_SCAN_PHASE_EVENTS;
            return IPCFP_OK;
        }
        cap = nm;
    }
    if (out.ext_matches && cap == cap_matches) {
        out.matches_p = out.ext_matches;
    } else {
        IPCFP_HIP(ctx, out.matches.alloc(cap));
        out.matches_p = out.matches.p;
    }
    WitnessView rec = view;
    rec.touched = touched_d;
    rc = launch_scan_pass2(ctx, rec, root, leaves, n, filter, has_actor, actor, cnt, offsets.p,
                           cap ? out.matches_p : nullptr, cap, out.has_p, n_idx, lo, tview.receipts ? &tview : nullptr);
    if (rc) return rc;
    IPCFP_HIP(ctx, sync_stream(ctx, ctx->stream));  // delivers nm / e1 when they were not waited for above
    if (e_table < e1) e1 = e_table;
    if (e1 != kNoEnumError) {  // PASS 2 ran on a tipset PASS 1 rejected: its output is discarded
        out.status = enum_error_code(e1);
        out.phase = IPCFP_SCAN_PHASE_EVENTS;
        return IPCFP_OK;
    }
    out.n_idx = n_idx;
    out.n_matches = nm;
    out.status = IPCFP_ST_TRUE;
    return IPCFP_OK;
}

}  // namespace ipcfp

extern "C" {

int ipcfp_scan_events(ipcfp_ctx_t* ctx, ipcfp_witness_t* w, const uint8_t* receipts_root40,
                      const ipcfp_event_filter_t* filter, int has_actor, uint64_t actor, ipcfp_status_t* status_out,
                      uint8_t* receipt_has_match, uint64_t cap_receipts, uint64_t* n_receipts,
                      ipcfp_event_match_t* matches, uint64_t cap_matches, uint64_t* n_matches, uint32_t* touched_bits) {
    if (!ctx || !w || w->ctx != ctx || !receipts_root40 || !filter || !status_out || !n_receipts || !n_matches)
        return IPCFP_E_INVALID;
    IPCFP_ENTER(ctx);
    *n_receipts = *n_matches = 0;
    *status_out = IPCFP_ST_ERR;
    const uint32_t words = div_up(w->n, 32);
    DevBuf<uint32_t> touched;
    if (touched_bits) {
        IPCFP_HIP(ctx, touched.alloc(words));
        IPCFP_HIP(ctx, hipMemsetAsync(touched.p, 0, size_t(words) * 4, ctx->stream));
    }
    ScanResult res;
    int rc = scan_events_device(ctx, w, key_from_slot(receipts_root40), *filter, has_actor, actor,
                                touched_bits ? touched.p : nullptr, res, matches ? cap_matches : 0);
    if (rc) return rc;
    *status_out = ipcfp_status_t(res.status);
    w->last_scan_phase = res.status == IPCFP_ST_TRUE ? 0u : res.phase;
    if (res.status != IPCFP_ST_TRUE) return IPCFP_OK;
    *n_receipts = res.n_idx;
    *n_matches = res.n_matches;
    bool queued = false;  // (a sizing call copies nothing back: the scan's own synchronisation was the last one it needs)
    if (receipt_has_match && res.n_idx && cap_receipts) {
        IPCFP_HIP(ctx, hipMemcpyAsync(receipt_has_match, res.has_p, res.n_idx < cap_receipts ? res.n_idx : cap_receipts,
                                      hipMemcpyDeviceToHost, ctx->stream));
        queued = true;
    }
    if (matches && res.n_matches && cap_matches) {
        IPCFP_HIP(ctx, hipMemcpyAsync(matches, res.matches_p,
                                      (res.n_matches < cap_matches ? res.n_matches : cap_matches) * sizeof(ipcfp_event_match_t),
                                      hipMemcpyDeviceToHost, ctx->stream));
        queued = true;
    }
    if (touched_bits) {
        IPCFP_HIP(ctx, hipMemcpyAsync(touched_bits, touched.p, size_t(words) * 4, hipMemcpyDeviceToHost, ctx->stream));
        queued = true;
    }
    if (queued) IPCFP_HIP(ctx, sync_stream(ctx, ctx->stream));
    ctl_preprime(ctx);
    return IPCFP_OK;
}

// The same scan with DEVICE outputs (a multi-GPU host all-gathers them without a round trip through host memory):
// receipt_has_match_d (cap_receipts bytes) and matches_d (cap_matches records) are HBM buffers of the caller, or
// null; the three scalars still come back to the host.
int ipcfp_scan_events_device(ipcfp_ctx_t* ctx, ipcfp_witness_t* w, const uint8_t* receipts_root40,
                             const ipcfp_event_filter_t* filter, int has_actor, uint64_t actor, ipcfp_status_t* status_out,
                             void* receipt_has_match_d, uint64_t cap_receipts, uint64_t* n_receipts, void* matches_d,
                             uint64_t cap_matches, uint64_t* n_matches, void* summary_d) {
    if (!ctx || !w || w->ctx != ctx || !receipts_root40 || !filter || !status_out || !n_receipts || !n_matches)
        return IPCFP_E_INVALID;
    IPCFP_ENTER(ctx);
    *n_receipts = *n_matches = 0;
    *status_out = IPCFP_ST_ERR;
    ScanResult res;
    res.ext_has = static_cast<uint8_t*>(receipt_has_match_d);
    res.ext_has_cap = receipt_has_match_d ? cap_receipts : 0;
    res.ext_matches = cap_matches ? static_cast<ipcfp_event_match_t*>(matches_d) : nullptr;
    int rc = scan_events_device(ctx, w, key_from_slot(receipts_root40), *filter, has_actor, actor, nullptr, res,
                                matches_d ? cap_matches : 0);
    if (rc) return rc;
    *status_out = ipcfp_status_t(res.status);
    w->last_scan_phase = res.status == IPCFP_ST_TRUE ? 0u : res.phase;
    if (summary_d) {  // {status | phase << 8, n_matches} for the step message of a multi-GPU host
        const uint64_t sm[2] = {uint64_t(res.status) | (uint64_t(w->last_scan_phase) << 8), res.status == IPCFP_ST_TRUE ? res.n_matches : 0};
        IPCFP_HIP(ctx, h2d_small(ctx, summary_d, sm, sizeof sm, ctx->stream));
    }
    if (res.status != IPCFP_ST_TRUE) return IPCFP_OK;
    *n_receipts = res.n_idx;
    *n_matches = res.n_matches;
    // PASS 2 wrote straight into the caller's buffers when they were large enough; otherwise a (truncating) copy
    bool copied = false;
    if (receipt_has_match_d && res.n_idx && res.has_p != receipt_has_match_d) {
        IPCFP_HIP(ctx, hipMemcpyAsync(receipt_has_match_d, res.has_p, res.n_idx < cap_receipts ? res.n_idx : cap_receipts,
                                      hipMemcpyDeviceToDevice, ctx->stream));
        copied = true;
    }
    if (matches_d && res.n_matches && res.matches_p != matches_d) {
        IPCFP_HIP(ctx, hipMemcpyAsync(matches_d, res.matches_p,
                                      (res.n_matches < cap_matches ? res.n_matches : cap_matches) * sizeof(ipcfp_event_match_t),
                                      hipMemcpyDeviceToDevice, ctx->stream));
        copied = true;
    }
    if (copied || summary_d) {
        IPCFP_HIP(ctx, sync_stream(ctx, ctx->stream));  // `res` returns its buffers to the pool on exit
    } else {
        // PASS 2 wrote straight into the caller's buffers and the counts came through the mailbox, which the tail kernel
        // publishes from its last workgroup WITHOUT a release fence (event_scan.hip k_scan_tail_fused): its writes are
        // visible to other streams and to blocking copies once the kernel has ENDED — a few hundred nanoseconds after the
        // host saw the mailbox.  The call is documented as synchronous (ipcfp.h), so it waits for that end: a query loop,
        // not an event (an event on a queue that has just gone idle is picked up ≈ 60 µs later).  ADVICE r4.
        IPCFP_HIP(ctx, settle_stream(ctx, ctx->stream));
    }
    ctl_preprime(ctx);
    return IPCFP_OK;
}

int ipcfp_witness_last_scan_phase(const ipcfp_witness_t* w) { return w ? int(w->last_scan_phase) : 0; }

}  // extern "C"
