// csrc/host/verify_fast.cpp — `verify_event_proof` for a batch of ONE tipset pair without a mid-call synchronisation.
//
// host/verify_events.cpp::verify_packed waits for the device twice before the verify kernel runs — for the AMT roots'
// shapes (to size the walk) and for the walk's anomaly flag — and each wait drains the stream, wakes the host and
// leaves the GPU idle until the next launch arrives (≈40 + ≈80 µs of a 1.3 ms step).  Here
//   * the root shapes reach the host through the MAILBOX — a page of coherent pinned host memory that k_enum_roots
//     writes while the stream keeps going — so the host sizes and queues everything else without draining anything;
//   * the anomaly flag is read at the END of the call, with the status bytes: every kernel behind the walk is queued
//     before the host knows whether the dense walk held.  They are safe on whatever it left (k_dense_leaves writes every
//     LeafRef it owes, kNoBlock when there is no value; readers skip those), and when the flag is set — a sparse AMT, a
//     lying count, a missing block, a decode error — the call's results are thrown away and verify_packed does the
//     whole batch again its own way (the general level-synchronous walk orders errors as the reference does);
//   * what the host did between the second wait and the verify kernel (is the execution order reachable, where are the
//     tables, exec_len, the inverse permutation) happens on the device (k_ctx_finish).
// The call synchronises once.  Same kernels otherwise, same results: tests/test_gpu_events.py, test_gpu_baseline_sizes.py,
// test_gpu_enum_shapes.py (every shape that leaves the dense path) run through this file first.
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <vector>

#include "../common.h"
#include "../kernels/claims_dev.h"
#include "../kernels/amt_enum.h"
#include "../kernels/event_table.h"
#include "../kernels/tipset_ctx.h"
#include "../kernels/types_dev.h"
#include "../kernels/launch.h"
#include "exec_state.h"

using namespace ipcfp;

namespace ipcfp {

int exec_state_prepare(ipcfp_ctx* ctx, ExecState& ex, uint32_t n_parents);

// → IPCFP_OK with *done = true: status_d (and where_d) hold the batch's results;
//   IPCFP_OK with *done = false: not this route's case, or the dense walk did not hold — nothing the caller may use;
//   anything else: an ABI error.
static int verify_packed_fast_queue(ipcfp_ctx* ctx, ipcfp_witness* w, std::vector<TipsetCtxDev>& tcs, const EventClaimPacked* claims_d,
                                    uint32_t n, const uint8_t* blob_d, uint64_t blob_len, const ipcfp_trust_policy_t* trust,
                                    const ipcfp_event_filter_t* filter, uint8_t* status_d, void* where_d, bool* done, ScanRide* ride);

int verify_packed_fast(ipcfp_ctx* ctx, ipcfp_witness* w, std::vector<TipsetCtxDev>& tcs, const EventClaimPacked* claims_d,
                       uint32_t n, const uint8_t* blob_d, uint64_t blob_len, const ipcfp_trust_policy_t* trust,
                       const ipcfp_event_filter_t* filter, uint8_t* status_d, void* where_d, bool* done, ScanRide* ride) {
    const int rc = verify_packed_fast_queue(ctx, w, tcs, claims_d, n, blob_d, blob_len, trust, filter, status_d, where_d, done, ride);
    if (rc != IPCFP_OK) {
        // An error return may have left kernels on the main and the aux stream that still use this call's pooled scratch
        // (the leaves, the receipts' event records, the deferred re-hash): the buffers went back to the pool when the
        // function returned, and nothing may be handed out again before those kernels are through.
        (void)upload_task_wait(ctx);
        (void)hipStreamSynchronize(ctx->stream_aux);
        (void)hipStreamSynchronize(ctx->stream);
        if (w->bt_valid) w->bt_joined = true;
    }
    return rc;
}

static int verify_packed_fast_queue(ipcfp_ctx* ctx, ipcfp_witness* w, std::vector<TipsetCtxDev>& tcs, const EventClaimPacked* claims_d,
                                    uint32_t n, const uint8_t* blob_d, uint64_t blob_len, const ipcfp_trust_policy_t* trust,
                                    const ipcfp_event_filter_t* filter, uint8_t* status_d, void* where_d, bool* done, ScanRide* ride) {
    static const ipcfp_trust_policy_t accept_all = {0, 0, 0, 0};
    const bool enabled = ctx->fast_verify != 0;  // (env IPCFP_FAST_VERIFY / ipcfp_ctx_set_tuning "fast_verify")
    *done = false;
    if (!enabled || !ctx->mailbox || tcs.size() != 1 || !w->use_event_table || ctx->stream_aux == ctx->stream) return IPCFP_OK;
    const TipsetCtxDev& in = tcs[0];
    if (!(in.flags & TC_PARENTS_PARSED) || !(in.flags & TC_CHILD_PARSED) || in.n_parents == 0 || in.n_parents > kMaxParents) return IPCFP_OK;
    for (auto& e : w->enum_cache)  // a scan enumerated the receipts already: verify_packed shares that enumeration
        if (e->vkind == VK_RECEIPT && e->lo == w->receipt_lo && e->hi == w->receipt_hi) return IPCFP_OK;
    const WitnessView view = witness_view(w);
    const uint32_t P = in.n_parents, n_roots = 2 * P, n_all = n_roots + 1;
    // the block-order event parse runs beside everything below (aux stream), counting the last scan's filter
    int rc = ctx->has_scan_hint ? block_table_prefetch(ctx, w, &ctx->scan_hint.filter, int(ctx->scan_hint.has_actor), ctx->scan_hint.actor)
                                : block_table_prefetch(ctx, w, nullptr, 0, 0);
    if (rc) return rc;
    // Where the head of the call is queued.  `ctx->stream` IS that stream until the hand-back, so every helper that queues
    // "on the call's stream" (control words, small copies, the launchers) follows without knowing.
    //   * head stream (IPCFP_HEAD_STREAM=1, measured and off): the tipset prologue runs BESIDE the CID index's inserts — it waits for the point where
    //     the table was cleared, not for the inserts behind it, and its lookups wait for their keys (tipset_prepare.hip
    //     LiveIndex); the main stream takes over with k_enum_roots.  Not when a block may exceed the prologue's LDS stage
    //     (the general companion wants the finished index).
    //   * narrow stream (IPCFP_RESERVE_CUS, measured and off): prologue, roots and the narrow interior levels on reserved
    //     CUs, behind the inserts; launch_dense_walk hands back.
    struct StreamSwap {
        ipcfp_ctx* c;
        hipStream_t saved;
        ~StreamSwap() {
            if (c->stream != saved) (void)hipStreamSynchronize(c->stream);  // (left early: nothing handed the work over)
            c->stream = saved;
        }
    } swap{ctx, ctx->stream};
    const bool need_general = uint64_t(w->max_block_len) + 32u > uint64_t(kPrologueStageChunks) * 16u;
    const bool head = ctx->stream_head && w->index_event && w->index_done.p && !need_general;
    if (head) {
        IPCFP_HIP(ctx, hipStreamWaitEvent(ctx->stream_head, w->index_event, 0));  // the witness is in place, the table cleared
        IPCFP_HIP(ctx, hipStreamWaitEvent(ctx->stream_head, ctx->ctl_event, 0));  // the control block re-initialised
        ctx->stream = ctx->stream_head;
    } else if (ctx->stream_narrow) {
        IPCFP_HIP(ctx, hipEventRecord(ctx->narrow_event, ctx->stream));  // behind the index build (and the preprimed control block)
        IPCFP_HIP(ctx, hipStreamWaitEvent(ctx->stream_narrow, ctx->narrow_event, 0));
        ctx->stream = ctx->stream_narrow;
    }
    // The context on the device: a slice of the control block's ZERO half when there is room (re-initialised at the end of
    // the previous call) — its inputs then travel as a kernel argument of the prologue, which writes them in; no copy
    // kernel at the head of the call (17.7 µs beside the side streams' grids: profiles/r03_last_commit_timeline.txt).
    struct TcsDev {
        DevBuf<TipsetCtxDev> own;
        TipsetCtxDev* p = nullptr;
    } tcs_d;
    TipsetInputs inputs;
    std::memcpy(&inputs, &tcs[0], sizeof inputs);
    const bool head_or_general = (ctx->stream_head && w->index_event && w->index_done.p) ||
                                 uint64_t(w->max_block_len) + 32u > uint64_t(kPrologueStageChunks) * 16u;
    tcs_d.p = head_or_general ? nullptr : static_cast<TipsetCtxDev*>(ctl_take(ctx, sizeof(TipsetCtxDev), false));
    const bool inline_inputs = tcs_d.p != nullptr;
    if (!inline_inputs) {
        IPCFP_HIP(ctx, tcs_d.own.alloc(1));
        tcs_d.p = tcs_d.own.p;
        TipsetCtxDev init = tcs[0];
        for (uint32_t b = 0; b < IPCFP_MAX_PARENTS; ++b) init.txmeta_block[b] = 0;  // (nothing left to re-hash yet)
        IPCFP_HIP(ctx, h2d_small(ctx, tcs_d.p, &init, sizeof(TipsetCtxDev), ctx->stream));
    }
    ExecState ex;
    rc = exec_state_prepare(ctx, ex, P);
    if (rc) return rc;
    PrepareJob job{tcs_d.p, ex.roots.p, ex.err.p};
    DevBuf<EnumNode> frontier;
    DevBuf<DenseNode> dense_frontier;
    DevBuf<uint32_t> small_own;
    DevBuf<uint64_t> info_own;
    uint32_t* small = nullptr;  // [0] = max height (unused here), [2] = anomaly flag (the dense walk, the live lookups)
    uint64_t* info_d = nullptr;
    IPCFP_HIP(ctx, frontier.alloc(n_all));
    IPCFP_HIP(ctx, dense_frontier.alloc(n_all));
    IPCFP_HIP(ctx, ctl_words(ctx, small_own, small, 4, false));
    IPCFP_HIP(ctx, ctl_words(ctx, info_own, info_d, 2 * size_t(n_all), false));
    std::unique_ptr<ProfileScope> prof(new ProfileScope(ctx, IPCFP_K_TIPSET_PROLOGUE));
    // (the TxMeta re-hashes — ≈ 3 k one-lane instructions per parent that gate nothing, they only ever add an error — are
    // left to a launch of their own on the aux stream, joined at the end of the call; not when a block may need the
    // general companion, which hashes inline)
    const bool defer_rehash = !need_general && ctx->rehash_event != nullptr;
    rc = head ? launch_tipset_prepare(ctx, view, &job, nullptr, 1, false, w->index_done.p, w->index_wgs, small + 2, defer_rehash)
              : launch_tipset_prepare(ctx, view, &job, nullptr, 1, need_general, nullptr, 0, nullptr, defer_rehash,
                                      inline_inputs ? &inputs : nullptr);
    if (rc) return rc;
    if (head) {  // hand back: the main stream (behind the inserts by its own order) waits for the prologue
        IPCFP_HIP(ctx, hipEventRecord(ctx->head_event, ctx->stream));
        IPCFP_HIP(ctx, hipStreamWaitEvent(swap.saved, ctx->head_event, 0));
        ctx->stream = swap.saved;
        for (auto& r : ctx->pending)
            if (r.stream == ctx->stream_head) r.stream = ctx->stream;
        if (prof) prof->stream = ctx->stream;
    }
    // ---- the roots; their shapes come back through the mailbox ----
    const unsigned long long seq = ++ctx->mailbox_seq;
    rc = launch_enum_roots(ctx, view, ex.roots.p, n_all, VK_CID, frontier.p, small, ex.err.p, info_d, ctx->mailbox_dev, seq,
                           dense_frontier.p);
    if (rc) return rc;
    prof.reset();
    {
        const auto deadline = std::chrono::steady_clock::now() + std::chrono::seconds(5);
        uint32_t spins = 0;
        while (__atomic_load_n(ctx->mailbox, __ATOMIC_ACQUIRE) != seq) {
            if ((++spins & 1023u) == 0 && std::chrono::steady_clock::now() > deadline) {
                IPCFP_HIP(ctx, sync_stream(ctx, ctx->stream));  // (a failed launch surfaces here)
                if (__atomic_load_n(ctx->mailbox, __ATOMIC_ACQUIRE) != seq)
                    return set_error(ctx, IPCFP_E_HIP, "the AMT roots' shapes never reached the mailbox");
            }
        }
    }
    std::vector<uint64_t> root_info(2 * size_t(n_all));
    for (size_t i = 0; i < root_info.size(); ++i) root_info[i] = __atomic_load_n(ctx->mailbox + 1 + i, __ATOMIC_RELAXED);
    DensePlan plan;
    dense_plan(root_info, n_roots, VK_CID, /*want_keys=*/true, 0, ~0ULL, 1, VK_RECEIPT, w->receipt_lo, w->receipt_hi, plan);
    if (!plan.ok || plan.n_use != n_all || plan.n_leaves == 0 || plan.n_extra == 0) {
        IPCFP_HIP(ctx, sync_stream(ctx, ctx->stream));  // (the kernels above still use this call's scratch)
        return IPCFP_OK;
    }
    // ---- everything else, queued in one go ----
    const uint32_t n_msgs = uint32_t(plan.n_leaves), n_rcpt = uint32_t(plan.n_extra);
    DevBuf<DenseNode> a, b;
    DevBuf<LeafRef> rleaves;
    IPCFP_HIP(ctx, a.alloc(plan.biggest));
    IPCFP_HIP(ctx, b.alloc(plan.biggest));
    IPCFP_HIP(ctx, ex.keys.alloc(n_msgs));
    IPCFP_HIP(ctx, rleaves.alloc(n_rcpt));
    // the execution order's hash table and first-occurrence flags: sized now, cleared by the walk's last kernel on the way
    uint32_t size = 64;
    while (size < 2ull * n_msgs) size <<= 1;
    ex.mask = size - 1;
    ex.raw_len = n_msgs;
    IPCFP_HIP(ctx, ex.slots.alloc(size));
    IPCFP_HIP(ctx, ex.first.alloc(n_msgs));
    IPCFP_HIP(ctx, ex.pos.alloc(n_msgs));
    IPCFP_HIP(ctx, ex.inv.alloc(n_msgs));
    const DenseClear clear{ex.slots.p, size, ex.first.p, n_msgs};
    prof.reset(new ProfileScope(ctx, IPCFP_K_AMT_WALK));
    // the receipt leaves are consumed on the aux stream (k_receipt_events), the message keys on the main stream: the two
    // leaf kernels fork accordingly and run side by side (IPCFP_LEAVES_AUX=0: both on the main stream, one after the other)
    static const bool leaves_aux = [] {
        const char* e = std::getenv("IPCFP_LEAVES_AUX");
        return !(e && std::atoi(e) == 0);
    }();
    rc = launch_dense_walk(ctx, view, dense_frontier.p, plan, a.p, b.p, nullptr, ex.keys.p, rleaves.p, small + 2,
                           leaves_aux ? ctx->stream_aux : nullptr, leaves_aux ? ctx->main_event : nullptr,
                           ctx->stream != swap.saved ? swap.saved : nullptr, ctx->narrow_event, ctx->narrow_max_wg, &clear);
    if (rc) return rc;
    // (the main stream now waits for everything the narrow stream was given: its small copies are the main stream's)
    for (auto& r : ctx->pending)
        if (ctx->stream_narrow && r.stream == ctx->stream_narrow) r.stream = ctx->stream;
    if (prof) prof->stream = ctx->stream;  // (started on the narrow stream, ends on the main one)
    prof.reset();
    if (ctx->k1_defer == 1 && (rc = k1_flush(ctx, true))) return rc;
    // the receipts' event records: aux stream, behind the block-order parse and behind the leaves just queued
    std::unique_ptr<EventTableCached> table(new EventTableCached());
    table->lo = w->receipt_lo;
    table->hi = w->receipt_hi;
    table->n = n_rcpt;
    table->events = w->bt_events.p;
    IPCFP_HIP(ctx, table->receipts.alloc(n_rcpt));
    IPCFP_HIP(ctx, table->err_word.alloc(1));
    table->has_counts = w->bt_has_filter;
    if (table->has_counts) {
        table->counts_filter = w->bt_filter;
        IPCFP_HIP(ctx, table->counts.alloc(n_rcpt));
    }
    if (!leaves_aux) {  // (forked: the leaves are already on the aux stream, in order before what follows)
        IPCFP_HIP(ctx, hipEventRecord(ctx->main_event, ctx->stream));
        IPCFP_HIP(ctx, hipStreamWaitEvent(ctx->stream_aux, ctx->main_event, 0));
    }
    IPCFP_HIP(ctx, hipMemsetAsync(table->err_word.p, 0xff, 8, ctx->stream_aux));  // kNoEnumError
    rc = launch_receipt_events(ctx, view, rleaves.p, n_rcpt, table->has_counts ? &w->bt_filter.filter : nullptr,
                               int(w->bt_filter.has_actor), w->bt_filter.actor, w->bt_blocks.p, table->receipts.p,
                               table->has_counts ? table->counts.p : nullptr, table->err_word.p, ctx->stream_aux);
    if (rc) return rc;
    IPCFP_HIP(ctx, hipEventRecord(ctx->aux_event, ctx->stream_aux));
    w->bt_joined = false;
    if (defer_rehash) {  // behind the receipts' event records, which the verify kernel waits for — this it does not wait for
        rc = launch_txmeta_rehash(ctx, ctx->stream_aux, view, tcs_d.p, ex.err.p);
        if (rc) return rc;
        IPCFP_HIP(ctx, hipEventRecord(ctx->rehash_event, ctx->stream_aux));
    }
    // the execution order: first-seen dedupe of the message CIDs, positions, inverse — main stream, beside the above
    prof.reset(new ProfileScope(ctx, IPCFP_K_EXEC_ORDER));
    // (table and flags were cleared by the walk's k_dense_link_leaves: launch_dense_walk `clear`)
    rc = launch_exec_insert_flags(ctx, ex.keys.p, n_msgs, ex.slots.p, ex.mask, ex.first.p);
    if (rc) return rc;
    DevBuf<uint64_t> tiles;
    IPCFP_HIP(ctx, tiles.alloc(size_t(div_up(n_msgs, 256)) + 2));
    IPCFP_HIP(ctx, ctl_words(ctx, ex.total_own, ex.total.p, 1, false));
    CtxFinish fin{};
    fin.err = ex.err.p;
    fin.total = ex.total.p;
    fin.first = ex.first.p;
    fin.pos = ex.pos.p;
    fin.inv = ex.inv.p;
    fin.slots = ex.slots.p;
    fin.keys = ex.keys.p;
    fin.mask = ex.mask;
    fin.raw_len = n_msgs;
    fin.receipt_leaves = rleaves.p;
    fin.n_receipt_leaves = n_rcpt;
    fin.receipt_first = w->receipt_lo;
    fin.receipt_recs = table->receipts.p;
    fin.event_recs = table->events;
    rc = launch_exec_finish_fused(ctx, tcs_d.p, fin, ex.first.p, ex.pos.p, tiles.p, ex.total.p, /*flags_ready=*/true);
    if (rc) return rc;
    prof.reset();
    rc = event_table_join(ctx, w);
    if (rc) return rc;
    if (ctx->k1_defer == 2 && (rc = k1_flush(ctx, true))) return rc;
    if ((rc = claims_ready(ctx))) return rc;  // claims that were crossing PCIe beside all of the above are in HBM
    rc = launch_verify_events(ctx, view, claims_d, n, tcs_d.p, 1, blob_d, blob_len, trust ? *trust : accept_all, filter, status_d,
                              where_d, /*tabulated=*/true);
    if (rc) return rc;
    if ((rc = k1_flush(ctx, true))) return rc;  // (mode 3, and whatever is still noted)
    // ---- a scan riding on this call (ipcfp_verify_and_scan_device): its tail right behind the verify kernel ----
    DevBuf<uint32_t> ride_counts, ride_offsets;
    DevBuf<unsigned long long> ride_err_own;
    unsigned long long ride_seq = 0;
    if (ride && ctx->scan_fused != 0 && ride->cap_matches <= (1ull << 26) && (!ride->has_d || ride->cap_receipts >= n_rcpt)) {
        const ScanParams want = scan_params_of(ride->filter, ride->has_actor, ride->actor);
        const EventTableView tview = table->view();
        const uint32_t* cnt = nullptr;
        const unsigned long long* table_err = nullptr;
        unsigned long long* err_p = nullptr;
        IPCFP_HIP(ctx, ctl_words(ctx, ride_err_own, err_p, 1, true));
        if (table->has_counts && std::memcmp(&want, &table->counts_filter, sizeof want) == 0) {
            cnt = table->counts.p;  // counted while the table was built (the context's scan hint was this filter)
            table_err = table->err_word.p;
        } else {
            IPCFP_HIP(ctx, ride_counts.alloc(n_rcpt));
            rc = launch_count_from_table(ctx, view, rleaves.p, n_rcpt, ride->filter, ride->has_actor, ride->actor, tview, ride_counts.p, err_p);
            if (rc) return rc;
            cnt = ride_counts.p;
        }
        IPCFP_HIP(ctx, ride_offsets.alloc(n_rcpt));
        unsigned long long* tail_scratch = nullptr;
        rc = scan_tail_scratch(ctx, div_up(n_rcpt, 1024), &tail_scratch);
        if (rc) return rc;
        ride_seq = ++ctx->mailbox_seq;
        rc = launch_scan_tail_fused(ctx, view, rleaves.p, n_rcpt, w->receipt_lo, ride->filter, ride->has_actor, ride->actor, tview, cnt,
                                    ride_offsets.p, ride->cap_matches ? ride->matches_d : nullptr, ride->matches_d ? ride->cap_matches : 0,
                                    ride->has_d, ride->has_d ? n_rcpt : 0, w->receipt_lo, tail_scratch, ++ctx->scan_epoch, table_err,
                                    err_p, ctx->mailbox_dev, ride_seq);
        if (rc) return rc;
        ctx->scan_hint = want;
        ctx->has_scan_hint = true;
    }
    if (defer_rehash) IPCFP_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->rehash_event, 0));  // its errors are in before the flags are read
    // ---- the one synchronisation: did the dense walk hold? ----
    uint32_t bad = 0;
    unsigned long long e = kNoEnumError;
    TipsetCtxDev facts;
    IPCFP_HIP(ctx, ctl_read(ctx, &bad, small + 2, 4));
    IPCFP_HIP(ctx, ctl_read(ctx, &e, ex.err.p, 8));
    IPCFP_HIP(ctx, ctl_read(ctx, &facts, tcs_d.p, sizeof facts));  // (rides on the control block's one read-back when the context lives there)
    IPCFP_HIP(ctx, sync_stream(ctx, ctx->stream, /*last_of_call=*/true));  // (a fallback below takes fresh words: still fine)
    IPCFP_HIP(ctx, hipGetLastError());
    if (bad || e != kNoEnumError || facts.child_status != IPCFP_ST_TRUE) {
        // (the aux stream may still be writing this call's buffers)
        IPCFP_HIP(ctx, hipStreamSynchronize(ctx->stream_aux));
        w->bt_joined = true;
        return IPCFP_OK;  // *done == false: verify_packed decides, in the reference's order of errors
    }
    tcs[0] = facts;
    // the enumeration of the receipts and their event table now belong to the witness: the scan that follows finds them
    {
        std::unique_ptr<EnumCached> en(new EnumCached());
        std::memcpy(en->root, facts.receipts_root.w, 40);
        en->version = 0;
        en->vkind = VK_RECEIPT;
        en->lo = w->receipt_lo;
        en->hi = w->receipt_hi;
        en->n = n_rcpt;
        en->error = kNoEnumError;
        en->dense = true;
        en->leaves.p = reinterpret_cast<uint8_t*>(rleaves.p);
        en->leaves.count = rleaves.count * sizeof(LeafRef);
        en->leaves.cap = rleaves.cap;
        en->leaves.owner = rleaves.owner;
        rleaves.p = nullptr;
        rleaves.count = rleaves.cap = 0;
        rleaves.owner = nullptr;
        w->enum_cache.push_back(std::move(en));
        std::memcpy(table->root, facts.receipts_root.w, 40);
        w->table_cache.push_back(std::move(table));
    }
    if (ride_seq) {  // (published before the synchronisation above returned: the tail ran ahead of it on the same stream)
        if (__atomic_load_n(ctx->mailbox, __ATOMIC_ACQUIRE) != ride_seq) return set_error(ctx, IPCFP_E_HIP, "the riding scan's results never reached the mailbox");
        const uint64_t nm = __atomic_load_n(ctx->mailbox + 1, __ATOMIC_RELAXED);
        unsigned long long e1 = __atomic_load_n(ctx->mailbox + 3, __ATOMIC_RELAXED);
        const unsigned long long et = __atomic_load_n(ctx->mailbox + 2, __ATOMIC_RELAXED);
        const uint64_t walk = __atomic_load_n(ctx->mailbox + 4, __ATOMIC_RELAXED);
        if (et < e1) e1 = et;
        if (e1 != kNoEnumError) {
            ride->status = enum_error_code(e1);
            ride->done = true;
        } else if (!walk) {  // (receipts the table does not cover: the caller's ordinary scan walks them)
            ride->status = IPCFP_ST_TRUE;
            ride->n_idx = n_rcpt;
            ride->n_matches = nm;
            ride->done = true;
        }
    }
    *done = true;
    return IPCFP_OK;
}

}  // namespace ipcfp
