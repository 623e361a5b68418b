// csrc/host/verify_events.cpp — `verify_event_proof` batches and `reconstruct_execution_order`.
//
// Host side of src/proofs/events/verifier.rs:51-74: parse the claim strings once, group the
// proofs by tipset context (parent_tipset_cids, child_block_cid), prepare each context on the
// device (header facts + execution order — both recomputed PER PROOF by the reference,
// events/verifier.rs:105,115,190), then verify the whole batch with one kernel.
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <unordered_map>
#include <vector>

#include "../common.h"
#include "../kernels/claims_dev.h"
#include "../kernels/amt_enum.h"
#include "../kernels/event_table.h"
#include "../kernels/tipset_ctx.h"
#include "../kernels/types_dev.h"
#include "../kernels/launch.h"
#include "cidstr.h"
#include "exec_state.h"
#include "tipset_wide.h"
#include "pack_claims.h"

using namespace ipcfp;

namespace ipcfp {

void parse_cid_claim(const char* s, CidKey& key, bool& parsed, bool& canonical);
CidKey key_from_slot(const uint8_t* slot40);

int exec_state_prepare(ipcfp_ctx* ctx, ExecState& ex, uint32_t n_parents) {
    IPCFP_HIP(ctx, ex.roots.alloc(2 * size_t(n_parents) + 1));
    IPCFP_HIP(ctx, ctl_words(ctx, ex.err_own, ex.err.p, 1, true));  // kNoEnumError
    return IPCFP_OK;
}

// Reconstruct the execution order of the context stored at ctx_d (device) on the device.
int build_exec_order(ipcfp_ctx* ctx, const WitnessView& view, const TipsetCtxDev* ctx_d, uint32_t n_parents,
                     ExecState& ex, int verify_txmeta, bool host_len, bool prepared, EnumExtra* extra,
                     const std::function<int()>* after_enum) {
    int rc;
    if (!prepared) {
        rc = exec_state_prepare(ctx, ex, n_parents);
        if (rc) return rc;
        rc = launch_exec_roots(ctx, view, ctx_d, ex.roots.p, ex.err.p, verify_txmeta, n_parents);
        if (rc) return rc;
    }
    AmtEnumResult en;
    rc = amt_enumerate(ctx, view, ex.roots.p, 2 * n_parents, VK_CID, ex.err.p, en, 0, ~0ULL, &ex.keys,
                       prepared ? extra : nullptr);
    if (rc) return rc;
    if (after_enum && *after_enum) {
        rc = (*after_enum)();
        if (rc) return rc;
    }
    const unsigned long long e = en.error;  // read back by the enumerator: stage-1 errors and its own, merged
    ex.status = e == kNoEnumError ? uint32_t(IPCFP_ST_TRUE) : enum_error_code(e);
    ex.raw_len = ex.exec_len = 0;
    if (ex.status != IPCFP_ST_TRUE) return IPCFP_OK;
    const uint32_t n = uint32_t(en.n_leaves);
    ex.raw_len = n;
    uint32_t size = 64;
    while (size < 2ull * n) size <<= 1;
    ex.mask = size - 1;
    if (!en.keys_written) IPCFP_HIP(ctx, ex.keys.alloc(n));
    IPCFP_HIP(ctx, ex.slots.alloc(size));
    IPCFP_HIP(ctx, ex.first.alloc(n));
    IPCFP_HIP(ctx, ex.pos.alloc(n));
    IPCFP_HIP(ctx, hipMemsetAsync(ex.slots.p, 0xff, size_t(size) * 8, ctx->stream));
    rc = launch_exec_dedup(ctx, view, en.keys_written ? nullptr : en.leaves.p, n, ex.keys.p, ex.slots.p, ex.mask, ex.first.p);
    if (rc) return rc;
    DevBuf<uint64_t> scratch;
    IPCFP_HIP(ctx, scratch.alloc(size_t(div_up(n, 1024)) + 2));
    IPCFP_HIP(ctx, ctl_words(ctx, ex.total_own, ex.total.p, 1, false));
    rc = launch_scan_u32(ctx, ex.first.p, n, ex.pos.p, ex.total.p, scratch.p);
    if (rc) return rc;
    // `en.leaves` and `scratch` go back to the pool here; reuse is ordered on the one stream
    if (!host_len) return IPCFP_OK;
    uint64_t distinct = 0;
    IPCFP_HIP(ctx, ctl_read(ctx, &distinct, ex.total.p, 8));
    IPCFP_HIP(ctx, sync_stream(ctx, ctx->stream));
    ex.exec_len = distinct;
    return IPCFP_OK;
}


// Shared device path of the string and the packed entry points: prepare every tipset context
// (header facts, execution order) and verify the batch.  `claims_d`, `blob_d`, `status_d` are device.
int verify_packed_fast(ipcfp_ctx* ctx, ipcfp_witness* w, std::vector<TipsetCtxDev>& tcs, const EventClaimPacked* claims_d,
                       uint32_t n, const uint8_t* blob_d, uint64_t blob_len, const ipcfp_trust_policy_t* trust,
                       const ipcfp_event_filter_t* filter, uint8_t* status_d, void* where_d, bool* done, ScanRide* ride);  // verify_fast.cpp

int verify_packed(ipcfp_ctx* ctx, ipcfp_witness* w, std::vector<TipsetCtxDev>& tcs, const EventClaimPacked* claims_d,
                  uint32_t n, const uint8_t* blob_d, uint64_t blob_len, const ipcfp_trust_policy_t* trust,
                  const ipcfp_event_filter_t* filter, uint8_t* status_d, void* where_d = nullptr, ScanRide* ride = nullptr) {
    static const ipcfp_trust_policy_t accept_all = {0, 0, 0, 0};
    {   // one tipset pair, receipts not enumerated yet: the route without a mid-call synchronisation (verify_fast.cpp);
        // whenever its dense walk does not hold, everything is done again below
        const std::vector<TipsetCtxDev> saved = tcs;
        bool done = false;
        int rc_fast = verify_packed_fast(ctx, w, tcs, claims_d, n, blob_d, blob_len, trust, filter, status_d, where_d, &done, ride);
        if (rc_fast) return rc_fast;
        if (done) return IPCFP_OK;
        if (int rc_k1 = k1_flush(ctx)) return rc_k1;  // (a noted K1 launch the fast route did not get to queue)
        tcs = saved;
    }
    const WitnessView view = witness_view(w);
    // when no scan has tabulated the events yet, the block-order parse runs beside the whole tipset prologue
    int rc_bt = ctx->has_scan_hint ? block_table_prefetch(ctx, w, &ctx->scan_hint.filter, int(ctx->scan_hint.has_actor), ctx->scan_hint.actor)
                                   : block_table_prefetch(ctx, w, nullptr, 0, 0);
    if (rc_bt) return rc_bt;
    DevBuf<TipsetCtxDev> tcs_d;
    IPCFP_HIP(ctx, tcs_d.alloc(tcs.size()));
    IPCFP_HIP(ctx, h2d_small(ctx, tcs_d.p, tcs.data(), tcs.size() * sizeof(TipsetCtxDev), ctx->stream));
    // The tipset prologue — header facts of every context and stage 1 of every execution order (parent
    // headers, TxMeta re-hash, message AMT roots) — is one launch.  The execution order is prepared for every
    // context whose claim strings parsed, before the header facts are known on the host; a context that fails
    // steps 1-2 simply never looks at it.
    std::vector<std::unique_ptr<ExecState>> execs(tcs.size());
    std::vector<PrepareJob> jobs(tcs.size());
    int rc;
    for (size_t k = 0; k < tcs.size(); ++k) {
        const TipsetCtxDev& in = tcs[k];
        jobs[k] = PrepareJob{tcs_d.p + k, nullptr, nullptr};
        if (!(in.flags & TC_PARENTS_PARSED) || !(in.flags & TC_CHILD_PARSED) || in.n_parents == 0) continue;
        execs[k].reset(new ExecState());
        rc = exec_state_prepare(ctx, *execs[k], in.n_parents);
        if (rc) return rc;
        jobs[k].roots = execs[k]->roots.p;
        jobs[k].err = execs[k]->err.p;
    }
    // (a context whose tipset key is wider than the inline form has a prologue launch of its own: tipset_wide.h)
    std::vector<PrepareJob> narrow;
    for (size_t k = 0; k < tcs.size(); ++k) {
        if (!tipset_is_wide(tcs[k].n_parents)) {
            narrow.push_back(jobs[k]);
            continue;
        }
        rc = launch_tipset_prepare_wide(ctx, view, &jobs[k], tcs[k].n_parents);
        if (rc) return rc;
    }
    DevBuf<PrepareJob> jobs_d;  // (a handful of jobs travel as a kernel argument instead)
    if (narrow.size() > kInlineJobs) {
        IPCFP_HIP(ctx, jobs_d.alloc(narrow.size()));
        IPCFP_HIP(ctx, h2d_small(ctx, jobs_d.p, narrow.data(), narrow.size() * sizeof(PrepareJob), ctx->stream));
    }
    rc = launch_tipset_prepare(ctx, view, narrow.data(), jobs_d.p, uint32_t(narrow.size()),
                               /*need_general=*/uint64_t(w->max_block_len) + 32u > uint64_t(kPrologueStageChunks) * 16u);
    if (rc) return rc;
    // the header facts come back with the first synchronisation below (the enumerator's), not one of their own
    std::vector<TipsetCtxDev> facts(tcs.size());
    IPCFP_HIP(ctx, d2h_small(ctx, facts.data(), tcs_d.p, tcs.size() * sizeof(TipsetCtxDev), ctx->stream));
    // The receipts AMT of a context rides along with its message AMTs (same per-level launches, same two
    // synchronisations) unless the witness already holds an enumeration of receipts — a scan that ran before.
    bool receipts_known = false;
    for (auto& e : w->enum_cache) receipts_known = receipts_known || (e->vkind == VK_RECEIPT && e->lo == w->receipt_lo && e->hi == w->receipt_hi);
    std::vector<std::unique_ptr<AmtEnumResult>> rec_en(tcs.size());
    std::vector<EnumExtra> rec_extra(tcs.size());
    bool synced = false;
    for (size_t k = 0; k < tcs.size(); ++k) {
        if (!execs[k]) continue;
        EnumExtra* extra = nullptr;
        if (!receipts_known) {
            rec_en[k].reset(new AmtEnumResult());
            rec_extra[k].vkind = VK_RECEIPT;
            rec_extra[k].lo = w->receipt_lo;
            rec_extra[k].hi = w->receipt_hi;
            rec_extra[k].out = rec_en[k].get();
            extra = &rec_extra[k];
        }
        // With the receipts on board, their event table (k_receipt_events) does not wait for the execution-order
        // hash kernels: it is queued on the aux stream — idle once the block-order parse is through — the moment the
        // enumeration is back, and joined before the verify kernel.
        const std::function<int()> start_table = [&, k]() -> int {
            const TipsetCtxDev& f = facts[k];
            if (!rec_en[k] || !rec_extra[k].done || f.child_status != IPCFP_ST_TRUE) return IPCFP_OK;
            int r = enum_cache_put(ctx, w, f.receipts_root, 0, VK_RECEIPT, w->receipt_lo, w->receipt_hi, *rec_en[k]);
            if (r) return r;
            rec_extra[k].done = false;  // (handed over)
            if (!w->use_event_table) return IPCFP_OK;
            const EnumCached* en_r = nullptr;
            r = amt_enumerate_cached(ctx, w, f.receipts_root, 0, VK_RECEIPT, &en_r, w->receipt_lo, w->receipt_hi);
            if (r) return r;
            if (en_r->error != kNoEnumError || !en_r->dense || !en_r->n) return IPCFP_OK;
            const EventTableCached* table = nullptr;
            return event_table_get(ctx, w, f.receipts_root, en_r, &table, /*on_aux=*/true);
        };
        rc = build_exec_order(ctx, view, tcs_d.p + k, tcs[k].n_parents, *execs[k], 1, /*host_len=*/false, /*prepared=*/true, extra,
                              extra ? &start_table : nullptr);
        if (rc) return rc;
        synced = true;
    }
    if (!synced) IPCFP_HIP(ctx, sync_stream(ctx, ctx->stream));
    for (size_t k = 0; k < tcs.size(); ++k) {
        TipsetCtxDev& tc = tcs[k];
        tc = facts[k];
        tc.exec_status = IPCFP_ST_ERR_BAD_CLAIM;
        tc.exec_slots = nullptr;
        tc.exec_inv = nullptr;
        tc.receipt_leaves = nullptr;
        tc.n_receipt_leaves = 0;
        tc.receipt_first = 0;
        tc.receipt_recs = nullptr;
        tc.event_recs = nullptr;
        // the execution order is only reached when steps 1-2 can pass for some proof of this context
        const bool reachable = (tc.flags & TC_PARENTS_PARSED) && (tc.flags & TC_CHILD_PARSED) &&
                               tc.child_status == IPCFP_ST_TRUE && tc.parents_match && tc.n_parents > 0 &&
                               tc.parent0_status == IPCFP_ST_TRUE;
        if (!reachable || !execs[k]) continue;
        tc.exec_status = execs[k]->status;
        tc.exec_mask = execs[k]->mask;
        tc.exec_slots = execs[k]->slots.p;
        tc.exec_keys = execs[k]->keys.p;
        tc.exec_pos = execs[k]->pos.p;
        tc.exec_inv = nullptr;
        if (execs[k]->status == IPCFP_ST_TRUE && execs[k]->total.p) {
            IPCFP_HIP(ctx, execs[k]->inv.alloc(execs[k]->raw_len));
            tc.exec_inv = execs[k]->inv.p;
        }
        tc.exec_len = 0;  // patched on the device below
        // receipts AMT of this context: enumerated once (shared with ipcfp_scan_events through the witness cache) —
        // along with the message AMTs above when it was, else by itself here
        if (rec_en[k] && rec_extra[k].done && tc.child_status == IPCFP_ST_TRUE) {
            bool have = false;
            for (auto& e : w->enum_cache)
                have = have || (e->version == 0 && e->vkind == VK_RECEIPT && e->lo == w->receipt_lo && e->hi == w->receipt_hi &&
                                std::memcmp(e->root, tc.receipts_root.w, 40) == 0);
            if (!have) {
                rc = enum_cache_put(ctx, w, tc.receipts_root, 0, VK_RECEIPT, w->receipt_lo, w->receipt_hi, *rec_en[k]);
                if (rc) return rc;
            }
        }
        const EnumCached* rc_enum = nullptr;
        rc = amt_enumerate_cached(ctx, w, tc.receipts_root, 0, VK_RECEIPT, &rc_enum, w->receipt_lo, w->receipt_hi);
        if (rc) return rc;
        if (rc_enum->error == kNoEnumError && rc_enum->dense) {
            tc.receipt_leaves = reinterpret_cast<const LeafRef*>(rc_enum->leaves.p);
            tc.n_receipt_leaves = rc_enum->n;
            tc.receipt_first = rc_enum->n ? w->receipt_lo : 0;
            // ... and their events tabulated once (shared with the scan through the witness cache)
            if (w->use_event_table && rc_enum->n) {
                const EventTableCached* table = nullptr;
                rc = event_table_get(ctx, w, tc.receipts_root, rc_enum, &table);
                if (rc) return rc;
                tc.receipt_recs = table->receipts.p;
                tc.event_recs = table->events;
            }
        }
    }
    IPCFP_HIP(ctx, h2d_small(ctx, tcs_d.p, tcs.data(), tcs.size() * sizeof(TipsetCtxDev), ctx->stream));
    for (size_t k = 0; k < tcs.size(); ++k)
        if (tcs[k].exec_slots && execs[k]->status == IPCFP_ST_TRUE && execs[k]->total.p) {
            rc = launch_exec_finish(ctx, tcs_d.p + k, execs[k]->total.p, execs[k]->first.p, execs[k]->pos.p,
                                    uint32_t(execs[k]->raw_len), execs[k]->inv.p);
            if (rc) return rc;
        }
    rc = event_table_join(ctx, w);
    if (rc) return rc;
    bool tabulated = false;
    for (auto& tc : tcs) tabulated = tabulated || tc.receipt_recs != nullptr;
    if ((rc = claims_ready(ctx))) return rc;  // the claims are in HBM
    rc = launch_verify_events(ctx, view, claims_d, n, tcs_d.p, uint32_t(tcs.size()), blob_d, blob_len,
                              trust ? *trust : accept_all, filter, status_d, where_d, tabulated);
    if (rc) return rc;
    IPCFP_HIP(ctx, sync_stream(ctx, ctx->stream));  // contexts / exec tables are released on return
    return IPCFP_OK;
}

}  // namespace ipcfp


extern "C" {

static int verify_event_proofs_impl(ipcfp_ctx_t* ctx, ipcfp_witness_t* w, const ipcfp_event_proof_t* proofs, uint64_t n,
                                    const ipcfp_trust_policy_t* trust, const ipcfp_event_filter_t* filter,
                                    ipcfp_status_t* status, ipcfp_value_loc_t* event_loc);

int ipcfp_verify_event_proofs(ipcfp_ctx_t* ctx, ipcfp_witness_t* w, const ipcfp_event_proof_t* proofs, uint64_t n,
                              const ipcfp_trust_policy_t* trust, const ipcfp_event_filter_t* filter,
                              ipcfp_status_t* status) {
    return verify_event_proofs_impl(ctx, w, proofs, n, trust, filter, status, nullptr);
}

// The same, also reporting WHERE each proof's event lies (event_loc[i].block == 0xffffffff when the proof did not get
// that far): the door for an arbitrary host `check_event(&ActorEvent)` closure (events/verifier.rs:51-56,247-251) —
// the wrapper reads the located bytes (ipcfp_witness_read_values), runs the closure over the proofs whose status is
// TRUE, and turns a `false` into IPCFP_ST_FALSE_FILTER.
int ipcfp_verify_event_proofs_located(ipcfp_ctx_t* ctx, ipcfp_witness_t* w, const ipcfp_event_proof_t* proofs, uint64_t n,
                                      const ipcfp_trust_policy_t* trust, const ipcfp_event_filter_t* filter,
                                      ipcfp_status_t* status, ipcfp_value_loc_t* event_loc) {
    if (n && !event_loc) return IPCFP_E_INVALID;
    return verify_event_proofs_impl(ctx, w, proofs, n, trust, filter, status, event_loc);
}

// verify_event_proof with an arbitrary host predicate (events/verifier.rs:51-56, applied at :247-251): the fold
// "still true and the predicate declines => Ok(false)" happens here, behind the ABI.
int ipcfp_verify_event_proofs_with(ipcfp_ctx_t* ctx, ipcfp_witness_t* w, const ipcfp_event_proof_t* proofs, uint64_t n,
                                   const ipcfp_trust_policy_t* trust, const ipcfp_event_filter_t* filter,
                                   ipcfp_check_event_fn check_event, void* user, ipcfp_status_t* status) {
    if (!check_event) return verify_event_proofs_impl(ctx, w, proofs, n, trust, filter, status, nullptr);
    if (n == 0) return ctx && w && w->ctx == ctx ? IPCFP_OK : IPCFP_E_INVALID;
    std::vector<ipcfp_value_loc_t> loc(n);
    int rc = verify_event_proofs_impl(ctx, w, proofs, n, trust, filter, status, loc.data());
    if (rc) return rc;
    // the events of the proofs that are still true, compacted (a proof that failed earlier never reaches the predicate)
    std::vector<uint64_t> who;
    std::vector<ipcfp_value_loc_t> want;
    uint64_t stride = 16;
    for (uint64_t i = 0; i < n; ++i)
        if (status[i] == IPCFP_ST_TRUE) {
            // a proof that verified HAS a located event; one without would skip the predicate and stay true (fail open)
            if (loc[i].block == 0xffffffffu)
                return set_error(ctx, IPCFP_E_HIP, "internal: proof %llu verified without a located event", (unsigned long long)i);
            who.push_back(i);
            want.push_back(loc[i]);
            stride = std::max<uint64_t>(stride, (uint64_t(loc[i].len) + 15) & ~uint64_t(15));
        }
    // chunks bound the host buffer (an event is < 64 KB, usually ~130 B)
    const uint64_t per = std::max<uint64_t>(1, (uint64_t(128) << 20) / stride);
    std::vector<uint8_t> bytes;
    for (uint64_t at = 0; at < who.size(); at += per) {
        const uint64_t m = std::min<uint64_t>(per, who.size() - at);
        bytes.resize(m * stride);
        rc = ipcfp_witness_read_values(ctx, w, want.data() + at, m, bytes.data(), stride);
        if (rc) return rc;
        for (uint64_t k = 0; k < m; ++k)
            if (!check_event(user, who[at + k], bytes.data() + k * stride, want[at + k].len)) status[who[at + k]] = IPCFP_ST_FALSE_FILTER;
    }
    return IPCFP_OK;
}

static int verify_event_proofs_impl(ipcfp_ctx_t* ctx, ipcfp_witness_t* w, const ipcfp_event_proof_t* proofs, uint64_t n,
                                    const ipcfp_trust_policy_t* trust, const ipcfp_event_filter_t* filter,
                                    ipcfp_status_t* status, ipcfp_value_loc_t* event_loc) {
    if (!ctx || !w || w->ctx != ctx || (n && (!proofs || !status))) return IPCFP_E_INVALID;
    if (n >= 0xffffffffULL) return set_error(ctx, IPCFP_E_UNSUPPORTED, "batch too large");
    if (n == 0) return IPCFP_OK;
    IPCFP_ENTER(ctx);

    // ---- parse claims, group by tipset context (host/pack_claims.cpp: parallel over claim ranges) ----
    PackedEvents pk;
    {
        std::string perr;
        const int prc = pack_event_claims_host(proofs, n, pk, perr);
        if (prc) return set_error(ctx, prc, "%s", perr.c_str());
    }
    std::vector<EventClaimPacked>& packed = pk.claims;
    std::vector<uint8_t>& blob = pk.blob;
    std::vector<TipsetCtxDev> tcs(pk.tipsets.size());
    WideParents wide;
    for (size_t k = 0; k < tcs.size(); ++k)
        if (int rc_t = tipset_inputs(ctx, pk.tipsets[k], tcs[k], wide)) return rc_t;

    // ---- upload, then the shared device path ----
    DevBuf<EventClaimPacked> cd;
    DevBuf<uint8_t> bd, sd;
    IPCFP_HIP(ctx, cd.alloc(n));
    IPCFP_HIP(ctx, bd.alloc(blob.size() + 64));
    IPCFP_HIP(ctx, sd.alloc(n));
    IPCFP_HIP(ctx, hipMemcpyAsync(cd.p, packed.data(), n * sizeof(EventClaimPacked), hipMemcpyHostToDevice, ctx->stream));
    if (!blob.empty())
        IPCFP_HIP(ctx, hipMemcpyAsync(bd.p, blob.data(), blob.size(), hipMemcpyHostToDevice, ctx->stream));
    DevBuf<ipcfp_value_loc_t> ld;
    if (event_loc) IPCFP_HIP(ctx, ld.alloc(n));
    int rc = verify_packed(ctx, w, tcs, cd.p, uint32_t(n), bd.p, blob.size(), trust, filter, sd.p, event_loc ? ld.p : nullptr);
    if (rc) return rc;
    if (event_loc) IPCFP_HIP(ctx, hipMemcpyAsync(event_loc, ld.p, n * sizeof(ipcfp_value_loc_t), hipMemcpyDeviceToHost, ctx->stream));
    IPCFP_HIP(ctx, hipMemcpyAsync(status, sd.p, n, hipMemcpyDeviceToHost, ctx->stream));
    IPCFP_HIP(ctx, sync_stream(ctx, ctx->stream));
    return IPCFP_OK;
}

int ipcfp_verify_event_claims_device(ipcfp_ctx_t* ctx, ipcfp_witness_t* w, const ipcfp_tipset_ref_t* tipsets,
                                     uint32_t n_tipsets, const void* claims_d, uint64_t n, const void* blob_d,
                                     uint64_t blob_len, const ipcfp_trust_policy_t* trust,
                                     const ipcfp_event_filter_t* filter, void* status_d) {
    if (!ctx || !w || w->ctx != ctx || (n && (!claims_d || !status_d || !tipsets))) return IPCFP_E_INVALID;
    if (n >= 0xffffffffULL) return set_error(ctx, IPCFP_E_UNSUPPORTED, "batch too large");
    if (n == 0) return IPCFP_OK;
    IPCFP_ENTER(ctx);
    std::vector<TipsetCtxDev> tcs(n_tipsets);
    WideParents wide;
    for (uint32_t k = 0; k < n_tipsets; ++k)
        if (int rc_t = tipset_inputs(ctx, tipsets[k], tcs[k], wide)) return rc_t;
    int rc = verify_packed(ctx, w, tcs, static_cast<const EventClaimPacked*>(claims_d), uint32_t(n),
                           static_cast<const uint8_t*>(blob_d), blob_len, trust, filter, static_cast<uint8_t*>(status_d));
    if (rc) return rc;
    IPCFP_HIP(ctx, sync_stream(ctx, ctx->stream));
    ctl_preprime(ctx);
    return IPCFP_OK;
}

// ipcfp_verify_event_claims_device and ipcfp_scan_events_device (of tipsets[0]'s child: its receipts AMT) in ONE call:
// the scan's tail is queued behind the verify kernel and both come back with one synchronisation — two host round
// trips and the gap between two calls less (≈ 80 µs of a 1.1 ms pass).  Same results as the two calls in that order.
int ipcfp_verify_and_scan_device(ipcfp_ctx_t* ctx, ipcfp_witness_t* w, const ipcfp_tipset_ref_t* tipsets, uint32_t n_tipsets,
                                 const void* claims_d, uint64_t n, const void* blob_d, uint64_t blob_len,
                                 const ipcfp_trust_policy_t* trust, const ipcfp_event_filter_t* check_filter, void* status_d,
                                 const ipcfp_event_filter_t* scan_filter, int has_actor, uint64_t actor,
                                 ipcfp_status_t* scan_status, void* receipt_has_match_d, uint64_t cap_receipts,
                                 uint64_t* n_receipts, void* matches_d, uint64_t cap_matches, uint64_t* n_matches) {
    if (!ctx || !w || w->ctx != ctx || !tipsets || n_tipsets == 0 || (n && (!claims_d || !status_d)) || !scan_filter || !scan_status ||
        !n_receipts || !n_matches)
        return IPCFP_E_INVALID;
    if (n >= 0xffffffffULL) return set_error(ctx, IPCFP_E_UNSUPPORTED, "batch too large");
    IPCFP_ENTER(ctx);
    *scan_status = IPCFP_ST_ERR;
    *n_receipts = *n_matches = 0;
    std::vector<TipsetCtxDev> tcs(n_tipsets);
    WideParents wide;
    for (uint32_t k = 0; k < n_tipsets; ++k)
        if (int rc_t = tipset_inputs(ctx, tipsets[k], tcs[k], wide)) return rc_t;
    ScanRide ride;
    ride.filter = *scan_filter;
    ride.has_actor = has_actor;
    ride.actor = actor;
    ride.has_d = static_cast<uint8_t*>(receipt_has_match_d);
    ride.cap_receipts = receipt_has_match_d ? cap_receipts : 0;
    ride.matches_d = cap_matches ? static_cast<ipcfp_event_match_t*>(matches_d) : nullptr;
    ride.cap_matches = matches_d ? cap_matches : 0;
    const CidKey child0 = tcs[0].child;
    if (n) {
        int rc = verify_packed(ctx, w, tcs, static_cast<const EventClaimPacked*>(claims_d), uint32_t(n),
                               static_cast<const uint8_t*>(blob_d), blob_len, trust, check_filter, static_cast<uint8_t*>(status_d),
                               nullptr, &ride);
        if (rc) return rc;
        IPCFP_HIP(ctx, sync_stream(ctx, ctx->stream));
    }
    if (ride.done) {
        *scan_status = ipcfp_status_t(ride.status);
        if (ride.status == IPCFP_ST_TRUE) {
            *n_receipts = ride.n_idx;
            *n_matches = ride.n_matches;
        }
        ctl_preprime(ctx);
        return IPCFP_OK;
    }
    // the ride did not happen: the ordinary scan of the child's receipts AMT (HeaderLite.parent_message_receipts:
    // src/proofs/common/decode.rs:100-118), with whatever the verify call left cached
    const WitnessView view = witness_view(w);
    CidKey receipts_root;
    {
        std::vector<TipsetCtxDev> one(1);
        std::memset(&one[0], 0, sizeof(TipsetCtxDev));
        one[0].flags = TC_PARENTS_PARSED | TC_CHILD_PARSED;
        one[0].child = child0;
        DevBuf<TipsetCtxDev> tc_d;
        IPCFP_HIP(ctx, tc_d.alloc(1));
        IPCFP_HIP(ctx, h2d_small(ctx, tc_d.p, one.data(), sizeof(TipsetCtxDev), ctx->stream));
        int rc = launch_ctx_headers(ctx, view, tc_d.p, 1);
        if (rc) return rc;
        IPCFP_HIP(ctx, d2h_small(ctx, one.data(), tc_d.p, sizeof(TipsetCtxDev), ctx->stream));
        IPCFP_HIP(ctx, sync_stream(ctx, ctx->stream));
        if (one[0].child_status != IPCFP_ST_TRUE) {
            *scan_status = ipcfp_status_t(one[0].child_status);
            ctl_preprime(ctx);
            return IPCFP_OK;
        }
        receipts_root = one[0].receipts_root;
    }
    uint8_t root40[IPCFP_CID_SLOT];
    std::memcpy(root40, receipts_root.w, IPCFP_CID_SLOT);
    return ipcfp_scan_events_device(ctx, w, root40, scan_filter, has_actor, actor, scan_status, receipt_has_match_d, cap_receipts,
                                    n_receipts, matches_d, cap_matches, n_matches, nullptr);
}

// Packed claims in HOST memory: upload, verify, status bytes back (the T2 window of the benchmarks: PCIe inclusive).
int ipcfp_verify_event_claims(ipcfp_ctx_t* ctx, ipcfp_witness_t* w, const ipcfp_tipset_ref_t* tipsets, uint32_t n_tipsets,
                              const ipcfp_event_claim_t* claims, uint64_t n, const uint8_t* blob, uint64_t blob_len,
                              const ipcfp_trust_policy_t* trust, const ipcfp_event_filter_t* filter,
                              ipcfp_status_t* status) {
    if (!ctx || !w || w->ctx != ctx || (n && (!claims || !status || !tipsets)) || (blob_len && !blob)) return IPCFP_E_INVALID;
    if (n >= 0xffffffffULL) return set_error(ctx, IPCFP_E_UNSUPPORTED, "batch too large");
    if (n == 0) return IPCFP_OK;
    IPCFP_ENTER(ctx);
    std::vector<TipsetCtxDev> tcs(n_tipsets);
    WideParents wide;
    for (uint32_t k = 0; k < n_tipsets; ++k)
        if (int rc_t = tipset_inputs(ctx, tipsets[k], tcs[k], wide)) return rc_t;
    DevBuf<EventClaimPacked> cd;
    DevBuf<uint8_t> bd, sd;
    IPCFP_HIP(ctx, cd.alloc(n));
    IPCFP_HIP(ctx, bd.alloc(blob_len + 64));
    IPCFP_HIP(ctx, sd.alloc(n));
    // The claims cross PCIe on a thread of their own while this one queues the tipset prologue, the AMT walk and the
    // execution order, none of which reads a claim; launch_verify_events' callers wait for the copy (upload_task_wait).
    // (IPCFP_UPLOAD_MODE=1, or a batch too small to matter: uploaded here, first)
    int rc = IPCFP_OK;
    static const bool beside = [] {
        const char* e = std::getenv("IPCFP_UPLOAD_MODE");
        return !(e && std::atoi(e) != 0);
    }();
    if (beside && n * sizeof(EventClaimPacked) >= (size_t(8) << 20)) {
        IPCFP_HIP(ctx, hipStreamSynchronize(ctx->stream));  // (nothing queued earlier may still use the buffers just taken)
        ctx->upload_task = upload_task_start(ctx, cd.p, claims, n * sizeof(EventClaimPacked), bd.p, blob, blob_len);
    }
    if (!ctx->upload_task) {  // small batch, IPCFP_UPLOAD_MODE=1, or no thread to be had: uploaded here, first
        rc = upload(ctx, cd.p, claims, n * sizeof(EventClaimPacked), ctx->stream);
        if (rc) return rc;
        if (blob_len) {
            rc = upload(ctx, bd.p, blob, blob_len, ctx->stream);
            if (rc) return rc;
        }
    }
    rc = verify_packed(ctx, w, tcs, cd.p, uint32_t(n), bd.p, blob_len, trust, filter, sd.p);
    {
        const int rc_up = upload_task_wait(ctx);  // (whatever happened: the copy must be over before cd / bd go back to the pool)
        if (rc == IPCFP_OK) rc = rc_up;
    }
    if (rc) return rc;
    IPCFP_HIP(ctx, hipMemcpyAsync(status, sd.p, n, hipMemcpyDeviceToHost, ctx->stream));
    IPCFP_HIP(ctx, sync_stream(ctx, ctx->stream));
    return IPCFP_OK;
}

// The claims of the receipts [receipt_lo, receipt_hi) out of a batch in exec_index order, which stays where it is in host
// memory: two binary searches find the records, they are uploaded as they are, the window of the blob they point into is
// found ON THE DEVICE (one reduction, one read-back), uploaded beside the walk, and the records' offsets are rebased in
// front of the verify kernel (kernels/claims_compact.hip).  No pass over the batch on the host.
int ipcfp_verify_event_claims_range(ipcfp_ctx_t* ctx, ipcfp_witness_t* w, const ipcfp_tipset_ref_t* tipsets, uint32_t n_tipsets,
                                    const ipcfp_event_claim_t* claims, uint64_t n, const uint8_t* blob, uint64_t blob_len,
                                    uint64_t receipt_lo, uint64_t receipt_hi, int last_shard, const ipcfp_trust_policy_t* trust,
                                    const ipcfp_event_filter_t* filter, uint64_t* first_out, uint64_t* count_out,
                                    ipcfp_status_t* status) {
    if (!ctx || !w || w->ctx != ctx || !first_out || !count_out || (n && (!claims || !status || !tipsets)) || (blob_len && !blob) ||
        receipt_lo > receipt_hi)
        return IPCFP_E_INVALID;
    if (n >= 0xffffffffULL) return set_error(ctx, IPCFP_E_UNSUPPORTED, "batch too large");
    auto lower_bound = [&](uint64_t key) {
        uint64_t lo = 0, hi = n;
        while (lo < hi) {
            const uint64_t mid = lo + (hi - lo) / 2;
            if (claims[mid].exec_index < key) lo = mid + 1; else hi = mid;
        }
        return lo;
    };
    const uint64_t a = lower_bound(receipt_lo), b = last_shard ? n : lower_bound(receipt_hi);
    *first_out = a;
    *count_out = b > a ? b - a : 0;
    if (b <= a) return IPCFP_OK;
    const uint64_t m = b - a;
    // (the order is the caller's promise; what can be checked for nothing is checked: the slice's ends)
    if (claims[a].exec_index > claims[b - 1].exec_index || (a && claims[a - 1].exec_index > claims[a].exec_index) ||
        (b < n && claims[b - 1].exec_index > claims[b].exec_index))
        return set_error(ctx, IPCFP_E_INVALID, "the claim batch is not in exec_index order (ipcfp_route_event_claims takes any order)");
    IPCFP_ENTER(ctx);
    std::vector<TipsetCtxDev> tcs(n_tipsets);
    WideParents wide;
    for (uint32_t k = 0; k < n_tipsets; ++k)
        if (int rc_t = tipset_inputs(ctx, tipsets[k], tcs[k], wide)) return rc_t;
    // Two ways to the window of the blob the slice points into.
    //   guessed   from the slice's two ends on the host (a bundle's blob is written in claim order: the first records'
    //             smallest offset, the last records' largest end) — records AND window then cross PCIe on a thread of their
    //             own while this one queues the tipset prologue, the AMT walks and the execution order, and the rebase
    //             kernel in front of the verify kernel says whether any record lay inside the batch's blob but outside
    //             the guess (`miss`);
    //   exact     records first, one reduction over them on the device, one synchronisation, then the window — the
    //             fallback when the guess missed (a blob in another order), and the only way for a slice too small to be
    //             worth a thread.
    auto window_of = [&](uint64_t i0, uint64_t i1, uint64_t& lo, uint64_t& hi) {
        for (uint64_t i = i0; i < i1; ++i) {
            const ipcfp_event_claim_t& c = claims[i];
            if (c.n_topics) {
                lo = std::min<uint64_t>(lo, c.topics_off);
                hi = std::max<uint64_t>(hi, uint64_t(c.topics_off) + 33ull * c.n_topics);
            }
            if (c.data_len) {
                lo = std::min<uint64_t>(lo, c.data_off);
                hi = std::max<uint64_t>(hi, uint64_t(c.data_off) + c.data_len);
            }
        }
    };
    auto attempt = [&](bool guessed, bool* missed) -> int {
        DevBuf<EventClaimPacked> cd;
        DevBuf<uint8_t> bd, sd;
        DevBuf<unsigned long long> win_d;
        DevBuf<uint32_t> miss_d, order_d;
        IPCFP_HIP(ctx, cd.alloc(m));
        IPCFP_HIP(ctx, order_d.alloc(1));
        IPCFP_HIP(ctx, hipMemsetAsync(order_d.p, 0, 4, ctx->stream));
        IPCFP_HIP(ctx, sd.alloc(m));
        uint64_t o0 = ~0ull, o1 = 0;
        int rc = IPCFP_OK;
        if (guessed) {
            constexpr uint64_t kEnds = 256;  // records looked at on either end
            window_of(a, std::min(b, a + kEnds), o0, o1);
            window_of(b > a + kEnds ? std::max(a + kEnds, b - kEnds) : b, b, o0, o1);
            IPCFP_HIP(ctx, miss_d.alloc(1));
            IPCFP_HIP(ctx, hipMemsetAsync(miss_d.p, 0, 4, ctx->stream));
        } else {
            IPCFP_HIP(ctx, win_d.alloc(2));
            const unsigned long long win0[2] = {~0ull, 0ull};
            IPCFP_HIP(ctx, h2d_small(ctx, win_d.p, win0, sizeof win0, ctx->stream));
            rc = upload(ctx, cd.p, claims + a, m * sizeof(EventClaimPacked), ctx->stream);
            if (rc) return rc;
            rc = launch_claims_window(ctx, cd.p, uint32_t(m), win_d.p);
            if (rc) return rc;
            unsigned long long win[2] = {0, 0};
            IPCFP_HIP(ctx, d2h_small(ctx, win, win_d.p, sizeof win, ctx->stream));
            IPCFP_HIP(ctx, sync_stream(ctx, ctx->stream));
            o0 = win[0];
            o1 = win[1];
        }
        if (o0 == ~0ull) o0 = 0;
        if (o0 > blob_len) o0 = blob_len;  // (records that point outside the batch's blob: the window is clamped, they are refused)
        if (o1 > blob_len) o1 = blob_len;
        if (o1 < o0) o1 = o0;
        const uint64_t wlen = o1 - o0;
        IPCFP_HIP(ctx, bd.alloc(wlen + 64));
        if (guessed) {
            IPCFP_HIP(ctx, hipStreamSynchronize(ctx->stream));  // (nothing queued earlier may still use the buffers just taken)
            ctx->upload_task = upload_task_start(ctx, cd.p, claims + a, m * sizeof(EventClaimPacked), bd.p, blob + o0, wlen);
            if (!ctx->upload_task) {
                rc = upload(ctx, cd.p, claims + a, m * sizeof(EventClaimPacked), ctx->stream);
                if (!rc && wlen) rc = upload(ctx, bd.p, blob + o0, wlen, ctx->stream);
                if (rc) return rc;
            }
        } else {
            if (wlen >= (size_t(4) << 20)) ctx->upload_task = upload_task_start(ctx, bd.p, blob + o0, wlen, nullptr, nullptr, 0);
            if (!ctx->upload_task && wlen) {
                rc = upload(ctx, bd.p, blob + o0, wlen, ctx->stream);
                if (rc) return rc;
            }
        }
        ctx->claims_rebase.pending = true;  // (queued by claims_ready: behind the copy, in front of whichever kernel reads the claims first)
        ctx->claims_rebase.claims_d = cd.p;
        ctx->claims_rebase.n = uint32_t(m);
        ctx->claims_rebase.base = o0;
        ctx->claims_rebase.blob_len = wlen;
        ctx->claims_rebase.full_len = blob_len;
        ctx->claims_rebase.miss_d = guessed ? miss_d.p : nullptr;
        ctx->claims_rebase.order_d = order_d.p;
        ctx->claims_rebase.key_lo = receipt_lo;
        ctx->claims_rebase.key_hi = last_shard ? ~0ull : receipt_hi;
        rc = verify_packed(ctx, w, tcs, cd.p, uint32_t(m), bd.p, wlen, trust, filter, sd.p);
        {
            const int rc_up = upload_task_wait(ctx);
            if (rc == IPCFP_OK) rc = rc_up;
            if (ctx->claims_rebase.pending) {  // no route reached its verify kernel: nothing may be reported
                ctx->claims_rebase.pending = false;
                if (rc == IPCFP_OK) rc = set_error(ctx, IPCFP_E_INVALID, "the claim slice was never rebased (no route reached its verify kernel)");
            }
        }
        if (rc) {
            (void)hipStreamSynchronize(ctx->stream);
            return rc;
        }
        uint32_t miss = 0, disorder = 0;
        if (guessed) IPCFP_HIP(ctx, d2h_small(ctx, &miss, miss_d.p, 4, ctx->stream));
        IPCFP_HIP(ctx, d2h_small(ctx, &disorder, order_d.p, 4, ctx->stream));
        IPCFP_HIP(ctx, hipMemcpyAsync(status, sd.p, m, hipMemcpyDeviceToHost, ctx->stream));
        IPCFP_HIP(ctx, sync_stream(ctx, ctx->stream));
        *missed = miss != 0;
        if (disorder)  // (checked over the whole slice on the device, where the records are: the binary searches above trusted it)
            return set_error(ctx, IPCFP_E_INVALID, "the claim batch is not in exec_index order (ipcfp_route_event_claims takes any order)");
        return IPCFP_OK;
    };
    static const bool allow_guess = [] {
        const char* e = std::getenv("IPCFP_CLAIMS_WINDOW_GUESS");
        return !(e && std::atoi(e) == 0);
    }();
    bool missed = false;
    if (allow_guess && m * sizeof(EventClaimPacked) >= (size_t(1) << 20)) {  // (below: a thread and a join cost what the synchronisation does)
        const int rc = attempt(true, &missed);
        if (rc || !missed) return rc;
    }
    return attempt(false, &missed);
}

// reconstruct_execution_order(bs, parent_hdr_cids) (src/proofs/events/utils.rs:16-30).
//   *status_out  IPCFP_ST_TRUE or the ERR_* the reference's `?` would surface first
//   *count       number of messages in execution order
//   out_cids40   receives min(*count, cap) CIDs (40-byte slots), nullable
int ipcfp_exec_order(ipcfp_ctx_t* ctx, ipcfp_witness_t* w, const uint8_t* parent_cids40, uint32_t n_parents,
                     ipcfp_status_t* status_out, uint8_t* out_cids40, uint64_t cap, uint64_t* count) {
    if (!ctx || !w || w->ctx != ctx || !status_out || !count || (n_parents && !parent_cids40)) return IPCFP_E_INVALID;
    IPCFP_ENTER(ctx);
    TipsetCtxDev tc;
    WideParents wide;
    if (int rc_t = tipset_inputs_list(ctx, TC_PARENTS_PARSED | TC_CHILD_PARSED, parent_cids40, n_parents, nullptr, tc, wide)) return rc_t;
    DevBuf<TipsetCtxDev> tc_d;
    IPCFP_HIP(ctx, tc_d.alloc(1));
    IPCFP_HIP(ctx, hipMemcpyAsync(tc_d.p, &tc, sizeof tc, hipMemcpyHostToDevice, ctx->stream));
    ExecState ex;
    int rc = build_exec_order(ctx, witness_view(w), tc_d.p, n_parents, ex);
    if (rc) return rc;
    *status_out = ipcfp_status_t(ex.status);
    *count = ex.status == IPCFP_ST_TRUE ? ex.exec_len : 0;
    if (ex.status == IPCFP_ST_TRUE && out_cids40 && ex.exec_len) {
        DevBuf<CidKey> out;
        IPCFP_HIP(ctx, out.alloc(ex.exec_len));
        rc = launch_exec_compact(ctx, ex.keys.p, uint32_t(ex.raw_len), ex.first.p, ex.pos.p, out.p);
        if (rc) return rc;
        const uint64_t take = ex.exec_len < cap ? ex.exec_len : cap;
        IPCFP_HIP(ctx, hipMemcpyAsync(out_cids40, out.p, take * IPCFP_CID_SLOT, hipMemcpyDeviceToHost, ctx->stream));
        IPCFP_HIP(ctx, sync_stream(ctx, ctx->stream));
    }
    return IPCFP_OK;
}

}  // extern "C"
