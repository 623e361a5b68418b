// csrc/host/claims_compact.cpp — EventProof claims in transport form (include/ipcfp.h ipcfp_event_claim_compact_t):
// the host-side conversion, the device expansion's entry point and the PCIe-inclusive verify call that uses both.
//
// Reference counterpart: the `event_proofs` of a UnifiedProofBundle (src/proofs/common/bundle.rs:36-45,
// src/proofs/events/bundle.rs:5-23) handed to `verify_proof_bundle` (src/proofs/verifier.rs:49-54).
#include <cstring>
#include <map>
#include <tuple>
#include <vector>

#include "../common.h"
#include "../kernels/claims_dev.h"
#include "../kernels/launch.h"
#include "../kernels/tipset_ctx.h"
#include "exec_state.h"
#include "tipset_wide.h"

namespace ipcfp {

int verify_packed(ipcfp_ctx* ctx, ipcfp_witness* w, std::vector<TipsetCtxDev>& tcs, const EventClaimPacked* claims_d, uint32_t n,
                  const uint8_t* blob_d, uint64_t blob_len, const ipcfp_trust_policy_t* trust, const ipcfp_event_filter_t* filter,
                  uint8_t* status_d, void* where_d = nullptr, ScanRide* ride = nullptr);

int claims_ready(ipcfp_ctx* ctx) {
    int rc = upload_task_wait(ctx);
    if (ctx->claims_rebase.pending) {  // a slice of a larger batch: its blob offsets become offsets into what was uploaded
        ipcfp_ctx::ClaimsRebase& rb = ctx->claims_rebase;
        rb.pending = false;
        if (rc) return rc;
        rc = launch_rebase_claims(ctx, rb.claims_d, rb.n, rb.base, rb.blob_len, rb.full_len, rb.miss_d, rb.order_d, rb.key_lo, rb.key_hi);
    }
    ipcfp_ctx::ClaimsExpand& x = ctx->claims_expand;
    if (!x.pending) return rc;
    x.pending = false;
    if (rc) return rc;
    return launch_expand_claims(ctx, x.compact_d, x.n, x.groups_d, x.n_groups, x.cblob_d, x.cblob_len, x.claims_out_d, x.blob_out_d,
                                x.cap_blob, x.scratch_u32, x.scan_scratch);
}

}  // namespace ipcfp

using namespace ipcfp;

extern "C" {

int ipcfp_compact_event_claims(const ipcfp_event_claim_t* claims, uint64_t n, const uint8_t* blob, uint64_t blob_len,
                               ipcfp_event_claim_group_t* groups, uint32_t* n_groups, ipcfp_event_claim_compact_t* out,
                               uint8_t* out_blob, uint64_t cap_blob, uint64_t* out_blob_len) {
    if (!n_groups || !out_blob_len || (n && (!claims || !groups || !out)) || (blob_len && !blob)) return IPCFP_E_INVALID;
    static const uint8_t kStd[6] = {0x01, 0x71, 0xa0, 0xe4, 0x02, 0x20};
    std::map<std::tuple<int64_t, int64_t, uint32_t>, uint32_t> seen;
    uint64_t at = 0;
    for (uint64_t i = 0; i < n; ++i) {
        const ipcfp_event_claim_t& c = claims[i];
        if (c.exec_index > 0xffffffffull || c.event_index > 0xffffffffull || c.n_topics > IPCFP_COMPACT_MAX_TOPICS ||
            c.data_len > 0xffffu || (c.flags & ~3u))
            return IPCFP_E_UNSUPPORTED;
        if (uint64_t(c.topics_off) + 33ull * c.n_topics > blob_len || uint64_t(c.data_off) + c.data_len > blob_len) return IPCFP_E_UNSUPPORTED;
        if (c.flags & IPCFP_CLAIM_MSG_PARSED) {
            if (std::memcmp(c.message_cid, kStd, 6) != 0 || c.message_cid[38] || c.message_cid[39]) return IPCFP_E_UNSUPPORTED;
        }
        const auto key = std::make_tuple(c.parent_epoch, c.child_epoch, c.tipset);
        auto it = seen.find(key);
        if (it == seen.end()) {
            if (seen.size() >= IPCFP_COMPACT_MAX_GROUPS) return IPCFP_E_UNSUPPORTED;
            const uint32_t g = uint32_t(seen.size());
            groups[g] = ipcfp_event_claim_group_t{c.parent_epoch, c.child_epoch, c.tipset, 0u};
            it = seen.emplace(key, g).first;
        }
        const uint64_t need = 32ull * c.n_topics + c.data_len;
        if (at + need > cap_blob) return IPCFP_E_INVALID;
        ipcfp_event_claim_compact_t& o = out[i];
        std::memset(&o, 0, sizeof o);
        o.emitter = c.emitter;
        o.exec_index = uint32_t(c.exec_index);
        o.event_index = uint32_t(c.event_index);
        if (c.flags & IPCFP_CLAIM_MSG_PARSED) std::memcpy(o.message_digest, c.message_cid + 6, 32);
        o.data_len = uint16_t(c.data_len);
        o.n_topics = uint8_t(c.n_topics);
        o.flags = uint8_t(c.flags);
        o.group = uint8_t(it->second);
        for (uint32_t t = 0; t < c.n_topics; ++t) {
            const uint8_t* src = blob + c.topics_off + 33ull * t;
            if (src[0] > 1) return IPCFP_E_UNSUPPORTED;  // (the flag byte is 0 or 1 in the plain form)
            if (src[0]) o.topic_flags |= uint8_t(1u << t);
            std::memcpy(out_blob + at + 32ull * t, src + 1, 32);
        }
        if (c.data_len) std::memcpy(out_blob + at + 32ull * c.n_topics, blob + c.data_off, c.data_len);
        at += need;
    }
    *n_groups = uint32_t(seen.size());
    *out_blob_len = at;
    return IPCFP_OK;
}

int ipcfp_expand_event_claims_device(ipcfp_ctx_t* ctx, const ipcfp_event_claim_group_t* groups, uint32_t n_groups,
                                     const void* compact_d, uint64_t n, const void* cblob_d, uint64_t cblob_len,
                                     void* claims_out_d, void* blob_out_d, uint64_t cap_blob, uint64_t* blob_len_out) {
    if (!ctx || (n && (!compact_d || !claims_out_d || !groups)) || (cblob_len && !cblob_d)) return IPCFP_E_INVALID;
    if (n >= 0xffffffffULL || n_groups > IPCFP_COMPACT_MAX_GROUPS) return set_error(ctx, IPCFP_E_UNSUPPORTED, "batch too large");
    if (cblob_len + 8 * n >= 0xf0000000ull) return set_error(ctx, IPCFP_E_UNSUPPORTED, "claim blob of %llu bytes (offsets are u32)", (unsigned long long)cblob_len);
    if (blob_len_out) *blob_len_out = 0;
    if (n == 0) return IPCFP_OK;
    IPCFP_ENTER(ctx);
    DevBuf<ipcfp_event_claim_group_t> gd;
    DevBuf<uint32_t> su;
    DevBuf<uint64_t> ss;
    IPCFP_HIP(ctx, gd.alloc(n_groups ? n_groups : 1));
    IPCFP_HIP(ctx, su.alloc(4 * size_t(n)));
    IPCFP_HIP(ctx, ss.alloc(size_t(div_up(uint32_t(n), 1024)) + 2));
    if (n_groups) IPCFP_HIP(ctx, hipMemcpyAsync(gd.p, groups, n_groups * sizeof *groups, hipMemcpyHostToDevice, ctx->stream));
    int rc = launch_expand_claims(ctx, compact_d, uint32_t(n), gd.p, n_groups, static_cast<const uint8_t*>(cblob_d), cblob_len, claims_out_d,
                                  static_cast<uint8_t*>(blob_out_d), cap_blob, su.p, ss.p);
    if (rc) {
        (void)hipStreamSynchronize(ctx->stream);
        return rc;
    }
    uint64_t total = 0;
    IPCFP_HIP(ctx, d2h_small(ctx, &total, ss.p + div_up(uint32_t(n), 1024) + 1, sizeof total, ctx->stream));
    IPCFP_HIP(ctx, sync_stream(ctx, ctx->stream));  // (the scratch goes back to the pool on return)
    if (blob_len_out) *blob_len_out = total;
    return IPCFP_OK;
}

int ipcfp_verify_event_claims_compact(ipcfp_ctx_t* ctx, ipcfp_witness_t* w, const ipcfp_tipset_ref_t* tipsets,
                                      uint32_t n_tipsets, const ipcfp_event_claim_group_t* groups, uint32_t n_groups,
                                      const ipcfp_event_claim_compact_t* claims, uint64_t n, const uint8_t* cblob,
                                      uint64_t cblob_len, const ipcfp_trust_policy_t* trust,
                                      const ipcfp_event_filter_t* filter, ipcfp_status_t* status) {
    if (!ctx || !w || w->ctx != ctx || (n && (!claims || !status || !tipsets || !groups)) || (cblob_len && !cblob)) return IPCFP_E_INVALID;
    if (n >= 0xffffffffULL || n_groups > IPCFP_COMPACT_MAX_GROUPS) return set_error(ctx, IPCFP_E_UNSUPPORTED, "batch too large");
    if (cblob_len + 8 * n >= 0xf0000000ull) return set_error(ctx, IPCFP_E_UNSUPPORTED, "claim blob of %llu bytes (offsets are u32)", (unsigned long long)cblob_len);
    if (n == 0) return IPCFP_OK;
    IPCFP_ENTER(ctx);
    std::vector<TipsetCtxDev> tcs(n_tipsets);
    WideParents wide;
    for (uint32_t k = 0; k < n_tipsets; ++k)
        if (int rc_t = tipset_inputs(ctx, tipsets[k], tcs[k], wide)) return rc_t;
    const uint64_t cap_blob = cblob_len + 8 * n + 64;
    DevBuf<uint8_t> cc, cb, bd, sd;
    DevBuf<EventClaimPacked> cd;
    DevBuf<ipcfp_event_claim_group_t> gd;
    DevBuf<uint32_t> su;
    DevBuf<uint64_t> ss;
    IPCFP_HIP(ctx, cc.alloc(n * sizeof(ipcfp_event_claim_compact_t)));
    IPCFP_HIP(ctx, cb.alloc(cblob_len + 64));
    IPCFP_HIP(ctx, cd.alloc(n));
    IPCFP_HIP(ctx, bd.alloc(cap_blob));
    IPCFP_HIP(ctx, sd.alloc(n));
    IPCFP_HIP(ctx, gd.alloc(n_groups ? n_groups : 1));
    IPCFP_HIP(ctx, su.alloc(4 * size_t(n)));
    IPCFP_HIP(ctx, ss.alloc(size_t(div_up(uint32_t(n), 1024)) + 2));
    // The records cross PCIe on a thread of their own while this one queues the tipset prologue, the AMT walk and the
    // execution order (ipcfp_verify_event_claims); whoever queues the verify kernel calls claims_ready, which joins the
    // copy and queues the expansion in front of it.
    int rc = IPCFP_OK;
    IPCFP_HIP(ctx, hipStreamSynchronize(ctx->stream));  // (nothing queued earlier may still use the buffers just taken)
    if (n_groups) IPCFP_HIP(ctx, hipMemcpyAsync(gd.p, groups, n_groups * sizeof *groups, hipMemcpyHostToDevice, ctx->stream));
    if (n * sizeof(ipcfp_event_claim_compact_t) >= (size_t(8) << 20))
        ctx->upload_task = upload_task_start(ctx, cc.p, claims, n * sizeof(ipcfp_event_claim_compact_t), cb.p, cblob, cblob_len);
    if (!ctx->upload_task) {
        rc = upload(ctx, cc.p, claims, n * sizeof(ipcfp_event_claim_compact_t), ctx->stream);
        if (!rc && cblob_len) rc = upload(ctx, cb.p, cblob, cblob_len, ctx->stream);
        if (rc) return rc;
    }
    ipcfp_ctx::ClaimsExpand& x = ctx->claims_expand;
    x.pending = true;
    x.compact_d = cc.p;
    x.groups_d = gd.p;
    x.n = uint32_t(n);
    x.n_groups = n_groups;
    x.cblob_d = cb.p;
    x.cblob_len = cblob_len;
    x.cap_blob = cap_blob;
    x.claims_out_d = cd.p;
    x.blob_out_d = bd.p;
    x.scratch_u32 = su.p;
    x.scan_scratch = ss.p;
    rc = verify_packed(ctx, w, tcs, cd.p, uint32_t(n), bd.p, cap_blob, trust, filter, sd.p);
    {
        const int rc_up = upload_task_wait(ctx);  // (whatever happened: the copy must be over before the buffers go back to the pool)
        if (rc == IPCFP_OK) rc = rc_up;
        if (x.pending) {  // no route reached its verify kernel: nothing was expanded, nothing may be reported
            x.pending = false;
            if (rc == IPCFP_OK) rc = set_error(ctx, IPCFP_E_INVALID, "compact claims were never expanded (no route reached its verify kernel)");
        }
    }
    if (rc) {
        (void)hipStreamSynchronize(ctx->stream);
        return rc;
    }
    IPCFP_HIP(ctx, hipMemcpyAsync(status, sd.p, n, hipMemcpyDeviceToHost, ctx->stream));
    IPCFP_HIP(ctx, sync_stream(ctx, ctx->stream));
    return IPCFP_OK;
}

}  // extern "C"
