// csrc/kernels/verify_table.hip — `verify_single_proof` (src/proofs/events/verifier.rs:92-290) for the claims whose
// receipt was enumerated and whose events are tabulated (event_table.h): steps 1-3 are table lookups (execution-order
// hash table), step 4 compares the claim with bytes at addresses the records name.  No block is parsed here, so the
// kernel needs neither the CBOR reader nor LDS and runs at full occupancy; a claim the table does not cover is
// marked kStPending and taken by the general walker (k_verify_events) right behind.
#include <hip/hip_runtime.h>

#include "../common.h"
#include "launch.h"
#include "verify_dev.h"

namespace ipcfp {

typedef unsigned long long u64x2_t __attribute__((ext_vector_type(2)));

__device__ __forceinline__ u64x2_t ld16(const uint8_t* __restrict__ p) {
    u64x2_t v;
    __builtin_memcpy(&v, p, 16);  // (any alignment: gfx950 runs in unaligned-access mode)
    return v;
}
__device__ __forceinline__ bool differ32(const u64x2_t a[2], const u64x2_t b[2]) {
    return (((a[0].x ^ b[0].x) | (a[0].y ^ b[0].y)) | ((a[1].x ^ b[1].x) | (a[1].y ^ b[1].y))) != 0;
}

// The claimed bytes of the usual shape — at most two topics, 32 bytes of data — are fetched from the blob right after the
// claim itself, beside the record loads, instead of piece by piece between the (random, slow) reads of the event: a line
// of the blob fetched once per wavefront instead of once per piece (every access costs a whole 128-byte line:
// profiles/r03_fetch_calib.txt), and the chain of dependent loads of a claim is four long instead of eight.
struct ClaimedBytes {
    u64x2_t topic[2][2];
    uint32_t flag[2];  // the matchable byte in front of each topic
    u64x2_t data[2];
};

__device__ __forceinline__ void load_claimed(const EventClaimPacked& c, const uint8_t* __restrict__ blob, ClaimedBytes& p) {
    p = ClaimedBytes{};
    if (c.n_topics >= 1 && c.n_topics <= 2) {
        const uint8_t* q = blob + c.topics_off;
        p.flag[0] = q[0];
        p.topic[0][0] = ld16(q + 1);
        p.topic[0][1] = ld16(q + 17);
        if (c.n_topics == 2) {
            p.flag[1] = q[33];
            p.topic[1][0] = ld16(q + 34);
            p.topic[1][1] = ld16(q + 50);
        }
    }
    if ((c.flags & EC_DATA_MATCHABLE) && c.data_len == 32) {
        p.data[0] = ld16(blob + c.data_off);
        p.data[1] = ld16(blob + c.data_off + 16);
    }
}

// verify_event_record (verify_dev.h) with the claimed bytes in registers and the event's bytes fetched in one batch; the
// verdicts are taken in the reference's order (events/verifier.rs:257-290, then :247-251).  Other shapes: the general form.
__device__ __forceinline__ uint32_t verify_event_record_batched(const WitnessView& w, const EventClaimPacked& c, const EventRec& e,
                                                                const ClaimedBytes& p, const uint8_t* __restrict__ blob,
                                                                const ipcfp_event_filter_t& filter, bool has_filter) {
    if (e.emitter != c.emitter) return IPCFP_ST_FALSE_EMITTER;                                                // :262
    if (!(e.base_flags & kEvIsLog)) return IPCFP_ST_FALSE_NOT_EVM_LOG;                                        // :267
    const uint32_t nt = uint32_t(e.base_flags >> kEvTopicShift) & 0xffu;
    if (nt != c.n_topics) return IPCFP_ST_FALSE_TOPIC_COUNT;                                                  // :272
    if (nt > 2) return verify_event_record(w, c, e, blob, filter, has_filter);
    const uint8_t* item = w.arena + (e.base_flags & kEvBaseMask);
    const bool case_a = (e.base_flags & kEvCaseA) != 0;
    const uint32_t rel0 = e.topic_rel[0], rel1 = case_a ? uint32_t(e.topic_rel[0]) + 32u : uint32_t(e.topic_rel[1]);
    const bool data_shape = (c.flags & EC_DATA_MATCHABLE) && c.data_len == e.data_len;
    const bool data_batched = data_shape && c.data_len == 32;
    u64x2_t t0[2] = {}, t1[2] = {}, d[2] = {};
    if (nt >= 1) {
        t0[0] = ld16(item + rel0);
        t0[1] = ld16(item + rel0 + 16);
    }
    if (nt == 2) {
        t1[0] = ld16(item + rel1);
        t1[1] = ld16(item + rel1 + 16);
    }
    if (data_batched) {
        d[0] = ld16(item + e.data_rel);
        d[1] = ld16(item + e.data_rel + 16);
    }
    if (nt >= 1 && (!p.flag[0] || differ32(p.topic[0], t0))) return IPCFP_ST_FALSE_TOPIC;                     // :276-281
    if (nt == 2 && (!p.flag[1] || differ32(p.topic[1], t1))) return IPCFP_ST_FALSE_TOPIC;
    if (!data_shape) return IPCFP_ST_FALSE_DATA;                                                              // :284-287
    if (data_batched ? differ32(p.data, d) : !bytes_equal_global(item + e.data_rel, blob + c.data_off, c.data_len))
        return IPCFP_ST_FALSE_DATA;
    if (has_filter) {                                                                                         // :247-251
        if (nt < 2) return IPCFP_ST_FALSE_FILTER;
        u64x2_t f0[2], f1[2];
        __builtin_memcpy(f0, filter.topic0, 32);
        __builtin_memcpy(f1, filter.topic1, 32);
        if (differ32(t0, f0) || differ32(t1, f1)) return IPCFP_ST_FALSE_FILTER;
    }
    return IPCFP_ST_TRUE;
}

__device__ __forceinline__ uint32_t verify_table_one(const WitnessView& w, const EventClaimPacked& c, const TipsetCtxDev& tc,
                                                     const uint8_t* __restrict__ blob, const ipcfp_trust_policy_t& trust,
                                                     const ipcfp_event_filter_t& filter, bool has_filter, ValueLoc* where) {
    // everything that depends on the claim alone is in flight before the first verdict is taken
    ClaimedBytes p;
    load_claimed(c, blob, p);
    const bool tabulated = tc.receipt_leaves && tc.receipt_recs && c.exec_index >= tc.receipt_first &&
                           c.exec_index - tc.receipt_first < tc.n_receipt_leaves;
    ReceiptRec rr{RK_WALK, 0, 0, kNoBlock, 0};
    if (tabulated) rr = tc.receipt_recs[c.exec_index - tc.receipt_first];
    const uint32_t st = verify_event_prefix(c, tc, trust);
    if (st != IPCFP_ST_TRUE) return st;
    // verify_event_from_table (verify_dev.h), same rules in the same order
    if (!tabulated || rr.kind == RK_WALK) return kStPending;
    if (rr.kind == RK_NO_EVENTS) return IPCFP_ST_FALSE_NO_EVENTS_ROOT;                                        // :229
    if (rr.kind >= 64) return rr.kind;                                                                        // :234 Err
    if (c.event_index == ~0ULL) return IPCFP_ST_ERR;                                                          // > MAX_INDEX
    if (c.event_index >= 64 || !((rr.bitmap >> c.event_index) & 1ull)) return IPCFP_ST_FALSE_NO_EVENT;        // :237
    const EventRec e = tc.event_recs[rr.first + __popcll(rr.bitmap & ((1ull << c.event_index) - 1ull))];
    if (where) *where = ValueLoc{rr.block, uint32_t((e.base_flags & kEvBaseMask) - w.off[rr.block]), e.ev_len};
    return verify_event_record_batched(w, c, e, p, blob, filter, has_filter);
}

__global__ __launch_bounds__(256) void k_verify_events_table(WitnessView w, const EventClaimPacked* __restrict__ claims, uint32_t n,
                                                             const TipsetCtxDev* __restrict__ ctxs, uint32_t n_ctxs,
                                                             const uint8_t* __restrict__ blob, uint64_t blob_len,
                                                             ipcfp_trust_policy_t trust, ipcfp_event_filter_t filter,
                                                             int has_filter, uint8_t* __restrict__ status,
                                                             ValueLoc* __restrict__ where) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = t < n;
    EventClaimPacked c;
    if (live) c = claims[t];
    else c.context = 0;
    ValueLoc loc{kNoBlock, 0, 0};
    uint32_t st = IPCFP_ST_ERR_BAD_CLAIM;
    const bool inb = live && claim_in_bounds(c, n_ctxs, blob_len);
    // The proofs of a wavefront nearly always name ONE tipset pair: its context (≈700 bytes of header facts and
    // table pointers) is then read through the scalar unit once per wavefront instead of by 64 lanes apiece.
    const uint32_t ctx0 = __builtin_amdgcn_readfirstlane(inb ? c.context : 0xffffffffu);
    if (__all(!inb || c.context == ctx0)) {
        if (inb) st = verify_table_one(w, c, ctxs[ctx0], blob, trust, filter, has_filter != 0, where ? &loc : nullptr);
    } else if (inb) {
        st = verify_table_one(w, c, ctxs[c.context], blob, trust, filter, has_filter != 0, where ? &loc : nullptr);
    }
    if (!live) return;
    status[t] = uint8_t(st);
    if (where) where[t] = loc;
}

int launch_verify_events_table(ipcfp_ctx* ctx, const WitnessView& w, const EventClaimPacked* claims_d, uint32_t n,
                               const TipsetCtxDev* ctxs_d, uint32_t n_ctxs, const uint8_t* blob_d, uint64_t blob_len,
                               const ipcfp_trust_policy_t& trust, const ipcfp_event_filter_t& filter, int has_filter,
                               uint8_t* status_d, void* where_d) {
    if (n == 0) return IPCFP_OK;
    hipLaunchKernelGGL(k_verify_events_table, dim3(div_up(n, 256)), dim3(256), 0, ctx->stream, w, claims_d, n, ctxs_d, n_ctxs,
                       blob_d, blob_len, trust, filter, has_filter, status_d, static_cast<ValueLoc*>(where_d));
    IPCFP_HIP(ctx, hipGetLastError());
    return IPCFP_OK;
}

}  // namespace ipcfp
