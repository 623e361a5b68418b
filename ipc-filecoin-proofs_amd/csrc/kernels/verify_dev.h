// csrc/kernels/verify_dev.h — the checks of `verify_single_proof` (src/proofs/events/verifier.rs:92-290) that every
// device route shares: steps 1-3 (trust anchors, header consistency, execution order) and step 4 on a tabulated
// event (event_table.h).  Used by k_verify_events_table (verify_table.hip: parses nothing) and by k_verify_events
// (verify_events.hip: the general walker).
#pragma once
#include "claims_dev.h"
#include "event_table.h"
#include "amt_enum.h"
#include "event_table.h"
#include "tipset_ctx.h"
#include "types_dev.h"

namespace ipcfp {

// internal status byte: "not settled from the event table — the general walker decides" (never leaves the library)

constexpr unsigned long long kEmptySlot64 = ~0ULL;

__device__ __forceinline__ uint32_t exec_find(const unsigned long long* slots, uint32_t mask, const CidKey* keys,
                                              const CidKey& key) {
    const uint64_t h = cid_hash64(key);
    uint32_t s = uint32_t(h >> 32) & mask;
    for (;;) {
        const unsigned long long cur = slots[s];
        if (cur == kEmptySlot64) return kNoBlock;
        if (uint32_t(cur >> 32) == uint32_t(h) && cid_equal(keys[uint32_t(cur)], key)) return uint32_t(cur);
        s = (s + 1) & mask;
    }
}

__device__ __forceinline__ bool ev_trusted(const ipcfp_trust_policy_t& t, long long epoch) {
    if (t.kind == 0) return true;
    if (t.ec_chain_empty) return false;
    return epoch >= t.min_epoch && epoch <= t.max_epoch;
}

// verify_event_data_matches (+ the built-in check_event) on a tabulated event: bytes at known addresses
__device__ __forceinline__ bool bytes_equal_global(const uint8_t* __restrict__ a, const uint8_t* __restrict__ b, uint32_t n) {
    uint64_t diff = 0;
    uint32_t i = 0;
    // 16 bytes per load instruction (any alignment: gfx950 runs in unaligned-access mode): every lane compares at
    // addresses of its own, so what these loops cost is load INSTRUCTIONS, not bytes
    typedef unsigned long long u64x2_t __attribute__((ext_vector_type(2)));
    for (; i + 16 <= n; i += 16) {
        u64x2_t x, y;
        __builtin_memcpy(&x, a + i, 16);
        __builtin_memcpy(&y, b + i, 16);
        diff |= (x.x ^ y.x) | (x.y ^ y.y);
    }
    for (; i + 8 <= n; i += 8) {
        uint64_t x, y;
        __builtin_memcpy(&x, a + i, 8);
        __builtin_memcpy(&y, b + i, 8);
        diff |= x ^ y;
    }
    for (; i < n; ++i) diff |= uint64_t(a[i] ^ b[i]);
    return diff == 0;
}

__device__ __forceinline__ uint32_t verify_event_record(const WitnessView& w, const EventClaimPacked& c, const EventRec& e,
                                                        const uint8_t* __restrict__ blob, const ipcfp_event_filter_t& filter,
                                                        bool has_filter) {
    if (e.emitter != c.emitter) return IPCFP_ST_FALSE_EMITTER;                                                // :262
    if (!(e.base_flags & kEvIsLog)) return IPCFP_ST_FALSE_NOT_EVM_LOG;                                        // :267
    const uint32_t nt = uint32_t(e.base_flags >> kEvTopicShift) & 0xffu;
    if (nt != c.n_topics) return IPCFP_ST_FALSE_TOPIC_COUNT;                                                  // :272
    const uint8_t* item = w.arena + (e.base_flags & kEvBaseMask);
    const bool case_a = (e.base_flags & kEvCaseA) != 0;
    for (uint32_t i = 0; i < nt; ++i) {                                                                       // :276-281
        const uint8_t* claimed = blob + c.topics_off + 33u * i;
        if (!claimed[0]) return IPCFP_ST_FALSE_TOPIC;  // the claimed string is not "0x" + 64 hex digits
        const uint32_t rel = case_a ? uint32_t(e.topic_rel[0]) + 32u * i : uint32_t(e.topic_rel[i & 3u]);
        if (!bytes_equal_global(item + rel, claimed + 1, 32)) return IPCFP_ST_FALSE_TOPIC;
    }
    if (!(c.flags & EC_DATA_MATCHABLE) || c.data_len != e.data_len) return IPCFP_ST_FALSE_DATA;               // :284-287
    if (!bytes_equal_global(item + e.data_rel, blob + c.data_off, c.data_len)) return IPCFP_ST_FALSE_DATA;
    if (has_filter) {                                                                                         // :247-251
        if (nt < 2) return IPCFP_ST_FALSE_FILTER;
        const uint32_t r1 = case_a ? uint32_t(e.topic_rel[0]) + 32u : uint32_t(e.topic_rel[1]);
        if (!bytes_equal_global(item + e.topic_rel[0], filter.topic0, 32) || !bytes_equal_global(item + r1, filter.topic1, 32))
            return IPCFP_ST_FALSE_FILTER;
    }
    return IPCFP_ST_TRUE;
}

// Steps 1-3 of verify_single_proof: TRUE when the proof may go on to its receipt.
__device__ __forceinline__ uint32_t verify_event_prefix(const EventClaimPacked& c, const TipsetCtxDev& tc,
                                                        const ipcfp_trust_policy_t& trust) {
    // Step 1: verify_trust_anchors (events/verifier.rs:124-144)
    if (!(tc.flags & TC_PARENTS_PARSED) || !(tc.flags & TC_CHILD_PARSED)) return IPCFP_ST_ERR_BAD_CLAIM;  // :130-131
    if (!ev_trusted(trust, c.parent_epoch)) return IPCFP_ST_FALSE_UNTRUSTED_PARENT;                          // :134
    if (!ev_trusted(trust, c.child_epoch)) return IPCFP_ST_FALSE_UNTRUSTED_CHILD;                            // :139
    // Step 2: verify_header_consistency (:147-181)
    if (tc.child_status != IPCFP_ST_TRUE) return tc.child_status;                                             // :155-158
    if (!tc.parents_match) return IPCFP_ST_FALSE_PARENTS_MISMATCH;                                            // :161
    if (tc.child_height != c.child_epoch) return IPCFP_ST_FALSE_CHILD_EPOCH;                                  // :166
    if (tc.n_parents == 0) return IPCFP_ST_ERR_EMPTY_PARENTS;                                                 // :172 (panic)
    if (tc.parent0_status != IPCFP_ST_TRUE) return tc.parent0_status;                                         // :171-174
    if (tc.parent0_height != c.parent_epoch) return IPCFP_ST_FALSE_PARENT_EPOCH;                              // :176
    // Step 3: verify_execution_order (:184-204)
    if (tc.exec_status != IPCFP_ST_TRUE) return tc.exec_status;                                               // :190
    if (!(c.flags & EC_MSG_PARSED)) return IPCFP_ST_ERR_BAD_CLAIM;                                            // :193
    // `exec_order.iter().position(|c| c == msg)` == exec_index: the order holds every message once, so the claim is
    // right iff the message AT exec_index is the claimed one — two reads that run along with the claims (exec_inv).
    // Only a claim that fails this asks the hash table whether the message is somewhere else (:199) or nowhere (:194).
    if (tc.exec_inv && c.exec_index < tc.exec_len && cid_equal(tc.exec_keys[tc.exec_inv[c.exec_index]], c.message))
        return IPCFP_ST_TRUE;
    const uint32_t raw = tc.exec_slots ? exec_find(tc.exec_slots, tc.exec_mask, tc.exec_keys, c.message) : kNoBlock;
    if (raw == kNoBlock) return IPCFP_ST_FALSE_MSG_NOT_IN_EXEC;                                               // :194
    if (uint64_t(tc.exec_pos[raw]) != c.exec_index) return IPCFP_ST_FALSE_EXEC_INDEX;                         // :199
    return IPCFP_ST_TRUE;
}

// Step 4 (verify_receipt_and_event, :207-254) when the receipt was enumerated and its events tabulated: what
// `Amt::load(events_root)`, `get(event_index)`, extract_evm_log and the compares observe is all in the records.
// `settled` = false: the table does not cover this claim (receipt outside the enumeration, RK_WALK) — nothing decided.
// `where` (nullable) receives the location of the StampedEvent once the proof has reached it.
__device__ __forceinline__ uint32_t verify_event_from_table(const WitnessView& w, const EventClaimPacked& c,
                                                            const TipsetCtxDev& tc, const uint8_t* __restrict__ blob,
                                                            const ipcfp_event_filter_t& filter, bool has_filter,
                                                            ValueLoc* where, bool& settled) {
    settled = false;
    if (!tc.receipt_leaves || !tc.receipt_recs || c.exec_index < tc.receipt_first ||
        c.exec_index - tc.receipt_first >= tc.n_receipt_leaves)
        return kStPending;
    const ReceiptRec rr = tc.receipt_recs[c.exec_index - tc.receipt_first];
    if (rr.kind == RK_WALK) return kStPending;
    settled = true;
    if (rr.kind == RK_NO_EVENTS) return IPCFP_ST_FALSE_NO_EVENTS_ROOT;                                        // :229
    if (rr.kind >= 64) return rr.kind;                                                                        // :234 Err
    if (c.event_index == ~0ULL) return IPCFP_ST_ERR;                                                          // > MAX_INDEX
    if (c.event_index >= 64 || !((rr.bitmap >> c.event_index) & 1ull)) return IPCFP_ST_FALSE_NO_EVENT;        // :237
    const EventRec e = tc.event_recs[rr.first + __popcll(rr.bitmap & ((1ull << c.event_index) - 1ull))];
    if (where) *where = ValueLoc{rr.block, uint32_t((e.base_flags & kEvBaseMask) - w.off[rr.block]), e.ev_len};
    return verify_event_record(w, c, e, blob, filter, has_filter);
}

// A packed claim that points outside the tipset table or the blob (only a caller of the packed entry points can
// build one; the string lowering cannot) is answered with ERR_BAD_CLAIM instead of being followed.
__device__ __forceinline__ bool claim_in_bounds(const EventClaimPacked& c, uint32_t n_ctxs, uint64_t blob_len) {
    return c.context < n_ctxs && c.n_topics <= (1u << 20) && uint64_t(c.topics_off) + 33ull * c.n_topics <= blob_len &&
           uint64_t(c.data_off) + c.data_len <= blob_len;
}

}  // namespace ipcfp
