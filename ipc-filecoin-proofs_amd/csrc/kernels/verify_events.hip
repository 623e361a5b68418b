// csrc/kernels/verify_events.hip — batch `verify_event_proof`: tipset-context preparation,
// execution-order reconstruction, and one proof per lane.
//
// Replaces src/proofs/events/verifier.rs:51-290 and src/proofs/events/utils.rs:16-30,48-94.
// Check order and every Ok(false)/Err outcome follow SURVEY.md A.10; the status byte names the
// reference line that decided.
#define IPCFP_LINE_STAGE 1  // k_verify_events parses whole blocks front to back: see cbor_dev.h
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "../common.h"
#include "blake2b_dev.h"
#include "claims_dev.h"
#include "event_log_dev.h"
#include "walk_dev.h"
#include "amt_enum.h"
#include "event_table.h"
#include "tipset_ctx.h"
#include "types_dev.h"
#include "launch.h"
#include "scan_dev.h"
#include "verify_dev.h"

namespace ipcfp {

// stage 2: leaf values (tag-42 links, already validated) → message CID keys
__global__ __launch_bounds__(256) void k_exec_keys(WitnessView w, const LeafRef* __restrict__ leaves, uint32_t n,
                                                   CidKey* __restrict__ keys) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const LeafRef l = leaves[t];
    Rd r;
    r.init(w.arena + w.off[l.block] + l.off, l.len);
    CidKey k;
    r.read_link_key(k);
    keys[t] = k;
}

// stage 3: `seen.insert(c)` — the table keeps, per distinct CID, the SMALLEST raw position.
// A slot is one u64: fingerprint (low half of the key's 64-bit hash) in the high word, raw position in the low
// word.  A probe compares fingerprints first and reads the 40-byte key behind a slot only when they agree — at load
// 0.5 half of all inserts pass an occupied slot, and each of those used to be a random 40-byte read.

__global__ __launch_bounds__(256) void k_exec_insert(const CidKey* __restrict__ keys, uint32_t n,
                                                     unsigned long long* __restrict__ slots, uint32_t mask) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const CidKey key = keys[i];
    const uint64_t h = cid_hash64(key);
    const unsigned long long mine = ((unsigned long long)uint32_t(h) << 32) | i;
    uint32_t s = uint32_t(h >> 32) & mask;
    for (;;) {
        unsigned long long cur = slots[s];
        if (cur == kEmptySlot64) {
            cur = atomicCAS(&slots[s], kEmptySlot64, mine);
            if (cur == kEmptySlot64) return;
        }
        if (uint32_t(cur >> 32) == uint32_t(h) && cid_equal(keys[uint32_t(cur)], key)) {
            atomicMin(&slots[s], mine);  // same fingerprint: the minimum is the smaller position
            return;
        }
        s = (s + 1) & mask;
    }
}

// stages 3+4 in ONE probe per position (host/verify_fast.cpp): `first` (zeroed by the caller) is kept right while the
// table fills.  A position that takes an empty slot counts itself; one that lowers a slot's minimum counts itself and
// un-counts the position it displaced.  Additions commute, so whatever order the wavefronts run in, the holder of each
// slot's final minimum ends at 1 and every other position of that CID at 0 — the second random probe of every position
// (k_exec_first: another 128-byte line apiece) becomes a streaming read of the flags.
// (CAS_FIRST as in k_index_insert: the probe is the compare-and-swap itself)
template <bool CAS_FIRST>
__global__ __launch_bounds__(256) void k_exec_insert_flags(const CidKey* __restrict__ keys, uint32_t n,
                                                           unsigned long long* __restrict__ slots, uint32_t mask,
                                                           uint32_t* __restrict__ first) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const CidKey key = keys[i];
    const uint64_t h = cid_hash64(key);
    const unsigned long long mine = ((unsigned long long)uint32_t(h) << 32) | i;
    uint32_t s = uint32_t(h >> 32) & mask;
    for (;;) {
        unsigned long long cur = CAS_FIRST ? kEmptySlot64 : slots[s];
        if (cur == kEmptySlot64) {
            cur = atomicCAS(&slots[s], kEmptySlot64, mine);
            if (cur == kEmptySlot64) {
                atomicAdd(&first[i], 1u);
                return;
            }
        }
        if (uint32_t(cur >> 32) == uint32_t(h) && cid_equal(keys[uint32_t(cur)], key)) {
            const unsigned long long old = atomicMin(&slots[s], mine);  // (only positions of THIS CID ever lower this slot)
            if (old > mine) {
                atomicAdd(&first[i], 1u);
                atomicAdd(&first[uint32_t(old)], 0xffffffffu);
            }
            return;
        }
        s = (s + 1) & mask;
    }
}

// stage 4: first[i] = 1 iff position i is the first occurrence of its CID (`if seen.insert(*c) { out.push(*c) }`)
__global__ __launch_bounds__(256) void k_exec_first(const CidKey* __restrict__ keys, uint32_t n,
                                                    const unsigned long long* __restrict__ slots, uint32_t mask,
                                                    uint32_t* __restrict__ first) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const CidKey key = keys[i];
    const uint64_t h = cid_hash64(key);
    uint32_t s = uint32_t(h >> 32) & mask;
    uint32_t f = 0;
    for (;;) {
        const unsigned long long cur = slots[s];
        if (cur == kEmptySlot64) break;  // cannot happen after stage 3; kept as a stop
        if (uint32_t(cur >> 32) == uint32_t(h)) {
            if (uint32_t(cur) == i) {  // the slot is this position's own: no key to read
                f = 1;
                break;
            }
            if (cid_equal(keys[uint32_t(cur)], key)) break;  // an earlier position holds the same CID
        }
        s = (s + 1) & mask;
    }
    first[i] = f;
}

__device__ __forceinline__ void ctx_finish_fields(TipsetCtxDev* __restrict__ c, const CtxFinish& a) {
    c->exec_status = IPCFP_ST_ERR_BAD_CLAIM;
    c->exec_slots = nullptr;
    c->exec_inv = nullptr;
    c->receipt_leaves = nullptr;
    c->n_receipt_leaves = 0;
    c->receipt_first = 0;
    c->receipt_recs = nullptr;
    c->event_recs = nullptr;
    // the execution order is only reached when steps 1-2 can pass for some proof of this context
    const bool reachable = (c->flags & TC_PARENTS_PARSED) && (c->flags & TC_CHILD_PARSED) && c->child_status == IPCFP_ST_TRUE &&
                           c->parents_match && c->n_parents > 0 && c->parent0_status == IPCFP_ST_TRUE;
    if (!reachable) return;
    const unsigned long long e = *a.err;
    const uint32_t status = e == kNoEnumError ? uint32_t(IPCFP_ST_TRUE) : enum_error_code(e);
    c->exec_status = status;
    c->exec_mask = a.mask;
    c->exec_slots = a.slots;
    c->exec_keys = a.keys;
    c->exec_pos = a.pos;
    c->exec_inv = status == IPCFP_ST_TRUE ? a.inv : nullptr;
    c->exec_len = status == IPCFP_ST_TRUE ? *a.total : 0;
    // the receipts AMT rode along with the message AMTs: a table lookup per claim
    c->receipt_leaves = a.receipt_leaves;
    c->n_receipt_leaves = a.n_receipt_leaves;
    c->receipt_first = a.n_receipt_leaves ? a.receipt_first : 0;
    c->receipt_recs = a.receipt_recs;
    c->event_recs = a.event_recs;
}

// stages 4-6 in two kernels instead of five (host/verify_fast.cpp): first-occurrence flags with their tile sums, then —
// behind the scan of the 256-item tiles' sums (scan.hip k_scan_tiles_u64) — positions, the inverse permutation and the
// context's tail (k_ctx_finish's part) in one pass
__global__ __launch_bounds__(256) void k_exec_first_sums(const CidKey* __restrict__ keys, uint32_t n,
                                                         const unsigned long long* __restrict__ slots, uint32_t mask,
                                                         uint32_t* __restrict__ first, uint64_t* __restrict__ tile_sums) {
    __shared__ uint64_t smem[17];
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t f = 0;
    if (i < n) {
        const CidKey key = keys[i];
        const uint64_t h = cid_hash64(key);
        uint32_t s = uint32_t(h >> 32) & mask;
        for (;;) {
            const unsigned long long cur = slots[s];
            if (cur == kEmptySlot64) break;  // cannot happen after stage 3; kept as a stop
            if (uint32_t(cur >> 32) == uint32_t(h)) {
                if (uint32_t(cur) == i) {  // the slot is this position's own: no key to read
                    f = 1;
                    break;
                }
                if (cid_equal(keys[uint32_t(cur)], key)) break;  // an earlier position holds the same CID
            }
            s = (s + 1) & mask;
        }
        first[i] = f;
    }
    uint64_t total;
    (void)block_exclusive_scan(uint64_t(f), smem, &total);
    if (threadIdx.x == 0) tile_sums[blockIdx.x] = total;
}

// the tile sums of flags that are already there (k_exec_insert_flags)
__global__ __launch_bounds__(256) void k_exec_flag_sums(const uint32_t* __restrict__ first, uint32_t n,
                                                        uint64_t* __restrict__ tile_sums) {
    __shared__ uint64_t smem[17];
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t total;
    (void)block_exclusive_scan(uint64_t(i < n ? first[i] : 0u), smem, &total);
    if (threadIdx.x == 0) tile_sums[blockIdx.x] = total;
}

__global__ __launch_bounds__(256) void k_exec_apply_finish(const uint32_t* __restrict__ first, uint32_t n,
                                                           const uint64_t* __restrict__ tile_base, uint32_t* __restrict__ pos,
                                                           TipsetCtxDev* __restrict__ c, CtxFinish a) {
    __shared__ uint64_t smem[17];
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t f = i < n ? first[i] : 0u;
    uint64_t total;
    const uint64_t ex = block_exclusive_scan(uint64_t(f), smem, &total) + tile_base[blockIdx.x];
    if (i < n) {
        pos[i] = uint32_t(ex);
        if (f) a.inv[uint32_t(ex)] = i;
    }
    if (i == 0) ctx_finish_fields(c, a);
}

// the distinct CIDs in execution order (ipcfp_exec_order)
__global__ __launch_bounds__(256) void k_exec_compact(const CidKey* __restrict__ keys, uint32_t n,
                                                      const uint32_t* __restrict__ first,
                                                      const uint32_t* __restrict__ pos, CidKey* __restrict__ out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (first[i]) out[pos[i]] = keys[i];
}

// ---------------------------------------------------------------------------
// one proof per lane (the checks every route shares live in verify_dev.h)
// ---------------------------------------------------------------------------
// `where` (nullable) receives the location of the StampedEvent the claim names once the proof has reached it
// (block = 0xffffffff otherwise): a host `check_event` closure runs over those bytes (events/verifier.rs:247-251).
__device__ __forceinline__ uint32_t verify_event_one(const WitnessView& w, const EventClaimPacked& c,
                                                     const TipsetCtxDev& tc, const uint8_t* __restrict__ blob,
                                                     const ipcfp_trust_policy_t& trust, const ipcfp_event_filter_t& filter,
                                                     bool has_filter, ValueLoc* where) {
    // Steps 1-3: trust anchors, header consistency, execution order (verify_dev.h)
    const uint32_t pre = verify_event_prefix(c, tc, trust);
    if (pre != IPCFP_ST_TRUE) return pre;
    // Step 4: verify_receipt_and_event (:207-254)
    uint32_t st;
    ValueLoc rloc;
    if (tc.receipt_leaves && c.exec_index >= tc.receipt_first && c.exec_index - tc.receipt_first < tc.n_receipt_leaves) {
        // the receipts AMT was enumerated (and thereby fully validated) for this context: load + get
        // of a present index cannot fail and yields exactly this leaf                                        // :220-224
        const uint64_t slot = c.exec_index - tc.receipt_first;
        if (tc.receipt_recs) {
            // ... and its events were tabulated (event_table.h): normally k_verify_events_table has settled the claim
            bool settled;
            const uint32_t ts = verify_event_from_table(w, c, tc, blob, filter, has_filter, where, settled);
            if (settled) return ts;
        }
        const LeafRef l = tc.receipt_leaves[slot];
        if (l.block == kNoBlock) return IPCFP_ST_ERR;  // (a leaf the enumeration could not produce: the host redoes the call)
        rloc = ValueLoc{l.block, l.off, l.len};
    } else {
        AmtRootInfo receipts;
        st = amt_load(w, tc.receipts_root, 0, VK_RECEIPT, receipts);                                          // :220
        if (st != IPCFP_ST_TRUE) return st;
        st = amt_get(w, receipts, VK_RECEIPT, c.exec_index, rloc);                                            // :224
        if (st == IPCFP_ST_NOT_FOUND) return IPCFP_ST_FALSE_NO_RECEIPT;
        if (st != IPCFP_ST_TRUE) return st;
    }
    Rd rr;
    rr.init(w.arena + w.off[rloc.block] + rloc.off, rloc.len);
    uint32_t o, l;
    rr.expect_array(4);
    (void)rr.read_uint();
    rr.read_bytes(o, l);
    (void)rr.read_uint();
    if (rr.at_null()) return IPCFP_ST_FALSE_NO_EVENTS_ROOT;                                                   // :229
    CidKey events_root;
    rr.read_link_key(events_root);
    ValueLoc eloc;
    st = amt_load_get(w, events_root, 3, VK_STAMPED_EVENT, c.event_index, eloc);                              // :234-237
    if (st == IPCFP_ST_NOT_FOUND) return IPCFP_ST_FALSE_NO_EVENT;
    if (st != IPCFP_ST_TRUE) return st;
    if (where) *where = eloc;
    // verify_event_data_matches (:257-290)
    Rd er;
    er.init(w.arena + w.off[eloc.block] + eloc.off, eloc.len);
    uint64_t emitter;
    EvmLogLoc log;
    decode_event_log(er, emitter, log);
    if (emitter != c.emitter) return IPCFP_ST_FALSE_EMITTER;                                                  // :262
    if (!log.is_log) return IPCFP_ST_FALSE_NOT_EVM_LOG;                                                       // :267
    if (log.n_topics != c.n_topics) return IPCFP_ST_FALSE_TOPIC_COUNT;                                        // :272
    for (uint32_t i = 0; i < log.n_topics; ++i) {                                                             // :276-281
        const uint8_t* claimed = blob + c.topics_off + 33u * i;
        if (!claimed[0]) return IPCFP_ST_FALSE_TOPIC;  // the claimed string is not "0x" + 64 hex digits
        if (!er.equal32(log.topic_at(i), claimed + 1)) return IPCFP_ST_FALSE_TOPIC;
    }
    if (!(c.flags & EC_DATA_MATCHABLE) || c.data_len != log.data.len) return IPCFP_ST_FALSE_DATA;             // :284-287
    if (!er.equal_bytes(log.data.off, blob + c.data_off, c.data_len)) return IPCFP_ST_FALSE_DATA;
    if (has_filter && !log_matches(er, log, filter)) return IPCFP_ST_FALSE_FILTER;                             // :247-251
    return IPCFP_ST_TRUE;
}

__global__ __launch_bounds__(256, IPCFP_WALK_WAVES) void k_verify_events(WitnessView w, const EventClaimPacked* __restrict__ claims,
                                                       uint32_t n, const TipsetCtxDev* __restrict__ ctxs, uint32_t n_ctxs,
                                                       const uint8_t* __restrict__ blob, uint64_t blob_len,
                                                       ipcfp_trust_policy_t trust,
                                                       ipcfp_event_filter_t filter, int has_filter,
                                                       uint8_t* __restrict__ status, ValueLoc* __restrict__ where,
                                                       int pending_only) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    if (pending_only && status[t] != kStPending) return;  // settled from the event table (verify_table.hip)
    const EventClaimPacked& c = claims[t];
    ValueLoc loc{kNoBlock, 0, 0};
    uint32_t st = IPCFP_ST_ERR_BAD_CLAIM;
    if (claim_in_bounds(c, n_ctxs, blob_len))
        st = verify_event_one(w, c, ctxs[c.context], blob, trust, filter, has_filter != 0, where ? &loc : nullptr);
    status[t] = uint8_t(st);
    if (where) where[t] = loc;
}

// exec_len of a context = the total of the first-occurrence scan, copied on the device so the host
// never waits for it
// ... and the inverse of exec_pos: inv[execution index] = raw position of the message's first occurrence.  A claim
// names (exec_index, message): `exec_keys[inv[exec_index]] == message` settles `position(message) == exec_index`
// with two reads that run along with the claims instead of a hash probe (three reads somewhere in 60 MB).
__global__ __launch_bounds__(256) void k_exec_finish(TipsetCtxDev* __restrict__ c, const uint64_t* __restrict__ total,
                                                     const uint32_t* __restrict__ first, const uint32_t* __restrict__ pos,
                                                     uint32_t n, uint32_t* __restrict__ inv) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) c->exec_len = *total;
    if (i < n && first[i]) inv[pos[i]] = i;
}

// The tail of a tipset context ON THE DEVICE (host/verify_fast.cpp): what host/verify_events.cpp::verify_packed does between
// its second synchronisation and the verify kernel — decide whether the execution order is reachable, point the context
// at the tables, copy exec_len, invert exec_pos — without the host having seen the header facts.  Same rules, same order.
__global__ __launch_bounds__(256) void k_ctx_finish(TipsetCtxDev* __restrict__ c, CtxFinish a) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < a.raw_len && a.first[i]) a.inv[a.pos[i]] = i;
    if (i == 0) ctx_finish_fields(c, a);
}

// first flags + tile sums, scan of the tile sums, positions + inverse + the context's tail.  tile_d: div_up(n, 256) + 1 words.
int launch_exec_finish_fused(ipcfp_ctx* ctx, TipsetCtxDev* ctx_d, const CtxFinish& a, uint32_t* first_d, uint32_t* pos_d,
                             uint64_t* tile_d, uint64_t* total_d, bool flags_ready) {
    const uint32_t n = a.raw_len, ntiles = div_up(n ? n : 1, 256);
    if (flags_ready) hipLaunchKernelGGL(k_exec_flag_sums, dim3(ntiles), dim3(256), 0, ctx->stream, first_d, n, tile_d);
    else hipLaunchKernelGGL(k_exec_first_sums, dim3(ntiles), dim3(256), 0, ctx->stream, a.keys, n, a.slots, a.mask, first_d, tile_d);
    int rc = launch_scan_tiles_u64(ctx, tile_d, ntiles, total_d);
    if (rc) return rc;
    hipLaunchKernelGGL(k_exec_apply_finish, dim3(ntiles), dim3(256), 0, ctx->stream, first_d, n, tile_d, pos_d, ctx_d, a);
    IPCFP_HIP(ctx, hipGetLastError());
    return IPCFP_OK;
}

int launch_ctx_finish(ipcfp_ctx* ctx, TipsetCtxDev* ctx_d, const CtxFinish& a) {
    hipLaunchKernelGGL(k_ctx_finish, dim3(a.raw_len ? div_up(a.raw_len, 256) : 1), dim3(256), 0, ctx->stream, ctx_d, a);
    IPCFP_HIP(ctx, hipGetLastError());
    return IPCFP_OK;
}

// ------------------------------ launchers -----------------------------------
int launch_exec_finish(ipcfp_ctx* ctx, TipsetCtxDev* ctx_d, const uint64_t* total_d, const uint32_t* first_d,
                       const uint32_t* pos_d, uint32_t n, uint32_t* inv_d) {
    hipLaunchKernelGGL(k_exec_finish, dim3(n ? div_up(n, 256) : 1), dim3(256), 0, ctx->stream, ctx_d, total_d, first_d, pos_d,
                       n, inv_d);
    IPCFP_HIP(ctx, hipGetLastError());
    return IPCFP_OK;
}

int launch_exec_dedup(ipcfp_ctx* ctx, const WitnessView& w, const LeafRef* leaves_d, uint32_t n, CidKey* keys_d,
                      unsigned long long* slots_d, uint32_t mask, uint32_t* first_d) {
    if (n == 0) return IPCFP_OK;
    const dim3 g(div_up(n, 256)), b(256);
    if (leaves_d) hipLaunchKernelGGL(k_exec_keys, g, b, 0, ctx->stream, w, leaves_d, n, keys_d);  // else: keys came with the enumeration
    hipLaunchKernelGGL(k_exec_insert, g, b, 0, ctx->stream, keys_d, n, slots_d, mask);
    hipLaunchKernelGGL(k_exec_first, g, b, 0, ctx->stream, keys_d, n, slots_d, mask, first_d);
    IPCFP_HIP(ctx, hipGetLastError());
    return IPCFP_OK;
}

int launch_exec_insert(ipcfp_ctx* ctx, const CidKey* keys_d, uint32_t n, unsigned long long* slots_d, uint32_t mask) {
    if (n == 0) return IPCFP_OK;
    hipLaunchKernelGGL(k_exec_insert, dim3(div_up(n, 256)), dim3(256), 0, ctx->stream, keys_d, n, slots_d, mask);
    IPCFP_HIP(ctx, hipGetLastError());
    return IPCFP_OK;
}

// first_d must be zero (hipMemsetAsync on the same stream) when this is queued
int launch_exec_insert_flags(ipcfp_ctx* ctx, const CidKey* keys_d, uint32_t n, unsigned long long* slots_d, uint32_t mask,
                             uint32_t* first_d) {
    if (n == 0) return IPCFP_OK;
    static const bool cas_first = [] {
        const char* e = std::getenv("IPCFP_INDEX_CAS_FIRST");
        return !(e && std::atoi(e) == 0);
    }();
    if (cas_first)
        hipLaunchKernelGGL(k_exec_insert_flags<true>, dim3(div_up(n, 256)), dim3(256), 0, ctx->stream, keys_d, n, slots_d, mask, first_d);
    else
        hipLaunchKernelGGL(k_exec_insert_flags<false>, dim3(div_up(n, 256)), dim3(256), 0, ctx->stream, keys_d, n, slots_d, mask, first_d);
    IPCFP_HIP(ctx, hipGetLastError());
    return IPCFP_OK;
}

int launch_exec_compact(ipcfp_ctx* ctx, const CidKey* keys_d, uint32_t n, const uint32_t* first_d,
                        const uint32_t* pos_d, CidKey* out_d) {
    if (n == 0) return IPCFP_OK;
    hipLaunchKernelGGL(k_exec_compact, dim3(div_up(n, 256)), dim3(256), 0, ctx->stream, keys_d, n, first_d, pos_d,
                       out_d);
    IPCFP_HIP(ctx, hipGetLastError());
    return IPCFP_OK;
}

int launch_verify_events(ipcfp_ctx* ctx, const WitnessView& w, const EventClaimPacked* claims_d, uint32_t n,
                         const TipsetCtxDev* ctxs_d, uint32_t n_ctxs, const uint8_t* blob_d, uint64_t blob_len,
                         const ipcfp_trust_policy_t& trust, const ipcfp_event_filter_t* filter, uint8_t* status_d,
                         void* where_d, bool tabulated) {
    if (n == 0) return IPCFP_OK;
    ipcfp_event_filter_t f{};
    if (filter) f = *filter;
    {
        ProfileScope prof(ctx, IPCFP_K_EVENT_VERIFY);
        // claims whose receipt's events are tabulated are settled by a kernel that parses nothing; the general
        // walker then only serves what that one left pending
        if (tabulated) {
            int rc = launch_verify_events_table(ctx, w, claims_d, n, ctxs_d, n_ctxs, blob_d, blob_len, trust, f, filter ? 1 : 0,
                                                status_d, where_d);
            if (rc) return rc;
        }
        hipLaunchKernelGGL(k_verify_events, dim3(div_up(n, 256)), dim3(256), 0, ctx->stream, w, claims_d, n, ctxs_d,
                           n_ctxs, blob_d, blob_len, trust, f, filter ? 1 : 0, status_d, static_cast<ValueLoc*>(where_d),
                           tabulated ? 1 : 0);
    }
    IPCFP_HIP(ctx, hipGetLastError());
    return IPCFP_OK;
}

}  // namespace ipcfp
