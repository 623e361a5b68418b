// csrc/kernels/amt_enum.h — level-synchronous `Amt::for_each` over the HBM-resident witness.
//
// The reference walks the big AMTs sequentially and depth-first
// (`bls_amt.for_each`, `secp_amt.for_each` src/proofs/events/utils.rs:76-90; the receipts list that
// drives `find_matching_events`, src/proofs/events/generator.rs:199-239).  Here every node of a level
// is decoded by its own lane, children are placed with a prefix sum so ascending-index order is
// preserved, and each node is decoded exactly once per pass.  Several AMTs (e.g. the BLS and secp
// trees of every parent block) are enumerated TOGETHER: roots lower than the tallest tree ride along
// as pass-through entries until their level comes up, so the output is the concatenation of the
// for_each sequences in root order.
//
// Errors: `for_each` aborts at the first failing node in depth-first order.  Every failure here is
// recorded as (sequence number, base index of the failing node, status code) packed into one u64
// and combined with atomicMin — the minimum IS the first failure in depth-first order, because a
// failing node has no descendants and every node with a smaller base index is visited earlier.
#pragma once
#include <cstdint>
#include <vector>

#include "../common.h"
#include "amt_types.h"
#include "walk_dev.h"

namespace ipcfp {

}  // namespace ipcfp

struct ipcfp_ctx;
struct ipcfp_witness;

namespace ipcfp {

// Host-side driver state: device buffers reused across levels.
struct AmtEnumResult {
    DevBuf<LeafRef> leaves;
    uint64_t n_leaves = 0;
    uint64_t error = kNoEnumError;  // packed first error (after the final sync)
    bool dense = false;             // every AMT held exactly the indices 0..count-1 (the fast path succeeded)
    bool keys_written = false;      // the values (links) went to the caller's key buffer instead of `leaves`
};

// An enumeration may be restricted to the indices [lo, hi) (a receipt-range shard of one tipset):
// subtrees that hold no index of the range are neither resolved nor loaded — their blocks live in another
// shard's witness — while every node that is visited is validated completely.  Meant for ONE root.

// One more AMT riding along with a call (dense fast path only): its spec is roots_d[n_roots], written by the caller
// or by a kernel queued before; its values have their own type and index range and go to a result of their own.
// `done` stays false when the fast path did not take it (not dense, failed to load, anomaly): the caller then
// enumerates it by itself.  This is how the verify path walks the receipts AMT in the same launches as the
// message AMTs of the execution order instead of in a second chain of per-level launches.
struct EnumExtra {
    int vkind = 0;
    uint64_t lo = 0, hi = ~0ULL;
    AmtEnumResult* out = nullptr;
    bool done = false;
};

// ---- the dense fast path (amt_enum.hip): the tree's shape follows from the roots' (height, bit width, count) ----
struct DenseRoot {
    uint32_t height, bit_width;
    uint64_t count;
    uint64_t lo, hi;  // the indices to enumerate: [lo, hi) with lo < hi <= count, or (0, 0) for an empty tree
    uint32_t vkind;   // value type of this tree
    uint32_t out_sel; // where its values go: 0 = keys_out (links as witness keys), 1 = leaves, 2 = leaves of the extra root
    uint64_t out_off; // ... from this element on
};
constexpr uint32_t kMaxDenseRoots = 2 * IPCFP_MAX_PARENTS + 1;  // BLS + secp per parent block, + the receipts AMT
constexpr uint32_t kMaxDenseLevels = 24;
// interior levels of at most this many entries at the top of the walk share ONE single-workgroup launch (k_dense_top)
constexpr uint32_t kDenseTopMax = 1024;
struct DenseRoots {  // travels as a kernel ARGUMENT (1.6 KB): no copy at the head of the walk
    DenseRoot r[kMaxDenseRoots];
    uint32_t n;
};
struct DensePlan {
    bool ok = false;          // every root is a dense candidate: the walk below may be launched
    uint32_t n_use = 0;       // roots in the walk (the call's own + the extra one when it joined)
    uint32_t max_height = 0;
    uint64_t n_level[kMaxDenseLevels] = {};  // nodes per level (node height)
    uint64_t n_leaves = 0, n_extra = 0, biggest = 0;
    DenseRoots roots{};
};
// what k_dense_link_leaves clears on the side (launch_dense_walk `clear`; only when the plan has key values)
struct DenseClear {
    unsigned long long* slots;  // n_slots × ~0
    uint64_t n_slots;
    uint32_t* words;            // n_words × 0
    uint64_t n_words;
};
void dense_plan(const std::vector<uint64_t>& root_info, uint32_t n_roots, int vkind, bool want_keys, uint64_t lo, uint64_t hi,
                uint32_t has_extra, int extra_vkind, uint64_t extra_lo, uint64_t extra_hi, DensePlan& plan);
int launch_dense_walk(ipcfp_ctx* ctx, const WitnessView& view, const DenseNode* frontier, const DensePlan& plan,
                      DenseNode* a, DenseNode* b, LeafRef* leaves_main, CidKey* keys_main, LeafRef* leaves_extra, uint32_t* anomaly_d,
                      hipStream_t leaves_stream = nullptr,  // non-null (with fork_event): k_dense_leaves runs there, beside
                      hipEvent_t fork_event = nullptr,      // k_dense_link_leaves on the main stream
                      // non-null: the caller runs on the narrow stream (ctx->stream is that stream); the first level of more
                      // than `narrow_max_wg` workgroups — the leaves at the latest — hands over to `wide_stream` through
                      // `wide_event`, and ctx->stream is `wide_stream` on return
                      hipStream_t wide_stream = nullptr, hipEvent_t wide_event = nullptr, uint32_t narrow_max_wg = 0,
                      const DenseClear* clear = nullptr);

int launch_enum_roots(ipcfp_ctx* ctx, const WitnessView& view, const AmtRootSpec* roots_d, uint32_t n_all, int vkind,
                      EnumNode* frontier_d, uint32_t* max_height_d, unsigned long long* err_d, uint64_t* root_info_d,
                      unsigned long long* mailbox, unsigned long long mailbox_seq, DenseNode* dense_frontier_d);
struct TipsetCtxDev;
int launch_txmeta_rehash(ipcfp_ctx* ctx, hipStream_t stream, const WitnessView& view, const TipsetCtxDev* ctx_d, unsigned long long* err_d);

// Enumerate `n_roots` AMTs (device array `roots_d`) whose values have type `vkind`.
// `err_d` is a device u64 initialised by the caller (kNoEnumError or earlier-stage errors); the
// enumerator atomicMin's into it.  Synchronises the stream twice on the dense path (root shapes; anomaly flag
// + error word), once more per level on the general path.
int amt_enumerate(ipcfp_ctx* ctx, const WitnessView& view, const AmtRootSpec* roots_d, uint32_t n_roots, int vkind,
                  unsigned long long* err_d, AmtEnumResult& out, uint64_t lo = 0, uint64_t hi = ~0ULL,
                  DevBuf<CidKey>* keys_out = nullptr,  // VK_CID on the dense path: the links as witness keys, no LeafRefs
                  EnumExtra* extra = nullptr);

// hand a finished enumeration of one AMT to the witness's cache (what amt_enumerate_cached would have produced)
int enum_cache_put(ipcfp_ctx* ctx, ipcfp_witness* w, const CidKey& root, int version, int vkind, uint64_t lo, uint64_t hi,
                   AmtEnumResult& en);

// Enumerate one AMT of the witness, or return the cached enumeration (owned by the witness; valid
// until ipcfp_witness_rebuild_index).
int amt_enumerate_cached(ipcfp_ctx* ctx, ipcfp_witness* w, const CidKey& root, int version, int vkind,
                         const EnumCached** out, uint64_t lo = 0, uint64_t hi = ~0ULL);

}  // namespace ipcfp
