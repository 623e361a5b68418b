// csrc/kernels/event_scan.hip — K6: the two-pass event-filter scan, one receipt per lane.
//
// Replaces find_matching_events (src/proofs/events/generator.rs:180-307):
//   PASS 1 (:209-239)  for every receipt with an events_root: Amt::<StampedEvent>::load + for_each,
//                      emitter filter (:220-224), extract_evm_log (common/evm.rs:13-59),
//                      EventMatcher::matches_log (:38-40)  →  has_matching per receipt
//   PASS 2 (:242-301)  for matching receipts only: r_amt.get(i) (records the receipt path), the events
//                      AMT walked again on a recorder, one EventProof per matching event in index order.
// The receipt list the reference takes from RPC (:199-204) is the receipts AMT enumerated in index
// order (amt_enum.hip).  K8: with `touched` set in the WitnessView, PASS 2 marks exactly the blocks
// the reference's RecordingBlockStores see (rec_receipts + one rec_events per matching receipt,
// src/proofs/common/blockstore.rs:26-30); PASS 1 runs untracked, like the throw-away recorder.
#include <hip/hip_runtime.h>

#include "../common.h"
#include "amt_enum.h"
#include "event_table.h"
#include "event_log_dev.h"
#include "walk_dev.h"
#include "launch.h"
#include "scan_dev.h"

namespace ipcfp {

struct EventMatch {  // == ipcfp_event_match_t
    uint64_t exec_index, event_index, emitter;
    ValueLoc event;
    uint32_t pad;
};

// Amt::for_each on one lane (depth-first, explicit stack).  For every value in ascending index
// order `f(index, block, r)` is called with the reader positioned on the value; f MUST consume
// exactly that item, validating it as the AMT's value type (it is the type check serde performs).
// Returns TRUE or the first ERR_* in traversal order.
template <typename F>
__device__ __forceinline__ uint32_t amt_for_each_lane_tall(const WitnessView& w, const AmtRootInfo& root, int vkind, F&& f);

template <typename F>
__device__ __forceinline__ uint32_t amt_for_each_lane(const WitnessView& w, const AmtRootInfo& root, int vkind, F&& f) {
    // FVM event AMTs (bit width 5) are 1-2 levels deep: an explicit stack of 8 levels in registers / scratch.  The root's
    // bit width is the WITNESS's choice (Amt::load takes whatever the root block says, events/verifier.rs:215), so a tree
    // may be up to 64 / bit_width levels tall: those take the stackless walk below — same visits, same order.
    constexpr int kMaxDepth = 8;
    if (root.height >= kMaxDepth) return amt_for_each_lane_tall(w, root, vkind, f);
    uint32_t blk[kMaxDepth], noff[kMaxDepth], next_sub[kMaxDepth];
    uint64_t base[kMaxDepth];
    int depth = 0;
    blk[0] = root.block;
    noff[0] = root.node_off;
    next_sub[0] = 0;
    base[0] = 0;
    const uint32_t bw = root.bit_width;
    while (depth >= 0) {
        const uint64_t height = root.height - uint64_t(depth);
        Rd r = open_block(w, blk[depth]);
        r.pos = noff[depth];
        if (next_sub[depth] == 0 && depth > 0) {  // first visit of a child: decode the whole node
            // (CollapsedNode::expand checks; the root node was validated by amt_load)
            AmtNode nd;
            Rd v = r;
            amt_read_node(v, bw, vkind, ~0u, nd);
            if (noff[depth] == 0) v.finish();
            if (!v.ok()) return IPCFP_ST_ERR_DECODE;
        }
        r.expect_array(3);
        uint32_t bo, bl;
        r.read_bytes(bo, bl);
        const uint64_t nl = r.read_array();
        const uint32_t width = 1u << bw;
        if (nl == 0) {  // Leaf: every value, ascending
            const uint64_t nv = r.read_array();
            uint32_t sub = 0;
            for (uint64_t j = 0; j < nv; ++j) {
                while (!((r.at(bo + (sub >> 3)) >> (sub & 7)) & 1u)) ++sub;
                f(base[depth] + sub, blk[depth], r);
                ++sub;
            }
            --depth;
            continue;
        }
        if (height == 0) return IPCFP_ST_ERR_DECODE;  // a link node at height 0
        // Link node: next set bit at or after next_sub
        uint32_t sub = next_sub[depth];
        uint32_t ordinal = 0;
        for (uint32_t i = 0; i < sub && i < width; ++i) ordinal += (r.at(bo + (i >> 3)) >> (i & 7)) & 1u;
        while (sub < width && !((r.at(bo + (sub >> 3)) >> (sub & 7)) & 1u)) ++sub;
        if (sub >= width) {
            --depth;
            continue;
        }
        next_sub[depth] = sub + 1;
        CidKey key;
        for (uint32_t k = 0; k <= ordinal; ++k) r.read_link_key(key);  // the ordinal-th link
        const uint32_t child = witness_find(w, key);
        if (child == kNoBlock) return IPCFP_ST_ERR_MISSING_BLOCK;
        const uint64_t span = amt_span(bw, height);
        base[depth + 1] = base[depth] + uint64_t(sub) * span;
        ++depth;
        blk[depth] = child;
        noff[depth] = 0;
        next_sub[depth] = 0;
    }
    return IPCFP_ST_TRUE;
}

// The same traversal for trees of ANY height (≤ 64 / bit_width levels, amt_load has checked that) without a per-level
// stack: the path from the root is kept as packed digits (the slot taken at each depth, bit_width bits apiece: at most
// 72 bits), and going back UP a level is a descent from the root along the digits — O(height) block visits per pop, for
// trees nobody builds except to see what the verifier does with them.  Nodes are validated when first entered, leaves
// visited in ascending index order, the first error in traversal order returned: as amt_for_each_lane.
template <typename F>
__device__ __forceinline__ uint32_t amt_for_each_lane_tall(const WitnessView& w, const AmtRootInfo& root, int vkind, F&& f) {
    const uint32_t bw = root.bit_width, width = 1u << bw;
    unsigned __int128 path = 0;
    auto digit = [&](uint32_t k) { return uint32_t(path >> (k * bw)) & (width - 1u); };
    uint32_t depth = 0, blk = root.block, noff = root.node_off, next = 0;
    uint64_t base = 0;
    bool fresh = false;  // (the root node was validated by amt_load)
    for (;;) {
        const uint64_t height = root.height - uint64_t(depth);
        Rd r = open_block(w, blk);
        r.pos = noff;
        if (fresh) {  // CollapsedNode::expand checks, once per node
            AmtNode nd;
            Rd v = r;
            amt_read_node(v, bw, vkind, ~0u, nd);
            v.finish();
            if (!v.ok()) return IPCFP_ST_ERR_DECODE;
            fresh = false;
        }
        r.expect_array(3);
        uint32_t bo, bl;
        r.read_bytes(bo, bl);
        const uint64_t nl = r.read_array();
        bool descend = false;
        if (nl == 0) {  // Leaf: every value, ascending
            const uint64_t nv = r.read_array();
            uint32_t sub = 0;
            for (uint64_t j = 0; j < nv; ++j) {
                while (!((r.at(bo + (sub >> 3)) >> (sub & 7)) & 1u)) ++sub;
                f(base + sub, blk, r);
                ++sub;
            }
        } else {
            if (height == 0) return IPCFP_ST_ERR_DECODE;  // a link node at height 0
            uint32_t sub = next, ordinal = 0;
            for (uint32_t i = 0; i < sub && i < width; ++i) ordinal += (r.at(bo + (i >> 3)) >> (i & 7)) & 1u;
            while (sub < width && !((r.at(bo + (sub >> 3)) >> (sub & 7)) & 1u)) ++sub;
            if (sub < width) {
                CidKey key;
                for (uint32_t k = 0; k <= ordinal; ++k) r.read_link_key(key);  // the ordinal-th link
                const uint32_t child = witness_find(w, key);
                if (child == kNoBlock) return IPCFP_ST_ERR_MISSING_BLOCK;
                path &= ~((unsigned __int128)(width - 1u) << (depth * bw));
                path |= (unsigned __int128)sub << (depth * bw);
                base += uint64_t(sub) * amt_span(bw, height);
                ++depth;
                blk = child;
                noff = 0;
                next = 0;
                fresh = true;
                descend = true;
            }
        }
        if (descend) continue;
        // this node is done: back to its parent — found again from the root along the digits
        if (depth == 0) return IPCFP_ST_TRUE;
        --depth;
        next = digit(depth) + 1u;
        blk = root.block;
        noff = root.node_off;
        base = 0;
        for (uint32_t k = 0; k < depth; ++k) {
            Rd q = open_block(w, blk);
            q.pos = noff;
            q.expect_array(3);
            uint32_t qo, ql;
            q.read_bytes(qo, ql);
            (void)q.read_array();
            const uint32_t sub = digit(k);
            uint32_t ordinal = 0;
            for (uint32_t i = 0; i < sub; ++i) ordinal += (q.at(qo + (i >> 3)) >> (i & 7)) & 1u;
            CidKey key;
            for (uint32_t j = 0; j <= ordinal; ++j) q.read_link_key(key);
            const uint32_t child = witness_find(w, key);
            if (child == kNoBlock || !q.ok()) return IPCFP_ST_ERR_MISSING_BLOCK;  // (it resolved on the way down)
            base += uint64_t(sub) * amt_span(bw, root.height - uint64_t(k));
            blk = child;
            noff = 0;
        }
    }
}

// Amt::<V>::load + for_each fused for the overwhelmingly common shape — a v3 AMT whose root is a leaf (height 0:
// up to 2^bit_width values, e.g. ≤ 32 events of one receipt) — in ONE pass over the root block: the callback is
// the per-value type check, so decoding the node first (amt_load) and visiting it afterwards (for_each) would
// parse every value twice.  *handled = false (nothing consumed, f never called) when the root is taller or holds
// links: the caller then takes amt_load + amt_for_each_lane.  Every decode failure is ERR_DECODE on both routes.
template <typename F>
__device__ __forceinline__ uint32_t amt3_for_each_leaf_root(const WitnessView& w, const CidKey& root, bool& handled, F&& f) {
    handled = true;
    const uint32_t b = witness_find(w, root);
    if (b == kNoBlock) return IPCFP_ST_ERR_MISSING_BLOCK;
    Rd r = open_block(w, b);
    r.expect_array(4);
    const uint64_t bw = r.read_uint();
    if (r.ok() && (bw < 1 || bw > kAmtMaxBitWidth)) r.fail();
    const uint64_t height = r.read_uint();
    (void)r.read_uint();  // count: not checked by load or for_each
    if (!r.ok()) return IPCFP_ST_ERR_DECODE;
    if (height != 0) {
        handled = false;
        return IPCFP_ST_TRUE;
    }
    AmtNode nd;  // only the bitmap words are used here
    nd.width = 1u << uint32_t(bw);
    r.expect_array(3);
    uint32_t bo, bl;
    r.read_bytes(bo, bl);
    if (!r.ok()) return IPCFP_ST_ERR_DECODE;
    const bool bmap_len_ok = bl == (nd.width + 7) / 8;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint32_t lo = 8u * uint32_t(k);
        uint64_t v = 0;
        if (bl > lo) {
            v = r.peek64(bo + lo);
            const uint32_t valid = bl - lo;
            if (valid < 8) v &= (1ull << (8u * valid)) - 1ull;
        }
        nd.b[k] = v;
    }
    if (nd.width < 64) nd.b[0] &= (1ull << nd.width) - 1ull;
    const uint64_t nl = r.read_array();
    if (!r.ok()) return IPCFP_ST_ERR_DECODE;
    if (nl != 0) {  // links in a height-0 root: an error on the general route as well, let it say which
        handled = false;
        return IPCFP_ST_TRUE;
    }
    const uint64_t nv = r.read_array();
    if (!r.ok() || !bmap_len_ok || nv != nd.popcount()) return IPCFP_ST_ERR_DECODE;
    uint32_t sub = 0;
    for (uint64_t j = 0; j < nv && r.ok(); ++j) {
        while (!nd.bit(sub)) ++sub;  // nv == popcount: a set bit exists
        f(uint64_t(sub), b, r);
        ++sub;
    }
    r.finish();
    return r.ok() ? uint32_t(IPCFP_ST_TRUE) : uint32_t(IPCFP_ST_ERR_DECODE);
}

// decode a receipt value; returns false when events_root is null
__device__ __forceinline__ bool receipt_events_root(const WitnessView& w, const LeafRef& l, CidKey& root) {
    Rd r;
    r.init(w.arena + w.off[l.block] + l.off, l.len);
    uint32_t o, n;
    r.expect_array(4);
    (void)r.read_uint();
    r.read_bytes(o, n);
    (void)r.read_uint();
    if (r.at_null()) return false;
    r.read_link_key(root);
    return true;
}

// PASS 1: counts[t] = number of matching events of receipt leaf t
__global__ __launch_bounds__(256, IPCFP_WALK_WAVES) void k_scan_pass1(WitnessView w, const LeafRef* __restrict__ receipts, uint32_t n,
                                                    ScanParams sp, uint32_t* __restrict__ counts,
                                                    unsigned long long* __restrict__ err) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    uint32_t c = 0;
    CidKey ev_root;
    if (receipt_events_root(w, receipts[t], ev_root)) {
        auto visit = [&](uint64_t, uint32_t, Rd& er) {
            uint64_t emitter;
            EvmLogLoc log;
            decode_event_log(er, emitter, log);                // parses (and type-checks) the StampedEvent
            if (sp.has_actor && emitter != sp.actor) return;  // :220-224
            if (log_matches(er, log, sp.filter)) ++c;         // :227-231
        };
        bool handled;
        uint32_t st = amt3_for_each_leaf_root(w, ev_root, handled, visit);  // generator.rs:215 + :218
        if (!handled) {
            c = 0;
            AmtRootInfo info;
            st = amt_load(w, ev_root, 3, VK_STAMPED_EVENT, info);
            if (st == IPCFP_ST_TRUE) st = amt_for_each_lane(w, info, VK_STAMPED_EVENT, visit);
        }
        if (st != IPCFP_ST_TRUE) {
            atomicMin(err, (unsigned long long)pack_enum_error(1, t, st));
            c = 0;
        }
    }
    counts[t] = c;
}

// PASS 2: matching receipts write their matches in order; the recorded blocks are marked in w.touched
// (only the few matching receipts do any work here, so occupancy is irrelevant and the register allocator
// gets the whole file: at 4 waves/SIMD this kernel spilled ≈1 KB per lane)
__global__ __launch_bounds__(256, 2) void k_scan_pass2(WitnessView w, CidKey receipts_root,
                                                    const LeafRef* __restrict__ receipts, uint32_t n, ScanParams sp,
                                                    const uint32_t* __restrict__ counts,
                                                    const uint32_t* __restrict__ offsets,
                                                    EventMatch* __restrict__ matches, uint64_t matches_cap,
                                                    uint8_t* __restrict__ has_match, uint64_t has_cap, uint64_t has_base,
                                                    const ReceiptRec* __restrict__ skip_tabulated) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    const bool recording = w.touched != nullptr;
    if (t == 0 && recording) {  // `Amtv0::load(&receipts_root, &rec_receipts)` records the root even without matches (:195-196)
        AmtRootInfo info;
        (void)amt_load(w, receipts_root, 0, VK_RECEIPT, info);
    }
    if (t >= n) return;
    const LeafRef leaf = receipts[t];
    const uint32_t c = counts[t];
    // one byte per receipt index; a shard's map starts at its first index (has_base = receipt_lo)
    if (leaf.index >= has_base && leaf.index - has_base < has_cap) has_match[leaf.index - has_base] = c ? 1 : 0;
    if (c == 0) return;
    if (skip_tabulated && skip_tabulated[t].kind == RK_TABLE) return;  // k_scan_pass2_table wrote its matches
    // `r_amt.get(i)` on the recorder (:249).  Its only observable effect is the recorded path: the index
    // came out of this very AMT's enumeration, which validated every node, so the get cannot return None
    // or Err — without a recorder there is nothing to do.
    if (recording) {
        AmtRootInfo rinfo;
        if (amt_load(w, receipts_root, 0, VK_RECEIPT, rinfo) == IPCFP_ST_TRUE) {
            ValueLoc rl;
            (void)amt_get(w, rinfo, VK_RECEIPT, leaf.index, rl);
        }
    }
    CidKey ev_root;
    if (!receipt_events_root(w, leaf, ev_root)) return;
    uint32_t k = 0;
    const uint32_t o = offsets[t];
    auto visit = [&](uint64_t j, uint32_t b, Rd& er) {
        const uint32_t start = er.pos;
        uint64_t emitter;
        EvmLogLoc log;
        decode_event_log(er, emitter, log);
        if (sp.has_actor && emitter != sp.actor) return;
        if (!log_matches(er, log, sp.filter)) return;
        if (matches && k < c && uint64_t(o) + k < matches_cap) matches[o + k] = EventMatch{leaf.index, j, emitter, ValueLoc{b, start, er.pos - start}, 0};
        ++k;
    };
    // the same blocks in the same order as PASS 1, which found them sound: nothing here can fail (:259-:262)
    bool handled;
    (void)amt3_for_each_leaf_root(w, ev_root, handled, visit);
    if (!handled) {
        k = 0;
        AmtRootInfo info;
        if (amt_load(w, ev_root, 3, VK_STAMPED_EVENT, info) != IPCFP_ST_TRUE) return;
        (void)amt_for_each_lane(w, info, VK_STAMPED_EVENT, visit);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// The event table (event_table.h): PASS 1 that also leaves a record per event.
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ bool rec_matches(const uint8_t* __restrict__ arena, const EventRec& e, const ScanParams& sp) {
    if (sp.has_actor && e.emitter != sp.actor) return false;
    const uint32_t nt = uint32_t(e.base_flags >> kEvTopicShift) & 0xffu;
    if (!(e.base_flags & kEvIsLog) || nt < 2) return false;
    const uint8_t* item = arena + (e.base_flags & kEvBaseMask);
    const uint8_t* t0 = item + e.topic_rel[0];
    const uint8_t* t1 = (e.base_flags & kEvCaseA) ? t0 + 32 : item + e.topic_rel[1];
    uint64_t diff = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        uint64_t a, b, fa, fb;
        __builtin_memcpy(&a, t0 + 8 * k, 8);
        __builtin_memcpy(&b, t1 + 8 * k, 8);
        __builtin_memcpy(&fa, sp.filter.topic0 + 8 * k, 8);
        __builtin_memcpy(&fb, sp.filter.topic1 + 8 * k, 8);
        diff |= (a ^ fa) | (b ^ fb);
    }
    return diff == 0;
}

// One receipt per lane: events_root → block id → the block's record (k_block_events) → ReceiptRec and, when the table
// was built with this scan's filter, the match count.  No block is parsed here except the receipt value itself; a
// receipt whose block the table does not cover is marked for k_receipt_walk (counts[t] = kWalkPending).
constexpr uint32_t kWalkPending = 0xffffffffu;

__global__ __launch_bounds__(256) void k_receipt_events(WitnessView w, const LeafRef* __restrict__ receipts, uint32_t n,
                                                        int count_matches, const BlockRec* __restrict__ brecs,
                                                        ReceiptRec* __restrict__ rrecs, uint32_t* __restrict__ counts,
                                                        unsigned long long* __restrict__ err) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    ReceiptRec rr{RK_NO_EVENTS, 0, 0, kNoBlock, 0};
    uint32_t c = 0;
    CidKey ev_root;
    if (receipts[t].block == kNoBlock) {
        // a leaf the enumeration could not produce (its anomaly flag is on its way to the host, which then redoes the
        // whole walk the general way — host/verify_fast.cpp): nothing to read, nothing decided
        rr.kind = RK_WALK;
    } else if (receipt_events_root(w, receipts[t], ev_root)) {
        const uint32_t b = witness_find(w, ev_root);
        rr.block = b;
        if (b == kNoBlock) {
            rr.kind = IPCFP_ST_ERR_MISSING_BLOCK;
            atomicMin(err, (unsigned long long)pack_enum_error(1, t, rr.kind));
        } else {
            const BlockRec br = brecs[b];
            if ((br.kind_matches & 0xffu) == RK_TABLE) {
                rr.kind = RK_TABLE;
                rr.first = br.first;
                rr.bitmap = br.bitmap;
                c = count_matches ? br.kind_matches >> 8 : 0u;
            } else {
                rr.kind = RK_WALK;
                c = kWalkPending;
            }
        }
    }
    rrecs[t] = rr;
    if (counts) counts[t] = c;
}

// The general route for the receipts k_receipt_events could not settle from the table: names decode errors, counts
// the matches of tall / wide / oversized trees.  Every other lane leaves at once.
__global__ __launch_bounds__(256, IPCFP_WALK_WAVES) void k_receipt_walk(WitnessView w, const LeafRef* __restrict__ receipts, uint32_t n,
                                                      ScanParams sp, int count_matches, const ReceiptRec* __restrict__ rrecs,
                                                      uint32_t* __restrict__ counts, unsigned long long* __restrict__ err) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    if (counts ? counts[t] != kWalkPending : rrecs[t].kind != RK_WALK) return;
    uint32_t c = 0;
    CidKey ev_root;
    if (receipts[t].block != kNoBlock && receipt_events_root(w, receipts[t], ev_root)) {
        auto visit = [&](uint64_t, uint32_t, Rd& er) {
            uint64_t emitter;
            EvmLogLoc log;
            decode_event_log(er, emitter, log);
            if (!count_matches || (sp.has_actor && emitter != sp.actor)) return;
            if (log_matches(er, log, sp.filter)) ++c;
        };
        bool handled;
        uint32_t st = amt3_for_each_leaf_root(w, ev_root, handled, visit);
        if (!handled) {
            c = 0;
            AmtRootInfo info;
            st = amt_load(w, ev_root, 3, VK_STAMPED_EVENT, info);
            if (st == IPCFP_ST_TRUE) st = amt_for_each_lane(w, info, VK_STAMPED_EVENT, visit);
        }
        if (st != IPCFP_ST_TRUE) {  // the scan stops here; a verify call walks this receipt itself (kind stays RK_WALK)
            atomicMin(err, (unsigned long long)pack_enum_error(1, t, st));
            c = 0;
        }
    }
    if (counts) counts[t] = c;
}

// the match count of a (new) filter from an existing table; receipts the table does not cover are walked
__global__ __launch_bounds__(256, IPCFP_WALK_WAVES) void k_count_from_table(WitnessView w, const LeafRef* __restrict__ receipts, uint32_t n,
                                                          ScanParams sp, const ReceiptRec* __restrict__ rrecs,
                                                          const EventRec* __restrict__ erecs, uint32_t* __restrict__ counts,
                                                          unsigned long long* __restrict__ err) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const ReceiptRec rr = rrecs[t];
    uint32_t c = 0;
    if (rr.kind == RK_TABLE) {
        const uint32_t ne = __popcll(rr.bitmap);
        for (uint32_t j = 0; j < ne; ++j)
            if (rec_matches(w.arena, erecs[rr.first + j], sp)) ++c;
    } else if (rr.kind == RK_WALK) {
        CidKey ev_root;
        if (receipt_events_root(w, receipts[t], ev_root)) {
            auto visit = [&](uint64_t, uint32_t, Rd& er) {
                uint64_t emitter;
                EvmLogLoc log;
                decode_event_log(er, emitter, log);
                if (sp.has_actor && emitter != sp.actor) return;
                if (log_matches(er, log, sp.filter)) ++c;
            };
            bool handled;
            uint32_t st = amt3_for_each_leaf_root(w, ev_root, handled, visit);
            if (!handled) {
                c = 0;
                AmtRootInfo info;
                st = amt_load(w, ev_root, 3, VK_STAMPED_EVENT, info);
                if (st == IPCFP_ST_TRUE) st = amt_for_each_lane(w, info, VK_STAMPED_EVENT, visit);
            }
            if (st != IPCFP_ST_TRUE) {
                atomicMin(err, (unsigned long long)pack_enum_error(1, t, st));
                c = 0;
            }
        }
    } else if (rr.kind >= 64) {
        atomicMin(err, (unsigned long long)pack_enum_error(1, t, rr.kind));
    }
    counts[t] = c;
}

// PASS 2 without a recorder: the matches of tabulated receipts come straight from the records
__global__ __launch_bounds__(256) void k_scan_pass2_table(WitnessView w, const LeafRef* __restrict__ receipts, uint32_t n,
                                                          ScanParams sp, const ReceiptRec* __restrict__ rrecs,
                                                          const EventRec* __restrict__ erecs,
                                                          const uint32_t* __restrict__ counts,
                                                          const uint32_t* __restrict__ offsets,
                                                          EventMatch* __restrict__ matches, uint64_t matches_cap,
                                                          uint8_t* __restrict__ has_match, uint64_t has_cap, uint64_t has_base) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const uint32_t c = counts[t];
    const uint64_t index = receipts[t].index;
    if (index >= has_base && index - has_base < has_cap) has_match[index - has_base] = c ? 1 : 0;
    if (c == 0) return;
    const ReceiptRec rr = rrecs[t];
    if (rr.kind != RK_TABLE || !matches) return;  // the rest is k_scan_pass2's (general walk)
    uint32_t k = 0, ord = 0;
    const uint32_t o = offsets[t];
    const uint64_t block_base = w.off[rr.block];
    for (uint32_t j = 0; j < 64; ++j) {
        if (!((rr.bitmap >> j) & 1ull)) continue;
        const EventRec e = erecs[rr.first + ord++];
        if (!rec_matches(w.arena, e, sp)) continue;
        if (k < c && uint64_t(o) + k < matches_cap)
            matches[o + k] = EventMatch{index, j, e.emitter,
                                        ValueLoc{rr.block, uint32_t((e.base_flags & kEvBaseMask) - block_base), e.ev_len}, 0};
        ++k;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// The scan's tail in ONE launch (find_matching_events' second half, events/generator.rs:242-301): exclusive prefix sum of
// the per-receipt match counts, the has-match map, the match records of tabulated receipts in (exec_index, event_index)
// order, and the call's results — match total and error words — in the MAILBOX (pinned host memory: the host reads them
// without a stream synchronisation or a copy).  Was: tile sums, scan of tile sums, apply, pass 2, a read-back copy and
// two host round trips — ≈ 160 µs of a 1.07 ms step for ≈ 45 µs of kernels (profiles/r03_last_commit_timeline.txt).
//
// The prefix sum is a single pass with DECOUPLED LOOK-BACK: a workgroup takes a ticket (tiles in ticket order: a tile only
// ever waits for tiles that are already running), sums its 1 024 counts, publishes the aggregate, and wave 0 looks back
// over its predecessors' published aggregates / inclusive prefixes, 64 at a time.  A tile's state word carries the
// call's EPOCH in its high bits, so the state array is never cleared: a word of another epoch is "nothing yet".
//   state[tile] = epoch << 34 | flag << 32 | value     flag 1: aggregate of this tile, 2: inclusive prefix up to it
// The workgroup that finishes LAST (a completion counter behind a device-scope fence: every other tile's matches and map
// bytes are then written) publishes the mailbox and resets ticket and counter for the next call.
struct ScanTailCtl {
    unsigned long long* state;    // n_tiles words (context-owned, never cleared)
    unsigned int* ticket;         // [0] ticket, [1] completion count, both left at 0
    unsigned long long* total;    // [0]: the match total, written by the last tile in ticket order
    unsigned long long epoch;     // < 2^30, this call's
    const unsigned long long* err_a;  // nullable: error words of this call's earlier kernels, forwarded to the mailbox
    const unsigned long long* err_b;
    unsigned long long* mailbox;  // pinned host: [0] seq, [1] total, [2] *err_a, [3] *err_b
    unsigned long long seq;
};
constexpr uint32_t kScanTile = 1024;

__global__ __launch_bounds__(256) void k_scan_tail_fused(WitnessView w, const LeafRef* __restrict__ receipts, uint32_t n,
                                                         uint64_t dense_first, ScanParams sp, const ReceiptRec* __restrict__ rrecs,
                                                         const EventRec* __restrict__ erecs, const uint32_t* __restrict__ counts,
                                                         uint32_t* __restrict__ offsets, EventMatch* __restrict__ matches,
                                                         uint64_t matches_cap, uint8_t* __restrict__ has_match, uint64_t has_cap,
                                                         uint64_t has_base, ScanTailCtl ctl, unsigned int* __restrict__ untabulated) {
    __shared__ uint64_t smem[17];
    __shared__ uint32_t s_tile;
    __shared__ uint64_t s_prefix;
    const uint32_t n_tiles = (n + kScanTile - 1) / kScanTile;
    if (threadIdx.x == 0) s_tile = atomicAdd(ctl.ticket, 1u);
    __syncthreads();
    const uint32_t tile = s_tile;
    const uint32_t base = tile * kScanTile + threadIdx.x * 4u;
    uint32_t c[4];
    uint64_t s = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        c[k] = base + k < n ? counts[base + k] : 0u;
        s += c[k];
    }
    uint64_t tile_total;
    const uint64_t ex = block_exclusive_scan(s, smem, &tile_total);
    const unsigned long long tag = ctl.epoch << 34;
    // ---- publish, look back (wave 0) ----
    if (threadIdx.x < 64) {
        const uint32_t lane = threadIdx.x;
        if (lane == 0)
            __hip_atomic_store(ctl.state + tile, tag | ((tile == 0 ? 2ull : 1ull) << 32) | (tile_total & 0xffffffffull),
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        uint64_t prefix = 0;
        int32_t hi = int32_t(tile) - 1;  // the nearest predecessor not yet accounted for
        while (hi >= 0) {
            const int32_t j = hi - int32_t(lane);
            unsigned long long v = 0;
            bool ready = true;
            if (j >= 0) {
                v = __hip_atomic_load(ctl.state + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ready = (v >> 34) == ctl.epoch && ((v >> 32) & 3ull) != 0ull;
            }
            // the window [hi - 63, hi] can be used up to its first not-yet-published word (from hi downwards)
            const uint64_t not_ready = __ballot(!ready);
            const uint32_t usable = not_ready ? uint32_t(__builtin_ctzll(not_ready)) : 64u;  // lanes 0 .. usable-1
            const uint64_t is_prefix = __ballot(ready && j >= 0 && ((v >> 32) & 3ull) == 2ull);
            const uint64_t in_use = usable >= 64u ? ~0ull : ((1ull << usable) - 1ull);
            const uint64_t pfx_in = is_prefix & in_use;
            const uint32_t stop = pfx_in ? uint32_t(__builtin_ctzll(pfx_in)) : 64u;  // first inclusive prefix met
            const uint32_t take = stop < 64u ? stop + 1u : usable;                   // lanes 0 .. take-1 contribute
            uint64_t mine = (lane < take && j >= 0) ? (v & 0xffffffffull) : 0ull;
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) mine += __shfl_xor(mine, d, 64);
            prefix += mine;
            if (stop < 64u) break;
            hi -= int32_t(take);
            if (take == 0) __builtin_amdgcn_s_sleep(1);
        }
        if (lane == 0) {
            if (tile != 0)
                __hip_atomic_store(ctl.state + tile, tag | (2ull << 32) | ((prefix + tile_total) & 0xffffffffull), __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
            s_prefix = prefix;
            // (an atomic exchange that RETURNS has been performed at the coherence point before this lane goes on to the
            // completion counter below — no fence, which on this chip is a write-back of the whole L2.  The empty asm
            // consumes the returned value: with the result discarded the compiler may emit the no-return form, which
            // orders nothing against the later read-modify-write of another address — ADVICE r4)
            if (tile == n_tiles - 1) {
                const unsigned long long before = atomicExch(ctl.total, (unsigned long long)(prefix + tile_total));
                asm volatile("" ::"v"(before));
            }
        }
    }
    __syncthreads();
    // ---- the map, the offsets, the matches of this tile ----
    uint64_t o = s_prefix + ex;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint32_t t = base + k;
        if (t >= n) break;
        const uint64_t index = dense_first != ~0ull ? dense_first + t : receipts[t].index;
        if (index >= has_base && index - has_base < has_cap) has_match[index - has_base] = c[k] ? 1 : 0;
        if (c[k]) {
            offsets[t] = uint32_t(o);
            const ReceiptRec rr = rrecs[t];
            if (rr.kind != RK_TABLE) {
                atomicAdd(untabulated, 1u);  // k_scan_pass2 (the general walk) is queued behind this launch for it
            } else if (matches) {
                uint32_t m = 0, ord = 0;
                const uint64_t block_base = w.off[rr.block];
                for (uint32_t j = 0; j < 64; ++j) {
                    if (!((rr.bitmap >> j) & 1ull)) continue;
                    const EventRec e = erecs[rr.first + ord++];
                    if (!rec_matches(w.arena, e, sp)) continue;
                    if (m < c[k] && o + m < matches_cap)
                        matches[o + m] = EventMatch{index, j, e.emitter,
                                                    ValueLoc{rr.block, uint32_t((e.base_flags & kEvBaseMask) - block_base), e.ev_len}, 0};
                    ++m;
                }
            }
        }
        o += c[k];
    }
    // ---- the last workgroup to finish reports ----
    // No fence: the state words carry everything the tiles tell each other, the counters below are read-modify-writes at
    // the coherence point, and what the tiles wrote for LATER kernels and copies (matches, map) is ordered by the end of
    // this launch — a device-scope release per tile is an L2 write-back per tile on a chip with one L2 per XCD (the first
    // version of this kernel, with acquire / release: 159 µs).
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned int done = atomicAdd(ctl.ticket + 1, 1u);
        if (done == n_tiles - 1) {
            const unsigned long long total = atomicAdd(ctl.total, 0ull);
            const unsigned long long ea = ctl.err_a ? __hip_atomic_load(ctl.err_a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : ~0ull;
            const unsigned long long eb = ctl.err_b ? __hip_atomic_load(ctl.err_b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : ~0ull;
            const unsigned int walk = atomicExch(untabulated, 0u);
            (void)atomicExch(ctl.ticket, 0u);
            (void)atomicExch(ctl.ticket + 1, 0u);
            __hip_atomic_store(ctl.mailbox + 1, total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(ctl.mailbox + 2, ea, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(ctl.mailbox + 3, eb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(ctl.mailbox + 4, (unsigned long long)walk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(ctl.mailbox, ctl.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

// scratch_d: [ticket u32, done u32, untabulated u32, pad | total u64 | state n_tiles u64]  (context-owned; the counters are at
// a FIXED place and zero between calls, the state words are told apart by the epoch)
int launch_scan_tail_fused(ipcfp_ctx* ctx, const WitnessView& w, const LeafRef* receipts_d, uint32_t n, uint64_t dense_first,
                           const ipcfp_event_filter_t& filter, int has_actor, uint64_t actor, const EventTableView& table,
                           const uint32_t* counts_d, uint32_t* offsets_d, void* matches_d, uint64_t matches_cap, uint8_t* has_match_d,
                           uint64_t has_cap, uint64_t has_base, unsigned long long* scratch_d, unsigned long long epoch,
                           const unsigned long long* err_a_d, const unsigned long long* err_b_d, unsigned long long* mailbox_dev,
                           unsigned long long seq) {
    const uint32_t n_tiles = div_up(n, kScanTile);
    ScanTailCtl ctl;
    ctl.ticket = reinterpret_cast<unsigned int*>(scratch_d);
    ctl.total = scratch_d + 2;
    ctl.state = scratch_d + 3;
    ctl.epoch = epoch & ((1ull << 30) - 1ull);
    ctl.err_a = err_a_d;
    ctl.err_b = err_b_d;
    ctl.mailbox = mailbox_dev;
    ctl.seq = seq;
    ScanParams sp{filter, actor, has_actor ? 1u : 0u, 0};
    {
        ProfileScope prof(ctx, IPCFP_K_REPLAY);
        hipLaunchKernelGGL(k_scan_tail_fused, dim3(n_tiles), dim3(256), 0, ctx->stream, w, receipts_d, n, dense_first, sp,
                           table.receipts, table.events, counts_d, offsets_d, static_cast<EventMatch*>(matches_d), matches_cap,
                           has_match_d, has_cap, has_base, ctl, ctl.ticket + 2);
    }
    IPCFP_HIP(ctx, hipGetLastError());
    return IPCFP_OK;
}

int launch_scan_pass1(ipcfp_ctx* ctx, const WitnessView& w, const LeafRef* receipts_d, uint32_t n,
                      const ipcfp_event_filter_t& filter, int has_actor, uint64_t actor, uint32_t* counts_d,
                      unsigned long long* err_d) {
    if (n == 0) return IPCFP_OK;
    ScanParams sp{filter, actor, has_actor ? 1u : 0u, 0};
    {
        ProfileScope prof(ctx, IPCFP_K_EVENT_SCAN);
        hipLaunchKernelGGL(k_scan_pass1, dim3(div_up(n, 256)), dim3(256), 0, ctx->stream, w, receipts_d, n, sp, counts_d,
                           err_d);
    }
    IPCFP_HIP(ctx, hipGetLastError());
    return IPCFP_OK;
}

int launch_receipt_events(ipcfp_ctx* ctx, const WitnessView& w, const LeafRef* receipts_d, uint32_t n,
                          const ipcfp_event_filter_t* filter, int has_actor, uint64_t actor, const BlockRec* brecs_d,
                          ReceiptRec* rrecs_d, uint32_t* counts_d, unsigned long long* err_d, hipStream_t stream) {
    if (n == 0) return IPCFP_OK;
    if (!stream) stream = ctx->stream;
    ScanParams sp{};
    if (filter) sp = ScanParams{*filter, actor, has_actor ? 1u : 0u, 0};
    {
        ProfileScope prof(ctx, IPCFP_K_EVENT_SCAN, stream);
        hipLaunchKernelGGL(k_receipt_events, dim3(div_up(n, 256)), dim3(256), 0, stream, w, receipts_d, n,
                           filter ? 1 : 0, brecs_d, rrecs_d, counts_d, err_d);
        hipLaunchKernelGGL(k_receipt_walk, dim3(div_up(n, 256)), dim3(256), 0, stream, w, receipts_d, n, sp,
                           filter ? 1 : 0, rrecs_d, counts_d, err_d);
    }
    IPCFP_HIP(ctx, hipGetLastError());
    return IPCFP_OK;
}

int launch_count_from_table(ipcfp_ctx* ctx, const WitnessView& w, const LeafRef* receipts_d, uint32_t n,
                            const ipcfp_event_filter_t& filter, int has_actor, uint64_t actor, const EventTableView& table,
                            uint32_t* counts_d, unsigned long long* err_d) {
    if (n == 0) return IPCFP_OK;
    ScanParams sp{filter, actor, has_actor ? 1u : 0u, 0};
    {
        ProfileScope prof(ctx, IPCFP_K_EVENT_SCAN);
        hipLaunchKernelGGL(k_count_from_table, dim3(div_up(n, 256)), dim3(256), 0, ctx->stream, w, receipts_d, n, sp,
                           table.receipts, table.events, counts_d, err_d);
    }
    IPCFP_HIP(ctx, hipGetLastError());
    return IPCFP_OK;
}

int launch_scan_pass2(ipcfp_ctx* ctx, const WitnessView& w, const CidKey& receipts_root, const LeafRef* receipts_d,
                      uint32_t n, const ipcfp_event_filter_t& filter, int has_actor, uint64_t actor,
                      const uint32_t* counts_d, const uint32_t* offsets_d, void* matches_d, uint64_t matches_cap,
                      uint8_t* has_match_d, uint64_t has_cap, uint64_t has_base, const EventTableView* table) {
    ScanParams sp{filter, actor, has_actor ? 1u : 0u, 0};
    const uint32_t threads = n ? n : 1;
    {
        ProfileScope prof(ctx, IPCFP_K_REPLAY);
        // without a recorder the matches of tabulated receipts are read off the records; the walking kernel then
        // only serves the receipts the table does not cover
        const bool use_table = table && table->receipts && !w.touched && n;
        if (use_table)
            hipLaunchKernelGGL(k_scan_pass2_table, dim3(div_up(n, 256)), dim3(256), 0, ctx->stream, w, receipts_d, n, sp,
                               table->receipts, table->events, counts_d, offsets_d, static_cast<EventMatch*>(matches_d),
                               matches_cap, has_match_d, has_cap, has_base);
        hipLaunchKernelGGL(k_scan_pass2, dim3(div_up(threads, 256)), dim3(256), 0, ctx->stream, w, receipts_root,
                           receipts_d, n, sp, counts_d, offsets_d, static_cast<EventMatch*>(matches_d), matches_cap,
                           has_match_d, has_cap, has_base, use_table ? table->receipts : nullptr);
    }
    IPCFP_HIP(ctx, hipGetLastError());
    return IPCFP_OK;
}

}  // namespace ipcfp
