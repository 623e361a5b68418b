// csrc/kernels/amt_enum.hip — level-synchronous `Amt::for_each` (see amt_enum.h).
#include "amt_enum.h"

#include <hip/hip_runtime.h>

#include <cstdlib>
#include <cstring>
#include <vector>

#include "../common.h"
#include "blake2b_dev.h"
#include "launch.h"
#include "tipset_ctx.h"

namespace ipcfp {

__device__ __forceinline__ void enum_error(unsigned long long* err, uint32_t seq, uint64_t base, uint32_t code) {
    atomicMin(err, (unsigned long long)pack_enum_error(seq, base, code));
}

// Amt::load of a root whose node is the canonical DENSE link node (defined behind the dense walk's helpers, below): TRUE with
// `info` filled, or "not this shape" — the item-by-item amt_load then decides (and is the only one to report an error).
__device__ __forceinline__ bool amt_load_dense_root(const WitnessView& w, const CidKey& root, int version, AmtRootInfo& info);

// one root → its frontier entry and its shape {height | bw << 32, count} (~0: the root did not load)
__device__ __forceinline__ void enum_root_one(const WitnessView& w, const AmtRootSpec& spec, int vkind, EnumNode* __restrict__ slot,
                                              uint32_t* __restrict__ max_height, unsigned long long* __restrict__ err,
                                              uint64_t& info0, uint64_t& info1, DenseNode* __restrict__ dense_slot) {
    EnumNode e{kNoBlock, 0, 0, spec.seq, 0, 0, 0};
    DenseNode d{0, 0, kNoBlock, 0, spec.seq, 0};
    info0 = ~0ULL;  // dead
    info1 = 0;
    if (!spec.skip) {
        AmtRootInfo info;
        const uint32_t st = amt_load_dense_root(w, spec.root, int(spec.version), info)
                                ? uint32_t(IPCFP_ST_TRUE)
                                : amt_load(w, spec.root, int(spec.version), spec.kind_p1 ? int(spec.kind_p1) - 1 : vkind, info);
        if (st != IPCFP_ST_TRUE) {
            if (!spec.kind_p1) enum_error(err, spec.seq, 0, st);  // (the extra root: left to its own enumeration)
        } else {
            e.block = info.block;
            e.node_off = info.node_off;
            e.height = uint16_t(info.height);
            e.bit_width = uint8_t(info.bit_width);
            atomicMax(max_height, uint32_t(info.height));
            info0 = uint64_t(uint32_t(info.height)) | (uint64_t(info.bit_width) << 32);
            info1 = info.count;
            d.block = info.block;
            d.goff = w.off[info.block] + info.node_off;
            d.rem = w.len[info.block] - info.node_off;
        }
    }
    *slot = e;
    if (dense_slot) *dense_slot = d;
}

// The TxMeta re-hash of parent block b, left here by the tipset prologue (tipset_ctx.h txmeta_block): put_cbor(&(bls_root,
// secp_root), Blake2b256) must give the CID the header named (src/proofs/events/utils.rs:65-72; the verify path always
// checks).  The block was decoded by the prologue already; its CID is the key it was found under.
__device__ __forceinline__ void txmeta_rehash_lane(const WitnessView& w, const TipsetCtxDev& c, uint32_t b, unsigned long long* __restrict__ err) {
    if (b >= c.n_parents || b >= IPCFP_MAX_PARENTS) return;
    if (c.txmeta_block[b] == 0u) return;  // (1 + block id: tipset_ctx.h)
    const uint32_t tb = c.txmeta_block[b] - 1u;
    const uint32_t seq = c.n_parents + 3 * b;
    Rd r;
    r.init(w.arena + w.off[tb], w.len[tb]);
    uint32_t o[2], l[2];
    r.expect_array(2);
    r.read_link(o[0], l[0]);
    r.read_link(o[1], l[1]);
    r.finish();
    if (!r.ok() || l[0] > 64u || l[1] > 64u) {  // (the prologue took this very block: not reachable)
        enum_error(err, seq, 0, IPCFP_ST_ERR_DECODE);
        return;
    }
    uint64_t d[4];
    if (w.len[tb] == 87u && o[0] == 6u && l[0] == 38u && o[1] == 49u && l[1] == 38u) {
        // `82 | d8 2a 58 27 00 ‖ 38 | d8 2a 58 27 00 ‖ 38`: the canonical re-encoding IS the block — one compression
        // straight from its bytes (blocks sit on 128-byte lines with tail slack), no byte buffer in scratch
        uint64_t m[16];
        const ulonglong2* q = reinterpret_cast<const ulonglong2*>(w.arena + w.off[tb]);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const ulonglong2 v = k < 6 ? q[k] : ulonglong2{0ull, 0ull};
            m[2 * k] = v.x;
            m[2 * k + 1] = v.y;
        }
        b2b::mask_tail(m, 87u);
        uint64_t h[8];
        b2b::init256(h);
        b2b::compress<0>(h, m, 87ull, true);
        d[0] = h[0];
        d[1] = h[1];
        d[2] = h[2];
        d[3] = h[3];
    } else {
    uint8_t enc[200];
    uint32_t n = 0;
    enc[n++] = 0x82;
    for (int k = 0; k < 2; ++k) {
        enc[n++] = 0xd8;
        enc[n++] = 0x2a;
        const uint32_t bl = l[k] + 1;
        if (bl < 24) enc[n++] = uint8_t(0x40 | bl);
        else { enc[n++] = 0x58; enc[n++] = uint8_t(bl); }
        enc[n++] = 0x00;
        for (uint32_t i = 0; i < l[k]; ++i) enc[n++] = uint8_t(r.at(o[k] + i));
    }
    blake2b256_small(enc, n, d);
    }
    CidKey re;
    re.w[0] = 0x00002002e4a07101ULL | (d[0] << 48);
    re.w[1] = (d[0] >> 16) | (d[1] << 48);
    re.w[2] = (d[1] >> 16) | (d[2] << 48);
    re.w[3] = (d[2] >> 16) | (d[3] << 48);
    re.w[4] = d[3] >> 16;
    if (!cid_equal(re, load_cid_slot(w.cids, tb))) enum_error(err, seq, 0, IPCFP_ST_ERR_TXMETA_MISMATCH);
}

// one lane per parent block.  Nothing waits for this launch but the end of the call (host/verify_fast.cpp queues it on
// the aux stream behind the receipts' event records): as a second workgroup of k_enum_roots it took 123 µs and held
// the first level of the walk back for as long.
__global__ __launch_bounds__(64) void k_txmeta_rehash(WitnessView w, const TipsetCtxDev* __restrict__ c, unsigned long long* __restrict__ err) {
    txmeta_rehash_lane(w, *c, threadIdx.x, err);
}

// roots → frontier (one entry per root, in root order)
__global__ __launch_bounds__(128) void k_enum_roots(WitnessView w, const AmtRootSpec* __restrict__ roots, uint32_t n,
                                                   int vkind, EnumNode* __restrict__ frontier,
                                                   uint32_t* __restrict__ max_height,
                                                   unsigned long long* __restrict__ err,
                                                   uint64_t* __restrict__ root_info /* n × {height|bw<<32, count} */,
                                                   unsigned long long* __restrict__ mailbox, unsigned long long mailbox_seq,
                                                   DenseNode* __restrict__ dense_frontier) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t info0 = ~0ULL, info1 = 0;
    if (t < n) {
        enum_root_one(w, roots[t], vkind, frontier + t, max_height, err, info0, info1, dense_frontier ? dense_frontier + t : nullptr);
        root_info[2 * t] = info0;
        root_info[2 * t + 1] = info1;
    }
    // The mailbox (host/verify_fast.cpp): PINNED HOST memory the device writes while the stream keeps going.  The root
    // shapes are all the host needs to size and queue the rest of the walk, so it polls these words instead of draining
    // the stream with a synchronisation: [0] = sequence number (written last), [1 + 2t ..] = the shapes.  Single-block
    // launches only (n ≤ 128 roots: 2 · IPCFP_MAX_PARENTS + 1 fit), so that "every root is through" is a __syncthreads.
    if (mailbox && gridDim.x == 1) {
        if (t < n) {
            __hip_atomic_store(mailbox + 1 + 2 * t, (unsigned long long)info0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(mailbox + 2 + 2 * t, (unsigned long long)info1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        __threadfence_system();
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_store(mailbox, mailbox_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// decode the node of entry e (validating it completely); false ⇒ decode error
__device__ __forceinline__ bool enum_read_node(const WitnessView& w, const EnumNode& e, int vkind, AmtNode& nd) {
    Rd r = open_block(w, e.block);
    r.pos = e.node_off;
    amt_read_node(r, e.bit_width, vkind, ~0u, nd);
    if (e.node_off == 0) r.finish();
    return r.ok();
}

// interior level L ≥ 1: how many entries does each frontier entry contribute to the next level?
// `bound` (nullable): device-side number of valid frontier entries when the host launched with a
// PREDICTED size (speculative path); entries past it count as empty.
// An index range [lo, hi) restricts an enumeration to the values inside it (a receipt-range shard of one tipset,
// SURVEY.md §8e): a child whose span does not meet the range is not counted, not resolved and not loaded — its
// block belongs to another shard's witness.  Nodes that ARE visited are still validated completely.
struct EnumRange {
    uint64_t lo, hi;
};
__device__ __forceinline__ bool span_meets(uint64_t base, uint64_t span, const EnumRange& rg) {
    const uint64_t end = base + span < base ? ~0ULL : base + span;  // saturating
    return base < rg.hi && end > rg.lo;
}
// slots of node `e` (bitmap in nd) whose child span meets the range
__device__ __forceinline__ uint32_t links_in_range(const AmtNode& nd, const EnumNode& e, const EnumRange& rg) {
    const uint64_t span = amt_span(e.bit_width, e.height);
    uint32_t c = 0;
    for (uint32_t sub = 0; sub < nd.width; ++sub)
        if (nd.bit(sub) && span_meets(e.base + uint64_t(sub) * span, span, rg)) ++c;
    return c;
}

__global__ __launch_bounds__(256) void k_enum_count(WitnessView w, const EnumNode* __restrict__ frontier, uint32_t n,
                                                    uint32_t level, int vkind, uint32_t* __restrict__ counts,
                                                    unsigned long long* __restrict__ err,
                                                    const uint64_t* __restrict__ bound, EnumRange rg) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    if (bound && t >= *bound) {
        counts[t] = 0;
        return;
    }
    const EnumNode e = frontier[t];
    uint32_t c = 0;
    if (e.block != kNoBlock) {
        if (e.leaf_ready || e.height < level) {
            c = 1;  // rides along
        } else {
            AmtNode nd;
            if (!enum_read_node(w, e, vkind, nd)) enum_error(err, e.seq, e.base, IPCFP_ST_ERR_DECODE);
            else c = nd.nlinks ? links_in_range(nd, e, rg) : 1;  // a Leaf above height 0 is carried down as-is
        }
    }
    counts[t] = c;
}

// One lane per OUTPUT entry (child): lane j finds its parent with a binary search over the exclusive
// offsets, steps over the links before its own (the node was validated by k_enum_count) and resolves
// one link.  A node's 8-32 children are thus resolved by as many lanes side by side instead of one lane
// chasing 8-32 hash probes in sequence.
__global__ __launch_bounds__(256) void k_enum_expand(WitnessView w, const EnumNode* __restrict__ frontier, uint32_t n,
                                                     uint32_t level, const uint32_t* __restrict__ offsets,
                                                     uint32_t total, EnumNode* __restrict__ next,
                                                     unsigned long long* __restrict__ err,
                                                     const uint64_t* __restrict__ bound, EnumRange rg) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= total) return;
    if (bound && j >= *bound) return;
    // parent = last t with offsets[t] <= j  (entries that contribute nothing share their successor's offset)
    uint32_t lo = 0, hi = n;
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (offsets[mid] <= j) lo = mid;
        else hi = mid;
    }
    EnumNode e = frontier[lo];
    const uint32_t k = j - offsets[lo];
    if (e.leaf_ready || e.height < level) {  // rides along (k == 0)
        next[j] = e;
        return;
    }
    Rd r = open_block(w, e.block);
    r.pos = e.node_off;
    r.expect_array(3);
    uint32_t bo, bl;
    r.read_bytes(bo, bl);
    const uint64_t nl = r.read_array();
    if (nl == 0) {  // a Leaf above height 0: carried down as-is
        e.leaf_ready = 1;
        next[j] = e;
        return;
    }
    // the k-th set bit whose child meets the range names the slot; step over the links before ours
    const uint64_t span = amt_span(e.bit_width, e.height);
    uint32_t sub = 0, seen = 0, before = 0;
    for (;; ++sub) {
        if ((r.at(bo + (sub >> 3)) >> (sub & 7)) & 1u) {
            if (span_meets(e.base + uint64_t(sub) * span, span, rg)) {
                if (seen == k) break;
                ++seen;
            }
            ++before;
        }
    }
    for (uint32_t q = 0; q < before; ++q) {
        uint32_t m;
        uint64_t a;
        r.head(m, a);  // tag 42
        r.head(m, a);  // byte-string header
        r.pos += uint32_t(a);
    }
    CidKey key;
    r.read_link_key(key);
    EnumNode c{kNoBlock, 0, e.base + uint64_t(sub) * span, e.seq, uint16_t(e.height - 1), e.bit_width, 0};
    const uint32_t b = witness_find(w, key);
    if (b == kNoBlock) enum_error(err, e.seq, c.base, IPCFP_ST_ERR_MISSING_BLOCK);
    else c.block = b;
    next[j] = c;
}

// leaf level: number of values per entry
__global__ __launch_bounds__(256) void k_enum_count_leaf(WitnessView w, const EnumNode* __restrict__ frontier,
                                                         uint32_t n, int vkind, uint32_t* __restrict__ counts,
                                                         unsigned long long* __restrict__ err,
                                                         const uint64_t* __restrict__ bound, EnumRange rg) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    if (bound && t >= *bound) {
        counts[t] = 0;
        return;
    }
    const EnumNode e = frontier[t];
    uint32_t c = 0;
    if (e.block != kNoBlock) {
        AmtNode nd;
        if (!enum_read_node(w, e, vkind, nd)) enum_error(err, e.seq, e.base, IPCFP_ST_ERR_DECODE);
        else if (nd.nlinks) enum_error(err, e.seq, e.base, IPCFP_ST_ERR_DECODE);  // link node at height 0
        else
            for (uint32_t sub = 0; sub < nd.width; ++sub)
                if (nd.bit(sub) && e.base + sub >= rg.lo && e.base + sub < rg.hi) ++c;
    }
    counts[t] = c;
}

__global__ __launch_bounds__(256) void k_enum_emit(WitnessView w, const EnumNode* __restrict__ frontier, uint32_t n,
                                                   int vkind, const uint32_t* __restrict__ counts,
                                                   const uint32_t* __restrict__ offsets,
                                                   LeafRef* __restrict__ leaves, uint64_t cap, EnumRange rg) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    if (counts[t] == 0) return;
    if (uint64_t(offsets[t]) + counts[t] > cap) return;  // speculative path: never write past the prediction
    const EnumNode e = frontier[t];
    const uint32_t o = offsets[t];
    Rd r = open_block(w, e.block);
    r.pos = e.node_off;
    r.expect_array(3);
    uint32_t bo, bl;
    r.read_bytes(bo, bl);
    (void)r.read_array();  // no links in a leaf
    const uint64_t nv = r.read_array();
    uint32_t sub = 0, k = 0;
    for (uint64_t j = 0; j < nv; ++j) {
        while (!((r.at(bo + (sub >> 3)) >> (sub & 7)) & 1u)) ++sub;
        const uint32_t start = r.pos;
        check_value(r, vkind);
        if (e.base + sub >= rg.lo && e.base + sub < rg.hi) leaves[o + k++] = LeafRef{e.block, start, r.pos - start, e.seq, e.base + sub};
        ++sub;
    }
}

// ---------------------------------------------------------------------------------------------
// Dense fast path.  fvm_ipld_amt writes message lists, receipts and events with indices 0..count-1, so
// the whole shape of the tree follows from the root's (height, bit width, count): node p of a level has
// min(W, remaining) entries with the low bits of its bitmap set, and child q of a level hangs off parent
// q / W at slot q % W.  With the shape known there is nothing to count or prefix-sum: ONE kernel per
// level (the first child's lane validates the parent completely, every lane resolves its own link) and
// ONE leaf kernel that validates and emits.  Anything that is not exactly that shape — a sparse node, a
// lying count, a missing block, a decode error — raises `anomaly` and the caller redoes the walk with
// the general level-synchronous path, which also knows how to order errors.
// ---------------------------------------------------------------------------------------------
// nodes of `level` (node height) the whole tree has
__host__ __device__ inline uint64_t dense_total(const DenseRoot& r, uint32_t level) {
    if (r.height < level) return 1;  // rides along until its own level
    const uint64_t shift = uint64_t(r.bit_width) * (level + 1);
    if (shift >= 64) return 1;
    const uint64_t n = (r.count + (1ULL << shift) - 1) >> shift;
    return n ? n : 1;
}
// first node of `level` that holds an index of [lo, hi), and how many such nodes there are
__host__ __device__ inline uint64_t dense_first(const DenseRoot& r, uint32_t level) {
    const uint64_t shift = uint64_t(r.bit_width) * (level + 1);
    return (r.height < level || shift >= 64 || r.hi == 0) ? 0 : (r.lo >> shift);
}
__host__ __device__ inline uint64_t dense_nodes(const DenseRoot& r, uint32_t level) {
    const uint64_t shift = uint64_t(r.bit_width) * (level + 1);
    if (r.height < level || shift >= 64 || r.hi == 0) return 1;
    return ((r.hi - 1) >> shift) - (r.lo >> shift) + 1;
}

// ---- the dense walk's kernels ------------------------------------------------------------------------------------
// Every lane's work is a short chain of DEPENDENT memory latencies — the levels are a few thousand lanes at most — so
// what these kernels are made for is the number of hops, not bytes or instructions:
//   * a frontier entry (DenseNode) carries the arena offset and length of its node, so a child's lane starts reading its
//     parent without asking the off / len tables first;
//   * the node header is not parsed to find out where things are: with the shape known (m entries, the low m bitmap
//     bits) the canonical spelling of the header is COMPUTED, the 16 bytes at the node's start are compared with it, and
//     the lane's own link sits at `header + 43·k` — header and link are fetched side by side, one latency;
//   * after the hash slot, the candidate block's CID, offset and length are fetched side by side as well.
// Three hops per level (node bytes → hash slot → block facts) instead of six.  Anything that is not spelled
// canonically — a non-minimal length, a bit width above 6, a link that is not the standard 43 bytes — is an anomaly and
// the walk is redone by the general level-synchronous path, like every other shape that leaves the dense route.

__device__ __forceinline__ uint64_t load_u64_any(const uint8_t* p) {
    uint64_t v;
    __builtin_memcpy(&v, p, 8);  // one unaligned 8-byte load (gfx950 runs in unaligned-access mode)
    return v;
}

// The canonical header of an AMT node whose bitmap is the low m bits of a W-bit map: `83 | 4x bitmap | 8m (links) …` for a
// link node, `83 | 4x bitmap | 80 | 8m (values) …` for a leaf.  → its length (0: this route does not spell such a node)
// and its bytes as two little-endian words.
__device__ __forceinline__ uint32_t dense_header(uint32_t W, uint32_t m, bool leaf, uint64_t& h0, uint64_t& h1) {
    const uint32_t bl = (W + 7u) / 8u;
    h0 = h1 = 0;
    if (bl > 8u || m > 64u || m == 0u) return 0;
    uint32_t n = 0;
    auto put = [&](uint32_t byte) {  // (no byte array: a dynamically indexed one would live in scratch)
        if (n < 8u) h0 |= uint64_t(byte & 0xffu) << (8u * n);
        else h1 |= uint64_t(byte & 0xffu) << (8u * (n - 8u));
        ++n;
    };
    put(0x83);
    put(0x40u | bl);
    for (uint32_t i = 0; i < bl; ++i) {
        const uint32_t lo = i * 8u;
        put(m >= lo + 8u ? 0xffu : (m > lo ? (1u << (m - lo)) - 1u : 0u));
    }
    if (leaf) put(0x80);  // a leaf holds no links
    if (m < 24u) put(0x80u | m);
    else {
        put(0x98);
        put(m);
    }
    return n;
}
__device__ __forceinline__ bool header_matches(const uint8_t* node, uint32_t hdr, uint64_t h0, uint64_t h1) {
    const uint64_t g0 = load_u64_any(node), g1 = load_u64_any(node + 8);
    const uint64_t m0 = hdr >= 8 ? ~0ull : ((1ull << (8u * hdr)) - 1ull);
    const uint64_t m1 = hdr <= 8 ? 0ull : (hdr >= 16 ? ~0ull : ((1ull << (8u * (hdr - 8u))) - 1ull));
    return ((g0 ^ h0) & m0) == 0 && ((g1 ^ h1) & m1) == 0;
}
// the standard 43-byte link at p:   d8 2a | 58 27 | 00 | 01 71 a0 e4 02 20 | digest[32]
__device__ __forceinline__ bool std_link_at(const uint8_t* p) {
    return load_u64_any(p) == 0xa071010027582ad8ull && (load_u64_any(p + 8) & 0xffffffull) == 0x2002e4ull;
}
__device__ __forceinline__ CidKey std_link_key(const uint8_t* p) {
    CidKey k;
#pragma unroll
    for (int j = 0; j < 5; ++j) k.w[j] = load_u64_any(p + 5 + 8 * j);
    k.w[4] &= (1ull << 48) - 1ull;  // 38 = 4·8 + 6 bytes
    return k;
}

// The root block of a big list — `83 | height | count | node` (Amtv0) or `84 | bit width | height | count | node` (Amt) with
// the node a canonical dense LINK node: `83 | 4x bitmap = low m bits | 8m / 98 m | m standard 43-byte links | 80` and nothing
// behind it — settled from a dozen loads that do not depend on each other.  Exactly what amt_load accepts for these bytes
// (array heads, minimal uints, bitmap of the width's length whose popcount is the link count, well-formed links, no values,
// nothing after the node, height ≤ 64 / bit width); any other spelling or shape returns false and amt_load takes the block.
// Item by item the ten message-list roots and the receipts root of a tipset were 3.7 k instructions of ONE wavefront at
// 30-50 cycles each beside the side streams: 79 µs on the verify call's critical path (profiles/r04_final_timeline.txt).
__device__ __forceinline__ bool amt_load_dense_root(const WitnessView& w, const CidKey& root, int version, AmtRootInfo& info) {
    const uint32_t b = witness_find(w, root);
    if (b == kNoBlock) return false;
    const uint8_t* p = w.arena + w.off[b];
    const uint32_t len = w.len[b];
    if (len < 8u) return false;
    const uint64_t g0 = load_u64_any(p), g1 = load_u64_any(p + 8);  // (blocks sit on 128-byte lines with tail slack)
    auto byte_at = [&](uint32_t i) { return uint32_t((i < 8u ? g0 >> (8u * i) : g1 >> (8u * (i - 8u))) & 0xffull); };
    uint32_t pos = 1, bw = 3;
    if (version == 0) {
        if (byte_at(0) != 0x83u) return false;
    } else {
        if (byte_at(0) != 0x84u) return false;
        bw = byte_at(1);
        if (bw < 1u || bw > 6u) return false;  // (an immediate uint; the dense walk spells headers up to width 6)
        pos = 2;
    }
    const uint32_t height = byte_at(pos);
    if (height < 1u || height > 23u || height > 64u / bw) return false;  // immediate uint; a link node at the root
    ++pos;
    // count: a minimal unsigned integer
    const uint32_t cb = byte_at(pos);
    uint64_t count;
    uint32_t cl;
    if (cb < 0x18u) {
        count = cb;
        cl = 1;
    } else if (cb == 0x18u) {
        count = byte_at(pos + 1);
        cl = 2;
        if (count < 24u) return false;
    } else if (cb == 0x19u) {
        count = (uint64_t(byte_at(pos + 1)) << 8) | byte_at(pos + 2);
        cl = 3;
        if (count < 256u) return false;
    } else if (cb == 0x1au) {
        count = (uint64_t(byte_at(pos + 1)) << 24) | (uint64_t(byte_at(pos + 2)) << 16) | (uint64_t(byte_at(pos + 3)) << 8) | byte_at(pos + 4);
        cl = 5;
        if (count < 65536u) return false;
    } else {
        return false;  // (≥ 2^32 entries, or not an unsigned integer: the long way)
    }
    pos += cl;
    if (pos > 8u) return false;  // (cannot happen: 2 + 1 + 5)
    const uint32_t W = 1u << bw;
    const uint64_t span = amt_span(bw, height);  // indices under one link of the root
    if (count == 0 || span == ~0ULL) return false;
    const uint64_t m64 = (count + span - 1) / span;
    if (m64 > W) return false;  // (more entries than the height holds: not a dense tree)
    const uint32_t m = uint32_t(m64);
    uint64_t h0, h1;
    const uint32_t hdr = dense_header(W, m, false, h0, h1);
    const uint8_t* node = p + pos;
    const uint32_t end = pos + hdr + 43u * m + 1u;  // … links, then the empty values array
    if (hdr == 0 || end != len || !header_matches(node, hdr, h0, h1)) return false;
    bool ok = node[hdr + 43u * m] == 0x80u;
    for (uint32_t i = 0; i < m; ++i) ok = ok && std_link_at(node + hdr + 43u * i);
    if (!ok) return false;
    info.block = b;
    info.node_off = pos;
    info.bit_width = bw;
    info.height = height;
    info.count = count;
    return true;
}

// which root does frontier entry j of `level` belong to (dense trees side by side), and where do its entries start?
struct DenseWhere {
    DenseRoot dr;
    uint32_t upper_off, here_off;
};
__device__ __forceinline__ DenseWhere dense_where(const DenseRoot* roots, uint32_t j, uint32_t level_here) {
    DenseWhere o;
    uint32_t r = 0;
    o.upper_off = o.here_off = 0;
    o.dr = roots[0];
    uint64_t n_here = dense_nodes(o.dr, level_here);
    while (j >= o.here_off + n_here) {
        o.here_off += uint32_t(n_here);
        o.upper_off += uint32_t(dense_nodes(o.dr, level_here + 1));
        o.dr = roots[++r];
        n_here = dense_nodes(o.dr, level_here);
    }
    return o;
}

// frontier entering `level` (≥ 1) → frontier entering level-1; one lane per child
__device__ __forceinline__ void dense_level_one(const WitnessView& w, const DenseNode* __restrict__ cur, const DenseRoot* roots,
                                                uint32_t level, uint32_t j, DenseNode* __restrict__ next,
                                                uint32_t* __restrict__ anomaly) {
    const DenseWhere wh = dense_where(roots, j, level - 1);
    const DenseRoot& dr = wh.dr;
    if (dr.height < level) {  // rides along
        next[j] = cur[wh.upper_off];
        return;
    }
    const uint64_t q = dense_first(dr, level - 1) + (j - wh.here_off);  // absolute node number on the child level
    const uint32_t W = 1u << dr.bit_width;
    const uint64_t p = q >> dr.bit_width;
    const uint32_t k = uint32_t(q) & (W - 1u);
    const DenseNode e = cur[wh.upper_off + uint32_t(p - dense_first(dr, level))];
    DenseNode c{0, e.base + uint64_t(k) * amt_span(dr.bit_width, level), kNoBlock, 0, e.seq, 1};
    if (e.block == kNoBlock) {
        next[j] = c;
        return;
    }
    const uint64_t remaining = dense_total(dr, level - 1) - p * W;
    const uint32_t m = remaining < W ? uint32_t(remaining) : W;  // links this parent must hold
    // a parent on the edge of the range has children without a lane: the first / last lane it does have stands in
    const uint64_t n_here = dense_nodes(dr, level - 1);
    const bool edge_first = j == wh.here_off && k > 0;
    const bool edge_last = j + 1 == wh.here_off + n_here && k + 1 < m;
    // The parent is validated BY ITS CHILDREN, each lane a share: every lane the header, lane k link k (at header +
    // 43·k: right iff links 0..k-1 are standard 43-byte links, which lanes 0..k-1 say of theirs), lane m-1 what follows
    // the last link (an empty values array and, in a child block, the end of the block).  Together: everything
    // CollapsedNode::expand checks.
    uint64_t h0, h1;
    const uint32_t hdr = dense_header(W, m, false, h0, h1);
    const uint8_t* node = w.arena + e.goff;
    const uint32_t mine = hdr + 43u * k;
    bool ok = hdr != 0 && mine + 43u <= e.rem && header_matches(node, hdr, h0, h1) && std_link_at(node + mine);
    if (ok && edge_first)
        for (uint32_t i = 0; i < k && ok; ++i) ok = std_link_at(node + hdr + 43u * i);
    if (ok && edge_last)
        for (uint32_t i = k + 1; i < m && ok; ++i) ok = hdr + 43u * (i + 1u) <= e.rem && std_link_at(node + hdr + 43u * i);
    if (ok && (k == m - 1 || edge_last)) {
        const uint32_t tail = hdr + 43u * m;  // the values array: empty
        ok = tail + 1u <= e.rem && node[tail] == 0x80u && (!e.whole || tail + 1u == e.rem);
    }
    if (!ok) {
        atomicOr(anomaly, 1u);
        next[j] = c;
        return;
    }
    const CidKey key = std_link_key(node + mine);
    // Blockstore::get: hash slot, then the candidate's CID, offset and length side by side
    uint32_t s = cid_hash(key) & w.mask;
    for (;;) {
        const uint32_t b = w.slots[s];
        if (b == kNoBlock) {
            atomicOr(anomaly, 1u);  // a missing block: the general path names it
            break;
        }
        const CidKey have = load_cid_slot(w.cids, b);
        const uint64_t off = w.off[b];
        const uint32_t len = w.len[b];
        if (cid_equal(have, key)) {
            if (w.touched) atomicOr(&w.touched[b >> 5], 1u << (b & 31));
            c.block = b;
            c.goff = off;
            c.rem = len;
            break;
        }
        s = (s + 1) & w.mask;
    }
    next[j] = c;
}

__global__ __launch_bounds__(256) void k_dense_level(WitnessView w, const DenseNode* __restrict__ cur, const DenseRoots roots_arg,
                                                     uint32_t level, uint32_t n_next, DenseNode* __restrict__ next,
                                                     uint32_t* __restrict__ anomaly) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < n_next) dense_level_one(w, cur, roots_arg.r, level, j, next, anomaly);
}

// The NARROW interior levels at the top of the trees (a few hundred entries: 14, 71 and 495 for the 1M-receipt tipset) in ONE
// launch of ONE workgroup, a __syncthreads between levels — the frontier of a level is written and read back by this
// workgroup alone, so workgroup-scope ordering is all it needs; no grid barrier (the persistent all-levels form lost to its
// barriers: launch_dense_walk).  What a level of so few entries costs beside K1 and the event parse is not its three
// dependent loads but getting a workgroup ONTO the chip (k_enum_roots: one wavefront, 29 loads, 40-55 µs in the step): here
// that is paid once for the levels together.  `n[i]`: entries of the frontier entering level_hi - 1 - i.
struct DenseTopCounts {
    uint32_t n[8];
};
__global__ __launch_bounds__(256) void k_dense_top(WitnessView w, const DenseNode* frontier, const DenseRoots roots_arg, uint32_t level_hi,
                                                   uint32_t n_levels, DenseTopCounts counts, DenseNode* a, DenseNode* b,
                                                   uint32_t* __restrict__ anomaly) {
    const DenseNode* src = frontier;
    for (uint32_t i = 0; i < n_levels; ++i) {
        const uint32_t n_next = counts.n[i];
        for (uint32_t j = threadIdx.x; j < n_next; j += blockDim.x) dense_level_one(w, src, roots_arg.r, level_hi - i, j, a, anomaly);
        __syncthreads();  // (the level's entries are in place for every wavefront of the workgroup)
        src = a;
        DenseNode* t = a;
        a = b;
        b = t;
    }
}

// leaf level of the trees whose values are LINKS taken as witness keys (Amtv0<Cid>: the message lists) — one lane per
// VALUE.  The leaf is validated by its values' lanes exactly as an interior node by its children's.
__global__ __launch_bounds__(256) void k_dense_link_leaves(WitnessView w, const DenseNode* __restrict__ cur, const DenseRoots roots_arg,
                                                           uint32_t n_key_roots, uint64_t n_values, uint32_t* __restrict__ anomaly,
                                                           CidKey* __restrict__ keys_main, DenseClear clear) {
    const uint64_t v = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    // (a passenger: the execution order's hash table and first-occurrence flags, which the kernels right behind this one
    // fill, are cleared by this grid instead of by two fill kernels of their own in front of them)
    {
        const uint64_t stride = uint64_t(gridDim.x) * blockDim.x;
        for (uint64_t j = v; j < clear.n_slots; j += stride) clear.slots[j] = ~0ull;
        for (uint64_t j = v; j < clear.n_words; j += stride) clear.words[j] = 0u;
    }
    if (v >= n_values) return;
    // which root (the key trees come first among the roots), which value
    const DenseRoot* roots = roots_arg.r;
    uint32_t r = 0, node_off0 = 0;
    uint64_t voff = 0;
    DenseRoot dr = roots[0];
    while (r + 1 < n_key_roots && v >= voff + (dr.hi - dr.lo)) {
        voff += dr.hi - dr.lo;
        node_off0 += uint32_t(dense_nodes(dr, 0));
        dr = roots[++r];
    }
    const uint64_t idx = dr.lo + (v - voff);
    const uint32_t W = 1u << dr.bit_width;
    const uint64_t pnode = idx >> dr.bit_width;
    const uint32_t k = uint32_t(idx) & (W - 1u);
    const DenseNode e = cur[node_off0 + uint32_t(pnode - dense_first(dr, 0))];
    if (e.block == kNoBlock) return;  // reported where the link failed to resolve
    const uint64_t remaining = dr.count - pnode * W;
    const uint32_t m = remaining < W ? uint32_t(remaining) : W;
    const bool edge_first = idx == dr.lo && k > 0;
    const bool edge_last = idx + 1 == dr.hi && k + 1 < m;
    uint64_t h0, h1;
    const uint32_t hdr = dense_header(W, m, true, h0, h1);
    const uint8_t* node = w.arena + e.goff;
    const uint32_t mine = hdr + 43u * k;
    bool ok = hdr != 0 && mine + 43u <= e.rem && header_matches(node, hdr, h0, h1) && std_link_at(node + mine);
    if (ok && edge_first)
        for (uint32_t i = 0; i < k && ok; ++i) ok = std_link_at(node + hdr + 43u * i);
    if (ok && edge_last)
        for (uint32_t i = k + 1; i < m && ok; ++i) ok = hdr + 43u * (i + 1u) <= e.rem && std_link_at(node + hdr + 43u * i);
    if (ok && (k == m - 1 || edge_last)) ok = !e.whole || hdr + 43u * m == e.rem;  // nothing after the last value
    if (!ok) {
        atomicOr(anomaly, 1u);
        return;
    }
    keys_main[dr.out_off + (idx - dr.lo)] = std_link_key(node + mine);
}

// leaf level of every other tree: one lane per leaf node validates it in one pass and writes its values' locations
__global__ __launch_bounds__(256) void k_dense_leaves(WitnessView w, const DenseNode* __restrict__ cur, const DenseRoots roots_arg,
                                                      uint32_t n_nodes, uint32_t first_node, LeafRef* __restrict__ leaves_main,
                                                      uint32_t* __restrict__ anomaly, LeafRef* __restrict__ leaves_extra) {
    const uint32_t t = first_node + blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_nodes) return;
    const DenseWhere wh = dense_where(roots_arg.r, t, 0);
    const DenseRoot& dr = wh.dr;
    if (dr.out_sel == 0 && dr.count != 0) return;  // (a key tree with values: k_dense_link_leaves validates its leaves)
    const uint64_t leaf_off = dr.out_off;
    const int vkind = int(dr.vkind);
    LeafRef* const leaves = dr.out_sel == 1 ? leaves_main : (dr.out_sel == 2 ? leaves_extra : nullptr);
    const uint64_t p = dense_first(dr, 0) + (t - wh.here_off);  // absolute leaf-node number
    const DenseNode e = cur[t];
    const uint32_t W = 1u << dr.bit_width;
    const uint64_t remaining = dr.count - p * W;
    const uint32_t m = remaining < W ? uint32_t(remaining) : W;
    // Whatever happens to this node, every LeafRef it is responsible for is WRITTEN (kNoBlock when there is no value): the
    // kernels queued behind — before the host has seen the anomaly flag — read them (host/verify_fast.cpp).
    auto blank = [&]() {
        if (!leaves) return;
        for (uint32_t i = 0; i < m; ++i) {
            const uint64_t idx = p * W + i;
            if (idx >= dr.lo && idx < dr.hi) leaves[leaf_off + (idx - dr.lo)] = LeafRef{kNoBlock, 0, 0, e.seq, idx};
        }
    };
    if (e.block == kNoBlock) {  // reported where the link failed to resolve
        blank();
        return;
    }
    const uint32_t node_off = uint32_t(e.goff - w.off[e.block]);  // (LeafRefs are relative to the block)
    Rd rd;
    rd.init(w.arena + e.goff, e.rem);
    rd.expect_array(3);
    uint32_t bo, bl;
    rd.read_bytes(bo, bl);
    bool ok = rd.ok() && bl == (W + 7) / 8;
    if (ok) {  // bitmap = the low m bits (bits at or above the width do not count when W < 8)
        for (uint32_t i = 0; i < bl; ++i) {
            const uint32_t lo = i * 8u;
            uint32_t want = m >= lo + 8 ? 0xffu : (m > lo ? (1u << (m - lo)) - 1u : 0u);
            uint32_t have = rd.at(bo + i);
            if (W < 8) have &= (1u << W) - 1u;
            ok = ok && have == want;
        }
    }
    ok = ok && rd.read_array() == 0;  // a leaf holds no links
    const uint64_t nv = rd.read_array();
    ok = ok && rd.ok() && nv == m;
    // every value of the node is type-checked (serde decodes the whole node); the ones inside [lo, hi) are emitted
    for (uint32_t i = 0; ok && i < m; ++i) {
        const uint32_t start = rd.pos;
        const uint64_t idx = p * W + i;
        check_value(rd, vkind);
        ok = rd.ok();
        if (ok && leaves && idx >= dr.lo && idx < dr.hi)
            leaves[leaf_off + (idx - dr.lo)] = LeafRef{e.block, node_off + start, rd.pos - start, e.seq, e.base + i};
    }
    if (ok && e.whole) {
        rd.finish();
        ok = rd.ok();
    }
    if (!ok) {
        atomicOr(anomaly, 1u);
        blank();
    }
}

__global__ void k_enum_check(const uint64_t* __restrict__ actual, uint64_t expected, uint32_t* __restrict__ mismatch) {
    if (threadIdx.x == 0 && blockIdx.x == 0 && *actual != expected) atomicOr(mismatch, 1u);
}

int launch_scan_u32(ipcfp_ctx* ctx, const uint32_t* in_d, uint32_t n, uint32_t* out_d, uint64_t* total_d,
                    uint64_t* scratch_d);

// Entries of the frontier that ENTERS `level`, predicted from the roots' (height, bit width, count)
// under the assumption that every AMT is dense (indices 0..count-1): what fvm_ipld_amt writes for
// message lists, receipts and events.  A sparse or lying root only makes the prediction wrong, which the
// device detects (k_enum_check) and the caller answers by walking level by level.
static uint64_t predicted_frontier(const std::vector<uint64_t>& info, uint32_t level, uint32_t max_height) {
    if (level == max_height) return info.size() / 2;
    uint64_t total = 0;
    for (size_t i = 0; i + 1 < info.size(); i += 2) {
        if (info[i] == ~0ULL) continue;  // dead root
        const uint32_t h = uint32_t(info[i]), bw = uint32_t(info[i] >> 32);
        const uint64_t c = info[i + 1];
        if (h <= level || c == 0) {
            total += 1;  // rides along / is the root itself / an empty root carried down
        } else {
            const uint64_t shift = uint64_t(bw) * (level + 1);
            total += shift >= 64 ? 1 : ((c + (1ULL << shift) - 1) >> shift);
        }
    }
    return total;
}

static uint64_t amt_span_host(uint32_t bw, uint64_t height) {
    const uint64_t shift = uint64_t(bw) * height;
    return shift >= 64 ? ~0ULL : (1ULL << shift);
}

// The dense plan: what the roots' (height, bit width, count) say about every level (host side; amt_enum.h DensePlan).
// root_info: 2 words per root as k_enum_roots writes them ({height | bw << 32, count}; ~0 = a root that failed to load);
// the optional extra root is entry n_roots.
void dense_plan(const std::vector<uint64_t>& root_info, uint32_t n_roots, int vkind, bool want_keys, uint64_t lo, uint64_t hi,
                uint32_t has_extra, int extra_vkind, uint64_t extra_lo, uint64_t extra_hi, DensePlan& plan) {
    plan = DensePlan{};
    const uint32_t n_all = n_roots + (has_extra ? 1u : 0u);
    if (n_roots == 0 || n_all > kMaxDenseRoots || root_info.size() < 2 * size_t(n_all)) return;
    auto shape = [&](uint32_t i, uint64_t rlo, uint64_t rhi, DenseRoot& d) -> bool {  // false: not a dense candidate
        if (root_info[2 * size_t(i)] == ~0ULL) return false;  // a dead root: let the general path sort it out
        d.height = uint32_t(root_info[2 * size_t(i)]);
        d.bit_width = uint32_t(root_info[2 * size_t(i)] >> 32);
        d.count = root_info[2 * size_t(i) + 1];
        if (d.count == 0 && d.height > 0) return false;  // empty tree with a tall root
        if (d.count > amt_span_host(d.bit_width, d.height + 1)) return false;
        d.lo = rlo < d.count ? rlo : d.count;
        d.hi = rhi < d.count ? rhi : d.count;
        if (d.count && d.lo >= d.hi) return false;  // nothing of this tree in the range
        if (d.count == 0) d.lo = d.hi = 0;
        return true;
    };
    bool try_dense = true;
    uint64_t n_leaves = 0;
    for (uint32_t i = 0; i < n_roots && try_dense; ++i) {
        try_dense = shape(i, lo, hi, plan.roots.r[i]);
        plan.roots.r[i].vkind = uint32_t(vkind);
        plan.roots.r[i].out_sel = want_keys ? 0u : 1u;
        plan.roots.r[i].out_off = n_leaves;
        n_leaves += plan.roots.r[i].hi - plan.roots.r[i].lo;
    }
    // the extra root joins when it is a dense candidate itself; otherwise the call goes on without it
    uint32_t n_use = n_roots;
    uint64_t n_extra = 0;
    if (has_extra && try_dense && shape(n_roots, extra_lo, extra_hi, plan.roots.r[n_roots])) {
        plan.roots.r[n_roots].vkind = uint32_t(extra_vkind);
        plan.roots.r[n_roots].out_sel = 2u;
        plan.roots.r[n_roots].out_off = 0;
        n_extra = plan.roots.r[n_roots].hi - plan.roots.r[n_roots].lo;
        n_use = n_all;
    }
    if (!try_dense) return;
    uint32_t max_height = 0;  // the tallest of the roots in use
    for (uint32_t i = 0; i < n_use; ++i) max_height = plan.roots.r[i].height > max_height ? plan.roots.r[i].height : max_height;
    if (max_height >= kMaxDenseLevels) return;
    uint64_t biggest = n_use;
    for (uint32_t level = 0; level <= max_height; ++level) {
        uint64_t nl = 0;
        for (uint32_t i = 0; i < n_use; ++i) nl += dense_nodes(plan.roots.r[i], level);
        if (nl >= 0x7fffffffULL) return;
        plan.n_level[level] = nl;
        biggest = nl > biggest ? nl : biggest;
    }
    if (n_leaves >= 0x7fffffffULL || n_extra >= 0x7fffffffULL || plan.n_level[max_height] != n_use) return;
    plan.ok = true;
    plan.n_use = n_use;
    plan.roots.n = n_use;
    plan.max_height = max_height;
    plan.n_leaves = n_leaves;
    plan.n_extra = n_extra;
    plan.biggest = biggest;
}

// The launches of the dense walk: one kernel per interior level, one leaf kernel.  `frontier`: the roots' entries
// (k_enum_roots); a / b: two frontier buffers of plan.biggest entries.  The root shapes travel as a kernel argument.
int launch_dense_walk(ipcfp_ctx* ctx, const WitnessView& view, const DenseNode* frontier, const DensePlan& plan,
                      DenseNode* a, DenseNode* b, LeafRef* leaves_main, CidKey* keys_main, LeafRef* leaves_extra, uint32_t* anomaly_d,
                      hipStream_t leaves_stream, hipEvent_t fork_event, hipStream_t wide_stream, hipEvent_t wide_event,
                      uint32_t narrow_max_wg, const DenseClear* clear) {
    const DenseNode* src = frontier;
    auto widen = [&]() -> hipError_t {  // the narrow stream's part ends here
        if (!wide_stream || ctx->stream == wide_stream) return hipSuccess;
        hipError_t e = hipEventRecord(wide_event, ctx->stream);
        if (e == hipSuccess) e = hipStreamWaitEvent(wide_stream, wide_event, 0);
        ctx->stream = wide_stream;
        return e;
    };
    // (All interior levels as ONE persistent launch — a workgroup per CU, a grid barrier per level — was built and is
    // slower: 382 µs against 238 µs for the six launches.  What stretches a level beside K1 and the event parse is the
    // latency of its three dependent loads under their memory traffic, not the dispatch; a barrier adds an L2
    // write-back and an L1 invalidate per level on top.  profiles/r03_experiments.md)
    uint32_t level_from = plan.max_height;
    {   // the narrow levels at the top: one launch (k_dense_top).  IPCFP_DENSE_TOP = the widest level it takes (0: none)
        static const uint32_t top_max = [] {
            const char* e = std::getenv("IPCFP_DENSE_TOP");
            return e ? uint32_t(std::atoi(e)) : kDenseTopMax;
        }();
        DenseTopCounts counts{};
        uint32_t n_top = 0;
        while (n_top < 8 && level_from - n_top >= 1 && plan.n_level[level_from - n_top - 1] <= top_max) {
            counts.n[n_top] = uint32_t(plan.n_level[level_from - n_top - 1]);
            ++n_top;
        }
        if (n_top >= 2) {  // (one level alone is k_dense_level's)
            hipLaunchKernelGGL(k_dense_top, dim3(1), dim3(256), 0, ctx->stream, view, src, plan.roots, level_from, n_top, counts, a, b, anomaly_d);
            // the last level written: a for an odd number of levels, b for an even one — and the next writes the other
            if (n_top & 1u) {
                src = a;
                DenseNode* t = a;
                a = b;
                b = t;
            } else {
                src = b;
            }
            level_from -= n_top;
        }
    }
    for (uint32_t level = level_from; level >= 1; --level) {
        const uint32_t nn = uint32_t(plan.n_level[level - 1]);
        if (div_up(nn, 256) > narrow_max_wg) IPCFP_HIP(ctx, widen());
        hipLaunchKernelGGL(k_dense_level, dim3(div_up(nn, 256)), dim3(256), 0, ctx->stream, view, src, plan.roots, level, nn, a, anomaly_d);
        src = a;
        DenseNode* t = a;
        a = b;
        b = t;  // `src` now lives in b; the next level writes a
    }
    // the key trees (links taken as witness keys) come first among the roots: one lane per value; every other tree: one
    // lane per leaf node
    uint32_t n_key_roots = 0, key_nodes = 0;
    uint64_t n_key_values = 0;
    while (n_key_roots < plan.n_use && plan.roots.r[n_key_roots].out_sel == 0) {
        n_key_values += plan.roots.r[n_key_roots].hi - plan.roots.r[n_key_roots].lo;
        key_nodes += uint32_t(dense_nodes(plan.roots.r[n_key_roots], 0));
        ++n_key_roots;
    }
    for (uint32_t i = n_key_roots; i < plan.n_use; ++i)
        if (plan.roots.r[i].out_sel == 0) return set_error(ctx, IPCFP_E_INVALID, "dense walk: the key trees must come first");
    // The two leaf kernels read the same frontier and write different outputs.  A caller whose consumers sit on two
    // streams (host/verify_fast.cpp: the message keys feed the execution-order kernels on the main stream, the receipt
    // leaves feed k_receipt_events on the aux stream) forks here: k_dense_leaves goes to `leaves_stream`, ordered behind
    // the last interior level by `fork_event`, and runs beside k_dense_link_leaves instead of after it.
    IPCFP_HIP(ctx, widen());
    hipStream_t ls = ctx->stream;
    if (leaves_stream && leaves_stream != ctx->stream && fork_event) {
        IPCFP_HIP(ctx, hipEventRecord(fork_event, ctx->stream));
        IPCFP_HIP(ctx, hipStreamWaitEvent(leaves_stream, fork_event, 0));
        ls = leaves_stream;
    }
    // (every leaf node gets a lane here: those of key trees that have values leave at once, an EMPTY key tree's root is
    // validated here — no value lane looks at it)
    const uint32_t n_nodes = uint32_t(plan.n_level[0]);
    (void)key_nodes;
    hipLaunchKernelGGL(k_dense_leaves, dim3(div_up(n_nodes, 256)), dim3(256), 0, ls, view, src, plan.roots, n_nodes, 0u,
                       leaves_main, anomaly_d, leaves_extra);
    if (clear && !n_key_values) {  // (no lane to take the passenger)
        if (clear->n_slots) IPCFP_HIP(ctx, hipMemsetAsync(clear->slots, 0xff, clear->n_slots * 8, ctx->stream));
        if (clear->n_words) IPCFP_HIP(ctx, hipMemsetAsync(clear->words, 0, clear->n_words * 4, ctx->stream));
    }
    if (n_key_values)
        hipLaunchKernelGGL(k_dense_link_leaves, dim3(div_up(n_key_values, 256)), dim3(256), 0, ctx->stream, view, src, plan.roots,
                           n_key_roots, n_key_values, anomaly_d, keys_main, clear ? *clear : DenseClear{nullptr, 0, nullptr, 0});
    IPCFP_HIP(ctx, hipGetLastError());
    return IPCFP_OK;
}

int amt_enumerate(ipcfp_ctx* ctx, const WitnessView& view, const AmtRootSpec* roots_d, uint32_t n_roots, int vkind,
                  unsigned long long* err_d, AmtEnumResult& out, uint64_t lo, uint64_t hi, DevBuf<CidKey>* keys_out,
                  EnumExtra* extra) {
    out.keys_written = false;
    if (extra) {
        extra->done = false;
        if (!extra->out || n_roots == 0) extra = nullptr;
    }
    const uint32_t n_all = n_roots + (extra ? 1u : 0u);  // the extra root's spec is roots_d[n_roots]
    const EnumRange rg{lo, hi};
    const bool whole = lo == 0 && hi == ~0ULL;
    out.n_leaves = 0;
    out.error = kNoEnumError;
    out.leaves.release();
    if (n_roots == 0) return IPCFP_OK;
    ProfileScope prof(ctx, IPCFP_K_EXEC_ORDER);
    DevBuf<EnumNode> cur, nxt;
    DevBuf<uint32_t> counts, offsets, small_own;
    DevBuf<uint64_t> scratch, total_d, root_info_own;
    DevBuf<DenseNode> dense_cur;  // the roots again, as the dense walk wants them
    IPCFP_HIP(ctx, cur.alloc(n_all));
    IPCFP_HIP(ctx, dense_cur.alloc(n_all));
    IPCFP_HIP(ctx, total_d.alloc(2));
    uint32_t* small = nullptr;      // [0] = max height; [2] = anomaly flag of the dense path
    uint64_t* root_info_d = nullptr;
    IPCFP_HIP(ctx, ctl_words(ctx, small_own, small, 4, false));
    IPCFP_HIP(ctx, ctl_words(ctx, root_info_own, root_info_d, 2 * size_t(n_all), false));
    hipLaunchKernelGGL(k_enum_roots, dim3(div_up(n_all, 64)), dim3(64), 0, ctx->stream, view, roots_d, n_all, vkind,
                       cur.p, small, err_d, root_info_d, static_cast<unsigned long long*>(nullptr), 0ull, dense_cur.p);
    uint32_t max_height = 0;
    std::vector<uint64_t> root_info(2 * size_t(n_all));
    IPCFP_HIP(ctx, ctl_read(ctx, &max_height, small, 4));
    IPCFP_HIP(ctx, ctl_read(ctx, root_info.data(), root_info_d, root_info.size() * 8));
    IPCFP_HIP(ctx, sync_stream(ctx, ctx->stream));

    // ---- dense fast path: the tree's shape follows from the roots; one kernel per level ----
    out.dense = false;
    {
        const bool want_keys = keys_out && vkind == VK_CID;
        DensePlan plan;
        dense_plan(root_info, n_roots, vkind, want_keys, lo, hi, extra ? 1u : 0u, extra ? extra->vkind : 0, extra ? extra->lo : 0,
                   extra ? extra->hi : 0, plan);
        if (plan.ok) {
            const uint32_t n_use = plan.n_use;
            DevBuf<DenseNode> a, b;
            struct { uint32_t* p; } anomaly{small + 2};  // still zero: nothing has written it
            IPCFP_HIP(ctx, a.alloc(plan.biggest));
            IPCFP_HIP(ctx, b.alloc(plan.biggest));
            if (want_keys) IPCFP_HIP(ctx, keys_out->alloc(plan.n_leaves));
            else IPCFP_HIP(ctx, out.leaves.alloc(plan.n_leaves));
            if (n_use > n_roots) IPCFP_HIP(ctx, extra->out->leaves.alloc(plan.n_extra));
            int rc = launch_dense_walk(ctx, view, dense_cur.p, plan, a.p, b.p, want_keys ? nullptr : out.leaves.p,
                                       want_keys ? keys_out->p : nullptr, n_use > n_roots ? extra->out->leaves.p : nullptr, anomaly.p);
            if (rc) return rc;
            uint32_t bad = 0;
            unsigned long long e = kNoEnumError;  // err_d is untouched here: report what earlier stages left in it
            IPCFP_HIP(ctx, ctl_read(ctx, &bad, anomaly.p, 4));
            IPCFP_HIP(ctx, ctl_read(ctx, &e, err_d, 8));
            IPCFP_HIP(ctx, sync_stream(ctx, ctx->stream));
            IPCFP_HIP(ctx, hipGetLastError());
            if (!bad) {
                out.n_leaves = plan.n_leaves;
                out.error = e;
                out.dense = true;
                out.keys_written = want_keys;
                if (n_use > n_roots) {
                    extra->out->n_leaves = plan.n_extra;
                    extra->out->error = kNoEnumError;
                    extra->out->dense = true;
                    extra->out->keys_written = false;
                    extra->done = true;
                }
                return IPCFP_OK;
            }
            out.leaves.release();
            if (want_keys) keys_out->release();
            if (n_use > n_roots) extra->out->leaves.release();
            // the anomaly may be the extra root's: the general paths below walk the call's own roots only
        }
        // what follows knows nothing of the extra root
        max_height = 0;
        for (uint32_t i = 0; i < n_roots; ++i)
            if (root_info[2 * size_t(i)] != ~0ULL) {
                const uint32_t h = uint32_t(root_info[2 * size_t(i)]);
                max_height = h > max_height ? h : max_height;
            }
        root_info.resize(2 * size_t(n_roots));
    }

    // ---- speculative pass: every level launched back to back with PREDICTED sizes, one sync at the end ----
    if (whole) {
        uint64_t pred_leaves = 0;
        bool sane = true;
        for (size_t i = 0; i + 1 < root_info.size(); i += 2)
            if (root_info[i] != ~0ULL) pred_leaves += root_info[i + 1];
        std::vector<uint64_t> pred(max_height + 1);
        for (uint32_t level = 0; level <= max_height; ++level) {
            pred[level] = predicted_frontier(root_info, level, max_height);
            sane = sane && pred[level] < 0x7fffffffULL;
        }
        sane = sane && pred_leaves < 0x7fffffffULL;
        if (sane) {
            DevBuf<EnumNode> a, b;
            DevBuf<uint32_t> cnt, offs, mismatch;
            DevBuf<uint64_t> scr, totals;
            uint64_t biggest = n_roots;
            for (auto v : pred) biggest = v > biggest ? v : biggest;
            IPCFP_HIP(ctx, a.alloc(biggest));
            IPCFP_HIP(ctx, b.alloc(biggest));
            IPCFP_HIP(ctx, cnt.alloc(biggest));
            IPCFP_HIP(ctx, offs.alloc(biggest));
            IPCFP_HIP(ctx, scr.alloc(size_t(div_up(biggest, 1024)) + 2));
            IPCFP_HIP(ctx, totals.alloc(max_height + 3));
            IPCFP_HIP(ctx, mismatch.alloc(1));
            IPCFP_HIP(ctx, hipMemsetAsync(mismatch.p, 0, 4, ctx->stream));
            IPCFP_HIP(ctx, hipMemcpyAsync(a.p, cur.p, size_t(n_roots) * sizeof(EnumNode), hipMemcpyDeviceToDevice, ctx->stream));
            IPCFP_HIP(ctx, out.leaves.alloc(pred_leaves));
            // err_d may already hold earlier-stage errors; the speculative pass works on a copy
            DevBuf<unsigned long long> err_spec;
            IPCFP_HIP(ctx, err_spec.alloc(1));
            IPCFP_HIP(ctx, hipMemcpyAsync(err_spec.p, err_d, 8, hipMemcpyDeviceToDevice, ctx->stream));
            const uint64_t* bound = nullptr;  // the top frontier (the roots) is exact
            uint32_t slot = 0;
            int rc = IPCFP_OK;
            for (uint32_t level = max_height;; --level, ++slot) {
                const uint32_t np = uint32_t(pred[level]);
                uint64_t* tot = totals.p + slot;
                if (np == 0) {
                    IPCFP_HIP(ctx, hipMemsetAsync(tot, 0, 8, ctx->stream));
                } else if (level >= 1) {
                    hipLaunchKernelGGL(k_enum_count, dim3(div_up(np, 256)), dim3(256), 0, ctx->stream, view, a.p, np, level,
                                       vkind, cnt.p, err_spec.p, bound, rg);
                    rc = launch_scan_u32(ctx, cnt.p, np, offs.p, tot, scr.p);
                } else {
                    hipLaunchKernelGGL(k_enum_count_leaf, dim3(div_up(np, 256)), dim3(256), 0, ctx->stream, view, a.p, np,
                                       vkind, cnt.p, err_spec.p, bound, rg);
                    rc = launch_scan_u32(ctx, cnt.p, np, offs.p, tot, scr.p);
                }
                if (rc) return rc;
                const uint64_t expect = level >= 1 ? pred[level - 1] : pred_leaves;
                hipLaunchKernelGGL(k_enum_check, dim3(1), dim3(64), 0, ctx->stream, tot, expect, mismatch.p);
                if (level >= 1) {
                    if (expect && np)
                        hipLaunchKernelGGL(k_enum_expand, dim3(div_up(expect, 256)), dim3(256), 0, ctx->stream, view, a.p, np,
                                           level, offs.p, uint32_t(expect), b.p, err_spec.p, tot, rg);
                    a.swap(b);
                    bound = tot;
                } else {
                    if (pred_leaves && np)
                        hipLaunchKernelGGL(k_enum_emit, dim3(div_up(np, 256)), dim3(256), 0, ctx->stream, view, a.p, np, vkind,
                                           cnt.p, offs.p, out.leaves.p, pred_leaves, rg);
                    break;
                }
            }
            uint32_t bad = 0;
            unsigned long long e = kNoEnumError;
            IPCFP_HIP(ctx, d2h_small(ctx, &bad, mismatch.p, 4, ctx->stream));
            IPCFP_HIP(ctx, d2h_small(ctx, &e, err_spec.p, 8, ctx->stream));
            IPCFP_HIP(ctx, sync_stream(ctx, ctx->stream));
            IPCFP_HIP(ctx, hipGetLastError());
            if (!bad) {
                IPCFP_HIP(ctx, hipMemcpyAsync(err_d, err_spec.p, 8, hipMemcpyDeviceToDevice, ctx->stream));
                out.n_leaves = pred_leaves;
                out.error = e;
                return IPCFP_OK;
            }
            out.leaves.release();  // prediction failed (sparse AMT, decode error, lying count): walk level by level
        }
    }
    uint32_t n = n_roots;
    for (uint32_t level = max_height;; --level) {
        IPCFP_HIP(ctx, counts.alloc(n));
        IPCFP_HIP(ctx, offsets.alloc(n));
        IPCFP_HIP(ctx, scratch.alloc(size_t(div_up(n, 1024)) + 2));
        if (level >= 1)
            hipLaunchKernelGGL(k_enum_count, dim3(div_up(n, 256)), dim3(256), 0, ctx->stream, view, cur.p, n, level,
                               vkind, counts.p, err_d, static_cast<const uint64_t*>(nullptr), rg);
        else
            hipLaunchKernelGGL(k_enum_count_leaf, dim3(div_up(n, 256)), dim3(256), 0, ctx->stream, view, cur.p, n,
                               vkind, counts.p, err_d, static_cast<const uint64_t*>(nullptr), rg);
        int rc = launch_scan_u32(ctx, counts.p, n, offsets.p, total_d.p, scratch.p);
        if (rc) return rc;
        uint64_t total = 0;
        IPCFP_HIP(ctx, d2h_small(ctx, &total, total_d.p, 8, ctx->stream));
        IPCFP_HIP(ctx, sync_stream(ctx, ctx->stream));
        if (total >= 0x7fffffffULL)
            return set_error(ctx, IPCFP_E_UNSUPPORTED, "AMT enumeration expands to %llu entries",
                             (unsigned long long)total);
        if (level >= 1) {
            IPCFP_HIP(ctx, nxt.alloc(total));
            if (total)
                hipLaunchKernelGGL(k_enum_expand, dim3(div_up(total, 256)), dim3(256), 0, ctx->stream, view, cur.p, n, level,
                                   offsets.p, uint32_t(total), nxt.p, err_d, static_cast<const uint64_t*>(nullptr), rg);
            cur.swap(nxt);  // the old frontier returns to the pool; reuse is stream-ordered
            n = uint32_t(total);
            if (n == 0) break;
        } else {
            IPCFP_HIP(ctx, out.leaves.alloc(total));
            out.n_leaves = total;
            if (total)
                hipLaunchKernelGGL(k_enum_emit, dim3(div_up(n, 256)), dim3(256), 0, ctx->stream, view, cur.p, n, vkind,
                                   counts.p, offsets.p, out.leaves.p, total, rg);
            break;
        }
    }
    unsigned long long e = kNoEnumError;
    IPCFP_HIP(ctx, d2h_small(ctx, &e, err_d, 8, ctx->stream));
    IPCFP_HIP(ctx, sync_stream(ctx, ctx->stream));
    IPCFP_HIP(ctx, hipGetLastError());
    out.error = e;
    return IPCFP_OK;
}

__global__ __launch_bounds__(256) void k_check_dense(const LeafRef* __restrict__ leaves, uint32_t n, uint64_t first,
                                                     uint32_t* __restrict__ flag) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const bool bad = i < n && leaves[i].index != first + uint64_t(i);
    if (__any(bad) && (threadIdx.x & 63) == 0) atomicOr(flag, 1u);
}

// move an enumeration result into a cache entry (the density check runs only for results of the general paths)
static int enum_cache_fill(ipcfp_ctx* ctx, EnumCached* e, AmtEnumResult& en, uint64_t lo, uint32_t* flag_p) {
    e->n = en.n_leaves;
    e->error = en.error;
    uint32_t not_dense = 0;
    if (en.n_leaves && !en.dense) {
        const uint32_t n = uint32_t(en.n_leaves);
        hipLaunchKernelGGL(k_check_dense, dim3(div_up(n, 256)), dim3(256), 0, ctx->stream, en.leaves.p, n, lo, flag_p);
        IPCFP_HIP(ctx, ctl_read(ctx, &not_dense, flag_p, 4));
        IPCFP_HIP(ctx, sync_stream(ctx, ctx->stream));
    }
    e->dense = !not_dense;
    // move the leaves into the cache entry (byte-typed buffer)
    e->leaves.p = reinterpret_cast<uint8_t*>(en.leaves.p);
    e->leaves.count = en.leaves.count * sizeof(LeafRef);
    e->leaves.cap = en.leaves.cap;
    e->leaves.owner = en.leaves.owner;
    en.leaves.p = nullptr;
    en.leaves.count = en.leaves.cap = 0;
    en.leaves.owner = nullptr;
    return IPCFP_OK;
}

// k_enum_roots for a caller that queues the walk itself (host/verify_fast.cpp): the shapes also go to `mailbox` (pinned host)
int launch_enum_roots(ipcfp_ctx* ctx, const WitnessView& view, const AmtRootSpec* roots_d, uint32_t n_all, int vkind,
                      EnumNode* frontier_d, uint32_t* max_height_d, unsigned long long* err_d, uint64_t* root_info_d,
                      unsigned long long* mailbox, unsigned long long mailbox_seq, DenseNode* dense_frontier_d) {
    if (n_all == 0 || n_all > 128) return set_error(ctx, IPCFP_E_INVALID, "launch_enum_roots: %u roots", n_all);
    hipLaunchKernelGGL(k_enum_roots, dim3(1), dim3(n_all <= 64 ? 64 : 128), 0, ctx->stream, view, roots_d, n_all, vkind, frontier_d, max_height_d, err_d,
                       root_info_d, mailbox, mailbox_seq, dense_frontier_d);
    IPCFP_HIP(ctx, hipGetLastError());
    return IPCFP_OK;
}

// the TxMeta re-hashes the tipset prologue left behind (tipset_ctx.h txmeta_block), on `stream`
int launch_txmeta_rehash(ipcfp_ctx* ctx, hipStream_t stream, const WitnessView& view, const TipsetCtxDev* ctx_d, unsigned long long* err_d) {
    hipLaunchKernelGGL(k_txmeta_rehash, dim3(1), dim3(64), 0, stream, view, ctx_d, err_d);
    IPCFP_HIP(ctx, hipGetLastError());
    return IPCFP_OK;
}

int amt_enumerate_cached(ipcfp_ctx* ctx, ipcfp_witness* w, const CidKey& root, int version, int vkind,
                         const EnumCached** out, uint64_t lo, uint64_t hi) {
    for (auto& e : w->enum_cache)
        if (e->version == version && e->vkind == vkind && e->lo == lo && e->hi == hi &&
            std::memcmp(e->root, root.w, 40) == 0) {
            *out = e.get();
            return IPCFP_OK;
        }
    std::unique_ptr<EnumCached> e(new EnumCached());
    std::memcpy(e->root, root.w, 40);
    e->version = version;
    e->vkind = vkind;
    e->lo = lo;
    e->hi = hi;
    const WitnessView view = witness_view(w);
    DevBuf<AmtRootSpec> roots;
    DevBuf<unsigned long long> err_own;
    DevBuf<uint32_t> flag_own;
    unsigned long long* err_p = nullptr;  // kNoEnumError
    uint32_t* flag_p = nullptr;
    IPCFP_HIP(ctx, roots.alloc(1));
    IPCFP_HIP(ctx, ctl_words(ctx, err_own, err_p, 1, true));
    IPCFP_HIP(ctx, ctl_words(ctx, flag_own, flag_p, 1, false));
    struct { unsigned long long* p; } err{err_p};
    struct { uint32_t* p; } flag{flag_p};
    AmtRootSpec spec{};
    spec.root = root;
    spec.version = uint32_t(version);
    IPCFP_HIP(ctx, h2d_small(ctx, roots.p, &spec, sizeof spec, ctx->stream));
    AmtEnumResult en;
    int rc = amt_enumerate(ctx, view, roots.p, 1, vkind, err.p, en, lo, hi);
    if (rc) return rc;
    rc = enum_cache_fill(ctx, e.get(), en, lo, flag.p);
    if (rc) return rc;
    *out = e.get();
    w->enum_cache.push_back(std::move(e));
    return IPCFP_OK;
}

int enum_cache_put(ipcfp_ctx* ctx, ipcfp_witness* w, const CidKey& root, int version, int vkind, uint64_t lo, uint64_t hi,
                   AmtEnumResult& en) {
    std::unique_ptr<EnumCached> e(new EnumCached());
    std::memcpy(e->root, root.w, 40);
    e->version = version;
    e->vkind = vkind;
    e->lo = lo;
    e->hi = hi;
    DevBuf<uint32_t> flag_own;
    uint32_t* flag_p = nullptr;
    IPCFP_HIP(ctx, ctl_words(ctx, flag_own, flag_p, 1, false));
    int rc = enum_cache_fill(ctx, e.get(), en, lo, flag_p);
    if (rc) return rc;
    w->enum_cache.push_back(std::move(e));
    return IPCFP_OK;
}

}  // namespace ipcfp
