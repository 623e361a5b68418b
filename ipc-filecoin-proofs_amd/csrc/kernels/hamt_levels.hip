// csrc/kernels/hamt_levels.hip — K7 for a batch, LEVEL BY LEVEL: every node the batch visits is decoded ONCE.
//
// `Hamt::get` decodes each node on a key's path completely (serde decodes the whole `[bitfield, [pointer…]]`:
// fvm_ipld_hamt 0.10.4; src/proofs/common/decode.rs:29-39, src/proofs/storage/decode.rs:79-96), and the reference does
// so per lookup: 66 k actor gets in a 4 M-actor state tree decode ≈ 330 k nodes of which ≈ 95 k are distinct — the root
// 66 k times.  One query per lane walking by itself (walk.hip k_hamt_get) repeats exactly that: 8.5× the algorithmic
// bytes fetched, 191 k instructions per wavefront (profiles/r03_hamt_storage_pmc.txt).  A node's decode is a pure
// function of the block, so here the batch advances one tree level per step:
//
//   k_hamt_lv_start    lane = query: SHA-256 of the key (kept: 32 B per query), current node = the root block
//   k_hamt_lv_parse    lane = one node of this level's WORK LIST (the distinct blocks its queries stand on): the whole
//                      node is validated for the HAMT's value type and left as a HamtNodeRec (hamt_table.h) in a table
//                      indexed by block id — or as "not tabulated", which decides nothing (see below)
//   k_hamt_lv_advance  lane = query: take the level's hash bits, bitfield popcount, the pointer's offset from the record;
//                      a link → CID → block id (index probe) → CLAIM it for the next level's work list (a bitmap over the
//                      blocks: the first claimant appends it); a bucket → compare the ≤ 3 keys → settled
//
// All queries of a level have consumed the same number of hash bits, so a level is uniform.  The host queues a fixed
// number of levels (from the witness size) with no synchronisation; whatever is still unsettled afterwards — a deeper
// tree, a node the table does not cover (> 32 pointers, a bitfield over 64 bits, offsets beyond 64 KB, ANY decode
// problem) — is left kStPending and the per-query walker (k_hamt_get, pending_only) takes it from the root, so an
// outcome never depends on this path: it only ever answers what the walk would answer.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>

#include "../common.h"
#include "launch.h"
#include "hamt_outline.h"
#include "walk_dev.h"

namespace ipcfp {

struct HamtLevels {
    uint32_t* cur;        // n          the block each query stands on (kNoBlock: settled, or left to the walker)
    uint32_t* hash;       // n × 8      SHA-256 of the key, eight big-endian words
    HamtNodeRec* recs;    // n_blocks   indexed by block id; valid where `claimed` says so and the level has been parsed
    uint32_t* claimed;    // ⌈n_blocks / 32⌉ bits: the block is (or was) on a work list
    // Work lists of even / odd levels (capacity of each: min(n, n_blocks)), in TWO size classes when `split`: class 0 the
    // nodes of the small-stage parse instance (len + 24 ≤ kCoopSmallStage), class 1 everything longer.  Round 5 kept one list
    // per level and let both parse instances walk all of it, each skipping the other's nodes: at the bucket level the small
    // instance read 29.5 k entries and lengths to find nothing, at the overflow level the big one did (≈ 60 µs of a 520 µs
    // call: profiles/r05_experiments.md).  The claimant knows the block, so it files it where its parser will look.
    // An entry is {block, length, arena offset}: whoever files a block knows it (the claimant has just probed the index for
    // it) and reads its length for the size class anyway — so the parse's wavefront starts staging after ONE dependent read
    // (its list entry) instead of three (entry → length / offset → bytes): a level's parse is a few thousand wavefronts whose
    // time is that chain.
    uint4* work[2][3];    // [level parity][size class]; .x block, .y length, .z / .w arena offset (low / high word)
    const uint32_t* plain;  // plain_list: the one list, block ids only
    uint32_t* count;      // entries of level l's lists: count[3 l + class] (kHamtClasses counters per level)
    uint32_t split;       // 1: lists by size class (the 32-lane parse of the state tree); 0: everything in class 0
    uint32_t plain_list;  // 1: ONE list work[0][0] / count[0] whatever the level and class (launch_hamt_outline_list)
    uint32_t cap;         // capacity of every list
    uint32_t* top_overflow;  // set when a fused top level's list was full (k_hamt_lv_advance_top then leaves everything to the walker)
    HamtEntryTab* etabs;  // entry tables of the visited nodes that hold entries (`etab_cap` of them; null: none kept)
    uint32_t* etab_of;    // n_blocks: 1 + the node's table, 0: none (written for every node the 32-lane parse takes)
    uint32_t etab_cap;
    uint32_t* child;      // n_blocks × 32: the block behind pointer p where p is a standard link — filled by the 32-lane parse
                          // (HamtNodeRec::pad bit 0), so that a query's step down is one word instead of link bytes → index
                          // probe → CID compare (three dependent random reads of the eight a step was)
};

// Blockstore::get → block id WITHOUT recording the read (the parse resolves every link of a node; which of them a query
// follows — and so which block the RecordingBlockStore would have seen — is decided by k_hamt_lv_advance)
__device__ __forceinline__ uint32_t witness_find_quiet(const WitnessView& w, const CidKey& key) {
    uint32_t s = cid_hash(key) & w.mask;
    for (;;) {
        const uint32_t b = w.slots[s];
        if (b == kNoBlock) return kNoBlock;
        if (cid_equal(load_cid_slot(w.cids, b), key)) return b;
        s = (s + 1) & w.mask;
    }
}

// EVERY lane of the workgroup calls this (`want`: the lane has a block to claim for the level's work list).  All queries
// of the upper levels stand on a handful of nodes — level 0's 66 k claims are 32 distinct blocks, i.e. 66 k atomics on the
// same few words of the L2, and every appended block one more atomic on the level's ONE counter — so claims are thinned
// out on the way: a workgroup-wide set in LDS lets one lane per distinct block through, and the lanes of a wavefront that
// win their block append with one counter update between them.
constexpr uint32_t kCoopBigStage = 6912, kCoopBigEntries = 96, kCoopSmallStage = 1536, kCoopSmallEntries = 24;
// … and a THIRD size class between them (round 6): nine of ten bucket nodes of a 4 M-actor tree are 2-4 KB, and an instance
// whose stage is sized for them keeps 14 wavefronts on a CU where the 6.9 KB one keeps 9.  Round 5 measured such an instance
// beside the big one on two streams and dropped it (158 vs 173 µs, the extra launches and the fork cost more); with a list
// per class and count-driven launches an instance costs its own nodes and ≈ 4 µs when it has none.
constexpr uint32_t kCoopMidStage = 4352, kCoopMidEntries = 64;
constexpr uint32_t kHamtClasses = 3;
__device__ __forceinline__ uint32_t hamt_class_of_len(uint32_t len) {
    return len + 24u <= kCoopSmallStage ? 0u : (len + 24u <= kCoopMidStage ? 1u : 2u);
}
__device__ __forceinline__ uint32_t hamt_size_class(const WitnessView& w, const HamtLevels& L, uint32_t block) {
    return L.split ? hamt_class_of_len(w.len[block]) : 0u;
}
__device__ __forceinline__ uint4 hamt_work_entry(const WitnessView& w, uint32_t block, uint32_t len) {
    const uint64_t off = w.off[block];
    return make_uint4(block, len, uint32_t(off), uint32_t(off >> 32));
}
constexpr uint32_t kClaimSet = 512;  // LDS slots per 256-thread workgroup (a power of two ≥ 2 × the workgroup)
__device__ __forceinline__ void hamt_claim(const WitnessView& w, const HamtLevels& L, uint32_t block, uint32_t level, bool want, uint32_t* set) {
    for (uint32_t i = threadIdx.x; i < kClaimSet; i += blockDim.x) set[i] = kNoBlock;
    __syncthreads();
    bool first = false;
    if (want) {
        uint32_t s = (block * 2654435761u) >> 23;  // 9 bits
        for (;;) {
            const uint32_t seen = atomicCAS(&set[s], kNoBlock, block);
            if (seen == kNoBlock) {
                first = true;
                break;
            }
            if (seen == block) break;
            s = (s + 1u) & (kClaimSet - 1u);
        }
    }
    bool won = false;
    if (first) {
        const uint32_t bit = 1u << (block & 31u);
        uint32_t* word = L.claimed + (block >> 5);
        // (a stale miss of the plain look only costs the atomic)
        won = !(__builtin_nontemporal_load(word) & bit) && !(atomicOr(word, bit) & bit);
    }
    const uint32_t blen = won ? w.len[block] : 0u;
    const uint32_t cls = won && L.split ? hamt_class_of_len(blen) : 0u;
    const uint32_t lane = threadIdx.x & 63u;
#pragma unroll
    for (uint32_t c = 0; c < kHamtClasses; ++c) {  // (one counter update per wavefront and class)
        const uint64_t winners = __ballot(won && cls == c);
        if (!winners) continue;
        const uint32_t leader = uint32_t(__ffsll((long long)winners)) - 1u;
        uint32_t base = 0;
        if (lane == leader) base = atomicAdd(L.count + kHamtClasses * level + c, uint32_t(__popcll(winners)));
        base = __shfl(base, leader, 64);
        if (won && cls == c) L.work[level & 1u][c][base + uint32_t(__popcll(winners & ((1ull << lane) - 1ull)))] = hamt_work_entry(w, block, blen);
    }
}

__global__ __launch_bounds__(256) void k_hamt_lv_start(WitnessView w, CidKey root, HamtLevels L, const uint8_t* __restrict__ keys,
                                                       const uint32_t* __restrict__ key_off, const uint32_t* __restrict__ key_len,
                                                       uint32_t n, uint8_t* __restrict__ status, ValueLoc* __restrict__ loc,
                                                       uint32_t n_clear /* counters + bitmap words, contiguous from L.count */) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    {   // The call's counters and claim bitmap start out clear but for the root (every lane knows it: the same probe).  A
        // hipMemsetAsync of this unaligned range is three fill kernels of the runtime, ≈ 25 µs in front of a 0.6 ms call.
        const uint32_t rb0 = witness_find(w, root);
        const uint32_t stride = gridDim.x * blockDim.x, n_count = uint32_t(L.claimed - L.count);
        const uint32_t rcls = rb0 != kNoBlock ? hamt_size_class(w, L, rb0) : 0u;
        for (uint32_t j = t; j < n_clear; j += stride) {
            uint32_t v = 0;
            if (rb0 != kNoBlock) {
                if (j == rcls) v = 1u;                                         // count[0 + class]: the root is level 0's work list
                else if (j == n_count + (rb0 >> 5)) v = 1u << (rb0 & 31u);    // … and claimed
            }
            L.count[j] = v;
        }
    }
    if (t >= n) return;
    uint32_t h[8];
    sha256::hash_bytes(keys + key_off[t], key_len[t], h);
    uint4* hp = reinterpret_cast<uint4*>(L.hash + size_t(t) * 8);
    hp[0] = make_uint4(h[0], h[1], h[2], h[3]);
    hp[1] = make_uint4(h[4], h[5], h[6], h[7]);
    const uint32_t rb = witness_find(w, root);  // (every lane: the same probe, broadcast)
    L.cur[t] = rb;
    status[t] = uint8_t(rb == kNoBlock ? uint32_t(IPCFP_ST_ERR_MISSING_BLOCK) : kStPending);
    if (loc) loc[t] = ValueLoc{kNoBlock, 0, 0};
    if (t == 0 && rb != kNoBlock) {
        L.work[0][hamt_size_class(w, L, rb)][0] = hamt_work_entry(w, rb, w.len[rb]);
        L.recs[rb].status = 0;  // (until a parse of THIS call says otherwise: the fused top launches one instance only)
    }
}

// lane = one node of level `level`'s work list
__global__ __launch_bounds__(256, IPCFP_WALK_WAVES) void k_hamt_lv_parse(WitnessView w, HamtLevels L, uint32_t level, int vkind) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= L.count[kHamtClasses * level]) return;
    const uint32_t block = L.work[level & 1u][0][i].x;
    HamtNodeRec* out = L.recs + block;
    Rd r = open_block(w, block);
    uint32_t status = 0, std_links = 0, np32 = 0;
    uint64_t bf = 0;
    do {
        r.expect_array(2);
        uint32_t bo, bl;
        r.read_bytes(bo, bl);
        if (!r.ok() || bl > 8) break;
        if (bl) {  // big-endian integer, leading zeros stripped: the last byte holds bits 0..7
            const uint64_t v = r.peek64(bo);
            bf = __builtin_bswap64(v) >> (64u - 8u * bl);
        }
        const uint64_t np = r.read_array();
        if (!r.ok() || np > kHamtTablePointers) break;
        np32 = uint32_t(np);
        bool fits = true;
        for (uint32_t p = 0; p < np32 && r.ok(); ++p) {
            const uint32_t at = r.pos;
            fits = fits && at <= 0xffffu;
            out->ptr_off[p] = uint16_t(at);
            const uint32_t b0 = r.peek();
            if ((b0 >> 5) == 6) {
                uint32_t o, l;
                r.read_link(o, l);
                if (r.ok() && l == 38 && o == at + 5) std_links |= 1u << p;  // (read_link's own fast path saw the standard form)
            } else if ((b0 >> 5) == 4) {
                const uint64_t nkv = r.read_array();
                for (uint64_t k = 0; k < nkv && r.ok(); ++k) {
                    uint32_t ko, kl, vstart;
                    if (vkind == VK_ACTOR_STATE && actor_entry_fast(r, ko, kl, vstart)) continue;
                    r.expect_array(2);
                    r.read_bytes(ko, kl);
                    check_value(r, vkind);
                }
            } else {
                r.fail();
            }
        }
        r.finish();
        if (r.ok() && fits) status = 1;
    } while (false);
    // a standard link must really be one: l == 38 at at + 5 is also what the long way reports for that spelling only
    out->status = uint8_t(status);
    out->kinds_ok = uint8_t(status);
    out->np = uint8_t(np32);
    out->pad = 0;
    out->std_links = std_links;
    out->bitfield = bf;
}

// ---- the same for `Hamt<_, ActorState>` (the state tree: src/proofs/common/decode.rs:29-39), THIRTY-TWO LANES PER NODE ----
// One lane per node (above) is a chain of dependent loads from HBM — a 4-5 KB state-tree node is ≈ 80 pointers and bucket
// entries, each a few round trips — run by the few hundred wavefronts a level's work list fills: 0.19 ms per level on
// config 4's tree (profiles/r04_experiments.md), and the lanes of a wavefront sit in different kinds of pointers, so the
// wavefront executes both the link and the bucket path for every one.  Here a wavefront takes two nodes:
//   stage   the 32 lanes of a group copy their node into LDS with coalesced 16-byte loads (one burst, no chain);
//   outline lane 0 of the group walks the node's OUTLINE out of LDS — item headers only: where every pointer, every
//           bucket entry's ActorState and its optional address start — for the spellings every encoder writes;
//   check   all 32 lanes validate the pieces side by side: each link's 11-byte CIDv1 / dag-cbor / blake2b-256 prefix, each
//           TokenAmount's sign byte and length, each delegated address (check_address);
//   record  the HamtNodeRec, offsets written by all lanes.
// Whatever is not exactly `82 | 4x bitfield | 8x/98 pointers | (std link | 8x bucket of [82, key, 85, std link, std link,
// uint, bytes, f6 | bytes])`, or does not fit the 7.4 KB stage, is left "not tabulated" — the walker decides; a node this
// kernel accepts is one the item-by-item decode accepts, with the same record.
// LDS: 2 × 7424 B of stage + 1.2 KB of outline = 16.0 KB per wavefront, so that FOUR wavefronts share the 64 KB a CU
// hands out (one 35 KB workgroup per CU was measured: 8 192 wavefronts of 26 µs took 0.79 ms).  The outline reads 8
// bytes per LDS round trip: a wavefront's time IS the outline lane's chain of dependent LDS reads.
constexpr uint32_t kCoopLanes = 32, kCoopNodes = 2, kCoopParallelMin = 2048, kCoopParallelMinBuckets = 512;
// Two instances of the kernel share a level's work list: nodes of up to kCoopSmallStage - 24 bytes (the link nodes of the upper
// levels, the overflow nodes under a full bucket: 0.3-1.4 KB) go to the one with a 1.5 KB stage — 3.7 KB of LDS per wavefront,
// so that a CU keeps 32 of them resident instead of 9: such a wavefront's time is three dependent random reads (work list →
// length / offset → the node) and what hides them is the number of wavefronts in flight — everything else to the one with
// the 6.9 KB stage.  Each skips the other's nodes.
// (kCoopBigStage / kCoopBigEntries / kCoopSmallStage / kCoopSmallEntries: defined above hamt_claim, which files by them)


// the 8 bytes at S[p, p + 8) as a little-endian word (one aligned two-word LDS read)
__device__ __forceinline__ uint64_t lds_peek64(const uint8_t* S, uint32_t p) {
    const uint64_t* q = reinterpret_cast<const uint64_t*>(S + (p & ~7u));
    const uint64_t lo = q[0], hi = q[1];
    const uint32_t sh = (p & 7u) * 8u;
    return (lo >> sh) | ((hi << 1) << (63u - sh));
}

__device__ __forceinline__ bool lds_std_link_prefix(const uint8_t* S, uint32_t at) {
    // d8 2a | 58 27 | 00 | 01 71 a0 e4 02 20
    return lds_peek64(S, at) == 0xa071010027582ad8ull && (lds_peek64(S, at + 8u) & 0xffffffull) == 0x2002e4ull;
}

// one well-formed binary CID in S[off, off + len)?   (cbor_dev.h Rd::cid_ok)
__device__ __forceinline__ bool lds_cid_ok(const uint8_t* S, uint32_t off, uint32_t len) {
    if (len == 34 && S[off] == 0x12 && S[off + 1] == 0x20) return true;  // CIDv0
    uint32_t q = off;
    const uint32_t end = off + len;
    uint64_t field[4];
#pragma unroll
    for (int f = 0; f < 4; ++f) {
        uint64_t v = 0;
        bool done = false;
        for (int shift = 0; shift < 63; shift += 7) {
            if (q >= end) return false;
            const uint32_t c = S[q];
            ++q;
            v |= uint64_t(c & 0x7f) << shift;
            if (!(c & 0x80)) {
                if (c == 0 && shift > 0) return false;  // non-minimal varint
                done = true;
                break;
            }
        }
        if (!done) return false;
        field[f] = v;
    }
    if (field[0] != 1) return false;
    if (field[3] > 64) return false;
    return uint64_t(end - q) == field[3];
}

// the tag-42 link at S[at] whose OUTLINE (d8 2a, a byte-string header) the outline pass has seen: standard prefix, or
// the long way — 0x00 multibase byte, one well-formed CID (cbor_dev.h Rd::read_link).  *std: the 43-byte standard form.
__device__ __forceinline__ bool lds_link_ok(const uint8_t* S, uint32_t at, bool* std_form) {
    *std_form = lds_std_link_prefix(S, at);
    if (*std_form) return true;
    {   // the short spelling (cbor_dev.h link_fast_len): d8 2a | 4l | 00 01 codec code n — e.g. the builtin actors' code CIDs
        const uint64_t w0 = lds_peek64(S, at);
        const uint32_t hb = uint32_t(w0 >> 16) & 0xffu, l = hb - 0x40u, n = uint32_t(w0 >> 56);
        if (hb >= 0x45u && hb <= 0x57u && ((w0 >> 24) & 0xffffull) == 0x0100ull && ((w0 >> 24) & 0x80800000ull) == 0ull && n <= 40u && l == 5u + n)
            return true;
    }
    const uint32_t hb = S[at + 2];
    const uint32_t h = hb == 0x58u ? 4u : 3u, l = hb == 0x58u ? uint32_t(S[at + 3]) : hb - 0x40u;
    return l >= 1u && S[at + h] == 0x00u && lds_cid_ok(S, at + h + 1u, l - 1u);
}

// Address::from_bytes shape of S[off, off + n)   (cbor_dev.h check_address)
__device__ __forceinline__ bool lds_address_ok(const uint8_t* S, uint32_t off, uint32_t n) {
    if (n < 1) return false;
    const uint32_t proto = S[off];
    if (proto == 0 || proto == 4) {
        uint32_t pos = 1;
        bool term = false;
        for (int k = 0; k < 10; ++k) {
            if (pos >= n) return false;
            if (!(S[off + pos++] & 0x80)) {
                term = true;
                break;
            }
        }
        if (!term) return false;
        return proto == 0 ? pos == n : n - pos <= 54;
    }
    if (proto == 1 || proto == 2) return n == 21;
    if (proto == 3) return n == 49;
    return false;
}

// One wavefront's pair of nodes: entries 2 · pair and 2 · pair + 1 of the level's list of this instance's size class
// (`n_list` entries).  `emit_children` (the fused top levels): a node of links alone also FILES the blocks behind its links
// as the next level's work — every child, not only the ones a query will step to: the top of a state tree is 1 + 32 + 1 024
// link nodes that 66 k queries visit all of anyway, and listing them here takes the advance kernel (and its 66 k claims on
// 32 blocks) out of every top level.  A record is a pure function of its block: parsing a node no query visits changes nothing.
template <uint32_t kCoopStage, uint32_t kCoopMaxEntries, uint32_t CLS, bool EMIT, uint32_t LANES>
__device__ __forceinline__ void hamt_parse_actor_pair(const WitnessView& w, const HamtLevels& L, uint32_t level, uint32_t pair,
                                                      uint32_t n_list, uint32_t slot_base) {
    // LANES lanes per node, 64 / LANES nodes per wavefront.  32: the form the outline's parallel phases were written for.  8
    // (round 6, the SHORT nodes below the fused top): an overflow node is ≈ 500 bytes and five entries that one lane reads
    // front to back out of LDS — with 32 lanes on it, 31 wait; eight groups of eight read eight nodes side by side.
    static_assert(LANES == 32u || LANES == 8u, "group size");
    static_assert(!EMIT || LANES == 32u, "the fused top lists one child per lane");
    constexpr uint32_t NODES = 64u / LANES;
    constexpr uint64_t GMASK = (1ull << LANES) - 1ull;
    __shared__ __attribute__((aligned(16))) uint8_t stage[NODES][kCoopStage];
    __shared__ uint16_t s_ptr[NODES][kHamtTablePointers];   // pointer starts
    __shared__ uint16_t s_val[NODES][kCoopMaxEntries];      // per bucket entry: where its ActorState (0x85) starts
    __shared__ uint16_t s_l2[NODES][kCoopMaxEntries];       // … its second link (`state`) …
    __shared__ uint16_t s_adr[NODES][kCoopMaxEntries];      // … and its delegated_address item
    __shared__ uint16_t s_end[NODES][kCoopMaxEntries];      // … and where the entry ends (the parallel outline's tiling check)
    __shared__ uint8_t s_gn[NODES][kCoopMaxEntries + 4], s_gc[NODES][kCoopMaxEntries + 4], s_gb[NODES][kCoopMaxEntries + 4];
    __shared__ uint8_t s_klen[NODES][kCoopMaxEntries + 4], s_first[NODES][kHamtTablePointers];  // (the entry table's extras)
    __shared__ uint32_t s_np[NODES], s_ne[NODES], s_links[NODES], s_lall[NODES], s_slot[NODES];
    __shared__ uint64_t s_bf[NODES];
    const uint32_t lane = threadIdx.x & 63u, g = lane / LANES, sub = lane % LANES;
    const uint32_t i = pair * NODES + g;
    constexpr bool SMALL = CLS == 0u;
    const uint32_t cls = L.plain_list ? 0u : CLS;
    const bool listed = i < n_list;
    uint4 we = make_uint4(0, 0, 0, 0);
    if (listed) {
        if (L.plain_list) {
            const uint32_t b = L.plain[i];
            we = hamt_work_entry(w, b, w.len[b]);
        } else {
            we = L.work[level & 1u][cls][i];
        }
    }
    const uint32_t block = we.x, len = we.y;
    // (a split list holds this instance's nodes only; an unsplit one — the outline over a plain list — is the big instance's)
    const bool have = listed && (L.split ? true : (len + 24u <= kCoopSmallStage) == SMALL);  // (unsplit: small or not small)
    const bool staged = have && len >= 3u && len + 24u <= kCoopStage;
    uint8_t* S = stage[g];
    if (staged) {
        const uint4* src = reinterpret_cast<const uint4*>(w.arena + (uint64_t(we.z) | (uint64_t(we.w) << 32)));  // line-aligned, padded to a line
        uint4* dst = reinterpret_cast<uint4*>(S);
        const uint32_t chunks = (len + 15u) >> 4;
        for (uint32_t c = sub; c < chunks; c += LANES) dst[c] = src[c];
    }
    if (sub == 0) {
        s_np[g] = 0xffffffffu;  // "no outline"
        s_links[g] = 0u;
    }
    __syncthreads();
    // ---- outline, all 32 lanes (hamt_outline.h): anchors → entries forward → gaps → the pieces must tile the node ----
    // Worth its phases (a dozen barriers and prefix sums) only for a node with many entries: the link nodes of the upper
    // levels (32 links, 1.4 KB) and the small overflow nodes below the bucket level are ≈ 300 instructions front to back
    // (measured: levels of such nodes 11 µs sequential, 26-31 µs parallel) — lane 0 reads those, as in the first form.
    // Mid-sized nodes whose first pointer is a BUCKET (the overflow nodes under a full bucket: 4-20 entries) also gain a little
    // (a level of them: 115 µs front to back, 90 µs in parallel); which route reads a node never changes its record.
    const outline::Header hd = staged ? outline::header(S, len) : outline::Header{false, 0, 0, 0};
    const bool wide = LANES == 32u && staged && hd.ok && (len >= kCoopParallelMin || (len >= kCoopParallelMinBuckets && hd.pos0 < len && S[hd.pos0] != 0xd8u));
    bool fast_ok = false;
    if (__ballot(wide) != 0ull) {  // (wave-uniform: a wavefront of two small nodes skips the phases altogether)
    uint32_t cnt = 0, a_from = 0, a_to = 0;
    uint64_t packed = 0;  // the lane's first four anchors (one pass); more than four: the group scans again, writing
    if (wide && hd.ok) {
        const uint32_t span = (((len + LANES - 1u) / LANES) + 7u) & ~7u;  // a lane's share of the node, whole words
        const uint32_t scan_end = len >= 2u ? len - 2u : 0u;
        a_from = sub * span > hd.pos0 ? sub * span : hd.pos0;
        a_to = (sub + 1u) * span < scan_end ? (sub + 1u) * span : scan_end;
        if (a_from < a_to) cnt = outline::scan_anchors_packed(S, a_from, a_to, packed);
    }
    const bool crowded = ((__ballot(cnt > 4u) >> (g * LANES)) & GMASK) != 0ull;
    uint32_t incl = cnt;
#pragma unroll
    for (uint32_t d = 1; d < LANES; d <<= 1) {
        const uint32_t up = __shfl_up(incl, d, LANES);
        if (sub >= d) incl += up;
    }
    const uint32_t na = __shfl(incl, LANES - 1u, LANES);  // anchors of the node
    const bool fast = wide && hd.ok && na <= kCoopMaxEntries;
    if (fast && cnt) {
        if (crowded) (void)outline::scan_anchors<true>(S, a_from, a_to, s_val[g] + (incl - cnt), cnt);
        else
            for (uint32_t k = 0; k < cnt; ++k) s_val[g][incl - cnt + k] = uint16_t(packed >> (16u * k));
    }
    __syncthreads();
    bool okl = true;
    // Every anchored entry forward to its end.  An anchor that does not parse is DROPPED, not held against the node: the
    // three bytes also occur where a digest ends in 0x85 in front of another link (`… 85 | d8 2a 58 27 …`: one link
    // node in eight has such a pair) — a sequence number can never start with that link's 0xd8.  What is kept must still
    // tile the node below, so an entry that is wrongly dropped, or a stray anchor that does parse, sends the node to the
    // sequential reader.
    uint32_t ne_kept = 0;
    for (uint32_t r = 0; r * LANES < (fast ? na : 0u); ++r) {
        const uint32_t e = r * LANES + sub;
        uint32_t a = 0, l2 = 0, adr = 0, end = 0;
        bool keep = false;
        if (e < na) {
            a = s_val[g][e];
            keep = outline::entry_forward(S, a, len, l2, adr, end);
        }
        const uint32_t kept = uint32_t((__ballot(keep) >> (g * LANES)) & GMASK);
        __syncthreads();  // (every anchor of this round has been read: the kept ones move down)
        if (keep) {
            const uint32_t at = ne_kept + uint32_t(__popc(kept & ((1u << sub) - 1u)));
            s_val[g][at] = uint16_t(a);
            s_l2[g][at] = uint16_t(l2);
            s_adr[g][at] = uint16_t(adr);
            s_end[g][at] = uint16_t(end);
        }
        ne_kept += uint32_t(__popc(kept));
    }
    __syncthreads();
    if (fast) {  // every gap: how many pointers start in it, the bucket header's count (the tail is gap `na`)
        for (uint32_t e = sub; e <= ne_kept; e += LANES) {
            uint32_t n_ptr = 0, count = 0, kl = 0;
            const uint32_t from = e ? uint32_t(s_end[g][e - 1u]) : hd.pos0, target = e < ne_kept ? uint32_t(s_val[g][e]) : len;
            const bool ok = outline::gap_walk(S, from, target, len, e == ne_kept, e == 0u, n_ptr, count, nullptr, 0u, nullptr, &kl);
            okl = okl && ok;
            if (e < ne_kept) s_klen[g][e] = uint8_t(kl);
            s_gn[g][e] = uint8_t(ok ? n_ptr : 0u);
            s_gc[g][e] = uint8_t(ok ? count : 0u);
        }
    }
    __syncthreads();
    if (fast) {  // pointer numbers: a running sum over the gaps; the buckets' counts must hop from header to header
        uint32_t carry = 0;
        for (uint32_t r = 0; r * LANES <= ne_kept; ++r) {
            const uint32_t e = r * LANES + sub;
            const uint32_t v = e <= ne_kept ? uint32_t(s_gn[g][e]) : 0u;
            uint32_t run = v;
#pragma unroll
            for (uint32_t d = 1; d < LANES; d <<= 1) {
                const uint32_t up = __shfl_up(run, d, LANES);
                if (sub >= d) run += up;
            }
            if (e <= ne_kept) s_gb[g][e] = uint8_t(carry + run - v);
            carry += __shfl(run, LANES - 1u, LANES);
        }
        okl = okl && carry == hd.np;
        for (uint32_t e = sub; e < ne_kept; e += LANES) okl = okl && outline::bucket_spans(s_gc[g], e, ne_kept);
    }
    __syncthreads();
    if (fast) {  // the same walk again, now writing where every pointer starts
        uint32_t lm = 0;
        for (uint32_t e = sub; e <= ne_kept; e += LANES) {
            uint32_t n_ptr = 0, count = 0;
            const uint32_t from = e ? uint32_t(s_end[g][e - 1u]) : hd.pos0, target = e < ne_kept ? uint32_t(s_val[g][e]) : len;
            const uint32_t gn = s_gn[g][e], gb = s_gb[g][e];
            if (gn == 0u) continue;
            if (gn == 1u && s_gc[g][e] != 0u && gb < kHamtTablePointers) {  // the gap's one pointer is the bucket's header, at its start
                s_ptr[g][gb] = uint16_t(from);
                s_first[g][gb] = uint8_t(e);
                continue;
            }
            (void)outline::gap_walk(S, from, target, len, e == ne_kept, e == 0u, n_ptr, count, s_ptr[g], gb, &lm, nullptr, s_first[g], e);
        }
        if (lm) atomicOr(&s_links[g], lm);
    }
    {   // every lane of the group content?  Then the outline stands; else lane 0 reads the node front to back
        const uint64_t votes = __ballot(fast && okl);
        const uint64_t mine = GMASK << (g * LANES);
        fast_ok = (votes & mine) == mine;
        __syncthreads();
        if (fast_ok && sub == 0) {
            s_np[g] = hd.np;
            s_ne[g] = ne_kept;
            s_bf[g] = hd.bf;
            s_lall[g] = s_links[g];  // (every link; the check phase below leaves the STANDARD ones in s_links)
        }
    }
    }  // (the parallel phases)
    {
        if (staged && sub == 0) {
            if (!fast_ok) {
                outline::Result r{0, 0, 0, 0};
                if (outline::outline_sequential(S, len, r, s_ptr[g], s_val[g], s_l2[g], s_adr[g], kCoopMaxEntries, s_end[g], s_klen[g],
                                                s_first[g])) {
                    s_np[g] = r.np;
                    s_ne[g] = r.ne;
                    s_bf[g] = r.bf;
                    s_links[g] = r.links;
                    s_lall[g] = r.links;
                }
            }
        }
    }
    __syncthreads();
    // ---- check: every lane its share of the links, the entries' links and addresses ----
    const uint32_t np = s_np[g];
    bool good = staged && np != 0xffffffffu;
    if (good) {
        const uint32_t links = s_links[g], ne = s_ne[g];
        for (uint32_t p = sub; p < np; p += LANES)
            if ((links >> p) & 1u) {
                bool std_form = true;  // (a node already found bad is not used: its mask does not matter)
                good = good && lds_link_ok(S, s_ptr[g][p], &std_form);
                if (!std_form) atomicAnd(&s_links[g], ~(1u << p));  // (the record's mask names the STANDARD links only)
            }
        for (uint32_t e = sub; e < ne; e += LANES) {
            const uint32_t q = s_val[g][e];
            bool std_form;
            good = good && lds_link_ok(S, q + 1u, &std_form) && lds_link_ok(S, s_l2[g][e], &std_form);
            const uint32_t a = s_adr[g][e], ab = S[a];
            if (ab != 0xf6u) {
                const uint32_t off = ab == 0x58u ? a + 2u : a + 1u, n = ab == 0x58u ? uint32_t(S[a + 1u]) : ab - 0x40u;
                good = good && lds_address_ok(S, off, n);
            }
        }
    }
    // all thirty-two lanes of the group agree?
    const uint64_t votes = __ballot(good);
    const uint64_t mine = GMASK << (g * LANES);
    const bool node_ok = (votes & mine) == mine;
    if (have && sub == 0 && L.etab_of) {  // a table for the node's entries, if it has any (and the pool reaches that far)
        // The node's place in the pool is its place in the call's work lists — the earlier levels' counts + its index —
        // not a ticket from a counter: thirty thousand nodes of a level drawing tickets from ONE word of the L2 took
        // 270 µs where the level's parse takes 70 (profiles/r04_experiments.md).
        uint32_t slot = 0;
        if (node_ok && s_ne[g] != 0u) {
            // (`slot_base`: the lists of the earlier levels, both classes, and this level's class 0 in front of its class 1 —
            // summed once per workgroup, by all lanes at once, while the node is being staged: as a loop of dependent
            // reads by lane 0 here it stood in every wavefront's critical path)
            slot = slot_base + i < L.etab_cap ? slot_base + i + 1u : 0u;
        }
        s_slot[g] = slot;
        L.etab_of[block] = slot;
    }
    __syncthreads();  // (s_links: the lanes' atomicAnd before lane 0 reads it back; s_slot)
    if (!have) return;
    if (L.etab_of && s_slot[g]) {
        HamtEntryTab* T = L.etabs + (s_slot[g] - 1u);
        const uint32_t ne = s_ne[g], lall = s_lall[g];
        for (uint32_t e = sub; e < ne; e += LANES) {
            const uint32_t v = s_val[g][e], kl = s_klen[g][e];
            T->e[e] = HamtEntryTab::Entry{uint16_t(v - kl), uint16_t(v), uint16_t(uint32_t(s_end[g][e]) - v), uint8_t(kl), 0};
        }
        for (uint32_t p = sub; p < kHamtTablePointers; p += LANES) {  // pointer p
            const bool bucket = p < np && !((lall >> p) & 1u);
            const uint32_t cnt = bucket ? uint32_t(S[s_ptr[g][p]]) - 0x80u : 0u;
            T->first[p] = uint8_t(cnt ? s_first[g][p] : 0u);
            T->count[p] = uint8_t(cnt);
        }
        if (sub == 0) {
            T->links = lall;
            T->n_entries = ne;
        }
    }
    HamtNodeRec* out = L.recs + block;
    if (node_ok) {
        // offsets: sixteen dwords = thirty-two u16
        for (uint32_t q = sub; q < 16u; q += LANES) {
            const uint32_t lo = 2u * q < np ? s_ptr[g][2u * q] : 0u, hi = 2u * q + 1u < np ? s_ptr[g][2u * q + 1u] : 0u;
            reinterpret_cast<uint32_t*>(out->ptr_off)[q] = lo | (hi << 16);
        }
    }
    // Only for a node of links alone (the upper levels: every query that stands on it steps through one of them).  A bucket
    // node's few links lead to overflow nodes a handful of its queries follow: resolving all of them here cost the bucket
    // level's parse 28 µs (two more dependent reads in every wavefront) to save its advance nothing.
    const bool resolve = node_ok && L.child != nullptr && s_ne[g] == 0u;
    uint32_t c = kNoBlock;  // (LANES == 32: the block behind the lane's own pointer, what EMIT lists)
    if (resolve) {  // the block behind pointer p (the 38 CID bytes of a standard link start 5 bytes in)
        for (uint32_t p = sub; p < kHamtTablePointers; p += LANES) {
            c = kNoBlock;
            if (p < np && ((s_links[g] >> p) & 1u)) {
                const uint32_t at = uint32_t(s_ptr[g][p]) + 5u;
                CidKey key;
#pragma unroll
                for (int j = 0; j < 5; ++j) key.w[j] = lds_peek64(S, at + 8u * uint32_t(j));
                key.w[4] &= (1ull << 48) - 1ull;
                c = witness_find_quiet(w, key);
            }
            L.child[size_t(block) * kHamtTablePointers + p] = c;
        }
    }
    if (EMIT) {  // (an instance of its own — the fused top's: the other launches do not carry this code; the lanes still here vote)
        const bool em = resolve && c != kNoBlock;
        const uint32_t clen = em ? w.len[c] : 0u;
        const uint32_t ccls = em && L.split ? hamt_class_of_len(clen) : 0u;
        // a child the fused top's parse (the small-stage instance alone) will not take must not keep a record of another call
        if (em && ccls != 0u) L.recs[c].status = 0;
#pragma unroll
        for (uint32_t k = 0; k < kHamtClasses; ++k) {
            const uint64_t votes = __ballot(em && ccls == k);
            if (!votes) continue;
            const uint32_t leader = uint32_t(__ffsll((long long)votes)) - 1u;
            uint32_t base = 0;
            if (lane == leader) base = atomicAdd(L.count + kHamtClasses * (level + 1u) + k, uint32_t(__popcll(votes)));
            base = __shfl(base, leader, 64);
            if (em && ccls == k) {
                const uint32_t at = base + uint32_t(__popcll(votes & ((1ull << lane) - 1ull)));
                if (at < L.cap) L.work[(level + 1u) & 1u][k][at] = hamt_work_entry(w, c, clen);
                else *L.top_overflow = 1u;
            }
        }
    }
    if (sub == 0) {
        out->status = uint8_t(node_ok ? 1u : 0u);  // 0: not tabulated — the walker decides
        out->kinds_ok = uint8_t(node_ok ? 1u : 0u);
        out->np = uint8_t(node_ok ? np : 0u);
        out->pad = uint8_t(resolve ? 1u : 0u);  // bit 0: L.child holds this node's standard links
        out->std_links = node_ok ? s_links[g] : 0u;
        out->bitfield = node_ok ? s_bf[g] : 0ull;
    }
}

// A level's list of one size class, two nodes per wavefront, COUNT-DRIVEN: the grid is what the chip holds of this
// instance and strides over the list, so a list that turned out empty costs a launch of workgroups that read one word
// (round 5 sized every launch by the level's upper bound: 32 k workgroups to find an empty list).
// (EMIT: hamt_parse_actor_pair's child listing, the fused top's instance.  The short-node instance asks for FOUR wavefronts
// per SIMD, what round 5's form reached at 99 VGPRs: with the listing code and the count-driven loop the
// allocator otherwise took 152 and one wavefront per SIMD less, 85 → 100 µs for the overflow level.)
template <uint32_t kCoopStage, uint32_t kCoopMaxEntries, uint32_t CLS, bool EMIT, uint32_t LANES = kCoopLanes>
__global__ __launch_bounds__(64, LANES != 32u ? 3 : (CLS == 0u ? 5 : (CLS == 1u ? 4 : 3))) void k_hamt_lv_parse_actor(WitnessView w, HamtLevels L,
                                                                                                                    uint32_t level) {
    const uint32_t n_raw = L.count[L.plain_list ? 0u : kHamtClasses * level + CLS];
    const uint32_t n_list = !L.plain_list && n_raw > L.cap ? L.cap : n_raw;
    uint32_t slot_base = 0;
    if (L.etab_of && !L.plain_list) {  // lane l: the count of list l in front of this one (2 · levels + 1 < 64 lists)
        const uint32_t before = kHamtClasses * level + CLS;
        slot_base = threadIdx.x < before ? L.count[threadIdx.x] : 0u;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) slot_base += __shfl_xor(slot_base, d, 64);
    }
    for (uint32_t pair = blockIdx.x; pair * (64u / LANES) < n_list; pair += gridDim.x) {
        hamt_parse_actor_pair<kCoopStage, kCoopMaxEntries, CLS, EMIT, LANES>(w, L, level, pair, n_list, slot_base);
        __syncthreads();  // (the pair's LDS is the next pair's)
    }
}

// One query one level down from `block` (a node whose record is THIS call's).  → st (kStPending: not settled here),
// next (the node it stands on afterwards, or kNoBlock), hit (the value, when st is TRUE), via_child (the step took a link the
// parse had resolved: HamtLevels::child).
__device__ __forceinline__ void hamt_advance_step(const WitnessView& w, const HamtLevels& L, uint32_t level, uint32_t bit_width, int vkind,
                                                  const uint8_t* __restrict__ keys, const uint32_t* __restrict__ key_off,
                                                  const uint32_t* __restrict__ key_len, uint32_t t, uint32_t block, uint32_t& st,
                                                  uint32_t& next, ValueLoc& hit, bool& via_child) {
    const HamtNodeRec* rec = L.recs + block;
    const uint32_t etab = L.etab_of ? L.etab_of[block] : 0u;
    st = kStPending;
    next = kNoBlock;
    via_child = false;
    do {
        // status | kinds_ok << 8 | np << 16 | pad << 24, std_links, bitfield: the record's first 16 bytes in ONE read (three
        // reads at three points of the control flow were three round trips of the query's chain)
        const uint4 rec_head = *reinterpret_cast<const uint4*>(rec);
        const uint32_t head = rec_head.x;
        if ((head & 0xffu) != 1u) break;  // not tabulated: the walker decides (from the root)
        const uint32_t np = (head >> 16) & 0xffu;
        const uint32_t consumed = level * bit_width;
        if (consumed + bit_width > 256u) {  // HashBits::next after the node has decoded
            st = IPCFP_ST_ERR_MAX_DEPTH;
            break;
        }
        const uint4* hp = reinterpret_cast<const uint4*>(L.hash + size_t(t) * 8);
        const uint4 h0 = hp[0], h1 = hp[1];
        const uint32_t h[8] = {h0.x, h0.y, h0.z, h0.w, h1.x, h1.y, h1.z, h1.w};
        const uint32_t idx = sha256::take_bits(h, consumed, bit_width);
        const uint64_t bf = uint64_t(rec_head.z) | (uint64_t(rec_head.w) << 32);
        if (idx >= 64u || !((bf >> idx) & 1ull)) {
            st = IPCFP_ST_NOT_FOUND;
            break;
        }
        const uint32_t rank = uint32_t(__popcll(bf & ((1ull << idx) - 1ull)));
        if (rank >= np) {
            st = IPCFP_ST_ERR_DECODE;
            break;
        }
        const uint8_t* g = w.arena + w.off[block];
        if (etab && !((L.etabs[etab - 1u].links >> rank) & 1u)) {  // a bucket of a node whose entries the parse kept
            const HamtEntryTab* T = L.etabs + (etab - 1u);
            const uint32_t c = T->count[rank], e0 = T->first[rank];
            const uint8_t* key = keys + key_off[t];
            const uint32_t key_len_t = key_len[t];
            bool found = false;
            for (uint32_t k = 0; k < c && !found; ++k) {
                const HamtEntryTab::Entry en = T->e[e0 + k];
                if (en.key_len != key_len_t) continue;
                uint64_t diff = 0;
                for (uint32_t i = 0; i < key_len_t; i += 8u) {
                    const uint32_t valid = key_len_t - i;
                    uint64_t a, b = 0;
                    __builtin_memcpy(&a, g + en.key_off + i, 8);  // (the arena has slack behind every block)
                    if (valid >= 8u) __builtin_memcpy(&b, key + i, 8);
                    else
                        for (uint32_t j = 0; j < valid; ++j) b |= uint64_t(key[i + j]) << (8u * j);  // never read past the key
                    uint64_t d = a ^ b;
                    if (valid < 8u) d &= (1ull << (8u * valid)) - 1ull;
                    diff |= d;
                }
                if (diff == 0) {
                    found = true;
                    hit = ValueLoc{block, en.val_off, en.val_len};
                }
            }
            st = found ? uint32_t(IPCFP_ST_TRUE) : uint32_t(IPCFP_ST_NOT_FOUND);
            break;
        }
        const uint32_t off = rec->ptr_off[rank];
        CidKey link;
        bool is_link = true;
        if (((rec_head.y >> rank) & 1u) && (head & (1u << 24))) {  // resolved by the parse: one word
            next = L.child[size_t(block) * kHamtTablePointers + rank];
            via_child = true;
            if (next == kNoBlock) st = IPCFP_ST_ERR_MISSING_BLOCK;
            else if (w.touched) atomicOr(&w.touched[next >> 5], 1u << (next & 31));  // (what witness_find records)
            break;
        }
        if ((rec_head.y >> rank) & 1u) {
#pragma unroll
            for (int j = 0; j < 5; ++j) __builtin_memcpy(&link.w[j], g + off + 5 + 8 * j, 8);  // unaligned 8-byte loads
            link.w[4] &= (1ull << 48) - 1ull;
        } else {
            Rd r;
            r.init(g, w.len[block]);
            r.pos = off;
            if ((r.peek() >> 5) == 6) {
                uint32_t o, l;
                r.read_link(o, l);
                if (!r.ok()) break;  // (cannot happen: the node was validated) → the walker
                link = r.key_any(o, l);
            } else {
                // a bucket `[[key, value]…]` of a validated node: find the key
                is_link = false;
                const uint8_t* key = keys + key_off[t];
                const uint32_t key_len_t = key_len[t];
                const uint64_t nkv = r.read_array();
                bool found = false;
                for (uint64_t k = 0; k < nkv && r.ok() && !found; ++k) {
                    uint32_t ko, kl, vstart;
                    if (!(vkind == VK_ACTOR_STATE && actor_entry_fast(r, ko, kl, vstart))) {
                        r.expect_array(2);
                        r.read_bytes(ko, kl);
                        vstart = r.pos;
                        r.skip();
                    }
                    if (r.ok() && kl == key_len_t && r.equal_bytes(ko, key, kl)) {
                        found = true;
                        hit = ValueLoc{block, vstart, r.pos - vstart};
                    }
                }
                if (!r.ok()) break;  // → the walker
                st = found ? uint32_t(IPCFP_ST_TRUE) : uint32_t(IPCFP_ST_NOT_FOUND);
            }
        }
        if (is_link) {
            next = witness_find(w, link);
            if (next == kNoBlock) st = IPCFP_ST_ERR_MISSING_BLOCK;
        }
    } while (false);
}

// lane = query: one level down
__global__ __launch_bounds__(256) void k_hamt_lv_advance(WitnessView w, HamtLevels L, uint32_t level, uint32_t bit_width, int vkind,
                                                         const uint8_t* __restrict__ keys, const uint32_t* __restrict__ key_off,
                                                         const uint32_t* __restrict__ key_len, uint32_t n,
                                                         uint8_t* __restrict__ status, ValueLoc* __restrict__ loc) {
    __shared__ uint32_t s_claims[kClaimSet];
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t block = t < n ? L.cur[t] : kNoBlock;
    const bool live = block != kNoBlock;  // (else: settled, left to the walker, or beyond the batch — still a party to the workgroup's claims)
    uint32_t st = kStPending, next = kNoBlock;
    ValueLoc hit{kNoBlock, 0, 0};
    bool via_child = false;
    if (live) hamt_advance_step(w, L, level, bit_width, vkind, keys, key_off, key_len, t, block, st, next, hit, via_child);
    hamt_claim(w, L, next, level + 1u, next != kNoBlock, s_claims);  // (the ONE place: every lane of the workgroup comes through here)
    if (!live) return;
    if (next != kNoBlock) {
        L.cur[t] = next;
        return;
    }
    L.cur[t] = kNoBlock;
    if (st != kStPending) {
        status[t] = uint8_t(st);
        if (loc && st == IPCFP_ST_TRUE) loc[t] = hit;
    }
}

// lane = query: the `top` FUSED levels in one go.  Their nodes were listed by the parse itself (hamt_parse_actor_pair
// emit_children), level after level, so every record on the way down is this call's as long as each step takes a link the
// parse resolved; a step that goes another way (a bucket node this high, a link in another spelling) ends the query's part
// here and the walker takes it from the root.  The node a query stands on after the last fused level is claimed for level
// `top`'s lists — from there on the batch advances level by level as before.
__global__ __launch_bounds__(256) void k_hamt_lv_advance_top(WitnessView w, HamtLevels L, uint32_t top, uint32_t bit_width, int vkind,
                                                             const uint8_t* __restrict__ keys, const uint32_t* __restrict__ key_off,
                                                             const uint32_t* __restrict__ key_len, uint32_t n,
                                                             uint8_t* __restrict__ status, ValueLoc* __restrict__ loc) {
    __shared__ uint32_t s_claims[kClaimSet];
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t block = t < n ? L.cur[t] : kNoBlock;
    const bool live = block != kNoBlock && *L.top_overflow == 0u;  // (a full list: nothing of the top is trusted, the walker decides)
    uint32_t st = kStPending, next = kNoBlock;
    ValueLoc hit{kNoBlock, 0, 0};
    if (live)
        for (uint32_t lv = 0; lv < top; ++lv) {
            bool via_child = false;
            hamt_advance_step(w, L, lv, bit_width, vkind, keys, key_off, key_len, t, block, st, next, hit, via_child);
            if (next == kNoBlock) break;  // settled, or left pending
            if (lv + 1u == top) break;    // stands on a node of level `top`: claimed below
            if (!via_child) {             // the child was not listed by the parse: its record is not this call's
                next = kNoBlock;
                st = kStPending;
                break;
            }
            block = next;
        }
    hamt_claim(w, L, next, top, next != kNoBlock, s_claims);
    if (t >= n || L.cur[t] == kNoBlock) return;
    if (next != kNoBlock) {
        L.cur[t] = next;
        return;
    }
    L.cur[t] = kNoBlock;
    if (st != kStPending) {
        status[t] = uint8_t(st);
        if (loc && st == IPCFP_ST_TRUE) loc[t] = hit;
    }
}

// Scratch of one call: [cur n | hash 8n | 4 work lists of cap 16-byte entries | count 2 (levels + 2) | claimed words | 8 spare] u32 + child table + etab_of.
size_t hamt_levels_scratch_words(uint32_t n, uint32_t n_blocks, uint32_t levels) {
    const size_t cap = n < n_blocks ? n : n_blocks;
    return size_t(n) * 9 + 4 + cap * 24 + 3 * size_t(levels + 2) + div_up(n_blocks, 32) + 8 + size_t(n_blocks) * kHamtTablePointers + size_t(n_blocks);
}

// The 32-lane outline over ANY list of blocks (not a level of a walk): every block of `work_d[0 .. *count_d)` whose length
// the big-stage instance takes (kHamtOutlineMinLen ≤ len, hamt_table.h) gets its HamtNodeRec in recs_d[block] — status 1
// and kinds_ok = HK_ACTOR_STATE when it is a state-tree node in the spellings the outline reads, status 0 ("not tabulated:
// the walker decides") otherwise.  The per-call node table of the storage proofs uses it for the 4-5 KB nodes, which one
// lane per block parses for a third of a millisecond while the rest of the chip idles (host/verify_storage.cpp).
int launch_hamt_outline_list(ipcfp_ctx* ctx, hipStream_t stream, const WitnessView& w, void* recs_d, uint32_t* work_d, uint32_t* count_d,
                             uint32_t bound) {
    if (bound == 0) return IPCFP_OK;
    HamtLevels L{};
    L.recs = static_cast<HamtNodeRec*>(recs_d);
    L.plain = work_d;
    L.count = count_d;
    L.plain_list = 1u;
    L.cap = bound;
    static_assert(kHamtOutlineMinLen + 24u > kCoopSmallStage, "every listed block is the big-stage instance's");
    hipLaunchKernelGGL((k_hamt_lv_parse_actor<kCoopBigStage, kCoopBigEntries, 2u, false>), dim3(std::min(div_up(bound, kCoopNodes), 32768u)), dim3(64), 0,
                       stream, w, L, 0u);
    IPCFP_HIP(ctx, hipGetLastError());
    return IPCFP_OK;
}

int launch_hamt_get_levels(ipcfp_ctx* ctx, const WitnessView& w, const CidKey& root, uint32_t bit_width, int vkind,
                           const uint8_t* keys_d, const uint32_t* key_off_d, const uint32_t* key_len_d, uint32_t n,
                           uint8_t* status_d, void* loc_d, uint32_t levels, uint32_t* scratch_d, void* recs_d, bool coop, void* etabs_d,
                           uint32_t etab_cap) {
    if (n == 0) return IPCFP_OK;
    const uint32_t cap = n < w.n ? n : w.n;
    const uint32_t words = div_up(w.n, 32);
    const bool actor = vkind == VK_ACTOR_STATE && coop;
    HamtLevels L{};
    L.cur = scratch_d;
    L.hash = scratch_d + size_t(n);
    L.work[0][0] = reinterpret_cast<uint4*>(scratch_d + ((size_t(n) * 9 + 3) & ~size_t(3)));  // (16-byte entries: scratch_d is hipMalloc'd)
    for (uint32_t k = 1; k < 2u * kHamtClasses; ++k) L.work[k / kHamtClasses][k % kHamtClasses] = L.work[0][0] + size_t(k) * cap;
    L.count = reinterpret_cast<uint32_t*>(L.work[0][0] + size_t(2u * kHamtClasses) * cap);
    L.claimed = L.count + kHamtClasses * (levels + 2);
    L.split = actor ? 1u : 0u;
    L.cap = cap;
    L.top_overflow = L.claimed + words;  // (the first of the eight spare words behind the bitmap: cleared with it)
    L.child = actor ? L.claimed + words + 8 : nullptr;  // (the 32-lane parse fills it)
    const bool tabs = L.child != nullptr && etabs_d != nullptr && etab_cap != 0;
    L.etab_of = tabs ? L.claimed + words + 8 + size_t(w.n) * kHamtTablePointers : nullptr;
    L.etabs = tabs ? static_cast<HamtEntryTab*>(etabs_d) : nullptr;
    L.etab_cap = tabs ? etab_cap : 0u;
    L.recs = static_cast<HamtNodeRec*>(recs_d);
    // counters, bitmap and the spare words are contiguous: cleared by the first launch
    ValueLoc* loc = static_cast<ValueLoc*>(loc_d);
    hipLaunchKernelGGL(k_hamt_lv_start, dim3(div_up(n, 256)), dim3(256), 0, ctx->stream, w, root, L, keys_d, key_off_d, key_len_d,
                       n, status_d, loc, kHamtClasses * (levels + 2u) + words + 8u);
    // The FUSED TOP (round 6).  The upper levels of a big tree are a handful of link nodes every query passes: level l holds
    // at most 32^l of them.  Round 5 ran a parse / parse / advance triple per level there — 130 µs for 1 + 32 + 1 024 nodes
    // of a 4 M-actor tree, the advance kernels being 66 k queries claiming the same few blocks.  Now the parse lists a link
    // node's children itself (level after level, the small-stage instance alone: a link node is 1.4 KB) and ONE advance
    // launch takes every query down all of them.  How many levels: all but the last three the call queues (the bucket level
    // and two of overflow nodes), at most three, and only while a level's worst case fits a list.
    static const int top_env = [] { const char* e = std::getenv("IPCFP_HAMT_TOP"); return e ? std::atoi(e) : -1; }();
    uint32_t top = 0;
    if (actor && levels >= 4 && bit_width == 5 && top_env != 0) {
        top = std::min(levels - 3u, 3u);
        if (top_env > 0) top = std::min(uint32_t(top_env), levels - 1u);
        while (top > 0 && (1ull << (bit_width * (top - 1u))) > cap) --top;
    }
    auto parse_grid = [&](uint32_t lv, uint32_t per_cu, uint32_t nodes_per_wg = kCoopNodes) {
        uint64_t fan = 1;  // level l holds at most min(n, 2^(bit_width · l)) distinct nodes
        for (uint32_t k = 0; k < lv && fan < cap; ++k) fan <<= bit_width;
        const uint32_t bound = fan < cap ? uint32_t(fan) : cap;
        (void)per_cu;
        // One workgroup per pair up to a cap the loop covers: the dispatcher backfills workgroups as they retire and so
        // balances nodes of unequal length; a grid of "what the chip holds" striding over the list was 255 µs for the
        // bucket level against 173 (profiles/r06_experiments.md).  An empty list is one word read per workgroup.
        return std::min(div_up(bound, nodes_per_wg), 32768u);
    };
    for (uint32_t lv = 0; lv < top; ++lv) {
        if (lv + 1u < top)
            hipLaunchKernelGGL((k_hamt_lv_parse_actor<kCoopSmallStage, kCoopSmallEntries, 0u, true>), dim3(parse_grid(lv, 32)), dim3(64), 0,
                               ctx->stream, w, L, lv);
        else
            hipLaunchKernelGGL((k_hamt_lv_parse_actor<kCoopSmallStage, kCoopSmallEntries, 0u, false>), dim3(parse_grid(lv, 32)), dim3(64), 0,
                               ctx->stream, w, L, lv);
    }
    if (top)
        hipLaunchKernelGGL(k_hamt_lv_advance_top, dim3(div_up(n, 256)), dim3(256), 0, ctx->stream, w, L, top, bit_width, vkind, keys_d, key_off_d,
                           key_len_d, n, status_d, loc);
    // Below the fused top the SHORT nodes (the overflow nodes under full buckets) keep the 32-lane form, two nodes per
    // wavefront.  IPCFP_HAMT_SMALL_LANE=1 (A/B): one lane each with the line-staged reader (hamt_table_lane.hip) — measured
    // and off: 182 µs for the overflow level against ≈ 100, and its nodes have no entry table, so the advance behind it
    // walks the bucket with the reader (42 µs against 12): profiles/r06_experiments.md.
    static const bool small_lane = [] { const char* e = std::getenv("IPCFP_HAMT_SMALL_LANE"); return e && std::atoi(e) == 1; }();
    for (uint32_t lv = top; lv < levels; ++lv) {
        if (actor) {
            // Which class a level's nodes fall into is the tree's business, but the odds are known: the first level below
            // the fused top of a big tree is the BUCKET level (2-8 KB nodes), the levels under it hold the short overflow
            // nodes.  The unlikely class gets a narrow grid — the kernel strides, so a surprise costs time, not results — and
            // an empty list then costs ≈ 4 µs instead of ≈ 9 (32 k workgroups that read one word each).
            const bool bucket_level = top > 0 && lv == top;
            const uint32_t narrow = 1024u;
            static const bool small8 = [] { const char* e = std::getenv("IPCFP_HAMT_SMALL8"); return !(e && std::atoi(e) == 0); }();
            const bool eight = small8 && top > 0;  // (eight lanes per short node below a fused top: hamt_parse_actor_pair LANES)
            const uint32_t g_small = top > 0 && bucket_level ? std::min(parse_grid(lv, 32, eight ? 8u : kCoopNodes), narrow)
                                                             : parse_grid(lv, 32, eight ? 8u : kCoopNodes);
            const uint32_t g_long = top > 0 && !bucket_level ? std::min(parse_grid(lv, 14), narrow) : parse_grid(lv, 14);
            if (small_lane && top > 0) {
                uint64_t fan = 1;
                for (uint32_t k = 0; k < lv && fan < cap; ++k) fan <<= bit_width;
                const int rc = launch_hamt_lv_parse_lane(ctx, w, L.work[lv & 1u][0], L.count + kHamtClasses * lv, cap, fan < cap ? uint32_t(fan) : cap,
                                                         HK_ACTOR_STATE, recs_d, L.etab_of);
                if (rc) return rc;
            } else if (eight) {
                hipLaunchKernelGGL((k_hamt_lv_parse_actor<kCoopSmallStage, kCoopSmallEntries, 0u, false, 8u>), dim3(g_small), dim3(64), 0,
                                   ctx->stream, w, L, lv);
            } else {
                hipLaunchKernelGGL((k_hamt_lv_parse_actor<kCoopSmallStage, kCoopSmallEntries, 0u, false>), dim3(g_small), dim3(64), 0, ctx->stream,
                                   w, L, lv);
            }
            hipLaunchKernelGGL((k_hamt_lv_parse_actor<kCoopMidStage, kCoopMidEntries, 1u, false>), dim3(g_long), dim3(64), 0, ctx->stream, w, L,
                               lv);
            hipLaunchKernelGGL((k_hamt_lv_parse_actor<kCoopBigStage, kCoopBigEntries, 2u, false>), dim3(g_long), dim3(64), 0, ctx->stream, w, L,
                               lv);
        } else {
            uint64_t fan = 1;
            for (uint32_t k = 0; k < lv && fan < cap; ++k) fan <<= bit_width;
            const uint32_t bound = fan < cap ? uint32_t(fan) : cap;
            hipLaunchKernelGGL(k_hamt_lv_parse, dim3(div_up(bound, 256)), dim3(256), 0, ctx->stream, w, L, lv, vkind);
        }
        hipLaunchKernelGGL(k_hamt_lv_advance, dim3(div_up(n, 256)), dim3(256), 0, ctx->stream, w, L, lv, bit_width, vkind, keys_d,
                           key_off_d, key_len_d, n, status_d, loc);
    }
    IPCFP_HIP(ctx, hipGetLastError());
    return IPCFP_OK;
}

}  // namespace ipcfp
