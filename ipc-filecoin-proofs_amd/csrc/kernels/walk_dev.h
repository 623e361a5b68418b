// csrc/kernels/walk_dev.h — AMT / HAMT path walks on the HBM-resident witness, one query
// per lane.
//
// Device counterparts of (crates fvm_ipld_amt 0.7.4 / fvm_ipld_hamt 0.10.4, restated in
// oracle/amt.cpp, oracle/hamt.cpp):
//   Amtv0::<V>::load(root).get(i)   src/proofs/events/verifier.rs:220-226
//   Amt::<V>::load(root).get(i)     src/proofs/events/verifier.rs:234-239
//   Hamt::load_with_bit_width(root, bw).get(key)
//                                   src/proofs/common/decode.rs:29-39, src/proofs/storage/decode.rs:79-96
// Semantics kept exact: `load` reads and fully decodes the root block; every visited node
// is decoded COMPLETELY (all links are valid CIDs, all values type-check, counts match the
// bitmap) before the next link is followed; a missing child block is Err, a clear bit is
// None.  Node bitmaps are tiny (1-4 bytes for AMT, ≤ 4 for a bit-width-5 HAMT), so the
// child select is a mask + popcount in registers.
#pragma once
#include "cbor_dev.h"
#include "hamt_table.h"
#include "sha256_dev.h"

// The walk / scan / verify kernels are chains of dependent loads (one parser per lane): what hides
// their latency is the number of wavefronts in flight, so they are compiled for this many waves per
// SIMD (the register allocator spills the rest to scratch).  Measured on the 1M-receipt tipset —
// DESIGN.md §4.
#ifndef IPCFP_WALK_WAVES
#define IPCFP_WALK_WAVES 4
#endif

namespace ipcfp {

__device__ __forceinline__ Rd open_block(const WitnessView& w, uint32_t b) {
    Rd r;
    r.init(w.arena + w.off[b], w.len[b]);
    return r;
}

// ---------------------------------------------------------------------------
// AMT
// ---------------------------------------------------------------------------
constexpr uint32_t kAmtMaxBitWidth = 8;

struct AmtNode {
    uint32_t width;       // 1 << bit_width
    // up to 256 bits as four words, bit i ⇔ b[i/64] >> (i%64) (LSB-first bytes ⇒ same bit order).  Always
    // accessed with compile-time word indices (select chains): a dynamically indexed array would live
    // in scratch memory, and every walk kernel carries one of these per lane.
    uint64_t b[4];
    uint32_t nlinks, nvalues;
    // position bookkeeping for the entry the caller asked for
    uint32_t want;        // ordinal (rank) of the wanted link/value, or ~0u
    uint32_t want_off, want_len;  // value: item offset/len; link: CID bytes offset/len
    __device__ __forceinline__ uint64_t word(uint32_t k) const {
        // masks, not selects: LLVM folds a select of loads into a load from a selected ADDRESS, which is
        // exactly the dynamic index this is here to avoid
        return (b[0] & (k == 0 ? ~0ull : 0ull)) | (b[1] & (k == 1 ? ~0ull : 0ull)) | (b[2] & (k == 2 ? ~0ull : 0ull)) |
               (b[3] & (k == 3 ? ~0ull : 0ull));
    }
    __device__ __forceinline__ bool bit(uint32_t i) const { return (word(i >> 6) >> (i & 63)) & 1ull; }
    __device__ __forceinline__ uint32_t rank(uint32_t i) const {
        uint32_t r = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t lo = uint32_t(k) * 64u;
            if (i >= lo + 64) r += __popcll(b[k]);
            else if (i > lo) r += __popcll(b[k] & ((1ull << (i - lo)) - 1ull));
        }
        return r;
    }
    __device__ __forceinline__ uint32_t popcount() const {
        return __popcll(b[0]) + __popcll(b[1]) + __popcll(b[2]) + __popcll(b[3]);
    }
    // the low m bits set and nothing else (m ≤ 256)
    __device__ __forceinline__ bool is_low(uint32_t m) const {
        bool ok = true;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t lo = uint32_t(k) * 64u;
            const uint64_t want = m >= lo + 64 ? ~0ull : (m > lo ? (1ull << (m - lo)) - 1ull : 0ull);
            ok = ok && b[k] == want;
        }
        return ok;
    }
};

// Decode `[bmap, [links…], [values…]]` at r (CollapsedNode) and apply the expand() checks.
// PASS 1 (sub == ~0u): only decode/validate.  With sub != ~0u the ordinal rank(sub) entry's
// location is recorded (link CID bytes or value item).
__device__ __forceinline__ void amt_read_node(Rd& r, uint32_t bw, int vkind, uint32_t sub, AmtNode& nd) {
    nd.width = 1u << bw;
    nd.nlinks = nd.nvalues = 0;
    nd.want = ~0u;
    nd.want_off = nd.want_len = 0;
    r.expect_array(3);
    uint32_t bo, bl;
    r.read_bytes(bo, bl);
    if (!r.ok()) return;
    const uint32_t need = (nd.width + 7) / 8;
    const bool bmap_len_ok = bl == need;
    // keep the bits we can hold; a wrong length is an error AFTER links/values decode (order is
    // irrelevant: every failure here is ERR_DECODE)
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint32_t lo = 8u * uint32_t(k);
        uint64_t v = 0;
        if (bl > lo) {  // unaligned 8-byte fetch, bytes past the string masked off
            v = r.peek64(bo + lo);
            const uint32_t valid = bl - lo;
            if (valid < 8) v &= (1ull << (8u * valid)) - 1ull;
        }
        nd.b[k] = v;
    }
    if (nd.width < 64) nd.b[0] &= (1ull << nd.width) - 1ull;  // bits ≥ width are ignored (bw < 3)
    if (bmap_len_ok && sub != ~0u && sub < nd.width && nd.bit(sub)) nd.want = nd.rank(sub);
    const uint64_t nl = r.read_array();
    if (!r.ok()) return;
    for (uint64_t i = 0; i < nl && r.ok(); ++i) {
        uint32_t o, l;
        r.read_link(o, l);
        if (i == nd.want) {
            nd.want_off = o;
            nd.want_len = l;
        }
    }
    const uint64_t nv = r.read_array();
    if (!r.ok()) return;
    for (uint64_t i = 0; i < nv && r.ok(); ++i) {
        const uint32_t start = r.pos;
        check_value(r, vkind);
        if (i == nd.want) {
            nd.want_off = start;
            nd.want_len = r.pos - start;
        }
    }
    if (!r.ok()) return;
    nd.nlinks = nl > 0xffffffffULL ? 0xffffffffu : uint32_t(nl);
    nd.nvalues = nv > 0xffffffffULL ? 0xffffffffu : uint32_t(nv);
    if (nl && nv) return r.fail();    // LinksAndValues
    if (!bmap_len_ok) return r.fail();
    const uint64_t have = nl ? nl : nv;
    if (have != nd.popcount()) return r.fail();
}

struct AmtRootInfo {
    uint32_t block;
    uint32_t node_off;  // offset of the inline root node
    uint32_t bit_width;
    uint64_t height, count;
};

// Amt::load / Amtv0::load.  Returns a status: TRUE on success, ERR_* otherwise.  `hint` (optional)
// names a slot of the ROOT node whose entry location is recorded in *root_node while the node is
// being validated, so that a following get() on a height-0 tree need not parse the block again.
__device__ __forceinline__ uint32_t amt_load(const WitnessView& w, const CidKey& root, int version, int vkind,
                                             AmtRootInfo& info, uint32_t hint = ~0u, AmtNode* root_node = nullptr) {
    const uint32_t b = witness_find(w, root);
    if (b == kNoBlock) return IPCFP_ST_ERR_MISSING_BLOCK;
    Rd r = open_block(w, b);
    info.block = b;
    if (version == 0) {
        r.expect_array(3);
        info.bit_width = 3;
    } else {
        r.expect_array(4);
        const uint64_t bw = r.read_uint();
        if (r.ok() && (bw < 1 || bw > kAmtMaxBitWidth)) r.fail();
        info.bit_width = uint32_t(bw);
    }
    info.height = r.read_uint();
    info.count = r.read_uint();
    info.node_off = r.pos;
    if (!r.ok()) return IPCFP_ST_ERR_DECODE;
    AmtNode nd;
    amt_read_node(r, info.bit_width, vkind, hint, nd);
    r.finish();
    if (!r.ok()) return IPCFP_ST_ERR_DECODE;
    if (info.height > 64 / info.bit_width) return IPCFP_ST_ERR_DECODE;  // MaxHeight
    if (root_node) *root_node = nd;
    return IPCFP_ST_TRUE;
}

__device__ __forceinline__ uint64_t amt_span(uint32_t bw, uint64_t height) {
    const uint64_t shift = uint64_t(bw) * height;
    return shift >= 64 ? ~0ULL : (1ULL << shift);
}

// Amt::get(index) on a loaded root.  TRUE ⇒ loc set; NOT_FOUND ⇒ None; ERR_*.
__device__ __forceinline__ uint32_t amt_get(const WitnessView& w, const AmtRootInfo& root, int vkind, uint64_t index,
                                            ValueLoc& loc) {
    if (index == ~0ULL) return IPCFP_ST_ERR;  // > MAX_INDEX
    if (index >= amt_span(root.bit_width, root.height + 1)) return IPCFP_ST_NOT_FOUND;
    uint32_t block = root.block;
    uint32_t node_off = root.node_off;
    uint64_t height = root.height;
    uint64_t i = index;
    const uint32_t bw = root.bit_width;
    for (;;) {
        Rd r = open_block(w, block);
        r.pos = node_off;
        const uint64_t span = amt_span(bw, height);
        // which entry of this node do we need?  (decided before parsing so one pass finds it)
        // Leaf: entry i; Link node: entry i / span.  The node kind is known only after parsing
        // (links non-empty), so parse with the link-candidate and re-derive for leaves.
        const uint64_t sub64 = i / span;
        const uint32_t sub_link = sub64 < 256 ? uint32_t(sub64) : 0xfffffffeu;
        const uint32_t sub_leaf = i < 256 ? uint32_t(i) : 0xfffffffeu;
        AmtNode nd;
        // First assume a link node unless height == 0 (span == 1 makes both candidates equal).
        amt_read_node(r, bw, vkind, height == 0 ? sub_leaf : sub_link, nd);
        if (node_off == 0) r.finish();  // a child block holds exactly one node; the root's tail was checked by load
        if (!r.ok()) return IPCFP_ST_ERR_DECODE;
        if (nd.nlinks == 0) {
            // Node::Leaf — `vals.get(i)`; at height > 0 the wanted ordinal must be re-derived for i
            if (i >= nd.width || !nd.bit(uint32_t(i))) return IPCFP_ST_NOT_FOUND;
            if (height != 0) {
                Rd r2 = open_block(w, block);
                r2.pos = node_off;
                amt_read_node(r2, bw, vkind, uint32_t(i), nd);
            }
            loc.block = block;
            loc.off = nd.want_off;
            loc.len = nd.want_len;
            return IPCFP_ST_TRUE;
        }
        if (height == 0) return IPCFP_ST_ERR_DECODE;  // link node at height 0
        if (sub64 >= nd.width || !nd.bit(uint32_t(sub64))) return IPCFP_ST_NOT_FOUND;
        const CidKey key = r.key_any(nd.want_off, nd.want_len);
        const uint32_t child = witness_find(w, key);
        if (child == kNoBlock) return IPCFP_ST_ERR_MISSING_BLOCK;
        block = child;
        node_off = 0;
        i = i % span;
        height -= 1;
    }
}

// Amt::load(root).get(index) in one go.  Identical outcomes to amt_load + amt_get; a height-0 tree
// whose root is a leaf (the usual events AMT: ≤ 32 events) is parsed once instead of twice.
__device__ __forceinline__ uint32_t amt_load_get(const WitnessView& w, const CidKey& root, int version, int vkind,
                                                 uint64_t index, ValueLoc& loc) {
    AmtRootInfo info;
    AmtNode nd;
    const uint32_t hint = index < 256 ? uint32_t(index) : 0xfffffffeu;
    const uint32_t st = amt_load(w, root, version, vkind, info, hint, &nd);
    if (st != IPCFP_ST_TRUE) return st;
    if (info.height == 0 && nd.nlinks == 0) {
        if (index == ~0ULL) return IPCFP_ST_ERR;
        if (index >= nd.width || !nd.bit(uint32_t(index))) return IPCFP_ST_NOT_FOUND;
        loc.block = info.block;
        loc.off = nd.want_off;
        loc.len = nd.want_len;
        return IPCFP_ST_TRUE;
    }
    return amt_get(w, info, vkind, index, loc);
}

// ---------------------------------------------------------------------------
// HAMT
// ---------------------------------------------------------------------------
// Hamt::load_with_bit_width(root, bw).get(key): TRUE ⇒ loc set; NOT_FOUND; ERR_*.
__device__ __forceinline__ uint32_t hamt_get(const WitnessView& w, const CidKey& root, uint32_t bit_width, int vkind,
                                             const uint8_t* key, uint32_t key_len, ValueLoc& loc) {
    if (bit_width < 1 || bit_width > 8) return IPCFP_ST_ERR_DECODE;
    uint32_t h[8];
    sha256::hash_bytes(key, key_len, h);
    uint32_t consumed = 0;
    uint32_t block = witness_find(w, root);
    if (block == kNoBlock) return IPCFP_ST_ERR_MISSING_BLOCK;
    for (;;) {
        Rd r = open_block(w, block);
        // HashBits::next happens after the node is decoded; compute idx first only when bits remain
        const bool depth_ok = consumed + bit_width <= 256;
        const uint32_t idx = depth_ok ? sha256::take_bits(h, consumed, bit_width) : 0;
        // ---- decode the whole node: [bitfield bytes, [pointer…]] ----
        r.expect_array(2);
        uint32_t bo, bl;
        r.read_bytes(bo, bl);
        if (r.ok() && bl > 32) r.fail();
        if (!r.ok()) return IPCFP_ST_ERR_DECODE;
        // big-endian integer, leading zeros stripped: byte k (from the END) holds bits 8k..8k+7
        bool bit_set = false;
        uint32_t rank = 0;
        for (uint32_t k = 0; k < bl; ++k) {
            const uint32_t byte = r.at(bo + bl - 1 - k);
            const uint32_t lo = 8 * k;
            if (idx >= lo + 8) rank += __popc(byte);
            else if (idx >= lo) {
                rank += __popc(byte & ((1u << (idx - lo)) - 1u));
                bit_set = (byte >> (idx - lo)) & 1u;
            }
        }
        const uint64_t np = r.read_array();
        if (!r.ok()) return IPCFP_ST_ERR_DECODE;
        // outcome of the wanted pointer, filled while scanning
        bool ptr_is_link = false;
        uint32_t link_off = 0, link_len = 0;
        bool found = false;
        ValueLoc hit{};
        for (uint64_t pi = 0; pi < np && r.ok(); ++pi) {
            const bool wanted = bit_set && pi == rank;
            const uint32_t b0 = r.peek();
            if ((b0 >> 5) == 6) {
                uint32_t o, l;
                r.read_link(o, l);
                if (wanted) {
                    ptr_is_link = true;
                    link_off = o;
                    link_len = l;
                }
            } else if ((b0 >> 5) == 4) {
                const uint64_t nkv = r.read_array();
                for (uint64_t k = 0; k < nkv && r.ok(); ++k) {
                    uint32_t ko, kl, vstart;
                    if (!(vkind == VK_ACTOR_STATE && actor_entry_fast(r, ko, kl, vstart))) {
                        r.expect_array(2);
                        r.read_bytes(ko, kl);
                        vstart = r.pos;
                        check_value(r, vkind);
                    }
                    if (wanted && r.ok() && !found && kl == key_len) {
                        bool eq = true;
                        for (uint32_t c = 0; c < kl; ++c) eq &= r.at(ko + c) == key[c];
                        if (eq) {
                            found = true;
                            hit.block = block;
                            hit.off = vstart;
                            hit.len = r.pos - vstart;
                        }
                    }
                }
            } else {
                r.fail();
            }
        }
        r.finish();
        if (!r.ok()) return IPCFP_ST_ERR_DECODE;
        if (!depth_ok) return IPCFP_ST_ERR_MAX_DEPTH;
        consumed += bit_width;
        if (!bit_set) return IPCFP_ST_NOT_FOUND;
        if (uint64_t(rank) >= np) return IPCFP_ST_ERR_DECODE;
        if (!ptr_is_link) {
            if (!found) return IPCFP_ST_NOT_FOUND;
            loc = hit;
            return IPCFP_ST_TRUE;
        }
        const CidKey ck = r.key_any(link_off, link_len);
        block = witness_find(w, ck);
        if (block == kNoBlock) return IPCFP_ST_ERR_MISSING_BLOCK;
    }
}

// ---------------------------------------------------------------------------
// A bucket of a storage HAMT searched with plain 8-byte reads (round 6).  The node table has validated the node and
// type-checked its values (kinds_ok): what is left for a claim is to find its 32-byte key among ≤ 3 entries spelled
// `82 58 20 <key> <value>`, value = `8n` | `98 nn` and n elements of one or two bytes.  The windowed reader's item heads
// (≈ 80 instructions each, four per entry) and the walk over the MATCHED value — which left_pad_32 walks again — were
// 0.4 of configs[4]'s 0.88 ms kernel (profiles/r06_experiments.md).  Reads may run ≤ 8 bytes past the block: the arena has
// blocks on 128-byte lines and 256 bytes of tail slack (witness.cpp kTailSlack).
// 1: found, `vstart` = the value's offset in the block; 0: not in the bucket; 2: spelled some other way — take the reader.
// ---------------------------------------------------------------------------
__device__ __forceinline__ uint64_t raw_ld64(const uint8_t* p) {
    uint64_t x;
    __builtin_memcpy(&x, p, 8);
    return x;
}

struct Raw16 {
    uint64_t lo, hi;
};
__device__ __forceinline__ Raw16 raw_ld128(const uint8_t* p) {
    Raw16 x;
    __builtin_memcpy(&x, p, 16);
    return x;
}

__device__ __forceinline__ uint32_t bucket_find32_raw(const uint8_t* __restrict__ g, uint32_t blen, uint32_t off, const uint64_t kw[4],
                                                      uint32_t& vstart) {
    // An entry's head and key are the 36 bytes from the byte BEFORE it on (the bucket's own head, for the first entry): three
    // 16-byte reads.  What bounds this kernel is the number of load instructions whose 64 lanes go to 64 different lines —
    // the CU's L1 looks up one line per clock, 112 such loads per wavefront were its 545 µs (profiles/r06_experiments.md) —
    // so the same bytes in fewer, wider loads.
    uint32_t nkv = 0, pos = off + 1u;
    for (uint32_t k = 0; k == 0 || k < nkv; ++k) {
        if (pos + 36u > blen) return 2;
        const Raw16 a = raw_ld128(g + pos - 1u), b = raw_ld128(g + pos + 15u), c = raw_ld128(g + pos + 31u);
        if (k == 0) {
            const uint32_t hb = uint32_t(a.lo) & 0xffu;
            if (hb < 0x81u || hb > 0x83u) return 2;
            nkv = hb - 0x80u;
        }
        if ((uint32_t(a.lo >> 8) & 0xffffffu) != 0x205882u) return 2;
        // key byte i = stream byte 4 + i: word j = the high half of stream word j and the low half of word j + 1
        const uint64_t diff = (((a.lo >> 32) | (a.hi << 32)) ^ kw[0]) | (((a.hi >> 32) | (b.lo << 32)) ^ kw[1]) |
                              (((b.lo >> 32) | (b.hi << 32)) ^ kw[2]) | (((b.hi >> 32) | (c.lo << 32)) ^ kw[3]);
        pos += 35u;
        if (diff == 0) {
            vstart = pos;
            return 1;
        }
        if (k + 1u == nkv) break;
        const uint32_t hv = g[pos];  // the value of another key: over it
        uint32_t n;
        if (hv >= 0x80u && hv < 0x98u) {
            n = hv - 0x80u;
            pos += 1u;
        } else if (hv == 0x98u) {
            n = g[pos + 1u];
            pos += 2u;
        } else {
            return 2;
        }
        uint32_t cur = 0, bad = 0;
        uint32_t q = n >> 2;
        for (; q && pos < blen; --q) {
            const uint64_t w8 = raw_ld64(g + pos);
            vec_u8_take<4>(uint32_t(w8), uint32_t(w8 >> 32), cur, pos, bad);
        }
        if (q) return 2;
        if (n & 3u) {
            const uint64_t w8 = raw_ld64(g + pos);
            const uint32_t r = n & 3u;
            if (r == 1) vec_u8_take<1>(uint32_t(w8), uint32_t(w8 >> 32), cur, pos, bad);
            else if (r == 2) vec_u8_take<2>(uint32_t(w8), uint32_t(w8 >> 32), cur, pos, bad);
            else vec_u8_take<3>(uint32_t(w8), uint32_t(w8 >> 32), cur, pos, bad);
        }
        if (bad || pos > blen) return 2;
    }
    return 0;
}

// ---------------------------------------------------------------------------
// HAMT get over the node table (hamt_table.h): same outcomes as hamt_get above, or kTablePunt when the walk meets a block
// the table does not cover (the caller then walks).  `kbit`: the HK_* bit of the HAMT's value kind.
// ---------------------------------------------------------------------------
constexpr uint32_t kTablePunt = 0xfdu;  // not an ipcfp_status_t

__device__ __forceinline__ uint32_t table_hamt_get(const WitnessView& w, const HamtNodeRec* __restrict__ table, const CidKey& root,
                                                   uint32_t bit_width, uint32_t kbit, const uint8_t* key, uint32_t key_len,
                                                   ValueLoc& loc, uint32_t root_block = kNoBlock,
                                                   const uint32_t* __restrict__ root_children = nullptr,
                                                   const uint64_t* key_words = nullptr) {
    // `key_words` (optional, key_len == 32): the key's bytes as four little-endian words the caller holds in registers — the
    // hash and the bucket's compare then read nothing of `key`.
    // `root_block` / `root_children` (optional): the root's block id and the blocks behind its pointers as SOMEBODY ELSE has
    // already resolved them (kNoBlock where not: the link is then read and looked up here) — the 256 storage proofs of one
    // contract all start at the same root and step through one of its 32 links (verify_storage.hip k_storage_run_children).
    if (bit_width < 1 || bit_width > 8) return IPCFP_ST_ERR_DECODE;
    uint32_t h[8];
    if (key_words && key_len == 32) sha256::hash32_words(key_words, h);
    else sha256::hash_bytes(key, key_len, h);
    uint32_t consumed = 0;
    uint32_t block = root_block != kNoBlock ? root_block : witness_find(w, root);
    if (block == kNoBlock) return IPCFP_ST_ERR_MISSING_BLOCK;
    for (;;) {
        const HamtNodeRec* rec = table + block;
        const uint4 rh = *reinterpret_cast<const uint4*>(rec);  // the record's first 16 bytes in one read
        const uint32_t head = rh.x;                              // status | kinds_ok << 8 | np << 16
        const uint32_t rec_std_links = rh.y;
        if ((head & 0xffu) != 1u) return kTablePunt;
        if (!((head >> 8) & kbit)) return IPCFP_ST_ERR_DECODE;  // a value of another type in one of the node's buckets
        const uint32_t np = (head >> 16) & 0xffu;
        // HashBits::next happens after the node is decoded
        if (consumed + bit_width > 256) return IPCFP_ST_ERR_MAX_DEPTH;
        const uint32_t idx = sha256::take_bits(h, consumed, bit_width);
        consumed += bit_width;
        const uint64_t bf = uint64_t(rh.z) | (uint64_t(rh.w) << 32);
        if (idx >= 64u || !((bf >> idx) & 1ull)) return IPCFP_ST_NOT_FOUND;
        const uint32_t rank = uint32_t(__popcll(bf & ((1ull << idx) - 1ull)));
        if (rank >= np) return IPCFP_ST_ERR_DECODE;
        if (root_children && consumed == bit_width && ((rec_std_links >> rank) & 1u)) {  // the root's link, resolved per run
            const uint32_t cb = root_children[rank];
            if (cb != kNoBlock) {
                block = cb;
                continue;
            }
        }
        const uint32_t off = rec->ptr_off[rank];
        const uint8_t* g = w.arena + w.off[block];
        CidKey link;
        if ((rec_std_links >> rank) & 1u) {
            // the standard 43-byte link: its 38 CID bytes lie at off + 5
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
                if (!r.ok()) return kTablePunt;  // (cannot happen: the table validated it)
                link = r.key_any(o, l);
            } else {
                // a bucket: `[[key, value]…]`, validated; find the key, skip the values
                if (kbit == HK_VEC_U8 && key_len == 32) {  // a storage slot: plain reads first
                    uint64_t kw[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) kw[j] = key_words ? key_words[j] : raw_ld64(key + 8 * j);
                    uint32_t vstart = 0;
                    const uint32_t f = bucket_find32_raw(g, w.len[block], off, kw, vstart);
                    if (f == 0) return IPCFP_ST_NOT_FOUND;
                    if (f == 1) {
                        loc.block = block;
                        loc.off = vstart;
                        loc.len = w.len[block] - vstart;  // (the value ends before the block does; its reader stops at its own end)
                        return IPCFP_ST_TRUE;
                    }
                }
                const uint64_t nkv = r.read_array();
                for (uint64_t k = 0; k < nkv && r.ok(); ++k) {
                    r.expect_array(2);
                    uint32_t ko, kl;
                    r.read_bytes(ko, kl);
                    const uint32_t vstart = r.pos;
                    const bool eq = r.ok() && kl == key_len && r.equal_bytes(ko, key, kl);
                    // Behind the key: the value, which the table has type-checked as `kbit`'s kind — so the typed walk
                    // reaches its end, in a fraction of the generic skip's instructions (a Vec<u8> of 32 elements is 33 item
                    // headers the generic way, ≈ 2 k instructions per entry per claim, and eight-byte steps the typed way).
                    if (kbit == HK_VEC_U8) check_vec_u8(r);
                    else if (kbit == HK_ACTOR_STATE) check_actor_state(r);
                    else r.skip();
                    if (eq && r.ok()) {
                        loc.block = block;
                        loc.off = vstart;
                        loc.len = r.pos - vstart;
                        return IPCFP_ST_TRUE;
                    }
                }
                return r.ok() ? uint32_t(IPCFP_ST_NOT_FOUND) : kTablePunt;
            }
        }
        block = witness_find(w, link);
        if (block == kNoBlock) return IPCFP_ST_ERR_MISSING_BLOCK;
    }
}

}  // namespace ipcfp
