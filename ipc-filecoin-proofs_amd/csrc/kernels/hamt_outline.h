// csrc/kernels/hamt_outline.h — the OUTLINE of a state-tree HAMT node (`Hamt<_, ActorState>`: src/proofs/common/decode.rs:29-39):
// where every pointer, every bucket entry's ActorState, its second link and its delegated-address item start — item headers
// only, for the spellings every encoder writes:
//
//   82 | 4x bitfield | 8n / 98 n pointers | pointer = link (d8 2a 4l/58 l …)  |  bucket 8c of c entries
//   entry = 82 | key: 4k … / 58 k … | 85 | link (code) | link (state) | uint (sequence) | bytes (balance) | f6 / bytes (address)
//
// Two ways to find it, same result:
//   * outline_sequential — one reader front to back (round 4's first form: lane 0 of the group, ≈ 4.6 k instructions for a
//     4-5 KB node of ≈ 100 entries; the level-by-level walk of configs[3] was bound by exactly these instructions,
//     profiles/r04_experiments.md);
//   * the PARALLEL outline, built from the pieces below by 32 lanes (hamt_levels.hip k_hamt_lv_parse_actor): every entry
//     contains the three bytes `85 d8 2a` (its ActorState's array header and the tag of its first link), so the lanes scan
//     the node for them side by side (anchors), parse every anchored entry FORWARD to its end, then every GAP between one
//     entry's end and the next anchor (key, bucket header, link pointers) — and accept the node only if the pieces tile it
//     exactly.  An anchor that is no entry (the three bytes inside a digest or a balance) breaks the tiling: the
//     sequential outline then takes the node, so the outcome is the sequential one by construction.
//
// Everything here is plain C++ over a byte buffer (`S`: 8-byte aligned, readable 24 bytes beyond `len`) so that the same
// code runs in LDS on the device and in tests/native/outline_harness.cpp on the host (tests/test_hamt_outline.py).
#pragma once
#include <cstdint>

#if defined(__HIPCC__)
#define IPCFP_OL_FN __host__ __device__ __forceinline__
#else
#define IPCFP_OL_FN inline
#endif

namespace ipcfp {
namespace outline {

constexpr uint32_t kMaxPointers = 32, kMaxEntries = 96;

// the 8 bytes at S[p, p + 8) as a little-endian word (two aligned word reads)
IPCFP_OL_FN uint64_t peek64(const uint8_t* S, uint32_t p) {
    const uint64_t* q = reinterpret_cast<const uint64_t*>(S + (p & ~7u));
    const uint64_t lo = q[0], hi = q[1];
    const uint32_t sh = (p & 7u) * 8u;
    return (lo >> sh) | ((hi << 1) << (63u - sh));
}

// encoded length of the link whose first bytes are w (d8 2a | 4l / 58 l | …), 0: not that outline
IPCFP_OL_FN uint32_t link_len(uint64_t w) {
    if ((w & 0xffffull) != 0x2ad8ull) return 0u;
    const uint32_t hb = uint32_t(w >> 16) & 0xffu;
    if (hb >= 0x41u && hb <= 0x57u) return 3u + (hb - 0x40u);
    if (hb == 0x58u) return 4u + (uint32_t(w >> 24) & 0xffu);
    return 0u;
}

struct Header {
    bool ok;
    uint64_t bf;
    uint32_t np, pos0;  // pointers, where the first one starts
};
IPCFP_OL_FN Header header(const uint8_t* S, uint32_t len) {
    Header h{false, 0, 0, 0};
    const uint64_t w0 = peek64(S, 0);
    bool ok = (w0 & 0xffu) == 0x82u;
    const uint32_t bl = (uint32_t(w0 >> 8) & 0xffu) - 0x40u;  // bitfield: bytes, at most 8
    ok = ok && bl <= 8u;
    if (ok && bl) h.bf = __builtin_bswap64(peek64(S, 2u)) >> (64u - 8u * bl);
    uint32_t pos = 2u + bl;
    if (ok) {
        const uint64_t w = peek64(S, pos);
        const uint32_t b = uint32_t(w) & 0xffu;
        if (b >= 0x80u && b <= 0x97u) {
            h.np = b - 0x80u;
            pos += 1u;
        } else if (b == 0x98u) {
            h.np = uint32_t(w >> 8) & 0xffu;
            pos += 2u;
        } else {
            ok = false;
        }
        ok = ok && h.np <= kMaxPointers && pos <= len;
    }
    h.ok = ok;
    h.pos0 = pos;
    return h;
}

// one entry from its `82`: key header → q (where the ActorState's 85 must be); false: not a key header
IPCFP_OL_FN bool key_end(uint64_t e0, uint32_t pos, uint32_t& q) {
    const uint32_t kb = uint32_t(e0 >> 8) & 0xffu;
    if (kb >= 0x40u && kb <= 0x57u) q = pos + 2u + (kb - 0x40u);
    else if (kb == 0x58u) q = pos + 3u + (uint32_t(e0 >> 16) & 0xffu);
    else return false;
    return true;
}

// the part of an entry behind its key, from the ActorState's array header at `a`: 85 | link | link | sequence | balance |
// address.  l2 / adr: where the second link and the address item start; end: the first byte behind the entry.
IPCFP_OL_FN bool entry_forward(const uint8_t* S, uint32_t a, uint32_t len, uint32_t& l2, uint32_t& adr, uint32_t& end) {
    const uint64_t v0 = peek64(S, a);
    const uint32_t l1 = link_len(v0 >> 8);
    // (a link that runs beyond the node: whatever lies there is not read — the entry could only end beyond the node too)
    const uint32_t ll2 = l1 && a + 1u + l1 < len ? link_len(peek64(S, a + 1u + l1)) : 0u;
    bool ok = (v0 & 0xffu) == 0x85u && l1 != 0u && ll2 != 0u;
    l2 = a + 1u + l1;
    uint32_t q = a + 1u + l1 + ll2;
    ok = ok && q < len;
    if (!ok) return false;
    const uint32_t sb = uint32_t(peek64(S, q)) & 0xffu;  // sequence: an unsigned integer in any width
    ok = sb <= 0x1bu;
    q += 1u + (sb < 0x18u ? 0u : (1u << ((sb - 0x18u) & 3u)));
    if (!ok || q >= len) return false;
    const uint64_t b0 = peek64(S, q);  // balance: bytes, at most 128, sign byte 0 / 1
    const uint32_t bb = uint32_t(b0) & 0xffu;
    uint32_t l, sign;
    if (bb >= 0x40u && bb <= 0x57u) {
        l = bb - 0x40u;
        sign = uint32_t(b0 >> 8) & 0xffu;
        q += 1u;
    } else if (bb == 0x58u) {
        l = uint32_t(b0 >> 8) & 0xffu;
        sign = uint32_t(b0 >> 16) & 0xffu;
        q += 2u;
    } else {
        return false;
    }
    ok = l <= 128u && (l == 0u || sign <= 1u);
    q += l;
    if (!ok || q >= len) return false;
    adr = q;
    const uint64_t a0 = peek64(S, q);  // delegated_address: None, or address bytes (checked by the lanes)
    const uint32_t ab = uint32_t(a0) & 0xffu;
    if (ab == 0xf6u) q += 1u;
    else if (ab >= 0x40u && ab <= 0x57u) q += 1u + (ab - 0x40u);
    else if (ab == 0x58u) q += 2u + (uint32_t(a0 >> 8) & 0xffu);
    else return false;
    end = q;
    return q <= len;
}

// ---- front to back ----------------------------------------------------------------------------------------------------
struct Result {
    uint32_t np, ne, links;  // pointers, bucket entries, bit p: pointer p is a link
    uint64_t bf;
};
// ptr[p]: where pointer p starts; val / l2 / adr[e]: entry e's ActorState header, second link, address item
// (max_entries: what val / l2v / adrv hold)
// endv / klen / first_of (nullable, for the entry table of hamt_levels.hip): where entry e ends, the length of its key (the key's
// bytes end where the ActorState starts), and for a bucket pointer p the index of its first entry.
IPCFP_OL_FN bool outline_sequential(const uint8_t* S, uint32_t len, Result& r, uint16_t* ptr, uint16_t* val, uint16_t* l2v,
                                    uint16_t* adrv, uint32_t max_entries = kMaxEntries, uint16_t* endv = nullptr, uint8_t* klen = nullptr,
                                    uint8_t* first_of = nullptr) {
    const Header h = header(S, len);
    bool ok = h.ok;
    uint32_t pos = h.pos0, ne = 0, links = 0;
    const uint32_t np = h.np;
    for (uint32_t p = 0; ok && p < np; ++p) {
        ptr[p] = uint16_t(pos);
        const uint64_t w = peek64(S, pos);
        const uint32_t b = uint32_t(w) & 0xffu;
        if (b == 0xd8u) {  // a link: its bytes are checked by the lanes
            const uint32_t ll = link_len(w);
            ok = ll != 0u;
            links |= 1u << p;
            pos += ll;
        } else if (b >= 0x80u && b <= 0x97u) {  // a bucket of b - 0x80 entries
            const uint32_t nkv = b - 0x80u;
            pos += 1u;
            ok = ne + nkv <= max_entries;
            if (first_of && nkv) first_of[p] = uint8_t(ne);  // (an empty bucket has no first entry)
            for (uint32_t k = 0; ok && k < nkv; ++k) {
                const uint64_t e0 = peek64(S, pos);
                uint32_t q = 0;
                if (!key_end(e0, pos, q) || (e0 & 0xffu) != 0x82u || q >= len) {
                    ok = false;
                    break;
                }
                uint32_t l2 = 0, adr = 0, end = 0;
                if (!entry_forward(S, q, len, l2, adr, end)) {
                    ok = false;
                    break;
                }
                val[ne] = uint16_t(q);
                l2v[ne] = uint16_t(l2);
                adrv[ne] = uint16_t(adr);
                if (endv) endv[ne] = uint16_t(end);
                if (klen) klen[ne] = uint8_t(q - pos - ((uint32_t(e0 >> 8) & 0xffu) == 0x58u ? 3u : 2u));
                ++ne;
                pos = end;
            }
        } else {
            ok = false;
        }
        ok = ok && pos <= len;
    }
    ok = ok && pos == len;  // nothing after the node
    r.np = np;
    r.ne = ne;
    r.links = links;
    r.bf = h.bf;
    return ok;
}

// ---- the pieces of the parallel outline ----------------------------------------------------------------------------------
// Anchors (`85 d8 2a`) whose first byte lies in [from, to), to ≤ len - 2: count them, or write their positions (ascending).
// The zero-byte trick finds the 0x85 bytes of a word exactly; the two bytes behind come out of the 16-byte window.
template <bool WRITE>
IPCFP_OL_FN uint32_t scan_anchors(const uint8_t* S, uint32_t from, uint32_t to, uint16_t* out, uint32_t cap) {
    uint32_t n = 0;
    for (uint32_t p = from & ~7u; p < to; p += 8u) {
        const uint64_t* q = reinterpret_cast<const uint64_t*>(S + p);
        const uint64_t x = q[0], y = q[1];
        const uint64_t z = x ^ 0x8585858585858585ull;
        uint64_t m = ~(((z & 0x7f7f7f7f7f7f7f7full) + 0x7f7f7f7f7f7f7f7full) | z | 0x7f7f7f7f7f7f7f7full);  // 0x80 where x has 0x85
        while (m) {
            const uint32_t k = uint32_t(__builtin_ctzll(m)) >> 3;
            m &= m - 1ull;
            const uint32_t at = p + k;
            if (at < from || at >= to) continue;
            const uint64_t next = k == 7u ? y : ((x >> (8u * (k + 1u))) | ((y << 1) << (55u - 8u * k)));  // bytes at + 1 …
            if ((next & 0xffffull) != 0x2ad8ull) continue;
            if (WRITE && n < cap) out[n] = uint16_t(at);
            ++n;
        }
    }
    return n;
}

// One pass: the lane's first FOUR anchors packed 16 bits apiece (a lane's share of a node holds two or three entries), the
// count of all of them returned — a lane that meets more makes its group take the two-pass form above.
IPCFP_OL_FN uint32_t scan_anchors_packed(const uint8_t* S, uint32_t from, uint32_t to, uint64_t& packed) {
    uint32_t n = 0;
    packed = 0;
    for (uint32_t p = from & ~7u; p < to; p += 8u) {
        const uint64_t* q = reinterpret_cast<const uint64_t*>(S + p);
        const uint64_t x = q[0], y = q[1];
        const uint64_t z = x ^ 0x8585858585858585ull;
        uint64_t m = ~(((z & 0x7f7f7f7f7f7f7f7full) + 0x7f7f7f7f7f7f7f7full) | z | 0x7f7f7f7f7f7f7f7full);
        while (m) {
            const uint32_t k = uint32_t(__builtin_ctzll(m)) >> 3;
            m &= m - 1ull;
            const uint32_t at = p + k;
            if (at < from || at >= to) continue;
            const uint64_t next = k == 7u ? y : ((x >> (8u * (k + 1u))) | ((y << 1) << (55u - 8u * k)));
            if ((next & 0xffffull) != 0x2ad8ull) continue;
            if (n < 4u) packed |= uint64_t(at) << (16u * n);
            ++n;
        }
    }
    return n;
}

// The buckets' counts must hop from header to header: entry e opens a bucket of c = gc[e] entries, so the c - 1 entries
// behind it open none and the one behind those does (or the entries end there).  Checked for every header by itself, this
// is the walk `e = 0; while (e < ne) e += gc[e]` landing on every header and on ne (entry 0 always opens a bucket: gap_walk).
IPCFP_OL_FN bool bucket_spans(const uint8_t* gc, uint32_t e, uint32_t ne) {
    const uint32_t c = gc[e];
    if (!c) return true;
    if (e + c > ne) return false;
    for (uint32_t j = 1; j < c; ++j)
        if (gc[e + j]) return false;
    return e + c == ne || gc[e + c] != 0;
}

// The gap in front of entry e — from `from` (the previous entry's end, or the first pointer) to the entry's anchor — or, with
// tail, behind the last entry up to the end of the node: link pointers and empty buckets, then (not tail) either a bucket
// header `8c` + the entry's `82 key`, or — continuing the previous entry's bucket — `82 key` alone.
//   n_ptr  pointers that START in the gap (links, empty buckets, the header);   count: c of the header, 0 = none
//   ptr / links (nullable): the second pass writes the pointers' positions from index `first` on
//   key_len: of the entry's key;   first_of (nullable, with ptr): first_of[p] = entry for the bucket header's pointer p
IPCFP_OL_FN bool gap_walk(const uint8_t* S, uint32_t from, uint32_t target, uint32_t len, bool tail, bool first_gap, uint32_t& n_ptr,
                          uint32_t& count, uint16_t* ptr, uint32_t first, uint32_t* links, uint32_t* key_len = nullptr,
                          uint8_t* first_of = nullptr, uint32_t entry = 0) {
    uint32_t pos = from, n = 0;
    count = 0;
    n_ptr = 0;
    for (uint32_t it = 0; it <= kMaxPointers + 1u; ++it) {
        if (tail && pos == len) {
            n_ptr = n;
            return true;
        }
        if (pos >= len || first + n > kMaxPointers) return false;
        const uint64_t w = peek64(S, pos);
        const uint32_t b = uint32_t(w) & 0xffu;
        if (b == 0xd8u) {
            const uint32_t ll = link_len(w);
            if (!ll || first + n >= kMaxPointers) return false;
            if (ptr) {
                ptr[first + n] = uint16_t(pos);
                *links |= 1u << (first + n);
            }
            ++n;
            pos += ll;
            continue;
        }
        if (b == 0x80u) {  // an empty bucket
            if (first + n >= kMaxPointers) return false;
            if (ptr) ptr[first + n] = uint16_t(pos);
            ++n;
            pos += 1u;
            continue;
        }
        if (tail) return false;  // entries behind the last anchor: there are none
        // `82` opens a bucket of two entries OR an entry: a bucket's header is followed by an entry's `82`, an entry's
        // `82` by its key's byte-string header — one byte of look-ahead tells them apart
        const bool header = b >= 0x81u && b <= 0x97u && (b != 0x82u || (uint32_t(w >> 8) & 0xffu) == 0x82u);
        if (header) {
            if (first + n >= kMaxPointers) return false;
            if (ptr) ptr[first + n] = uint16_t(pos);
            if (first_of) first_of[first + n] = uint8_t(entry);
            ++n;
            count = b - 0x80u;
            pos += 1u;
        } else if (n != 0u || first_gap) {
            return false;  // an entry can only follow its bucket's header or another entry
        }
        const uint64_t e0 = peek64(S, pos);
        uint32_t q = 0;
        if ((e0 & 0xffu) != 0x82u || !key_end(e0, pos, q)) return false;
        n_ptr = n;
        if (key_len) *key_len = q - pos - ((uint32_t(e0 >> 8) & 0xffu) == 0x58u ? 3u : 2u);
        return q == target;
    }
    return false;
}

}  // namespace outline
}  // namespace ipcfp
