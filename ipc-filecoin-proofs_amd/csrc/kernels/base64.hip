// csrc/kernels/base64.hip — base64 (RFC 4648 standard alphabet, canonical padding) → witness arena.
//
// Replaces `deserialize_base64` (src/proofs/common/bundle.rs:30-37: `B64.decode(s)` with
// base64 0.21's STANDARD engine) for every `ProofBlock.data` of a bundle at once.  The JSON text is in
// HBM; the host has located each data string (offset, length) and — from the length and the last two
// characters — knows the decoded size, so the arena layout exists before the kernel runs.
//
// One lane per 16 characters → 12 bytes (three aligned dword stores; every block starts on a 128-byte
// line so 12·u is 4-byte aligned).  The lane finds its block by binary search over the per-block unit
// prefix.  The kernel is a pure stream: 4 B read per 3 B written.
//
// Validity (any violation makes the whole bundle an Err, as the serde error does): length % 4 == 0;
// only alphabet characters, except one or two '=' at the very end; the bits dropped by the padding are
// zero (`DecodeError::InvalidLastSymbol`).
#include <hip/hip_runtime.h>

#include "../common.h"
#include "blake2b_dev.h"
#include "launch.h"

namespace ipcfp {

// Four characters at once (SWAR on a dword; every byte must be < 0x80, checked by the caller through
// the `bad` word): per-byte range tests with the carry trick `(x + (0x80 - k)) & 0x80`, then the sextet
// is x + offset(class) with the class offsets accumulated per byte without inter-byte carries.
//   '+' 0x2b → 62 (+19)   '/' 0x2f → 63 (+16)   '0'..'9' → 52.. (+4)   'A'..'Z' → 0.. (-65)   'a'..'z' → 26.. (-71)
// Returns the four sextets in the four bytes of the result; `bad` accumulates 0x80 bits for bytes
// outside the alphabet.
__device__ __forceinline__ uint32_t b64_sextets4(uint32_t x, uint32_t& bad) {
    const uint32_t H = 0x80808080u, L = 0x01010101u;
    bad |= x & H;
    const uint32_t y = x & ~H;
    auto ge = [&](uint32_t k) { return ((y + (0x80u - k) * L) & H) >> 7; };  // 0/1 per byte
    const uint32_t g2b = ge(0x2b), g2c = ge(0x2c), g2f = ge(0x2f), g30 = ge(0x30), g3a = ge(0x3a), g41 = ge(0x41),
                   g5b = ge(0x5b), g61 = ge(0x61), g7b = ge(0x7b);
    // class indicators (0/1 per byte), mutually exclusive
    const uint32_t plus = g2b & ~g2c, slash = g2f & ~g30, digit = g30 & ~g3a, upper = g41 & ~g5b, lower = g61 & ~g7b;
    const uint32_t valid = plus | slash | digit | upper | lower;
    bad |= (valid ^ L) << 7;
    // sextet = y + offset (mod 256 per byte); offsets as positive bytes: +19, +16, +4, 256-65 = 191, 256-71 = 185.
    // y < 0x80 and the class is exclusive, so y + off never needs a carry OUT of the byte to be right
    // mod 256 — but it may produce one, so add the low 7 bits and patch the top bit (carry-less byte add).
    const uint32_t off = plus * 19u + slash * 16u + digit * 4u + upper * 191u + lower * 185u;
    const uint32_t sum = (y & 0x7f7f7f7fu) + (off & 0x7f7f7f7fu);
    return (sum ^ (off & H)) & 0x3f3f3f3fu;
}

// four sextet bytes (little-endian: s0 in byte 0) → three data bytes, little-endian in the result
__device__ __forceinline__ uint32_t b64_pack(uint32_t s) {
    const uint32_t s0 = s & 0xffu, s1 = (s >> 8) & 0xffu, s2 = (s >> 16) & 0xffu, s3 = s >> 24;
    const uint32_t b0 = (s0 << 2) | (s1 >> 4);
    const uint32_t b1 = ((s1 & 15u) << 4) | (s2 >> 2);
    const uint32_t b2 = ((s2 & 3u) << 6) | s3;
    return b0 | (b1 << 8) | (b2 << 16);
}

struct B64Span {
    uint64_t src;   // byte offset of the string body in the JSON text
    uint32_t len;   // characters
    uint32_t unit0; // index of this block's first 16-character unit
};

// One lane per 16-character unit; a wavefront owns 64 CONSECUTIVE units, so its first lane finds the
// block by binary search (wave-uniform → scalar loads) and the other lanes walk forward from there —
// a wavefront spans a handful of blocks at most.
__global__ __launch_bounds__(256) void k_base64_decode(const uint8_t* __restrict__ text, const B64Span* __restrict__ spans,
                                                       uint32_t n_blocks, uint32_t n_units,
                                                       const uint64_t* __restrict__ dst_off, uint8_t* __restrict__ arena,
                                                       unsigned long long* __restrict__ first_bad) {
    const uint32_t u = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t u_first = __builtin_amdgcn_readfirstlane(u & ~63u);
    if (u_first >= n_units) return;
    // last block whose unit0 <= u_first (uniform)
    uint32_t lo = 0, hi = n_blocks;
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (spans[mid].unit0 <= u_first) lo = mid;
        else hi = mid;
    }
    if (u >= n_units) return;
    // blocks with zero units (empty data) share unit0 with their successor: skip them as well
    while (lo + 1 < n_blocks && spans[lo + 1].unit0 <= u) ++lo;
    const B64Span sp = spans[lo];
    const uint32_t k = u - sp.unit0;          // unit inside the block
    const uint32_t c0 = k * 16u;              // first character of this unit
    const uint32_t nchar = min(16u, sp.len - c0);
    // 16 characters at an arbitrary byte offset: three aligned 8-byte words, funnel-shifted
    const uint64_t a = sp.src + c0;
    const uint64_t* wp = reinterpret_cast<const uint64_t*>(text + (a & ~7ull));
    const uint32_t sh = uint32_t(a & 7ull) * 8u;
    const uint64_t w0 = wp[0], w1 = wp[1], w2 = wp[2];  // the text buffer has 32 bytes of tail slack
    const uint64_t q0 = sh ? (w0 >> sh) | (w1 << (64u - sh)) : w0;
    const uint64_t q1 = sh ? (w1 >> sh) | (w2 << (64u - sh)) : w1;
    uint32_t grp[4] = {uint32_t(q0), uint32_t(q0 >> 32), uint32_t(q1), uint32_t(q1 >> 32)};
    uint32_t bad = 0, pads = 0;
    uint8_t* dst = arena + dst_off[lo] + uint64_t(k) * 12u;
    if (c0 + 16u < sp.len) {
        // interior unit: 16 alphabet characters → 12 bytes → three aligned dwords
        const uint32_t o0 = b64_pack(b64_sextets4(grp[0], bad)), o1 = b64_pack(b64_sextets4(grp[1], bad)),
                       o2 = b64_pack(b64_sextets4(grp[2], bad)), o3 = b64_pack(b64_sextets4(grp[3], bad));
        uint32_t* d4 = reinterpret_cast<uint32_t*>(dst);
        d4[0] = o0 | (o1 << 24);
        d4[1] = (o1 >> 8) | (o2 << 16);
        d4[2] = (o2 >> 16) | (o3 << 8);
    } else {
        // the block's last unit: 4..16 characters, '=' may close the final group ("xx==" or "xxx=")
        if (sp.len & 3u) bad = 0x80u;  // the host never schedules such a block, but stay safe
        uint32_t out[4] = {0, 0, 0, 0};
        for (uint32_t g = 0; g * 4u < nchar; ++g) {
            uint32_t w = grp[g];
            const bool final_group = g * 4u + 4u >= nchar;
            if (final_group) {
                const bool p3 = (w >> 24) == '=';
                const bool p2 = p3 && ((w >> 16) & 0xffu) == '=';
                pads = p2 ? 2u : p3 ? 1u : 0u;
                if (p2) w = (w & 0x0000ffffu) | (uint32_t('A') << 16) | (uint32_t('A') << 24);
                else if (p3) w = (w & 0x00ffffffu) | (uint32_t('A') << 24);
            }
            const uint32_t d = b64_pack(b64_sextets4(w, bad));
            // bits that the padding drops must be zero (DecodeError::InvalidLastSymbol): with the pad
            // characters read as 'A' that is "the dropped byte is zero"
            if (final_group && pads == 1 && ((d >> 16) & 0xffu)) bad |= 0x80u;
            if (final_group && pads == 2 && ((d >> 8) & 0xffu)) bad |= 0x80u;
            out[g] = d;
        }
        const uint32_t dw[3] = {out[0] | (out[1] << 24), (out[1] >> 8) | (out[2] << 16), (out[2] >> 16) | (out[3] << 8)};
        const uint32_t nbytes = (nchar >> 2) * 3u - pads;
        for (uint32_t i = 0; i < nbytes; ++i) dst[i] = uint8_t(dw[i >> 2] >> ((i & 3u) * 8u));
    }
    if (bad) atomicMin(first_bad, (unsigned long long)lo);
}

// ---------------------------------------------------------------------------------------------
// `ProofBlock.cid` as serde_json writes it: `[1,113,160,228,2,32, …]` — one lane per block parses the
// array body located by the host into the 40-byte CID slot and validates it the way
// `Cid::try_from(bytes)` does (CIDv0 = bare sha2-256 multihash, or v1 ‖ codec ‖ multihash with an
// exact digest length).  Grammar of the body: ws* ( u8 ( ws* ',' ws* u8 )* )? ws*, u8 = 0 | [1-9][0-9]*
// ≤ 255.  Anything else — a sign, a fraction, a nested value — is an Err in serde as well.
// error codes: 1 = not a CID byte array, 2 = a valid CID longer than the 40-byte slot (engine limit)
// ---------------------------------------------------------------------------------------------
struct CidSpan {
    uint64_t src;
    uint32_t len;
    uint32_t pad;
};

__device__ __forceinline__ bool uvarint_dev(const uint8_t* p, uint32_t n, uint32_t& pos, uint64_t& v) {
    // unsigned-varint as `Cid::read_bytes` reads it: at most 9 bytes, and a non-minimal form (a zero final byte
    // after the first) is rejected — the same rule as Rd::cid_ok and the host's cidstr.cpp
    v = 0;
    for (int i = 0; i < 9 && pos < n; ++i) {
        const uint8_t b = p[pos++];
        v |= uint64_t(b & 0x7f) << (7 * i);
        if (!(b & 0x80)) return !(b == 0 && i > 0);
    }
    return false;
}

__global__ __launch_bounds__(256) void k_parse_cid_arrays(const uint8_t* __restrict__ text, const CidSpan* __restrict__ spans,
                                                          uint32_t n, uint8_t* __restrict__ cids,
                                                          unsigned long long* __restrict__ first_bad) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const CidSpan sp = spans[t];
    uint8_t cid[104];  // the longest well-formed CID: 4 varints (≤ 9 B each) + 64 digest bytes
    uint32_t count = 0, err = 0;
    // states: 0 expect number (or nothing if the array is empty), 1 in number, 2 after number, 3 after comma
    uint32_t state = 0, value = 0, digits = 0;
    bool lead_zero = false;
    const uint8_t* p = text + sp.src;
    for (uint32_t i = 0; i <= sp.len && !err; ++i) {
        const uint32_t c = i < sp.len ? p[i] : ',' + 256u;  // sentinel closes a pending number
        const bool ws = c == ' ' || c == '\t' || c == '\n' || c == '\r';
        const bool digit = c - '0' < 10u;
        if (state == 1) {
            if (digit) {
                if (lead_zero) err = 1;  // "01"
                value = value * 10u + (c - '0');
                if (++digits > 3 || value > 255u) err = 1;
                continue;
            }
            if (count < 104) cid[count] = uint8_t(value);
            ++count;
            state = 2;
        }
        if (c == ',' + 256u) {
            if (state == 3) err = 1;              // trailing comma
            break;
        }
        if (ws) continue;
        if (digit) {
            if (state == 2) err = 1;              // two numbers without a comma
            state = 1;
            value = c - '0';
            digits = 1;
            lead_zero = c == '0';
        } else if (c == ',') {
            if (state != 2) err = 1;
            state = 3;
        } else {
            err = 1;
        }
    }
    if (!err) {
        if (count > 104) err = 1;
        else if (!(count == 34 && cid[0] == 0x12 && cid[1] == 0x20)) {
            uint32_t pos = 0;
            uint64_t version, codec, code, size;
            if (!uvarint_dev(cid, count, pos, version) || version != 1 || !uvarint_dev(cid, count, pos, codec) ||
                !uvarint_dev(cid, count, pos, code) || !uvarint_dev(cid, count, pos, size) || size > 64 ||
                count - pos != size)
                err = 1;
        }
    }
    if (err) {
        atomicMin(first_bad, ((unsigned long long)t << 2) | err);
        return;
    }
    uint8_t* out = cids + size_t(t) * IPCFP_CID_SLOT;
    if (count > IPCFP_CID_SLOT) {  // longer than the slot: the fold ff | len | blake2b-256(cid) (include/ipcfp.h "CIDs")
        uint64_t d[4];
        blake2b256_small(cid, count, d);
        out[0] = 0xff;
        out[1] = uint8_t(count);
        for (uint32_t i = 0; i < 32; ++i) out[2 + i] = uint8_t(d[i >> 3] >> (8u * (i & 7u)));
        for (uint32_t i = 34; i < IPCFP_CID_SLOT; ++i) out[i] = 0;
        return;
    }
    for (uint32_t i = 0; i < IPCFP_CID_SLOT; ++i) out[i] = i < count ? cid[i] : 0;
}

int launch_base64_decode(ipcfp_ctx* ctx, const uint8_t* text_d, const void* spans_d, uint32_t n_blocks, uint32_t n_units,
                         const uint64_t* dst_off_d, uint8_t* arena_d, unsigned long long* first_bad_d) {
    if (n_units == 0) return IPCFP_OK;
    ProfileScope prof(ctx, IPCFP_K_BASE64);
    hipLaunchKernelGGL(k_base64_decode, dim3(div_up(n_units, 256)), dim3(256), 0, ctx->stream, text_d,
                       static_cast<const B64Span*>(spans_d), n_blocks, n_units, dst_off_d, arena_d, first_bad_d);
    IPCFP_HIP(ctx, hipGetLastError());
    return IPCFP_OK;
}

int launch_parse_cid_arrays(ipcfp_ctx* ctx, const uint8_t* text_d, const void* spans_d, uint32_t n, uint8_t* cids_d,
                            unsigned long long* first_bad_d) {
    if (n == 0) return IPCFP_OK;
    hipLaunchKernelGGL(k_parse_cid_arrays, dim3(div_up(n, 256)), dim3(256), 0, ctx->stream, text_d,
                       static_cast<const CidSpan*>(spans_d), n, cids_d, first_bad_d);
    IPCFP_HIP(ctx, hipGetLastError());
    return IPCFP_OK;
}

}  // namespace ipcfp
