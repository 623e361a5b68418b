// csrc/kernels/blake2b_dev.h — BLAKE2b compression for gfx950, one hash per lane.
//
// Algorithm: RFC 7693 §3.2 (the function behind multihash code 0xb220,
// `Code::Blake2b256`, reference call site src/proofs/events/utils.rs:65).
//
// CDNA4 mapping.  Blake2b is pure 64-bit integer work: per 128-byte chunk,
// 12 rounds × 8 G, each G = 6 u64 adds + 4 u64 xors + 4 rotates.  gfx950 has
// no 64-bit vector rotate, so on the 32-bit VALU a G costs ≈26 ops
// (add = v_add_co+v_addc, xor = 2×v_xor, rot32 = register rename,
// rot24/rot16 = 2×v_perm_b32/v_alignbit_b32, rot63 = 2×v_alignbit_b32):
// ≈2500 VALU ops per chunk ≈ 19.5 ops/byte.  The state of ONE hash has only
// 4-way (G-column) parallelism and the chunk chain is serial, so a wavefront
// runs 64 INDEPENDENT hashes, one per lane, with all 12 rounds unrolled so the
// sigma schedule is compile-time register naming (no LDS, no indexing).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace ipcfp {
namespace b2b {

// 64-bit rotates on 32-bit halves.  hipcc lowers a generic u64 rotate to a 64-bit
// shift + 32-bit shift + or; written on halves every rotate is two v_alignbit_b32
// (rot32 is a register rename).  alignbit(hi, lo, s) = ({hi,lo} >> s)[31:0].
template <int N>
__device__ __forceinline__ uint64_t rotr(uint64_t x) {
    const uint32_t lo = uint32_t(x), hi = uint32_t(x >> 32);
    uint32_t nlo, nhi;
    if constexpr (N == 32) {
        nlo = hi;
        nhi = lo;
    } else if constexpr (N < 32) {
        nlo = __builtin_amdgcn_alignbit(hi, lo, N);
        nhi = __builtin_amdgcn_alignbit(lo, hi, N);
    } else {
        nlo = __builtin_amdgcn_alignbit(lo, hi, N - 32);
        nhi = __builtin_amdgcn_alignbit(hi, lo, N - 32);
    }
    return (uint64_t(nhi) << 32) | nlo;
}

// 64-bit add.  MODE 0: let hipcc pick (v_lshl_add_u64 on gfx950).
// MODE 1: explicit v_add_co_u32 / v_addc_co_u32 pair (A/B-measured on MI355X).
template <int MODE>
__device__ __forceinline__ uint64_t add64(uint64_t a, uint64_t b) {
    if constexpr (MODE != 1) {
        return a + b;
    } else {
        uint32_t lo, hi;
        asm("v_add_co_u32 %0, vcc, %2, %4\n\tv_addc_co_u32 %1, vcc, %3, %5, vcc"
            : "=&v"(lo), "=v"(hi)
            : "v"(uint32_t(a)), "v"(uint32_t(a >> 32)), "v"(uint32_t(b)), "v"(uint32_t(b >> 32))
            : "vcc");
        return (uint64_t(hi) << 32) | lo;
    }
}

#define IPCFP_B2B_IV0 0x6a09e667f3bcc908ULL
#define IPCFP_B2B_IV1 0xbb67ae8584caa73bULL
#define IPCFP_B2B_IV2 0x3c6ef372fe94f82bULL
#define IPCFP_B2B_IV3 0xa54ff53a5f1d36f1ULL
#define IPCFP_B2B_IV4 0x510e527fade682d1ULL
#define IPCFP_B2B_IV5 0x9b05688c2b3e6c1fULL
#define IPCFP_B2B_IV6 0x1f83d9abfb41bd6bULL
#define IPCFP_B2B_IV7 0x5be0cd19137e2179ULL

__device__ __forceinline__ void init256(uint64_t h[8]) {
    h[0] = IPCFP_B2B_IV0 ^ 0x01010020ULL;  // digest_length = 32, fanout = depth = 1, no key
    h[1] = IPCFP_B2B_IV1;
    h[2] = IPCFP_B2B_IV2;
    h[3] = IPCFP_B2B_IV3;
    h[4] = IPCFP_B2B_IV4;
    h[5] = IPCFP_B2B_IV5;
    h[6] = IPCFP_B2B_IV6;
    h[7] = IPCFP_B2B_IV7;
}

// rotr 63 = rotl 1.  MODE 2: as (x << 1) + (x >> 63) — one 32-bit shift and one v_lshl_add_u64 instead of two
// v_alignbit_b32 (A/B-measured on MI355X: profiles/r02_k1_variants.log).
template <int MODE>
__device__ __forceinline__ uint64_t rotr63(uint64_t x) {
    if constexpr (MODE == 2) return (x << 1) + (x >> 63);
    else return rotr<63>(x);
}

#define IPCFP_B2B_G(a, b, c, d, x, y)          \
    a = add64<MODE>(add64<MODE>(a, b), (x));   \
    d = rotr<32>(d ^ a);                       \
    c = add64<MODE>(c, d);                     \
    b = rotr<24>(b ^ c);                       \
    a = add64<MODE>(add64<MODE>(a, b), (y));   \
    d = rotr<16>(d ^ a);                       \
    c = add64<MODE>(c, d);                     \
    b = rotr63<MODE>(b ^ c);

#define IPCFP_B2B_ROUND(s0, s1, s2, s3, s4, s5, s6, s7, s8, s9, s10, s11, s12, s13, s14, s15) \
    IPCFP_B2B_G(v0, v4, v8, v12, m[s0], m[s1])                                                \
    IPCFP_B2B_G(v1, v5, v9, v13, m[s2], m[s3])                                                \
    IPCFP_B2B_G(v2, v6, v10, v14, m[s4], m[s5])                                               \
    IPCFP_B2B_G(v3, v7, v11, v15, m[s6], m[s7])                                               \
    IPCFP_B2B_G(v0, v5, v10, v15, m[s8], m[s9])                                               \
    IPCFP_B2B_G(v1, v6, v11, v12, m[s10], m[s11])                                             \
    IPCFP_B2B_G(v2, v7, v8, v13, m[s12], m[s13])                                              \
    IPCFP_B2B_G(v3, v4, v9, v14, m[s14], m[s15])

// h ← F(h, m, t, last).  `t` is the byte counter (low word; inputs are < 2^64 B).
// `last` may differ per lane (it only flips v14).
template <int MODE>
__device__ __forceinline__ void compress(uint64_t h[8], const uint64_t m[16], uint64_t t, bool last) {
    uint64_t v0 = h[0], v1 = h[1], v2 = h[2], v3 = h[3], v4 = h[4], v5 = h[5], v6 = h[6], v7 = h[7];
    uint64_t v8 = IPCFP_B2B_IV0, v9 = IPCFP_B2B_IV1, v10 = IPCFP_B2B_IV2, v11 = IPCFP_B2B_IV3;
    uint64_t v12 = IPCFP_B2B_IV4 ^ t, v13 = IPCFP_B2B_IV5;
    uint64_t v14 = last ? ~IPCFP_B2B_IV6 : IPCFP_B2B_IV6, v15 = IPCFP_B2B_IV7;
    IPCFP_B2B_ROUND(0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15)
    IPCFP_B2B_ROUND(14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3)
    IPCFP_B2B_ROUND(11, 8, 12, 0, 5, 2, 15, 13, 10, 14, 3, 6, 7, 1, 9, 4)
    IPCFP_B2B_ROUND(7, 9, 3, 1, 13, 12, 11, 14, 2, 6, 5, 10, 4, 0, 15, 8)
    IPCFP_B2B_ROUND(9, 0, 5, 7, 2, 4, 10, 15, 14, 1, 11, 12, 6, 8, 3, 13)
    IPCFP_B2B_ROUND(2, 12, 6, 10, 0, 11, 8, 3, 4, 13, 7, 5, 15, 14, 1, 9)
    IPCFP_B2B_ROUND(12, 5, 1, 15, 14, 13, 4, 10, 0, 7, 6, 3, 9, 2, 8, 11)
    IPCFP_B2B_ROUND(13, 11, 7, 14, 12, 1, 3, 9, 5, 0, 15, 4, 8, 6, 2, 10)
    IPCFP_B2B_ROUND(6, 15, 14, 9, 11, 3, 0, 8, 12, 2, 13, 7, 1, 4, 10, 5)
    IPCFP_B2B_ROUND(10, 2, 8, 4, 7, 6, 1, 5, 15, 11, 9, 14, 3, 12, 13, 0)
    IPCFP_B2B_ROUND(0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15)
    IPCFP_B2B_ROUND(14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3)
    h[0] ^= v0 ^ v8;
    h[1] ^= v1 ^ v9;
    h[2] ^= v2 ^ v10;
    h[3] ^= v3 ^ v11;
    h[4] ^= v4 ^ v12;
    h[5] ^= v5 ^ v13;
    h[6] ^= v6 ^ v14;
    h[7] ^= v7 ^ v15;
}

// The same compression with the 16 message words staged in LDS ([word][lane], conflict-free): 32 VGPRs fewer per
// lane, i.e. one or two more wavefronts per SIMD (north-star's "message schedule staged in LDS"; A/B-measured —
// profiles/r02_k1_variants.log).  `lm` points at this lane's word 0; word w is lm[w * stride].
__device__ __forceinline__ void compress_lds(uint64_t h[8], const uint64_t* lm, uint32_t stride, uint64_t t, bool last) {
    constexpr int MODE = 0;
    uint64_t v0 = h[0], v1 = h[1], v2 = h[2], v3 = h[3], v4 = h[4], v5 = h[5], v6 = h[6], v7 = h[7];
    uint64_t v8 = IPCFP_B2B_IV0, v9 = IPCFP_B2B_IV1, v10 = IPCFP_B2B_IV2, v11 = IPCFP_B2B_IV3;
    uint64_t v12 = IPCFP_B2B_IV4 ^ t, v13 = IPCFP_B2B_IV5;
    uint64_t v14 = last ? ~IPCFP_B2B_IV6 : IPCFP_B2B_IV6, v15 = IPCFP_B2B_IV7;
#define m(i) lm[(i) * stride]
#define IPCFP_B2B_ROUND_L(s0, s1, s2, s3, s4, s5, s6, s7, s8, s9, s10, s11, s12, s13, s14, s15) \
    IPCFP_B2B_G(v0, v4, v8, v12, m(s0), m(s1))                                                  \
    IPCFP_B2B_G(v1, v5, v9, v13, m(s2), m(s3))                                                  \
    IPCFP_B2B_G(v2, v6, v10, v14, m(s4), m(s5))                                                 \
    IPCFP_B2B_G(v3, v7, v11, v15, m(s6), m(s7))                                                 \
    IPCFP_B2B_G(v0, v5, v10, v15, m(s8), m(s9))                                                 \
    IPCFP_B2B_G(v1, v6, v11, v12, m(s10), m(s11))                                               \
    IPCFP_B2B_G(v2, v7, v8, v13, m(s12), m(s13))                                                \
    IPCFP_B2B_G(v3, v4, v9, v14, m(s14), m(s15))
    IPCFP_B2B_ROUND_L(0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15)
    IPCFP_B2B_ROUND_L(14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3)
    IPCFP_B2B_ROUND_L(11, 8, 12, 0, 5, 2, 15, 13, 10, 14, 3, 6, 7, 1, 9, 4)
    IPCFP_B2B_ROUND_L(7, 9, 3, 1, 13, 12, 11, 14, 2, 6, 5, 10, 4, 0, 15, 8)
    IPCFP_B2B_ROUND_L(9, 0, 5, 7, 2, 4, 10, 15, 14, 1, 11, 12, 6, 8, 3, 13)
    IPCFP_B2B_ROUND_L(2, 12, 6, 10, 0, 11, 8, 3, 4, 13, 7, 5, 15, 14, 1, 9)
    IPCFP_B2B_ROUND_L(12, 5, 1, 15, 14, 13, 4, 10, 0, 7, 6, 3, 9, 2, 8, 11)
    IPCFP_B2B_ROUND_L(13, 11, 7, 14, 12, 1, 3, 9, 5, 0, 15, 4, 8, 6, 2, 10)
    IPCFP_B2B_ROUND_L(6, 15, 14, 9, 11, 3, 0, 8, 12, 2, 13, 7, 1, 4, 10, 5)
    IPCFP_B2B_ROUND_L(10, 2, 8, 4, 7, 6, 1, 5, 15, 11, 9, 14, 3, 12, 13, 0)
    IPCFP_B2B_ROUND_L(0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15)
    IPCFP_B2B_ROUND_L(14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3)
#undef IPCFP_B2B_ROUND_L
#undef m
    h[0] ^= v0 ^ v8;
    h[1] ^= v1 ^ v9;
    h[2] ^= v2 ^ v10;
    h[3] ^= v3 ^ v11;
    h[4] ^= v4 ^ v12;
    h[5] ^= v5 ^ v13;
    h[6] ^= v6 ^ v14;
    h[7] ^= v7 ^ v15;
}

// Load one 128-byte chunk (16 little-endian u64 words) with eight 16-byte loads.
// `p` must be 16-byte aligned.
__device__ __forceinline__ void load_chunk(uint64_t m[16], const uint8_t* p) {
    const ulonglong2* q = reinterpret_cast<const ulonglong2*>(p);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        ulonglong2 w = q[k];
        m[2 * k] = w.x;
        m[2 * k + 1] = w.y;
    }
}

// Zero every byte of the chunk at position >= rem (rem in 0..128).
__device__ __forceinline__ void mask_tail(uint64_t m[16], uint32_t rem) {
#pragma unroll
    for (int w = 0; w < 16; ++w) {
        const int k = int(rem) - 8 * w;  // valid bytes in word w
        uint64_t mask = (k >= 8) ? ~0ULL : ((k <= 0) ? 0ULL : ((1ULL << (8 * k)) - 1ULL));
        m[w] &= mask;
    }
}

}  // namespace b2b
// Blake2b-256 of a short byte buffer on one lane (the TxMeta re-hash, events/utils.rs:65)
__device__ __forceinline__ void blake2b256_small(const uint8_t* buf, uint32_t len, uint64_t out[4]) {
    uint64_t h[8];
    b2b::init256(h);
    uint32_t done = 0;
    uint64_t t = 0;
    for (;;) {
        const uint32_t left = len - done;
        const bool last = left <= 128;
        const uint32_t take = last ? left : 128;
        uint64_t m[16];
#pragma unroll
        for (int w = 0; w < 16; ++w) {
            uint64_t v = 0;
            for (int b = 0; b < 8; ++b) {
                const uint32_t idx = 8u * w + b;
                if (idx < take) v |= uint64_t(buf[done + idx]) << (8 * b);
            }
            m[w] = v;
        }
        t += take;
        b2b::compress<0>(h, m, t, last);
        done += take;
        if (last) break;
    }
    out[0] = h[0];
    out[1] = h[1];
    out[2] = h[2];
    out[3] = h[3];
}


}  // namespace ipcfp
