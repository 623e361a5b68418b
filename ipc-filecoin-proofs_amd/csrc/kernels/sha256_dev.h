// csrc/kernels/sha256_dev.h — SHA-256 for gfx950, one hash per lane.
//
// fvm_ipld_hamt 0.10 hashes every HAMT key with SHA-256 and consumes the digest
// `bit_width` bits at a time, MSB first (reference call sites:
// src/proofs/common/decode.rs:29-39, src/proofs/storage/decode.rs:79-96).  Keys on
// this path are short (ID addresses 2-11 B, storage slots 32 B): one compression.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace ipcfp {
namespace sha256 {

__device__ __forceinline__ uint32_t rotr(uint32_t x, int n) { return __builtin_amdgcn_alignbit(x, x, n); }

__device__ __forceinline__ void compress(uint32_t h[8], uint32_t w[16]) {
    constexpr uint32_t K[64] = {
        0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5,
        0xd807aa98, 0x12835b01, 0x243185be, 0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174,
        0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da,
        0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967,
        0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85,
        0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070,
        0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3,
        0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208, 0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};
    uint32_t a = h[0], b = h[1], c = h[2], d = h[3], e = h[4], f = h[5], g = h[6], hh = h[7];
#pragma unroll
    for (int i = 0; i < 64; ++i) {
        if (i >= 16) {
            const uint32_t w15 = w[(i - 15) & 15], w2 = w[(i - 2) & 15];
            const uint32_t s0 = rotr(w15, 7) ^ rotr(w15, 18) ^ (w15 >> 3);
            const uint32_t s1 = rotr(w2, 17) ^ rotr(w2, 19) ^ (w2 >> 10);
            w[i & 15] = w[i & 15] + s0 + w[(i - 7) & 15] + s1;
        }
        const uint32_t S1 = rotr(e, 6) ^ rotr(e, 11) ^ rotr(e, 25);
        const uint32_t ch = (e & f) ^ (~e & g);
        const uint32_t t1 = hh + S1 + ch + K[i] + w[i & 15];
        const uint32_t S0 = rotr(a, 2) ^ rotr(a, 13) ^ rotr(a, 22);
        const uint32_t mj = (a & b) ^ (a & c) ^ (b & c);
        const uint32_t t2 = S0 + mj;
        hh = g;
        g = f;
        f = e;
        e = d + t1;
        d = c;
        c = b;
        b = a;
        a = t1 + t2;
    }
    h[0] += a;
    h[1] += b;
    h[2] += c;
    h[3] += d;
    h[4] += e;
    h[5] += f;
    h[6] += g;
    h[7] += hh;
}

__device__ __forceinline__ void init(uint32_t h[8]) {
    h[0] = 0x6a09e667;
    h[1] = 0xbb67ae85;
    h[2] = 0x3c6ef372;
    h[3] = 0xa54ff53a;
    h[4] = 0x510e527f;
    h[5] = 0x9b05688c;
    h[6] = 0x1f83d9ab;
    h[7] = 0x5be0cd19;
}

// SHA-256 of `len` bytes (any length), byte loads.  h[] receives the digest as
// eight big-endian words: digest byte 4k+j = (h[k] >> (24 - 8j)) & 0xff, so HAMT
// hash bit b (MSB first) is bit (31 - b%32) of h[b/32].
__device__ __forceinline__ void hash_bytes(const uint8_t* __restrict__ p, uint32_t len, uint32_t h[8]) {
    init(h);
    const uint64_t bitlen = uint64_t(len) * 8ull;
    uint32_t pos = 0;
    bool done = false, pad_started = false;
    while (!done) {
        uint32_t w[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            uint32_t v = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint32_t idx = pos + 4u * k + j;
                uint32_t byte = 0;
                if (idx < len) byte = p[idx];
                else if (idx == len) byte = 0x80;
                v = (v << 8) | byte;
            }
            w[k] = v;
        }
        (void)pad_started;
        // length goes in the last 8 bytes of the block in which byte `len` (0x80) and 8 more bytes fit
        const uint32_t block_end = pos + 64u;
        if (len + 9u <= block_end) {
            w[14] = uint32_t(bitlen >> 32);
            w[15] = uint32_t(bitlen);
            done = true;
        }
        compress(h, w);
        pos += 64u;
    }
}

// SHA-256 of exactly 32 bytes held as four little-endian 64-bit words (byte i = word i/8, bits 8·(i%8)…): one compression
// whose second half is the padding — no byte loads, no length logic.
__device__ __forceinline__ void hash32_words(const uint64_t kw[4], uint32_t h[8]) {
    init(h);
    uint32_t w[16];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        w[2 * j] = __builtin_bswap32(uint32_t(kw[j]));
        w[2 * j + 1] = __builtin_bswap32(uint32_t(kw[j] >> 32));
    }
    w[8] = 0x80000000u;
#pragma unroll
    for (int k = 9; k < 15; ++k) w[k] = 0;
    w[15] = 256u;
    compress(h, w);
}

// bits [bit_pos, bit_pos + width) of the digest, MSB first (HashBits::next).  width ≤ 8.
__device__ __forceinline__ uint32_t take_bits(const uint32_t h[8], uint32_t bit_pos, uint32_t width) {
    // gather a 64-bit window starting at word bit_pos/32 without dynamic register indexing
    const uint32_t wi = bit_pos >> 5;
    uint32_t hi = 0, lo = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        if ((uint32_t)k == wi) hi = h[k];
        if ((uint32_t)k == wi + 1) lo = h[k];
    }
    const uint64_t win = (uint64_t(hi) << 32) | lo;
    const uint32_t sh = 64u - (bit_pos & 31u) - width;
    return uint32_t(win >> sh) & ((1u << width) - 1u);
}

}  // namespace sha256
}  // namespace ipcfp
