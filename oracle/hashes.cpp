// oracle/hashes.cpp — TEST INFRASTRUCTURE (see hashes.hpp header for provenance).
#include "hashes.hpp"

#include <cstring>

namespace orc {

// ----------------------------------------------------------------------------
// BLAKE2b-256 (RFC 7693 §3).  multihash code 0xb220 = blake2b with a 32-byte
// digest *parameter* (h[0] ^= 0x01010020), not a truncation of BLAKE2b-512.
// ----------------------------------------------------------------------------
static const uint64_t B2B_IV[8] = {
    0x6a09e667f3bcc908ULL, 0xbb67ae8584caa73bULL, 0x3c6ef372fe94f82bULL, 0xa54ff53a5f1d36f1ULL,
    0x510e527fade682d1ULL, 0x9b05688c2b3e6c1fULL, 0x1f83d9abfb41bd6bULL, 0x5be0cd19137e2179ULL};

static const uint8_t B2B_SIGMA[12][16] = {
    {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15},
    {14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3},
    {11, 8, 12, 0, 5, 2, 15, 13, 10, 14, 3, 6, 7, 1, 9, 4},
    {7, 9, 3, 1, 13, 12, 11, 14, 2, 6, 5, 10, 4, 0, 15, 8},
    {9, 0, 5, 7, 2, 4, 10, 15, 14, 1, 11, 12, 6, 8, 3, 13},
    {2, 12, 6, 10, 0, 11, 8, 3, 4, 13, 7, 5, 15, 14, 1, 9},
    {12, 5, 1, 15, 14, 13, 4, 10, 0, 7, 6, 3, 9, 2, 8, 11},
    {13, 11, 7, 14, 12, 1, 3, 9, 5, 0, 15, 4, 8, 6, 2, 10},
    {6, 15, 14, 9, 11, 3, 0, 8, 12, 2, 13, 7, 1, 4, 10, 5},
    {10, 2, 8, 4, 7, 6, 1, 5, 15, 11, 9, 14, 3, 12, 13, 0},
    {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15},
    {14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3}};

static inline uint64_t rotr64(uint64_t x, unsigned n) { return (x >> n) | (x << (64 - n)); }
static inline uint64_t load64le(const uint8_t* p) {
    uint64_t v;
    std::memcpy(&v, p, 8);  // host is little-endian (x86-64)
    return v;
}

static void b2b_compress(uint64_t h[8], const uint8_t block[128], uint64_t t, bool last) {
    uint64_t m[16], v[16];
    for (int i = 0; i < 16; ++i) m[i] = load64le(block + 8 * i);
    for (int i = 0; i < 8; ++i) {
        v[i] = h[i];
        v[i + 8] = B2B_IV[i];
    }
    v[12] ^= t;  // low 64 bits of the byte counter; the high word stays 0 (< 2^64 bytes)
    if (last) v[14] = ~v[14];
#define ORC_G(a, b, c, d, x, y)        \
    v[a] = v[a] + v[b] + (x);          \
    v[d] = rotr64(v[d] ^ v[a], 32);    \
    v[c] = v[c] + v[d];                \
    v[b] = rotr64(v[b] ^ v[c], 24);    \
    v[a] = v[a] + v[b] + (y);          \
    v[d] = rotr64(v[d] ^ v[a], 16);    \
    v[c] = v[c] + v[d];                \
    v[b] = rotr64(v[b] ^ v[c], 63);
    for (int r = 0; r < 12; ++r) {
        const uint8_t* s = B2B_SIGMA[r];
        ORC_G(0, 4, 8, 12, m[s[0]], m[s[1]]);
        ORC_G(1, 5, 9, 13, m[s[2]], m[s[3]]);
        ORC_G(2, 6, 10, 14, m[s[4]], m[s[5]]);
        ORC_G(3, 7, 11, 15, m[s[6]], m[s[7]]);
        ORC_G(0, 5, 10, 15, m[s[8]], m[s[9]]);
        ORC_G(1, 6, 11, 12, m[s[10]], m[s[11]]);
        ORC_G(2, 7, 8, 13, m[s[12]], m[s[13]]);
        ORC_G(3, 4, 9, 14, m[s[14]], m[s[15]]);
    }
#undef ORC_G
    for (int i = 0; i < 8; ++i) h[i] ^= v[i] ^ v[i + 8];
}

void blake2b256(const uint8_t* data, size_t len, uint8_t out[32]) {
    uint64_t h[8];
    for (int i = 0; i < 8; ++i) h[i] = B2B_IV[i];
    h[0] ^= 0x01010000ULL ^ 32ULL;  // fanout=1, depth=1, key_len=0, digest_len=32
    uint64_t t = 0;
    // All but the last block: a message that is an exact multiple of 128 bytes
    // ends on a FULL final block (no extra empty block); the empty message is one
    // all-zero final block with t = 0.
    while (len > 128) {
        t += 128;
        b2b_compress(h, data, t, false);
        data += 128;
        len -= 128;
    }
    uint8_t last[128];
    std::memset(last, 0, sizeof last);
    if (len) std::memcpy(last, data, len);
    t += len;
    b2b_compress(h, last, t, true);
    std::memcpy(out, h, 32);  // little-endian words
}

// ----------------------------------------------------------------------------
// Keccak-256 (Keccak-f[1600], rate 1088 bits, original 0x01 domain byte).
// ----------------------------------------------------------------------------
static const uint64_t KECCAK_RC[24] = {
    0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808aULL, 0x8000000080008000ULL,
    0x000000000000808bULL, 0x0000000080000001ULL, 0x8000000080008081ULL, 0x8000000000008009ULL,
    0x000000000000008aULL, 0x0000000000000088ULL, 0x0000000080008009ULL, 0x000000008000000aULL,
    0x000000008000808bULL, 0x800000000000008bULL, 0x8000000000008089ULL, 0x8000000000008003ULL,
    0x8000000000008002ULL, 0x8000000000000080ULL, 0x000000000000800aULL, 0x800000008000000aULL,
    0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};
static const int KECCAK_ROT[24] = {1,  3,  6,  10, 15, 21, 28, 36, 45, 55, 2,  14,
                                   27, 41, 56, 8,  25, 43, 62, 18, 39, 61, 20, 44};
static const int KECCAK_PIL[24] = {10, 7,  11, 17, 18, 3, 5,  16, 8,  21, 24, 4,
                                   15, 23, 19, 13, 12, 2, 20, 14, 22, 9,  6,  1};

static inline uint64_t rotl64(uint64_t x, unsigned n) { return (x << n) | (x >> (64 - n)); }

static void keccak_f1600(uint64_t st[25]) {
    for (int round = 0; round < 24; ++round) {
        uint64_t bc[5];
        for (int i = 0; i < 5; ++i) bc[i] = st[i] ^ st[i + 5] ^ st[i + 10] ^ st[i + 15] ^ st[i + 20];
        for (int i = 0; i < 5; ++i) {
            uint64_t t = bc[(i + 4) % 5] ^ rotl64(bc[(i + 1) % 5], 1);
            for (int j = 0; j < 25; j += 5) st[j + i] ^= t;
        }
        uint64_t t = st[1];
        for (int i = 0; i < 24; ++i) {
            int j = KECCAK_PIL[i];
            uint64_t b = st[j];
            st[j] = rotl64(t, KECCAK_ROT[i]);
            t = b;
        }
        for (int j = 0; j < 25; j += 5) {
            for (int i = 0; i < 5; ++i) bc[i] = st[j + i];
            for (int i = 0; i < 5; ++i) st[j + i] ^= (~bc[(i + 1) % 5]) & bc[(i + 2) % 5];
        }
        st[0] ^= KECCAK_RC[round];
    }
}

void keccak256(const uint8_t* data, size_t len, uint8_t out[32]) {
    const size_t rate = 136;
    uint64_t st[25];
    std::memset(st, 0, sizeof st);
    while (len >= rate) {
        for (size_t i = 0; i < rate / 8; ++i) st[i] ^= load64le(data + 8 * i);
        keccak_f1600(st);
        data += rate;
        len -= rate;
    }
    uint8_t last[136];
    std::memset(last, 0, sizeof last);
    if (len) std::memcpy(last, data, len);
    last[len] ^= 0x01;
    last[rate - 1] ^= 0x80;
    for (size_t i = 0; i < rate / 8; ++i) st[i] ^= load64le(last + 8 * i);
    keccak_f1600(st);
    std::memcpy(out, st, 32);
}

// ----------------------------------------------------------------------------
// SHA-256 (FIPS 180-4).
// ----------------------------------------------------------------------------
static const uint32_t SHA256_K[64] = {
    0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5,
    0xd807aa98, 0x12835b01, 0x243185be, 0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174,
    0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da,
    0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967,
    0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85,
    0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070,
    0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3,
    0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208, 0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};

static inline uint32_t rotr32(uint32_t x, unsigned n) { return (x >> n) | (x << (32 - n)); }

static void sha256_compress(uint32_t h[8], const uint8_t blk[64]) {
    uint32_t w[64];
    for (int i = 0; i < 16; ++i)
        w[i] = (uint32_t(blk[4 * i]) << 24) | (uint32_t(blk[4 * i + 1]) << 16) |
               (uint32_t(blk[4 * i + 2]) << 8) | uint32_t(blk[4 * i + 3]);
    for (int i = 16; i < 64; ++i) {
        uint32_t s0 = rotr32(w[i - 15], 7) ^ rotr32(w[i - 15], 18) ^ (w[i - 15] >> 3);
        uint32_t s1 = rotr32(w[i - 2], 17) ^ rotr32(w[i - 2], 19) ^ (w[i - 2] >> 10);
        w[i] = w[i - 16] + s0 + w[i - 7] + s1;
    }
    uint32_t a = h[0], b = h[1], c = h[2], d = h[3], e = h[4], f = h[5], g = h[6], hh = h[7];
    for (int i = 0; i < 64; ++i) {
        uint32_t S1 = rotr32(e, 6) ^ rotr32(e, 11) ^ rotr32(e, 25);
        uint32_t ch = (e & f) ^ (~e & g);
        uint32_t t1 = hh + S1 + ch + SHA256_K[i] + w[i];
        uint32_t S0 = rotr32(a, 2) ^ rotr32(a, 13) ^ rotr32(a, 22);
        uint32_t mj = (a & b) ^ (a & c) ^ (b & c);
        uint32_t t2 = S0 + mj;
        hh = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
    }
    h[0] += a; h[1] += b; h[2] += c; h[3] += d; h[4] += e; h[5] += f; h[6] += g; h[7] += hh;
}

void sha256(const uint8_t* data, size_t len, uint8_t out[32]) {
    uint32_t h[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a,
                     0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};
    const uint64_t bitlen = uint64_t(len) * 8;
    while (len >= 64) {
        sha256_compress(h, data);
        data += 64;
        len -= 64;
    }
    uint8_t last[128];
    std::memset(last, 0, sizeof last);
    if (len) std::memcpy(last, data, len);
    last[len] = 0x80;
    const size_t total = (len < 56) ? 64 : 128;
    for (int i = 0; i < 8; ++i) last[total - 1 - i] = uint8_t(bitlen >> (8 * i));
    sha256_compress(h, last);
    if (total == 128) sha256_compress(h, last + 64);
    for (int i = 0; i < 8; ++i) {
        out[4 * i] = uint8_t(h[i] >> 24);
        out[4 * i + 1] = uint8_t(h[i] >> 16);
        out[4 * i + 2] = uint8_t(h[i] >> 8);
        out[4 * i + 3] = uint8_t(h[i]);
    }
}

}  // namespace orc
