// tools/synth/hash_min.hpp — compact CPU hashes for the synthetic-tipset GENERATOR.
//
// The generator is the writer side of the test inputs (it must produce real CIDs, HAMT
// placements and Keccak topics).  It deliberately does NOT link oracle/ (which only the
// checkers may use) and is written independently of it — loop/table style here, unrolled
// macro style there — so the two cross-check each other through the parity tests.
#pragma once
#include <cstdint>
#include <cstring>

namespace synth {

inline uint64_t ror64(uint64_t v, int s) { return (v >> s) | (v << (64 - s)); }
inline uint64_t rol64(uint64_t v, int s) { return s ? ((v << s) | (v >> (64 - s))) : v; }

// ---- BLAKE2b-256 (RFC 7693) ----
struct Blake2b256 {
    uint64_t h[8];
    uint64_t t = 0;
    uint8_t buf[128];
    size_t fill = 0;
    static const uint64_t* iv() {
        static const uint64_t v[8] = {0x6a09e667f3bcc908ULL, 0xbb67ae8584caa73bULL, 0x3c6ef372fe94f82bULL,
                                      0xa54ff53a5f1d36f1ULL, 0x510e527fade682d1ULL, 0x9b05688c2b3e6c1fULL,
                                      0x1f83d9abfb41bd6bULL, 0x5be0cd19137e2179ULL};
        return v;
    }
    Blake2b256() {
        for (int i = 0; i < 8; ++i) h[i] = iv()[i];
        h[0] ^= 0x01010020ULL;
    }
    void block(const uint8_t* b, bool fin) {
        static const uint8_t S[10][16] = {
            {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15}, {14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3},
            {11, 8, 12, 0, 5, 2, 15, 13, 10, 14, 3, 6, 7, 1, 9, 4}, {7, 9, 3, 1, 13, 12, 11, 14, 2, 6, 5, 10, 4, 0, 15, 8},
            {9, 0, 5, 7, 2, 4, 10, 15, 14, 1, 11, 12, 6, 8, 3, 13}, {2, 12, 6, 10, 0, 11, 8, 3, 4, 13, 7, 5, 15, 14, 1, 9},
            {12, 5, 1, 15, 14, 13, 4, 10, 0, 7, 6, 3, 9, 2, 8, 11}, {13, 11, 7, 14, 12, 1, 3, 9, 5, 0, 15, 4, 8, 6, 2, 10},
            {6, 15, 14, 9, 11, 3, 0, 8, 12, 2, 13, 7, 1, 4, 10, 5}, {10, 2, 8, 4, 7, 6, 1, 5, 15, 11, 9, 14, 3, 12, 13, 0}};
        static const uint8_t Q[8][4] = {{0, 4, 8, 12}, {1, 5, 9, 13}, {2, 6, 10, 14}, {3, 7, 11, 15},
                                        {0, 5, 10, 15}, {1, 6, 11, 12}, {2, 7, 8, 13}, {3, 4, 9, 14}};
        uint64_t m[16], v[16];
        std::memcpy(m, b, 128);
        for (int i = 0; i < 8; ++i) { v[i] = h[i]; v[8 + i] = iv()[i]; }
        v[12] ^= t;
        if (fin) v[14] = ~v[14];
        for (int r = 0; r < 12; ++r) {
            const uint8_t* s = S[r % 10];
            for (int q = 0; q < 8; ++q) {
                uint64_t &a = v[Q[q][0]], &bb = v[Q[q][1]], &c = v[Q[q][2]], &d = v[Q[q][3]];
                a += bb + m[s[2 * q]];     d = ror64(d ^ a, 32);
                c += d;                    bb = ror64(bb ^ c, 24);
                a += bb + m[s[2 * q + 1]]; d = ror64(d ^ a, 16);
                c += d;                    bb = ror64(bb ^ c, 63);
            }
        }
        for (int i = 0; i < 8; ++i) h[i] ^= v[i] ^ v[8 + i];
    }
    void update(const uint8_t* p, size_t n) {
        while (n) {
            if (fill == 128) {  // only flush when more input follows: the last block is flagged final
                t += 128;
                block(buf, false);
                fill = 0;
            }
            size_t take = 128 - fill < n ? 128 - fill : n;
            std::memcpy(buf + fill, p, take);
            fill += take; p += take; n -= take;
        }
    }
    void final(uint8_t out[32]) {
        t += fill;
        std::memset(buf + fill, 0, 128 - fill);
        block(buf, true);
        std::memcpy(out, h, 32);
    }
};
inline void blake2b256(const uint8_t* p, size_t n, uint8_t out[32]) {
    Blake2b256 s;
    s.update(p, n);
    s.final(out);
}

// ---- SHA-256 ----
inline void sha256(const uint8_t* p, size_t n, uint8_t out[32]) {
    static const uint32_t K[64] = {
        0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01,
        0x243185be, 0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc,
        0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da, 0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147,
        0x06ca6351, 0x14292967, 0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85,
        0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070, 0x19a4c116, 0x1e376c08,
        0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208,
        0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};
    uint32_t H[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};
    uint8_t tail[128];
    const size_t full = n / 64 * 64, rem = n - full;
    std::memset(tail, 0, sizeof tail);
    std::memcpy(tail, p + full, rem);
    tail[rem] = 0x80;
    const size_t tl = rem < 56 ? 64 : 128;
    const uint64_t bits = uint64_t(n) * 8;
    for (int i = 0; i < 8; ++i) tail[tl - 1 - i] = uint8_t(bits >> (8 * i));
    auto R = [](uint32_t x, int s) { return (x >> s) | (x << (32 - s)); };
    auto comp = [&](const uint8_t* b) {
        uint32_t w[64];
        for (int i = 0; i < 16; ++i) w[i] = uint32_t(b[4 * i]) << 24 | uint32_t(b[4 * i + 1]) << 16 | uint32_t(b[4 * i + 2]) << 8 | b[4 * i + 3];
        for (int i = 16; i < 64; ++i)
            w[i] = w[i - 16] + (R(w[i - 15], 7) ^ R(w[i - 15], 18) ^ (w[i - 15] >> 3)) + w[i - 7] +
                   (R(w[i - 2], 17) ^ R(w[i - 2], 19) ^ (w[i - 2] >> 10));
        uint32_t s[8];
        std::memcpy(s, H, 32);
        for (int i = 0; i < 64; ++i) {
            uint32_t t1 = s[7] + (R(s[4], 6) ^ R(s[4], 11) ^ R(s[4], 25)) + ((s[4] & s[5]) ^ (~s[4] & s[6])) + K[i] + w[i];
            uint32_t t2 = (R(s[0], 2) ^ R(s[0], 13) ^ R(s[0], 22)) + ((s[0] & s[1]) ^ (s[0] & s[2]) ^ (s[1] & s[2]));
            for (int k = 7; k > 0; --k) s[k] = s[k - 1];
            s[4] += t1;
            s[0] = t1 + t2;
        }
        for (int i = 0; i < 8; ++i) H[i] += s[i];
    };
    for (size_t o = 0; o < full; o += 64) comp(p + o);
    comp(tail);
    if (tl == 128) comp(tail + 64);
    for (int i = 0; i < 8; ++i) { out[4 * i] = H[i] >> 24; out[4 * i + 1] = H[i] >> 16; out[4 * i + 2] = H[i] >> 8; out[4 * i + 3] = H[i]; }
}

// ---- Keccak-256 (pad 0x01) ----
inline void keccak256(const uint8_t* p, size_t n, uint8_t out[32]) {
    uint64_t A[5][5] = {};  // A[x][y]
    auto perm = [&]() {
        uint64_t rc = 1;  // LFSR-generated round constants
        auto lfsr = [&](uint8_t& st) { bool r = st & 1; st = uint8_t((st & 0x80) ? ((st << 1) ^ 0x71) : (st << 1)); return r; };
        uint8_t st = 1;
        (void)rc;
        for (int round = 0; round < 24; ++round) {
            uint64_t C[5], D[5], B[5][5];
            for (int x = 0; x < 5; ++x) C[x] = A[x][0] ^ A[x][1] ^ A[x][2] ^ A[x][3] ^ A[x][4];
            for (int x = 0; x < 5; ++x) D[x] = C[(x + 4) % 5] ^ rol64(C[(x + 1) % 5], 1);
            for (int x = 0; x < 5; ++x) for (int y = 0; y < 5; ++y) A[x][y] ^= D[x];
            int x = 1, y = 0;
            B[0][0] = A[0][0];
            for (int t = 0; t < 24; ++t) {
                B[y][(2 * x + 3 * y) % 5] = rol64(A[x][y], ((t + 1) * (t + 2) / 2) % 64);
                int nx = y, ny = (2 * x + 3 * y) % 5;
                x = nx; y = ny;
            }
            for (int xx = 0; xx < 5; ++xx) for (int yy = 0; yy < 5; ++yy) A[xx][yy] = B[xx][yy] ^ (~B[(xx + 1) % 5][yy] & B[(xx + 2) % 5][yy]);
            uint64_t c = 0;
            for (int j = 0; j < 7; ++j) if (lfsr(st)) c ^= 1ULL << ((1 << j) - 1);
            A[0][0] ^= c;
        }
    };
    const size_t rate = 136;
    uint8_t blk[136];
    size_t o = 0;
    for (;;) {
        size_t take = n - o < rate ? n - o : rate;
        std::memset(blk, 0, rate);
        std::memcpy(blk, p + o, take);
        const bool last = take < rate;
        if (last) { blk[take] ^= 0x01; blk[rate - 1] ^= 0x80; }
        for (size_t i = 0; i < rate / 8; ++i) { uint64_t w; std::memcpy(&w, blk + 8 * i, 8); A[i % 5][i / 5] ^= w; }
        perm();
        o += take;
        if (last) break;
    }
    for (int i = 0; i < 4; ++i) std::memcpy(out + 8 * i, &A[i % 5][i / 5], 8);
}

}  // namespace synth
