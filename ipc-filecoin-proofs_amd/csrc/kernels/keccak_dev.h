// csrc/kernels/keccak_dev.h — Keccak-f[1600] / Keccak-256 for gfx950, one hash per lane.
//
// Function: original Keccak-256 (pad 0x01 … 0x80, rate 136 B) as produced by
// sha3 0.10 `Keccak256` — reference call sites src/proofs/common/evm.rs:62-69
// (hash_event_signature), :81-88 (keccak256), src/proofs/storage/utils.rs:5-12
// (compute_mapping_slot).
//
// The 25-lane state lives in 50 VGPRs per work-item; all 24 rounds are unrolled
// so rho/pi are register renames and the rotates become v_alignbit_b32 pairs.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace ipcfp {
namespace keccak {

__device__ __forceinline__ uint64_t rotl(uint64_t x, int n) { return (x << n) | (x >> (64 - n)); }

__device__ __forceinline__ void f1600(uint64_t s[25]) {
    constexpr uint64_t RC[24] = {
        0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808aULL, 0x8000000080008000ULL,
        0x000000000000808bULL, 0x0000000080000001ULL, 0x8000000080008081ULL, 0x8000000000008009ULL,
        0x000000000000008aULL, 0x0000000000000088ULL, 0x0000000080008009ULL, 0x000000008000000aULL,
        0x000000008000808bULL, 0x800000000000008bULL, 0x8000000000008089ULL, 0x8000000000008003ULL,
        0x8000000000008002ULL, 0x8000000000000080ULL, 0x000000000000800aULL, 0x800000008000000aULL,
        0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};
#pragma unroll
    for (int r = 0; r < 24; ++r) {
        // theta
        uint64_t c0 = s[0] ^ s[5] ^ s[10] ^ s[15] ^ s[20];
        uint64_t c1 = s[1] ^ s[6] ^ s[11] ^ s[16] ^ s[21];
        uint64_t c2 = s[2] ^ s[7] ^ s[12] ^ s[17] ^ s[22];
        uint64_t c3 = s[3] ^ s[8] ^ s[13] ^ s[18] ^ s[23];
        uint64_t c4 = s[4] ^ s[9] ^ s[14] ^ s[19] ^ s[24];
        uint64_t d0 = c4 ^ rotl(c1, 1), d1 = c0 ^ rotl(c2, 1), d2 = c1 ^ rotl(c3, 1);
        uint64_t d3 = c2 ^ rotl(c4, 1), d4 = c3 ^ rotl(c0, 1);
#pragma unroll
        for (int y = 0; y < 25; y += 5) {
            s[y] ^= d0;
            s[y + 1] ^= d1;
            s[y + 2] ^= d2;
            s[y + 3] ^= d3;
            s[y + 4] ^= d4;
        }
        // rho + pi  (B[y][2x+3y] = rot(A[x][y], r[x][y])), written out
        uint64_t b[25];
        b[0] = s[0];
        b[10] = rotl(s[1], 1);
        b[20] = rotl(s[2], 62);
        b[5] = rotl(s[3], 28);
        b[15] = rotl(s[4], 27);
        b[16] = rotl(s[5], 36);
        b[1] = rotl(s[6], 44);
        b[11] = rotl(s[7], 6);
        b[21] = rotl(s[8], 55);
        b[6] = rotl(s[9], 20);
        b[7] = rotl(s[10], 3);
        b[17] = rotl(s[11], 10);
        b[2] = rotl(s[12], 43);
        b[12] = rotl(s[13], 25);
        b[22] = rotl(s[14], 39);
        b[23] = rotl(s[15], 41);
        b[8] = rotl(s[16], 45);
        b[18] = rotl(s[17], 15);
        b[3] = rotl(s[18], 21);
        b[13] = rotl(s[19], 8);
        b[14] = rotl(s[20], 18);
        b[24] = rotl(s[21], 2);
        b[9] = rotl(s[22], 61);
        b[19] = rotl(s[23], 56);
        b[4] = rotl(s[24], 14);
        // chi
#pragma unroll
        for (int y = 0; y < 25; y += 5) {
            s[y] = b[y] ^ (~b[y + 1] & b[y + 2]);
            s[y + 1] = b[y + 1] ^ (~b[y + 2] & b[y + 3]);
            s[y + 2] = b[y + 2] ^ (~b[y + 3] & b[y + 4]);
            s[y + 3] = b[y + 3] ^ (~b[y + 4] & b[y]);
            s[y + 4] = b[y + 4] ^ (~b[y] & b[y + 1]);
        }
        // iota
        s[0] ^= RC[r];
    }
}

// Keccak-256 of `len` bytes read one byte at a time (general, any alignment).
// The short inputs on the hot path (event signatures ≤ ~64 B, 64-byte mapping
// keys) make this a single permutation; the 64-byte case has its own kernel.
__device__ __forceinline__ void hash_bytes(const uint8_t* __restrict__ p, uint32_t len, uint64_t out[4]) {
    uint64_t s[25];
#pragma unroll
    for (int i = 0; i < 25; ++i) s[i] = 0;
    uint32_t pos = 0;
    for (;;) {
        const uint32_t take = (len - pos) < 136u ? (len - pos) : 136u;
        // absorb `take` bytes into words 0..16
#pragma unroll
        for (int w = 0; w < 17; ++w) {
            uint64_t v = 0;
            const uint32_t b0 = 8u * w;
            if (b0 < take) {
                const uint32_t nb = (take - b0) < 8u ? (take - b0) : 8u;
                for (uint32_t k = 0; k < nb; ++k) v |= uint64_t(p[pos + b0 + k]) << (8 * k);
            }
            s[w] ^= v;
        }
        pos += take;
        if (take < 136u) {
            // pad10*1 with the 0x01 domain byte, inside this block
            const uint32_t w = take >> 3, sh = (take & 7u) * 8u;
#pragma unroll
            for (int k = 0; k < 17; ++k)
                if ((uint32_t)k == w) s[k] ^= uint64_t(0x01) << sh;
            s[16] ^= 0x8000000000000000ULL;
            f1600(s);
            break;
        }
        f1600(s);
    }
    out[0] = s[0];
    out[1] = s[1];
    out[2] = s[2];
    out[3] = s[3];
}

}  // namespace keccak
}  // namespace ipcfp
