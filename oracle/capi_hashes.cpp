// oracle/capi_hashes.cpp — TEST INFRASTRUCTURE: ctypes-facing entry points for the
// scalar hash restatements (see hashes.hpp).  Batch forms take the same
// (bytes, off[], len[]) table the product's C ABI takes, so a parity test feeds
// identical buffers to both sides.
#include <omp.h>

#include <cstdint>
#include <cstring>

#include "amt.hpp"
#include "hashes.hpp"

extern "C" {

void orc_blake2b256(const uint8_t* data, uint64_t len, uint8_t* out32) { orc::blake2b256(data, len, out32); }
void orc_keccak256(const uint8_t* data, uint64_t len, uint8_t* out32) { orc::keccak256(data, len, out32); }
void orc_sha256(const uint8_t* data, uint64_t len, uint8_t* out32) { orc::sha256(data, len, out32); }

// kind: 0 = blake2b-256, 1 = keccak-256, 2 = sha-256
void orc_hash_batch(int kind, const uint8_t* bytes, const uint64_t* off, const uint32_t* len, uint64_t n,
                    uint8_t* out32) {
    for (uint64_t i = 0; i < n; ++i) {
        const uint8_t* p = bytes + off[i];
        uint8_t* o = out32 + 32 * i;
        if (kind == 0) orc::blake2b256(p, len[i], o);
        else if (kind == 1) orc::keccak256(p, len[i], o);
        else orc::sha256(p, len[i], o);
    }
}

// CID check the way Block::cid / put_cbor define it: the block's
// bytes must Blake2b-256-hash to the digest carried by its claimed CID.
// ok[i] = 1 iff digest matches expect32[i].  Returns the number of matches.
uint64_t orc_blake2b256_verify(const uint8_t* bytes, const uint64_t* off, const uint32_t* len,
                               const uint8_t* expect32, uint64_t n, uint8_t* ok) {
    uint64_t good = 0;
    for (uint64_t i = 0; i < n; ++i) {
        uint8_t d[32];
        orc::blake2b256(bytes + off[i], len[i], d);
        ok[i] = std::memcmp(d, expect32 + 32 * i, 32) == 0;
        good += ok[i];
    }
    return good;
}

// the same check on `threads` OpenMP threads (0 = every processor): baseline variant B2 all-cores
uint64_t orc_blake2b256_verify_mt(const uint8_t* bytes, const uint64_t* off, const uint32_t* len,
                                  const uint8_t* expect32, uint64_t n, uint8_t* ok, int threads) {
    orc::use_threads(threads);
    uint64_t good = 0;
#pragma omp parallel for schedule(dynamic, 1024) reduction(+ : good)
    for (int64_t i = 0; i < int64_t(n); ++i) {
        uint8_t d[32];
        orc::blake2b256(bytes + off[i], len[i], d);
        ok[i] = std::memcmp(d, expect32 + 32 * i, 32) == 0;
        good += ok[i];
    }
    return good;
}

}  // extern "C"
