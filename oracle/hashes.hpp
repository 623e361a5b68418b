// oracle/hashes.hpp — TEST INFRASTRUCTURE, NOT PRODUCT CODE.
//
// Scalar CPU restatement of the three hash functions the reference's hot path
// reaches through un-vendored crates (the crates are NOT under /root/reference;
// Cargo.lock is git-ignored, so versions are semver ranges only):
//
//   * Blake2b-256  — multihash-codetable 0.1.4 `Code::Blake2b256`
//                    call site: /root/reference/src/proofs/events/utils.rs:65
//                    (`bs.put_cbor(&(bls_root, secp_root), Code::Blake2b256)`)
//   * Keccak-256   — sha3 0.10 `Keccak256`
//                    call sites: /root/reference/src/proofs/common/evm.rs:62-69,81-88
//   * SHA-256      — fvm_ipld_hamt 0.10.4 default `Sha256` key hasher
//                    call sites: /root/reference/src/proofs/common/decode.rs:29-39,
//                                /root/reference/src/proofs/storage/decode.rs:79-96
//
// PARITY STATUS: the reference holds no tests or golden vectors for this path
// ("parity unpinned" by the reference).  These functions are pinned instead by
// RFC 7693 / FIPS 180-4 / Keccak-team known answers and by Python's hashlib
// (tests/test_oracle_hashes.py), and by five well-known Filecoin CIDs
// (tests/test_oracle_kat.py).
//
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use
// anything in oracle/.
#pragma once
#include <cstddef>
#include <cstdint>

namespace orc {

// BLAKE2b with digest_length = 32 in the parameter block (RFC 7693), no key.
void blake2b256(const uint8_t* data, size_t len, uint8_t out[32]);
// Original Keccak (pad 0x01 .. 0x80), rate 136, 32-byte output.
void keccak256(const uint8_t* data, size_t len, uint8_t out[32]);
// FIPS 180-4 SHA-256.
void sha256(const uint8_t* data, size_t len, uint8_t out[32]);

}  // namespace orc
