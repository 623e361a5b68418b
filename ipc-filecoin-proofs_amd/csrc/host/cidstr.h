// csrc/host/cidstr.h — CID / hex string forms on the host side of the engine.
// Counterpart of `Cid::try_from(&str)` / `Cid::to_string()` (cid 0.11) as used at
// src/proofs/common/witness.rs:60-72 and src/proofs/storage/verifier.rs:110,126,144.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace ipcfp {

// Parse a CID string (multibase b/B base32, f/F base16, z base58btc, or a bare CIDv0 "Qm…").
// On success `out` holds the binary CID (any length).  Returns false when the reference's
// `Cid::try_from` would return Err.
bool cid_from_string(const char* s, std::vector<uint8_t>& out);
// Cid::to_string(): CIDv1 → "b" + base32-lower; CIDv0 → base58btc.
std::string cid_to_string(const uint8_t* cid, size_t len);
// true iff p[0..n) is exactly one well-formed binary CID
bool cid_binary_ok(const uint8_t* p, size_t n);
// A binary CID as its 40-byte ABI slot: zero padded, or — longer than the slot — folded to ff | len | blake2b-256(cid)
// (include/ipcfp.h "CIDs"; the device folds the long links it reads the same way: kernels/cbor_dev.h long_cid_fold)
void cid_to_slot(const uint8_t* cid, size_t len, uint8_t slot40[40]);
void blake2b256_host(const uint8_t* data, size_t len, uint8_t out32[32]);
// hex digits → bytes; false on odd length / non-hex
bool hex_decode(const char* s, size_t n, std::vector<uint8_t>& out);

}  // namespace ipcfp
