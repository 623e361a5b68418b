// oracle/cid.hpp — TEST INFRASTRUCTURE.  CID binary/string forms.
//
// Restates `cid` 0.11 + `multihash` behaviour the reference relies on (crates not under
// /root/reference, Cargo.toml:11,15):
//   Cid::try_from(&str)   src/proofs/common/witness.rs:60-72,
//                         src/proofs/events/verifier.rs:109,131,151,193
//   Cid::to_string()      src/proofs/storage/verifier.rs:110,126,144
//   Cid: Ord / Eq / Hash  src/proofs/common/blockstore.rs:10 (BTreeSet<Cid>),
//                         src/proofs/events/utils.rs:54 (HashSet<Cid>)
// Binary form: CIDv1 = varint(1) varint(codec) multihash;
// multihash = varint(code) varint(size) digest[size], size ≤ 64.  CIDv0 = 12 20 ‖ 32 B.
// String form: CIDv1 → multibase; CIDv0 → bare base58btc ("Qm…", 46 chars).
// ⚠ Deviation kept deliberately strict: bytes after the multihash are an error here
// (cid 0.11's `TryFrom<&[u8]>` ignores them); the HIP path applies the same rule.
#pragma once
#include <algorithm>
#include <cstdint>
#include <string>
#include <vector>

#include "cbor.hpp"

namespace orc {

struct Cid {
    Bytes b;  // canonical binary form
    bool operator==(const Cid& o) const { return b == o.b; }
    bool operator!=(const Cid& o) const { return b != o.b; }
    // `Cid: Ord` is derived field-wise (version, codec, hash{code,size,digest}).  For CIDs of one
    // type this is the lexicographic order of the digest; across types compare the parsed fields.
    bool operator<(const Cid& o) const;
};

inline bool Cid::operator<(const Cid& o) const {
    CidParts x, y;
    if (!cid_parse_binary(b.data(), b.size(), x) || !cid_parse_binary(o.b.data(), o.b.size(), y)) return b < o.b;
    if (x.version != y.version) return x.version < y.version;
    if (x.codec != y.codec) return x.codec < y.codec;
    if (x.mh_code != y.mh_code) return x.mh_code < y.mh_code;
    if (x.mh_size != y.mh_size) return x.mh_size < y.mh_size;
    return std::lexicographical_compare(x.digest, x.digest + x.mh_size, y.digest, y.digest + y.mh_size);
}

inline Cid cid_from_link(const uint8_t* p, size_t n) {
    CidParts parts;
    if (!cid_parse_binary(p, n, parts)) decode_err("malformed CID in link");
    return Cid{Bytes(p, p + n)};
}

inline Cid read_cid(Reader& r) {
    const uint8_t* p; size_t n;
    r.read_link(p, n);
    return cid_from_link(p, n);
}
// the same through a try-reader: a malformed CID marks the reader instead of throwing
inline Cid read_cid(TryReader& r) {
    const uint8_t* p; size_t n;
    r.read_link(p, n);
    CidParts parts;
    if (!r.bad && !cid_parse_binary(p, n, parts)) r.fail("malformed CID in link");
    return r.bad ? Cid{} : Cid{Bytes(p, p + n)};
}

// Filecoin chain CID for a DAG-CBOR block: CIDv1, 0x71, blake2b-256.
Cid cid_for_block(const uint8_t* data, size_t len);

// ---- strings ----
std::string base32_lower(const uint8_t* p, size_t n);
bool base32_decode(const std::string& s, size_t from, Bytes& out, bool upper = false);  // one case per call (multibase b / B), no padding
std::string base58btc(const uint8_t* p, size_t n);
bool base58btc_decode(const std::string& s, size_t from, Bytes& out);

std::string cid_to_string(const Cid& c);            // Cid::to_string
bool cid_from_string(const std::string& s, Cid& out);  // Cid::try_from(&str); false ⇒ Err

}  // namespace orc
