// oracle/hamt.hpp — TEST INFRASTRUCTURE.  HAMT v3 reader.
//
// Restates fvm_ipld_hamt 0.10.4 `Hamt<BS, V, BytesKey, Sha256>` (crate NOT under
// /root/reference, Cargo.toml:20) as the reference calls it:
//   Hamt::<_, ActorState>::load_with_bit_width(.., 5).get(&BytesKey(addr))  src/proofs/common/decode.rs:29-39
//   Hamt::<_, Vec<u8>>::load_with_bit_width(..).get(&BytesKey(slot))        src/proofs/storage/decode.rs:79-96
// Wire format: node = [bitfield: bytes, pointers: [P…]]; the root block IS a node;
// bitfield = 256-bit big-endian integer with leading zero bytes stripped; bit idx ⇔ value bit 2^idx;
// pointer = tag-42 link | array of [key: bytes, value] pairs (bucket).
// get: h = SHA-256(key); take bit_width bits MSB-first per level; bit clear ⇒ None;
// child #popcount(bitfield & (2^idx − 1)); Link ⇒ load (missing ⇒ Err) and descend;
// bucket ⇒ linear search for the exact key; hash bits exhausted ⇒ Err(MaxDepth).
// serde decodes a node completely, so every pointer and every bucket entry of a visited
// node is type-checked (`check_value`), not only the one followed.
#pragma once
#include "amt.hpp"

namespace orc {

// Hamt::load_with_bit_width(root).get(key).  true ⇒ Some(value at loc), false ⇒ None.
bool hamt_get(const Blockstore& bs, const Cid& root, uint32_t bit_width, const uint8_t* key, size_t key_len,
              const ValueChecker& check_value, ValueLoc& loc);

}  // namespace orc
