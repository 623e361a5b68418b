// oracle/store.hpp — TEST INFRASTRUCTURE.  Blockstores.
//
// Restates:
//   * `fvm_ipld_blockstore::MemoryBlockstore` as the verifiers use it
//     (src/proofs/events/verifier.rs:79-89, src/proofs/storage/verifier.rs:68-78):
//     a `HashMap<Cid, Vec<u8>>`; `put_keyed` does NOT hash the data;
//     inserting an existing CID replaces the block (last wins).
//   * `RecordingBlockStore` (src/proofs/common/blockstore.rs:8-39): `get` records the
//     CID BEFORE delegating (so a miss is recorded too); `take_seen` copies the
//     BTreeSet in `Cid: Ord` order and does not drain it.
#pragma once
#include <map>
#include <set>
#include <unordered_map>

#include "cid.hpp"

namespace orc {

struct BytesHash {
    size_t operator()(const Bytes& b) const {
        uint64_t h = 1469598103934665603ULL;
        for (uint8_t c : b) { h ^= c; h *= 1099511628211ULL; }
        return size_t(h);
    }
};

struct Blockstore {
    virtual ~Blockstore() = default;
    // Ok(Some(bytes)) → pointer; Ok(None) → nullptr.
    virtual const Bytes* get(const Cid& c) const = 0;
};

struct MemoryBlockstore : Blockstore {
    std::unordered_map<Bytes, Bytes, BytesHash> map;
    void put_keyed(const Cid& c, const uint8_t* data, size_t len) { map[c.b] = Bytes(data, data + len); }
    const Bytes* get(const Cid& c) const override {
        auto it = map.find(c.b);
        return it == map.end() ? nullptr : &it->second;
    }
};

struct RecordingBlockStore : Blockstore {
    const Blockstore& inner;
    mutable std::set<Cid> seen;
    explicit RecordingBlockStore(const Blockstore& in) : inner(in) {}
    const Bytes* get(const Cid& c) const override {
        seen.insert(c);
        return inner.get(c);
    }
    std::vector<Cid> take_seen() const { return std::vector<Cid>(seen.begin(), seen.end()); }
};

// `bs.get(cid)?.ok_or_else(|| anyhow!("missing …"))?`
inline const Bytes& must_get(const Blockstore& bs, const Cid& c, const char* what) {
    const Bytes* b = bs.get(c);
    if (!b) throw Err(IPCFP_ST_ERR_MISSING_BLOCK, std::string("missing ") + what);
    return *b;
}

}  // namespace orc
