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
    // One `HashMap<Cid, Vec<u8>>`, kept as 2^k independent sub-maps selected by hash bits: the reference's map is
    // the k = 0 case; the all-cores baseline (BASELINE.md variant B2) builds the sub-maps on separate threads.
    using Map = std::unordered_map<Bytes, Bytes, BytesHash>;
    std::vector<Map> shards;
    size_t mask = 0;
    explicit MemoryBlockstore(size_t n_shards = 1) { reshard(n_shards); }
    void reshard(size_t n_shards) {  // n_shards: a power of two; drops the content
        shards.assign(n_shards, Map());
        mask = n_shards - 1;
    }
    static size_t shard_bits(const Bytes& key) { return BytesHash()(key) >> 40; }
    Map& shard_of(const Bytes& key) { return shards[shard_bits(key) & mask]; }
    void put_keyed(const Cid& c, const uint8_t* data, size_t len) { shard_of(c.b)[c.b] = Bytes(data, data + len); }
    const Bytes* get(const Cid& c) const override {
        const Map& m = shards[shard_bits(c.b) & mask];
        auto it = m.find(c.b);
        return it == m.end() ? nullptr : &it->second;
    }
    size_t size() const {
        size_t n = 0;
        for (const Map& m : shards) n += m.size();
        return n;
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
