// oracle/capi_verify.cpp — TEST INFRASTRUCTURE: ctypes-facing entry points of the restated
// verifiers.  They take the same tables / claim structs as the product's C ABI (include/ipcfp.h)
// so a parity test hands identical inputs to both sides.
#include <omp.h>

#include <cstring>
#include <memory>
#include <unordered_map>

#include "verify.hpp"

using namespace orc;

namespace {

struct Store {
    MemoryBlockstore bs;
    // kept so the "as written" baseline can rebuild the store per proof (verifier.rs:19-28)
    const uint8_t* bytes;
    const uint64_t* off;
    const uint32_t* len;
    const uint8_t* cids40;
    uint64_t n;
    // mode 2 of orc_verify_event_proofs: execution orders kept across calls, keyed by the tipset key strings
    std::unordered_map<std::string, ExecCache> exec_caches;
};

size_t cid_slot_len(const uint8_t* slot) {
    // binary CIDs are self-delimiting: parse to find the length inside the 40-byte slot
    for (size_t l = 1; l <= IPCFP_CID_SLOT; ++l) {
        CidParts parts;
        if (cid_parse_binary(slot, l, parts)) return l;
    }
    return 0;
}

// threads == 1: the reference's loop (events/verifier.rs:79-89).  Otherwise the store is 2^k sub-maps filled by
// separate threads, each visiting its blocks in input order (duplicate CID: the last block still wins).
void load_store(MemoryBlockstore& bs, const uint8_t* bytes, const uint64_t* off, const uint32_t* len,
                const uint8_t* cids40, uint64_t n, int threads = 1) {
    if (threads == 1) {
        bs.reshard(1);
        bs.shards[0].reserve(size_t(n) * 2);
        for (uint64_t i = 0; i < n; ++i) {
            const uint8_t* slot = cids40 + IPCFP_CID_SLOT * i;
            const size_t l = cid_slot_len(slot);
            Cid c{Bytes(slot, slot + (l ? l : IPCFP_CID_SLOT))};
            bs.put_keyed(c, bytes + off[i], len[i]);
        }
        return;
    }
    const int nt = use_threads(threads);
    size_t ns = 1;
    while (ns < size_t(nt) * 4) ns <<= 1;
    bs.reshard(ns);
    std::vector<uint32_t> shard_of(n);
    std::vector<uint8_t> klen(n);
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < int64_t(n); ++i) {
        const uint8_t* slot = cids40 + IPCFP_CID_SLOT * uint64_t(i);
        const size_t l = cid_slot_len(slot);
        klen[size_t(i)] = uint8_t(l ? l : IPCFP_CID_SLOT);
        shard_of[size_t(i)] = uint32_t(MemoryBlockstore::shard_bits(Bytes(slot, slot + klen[size_t(i)])) & bs.mask);
    }
    // block ids grouped by sub-map, ascending inside a group (counting sort)
    std::vector<uint64_t> start(ns + 1, 0);
    for (uint64_t i = 0; i < n; ++i) ++start[shard_of[i] + 1];
    for (size_t s = 0; s < ns; ++s) start[s + 1] += start[s];
    std::vector<uint32_t> ids(n);
    {
        std::vector<uint64_t> cur(start.begin(), start.end() - 1);
        for (uint64_t i = 0; i < n; ++i) ids[cur[shard_of[i]]++] = uint32_t(i);
    }
#pragma omp parallel for schedule(dynamic, 1)
    for (int64_t s = 0; s < int64_t(ns); ++s) {
        MemoryBlockstore::Map& m = bs.shards[size_t(s)];
        m.reserve((start[size_t(s) + 1] - start[size_t(s)]) * 2);
        for (uint64_t q = start[size_t(s)]; q < start[size_t(s) + 1]; ++q) {
            const uint64_t i = ids[q];
            const uint8_t* slot = cids40 + IPCFP_CID_SLOT * i;
            m[Bytes(slot, slot + klen[i])] = Bytes(bytes + off[i], bytes + off[i] + len[i]);
        }
    }
}

template <typename F>
uint8_t guarded(F&& f) {
    try {
        return f();
    } catch (const Err& e) {
        return e.status;
    }
}

}  // namespace

extern "C" {

void* orc_store_create(const uint8_t* bytes, const uint64_t* off, const uint32_t* len, const uint8_t* cids40,
                       uint64_t n) {
    auto* s = new Store();
    s->bytes = bytes; s->off = off; s->len = len; s->cids40 = cids40; s->n = n;
    load_store(s->bs, bytes, off, len, cids40, n);
    return s;
}
// the same store built on `threads` OpenMP threads (0 = every processor): baseline variant B2 all-cores
void* orc_store_create_mt(const uint8_t* bytes, const uint64_t* off, const uint32_t* len, const uint8_t* cids40,
                          uint64_t n, int threads) {
    auto* s = new Store();
    s->bytes = bytes; s->off = off; s->len = len; s->cids40 = cids40; s->n = n;
    load_store(s->bs, bytes, off, len, cids40, n, threads);
    return s;
}
// The store keyed by CIDs of ANY length (the 40-byte slot tables above cannot name a CID with a 64-byte digest):
// block i's CID is cid_bytes[cid_off[i] .. cid_off[i] + cid_len[i]).  `load_witness_store` takes whatever `Cid` the
// bundle holds (src/proofs/events/verifier.rs:79-89, src/proofs/common/witness.rs:60-72).  The as-written baseline's
// per-proof rebuild (mode 0) is not available on such a store.
void* orc_store_create_var(const uint8_t* bytes, const uint64_t* off, const uint32_t* len, const uint8_t* cid_bytes,
                           const uint64_t* cid_off, const uint32_t* cid_len, uint64_t n) {
    auto* s = new Store();
    s->bytes = bytes; s->off = off; s->len = len; s->cids40 = nullptr; s->n = 0;
    s->bs.reshard(1);
    s->bs.shards[0].reserve(size_t(n) * 2);
    for (uint64_t i = 0; i < n; ++i) {
        Cid c{Bytes(cid_bytes + cid_off[i], cid_bytes + cid_off[i] + cid_len[i])};
        s->bs.put_keyed(c, bytes + off[i], len[i]);
    }
    return s;
}
void orc_store_destroy(void* s) { delete static_cast<Store*>(s); }
uint64_t orc_store_size(void* s) { return static_cast<Store*>(s)->bs.size(); }
int orc_num_procs(void) { return omp_get_num_procs(); }
int orc_use_threads(int threads) { return use_threads(threads); }  // the count `threads` resolves to

// mode 0: exactly as written — sequential, exec order rebuilt per proof (events/verifier.rs:190).
// mode 1: "fair" baseline — exec order computed once per distinct parent tipset key and looked
//         up through a hash map; OpenMP over proofs with `threads` threads (0 = all).
// mode 2: mode 1 with the execution orders kept in the store across calls (a timing harness builds them with a
//         one-claim call and then times the per-claim verifier alone).
void orc_verify_event_proofs(void* store, const ipcfp_event_proof_t* proofs, uint64_t n,
                             const ipcfp_trust_policy_t* trust, const ipcfp_event_filter_t* filter, uint8_t* status,
                             int mode, int threads) {
    Store* s = static_cast<Store*>(store);
    if (mode == 0) {
        for (uint64_t i = 0; i < n; ++i)
            status[i] = guarded([&] { return verify_event_proof_one(s->bs, proofs[i], trust, filter); });
        return;
    }
    // one ExecCache per distinct tipset key (the strings of parent_tipset_cids)
    std::unordered_map<std::string, ExecCache> local_caches;
    std::unordered_map<std::string, ExecCache>& caches = mode == 2 ? s->exec_caches : local_caches;
    std::vector<const ExecCache*> which(n, nullptr);
    for (uint64_t i = 0; i < n; ++i) {
        // proofs of one bundle usually share the very same string array: skip the key building for them
        if (i > 0 && proofs[i].parent_tipset_cids == proofs[i - 1].parent_tipset_cids &&
            proofs[i].n_parent_tipset_cids == proofs[i - 1].n_parent_tipset_cids) {
            which[i] = which[i - 1];
            continue;
        }
        std::string key;
        for (uint32_t k = 0; k < proofs[i].n_parent_tipset_cids; ++k) {
            key += proofs[i].parent_tipset_cids[k] ? proofs[i].parent_tipset_cids[k] : "";
            key.push_back('\n');
        }
        auto it = caches.find(key);
        if (it == caches.end()) {
            std::vector<Cid> parents;
            bool parsed = true;
            for (uint32_t k = 0; k < proofs[i].n_parent_tipset_cids && parsed; ++k) {
                Cid c;
                parsed = proofs[i].parent_tipset_cids[k] && cid_from_string(proofs[i].parent_tipset_cids[k], c);
                parents.push_back(c);
            }
            // an unparsable key never reaches the execution order (Err at step 1): any cache will do
            it = caches.emplace(key, parsed ? build_exec_cache(s->bs, parents, threads) : ExecCache()).first;
        }
        which[i] = &it->second;
    }
    use_threads(threads);
#pragma omp parallel for schedule(dynamic, 256)
    for (int64_t i = 0; i < int64_t(n); ++i)
        status[i] = guarded([&] { return verify_event_proof_one(s->bs, proofs[i], trust, filter, which[i]); });
}

// mode 0: as written — the witness store is rebuilt for EVERY proof (verifier.rs:19-28 → storage/verifier.rs:30).
// mode 1: store built once, OpenMP over proofs.
void orc_verify_storage_proofs(void* store, const ipcfp_storage_proof_t* proofs, uint64_t n,
                               const ipcfp_trust_policy_t* trust, uint8_t* status, int mode, int threads) {
    Store* s = static_cast<Store*>(store);
    if (mode == 0) {
        for (uint64_t i = 0; i < n; ++i) {
            MemoryBlockstore fresh;
            load_store(fresh, s->bytes, s->off, s->len, s->cids40, s->n);
            status[i] = guarded([&] { return verify_storage_proof_one(fresh, proofs[i], trust); });
        }
        return;
    }
    use_threads(threads);
#pragma omp parallel for schedule(dynamic, 256)
    for (int64_t i = 0; i < int64_t(n); ++i)
        status[i] = guarded([&] { return verify_storage_proof_one(s->bs, proofs[i], trust); });
}

// ---- packed claims → the reference's string structs (full-size parity runs) ---------------------------
// Building millions of ctypes string structs in Python takes minutes; these two rebuild the strings in C++ from
// the packed claims of include/ipcfp.h (binary CIDs → Cid::to_string, bytes → "0x" + hex) and hand them to the
// very same verify_*_proof_one.  Only claims whose every flag is set can be expressed (the flags record facts
// about strings that no longer exist); anything else gets status 255.
static std::string hex0x_of(const uint8_t* p, size_t n) {
    static const char* d = "0123456789abcdef";
    std::string s = "0x";
    for (size_t i = 0; i < n; ++i) { s.push_back(d[p[i] >> 4]); s.push_back(d[p[i] & 15]); }
    return s;
}
static std::string slot_to_string(const uint8_t* slot40) {
    return cid_to_string(Cid{Bytes(slot40, slot40 + cid_slot_len(slot40))});
}

void orc_verify_storage_claims_packed(void* store, const ipcfp_storage_claim_t* claims, uint64_t n,
                                      const ipcfp_trust_policy_t* trust, uint8_t* status, int threads) {
    Store* s = static_cast<Store*>(store);
    use_threads(threads);
#pragma omp parallel for schedule(dynamic, 256)
    for (int64_t i = 0; i < int64_t(n); ++i) {
        const ipcfp_storage_claim_t& c = claims[i];
        if (c.flags != 63u) {
            status[i] = 255;
            continue;
        }
        const std::string child = slot_to_string(c.child), sr = slot_to_string(c.state_root),
                          as = slot_to_string(c.actor_state), st = slot_to_string(c.storage_root),
                          slot = hex0x_of(c.slot, 32), value = hex0x_of(c.value, 32);
        ipcfp_storage_proof_t p{};
        p.child_epoch = c.child_epoch;
        p.child_block_cid = child.c_str();
        p.parent_state_root = sr.c_str();
        p.actor_id = c.actor_id;
        p.actor_state_cid = as.c_str();
        p.storage_root = st.c_str();
        p.slot = slot.c_str();
        p.value = value.c_str();
        status[i] = guarded([&] { return verify_storage_proof_one(s->bs, p, trust); });
    }
}

void orc_verify_event_claims_packed(void* store, const ipcfp_tipset_ref_t* tipsets, uint32_t n_tipsets,
                                    const ipcfp_event_claim_t* claims, uint64_t n, const uint8_t* blob,
                                    const ipcfp_trust_policy_t* trust, const ipcfp_event_filter_t* filter,
                                    uint8_t* status, int threads) {
    Store* s = static_cast<Store*>(store);
    struct Ts {
        std::vector<std::string> parents;
        std::vector<const char*> parent_ptrs;
        std::string child;
        ExecCache cache;
        bool ok;
    };
    std::vector<Ts> ts(n_tipsets);
    for (uint32_t k = 0; k < n_tipsets; ++k) {
        ts[k].ok = tipsets[k].flags == 3u && (tipsets[k].n_parents <= IPCFP_MAX_PARENTS || tipsets[k].more_parents != nullptr);
        if (!ts[k].ok) continue;
        std::vector<Cid> parents;
        for (uint32_t j = 0; j < tipsets[k].n_parents; ++j) {
            // (a tipset key wider than the inline form keeps its tail behind more_parents: include/ipcfp.h)
            const uint8_t* slot = j < IPCFP_MAX_PARENTS ? tipsets[k].parents[j]
                                                        : tipsets[k].more_parents + size_t(j - IPCFP_MAX_PARENTS) * IPCFP_CID_SLOT;
            parents.push_back(Cid{Bytes(slot, slot + cid_slot_len(slot))});
            ts[k].parents.push_back(cid_to_string(parents.back()));
        }
        for (auto& str : ts[k].parents) ts[k].parent_ptrs.push_back(str.c_str());
        ts[k].child = slot_to_string(tipsets[k].child);
        ts[k].cache = build_exec_cache(s->bs, parents, threads);
    }
    use_threads(threads);
#pragma omp parallel for schedule(dynamic, 256)
    for (int64_t i = 0; i < int64_t(n); ++i) {
        const ipcfp_event_claim_t& c = claims[i];
        if (c.tipset >= n_tipsets || !ts[c.tipset].ok || c.flags != 3u || c.n_topics > 8) {
            status[i] = 255;
            continue;
        }
        const Ts& t = ts[c.tipset];
        std::string topic_s[8];
        const char* topic_p[8];
        bool expressible = true;
        for (uint32_t k = 0; k < c.n_topics; ++k) {
            const uint8_t* e = blob + c.topics_off + 33u * k;
            expressible = expressible && e[0] == 1;
            topic_s[k] = hex0x_of(e + 1, 32);
            topic_p[k] = topic_s[k].c_str();
        }
        if (!expressible) {
            status[i] = 255;
            continue;
        }
        const std::string msg = slot_to_string(c.message_cid), data = hex0x_of(blob + c.data_off, c.data_len);
        ipcfp_event_proof_t p{};
        p.parent_epoch = c.parent_epoch;
        p.child_epoch = c.child_epoch;
        p.parent_tipset_cids = t.parent_ptrs.data();
        p.n_parent_tipset_cids = uint32_t(t.parent_ptrs.size());
        p.child_block_cid = t.child.c_str();
        p.message_cid = msg.c_str();
        p.exec_index = c.exec_index;
        p.event_index = c.event_index;
        p.emitter = c.emitter;
        p.topics = topic_p;
        p.n_topics = c.n_topics;
        p.data = data.c_str();
        status[i] = guarded([&] { return verify_event_proof_one(s->bs, p, trust, filter, &t.cache); });
    }
}

// ---- primitives (parity targets of the device primitives) ------------------
static ValueChecker checker_for(int value_kind) {
    switch (value_kind) {
        case 0: return check_cid_value;
        case 1: return check_receipt;
        case 2: return check_stamped_event;
        case 3: return check_actor_state;
        case 4: return check_vec_u8;
        default: return [](Reader& r) { r.skip(); };
    }
}

// Amt::load(root).get(index).  status: TRUE / NOT_FOUND / ERR_*; value bytes copied to out (≤ cap).
void orc_amt_get(void* store, const uint8_t* root_cid40, int version, int value_kind, const uint64_t* index,
                 uint64_t n, uint8_t* status, uint8_t* out, uint32_t cap, uint32_t* out_len) {
    Store* s = static_cast<Store*>(store);
    Cid root{Bytes(root_cid40, root_cid40 + cid_slot_len(root_cid40))};
    ValueChecker chk = checker_for(value_kind);
    for (uint64_t i = 0; i < n; ++i) {
        out_len[i] = 0;
        status[i] = guarded([&]() -> uint8_t {
            AmtRoot r = amt_load(s->bs, root, version, chk);
            ValueLoc loc;
            if (!amt_get(s->bs, r, index[i], chk, loc)) return IPCFP_ST_NOT_FOUND;
            out_len[i] = uint32_t(loc.len);
            std::memcpy(out + size_t(cap) * i, loc.block->data() + loc.off, loc.len < cap ? loc.len : cap);
            return IPCFP_ST_TRUE;
        });
    }
}

// `threads`: 0 = every processor (capped by IPCFP_ORACLE_MAX_THREADS), else exactly that many (bench.py's sweep)
void orc_hamt_get_t(void* store, const uint8_t* root_cid40, uint32_t bit_width, int value_kind, const uint8_t* keys,
                    const uint32_t* key_off, const uint32_t* key_len, uint64_t n, uint8_t* status, uint8_t* out,
                    uint32_t cap, uint32_t* out_len, int threads) {
    Store* s = static_cast<Store*>(store);
    Cid root{Bytes(root_cid40, root_cid40 + cid_slot_len(root_cid40))};
    ValueChecker chk = checker_for(value_kind);
    use_threads(threads);
#pragma omp parallel for schedule(dynamic, 64)
    for (int64_t i = 0; i < int64_t(n); ++i) {
        out_len[i] = 0;
        status[i] = guarded([&]() -> uint8_t {
            ValueLoc loc;
            if (!hamt_get(s->bs, root, bit_width, keys + key_off[i], key_len[i], chk, loc)) return IPCFP_ST_NOT_FOUND;
            out_len[i] = uint32_t(loc.len);
            std::memcpy(out + size_t(cap) * i, loc.block->data() + loc.off, loc.len < cap ? loc.len : cap);
            return IPCFP_ST_TRUE;
        });
    }
}

void orc_hamt_get(void* store, const uint8_t* root_cid40, uint32_t bit_width, int value_kind, const uint8_t* keys,
                  const uint32_t* key_off, const uint32_t* key_len, uint64_t n, uint8_t* status, uint8_t* out,
                  uint32_t cap, uint32_t* out_len) {
    orc_hamt_get_t(store, root_cid40, bit_width, value_kind, keys, key_off, key_len, n, status, out, cap, out_len, 0);
}

// reconstruct_execution_order: returns the status; on success *count and up to cap CIDs (40-byte slots).
uint8_t orc_exec_order(void* store, const uint8_t* parent_cids40, uint32_t n_parents, uint8_t* out_cids40,
                       uint64_t cap, uint64_t* count) {
    Store* s = static_cast<Store*>(store);
    *count = 0;
    return guarded([&]() -> uint8_t {
        std::vector<Cid> parents;
        for (uint32_t i = 0; i < n_parents; ++i) {
            const uint8_t* slot = parent_cids40 + IPCFP_CID_SLOT * i;
            parents.push_back(Cid{Bytes(slot, slot + cid_slot_len(slot))});
        }
        std::vector<Cid> exec = reconstruct_execution_order(s->bs, parents);
        *count = exec.size();
        for (size_t i = 0; i < exec.size() && i < cap; ++i) {
            std::memset(out_cids40 + IPCFP_CID_SLOT * i, 0, IPCFP_CID_SLOT);
            std::memcpy(out_cids40 + IPCFP_CID_SLOT * i, exec[i].b.data(), exec[i].b.size());
        }
        return IPCFP_ST_TRUE;
    });
}

// find_matching_events.  receipt_has_match: one byte per receipt index (cap_receipts); matches as
// (exec_index, event_index, emitter) triples (cap_matches); touched CIDs in Cid order (cap_touched).
uint8_t orc_scan_events_mt(void* store, const uint8_t* receipts_root40, const ipcfp_event_filter_t* filter, int has_actor,
                           uint64_t actor, uint8_t* receipt_has_match, uint64_t cap_receipts, uint64_t* n_receipts,
                           uint64_t* match_triples, uint64_t cap_matches, uint64_t* n_matches, uint8_t* touched40,
                           uint64_t cap_touched, uint64_t* n_touched, int threads);
uint8_t orc_scan_events(void* store, const uint8_t* receipts_root40, const ipcfp_event_filter_t* filter, int has_actor,
                        uint64_t actor, uint8_t* receipt_has_match, uint64_t cap_receipts, uint64_t* n_receipts,
                        uint64_t* match_triples, uint64_t cap_matches, uint64_t* n_matches, uint8_t* touched40,
                        uint64_t cap_touched, uint64_t* n_touched) {
    return orc_scan_events_mt(store, receipts_root40, filter, has_actor, actor, receipt_has_match, cap_receipts,
                              n_receipts, match_triples, cap_matches, n_matches, touched40, cap_touched, n_touched, 1);
}
// threads: 1 = the sequential loops as written; 0 = every processor (baseline variant B2 all-cores)
uint8_t orc_scan_events_mt(void* store, const uint8_t* receipts_root40, const ipcfp_event_filter_t* filter, int has_actor,
                           uint64_t actor, uint8_t* receipt_has_match, uint64_t cap_receipts, uint64_t* n_receipts,
                           uint64_t* match_triples, uint64_t cap_matches, uint64_t* n_matches, uint8_t* touched40,
                           uint64_t cap_touched, uint64_t* n_touched, int threads) {
    Store* s = static_cast<Store*>(store);
    *n_receipts = *n_matches = *n_touched = 0;
    return guarded([&]() -> uint8_t {
        Cid root{Bytes(receipts_root40, receipts_root40 + cid_slot_len(receipts_root40))};
        std::vector<uint8_t> has;
        std::vector<ScanMatch> ms;
        std::vector<Cid> touched;
        scan_events(s->bs, root, *filter, has_actor != 0, actor, has, ms, touched40 ? &touched : nullptr, threads);
        *n_receipts = has.size();
        for (size_t i = 0; i < has.size() && i < cap_receipts; ++i) receipt_has_match[i] = has[i];
        *n_matches = ms.size();
        for (size_t i = 0; i < ms.size() && i < cap_matches; ++i) {
            match_triples[3 * i] = ms[i].exec_index;
            match_triples[3 * i + 1] = ms[i].event_index;
            match_triples[3 * i + 2] = ms[i].emitter;
        }
        *n_touched = touched.size();
        for (size_t i = 0; i < touched.size() && i < cap_touched; ++i) {
            std::memset(touched40 + IPCFP_CID_SLOT * i, 0, IPCFP_CID_SLOT);
            std::memcpy(touched40 + IPCFP_CID_SLOT * i, touched[i].b.data(), touched[i].b.size());
        }
        return IPCFP_ST_TRUE;
    });
}

// generate_event_proof: proofs as (exec, event, emitter) triples + message CIDs + topic/data blobs are
// re-derivable from the triples, so only the triples, the message CIDs and the witness order are returned.
uint8_t orc_generate_event_proof(void* store, const uint8_t* parent_cids40, uint32_t n_parents, const uint8_t* child40,
                                 const ipcfp_event_filter_t* filter, int has_actor, uint64_t actor,
                                 uint64_t* triples, uint8_t* msg_cids40, uint64_t cap_proofs, uint64_t* n_proofs,
                                 uint8_t* witness40, uint64_t cap_witness, uint64_t* n_witness) {
    Store* s = static_cast<Store*>(store);
    *n_proofs = *n_witness = 0;
    return guarded([&]() -> uint8_t {
        std::vector<Cid> parents;
        for (uint32_t i = 0; i < n_parents; ++i) {
            const uint8_t* slot = parent_cids40 + IPCFP_CID_SLOT * i;
            parents.push_back(Cid{Bytes(slot, slot + cid_slot_len(slot))});
        }
        Cid child{Bytes(child40, child40 + cid_slot_len(child40))};
        GeneratedEventBundle b = generate_event_proof(s->bs, parents, child, *filter, has_actor != 0, actor);
        *n_proofs = b.proofs.size();
        for (size_t i = 0; i < b.proofs.size() && i < cap_proofs; ++i) {
            triples[3 * i] = b.proofs[i].exec_index;
            triples[3 * i + 1] = b.proofs[i].event_index;
            triples[3 * i + 2] = b.proofs[i].emitter;
            std::memset(msg_cids40 + IPCFP_CID_SLOT * i, 0, IPCFP_CID_SLOT);
            std::memcpy(msg_cids40 + IPCFP_CID_SLOT * i, b.proofs[i].message_cid.b.data(), b.proofs[i].message_cid.b.size());
        }
        *n_witness = b.witness.size();
        for (size_t i = 0; i < b.witness.size() && i < cap_witness; ++i) {
            std::memset(witness40 + IPCFP_CID_SLOT * i, 0, IPCFP_CID_SLOT);
            std::memcpy(witness40 + IPCFP_CID_SLOT * i, b.witness[i].b.data(), b.witness[i].b.size());
        }
        return IPCFP_ST_TRUE;
    });
}

// generate_storage_proof: out3 = parent_state_root, actor_state, storage_root (3 × 40); value32; witness.
uint8_t orc_generate_storage_proof(void* store, const uint8_t* child40, uint64_t actor_id, const uint8_t* slot32,
                                   uint8_t* out3x40, uint8_t* value32, uint8_t* witness40, uint64_t cap_witness,
                                   uint64_t* n_witness) {
    Store* s = static_cast<Store*>(store);
    *n_witness = 0;
    return guarded([&]() -> uint8_t {
        Cid child{Bytes(child40, child40 + cid_slot_len(child40))};
        GeneratedStorageProof p = generate_storage_proof(s->bs, child, actor_id, slot32);
        std::memset(out3x40, 0, 3 * IPCFP_CID_SLOT);
        std::memcpy(out3x40, p.parent_state_root.b.data(), p.parent_state_root.b.size());
        std::memcpy(out3x40 + IPCFP_CID_SLOT, p.actor_state_cid.b.data(), p.actor_state_cid.b.size());
        std::memcpy(out3x40 + 2 * IPCFP_CID_SLOT, p.storage_root.b.data(), p.storage_root.b.size());
        std::memcpy(value32, p.value, 32);
        *n_witness = p.witness.size();
        for (size_t i = 0; i < p.witness.size() && i < cap_witness; ++i) {
            std::memset(witness40 + IPCFP_CID_SLOT * i, 0, IPCFP_CID_SLOT);
            std::memcpy(witness40 + IPCFP_CID_SLOT * i, p.witness[i].b.data(), p.witness[i].b.size());
        }
        return IPCFP_ST_TRUE;
    });
}

// string helpers for tests / fixtures
int orc_cid_to_string(const uint8_t* cid, uint32_t len, char* out, uint32_t cap) {
    std::string s = cid_to_string(Cid{Bytes(cid, cid + len)});
    if (s.size() + 1 > cap) return -1;
    std::memcpy(out, s.c_str(), s.size() + 1);
    return int(s.size());
}
int orc_cid_from_string(const char* s, uint8_t* out40) {
    Cid c;
    if (!cid_from_string(s, c) || c.b.size() > IPCFP_CID_SLOT) return -1;
    std::memset(out40, 0, IPCFP_CID_SLOT);
    std::memcpy(out40, c.b.data(), c.b.size());
    return int(c.b.size());
}
void orc_cid_for_block(const uint8_t* data, uint64_t len, uint8_t* out40) {
    Cid c = cid_for_block(data, len);
    std::memset(out40, 0, IPCFP_CID_SLOT);
    std::memcpy(out40, c.b.data(), c.b.size());
}

}  // extern "C"
