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
};

size_t cid_slot_len(const uint8_t* slot) {
    // binary CIDs are self-delimiting: parse to find the length inside the 40-byte slot
    for (size_t l = 1; l <= IPCFP_CID_SLOT; ++l) {
        CidParts parts;
        if (cid_parse_binary(slot, l, parts)) return l;
    }
    return 0;
}

void load_store(MemoryBlockstore& bs, const uint8_t* bytes, const uint64_t* off, const uint32_t* len,
                const uint8_t* cids40, uint64_t n) {
    bs.map.reserve(size_t(n) * 2);
    for (uint64_t i = 0; i < n; ++i) {
        const uint8_t* slot = cids40 + IPCFP_CID_SLOT * i;
        const size_t l = cid_slot_len(slot);
        Cid c{Bytes(slot, slot + (l ? l : IPCFP_CID_SLOT))};
        bs.put_keyed(c, bytes + off[i], len[i]);
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
void orc_store_destroy(void* s) { delete static_cast<Store*>(s); }

// mode 0: exactly as written — sequential, exec order rebuilt per proof (events/verifier.rs:190).
// mode 1: "fair" baseline — exec order computed once per distinct parent tipset key and looked
//         up through a hash map; OpenMP over proofs with `threads` threads (0 = all).
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
    std::unordered_map<std::string, ExecCache> caches;
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
            it = caches.emplace(key, parsed ? build_exec_cache(s->bs, parents) : ExecCache()).first;
        }
        which[i] = &it->second;
    }
    if (threads > 0) omp_set_num_threads(threads);
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
    if (threads > 0) omp_set_num_threads(threads);
#pragma omp parallel for schedule(dynamic, 256)
    for (int64_t i = 0; i < int64_t(n); ++i)
        status[i] = guarded([&] { return verify_storage_proof_one(s->bs, proofs[i], trust); });
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

void orc_hamt_get(void* store, const uint8_t* root_cid40, uint32_t bit_width, int value_kind, const uint8_t* keys,
                  const uint32_t* key_off, const uint32_t* key_len, uint64_t n, uint8_t* status, uint8_t* out,
                  uint32_t cap, uint32_t* out_len) {
    Store* s = static_cast<Store*>(store);
    Cid root{Bytes(root_cid40, root_cid40 + cid_slot_len(root_cid40))};
    ValueChecker chk = checker_for(value_kind);
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
uint8_t orc_scan_events(void* store, const uint8_t* receipts_root40, const ipcfp_event_filter_t* filter, int has_actor,
                        uint64_t actor, uint8_t* receipt_has_match, uint64_t cap_receipts, uint64_t* n_receipts,
                        uint64_t* match_triples, uint64_t cap_matches, uint64_t* n_matches, uint8_t* touched40,
                        uint64_t cap_touched, uint64_t* n_touched) {
    Store* s = static_cast<Store*>(store);
    *n_receipts = *n_matches = *n_touched = 0;
    return guarded([&]() -> uint8_t {
        Cid root{Bytes(receipts_root40, receipts_root40 + cid_slot_len(receipts_root40))};
        std::vector<uint8_t> has;
        std::vector<ScanMatch> ms;
        std::vector<Cid> touched;
        scan_events(s->bs, root, *filter, has_actor != 0, actor, has, ms, touched40 ? &touched : nullptr);
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
