// csrc/host/pack_claims.cpp — lowering of the reference's EventProof structs (strings) to the packed ABI form.
//
// `verify_single_proof` parses the claim strings of every proof itself (`Cid::try_from`,
// src/proofs/events/verifier.rs:130-131,193; hex compares :276-287).  The engine parses each string ONCE on
// the host and carries the two facts the reference observes about it — does it parse, and what are its bytes —
// in `ipcfp_event_claim_t` / `ipcfp_tipset_ref_t` + a byte blob (include/ipcfp.h).  Pure host code (no device,
// no context): proofs are grouped by tipset key in one sequential pass (consecutive proofs of a bundle share
// the very same string arrays, which makes it a pointer compare), then lowered in parallel by contiguous
// ranges, each range into its own blob; the blobs are concatenated and the offsets rebased.
#include "pack_claims.h"
#include "parallel.h"

#include <algorithm>
#include <cstring>
#include <thread>
#include <unordered_map>

#include "cidstr.h"

namespace ipcfp {

void parse_cid_claim(const char* s, CidKey& key, bool& parsed, bool& canonical);

namespace {

bool parse_hex0x(const char* s, std::vector<uint8_t>& out) {
    // the reference formats "0x" + lowercase hex and compares ignoring ASCII case, so a claimed
    // string matches iff it is '0' 'x'|'X' followed by hex digits of the same bytes
    if (!s || s[0] != '0' || !(s[1] == 'x' || s[1] == 'X')) return false;
    return hex_decode(s + 2, std::strlen(s + 2), out);
}

// everything of one claim except its tipset index
void lower_one(const ipcfp_event_proof_t& p, EventClaimPacked& c, std::vector<uint8_t>& blob) {
    c.parent_epoch = p.parent_epoch;
    c.child_epoch = p.child_epoch;
    c.exec_index = p.exec_index;
    c.event_index = p.event_index;
    c.emitter = p.emitter;
    c.flags = 0;
    bool parsed, canon;
    parse_cid_claim(p.message_cid, c.message, parsed, canon);
    if (parsed) c.flags |= EC_MSG_PARSED;
    // topics: n × [matchable, 32 bytes]
    c.n_topics = p.n_topics;
    c.topics_off = uint32_t(blob.size());
    std::vector<uint8_t> t;
    for (uint32_t k = 0; k < p.n_topics; ++k) {
        const bool ok = parse_hex0x(p.topics ? p.topics[k] : nullptr, t) && t.size() == 32;
        blob.push_back(ok ? 1 : 0);
        const size_t at = blob.size();
        blob.resize(at + 32, 0);
        if (ok) std::memcpy(blob.data() + at, t.data(), 32);
    }
    c.data_off = 0;
    c.data_len = 0;
    if (parse_hex0x(p.data, t)) {
        c.flags |= EC_DATA_MATCHABLE;
        c.data_off = uint32_t(blob.size());
        c.data_len = uint32_t(t.size());
        blob.insert(blob.end(), t.begin(), t.end());
    }
}

}  // namespace

int pack_event_claims_host(const ipcfp_event_proof_t* proofs, uint64_t n, PackedEvents& out, std::string& err) {
    out.tipsets.clear();
    out.more_parents.clear();
    out.claims.assign(n, EventClaimPacked{});
    out.blob.clear();
    // ---- pass 1 (sequential): tipset contexts ----
    std::unordered_map<std::string, uint32_t> ctx_index;
    const char* const* last_parents = nullptr;
    const char* last_child = nullptr;
    uint32_t last_np = 0, last_ctx = 0;
    bool have_last = false;
    for (uint64_t i = 0; i < n; ++i) {
        const ipcfp_event_proof_t& p = proofs[i];
        uint32_t ci;
        if (have_last && p.parent_tipset_cids == last_parents && p.child_block_cid == last_child &&
            p.n_parent_tipset_cids == last_np) {
            ci = last_ctx;  // same string arrays as the previous proof
        } else {
            // length-prefixed, so that no two distinct (parents…, child) tuples share a key: count, then
            // (length, bytes) per string — a null pointer is length 0xffffffff
            std::string key;
            auto put_u32 = [&key](uint32_t v) { key.append(reinterpret_cast<const char*>(&v), 4); };
            auto put_str = [&](const char* str) {
                if (!str) return put_u32(0xffffffffu);
                const size_t l = std::strlen(str);
                put_u32(uint32_t(l));
                key.append(str, l);
            };
            put_u32(p.n_parent_tipset_cids);
            for (uint32_t k = 0; k < p.n_parent_tipset_cids; ++k)
                put_str(p.parent_tipset_cids ? p.parent_tipset_cids[k] : nullptr);
            put_str(p.child_block_cid);
            auto it = ctx_index.find(key);
            if (it == ctx_index.end()) {
                if (p.n_parent_tipset_cids > uint32_t(IPCFP_MAX_PARENTS_WIDE)) {
                    err = "proof " + std::to_string(i) + " names " + std::to_string(p.n_parent_tipset_cids) +
                          " parent blocks (engine limit " + std::to_string(uint32_t(IPCFP_MAX_PARENTS_WIDE)) + ")";
                    return IPCFP_E_UNSUPPORTED;
                }
                ipcfp_tipset_ref_t tr;
                std::memset(&tr, 0, sizeof tr);
                tr.n_parents = p.n_parent_tipset_cids;
                uint8_t* more = nullptr;  // a key wider than the inline form keeps its tail in the handle
                if (tr.n_parents > uint32_t(IPCFP_MAX_PARENTS)) {
                    out.more_parents.emplace_back(new std::vector<uint8_t>(size_t(tr.n_parents - IPCFP_MAX_PARENTS) * IPCFP_CID_SLOT));
                    more = out.more_parents.back()->data();
                    tr.more_parents = more;
                }
                bool all = true;
                for (uint32_t k = 0; k < tr.n_parents; ++k) {
                    bool parsed, canon;
                    CidKey key40;
                    parse_cid_claim(p.parent_tipset_cids ? p.parent_tipset_cids[k] : nullptr, key40, parsed, canon);
                    std::memcpy(k < uint32_t(IPCFP_MAX_PARENTS) ? tr.parents[k] : more + size_t(k - IPCFP_MAX_PARENTS) * IPCFP_CID_SLOT, key40.w,
                                IPCFP_CID_SLOT);
                    all = all && parsed;
                }
                if (all) tr.flags |= TC_PARENTS_PARSED;
                bool parsed, canon;
                CidKey key40;
                parse_cid_claim(p.child_block_cid, key40, parsed, canon);
                std::memcpy(tr.child, key40.w, IPCFP_CID_SLOT);
                if (parsed) tr.flags |= TC_CHILD_PARSED;
                ci = uint32_t(out.tipsets.size());
                out.tipsets.push_back(tr);
                ctx_index.emplace(std::move(key), ci);
            } else {
                ci = it->second;
            }
            last_parents = p.parent_tipset_cids;
            last_child = p.child_block_cid;
            last_np = p.n_parent_tipset_cids;
            last_ctx = ci;
            have_last = true;
        }
        out.claims[i].context = ci;
    }
    // ---- pass 2 (parallel by contiguous ranges): message CID, topics, data ----
    unsigned hw = std::thread::hardware_concurrency();
    if (hw == 0) hw = 1;
    const uint64_t per_thread_min = 4096;
    const unsigned n_threads = unsigned(std::max<uint64_t>(1, std::min<uint64_t>({uint64_t(hw), 32, n / per_thread_min})));
    std::vector<std::vector<uint8_t>> blobs(n_threads);
    auto work = [&](unsigned t) {
        const uint64_t lo = n * t / n_threads, hi = n * (t + 1) / n_threads;
        std::vector<uint8_t>& b = blobs[t];
        b.reserve(size_t(hi - lo) * 100);
        for (uint64_t i = lo; i < hi; ++i) {
            const uint32_t ctx_keep = out.claims[i].context;
            lower_one(proofs[i], out.claims[i], b);
            out.claims[i].context = ctx_keep;
        }
    };
    if (!run_parts(n_threads, work)) {  // (host/parallel.h: a part that ran out of memory is an error code, not a terminate)
        err = "out of memory while lowering the claims";
        return IPCFP_E_NOMEM;
    }
    // ---- pass 3: one blob, offsets rebased ----
    uint64_t total = 0;
    for (auto& b : blobs) total += b.size();
    if (total >= 0xf0000000ULL) {
        err = "claim blob too large";
        return IPCFP_E_UNSUPPORTED;
    }
    out.blob.resize(total);
    uint64_t base = 0;
    for (unsigned t = 0; t < n_threads; ++t) {
        const uint64_t lo = n * t / n_threads, hi = n * (t + 1) / n_threads;
        if (!blobs[t].empty()) std::memcpy(out.blob.data() + base, blobs[t].data(), blobs[t].size());
        if (base)
            for (uint64_t i = lo; i < hi; ++i) {
                out.claims[i].topics_off += uint32_t(base);
                if (out.claims[i].flags & EC_DATA_MATCHABLE) out.claims[i].data_off += uint32_t(base);
            }
        base += blobs[t].size();
    }
    return IPCFP_OK;
}

}  // namespace ipcfp

namespace ipcfp {

// One StorageProof → ipcfp_storage_claim_t: the same lowering ipcfp_verify_storage_proofs performs
// (host/verify_storage.cpp), kept here as its own copy so that the exported, parallel packer could be added
// without touching that path; the two are unified once this one has run on a GPU-side test.
//   child_block_cid           must parse                                    (storage/verifier.rs:85)
//   parent_state_root, actor_state_cid, storage_root   compared as STRINGS with `Cid::to_string()` of the
//                             decoded value, so only a canonical spelling can match (:110,126,144)
//   slot   hex::decode_to_slice(slot.trim_start_matches("0x"), &mut [u8; 32])  (:155-157)
//   value  compared with "0x" + hex(padded) ignoring ASCII case               (:160-169)
static void lower_storage_one(const ipcfp_storage_proof_t& p, StorageClaimPacked& c) {
    std::memset(&c, 0, sizeof c);
    c.child_epoch = p.child_epoch;
    c.actor_id = p.actor_id;
    bool parsed, canon;
    parse_cid_claim(p.child_block_cid, c.child, parsed, canon);
    if (parsed) c.flags |= SC_CHILD_PARSED;
    parse_cid_claim(p.parent_state_root, c.state_root, parsed, canon);
    if (parsed && canon) c.flags |= SC_STATE_ROOT_CANON;
    parse_cid_claim(p.actor_state_cid, c.actor_state, parsed, canon);
    if (parsed && canon) c.flags |= SC_ACTOR_STATE_CANON;
    parse_cid_claim(p.storage_root, c.storage_root, parsed, canon);
    if (parsed && canon) c.flags |= SC_STORAGE_ROOT_CANON;
    std::vector<uint8_t> b;
    if (p.slot) {
        const char* s = p.slot;
        while (s[0] == '0' && s[1] == 'x') s += 2;
        if (std::strlen(s) == 64 && hex_decode(s, 64, b)) {
            std::memcpy(c.slot, b.data(), 32);
            c.flags |= SC_SLOT_PARSED;
        }
    }
    if (p.value && std::strlen(p.value) == 66 && p.value[0] == '0' && (p.value[1] == 'x' || p.value[1] == 'X')) {
        if (hex_decode(p.value + 2, 64, b)) {
            std::memcpy(c.value, b.data(), 32);
            c.flags |= SC_VALUE_MATCHABLE;
        }
    }
}

}  // namespace ipcfp

struct ipcfp_packed_events {
    ipcfp::PackedEvents p;
};

extern "C" {

int ipcfp_pack_event_proofs(const ipcfp_event_proof_t* proofs, uint64_t n, ipcfp_packed_events_t** out) {
    if (!out || (n && !proofs)) return IPCFP_E_INVALID;
    *out = nullptr;
    if (n >= 0xffffffffULL) return IPCFP_E_UNSUPPORTED;
    auto* h = new (std::nothrow) ipcfp_packed_events();
    if (!h) return IPCFP_E_NOMEM;
    std::string err;
    const int rc = ipcfp::pack_event_claims_host(proofs, n, h->p, err);
    if (rc) {
        delete h;
        return rc;
    }
    *out = h;
    return IPCFP_OK;
}
void ipcfp_packed_events_destroy(ipcfp_packed_events_t* p) { delete p; }
const ipcfp_tipset_ref_t* ipcfp_packed_events_tipsets(const ipcfp_packed_events_t* p, uint32_t* n) {
    if (n) *n = p ? uint32_t(p->p.tipsets.size()) : 0;
    return p ? p->p.tipsets.data() : nullptr;
}
const ipcfp_event_claim_t* ipcfp_packed_events_claims(const ipcfp_packed_events_t* p, uint64_t* n) {
    if (n) *n = p ? p->p.claims.size() : 0;
    return p ? reinterpret_cast<const ipcfp_event_claim_t*>(p->p.claims.data()) : nullptr;
}
const uint8_t* ipcfp_packed_events_blob(const ipcfp_packed_events_t* p, uint64_t* len) {
    if (len) *len = p ? p->p.blob.size() : 0;
    return p ? p->p.blob.data() : nullptr;
}

// Host-only, parallel: n StorageProof structs → n ipcfp_storage_claim_t written to `claims` (caller's array).
int ipcfp_pack_storage_proofs(const ipcfp_storage_proof_t* proofs, uint64_t n, ipcfp_storage_claim_t* claims) {
    if (n && (!proofs || !claims)) return IPCFP_E_INVALID;
    auto* out = reinterpret_cast<ipcfp::StorageClaimPacked*>(claims);
    unsigned hw = std::thread::hardware_concurrency();
    if (hw == 0) hw = 1;
    const unsigned n_threads = unsigned(std::max<uint64_t>(1, std::min<uint64_t>({uint64_t(hw), 32, n / 4096})));
    auto work = [&](unsigned t) {
        const uint64_t lo = n * t / n_threads, hi = n * (t + 1) / n_threads;
        for (uint64_t i = lo; i < hi; ++i) ipcfp::lower_storage_one(proofs[i], out[i]);
    };
    return ipcfp::run_parts(n_threads, work) ? IPCFP_OK : IPCFP_E_NOMEM;
}

}  // extern "C"
