// oracle/verify.hpp — TEST INFRASTRUCTURE.  The reference's verifiers and the event scan, restated.
#pragma once
#include "hamt.hpp"
#include "types.hpp"

namespace orc {

// TrustPolicy::{verify_parent_tipset, verify_child_header} (src/proofs/trust/mod.rs:53-78,
// src/cert.rs:52-64) with `.unwrap_or(false)` applied (src/proofs/verifier.rs:21-25,38-49).
bool trusted(const ipcfp_trust_policy_t* t, int64_t epoch);

// reconstruct_execution_order (src/proofs/events/utils.rs:16-30) → collect_exec_list(verify_txmeta = true) (:48-94)
std::vector<Cid> reconstruct_execution_order(const Blockstore& bs, const std::vector<Cid>& parent_hdr_cids);

// Baseline variant B2 ("fair"): the execution order of a tipset key computed once and
// looked up through a hash map instead of being rebuilt and linearly searched per proof.
struct ExecCache {
    bool ok = false;
    uint8_t err_status = 0;                                // Err of reconstruct_execution_order, if any
    // message CID → execution index; sub-maps selected by hash bits so the all-cores build fills them in parallel
    using Map = std::unordered_map<Bytes, uint64_t, BytesHash>;
    std::vector<Map> shards = std::vector<Map>(1);
    size_t mask = 0;
    uint64_t len = 0;                                      // number of distinct messages
    const uint64_t* find(const Bytes& key) const {
        const Map& m = shards[(BytesHash()(key) >> 40) & mask];
        auto it = m.find(key);
        return it == m.end() ? nullptr : &it->second;
    }
};

// verify_single_proof (src/proofs/events/verifier.rs:92-121) → status byte; Err is thrown.
// exec_cache == nullptr: exactly as written (execution order rebuilt for this proof, :190).
uint8_t verify_event_proof_one(const Blockstore& bs, const ipcfp_event_proof_t& p, const ipcfp_trust_policy_t* trust,
                               const ipcfp_event_filter_t* filter, const ExecCache* exec_cache = nullptr);
// threads == 1: reconstruct_execution_order as written, then indexed.  Otherwise (0 = every processor) the ten
// message AMTs are enumerated and de-duplicated on `threads` OpenMP threads — same order, same Err.
ExecCache build_exec_cache(const Blockstore& bs, const std::vector<Cid>& parents, int threads = 1);

// verify_storage_proof steps 2-6 (src/proofs/storage/verifier.rs:24-63) over an already loaded store.
uint8_t verify_storage_proof_one(const Blockstore& bs, const ipcfp_storage_proof_t& p,
                                 const ipcfp_trust_policy_t* trust);

// read_storage_slot (src/proofs/storage/decode.rs:36-97); true ⇒ Some(value)
bool read_storage_slot(const Blockstore& bs, const Cid& root, const uint8_t slot[32], Bytes& value);

// get_actor_state (src/proofs/common/decode.rs:17-42)
ActorState get_actor_state(const Blockstore& bs, const Cid& state_root, uint64_t actor_id);

// Address::new_id(id).to_bytes()
Bytes id_address_bytes(uint64_t id);

struct ScanMatch {
    uint64_t exec_index, event_index, emitter;
    std::vector<std::array<uint8_t, 32>> topics;
    Bytes data;
};
// find_matching_events (src/proofs/events/generator.rs:180-307) with the RPC receipt list replaced by
// the receipts AMT itself (for_each order).  `receipt_has_match[i]` = pass-1 verdict per receipt index;
// `matches` = pass-2 (exec_index, event_index, EventData) in emission order; `touched` (nullable) = the
// union of every RecordingBlockStore's take_seen() that generate_event_proof feeds to the collector
// for this step (rec_events per matching receipt + rec_receipts), sorted in `Cid: Ord`.
// `threads` != 1 (0 = every processor): the receipt list and PASS 1 run on OpenMP threads (baseline variant B2
// all-cores); outcomes — including which Err surfaces first — are those of the sequential loops.
void scan_events(const Blockstore& bs, const Cid& receipts_root, const ipcfp_event_filter_t& filter, bool has_actor,
                 uint64_t actor, std::vector<uint8_t>& receipt_has_match, std::vector<ScanMatch>& matches,
                 std::vector<Cid>* touched, int threads = 1);

// ---- generator side (offline: the RPC-backed store of the reference is any Blockstore) ----
struct GeneratedEventProof {
    uint64_t exec_index, event_index, emitter;
    Cid message_cid;
    std::vector<std::array<uint8_t, 32>> topics;
    Bytes data;
};
struct GeneratedEventBundle {
    std::vector<GeneratedEventProof> proofs;
    std::vector<Cid> witness;  // `WitnessCollector::materialize()` order: BTreeSet<Cid>
};
// generate_event_proof (src/proofs/events/generator.rs:60-107).  What the reference reads from the RPC
// tipset JSON (child.blocks[0].parent_message_receipts, parent.blocks[i].messages) is read from the
// headers in the store.
GeneratedEventBundle generate_event_proof(const Blockstore& bs, const std::vector<Cid>& parent_cids, const Cid& child_cid,
                                          const ipcfp_event_filter_t& filter, bool has_actor, uint64_t actor);

struct GeneratedStorageProof {
    Cid parent_state_root, actor_state_cid, storage_root;
    uint8_t value[32];
    std::vector<Cid> witness;  // materialize() order
};
// generate_storage_proof (src/proofs/storage/generator.rs:29-67)
GeneratedStorageProof generate_storage_proof(const Blockstore& bs, const Cid& child_cid, uint64_t actor_id,
                                             const uint8_t slot[32]);

}  // namespace orc
