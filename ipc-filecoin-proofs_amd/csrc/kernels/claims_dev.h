// csrc/kernels/claims_dev.h — packed (binary) proof claims as the device kernels read them.
//
// The reference's claims carry CIDs and hex values as strings
// (src/proofs/events/bundle.rs:5-23, src/proofs/storage/bundle.rs:5-14).  The host parses each
// string ONCE (host/claims.cpp) and records, per field, whether it parsed and whether it is the
// canonical `Cid::to_string()` form — the two facts the reference's `Cid::try_from(str)?` and
// `cid.to_string() == claim` observe — so the kernels reproduce Err / Ok(false) exactly.
#pragma once
#include <cstdint>

#include "ipcfp.h"
#include "witness_dev.h"

namespace ipcfp {

// ---- storage ----
enum : uint32_t {
    SC_CHILD_PARSED = 1u << 0,        // child_block_cid parses (else Err at storage/verifier.rs:85)
    SC_STATE_ROOT_CANON = 1u << 1,    // parent_state_root parses AND equals its canonical to_string()
    SC_ACTOR_STATE_CANON = 1u << 2,
    SC_STORAGE_ROOT_CANON = 1u << 3,
    SC_SLOT_PARSED = 1u << 4,         // slot hex decodes to exactly 32 bytes
    SC_VALUE_MATCHABLE = 1u << 5,     // value is "0x"/"0X" + 64 hex digits (else it can never compare equal)
};

struct StorageClaimPacked {
    long long child_epoch;
    uint64_t actor_id;
    CidKey child, state_root, actor_state, storage_root;
    uint8_t slot[32];
    uint8_t value[32];
    uint32_t flags;
    uint32_t pad;
};

static_assert(sizeof(StorageClaimPacked) == sizeof(ipcfp_storage_claim_t), "packed storage claim layout");
static_assert(SC_CHILD_PARSED == IPCFP_SCLAIM_CHILD_PARSED && SC_VALUE_MATCHABLE == IPCFP_SCLAIM_VALUE_MATCHABLE &&
                  SC_SLOT_PARSED == IPCFP_SCLAIM_SLOT_PARSED && SC_STATE_ROOT_CANON == IPCFP_SCLAIM_STATE_ROOT_CANON,
              "flags");

// ---- events ----
enum : uint32_t {
    EC_MSG_PARSED = 1u << 0,          // message_cid parses (else Err at events/verifier.rs:193)
    EC_DATA_MATCHABLE = 1u << 1,      // data is "0x" + an even number of hex digits
};

// A tipset context: everything `verify_single_proof` derives from (parent_tipset_cids,
// child_block_cid) alone, shared by all proofs that name the same pair.
enum : uint32_t {
    TC_PARENTS_PARSED = 1u << 0,      // every parent CID string parses (events/verifier.rs:130)
    TC_CHILD_PARSED = 1u << 1,        // child CID string parses (:131)
};

struct EventClaimPacked {   // layout == ipcfp_event_claim_t (include/ipcfp.h)
    long long parent_epoch, child_epoch;
    uint64_t exec_index, event_index, emitter;
    CidKey message;
    uint32_t context;       // index into the tipset-context table
    uint32_t flags;
    uint32_t n_topics;      // claimed topic count
    uint32_t topics_off;    // blob offset of n_topics × 33 bytes: [matchable flag, 32 topic bytes]
    uint32_t data_off;      // blob offset of the claimed data bytes
    uint32_t data_len;
};
static_assert(sizeof(EventClaimPacked) == sizeof(ipcfp_event_claim_t), "packed claim layout");
static_assert(EC_MSG_PARSED == IPCFP_CLAIM_MSG_PARSED && EC_DATA_MATCHABLE == IPCFP_CLAIM_DATA_MATCHABLE, "flags");
static_assert(TC_PARENTS_PARSED == IPCFP_TIPSET_PARENTS_PARSED && TC_CHILD_PARSED == IPCFP_TIPSET_CHILD_PARSED, "flags");

}  // namespace ipcfp
