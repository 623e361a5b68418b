// csrc/kernels/storage_runs.h — what consecutive storage claims share.
//
// `verify_storage_proof` (src/proofs/storage/verifier.rs:24-63) derives, for every proof anew, facts that depend only on
// (child_block_cid, parent_state_root, actor_id, actor_state_cid, storage_root): the child header's parent_state_root
// (:95-111), the actor's state CID behind the state root (:114-127, common/decode.rs:17-42), the EVM state's
// contract_state (:130-145, common/decode.rs:79-97) and which of the six layouts the storage root decodes as
// (storage/decode.rs:46-96).  A bundle asks them of one contract hundreds of times in a row (src/proofs/verifier.rs:19-28
// walks the proofs in order).  Here a RUN = a maximal stretch of consecutive claims that agree on those five fields; the
// facts are computed once per run (exact key comparison, pure functions of the witness) and every claim is then judged
// in the reference's order of checks from its run's record plus its own flags, slot and value.
#pragma once
#include <cstdint>

#include "witness_dev.h"

namespace ipcfp {

struct StorageRun {
    uint32_t first_claim;        // index of the run's first claim
    uint32_t hdr_status;         // TRUE or the ERR_* of get(child) + HeaderLite decode
    CidKey parent_state_root;
    uint32_t sr_status;          // StateRoot block: TRUE / ERR_MISSING_BLOCK / ERR_DECODE
    uint32_t actor_status;       // Hamt get + ActorState decode: TRUE / ERR_ACTOR_NOT_FOUND / ERR_* / kCoopPunt (undecided)
    CidKey actors;               // StateRoot.actors
    CidKey actor_state;          // ActorState.state of the run's actor
    uint32_t evm_status;         // get(actor_state_cid) + parse_evm_state: TRUE / ERR_MISSING_BLOCK / ERR_DECODE
    uint32_t root_kind;          // 0 = inline A1, 1 = A2, 2 = A3, 3 = HAMT (hamt_root, hamt_bw: B1 / B2 / C), 4 = root block missing
    CidKey contract_state;       // EvmState.contract_state
    CidKey hamt_root;
    uint32_t hamt_bw;
    uint32_t pad;
};

}  // namespace ipcfp
