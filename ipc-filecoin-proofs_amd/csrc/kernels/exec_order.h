// csrc/kernels/exec_order.h — device state of one tipset context: everything `verify_single_proof`
// derives from (parent_tipset_cids, child_block_cid) alone — header consistency facts
// (src/proofs/events/verifier.rs:147-181) and the reconstructed execution order
// (src/proofs/events/utils.rs:16-30,48-94) — computed ONCE per distinct pair instead of once per
// proof (the reference recomputes it for every proof: events/verifier.rs:190).
#pragma once
#include <cstdint>

#include "amt_enum.h"
#include "event_table.h"
#include "types_dev.h"

namespace ipcfp {

struct TipsetCtxDev {
    // inputs
    uint32_t flags;      // TC_* (claims_dev.h)
    uint32_t n_parents;
    CidKey child;
    CidKey parents[kMaxParents];
    // header facts (k_ctx_headers)
    uint32_t child_status;     // TRUE or ERR_* of `get(child)` + HeaderLite decode
    uint32_t parents_match;    // child_hdr.parents == parent_cids
    long long child_height;
    CidKey receipts_root;      // child_hdr.parent_message_receipts
    uint32_t parent0_status;   // TRUE or ERR_* for parent_cids[0]
    uint32_t pad0;
    long long parent0_height;
    // execution order (filled by the host after the enumeration)
    uint32_t exec_status;      // TRUE or the first ERR_* of reconstruct_execution_order
    uint32_t exec_mask;        // hash-table size - 1
    const unsigned long long* exec_slots;  // open addressing over message CIDs: {fingerprint, FIRST raw position}
    const CidKey* exec_keys;      // raw for_each sequence (with duplicates)
    const uint32_t* exec_pos;     // raw position → execution index (valid where the position is a first occurrence)
    const uint32_t* exec_inv;     // execution index → raw position of that message's first occurrence (k_exec_finish)
    uint64_t exec_len;            // number of distinct messages
    // receipts AMT enumerated once per context (amt_enum.hip): when it decoded without error and is
    // dense, `Amt::get(exec_index)` is a table lookup — every node on every path was already validated
    const LeafRef* receipt_leaves;
    uint64_t n_receipt_leaves;
    uint64_t receipt_first;       // index of receipt_leaves[0] (0, or the first receipt of a shard witness)
    // the event table of those receipts (event_table.h), aligned with receipt_leaves; null: walk every claim
    const ReceiptRec* receipt_recs;
    const EventRec* event_recs;
};

}  // namespace ipcfp
