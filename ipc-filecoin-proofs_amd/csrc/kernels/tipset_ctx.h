// csrc/kernels/tipset_ctx.h — device state of one tipset context: everything `verify_single_proof` derives from
// (parent_tipset_cids, child_block_cid) alone — header consistency facts (src/proofs/events/verifier.rs:147-181) and the
// reconstructed execution order (src/proofs/events/utils.rs:16-30,48-94) — computed ONCE per distinct pair instead of once
// per proof (the reference recomputes it for every proof: events/verifier.rs:190); and the job record of the tipset
// prologue.  Free of the walk primitives so that the prologue can be compiled with an LDS reader (tipset_prepare.hip).
#pragma once
#include <cstdint>

#include "amt_types.h"
#include "event_table.h"
#include "ipcfp.h"
#include "witness_dev.h"

namespace ipcfp {

// the INPUTS of a context: the head of TipsetCtxDev, which the tipset prologue can also take as a kernel argument
// (tipset_prepare.hip: one H2D copy less at the head of a verification call — 18 µs beside the side streams' grids)
struct TipsetInputs {
    uint32_t flags;      // TC_* (claims_dev.h)
    uint32_t n_parents;
    CidKey child;
    CidKey parents[IPCFP_MAX_PARENTS];
};

struct TipsetCtxDev {
    // inputs (= TipsetInputs)
    uint32_t flags;      // TC_* (claims_dev.h)
    uint32_t n_parents;
    CidKey child;
    CidKey parents[IPCFP_MAX_PARENTS];
    // header facts (k_ctx_headers)
    uint32_t child_status;     // TRUE or ERR_* of `get(child)` + HeaderLite decode
    uint32_t parents_match;    // child_hdr.parents == parent_cids
    long long child_height;
    CidKey receipts_root;      // child_hdr.parent_message_receipts
    uint32_t parent0_status;   // TRUE or ERR_* for parent_cids[0]
    uint32_t pad0;
    unsigned long long prologue_general;  // bit s: slot s of the tipset prologue is left to the general kernel (a block larger than the LDS stage)
    long long parent0_height;
    // 1 + the TxMeta block of parent b when its re-hash was LEFT to k_txmeta_rehash (amt_enum.hip; tipset_prepare.hip
    // roots_slot with `defer_rehash`); 0: nothing to re-hash (checked inline, or never reached) — so that a context
    // taken from zeroed memory needs no initialisation.
    uint32_t txmeta_block[IPCFP_MAX_PARENTS];
    // A tipset key of MORE than IPCFP_MAX_PARENTS blocks (the reference takes any: src/proofs/events/verifier.rs:147-181,
    // src/proofs/events/utils.rs:16-30): ALL n_parents keys in HBM, read through tipset_parent(); null for the keys every
    // chain has, whose parents travel inline (and as kernel arguments: TipsetInputs).  A wide context never takes the
    // single-launch prologue or the dense walk — host/tipset_wide.h.
    const CidKey* parents_wide;
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

static_assert(sizeof(TipsetInputs) == 8 + 40 * (1 + IPCFP_MAX_PARENTS), "TipsetInputs is the head of TipsetCtxDev");

// parent block b of a context's tipset key (b < n_parents)
__device__ __forceinline__ CidKey tipset_parent(const TipsetCtxDev& c, uint32_t b) {
    return c.parents_wide ? c.parents_wide[b] : c.parents[b < IPCFP_MAX_PARENTS ? b : 0u];
}

// what k_ctx_finish (verify_events.hip) writes into a context on the device
struct CtxFinish {
    const unsigned long long* err;   // packed first error of the execution-order reconstruction (kNoEnumError: none)
    const uint64_t* total;           // number of distinct messages
    const uint32_t* first;           // raw position → 1 iff first occurrence
    const uint32_t* pos;             // raw position → execution index
    uint32_t* inv;                   // execution index → raw position (filled here)
    const unsigned long long* slots;
    const CidKey* keys;
    uint32_t mask, raw_len;
    const LeafRef* receipt_leaves;
    uint64_t n_receipt_leaves, receipt_first;
    const ReceiptRec* receipt_recs;
    const EventRec* event_recs;
};

// One context's share of the tipset prologue (launch_tipset_prepare): two wavefronts decode the child and the first
// parent header, one wavefront per parent block decodes its header and TxMeta and re-hashes it.
struct PrepareJob {
    TipsetCtxDev* ctx;
    AmtRootSpec* roots;          // 2·P message-AMT roots + the receipts root; nullptr: no execution order for this context
    unsigned long long* err;
};
constexpr uint32_t kPrepareSlots = 2 + IPCFP_MAX_PARENTS;
static_assert(kPrepareSlots <= 64, "TipsetCtxDev::prologue_general is one bit per slot");
// the jobs travel as a kernel ARGUMENT when there are at most this many (a verification call has one context per
// distinct tipset pair — usually one): one H2D copy less at the head of the call
constexpr uint32_t kInlineJobs = 4;
struct PrepareJobs {
    PrepareJob inline_jobs[kInlineJobs];
    const PrepareJob* more;  // non-null: the jobs are in device memory instead
};
__device__ __forceinline__ PrepareJob prepare_job(const PrepareJobs& j, uint32_t i) { return j.more ? j.more[i] : j.inline_jobs[i]; }
// LDS stage of the prologue (tipset_prepare.hip): a block header is 0.6-2 KB, a TxMeta 90 bytes.  A witness whose
// largest block fits needs no general companion launch.
constexpr uint32_t kPrologueStageChunks = 512;  // 8 KB

}  // namespace ipcfp
