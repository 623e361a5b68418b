// csrc/kernels/event_table.h — every event of a receipt range, decoded ONCE per witness.
//
// The reference decodes an events AMT every time a proof touches it: once per receipt in each pass of
// find_matching_events (src/proofs/events/generator.rs:215-233, 259-297), and again per EventProof in
// verify_receipt_and_event (src/proofs/events/verifier.rs:234-239) + extract_evm_log (common/evm.rs:13-59).  On the
// device a decode is a chain of dependent loads through one block per lane — the cost of both the scan and the
// verify kernel.  Here the first pass over a receipt range leaves one fixed-size record per event (where the EVM
// log's topics and data lie inside the block, the emitter, the decode outcome), and everything after it — the
// match count of another filter, PASS 2's match list, verify_event_data_matches of a million claims — reads
// records and compares bytes at known addresses.  Like the enumerations it belongs to the witness and is dropped
// by ipcfp_witness_rebuild_index.
//
// Only the overwhelmingly common shape is tabulated: an events AMT whose root is a leaf of at most 64 slots
// (FVM writes bit width 5) holding events of less than 64 KB.  Anything else is marked RK_WALK and takes the
// general walkers (event_scan.hip, walk_dev.h), so outcomes never depend on the table.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "ipcfp.h"

namespace ipcfp {

enum : uint32_t {
    RK_NO_EVENTS = 0,  // receipt.events_root is null
    RK_TABLE = 1,      // events tabulated: `bitmap` names the present indices, records at [first, first + popcount)
    RK_WALK = 2,       // not tabulated (tall AMT, wide node, oversized event, record pool exhausted): walk it
    // values >= 64: the ERR_* status `Amt::load(events_root)` fails with (ipcfp_status_t)
};

struct ReceiptRec {
    uint32_t kind;
    uint32_t first;    // index of the first EventRec
    uint64_t bitmap;   // bit j ⇔ event index j exists
    uint32_t block;    // the events AMT's root block
    uint32_t pad;
};

struct EventRec {
    uint64_t base_flags;    // bits 0..47: arena offset of the StampedEvent item; 48..55: topic count; 56: is an EVM
                            // log (extract_evm_log → Some); 57: Case A (one concatenated "topics" value)
    uint64_t emitter;
    uint16_t topic_rel[4];  // offsets of t1..t4 from the item start (Case A: [0] = start of the concatenation)
    uint16_t data_rel;      // offset of the data bytes from the item start
    uint16_t ev_len;        // encoded length of the item
    uint32_t data_len;
};
static_assert(sizeof(EventRec) == 32 && sizeof(ReceiptRec) == 24, "record layouts");

constexpr uint64_t kEvBaseMask = (1ull << 48) - 1;
constexpr int kEvTopicShift = 48;
constexpr uint64_t kEvIsLog = 1ull << 56, kEvCaseA = 1ull << 57;

// The table is built in two steps.
//   k_block_events (block_events.hip)  EVERY block of the witness, in arena order, is parsed as if it were the root of
//     an events AMT: a wavefront copies the contiguous arena span of its 64 blocks into LDS with coalesced 16-byte
//     loads and every lane parses its own block out of LDS (the reader's chunk load is a ds_read_b128).  It needs
//     nothing but the arena — neither the CID index nor the receipts — so it runs on its own stream beside K1 and
//     the receipts enumeration.  Outcome per block: a BlockRec.  A block that is not exactly the tabulated shape
//     (or is no events AMT at all) is RK_WALK, which decides nothing.
//   k_receipt_events (event_scan.hip)  one receipt per lane: events_root → block id → BlockRec → ReceiptRec.
//     Receipts whose block is RK_WALK take the general walkers (k_receipt_walk), as before.
struct BlockRec {
    uint32_t kind_matches;  // bits 0..7: RK_TABLE | RK_WALK; bits 8..31: events matching the filter the table was built with
    uint32_t first;         // index of the first EventRec
    uint64_t bitmap;
};
static_assert(sizeof(BlockRec) == 16, "record layout");

// The EventRec pool is cut into kPoolParts equal partitions, each with its own fill counter on its own 128-byte line:
// a wavefront reserves the records of its blocks with ONE atomic on the counter of partition (wavefront number mod
// kPoolParts).  One counter for the whole pool made every wavefront of the chip wait in line at a single L2 address.
constexpr uint32_t kPoolParts = 256;  // at most; a small witness uses one partition per wavefront (pool_parts below)
__host__ __device__ inline uint32_t pool_parts(uint32_t n_blocks) {
    const uint32_t waves = (n_blocks + 63u) / 64u;
    return waves < 1u ? 1u : (waves > kPoolParts ? kPoolParts : waves);
}
constexpr uint32_t kPoolCounterStride = 32;  // uint32 words between two counters

// filter of one scan (EventMatcher + the optional emitter filter, src/proofs/events/generator.rs:25-40,220-224)
struct ScanParams {
    ipcfp_event_filter_t filter;
    uint64_t actor;
    uint32_t has_actor;
    uint32_t pad;
};

// device view handed to the kernels (null pointers: no table)
struct EventTableView {
    const ReceiptRec* receipts;  // one per enumerated receipt leaf, same order as the LeafRef table
    const EventRec* events;
};

}  // namespace ipcfp
