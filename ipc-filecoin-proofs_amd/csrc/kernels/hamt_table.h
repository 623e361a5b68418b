// csrc/kernels/hamt_table.h — every HAMT node of a witness decoded ONCE per call.
//
// `Hamt::get` (fvm_ipld_hamt 0.10.4; src/proofs/common/decode.rs:29-39, src/proofs/storage/decode.rs:79-96) decodes
// every node on the path completely — serde decodes the whole `[bitfield, [pointer…]]` — and the reference does so for
// every lookup anew: the 256 storage proofs of one contract decode its storage root 256 times
// (src/proofs/verifier.rs:19-28), 66 k actor gets decode the state tree's root 66 k times.  A node's decode is a pure
// function of the block, so a large batch first tabulates EVERY block of the witness as if it were a HAMT node — one
// lane per block in arena order, the pass the event table makes for events AMTs (event_table.h) — and a get is then
// SHA-256 + one record per level: bitfield popcount, the pointer's offset, a 43-byte link or a bucket of ≤ 3 entries.
//
// What a record says:
//   status    1: the block is `[bytes(≤ 8), [≤ 32 pointers]]`, every pointer a well-formed link or a bucket of
//                `[key bytes, value]` pairs with well-formed values, nothing after it;   0: anything else — NOT
//                tabulated, which decides nothing: a get that meets such a block is left to the walkers (walk_dev.h).
//   kinds_ok  bit k: every bucket value of the node also passes the TYPED check of value kind k (the element type the
//                HAMT is opened with: serde decodes values as ActorState, Vec<u8>, …) — a node that is well-formed
//                but holds a value of the wrong type is a decode error for that kind, exactly as when walked.
// So outcomes never depend on the table; it only ever answers what the walk would answer.
#pragma once
#include <cstdint>

namespace ipcfp {

constexpr uint32_t kHamtTablePointers = 32;
// The per-call node table is made by two kernels (hamt_table_lane.hip): blocks of at least this many bytes — the 4-5 KB
// bucket nodes of a state tree — go to the 32-lane outline (hamt_levels.hip), shorter ones to one lane each; a long block
// the outline does not take (it reads ActorState buckets only) gets its lane afterwards.
constexpr uint32_t kHamtOutlineMinLen = 2048;
enum : uint32_t { HK_ACTOR_STATE = 1u << 0, HK_VEC_U8 = 1u << 1, HK_ANY = 1u << 2,
                  HK_ITEM_BY_ITEM = 1u << 6 };  // (a request to the node parse, not a kind: no plain-read entry path — A/B runs)

struct HamtNodeRec {
    uint8_t status;      // 1: tabulated
    uint8_t kinds_ok;    // HK_* bits
    uint8_t np;          // pointers
    uint8_t pad;
    uint32_t std_links;  // bit p: pointer p is the standard 43-byte link (CID = 38 bytes at ptr_off[p] + 5)
    uint64_t bitfield;   // big-endian bytes of the node's bitfield as an integer (≤ 64 bits)
    uint16_t ptr_off[kHamtTablePointers];  // offset of pointer p inside the block
};
static_assert(sizeof(HamtNodeRec) == 80, "record layout");

// The bucket entries of ONE visited state-tree node (kernels/hamt_levels.hip): what the 32-lane parse knows anyway — where
// every entry's key and ActorState lie — kept for the queries that stand on the node, so that a query's bucket search is
// its pointer's ≤ 3 entries out of this table and one key compare each, instead of a CBOR reader walking the bucket again
// (6.1 k instructions and 100 dependent reads per wavefront of 64 queries: profiles/r04_experiments.md).
constexpr uint32_t kHamtTabEntries = 96;
struct HamtEntryTab {
    uint32_t links;                 // bit p: pointer p is a link (of any spelling) — everything else is a bucket
    uint32_t n_entries;
    uint8_t first[kHamtTablePointers];  // bucket pointer p: index of its first entry …
    uint8_t count[kHamtTablePointers];  // … and how many it holds (0: a link, or an empty bucket)
    struct Entry {
        uint16_t key_off, val_off, val_len;  // inside the block: the key's bytes, the ActorState item
        uint8_t key_len, pad;
    } e[kHamtTabEntries];
};
static_assert(sizeof(HamtEntryTab) == 8 + 64 + 8 * kHamtTabEntries, "entry table layout");

}  // namespace ipcfp
