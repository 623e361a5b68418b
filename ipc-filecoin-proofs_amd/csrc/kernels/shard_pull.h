// csrc/kernels/shard_pull.h — what the pull kernels (shard_pull.hip) and their host loop (host/shard_pull.cpp) share.
#pragma once
#include <cstdint>

#include "witness_dev.h"

namespace ipcfp {

enum PullKind : uint32_t {
    PK_LEAF = 0,        // copied, never expanded (leaves of trees that are needed whole)
    PK_HDR_CHILD = 1,   // child block header → receipts root            (src/proofs/events/verifier.rs:214-220)
    PK_HDR_PARENT = 2,  // parent block header → TxMeta                  (src/proofs/events/utils.rs:54-61)
    PK_TXMETA = 3,      // [bls_root, secp_root] → the two message AMTs  (utils.rs:61-90)
    PK_MSG_ROOT = 4,    // Amtv0<Cid> root, the whole tree
    PK_MSG_NODE = 5,
    PK_RCPT_ROOT = 6,   // Amtv0<Receipt> root: count → [lo, hi); the paths to it   (events/generator.rs:195-196,249)
    PK_RCPT_NODE = 7,   //   item.base = first index under the node, height in bits 8..15 of item.kind
    PK_EV_ROOT = 8,     // Amt<StampedEvent> root of a receipt in [lo, hi), the whole tree (events/generator.rs:215,259)
    PK_EV_NODE = 9,
};

struct PullItem {
    uint32_t id;    // block number in the BUNDLE
    uint32_t kind;  // PullKind | height << 8
    uint64_t base;
};

struct PullFrontier {
    PullItem* items;
    uint32_t cap;
    // N words, zero at the start: the role (PullKind | height << 8 | 1 << 31) a block was first emitted in.  A block of a
    // tree that is needed WHOLE (headers, TxMeta, message AMTs, events AMTs) links to the same children wherever it is
    // reached from, so a second emission in the same role adds nothing to the closure and is dropped: identical receipts
    // share one events root, and the bundle keeps one copy of it — without this the events-root level holds one item per
    // RECEIPT while the frontier is sized by the bundle's BLOCKS.  Receipts-AMT nodes carry a base and are never dropped.
    uint32_t* role;
};

struct PullCtl {
    uint32_t n_cur;     // items of the frontier being read
    uint32_t n_next;    // items of the frontier being written
    uint32_t n_copy;    // blocks claimed this round
    uint32_t n_pulled;  // blocks claimed so far
    uint32_t overflow;  // bit 0: a frontier was full, bit 1: the staging arena / the pulled list was
    unsigned long long stage_used;
    unsigned long long payload;  // bytes of the claimed blocks themselves (Σ len)
    unsigned long long lo, hi, n_receipts;
    uint32_t have_range;  // the receipts root was found and decoded: lo / hi / n_receipts are set
    uint32_t pad;
    // what the walk takes: hi, or — the LAST shard — everything from lo on.  The root's count is checked by neither
    // `Amt::load` nor `get` (a root that says 572 over 700 receipts answers get(650)), and a claim beyond the count has an
    // owner, the last rank: so the last rank holds whatever the tree has behind its lo, not what the count promises
    // (tools/gpu_fuzz_seeds.sh seed 1010: one flipped bit of the count).
    unsigned long long hi_walk;
};

struct PullSeeds {
    CidKey child;
    CidKey parents[IPCFP_MAX_PARENTS];
    uint32_t n_parents;
    const CidKey* parents_wide;  // n_parents > IPCFP_MAX_PARENTS: ALL the keys, in HBM (else null)
};

struct PullTables {
    const uint32_t* len;      // bundle tables (N entries), resident in HBM
    const uint64_t* goff;     // offset of block i in the host buffer
    uint32_t* resident;       // bitmap over N: claimed
    uint64_t* stage_off;      // N: where block i lies in the staging arena (valid once claimed)
    uint32_t* pulled;         // the claimed blocks in claim order
    uint32_t pulled_cap;
    uint8_t* stage;
    uint64_t stage_cap;
    uint64_t* copy_src;       // this round's copy list (at most one entry per frontier item)
    uint64_t* copy_dst;
    uint32_t* copy_len;
};

}  // namespace ipcfp
