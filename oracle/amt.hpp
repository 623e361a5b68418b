// oracle/amt.hpp — TEST INFRASTRUCTURE.  AMT v0 / v3 reader.
//
// Restates fvm_ipld_amt 0.7.4 (`Amtv0`, `Amt`; crate NOT under /root/reference,
// Cargo.toml:19) as the reference calls it:
//   Amtv0::<Cid,_>::load + for_each        src/proofs/events/utils.rs:76-90
//   Amtv0::<Receipt,_>::load + get         src/proofs/events/verifier.rs:220-226
//   Amt::<StampedEvent,_>::load + get      src/proofs/events/verifier.rs:234-239
//   Amt::<StampedEvent,_>::load + for_each src/proofs/events/generator.rs:215-233,259-297
// Wire format:
//   v3 root = [bit_width, height, count, node]; v0 root = [height, count, node], bit_width 3
//   node    = [bmap: bytes(ceil(2^bw/8)), links: [cid…], values: [V…]], bit i ⇔ bmap[i/8] & (1<<(i%8));
//             links/values compacted (one per set bit, ascending); a node has links XOR values.
// Decode errors (`CollapsedNode::expand`): wrong bmap length; links and values both non-empty;
// number of links/values != popcount(bmap).  Every value of a decoded leaf is type-checked
// by the caller's `check_value` (serde decodes the whole `Vec<V>`), so a malformed sibling
// value is an Err even when another index was asked for.
#pragma once
#include <functional>

#include "store.hpp"

namespace orc {

constexpr uint32_t kAmtMaxBitWidth = 8;  // engine limit (documented in DESIGN.md); FVM uses 3 and 5

// A located value: the bytes of one CBOR item inside a block.
struct ValueLoc {
    const Bytes* block = nullptr;
    size_t off = 0, len = 0;
};

// Type-checks (fully decodes) one value at the reader's position and advances past it.
using ValueChecker = std::function<void(Reader&)>;

struct AmtRoot {
    uint32_t bit_width = 3;
    uint64_t height = 0, count = 0;
    const Bytes* block = nullptr;  // the root block
    size_t node_off = 0;           // offset of the inline root node inside it
};

// Amt::load / Amtv0::load — reads only the root block.
AmtRoot amt_load(const Blockstore& bs, const Cid& root, int version /*0|3*/, const ValueChecker& check);
// Amt::get — true ⇒ Some(value at `loc`), false ⇒ None.  Throws Err.
bool amt_get(const Blockstore& bs, const AmtRoot& root, uint64_t index, const ValueChecker& check, ValueLoc& loc);
// Amt::for_each — ascending index order.
void amt_for_each(const Blockstore& bs, const AmtRoot& root, const ValueChecker& check,
                  const std::function<void(uint64_t, const ValueLoc&)>& f);


// Amt::for_each collected into a vector, in ascending index order: (index, value).  Baseline variant B2
// "all cores" (BASELINE.md §2): the subtrees below the top levels are walked by `threads` OpenMP threads
// (0 = every processor).  The outcome is the sequential for_each's: on failure the Err thrown is the one
// the depth-first traversal meets first.
struct AmtItem {
    uint64_t index;
    ValueLoc value;
};
void amt_collect(const Blockstore& bs, const AmtRoot& root, const ValueChecker& check, std::vector<AmtItem>& out,
                 int threads);

// omp_set_num_threads(threads), 0 = omp_get_num_procs(); returns the count in effect
int use_threads(int threads);

}  // namespace orc
