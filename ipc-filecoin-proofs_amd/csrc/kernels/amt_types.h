// csrc/kernels/amt_types.h — the plain records of the AMT enumerator (amt_enum.h), without the walk primitives: what a
// unit that only fills or reads them needs (the tipset prologue writes root specs, kernels read leaf tables).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "witness_dev.h"

namespace ipcfp {

struct AmtRootSpec {
    CidKey root;
    uint32_t version;  // 0 | 3
    uint32_t seq;      // error-ordering sequence number of this AMT (ascending in traversal order)
    uint32_t skip;     // 1 ⇒ do not load (an earlier stage already failed for it)
    uint32_t kind_p1;  // 0: the value type of the call; else value type + 1 — the EXTRA root of amt_enumerate (below),
                       // whose load failure is not an error of the call (it is then enumerated on its own)
};

// one frontier entry
struct EnumNode {
    uint32_t block;     // kNoBlock ⇒ dead entry (contributes nothing)
    uint32_t node_off;  // offset of the node inside the block
    uint64_t base;      // index of the node's first slot
    uint32_t seq;
    uint16_t height;    // node height (0 = leaf level)
    uint8_t bit_width;
    uint8_t leaf_ready; // a Leaf node met above height 0: carried down unchanged
};

// one frontier entry of the DENSE walk (amt_enum.hip): where the node's bytes lie, so that the level below starts
// reading them without first asking the off / len tables
struct DenseNode {
    uint64_t goff;    // arena offset of the node's first byte
    uint64_t base;    // index of the node's first slot
    uint32_t block;   // kNoBlock ⇒ dead entry
    uint32_t rem;     // bytes from the node's first byte to the end of its block
    uint32_t seq;
    uint32_t whole;   // 1: the node is the whole block (a child block); 0: the root node inside `[height, count, node]`
};

// one enumerated value, in for_each order
struct LeafRef {
    uint32_t block, off, len;
    uint32_t seq;
    uint64_t index;
};

constexpr uint64_t kNoEnumError = ~0ULL;
__host__ __device__ inline uint64_t pack_enum_error(uint32_t seq, uint64_t base, uint32_t code) {
    return (uint64_t(seq & 0xffffu) << 48) | ((base & 0xffffffffffULL) << 8) | (code & 0xffu);
}
__host__ __device__ inline uint32_t enum_error_code(uint64_t e) { return e == kNoEnumError ? 0u : uint32_t(e & 0xffu); }

}  // namespace ipcfp
