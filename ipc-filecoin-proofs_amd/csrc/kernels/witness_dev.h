// csrc/kernels/witness_dev.h — device-side view of the HBM-resident witness and the
// CID → block-id lookup every walk kernel uses instead of `Blockstore::get`
// (trait impls: src/proofs/common/blockstore.rs:26-39; MemoryBlockstore:
// src/proofs/events/verifier.rs:82-86).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace ipcfp {

constexpr uint32_t kNoBlock = 0xffffffffu;
// not an ipcfp_status_t: "not settled by the fast kernel — the general walker launched right behind takes it"
constexpr uint32_t kStPending = 0xfeu;

// a located value: the CBOR item of witness block `block` at [off, off + len)
struct ValueLoc {
    uint32_t block;  // witness block id
    uint32_t off;    // byte offset of the value's CBOR item inside the block
    uint32_t len;    // its encoded length
};

// (Raising the issue priority of the main-stream kernels with s_setprio, so that K1 and the block-order event parse
// beside them would not stretch their dependent steps, was measured and made every one of them SLOWER — AMT levels
// 15 -> 30 us, k_tipset_prepare 87 -> 200 us, with or without a neighbour: profiles/r02_experiments.md.)

struct WitnessView {
    const uint8_t* arena;    // every block on its own 128-byte line(s), + 256 B tail slack
    const uint64_t* off;     // n
    const uint32_t* len;     // n
    const uint8_t* cids;     // n × 40, zero padded
    const uint32_t* slots;   // open-addressing table of block ids (kNoBlock = empty)
    uint32_t mask;           // table size - 1 (power of two)
    uint32_t n;
    // K8 recording: when non-null, every lookup of a CID present in the witness sets
    // bit `block id` (RecordingBlockStore::get, src/proofs/common/blockstore.rs:26-30)
    uint32_t* touched;
};

// One block in SCHEDULE order = arena order (blocks sorted by 128-byte line count, each class contiguous): what K1
// and the block-order event parser read, coalesced.
struct K1Meta {
    uint64_t off;   // arena offset of the block
    uint32_t len;
    uint32_t id;    // block id (position in the caller's tables)
};

struct CidKey {
    uint64_t w[5];  // 40-byte slot as little-endian words
};

__device__ __forceinline__ uint32_t cid_hash(const CidKey& k) {
    uint64_t x = k.w[0] ^ ((k.w[1] << 13) | (k.w[1] >> 51)) ^ ((k.w[2] << 27) | (k.w[2] >> 37)) ^
                 ((k.w[3] << 41) | (k.w[3] >> 23)) ^ k.w[4];
    x *= 0x9E3779B97F4A7C15ULL;
    return uint32_t(x >> 32);
}

// the full 64-bit mix: high half selects the slot (cid_hash), low half is an independent fingerprint
__device__ __forceinline__ uint64_t cid_hash64(const CidKey& k) {
    uint64_t x = k.w[0] ^ ((k.w[1] << 13) | (k.w[1] >> 51)) ^ ((k.w[2] << 27) | (k.w[2] >> 37)) ^
                 ((k.w[3] << 41) | (k.w[3] >> 23)) ^ k.w[4];
    return x * 0x9E3779B97F4A7C15ULL;
}

__device__ __forceinline__ CidKey load_cid_slot(const uint8_t* cids, uint32_t i) {
    const uint64_t* p = reinterpret_cast<const uint64_t*>(cids + 40ull * i);
    CidKey k;
#pragma unroll
    for (int j = 0; j < 5; ++j) k.w[j] = p[j];
    return k;
}

__device__ __forceinline__ bool cid_equal(const CidKey& a, const CidKey& b) {
    return ((a.w[0] ^ b.w[0]) | (a.w[1] ^ b.w[1]) | (a.w[2] ^ b.w[2]) | (a.w[3] ^ b.w[3]) | (a.w[4] ^ b.w[4])) == 0;
}

// Build a key from `n` CID bytes at an arbitrary address (n ≤ 40; longer CIDs
// cannot be witness keys — ipcfp.h IPCFP_CID_SLOT).
__device__ __forceinline__ CidKey cid_key_from_bytes(const uint8_t* p, uint32_t n) {
    CidKey k;
#pragma unroll
    for (int j = 0; j < 5; ++j) {
        uint64_t v = 0;
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const uint32_t idx = 8u * j + b;
            if (idx < n) v |= uint64_t(p[idx]) << (8 * b);
        }
        k.w[j] = v;
    }
    return k;
}

// One key into the CID → block-id table (K4; kernels/cid_index.hip, and the fused index + prologue launch of
// tipset_prepare.hip).  Open addressing, linear probing; the probe IS the compare-and-swap (at load ≤ 0.5 most home
// slots are empty: one round trip to the table instead of a read followed by the CAS).  An entry never moves once
// placed; a duplicate CID raises the slot to the larger block id (last block wins).  Every atomic's result is
// consumed, so the lane has seen it performed before it goes on (the completion count of the fused launch relies on it).
template <bool CAS_FIRST>
__device__ __forceinline__ void index_insert_key(const uint8_t* __restrict__ cids, uint32_t* __restrict__ slots, uint32_t mask,
                                                 uint32_t i) {
    const CidKey key = load_cid_slot(cids, i);
    uint32_t s = cid_hash(key) & mask;
    for (;;) {
        uint32_t cur = CAS_FIRST ? kNoBlock : slots[s];
        if (cur == kNoBlock) {
            cur = atomicCAS(&slots[s], kNoBlock, i);
            if (cur == kNoBlock) return;  // claimed an empty slot
        }
        // slot owned by block `cur` (its CID identity never changes once claimed)
        if (cid_equal(load_cid_slot(cids, cur), key)) {
            const uint32_t old = atomicMax(&slots[s], i);  // duplicate CID: last block wins
            asm volatile("" ::"v"(old));
            return;
        }
        s = (s + 1) & mask;
    }
}

// Blockstore::get → block id, or kNoBlock when the CID is not in the witness.
__device__ __forceinline__ uint32_t witness_find(const WitnessView& w, const CidKey& key) {
    uint32_t s = cid_hash(key) & w.mask;
    for (;;) {
        const uint32_t b = w.slots[s];
        if (b == kNoBlock) return kNoBlock;
        if (cid_equal(load_cid_slot(w.cids, b), key)) {
            if (w.touched) atomicOr(&w.touched[b >> 5], 1u << (b & 31));
            return b;
        }
        s = (s + 1) & w.mask;
    }
}

}  // namespace ipcfp
