// csrc/kernels/storage_dev.h — read_storage_slot (src/proofs/storage/decode.rs:36-97) and
// left_pad_32 (src/proofs/common/evm.rs:91-100) on the device: the six-way layout sniff of the
// contract-state root, tried in the reference's order A1, A2, A3, B1, B2, C.
#pragma once
#include "types_dev.h"

namespace ipcfp {

__device__ __forceinline__ bool text_is(Rd& r, uint32_t off, uint32_t len, const char* s, uint32_t n) {
    if (len != n) return false;
    for (uint32_t i = 0; i < n; ++i)
        if (r.at(off + i) != uint8_t(s[i])) return false;
    return true;
}

struct SlotHit {
    bool found;
    uint32_t off, len;  // byte-string value inside the block
};

// SmallMap { v: [[key bytes, value bytes]…] } (storage/decode.rs:9-13): a CBOR map with the required
// field "v"; unknown fields skipped; a second "v" is a decode error.  When `search`, the first pair
// whose key equals the slot is recorded.
__device__ __forceinline__ void read_small_map(Rd& r, const uint8_t* slot, bool search, SlotHit& hit) {
    const uint64_t n = r.read_map();
    bool have_v = false;
    for (uint64_t i = 0; i < n && r.ok(); ++i) {
        uint32_t ko, kl;
        r.read_text(ko, kl);
        if (!r.ok()) return;
        if (text_is(r, ko, kl, "v", 1)) {
            if (have_v) return r.fail();
            have_v = true;
            const uint64_t np = r.read_array();
            for (uint64_t j = 0; j < np && r.ok(); ++j) {
                uint32_t ao, al, bo, bl;
                r.expect_array(2);
                r.read_bytes(ao, al);
                r.read_bytes(bo, bl);
                if (r.ok() && search && !hit.found && al == 32) {
                    bool eq = true;
                    for (int c = 0; c < 32; ++c) eq &= r.at(ao + c) == slot[c];
                    if (eq) {
                        hit.found = true;
                        hit.off = bo;
                        hit.len = bl;
                    }
                }
            }
        } else {
            r.skip();
        }
    }
    if (r.ok() && !have_v) r.fail();
}

// left_pad_32 of a byte range (len ≥ 32 keeps the LAST 32 bytes)
__device__ __forceinline__ void left_pad_32_bytes(Rd& r, uint32_t off, uint32_t len, uint8_t out[32]) {
    for (int i = 0; i < 32; ++i) out[i] = 0;
    if (len >= 32) {
        for (int i = 0; i < 32; ++i) out[i] = uint8_t(r.at(off + len - 32 + i));
    } else {
        for (uint32_t i = 0; i < len; ++i) out[32 - len + i] = uint8_t(r.at(off + i));
    }
}

// left_pad_32 of a serde Vec<u8> (CBOR array of u8) already type-checked by hamt_get
__device__ __forceinline__ void left_pad_32_vec(const WitnessView& w, const ValueLoc& loc, uint8_t out[32]) {
    Rd v;
    v.init(w.arena + w.off[loc.block] + loc.off, loc.len);
    const uint64_t n = v.read_array();
    for (int i = 0; i < 32; ++i) out[i] = 0;
    for (uint64_t i = 0; i < n && v.ok(); ++i) {
        const uint8_t x = uint8_t(v.read_uint());
        if (n >= 32) {
            if (i >= n - 32) out[i - (n - 32)] = x;
        } else {
            out[32 - n + i] = x;
        }
    }
}

// The layout sniff of read_storage_slot alone (storage/decode.rs:46-96): WHICH of the six decodes succeeds is a property
// of the root block, not of the slot.  → 0 = A1, 1 = A2, 2 = A3 (inline maps), 3 = a HAMT (root, bw: B1 / B2 / C),
// 4 = the root block is not in the witness.  Same attempts, same order as read_storage_slot_padded below.
__device__ __forceinline__ uint32_t sniff_storage_root(const WitnessView& w, const CidKey& root, CidKey& hamt_root, uint32_t& bw) {
    hamt_root = root;
    bw = 5;
    const uint32_t b = witness_find(w, root);
    if (b == kNoBlock) return 4;
    const uint8_t none[32] = {0};
    SlotHit hit{false, 0, 0};
    {
        Rd r = open_block(w, b);
        uint32_t o, l;
        r.expect_array(2);
        r.read_bytes(o, l);
        const uint64_t n = r.read_array();
        for (uint64_t i = 0; i < n && r.ok(); ++i) read_small_map(r, none, false, hit);
        r.finish();
        if (r.ok() && n > 0) return 0;
    }
    {
        Rd r = open_block(w, b);
        uint32_t o, l;
        r.expect_array(2);
        r.read_bytes(o, l);
        read_small_map(r, none, false, hit);
        r.finish();
        if (r.ok()) return 1;
    }
    {
        Rd r = open_block(w, b);
        read_small_map(r, none, false, hit);
        r.finish();
        if (r.ok()) return 2;
    }
    {
        Rd r = open_block(w, b);
        CidKey inner;
        r.expect_array(2);
        r.read_link_key(inner);
        const uint64_t v = r.read_uint();
        r.finish();
        if (r.ok()) {
            hamt_root = inner;
            bw = uint32_t(v);  // `bw as u32`
            return 3;
        }
    }
    {
        Rd r = open_block(w, b);
        CidKey inner;
        uint64_t v = 0;
        bool have_root = false, have_bw = false;
        const uint64_t n = r.read_map();
        for (uint64_t i = 0; i < n && r.ok(); ++i) {
            uint32_t ko, kl;
            r.read_text(ko, kl);
            if (!r.ok()) break;
            if (text_is(r, ko, kl, "root", 4)) {
                if (have_root) r.fail();
                have_root = true;
                r.read_link_key(inner);
            } else if (text_is(r, ko, kl, "bitwidth", 8)) {
                if (have_bw) r.fail();
                have_bw = true;
                v = r.read_uint();
            } else {
                r.skip();
            }
        }
        r.finish();
        if (r.ok() && have_root && have_bw) {
            hamt_root = inner;
            bw = uint32_t(v);
        }
    }
    return 3;
}

// read_storage_slot + left_pad_32.  TRUE ⇒ padded holds the 32-byte value (zero when absent).
__device__ __forceinline__ uint32_t read_storage_slot_padded(const WitnessView& w, const CidKey& root,
                                                             const uint8_t slot[32], uint8_t padded[32]) {
    for (int i = 0; i < 32; ++i) padded[i] = 0;
    const uint32_t b = witness_find(w, root);
    if (b == kNoBlock) return IPCFP_ST_ERR_MISSING_BLOCK;  // decode.rs:41-43
    // A1) InlineTupleList(bytes, Vec<SmallMap>)  decode.rs:46-55 — only the FIRST map is searched
    {
        Rd r = open_block(w, b);
        SlotHit hit{false, 0, 0};
        uint32_t o, l;
        r.expect_array(2);
        r.read_bytes(o, l);
        const uint64_t n = r.read_array();
        for (uint64_t i = 0; i < n && r.ok(); ++i) read_small_map(r, slot, i == 0, hit);
        r.finish();
        if (r.ok() && n > 0) {
            if (hit.found) left_pad_32_bytes(r, hit.off, hit.len, padded);
            return IPCFP_ST_TRUE;
        }
    }
    // A2) InlineTuple(bytes, SmallMap)  decode.rs:58-65
    {
        Rd r = open_block(w, b);
        SlotHit hit{false, 0, 0};
        uint32_t o, l;
        r.expect_array(2);
        r.read_bytes(o, l);
        read_small_map(r, slot, true, hit);
        r.finish();
        if (r.ok()) {
            if (hit.found) left_pad_32_bytes(r, hit.off, hit.len, padded);
            return IPCFP_ST_TRUE;
        }
    }
    // A3) SmallMap  decode.rs:68-75
    {
        Rd r = open_block(w, b);
        SlotHit hit{false, 0, 0};
        read_small_map(r, slot, true, hit);
        r.finish();
        if (r.ok()) {
            if (hit.found) left_pad_32_bytes(r, hit.off, hit.len, padded);
            return IPCFP_ST_TRUE;
        }
    }
    CidKey hamt_root = root;
    uint32_t bw = 5;
    bool wrapped = false;
    // B1) MapTuple(Cid, u64)  decode.rs:78-82
    {
        Rd r = open_block(w, b);
        CidKey inner;
        r.expect_array(2);
        r.read_link_key(inner);
        const uint64_t v = r.read_uint();
        r.finish();
        if (r.ok()) {
            wrapped = true;
            hamt_root = inner;
            bw = uint32_t(v);  // `bw as u32`
        }
    }
    // B2) MapStruct { root, bitwidth, .. }  decode.rs:85-89
    if (!wrapped) {
        Rd r = open_block(w, b);
        CidKey inner;
        uint64_t v = 0;
        bool have_root = false, have_bw = false;
        const uint64_t n = r.read_map();
        for (uint64_t i = 0; i < n && r.ok(); ++i) {
            uint32_t ko, kl;
            r.read_text(ko, kl);
            if (!r.ok()) break;
            if (text_is(r, ko, kl, "root", 4)) {
                if (have_root) r.fail();
                have_root = true;
                r.read_link_key(inner);
            } else if (text_is(r, ko, kl, "bitwidth", 8)) {
                if (have_bw) r.fail();
                have_bw = true;
                v = r.read_uint();
            } else {
                r.skip();
            }
        }
        r.finish();
        if (r.ok() && have_root && have_bw) {
            wrapped = true;
            hamt_root = inner;
            bw = uint32_t(v);
        }
    }
    // B1/B2 → wrapped HAMT; C) direct HAMT at this CID, bit width 5  decode.rs:92-96
    ValueLoc loc;
    const uint32_t st = hamt_get(w, hamt_root, bw, VK_VEC_U8, slot, 32, loc);
    if (st == IPCFP_ST_NOT_FOUND) return IPCFP_ST_TRUE;  // unwrap_or_default(): missing key means zero
    if (st != IPCFP_ST_TRUE) return st;
    left_pad_32_vec(w, loc, padded);
    return IPCFP_ST_TRUE;
}

}  // namespace ipcfp
