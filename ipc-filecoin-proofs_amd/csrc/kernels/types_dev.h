// csrc/kernels/types_dev.h — typed DAG-CBOR decodes of the chain objects on the hot path
// (device side; same shapes as oracle/types.cpp).
#pragma once
#include "header_dev.h"
#include "walk_dev.h"

namespace ipcfp {

constexpr uint32_t kMaxParents = IPCFP_MAX_PARENTS;  // tipset keys are small (mainnet: ≤ ~10 blocks); engine limit, see DESIGN.md

// Wave-cooperative staging of one block into LDS: block headers are ≈1 KB and a single lane that
// parses one from HBM pays a DRAM round trip per window word (≈190 µs for two headers on an otherwise
// idle chip).  All 64 lanes copy the block with 16-byte loads, then ONE lane parses it out of LDS.
// Every lane of the wavefront must call this.  Returns a reader over the LDS copy, or over HBM when
// the block does not fit.
__device__ __forceinline__ Rd open_block_staged(const WitnessView& w, uint32_t b, uint8_t* lds, uint32_t cap) {
    const uint32_t len = w.len[b];
    const uint8_t* src = w.arena + w.off[b];  // 128-byte aligned, padded to a line
    Rd r;
    if (len + 48u <= cap) {
        const uint32_t words = ((len + 15u) >> 4) + 2u;  // whole 16-byte chunks, plus two for window over-reads
        const uint4* s4 = reinterpret_cast<const uint4*>(src);
        uint4* d4 = reinterpret_cast<uint4*>(lds);
        for (uint32_t i = threadIdx.x & 63u; i < words; i += 64u) d4[i] = s4[i];
        __syncthreads();
        r.init(lds, len);
        r.stage = false;  // already in LDS
    } else {
        r.init(src, len);
    }
    return r;
}

// `bs.get(cid)?.ok_or(..)?` + from_slice::<HeaderLite>
__device__ __forceinline__ uint32_t load_header(const WitnessView& w, const CidKey& cid, HeaderLite& h, uint32_t& block) {
    block = witness_find(w, cid);
    if (block == kNoBlock) return IPCFP_ST_ERR_MISSING_BLOCK;
    Rd r = open_block(w, block);
    return decode_header(r, h);
}

// Address::new_id(id).to_bytes() = 00 ‖ uvarint(id)  (≤ 11 bytes)
__device__ __forceinline__ uint32_t id_address_bytes(uint64_t id, uint8_t out[12]) {
    uint32_t n = 0;
    out[n++] = 0;
    do {
        uint8_t c = id & 0x7f;
        id >>= 7;
        if (id) c |= 0x80;
        out[n++] = c;
    } while (id);
    return n;
}

// get_actor_state (src/proofs/common/decode.rs:17-42) → the actor's `state` CID.
__device__ __forceinline__ uint32_t get_actor_state(const WitnessView& w, const CidKey& state_root, uint64_t actor_id,
                                                    CidKey& actor_state) {
    const uint32_t b = witness_find(w, state_root);
    if (b == kNoBlock) return IPCFP_ST_ERR_MISSING_BLOCK;  // decode.rs:23-25
    Rd r = open_block(w, b);
    r.expect_array(3);  // StateRoot [version, actors, info]   decode.rs:26
    if (r.read_uint() > 5) r.fail();
    CidKey actors, info;
    r.read_link_key(actors);
    r.read_link_key(info);
    r.finish();
    if (!r.ok()) return IPCFP_ST_ERR_DECODE;
    uint8_t key[12];
    const uint32_t kl = id_address_bytes(actor_id, key);  // decode.rs:34
    ValueLoc loc;
    const uint32_t st = hamt_get(w, actors, 5, VK_ACTOR_STATE, key, kl, loc);  // decode.rs:29-37
    if (st == IPCFP_ST_NOT_FOUND) return IPCFP_ST_ERR_ACTOR_NOT_FOUND;          // decode.rs:39
    if (st != IPCFP_ST_TRUE) return st;
    Rd v;
    v.init(w.arena + w.off[loc.block] + loc.off, loc.len);
    CidKey code;
    v.expect_array(5);
    v.read_link_key(code);
    v.read_link_key(actor_state);
    return v.ok() ? IPCFP_ST_TRUE : IPCFP_ST_ERR_DECODE;
}

// one attempt of parse_evm_state: `fields`-tuple [bytecode cid, bytecode_hash bytes(32), contract_state cid,
// (reserved?), nonce u64, tombstone?]
__device__ __forceinline__ bool try_evm_state(const WitnessView& w, uint32_t block, int fields, CidKey& contract_state) {
    Rd r = open_block(w, block);
    CidKey bytecode;
    r.expect_array(uint64_t(fields));
    r.read_link_key(bytecode);
    uint32_t o, l;
    r.read_bytes(o, l);
    if (r.ok() && l != 32) r.fail();
    r.read_link_key(contract_state);
    if (fields == 6) {
        if (r.at_null()) r.read_null();
        else r.skip();
    }
    (void)r.read_uint();
    if (r.at_null()) r.read_null();
    else r.skip();
    r.finish();
    return r.ok();
}

// parse_evm_state (src/proofs/common/decode.rs:79-97): V6 first, then V5
__device__ __forceinline__ uint32_t parse_evm_state(const WitnessView& w, uint32_t block, CidKey& contract_state) {
    if (try_evm_state(w, block, 6, contract_state)) return IPCFP_ST_TRUE;
    if (try_evm_state(w, block, 5, contract_state)) return IPCFP_ST_TRUE;
    return IPCFP_ST_ERR_DECODE;
}

}  // namespace ipcfp
