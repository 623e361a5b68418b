// csrc/kernels/hamt_table_body.h — one block parsed as a HAMT node (hamt_table.h): the part the two forms of
// k_hamt_node_table share.  The including unit configures the reader (IPCFP_RD_RING / IPCFP_LINE_STAGE).
#pragma once
#include "cbor_dev.h"
#include "hamt_table.h"

namespace ipcfp {

// One bucket value at r.pos: which typed decodes does it pass, where does it end?  The typed checks run on the reader
// itself (a failed attempt rewinds: a value is far shorter than the ring's reach).  false: not even a well-formed item.
__device__ __forceinline__ bool value_kinds(Rd& r, uint32_t want, uint32_t& ok_kinds) {
    const uint32_t vstart = r.pos;
    ok_kinds = 0;
    if (want & HK_ACTOR_STATE) {
        check_actor_state(r);
        if (r.ok()) {
            ok_kinds = HK_ACTOR_STATE | HK_ANY;  // (an ActorState is no Vec<u8>: array(5) of a link …)
            return true;
        }
        if (r.err == kRdRingLost) return false;
        r.err = 0;
        r.pos = vstart;
    }
    if (want & HK_VEC_U8) {
        check_vec_u8(r);
        if (r.ok()) {
            ok_kinds = HK_VEC_U8 | HK_ANY;
            return true;
        }
        if (r.err == kRdRingLost) return false;
        r.err = 0;
        r.pos = vstart;
    }
    r.skip();
    ok_kinds = HK_ANY;
    return r.ok();
}

// `[bitfield bytes(≤ 8), [≤ 32 pointers]]` at r, every pointer a well-formed link or a bucket of `[key bytes, value]` pairs:
// → the record's fields; `writer`: this lane writes out->ptr_off (the ring form runs eight lanes in lockstep on one block).
__device__ __forceinline__ void hamt_node_parse(Rd& r, uint32_t kinds, bool writer, HamtNodeRec* __restrict__ out, uint32_t& status,
                                                uint32_t& kinds_ok, uint32_t& std_links, uint32_t& np32, uint64_t& bf) {
    status = 0;
    kinds_ok = (kinds & (HK_ACTOR_STATE | HK_VEC_U8)) | HK_ANY;
    std_links = 0;
    np32 = 0;
    bf = 0;
    do {
        r.expect_array(2);
        uint32_t bo, bl;
        r.read_bytes(bo, bl);
        if (!r.ok() || bl > 8) break;
        for (uint32_t k = 0; k < bl; ++k) bf |= uint64_t(r.at(bo + bl - 1 - k)) << (8u * k);  // big-endian, last byte = bits 0..7
        const uint64_t np = r.read_array();
        if (!r.ok() || np > kHamtTablePointers) break;
        np32 = uint32_t(np);
        bool fits = true;
        for (uint32_t p = 0; p < np32 && r.ok(); ++p) {
            r.ensure_span(104);  // (a window-staged reader: the link, or the bucket's head and first entry, in one refill for all lanes)
            const uint32_t at = r.pos;
            fits = fits && at <= 0xffffu;
            if (writer) out->ptr_off[p] = uint16_t(at);
            const uint32_t b0 = r.peek();
            if ((b0 >> 5) == 6) {
                uint32_t o, l;
                r.read_link(o, l);
                // the standard form: d8 2a | 58 27 | 00 | 01 71 a0 e4 02 20 | digest[32]
                if (r.ok() && l == 38 && o == at + 5 && r.peek64(at) == 0xa071010027582ad8ull &&
                    (r.peek64(at + 8) & 0xffffffull) == 0x2002e4ull)
                    std_links |= 1u << p;
            } else if ((b0 >> 5) == 4) {
                uint64_t nkv;
                if (b0 < 0x98u) {  // (the head of a short array is its one byte)
                    nkv = b0 - 0x80u;
                    r.pos += 1u;
                } else {
                    nkv = r.read_array();
                }
                for (uint64_t k = 0; k < nkv && r.ok(); ++k) {
                    if (k) r.ensure_span(104);
                    // A storage entry as every encoder writes it — `82 58 20 <32-byte slot>` and a Vec<u8> of one- and two-byte
                    // elements — from two fetches and one more per four elements.  Item by item it is four heads and an element
                    // loop, ≈ 2.7 k instructions, and one lane does that for the ≈ 8 entries of its node: 0.6 ms of configs[4]'s
                    // call for k_hamt_node_table_lane (profiles/r06_experiments.md).  Anything else takes that way as before.
                    if ((kinds & (HK_VEC_U8 | HK_ITEM_BY_ITEM)) == HK_VEC_U8) {
                        const uint32_t e0 = r.pos;
                        if (e0 + 36u <= r.n && (r.peek64(e0) & 0xffffffull) == 0x205882ull) {
                            const uint32_t end = vec_u8_end(r, e0 + 35u);
                            if (end) {
                                r.pos = end;
                                kinds_ok &= HK_VEC_U8 | HK_ANY;
                                continue;
                            }
                        }
                    }
                    r.expect_array(2);
                    uint32_t ko, kl;
                    r.read_bytes(ko, kl);
                    if (!r.ok()) break;
                    uint32_t vk;
                    if (!value_kinds(r, kinds, vk)) {
                        if (r.ok()) r.fail();
                        break;
                    }
                    kinds_ok &= vk;
                }
            } else {
                r.fail();
            }
        }
        r.finish();
        if (r.ok() && fits) status = 1;
    } while (false);
}

}  // namespace ipcfp
