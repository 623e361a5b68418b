// csrc/kernels/hamt_table.hip — the HAMT node table (hamt_table.h): k_hamt_node_table parses every block of the witness
// as a HAMT node, a GROUP OF EIGHT LANES per block, in arena order (what K1 and the event parse read: neighbouring
// groups, neighbouring blocks).
//
// The group drives ONE reader in lockstep and streams its block through a 1 KB ring in LDS (cbor_dev.h IPCFP_RD_RING):
// every step the eight lanes fetch the next 256 bytes with one coalesced load each, and the step after is already in
// flight while they parse.  A sequential parse of a 4-5 KB state-tree node costs one memory latency per 256 bytes,
// mostly hidden — one lane with a 16-byte window pays one per 16 bytes, and a pass of one lane per block over the
// 0.33 GB witness of configs 4/5 took 2.7 ms (123 GB/s; profiles/r03_experiments.md).  Eight lanes per block also means
// eight times the wavefronts to hide what latency is left.
#define IPCFP_RD_LDS 1
#define IPCFP_RD_RING 8
#include <hip/hip_runtime.h>

#include "../common.h"
#include "cbor_dev.h"
#include "hamt_table.h"
#include "launch.h"

namespace ipcfp {

constexpr uint32_t kGroup = IPCFP_RD_RING, kGroupsPerWave = 64 / kGroup, kRingChunks = 64;

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

__global__ __launch_bounds__(64, 4) void k_hamt_node_table(const uint8_t* __restrict__ arena, const K1Meta* __restrict__ meta,
                                                           uint32_t n, uint32_t kinds, HamtNodeRec* __restrict__ recs) {
    __shared__ rd_chunk_t rings[kGroupsPerWave][kRingChunks];
    const uint32_t grp = (threadIdx.x & 63u) / kGroup, sub = threadIdx.x & (kGroup - 1u);
    const uint32_t s = blockIdx.x * kGroupsPerWave + grp;
    if (s >= n) return;
    const K1Meta m = meta[s];
    HamtNodeRec* out = recs + m.id;
    Rd r;
    r.init_ring(arena + m.off, m.len, (IPCFP_RD_AS uint8_t*)rings[grp]);
    uint32_t status = 0, kinds_ok = kinds | HK_ANY, std_links = 0, np32 = 0;
    uint64_t bf = 0;
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
            const uint32_t at = r.pos;
            fits = fits && at <= 0xffffu;
            if (sub == 0) out->ptr_off[p] = uint16_t(at);
            const uint32_t b0 = r.peek();
            if ((b0 >> 5) == 6) {
                uint32_t o, l;
                r.read_link(o, l);
                // the standard form: d8 2a | 58 27 | 00 | 01 71 a0 e4 02 20 | digest[32]
                if (r.ok() && l == 38 && o == at + 5 && r.peek64(at) == 0xa071010027582ad8ull &&
                    (r.peek64(at + 8) & 0xffffffull) == 0x2002e4ull)
                    std_links |= 1u << p;
            } else if ((b0 >> 5) == 4) {
                const uint64_t nkv = r.read_array();
                for (uint64_t k = 0; k < nkv && r.ok(); ++k) {
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
    if (sub == 0) {
        out->status = uint8_t(status);
        out->kinds_ok = uint8_t(status ? kinds_ok : 0u);
        out->np = uint8_t(np32);
        out->pad = 0;
        out->std_links = std_links;
        out->bitfield = bf;
    }
}

int launch_hamt_node_table(ipcfp_ctx* ctx, const uint8_t* arena, const void* meta_d, uint32_t n, uint32_t kinds, void* recs_d) {
    if (n == 0) return IPCFP_OK;
    hipLaunchKernelGGL(k_hamt_node_table, dim3(div_up(n, kGroupsPerWave)), dim3(64), 0, ctx->stream, arena,
                       static_cast<const K1Meta*>(meta_d), n, kinds, static_cast<HamtNodeRec*>(recs_d));
    IPCFP_HIP(ctx, hipGetLastError());
    return IPCFP_OK;
}

}  // namespace ipcfp
