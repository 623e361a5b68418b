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

#include <cstdlib>

#include "../common.h"
#include "cbor_dev.h"
#include "hamt_table_body.h"
#include "launch.h"

namespace ipcfp {

constexpr uint32_t kGroup = IPCFP_RD_RING, kGroupsPerWave = 64 / kGroup, kRingChunks = 64;

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
    uint32_t status, kinds_ok, std_links, np32;
    uint64_t bf;
    hamt_node_parse(r, kinds, sub == 0, out, status, kinds_ok, std_links, np32, bf);
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
