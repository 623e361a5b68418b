// csrc/kernels/hamt_table_lane.hip — the HAMT node table (hamt_table.h), one block per LANE in arena order, the reader
// staging each 128-byte line it touches in the lane's LDS slot (cbor_dev.h IPCFP_LINE_STAGE) — the form that won for the
// block-order event parse (block_events.hip), against the eight-lanes-per-block ring reader of hamt_table.hip.
#define IPCFP_LINE_STAGE 1
#include <hip/hip_runtime.h>

#include "../common.h"
#include "hamt_table_body.h"
#include "launch.h"
#include "witness_dev.h"

namespace ipcfp {

__global__ __launch_bounds__(256, 4) void k_hamt_node_table_lane(const uint8_t* __restrict__ arena, const K1Meta* __restrict__ meta,
                                                                 uint32_t n, uint32_t kinds, HamtNodeRec* __restrict__ recs) {
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    bool live = s < n;
    K1Meta m{0, 0, 0};
    if (live) m = meta[s];
    live = live && m.len < kHamtOutlineMinLen;  // (the longer blocks are the 32-lane outline's: launch_hamt_node_table)
    HamtNodeRec* out = recs + m.id;
    Rd r;
    r.init(arena + m.off, live ? m.len : 0u);
    uint32_t status, kinds_ok, std_links, np32;
    uint64_t bf;
    hamt_node_parse(r, kinds, live, out, status, kinds_ok, std_links, np32, bf);
    if (live) {
        out->status = uint8_t(status);
        out->kinds_ok = uint8_t(status ? kinds_ok : 0u);
        out->np = uint8_t(np32);
        out->pad = 0;
        out->std_links = std_links;
        out->bitfield = bf;
    }
}

// the blocks the outline takes, as a work list (schedule order is by length class, longest first: they are a prefix, found
// by every lane for itself; one counter update per wavefront)
__global__ __launch_bounds__(256) void k_hamt_list_long(const K1Meta* __restrict__ meta, uint32_t n, uint32_t* __restrict__ work,
                                                        uint32_t* __restrict__ count) {
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    const bool mine = s < n && meta[s].len >= kHamtOutlineMinLen;
    const uint64_t votes = __ballot(mine);
    if (!votes) return;
    const uint32_t lane = threadIdx.x & 63u, leader = uint32_t(__ffsll((long long)votes)) - 1u;
    uint32_t base = 0;
    if (lane == leader) base = atomicAdd(count, uint32_t(__popcll(votes)));
    base = __shfl(base, leader, 64);
    if (mine) work[base + uint32_t(__popcll(votes & ((1ull << lane) - 1ull)))] = meta[s].id;
}

// the long blocks the outline left untabulated (status 0: not a state-tree node in its spellings — e.g. a storage node with
// one very full bucket chain), one lane each
__global__ __launch_bounds__(256, 4) void k_hamt_node_table_rest(WitnessView w, const uint32_t* __restrict__ work,
                                                                 const uint32_t* __restrict__ count, uint32_t kinds,
                                                                 HamtNodeRec* __restrict__ recs) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    bool live = i < *count;
    const uint32_t id = live ? work[i] : 0u;
    live = live && recs[id].status == 0;
    HamtNodeRec* out = recs + id;
    Rd r;
    r.init(w.arena + (live ? w.off[id] : 0ull), live ? w.len[id] : 0u);
    uint32_t status, kinds_ok, std_links, np32;
    uint64_t bf;
    hamt_node_parse(r, kinds, live, out, status, kinds_ok, std_links, np32, bf);
    if (live) {
        out->status = uint8_t(status);
        out->kinds_ok = uint8_t(status ? kinds_ok : 0u);
        out->np = uint8_t(np32);
        out->pad = 0;
        out->std_links = std_links;
        out->bitfield = bf;
    }
}

int launch_hamt_node_table_rest(ipcfp_ctx* ctx, hipStream_t stream, const WitnessView& w, const uint32_t* work_d, const uint32_t* count_d,
                                uint32_t bound, uint32_t kinds, void* recs_d) {
    if (bound == 0) return IPCFP_OK;
    hipLaunchKernelGGL(k_hamt_node_table_rest, dim3(div_up(bound, 256)), dim3(256), 0, stream, w, work_d, count_d, kinds,
                       static_cast<HamtNodeRec*>(recs_d));
    IPCFP_HIP(ctx, hipGetLastError());
    return IPCFP_OK;
}

int launch_hamt_list_long(ipcfp_ctx* ctx, const void* meta_d, uint32_t n, uint32_t* work_d, uint32_t* count_d) {
    if (n == 0) return IPCFP_OK;
    hipLaunchKernelGGL(k_hamt_list_long, dim3(div_up(n, 256)), dim3(256), 0, ctx->stream, static_cast<const K1Meta*>(meta_d), n, work_d, count_d);
    IPCFP_HIP(ctx, hipGetLastError());
    return IPCFP_OK;
}

int launch_hamt_node_table_lane(ipcfp_ctx* ctx, const uint8_t* arena, const void* meta_d, uint32_t n, uint32_t kinds, void* recs_d) {
    if (n == 0) return IPCFP_OK;
    hipLaunchKernelGGL(k_hamt_node_table_lane, dim3(div_up(n, 256)), dim3(256), 0, ctx->stream, arena, static_cast<const K1Meta*>(meta_d), n,
                       kinds, static_cast<HamtNodeRec*>(recs_d));
    IPCFP_HIP(ctx, hipGetLastError());
    return IPCFP_OK;
}

}  // namespace ipcfp
