// csrc/kernels/hamt_table_lane.hip — the HAMT node table (hamt_table.h), one block per LANE in arena order, the reader
// staging the 128 bytes it is about to read in the lane's LDS slot (cbor_dev.h IPCFP_LINE_STAGE) — the form that won for the
// block-order event parse (block_events.hip), against the eight-lanes-per-block ring reader of hamt_table.hip.
#ifndef IPCFP_LINE_STAGE
#define IPCFP_LINE_STAGE 2  // (a window the parse re-aims once per pointer and bucket entry: cbor_dev.h)
#endif
#include <hip/hip_runtime.h>

#include <algorithm>

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

// A level's SHORT nodes of a level-by-level walk (hamt_levels.hip: the work list of size class 0, entries {block, length,
// arena offset}), one lane per node with the same line-staged reader: the overflow nodes under a full bucket are ≈ 500
// bytes — four or five lines — and ≈ 30 k of them per level of configs[3]; two per wavefront with one lane reading out of
// a 1.5 KB stage was 85-110 µs for such a level (profiles/r05_experiments.md, r06_experiments.md).  The record is the
// levels path's: status 1 iff the node is well-formed AND every bucket value passes the typed check of `kind_bit`; no entry
// table (etab_of = 0: the advance searches the bucket with the reader), no resolved children (pad = 0).
__global__ __launch_bounds__(256, 4) void k_hamt_lv_parse_lane(const uint8_t* __restrict__ arena, const uint4* __restrict__ list,
                                                               const uint32_t* __restrict__ count, uint32_t cap, uint32_t kind_bit,
                                                               HamtNodeRec* __restrict__ recs, uint32_t* __restrict__ etab_of) {
    const uint32_t n_raw = *count, n = n_raw < cap ? n_raw : cap;
    for (uint32_t base = blockIdx.x * blockDim.x; base < n; base += gridDim.x * blockDim.x) {  // (uniform per workgroup)
        const uint32_t i = base + threadIdx.x;
        const bool live = i < n;
        const uint4 e = live ? list[i] : make_uint4(0, 0, 0, 0);
        HamtNodeRec* out = recs + e.x;
        Rd r;
        r.init(arena + (uint64_t(e.z) | (uint64_t(e.w) << 32)), live ? e.y : 0u);
        uint32_t status, kinds_ok, std_links, np32;
        uint64_t bf;
        hamt_node_parse(r, kind_bit, live, out, status, kinds_ok, std_links, np32, bf);
        if (live) {
            const uint32_t ok = status && (kinds_ok & kind_bit) ? 1u : 0u;
            out->status = uint8_t(ok);
            out->kinds_ok = uint8_t(ok);
            out->np = uint8_t(ok ? np32 : 0u);
            out->pad = 0;
            out->std_links = ok ? std_links : 0u;
            out->bitfield = ok ? bf : 0ull;
            if (etab_of) etab_of[e.x] = 0u;
        }
    }
}

int launch_hamt_lv_parse_lane(ipcfp_ctx* ctx, const WitnessView& w, const void* list_d, const uint32_t* count_d, uint32_t cap, uint32_t bound,
                              uint32_t kind_bit, void* recs_d, uint32_t* etab_of_d) {
    if (bound == 0) return IPCFP_OK;
    hipLaunchKernelGGL(k_hamt_lv_parse_lane, dim3(std::min(div_up(bound, 256), 4096u)), dim3(256), 0, ctx->stream, w.arena,
                       static_cast<const uint4*>(list_d), count_d, cap, kind_bit, static_cast<HamtNodeRec*>(recs_d), etab_of_d);
    IPCFP_HIP(ctx, hipGetLastError());
    return IPCFP_OK;
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
