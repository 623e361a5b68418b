// csrc/kernels/walk.hip — K5 (amt_get) and K7 (hamt_get) batch kernels: one query per lane.
//
// K5 replaces `receipts_amt.get(exec_index)` / `events_amt.get(event_index)` / `r_amt.get(i)`
// (src/proofs/events/verifier.rs:224,237; src/proofs/events/generator.rs:249);
// K7 replaces `actors.get(&key)` / `hamt.get(slot)` (src/proofs/common/decode.rs:35-39,
// src/proofs/storage/decode.rs:81,88,96).
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "../common.h"
#include "launch.h"
#include "walk_dev.h"

namespace ipcfp {

__global__ __launch_bounds__(256) void k_amt_get(WitnessView w, CidKey root, int version, int vkind,
                                                 const uint64_t* __restrict__ index, uint32_t n,
                                                 uint8_t* __restrict__ status, ValueLoc* __restrict__ loc) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    AmtRootInfo info;
    uint32_t st = amt_load(w, root, version, vkind, info);
    ValueLoc l{kNoBlock, 0, 0};
    if (st == IPCFP_ST_TRUE) st = amt_get(w, info, vkind, index[t], l);
    status[t] = uint8_t(st);
    if (loc) loc[t] = (st == IPCFP_ST_TRUE) ? l : ValueLoc{kNoBlock, 0, 0};
}

__global__ __launch_bounds__(256, IPCFP_WALK_WAVES) void k_hamt_get(WitnessView w, CidKey root, uint32_t bit_width, int vkind,
                                                  const uint8_t* __restrict__ keys,
                                                  const uint32_t* __restrict__ key_off,
                                                  const uint32_t* __restrict__ key_len, uint32_t n,
                                                  uint8_t* __restrict__ status, ValueLoc* __restrict__ loc, int pending_only,
                                                  uint32_t lanes) {
    // `lanes` queries per wavefront (hamt_lanes below): lane l of wavefront v takes query v·lanes + l, the other lanes idle
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t lane = g & 63u;
    const uint32_t t = (g >> 6) * lanes + lane;
    if (lane >= lanes || t >= n) return;
    if (pending_only && status[t] != kStPending) return;  // settled from the node table (k_hamt_get_table)
    ValueLoc l{kNoBlock, 0, 0};
    const uint32_t st = hamt_get(w, root, bit_width, vkind, keys + key_off[t], key_len[t], l);
    status[t] = uint8_t(st);
    if (loc) loc[t] = (st == IPCFP_ST_TRUE) ? l : ValueLoc{kNoBlock, 0, 0};
}

// K7 over the node table (hamt_table.h): one query per lane, one record per level.  What the table does not cover is left
// kStPending for k_hamt_get (pending_only).
__global__ __launch_bounds__(256) void k_hamt_get_table(WitnessView w, const HamtNodeRec* __restrict__ table, CidKey root,
                                                        uint32_t bit_width, uint32_t kbit, const uint8_t* __restrict__ keys,
                                                        const uint32_t* __restrict__ key_off, const uint32_t* __restrict__ key_len,
                                                        uint32_t n, uint8_t* __restrict__ status, ValueLoc* __restrict__ loc) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    ValueLoc l{kNoBlock, 0, 0};
    uint32_t st = table_hamt_get(w, table, root, bit_width, kbit, keys + key_off[t], key_len[t], l);
    if (st == kTablePunt) st = kStPending;
    status[t] = uint8_t(st);
    if (loc) loc[t] = (st == IPCFP_ST_TRUE) ? l : ValueLoc{kNoBlock, 0, 0};
}

// A walk is ≈ 190 k instructions per wavefront (≈ 5 node decodes of 1.4-5 KB through a 16-byte window; PMC in
// profiles/r03_hamt_storage_pmc.txt), and the instruction stream does not get shorter with fewer active lanes.  Measured
// on the 66 k-query batch of config 4 (one wavefront per SIMD at 64 queries each): 64 → 1.153 ms, 32 → 1.118 ms,
// 16 → 1.857 ms, 8 → 2.441 ms — a SIMD issues a lone wavefront's instruction every ≈ 14 cycles and four wavefronts' every
// ≈ 6, so halving the lanes once fills the gaps and halving them again only multiplies the instructions.  A batch that
// leaves the chip under two wavefronts per SIMD therefore runs 32 queries per wavefront (IPCFP_HAMT_LANES overrides).
uint32_t hamt_lanes(const ipcfp_ctx* ctx, uint32_t n) {
    static const int forced = [] {
        const char* e = std::getenv("IPCFP_HAMT_LANES");
        return e ? std::atoi(e) : 0;
    }();
    if (forced == 8 || forced == 16 || forced == 32 || forced == 64) return uint32_t(forced);
    const uint32_t simds = uint32_t(ctx->props.multiProcessorCount > 0 ? ctx->props.multiProcessorCount : 256) * 4u;
    return uint64_t(n) < uint64_t(simds) * 128u ? 32u : 64u;
}

uint32_t hamt_kind_bit(int vkind) {
    return vkind == VK_ACTOR_STATE ? uint32_t(HK_ACTOR_STATE) : vkind == VK_VEC_U8 ? uint32_t(HK_VEC_U8) : vkind == VK_ANY ? uint32_t(HK_ANY) : 0u;
}

int launch_hamt_get_table(ipcfp_ctx* ctx, const WitnessView& w, const void* table_d, const CidKey& root, uint32_t bit_width,
                          int vkind, const uint8_t* keys_d, const uint32_t* key_off_d, const uint32_t* key_len_d, uint32_t n,
                          uint8_t* status_d, void* loc_d) {
    if (n == 0) return IPCFP_OK;
    hipLaunchKernelGGL(k_hamt_get_table, dim3(div_up(n, 256)), dim3(256), 0, ctx->stream, w, static_cast<const HamtNodeRec*>(table_d),
                       root, bit_width, hamt_kind_bit(vkind), keys_d, key_off_d, key_len_d, n, status_d, static_cast<ValueLoc*>(loc_d));
    {
        const uint32_t lanes = hamt_lanes(ctx, n);
        hipLaunchKernelGGL(k_hamt_get, dim3(div_up(uint64_t(div_up(n, lanes)) * 64u, 256)), dim3(256), 0, ctx->stream, w, root,
                           bit_width, vkind, keys_d, key_off_d, key_len_d, n, status_d, static_cast<ValueLoc*>(loc_d), 1, lanes);
    }
    IPCFP_HIP(ctx, hipGetLastError());
    return IPCFP_OK;
}

int launch_amt_get(ipcfp_ctx* ctx, const WitnessView& w, const CidKey& root, int version, int vkind,
                   const uint64_t* index_d, uint32_t n, uint8_t* status_d, void* loc_d) {
    if (n == 0) return IPCFP_OK;
    {
        ProfileScope prof(ctx, IPCFP_K_AMT_GET);
        hipLaunchKernelGGL(k_amt_get, dim3(div_up(n, 256)), dim3(256), 0, ctx->stream, w, root, version, vkind,
                           index_d, n, status_d, static_cast<ValueLoc*>(loc_d));
    }
    IPCFP_HIP(ctx, hipGetLastError());
    return IPCFP_OK;
}

int launch_hamt_get(ipcfp_ctx* ctx, const WitnessView& w, const CidKey& root, uint32_t bit_width, int vkind,
                    const uint8_t* keys_d, const uint32_t* key_off_d, const uint32_t* key_len_d, uint32_t n,
                    uint8_t* status_d, void* loc_d, int pending_only) {
    if (n == 0) return IPCFP_OK;
    // (timed by the caller: host/primitives.cpp hamt_get_batch)
    const uint32_t lanes = hamt_lanes(ctx, n);
    hipLaunchKernelGGL(k_hamt_get, dim3(div_up(uint64_t(div_up(n, lanes)) * 64u, 256)), dim3(256), 0, ctx->stream, w, root, bit_width,
                       vkind, keys_d, key_off_d, key_len_d, n, status_d, static_cast<ValueLoc*>(loc_d), pending_only, lanes);
    IPCFP_HIP(ctx, hipGetLastError());
    return IPCFP_OK;
}

}  // namespace ipcfp
