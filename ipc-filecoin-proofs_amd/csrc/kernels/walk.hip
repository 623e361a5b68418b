// csrc/kernels/walk.hip — K5 (amt_get) and K7 (hamt_get) batch kernels: one query per lane.
//
// K5 replaces `receipts_amt.get(exec_index)` / `events_amt.get(event_index)` / `r_amt.get(i)`
// (src/proofs/events/verifier.rs:224,237; src/proofs/events/generator.rs:249);
// K7 replaces `actors.get(&key)` / `hamt.get(slot)` (src/proofs/common/decode.rs:35-39,
// src/proofs/storage/decode.rs:81,88,96).
#include <hip/hip_runtime.h>

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
                                                  uint8_t* __restrict__ status, ValueLoc* __restrict__ loc) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    ValueLoc l{kNoBlock, 0, 0};
    const uint32_t st = hamt_get(w, root, bit_width, vkind, keys + key_off[t], key_len[t], l);
    status[t] = uint8_t(st);
    if (loc) loc[t] = (st == IPCFP_ST_TRUE) ? l : ValueLoc{kNoBlock, 0, 0};
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
                    uint8_t* status_d, void* loc_d) {
    if (n == 0) return IPCFP_OK;
    {
        ProfileScope prof(ctx, IPCFP_K_HAMT_GET);
        hipLaunchKernelGGL(k_hamt_get, dim3(div_up(n, 256)), dim3(256), 0, ctx->stream, w, root, bit_width, vkind,
                           keys_d, key_off_d, key_len_d, n, status_d, static_cast<ValueLoc*>(loc_d));
    }
    IPCFP_HIP(ctx, hipGetLastError());
    return IPCFP_OK;
}

}  // namespace ipcfp
