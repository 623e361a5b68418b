// csrc/kernels/verify_table.hip — `verify_single_proof` (src/proofs/events/verifier.rs:92-290) for the claims whose
// receipt was enumerated and whose events are tabulated (event_table.h): steps 1-3 are table lookups (execution-order
// hash table), step 4 compares the claim with bytes at addresses the records name.  No block is parsed here, so the
// kernel needs neither the CBOR reader nor LDS and runs at full occupancy; a claim the table does not cover is
// marked kStPending and taken by the general walker (k_verify_events) right behind.
#include <hip/hip_runtime.h>

#include "../common.h"
#include "launch.h"
#include "verify_dev.h"

namespace ipcfp {

__device__ __forceinline__ uint32_t verify_table_one(const WitnessView& w, const EventClaimPacked& c, const TipsetCtxDev& tc,
                                                     const uint8_t* __restrict__ blob, const ipcfp_trust_policy_t& trust,
                                                     const ipcfp_event_filter_t& filter, bool has_filter, ValueLoc* where) {
    uint32_t st = verify_event_prefix(c, tc, trust);
    if (st == IPCFP_ST_TRUE) {
        bool settled;
        st = verify_event_from_table(w, c, tc, blob, filter, has_filter, where, settled);
    }
    return st;
}

__global__ __launch_bounds__(256) void k_verify_events_table(WitnessView w, const EventClaimPacked* __restrict__ claims, uint32_t n,
                                                             const TipsetCtxDev* __restrict__ ctxs, uint32_t n_ctxs,
                                                             const uint8_t* __restrict__ blob, uint64_t blob_len,
                                                             ipcfp_trust_policy_t trust, ipcfp_event_filter_t filter,
                                                             int has_filter, uint8_t* __restrict__ status,
                                                             ValueLoc* __restrict__ where) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = t < n;
    EventClaimPacked c;
    if (live) c = claims[t];
    else c.context = 0;
    ValueLoc loc{kNoBlock, 0, 0};
    uint32_t st = IPCFP_ST_ERR_BAD_CLAIM;
    const bool inb = live && claim_in_bounds(c, n_ctxs, blob_len);
    // The proofs of a wavefront nearly always name ONE tipset pair: its context (≈700 bytes of header facts and
    // table pointers) is then read through the scalar unit once per wavefront instead of by 64 lanes apiece.
    const uint32_t ctx0 = __builtin_amdgcn_readfirstlane(inb ? c.context : 0xffffffffu);
    if (__all(!inb || c.context == ctx0)) {
        if (inb) st = verify_table_one(w, c, ctxs[ctx0], blob, trust, filter, has_filter != 0, where ? &loc : nullptr);
    } else if (inb) {
        st = verify_table_one(w, c, ctxs[c.context], blob, trust, filter, has_filter != 0, where ? &loc : nullptr);
    }
    if (!live) return;
    status[t] = uint8_t(st);
    if (where) where[t] = loc;
}

int launch_verify_events_table(ipcfp_ctx* ctx, const WitnessView& w, const EventClaimPacked* claims_d, uint32_t n,
                               const TipsetCtxDev* ctxs_d, uint32_t n_ctxs, const uint8_t* blob_d, uint64_t blob_len,
                               const ipcfp_trust_policy_t& trust, const ipcfp_event_filter_t& filter, int has_filter,
                               uint8_t* status_d, void* where_d) {
    if (n == 0) return IPCFP_OK;
    hipLaunchKernelGGL(k_verify_events_table, dim3(div_up(n, 256)), dim3(256), 0, ctx->stream, w, claims_d, n, ctxs_d, n_ctxs,
                       blob_d, blob_len, trust, filter, has_filter, status_d, static_cast<ValueLoc*>(where_d));
    IPCFP_HIP(ctx, hipGetLastError());
    return IPCFP_OK;
}

}  // namespace ipcfp
