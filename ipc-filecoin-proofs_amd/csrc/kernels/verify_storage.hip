// csrc/kernels/verify_storage.hip — batch `verify_storage_proof`, one proof per lane.
//
// Replaces src/proofs/storage/verifier.rs:24-63 (steps 2-6; step 1, the per-proof rebuild of the
// witness store, src/proofs/verifier.rs:19-28, is gone: the witness is resident and indexed once)
// with read_storage_slot (src/proofs/storage/decode.rs:36-97), get_actor_state / parse_evm_state
// (src/proofs/common/decode.rs:17-42,79-97) and left_pad_32 (src/proofs/common/evm.rs:91-100).
// Check order and Ok(false)/Err outcomes follow SURVEY.md A.10 exactly; the status byte says
// which line decided.
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "../common.h"
#include "claims_dev.h"
#include "launch.h"
#include "storage_dev.h"

namespace ipcfp {

__device__ __forceinline__ bool trusted(const ipcfp_trust_policy_t& t, long long epoch) {
    if (t.kind == 0) return true;                       // AcceptAll
    if (t.ec_chain_empty) return false;                 // cert.rs:55-57
    return epoch >= t.min_epoch && epoch <= t.max_epoch;  // cert.rs:60-63
}

__device__ __forceinline__ uint32_t verify_storage_one(const WitnessView& w, const StorageClaimPacked& c,
                                                       const ipcfp_trust_policy_t& trust) {
    // Step 2: verify_trust_anchor (storage/verifier.rs:81-92)
    if (!(c.flags & SC_CHILD_PARSED)) return IPCFP_ST_ERR_BAD_CLAIM;             // :85
    if (!trusted(trust, c.child_epoch)) return IPCFP_ST_FALSE_UNTRUSTED_CHILD;    // :87
    // Step 3: verify_parent_state_root (:95-111)
    HeaderLite hdr;
    uint32_t hb;
    uint32_t st = load_header(w, c.child, hdr, hb);                               // :101-107
    if (st != IPCFP_ST_TRUE) return st;
    if (!((c.flags & SC_STATE_ROOT_CANON) && cid_equal(hdr.parent_state_root, c.state_root)))
        return IPCFP_ST_FALSE_STATE_ROOT;                                         // :110 string compare
    // Step 4: verify_actor_state (:114-127)
    CidKey actor_state;
    st = get_actor_state(w, c.state_root, c.actor_id, actor_state);               // :122
    if (st != IPCFP_ST_TRUE) return st;
    if (!((c.flags & SC_ACTOR_STATE_CANON) && cid_equal(actor_state, c.actor_state)))
        return IPCFP_ST_FALSE_ACTOR_STATE;                                        // :126
    // Step 5: verify_storage_root (:130-145)
    const uint32_t eb = witness_find(w, c.actor_state);                           // :136-138
    if (eb == kNoBlock) return IPCFP_ST_ERR_MISSING_BLOCK;
    CidKey contract_state;
    st = parse_evm_state(w, eb, contract_state);                                  // :141
    if (st != IPCFP_ST_TRUE) return st;
    if (!((c.flags & SC_STORAGE_ROOT_CANON) && cid_equal(contract_state, c.storage_root)))
        return IPCFP_ST_FALSE_STORAGE_ROOT;                                       // :144
    // Step 6: verify_storage_value (:148-170)
    if (!(c.flags & SC_SLOT_PARSED)) return IPCFP_ST_ERR_BAD_CLAIM;              // :155-157
    uint8_t padded[32];
    st = read_storage_slot_padded(w, c.storage_root, c.slot, padded);             // :160-165
    if (st != IPCFP_ST_TRUE) return st;
    if (!(c.flags & SC_VALUE_MATCHABLE)) return IPCFP_ST_FALSE_VALUE;            // claimed string can never equal "0x"+64 hex
    bool eq = true;
    for (int i = 0; i < 32; ++i) eq &= padded[i] == c.value[i];
    return eq ? IPCFP_ST_TRUE : IPCFP_ST_FALSE_VALUE;                             // :169
}

// WAVES = wavefronts per SIMD the register allocator must leave room for.  verify_storage_one inlines six
// layout attempts, each a Keccak + SHA-256 + HAMT walk: at 4 waves (128 VGPRs) it spills 384 bytes per
// lane to scratch, at 3 waves (168 VGPRs) it does not.  IPCFP_STORAGE_WAVES selects (default: see launch).
template <int WAVES>
__global__ __launch_bounds__(256, WAVES) void k_verify_storage(WitnessView w, const StorageClaimPacked* __restrict__ claims,
                                                               uint32_t n, ipcfp_trust_policy_t trust,
                                                               uint8_t* __restrict__ status) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    status[t] = uint8_t(verify_storage_one(w, claims[t], trust));
}

int launch_verify_storage(ipcfp_ctx* ctx, const WitnessView& w, const StorageClaimPacked* claims_d, uint32_t n,
                          const ipcfp_trust_policy_t& trust, uint8_t* status_d) {
    if (n == 0) return IPCFP_OK;
    {
        ProfileScope prof(ctx, IPCFP_K_STORAGE_VERIFY);
        static const int waves = [] {
            const char* e = std::getenv("IPCFP_STORAGE_WAVES");
            const int v = e ? std::atoi(e) : 4;
            return v >= 2 && v <= 4 ? v : 4;
        }();
        const dim3 g(div_up(n, 256)), blk(256);
        if (waves == 2) hipLaunchKernelGGL(k_verify_storage<2>, g, blk, 0, ctx->stream, w, claims_d, n, trust, status_d);
        else if (waves == 3) hipLaunchKernelGGL(k_verify_storage<3>, g, blk, 0, ctx->stream, w, claims_d, n, trust, status_d);
        else hipLaunchKernelGGL(k_verify_storage<4>, g, blk, 0, ctx->stream, w, claims_d, n, trust, status_d);
    }
    IPCFP_HIP(ctx, hipGetLastError());
    return IPCFP_OK;
}

}  // namespace ipcfp
