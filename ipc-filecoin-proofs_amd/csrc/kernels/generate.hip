// csrc/kernels/generate.hip — generator-side kernels: witness recording for generate_storage_proof and
// the small gathers generate_event_proof needs.
//
// Replaces generate_storage_proof steps 1-4 (src/proofs/storage/generator.rs:29-155): the child header,
// the state-tree walk, the EVM state and the storage-slot read, each on a RecordingBlockStore.  Here the
// recorder is the `touched` bitmap of the WitnessView: witness_find sets the bit of every block it
// returns, which is what `RecordingBlockStore::get` does for blocks that exist
// (src/proofs/common/blockstore.rs:26-30); a CID that is not in the store is an Err before the witness
// is ever materialised.
#include <hip/hip_runtime.h>

#include "../common.h"
#include "launch.h"
#include "storage_dev.h"

namespace ipcfp {

struct StorageSpec {  // StorageProofSpec { actor_id, slot } (src/proofs/generator.rs:12-15)
    uint64_t actor_id;
    uint8_t slot[32];
};

struct StorageGenOut {  // the fields create_proof_claim needs (storage/generator.rs:158-178)
    CidKey parent_state_root, actor_state, storage_root;
    uint8_t value[32];
    uint32_t status;
    uint32_t pad;
};

__global__ __launch_bounds__(256, IPCFP_WALK_WAVES) void k_generate_storage(WitnessView w, CidKey child,
                                                                            const StorageSpec* __restrict__ specs,
                                                                            uint32_t n, StorageGenOut* __restrict__ out) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    StorageGenOut o;
    for (int i = 0; i < 32; ++i) o.value[i] = 0;
    o.pad = 0;
    for (int j = 0; j < 5; ++j) o.parent_state_root.w[j] = o.actor_state.w[j] = o.storage_root.w[j] = 0;
    // Step 1: extract_and_verify_parent_state (:72-103)
    HeaderLite hdr;
    uint32_t hb;
    uint32_t st = load_header(w, child, hdr, hb);
    if (st == IPCFP_ST_TRUE) {
        o.parent_state_root = hdr.parent_state_root;
        // Step 3: load_actor_and_storage_root (:106-134)
        st = get_actor_state(w, hdr.parent_state_root, specs[t].actor_id, o.actor_state);
    }
    if (st == IPCFP_ST_TRUE) {
        const uint32_t eb = witness_find(w, o.actor_state);
        if (eb == kNoBlock) st = IPCFP_ST_ERR_MISSING_BLOCK;
        else st = parse_evm_state(w, eb, o.storage_root);
    }
    // Step 4: read_storage_value (:137-155)
    if (st == IPCFP_ST_TRUE) st = read_storage_slot_padded(w, o.storage_root, specs[t].slot, o.value);
    o.status = st;
    out[t] = o;
}

// collector.add_cid(..) for CIDs that are known to be needed: mark them, report the ones that are absent
__global__ void k_mark_cids(WitnessView w, const CidKey* __restrict__ keys, uint32_t n, uint32_t* __restrict__ missing) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    if (witness_find(w, keys[t]) == kNoBlock) atomicOr(missing, 1u);
}

// out[i] = table[index[i]]  (message CID of each generated proof = exec[exec_index])
__global__ void k_gather_keys(const CidKey* __restrict__ table, uint64_t table_len, const uint64_t* __restrict__ index,
                              uint32_t n, CidKey* __restrict__ out, uint32_t* __restrict__ out_of_range) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    if (index[t] >= table_len) {
        atomicOr(out_of_range, 1u);
        return;
    }
    out[t] = table[index[t]];
}

// cids of the listed block ids (for the host-side `Cid: Ord` sort of the materialised witness)
__global__ void k_gather_block_cids(const uint8_t* __restrict__ cids, const uint32_t* __restrict__ ids, uint32_t n,
                                    CidKey* __restrict__ out) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    out[t] = load_cid_slot(cids, ids[t]);
}

int launch_generate_storage(ipcfp_ctx* ctx, const WitnessView& w, const CidKey& child, const void* specs_d, uint32_t n,
                            void* out_d) {
    if (n == 0) return IPCFP_OK;
    hipLaunchKernelGGL(k_generate_storage, dim3(div_up(n, 256)), dim3(256), 0, ctx->stream, w, child,
                       static_cast<const StorageSpec*>(specs_d), n, static_cast<StorageGenOut*>(out_d));
    IPCFP_HIP(ctx, hipGetLastError());
    return IPCFP_OK;
}

int launch_mark_cids(ipcfp_ctx* ctx, const WitnessView& w, const CidKey* keys_d, uint32_t n, uint32_t* missing_d) {
    if (n == 0) return IPCFP_OK;
    hipLaunchKernelGGL(k_mark_cids, dim3(div_up(n, 64)), dim3(64), 0, ctx->stream, w, keys_d, n, missing_d);
    IPCFP_HIP(ctx, hipGetLastError());
    return IPCFP_OK;
}

int launch_gather_keys(ipcfp_ctx* ctx, const CidKey* table_d, uint64_t table_len, const uint64_t* index_d, uint32_t n,
                       CidKey* out_d, uint32_t* oor_d) {
    if (n == 0) return IPCFP_OK;
    hipLaunchKernelGGL(k_gather_keys, dim3(div_up(n, 256)), dim3(256), 0, ctx->stream, table_d, table_len, index_d, n, out_d,
                       oor_d);
    IPCFP_HIP(ctx, hipGetLastError());
    return IPCFP_OK;
}

int launch_gather_block_cids(ipcfp_ctx* ctx, const uint8_t* cids_d, const uint32_t* ids_d, uint32_t n, CidKey* out_d) {
    if (n == 0) return IPCFP_OK;
    hipLaunchKernelGGL(k_gather_block_cids, dim3(div_up(n, 256)), dim3(256), 0, ctx->stream, cids_d, ids_d, n, out_d);
    IPCFP_HIP(ctx, hipGetLastError());
    return IPCFP_OK;
}

}  // namespace ipcfp
