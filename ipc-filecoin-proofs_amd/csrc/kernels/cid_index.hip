// csrc/kernels/cid_index.hip — K4: the CID → block-id index of the witness store.
//
// Replaces `MemoryBlockstore::put_keyed` into a `HashMap<Cid, Vec<u8>>`
// (src/proofs/events/verifier.rs:82-86, src/proofs/storage/verifier.rs:69-75).  Same
// observable semantics: no hashing of the data on insert and a
// duplicate CID keeps the LAST block inserted (HashMap::insert overwrites).
//
// Layout: power-of-two open-addressing table of u32 block ids, load factor ≤ 0.5,
// linear probing; keys are compared against the cids[] array (5 × u64 per CID).
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "../common.h"
#include "launch.h"
#include "witness_dev.h"

namespace ipcfp {

// `done`: every workgroup counts itself when its keys are in (all its atomics have returned by then), so that a reader on
// another stream can tell a key that is not in the table YET from one that never will be (tipset_prepare.hip LiveIndex).
template <bool CAS_FIRST>
__global__ __launch_bounds__(256) void k_index_insert(const uint8_t* __restrict__ cids, uint32_t n,
                                                      uint32_t* __restrict__ slots, uint32_t mask, uint32_t* __restrict__ done) {
    // workgroups are handed out in blockIdx order: the even ones take the table of CIDs from the front, the odd ones from
    // the back, so that BOTH ends of the witness are in within the first microseconds — a recorded witness holds the
    // headers and roots first (they are read first), one built bottom-up holds them last, and the tipset prologue on the
    // head stream is waiting for exactly those keys
    const uint32_t wg = (blockIdx.x & 1u) ? gridDim.x - 1u - (blockIdx.x >> 1) : (blockIdx.x >> 1);
    const uint32_t i = wg * blockDim.x + threadIdx.x;
    if (i < n) index_insert_key<CAS_FIRST>(cids, slots, mask, i);
    if (done) {  // (only a context with a head stream has a reader for the count)
        __syncthreads();
        if (threadIdx.x == 0) atomicAdd(done, 1u);
    }
}

int launch_index_insert(ipcfp_ctx* ctx, const uint8_t* cids_d, uint32_t n, uint32_t* slots_d, uint32_t mask) {
    if (n == 0) return IPCFP_OK;
    hipLaunchKernelGGL(k_index_insert<true>, dim3(div_up(n, 256)), dim3(256), 0, ctx->stream, cids_d, n, slots_d, mask,
                       static_cast<uint32_t*>(nullptr));
    IPCFP_HIP(ctx, hipGetLastError());
    return IPCFP_OK;
}

// (The same table in two passes — every key first STORES its id on its home slot, then only the keys that find another
// id there resolve the collision with CAS / atomicMax — was built to spare 70 % of the atomics and is slower: 0.29 ms
// against 0.15 ms.  Random 4-byte stores and a second random read of the table cost more than the atomics they
// replace.  profiles/r03_experiments.md)
int witness_build_index(ipcfp_ctx* ctx, ipcfp_witness* w) {
    const uint32_t n = uint32_t(w->n);
    uint32_t size = 64;
    while (size < 2ull * n) size <<= 1;
    if (w->index_slots.count != size) IPCFP_HIP(ctx, w->index_slots.alloc(size));  // rebuilds reuse the table
    w->index_mask = size - 1;
    IPCFP_HIP(ctx, hipMemsetAsync(w->index_slots.p, 0xff, size_t(size) * 4, ctx->stream));
    w->index_wgs = div_up(n, 256);
    if (ctx->stream_head) {  // the insert workgroups count themselves done; "the table is cleared": where a lookup on the head stream may start
        if (!w->index_done.p) IPCFP_HIP(ctx, w->index_done.alloc_unpooled(1));
        IPCFP_HIP(ctx, hipMemsetAsync(w->index_done.p, 0, 4, ctx->stream));
        if (!w->index_event) IPCFP_HIP(ctx, hipEventCreateWithFlags(&w->index_event, hipEventDisableTiming));
        IPCFP_HIP(ctx, hipEventRecord(w->index_event, ctx->stream));
    }
    if (n == 0) return IPCFP_OK;
    {
        ProfileScope prof(ctx, IPCFP_K_CID_INDEX);
        static const bool cas_first = [] {
            const char* e = std::getenv("IPCFP_INDEX_CAS_FIRST");
            return !(e && std::atoi(e) == 0);
        }();
        if (cas_first)
            hipLaunchKernelGGL(k_index_insert<true>, dim3(div_up(n, 256)), dim3(256), 0, ctx->stream, w->cids.p, n,
                               w->index_slots.p, w->index_mask, w->index_done.p);
        else
            hipLaunchKernelGGL(k_index_insert<false>, dim3(div_up(n, 256)), dim3(256), 0, ctx->stream, w->cids.p, n,
                               w->index_slots.p, w->index_mask, w->index_done.p);
    }
    IPCFP_HIP(ctx, hipGetLastError());
    return IPCFP_OK;
}

}  // namespace ipcfp
