// csrc/kernels/block_events.hip — the event table's first step: every block of the witness parsed ONCE, in arena
// order (event_table.h).  This unit is the launcher; the kernel is block_events_linestage.hip (block_events_lane.inc +
// block_events_body.h).
//
// What the reference does per receipt and per proof — `Amt::<StampedEvent>::load(events_root)` + for_each / get +
// `extract_evm_log` (src/proofs/events/generator.rs:215-233,259-297; src/proofs/events/verifier.rs:234-239;
// src/proofs/common/evm.rs:13-59) — is a pure function of the events AMT's root BLOCK when that root is a leaf (the
// shape FVM writes for up to 32 events per message).  So the parse does not have to wait for the receipts: it is done
// for every block of the witness in the order the blocks lie in the arena, on a stream of its own, beside K1 and the
// receipts enumeration.
//
// The kernel decides nothing by itself: a block that is not exactly the tabulated shape — not an AMT at all (most
// interior nodes, headers, message lists), a taller tree, links in the root, a decode error, an oversized event, an
// exhausted record pool — is left RK_WALK, and a receipt that points at it takes the general walkers, which name
// the reference's outcome.  RK_TABLE promises: this block decodes as a height-0 Amt<StampedEvent> root whose
// values are the tabulated records.
//
// Three forms of the same parse were built and measured on one box (profiles/r02_experiments.md): a wavefront staging
// the contiguous arena span of its 64 blocks in LDS (510-770 µs: a 24-32 KB stage leaves 1.3 wavefronts per SIMD), one
// block per lane on global memory (452 µs at 7 wavefronts per SIMD), and one block per lane with the reader's
// per-lane line staging (341 µs at 4) — the one that is shipped.  The other two were removed in round 3.
#include <hip/hip_runtime.h>

#include "../common.h"
#include "event_table.h"
#include "launch.h"

namespace ipcfp {

void launch_block_events_linestage(hipStream_t stream, const uint8_t* arena, const K1Meta* meta, uint32_t n, const ScanParams& sp,
                                   int count_matches, BlockRec* brecs, EventRec* erecs, uint32_t cap_events, uint32_t* pool_used);

int launch_block_events(ipcfp_ctx* ctx, hipStream_t stream, const uint8_t* arena, const void* meta_d, uint32_t n,
                        const ipcfp_event_filter_t* filter, int has_actor, uint64_t actor, BlockRec* brecs_d,
                        EventRec* erecs_d, uint32_t cap_events, uint32_t* pool_used_d) {
    if (n == 0) return IPCFP_OK;
    ScanParams sp{};
    if (filter) sp = ScanParams{*filter, actor, has_actor ? 1u : 0u, 0};
    {
        ProfileScope prof(ctx, IPCFP_K_EVENT_SCAN, stream);
        launch_block_events_linestage(stream, arena, static_cast<const K1Meta*>(meta_d), n, sp, filter ? 1 : 0, brecs_d, erecs_d,
                                      cap_events, pool_used_d);
    }
    IPCFP_HIP(ctx, hipGetLastError());
    return IPCFP_OK;
}

}  // namespace ipcfp
