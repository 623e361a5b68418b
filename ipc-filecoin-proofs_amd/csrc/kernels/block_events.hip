// csrc/kernels/block_events.hip — the event table's first step: every block of the witness parsed ONCE, in arena
// order, out of LDS (event_table.h).
//
// What the reference does per receipt and per proof — `Amt::<StampedEvent>::load(events_root)` + for_each / get +
// `extract_evm_log` (src/proofs/events/generator.rs:215-233,259-297; src/proofs/events/verifier.rs:234-239;
// src/proofs/common/evm.rs:13-59) — is a pure function of the events AMT's root BLOCK when that root is a leaf (the
// shape FVM writes for up to 32 events per message).  So the parse does not have to wait for the receipts: it is done
// for every block of the witness in the order the blocks lie in the arena.
//
// Why in this order: a lane that parses "its receipt's" block reads a block somewhere in a 0.44 GB arena, 16 bytes
// at a time, every load a line of its own and every load waiting for the one before it (k_event_table: 370 µs for
// 1 M receipts, FETCH_SIZE 1.8x the witness).  In arena order the 64 blocks of a wavefront are one contiguous,
// line-aligned span: the wavefront copies it into LDS with coalesced 16-byte loads (1 KB per instruction, all
// independent), and each lane then parses its own block with the SAME strict reader, whose chunk loads are now
// ds_read_b128.  HBM sees every line exactly once; the parse is bound by LDS latency and VALU, not by DRAM round trips.
//
// The kernel decides nothing by itself: a block that is not exactly the tabulated shape — not an AMT at all (most
// interior nodes, headers, message lists), a taller tree, links in the root, a decode error, an oversized event, an
// exhausted record pool — is left RK_WALK, and a receipt that points at it takes the general walkers, which name
// the reference's outcome.  RK_TABLE promises: this block decodes as a height-0 Amt<StampedEvent> root whose
// values are the tabulated records.
#define IPCFP_RD_LDS 1
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "../common.h"
#include "cbor_dev.h"
#include "event_log_dev.h"
#include "event_table.h"
#include "launch.h"
#include "block_events_body.h"

namespace ipcfp {

// The stage of a wavefront is dynamic LDS: `stage_chunks` 16-byte chunks.  It trades lanes for occupancy — 32 KB
// holds 64 blocks of four lines in one batch but leaves 5 wavefronts per CU; 12-16 KB takes the average block
// (341 B) in two batches of half-empty waves at 10-13 wavefronts per CU (IPCFP_BLOCK_STAGE_KB, measured default).
constexpr uint32_t kDefaultStageKB = 24;

typedef __attribute__((address_space(3))) const uint8_t* lds_bytes_t;

__device__ __forceinline__ uint64_t readlane64(uint64_t v, uint32_t lane) {
    const uint32_t lo = __builtin_amdgcn_readlane(uint32_t(v), lane);
    const uint32_t hi = __builtin_amdgcn_readlane(uint32_t(v >> 32), lane);
    return (uint64_t(hi) << 32) | lo;
}

// One wavefront = one workgroup = 64 consecutive blocks of the schedule.
__global__ __launch_bounds__(64) void k_block_events(const uint8_t* __restrict__ arena, const K1Meta* __restrict__ meta,
                                                     uint32_t n, ScanParams sp, int count_matches,
                                                     BlockRec* __restrict__ brecs, EventRec* __restrict__ erecs,
                                                     uint32_t cap_events, uint32_t* __restrict__ pool_used,
                                                     uint32_t kStageChunks) {
    extern __shared__ rd_chunk_t stage[];
    const uint32_t lane = threadIdx.x;
    const uint32_t s = blockIdx.x * 64u + lane;
    const bool live = s < n;
    K1Meta m{0, 0, kNoBlock};
    if (live) m = meta[s];
    const uint32_t padded = m.len == 0 ? 128u : (m.len + 127u) & ~127u;  // what the block owns in the arena (repack.hip)
    // the block starts where its predecessor in the schedule ends (true for every arena repack.hip lays out; a lane
    // for which it is not simply starts a batch of its own)
    const uint64_t my_end = m.off + padded;
    const uint64_t prev_end = (uint64_t(uint32_t(__shfl_up(uint32_t(my_end >> 32), 1, 64))) << 32) |
                              uint32_t(__shfl_up(uint32_t(my_end), 1, 64));
    const bool follows = lane > 0 && prev_end == m.off;
    // what a batch must hold of this block when it is the last one: its bytes in whole chunks plus two chunks of the
    // reader's read-ahead (the last block's line padding is not needed; the arena's tail slack covers the read)
    const uint32_t tail = ((m.len + 15u) & ~15u) + 32u;
    const bool stageable = live && (m.off & 127ull) == 0 && m.len <= kStageChunks * 16u - 32u;
    BlockRec br{RK_WALK, 0, 0};
    uint64_t todo = __ballot(stageable);
    while (todo) {
        // ---- the batch: the longest run of lanes from the first one left whose blocks are contiguous and fit ----
        const uint32_t first = uint32_t(__ffsll((unsigned long long)todo)) - 1u;
        const uint64_t base_off = readlane64(m.off, first);
        const uint64_t rel64 = m.off - base_off;
        const bool fits = stageable && lane >= first && m.off >= base_off && rel64 + tail <= uint64_t(kStageChunks) * 16u &&
                          (lane == first || follows);
        const uint64_t breaks = __ballot(!fits) & ~((2ull << first) - 1ull);  // lanes above `first` that end the run
        const uint32_t stop = breaks ? uint32_t(__ffsll((unsigned long long)breaks)) - 1u : 64u;
        const bool mine = lane >= first && lane < stop;
        const uint32_t span = uint32_t(readlane64(rel64 + tail, stop - 1u));  // bytes, a multiple of 16, <= the stage
        // ---- stage the span ----
        {
            const rd_chunk_t* src = reinterpret_cast<const rd_chunk_t*>(arena + base_off);
            const uint32_t chunks = span >> 4;
            for (uint32_t q0 = 0; q0 < chunks; q0 += 512u) {
                rd_chunk_t v[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const uint32_t q = q0 + 64u * uint32_t(k) + lane;
                    v[k] = q < chunks ? src[q] : rd_chunk_t{0, 0};
                }
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const uint32_t q = q0 + 64u * uint32_t(k) + lane;
                    if (q < chunks) stage[q] = v[k];
                }
            }
        }
        __syncthreads();
        // ---- parse out of LDS (block_events_body.h) ----
        Rd r;
        r.init((lds_bytes_t)(stage) + (mine ? uint32_t(rel64) : 0u), mine ? m.len : 0u);
        block_events_parse(r, mine, m.off, sp, count_matches, erecs, cap_events, pool_parts(n), pool_used, blockIdx.x, lane, br);
        todo &= ~(((stop < 64u ? (1ull << stop) : 0ull) - 1ull) & ~((1ull << first) - 1ull));
        __syncthreads();  // the next batch overwrites the stage
    }
    if (live) brecs[m.id] = br;
}

// the per-lane variants (block_events_lane.inc): the same parse straight from the arena, one block per lane in arena
// order, without (1) or with (2) the reader's per-lane line staging — A/B material for IPCFP_BLOCK_EVENTS_MODE
void launch_block_events_plain(hipStream_t stream, const uint8_t* arena, const K1Meta* meta, uint32_t n, const ScanParams& sp,
                               int count_matches, BlockRec* brecs, EventRec* erecs, uint32_t cap_events, uint32_t* pool_used);
void launch_block_events_linestage(hipStream_t stream, const uint8_t* arena, const K1Meta* meta, uint32_t n, const ScanParams& sp,
                                   int count_matches, BlockRec* brecs, EventRec* erecs, uint32_t cap_events, uint32_t* pool_used);

int launch_block_events(ipcfp_ctx* ctx, hipStream_t stream, const uint8_t* arena, const void* meta_d, uint32_t n,
                        const ipcfp_event_filter_t* filter, int has_actor, uint64_t actor, BlockRec* brecs_d,
                        EventRec* erecs_d, uint32_t cap_events, uint32_t* pool_used_d) {
    if (n == 0) return IPCFP_OK;
    ScanParams sp{};
    if (filter) sp = ScanParams{*filter, actor, has_actor ? 1u : 0u, 0};
    static const int mode = [] {
        const char* e = std::getenv("IPCFP_BLOCK_EVENTS_MODE");
        return e ? std::atoi(e) : 2;
    }();
    static const uint32_t stage_chunks = [] {
        const char* e = std::getenv("IPCFP_BLOCK_STAGE_KB");
        int kb = e ? std::atoi(e) : int(kDefaultStageKB);
        if (kb < 2) kb = 2;
        if (kb > 64) kb = 64;
        return uint32_t(kb) * 64u;
    }();
    {
        ProfileScope prof(ctx, IPCFP_K_EVENT_SCAN, stream);
        const K1Meta* meta = static_cast<const K1Meta*>(meta_d);
        if (mode == 1)
            launch_block_events_plain(stream, arena, meta, n, sp, filter ? 1 : 0, brecs_d, erecs_d, cap_events, pool_used_d);
        else if (mode == 2)
            launch_block_events_linestage(stream, arena, meta, n, sp, filter ? 1 : 0, brecs_d, erecs_d, cap_events, pool_used_d);
        else
            hipLaunchKernelGGL(k_block_events, dim3(div_up(n, 64)), dim3(64), stage_chunks * 16u, stream, arena, meta, n, sp,
                               filter ? 1 : 0, brecs_d, erecs_d, cap_events, pool_used_d, stage_chunks);
    }
    IPCFP_HIP(ctx, hipGetLastError());
    return IPCFP_OK;
}

}  // namespace ipcfp
