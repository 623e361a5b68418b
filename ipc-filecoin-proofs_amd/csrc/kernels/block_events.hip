// csrc/kernels/block_events.hip — the event table's first step: every block of the witness parsed ONCE, in arena
// order (event_table.h): k_block_events and its launcher.
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
// One block per lane, the reader staging each 128-byte line it touches in the lane's LDS slot (cbor_dev.h
// IPCFP_LINE_STAGE).  Forms that were built, measured on one box and removed: a wavefront staging the contiguous arena span
// of its 64 blocks in LDS (510-770 µs: a 24-32 KB stage leaves 1.3 wavefronts per SIMD) and one block per lane straight on
// global memory (452 µs at 7 wavefronts per SIMD) in round 2 (profiles/r02_experiments.md); a 128-byte sliding window
// per lane refilled one event ahead from registers in round 5 (203 µs against 156 for this form: the window's
// bookkeeping is more instructions than the refills' latency costs at 4 wavefronts per SIMD; profiles/r05_experiments.md).
#define IPCFP_LINE_STAGE 1
#include <hip/hip_runtime.h>

#include "../common.h"
#include "cbor_dev.h"
#include "event_log_dev.h"
#include "event_table.h"
#include "launch.h"

namespace ipcfp {

// EVERY lane of the wavefront calls this (the record reservation is a wave-level prefix sum); the lanes with `mine`
// parse the block their reader sits on.  On success br becomes RK_TABLE; it is left alone otherwise (RK_WALK).
// The root of an events AMT in the shape FVM writes it — `84 bw 00 count 83 4x bitmap 80 8n`: bit width and height as
// immediate uints, the count immediate or one byte, a leaf node whose bitmap has the length the width asks for, no
// links, up to 255 values — settled from the 16 bytes at the head of the block without a branch.  Exactly what the item-
// by-item decode below accepts for these bytes (same bitmap, same value count, same position afterwards); any other
// spelling returns false with the reader untouched and the item-by-item decode takes the block.  Eight item headers
// are ≈ 550 instructions, and every lane of every wavefront of k_block_events pays them.
__device__ __forceinline__ bool amt_leaf_root_fast(Rd& r, uint32_t& nv, uint64_t& bits) {
    if (r.err || r.pos != 0u || r.n < 8u) return false;
    uint64_t w0, w1;
    r.peek128(0u, w0, w1);
    const uint32_t b0 = uint32_t(w0) & 0xffu, bw = uint32_t(w0 >> 8) & 0xffu, b2 = uint32_t(w0 >> 16) & 0xffu,
                   b3 = uint32_t(w0 >> 24) & 0xffu;
    bool ok = b0 == 0x84u && bw >= 1u && bw <= 6u && b2 == 0u;
    const bool c_imm = b3 < 0x18u, c_1 = b3 == 0x18u;
    ok = ok && (c_imm || c_1);
    const uint32_t o = c_imm ? 4u : 5u;                    // the node: 83, the bitmap's header, the bitmap
    const uint32_t bwq = (bw >= 1u && bw <= 6u) ? bw : 1u;
    const uint32_t width = 1u << bwq, bl = (width + 7u) / 8u;  // 1, 1, 1, 2, 4, 8 bytes
    const uint64_t nw = bytes_from(w0, w1, o);
    ok = ok && (uint32_t(nw) & 0xffffu) == (0x83u | ((0x40u + bl) << 8));
    const uint32_t bo = o + 2u;                            // 6 or 7
    uint64_t bm = bytes_from(w0, w1, bo);                  // the bitmap's bytes (byte 0 = indices 0..7) and what follows
    const uint32_t to = bo + bl;                           // links header, values header: byte 7..16
    ok = ok && to + 2u <= 16u;
    const uint32_t tq = to + 2u <= 16u ? to : 8u;
    const uint64_t tw = bytes_from(w0, w1, tq > 8u ? 8u : tq) >> (tq > 8u ? 8u * (tq - 8u) : 0u);
    const uint32_t l = uint32_t(tw) & 0xffu, v = uint32_t(tw >> 8) & 0xffu, v2 = uint32_t(tw >> 16) & 0xffu;
    const bool v_imm = (v >> 5) == 4u && (v & 31u) < 24u, v_1 = v == 0x98u && tq + 3u <= 16u;
    ok = ok && l == 0x80u && (v_imm || v_1);
    if (bl < 8u) bm &= (1ull << (8u * bl)) - 1ull;
    if (width < 64u) bm &= (1ull << width) - 1ull;
    const uint32_t nvals = v_imm ? (v & 31u) : v2;
    const uint32_t total = to + (v_imm ? 2u : 3u);
    ok = ok && total <= r.n && nvals == uint32_t(__popcll(bm));
    if (!ok) return false;
    nv = nvals;
    bits = bm;
    r.pos = total;
    return true;
}

__device__ __forceinline__ void block_events_parse(Rd& r, bool mine, uint64_t arena_off, const ScanParams& sp, int count_matches,
                                                   EventRec* __restrict__ erecs, uint32_t cap_events, uint32_t n_parts,
                                                   uint32_t* __restrict__ pool_used, uint32_t wave_no, uint32_t lane,
                                                   BlockRec& br) {
    // ---- header first, so the record segment can be reserved before the events are read ----
    uint32_t nv = 0;
    uint64_t bits = 0;
    bool table = false;
    if (mine && amt_leaf_root_fast(r, nv, bits)) {
        table = true;
    } else if (mine) {
        r.expect_array(4);
        const uint64_t bw = r.read_uint();
        if (r.ok() && (bw < 1 || bw > 6)) r.fail();
        const uint64_t height = r.read_uint();
        (void)r.read_uint();  // count: checked by neither load nor for_each
        if (r.ok() && height == 0) {
            const uint32_t width = 1u << uint32_t(bw);
            r.expect_array(3);
            uint32_t bo, bl;
            r.read_bytes(bo, bl);
            if (r.ok() && bl == (width + 7) / 8) {
                bits = r.peek64(bo);
                if (bl < 8) bits &= (1ull << (8u * bl)) - 1ull;
                if (width < 64) bits &= (1ull << width) - 1ull;
                const uint64_t nl = r.read_array();
                const uint64_t nvals = r.ok() && nl == 0 ? r.read_array() : ~0ull;
                if (r.ok() && nl == 0 && nvals == uint64_t(__popcll(bits))) {
                    table = true;
                    nv = uint32_t(nvals);
                }
            }
        }
    }
    // reserve nv records: one atomic per wavefront and batch
    uint32_t rec_first;
    {
        const uint32_t want = table ? nv : 0;
        uint32_t incl = want;
    #pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t up = __shfl_up(incl, d, 64);
            if (lane >= uint32_t(d)) incl += up;
        }
        const uint32_t wave_total = __shfl(incl, 63, 64);
        uint32_t base = 0;
        const uint32_t part = wave_no % n_parts, part_cap = cap_events / n_parts;
        if (lane == 63 && wave_total) base = atomicAdd(pool_used + part * kPoolCounterStride, wave_total);
        base = __shfl(base, 63, 64);
        rec_first = base + incl - want;
        if (table && uint64_t(rec_first) + nv > part_cap) table = false;  // partition exhausted: the receipt walks
        rec_first += part * part_cap;
    }
    if (table) {
        // ---- the events, each decoded once (the decode IS the per-value type check of Amt::load) ----
        bool oversize = false;
        uint32_t c = 0;
        for (uint32_t j = 0; j < nv && r.ok(); ++j) {
            const uint32_t start = r.pos;
            uint64_t emitter;
            EvmLogLoc log;
            decode_event_log(r, emitter, log);
            if (!r.ok()) break;
            EventRec e;
            const uint32_t len = r.pos - start;
            const uint64_t flags = (uint64_t(log.n_topics & 0xffu) << kEvTopicShift) | (log.is_log ? kEvIsLog : 0) |
                                   (log.case_a ? kEvCaseA : 0);
            oversize |= len > 0xffffu || log.n_topics > 255u;
            e.base_flags = ((arena_off + start) & kEvBaseMask) | flags;
            e.emitter = emitter;
    #pragma unroll
            for (int q = 0; q < 4; ++q) e.topic_rel[q] = uint16_t(log.topic_off[q] >= start ? log.topic_off[q] - start : 0);
            e.data_rel = uint16_t(log.data.present ? log.data.off - start : 0);
            e.ev_len = uint16_t(len);
            e.data_len = log.data.present ? log.data.len : 0;
            erecs[rec_first + j] = e;
            if (count_matches && !(sp.has_actor && emitter != sp.actor) && log_matches(r, log, sp.filter)) ++c;
        }
        r.finish();
        if (r.ok() && !oversize) br = BlockRec{uint32_t(RK_TABLE) | (c << 8), rec_first, bits};
    }
}

__global__ __launch_bounds__(256, 4) void k_block_events(const uint8_t* __restrict__ arena, const K1Meta* __restrict__ meta, uint32_t n,
                                                         ScanParams sp, int count_matches, BlockRec* __restrict__ brecs,
                                                         EventRec* __restrict__ erecs, uint32_t cap_events,
                                                         uint32_t* __restrict__ pool_used) {
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = s < n;
    K1Meta m{0, 0, kNoBlock};
    if (live) m = meta[s];
    BlockRec br{RK_WALK, 0, 0};
    Rd r;
    r.init(arena + m.off, live ? m.len : 0u);
    block_events_parse(r, live, m.off, sp, count_matches, erecs, cap_events, pool_parts(n), pool_used, s >> 6, threadIdx.x & 63u, br);
    if (live) brecs[m.id] = br;
}

int launch_block_events(ipcfp_ctx* ctx, hipStream_t stream, const uint8_t* arena, const void* meta_d, uint32_t n,
                        const ipcfp_event_filter_t* filter, int has_actor, uint64_t actor, BlockRec* brecs_d,
                        EventRec* erecs_d, uint32_t cap_events, uint32_t* pool_used_d) {
    if (n == 0) return IPCFP_OK;
    ScanParams sp{};
    if (filter) sp = ScanParams{*filter, actor, has_actor ? 1u : 0u, 0};
    {
        ProfileScope prof(ctx, IPCFP_K_EVENT_SCAN, stream);
        hipLaunchKernelGGL(k_block_events, dim3(div_up(n, 256)), dim3(256), 0, stream, arena, static_cast<const K1Meta*>(meta_d), n, sp,
                           filter ? 1 : 0, brecs_d, erecs_d, cap_events, pool_used_d);
    }
    IPCFP_HIP(ctx, hipGetLastError());
    return IPCFP_OK;
}

}  // namespace ipcfp
