// csrc/host/witness.cpp — the HBM-resident witness store and the K1 entry points.
//
// Reference counterpart: `load_witness_store` → `MemoryBlockstore`
// (src/proofs/events/verifier.rs:79-89, src/proofs/storage/verifier.rs:68-78), rebuilt
// per storage proof by the reference (src/proofs/verifier.rs:19-28); here it is built
// once per bundle and stays resident.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <new>
#include <thread>
#include <vector>

#include "../common.h"
#include "parallel.h"
#include "../kernels/launch.h"
#include "exec_state.h"

using namespace ipcfp;

namespace ipcfp {
int witness_build_index(ipcfp_ctx* ctx, ipcfp_witness* w);  // cid_index.hip
}

namespace ipcfp {

constexpr uint64_t kTailSlack = 256;  // K1 may read one 128-byte chunk past a block's padded end

// IPCFP_TRACE_CREATE=1: host timestamps of the phases of a witness creation on stderr (where the 1-2 ms of a from-host
// creation that are not bytes over PCIe go; tools/gpu_t2_trace.sh)
struct CreateTrace {
    bool on;
    std::chrono::steady_clock::time_point t0, last;
    CreateTrace() {
        static const bool env = [] {
            const char* e = std::getenv("IPCFP_TRACE_CREATE");
            return e && std::atoi(e) != 0;
        }();
        on = env;
        if (on) t0 = last = std::chrono::steady_clock::now();
    }
    void mark(const char* what) {
        if (!on) return;
        const auto now = std::chrono::steady_clock::now();
        std::fprintf(stderr, "[create] %-28s +%8.1f us  (at %8.1f)\n", what, std::chrono::duration<double, std::micro>(now - last).count(),
                     std::chrono::duration<double, std::micro>(now - t0).count());
        last = now;
    }
};

// Shared tail of the constructors: `raw_bytes_d/raw_off_d` hold the caller's layout on
// the device; build the aligned arena (adopting nothing: the witness owns its copy),
// the lane schedule and the CID index.
// `host_bytes` (nullable): the payload is still in HOST memory and crosses PCIe here, into raw_bytes_d, once everything
// that needs only the tables has been queued (the K1 schedule, the arena layout, the CID index run beside the copy).
// `cuts` (nullable; n_cuts + 1 entries each): the caller's blocks are back to back, in order, and blocks
// [cut_block[c], cut_block[c+1]) are the bytes [cut_byte[c], cut_byte[c+1]) — the payload then crosses PCIe piece by piece
// and every piece is re-laid out into the arena while the next one is still on the link.
int witness_finish_create_from(ipcfp_ctx* ctx, ipcfp_witness* w, const uint8_t* raw_bytes_d, const uint64_t* raw_off_d,
                               const uint32_t* len_d_src, const uint8_t* cids_d_src, const uint8_t* host_bytes,
                               uint64_t host_nbytes, const uint64_t* cut_block = nullptr, const uint64_t* cut_byte = nullptr,
                               uint32_t n_cuts = 0) {
    const uint32_t n = uint32_t(w->n);
    CreateTrace tr;
    if (const char* e = std::getenv("IPCFP_EVENT_TABLE")) w->use_event_table = std::atoi(e) != 0;
    IPCFP_HIP(ctx, w->off.alloc(n));
    IPCFP_HIP(ctx, w->len.alloc(n));
    IPCFP_HIP(ctx, w->cids.alloc(size_t(n) * IPCFP_CID_SLOT));
    IPCFP_HIP(ctx, w->order.alloc(n));
    IPCFP_HIP(ctx, w->ok_bits.alloc(div_up(n, 32)));
    IPCFP_HIP(ctx, w->cid_status.alloc(n));
    IPCFP_HIP(ctx, w->counters.alloc(8));
    IPCFP_HIP(ctx, hipMemcpyAsync(w->len.p, len_d_src, size_t(n) * 4, hipMemcpyDeviceToDevice, ctx->stream));
    IPCFP_HIP(ctx, hipMemcpyAsync(w->cids.p, cids_d_src, size_t(n) * IPCFP_CID_SLOT, hipMemcpyDeviceToDevice,
                                  ctx->stream));
    IPCFP_HIP(ctx, hipMemsetAsync(w->counters.p, 0, 8 * sizeof(unsigned long long), ctx->stream));
    IPCFP_HIP(ctx, hipMemsetAsync(w->ok_bits.p, 0, w->ok_bits.bytes() ? w->ok_bits.bytes() : 4, ctx->stream));
    IPCFP_HIP(ctx, hipMemsetAsync(w->cid_status.p, 0, n ? n : 1, ctx->stream));

    // schedule (block ids by chunk count, longest first) → physical layout in schedule order
    DevBuf<uint64_t> scratch, sched_off;
    DevBuf<uint32_t> bins, sched_len;
    IPCFP_HIP(ctx, scratch.alloc(size_t(div_up(n, 1024)) + 2));
    IPCFP_HIP(ctx, bins.alloc(256));
    IPCFP_HIP(ctx, sched_len.alloc(n));
    IPCFP_HIP(ctx, sched_off.alloc(n));
    IPCFP_HIP(ctx, w->k1_meta.alloc(size_t(n) * 2));
    IPCFP_HIP(ctx, w->k1_cids.alloc(size_t(n) * IPCFP_CID_SLOT));
    uint64_t* total_d = scratch.p + div_up(n, 1024) + 1;
    int rc = launch_k1_layout(ctx, w->len.p, n, bins.p, w->order.p, sched_len.p, sched_off.p, total_d, scratch.p,
                              w->off.p, w->k1_meta.p);
    if (rc) return rc;
    rc = launch_gather_cids(ctx, w->order.p, w->cids.p, n, w->k1_cids.p);
    if (rc) return rc;
    tr.mark("allocs + layout queued");
    uint64_t total = 0;
    uint64_t meta0[2] = {0, 0};  // K1Meta of the first lane of the schedule = the longest block: {off, len | id << 32}
    IPCFP_HIP(ctx, d2h_small(ctx, &total, total_d, sizeof total, ctx->stream));
    if (n) IPCFP_HIP(ctx, d2h_small(ctx, meta0, w->k1_meta.p, sizeof meta0, ctx->stream));
    IPCFP_HIP(ctx, sync_stream(ctx, ctx->stream));
    {   // the schedule sorts by 128-byte chunk COUNT (classes capped at 255): the first block bounds every length
        const uint32_t len0 = uint32_t(meta0[1]);
        const uint32_t chunks0 = len0 ? (len0 + 127u) / 128u : 1u;
        w->max_block_len = !n ? 0u : (chunks0 >= 255u ? 0xffffffffu : chunks0 * 128u);
    }
    tr.mark("layout sync");
    w->arena_bytes = total + kTailSlack;
    IPCFP_HIP(ctx, w->arena.alloc(w->arena_bytes));
    IPCFP_HIP(ctx, hipMemsetAsync(w->arena.p + total, 0, kTailSlack, ctx->stream));
    // the CID index needs the CIDs only: its inserts run while the payload crosses PCIe (ipcfp_witness_create)
    rc = witness_build_index(ctx, w);
    if (rc) return rc;
    tr.mark("arena alloc + index queued");
    if (host_bytes && host_nbytes) {
        static const bool ring = [] {
            const char* e = std::getenv("IPCFP_UPLOAD_MODE");
            return e && std::atoi(e) != 0;
        }();
        if (ring) {
            rc = upload(ctx, const_cast<uint8_t*>(raw_bytes_d), host_bytes, host_nbytes, ctx->stream);
            if (rc) return rc;
        } else {  // the runtime's blocking copy (host/upload.cpp), WITHOUT draining the main stream: nothing queued THERE reads or
            // writes raw_bytes_d.  The side streams are another matter — raw_bytes_d is pooled memory, the engine's streams do
            // not synchronise with the NULL stream, and an earlier asynchronous call may still have a kernel on one of them
            // that uses the chunk under its previous owner: they are drained first (idle streams: a few microseconds).
            if (ctx->stream_aux != ctx->stream) IPCFP_HIP(ctx, hipStreamSynchronize(ctx->stream_aux));
            if (ctx->stream_k1 != ctx->stream) IPCFP_HIP(ctx, hipStreamSynchronize(ctx->stream_k1));
            if (ctx->stream_copy) IPCFP_HIP(ctx, hipStreamSynchronize(ctx->stream_copy));
            if (n_cuts > 1 && cut_block && cut_byte) {
                for (uint32_t c = 0; c < n_cuts; ++c) {
                    const uint64_t b0 = cut_byte[c], b1 = cut_byte[c + 1], i0 = cut_block[c], i1 = cut_block[c + 1];
                    if (b1 > b0)
                        IPCFP_HIP(ctx, hipMemcpy(const_cast<uint8_t*>(raw_bytes_d) + b0, host_bytes + b0, b1 - b0, hipMemcpyHostToDevice));
                    // (the blocking copy is over: the piece is in HBM; its re-layout runs on the main stream beside the next copy)
                    if (i1 > i0) {
                        rc = launch_repack(ctx, raw_bytes_d, raw_off_d + i0, w->len.p + i0, w->off.p + i0, uint32_t(i1 - i0), w->arena.p);
                        if (rc) return rc;
                    }
                }
                tr.mark("payload copy + repack, pieces");
                IPCFP_HIP(ctx, sync_stream(ctx, ctx->stream));
                tr.mark("last piece's repack + sync");
                return IPCFP_OK;
            }
            IPCFP_HIP(ctx, hipMemcpy(const_cast<uint8_t*>(raw_bytes_d), host_bytes, host_nbytes, hipMemcpyHostToDevice));
        }
    }
    tr.mark("payload copy");
    rc = launch_repack(ctx, raw_bytes_d, raw_off_d, w->len.p, w->off.p, n, w->arena.p);
    if (rc) return rc;
    IPCFP_HIP(ctx, sync_stream(ctx, ctx->stream));
    tr.mark("repack + sync");
    return IPCFP_OK;
}

int witness_finish_create(ipcfp_ctx* ctx, ipcfp_witness* w, const uint8_t* raw_bytes_d, const uint64_t* raw_off_d,
                          const uint32_t* len_d_src, const uint8_t* cids_d_src) {
    return witness_finish_create_from(ctx, w, raw_bytes_d, raw_off_d, len_d_src, cids_d_src, nullptr, 0);
}

}  // namespace ipcfp

namespace ipcfp {
int k1_flush(ipcfp_ctx* ctx, bool gated) {
    ipcfp_witness* w = ctx->k1_deferred_w;
    if (!w) return IPCFP_OK;
    ctx->k1_deferred_w = nullptr;
    if (gated && ctx->k1_gate && ctx->stream_k1 != ctx->stream) {
        if (!ctx->k1_gate_event) IPCFP_HIP(ctx, hipEventCreateWithFlags(&ctx->k1_gate_event, hipEventDisableTiming));
        IPCFP_HIP(ctx, hipEventRecord(ctx->k1_gate_event, ctx->stream));
        IPCFP_HIP(ctx, hipStreamWaitEvent(ctx->stream_k1, ctx->k1_gate_event, 0));
    }
    return launch_blake2b256_cid(ctx, w->arena.p, w->k1_meta.p, w->k1_cids.p, uint32_t(w->n), w->ok_bits.p, w->cid_status.p,
                                 w->counters.p);
}
}  // namespace ipcfp

extern "C" {

int ipcfp_witness_create(ipcfp_ctx_t* ctx, const uint8_t* bytes, uint64_t nbytes, const uint64_t* off,
                         const uint32_t* len, const uint8_t* cids40, uint64_t n, ipcfp_witness_t** out) {
    if (!ctx || !out) return IPCFP_E_INVALID;
    *out = nullptr;
    if (n && (!off || !len || !cids40)) return set_error(ctx, IPCFP_E_INVALID, "null table pointer");
    if (nbytes && !bytes) return set_error(ctx, IPCFP_E_INVALID, "null bytes pointer");
    if (n >= 0xffffffffull) return set_error(ctx, IPCFP_E_UNSUPPORTED, "more than 2^32-2 blocks");
    // every block inside the buffer (nothing is uploaded before this is known); a million-block table is checked by a
    // few threads — on one it is ≈ 1.5 ms of a 15 ms upload
    uint64_t payload = 0;
    {
        const unsigned T = n >= (1u << 18) ? 4u : 1u;
        uint64_t sum[4] = {0, 0, 0, 0}, bad[4] = {~0ull, ~0ull, ~0ull, ~0ull};
        auto part = [&](unsigned t) {
            const uint64_t lo = n * t / T, hi = n * (t + 1) / T;
            uint64_t acc = 0;
            for (uint64_t i = lo; i < hi; ++i) {
                if (off[i] > nbytes || uint64_t(len[i]) > nbytes - off[i]) {
                    bad[t] = i;
                    break;
                }
                acc += len[i];
            }
            sum[t] = acc;
        };
        (void)run_parts(T, part);  // (host/parallel.h: nothing is thrown across the ABI; `part` itself allocates nothing)
        for (unsigned t = 0; t < T; ++t) {
            if (bad[t] != ~0ull) {
                const uint64_t i = bad[t];  // (the first offender: the parts are in index order)
                return set_error(ctx, IPCFP_E_INVALID, "block %llu [%llu,+%u) lies outside the %llu-byte buffer",
                                 (unsigned long long)i, (unsigned long long)off[i], len[i], (unsigned long long)nbytes);
            }
            payload += sum[t];
        }
    }
    IPCFP_ENTER(ctx);
    std::unique_ptr<ipcfp_witness> w(new (std::nothrow) ipcfp_witness());
    if (!w) return IPCFP_E_NOMEM;
    w->ctx = ctx;
    w->n = n;
    w->nbytes = payload;

    DevBuf<uint8_t> raw_bytes, raw_cids;
    DevBuf<uint64_t> raw_off;
    DevBuf<uint32_t> raw_len;
    IPCFP_HIP(ctx, raw_bytes.alloc(nbytes));
    IPCFP_HIP(ctx, raw_off.alloc(n));
    IPCFP_HIP(ctx, raw_len.alloc(n));
    IPCFP_HIP(ctx, raw_cids.alloc(n * IPCFP_CID_SLOT));
    int rc = IPCFP_OK;
    // tables first; the payload crosses PCIe inside witness_finish_create, beside the kernels that need only the tables.
    // (Two blocking copies at once — the payload on a thread of its own — deliver 38 GB/s together where one delivers 56:
    // tools/ubench/h2d_paths.)
    if (n) {
        rc = upload(ctx, raw_len.p, len, n * 4, ctx->stream);
        if (!rc) rc = upload(ctx, raw_off.p, off, n * 8, ctx->stream);
        if (!rc) rc = upload(ctx, raw_cids.p, cids40, n * IPCFP_CID_SLOT, ctx->stream);
    }
    if (!rc) rc = witness_finish_create_from(ctx, w.get(), raw_bytes.p, raw_off.p, raw_len.p, raw_cids.p, bytes, nbytes);
    if (rc) return rc;
    *out = w.release();
    return IPCFP_OK;
}

// The tables in transport form (include/ipcfp.h): no offset table, 32-byte digests + one prefix instead of 40-byte slots.
int ipcfp_witness_create_packed(ipcfp_ctx_t* ctx, const uint8_t* bytes, uint64_t nbytes, const uint32_t* len,
                                const uint8_t* digests32, uint64_t n, const uint8_t* cid_prefix, uint32_t prefix_len,
                                const uint32_t* esc_index, const uint8_t* esc_cids40, uint64_t n_esc,
                                ipcfp_witness_t** out) {
    if (!ctx || !out) return IPCFP_E_INVALID;
    *out = nullptr;
    if (n && (!len || !digests32)) return set_error(ctx, IPCFP_E_INVALID, "null table pointer");
    if (nbytes && !bytes) return set_error(ctx, IPCFP_E_INVALID, "null bytes pointer");
    if (prefix_len > 8 || (prefix_len && !cid_prefix)) return set_error(ctx, IPCFP_E_INVALID, "CID prefix longer than 8 bytes");
    if (n_esc && (!esc_index || !esc_cids40)) return set_error(ctx, IPCFP_E_INVALID, "null escape table");
    if (n >= 0xffffffffull || n_esc > n) return set_error(ctx, IPCFP_E_UNSUPPORTED, "more than 2^32-2 blocks");
    for (uint64_t e = 0; e < n_esc; ++e)
        if (esc_index[e] >= n || (e && esc_index[e] <= esc_index[e - 1]))
            return set_error(ctx, IPCFP_E_INVALID, "escape %llu: index out of range or not ascending", (unsigned long long)e);
    // the blocks fill the buffer exactly (nothing is uploaded before this is known)
    uint64_t payload = 0;
    constexpr unsigned kPieces = 4;  // the pieces the payload crosses PCIe in = the parts the lengths are summed in
    uint64_t cut_block[kPieces + 1] = {}, cut_byte[kPieces + 1] = {};
    const unsigned T = n >= (1u << 18) ? kPieces : 1u;
    {
        uint64_t sum[kPieces] = {};
        auto part = [&](unsigned t) {
            uint64_t acc = 0;
            for (uint64_t i = n * t / T, hi = n * (t + 1) / T; i < hi; ++i) acc += len[i];
            sum[t] = acc;
        };
        // (no exception may cross the C ABI: a thread that cannot be made — std::system_error, std::bad_alloc — leaves
        // its part to this thread, as shard_host.cpp's parallel_ranges does)
        std::thread pool[kPieces];
        bool started[kPieces] = {};
        for (unsigned t = 1; t < T; ++t) {
            try {
                pool[t] = std::thread(part, t);
                started[t] = true;
            } catch (...) {
            }
        }
        part(0);
        for (unsigned t = 1; t < T; ++t) {
            if (started[t]) pool[t].join();
            else part(t);
        }
        for (unsigned t = 0; t < T; ++t) {
            cut_block[t] = n * t / T;
            cut_byte[t] = payload;
            payload += sum[t];
        }
        cut_block[T] = n;
        cut_byte[T] = payload;
    }
    CreateTrace tr;
    tr.mark("(host) lengths summed");
    if (payload != nbytes)
        return set_error(ctx, IPCFP_E_INVALID, "the block lengths add up to %llu bytes, the buffer holds %llu",
                         (unsigned long long)payload, (unsigned long long)nbytes);
    IPCFP_ENTER(ctx);
    std::unique_ptr<ipcfp_witness> w(new (std::nothrow) ipcfp_witness());
    if (!w) return IPCFP_E_NOMEM;
    w->ctx = ctx;
    w->n = n;
    w->nbytes = payload;

    DevBuf<uint8_t> raw_bytes, raw_cids, dig, esc_c;
    DevBuf<uint64_t> raw_off, scan_scratch;
    DevBuf<uint32_t> raw_len, esc_i;
    IPCFP_HIP(ctx, raw_bytes.alloc(nbytes));
    IPCFP_HIP(ctx, raw_off.alloc(n));
    IPCFP_HIP(ctx, raw_len.alloc(n));
    IPCFP_HIP(ctx, raw_cids.alloc(n * IPCFP_CID_SLOT));
    IPCFP_HIP(ctx, dig.alloc(n * 32));
    IPCFP_HIP(ctx, scan_scratch.alloc(size_t(div_up(uint32_t(n), 1024)) + 2));
    tr.mark("raw allocs");
    int rc = IPCFP_OK;
    if (n) {
        rc = upload(ctx, raw_len.p, len, n * 4, ctx->stream);
        if (!rc) rc = upload(ctx, dig.p, digests32, n * 32, ctx->stream);
        if (!rc && n_esc) {
            IPCFP_HIP(ctx, esc_i.alloc(n_esc));
            IPCFP_HIP(ctx, esc_c.alloc(n_esc * IPCFP_CID_SLOT));
            rc = upload(ctx, esc_i.p, esc_index, n_esc * 4, ctx->stream);
            if (!rc) rc = upload(ctx, esc_c.p, esc_cids40, n_esc * IPCFP_CID_SLOT, ctx->stream);
        }
        if (!rc) rc = launch_tight_offsets(ctx, raw_len.p, uint32_t(n), raw_off.p, scan_scratch.p + div_up(uint32_t(n), 1024) + 1, scan_scratch.p);
        if (!rc) rc = launch_expand_cids(ctx, dig.p, uint32_t(n), cid_prefix, prefix_len, esc_i.p, esc_c.p, uint32_t(n_esc), raw_cids.p);
    }
    tr.mark("tables up + expand queued");
    if (!rc) rc = witness_finish_create_from(ctx, w.get(), raw_bytes.p, raw_off.p, raw_len.p, raw_cids.p, bytes, nbytes, cut_block, cut_byte, T);
    tr.mark("finish_create");
    if (rc) {
        (void)hipStreamSynchronize(ctx->stream);  // (kernels queued above may still read the tables going back to the pool)
        return rc;
    }
    *out = w.release();
    return IPCFP_OK;
}

int ipcfp_witness_create_device(ipcfp_ctx_t* ctx, const void* bytes_d, uint64_t nbytes, const void* off_d,
                                const void* len_d, const void* cids40_d, uint64_t n, ipcfp_witness_t** out) {
    if (!ctx || !out) return IPCFP_E_INVALID;
    *out = nullptr;
    if (n && (!off_d || !len_d || !cids40_d)) return set_error(ctx, IPCFP_E_INVALID, "null table pointer");
    if (n >= 0xffffffffull) return set_error(ctx, IPCFP_E_UNSUPPORTED, "more than 2^32-2 blocks");
    IPCFP_ENTER(ctx);
    std::unique_ptr<ipcfp_witness> w(new (std::nothrow) ipcfp_witness());
    if (!w) return IPCFP_E_NOMEM;
    w->ctx = ctx;
    w->n = n;
    w->nbytes = nbytes;  // upper bound; the device path does not sum lengths on the host
    int rc = witness_finish_create(ctx, w.get(), static_cast<const uint8_t*>(bytes_d), static_cast<const uint64_t*>(off_d),
                                   static_cast<const uint32_t*>(len_d), static_cast<const uint8_t*>(cids40_d));
    if (rc) return rc;
    *out = w.release();
    return IPCFP_OK;
}

void ipcfp_witness_destroy(ipcfp_witness_t* w) {
    if (!w) return;
    if (w->ctx) {
        if (w->ctx->k1_deferred_w == w) w->ctx->k1_deferred_w = nullptr;  // (nobody can ask for its results any more)
        (void)hipSetDevice(w->ctx->device);
        (void)hipStreamSynchronize(w->ctx->stream);
        (void)hipStreamSynchronize(w->ctx->stream_k1);
        (void)hipStreamSynchronize(w->ctx->stream_aux);
    }
    delete w;
}

uint64_t ipcfp_witness_block_count(const ipcfp_witness_t* w) { return w ? w->n : 0; }
uint64_t ipcfp_witness_byte_count(const ipcfp_witness_t* w) { return w ? w->nbytes : 0; }

static int k1_launch_now(ipcfp_ctx_t* ctx, ipcfp_witness_t* w);

int ipcfp_witness_verify_cids_async(ipcfp_ctx_t* ctx, ipcfp_witness_t* w) {
    if (!ctx || !w || w->ctx != ctx) return IPCFP_E_INVALID;
    IPCFP_ENTER(ctx);
    if (ctx->k1_defer) {  // noted; queued by the next event-verify call, or by whoever needs it first (k1_flush)
        if (ctx->k1_deferred_w && ctx->k1_deferred_w != w) {
            int rc = ipcfp::k1_flush(ctx);
            if (rc) return rc;
        }
        ctx->k1_deferred_w = w;
        return IPCFP_OK;
    }
    return k1_launch_now(ctx, w);
}

static int k1_launch_now(ipcfp_ctx_t* ctx, ipcfp_witness_t* w) {
    if (ctx->k1_after_be && w->use_event_table && ctx->stream_aux != ctx->stream && ctx->stream_k1 != ctx->stream) {
        // K1 and the block-order event parse are both bound by instruction issue: side by side each takes about as long
        // as the two in a row.  Only the parse has consumers waiting for it (the receipts' event records, then the verify
        // kernel), so it goes first with the chip's side share to itself and K1 fills in behind it.
        int rc = ctx->has_scan_hint ? block_table_prefetch(ctx, w, &ctx->scan_hint.filter, int(ctx->scan_hint.has_actor), ctx->scan_hint.actor)
                                    : block_table_prefetch(ctx, w, nullptr, 0, 0);
        if (rc) return rc;
        if (w->bt_valid && !w->bt_joined) IPCFP_HIP(ctx, hipStreamWaitEvent(ctx->stream_k1, ctx->aux_event, 0));
    }
    return launch_blake2b256_cid(ctx, w->arena.p, w->k1_meta.p, w->k1_cids.p, uint32_t(w->n), w->ok_bits.p,
                                 w->cid_status.p, w->counters.p);
}

int ipcfp_witness_verify_cids(ipcfp_ctx_t* ctx, ipcfp_witness_t* w, uint8_t* status, uint64_t* n_bad) {
    if (!ctx || !w || w->ctx != ctx) return IPCFP_E_INVALID;
    IPCFP_ENTER(ctx);
    int rc = ipcfp_witness_verify_cids_async(ctx, w);
    if (rc) return rc;
    if ((rc = ipcfp::k1_flush(ctx))) return rc;
    unsigned long long bad = 0;
    if (status && w->n)
        IPCFP_HIP(ctx, hipMemcpyAsync(status, w->cid_status.p, w->n, hipMemcpyDeviceToHost, ctx->stream_k1));
    IPCFP_HIP(ctx, d2h_small(ctx, &bad, w->counters.p, sizeof bad, ctx->stream_k1));
    IPCFP_HIP(ctx, sync_stream(ctx, ctx->stream_k1));
    if (n_bad) *n_bad = bad;
    return IPCFP_OK;
}

// Results of the last ipcfp_witness_verify_cids_async: waits for K1's stream, copies back.
int ipcfp_witness_cid_results(ipcfp_ctx_t* ctx, ipcfp_witness_t* w, uint8_t* status, uint64_t* n_bad) {
    if (!ctx || !w || w->ctx != ctx) return IPCFP_E_INVALID;
    IPCFP_ENTER(ctx);
    if (int rc = ipcfp::k1_flush(ctx)) return rc;
    unsigned long long bad = 0;
    if (status && w->n)
        IPCFP_HIP(ctx, hipMemcpyAsync(status, w->cid_status.p, w->n, hipMemcpyDeviceToHost, ctx->stream_k1));
    IPCFP_HIP(ctx, d2h_small(ctx, &bad, w->counters.p, sizeof bad, ctx->stream_k1));
    IPCFP_HIP(ctx, sync_stream(ctx, ctx->stream_k1));
    if (n_bad) *n_bad = bad;
    return IPCFP_OK;
}

int ipcfp_witness_rebuild_index(ipcfp_ctx_t* ctx, ipcfp_witness_t* w) {
    if (!ctx || !w || w->ctx != ctx) return IPCFP_E_INVALID;
    IPCFP_ENTER(ctx);
    if (int rc = ipcfp::k1_flush(ctx)) return rc;
    IPCFP_HIP(ctx, sync_stream(ctx, ctx->stream));
    w->enum_cache.clear();  // enumerations and event tables are derived from the index
    w->table_cache.clear();
    // the block table is not (it is a function of the arena alone), but a rebuild means "start over": it goes as well
    if (w->bt_valid && !w->bt_joined) IPCFP_HIP(ctx, hipStreamSynchronize(ctx->stream_aux));
    w->bt_valid = w->bt_joined = false;
    return witness_build_index(ctx, w);
}

void* ipcfp_witness_cid_bitmap_device(ipcfp_witness_t* w) { return w ? w->ok_bits.p : nullptr; }
void* ipcfp_witness_cid_status_device(ipcfp_witness_t* w) { return w ? w->cid_status.p : nullptr; }

// ---- batch hashes ---------------------------------------------------------
namespace {
enum HashKind { H_B2B, H_KECCAK, H_SHA256 };

int hash_batch(ipcfp_ctx_t* ctx, HashKind kind, const uint8_t* bytes, uint64_t nbytes, const uint64_t* off,
               const uint32_t* len, uint64_t n, uint8_t* out32) {
    if (!ctx) return IPCFP_E_INVALID;
    if (n == 0) return IPCFP_OK;
    if (!off || !len || !out32 || (nbytes && !bytes)) return set_error(ctx, IPCFP_E_INVALID, "null pointer");
    if (n >= 0xffffffffull) return set_error(ctx, IPCFP_E_UNSUPPORTED, "more than 2^32-2 messages");
    for (uint64_t i = 0; i < n; ++i)
        if (off[i] > nbytes || uint64_t(len[i]) > nbytes - off[i])
            return set_error(ctx, IPCFP_E_INVALID, "message %llu outside the buffer", (unsigned long long)i);
    IPCFP_ENTER(ctx);
    DevBuf<uint8_t> b, o;
    DevBuf<uint64_t> off_d;
    DevBuf<uint32_t> len_d;
    IPCFP_HIP(ctx, b.alloc(nbytes + kTailSlack));
    IPCFP_HIP(ctx, off_d.alloc(n));
    IPCFP_HIP(ctx, len_d.alloc(n));
    IPCFP_HIP(ctx, o.alloc(n * 32));
    if (nbytes) IPCFP_HIP(ctx, hipMemcpyAsync(b.p, bytes, nbytes, hipMemcpyHostToDevice, ctx->stream));
    IPCFP_HIP(ctx, hipMemcpyAsync(off_d.p, off, n * 8, hipMemcpyHostToDevice, ctx->stream));
    IPCFP_HIP(ctx, hipMemcpyAsync(len_d.p, len, n * 4, hipMemcpyHostToDevice, ctx->stream));
    int rc = IPCFP_OK;
    if (kind == H_B2B) {
        // Blake2b wants 16-byte aligned blocks: re-lay out on the device, then hash in chunk-count order.
        DevBuf<uint64_t> new_off, scratch, sched_off, meta;
        DevBuf<uint32_t> order, bins, sched_len;
        DevBuf<uint8_t> arena;
        IPCFP_HIP(ctx, new_off.alloc(n));
        IPCFP_HIP(ctx, scratch.alloc(size_t(div_up(n, 1024)) + 2));
        IPCFP_HIP(ctx, order.alloc(n));
        IPCFP_HIP(ctx, bins.alloc(256));
        IPCFP_HIP(ctx, sched_len.alloc(n));
        IPCFP_HIP(ctx, sched_off.alloc(n));
        IPCFP_HIP(ctx, meta.alloc(n * 2));
        uint64_t* total_d = scratch.p + div_up(n, 1024) + 1;
        rc = launch_k1_layout(ctx, len_d.p, uint32_t(n), bins.p, order.p, sched_len.p, sched_off.p, total_d, scratch.p,
                              new_off.p, meta.p);
        if (rc) return rc;
        uint64_t total = 0;
        IPCFP_HIP(ctx, d2h_small(ctx, &total, total_d, 8, ctx->stream));
        IPCFP_HIP(ctx, sync_stream(ctx, ctx->stream));
        IPCFP_HIP(ctx, arena.alloc(total + kTailSlack));
        rc = launch_repack(ctx, b.p, off_d.p, len_d.p, new_off.p, uint32_t(n), arena.p);
        if (rc) return rc;
        rc = launch_blake2b256_raw(ctx, arena.p, meta.p, uint32_t(n), o.p);
        if (rc) return rc;
        IPCFP_HIP(ctx, hipMemcpyAsync(out32, o.p, n * 32, hipMemcpyDeviceToHost, ctx->stream));
        IPCFP_HIP(ctx, sync_stream(ctx, ctx->stream));
        return IPCFP_OK;
    }
    rc = (kind == H_KECCAK) ? launch_keccak256(ctx, b.p, off_d.p, len_d.p, uint32_t(n), o.p)
                            : launch_sha256(ctx, b.p, off_d.p, len_d.p, uint32_t(n), o.p);
    if (rc) return rc;
    IPCFP_HIP(ctx, hipMemcpyAsync(out32, o.p, n * 32, hipMemcpyDeviceToHost, ctx->stream));
    IPCFP_HIP(ctx, sync_stream(ctx, ctx->stream));
    return IPCFP_OK;
}
}  // namespace

int ipcfp_blake2b256_batch(ipcfp_ctx_t* ctx, const uint8_t* bytes, uint64_t nbytes, const uint64_t* off,
                           const uint32_t* len, uint64_t n, uint8_t* out32) {
    return hash_batch(ctx, H_B2B, bytes, nbytes, off, len, n, out32);
}
int ipcfp_keccak256_batch(ipcfp_ctx_t* ctx, const uint8_t* bytes, uint64_t nbytes, const uint64_t* off,
                          const uint32_t* len, uint64_t n, uint8_t* out32) {
    return hash_batch(ctx, H_KECCAK, bytes, nbytes, off, len, n, out32);
}
int ipcfp_sha256_batch(ipcfp_ctx_t* ctx, const uint8_t* bytes, uint64_t nbytes, const uint64_t* off,
                       const uint32_t* len, uint64_t n, uint8_t* out32) {
    return hash_batch(ctx, H_SHA256, bytes, nbytes, off, len, n, out32);
}

}  // extern "C"
