// csrc/host/blockstore.cpp — the witness as a `fvm_ipld_blockstore::Blockstore`
// (trait impls in the reference: src/proofs/common/blockstore.rs:26-39, src/client/blockstore.rs:20-37,
// src/client/cached_blockstore.rs:53-85; `MemoryBlockstore` as the verifiers fill it,
// src/proofs/events/verifier.rs:79-89):
//     get(&Cid) -> Option<Vec<u8>>      ipcfp_witness_get        (an owned copy, as the trait returns)
//     has(&Cid) -> bool                 ipcfp_witness_has        (batched)
//     put_keyed(&Cid, &[u8])            ipcfp_witness_put_keyed  (no hashing — SURVEY.md A.9; an existing CID is replaced)
// so that unmodified fvm_ipld_amt / fvm_ipld_hamt callers can sit on top of the HBM-resident store (bindings/rust/ffi.rs
// implements the trait over these three).  The lookups run on the device through the same CID index the walk
// kernels use; a put re-lays the witness out (blocks are immutable once placed: puts are meant to be batched).
#include <algorithm>
#include <cstring>
#include <memory>
#include <new>
#include <vector>

#include <utility>

#include "../common.h"
#include "../kernels/launch.h"
#include "exec_state.h"

using namespace ipcfp;

namespace ipcfp {
int witness_finish_create(ipcfp_ctx* ctx, ipcfp_witness* w, const uint8_t* raw_bytes_d, const uint64_t* raw_off_d,
                          const uint32_t* len_d_src, const uint8_t* cids_d_src);
}

extern "C" {

int ipcfp_witness_has(ipcfp_ctx_t* ctx, ipcfp_witness_t* w, const uint8_t* cids40, uint64_t n, uint8_t* has,
                      uint32_t* block_ids) {
    if (!ctx || !w || w->ctx != ctx || (n && (!cids40 || (!has && !block_ids)))) return IPCFP_E_INVALID;
    if (n >= 0xffffffffULL) return set_error(ctx, IPCFP_E_UNSUPPORTED, "batch too large");
    if (n == 0) return IPCFP_OK;
    IPCFP_ENTER(ctx);
    DevBuf<CidKey> keys_d;
    DevBuf<uint32_t> ids_d;
    IPCFP_HIP(ctx, keys_d.alloc(n));
    IPCFP_HIP(ctx, ids_d.alloc(n));
    IPCFP_HIP(ctx, hipMemcpyAsync(keys_d.p, cids40, n * IPCFP_CID_SLOT, hipMemcpyHostToDevice, ctx->stream));
    int rc = launch_find_blocks(ctx, witness_view(w), keys_d.p, uint32_t(n), ids_d.p);
    if (rc) return rc;
    std::vector<uint32_t> ids(n);
    IPCFP_HIP(ctx, hipMemcpyAsync(ids.data(), ids_d.p, n * 4, hipMemcpyDeviceToHost, ctx->stream));
    IPCFP_HIP(ctx, sync_stream(ctx, ctx->stream));
    for (uint64_t i = 0; i < n; ++i) {
        if (has) has[i] = ids[i] != 0xffffffffu;
        if (block_ids) block_ids[i] = ids[i];
    }
    return IPCFP_OK;
}

int ipcfp_witness_get(ipcfp_ctx_t* ctx, ipcfp_witness_t* w, const uint8_t* cid40, uint8_t* out, uint64_t cap, uint64_t* len,
                      int* found) {
    if (!ctx || !w || w->ctx != ctx || !cid40 || !len || !found) return IPCFP_E_INVALID;
    *len = 0;
    *found = 0;
    uint32_t id = 0xffffffffu;
    int rc = ipcfp_witness_has(ctx, w, cid40, 1, nullptr, &id);
    if (rc) return rc;
    if (id == 0xffffffffu) return IPCFP_OK;  // Ok(None)
    IPCFP_ENTER(ctx);
    uint64_t off = 0;
    uint32_t blen = 0;
    IPCFP_HIP(ctx, d2h_small(ctx, &off, w->off.p + id, 8, ctx->stream));
    IPCFP_HIP(ctx, d2h_small(ctx, &blen, w->len.p + id, 4, ctx->stream));
    IPCFP_HIP(ctx, sync_stream(ctx, ctx->stream));
    *len = blen;
    *found = 1;
    const uint64_t take = blen < cap ? blen : cap;
    if (out && take) {
        IPCFP_HIP(ctx, hipMemcpyAsync(out, w->arena.p + off, take, hipMemcpyDeviceToHost, ctx->stream));
        IPCFP_HIP(ctx, sync_stream(ctx, ctx->stream));
    }
    return IPCFP_OK;
}

// bytes of located values (ipcfp_value_loc_t from the walk primitives, the scan's matches or
// ipcfp_verify_event_proofs_located): value i goes to out[i * stride .. ), truncated to stride
int ipcfp_witness_read_values(ipcfp_ctx_t* ctx, ipcfp_witness_t* w, const ipcfp_value_loc_t* locs, uint64_t n, uint8_t* out,
                              uint64_t stride) {
    if (!ctx || !w || w->ctx != ctx || (n && (!locs || !out || !stride))) return IPCFP_E_INVALID;
    if (n == 0) return IPCFP_OK;
    IPCFP_ENTER(ctx);
    // ONE gather kernel into a staging buffer and ONE copy back per chunk of values (a 1M-proof bundle used to be 2M tiny
    // copies); the kernel checks every location against its block before it reads (block < n, off + len <= block len)
    if (stride > (uint64_t(1) << 30)) return set_error(ctx, IPCFP_E_INVALID, "stride too large");
    const uint64_t per_chunk = std::max<uint64_t>(1, std::min<uint64_t>(n, (uint64_t(256) << 20) / stride));
    DevBuf<ipcfp_value_loc_t> locs_d;
    DevBuf<uint8_t> stage;
    DevBuf<unsigned long long> bad_d;
    IPCFP_HIP(ctx, locs_d.alloc(per_chunk));
    IPCFP_HIP(ctx, stage.alloc(per_chunk * stride));
    IPCFP_HIP(ctx, bad_d.alloc(1));
    const WitnessView view = witness_view(w);
    for (uint64_t at = 0; at < n; at += per_chunk) {
        const uint64_t m = std::min<uint64_t>(per_chunk, n - at);
        IPCFP_HIP(ctx, hipMemsetAsync(bad_d.p, 0xff, 8, ctx->stream));
        IPCFP_HIP(ctx, hipMemcpyAsync(locs_d.p, locs + at, m * sizeof(ipcfp_value_loc_t), hipMemcpyHostToDevice, ctx->stream));
        int rc = launch_gather_values(ctx, view, locs_d.p, uint32_t(m), stage.p, stride, bad_d.p);
        if (rc) return rc;
        unsigned long long bad = ~0ull;
        IPCFP_HIP(ctx, d2h_small(ctx, &bad, bad_d.p, 8, ctx->stream));
        IPCFP_HIP(ctx, hipMemcpyAsync(out + at * stride, stage.p, m * stride, hipMemcpyDeviceToHost, ctx->stream));
        IPCFP_HIP(ctx, sync_stream(ctx, ctx->stream));
        if (bad != ~0ull)
            return set_error(ctx, IPCFP_E_INVALID, "value %llu (block %u, off %u, len %u) does not lie inside a block of the witness",
                             (unsigned long long)(at + bad), locs[at + bad].block, locs[at + bad].off, locs[at + bad].len);
    }
    return IPCFP_OK;
}

int ipcfp_witness_put_keyed(ipcfp_ctx_t* ctx, ipcfp_witness_t* w, const uint8_t* cids40, const uint8_t* bytes,
                            const uint64_t* off, const uint32_t* len, uint64_t n) {
    if (!ctx || !w || w->ctx != ctx || (n && (!cids40 || !off || !len))) return IPCFP_E_INVALID;
    if (n == 0) return IPCFP_OK;
    if (w->n + n >= 0xffffffffULL) return set_error(ctx, IPCFP_E_UNSUPPORTED, "more than 2^32-2 blocks");
    IPCFP_ENTER(ctx);
    uint64_t nbytes = 0;
    for (uint64_t i = 0; i < n; ++i) {
        if (off[i] + len[i] < off[i]) return set_error(ctx, IPCFP_E_INVALID, "block %llu: off + len overflows", (unsigned long long)i);
        nbytes = std::max<uint64_t>(nbytes, off[i] + len[i]);
    }
    if (nbytes && !bytes) return IPCFP_E_INVALID;
    const uint64_t total = w->n + n;
    // the source table of the new layout: old blocks where they lie in the old arena, new blocks in an upload buffer —
    // both addressed absolutely (the repack kernel adds the offset to a null base)
    DevBuf<uint8_t> fresh, cids_d;
    DevBuf<uint64_t> off_d;
    DevBuf<uint32_t> len_d;
    IPCFP_HIP(ctx, fresh.alloc(nbytes + 16));
    IPCFP_HIP(ctx, off_d.alloc(total));
    IPCFP_HIP(ctx, len_d.alloc(total));
    IPCFP_HIP(ctx, cids_d.alloc(total * IPCFP_CID_SLOT));
    int rc = nbytes ? upload(ctx, fresh.p, bytes, nbytes, ctx->stream) : IPCFP_OK;
    if (rc) return rc;
    std::vector<uint64_t> abs_new(n);
    for (uint64_t i = 0; i < n; ++i) abs_new[i] = reinterpret_cast<uint64_t>(fresh.p) + off[i];
    IPCFP_HIP(ctx, hipMemcpyAsync(off_d.p + w->n, abs_new.data(), n * 8, hipMemcpyHostToDevice, ctx->stream));
    IPCFP_HIP(ctx, hipMemcpyAsync(len_d.p + w->n, len, n * 4, hipMemcpyHostToDevice, ctx->stream));
    IPCFP_HIP(ctx, hipMemcpyAsync(cids_d.p + w->n * IPCFP_CID_SLOT, cids40, n * IPCFP_CID_SLOT, hipMemcpyHostToDevice, ctx->stream));
    if (w->n) {
        IPCFP_HIP(ctx, hipMemcpyAsync(len_d.p, w->len.p, w->n * 4, hipMemcpyDeviceToDevice, ctx->stream));
        IPCFP_HIP(ctx, hipMemcpyAsync(cids_d.p, w->cids.p, w->n * IPCFP_CID_SLOT, hipMemcpyDeviceToDevice, ctx->stream));
        rc = launch_absolute_offsets(ctx, w->off.p, uint32_t(w->n), reinterpret_cast<uint64_t>(w->arena.p), off_d.p);
        if (rc) return rc;
    }
    IPCFP_HIP(ctx, sync_stream(ctx, ctx->stream));  // abs_new (host) was read by the copy above
    // build the new layout into a scratch witness, then move it into the caller's handle
    std::unique_ptr<ipcfp_witness> nw(new (std::nothrow) ipcfp_witness());
    if (!nw) return IPCFP_E_NOMEM;
    nw->ctx = ctx;
    nw->n = total;
    nw->nbytes = w->nbytes + nbytes;
    nw->receipt_lo = w->receipt_lo;
    nw->receipt_hi = w->receipt_hi;
    rc = witness_finish_create(ctx, nw.get(), nullptr, off_d.p, len_d.p, cids_d.p);
    if (rc) return rc;
    IPCFP_HIP(ctx, sync_stream(ctx, ctx->stream));
    IPCFP_HIP(ctx, sync_stream(ctx, ctx->stream_k1));
    w->enum_cache.clear();
    w->table_cache.clear();
    IPCFP_HIP(ctx, hipStreamSynchronize(ctx->stream_aux));  // the block table describes the old arena
    w->bt_valid = w->bt_joined = false;
    w->bt_blocks.release();
    w->bt_events.release();
    w->n = nw->n;
    w->nbytes = nw->nbytes;
    w->arena_bytes = nw->arena_bytes;
    w->max_block_len = nw->max_block_len;
    w->arena.swap(nw->arena);
    w->off.swap(nw->off);
    w->len.swap(nw->len);
    w->cids.swap(nw->cids);
    w->order.swap(nw->order);
    w->k1_meta.swap(nw->k1_meta);
    w->k1_cids.swap(nw->k1_cids);
    w->ok_bits.swap(nw->ok_bits);
    w->cid_status.swap(nw->cid_status);
    w->counters.swap(nw->counters);
    w->index_slots.swap(nw->index_slots);
    w->index_mask = nw->index_mask;
    w->index_done.swap(nw->index_done);
    std::swap(w->index_wgs, nw->index_wgs);
    std::swap(w->index_event, nw->index_event);
    w->uniform_chunks = nw->uniform_chunks;
    return IPCFP_OK;  // nw (the old buffers) is released here
}

}  // extern "C"
