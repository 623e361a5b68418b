// csrc/host/primitives.cpp — C-ABI entry points of the batch path-walk primitives.
#include <cstdlib>
#include <cstring>

#include "../common.h"
#include "../kernels/hamt_table.h"
#include "../kernels/launch.h"

using namespace ipcfp;

namespace ipcfp {

WitnessView witness_view(const ipcfp_witness* w, uint32_t* touched_bits) {
    WitnessView v;
    v.arena = w->arena.p;
    v.off = w->off.p;
    v.len = w->len.p;
    v.cids = w->cids.p;
    v.slots = w->index_slots.p;
    v.mask = w->index_mask;
    v.n = uint32_t(w->n);
    v.touched = touched_bits;
    return v;
}

CidKey key_from_slot(const uint8_t* slot40) {
    CidKey k;
    std::memcpy(k.w, slot40, 40);
    return k;
}

// a context-owned buffer of at least `need` bytes (hipFree of the old one waits for the device: nothing still reads it)
static hipError_t grow(void*& p, size_t& cap, size_t need) {
    if (need <= cap) return hipSuccess;
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
    const size_t want = need + need / 4;
    hipError_t e = hipMalloc(&p, want);
    if (e == hipSuccess) cap = want;
    return e;
}

// K7 for a batch.  Default for a batch of ≥ 1024 queries: LEVEL BY LEVEL (kernels/hamt_levels.hip) — every node the batch
// visits is decoded once, a query is SHA-256 + one record per level — with the per-query walker (k_hamt_get) behind it
// for whatever that leaves pending.  IPCFP_HAMT_LEVELS=0: the walker alone (round 3's path; small batches take it
// anyway).  IPCFP_HAMT_TABLE=1 (A/B measurements): tabulate EVERY block of the witness first (kernels/hamt_table.h).
int hamt_get_batch(ipcfp_ctx* ctx, ipcfp_witness* w, const CidKey& root, uint32_t bit_width, int vkind, const uint8_t* keys_d,
                   const uint32_t* key_off_d, const uint32_t* key_len_d, uint32_t n, uint8_t* status_d, void* loc_d) {
    const int forced_table = ctx->hamt_table, levels_mode = ctx->hamt_levels;
    const uint32_t kbit = hamt_kind_bit(vkind);
    ProfileScope prof(ctx, IPCFP_K_HAMT_GET);
    const WitnessView view = witness_view(w);
    if (kbit && forced_table == 1) {
        IPCFP_HIP(ctx, grow(ctx->hamt_recs, ctx->hamt_recs_bytes, size_t(w->n) * sizeof(HamtNodeRec)));
        int rc = launch_hamt_node_table(ctx, w->arena.p, w->k1_meta.p, uint32_t(w->n), kbit, ctx->hamt_recs);
        if (rc) return rc;
        return launch_hamt_get_table(ctx, view, ctx->hamt_recs, root, bit_width, vkind, keys_d, key_off_d, key_len_d, n, status_d, loc_d);
    }
    const bool by_levels = levels_mode != 0 && bit_width >= 1 && bit_width <= 8 && w->n > 0 && (n >= 1024 || levels_mode > 0);
    if (!by_levels) return launch_hamt_get(ctx, view, root, bit_width, vkind, keys_d, key_off_d, key_len_d, n, status_d, loc_d);
    // how deep can the tree be?  A HAMT of B blocks with fan-out 2^bw has about log(B) / bw interior levels; two more
    // for uneven buckets.  A deeper tree is not an error: its queries stay pending and the walker finishes them.
    uint32_t levels = 2;
    for (uint64_t reach = 1; reach < w->n && levels < 14; reach <<= bit_width) ++levels;
    if (levels_mode > 0) levels = uint32_t(levels_mode);
    IPCFP_HIP(ctx, grow(ctx->hamt_recs, ctx->hamt_recs_bytes, size_t(w->n) * sizeof(HamtNodeRec)));
    IPCFP_HIP(ctx, grow(ctx->hamt_scratch, ctx->hamt_scratch_bytes, hamt_levels_scratch_words(n, uint32_t(w->n), levels) * 4));
    // entry tables: one per visited node, in work-list order (a level's list holds at most min(n, blocks) nodes; the upper
    // levels' lists are short) — what lies beyond the pool keeps the reader's bucket search
    const uint32_t etab_cap = uint32_t(std::min<uint64_t>(3ull * n + 4096, w->n));
    IPCFP_HIP(ctx, grow(ctx->hamt_etabs, ctx->hamt_etabs_bytes, size_t(etab_cap) * sizeof(HamtEntryTab)));
    int rc = launch_hamt_get_levels(ctx, view, root, bit_width, vkind, keys_d, key_off_d, key_len_d, n, status_d, loc_d, levels,
                                    static_cast<uint32_t*>(ctx->hamt_scratch), ctx->hamt_recs, /*coop=*/ctx->hamt_coop != 0, ctx->hamt_etabs,
                                    etab_cap);
    if (rc) return rc;
    return launch_hamt_get(ctx, view, root, bit_width, vkind, keys_d, key_off_d, key_len_d, n, status_d, loc_d, /*pending_only=*/1);
}

}  // namespace ipcfp

extern "C" {

int ipcfp_amt_get(ipcfp_ctx_t* ctx, ipcfp_witness_t* w, const uint8_t* root_cid40, int version, int value_kind,
                  const uint64_t* index, uint64_t n, ipcfp_status_t* status, ipcfp_value_loc_t* loc) {
    if (!ctx || !w || w->ctx != ctx || !root_cid40 || (n && (!index || !status))) return IPCFP_E_INVALID;
    if (version != 0 && version != 3) return set_error(ctx, IPCFP_E_INVALID, "AMT version must be 0 or 3");
    if (n >= 0xffffffffULL) return set_error(ctx, IPCFP_E_UNSUPPORTED, "batch too large");
    if (n == 0) return IPCFP_OK;
    IPCFP_ENTER(ctx);
    DevBuf<uint64_t> idx;
    DevBuf<uint8_t> st;
    DevBuf<ipcfp_value_loc_t> lc;
    IPCFP_HIP(ctx, idx.alloc(n));
    IPCFP_HIP(ctx, st.alloc(n));
    IPCFP_HIP(ctx, lc.alloc(n));
    IPCFP_HIP(ctx, hipMemcpyAsync(idx.p, index, n * 8, hipMemcpyHostToDevice, ctx->stream));
    int rc = launch_amt_get(ctx, witness_view(w), key_from_slot(root_cid40), version, value_kind, idx.p, uint32_t(n),
                            st.p, lc.p);
    if (rc) return rc;
    IPCFP_HIP(ctx, hipMemcpyAsync(status, st.p, n, hipMemcpyDeviceToHost, ctx->stream));
    if (loc) IPCFP_HIP(ctx, hipMemcpyAsync(loc, lc.p, n * sizeof(ipcfp_value_loc_t), hipMemcpyDeviceToHost, ctx->stream));
    IPCFP_HIP(ctx, sync_stream(ctx, ctx->stream));
    return IPCFP_OK;
}

int ipcfp_hamt_get(ipcfp_ctx_t* ctx, ipcfp_witness_t* w, const uint8_t* root_cid40, uint32_t bit_width,
                   int value_kind, const uint8_t* keys, const uint32_t* key_off, const uint32_t* key_len, uint64_t n,
                   ipcfp_status_t* status, ipcfp_value_loc_t* loc) {
    if (!ctx || !w || w->ctx != ctx || !root_cid40 || (n && (!keys || !key_off || !key_len || !status)))
        return IPCFP_E_INVALID;
    if (n >= 0xffffffffULL) return set_error(ctx, IPCFP_E_UNSUPPORTED, "batch too large");
    if (n == 0) return IPCFP_OK;
    uint64_t kbytes = 0;
    for (uint64_t i = 0; i < n; ++i) {
        const uint64_t end = uint64_t(key_off[i]) + key_len[i];
        if (end > kbytes) kbytes = end;
    }
    IPCFP_ENTER(ctx);
    DevBuf<uint8_t> kb, st;
    DevBuf<uint32_t> ko, kl;
    DevBuf<ipcfp_value_loc_t> lc;
    IPCFP_HIP(ctx, kb.alloc(kbytes + 16));
    IPCFP_HIP(ctx, ko.alloc(n));
    IPCFP_HIP(ctx, kl.alloc(n));
    IPCFP_HIP(ctx, st.alloc(n));
    IPCFP_HIP(ctx, lc.alloc(n));
    if (kbytes) IPCFP_HIP(ctx, hipMemcpyAsync(kb.p, keys, kbytes, hipMemcpyHostToDevice, ctx->stream));
    IPCFP_HIP(ctx, hipMemcpyAsync(ko.p, key_off, n * 4, hipMemcpyHostToDevice, ctx->stream));
    IPCFP_HIP(ctx, hipMemcpyAsync(kl.p, key_len, n * 4, hipMemcpyHostToDevice, ctx->stream));
    int rc = hamt_get_batch(ctx, w, key_from_slot(root_cid40), bit_width, value_kind, kb.p, ko.p, kl.p, uint32_t(n), st.p, lc.p);
    if (rc) return rc;
    IPCFP_HIP(ctx, hipMemcpyAsync(status, st.p, n, hipMemcpyDeviceToHost, ctx->stream));
    if (loc) IPCFP_HIP(ctx, hipMemcpyAsync(loc, lc.p, n * sizeof(ipcfp_value_loc_t), hipMemcpyDeviceToHost, ctx->stream));
    IPCFP_HIP(ctx, sync_stream(ctx, ctx->stream));
    return IPCFP_OK;
}

// ipcfp_hamt_get with every buffer already in HBM (keys_d: the concatenated key bytes, readable up to the last key's
// end + 16 bytes of slack; key_off_d / key_len_d: u32[n]; status_d: u8[n]; loc_d: ipcfp_value_loc_t[n] or null).  What a
// multi-GPU host calls for its query-index range before it all-gathers the status bytes.
// Asynchronous: the results are complete after ipcfp_ctx_sync (or any later synchronising call on this context).
int ipcfp_hamt_get_device(ipcfp_ctx_t* ctx, ipcfp_witness_t* w, const uint8_t* root_cid40, uint32_t bit_width, int value_kind,
                          const void* keys_d, const void* key_off_d, const void* key_len_d, uint64_t n, void* status_d,
                          void* loc_d) {
    if (!ctx || !w || w->ctx != ctx || !root_cid40 || (n && (!keys_d || !key_off_d || !key_len_d || !status_d)))
        return IPCFP_E_INVALID;
    if (n >= 0xffffffffULL) return set_error(ctx, IPCFP_E_UNSUPPORTED, "batch too large");
    if (n == 0) return IPCFP_OK;
    IPCFP_ENTER(ctx);
    return hamt_get_batch(ctx, w, key_from_slot(root_cid40), bit_width, value_kind, static_cast<const uint8_t*>(keys_d),
                          static_cast<const uint32_t*>(key_off_d), static_cast<const uint32_t*>(key_len_d), uint32_t(n),
                          static_cast<uint8_t*>(status_d), loc_d);
}

}  // extern "C"
