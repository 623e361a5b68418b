// csrc/host/shard.cpp — row (e) of SURVEY.md §8: one proof batch over the GPUs of a node.
//
// The reference verifies a bundle sequentially (src/proofs/verifier.rs:19-28,49-54;
// src/proofs/events/verifier.rs:62-71).  Given a read-only witness every proof is independent, so:
//   cfg 2 (CID batch)            block-index range per rank, each rank holds only its blocks
//   cfg 3 (one tipset)           receipt-index range per rank: the rank's witness is its receipts' events AMTs,
//                                the receipts-AMT nodes on the paths to them, and — replicated — the headers,
//                                TxMeta blocks and message AMTs of the parents (the execution order is global)
//   cfg 4/5 (HAMT gets, storage) query-index range per rank over a replicated state tree
// and ONE collective closes a step: an all-gather of the per-rank verdict bytes / match bitmaps.  The collective
// is RCCL's ncclAllGather, called directly (librccl.so.1 is resolved at run time: a single-GPU host needs no RCCL).
#include <dlfcn.h>

#include <cstring>
#include <memory>
#include <new>
#include <vector>

#include "../common.h"
#include "../kernels/claims_dev.h"
#include "../kernels/amt_enum.h"
#include "../kernels/event_table.h"
#include "../kernels/tipset_ctx.h"
#include "../kernels/types_dev.h"
#include "../kernels/launch.h"
#include "exec_state.h"
#include "tipset_wide.h"

using namespace ipcfp;

namespace ipcfp {
int witness_finish_create(ipcfp_ctx* ctx, ipcfp_witness* w, const uint8_t* raw_bytes_d, const uint64_t* raw_off_d,
                          const uint32_t* len_d_src, const uint8_t* cids_d_src);
}

// ---- RCCL, resolved at run time ------------------------------------------------------------------------------
namespace {

struct NcclUniqueId {
    char internal[128];
};
using nccl_comm_t = void*;
constexpr int kNcclUint8 = 1;  // ncclUint8 (rccl.h: ncclInt8 = 0, ncclUint8 = 1)

struct Rccl {
    void* handle = nullptr;
    int (*GetUniqueId)(NcclUniqueId*) = nullptr;
    int (*CommInitRank)(nccl_comm_t*, int, NcclUniqueId, int) = nullptr;
    int (*CommDestroy)(nccl_comm_t) = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, nccl_comm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    std::string error;
};

Rccl& rccl() {
    static Rccl r;
    static bool tried = false;
    if (tried) return r;
    tried = true;
    // the copy already in the process first (a host that also runs torch.distributed has one), then the system's
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : names)
        if ((r.handle = dlopen(n, RTLD_NOW | RTLD_NOLOAD))) break;
    if (!r.handle)
        for (const char* n : names)
            if ((r.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
    if (!r.handle) {
        r.error = std::string("librccl.so.1 not found: ") + (dlerror() ? dlerror() : "");
        return r;
    }
    r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(dlsym(r.handle, "ncclGetUniqueId"));
    r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(dlsym(r.handle, "ncclCommInitRank"));
    r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(dlsym(r.handle, "ncclCommDestroy"));
    r.AllGather = reinterpret_cast<decltype(r.AllGather)>(dlsym(r.handle, "ncclAllGather"));
    r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(dlsym(r.handle, "ncclGetErrorString"));
    if (!r.GetUniqueId || !r.CommInitRank || !r.CommDestroy || !r.AllGather) {
        r.error = "librccl.so.1 lacks ncclGetUniqueId / ncclCommInitRank / ncclCommDestroy / ncclAllGather";
        r.handle = nullptr;
    }
    return r;
}

const char* rccl_err(int code) {
    Rccl& r = rccl();
    return r.GetErrorString ? r.GetErrorString(code) : "RCCL error";
}

}  // namespace

struct ipcfp_comm {
    ipcfp_ctx* ctx = nullptr;
    nccl_comm_t comm = nullptr;
    int n_ranks = 1, rank = 0;
};

extern "C" {

void ipcfp_shard_range(uint64_t n, uint32_t n_shards, uint32_t shard, uint64_t* lo, uint64_t* hi) {
    if (n_shards == 0) n_shards = 1;
    if (shard >= n_shards) shard = n_shards - 1;
    // floor(n * s / G) without overflow: q * s + (r * s) / G with n = q * G + r
    const uint64_t q = n / n_shards, r = n % n_shards;
    if (lo) *lo = q * shard + (r * shard) / n_shards;
    if (hi) *hi = q * (shard + 1) + (r * (uint64_t(shard) + 1)) / n_shards;
}

// The part of a plan every shard shares, walked with the recorder `rec`: child header (→ receipts root), parent
// headers, TxMeta, message AMTs (the execution order is global), then `Amtv0::load` of the receipts root for its count.
// *status_out != TRUE: the traversal failed there and nothing else is valid.
static int plan_replicated(ipcfp_ctx* ctx, const WitnessView& rec, const uint8_t* parent_cids40, uint32_t n_parents,
                           const uint8_t* child_cid40, TipsetCtxDev& tc, uint64_t& count, ipcfp_status_t* status_out,
                           uint64_t* capacity = nullptr) {
    WideParents wide;  // (a tipset key wider than the inline form: alive until every kernel below has run)
    if (int rc_t = tipset_inputs_list(ctx, TC_PARENTS_PARSED | TC_CHILD_PARSED, parent_cids40, n_parents, child_cid40, tc, wide)) return rc_t;
    DevBuf<TipsetCtxDev> tc_d;
    IPCFP_HIP(ctx, tc_d.alloc(1));
    IPCFP_HIP(ctx, hipMemcpyAsync(tc_d.p, &tc, sizeof tc, hipMemcpyHostToDevice, ctx->stream));
    int rc = launch_ctx_headers(ctx, rec, tc_d.p, 1);
    if (rc) return rc;
    IPCFP_HIP(ctx, d2h_small(ctx, &tc, tc_d.p, sizeof tc, ctx->stream));
    IPCFP_HIP(ctx, sync_stream(ctx, ctx->stream));
    if (tc.child_status != IPCFP_ST_TRUE) {
        *status_out = ipcfp_status_t(tc.child_status);
        return IPCFP_OK;
    }
    ExecState ex;
    rc = build_exec_order(ctx, rec, tc_d.p, n_parents, ex, /*verify_txmeta=*/1);
    if (rc) return rc;
    if (ex.status != IPCFP_ST_TRUE) {
        *status_out = ipcfp_status_t(ex.status);
        return IPCFP_OK;
    }
    // the receipts AMT's count decides the ranges: `Amtv0::load` of the root (recorded, like every load here)
    DevBuf<uint64_t> info_d;
    IPCFP_HIP(ctx, info_d.alloc(4));
    rc = launch_amt_root_info(ctx, rec, tc.receipts_root, 0, VK_RECEIPT, info_d.p);
    if (rc) return rc;
    uint64_t info[4] = {0, 0, 0, 0};  // status, height, count, bit width
    IPCFP_HIP(ctx, d2h_small(ctx, info, info_d.p, sizeof info, ctx->stream));
    IPCFP_HIP(ctx, sync_stream(ctx, ctx->stream));
    if (info[0] != IPCFP_ST_TRUE) {
        *status_out = ipcfp_status_t(info[0]);
        return IPCFP_OK;
    }
    count = info[2];
    if (capacity) {  // indices the tree can hold: 2^(bit width × (height + 1)), what `get` accepts whatever the count says
        const uint64_t bits = info[3] * (info[1] + 1);
        *capacity = bits >= 63 ? ~0ULL : (1ULL << bits);
    }
    // the tipset pair itself
    std::vector<CidKey> base;
    tipset_parent_keys(tc, wide, base);
    base.push_back(tc.child);
    base.push_back(tc.receipts_root);
    DevBuf<CidKey> base_d;
    DevBuf<uint32_t> missing_d;
    IPCFP_HIP(ctx, base_d.alloc(base.size()));
    IPCFP_HIP(ctx, missing_d.alloc(1));
    IPCFP_HIP(ctx, hipMemsetAsync(missing_d.p, 0, 4, ctx->stream));
    IPCFP_HIP(ctx, hipMemcpyAsync(base_d.p, base.data(), base.size() * sizeof(CidKey), hipMemcpyHostToDevice, ctx->stream));
    rc = launch_mark_cids(ctx, rec, base_d.p, uint32_t(base.size()), missing_d.p);
    if (rc) return rc;
    IPCFP_HIP(ctx, sync_stream(ctx, ctx->stream));  // (base / base_d are read by the kernel just queued)
    *status_out = IPCFP_ST_TRUE;
    return IPCFP_OK;
}

int ipcfp_shard_plan_tipset(ipcfp_ctx_t* ctx, ipcfp_witness_t* w, const uint8_t* parent_cids40, uint32_t n_parents,
                            const uint8_t* child_cid40, uint32_t n_shards, uint32_t shard, ipcfp_status_t* status_out,
                            uint64_t* receipt_lo, uint64_t* receipt_hi, uint64_t* n_receipts, uint32_t* block_ids,
                            uint64_t cap_blocks, uint64_t* n_blocks) {
    if (!ctx || !w || w->ctx != ctx || !child_cid40 || !status_out || !receipt_lo || !receipt_hi || !n_blocks ||
        (n_parents && !parent_cids40) || n_shards == 0 || shard >= n_shards)
        return IPCFP_E_INVALID;
    IPCFP_ENTER(ctx);
    *n_blocks = 0;
    *receipt_lo = *receipt_hi = 0;
    if (n_receipts) *n_receipts = 0;
    *status_out = IPCFP_ST_ERR;
    const uint32_t words = div_up(uint32_t(w->n), 32);
    DevBuf<uint32_t> touched;
    IPCFP_HIP(ctx, touched.alloc(words + 1));
    IPCFP_HIP(ctx, hipMemsetAsync(touched.p, 0, size_t(words + 1) * 4, ctx->stream));
    const WitnessView rec = witness_view(w, touched.p);
    TipsetCtxDev tc;
    uint64_t count = 0, capacity = 0;
    int rc = plan_replicated(ctx, rec, parent_cids40, n_parents, child_cid40, tc, count, status_out, &capacity);
    if (rc || *status_out != IPCFP_ST_TRUE) return rc;
    *status_out = IPCFP_ST_ERR;
    uint64_t lo, hi;
    ipcfp_shard_range(count, n_shards, shard, &lo, &hi);
    // The LAST shard owns every index from its lo on — the root's count is checked by neither Amt::load nor get, and a claim
    // beyond it is the last rank's (ipcfp_route_event_claims) — so its plan covers what the tree CAN hold, not what the
    // count says (an index that is not there costs its lane the walk down to the first missing bitmap bit).
    uint64_t hi_plan = hi;
    if (shard + 1u == n_shards && capacity > hi) hi_plan = capacity;
    if (hi_plan - lo >= 0xffffffffULL) {
        if (hi - lo >= 0xffffffffULL) return set_error(ctx, IPCFP_E_UNSUPPORTED, "shard too large");
        hi_plan = lo + 0xfffffffeULL;
    }
    rc = launch_plan_receipts(ctx, rec, tc.receipts_root, lo, uint32_t(hi_plan - lo));
    if (rc) return rc;
    std::vector<uint32_t> bits(words + 1);
    IPCFP_HIP(ctx, hipMemcpyAsync(bits.data(), touched.p, size_t(words + 1) * 4, hipMemcpyDeviceToHost, ctx->stream));
    IPCFP_HIP(ctx, sync_stream(ctx, ctx->stream));
    uint64_t nb = 0;
    for (uint32_t wd = 0; wd < words; ++wd) {
        uint32_t m = bits[wd];
        while (m) {
            const int b = __builtin_ctz(m);
            m &= m - 1;
            if (block_ids && nb < cap_blocks) block_ids[nb] = wd * 32 + uint32_t(b);
            ++nb;
        }
    }
    *n_blocks = nb;
    *receipt_lo = lo;
    *receipt_hi = hi;
    if (n_receipts) *n_receipts = count;
    *status_out = IPCFP_ST_TRUE;
    return IPCFP_OK;
}

int ipcfp_shard_plan_tipset_all(ipcfp_ctx_t* ctx, ipcfp_witness_t* w, const uint8_t* parent_cids40, uint32_t n_parents,
                                const uint8_t* child_cid40, uint32_t n_shards, ipcfp_status_t* status_out,
                                uint64_t* n_receipts, uint64_t* receipt_bounds, uint64_t* shard_off, uint32_t* block_ids,
                                uint64_t cap_ids, uint64_t* n_ids) {
    if (!ctx || !w || w->ctx != ctx || !child_cid40 || !status_out || !receipt_bounds || !shard_off || !n_ids ||
        (n_parents && !parent_cids40) || n_shards == 0 || n_shards > IPCFP_MAX_SHARDS)
        return IPCFP_E_INVALID;
    IPCFP_ENTER(ctx);
    *n_ids = 0;
    if (n_receipts) *n_receipts = 0;
    for (uint32_t s = 0; s <= n_shards; ++s) receipt_bounds[s] = shard_off[s] = 0;
    *status_out = IPCFP_ST_ERR;
    // bitmap s < G: what shard s alone needs; bitmap G: what every shard needs
    const uint32_t words = div_up(uint32_t(w->n), 32);
    DevBuf<uint32_t> touched;
    IPCFP_HIP(ctx, touched.alloc(size_t(words) * (n_shards + 1) + 1));
    IPCFP_HIP(ctx, hipMemsetAsync(touched.p, 0, (size_t(words) * (n_shards + 1) + 1) * 4, ctx->stream));
    uint32_t* common_d = touched.p + size_t(words) * n_shards;
    TipsetCtxDev tc;
    uint64_t count = 0, capacity = 0;
    int rc = plan_replicated(ctx, witness_view(w, common_d), parent_cids40, n_parents, child_cid40, tc, count, status_out, &capacity);
    if (rc || *status_out != IPCFP_ST_TRUE) return rc;
    *status_out = IPCFP_ST_ERR;
    if (count >= 0xffffffffULL) return set_error(ctx, IPCFP_E_UNSUPPORTED, "more than 2^32-2 receipts");
    const uint64_t n_plan = capacity > count ? (capacity < 0xfffffffeULL ? capacity : 0xfffffffeULL) : count;  // (the last shard: to the tree's capacity)
    for (uint32_t s = 0; s < n_shards; ++s) ipcfp_shard_range(count, n_shards, s, &receipt_bounds[s], &receipt_bounds[s + 1]);
    DevBuf<uint64_t> bounds_d;
    IPCFP_HIP(ctx, bounds_d.alloc(n_shards + 1));
    IPCFP_HIP(ctx, hipMemcpyAsync(bounds_d.p, receipt_bounds, size_t(n_shards + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
    rc = launch_plan_receipts_all(ctx, witness_view(w, touched.p), tc.receipts_root, uint32_t(n_plan), bounds_d.p, n_shards, words);
    if (rc) return rc;
    std::vector<uint32_t> bits(size_t(words) * (n_shards + 1));
    if (!bits.empty())
        IPCFP_HIP(ctx, hipMemcpyAsync(bits.data(), touched.p, bits.size() * 4, hipMemcpyDeviceToHost, ctx->stream));
    IPCFP_HIP(ctx, sync_stream(ctx, ctx->stream));
    const uint32_t* common = bits.data() + size_t(words) * n_shards;
    uint64_t at = 0;
    for (uint32_t s = 0; s < n_shards; ++s) {
        shard_off[s] = at;
        const uint32_t* own = bits.data() + size_t(words) * s;
        for (uint32_t wd = 0; wd < words; ++wd) {
            uint32_t m = own[wd] | common[wd];
            while (m) {
                const int b = __builtin_ctz(m);
                m &= m - 1;
                if (block_ids && at < cap_ids) block_ids[at] = wd * 32 + uint32_t(b);
                ++at;
            }
        }
    }
    shard_off[n_shards] = at;
    *n_ids = at;
    if (n_receipts) *n_receipts = count;
    *status_out = IPCFP_ST_TRUE;
    return IPCFP_OK;
}

int ipcfp_witness_create_subset(ipcfp_ctx_t* ctx, ipcfp_witness_t* src, const uint32_t* block_ids, uint64_t n,
                                uint64_t receipt_lo, uint64_t receipt_hi, ipcfp_witness_t** out) {
    if (!ctx || !src || src->ctx != ctx || !out || (n && !block_ids)) return IPCFP_E_INVALID;
    *out = nullptr;
    if (n >= 0xffffffffULL) return set_error(ctx, IPCFP_E_UNSUPPORTED, "more than 2^32-2 blocks");
    IPCFP_ENTER(ctx);
    std::unique_ptr<ipcfp_witness> w(new (std::nothrow) ipcfp_witness());
    if (!w) return IPCFP_E_NOMEM;
    w->ctx = ctx;
    w->n = n;
    w->nbytes = 0;
    w->receipt_lo = receipt_lo;
    w->receipt_hi = receipt_hi;
    DevBuf<uint32_t> ids_d, len_d, bad_d;
    DevBuf<uint64_t> off_d;
    DevBuf<uint8_t> cids_d;
    IPCFP_HIP(ctx, ids_d.alloc(n));
    IPCFP_HIP(ctx, len_d.alloc(n));
    IPCFP_HIP(ctx, off_d.alloc(n));
    IPCFP_HIP(ctx, cids_d.alloc(n * IPCFP_CID_SLOT));
    IPCFP_HIP(ctx, bad_d.alloc(1));
    IPCFP_HIP(ctx, hipMemsetAsync(bad_d.p, 0, 4, ctx->stream));
    if (n) IPCFP_HIP(ctx, hipMemcpyAsync(ids_d.p, block_ids, n * 4, hipMemcpyHostToDevice, ctx->stream));
    int rc = launch_subset_tables(ctx, ids_d.p, uint32_t(n), uint32_t(src->n), src->off.p, src->len.p, src->cids.p, off_d.p,
                                  len_d.p, cids_d.p, bad_d.p);
    if (rc) return rc;
    uint32_t bad = 0;
    IPCFP_HIP(ctx, d2h_small(ctx, &bad, bad_d.p, 4, ctx->stream));
    IPCFP_HIP(ctx, sync_stream(ctx, ctx->stream));
    if (bad) return set_error(ctx, IPCFP_E_INVALID, "a block id is outside the source witness");
    rc = witness_finish_create(ctx, w.get(), src->arena.p, off_d.p, len_d.p, cids_d.p);
    if (rc) return rc;
    *out = w.release();
    return IPCFP_OK;
}

int ipcfp_witness_set_receipt_range(ipcfp_witness_t* w, uint64_t lo, uint64_t hi) {
    if (!w || lo > hi) return IPCFP_E_INVALID;
    w->receipt_lo = lo;
    w->receipt_hi = hi;
    w->enum_cache.clear();
    w->table_cache.clear();
    return IPCFP_OK;
}

void ipcfp_witness_receipt_range(const ipcfp_witness_t* w, uint64_t* lo, uint64_t* hi) {
    if (lo) *lo = w ? w->receipt_lo : 0;
    if (hi) *hi = w ? w->receipt_hi : ~0ULL;
}

// ---- the collective ----------------------------------------------------------------------------------------
int ipcfp_comm_unique_id(uint8_t id[IPCFP_COMM_ID_BYTES]) {
    if (!id) return IPCFP_E_INVALID;
    Rccl& r = rccl();
    if (!r.handle) return IPCFP_E_UNSUPPORTED;
    NcclUniqueId u;
    if (r.GetUniqueId(&u) != 0) return IPCFP_E_HIP;
    std::memcpy(id, u.internal, IPCFP_COMM_ID_BYTES);
    return IPCFP_OK;
}

int ipcfp_comm_create(ipcfp_ctx_t* ctx, const uint8_t id[IPCFP_COMM_ID_BYTES], int n_ranks, int rank, ipcfp_comm_t** out) {
    if (!ctx || !id || !out || n_ranks < 1 || rank < 0 || rank >= n_ranks) return IPCFP_E_INVALID;
    *out = nullptr;
    Rccl& r = rccl();
    if (!r.handle) return set_error(ctx, IPCFP_E_UNSUPPORTED, "%s", r.error.c_str());
    IPCFP_ENTER(ctx);
    std::unique_ptr<ipcfp_comm> c(new (std::nothrow) ipcfp_comm());
    if (!c) return IPCFP_E_NOMEM;
    c->ctx = ctx;
    c->n_ranks = n_ranks;
    c->rank = rank;
    NcclUniqueId u;
    std::memcpy(u.internal, id, IPCFP_COMM_ID_BYTES);
    const int e = r.CommInitRank(&c->comm, n_ranks, u, rank);
    if (e != 0) return set_error(ctx, IPCFP_E_HIP, "ncclCommInitRank(%d of %d) failed: %s", rank, n_ranks, rccl_err(e));
    *out = c.release();
    return IPCFP_OK;
}

void ipcfp_comm_destroy(ipcfp_comm_t* c) {
    if (!c) return;
    if (c->comm) {
        (void)hipSetDevice(c->ctx->device);
        (void)hipStreamSynchronize(c->ctx->stream);
        (void)rccl().CommDestroy(c->comm);
    }
    delete c;
}

int ipcfp_comm_rank(const ipcfp_comm_t* c) { return c ? c->rank : -1; }
int ipcfp_comm_size(const ipcfp_comm_t* c) { return c ? c->n_ranks : 0; }

// the main stream waits (on the device, not the host) for what K1's stream has queued so far
static int join_k1(ipcfp_ctx* ctx) {
    if (ctx->stream_k1 == ctx->stream) return IPCFP_OK;
    if (!ctx->join_event) IPCFP_HIP(ctx, hipEventCreateWithFlags(&ctx->join_event, hipEventDisableTiming));
    IPCFP_HIP(ctx, hipEventRecord(ctx->join_event, ctx->stream_k1));
    IPCFP_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->join_event, 0));
    return IPCFP_OK;
}

int ipcfp_allgather_device(ipcfp_ctx_t* ctx, ipcfp_comm_t* c, const void* send_d, void* recv_d, uint64_t bytes_per_rank) {
    if (!ctx || !c || c->ctx != ctx || (bytes_per_rank && (!send_d || !recv_d))) return IPCFP_E_INVALID;
    if (bytes_per_rank == 0) return IPCFP_OK;
    IPCFP_ENTER(ctx);
    int rc = join_k1(ctx);  // K1 writes its bitmap on the engine's second stream
    if (rc) return rc;
    const int e = rccl().AllGather(send_d, recv_d, size_t(bytes_per_rank), kNcclUint8, c->comm, ctx->stream);
    if (e != 0) return set_error(ctx, IPCFP_E_HIP, "ncclAllGather failed: %s", rccl_err(e));
    return IPCFP_OK;
}

int ipcfp_allgather_segments(ipcfp_ctx_t* ctx, ipcfp_comm_t* c, const void* const* seg_d, const uint64_t* seg_bytes,
                             uint32_t n_seg, void* staging_d, void* recv_d, uint64_t bytes_per_rank) {
    if (!ctx || !staging_d || (n_seg && (!seg_d || !seg_bytes)) || (c && c->ctx != ctx)) return IPCFP_E_INVALID;
    uint64_t total = 0;
    for (uint32_t i = 0; i < n_seg; ++i) total += seg_bytes[i];
    if (total > bytes_per_rank) return set_error(ctx, IPCFP_E_INVALID, "segments (%llu B) exceed the per-rank width (%llu B)",
                                                 (unsigned long long)total, (unsigned long long)bytes_per_rank);
    IPCFP_ENTER(ctx);
    int rc = join_k1(ctx);
    if (rc) return rc;
    ProfileScope prof(ctx, IPCFP_K_ALLGATHER);  // message packing + the collective
    uint64_t at = 0;
    for (uint32_t i = 0; i < n_seg; ++i) {
        if (seg_bytes[i])
            IPCFP_HIP(ctx, hipMemcpyAsync(static_cast<uint8_t*>(staging_d) + at, seg_d[i], seg_bytes[i],
                                          hipMemcpyDeviceToDevice, ctx->stream));
        at += seg_bytes[i];
    }
    if (at < bytes_per_rank) IPCFP_HIP(ctx, hipMemsetAsync(static_cast<uint8_t*>(staging_d) + at, 0, bytes_per_rank - at, ctx->stream));
    if (!c || c->n_ranks == 1) {  // a single rank: the gathered result is the message itself
        if (recv_d && recv_d != staging_d)
            IPCFP_HIP(ctx, hipMemcpyAsync(recv_d, staging_d, bytes_per_rank, hipMemcpyDeviceToDevice, ctx->stream));
        return IPCFP_OK;
    }
    if (!recv_d) return IPCFP_E_INVALID;
    const int e = rccl().AllGather(staging_d, recv_d, size_t(bytes_per_rank), kNcclUint8, c->comm, ctx->stream);
    if (e != 0) return set_error(ctx, IPCFP_E_HIP, "ncclAllGather failed: %s", rccl_err(e));
    return IPCFP_OK;
}

}  // extern "C"
