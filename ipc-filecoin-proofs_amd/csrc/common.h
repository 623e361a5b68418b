// csrc/common.h — internal declarations shared by the host side and the kernel
// launchers of libipcfp.so.  Nothing here crosses the C ABI (include/ipcfp.h).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <memory>
#include <cstring>
#include <string>
#include <utility>
#include <vector>

#include "ipcfp.h"
#include "kernels/event_table.h"

namespace ipcfp {

// Caching device allocator, one per context.  hipMalloc/hipFree cost ≈100 µs each and serialise
// the device; verification calls need a dozen scratch buffers, so freed buffers are kept and reused
// (stream order on the context's single stream makes reuse safe; every entry point that hands
// buffers back has synchronised the stream first).
struct DevPool {
    std::vector<std::pair<void*, size_t>> free_;  // (pointer, capacity in bytes)
    size_t cached_bytes = 0;
    hipError_t take(void** out, size_t bytes, size_t* cap);
    void give(void* p, size_t cap);
    void drain();
    ~DevPool() { drain(); }
};
extern thread_local DevPool* g_tls_pool;

struct UploadRing;  // host/upload.cpp
struct UploadTask;  // host/upload.cpp
void upload_ring_destroy(UploadRing* r);

struct ProfiledLaunch {
    int kernel_id;
    hipEvent_t start, stop;
};

}  // namespace ipcfp

// The opaque context of the C ABI.
struct ipcfp_ctx {
    int device = -1;
    hipStream_t stream = nullptr;
    hipStream_t stream_k1 = nullptr;  // K1 (VALU-bound hashing) runs beside the latency-bound walk kernels
    hipStream_t stream_aux = nullptr; // the block-order event parse (k_block_events) runs beside both
    // the filter of the last scan on this context: a verify call that has to tabulate the events itself counts THIS
    // filter's matches while it is at it, so that the scan that follows finds its PASS 1 done (a guess about a
    // parameter, never about data: a different filter simply counts from the records)
    ipcfp::ScanParams scan_hint{};
    bool has_scan_hint = false;
    hipEvent_t aux_event = nullptr;   // main stream ← aux stream dependency (host/scan_events.cpp)
    std::string last_error;
    hipDeviceProp_t props{};
    ipcfp::DevPool pool;
    // --- tuning knobs (env IPCFP_B2B_MODE / IPCFP_B2B_WG; defaults are the measured best) ---
    int b2b_mode = 0;        // 0: hipcc-chosen u64 adds, 1: explicit add_co/addc pairs
    uint32_t b2b_wg = 64;    // K1 workgroup size (multiple of 64, <= 256)
    // --- per-kernel HIP-event timing (ipcfp_profile_*) ---
    bool profiling = false;
    int profile_only = -1;  // >= 0: bracket launches of this kernel id alone (the timed region of bench.py: K1)
    std::vector<ipcfp::ProfiledLaunch> launches;   // recorded, not yet read
    std::vector<hipEvent_t> free_events;           // recycled events
    uint64_t prof_count[IPCFP_K_COUNT] = {};
    double prof_ms[IPCFP_K_COUNT] = {};
    // --- small device→host read-backs go through pinned memory (ipcfp::d2h_small / sync_stream): a
    // hipMemcpyAsync into pageable memory is staged and waited for by the runtime, 20-80 µs apiece, and a
    // verification step reads back a dozen scalars (level sizes, error words, counts) ---
    uint8_t* pinned = nullptr;
    size_t pinned_cap = 0, pinned_used = 0;
    struct PendingRead {
        void* dst;
        size_t off, n;
        hipStream_t stream;
        bool hold = false;  // an H2D staged in the page: the slot stays taken until its stream is synchronised
    };
    std::vector<PendingRead> pending;
    int call_depth = 0;
    // --- the call's control block: every flag, counter and error word a verification call hands to its kernels lives
    // in ONE small device buffer, initialised by one copy from a pinned template (first half zeros, second half
    // 0xff) and read back by one copy — a dozen hipMemsetAsync / hipMemcpyAsync of a few bytes each cost ~10 µs
    // apiece as kernels of their own (profiles/r02_timeline_before_ctl.txt)
    uint8_t* ctl_dev = nullptr;
    uint8_t* ctl_host = nullptr;               // pinned: [template 2·kCtlHalf][mirror 2·kCtlHalf]
    uint32_t ctl_used_zero = 0, ctl_used_ff = 0;
    bool ctl_primed = false;
    bool ctl_preprimed = false;  // the block was re-initialised at the END of the last call that used it (ctl_preprime)
    struct CtlRead {
        void* dst;
        uint32_t off, n;
    };
    std::vector<CtlRead> ctl_reads;            // words wanted on the host at the next synchronisation of the main stream
    hipEvent_t join_event = nullptr;           // main stream ← K1 stream dependency (host/shard.cpp)
    hipEvent_t spin_event = nullptr;           // wait_stream's polling event
    bool spin_sync = true;                     // env IPCFP_SPIN_SYNC=0: always block in hipStreamSynchronize
    ipcfp::UploadRing* upload_ring = nullptr;  // pinned staging ring of ipcfp::upload (created on first use)
    hipStream_t stream_copy = nullptr;         // the stream of UploadTask's blocking copies (created on first use)
    ipcfp::UploadTask* upload_task = nullptr;  // claims crossing PCIe on a thread of their own (host/upload.cpp); whoever
                                               // queues a kernel that reads them calls claims_ready first
    // claims that arrive in transport form (host/claims_compact.cpp): expanded on the main stream once they are in HBM
    struct ClaimsExpand {
        bool pending = false;
        const void* compact_d = nullptr;
        const ipcfp_event_claim_group_t* groups_d = nullptr;
        uint32_t n = 0, n_groups = 0;
        const uint8_t* cblob_d = nullptr;
        uint64_t cblob_len = 0, cap_blob = 0;
        void* claims_out_d = nullptr;
        uint8_t* blob_out_d = nullptr;
        uint32_t* scratch_u32 = nullptr;
        uint64_t* scan_scratch = nullptr;
    } claims_expand;
    // claims that are a slice of a larger batch (ipcfp_verify_event_claims_slice): blob offsets rebased once they are in HBM
    struct ClaimsRebase {
        bool pending = false;
        void* claims_d = nullptr;
        uint32_t n = 0;
        uint64_t base = 0, blob_len = 0;
        uint64_t full_len = 0;       // the whole batch's blob (what a record may point into at all)
        uint32_t* miss_d = nullptr;  // set to 1 when a record lies inside the batch's blob but outside the uploaded window (nullable)
        // ipcfp_verify_event_claims_range: the slice was found by binary search in a batch the caller PROMISED to be in
        // exec_index order; the promise is checked where the records are anyway (nullable: no check)
        uint32_t* order_d = nullptr;  // set to 1 when a record's exec_index is below its predecessor's or outside [key_lo, key_hi)
        uint64_t key_lo = 0, key_hi = ~0ull;
    } claims_rebase;
    // --- the mailbox: a page of COHERENT pinned host memory a kernel writes while the stream keeps going (device →
    // host without a synchronisation; kernels/amt_enum.hip k_enum_roots, host/verify_fast.cpp) ---
    unsigned long long* mailbox = nullptr;      // host address
    unsigned long long* mailbox_dev = nullptr;  // the same page as the device addresses it
    unsigned long long mailbox_seq = 0;
    hipEvent_t main_event = nullptr;            // aux stream ← main stream dependency (host/verify_fast.cpp)
    hipEvent_t rehash_event = nullptr;          // main stream ← the deferred TxMeta re-hashes on the aux stream (verify_fast.cpp)
    // --- reserved CUs (env IPCFP_RESERVE_CUS = CUs per XCD, 0 = off; host/context.cpp): the latency-bound head of a
    // verify call — tipset prologue, AMT roots, the narrow interior levels — runs on a stream confined to the reserved
    // CUs while the two side streams (K1, the block-order event parse) are confined to all the others, so those few
    // wavefronts never share a SIMD's issue slots with the hash or the parse ---
    hipStream_t stream_narrow = nullptr;        // null: no reservation (the head runs on `stream`)
    // --- the head stream (env IPCFP_HEAD_STREAM=1; measured and off by default): a verify call's tipset prologue runs here, beside the CID
    // index's inserts on the main stream instead of behind them (its lookups wait for their keys: tipset_prepare.hip
    // LiveIndex).  `ctl_event`: the main stream's last re-initialisation of the control block, which the head waits for ---
    hipStream_t stream_head = nullptr;
    hipEvent_t ctl_event = nullptr;
    hipEvent_t head_event = nullptr;            // head stream → main stream hand-back
    hipEvent_t narrow_event = nullptr;          // main stream ↔ narrow stream hand-overs
    uint32_t narrow_max_wg = 64;                // a level of at most this many workgroups counts as narrow
    bool k1_after_be = false;                   // env IPCFP_K1_AFTER_BE: K1 is queued behind the block-order event parse
    // --- K1 queued late (env IPCFP_K1_DEFER, host/witness.cpp k1_flush): ipcfp_witness_verify_cids_async only notes the
    // request; the event-verify call that follows queues the launch at the point of ITS kernel sequence where the hash
    // kernel costs the critical path least (1: behind the AMT walk, 2: before the verify kernel, 3: behind it).  Every
    // entry point that reads K1's results, synchronises the context or rebuilds the index queues a noted launch first.
    int k1_defer = 0;
    bool k1_gate = false;                       // env IPCFP_K1_GATE: … and K1's stream WAITS for the main stream to get there
    hipEvent_t k1_gate_event = nullptr;
    struct ipcfp_witness* k1_deferred_w = nullptr;
    // --- scratch of the ASYNCHRONOUS batch gets (ipcfp_hamt_get_device: node records, work lists, key hashes).  Owned by
    // the context and only ever grown: the kernels of a call that has already returned may still be reading it, and the
    // next call's kernels follow them on the same stream (a pooled buffer would go back to the pool on return) ---
    // the scan tail's look-back state (kernels/event_scan.hip k_scan_tail_fused): context-owned, told apart by the epoch
    void* scan_scratch = nullptr;
    size_t scan_scratch_bytes = 0;
    unsigned long long scan_epoch = 0;
    int scan_fused = -1;   // 0: the scan's tail as separate launches with read-back copies (round 3's)
    int hamt_levels = -1;  // -1: level by level for batches of >= 1024 queries; 0: the per-query walker alone; k > 0: exactly k levels
    int hamt_coop = -1;    // 0: the level path parses every node with one lane (kernels/hamt_levels.hip k_hamt_lv_parse) also for ActorState trees
    int hamt_table = -1;   // 1: tabulate EVERY block first (hamt_table.h; A/B measurements)
    int fast_verify = -1;  // 0: verify_event_proof never takes the no-synchronisation route (host/verify_fast.cpp)
    void* hamt_recs = nullptr;
    size_t hamt_recs_bytes = 0;
    void* hamt_scratch = nullptr;
    size_t hamt_scratch_bytes = 0;
    void* hamt_etabs = nullptr;  // entry tables of the level path's visited nodes (kernels/hamt_table.h HamtEntryTab)
    size_t hamt_etabs_bytes = 0;
};

namespace ipcfp {

int set_error(ipcfp_ctx* ctx, int rc, const char* fmt, ...);
int k1_flush(ipcfp_ctx* ctx, bool gated = false);  // queue the noted K1 launch, if any (host/witness.cpp)

#define IPCFP_HIP(ctx, call)                                                                    \
    do {                                                                                        \
        hipError_t _e = (call);                                                                 \
        if (_e != hipSuccess)                                                                   \
            return ::ipcfp::set_error((ctx), IPCFP_E_HIP, "%s failed: %s (%s:%d)", #call,       \
                                      hipGetErrorString(_e), __FILE__, __LINE__);               \
    } while (0)

constexpr uint32_t kCtlHalf = 2048;  // (room for one TipsetCtxDev in the zero half: host/verify_fast.cpp)

// `bytes` (multiple of 8) of the call's control block, pre-set to zero (`ff` = false) or to 0xff bytes; nullptr when the
// block is used up or absent (the caller then allocates and memsets as before).
inline void* ctl_take(ipcfp_ctx* ctx, uint32_t bytes, bool ff) {
    if (!ctx->ctl_dev) return nullptr;
    bytes = (bytes + 7u) & ~7u;
    uint32_t& used = ff ? ctx->ctl_used_ff : ctx->ctl_used_zero;
    if (used + bytes > kCtlHalf) return nullptr;
    if (!ctx->ctl_primed) {
        if (hipMemcpyAsync(ctx->ctl_dev, ctx->ctl_host, 2 * kCtlHalf, hipMemcpyHostToDevice, ctx->stream) != hipSuccess) return nullptr;
        ctx->ctl_primed = true;
    }
    void* p = ctx->ctl_dev + (ff ? kCtlHalf : 0u) + used;
    used += bytes;
    ctx->ctl_preprimed = false;  // in use: whoever wants it fresh for the next call re-initialises it (ctl_preprime)
    return p;
}
// (only the OUTERMOST entry point may do this: an inner call's words are still in use by its caller)
// Re-initialise the control block NOW, at the end of a call (the stream is idle: its copy costs nothing), so that the
// next call does not start with a copy kernel — at the head of a verification call that copy waits its turn behind
// the side streams' grids (13-37 us measured).
inline void ctl_preprime(ipcfp_ctx* ctx) {
    if (!ctx->ctl_dev || ctx->ctl_preprimed || ctx->call_depth != 1) return;
    if (hipMemcpyAsync(ctx->ctl_dev, ctx->ctl_host, 2 * kCtlHalf, hipMemcpyHostToDevice, ctx->stream) == hipSuccess) {
        ctx->ctl_preprimed = true;
        if (ctx->ctl_event) (void)hipEventRecord(ctx->ctl_event, ctx->stream);
    }
}
// Queue ONE read-back of the whole block on the main stream; after sync_stream every word is available through ctl_value.
inline hipError_t ctl_fetch(ipcfp_ctx* ctx) {
    if (!ctx->ctl_dev || !ctx->ctl_primed) return hipSuccess;
    return hipMemcpyAsync(ctx->ctl_host + 2 * kCtlHalf, ctx->ctl_dev, 2 * kCtlHalf, hipMemcpyDeviceToHost, ctx->stream);
}
inline bool ctl_owns(const ipcfp_ctx* ctx, const void* dev_ptr) {
    const uint8_t* p = static_cast<const uint8_t*>(dev_ptr);
    return ctx->ctl_dev && p >= ctx->ctl_dev && p < ctx->ctl_dev + 2 * kCtlHalf;
}
template <typename T>
inline T ctl_value(const ipcfp_ctx* ctx, const void* dev_ptr) {
    T v;
    std::memcpy(&v, ctx->ctl_host + 2 * kCtlHalf + (static_cast<const uint8_t*>(dev_ptr) - ctx->ctl_dev), sizeof(T));
    return v;
}

// Queue an asynchronous read-back of n bytes; `dst` is valid after the next sync_stream on `s`.
inline hipError_t d2h_small(ipcfp_ctx* ctx, void* dst, const void* src_d, size_t n, hipStream_t s) {
    const size_t need = (n + 15) & ~size_t(15);
    if (!ctx->pinned || ctx->pinned_used + need > ctx->pinned_cap) return hipMemcpyAsync(dst, src_d, n, hipMemcpyDeviceToHost, s);
    const size_t off = ctx->pinned_used;
    ctx->pinned_used += need;
    ctx->pending.push_back({dst, off, n, s, false});
    return hipMemcpyAsync(ctx->pinned + off, src_d, n, hipMemcpyDeviceToHost, s);
}
// A few bytes host → device through the same pinned page (valid until the next sync_stream of that stream).
inline hipError_t h2d_small(ipcfp_ctx* ctx, void* dst_d, const void* src, size_t n, hipStream_t s) {
    const size_t need = (n + 15) & ~size_t(15);
    if (!ctx->pinned || ctx->pinned_used + need > ctx->pinned_cap) return hipMemcpyAsync(dst_d, src, n, hipMemcpyHostToDevice, s);
    const size_t off = ctx->pinned_used;
    ctx->pinned_used += need;
    std::memcpy(ctx->pinned + off, src, n);
    ctx->pending.push_back({nullptr, off, 0, s, true});
    return hipMemcpyAsync(dst_d, ctx->pinned + off, n, hipMemcpyHostToDevice, s);
}
// hipStreamSynchronize + delivery of the read-backs queued on that stream.  Every synchronisation of an
// engine stream goes through here.
// The wait itself: hipStreamSynchronize parks the thread and is woken by an interrupt — 30-70 µs before the host runs
// again, and a verification pass waits for the device several times (tree shapes, match counts).  Polling an event
// keeps the thread on the core and sees the completion within a few microseconds (ctx->spin_sync, default on; a
// wait that lasts longer than ~2 ms falls back to the blocking call).
hipError_t wait_stream(ipcfp_ctx* ctx, hipStream_t s);

// `last_of_call`: nothing of this call will read the control block on the device after this wait — so its
// re-initialisation for the NEXT call (ctl_preprime) is queued here, behind the read-back and in front of the wait, while
// the GPU is still awake.  Queued after the wait it is the first submission to an idle queue: ≈ 60 µs until the command
// processor picks it up (host API trace: profiles/r04_experiments.md), which the next call's first synchronisation —
// ipcfp_witness_rebuild_index — then sits out.
inline hipError_t sync_stream(ipcfp_ctx* ctx, hipStream_t s, bool last_of_call = false) {
    const bool ctl = s == ctx->stream && !ctx->ctl_reads.empty();
    if (ctl) (void)ctl_fetch(ctx);  // one copy for every control word the host asked for
    if (last_of_call && s == ctx->stream && ctx->ctl_dev && ctx->ctl_primed && !ctx->ctl_preprimed && ctx->call_depth == 1 &&
        hipMemcpyAsync(ctx->ctl_dev, ctx->ctl_host, 2 * kCtlHalf, hipMemcpyHostToDevice, ctx->stream) == hipSuccess) {
        ctx->ctl_preprimed = true;
        if (ctx->ctl_event) (void)hipEventRecord(ctx->ctl_event, ctx->stream);
    }
    const hipError_t e = wait_stream(ctx, s);
    if (ctl) {
        if (e == hipSuccess)
            for (auto& r : ctx->ctl_reads) std::memcpy(r.dst, ctx->ctl_host + 2 * kCtlHalf + r.off, r.n);
        ctx->ctl_reads.clear();
    }
    bool others = false;
    for (auto& r : ctx->pending) {
        if (r.stream == s) {
            if (r.dst && e == hipSuccess) std::memcpy(r.dst, ctx->pinned + r.off, r.n);
            r.dst = nullptr;  // delivered (or lost with the failed synchronisation): never written twice
            r.hold = false;
        } else if (r.dst || r.hold) {
            others = true;
        }
    }
    if (!others) {
        ctx->pending.clear();
        ctx->pinned_used = 0;
    }
    return e;
}

// Pageable (or pinned) host memory → HBM, stream-ordered on `s`; large transfers are staged by several threads
// through the context's pinned ring (host/upload.cpp).
int upload(ipcfp_ctx* ctx, void* dst_d, const void* src, size_t bytes, hipStream_t s);
UploadTask* upload_task_start(ipcfp_ctx* ctx, void* dst0, const void* src0, size_t bytes0, void* dst1, const void* src1, size_t bytes1);
int upload_task_wait(ipcfp_ctx* ctx);
// the claims a verify kernel is about to read are in HBM in the form it reads: the upload beside the walk is over
// (upload_task_wait) and, when they crossed PCIe in transport form, their expansion is queued on the main stream
int claims_ready(ipcfp_ctx* ctx);

// RAII bracket: records an event pair around a kernel launch when profiling is on.
struct ProfileScope {
    ipcfp_ctx* ctx;
    int kernel_id;
    hipStream_t stream;
    hipEvent_t start = nullptr, stop = nullptr;
    ProfileScope(ipcfp_ctx* c, int id, hipStream_t s = nullptr);
    ~ProfileScope();
};

// A device allocation owned by the engine.
// Error paths that leave a function while ANOTHER stream (or rounds queued ahead of the host) may still touch pooled
// scratch: the guard drains that stream before the DevBufs declared BEFORE it go back to the pool (declare it after
// them: destructors run in reverse order).  Disarmed on the ordinary path, where an event wait orders the reuse.
struct StreamDrainGuard {
    hipStream_t stream;
    bool armed = false;
    explicit StreamDrainGuard(hipStream_t s) : stream(s) {}
    StreamDrainGuard(const StreamDrainGuard&) = delete;
    StreamDrainGuard& operator=(const StreamDrainGuard&) = delete;
    ~StreamDrainGuard() {
        if (armed) (void)hipStreamSynchronize(stream);
    }
};

template <typename T>
struct DevBuf {
    T* p = nullptr;
    size_t count = 0;
    size_t cap = 0;          // capacity in bytes as handed out by the pool
    DevPool* owner = nullptr;  // pool the buffer returns to (null: plain hipFree)
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    ~DevBuf() { release(); }
    void release() {
        if (p) {
            if (owner) owner->give(p, cap);
            else (void)hipFree(p);
        }
        p = nullptr;
        count = 0;
        cap = 0;
        owner = nullptr;
    }
    // an allocation of its own (never pooled): buffers that another stream writes while the main stream's pooled
    // scratch is still in flight
    hipError_t alloc_unpooled(size_t n) {
        release();
        count = n;
        cap = (n ? n : 1) * sizeof(T);
        return hipMalloc(reinterpret_cast<void**>(&p), cap);
    }
    hipError_t alloc(size_t n) {
        release();
        count = n;
        size_t bytes = (n ? n : 1) * sizeof(T);
        if (g_tls_pool) {
            owner = g_tls_pool;
            return owner->take(reinterpret_cast<void**>(&p), bytes, &cap);
        }
        cap = bytes;
        return hipMalloc(reinterpret_cast<void**>(&p), bytes);
    }
    size_t bytes() const { return count * sizeof(T); }
    void swap(DevBuf& o) {
        std::swap(p, o.p);
        std::swap(count, o.count);
        std::swap(cap, o.cap);
        std::swap(owner, o.owner);
    }
};

// `count` control words for the kernels of this call, pre-set to zero or to 0xff bytes: a slice of the control block
// when there is room, else `own` (allocated and memset here).
template <typename T>
inline hipError_t ctl_words(ipcfp_ctx* ctx, DevBuf<T>& own, T*& p, size_t count, bool ff) {
    p = static_cast<T*>(ctl_take(ctx, uint32_t(count * sizeof(T)), ff));
    if (p) return hipSuccess;
    hipError_t e = own.alloc(count);
    if (e != hipSuccess) return e;
    p = own.p;
    return hipMemsetAsync(own.p, ff ? 0xff : 0, count * sizeof(T), ctx->stream);
}
// `*dst` = the device value at p after the next sync_stream of the main stream (p: a control word or any device address)
inline hipError_t ctl_read(ipcfp_ctx* ctx, void* dst, const void* p, size_t n) {
    if (ctl_owns(ctx, p)) {
        ctx->ctl_reads.push_back({dst, uint32_t(static_cast<const uint8_t*>(p) - ctx->ctl_dev), uint32_t(n)});
        return hipSuccess;
    }
    return d2h_small(ctx, dst, p, n, ctx->stream);
}

// Binds the calling thread to a context for the duration of one C-ABI call: device selection and
// the allocation pool DevBufs draw from.
struct CallScope {
    DevPool* prev;
    ipcfp_ctx* ctx;
    explicit CallScope(ipcfp_ctx* c) : prev(g_tls_pool), ctx(c) {
        g_tls_pool = &c->pool;
        if (c->call_depth++ == 0) {  // read-backs a failed call left behind point at dead stack frames
            c->pending.clear();
            c->pinned_used = 0;
            c->ctl_used_zero = c->ctl_used_ff = 0;
            c->ctl_primed = c->ctl_preprimed;
            c->ctl_reads.clear();
        }
    }
    ~CallScope() {
        g_tls_pool = prev;
        --ctx->call_depth;
    }
};
#define IPCFP_ENTER(ctx)                                 \
    IPCFP_HIP((ctx), hipSetDevice((ctx)->device));       \
    ::ipcfp::CallScope _ipcfp_call_scope(ctx)

inline uint32_t div_up(uint64_t a, uint64_t b) { return uint32_t((a + b - 1) / b); }

}  // namespace ipcfp

namespace ipcfp {
// A cached `Amt::for_each` enumeration of one AMT of the witness (amt_enum.hip): the leaf values in
// index order.  Valid until the witness index is rebuilt.
struct EnumCached {
    uint64_t root[5];
    int version = 0, vkind = 0;
    DevBuf<uint8_t> leaves;  // LeafRef[n]
    uint64_t n = 0;
    uint64_t error = ~0ULL;  // packed first error, ~0 = none
    bool dense = false;      // leaf i has index lo + i for every i
    uint64_t lo = 0, hi = ~0ULL;  // the index range the enumeration was restricted to
};
// The event table of one receipts AMT (range) of the witness (kernels/event_table.h).  Valid until the witness
// index is rebuilt.
struct EventTableCached {
    uint64_t root[5];
    uint64_t lo = 0, hi = ~0ULL;
    DevBuf<ReceiptRec> receipts;  // one per enumerated receipt leaf
    const EventRec* events = nullptr;  // the witness's block table owns the records (ipcfp_witness::bt_events)
    uint64_t n = 0;
    // match counts per receipt for the filter the block table was built with (has_counts), and the first failing
    // receipt of the build (packed like an enumeration error; a device word, read back by whoever reports it)
    DevBuf<uint32_t> counts;
    bool has_counts = false;
    ScanParams counts_filter{};
    DevBuf<unsigned long long> err_word;
    EventTableView view() const { return EventTableView{receipts.p, events}; }
};
}  // namespace ipcfp

// The opaque witness of the C ABI: the whole witness resident in HBM as SoA.
struct ipcfp_witness {
    ipcfp_ctx* ctx = nullptr;
    uint32_t last_scan_phase = 0;  // IPCFP_SCAN_PHASE_* of the last ipcfp_scan_events* on this witness (0: it returned TRUE)
    uint64_t n = 0;        // blocks
    uint64_t nbytes = 0;   // payload bytes (sum of len)
    uint64_t arena_bytes = 0;
    uint32_t max_block_len = 0xffffffffu;  // an upper bound of every block's length (from the head of the K1 schedule)
    ipcfp::DevBuf<uint8_t> arena;     // blocks on 128-byte lines, in K1 schedule order, + 256 B tail slack
    ipcfp::DevBuf<uint64_t> off;      // n
    ipcfp::DevBuf<uint32_t> len;      // n
    ipcfp::DevBuf<uint8_t> cids;      // n × 40
    ipcfp::DevBuf<uint32_t> order;    // n: block ids sorted by 128-byte chunk count (K1 lane schedule)
    ipcfp::DevBuf<uint64_t> k1_meta;  // n × {arena offset u64, len u32, block id u32} in schedule order
    ipcfp::DevBuf<uint8_t> k1_cids;   // n × 40: claimed CIDs in schedule order
    ipcfp::DevBuf<uint32_t> ok_bits;  // ceil(n/32)
    ipcfp::DevBuf<uint8_t> cid_status;  // n
    ipcfp::DevBuf<unsigned long long> counters;  // [0] = mismatches
    // CID → block-id index (K4)
    ipcfp::DevBuf<uint32_t> index_slots;  // table of block ids, 0xffffffff = empty
    uint32_t index_mask = 0;
    // the fill in progress (kernels/cid_index.hip): workgroups of k_index_insert that have finished / that there are, and
    // the point of the main stream where the table was cleared (a lookup on another stream may start from there: the
    // tipset prologue on the head stream, host/verify_fast.cpp)
    ipcfp::DevBuf<uint32_t> index_done;
    uint32_t index_wgs = 0;
    hipEvent_t index_event = nullptr;
    ~ipcfp_witness() {
        if (index_event) (void)hipEventDestroy(index_event);
    }
    bool uniform_chunks = false;  // every block has the same chunk count → identity order
    // a shard of one tipset (host/shard.cpp): enumerations of a receipts AMT are restricted to [receipt_lo, receipt_hi)
    uint64_t receipt_lo = 0, receipt_hi = ~0ULL;
    std::vector<std::unique_ptr<ipcfp::EnumCached>> enum_cache;
    std::vector<std::unique_ptr<ipcfp::EventTableCached>> table_cache;
    bool use_event_table = true;  // env IPCFP_EVENT_TABLE=0: every scan / claim walks the blocks (A/B measurements)
    // the block table (kernels/event_table.h, k_block_events): one BlockRec per block + the EventRec pool.  The
    // buffers are the witness's own and live as long as it does; `bt_valid` is dropped with the index.
    ipcfp::DevBuf<ipcfp::BlockRec> bt_blocks;
    ipcfp::DevBuf<ipcfp::EventRec> bt_events;
    ipcfp::DevBuf<uint32_t> bt_used;
    bool bt_valid = false;    // k_block_events has been queued (aux stream) for the current arena
    bool bt_joined = false;   // ... and the main stream waits for it
    bool bt_has_filter = false;
    ipcfp::ScanParams bt_filter{};  // the filter whose matches BlockRec::kind_matches counts
};
