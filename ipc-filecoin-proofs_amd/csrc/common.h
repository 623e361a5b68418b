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
    std::string last_error;
    hipDeviceProp_t props{};
    ipcfp::DevPool pool;
    // --- tuning knobs (env IPCFP_B2B_MODE / IPCFP_B2B_WG; defaults are the measured best) ---
    int b2b_mode = 0;        // 0: hipcc-chosen u64 adds, 1: explicit add_co/addc pairs
    uint32_t b2b_wg = 64;    // K1 workgroup size (multiple of 64, <= 256)
    // --- per-kernel HIP-event timing (ipcfp_profile_*) ---
    bool profiling = false;
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
    hipEvent_t join_event = nullptr;           // main stream ← K1 stream dependency (host/shard.cpp)
    hipEvent_t spin_event = nullptr;           // wait_stream's polling event
    bool spin_sync = true;                     // env IPCFP_SPIN_SYNC=0: always block in hipStreamSynchronize
    ipcfp::UploadRing* upload_ring = nullptr;  // pinned staging ring of ipcfp::upload (created on first use)
};

namespace ipcfp {

int set_error(ipcfp_ctx* ctx, int rc, const char* fmt, ...);

#define IPCFP_HIP(ctx, call)                                                                    \
    do {                                                                                        \
        hipError_t _e = (call);                                                                 \
        if (_e != hipSuccess)                                                                   \
            return ::ipcfp::set_error((ctx), IPCFP_E_HIP, "%s failed: %s (%s:%d)", #call,       \
                                      hipGetErrorString(_e), __FILE__, __LINE__);               \
    } while (0)

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

inline hipError_t sync_stream(ipcfp_ctx* ctx, hipStream_t s) {
    const hipError_t e = wait_stream(ctx, s);
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
    DevBuf<EventRec> events;
    uint64_t n = 0;
    EventTableView view() const { return EventTableView{receipts.p, events.p}; }
};
}  // namespace ipcfp

// The opaque witness of the C ABI: the whole witness resident in HBM as SoA.
struct ipcfp_witness {
    ipcfp_ctx* ctx = nullptr;
    uint64_t n = 0;        // blocks
    uint64_t nbytes = 0;   // payload bytes (sum of len)
    uint64_t arena_bytes = 0;
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
    bool uniform_chunks = false;  // every block has the same chunk count → identity order
    // a shard of one tipset (host/shard.cpp): enumerations of a receipts AMT are restricted to [receipt_lo, receipt_hi)
    uint64_t receipt_lo = 0, receipt_hi = ~0ULL;
    std::vector<std::unique_ptr<ipcfp::EnumCached>> enum_cache;
    std::vector<std::unique_ptr<ipcfp::EventTableCached>> table_cache;
    bool use_event_table = true;  // env IPCFP_EVENT_TABLE=0: every scan / claim walks the blocks (A/B measurements)
};
