// csrc/host/context.cpp — context lifecycle, error text and HIP-event kernel timing.
#include <algorithm>
#include <chrono>
#include <cstdarg>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../common.h"

namespace ipcfp {

int set_error(ipcfp_ctx* ctx, int rc, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (ctx) ctx->last_error = buf;
    return rc;
}

thread_local DevPool* g_tls_pool = nullptr;

hipError_t wait_stream(ipcfp_ctx* ctx, hipStream_t s) {
    // (the polling event was created on the context's device by ipcfp_ctx_create; without one: the blocking call)
    if (!ctx->spin_sync || !ctx->spin_event) return hipStreamSynchronize(s);
    // Nothing queued since the last wait?  Then there is nothing to record an event behind: an event on an IDLE queue is a
    // submission the command processor takes ≈ 60 µs to pick up, and the caller would sit that out for nothing.
    hipError_t e = hipStreamQuery(s);
    if (e != hipErrorNotReady) return e;
    e = hipEventRecord(ctx->spin_event, s);
    if (e != hipSuccess) return e;
    // poll for a bounded WALL-CLOCK time (2 ms: every wait of a verification step is shorter), then give the core back
    const auto deadline = std::chrono::steady_clock::now() + std::chrono::microseconds(2000);
    for (uint32_t spins = 0;; ++spins) {
        e = hipEventQuery(ctx->spin_event);
        if (e != hipErrorNotReady) return e;
        if ((spins & 63u) == 63u && std::chrono::steady_clock::now() > deadline) return hipStreamSynchronize(s);
    }
}

hipError_t DevPool::take(void** out, size_t bytes, size_t* cap) {
    const size_t want = (bytes + 255) & ~size_t(255);
    // best fit among cached buffers that are not wastefully large
    size_t best = free_.size();
    for (size_t i = 0; i < free_.size(); ++i)
        if (free_[i].second >= want && free_[i].second <= 2 * want + (1u << 20) &&
            (best == free_.size() || free_[i].second < free_[best].second))
            best = i;
    if (best != free_.size()) {
        *out = free_[best].first;
        *cap = free_[best].second;
        cached_bytes -= free_[best].second;
        free_[best] = free_.back();
        free_.pop_back();
        return hipSuccess;
    }
    *cap = want;
    hipError_t e = hipMalloc(out, want);
    if (e != hipSuccess && !free_.empty()) {  // give cached memory back and retry once
        drain();
        e = hipMalloc(out, want);
    }
    return e;
}

void DevPool::give(void* p, size_t cap) {
    // 288 GB of HBM: keeping 16 GB of scratch warm is cheap.  IPCFP_POOL_CACHE_MB (debugging): 0 gives every buffer back to
    // the runtime at once, so that a read past a buffer's end meets an unmapped page instead of a neighbour's slack.
    static const size_t kMaxCached = [] {
        const char* e = std::getenv("IPCFP_POOL_CACHE_MB");
        return e ? size_t(std::strtoull(e, nullptr, 10)) << 20 : size_t(16) << 30;
    }();
    if (cached_bytes + cap > kMaxCached) {
        (void)hipFree(p);
        return;
    }
    free_.emplace_back(p, cap);
    cached_bytes += cap;
}

void DevPool::drain() {
    for (auto& f : free_) (void)hipFree(f.first);
    free_.clear();
    cached_bytes = 0;
}

static hipEvent_t take_event(ipcfp_ctx* ctx) {
    if (!ctx->free_events.empty()) {
        hipEvent_t e = ctx->free_events.back();
        ctx->free_events.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    return e;
}

ProfileScope::ProfileScope(ipcfp_ctx* c, int id, hipStream_t s) : ctx(c), kernel_id(id), stream(s ? s : c->stream) {
    if (!ctx->profiling || (ctx->profile_only >= 0 && ctx->profile_only != id)) return;
    start = take_event(ctx);
    stop = take_event(ctx);
    if (start && stop) (void)hipEventRecord(start, stream);
}

ProfileScope::~ProfileScope() {
    if (!ctx->profiling || !start || !stop) return;
    (void)hipEventRecord(stop, stream);
    ctx->launches.push_back({kernel_id, start, stop});
}

// fold every recorded launch into the per-kernel sums (synchronises the stream)
static int drain_launches(ipcfp_ctx* ctx) {
    if (ctx->launches.empty()) return IPCFP_OK;
    IPCFP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    IPCFP_HIP(ctx, hipStreamSynchronize(ctx->stream_k1));
    IPCFP_HIP(ctx, hipStreamSynchronize(ctx->stream_aux));
    for (auto& l : ctx->launches) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, l.start, l.stop) == hipSuccess && l.kernel_id >= 0 &&
            l.kernel_id < IPCFP_K_COUNT) {
            ctx->prof_count[l.kernel_id] += 1;
            ctx->prof_ms[l.kernel_id] += ms;
        }
        ctx->free_events.push_back(l.start);
        ctx->free_events.push_back(l.stop);
    }
    ctx->launches.clear();
    return IPCFP_OK;
}

}  // namespace ipcfp

using namespace ipcfp;

extern "C" {

int ipcfp_abi_version(void) { return IPCFP_ABI_VERSION; }

const char* ipcfp_strerror(int rc) {
    switch (rc) {
        case IPCFP_OK: return "ok";
        case IPCFP_E_INVALID: return "invalid argument";
        case IPCFP_E_NO_DEVICE: return "no usable HIP device (this engine has no CPU fallback)";
        case IPCFP_E_HIP: return "HIP runtime error";
        case IPCFP_E_NOMEM: return "out of memory";
        case IPCFP_E_UNSUPPORTED: return "unsupported input";
        case IPCFP_E_PARSE: return "parse error";
        default: return "unknown error";
    }
}

int ipcfp_ctx_create(int device, ipcfp_ctx_t** out) {
    if (!out) return IPCFP_E_INVALID;
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return IPCFP_E_NO_DEVICE;
    if (device < 0 || device >= count) return IPCFP_E_INVALID;
    ipcfp_ctx* ctx = new (std::nothrow) ipcfp_ctx();
    if (!ctx) return IPCFP_E_NOMEM;
    ctx->device = device;
    if (hipSetDevice(device) != hipSuccess || hipGetDeviceProperties(&ctx->props, device) != hipSuccess ||
        hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) {
        delete ctx;
        return IPCFP_E_NO_DEVICE;
    }
    // K1 runs on a second stream beside the walk kernels (IPCFP_K1_STREAM: 0 = one stream, 1 = second stream
    // [default], 2 = second stream with a CU mask, 3 = low-priority second stream).  The VALU-bound hash kernel
    // and the latency-bound chain of walk kernels share the chip: measured on the 1M-receipt tipset the step
    // drops 2.61 → 2.50 ms (the walk kernels lose ≈0.1 ms to K1, K1's 0.27 ms disappears from the critical
    // path); CU masks and stream priorities add nothing over the plain second stream.  K1's results are
    // complete after ipcfp_ctx_sync / ipcfp_witness_verify_cids, which wait for both streams.
    // IPCFP_K1_STREAM=2: the second stream is confined to a share of the CUs (IPCFP_K1_CU_PERCENT, default 75;
    // the mask sets bits in an even pattern so every shader engine / XCD keeps free CUs), so the walk
    // kernels' chain of small launches always finds idle CUs while K1 grinds beside it.
    ctx->stream_k1 = ctx->stream;
    {
        const char* e = std::getenv("IPCFP_K1_STREAM");
        const int mode = e ? std::atoi(e) : 1;
        if (mode == 1 && hipStreamCreateWithFlags(&ctx->stream_k1, hipStreamNonBlocking) != hipSuccess)
            ctx->stream_k1 = ctx->stream;
        if (mode == 3) {
            // K1 on the LOWEST-priority stream, the walk kernels on the highest: the dispatcher serves the chain
            // of small latency-bound launches first and K1's wavefronts fill whatever is idle
            int least = 0, greatest = 0;
            hipStream_t hi = nullptr, lo = nullptr;
            if (hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess &&
                hipStreamCreateWithPriority(&hi, hipStreamNonBlocking, greatest) == hipSuccess &&
                hipStreamCreateWithPriority(&lo, hipStreamNonBlocking, least) == hipSuccess) {
                (void)hipStreamDestroy(ctx->stream);
                ctx->stream = hi;
                ctx->stream_k1 = lo;
            } else {
                if (hi) (void)hipStreamDestroy(hi);
                ctx->stream_k1 = ctx->stream;
            }
        }
        if (mode == 2) {
            int pct = 75;
            if (const char* f = std::getenv("IPCFP_K1_CU_PERCENT")) pct = std::atoi(f);
            if (pct < 10) pct = 10;
            if (pct > 100) pct = 100;
            const int cus = ctx->props.multiProcessorCount > 0 ? ctx->props.multiProcessorCount : 256;
            std::vector<uint32_t> mask(size_t((cus + 31) / 32), 0u);
            // Bresenham spread of pct% ones over the CU bits
            int acc = 0;
            for (int i = 0; i < cus; ++i) {
                acc += pct;
                if (acc >= 100) {
                    acc -= 100;
                    mask[size_t(i) >> 5] |= 1u << (i & 31);
                }
            }
            if (hipExtStreamCreateWithCUMask(&ctx->stream_k1, uint32_t(mask.size()), mask.data()) != hipSuccess)
                ctx->stream_k1 = ctx->stream;
        }
    }
    // the block-order event parse on a stream of its own (IPCFP_AUX_STREAM=0: on the main stream, in order)
    ctx->stream_aux = ctx->stream;
    {
        const char* e = std::getenv("IPCFP_AUX_STREAM");
        if (!(e && std::atoi(e) == 0) && hipStreamCreateWithFlags(&ctx->stream_aux, hipStreamNonBlocking) != hipSuccess)
            ctx->stream_aux = ctx->stream;
        if (hipEventCreateWithFlags(&ctx->aux_event, hipEventDisableTiming) != hipSuccess) {
            if (ctx->stream_aux != ctx->stream) (void)hipStreamDestroy(ctx->stream_aux);
            ctx->stream_aux = ctx->stream;
            ctx->aux_event = nullptr;
        }
    }
    // IPCFP_RESERVE_CUS=r: r CUs of every XCD are kept for the narrow stream; K1's stream and the aux stream are re-made
    // with the complementary mask.  (KFD hands the mask's bits out XCD by XCD — bit i belongs to XCD i % 8 — and inside
    // an XCD shader engine by shader engine, so bits [0, 8r) are r CUs per XCD spread over its engines.  Whatever the
    // mapping, the two masks are disjoint.)
    if (const char* e = std::getenv("IPCFP_RESERVE_CUS")) {
        const int cus = ctx->props.multiProcessorCount > 0 ? ctx->props.multiProcessorCount : 256;
        const int r = std::atoi(e);
        const int n_res = r * 8;
        if (r > 0 && n_res < cus && ctx->stream_aux != ctx->stream && ctx->stream_k1 != ctx->stream) {
            std::vector<uint32_t> keep(size_t((cus + 31) / 32), 0u), rest(size_t((cus + 31) / 32), 0u);
            for (int i = 0; i < cus; ++i) (i < n_res ? keep : rest)[size_t(i) >> 5] |= 1u << (i & 31);
            hipStream_t nar = nullptr, k1 = nullptr, aux = nullptr;
            if (hipExtStreamCreateWithCUMask(&nar, uint32_t(keep.size()), keep.data()) == hipSuccess &&
                hipExtStreamCreateWithCUMask(&k1, uint32_t(rest.size()), rest.data()) == hipSuccess &&
                hipExtStreamCreateWithCUMask(&aux, uint32_t(rest.size()), rest.data()) == hipSuccess &&
                hipEventCreateWithFlags(&ctx->narrow_event, hipEventDisableTiming) == hipSuccess) {
                (void)hipStreamDestroy(ctx->stream_k1);
                (void)hipStreamDestroy(ctx->stream_aux);
                ctx->stream_k1 = k1;
                ctx->stream_aux = aux;
                ctx->stream_narrow = nar;
            } else {
                if (nar) (void)hipStreamDestroy(nar);
                if (k1) (void)hipStreamDestroy(k1);
                if (aux) (void)hipStreamDestroy(aux);
            }
        }
    }
    // the head stream (common.h stream_head); it needs the mailbox route's single-context verify call to be of any use
    {
        const char* e = std::getenv("IPCFP_HEAD_STREAM");
        if (e && std::atoi(e) != 0 && !ctx->stream_narrow && ctx->stream_aux != ctx->stream) {  // measured: off (r03_experiments.md)
            if (hipStreamCreateWithFlags(&ctx->stream_head, hipStreamNonBlocking) != hipSuccess ||
                hipEventCreateWithFlags(&ctx->ctl_event, hipEventDisableTiming) != hipSuccess ||
                hipEventCreateWithFlags(&ctx->head_event, hipEventDisableTiming) != hipSuccess ||
                hipEventRecord(ctx->ctl_event, ctx->stream) != hipSuccess) {
                if (ctx->stream_head) (void)hipStreamDestroy(ctx->stream_head);
                ctx->stream_head = nullptr;
            }
        }
    }
    if (const char* e = std::getenv("IPCFP_NARROW_MAX_WG")) ctx->narrow_max_wg = uint32_t(std::max(0, std::atoi(e)));
    if (const char* e = std::getenv("IPCFP_K1_AFTER_BE")) ctx->k1_after_be = std::atoi(e) != 0;
    if (const char* e = std::getenv("IPCFP_K1_DEFER")) ctx->k1_defer = std::atoi(e);
    if (const char* e = std::getenv("IPCFP_K1_GATE")) ctx->k1_gate = std::atoi(e) != 0;
    if (const char* e = std::getenv("IPCFP_SPIN_SYNC")) ctx->spin_sync = std::atoi(e) != 0;
    if (const char* e = std::getenv("IPCFP_HAMT_LEVELS")) ctx->hamt_levels = std::atoi(e);
    if (const char* e = std::getenv("IPCFP_HAMT_TABLE")) ctx->hamt_table = std::atoi(e);
    if (const char* e = std::getenv("IPCFP_HAMT_COOP")) ctx->hamt_coop = std::atoi(e);
    if (const char* e = std::getenv("IPCFP_SCAN_FUSED")) ctx->scan_fused = std::atoi(e);
    if (const char* e = std::getenv("IPCFP_FAST_VERIFY")) ctx->fast_verify = std::atoi(e);
    // wait_stream's polling event belongs to THIS device (created here, right after hipSetDevice(device))
    if (hipEventCreateWithFlags(&ctx->spin_event, hipEventDisableTiming) != hipSuccess) ctx->spin_event = nullptr;
    if (const char* e = std::getenv("IPCFP_B2B_MODE")) ctx->b2b_mode = (std::atoi(e) >= 0 && std::atoi(e) <= 3) ? std::atoi(e) : 0;
    if (const char* e = std::getenv("IPCFP_B2B_WG")) {
        const int wg = std::atoi(e);
        if (wg == 64 || wg == 128 || wg == 192 || wg == 256) ctx->b2b_wg = uint32_t(wg);
    }
    if (hipHostMalloc(reinterpret_cast<void**>(&ctx->pinned), 64 * 1024, hipHostMallocDefault) == hipSuccess)
        ctx->pinned_cap = 64 * 1024;
    else
        ctx->pinned = nullptr;  // read-backs fall back to pageable copies
    // the mailbox page (IPCFP_MAILBOX=0: none — every caller then takes its synchronising route)
    {
        const char* e = std::getenv("IPCFP_MAILBOX");
        void* hp = nullptr;
        void* dp = nullptr;
        if (!(e && std::atoi(e) == 0) &&
            hipHostMalloc(&hp, 4096, hipHostMallocMapped | hipHostMallocCoherent) == hipSuccess) {
            if (hipHostGetDevicePointer(&dp, hp, 0) == hipSuccess &&
                hipEventCreateWithFlags(&ctx->main_event, hipEventDisableTiming) == hipSuccess) {
                if (hipEventCreateWithFlags(&ctx->rehash_event, hipEventDisableTiming) != hipSuccess) ctx->rehash_event = nullptr;
                std::memset(hp, 0, 4096);
                ctx->mailbox = static_cast<unsigned long long*>(hp);
                ctx->mailbox_dev = static_cast<unsigned long long*>(dp);
            } else {
                (void)hipHostFree(hp);
            }
        }
    }
    // the control block and its pinned template / mirror (IPCFP_CTL_BLOCK=0: individual memsets and copies)
    const char* ctl_env = std::getenv("IPCFP_CTL_BLOCK");
    if (!(ctl_env && std::atoi(ctl_env) == 0) &&
        hipHostMalloc(reinterpret_cast<void**>(&ctx->ctl_host), 4 * kCtlHalf, hipHostMallocDefault) == hipSuccess) {
        if (hipMalloc(reinterpret_cast<void**>(&ctx->ctl_dev), 2 * kCtlHalf) == hipSuccess) {
            std::memset(ctx->ctl_host, 0, kCtlHalf);
            std::memset(ctx->ctl_host + kCtlHalf, 0xff, kCtlHalf);
        } else {
            (void)hipHostFree(ctx->ctl_host);
            ctx->ctl_host = nullptr;
            ctx->ctl_dev = nullptr;
        }
    }
    *out = ctx;
    return IPCFP_OK;
}

void ipcfp_ctx_destroy(ipcfp_ctx_t* ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)upload_task_wait(ctx);  // a copy still crossing PCIe on its own thread writes into pool memory: join it first
    (void)hipStreamSynchronize(ctx->stream);
    (void)hipStreamSynchronize(ctx->stream_k1);
    (void)hipStreamSynchronize(ctx->stream_aux);
    if (ctx->scan_scratch) (void)hipFree(ctx->scan_scratch);
    if (ctx->hamt_recs) (void)hipFree(ctx->hamt_recs);
    if (ctx->hamt_scratch) (void)hipFree(ctx->hamt_scratch);
    if (ctx->hamt_etabs) (void)hipFree(ctx->hamt_etabs);
    for (auto& l : ctx->launches) {
        (void)hipEventDestroy(l.start);
        (void)hipEventDestroy(l.stop);
    }
    for (auto e : ctx->free_events) (void)hipEventDestroy(e);
    ctx->pool.drain();
    if (ctx->pinned) (void)hipHostFree(ctx->pinned);
    if (ctx->ctl_host) (void)hipHostFree(ctx->ctl_host);
    if (ctx->ctl_dev) (void)hipFree(ctx->ctl_dev);
    if (ctx->upload_ring) upload_ring_destroy(ctx->upload_ring);
    if (ctx->mailbox) (void)hipHostFree(ctx->mailbox);
    if (ctx->main_event) (void)hipEventDestroy(ctx->main_event);
    if (ctx->join_event) (void)hipEventDestroy(ctx->join_event);
    if (ctx->spin_event) (void)hipEventDestroy(ctx->spin_event);
    if (ctx->aux_event) (void)hipEventDestroy(ctx->aux_event);
    if (ctx->stream_aux != ctx->stream) (void)hipStreamDestroy(ctx->stream_aux);
    if (ctx->stream_k1 != ctx->stream) (void)hipStreamDestroy(ctx->stream_k1);
    if (ctx->rehash_event) (void)hipEventDestroy(ctx->rehash_event);
    if (ctx->stream_narrow) (void)hipStreamDestroy(ctx->stream_narrow);
    if (ctx->stream_copy) (void)hipStreamDestroy(ctx->stream_copy);
    if (ctx->stream_head) (void)hipStreamDestroy(ctx->stream_head);
    if (ctx->ctl_event) (void)hipEventDestroy(ctx->ctl_event);
    if (ctx->head_event) (void)hipEventDestroy(ctx->head_event);
    if (ctx->narrow_event) (void)hipEventDestroy(ctx->narrow_event);
    if (ctx->k1_gate_event) (void)hipEventDestroy(ctx->k1_gate_event);
    (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

const char* ipcfp_last_error(const ipcfp_ctx_t* ctx) { return ctx ? ctx->last_error.c_str() : "null context"; }

void* ipcfp_ctx_stream(ipcfp_ctx_t* ctx) { return ctx ? reinterpret_cast<void*>(ctx->stream) : nullptr; }

// Route selection for A/B measurements and for tests that must drive BOTH routes of one entry point through the same
// corpus; results never depend on it (every fast route answers what the general one would).
int ipcfp_ctx_set_tuning(ipcfp_ctx_t* ctx, const char* key, int64_t value) {
    if (!ctx || !key) return IPCFP_E_INVALID;
    const std::string k(key);
    if (k == "hamt_levels") ctx->hamt_levels = int(value);
    else if (k == "hamt_table") ctx->hamt_table = int(value);
    else if (k == "hamt_coop") ctx->hamt_coop = int(value);
    else if (k == "scan_fused") ctx->scan_fused = int(value);
    else if (k == "fast_verify") ctx->fast_verify = int(value);
    else return set_error(ctx, IPCFP_E_INVALID, "unknown tuning key '%s'", key);
    return IPCFP_OK;
}

int ipcfp_ctx_sync(ipcfp_ctx_t* ctx) {
    if (!ctx) return IPCFP_E_INVALID;
    IPCFP_HIP(ctx, hipSetDevice(ctx->device));  // (a multi-device process: the event record below is per device)
    if (int rc = ipcfp::k1_flush(ctx)) return rc;
    IPCFP_HIP(ctx, wait_stream(ctx, ctx->stream));
    IPCFP_HIP(ctx, wait_stream(ctx, ctx->stream_k1));
    if (ctx->stream_aux != ctx->stream) IPCFP_HIP(ctx, wait_stream(ctx, ctx->stream_aux));
    return IPCFP_OK;
}

int ipcfp_ctx_device_info(ipcfp_ctx_t* ctx, char name[64], int* cu_count, uint64_t* hbm_bytes) {
    if (!ctx) return IPCFP_E_INVALID;
    if (name) {
        // the marketing name comes from libdrm's amdgpu.ids, which a minimal image may lack
        std::strncpy(name, ctx->props.name[0] ? ctx->props.name : ctx->props.gcnArchName, 63);
        name[63] = 0;
    }
    if (cu_count) *cu_count = ctx->props.multiProcessorCount;
    if (hbm_bytes) *hbm_bytes = ctx->props.totalGlobalMem;
    return IPCFP_OK;
}

int ipcfp_profile_enable(ipcfp_ctx_t* ctx, int on) {
    if (!ctx) return IPCFP_E_INVALID;
    if (!on) {
        int rc = drain_launches(ctx);
        if (rc) return rc;
    }
    ctx->profiling = on != 0;
    ctx->profile_only = on >= 2 ? on - 2 : -1;
    return IPCFP_OK;
}

int ipcfp_profile_reset(ipcfp_ctx_t* ctx) {
    if (!ctx) return IPCFP_E_INVALID;
    int rc = drain_launches(ctx);
    if (rc) return rc;
    for (int k = 0; k < IPCFP_K_COUNT; ++k) {
        ctx->prof_count[k] = 0;
        ctx->prof_ms[k] = 0.0;
    }
    return IPCFP_OK;
}

int ipcfp_profile_read(ipcfp_ctx_t* ctx, int kernel_id, uint64_t* launches, double* total_ms) {
    if (!ctx || kernel_id < 0 || kernel_id >= IPCFP_K_COUNT) return IPCFP_E_INVALID;
    int rc = drain_launches(ctx);
    if (rc) return rc;
    if (launches) *launches = ctx->prof_count[kernel_id];
    if (total_ms) *total_ms = ctx->prof_ms[kernel_id];
    return IPCFP_OK;
}

}  // extern "C"
