// csrc/host/upload.cpp — pageable host memory → HBM at PCIe speed.
//
// The witness bytes and the claim tables the C ABI borrows are ordinary process memory (the reference's
// `Vec<u8>` blocks, src/proofs/common/bundle.rs:10-16).  Measured on the MI355X boxes (tools/ubench/h2d_paths,
// 512 MB, profiles/r03_h2d_paths.txt): a pinned source 57.6 GB/s (the DMA engine's ceiling), the runtime's BLOCKING
// hipMemcpy from pageable memory 56.5 GB/s, a ring of pinned chunks filled by 4-24 threads with each chunk's DMA queued
// at once 39-50 GB/s (2-16 MB chunks; 45 GB/s at the 4 MB this file used until round 3).  So a large transfer is the
// blocking copy (IPCFP_UPLOAD_MODE=1: the ring); it is not stream-ordered — the stream is drained first, which costs
// nothing where uploads happen (the head of a call) — and a caller that wants it beside other work runs it on a
// thread of its own (UploadTask below: the claims of a verify call cross PCIe while the tipset's AMTs are walked).
// The ring: a few threads copy chunks into pinned buffers owned by the context and every chunk's DMA is queued on
// the caller's stream the moment it is staged, so the host copy of chunk k+1 overlaps the DMA of chunk k.
#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <cstring>
#include <new>
#include <thread>
#include <vector>

#include "../common.h"
#include "parallel.h"

namespace ipcfp {

namespace {
constexpr size_t kChunk = size_t(4) << 20;
constexpr unsigned kMaxThreads = 8;
constexpr unsigned kSlotsPerThread = 2;
}  // namespace

struct UploadRing {
    uint8_t* buf[kMaxThreads * kSlotsPerThread] = {};
    hipEvent_t done[kMaxThreads * kSlotsPerThread] = {};
    bool recorded[kMaxThreads * kSlotsPerThread] = {};
    unsigned threads = 0;
    ~UploadRing() {
        for (unsigned i = 0; i < kMaxThreads * kSlotsPerThread; ++i) {
            if (buf[i]) (void)hipHostFree(buf[i]);
            if (done[i]) (void)hipEventDestroy(done[i]);
        }
    }
};

void upload_ring_destroy(UploadRing* r) { delete r; }

static UploadRing* ring_of(ipcfp_ctx* ctx) {
    if (ctx->upload_ring) return ctx->upload_ring;
    unsigned hw = std::thread::hardware_concurrency();
    unsigned t = std::max(1u, std::min(kMaxThreads, hw / 2));
    if (const char* e = std::getenv("IPCFP_UPLOAD_THREADS")) t = std::max(1u, std::min(kMaxThreads, unsigned(std::atoi(e))));
    UploadRing* r = new UploadRing();
    r->threads = t;
    for (unsigned i = 0; i < t * kSlotsPerThread; ++i) {
        if (hipHostMalloc(reinterpret_cast<void**>(&r->buf[i]), kChunk, hipHostMallocDefault) != hipSuccess ||
            hipEventCreateWithFlags(&r->done[i], hipEventDisableTiming) != hipSuccess) {
            delete r;
            return nullptr;
        }
    }
    ctx->upload_ring = r;
    return r;
}

int upload(ipcfp_ctx* ctx, void* dst_d, const void* src, size_t bytes, hipStream_t s) {
    if (bytes == 0) return IPCFP_OK;
    static const int mode = [] {
        const char* e = std::getenv("IPCFP_UPLOAD_MODE");
        return e ? std::atoi(e) : 0;
    }();
    if (bytes >= 2 * kChunk && mode == 0) {  // the runtime's blocking copy: nothing queued on `s` may still use dst
        IPCFP_HIP(ctx, hipStreamSynchronize(s));
        IPCFP_HIP(ctx, hipMemcpy(dst_d, src, bytes, hipMemcpyHostToDevice));
        return IPCFP_OK;
    }
    UploadRing* r = bytes >= 2 * kChunk ? ring_of(ctx) : nullptr;
    if (!r) {  // small transfer (or no pinned memory to be had): the runtime's own staging
        IPCFP_HIP(ctx, hipMemcpyAsync(dst_d, src, bytes, hipMemcpyHostToDevice, s));
        return IPCFP_OK;
    }
    const size_t n_chunks = (bytes + kChunk - 1) / kChunk;
    const unsigned T = unsigned(std::min<size_t>(r->threads, n_chunks));
    std::atomic<int> failed{0};
    auto work = [&](unsigned t) {
        if (hipSetDevice(ctx->device) != hipSuccess) {
            failed = 1;
            return;
        }
        unsigned turn = 0;
        for (size_t c = t; c < n_chunks && !failed; c += T, ++turn) {
            const unsigned slot = t * kSlotsPerThread + (turn % kSlotsPerThread);
            if (r->recorded[slot] && hipEventSynchronize(r->done[slot]) != hipSuccess) failed = 1;
            const size_t off = c * kChunk, len = std::min(kChunk, bytes - off);
            std::memcpy(r->buf[slot], static_cast<const uint8_t*>(src) + off, len);
            if (hipMemcpyAsync(static_cast<uint8_t*>(dst_d) + off, r->buf[slot], len, hipMemcpyHostToDevice, s) != hipSuccess ||
                hipEventRecord(r->done[slot], s) != hipSuccess)
                failed = 1;
            r->recorded[slot] = true;
        }
    };
    if (!run_parts(T, work)) failed = 1;  // (host/parallel.h: a thread that cannot be made does not throw across the ABI)
    if (failed) return set_error(ctx, IPCFP_E_HIP, "staged upload of %zu bytes failed: %s", bytes, hipGetErrorString(hipGetLastError()));
    return IPCFP_OK;
}

// ---- an upload beside the caller's own work: blocking copies on a thread of their own --------------------------------
struct UploadTask {
    std::thread th;
    std::atomic<int> err{0};  // hipError_t of the first failed copy
};

UploadTask* upload_task_start(ipcfp_ctx* ctx, void* dst0, const void* src0, size_t bytes0, void* dst1, const void* src1, size_t bytes1) {
    UploadTask* t = new (std::nothrow) UploadTask();
    if (!t) return nullptr;
    const int device = ctx->device;
    // (a stream of its own: blocking copies on the NULL stream from two threads would queue up behind each other)
    if (!ctx->stream_copy && hipStreamCreateWithFlags(&ctx->stream_copy, hipStreamNonBlocking) != hipSuccess) ctx->stream_copy = nullptr;
    hipStream_t cs = ctx->stream_copy;
    try {  // (std::thread throws std::system_error when the thread cannot be made: nothing may cross the C ABI)
        t->th = std::thread([=] {
            hipError_t e = hipSetDevice(device);
            auto copy = [&](void* d, const void* s, size_t n) {
                return cs ? hipMemcpyWithStream(d, s, n, hipMemcpyHostToDevice, cs) : hipMemcpy(d, s, n, hipMemcpyHostToDevice);
            };
            if (e == hipSuccess && bytes0) e = copy(dst0, src0, bytes0);
            if (e == hipSuccess && bytes1) e = copy(dst1, src1, bytes1);
            t->err = int(e);
        });
    } catch (...) {
        delete t;
        return nullptr;  // the caller uploads synchronously instead
    }
    return t;
}

// the data is in HBM when this returns IPCFP_OK (idempotent: a finished task is gone)
int upload_task_wait(ipcfp_ctx* ctx) {
    UploadTask* t = ctx->upload_task;
    if (!t) return IPCFP_OK;
    ctx->upload_task = nullptr;
    t->th.join();
    const int e = t->err;
    delete t;
    if (e) return set_error(ctx, IPCFP_E_HIP, "upload beside the walk failed: %s", hipGetErrorString(hipError_t(e)));
    return IPCFP_OK;
}

}  // namespace ipcfp
