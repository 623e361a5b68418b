// csrc/host/upload.cpp — pageable host memory → HBM at PCIe speed.
//
// The witness bytes and the claim tables the C ABI borrows are ordinary process memory (the reference's
// `Vec<u8>` blocks, src/proofs/common/bundle.rs:10-16).  hipMemcpyAsync from pageable memory is staged by the
// runtime through its own pinned buffer on the calling thread, which is the bottleneck of the PCIe-inclusive
// window (T2) for a 0.44 GB witness.  Here a few threads copy chunks into a ring of pinned buffers owned by
// the context and every chunk's DMA is queued on the caller's stream the moment it is staged, so the host
// copy of chunk k+1 overlaps the DMA of chunk k.  Ordering: all DMAs are on `s`; the call returns when the
// last chunk has been QUEUED (stream-ordered like a plain hipMemcpyAsync from pinned memory).
#include <algorithm>
#include <atomic>
#include <cstring>
#include <thread>
#include <vector>

#include "../common.h"

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
    std::vector<std::thread> pool;
    for (unsigned t = 1; t < T; ++t) pool.emplace_back(work, t);
    work(0);
    for (auto& th : pool) th.join();
    if (failed) return set_error(ctx, IPCFP_E_HIP, "staged upload of %zu bytes failed: %s", bytes, hipGetErrorString(hipGetLastError()));
    return IPCFP_OK;
}

}  // namespace ipcfp
