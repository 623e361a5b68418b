// tools/ubench/h2d_paths.hip — pageable host memory → HBM: what each route delivers on this box (GB/s over 512 MB).
//   pageable     one hipMemcpy from malloc'd memory (the runtime's own staging)
//   pinned       one hipMemcpyAsync from hipHostMalloc'd memory (the DMA engine alone: the ceiling)
//   registered   hipHostRegister of the malloc'd buffer (timed separately), then one hipMemcpyAsync
//   staged T/C   T threads copy C-MB chunks into a ring of pinned buffers (2 per thread), each chunk's DMA queued at once
//                (csrc/host/upload.cpp with its two knobs)
// build: hipcc -O2 --offload-arch=gfx950 -o tools/ubench/h2d_paths tools/ubench/h2d_paths.hip -lpthread
#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#define CK(x)                                                                   \
    do {                                                                        \
        hipError_t e_ = (x);                                                    \
        if (e_ != hipSuccess) {                                                 \
            std::printf("%s: %s\n", #x, hipGetErrorString(e_));                 \
            std::exit(1);                                                       \
        }                                                                       \
    } while (0)

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

static double staged(uint8_t* dst, const uint8_t* src, size_t bytes, unsigned T, size_t chunk, hipStream_t s) {
    std::vector<uint8_t*> buf(T * 2);
    std::vector<hipEvent_t> ev(T * 2);
    for (unsigned i = 0; i < T * 2; ++i) {
        CK(hipHostMalloc(reinterpret_cast<void**>(&buf[i]), chunk, hipHostMallocDefault));
        CK(hipEventCreateWithFlags(&ev[i], hipEventDisableTiming));
        std::memset(buf[i], 0, chunk);
    }
    double best = 1e9;
    for (int rep = 0; rep < 3; ++rep) {
        std::vector<char> rec(T * 2, 0);
        const size_t n_chunks = (bytes + chunk - 1) / chunk;
        CK(hipDeviceSynchronize());
        const double t0 = now();
        auto work = [&](unsigned t) {
            (void)hipSetDevice(0);
            unsigned turn = 0;
            for (size_t c = t; c < n_chunks; c += T, ++turn) {
                const unsigned slot = t * 2 + (turn & 1);
                if (rec[slot]) (void)hipEventSynchronize(ev[slot]);
                const size_t off = c * chunk, len = std::min(chunk, bytes - off);
                std::memcpy(buf[slot], src + off, len);
                (void)hipMemcpyAsync(dst + off, buf[slot], len, hipMemcpyHostToDevice, s);
                (void)hipEventRecord(ev[slot], s);
                rec[slot] = 1;
            }
        };
        std::vector<std::thread> pool;
        for (unsigned t = 1; t < T; ++t) pool.emplace_back(work, t);
        work(0);
        for (auto& th : pool) th.join();
        CK(hipStreamSynchronize(s));
        best = std::min(best, now() - t0);
    }
    for (unsigned i = 0; i < T * 2; ++i) {
        (void)hipHostFree(buf[i]);
        (void)hipEventDestroy(ev[i]);
    }
    return best;
}

int main() {
    const size_t bytes = size_t(512) << 20;
    uint8_t* src = static_cast<uint8_t*>(std::malloc(bytes));
    for (size_t i = 0; i < bytes; i += 4096) src[i] = uint8_t(i >> 12);
    uint8_t *dst, *pin;
    CK(hipMalloc(reinterpret_cast<void**>(&dst), bytes));
    CK(hipHostMalloc(reinterpret_cast<void**>(&pin), bytes, hipHostMallocDefault));
    std::memcpy(pin, src, bytes);
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    auto gbps = [&](double t) { return double(bytes) / t / 1e9; };
    double best = 1e9;
    for (int r = 0; r < 3; ++r) {
        const double t0 = now();
        CK(hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice));
        best = std::min(best, now() - t0);
    }
    std::printf("pageable hipMemcpy      %6.1f GB/s\n", gbps(best));
    best = 1e9;
    for (int r = 0; r < 3; ++r) {
        const double t0 = now();
        CK(hipMemcpyWithStream(dst, src, bytes, hipMemcpyHostToDevice, s));
        best = std::min(best, now() - t0);
    }
    std::printf("pageable hipMemcpyWithStream %6.1f GB/s\n", gbps(best));
    best = 1e9;
    for (int r = 0; r < 3; ++r) {
        const double t0 = now();
        CK(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, s));
        CK(hipStreamSynchronize(s));
        best = std::min(best, now() - t0);
    }
    std::printf("pageable hipMemcpyAsync+sync %6.1f GB/s\n", gbps(best));
    {   // two blocking copies at once: the null stream on this thread, a stream of its own on another
        uint8_t* dst2;
        CK(hipMalloc(reinterpret_cast<void**>(&dst2), bytes / 4));
        hipStream_t s2;
        CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
        const double t0 = now();
        std::thread th([&] {
            (void)hipSetDevice(0);
            (void)hipMemcpyWithStream(dst, src, bytes, hipMemcpyHostToDevice, s2);
        });
        CK(hipMemcpy(dst2, src, bytes / 4, hipMemcpyHostToDevice));
        const double t_small = now() - t0;
        th.join();
        std::printf("two at once: 128 MB on the null stream done after %.1f ms, 512 MB on a thread's stream after %.1f ms (%.1f GB/s together)\n",
                    t_small * 1e3, (now() - t0) * 1e3, double(bytes + bytes / 4) / (now() - t0) / 1e9);
        (void)hipFree(dst2);
    }
    best = 1e9;
    for (int r = 0; r < 3; ++r) {
        CK(hipDeviceSynchronize());
        const double t0 = now();
        CK(hipMemcpyAsync(dst, pin, bytes, hipMemcpyHostToDevice, s));
        CK(hipStreamSynchronize(s));
        best = std::min(best, now() - t0);
    }
    std::printf("pinned hipMemcpyAsync   %6.1f GB/s\n", gbps(best));
    {
        const double t0 = now();
        CK(hipHostRegister(src, bytes, hipHostRegisterDefault));
        const double treg = now() - t0;
        best = 1e9;
        for (int r = 0; r < 3; ++r) {
            CK(hipDeviceSynchronize());
            const double t1 = now();
            CK(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, s));
            CK(hipStreamSynchronize(s));
            best = std::min(best, now() - t1);
        }
        const double t2 = now();
        CK(hipHostUnregister(src));
        std::printf("registered              %6.1f GB/s  (hipHostRegister %.1f ms, unregister %.1f ms)\n", gbps(best), treg * 1e3,
                    (now() - t2) * 1e3);
    }
    for (unsigned T : {4u, 8u})
        for (size_t mb : {4u, 16u})
            std::printf("staged T=%2u chunk=%2zu MB  %6.1f GB/s\n", T, mb, gbps(staged(dst, src, bytes, T, mb << 20, s)));
    return 0;
}
