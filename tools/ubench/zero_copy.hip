// tools/ubench/zero_copy.hip — a KERNEL reading pinned / registered host memory over PCIe (no DMA engine, no host gather):
// what a device-driven pull of witness blocks out of a bundle in host memory can expect on this box.
//   stream        256 MB contiguous, 16 bytes per lane, coalesced
//   gather B      300 000 blocks of B bytes (B = 128, 352, 512, 1024) at random 128-byte-aligned offsets of a 640 MB buffer,
//                 half a wavefront per block, 16 bytes per lane (k_repack's shape), written back to back into HBM
// for hipHostMalloc'd memory and for a malloc'd buffer after hipHostRegister (timed).
// build: hipcc -O2 --offload-arch=gfx950 -o tools/ubench/zero_copy tools/ubench/zero_copy.hip
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x)                                                   \
    do {                                                        \
        hipError_t e_ = (x);                                    \
        if (e_ != hipSuccess) {                                 \
            std::printf("%s: %s\n", #x, hipGetErrorString(e_)); \
            std::exit(1);                                       \
        }                                                       \
    } while (0)

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

__global__ void k_stream(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n16) {
    for (size_t i = blockIdx.x * size_t(blockDim.x) + threadIdx.x; i < n16; i += size_t(gridDim.x) * blockDim.x) dst[i] = src[i];
}

// one half-wavefront (32 lanes x 16 B = 512 B per step) per block
__global__ void k_gather(const uint8_t* __restrict__ src, const uint64_t* __restrict__ off, uint32_t blen, uint32_t n,
                         uint8_t* __restrict__ dst) {
    const uint32_t g = (blockIdx.x * blockDim.x + threadIdx.x) >> 5, sub = threadIdx.x & 31u;
    if (g >= n) return;
    const uint4* s = reinterpret_cast<const uint4*>(src + off[g]);
    uint4* d = reinterpret_cast<uint4*>(dst + size_t(g) * blen);
    for (uint32_t c = sub; c < blen / 16u; c += 32u) d[c] = s[c];
}

static void run(const char* name, uint8_t* host_d /* device-visible address of the host buffer */, size_t bytes, uint8_t* dev,
                uint64_t* off_d, std::vector<uint64_t>& off_h, uint32_t n) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    float ms;
    const size_t sbytes = 256u << 20;
    for (int rep = 0; rep < 2; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_stream, dim3(4096), dim3(256), 0, 0, reinterpret_cast<const uint4*>(host_d), reinterpret_cast<uint4*>(dev), sbytes / 16);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
    }
    std::printf("%-10s stream 256 MB            %6.2f ms  %5.1f GB/s\n", name, ms, sbytes / ms / 1e6);
    for (uint32_t blen : {128u, 352u, 512u, 1024u}) {
        const uint32_t b16 = (blen + 15u) & ~15u;
        for (uint32_t i = 0; i < n; ++i) off_h[i] = (uint64_t(rand()) * 2654435761ull % (bytes - 4096)) & ~127ull;
        CK(hipMemcpy(off_d, off_h.data(), n * 8, hipMemcpyHostToDevice));
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(k_gather, dim3((n * 32 + 255) / 256), dim3(256), 0, 0, host_d, off_d, b16, n, dev);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&ms, e0, e1));
        }
        std::printf("%-10s gather %6u x %4u B   %6.2f ms  %5.1f GB/s\n", name, n, b16, ms, double(n) * b16 / ms / 1e6);
    }
}

int main() {
    const size_t bytes = 640u << 20;
    const uint32_t n = 300000;
    uint8_t* dev;
    uint64_t* off_d;
    CK(hipMalloc(&dev, 512u << 20));
    CK(hipMalloc(&off_d, n * 8));
    std::vector<uint64_t> off_h(n);
    {
        uint8_t* p;
        CK(hipHostMalloc(reinterpret_cast<void**>(&p), bytes, hipHostMallocDefault));
        std::memset(p, 1, bytes);
        uint8_t* pd;
        CK(hipHostGetDevicePointer(reinterpret_cast<void**>(&pd), p, 0));
        run("pinned", pd, bytes, dev, off_d, off_h, n);
        CK(hipHostFree(p));
    }
    {
        uint8_t* p;
        CK(hipHostMalloc(reinterpret_cast<void**>(&p), bytes, hipHostMallocNonCoherent));
        std::memset(p, 1, bytes);
        uint8_t* pd;
        CK(hipHostGetDevicePointer(reinterpret_cast<void**>(&pd), p, 0));
        run("noncoh", pd, bytes, dev, off_d, off_h, n);
        CK(hipHostFree(p));
    }
    {
        uint8_t* p = static_cast<uint8_t*>(std::malloc(bytes));
        std::memset(p, 1, bytes);
        const double t0 = now();
        CK(hipHostRegister(p, bytes, hipHostRegisterDefault));
        const double t1 = now();
        std::printf("hipHostRegister of %zu MB: %.1f ms\n", bytes >> 20, (t1 - t0) * 1e3);
        uint8_t* pd;
        CK(hipHostGetDevicePointer(reinterpret_cast<void**>(&pd), p, 0));
        run("registered", pd, bytes, dev, off_d, off_h, n);
        CK(hipHostUnregister(p));
        std::free(p);
    }
    // the DMA engine beside it, same buffer sizes: one pageable hipMemcpy of 256 MB
    {
        uint8_t* p = static_cast<uint8_t*>(std::malloc(256u << 20));
        std::memset(p, 1, 256u << 20);
        CK(hipMemcpy(dev, p, 256u << 20, hipMemcpyHostToDevice));
        const double t0 = now();
        CK(hipMemcpy(dev, p, 256u << 20, hipMemcpyHostToDevice));
        const double t1 = now();
        std::printf("pageable hipMemcpy 256 MB      %6.2f ms  %5.1f GB/s\n", (t1 - t0) * 1e3, (256u << 20) / (t1 - t0) / 1e9);
        std::free(p);
    }
    return 0;
}
