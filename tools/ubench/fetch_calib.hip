// tools/ubench/fetch_calib.hip — what rocprofv3's FETCH_SIZE reports on gfx950 for the access patterns of this engine.
//
// guides/MI355X_MICROARCH.md §HBM: FETCH_SIZE is ½ of the bytes of a wide coalesced streaming read; "other access widths
// are uncalibrated: calibrate on a known byte count in your own access pattern".  The engine's kernels read memory three
// ways, each reproduced here over a 2 GiB buffer (8× the Infinity Cache) with an EXACTLY known number of distinct
// 128-byte lines touched:
//   stream     every lane 16 B, a wavefront 1 KiB contiguous                       (K1's payload loads, staging copies)
//   lane-seq   every lane walks ITS OWN 384-byte record 16 B at a time             (one CBOR parser per lane: the walk kernels)
//   rand16     every lane reads 16 B at a random 128-byte line                     (hash-table probes)
//   rand64     every lane reads 64 B at a random 128-byte line                     (records, event bytes)
//   rand128    every lane reads a whole random 128-byte line
// Run:  rocprofv3 --pmc FETCH_SIZE -- ./fetch_calib   and compare FETCH_SIZE (KB) per kernel with the bytes printed.
// Build: hipcc --offload-arch=gfx950 -O3 -o fetch_calib fetch_calib.hip
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));

__global__ void k_stream(const u64x2* __restrict__ p, size_t n16, unsigned long long* out) {
    unsigned long long acc = 0;
    for (size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n16; i += size_t(gridDim.x) * blockDim.x) {
        const u64x2 v = p[i];
        acc ^= v.x ^ v.y;
    }
    if (acc == 0x1234567812345678ull) out[0] = acc;
}

// lane t owns record t (384 B = three 128-byte lines) and reads it front to back, one dependent 16-byte load at a time
__global__ void k_lane_seq(const u64x2* __restrict__ p, size_t n_rec, unsigned long long* out) {
    const size_t t = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (t >= n_rec) return;
    unsigned long long acc = 0;
    const u64x2* rec = p + t * 24;
    unsigned skip = 0;  // (always 0: it only makes every load's address depend on the load before it, as a parser's does)
    for (int k = 0; k < 24; ++k) {
        const u64x2 v = rec[k + skip];
        acc ^= v.x ^ v.y;
        skip = unsigned(acc == 0x0123456789abcdefull);
    }
    if (acc == 0x1234567812345678ull) out[0] = acc;
}

__device__ __forceinline__ unsigned long long mix(unsigned long long x) {
    x ^= x >> 33;
    x *= 0xff51afd7ed558ccdull;
    x ^= x >> 33;
    x *= 0xc4ceb9fe1a85ec53ull;
    return x ^ (x >> 33);
}

// lane t reads CHUNKS 16-byte chunks at the start of line perm(t) — a permutation of the lines, so every line is touched
// exactly once per launch (n_lines must be a power of two; perm = odd multiplier + xor, a bijection mod 2^k)
template <int CHUNKS>
__global__ void k_rand(const u64x2* __restrict__ p, size_t n_lines, unsigned long long* out) {
    const size_t t = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (t >= n_lines) return;
    const size_t line = ((t * 0x9E3779B97F4A7C15ull) ^ 0x5bd1e995ull) & (n_lines - 1);
    const u64x2* q = p + line * 8;
    unsigned long long acc = 0;
#pragma unroll
    for (int k = 0; k < CHUNKS; ++k) {
        const u64x2 v = q[k];
        acc ^= v.x ^ v.y;
    }
    if (acc == 0x1234567812345678ull) out[0] = acc;
}

#define CK(x)                                                                 \
    do {                                                                      \
        hipError_t e = (x);                                                   \
        if (e != hipSuccess) {                                                \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e));           \
            return 1;                                                         \
        }                                                                     \
    } while (0)

int main() {
    const size_t bytes = size_t(2) << 30;
    void* buf = nullptr;
    unsigned long long* out = nullptr;
    CK(hipMalloc(&buf, bytes));
    CK(hipMalloc(reinterpret_cast<void**>(&out), 64));
    CK(hipMemset(buf, 0x5a, bytes));
    const u64x2* p = static_cast<const u64x2*>(buf);
    const size_t n16 = bytes / 16, n_lines = bytes / 128, n_rec = bytes / 384;
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(k_stream, dim3(8192), dim3(256), 0, 0, p, n16, out);
        hipLaunchKernelGGL(k_lane_seq, dim3(unsigned((n_rec + 255) / 256)), dim3(256), 0, 0, p, n_rec, out);
        hipLaunchKernelGGL(k_rand<1>, dim3(unsigned((n_lines + 255) / 256)), dim3(256), 0, 0, p, n_lines, out);
        hipLaunchKernelGGL(k_rand<4>, dim3(unsigned((n_lines + 255) / 256)), dim3(256), 0, 0, p, n_lines, out);
        hipLaunchKernelGGL(k_rand<8>, dim3(unsigned((n_lines + 255) / 256)), dim3(256), 0, 0, p, n_lines, out);
    }
    CK(hipDeviceSynchronize());
    printf("bytes_requested k_stream %zu k_lane_seq %zu k_rand<1> %zu k_rand<4> %zu k_rand<8> %zu\n", bytes, n_rec * 384, n_lines * 16,
           n_lines * 64, n_lines * 128);
    printf("lines_touched_x128 k_stream %zu k_lane_seq %zu k_rand<1> %zu k_rand<4> %zu k_rand<8> %zu\n", bytes, n_rec * 384, n_lines * 128,
           n_lines * 128, n_lines * 128);
    return 0;
}
