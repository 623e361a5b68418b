// csrc/kernels/repack.hip — builds the HBM-resident witness arena.
//
// The reference keeps the witness as `HashMap<Cid, Vec<u8>>`
// (`load_witness_store`, src/proofs/events/verifier.rs:79-89,
// src/proofs/storage/verifier.rs:68-78).  The engine keeps ONE byte arena with
// every block starting on a 128-byte line, so the hash and walk kernels can
// use 16-byte loads and no line is shared by two blocks, plus the (offset, length) table.  When the caller's blocks
// are packed at arbitrary offsets this kernel re-lays them out; it runs at copy
// speed and only at witness creation.
#include <hip/hip_runtime.h>

#include "../common.h"
#include "launch.h"
#include "scan_dev.h"

namespace ipcfp {

// Every block starts on a 128-byte boundary (one HBM/L2 line): K1 reads whole 128-byte chunks, so
// with line-aligned blocks no line is shared between two lanes' blocks and each is fetched exactly
// once (16-byte alignment re-fetched the shared boundary lines: 839 MB of traffic for 522 MB of
// distinct lines on the tipset witness).  An empty block still owns one line.
__device__ __forceinline__ uint64_t round16(uint32_t x) { return x == 0 ? 128ull : (uint64_t(x) + 127ull) & ~127ull; }

// pass 1: per-1024-element tile sums of round16(len)
__global__ __launch_bounds__(256) void k_tile_sums(const uint32_t* __restrict__ len, uint32_t n,
                                                   uint64_t* __restrict__ tile_sums) {
    __shared__ uint64_t smem[17];
    const uint32_t base = blockIdx.x * 1024u + threadIdx.x * 4u;
    uint64_t s = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (base + k < n) s += round16(len[base + k]);
    uint64_t total;
    (void)block_exclusive_scan(s, smem, &total);
    if (threadIdx.x == 0) tile_sums[blockIdx.x] = total;
}

// pass 2: exclusive scan of the tile sums in place (single workgroup, serial carry over 1024-wide strips)
__global__ __launch_bounds__(1024) void k_scan_tiles(uint64_t* __restrict__ tile_sums, uint32_t ntiles,
                                                     uint64_t* __restrict__ total_out) {
    __shared__ uint64_t smem[17];
    uint64_t carry = 0;
    for (uint32_t base = 0; base < ntiles; base += 1024) {
        const uint32_t i = base + threadIdx.x;
        const uint64_t v = i < ntiles ? tile_sums[i] : 0;
        uint64_t total;
        const uint64_t ex = block_exclusive_scan(v, smem, &total);
        if (i < ntiles) tile_sums[i] = carry + ex;
        carry += total;
    }
    if (threadIdx.x == 0) *total_out = carry;
}

// pass 3: final offsets
__global__ __launch_bounds__(256) void k_apply_offsets(const uint32_t* __restrict__ len, uint32_t n,
                                                       const uint64_t* __restrict__ tile_base,
                                                       uint64_t* __restrict__ new_off) {
    __shared__ uint64_t smem[17];
    const uint32_t base = blockIdx.x * 1024u + threadIdx.x * 4u;
    uint64_t r[4];
    uint64_t s = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        r[k] = (base + k < n) ? round16(len[base + k]) : 0;
        s += r[k];
    }
    uint64_t total;
    uint64_t ex = block_exclusive_scan(s, smem, &total) + tile_base[blockIdx.x];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (base + k < n) new_off[base + k] = ex;
        ex += r[k];
    }
}

// one wavefront per block; 4-byte moves when source and destination agree mod 4
__global__ __launch_bounds__(256) void k_repack(const uint8_t* __restrict__ src, const uint64_t* __restrict__ old_off,
                                                const uint32_t* __restrict__ len,
                                                const uint64_t* __restrict__ new_off, uint32_t n,
                                                uint8_t* __restrict__ dst) {
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint32_t nwaves = (gridDim.x * blockDim.x) >> 6;
    for (uint32_t i = wave; i < n; i += nwaves) {
        const uint8_t* s = src + old_off[i];
        uint8_t* d = dst + new_off[i];
        const uint32_t L = len[i];
        const uint32_t padded = L == 0 ? 128u : (L + 127u) & ~127u;
        if ((reinterpret_cast<uintptr_t>(s) & 3u) == 0) {
            const uint32_t words = L >> 2;
            const uint32_t* s4 = reinterpret_cast<const uint32_t*>(s);
            uint32_t* d4 = reinterpret_cast<uint32_t*>(d);
            for (uint32_t k = lane; k < words; k += 64) d4[k] = s4[k];
            for (uint32_t k = (words << 2) + lane; k < padded; k += 64) d[k] = k < L ? s[k] : 0;
        } else {
            for (uint32_t k = lane; k < padded; k += 64) d[k] = k < L ? s[k] : 0;
        }
    }
}

__global__ void k_check_aligned(const uint64_t* __restrict__ off, uint32_t n, uint32_t* __restrict__ flag) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const bool bad = i < n && (off[i] & 15ull) != 0;
    if (__any(bad) && (threadIdx.x & 63) == 0) atomicOr(flag, 1u);
}

int launch_aligned_offsets(ipcfp_ctx* ctx, const uint32_t* len_d, uint32_t n, uint64_t* new_off_d,
                           uint64_t* total_d, uint64_t* scratch_d) {
    const uint32_t ntiles = div_up(n, 1024);
    if (n == 0) {
        IPCFP_HIP(ctx, hipMemsetAsync(total_d, 0, sizeof(uint64_t), ctx->stream));
        return IPCFP_OK;
    }
    hipLaunchKernelGGL(k_tile_sums, dim3(ntiles), dim3(256), 0, ctx->stream, len_d, n, scratch_d);
    hipLaunchKernelGGL(k_scan_tiles, dim3(1), dim3(1024), 0, ctx->stream, scratch_d, ntiles, total_d);
    hipLaunchKernelGGL(k_apply_offsets, dim3(ntiles), dim3(256), 0, ctx->stream, len_d, n, scratch_d, new_off_d);
    IPCFP_HIP(ctx, hipGetLastError());
    return IPCFP_OK;
}

int launch_repack(ipcfp_ctx* ctx, const uint8_t* src, const uint64_t* old_off, const uint32_t* len,
                  const uint64_t* new_off, uint32_t n, uint8_t* dst) {
    if (n == 0) return IPCFP_OK;
    const uint32_t waves = n < 8192u * 4u ? n : 8192u * 4u;
    hipLaunchKernelGGL(k_repack, dim3(div_up(waves, 4)), dim3(256), 0, ctx->stream, src, old_off, len, new_off, n,
                       dst);
    IPCFP_HIP(ctx, hipGetLastError());
    return IPCFP_OK;
}

int launch_check_aligned(ipcfp_ctx* ctx, const uint64_t* off_d, uint32_t n, uint32_t* flag_d) {
    if (n == 0) return IPCFP_OK;
    hipLaunchKernelGGL(k_check_aligned, dim3(div_up(n, 256)), dim3(256), 0, ctx->stream, off_d, n, flag_d);
    IPCFP_HIP(ctx, hipGetLastError());
    return IPCFP_OK;
}

}  // namespace ipcfp
