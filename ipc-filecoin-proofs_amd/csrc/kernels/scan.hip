// csrc/kernels/scan.hip — device-wide exclusive prefix sum of u32 counts (three small kernels:
// tile sums, scan of tile sums, apply).  Used wherever a level of a tree walk places its children
// or matches in index order (amt_enum.hip, event_scan.hip, exec-order dedup).
#include <hip/hip_runtime.h>

#include "../common.h"
#include "launch.h"
#include "scan_dev.h"

namespace ipcfp {

__global__ __launch_bounds__(256) void k_scan_tile_sums(const uint32_t* __restrict__ in, uint32_t n,
                                                        uint64_t* __restrict__ tile_sums) {
    __shared__ uint64_t smem[17];
    const uint32_t base = blockIdx.x * 1024u + threadIdx.x * 4u;
    uint64_t s = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (base + k < n) s += in[base + k];
    uint64_t total;
    (void)block_exclusive_scan(s, smem, &total);
    if (threadIdx.x == 0) tile_sums[blockIdx.x] = total;
}

// One workgroup of FOUR wavefronts (four items per lane), not sixteen: a 1024-thread workgroup needs sixteen free
// wavefront slots on ONE compute unit at the same moment, and beside K1's grid it waited 143 µs for them
// (profiles/r03_experiments.md); four find room anywhere.
__global__ __launch_bounds__(256) void k_scan_tiles_u64(uint64_t* __restrict__ tile_sums, uint32_t ntiles,
                                                        uint64_t* __restrict__ total_out) {
    __shared__ uint64_t smem[17];
    uint64_t carry = 0;
    for (uint32_t base = 0; base < ntiles; base += 1024) {
        const uint32_t i0 = base + threadIdx.x * 4u;
        uint64_t v[4];
        uint64_t s = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            v[k] = i0 + k < ntiles ? tile_sums[i0 + k] : 0;
            s += v[k];
        }
        uint64_t total;
        uint64_t ex = carry + block_exclusive_scan(s, smem, &total);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (i0 + k < ntiles) tile_sums[i0 + k] = ex;
            ex += v[k];
        }
        carry += total;
    }
    if (threadIdx.x == 0) *total_out = carry;
}

__global__ __launch_bounds__(256) void k_scan_apply(const uint32_t* __restrict__ in, uint32_t n,
                                                    const uint64_t* __restrict__ tile_base,
                                                    uint32_t* __restrict__ out) {
    __shared__ uint64_t smem[17];
    const uint32_t base = blockIdx.x * 1024u + threadIdx.x * 4u;
    uint32_t r[4];
    uint64_t s = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        r[k] = (base + k < n) ? in[base + k] : 0;
        s += r[k];
    }
    uint64_t total;
    uint64_t ex = block_exclusive_scan(s, smem, &total) + tile_base[blockIdx.x];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (base + k < n) out[base + k] = uint32_t(ex);
        ex += r[k];
    }
}

// n ≤ 4096: the whole scan in one workgroup (four wavefronts, sixteen items per lane), one launch (most tree levels are
// this small)
__global__ __launch_bounds__(256) void k_scan_small(const uint32_t* __restrict__ in, uint32_t n,
                                                    uint32_t* __restrict__ out, uint64_t* __restrict__ total_out) {
    __shared__ uint64_t smem[17];
    const uint32_t base = threadIdx.x * 16u;
    uint32_t r[16];
    uint64_t s = 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        r[k] = (base + k < n) ? in[base + k] : 0;
        s += r[k];
    }
    uint64_t total;
    uint64_t ex = block_exclusive_scan(s, smem, &total);
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        if (base + k < n) out[base + k] = uint32_t(ex);
        ex += r[k];
    }
    if (threadIdx.x == 0) *total_out = total;
}

// exclusive scan IN PLACE of ntiles u64 tile sums; *total_d = their sum (one workgroup)
int launch_scan_tiles_u64(ipcfp_ctx* ctx, uint64_t* tile_sums_d, uint32_t ntiles, uint64_t* total_d) {
    hipLaunchKernelGGL(k_scan_tiles_u64, dim3(1), dim3(256), 0, ctx->stream, tile_sums_d, ntiles, total_d);
    IPCFP_HIP(ctx, hipGetLastError());
    return IPCFP_OK;
}

// out[i] = sum of in[0..i); *total_d = sum of all.  scratch_d must hold div_up(n,1024)+1 u64.
int launch_scan_u32(ipcfp_ctx* ctx, const uint32_t* in_d, uint32_t n, uint32_t* out_d, uint64_t* total_d,
                    uint64_t* scratch_d) {
    if (n == 0) {
        IPCFP_HIP(ctx, hipMemsetAsync(total_d, 0, sizeof(uint64_t), ctx->stream));
        return IPCFP_OK;
    }
    if (n <= 4096) {
        hipLaunchKernelGGL(k_scan_small, dim3(1), dim3(256), 0, ctx->stream, in_d, n, out_d, total_d);
        IPCFP_HIP(ctx, hipGetLastError());
        return IPCFP_OK;
    }
    const uint32_t ntiles = div_up(n, 1024);
    hipLaunchKernelGGL(k_scan_tile_sums, dim3(ntiles), dim3(256), 0, ctx->stream, in_d, n, scratch_d);
    hipLaunchKernelGGL(k_scan_tiles_u64, dim3(1), dim3(256), 0, ctx->stream, scratch_d, ntiles, total_d);
    hipLaunchKernelGGL(k_scan_apply, dim3(ntiles), dim3(256), 0, ctx->stream, in_d, n, scratch_d, out_d);
    IPCFP_HIP(ctx, hipGetLastError());
    return IPCFP_OK;
}

}  // namespace ipcfp
