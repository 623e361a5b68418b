// csrc/kernels/scan_dev.h — wave64 / workgroup prefix-sum building blocks.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace ipcfp {

// Inclusive scan across the 64 lanes of a wavefront.
template <typename T>
__device__ __forceinline__ T wave_inclusive_scan(T v) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        T up = __shfl_up(v, d, 64);
        if (lane >= d) v += up;
    }
    return v;
}

// Exclusive scan across a workgroup of up to 1024 threads (16 wavefronts).
// `smem` must hold 17 elements of T.  Returns the exclusive prefix of `v`;
// *total receives the workgroup sum (valid in every thread).
template <typename T>
__device__ __forceinline__ T block_exclusive_scan(T v, T* smem, T* total) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int nwaves = (blockDim.x + 63) >> 6;
    T inc = wave_inclusive_scan(v);
    if (lane == 63) smem[wave] = inc;
    __syncthreads();
    if (threadIdx.x == 0) {
        T run = 0;
        for (int w = 0; w < nwaves; ++w) {
            T s = smem[w];
            smem[w] = run;
            run += s;
        }
        smem[16] = run;
    }
    __syncthreads();
    T base = smem[wave];
    *total = smem[16];
    __syncthreads();  // smem may be reused by the caller
    return base + inc - v;
}

}  // namespace ipcfp
