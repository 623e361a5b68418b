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
// what a block adds to the running offset: its padded size in the arena (PAD), or its length (blocks back to back in the
// caller's buffer: ipcfp_witness_create_packed rebuilds the offset table it did not upload)
template <bool PAD>
__device__ __forceinline__ uint64_t step_of(uint32_t x) { return PAD ? round16(x) : uint64_t(x); }

// pass 1: per-1024-element tile sums of round16(len)
template <bool PAD>
__global__ __launch_bounds__(256) void k_tile_sums(const uint32_t* __restrict__ len, uint32_t n,
                                                   uint64_t* __restrict__ tile_sums) {
    __shared__ uint64_t smem[17];
    const uint32_t base = blockIdx.x * 1024u + threadIdx.x * 4u;
    uint64_t s = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (base + k < n) s += step_of<PAD>(len[base + k]);
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
template <bool PAD>
__global__ __launch_bounds__(256) void k_apply_offsets(const uint32_t* __restrict__ len, uint32_t n,
                                                       const uint64_t* __restrict__ tile_base,
                                                       uint64_t* __restrict__ new_off) {
    __shared__ uint64_t smem[17];
    const uint32_t base = blockIdx.x * 1024u + threadIdx.x * 4u;
    uint64_t r[4];
    uint64_t s = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        r[k] = (base + k < n) ? step_of<PAD>(len[base + k]) : 0;
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

// Half a wavefront per block, 16 destination bytes per lane: a block of up to 512 bytes — nearly every block of a
// Filecoin witness — is ONE round of loads and one of stores, two blocks in flight per wavefront.  The destination is
// line-aligned; the source starts anywhere, so a lane reads the three ALIGNED 8-byte words around its 16 bytes and
// funnel-shifts them (only words that hold at least one byte of the block are touched: nothing is read beyond the
// 8-byte word of the block's last byte).  (Round 3: a wavefront per block, 4 bytes per lane and a byte-wise tail —
// 222-477 µs for the 0.44 GB tipset witness, one block of ≈ 340 bytes in flight per wavefront.)
__global__ __launch_bounds__(256) void k_repack(const uint8_t* __restrict__ src, const uint64_t* __restrict__ old_off,
                                                const uint32_t* __restrict__ len,
                                                const uint64_t* __restrict__ new_off, uint32_t n,
                                                uint8_t* __restrict__ dst) {
    const uint32_t sub = threadIdx.x & 31;
    const uint32_t group = (blockIdx.x * blockDim.x + threadIdx.x) >> 5;
    const uint32_t ngroups = (gridDim.x * blockDim.x) >> 5;
    for (uint32_t i = group; i < n; i += ngroups) {
        const uint64_t so = old_off[i];
        const uint32_t L = len[i];
        uint8_t* d = dst + new_off[i];
        const uint32_t padded = L == 0 ? 128u : (L + 127u) & ~127u;
        const uintptr_t s = reinterpret_cast<uintptr_t>(src) + so;
        const uintptr_t last_word = L ? ((s + L - 1u) & ~uintptr_t(7)) : 0;  // the aligned word of the block's last byte
        for (uint32_t u = sub * 16u; u < padded; u += 512u) {
            uint64_t w0 = 0, w1 = 0;
            if (u < L) {
                const uintptr_t a = s + u;
                const uintptr_t q = a & ~uintptr_t(7);
                const uint32_t sh = uint32_t(a & 7u) * 8u;
                const uint64_t q0 = *reinterpret_cast<const uint64_t*>(q);
                const uint64_t q1 = q + 8u <= last_word ? *reinterpret_cast<const uint64_t*>(q + 8u) : 0ull;
                const uint64_t q2 = q + 16u <= last_word ? *reinterpret_cast<const uint64_t*>(q + 16u) : 0ull;
                w0 = (q0 >> sh) | ((q1 << 1) << (63u - sh));
                w1 = (q1 >> sh) | ((q2 << 1) << (63u - sh));
                const uint32_t valid = L - u;  // bytes of this unit that belong to the block (≥ 1)
                if (valid < 8u) {
                    w0 &= (1ull << (8u * valid)) - 1ull;
                    w1 = 0;
                } else if (valid < 16u) {
                    w1 &= valid == 8u ? 0ull : (1ull << (8u * (valid - 8u))) - 1ull;
                }
            }
            *reinterpret_cast<ulonglong2*>(d + u) = make_ulonglong2(w0, w1);
        }
    }
}

__global__ void k_check_aligned(const uint64_t* __restrict__ off, uint32_t n, uint32_t* __restrict__ flag) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const bool bad = i < n && (off[i] & 15ull) != 0;
    if (__any(bad) && (threadIdx.x & 63) == 0) atomicOr(flag, 1u);
}

template <bool PAD>
static int launch_offsets(ipcfp_ctx* ctx, const uint32_t* len_d, uint32_t n, uint64_t* new_off_d, uint64_t* total_d,
                          uint64_t* scratch_d) {
    const uint32_t ntiles = div_up(n, 1024);
    if (n == 0) {
        IPCFP_HIP(ctx, hipMemsetAsync(total_d, 0, sizeof(uint64_t), ctx->stream));
        return IPCFP_OK;
    }
    hipLaunchKernelGGL(k_tile_sums<PAD>, dim3(ntiles), dim3(256), 0, ctx->stream, len_d, n, scratch_d);
    hipLaunchKernelGGL(k_scan_tiles, dim3(1), dim3(1024), 0, ctx->stream, scratch_d, ntiles, total_d);
    hipLaunchKernelGGL(k_apply_offsets<PAD>, dim3(ntiles), dim3(256), 0, ctx->stream, len_d, n, scratch_d, new_off_d);
    IPCFP_HIP(ctx, hipGetLastError());
    return IPCFP_OK;
}

int launch_aligned_offsets(ipcfp_ctx* ctx, const uint32_t* len_d, uint32_t n, uint64_t* new_off_d,
                           uint64_t* total_d, uint64_t* scratch_d) {
    return launch_offsets<true>(ctx, len_d, n, new_off_d, total_d, scratch_d);
}

int launch_tight_offsets(ipcfp_ctx* ctx, const uint32_t* len_d, uint32_t n, uint64_t* off_d, uint64_t* total_d,
                         uint64_t* scratch_d) {
    return launch_offsets<false>(ctx, len_d, n, off_d, total_d, scratch_d);
}

// cids40[i] = prefix ‖ digests32[i], zero-padded to the 40-byte slot (one lane per 8-byte word of the table: coalesced
// stores; a slot's five words take bytes of the prefix and of at most two digest words)
__global__ __launch_bounds__(256) void k_expand_cids(const uint8_t* __restrict__ digests32, uint32_t n, uint64_t prefix,
                                                     uint32_t prefix_len, uint64_t* __restrict__ cids40) {
    const uint64_t t = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (t >= uint64_t(n) * 5u) return;
    const uint32_t i = uint32_t(t / 5u), j = uint32_t(t % 5u);
    const uint8_t* d = digests32 + size_t(i) * 32u;
    uint64_t v = 0;
#pragma unroll
    for (uint32_t b = 0; b < 8; ++b) {
        const uint32_t pos = 8u * j + b;  // byte of the slot
        uint32_t byte = 0;
        if (pos < prefix_len) byte = uint32_t(prefix >> (8u * pos)) & 0xffu;
        else if (pos < prefix_len + 32u) byte = d[pos - prefix_len];
        v |= uint64_t(byte) << (8u * b);
    }
    cids40[t] = v;
}

// the blocks whose CID is of another form: their slots as given
__global__ __launch_bounds__(256) void k_escape_cids(const uint32_t* __restrict__ index, const uint64_t* __restrict__ slots,
                                                     uint32_t n_esc, uint32_t n, uint64_t* __restrict__ cids40) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_esc * 5u) return;
    const uint32_t e = t / 5u, j = t % 5u;
    const uint32_t i = index[e];
    if (i < n) cids40[size_t(i) * 5u + j] = slots[size_t(e) * 5u + j];
}

int launch_expand_cids(ipcfp_ctx* ctx, const uint8_t* digests32_d, uint32_t n, const uint8_t* prefix, uint32_t prefix_len,
                       const uint32_t* esc_index_d, const uint8_t* esc_cids40_d, uint32_t n_esc, uint8_t* cids40_d) {
    if (n == 0) return IPCFP_OK;
    uint64_t p = 0;
    for (uint32_t b = 0; b < prefix_len && b < 8; ++b) p |= uint64_t(prefix[b]) << (8u * b);
    const uint64_t words = uint64_t(n) * 5u;
    hipLaunchKernelGGL(k_expand_cids, dim3(uint32_t((words + 255) / 256)), dim3(256), 0, ctx->stream, digests32_d, n, p, prefix_len,
                       reinterpret_cast<uint64_t*>(cids40_d));
    if (n_esc)
        hipLaunchKernelGGL(k_escape_cids, dim3(div_up(n_esc * 5u, 256)), dim3(256), 0, ctx->stream, esc_index_d,
                           reinterpret_cast<const uint64_t*>(esc_cids40_d), n_esc, n, reinterpret_cast<uint64_t*>(cids40_d));
    IPCFP_HIP(ctx, hipGetLastError());
    return IPCFP_OK;
}

int launch_repack(ipcfp_ctx* ctx, const uint8_t* src, const uint64_t* old_off, const uint32_t* len,
                  const uint64_t* new_off, uint32_t n, uint8_t* dst) {
    if (n == 0) return IPCFP_OK;
    const uint32_t groups = n < 8192u * 8u ? n : 8192u * 8u;  // half-wavefronts
    hipLaunchKernelGGL(k_repack, dim3(div_up(groups, 8)), dim3(256), 0, ctx->stream, src, old_off, len, new_off, n,
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
