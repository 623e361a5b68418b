// csrc/kernels/blake2b_cid.hip — K1: Blake2b-256 CID check of every witness block.
//
// Replaces `Block::cid` / `CborStore::put_cbor(.., Code::Blake2b256)` — reference
// call site src/proofs/events/utils.rs:65-72 (TxMeta re-hash), generalised to every
// witness block (README.md:401 "CID verification").
//
// Work mapping (gfx950): one hash per LANE, 64 independent hashes per wavefront
// (see blake2b_dev.h for why not one hash per wavefront).  Lanes follow the
// SCHEDULE: block ids sorted by 128-byte chunk count (longest first), so the 64
// lanes of a wavefront walk chunk chains of (almost) equal length and exec-mask
// divergence is confined to bucket edges.  The arena and K1's metadata are laid
// out PHYSICALLY in schedule order: neighbouring lanes read neighbouring blocks,
// so the 128-byte lines two blocks share are fetched once, and the per-lane
// metadata (offset, length, id, claimed CID) is read coalesced.  (A length-sorted
// schedule over an arrival-order arena re-fetched boundary lines and gathered
// metadata: 970 MB of HBM traffic for 512 MB of algorithmic bytes on the tipset
// witness — profiles/r01_rocprofv3_pmc_tipset.txt.)
//
// Memory: every lane streams its own block with 16-byte loads, eight per
// 128-byte chunk; the next chunk is loaded into a second register set while the
// current one is compressed, so the ≈900-cycle HBM latency hides under ≈5000
// cycles of VALU work per chunk.  Each 128-byte line is fetched once and fully
// used.  Algorithmic bytes per block: len + 40 (claimed CID) + 16 (offset, len,
// id) — DESIGN.md §K1.
#include <hip/hip_runtime.h>

#include "../common.h"
#include "blake2b_dev.h"
#include "launch.h"

// Register budget: the default (130 VGPRs → 3 waves/SIMD).  Forcing 4 waves/SIMD (≤128 VGPRs) was measured
// and is not faster (2.06 vs 2.10-2.15 TB/s at 4 M × 1 KiB): the VALU pipe is already ≈85-95 % busy.
#ifndef IPCFP_K1_WAVES
#define IPCFP_K1_WAVES 1
#endif

namespace ipcfp {

// chunks a lane must compress for a block of `len` bytes (≥ 1: the empty message
// is one all-zero final chunk)
__device__ __forceinline__ uint32_t chunk_count(uint32_t len) { return len == 0 ? 1u : (len + 127u) >> 7; }

// Shared body: hash block `i`, leaving the state in h[].
template <int MODE>
__device__ __forceinline__ void hash_block(const uint8_t* __restrict__ arena, uint64_t o, uint32_t L,
                                           uint64_t h[8]) {
    b2b::init256(h);
    const uint8_t* p = arena + o;
    const uint32_t nfull = chunk_count(L) - 1;  // non-final chunks
    uint64_t m[16];
    b2b::load_chunk(m, p);
    uint64_t t = 0;
    for (uint32_t c = 0; c < nfull; ++c) {
        uint64_t mn[16];
        b2b::load_chunk(mn, p + 128ull * (c + 1));  // c+1 <= nfull: inside the block's padded span
        t += 128;
        b2b::compress<MODE>(h, m, t, false);
#pragma unroll
        for (int k = 0; k < 16; ++k) m[k] = mn[k];
    }
    const uint32_t rem = L - nfull * 128u;  // 0 (empty message) .. 128
    b2b::mask_tail(m, rem);
    t += rem;
    b2b::compress<MODE>(h, m, t, true);
}

// The LDS-message variant of hash_block: the chunk being compressed lives in LDS ([word][lane]), the next one is
// prefetched into registers while it is compressed.
template <int WG>
__device__ __forceinline__ void hash_block_lds(const uint8_t* __restrict__ arena, uint64_t o, uint32_t L, uint64_t h[8],
                                               uint64_t (*sm)[WG]) {
    b2b::init256(h);
    const uint8_t* p = arena + o;
    const uint32_t nfull = chunk_count(L) - 1;  // non-final chunks
    uint64_t* lm = &sm[0][threadIdx.x];
    uint64_t m[16];
    b2b::load_chunk(m, p);
    uint64_t t = 0;
    for (uint32_t c = 0; c < nfull; ++c) {
#pragma unroll
        for (int k = 0; k < 16; ++k) lm[k * WG] = m[k];
        b2b::load_chunk(m, p + 128ull * (c + 1));  // in flight while the staged chunk is compressed
        t += 128;
        b2b::compress_lds(h, lm, WG, t, false);
    }
    const uint32_t rem = L - nfull * 128u;
    b2b::mask_tail(m, rem);
    t += rem;
#pragma unroll
    for (int k = 0; k < 16; ++k) lm[k * WG] = m[k];
    b2b::compress_lds(h, lm, WG, t, true);
}

// K1's per-lane metadata (K1Meta, witness_dev.h) is stored in SCHEDULE order so a wavefront reads it coalesced.

template <int MODE>
__global__ __launch_bounds__(256, IPCFP_K1_WAVES) void k_blake2b256_cid(const uint8_t* __restrict__ arena,
                                                       const K1Meta* __restrict__ meta,
                                                       const uint8_t* __restrict__ sched_cids40, uint32_t n,
                                                       uint32_t* __restrict__ ok_bits,
                                                       uint8_t* __restrict__ status,
                                                       unsigned long long* __restrict__ counters) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const K1Meta mt = meta[t];
    const uint32_t i = mt.id;

    // claimed CID: 01 <codec> a0 e4 02 20 ‖ digest[32] ‖ 00 00   (40-byte slot)
    const uint64_t* cw = reinterpret_cast<const uint64_t*>(sched_cids40 + 40ull * t);
    const uint64_t w0 = cw[0], w1 = cw[1], w2 = cw[2], w3 = cw[3], w4 = cw[4];
    const bool is_b2b = ((w0 & 0x0000FFFFFFFF00FFULL) == 0x00002002e4a00001ULL) && ((w0 & 0x8000ULL) == 0) &&
                        ((w4 >> 48) == 0);
    uint8_t st = IPCFP_CID_UNCHECKED;
    if (is_b2b) {
        uint64_t h[8];
        hash_block<MODE>(arena, mt.off, mt.len, h);
        const uint64_t e0 = (w0 >> 48) | (w1 << 16);
        const uint64_t e1 = (w1 >> 48) | (w2 << 16);
        const uint64_t e2 = (w2 >> 48) | (w3 << 16);
        const uint64_t e3 = (w3 >> 48) | (w4 << 16);
        const bool ok = ((h[0] ^ e0) | (h[1] ^ e1) | (h[2] ^ e2) | (h[3] ^ e3)) == 0;
        st = ok ? IPCFP_CID_OK : IPCFP_CID_MISMATCH;
        if (ok) atomicOr(&ok_bits[i >> 5], 1u << (i & 31));
        else atomicAdd(&counters[0], 1ull);
    }
    status[i] = st;
}

// K1 with the message words staged in LDS (IPCFP_B2B_MODE=3): same outputs, 64 threads per workgroup.
__global__ __launch_bounds__(64, 5) void k_blake2b256_cid_lds(const uint8_t* __restrict__ arena, const K1Meta* __restrict__ meta,
                                                              const uint8_t* __restrict__ sched_cids40, uint32_t n,
                                                              uint32_t* __restrict__ ok_bits, uint8_t* __restrict__ status,
                                                              unsigned long long* __restrict__ counters) {
    __shared__ uint64_t sm[16][64];
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const K1Meta mt = meta[t];
    const uint32_t i = mt.id;
    const uint64_t* cw = reinterpret_cast<const uint64_t*>(sched_cids40 + 40ull * t);
    const uint64_t w0 = cw[0], w1 = cw[1], w2 = cw[2], w3 = cw[3], w4 = cw[4];
    const bool is_b2b = ((w0 & 0x0000FFFFFFFF00FFULL) == 0x00002002e4a00001ULL) && ((w0 & 0x8000ULL) == 0) &&
                        ((w4 >> 48) == 0);
    uint8_t st = IPCFP_CID_UNCHECKED;
    if (is_b2b) {
        uint64_t h[8];
        hash_block_lds<64>(arena, mt.off, mt.len, h, sm);
        const uint64_t e0 = (w0 >> 48) | (w1 << 16);
        const uint64_t e1 = (w1 >> 48) | (w2 << 16);
        const uint64_t e2 = (w2 >> 48) | (w3 << 16);
        const uint64_t e3 = (w3 >> 48) | (w4 << 16);
        const bool ok = ((h[0] ^ e0) | (h[1] ^ e1) | (h[2] ^ e2) | (h[3] ^ e3)) == 0;
        st = ok ? IPCFP_CID_OK : IPCFP_CID_MISMATCH;
        if (ok) atomicOr(&ok_bits[i >> 5], 1u << (i & 31));
        else atomicAdd(&counters[0], 1ull);
    }
    status[i] = st;
}

// Raw digests (ipcfp_blake2b256_batch): out32[id] = Blake2b-256(block).
template <int MODE>
__global__ __launch_bounds__(256) void k_blake2b256_raw(const uint8_t* __restrict__ arena,
                                                       const K1Meta* __restrict__ meta, uint32_t n,
                                                       uint64_t* __restrict__ out32) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const K1Meta mt = meta[t];
    uint64_t h[8];
    hash_block<MODE>(arena, mt.off, mt.len, h);
    uint64_t* o = out32 + 4ull * mt.id;
    o[0] = h[0];
    o[1] = h[1];
    o[2] = h[2];
    o[3] = h[3];
}

// ---- physical layout in schedule order ----
__global__ __launch_bounds__(256) void k_gather_len(const uint32_t* __restrict__ order, const uint32_t* __restrict__ len,
                                                    uint32_t n, uint32_t* __restrict__ sched_len) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) sched_len[t] = len[order[t]];
}

__global__ __launch_bounds__(256) void k_place(const uint32_t* __restrict__ order, const uint32_t* __restrict__ sched_len,
                                               const uint64_t* __restrict__ sched_off, uint32_t n,
                                               uint64_t* __restrict__ off_by_id, K1Meta* __restrict__ meta) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const uint32_t id = order[t];
    off_by_id[id] = sched_off[t];
    meta[t] = K1Meta{sched_off[t], sched_len[t], id};
}

__global__ __launch_bounds__(256) void k_gather_cids(const uint32_t* __restrict__ order, const uint8_t* __restrict__ cids,
                                                     uint32_t n, uint8_t* __restrict__ sched_cids) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;  // one u64 word per thread: 5 words per CID
    if (t >= n * 5u) return;
    const uint32_t b = t / 5u, wd = t % 5u;
    reinterpret_cast<uint64_t*>(sched_cids)[t] = reinterpret_cast<const uint64_t*>(cids)[uint64_t(order[b]) * 5u + wd];
}

// ---- lane schedule: counting sort of block ids by chunk count, longest first ----
// class = min(chunks, 255); 256 bins.
__global__ void k_chunk_histogram(const uint32_t* __restrict__ len, uint32_t n, uint32_t* __restrict__ bins) {
    __shared__ uint32_t local[256];
    local[threadIdx.x] = 0;
    __syncthreads();
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        uint32_t c = chunk_count(len[i]);
        atomicAdd(&local[c > 255 ? 255 : c], 1u);
    }
    __syncthreads();
    if (local[threadIdx.x]) atomicAdd(&bins[threadIdx.x], local[threadIdx.x]);
}

// bins[c] ← start position of class c when classes are laid out 255,254,…,0.
__global__ void k_chunk_bin_starts(uint32_t* __restrict__ bins) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        uint32_t run = 0;
        for (int c = 255; c >= 0; --c) {
            uint32_t k = bins[c];
            bins[c] = run;
            run += k;
        }
    }
}

// Positions inside a class are handed out per TILE of 4096 consecutive blocks: the workgroup counts its tile's classes in
// LDS, reserves one range per class with ONE global atomic each, and hands the positions out from LDS cursors, aggregated
// per wavefront so that neighbouring blocks stay neighbours (DRAM page locality).  (Round 3 took the positions from the
// global cursors directly, one atomic per wavefront and class: a Filecoin witness has three or four classes that matter
// — blocks of 2-4 lines — so 20 k wavefronts queued up on the same four words of the L2: 630-790 µs for 1.3 M blocks,
// profiles/r03_final_kernel_stats.txt, every microsecond of it in front of a from-host witness's payload copy.)
constexpr uint32_t kScatterTile = 4096;
__global__ __launch_bounds__(256) void k_chunk_scatter(const uint32_t* __restrict__ len, uint32_t n, uint32_t* __restrict__ cursors,
                                                       uint32_t* __restrict__ order) {
    __shared__ uint32_t hist[256], cur[256];
    hist[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t tile0 = blockIdx.x * kScatterTile;
    const uint32_t lane = threadIdx.x & 63;
    // pass 1: the tile's class counts (wave-aggregated LDS atomics: a wavefront's 64 blocks are of 1-3 classes)
    for (uint32_t k = 0; k < kScatterTile / 256; ++k) {
        const uint32_t i = tile0 + k * 256 + threadIdx.x;
        const bool active = i < n;
        uint32_t c = 0;
        if (active) {
            c = chunk_count(len[i]);
            c = c > 255 ? 255 : c;
        }
        unsigned long long todo = __ballot(active);
        while (todo) {
            const int leader = __ffsll((long long)todo) - 1;
            const uint32_t lc = __shfl(c, leader, 64);
            const unsigned long long same = __ballot(active && c == lc) & todo;
            if ((int)lane == leader) atomicAdd(&hist[lc], (uint32_t)__popcll(same));
            todo &= ~same;
        }
    }
    __syncthreads();
    {   // one range per class present in the tile
        const uint32_t h = hist[threadIdx.x];
        cur[threadIdx.x] = h ? atomicAdd(&cursors[threadIdx.x], h) : 0u;
    }
    __syncthreads();
    // pass 2: positions from the LDS cursors
    for (uint32_t k = 0; k < kScatterTile / 256; ++k) {
        const uint32_t i = tile0 + k * 256 + threadIdx.x;
        const bool active = i < n;
        uint32_t c = 0;
        if (active) {
            c = chunk_count(len[i]);
            c = c > 255 ? 255 : c;
        }
        unsigned long long todo = __ballot(active);
        while (todo) {
            const int leader = __ffsll((long long)todo) - 1;
            const uint32_t lc = __shfl(c, leader, 64);
            const unsigned long long same = __ballot(active && c == lc) & todo;
            uint32_t base = 0;
            if ((int)lane == leader) base = atomicAdd(&cur[lc], (uint32_t)__popcll(same));
            base = __shfl(base, leader, 64);
            if (active && c == lc) {
                const uint32_t rank = __popcll(same & ((1ull << lane) - 1ull));
                order[base + rank] = i;
            }
            todo &= ~same;
        }
    }
}

// ------------------------------ launchers -----------------------------------
int launch_chunk_order(ipcfp_ctx* ctx, const uint32_t* len_d, uint32_t n, uint32_t* bins_d /*256*/,
                       uint32_t* order_d) {
    IPCFP_HIP(ctx, hipMemsetAsync(bins_d, 0, 256 * sizeof(uint32_t), ctx->stream));
    if (n == 0) return IPCFP_OK;
    const uint32_t hb = div_up(n, 256) < 2048 ? div_up(n, 256) : 2048;
    hipLaunchKernelGGL(k_chunk_histogram, dim3(hb), dim3(256), 0, ctx->stream, len_d, n, bins_d);
    hipLaunchKernelGGL(k_chunk_bin_starts, dim3(1), dim3(64), 0, ctx->stream, bins_d);
    hipLaunchKernelGGL(k_chunk_scatter, dim3(div_up(n, kScatterTile)), dim3(256), 0, ctx->stream, len_d, n, bins_d,
                       order_d);
    IPCFP_HIP(ctx, hipGetLastError());
    return IPCFP_OK;
}

// Lay the blocks out physically in schedule order: order[] from the counting sort, offsets from a
// prefix sum over the schedule, off_by_id[] for every other kernel, K1's metadata coalesced.
// sched_len_d / sched_off_d are scratch (n each); *total_d receives the arena payload size.
int launch_k1_layout(ipcfp_ctx* ctx, const uint32_t* len_d, uint32_t n, uint32_t* bins_d, uint32_t* order_d,
                     uint32_t* sched_len_d, uint64_t* sched_off_d, uint64_t* total_d, uint64_t* scan_scratch_d,
                     uint64_t* off_by_id_d, void* meta_d) {
    int rc = launch_chunk_order(ctx, len_d, n, bins_d, order_d);
    if (rc) return rc;
    if (n == 0) return launch_aligned_offsets(ctx, sched_len_d, 0, sched_off_d, total_d, scan_scratch_d);
    const dim3 g(div_up(n, 256)), b(256);
    hipLaunchKernelGGL(k_gather_len, g, b, 0, ctx->stream, order_d, len_d, n, sched_len_d);
    rc = launch_aligned_offsets(ctx, sched_len_d, n, sched_off_d, total_d, scan_scratch_d);
    if (rc) return rc;
    hipLaunchKernelGGL(k_place, g, b, 0, ctx->stream, order_d, sched_len_d, sched_off_d, n, off_by_id_d,
                       static_cast<K1Meta*>(meta_d));
    IPCFP_HIP(ctx, hipGetLastError());
    return IPCFP_OK;
}

int launch_gather_cids(ipcfp_ctx* ctx, const uint32_t* order_d, const uint8_t* cids_d, uint32_t n, uint8_t* sched_cids_d) {
    if (n == 0) return IPCFP_OK;
    hipLaunchKernelGGL(k_gather_cids, dim3(div_up(uint64_t(n) * 5, 256)), dim3(256), 0, ctx->stream, order_d, cids_d, n,
                       sched_cids_d);
    IPCFP_HIP(ctx, hipGetLastError());
    return IPCFP_OK;
}

// K1 runs on the context's second stream (results are independent of everything the main stream does
// after witness creation); ipcfp_ctx_sync / profile reads wait for both.
int launch_blake2b256_cid(ipcfp_ctx* ctx, const uint8_t* arena, const void* meta, const uint8_t* sched_cids40, uint32_t n,
                          uint32_t* ok_bits, uint8_t* status, unsigned long long* counters) {
    hipStream_t s = ctx->stream_k1;
    IPCFP_HIP(ctx, hipMemsetAsync(ok_bits, 0, size_t(div_up(n, 32)) * 4, s));
    IPCFP_HIP(ctx, hipMemsetAsync(counters, 0, sizeof(unsigned long long), s));
    if (n == 0) return IPCFP_OK;
    {
        ProfileScope prof(ctx, IPCFP_K_BLAKE2B_CID, s);
        const uint32_t wg = ctx->b2b_wg;
        const K1Meta* m = static_cast<const K1Meta*>(meta);
        if (ctx->b2b_mode == 1)
            hipLaunchKernelGGL(k_blake2b256_cid<1>, dim3(div_up(n, wg)), dim3(wg), 0, s, arena, m, sched_cids40, n, ok_bits,
                               status, counters);
        else if (ctx->b2b_mode == 2)
            hipLaunchKernelGGL(k_blake2b256_cid<2>, dim3(div_up(n, wg)), dim3(wg), 0, s, arena, m, sched_cids40, n, ok_bits,
                               status, counters);
        else if (ctx->b2b_mode == 3)
            hipLaunchKernelGGL(k_blake2b256_cid_lds, dim3(div_up(n, 64)), dim3(64), 0, s, arena, m, sched_cids40, n, ok_bits,
                               status, counters);
        else
            hipLaunchKernelGGL(k_blake2b256_cid<0>, dim3(div_up(n, wg)), dim3(wg), 0, s, arena, m, sched_cids40, n, ok_bits,
                               status, counters);
    }
    IPCFP_HIP(ctx, hipGetLastError());
    return IPCFP_OK;
}

int launch_blake2b256_raw(ipcfp_ctx* ctx, const uint8_t* arena, const void* meta, uint32_t n, uint8_t* out32) {
    if (n == 0) return IPCFP_OK;
    {
        ProfileScope prof(ctx, IPCFP_K_BLAKE2B_RAW);
        const uint32_t wg = ctx->b2b_wg;
        const K1Meta* m = static_cast<const K1Meta*>(meta);
        if (ctx->b2b_mode == 1)
            hipLaunchKernelGGL(k_blake2b256_raw<1>, dim3(div_up(n, wg)), dim3(wg), 0, ctx->stream, arena, m, n,
                               reinterpret_cast<uint64_t*>(out32));
        else
            hipLaunchKernelGGL(k_blake2b256_raw<0>, dim3(div_up(n, wg)), dim3(wg), 0, ctx->stream, arena, m, n,
                               reinterpret_cast<uint64_t*>(out32));
    }
    IPCFP_HIP(ctx, hipGetLastError());
    return IPCFP_OK;
}

}  // namespace ipcfp
