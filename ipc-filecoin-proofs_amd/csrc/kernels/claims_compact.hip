// csrc/kernels/claims_compact.hip — EventProof claims in transport form (include/ipcfp.h ipcfp_event_claim_compact_t)
// expanded on the device to the packed form the verify kernels read (claims_dev.h EventClaimPacked + blob).
//
// The claims of a bundle are the `event_proofs: Vec<EventProof>` of src/proofs/common/bundle.rs:36-45
// (fields: src/proofs/events/bundle.rs:5-23).  From host memory they are upload-bound: ≈ 200 bytes per claim in the packed
// form, 152 in the compact one, and the 4 ms they took of a 15 ms pass (profiles/r03_bench_final.json window_T2) are bytes
// over a 56 GB/s link.  Expansion is two prefix sums and one pass of byte moves: ≈ 0.3 GB of HBM traffic for a million claims.
#include <hip/hip_runtime.h>

#include "../common.h"
#include "claims_dev.h"
#include "launch.h"

namespace ipcfp {

struct ClaimCompact {  // == ipcfp_event_claim_compact_t
    uint64_t emitter;
    uint32_t exec_index, event_index;
    uint8_t digest[32];
    uint16_t data_len;
    uint8_t n_topics, topic_flags, flags, group;
    uint16_t reserved;
};
static_assert(sizeof(ClaimCompact) == sizeof(ipcfp_event_claim_compact_t) && sizeof(ClaimCompact) == 56, "compact claim layout");

struct ClaimGroups {  // travels as a kernel argument when it is small, else in device memory
    const ipcfp_event_claim_group_t* table;
    uint32_t n;
};

// per claim: bytes of its segment in the compact blob and in the packed one (0 / 0 for a record that is out of range)
__global__ __launch_bounds__(256) void k_compact_sizes(const ClaimCompact* __restrict__ cc, uint32_t n, uint32_t n_groups,
                                                       uint32_t* __restrict__ csize, uint32_t* __restrict__ psize) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t nt = cc[i].n_topics, dl = cc[i].data_len;
    // (a record that names no group still OWNS its segment of the compact blob: the records behind it find theirs)
    const bool shaped = nt <= IPCFP_COMPACT_MAX_TOPICS;
    csize[i] = shaped ? 32u * nt + dl : 0u;
    psize[i] = shaped && cc[i].group < n_groups ? 33u * nt + dl : 0u;
}

__global__ __launch_bounds__(256) void k_compact_expand(const ClaimCompact* __restrict__ cc, uint32_t n, ClaimGroups groups,
                                                        const uint8_t* __restrict__ cblob, uint64_t cblob_len,
                                                        const uint32_t* __restrict__ coff, const uint32_t* __restrict__ poff,
                                                        const uint64_t* __restrict__ total_c, const uint64_t* __restrict__ total_p,
                                                        EventClaimPacked* __restrict__ out, uint8_t* __restrict__ blob_out,
                                                        uint64_t cap_blob) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const ClaimCompact c = cc[i];
    const uint32_t nt = c.n_topics, dl = c.data_len;
    const uint32_t co = coff[i], po = poff[i];
    // The offsets are 32-bit prefix sums of sizes an untrusted record declares (up to 8 * 33 + 65 535 bytes each): once
    // the 64-bit totals pass 2^32 they have wrapped somewhere, a wrapped offset can pass the range checks below and two
    // records would move bytes into one segment.  Nobody can then say which records still own theirs: all of them are
    // ERR_BAD_CLAIM.  Below 2^32 nothing wrapped and the per-record checks are exact.
    const bool wrapped = (*total_c >> 32) != 0 || (*total_p >> 32) != 0;
    const bool ok = !wrapped && nt <= IPCFP_COMPACT_MAX_TOPICS && c.group < groups.n &&
                    uint64_t(co) + 32u * nt + dl <= cblob_len && uint64_t(po) + 33u * nt + dl <= cap_blob;
    EventClaimPacked p;
    p.parent_epoch = ok ? groups.table[c.group].parent_epoch : 0;
    p.child_epoch = ok ? groups.table[c.group].child_epoch : 0;
    p.exec_index = c.exec_index;
    p.event_index = c.event_index;
    p.emitter = c.emitter;
#pragma unroll
    for (int j = 0; j < 5; ++j) p.message.w[j] = 0;
    if (c.flags & EC_MSG_PARSED) {  // 01 71 a0 e4 02 20 ‖ digest, zero-padded to the 40-byte slot
        uint8_t slot[40];
        slot[0] = 0x01; slot[1] = 0x71; slot[2] = 0xa0; slot[3] = 0xe4; slot[4] = 0x02; slot[5] = 0x20;
#pragma unroll
        for (int b = 0; b < 32; ++b) slot[6 + b] = c.digest[b];
        slot[38] = slot[39] = 0;
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            uint64_t v = 0;
#pragma unroll
            for (int b = 0; b < 8; ++b) v |= uint64_t(slot[8 * j + b]) << (8 * b);
            p.message.w[j] = v;
        }
    }
    p.context = ok ? groups.table[c.group].tipset : 0xffffffffu;  // out of range: ERR_BAD_CLAIM, never followed
    p.flags = c.flags;
    p.n_topics = ok ? nt : 0u;
    p.topics_off = ok ? po : 0u;
    p.data_off = ok ? po + 33u * nt : 0u;
    p.data_len = ok ? dl : 0u;
    out[i] = p;
    if (!ok) return;
    const uint8_t* src = cblob + co;
    uint8_t* dst = blob_out + po;
    for (uint32_t t = 0; t < nt; ++t) {
        dst[33u * t] = (c.topic_flags >> t) & 1u;
        for (uint32_t b = 0; b < 32; ++b) dst[33u * t + 1u + b] = src[32u * t + b];
    }
    for (uint32_t b = 0; b < dl; ++b) dst[33u * nt + b] = src[32u * nt + b];
}

// ipcfp_verify_event_claims_slice: the records are consecutive claims of a larger batch and their blob offsets count from
// the start of THAT batch's blob, of which bytes [base, base + blob_len) were uploaded: rebase them; a claim whose topics
// or data do not lie inside that window is marked out of range (ERR_BAD_CLAIM, never followed).
// `miss` (nullable): the window was GUESSED (from the slice's ends, before the records were in HBM) — a record that lies inside
// the batch's blob [0, full_len) but outside the window sets *miss, and the caller repeats the call with the exact window.
// `order` (nullable): the slice came out of a binary search over a batch promised to be in exec_index order — a record below
// its predecessor or outside [key_lo, key_hi) breaks the promise (the search may have routed claims to the wrong shard) and
// sets *order; the caller returns IPCFP_E_INVALID instead of verdicts.  (exec_index is not rewritten here: the
// neighbour's read races with nothing.)
__global__ __launch_bounds__(256) void k_rebase_claims(EventClaimPacked* __restrict__ claims, uint32_t n, uint64_t base, uint64_t blob_len,
                                                       uint64_t full_len, uint32_t* __restrict__ miss, uint32_t* __restrict__ order,
                                                       uint64_t key_lo, uint64_t key_hi) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    EventClaimPacked c = claims[i];
    if (order && (c.exec_index < key_lo || c.exec_index >= key_hi || (i && claims[i - 1].exec_index > c.exec_index))) *order = 1u;
    const uint64_t t0 = c.topics_off, t1 = t0 + 33ull * c.n_topics, d0 = c.data_off, d1 = d0 + c.data_len;
    const bool t_ok = c.n_topics == 0 || (t0 >= base && t1 <= base + blob_len);
    const bool d_ok = c.data_len == 0 || (d0 >= base && d1 <= base + blob_len);
    if (t_ok && d_ok) {
        c.topics_off = c.n_topics ? uint32_t(t0 - base) : 0u;
        c.data_off = c.data_len ? uint32_t(d0 - base) : 0u;
    } else {
        if (miss && (c.n_topics == 0 || t1 <= full_len) && (c.data_len == 0 || d1 <= full_len)) *miss = 1u;  // (any lane: the same value)
        c.context = 0xffffffffu;
        c.n_topics = c.topics_off = c.data_off = c.data_len = 0u;
    }
    claims[i] = c;
}

// the window of the blob a run of records points into: win[0] = min start, win[1] = max end (win starts {~0, 0})
__global__ __launch_bounds__(256) void k_claims_window(const EventClaimPacked* __restrict__ claims, uint32_t n, unsigned long long* __restrict__ win) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long lo = ~0ull, hi = 0ull;
    if (i < n) {
        const EventClaimPacked c = claims[i];
        if (c.n_topics) {
            lo = c.topics_off;
            hi = uint64_t(c.topics_off) + 33ull * c.n_topics;
        }
        if (c.data_len) {
            lo = lo < c.data_off ? lo : (unsigned long long)c.data_off;
            const unsigned long long e = uint64_t(c.data_off) + c.data_len;
            hi = hi > e ? hi : e;
        }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const unsigned long long l2 = __shfl_xor(lo, d, 64), h2 = __shfl_xor(hi, d, 64);
        lo = lo < l2 ? lo : l2;
        hi = hi > h2 ? hi : h2;
    }
    if ((threadIdx.x & 63u) == 0) {
        if (lo != ~0ull) atomicMin(win, lo);
        if (hi) atomicMax(win + 1, hi);
    }
}

int launch_claims_window(ipcfp_ctx* ctx, const void* claims_d, uint32_t n, unsigned long long* win_d) {
    if (n == 0) return IPCFP_OK;
    hipLaunchKernelGGL(k_claims_window, dim3(div_up(n, 256)), dim3(256), 0, ctx->stream, static_cast<const EventClaimPacked*>(claims_d), n, win_d);
    IPCFP_HIP(ctx, hipGetLastError());
    return IPCFP_OK;
}

int launch_rebase_claims(ipcfp_ctx* ctx, void* claims_d, uint32_t n, uint64_t base, uint64_t blob_len, uint64_t full_len, uint32_t* miss_d,
                         uint32_t* order_d, uint64_t key_lo, uint64_t key_hi) {
    if (n == 0) return IPCFP_OK;
    hipLaunchKernelGGL(k_rebase_claims, dim3(div_up(n, 256)), dim3(256), 0, ctx->stream, static_cast<EventClaimPacked*>(claims_d), n, base,
                       blob_len, full_len, miss_d, order_d, key_lo, key_hi);
    IPCFP_HIP(ctx, hipGetLastError());
    return IPCFP_OK;
}

// scratch_u32: 4 × n words (sizes and offsets); scan_scratch: div_up(n, 1024) + 2 u64 (the last one receives the packed blob's length)
int launch_expand_claims(ipcfp_ctx* ctx, const void* compact_d, uint32_t n, const ipcfp_event_claim_group_t* groups_d, uint32_t n_groups,
                         const uint8_t* cblob_d, uint64_t cblob_len, void* claims_out_d, uint8_t* blob_out_d, uint64_t cap_blob,
                         uint32_t* scratch_u32, uint64_t* scan_scratch) {
    if (n == 0) return IPCFP_OK;
    const ClaimCompact* cc = static_cast<const ClaimCompact*>(compact_d);
    uint32_t *csize = scratch_u32, *psize = scratch_u32 + n, *coff = scratch_u32 + 2 * size_t(n), *poff = scratch_u32 + 3 * size_t(n);
    uint64_t* total_c = scan_scratch + div_up(n, 1024);
    uint64_t* total_p = total_c + 1;
    hipLaunchKernelGGL(k_compact_sizes, dim3(div_up(n, 256)), dim3(256), 0, ctx->stream, cc, n, n_groups, csize, psize);
    int rc = launch_scan_u32(ctx, csize, n, coff, total_c, scan_scratch);
    if (rc) return rc;
    rc = launch_scan_u32(ctx, psize, n, poff, total_p, scan_scratch);
    if (rc) return rc;
    hipLaunchKernelGGL(k_compact_expand, dim3(div_up(n, 256)), dim3(256), 0, ctx->stream, cc, n, ClaimGroups{groups_d, n_groups}, cblob_d,
                       cblob_len, coff, poff, total_c, total_p, static_cast<EventClaimPacked*>(claims_out_d), blob_out_d, cap_blob);
    IPCFP_HIP(ctx, hipGetLastError());
    return IPCFP_OK;
}

}  // namespace ipcfp
