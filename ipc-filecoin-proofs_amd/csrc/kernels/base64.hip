// csrc/kernels/base64.hip — base64 (RFC 4648 standard alphabet, canonical padding) → witness arena.
//
// Replaces `deserialize_base64` (src/proofs/common/bundle.rs:30-37: `B64.decode(s)` with
// base64 0.21's STANDARD engine) for every `ProofBlock.data` of a bundle at once.  The JSON text is in
// HBM; the host has located each data string (offset, length) and — from the length and the last two
// characters — knows the decoded size, so the arena layout exists before the kernel runs.
//
// One lane per 16 characters → 12 bytes (three aligned dword stores; every block starts on a 128-byte
// line so 12·u is 4-byte aligned).  The lane finds its block by binary search over the per-block unit
// prefix.  The kernel is a pure stream: 4 B read per 3 B written.
//
// Validity (any violation makes the whole bundle an Err, as the serde error does): length % 4 == 0;
// only alphabet characters, except one or two '=' at the very end; the bits dropped by the padding are
// zero (`DecodeError::InvalidLastSymbol`).
#include <hip/hip_runtime.h>

#include "../common.h"
#include "launch.h"

namespace ipcfp {

// character → sextet, 0xff when outside the alphabet; branch-free range compares
__device__ __forceinline__ uint32_t b64_val(uint32_t c) {
    uint32_t v = 0xffu;
    v = (c - 'A' < 26u) ? c - 'A' : v;
    v = (c - 'a' < 26u) ? c - 'a' + 26u : v;
    v = (c - '0' < 10u) ? c - '0' + 52u : v;
    v = c == '+' ? 62u : v;
    v = c == '/' ? 63u : v;
    return v;
}

// 4 characters (little-endian in `w`) → 3 bytes (little-endian in the result), bit 31 set on a bad character
__device__ __forceinline__ uint32_t b64_group(uint32_t w) {
    const uint32_t s0 = b64_val(w & 0xffu), s1 = b64_val((w >> 8) & 0xffu), s2 = b64_val((w >> 16) & 0xffu),
                   s3 = b64_val(w >> 24);
    const uint32_t bad = (s0 | s1 | s2 | s3) & 0x80u;
    const uint32_t b0 = ((s0 << 2) | (s1 >> 4)) & 0xffu;
    const uint32_t b1 = ((s1 << 4) | (s2 >> 2)) & 0xffu;
    const uint32_t b2 = ((s2 << 6) | s3) & 0xffu;
    return b0 | (b1 << 8) | (b2 << 16) | (bad << 24);
}

struct B64Span {
    uint64_t src;   // byte offset of the string body in the JSON text
    uint32_t len;   // characters
    uint32_t unit0; // index of this block's first 16-character unit
};

__global__ __launch_bounds__(256) void k_base64_decode(const uint8_t* __restrict__ text, const B64Span* __restrict__ spans,
                                                       uint32_t n_blocks, uint32_t n_units,
                                                       const uint64_t* __restrict__ dst_off, uint8_t* __restrict__ arena,
                                                       unsigned long long* __restrict__ first_bad) {
    const uint32_t u = blockIdx.x * blockDim.x + threadIdx.x;
    if (u >= n_units) return;
    // last block whose unit0 <= u
    uint32_t lo = 0, hi = n_blocks;
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (spans[mid].unit0 <= u) lo = mid;
        else hi = mid;
    }
    const B64Span sp = spans[lo];
    const uint32_t k = u - sp.unit0;          // unit inside the block
    const uint32_t c0 = k * 16u;              // first character of this unit
    const uint32_t nchar = min(16u, sp.len - c0);
    // 16 characters at an arbitrary byte offset: three aligned 8-byte words, funnel-shifted
    const uint64_t a = sp.src + c0;
    const uint64_t* wp = reinterpret_cast<const uint64_t*>(text + (a & ~7ull));
    const uint32_t sh = uint32_t(a & 7ull) * 8u;
    const uint64_t w0 = wp[0], w1 = wp[1], w2 = sh ? wp[2] : 0;  // the text buffer has 16 bytes of tail slack
    const uint64_t q0 = sh ? (w0 >> sh) | (w1 << (64u - sh)) : w0;
    const uint64_t q1 = sh ? (w1 >> sh) | (w2 << (64u - sh)) : w1;
    uint32_t grp[4] = {uint32_t(q0), uint32_t(q0 >> 32), uint32_t(q1), uint32_t(q1 >> 32)};
    const bool last_unit = c0 + 16u >= sp.len;
    bool bad = (sp.len & 3u) != 0 && last_unit;  // the host never schedules such a block, but stay safe
    uint32_t out[4];
    uint32_t pads = 0;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        uint32_t w = grp[g];
        const bool present = uint32_t(g) * 4u < nchar;
        const bool final_group = last_unit && uint32_t(g) * 4u + 4u >= nchar && present;
        if (final_group) {  // '=' may close the final group: "xx==" or "xxx="
            const bool p3 = (w >> 24) == '=';
            const bool p2 = p3 && ((w >> 16) & 0xffu) == '=';
            pads = p2 ? 2u : p3 ? 1u : 0u;
            if (p2) w = (w & 0x0000ffffu) | (uint32_t('A') << 16) | (uint32_t('A') << 24);
            else if (p3) w = (w & 0x00ffffffu) | (uint32_t('A') << 24);
        }
        const uint32_t d = present ? b64_group(w) : 0u;
        bad |= (d >> 31) != 0;
        if (final_group) {
            // bits that the padding drops must be zero: 1 pad → low 2 bits of s2 → byte 2 == 0 with s3 = 'A';
            // 2 pads → low 4 bits of s1 → byte 1 == 0
            if (pads == 1) bad |= ((d >> 16) & 0xffu) != 0;
            if (pads == 2) bad |= ((d >> 8) & 0xffu) != 0;
        }
        out[g] = d & 0x00ffffffu;
    }
    if (bad) atomicMin(first_bad, (unsigned long long)lo);
    // 4 × 3 bytes → 3 dwords
    const uint32_t d0 = out[0] | (out[1] << 24);
    const uint32_t d1 = (out[1] >> 8) | (out[2] << 16);
    const uint32_t d2 = (out[2] >> 16) | (out[3] << 8);
    uint8_t* dst = arena + dst_off[lo] + uint64_t(k) * 12u;
    const uint32_t nbytes = (nchar >> 2) * 3u - pads;  // bytes this unit produces
    if (nbytes == 12) {
        uint32_t* d4 = reinterpret_cast<uint32_t*>(dst);
        d4[0] = d0;
        d4[1] = d1;
        d4[2] = d2;
    } else {
        const uint32_t dw[3] = {d0, d1, d2};
        for (uint32_t i = 0; i < nbytes; ++i) dst[i] = uint8_t(dw[i >> 2] >> ((i & 3u) * 8u));
    }
}

int launch_base64_decode(ipcfp_ctx* ctx, const uint8_t* text_d, const void* spans_d, uint32_t n_blocks, uint32_t n_units,
                         const uint64_t* dst_off_d, uint8_t* arena_d, unsigned long long* first_bad_d) {
    if (n_units == 0) return IPCFP_OK;
    ProfileScope prof(ctx, IPCFP_K_BASE64);
    hipLaunchKernelGGL(k_base64_decode, dim3(div_up(n_units, 256)), dim3(256), 0, ctx->stream, text_d,
                       static_cast<const B64Span*>(spans_d), n_blocks, n_units, dst_off_d, arena_d, first_bad_d);
    IPCFP_HIP(ctx, hipGetLastError());
    return IPCFP_OK;
}

}  // namespace ipcfp
