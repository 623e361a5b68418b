// csrc/kernels/hash_short.hip — K2 (Keccak-256) and K3 (SHA-256) batch kernels over an
// (offset, length) table; one message per lane.
//
// K2 replaces hash_event_signature / keccak256 (src/proofs/common/evm.rs:62-69,81-88)
// and compute_mapping_slot (src/proofs/storage/utils.rs:5-12); K3 is the HAMT key
// hash of fvm_ipld_hamt (src/proofs/common/decode.rs:29-39).
#include <hip/hip_runtime.h>

#include "../common.h"
#include "keccak_dev.h"
#include "launch.h"
#include "sha256_dev.h"

namespace ipcfp {

__global__ __launch_bounds__(64) void k_keccak256(const uint8_t* __restrict__ bytes, const uint64_t* __restrict__ off,
                                                  const uint32_t* __restrict__ len, uint32_t n,
                                                  uint64_t* __restrict__ out32) {
    const uint32_t i = blockIdx.x * 64u + threadIdx.x;
    if (i >= n) return;
    uint64_t d[4];
    keccak::hash_bytes(bytes + off[i], len[i], d);
    uint64_t* o = out32 + 4ull * i;
    o[0] = d[0];
    o[1] = d[1];
    o[2] = d[2];
    o[3] = d[3];
}

__global__ __launch_bounds__(64) void k_sha256(const uint8_t* __restrict__ bytes, const uint64_t* __restrict__ off,
                                               const uint32_t* __restrict__ len, uint32_t n,
                                               uint32_t* __restrict__ out32) {
    const uint32_t i = blockIdx.x * 64u + threadIdx.x;
    if (i >= n) return;
    uint32_t h[8];
    sha256::hash_bytes(bytes + off[i], len[i], h);
    uint32_t* o = out32 + 8ull * i;
#pragma unroll
    for (int k = 0; k < 8; ++k) o[k] = __builtin_bswap32(h[k]);  // digest bytes are big-endian words
}

int launch_keccak256(ipcfp_ctx* ctx, const uint8_t* bytes, const uint64_t* off, const uint32_t* len, uint32_t n,
                     uint8_t* out32) {
    if (n == 0) return IPCFP_OK;
    {
        ProfileScope prof(ctx, IPCFP_K_KECCAK256);
        hipLaunchKernelGGL(k_keccak256, dim3(div_up(n, 64)), dim3(64), 0, ctx->stream, bytes, off, len, n,
                           reinterpret_cast<uint64_t*>(out32));
    }
    IPCFP_HIP(ctx, hipGetLastError());
    return IPCFP_OK;
}

int launch_sha256(ipcfp_ctx* ctx, const uint8_t* bytes, const uint64_t* off, const uint32_t* len, uint32_t n,
                  uint8_t* out32) {
    if (n == 0) return IPCFP_OK;
    {
        ProfileScope prof(ctx, IPCFP_K_SHA256);
        hipLaunchKernelGGL(k_sha256, dim3(div_up(n, 64)), dim3(64), 0, ctx->stream, bytes, off, len, n,
                           reinterpret_cast<uint32_t*>(out32));
    }
    IPCFP_HIP(ctx, hipGetLastError());
    return IPCFP_OK;
}

}  // namespace ipcfp
