// csrc/kernels/shard.hip — one tipset over several GPUs: the planner's marking pass and the
// gathers that cut a shard's witness out of the whole one.
//
// The reference verifies a bundle proof by proof on one thread (src/proofs/verifier.rs:19-28,49-54,
// src/proofs/events/verifier.rs:62-71).  Proofs are independent given a read-only witness, so rank r of G takes
// the receipts [lo, hi) of the tipset — their events AMTs, the receipts-AMT nodes on the paths to them (the upper
// nodes end up on every rank) — plus what every proof needs: the headers, the TxMeta blocks and the message AMTs
// the execution order is rebuilt from.  Which blocks those are is found the way the reference's generator finds a
// witness: by walking with a RecordingBlockStore (src/proofs/common/blockstore.rs:26-30) — here the `touched`
// bitmap of the WitnessView.
#include <hip/hip_runtime.h>

#include "../common.h"
#include "amt_enum.h"
#include "event_log_dev.h"
#include "walk_dev.h"
#include "launch.h"

namespace ipcfp {

// An events AMT taller than the explicit stack below (the root's bit width is the witness's choice: up to 64 / bit_width
// levels): the same walk without a per-level stack — the path is kept as packed digits and a step back UP is a descent
// from the root along them (event_scan.hip amt_for_each_lane_tall).  A node that fails to decode is recorded and not
// descended into.
__device__ __noinline__ void plan_tall_events(const WitnessView& rec, const AmtRootInfo& einfo) {
    const uint32_t bw = einfo.bit_width, width = 1u << bw;
    unsigned __int128 path = 0;
    uint32_t depth = 0, blk = einfo.block, noff = einfo.node_off, next = 0;
    for (;;) {
        const uint64_t height = einfo.height - uint64_t(depth);
        Rd nr = open_block(rec, blk);
        nr.pos = noff;
        nr.expect_array(3);
        uint32_t bo, bl;
        nr.read_bytes(bo, bl);
        const uint64_t nl = nr.read_array();
        bool descend = false;
        if (nr.ok() && nl != 0 && height != 0 && bl == (width + 7) / 8) {
            uint32_t sub = next, ordinal = 0;
            for (uint32_t i = 0; i < sub && i < width; ++i) ordinal += (nr.at(bo + (i >> 3)) >> (i & 7)) & 1u;
            while (sub < width && !((nr.at(bo + (sub >> 3)) >> (sub & 7)) & 1u)) ++sub;
            while (sub < width && ordinal < nl && !descend) {
                Rd lr = nr;
                CidKey key;
                for (uint32_t k = 0; k <= ordinal && lr.ok(); ++k) lr.read_link_key(key);
                if (!lr.ok()) break;
                const uint32_t child = witness_find(rec, key);  // records it
                if (child != kNoBlock) {
                    path &= ~((unsigned __int128)(width - 1u) << (depth * bw));
                    path |= (unsigned __int128)sub << (depth * bw);
                    ++depth;
                    blk = child;
                    noff = 0;
                    next = 0;
                    descend = true;
                } else {  // a link that does not resolve: on to the next set bit of this node
                    ++sub;
                    ++ordinal;
                    while (sub < width && !((nr.at(bo + (sub >> 3)) >> (sub & 7)) & 1u)) ++sub;
                }
            }
        }
        if (descend) continue;
        if (depth == 0) return;
        --depth;
        next = (uint32_t(path >> (depth * bw)) & (width - 1u)) + 1u;
        blk = einfo.block;
        noff = einfo.node_off;
        for (uint32_t k = 0; k < depth; ++k) {
            Rd q = open_block(rec, blk);
            q.pos = noff;
            q.expect_array(3);
            uint32_t qo, ql;
            q.read_bytes(qo, ql);
            (void)q.read_array();
            const uint32_t sub = uint32_t(path >> (k * bw)) & (width - 1u);
            uint32_t ordinal = 0;
            for (uint32_t i = 0; i < sub; ++i) ordinal += (q.at(qo + (i >> 3)) >> (i & 7)) & 1u;
            CidKey key;
            for (uint32_t j = 0; j <= ordinal; ++j) q.read_link_key(key);
            const uint32_t child = witness_find(rec, key);
            if (child == kNoBlock || !q.ok()) return;  // (it resolved on the way down)
            blk = child;
            noff = 0;
        }
    }
}

// the path to receipt `index` in the receipts AMT, and the whole events AMT of that receipt, recorded in rec.touched
__device__ __forceinline__ void plan_one_receipt(const WitnessView& rec, const CidKey& receipts_root, uint64_t index) {
    AmtRootInfo rinfo;
    if (amt_load(rec, receipts_root, 0, VK_RECEIPT, rinfo) != IPCFP_ST_TRUE) return;
    ValueLoc rl;
    if (amt_get(rec, rinfo, VK_RECEIPT, index, rl) != IPCFP_ST_TRUE) return;
    Rd r;
    r.init(rec.arena + rec.off[rl.block] + rl.off, rl.len);
    uint32_t o, l;
    r.expect_array(4);
    (void)r.read_uint();
    r.read_bytes(o, l);
    (void)r.read_uint();
    if (!r.ok() || r.at_null()) return;
    CidKey ev_root;
    r.read_link_key(ev_root);
    if (!r.ok()) return;
    AmtRootInfo einfo;
    if (amt_load(rec, ev_root, 3, VK_STAMPED_EVENT, einfo) != IPCFP_ST_TRUE) return;
    if (einfo.height == 0) return;  // the root block is the whole tree
    // depth-first over the links; a node that fails to decode is recorded and not descended into
    constexpr int kMaxDepth = 8;
    if (einfo.height >= kMaxDepth) return plan_tall_events(rec, einfo);
    uint32_t blk[kMaxDepth], noff[kMaxDepth], next_sub[kMaxDepth];
    int depth = 0;
    blk[0] = einfo.block;
    noff[0] = einfo.node_off;
    next_sub[0] = 0;
    const uint32_t bw = einfo.bit_width, width = 1u << bw;
    while (depth >= 0) {
        const uint64_t height = einfo.height - uint64_t(depth);
        Rd nr = open_block(rec, blk[depth]);
        nr.pos = noff[depth];
        nr.expect_array(3);
        uint32_t bo, bl;
        nr.read_bytes(bo, bl);
        const uint64_t nl = nr.read_array();
        if (!nr.ok() || nl == 0 || height == 0 || bl != (width + 7) / 8) {
            --depth;
            continue;
        }
        uint32_t sub = next_sub[depth], ordinal = 0;
        for (uint32_t i = 0; i < sub && i < width; ++i) ordinal += (nr.at(bo + (i >> 3)) >> (i & 7)) & 1u;
        while (sub < width && !((nr.at(bo + (sub >> 3)) >> (sub & 7)) & 1u)) ++sub;
        if (sub >= width || ordinal >= nl) {
            --depth;
            continue;
        }
        next_sub[depth] = sub + 1;
        CidKey key;
        for (uint32_t k = 0; k <= ordinal && nr.ok(); ++k) nr.read_link_key(key);
        if (!nr.ok()) {
            --depth;
            continue;
        }
        const uint32_t child = witness_find(rec, key);  // records it
        if (child == kNoBlock || depth + 1 >= kMaxDepth) continue;
        ++depth;
        blk[depth] = child;
        noff[depth] = 0;
        next_sub[depth] = 0;
    }
}

// lane i: receipt lo + i, recorded in the one bitmap of the view
__global__ __launch_bounds__(256) void k_plan_receipts(WitnessView rec, CidKey receipts_root, uint64_t lo, uint32_t n) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    plan_one_receipt(rec, receipts_root, lo + t);
}

// Every shard's plan in one pass (ipcfp_shard_plan_tipset_all): lane i takes receipt i and records into the bitmap of
// the shard that owns it — rec.touched is the first of n_shards bitmaps of `words` words each; bounds[s] .. bounds[s+1]
// are shard s's receipts (ipcfp_shard_range).
__global__ __launch_bounds__(256) void k_plan_receipts_all(WitnessView rec, CidKey receipts_root, uint32_t n,
                                                           const uint64_t* __restrict__ bounds, uint32_t n_shards,
                                                           uint32_t words) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    uint32_t lo = 0, hi = n_shards;  // the shard s with bounds[s] <= t < bounds[s + 1]
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (bounds[mid] <= t) lo = mid; else hi = mid;
    }
    WitnessView mine = rec;
    mine.touched = rec.touched + size_t(lo) * words;
    plan_one_receipt(mine, receipts_root, t);
}

// Amt::load of one root → {status, height, count, bit width}
__global__ void k_amt_root_info(WitnessView w, CidKey root, int version, int vkind, uint64_t* __restrict__ out) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    AmtRootInfo info;
    info.height = info.count = 0;
    info.bit_width = 0;
    const uint32_t st = amt_load(w, root, version, vkind, info);
    out[0] = st;
    out[1] = info.height;
    out[2] = info.count;
    out[3] = info.bit_width;
}

// tables of a sub-witness: entry i describes block ids[i] of the source witness
__global__ __launch_bounds__(256) void k_subset_tables(const uint32_t* __restrict__ ids, uint32_t n, uint32_t n_src,
                                                       const uint64_t* __restrict__ src_off, const uint32_t* __restrict__ src_len,
                                                       const uint8_t* __restrict__ src_cids, uint64_t* __restrict__ off,
                                                       uint32_t* __restrict__ len, uint8_t* __restrict__ cids,
                                                       uint32_t* __restrict__ bad) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const uint32_t id = ids[t];
    if (id >= n_src) {
        atomicOr(bad, 1u);
        off[t] = 0;
        len[t] = 0;
        return;
    }
    off[t] = src_off[id];
    len[t] = src_len[id];
    const uint64_t* s = reinterpret_cast<const uint64_t*>(src_cids + 40ull * id);
    uint64_t* d = reinterpret_cast<uint64_t*>(cids + 40ull * t);
#pragma unroll
    for (int k = 0; k < 5; ++k) d[k] = s[k];
}

// Blockstore::has / get: CID → block id (kNoBlock: absent)
__global__ __launch_bounds__(256) void k_find_blocks(WitnessView w, const CidKey* __restrict__ keys, uint32_t n,
                                                     uint32_t* __restrict__ ids) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) ids[t] = witness_find(w, keys[t]);
}

// out[i] = base + off[i]: arena offsets as absolute device addresses (ipcfp_witness_put_keyed's source table)
__global__ __launch_bounds__(256) void k_absolute_offsets(const uint64_t* __restrict__ off, uint32_t n, uint64_t base,
                                                          uint64_t* __restrict__ out) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) out[t] = base + off[t];
}

// ipcfp_witness_read_values: value i (block, off, len) → out[i * stride ..), truncated to stride; one wavefront per
// value, 16 lanes-bytes at a time.  A location that does not lie inside its block (block >= n, off + len past the
// block's end) is never dereferenced: the first such index goes to *first_bad.
__global__ __launch_bounds__(256) void k_gather_values(WitnessView w, const ValueLoc* __restrict__ locs, uint32_t n,
                                                       uint8_t* __restrict__ out, uint64_t stride,
                                                       unsigned long long* __restrict__ first_bad) {
    const uint32_t v = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (v >= n) return;
    const ValueLoc l = locs[v];
    if (l.block == kNoBlock) return;
    if (l.block >= w.n || uint64_t(l.off) + l.len > w.len[l.block]) {
        if ((threadIdx.x & 63u) == 0) atomicMin(first_bad, (unsigned long long)v);
        return;
    }
    const uint8_t* src = w.arena + w.off[l.block] + l.off;
    uint8_t* dst = out + uint64_t(v) * stride;
    const uint64_t take = l.len < stride ? l.len : stride;
    for (uint64_t i = threadIdx.x & 63u; i < take; i += 64) dst[i] = src[i];
}

int launch_gather_values(ipcfp_ctx* ctx, const WitnessView& w, const void* locs_d, uint32_t n, uint8_t* out_d, uint64_t stride,
                         unsigned long long* first_bad_d) {
    if (n == 0) return IPCFP_OK;
    hipLaunchKernelGGL(k_gather_values, dim3(div_up(n, 4)), dim3(256), 0, ctx->stream, w, static_cast<const ValueLoc*>(locs_d),
                       n, out_d, stride, first_bad_d);
    IPCFP_HIP(ctx, hipGetLastError());
    return IPCFP_OK;
}

int launch_find_blocks(ipcfp_ctx* ctx, const WitnessView& w, const CidKey* keys_d, uint32_t n, uint32_t* ids_d) {
    if (n == 0) return IPCFP_OK;
    hipLaunchKernelGGL(k_find_blocks, dim3(div_up(n, 256)), dim3(256), 0, ctx->stream, w, keys_d, n, ids_d);
    IPCFP_HIP(ctx, hipGetLastError());
    return IPCFP_OK;
}

int launch_absolute_offsets(ipcfp_ctx* ctx, const uint64_t* off_d, uint32_t n, uint64_t base, uint64_t* out_d) {
    if (n == 0) return IPCFP_OK;
    hipLaunchKernelGGL(k_absolute_offsets, dim3(div_up(n, 256)), dim3(256), 0, ctx->stream, off_d, n, base, out_d);
    IPCFP_HIP(ctx, hipGetLastError());
    return IPCFP_OK;
}

int launch_plan_receipts(ipcfp_ctx* ctx, const WitnessView& rec, const CidKey& receipts_root, uint64_t lo, uint32_t n) {
    if (n == 0) return IPCFP_OK;
    hipLaunchKernelGGL(k_plan_receipts, dim3(div_up(n, 256)), dim3(256), 0, ctx->stream, rec, receipts_root, lo, n);
    IPCFP_HIP(ctx, hipGetLastError());
    return IPCFP_OK;
}

int launch_plan_receipts_all(ipcfp_ctx* ctx, const WitnessView& rec, const CidKey& receipts_root, uint32_t n,
                             const uint64_t* bounds_d, uint32_t n_shards, uint32_t words) {
    if (n == 0) return IPCFP_OK;
    hipLaunchKernelGGL(k_plan_receipts_all, dim3(div_up(n, 256)), dim3(256), 0, ctx->stream, rec, receipts_root, n, bounds_d,
                       n_shards, words);
    IPCFP_HIP(ctx, hipGetLastError());
    return IPCFP_OK;
}

int launch_amt_root_info(ipcfp_ctx* ctx, const WitnessView& w, const CidKey& root, int version, int vkind, uint64_t* out_d) {
    hipLaunchKernelGGL(k_amt_root_info, dim3(1), dim3(64), 0, ctx->stream, w, root, version, vkind, out_d);
    IPCFP_HIP(ctx, hipGetLastError());
    return IPCFP_OK;
}

int launch_subset_tables(ipcfp_ctx* ctx, const uint32_t* ids_d, uint32_t n, uint32_t n_src, const uint64_t* src_off,
                         const uint32_t* src_len, const uint8_t* src_cids, uint64_t* off_d, uint32_t* len_d,
                         uint8_t* cids_d, uint32_t* bad_d) {
    if (n == 0) return IPCFP_OK;
    hipLaunchKernelGGL(k_subset_tables, dim3(div_up(n, 256)), dim3(256), 0, ctx->stream, ids_d, n, n_src, src_off, src_len,
                       src_cids, off_d, len_d, cids_d, bad_d);
    IPCFP_HIP(ctx, hipGetLastError());
    return IPCFP_OK;
}

}  // namespace ipcfp
