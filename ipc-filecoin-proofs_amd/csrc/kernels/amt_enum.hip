// csrc/kernels/amt_enum.hip — level-synchronous `Amt::for_each` (see amt_enum.h).
#include "amt_enum.h"

#include <hip/hip_runtime.h>

#include <cstring>
#include <vector>

#include "../common.h"
#include "launch.h"

namespace ipcfp {

__device__ __forceinline__ void enum_error(unsigned long long* err, uint32_t seq, uint64_t base, uint32_t code) {
    atomicMin(err, (unsigned long long)pack_enum_error(seq, base, code));
}

// roots → frontier (one entry per root, in root order)
__global__ __launch_bounds__(64) void k_enum_roots(WitnessView w, const AmtRootSpec* __restrict__ roots, uint32_t n,
                                                   int vkind, EnumNode* __restrict__ frontier,
                                                   uint32_t* __restrict__ max_height,
                                                   unsigned long long* __restrict__ err,
                                                   uint64_t* __restrict__ root_info /* n × {height|bw<<32, count} */) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const AmtRootSpec spec = roots[t];
    EnumNode e{kNoBlock, 0, 0, spec.seq, 0, 0, 0};
    root_info[2 * t] = ~0ULL;  // dead
    root_info[2 * t + 1] = 0;
    if (!spec.skip) {
        AmtRootInfo info;
        const uint32_t st = amt_load(w, spec.root, int(spec.version), vkind, info);
        if (st != IPCFP_ST_TRUE) {
            enum_error(err, spec.seq, 0, st);
        } else {
            e.block = info.block;
            e.node_off = info.node_off;
            e.height = uint16_t(info.height);
            e.bit_width = uint8_t(info.bit_width);
            atomicMax(max_height, uint32_t(info.height));
            root_info[2 * t] = uint64_t(uint32_t(info.height)) | (uint64_t(info.bit_width) << 32);
            root_info[2 * t + 1] = info.count;
        }
    }
    frontier[t] = e;
}

// decode the node of entry e (validating it completely); false ⇒ decode error
__device__ __forceinline__ bool enum_read_node(const WitnessView& w, const EnumNode& e, int vkind, AmtNode& nd) {
    Rd r = open_block(w, e.block);
    r.pos = e.node_off;
    amt_read_node(r, e.bit_width, vkind, ~0u, nd);
    if (e.node_off == 0) r.finish();
    return r.ok();
}

// interior level L ≥ 1: how many entries does each frontier entry contribute to the next level?
// `bound` (nullable): device-side number of valid frontier entries when the host launched with a
// PREDICTED size (speculative path); entries past it count as empty.
__global__ __launch_bounds__(256) void k_enum_count(WitnessView w, const EnumNode* __restrict__ frontier, uint32_t n,
                                                    uint32_t level, int vkind, uint32_t* __restrict__ counts,
                                                    unsigned long long* __restrict__ err,
                                                    const uint64_t* __restrict__ bound) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    if (bound && t >= *bound) {
        counts[t] = 0;
        return;
    }
    const EnumNode e = frontier[t];
    uint32_t c = 0;
    if (e.block != kNoBlock) {
        if (e.leaf_ready || e.height < level) {
            c = 1;  // rides along
        } else {
            AmtNode nd;
            if (!enum_read_node(w, e, vkind, nd)) enum_error(err, e.seq, e.base, IPCFP_ST_ERR_DECODE);
            else c = nd.nlinks ? nd.nlinks : 1;  // a Leaf above height 0 is carried down as-is
        }
    }
    counts[t] = c;
}

// One lane per OUTPUT entry (child): lane j finds its parent with a binary search over the exclusive
// offsets, steps over the links before its own (the node was validated by k_enum_count) and resolves
// one link.  A node's 8-32 children are thus resolved by as many lanes side by side instead of one lane
// chasing 8-32 hash probes in sequence.
__global__ __launch_bounds__(256) void k_enum_expand(WitnessView w, const EnumNode* __restrict__ frontier, uint32_t n,
                                                     uint32_t level, const uint32_t* __restrict__ offsets,
                                                     uint32_t total, EnumNode* __restrict__ next,
                                                     unsigned long long* __restrict__ err,
                                                     const uint64_t* __restrict__ bound) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= total) return;
    if (bound && j >= *bound) return;
    // parent = last t with offsets[t] <= j  (entries that contribute nothing share their successor's offset)
    uint32_t lo = 0, hi = n;
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (offsets[mid] <= j) lo = mid;
        else hi = mid;
    }
    EnumNode e = frontier[lo];
    const uint32_t k = j - offsets[lo];
    if (e.leaf_ready || e.height < level) {  // rides along (k == 0)
        next[j] = e;
        return;
    }
    Rd r = open_block(w, e.block);
    r.pos = e.node_off;
    r.expect_array(3);
    uint32_t bo, bl;
    r.read_bytes(bo, bl);
    const uint64_t nl = r.read_array();
    if (nl == 0) {  // a Leaf above height 0: carried down as-is
        e.leaf_ready = 1;
        next[j] = e;
        return;
    }
    // the k-th set bit of the bitmap names the slot; step over the k links before ours
    uint32_t sub = 0, seen = 0;
    for (;; ++sub) {
        if ((r.at(bo + (sub >> 3)) >> (sub & 7)) & 1u) {
            if (seen == k) break;
            ++seen;
        }
    }
    for (uint32_t q = 0; q < k; ++q) {
        uint32_t m;
        uint64_t a;
        r.head(m, a);  // tag 42
        r.head(m, a);  // byte-string header
        r.pos += uint32_t(a);
    }
    CidKey key;
    r.read_link_key(key);
    const uint64_t span = amt_span(e.bit_width, e.height);
    EnumNode c{kNoBlock, 0, e.base + uint64_t(sub) * span, e.seq, uint16_t(e.height - 1), e.bit_width, 0};
    const uint32_t b = witness_find(w, key);
    if (b == kNoBlock) enum_error(err, e.seq, c.base, IPCFP_ST_ERR_MISSING_BLOCK);
    else c.block = b;
    next[j] = c;
}

// leaf level: number of values per entry
__global__ __launch_bounds__(256) void k_enum_count_leaf(WitnessView w, const EnumNode* __restrict__ frontier,
                                                         uint32_t n, int vkind, uint32_t* __restrict__ counts,
                                                         unsigned long long* __restrict__ err,
                                                         const uint64_t* __restrict__ bound) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    if (bound && t >= *bound) {
        counts[t] = 0;
        return;
    }
    const EnumNode e = frontier[t];
    uint32_t c = 0;
    if (e.block != kNoBlock) {
        AmtNode nd;
        if (!enum_read_node(w, e, vkind, nd)) enum_error(err, e.seq, e.base, IPCFP_ST_ERR_DECODE);
        else if (nd.nlinks) enum_error(err, e.seq, e.base, IPCFP_ST_ERR_DECODE);  // link node at height 0
        else c = nd.nvalues;
    }
    counts[t] = c;
}

__global__ __launch_bounds__(256) void k_enum_emit(WitnessView w, const EnumNode* __restrict__ frontier, uint32_t n,
                                                   int vkind, const uint32_t* __restrict__ counts,
                                                   const uint32_t* __restrict__ offsets,
                                                   LeafRef* __restrict__ leaves, uint64_t cap) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    if (counts[t] == 0) return;
    if (uint64_t(offsets[t]) + counts[t] > cap) return;  // speculative path: never write past the prediction
    const EnumNode e = frontier[t];
    const uint32_t o = offsets[t];
    Rd r = open_block(w, e.block);
    r.pos = e.node_off;
    r.expect_array(3);
    uint32_t bo, bl;
    r.read_bytes(bo, bl);
    (void)r.read_array();  // no links in a leaf
    const uint64_t nv = r.read_array();
    uint32_t sub = 0;
    for (uint64_t j = 0; j < nv; ++j) {
        while (!((r.at(bo + (sub >> 3)) >> (sub & 7)) & 1u)) ++sub;
        const uint32_t start = r.pos;
        check_value(r, vkind);
        leaves[o + uint32_t(j)] = LeafRef{e.block, start, r.pos - start, e.seq, e.base + sub};
        ++sub;
    }
}

__global__ void k_enum_check(const uint64_t* __restrict__ actual, uint64_t expected, uint32_t* __restrict__ mismatch) {
    if (threadIdx.x == 0 && blockIdx.x == 0 && *actual != expected) atomicOr(mismatch, 1u);
}

int launch_scan_u32(ipcfp_ctx* ctx, const uint32_t* in_d, uint32_t n, uint32_t* out_d, uint64_t* total_d,
                    uint64_t* scratch_d);

// Entries of the frontier that ENTERS `level`, predicted from the roots' (height, bit width, count)
// under the assumption that every AMT is dense (indices 0..count-1): what fvm_ipld_amt writes for
// message lists, receipts and events.  A sparse or lying root only makes the prediction wrong, which the
// device detects (k_enum_check) and the caller answers by walking level by level.
static uint64_t predicted_frontier(const std::vector<uint64_t>& info, uint32_t level, uint32_t max_height) {
    if (level == max_height) return info.size() / 2;
    uint64_t total = 0;
    for (size_t i = 0; i + 1 < info.size(); i += 2) {
        if (info[i] == ~0ULL) continue;  // dead root
        const uint32_t h = uint32_t(info[i]), bw = uint32_t(info[i] >> 32);
        const uint64_t c = info[i + 1];
        if (h <= level || c == 0) {
            total += 1;  // rides along / is the root itself / an empty root carried down
        } else {
            const uint64_t shift = uint64_t(bw) * (level + 1);
            total += shift >= 64 ? 1 : ((c + (1ULL << shift) - 1) >> shift);
        }
    }
    return total;
}

int amt_enumerate(ipcfp_ctx* ctx, const WitnessView& view, const AmtRootSpec* roots_d, uint32_t n_roots, int vkind,
                  unsigned long long* err_d, AmtEnumResult& out) {
    out.n_leaves = 0;
    out.error = kNoEnumError;
    out.leaves.release();
    if (n_roots == 0) return IPCFP_OK;
    ProfileScope prof(ctx, IPCFP_K_EXEC_ORDER);
    DevBuf<EnumNode> cur, nxt;
    DevBuf<uint32_t> counts, offsets, small;
    DevBuf<uint64_t> scratch, total_d;
    IPCFP_HIP(ctx, cur.alloc(n_roots));
    IPCFP_HIP(ctx, small.alloc(4));
    IPCFP_HIP(ctx, total_d.alloc(2));
    IPCFP_HIP(ctx, hipMemsetAsync(small.p, 0, 16, ctx->stream));
    DevBuf<uint64_t> root_info_d;
    IPCFP_HIP(ctx, root_info_d.alloc(2 * size_t(n_roots)));
    hipLaunchKernelGGL(k_enum_roots, dim3(div_up(n_roots, 64)), dim3(64), 0, ctx->stream, view, roots_d, n_roots, vkind,
                       cur.p, small.p, err_d, root_info_d.p);
    uint32_t max_height = 0;
    std::vector<uint64_t> root_info(2 * size_t(n_roots));
    IPCFP_HIP(ctx, hipMemcpyAsync(&max_height, small.p, 4, hipMemcpyDeviceToHost, ctx->stream));
    IPCFP_HIP(ctx, hipMemcpyAsync(root_info.data(), root_info_d.p, root_info.size() * 8, hipMemcpyDeviceToHost, ctx->stream));
    IPCFP_HIP(ctx, hipStreamSynchronize(ctx->stream));

    // ---- speculative pass: every level launched back to back with PREDICTED sizes, one sync at the end ----
    {
        uint64_t pred_leaves = 0;
        bool sane = true;
        for (size_t i = 0; i + 1 < root_info.size(); i += 2)
            if (root_info[i] != ~0ULL) pred_leaves += root_info[i + 1];
        std::vector<uint64_t> pred(max_height + 1);
        for (uint32_t level = 0; level <= max_height; ++level) {
            pred[level] = predicted_frontier(root_info, level, max_height);
            sane = sane && pred[level] < 0x7fffffffULL;
        }
        sane = sane && pred_leaves < 0x7fffffffULL;
        if (sane) {
            DevBuf<EnumNode> a, b;
            DevBuf<uint32_t> cnt, offs, mismatch;
            DevBuf<uint64_t> scr, totals;
            uint64_t biggest = n_roots;
            for (auto v : pred) biggest = v > biggest ? v : biggest;
            IPCFP_HIP(ctx, a.alloc(biggest));
            IPCFP_HIP(ctx, b.alloc(biggest));
            IPCFP_HIP(ctx, cnt.alloc(biggest));
            IPCFP_HIP(ctx, offs.alloc(biggest));
            IPCFP_HIP(ctx, scr.alloc(size_t(div_up(biggest, 1024)) + 2));
            IPCFP_HIP(ctx, totals.alloc(max_height + 3));
            IPCFP_HIP(ctx, mismatch.alloc(1));
            IPCFP_HIP(ctx, hipMemsetAsync(mismatch.p, 0, 4, ctx->stream));
            IPCFP_HIP(ctx, hipMemcpyAsync(a.p, cur.p, size_t(n_roots) * sizeof(EnumNode), hipMemcpyDeviceToDevice, ctx->stream));
            IPCFP_HIP(ctx, out.leaves.alloc(pred_leaves));
            // err_d may already hold earlier-stage errors; the speculative pass works on a copy
            DevBuf<unsigned long long> err_spec;
            IPCFP_HIP(ctx, err_spec.alloc(1));
            IPCFP_HIP(ctx, hipMemcpyAsync(err_spec.p, err_d, 8, hipMemcpyDeviceToDevice, ctx->stream));
            const uint64_t* bound = nullptr;  // the top frontier (the roots) is exact
            uint32_t slot = 0;
            int rc = IPCFP_OK;
            for (uint32_t level = max_height;; --level, ++slot) {
                const uint32_t np = uint32_t(pred[level]);
                uint64_t* tot = totals.p + slot;
                if (np == 0) {
                    IPCFP_HIP(ctx, hipMemsetAsync(tot, 0, 8, ctx->stream));
                } else if (level >= 1) {
                    hipLaunchKernelGGL(k_enum_count, dim3(div_up(np, 256)), dim3(256), 0, ctx->stream, view, a.p, np, level,
                                       vkind, cnt.p, err_spec.p, bound);
                    rc = launch_scan_u32(ctx, cnt.p, np, offs.p, tot, scr.p);
                } else {
                    hipLaunchKernelGGL(k_enum_count_leaf, dim3(div_up(np, 256)), dim3(256), 0, ctx->stream, view, a.p, np,
                                       vkind, cnt.p, err_spec.p, bound);
                    rc = launch_scan_u32(ctx, cnt.p, np, offs.p, tot, scr.p);
                }
                if (rc) return rc;
                const uint64_t expect = level >= 1 ? pred[level - 1] : pred_leaves;
                hipLaunchKernelGGL(k_enum_check, dim3(1), dim3(64), 0, ctx->stream, tot, expect, mismatch.p);
                if (level >= 1) {
                    if (expect && np)
                        hipLaunchKernelGGL(k_enum_expand, dim3(div_up(expect, 256)), dim3(256), 0, ctx->stream, view, a.p, np,
                                           level, offs.p, uint32_t(expect), b.p, err_spec.p, tot);
                    a.swap(b);
                    bound = tot;
                } else {
                    if (pred_leaves && np)
                        hipLaunchKernelGGL(k_enum_emit, dim3(div_up(np, 256)), dim3(256), 0, ctx->stream, view, a.p, np, vkind,
                                           cnt.p, offs.p, out.leaves.p, pred_leaves);
                    break;
                }
            }
            uint32_t bad = 0;
            unsigned long long e = kNoEnumError;
            IPCFP_HIP(ctx, hipMemcpyAsync(&bad, mismatch.p, 4, hipMemcpyDeviceToHost, ctx->stream));
            IPCFP_HIP(ctx, hipMemcpyAsync(&e, err_spec.p, 8, hipMemcpyDeviceToHost, ctx->stream));
            IPCFP_HIP(ctx, hipStreamSynchronize(ctx->stream));
            IPCFP_HIP(ctx, hipGetLastError());
            if (!bad) {
                IPCFP_HIP(ctx, hipMemcpyAsync(err_d, err_spec.p, 8, hipMemcpyDeviceToDevice, ctx->stream));
                out.n_leaves = pred_leaves;
                out.error = e;
                return IPCFP_OK;
            }
            out.leaves.release();  // prediction failed (sparse AMT, decode error, lying count): walk level by level
        }
    }
    uint32_t n = n_roots;
    for (uint32_t level = max_height;; --level) {
        IPCFP_HIP(ctx, counts.alloc(n));
        IPCFP_HIP(ctx, offsets.alloc(n));
        IPCFP_HIP(ctx, scratch.alloc(size_t(div_up(n, 1024)) + 2));
        if (level >= 1)
            hipLaunchKernelGGL(k_enum_count, dim3(div_up(n, 256)), dim3(256), 0, ctx->stream, view, cur.p, n, level,
                               vkind, counts.p, err_d, static_cast<const uint64_t*>(nullptr));
        else
            hipLaunchKernelGGL(k_enum_count_leaf, dim3(div_up(n, 256)), dim3(256), 0, ctx->stream, view, cur.p, n,
                               vkind, counts.p, err_d, static_cast<const uint64_t*>(nullptr));
        int rc = launch_scan_u32(ctx, counts.p, n, offsets.p, total_d.p, scratch.p);
        if (rc) return rc;
        uint64_t total = 0;
        IPCFP_HIP(ctx, hipMemcpyAsync(&total, total_d.p, 8, hipMemcpyDeviceToHost, ctx->stream));
        IPCFP_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (total >= 0x7fffffffULL)
            return set_error(ctx, IPCFP_E_UNSUPPORTED, "AMT enumeration expands to %llu entries",
                             (unsigned long long)total);
        if (level >= 1) {
            IPCFP_HIP(ctx, nxt.alloc(total));
            if (total)
                hipLaunchKernelGGL(k_enum_expand, dim3(div_up(total, 256)), dim3(256), 0, ctx->stream, view, cur.p, n, level,
                                   offsets.p, uint32_t(total), nxt.p, err_d, static_cast<const uint64_t*>(nullptr));
            cur.swap(nxt);  // the old frontier returns to the pool; reuse is stream-ordered
            n = uint32_t(total);
            if (n == 0) break;
        } else {
            IPCFP_HIP(ctx, out.leaves.alloc(total));
            out.n_leaves = total;
            if (total)
                hipLaunchKernelGGL(k_enum_emit, dim3(div_up(n, 256)), dim3(256), 0, ctx->stream, view, cur.p, n, vkind,
                                   counts.p, offsets.p, out.leaves.p, total);
            break;
        }
    }
    unsigned long long e = kNoEnumError;
    IPCFP_HIP(ctx, hipMemcpyAsync(&e, err_d, 8, hipMemcpyDeviceToHost, ctx->stream));
    IPCFP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    IPCFP_HIP(ctx, hipGetLastError());
    out.error = e;
    return IPCFP_OK;
}

__global__ __launch_bounds__(256) void k_check_dense(const LeafRef* __restrict__ leaves, uint32_t n,
                                                     uint32_t* __restrict__ flag) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const bool bad = i < n && leaves[i].index != uint64_t(i);
    if (__any(bad) && (threadIdx.x & 63) == 0) atomicOr(flag, 1u);
}

int amt_enumerate_cached(ipcfp_ctx* ctx, ipcfp_witness* w, const CidKey& root, int version, int vkind,
                         const EnumCached** out) {
    for (auto& e : w->enum_cache)
        if (e->version == version && e->vkind == vkind && std::memcmp(e->root, root.w, 40) == 0) {
            *out = e.get();
            return IPCFP_OK;
        }
    std::unique_ptr<EnumCached> e(new EnumCached());
    std::memcpy(e->root, root.w, 40);
    e->version = version;
    e->vkind = vkind;
    WitnessView view;
    view.arena = w->arena.p;
    view.off = w->off.p;
    view.len = w->len.p;
    view.cids = w->cids.p;
    view.slots = w->index_slots.p;
    view.mask = w->index_mask;
    view.n = uint32_t(w->n);
    view.touched = nullptr;
    DevBuf<AmtRootSpec> roots;
    DevBuf<unsigned long long> err;
    DevBuf<uint32_t> flag;
    IPCFP_HIP(ctx, roots.alloc(1));
    IPCFP_HIP(ctx, err.alloc(1));
    IPCFP_HIP(ctx, flag.alloc(1));
    AmtRootSpec spec{};
    spec.root = root;
    spec.version = uint32_t(version);
    unsigned long long e0 = kNoEnumError;
    IPCFP_HIP(ctx, hipMemcpyAsync(roots.p, &spec, sizeof spec, hipMemcpyHostToDevice, ctx->stream));
    IPCFP_HIP(ctx, hipMemcpyAsync(err.p, &e0, 8, hipMemcpyHostToDevice, ctx->stream));
    IPCFP_HIP(ctx, hipMemsetAsync(flag.p, 0, 4, ctx->stream));
    AmtEnumResult en;
    int rc = amt_enumerate(ctx, view, roots.p, 1, vkind, err.p, en);
    if (rc) return rc;
    e->n = en.n_leaves;
    e->error = en.error;
    uint32_t not_dense = 0;
    if (en.n_leaves) {
        const uint32_t n = uint32_t(en.n_leaves);
        hipLaunchKernelGGL(k_check_dense, dim3(div_up(n, 256)), dim3(256), 0, ctx->stream, en.leaves.p, n, flag.p);
        IPCFP_HIP(ctx, hipMemcpyAsync(&not_dense, flag.p, 4, hipMemcpyDeviceToHost, ctx->stream));
        IPCFP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    e->dense = !not_dense;
    // move the leaves into the cache entry (byte-typed buffer)
    e->leaves.p = reinterpret_cast<uint8_t*>(en.leaves.p);
    e->leaves.count = en.leaves.count * sizeof(LeafRef);
    e->leaves.cap = en.leaves.cap;
    e->leaves.owner = en.leaves.owner;
    en.leaves.p = nullptr;
    en.leaves.count = en.leaves.cap = 0;
    en.leaves.owner = nullptr;
    *out = e.get();
    w->enum_cache.push_back(std::move(e));
    return IPCFP_OK;
}

}  // namespace ipcfp
