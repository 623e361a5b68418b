// csrc/kernels/shard_pull.hip — a rank PULLS its receipt-range shard out of a bundle that lies in HOST memory
// (ipcfp_witness_create_shard_pull, host/shard_pull.cpp): nobody holds the whole witness in HBM, nobody cuts block
// lists on the host.
//
// The loops being cut are the reference's sequential ones (src/proofs/verifier.rs:19-28,49-54,
// src/proofs/events/verifier.rs:62-71); what a rank needs for the receipts [lo, hi) is what the reference's generator
// records for them (src/proofs/events/generator.rs:122-177,195-301, src/proofs/events/utils.rs:48-94): child header →
// receipts AMT (the paths to [lo, hi)) → those receipts' events AMTs; parent headers → TxMeta → both message AMTs
// (whole: the execution order is global).  That set is a reachability closure, so it is found level by level:
//
//   round k   CLAIM   every block of the frontier that is not in HBM yet gets a place in the staging arena
//             COPY    the device reads those blocks straight out of the (pinned / registered) host buffer — half a
//                     wavefront per block, 48 GB/s for ≈ 350-byte blocks on this box against 56 GB/s for one bulk DMA
//                     (tools/ubench/zero_copy.hip) — so the only bytes that cross PCIe are the shard's own
//             EXPAND  one lane per frontier item parses its block by KIND and appends the blocks it links to (resolved by
//                     the index over the bundle's CID table, which is all of the bundle that was uploaded) to round k + 1
//
// A block that does not parse as its kind is simply not expanded: it is in the shard, and verify / scan give the
// reference's verdict about it there.  A link that resolves to no block of the bundle is nothing to pull (the verdict will
// be "missing block", as without the cut).  Expansion is per tree POSITION (a block reached on two paths is copied once
// and expanded twice, as a tree walk does); the frontier's capacity bounds what a hostile DAG can ask for.
#include <hip/hip_runtime.h>

#include <algorithm>

#include "../common.h"
#include "header_dev.h"
#include "launch.h"
#include "shard_pull.h"
#include "walk_dev.h"

namespace ipcfp {

namespace {

__device__ __forceinline__ void pull_emit(const PullFrontier& next, PullCtl* ctl, uint32_t id, uint32_t kind, uint32_t height,
                                          uint64_t base) {
    if (id == kNoBlock) return;
    if (kind != PK_RCPT_NODE && kind != PK_RCPT_ROOT) {
        const uint32_t sig = kind | (height << 8) | 0x80000000u;
        if (atomicCAS(&next.role[id], 0u, sig) == sig) return;  // emitted in this role before: same children, nothing new
    }
    const uint32_t at = atomicAdd(&ctl->n_next, 1u);
    if (at >= next.cap) {  // (n_next runs on past cap; k_pull_round_end never hands such a count to a reader)
        atomicOr(&ctl->overflow, 1u);
        return;
    }
    next.items[at] = PullItem{id, kind | (height << 8), base};
}

// `[bmap, [links…], [values…]]` at r, a node at `height` of a tree that is needed whole: every link → an item of
// `node_kind` one level down, or PK_LEAF when that level is the leaves (values only: copied, never parsed — a walk that
// finds links at height 0 fails there and follows none of them)
__device__ __forceinline__ void expand_node_all(const WitnessView& w, Rd& r, const PullFrontier& next, PullCtl* ctl,
                                                uint32_t node_kind, uint32_t height) {
    if (height == 0) return;
    r.expect_array(3);
    uint32_t bo, bl;
    r.read_bytes(bo, bl);
    const uint64_t nl = r.read_array();
    const uint32_t child_kind = height == 1u ? uint32_t(PK_LEAF) : node_kind;
    for (uint64_t k = 0; k < nl && r.ok(); ++k) {
        CidKey key;
        if (!r.read_link_key(key)) break;
        pull_emit(next, ctl, witness_find(w, key), child_kind, height - 1u, 0);
    }
}

// a node of the receipts AMT (v0: bit width 3) at `height` whose first index is `base`: the links whose subtree meets
// [lo, hi), or — a leaf — the events roots of the receipts inside it
__device__ __forceinline__ void expand_receipts_node(const WitnessView& w, Rd& r, const PullFrontier& next, PullCtl* ctl,
                                                     uint32_t height, uint64_t base, uint64_t lo, uint64_t hi) {
    r.expect_array(3);
    uint32_t bo, bl;
    r.read_bytes(bo, bl);
    if (!r.ok() || bl != 1u) return;
    const uint32_t bmap = r.at(bo);
    const uint64_t nl = r.read_array();
    if (!r.ok()) return;
    if (height > 0) {
        if (height > 20u) return;                       // (8^21 indices: no such tree; Amt::load refuses it as well)
        const uint64_t span = 1ull << (3u * height);    // indices under one link
        uint32_t k = 0;
        for (uint32_t j = 0; j < 8u && k < nl && r.ok(); ++j) {
            if (!((bmap >> j) & 1u)) continue;
            CidKey key;
            if (!r.read_link_key(key)) break;
            ++k;
            const uint64_t first = base + uint64_t(j) * span;
            if (first < hi && first + span > lo) pull_emit(next, ctl, witness_find(w, key), PK_RCPT_NODE, height - 1u, first);
        }
        return;
    }
    if (nl != 0) return;
    const uint64_t nv = r.read_array();
    uint32_t k = 0;
    for (uint32_t j = 0; j < 8u && k < nv && r.ok(); ++j) {
        if (!((bmap >> j) & 1u)) continue;
        ++k;
        // Receipt [exit_code, return_data, gas_used, events_root | null]
        r.expect_array(4);
        (void)r.read_uint();
        uint32_t o, l;
        r.read_bytes(o, l);
        (void)r.read_uint();
        if (!r.ok()) break;
        if (r.at_null()) {
            r.read_null();
            continue;
        }
        CidKey key;
        if (!r.read_link_key(key)) break;
        const uint64_t index = base + j;
        if (index >= lo && index < hi) pull_emit(next, ctl, witness_find(w, key), PK_EV_ROOT, 0, 0);
    }
}

}  // namespace

// the tipset key → the first frontier
__global__ void k_pull_seed(WitnessView w, PullSeeds seeds, PullFrontier first, PullCtl* __restrict__ ctl) {
    if (blockIdx.x != 0) return;
    for (uint32_t t = threadIdx.x; t <= seeds.n_parents; t += blockDim.x) {
        const bool child = t == seeds.n_parents;
        const CidKey key = child ? seeds.child : (seeds.parents_wide ? seeds.parents_wide[t] : seeds.parents[t < IPCFP_MAX_PARENTS ? t : 0u]);
        pull_emit(first, ctl, witness_find(w, key), child ? PK_HDR_CHILD : PK_HDR_PARENT, 0, 0);
    }
}

// CLAIM: lane = frontier item.  A block nobody has claimed yet gets the next lines of the staging arena and an entry of
// this round's copy list; the three counters are advanced once per wavefront (every lane of the chip on one address of
// the L2 was 787 µs in another kernel of this library: profiles/r04_experiments.md).
__global__ __launch_bounds__(256) void k_pull_claim(PullFrontier cur, PullTables t, PullCtl* __restrict__ ctl) {
    if (ctl->overflow) return;  // a buffer was outgrown: the rounds queued ahead do nothing, the host reports it
    const uint32_t n_items = min(ctl->n_cur, cur.cap);
    const uint32_t lane = threadIdx.x & 63u;
    // (whole wavefronts stride through the frontier: the shuffles below need all 64 lanes in every pass)
    for (uint32_t i0 = (blockIdx.x * blockDim.x + threadIdx.x) & ~63u; i0 < n_items; i0 += gridDim.x * blockDim.x) {
    const uint32_t i = i0 + lane;
    bool mine = false;
    uint32_t id = 0, len = 0;
    if (i < n_items) {
        id = cur.items[i].id;
        const uint32_t bit = 1u << (id & 31u);
        mine = (atomicOr(&t.resident[id >> 5], bit) & bit) == 0u;
        if (mine) len = t.len[id];
    }
    const uint64_t want = mine ? (len == 0 ? 128ull : (uint64_t(len) + 127ull) & ~127ull) : 0ull;
    uint64_t incl = want, raw = mine ? uint64_t(len) : 0ull;
    uint32_t cnt = mine ? 1u : 0u;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint64_t up = __shfl_up(incl, d, 64), ur = __shfl_up(raw, d, 64);
        const uint32_t uc = __shfl_up(cnt, d, 64);
        if (lane >= uint32_t(d)) {
            incl += up;
            raw += ur;
            cnt += uc;
        }
    }
    const uint64_t wave_bytes = __shfl(incl, 63, 64);
    const uint32_t wave_cnt = __shfl(cnt, 63, 64);
    if (wave_cnt == 0) continue;
    unsigned long long base_bytes = 0;
    uint32_t base_copy = 0, base_pulled = 0;
    if (lane == 63) {
        base_bytes = atomicAdd(&ctl->stage_used, (unsigned long long)wave_bytes);
        (void)atomicAdd(&ctl->payload, (unsigned long long)raw);
        base_copy = atomicAdd(&ctl->n_copy, wave_cnt);
        base_pulled = atomicAdd(&ctl->n_pulled, wave_cnt);
    }
    base_bytes = __shfl(base_bytes, 63, 64);
    base_copy = __shfl(base_copy, 63, 64);
    base_pulled = __shfl(base_pulled, 63, 64);
    if (!mine) continue;
    const uint64_t dst = base_bytes + incl - want;
    const uint32_t k = base_copy + cnt - 1u, p = base_pulled + cnt - 1u;
    if (dst + want > t.stage_cap || p >= t.pulled_cap) {
        atomicOr(&ctl->overflow, 2u);
        continue;
    }
    t.stage_off[id] = dst;
    t.pulled[p] = id;
    t.copy_src[k] = t.goff[id];
    t.copy_dst[k] = dst;
    t.copy_len[k] = len;
    }
}

// COPY: this round's blocks, host memory → staging arena: half a wavefront per block, 16 destination bytes per lane.
// Every SOURCE byte crosses PCIe once: lane j loads the ALIGNED 16-byte chunk j of the block's span, takes chunk j + 1
// from its neighbour (a shuffle; the last lane of the half loads it itself) and funnel-shifts the pair by the block's
// misalignment.  (k_repack's three overlapping 8-byte words per lane are fine out of HBM, where the overlap is an L2 hit;
// host memory is not cached, and the overlap went over the link again: 36 GB/s against 48 for this form.)  Only chunks
// that hold at least one byte of the block are read, so the host buffer needs no slack beyond its last 16-byte chunk.
__global__ __launch_bounds__(256) void k_pull_copy(const uint8_t* __restrict__ src, const uint64_t* __restrict__ src_off,
                                                   const uint32_t* __restrict__ len, const uint64_t* __restrict__ dst_off,
                                                   const PullCtl* __restrict__ ctl, uint8_t* __restrict__ dst) {
    if (ctl->overflow) return;
    const uint32_t n = ctl->n_copy;
    const uint32_t sub = threadIdx.x & 31;
    const uint32_t group = (blockIdx.x * blockDim.x + threadIdx.x) >> 5;
    const uint32_t ngroups = (gridDim.x * blockDim.x) >> 5;
    const uint32_t rounds = (n + ngroups - 1) / ngroups;  // (every lane of a wavefront runs the same number of rounds: shuffles)
    for (uint32_t it = 0; it < rounds; ++it) {
        const uint32_t i = group + it * ngroups;
        const bool live = i < n;
        const uint64_t so = live ? src_off[i] : 0;
        const uint32_t L = live ? len[i] : 0;
        uint8_t* d = dst + (live ? dst_off[i] : 0);
        const uint32_t padded = !live ? 0u : (L == 0 ? 128u : (L + 127u) & ~127u);
        const uintptr_t s = reinterpret_cast<uintptr_t>(src) + so;
        const uint32_t mis = uint32_t(s & 15u);
        const ulonglong2* chunks = reinterpret_cast<const ulonglong2*>(s - mis);
        const uint32_t span = live ? mis + L : 0u;  // bytes from the first chunk's start to the block's end
        const uint32_t steps = (padded + 511u) / 512u;
        const uint32_t max_steps = max(__shfl(steps, 0, 64), __shfl(steps, 32, 64));  // the two halves of the wavefront in step
        for (uint32_t st = 0; st < max_steps; ++st) {
            const uint32_t j = st * 32u + sub;  // destination chunk
            ulonglong2 c = make_ulonglong2(0, 0);
            if (16u * j < span) c = chunks[j];
            ulonglong2 nx;
            nx.x = __shfl_down(c.x, 1, 32);
            nx.y = __shfl_down(c.y, 1, 32);
            if (sub == 31u) {
                nx = make_ulonglong2(0, 0);
                if (16u * (j + 1u) < span) nx = chunks[j + 1u];
            }
            // 16 bytes that start `mis` bytes into chunk j
            const uint64_t w0 = mis < 8u ? c.x : c.y, w1 = mis < 8u ? c.y : nx.x, w2 = mis < 8u ? nx.x : nx.y;
            const uint32_t sh = (mis & 7u) * 8u;
            uint64_t o0 = (w0 >> sh) | ((w1 << 1) << (63u - sh));
            uint64_t o1 = (w1 >> sh) | ((w2 << 1) << (63u - sh));
            const uint32_t u = 16u * j;  // destination byte
            if (u < padded) {
                const uint32_t valid = u < L ? L - u : 0u;
                if (valid < 8u) {
                    o0 &= valid ? (1ull << (8u * valid)) - 1ull : 0ull;
                    o1 = 0;
                } else if (valid < 16u) {
                    o1 &= valid == 8u ? 0ull : (1ull << (8u * (valid - 8u))) - 1ull;
                }
                *reinterpret_cast<ulonglong2*>(d + u) = make_ulonglong2(o0, o1);
            }
        }
    }
}

// EXPAND: lane = frontier item; its block is in the staging arena now.
__device__ __forceinline__ void pull_expand_item(const WitnessView& w, const PullItem& it, const PullFrontier& next, PullCtl* ctl,
                                                 uint32_t n_shards, uint32_t shard) {
    const uint32_t kind = it.kind & 0xffu, height = (it.kind >> 8) & 0xffu;
    if (kind == PK_LEAF) return;
    Rd r = open_block(w, it.id);
    switch (kind) {
        case PK_HDR_CHILD:
        case PK_HDR_PARENT: {
            HeaderLite h;
            if (decode_header(r, h) != IPCFP_ST_TRUE) return;
            if (kind == PK_HDR_CHILD) pull_emit(next, ctl, witness_find(w, h.parent_message_receipts), PK_RCPT_ROOT, 0, 0);
            else pull_emit(next, ctl, witness_find(w, h.messages), PK_TXMETA, 0, 0);
            return;
        }
        case PK_TXMETA: {  // [bls_root, secp_root]  (src/proofs/events/utils.rs:61)
            CidKey a, b;
            r.expect_array(2);
            r.read_link_key(a);
            r.read_link_key(b);
            if (!r.ok()) return;
            pull_emit(next, ctl, witness_find(w, a), PK_MSG_ROOT, 0, 0);
            pull_emit(next, ctl, witness_find(w, b), PK_MSG_ROOT, 0, 0);
            return;
        }
        case PK_MSG_ROOT: {  // Amtv0 root [height, count, node]: the whole tree is needed
            r.expect_array(3);
            const uint64_t ht = r.read_uint();
            (void)r.read_uint();
            if (!r.ok() || ht > 64u) return;
            expand_node_all(w, r, next, ctl, PK_MSG_NODE, uint32_t(ht));
            return;
        }
        case PK_MSG_NODE:
        case PK_EV_NODE:
            expand_node_all(w, r, next, ctl, kind, height);
            return;
        case PK_RCPT_ROOT: {  // Amtv0 root of the receipts: its count decides the cut
            r.expect_array(3);
            const uint64_t ht = r.read_uint();
            const uint64_t count = r.read_uint();
            if (!r.ok() || ht > 20u) return;
            const uint64_t q = count / n_shards, rem = count % n_shards;  // == ipcfp_shard_range
            const uint64_t lo = q * shard + (rem * shard) / n_shards;
            const uint64_t hi = q * (shard + 1u) + (rem * (uint64_t(shard) + 1u)) / n_shards;
            ctl->lo = lo;
            ctl->hi = hi;
            ctl->n_receipts = count;
            ctl->have_range = 1u;
            const uint64_t hi_walk = shard + 1u == n_shards ? ~0ull : hi;
            ctl->hi_walk = hi_walk;
            expand_receipts_node(w, r, next, ctl, uint32_t(ht), 0, lo, hi_walk);
            return;
        }
        case PK_RCPT_NODE:
            expand_receipts_node(w, r, next, ctl, height, it.base, ctl->lo, ctl->hi_walk);
            return;
        case PK_EV_ROOT: {  // Amt (v3) root [bit_width, height, count, node]: taller than a leaf ⇒ every node below it
            r.expect_array(4);
            (void)r.read_uint();
            const uint64_t ht = r.read_uint();
            (void)r.read_uint();
            if (!r.ok() || ht > 64u) return;
            expand_node_all(w, r, next, ctl, PK_EV_NODE, uint32_t(ht));
            return;
        }
        default:
            return;
    }
}

__global__ __launch_bounds__(256) void k_pull_expand(WitnessView w, PullFrontier cur, PullFrontier next, PullCtl* __restrict__ ctl,
                                                     uint32_t n_shards, uint32_t shard) {
    if (ctl->overflow) return;
    const uint32_t n_items = min(ctl->n_cur, cur.cap);
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n_items; i += gridDim.x * blockDim.x)
        pull_expand_item(w, cur.items[i], next, ctl, n_shards, shard);
}

// between two rounds: the next frontier's size goes to the host (mailbox), the round counters start again
__global__ void k_pull_round_end(PullCtl* __restrict__ ctl, uint32_t cap, unsigned long long* __restrict__ mailbox,
                                 unsigned long long seq) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    // an outgrown frontier ends the walk HERE: no later round reads past `cap` items (pull_emit counts on beyond it), and
    // the host, which has rounds queued ahead of what it has read, is told through the mailbox's overflow word
    const uint32_t n_next = ctl->overflow ? 0u : min(ctl->n_next, cap);
    ctl->n_cur = n_next;
    ctl->n_next = 0;
    ctl->n_copy = 0;
    // (rounds are queued ahead of the host's reading: each publishes into the slot of its own parity, the sequence number last)
    unsigned long long* slot = mailbox + 8u * (seq & 3ull);
    __hip_atomic_store(slot + 1, (unsigned long long)n_next, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(slot + 2, (unsigned long long)ctl->overflow, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(slot, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

int launch_pull_seed(ipcfp_ctx* ctx, const WitnessView& w, const PullSeeds& seeds, const PullFrontier& first, PullCtl* ctl_d) {
    hipLaunchKernelGGL(k_pull_seed, dim3(1), dim3(64), 0, ctx->stream, w, seeds, first, ctl_d);
    IPCFP_HIP(ctx, hipGetLastError());
    return IPCFP_OK;
}

// One round, sized by `n_hint` (the host's last word on the frontier: the exact size when it has waited for the round
// before, an upper bound / guess when rounds are queued ahead — every kernel takes the true count from the device and
// strides, so a wrong guess costs time, not results).
int launch_pull_round(ipcfp_ctx* ctx, const WitnessView& w, const uint8_t* host_bytes_dev, const PullTables& t, const PullFrontier& cur,
                      uint32_t n_hint, const PullFrontier& next, PullCtl* ctl_d, uint32_t n_shards, uint32_t shard,
                      unsigned long long* mailbox_dev, unsigned long long seq) {
    if (n_hint) {
        const uint32_t wgs = std::min(div_up(n_hint, 256), 4096u);
        hipLaunchKernelGGL(k_pull_claim, dim3(wgs), dim3(256), 0, ctx->stream, cur, t, ctl_d);
        const uint32_t groups = std::min(n_hint, 8192u * 8u);  // half-wavefronts (at most one new block per item)
        hipLaunchKernelGGL(k_pull_copy, dim3(div_up(groups, 8)), dim3(256), 0, ctx->stream, host_bytes_dev, t.copy_src, t.copy_len,
                           t.copy_dst, ctl_d, t.stage);
        hipLaunchKernelGGL(k_pull_expand, dim3(wgs), dim3(256), 0, ctx->stream, w, cur, next, ctl_d, n_shards, shard);
    }
    hipLaunchKernelGGL(k_pull_round_end, dim3(1), dim3(64), 0, ctx->stream, ctl_d, next.cap, mailbox_dev, seq);
    IPCFP_HIP(ctx, hipGetLastError());
    return IPCFP_OK;
}

}  // namespace ipcfp
