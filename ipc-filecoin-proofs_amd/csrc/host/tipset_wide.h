// csrc/host/tipset_wide.h — a tipset context's INPUTS from what crosses the ABI, for tipset keys of any length.
//
// The reference takes any tipset key (src/proofs/events/verifier.rs:147-181 compares `child_hdr.parents` with the claimed
// list whatever its length, src/proofs/events/utils.rs:16-30 walks every parent header).  Keys of up to IPCFP_MAX_PARENTS
// blocks — every key a chain has produced — travel inline (TipsetCtxDev::parents, kernel arguments).  A longer key is kept
// whole in HBM (TipsetCtxDev::parents_wide) and such a context takes the general route: k_tipset_prepare_wide / k_exec_roots
// with one workgroup per parent block, the level-by-level enumerator (the dense walk plans at most kMaxDenseRoots trees).
#pragma once
#include <cstring>
#include <memory>
#include <vector>

#include "../common.h"
#include "../kernels/tipset_ctx.h"

namespace ipcfp {

CidKey key_from_slot(const uint8_t* slot40);

// the device copies of the wide keys of one call (alive until the call's entry point returns)
struct WideParents {
    std::vector<std::unique_ptr<DevBuf<CidKey>>> bufs;
    std::vector<std::unique_ptr<std::vector<CidKey>>> host;  // the upload's source must outlive the copy
};

inline bool tipset_is_wide(uint32_t n_parents) { return n_parents > IPCFP_MAX_PARENTS; }

// parents: `inline40` holds the first min(n, IPCFP_MAX_PARENTS) slots, `more40` the rest (nullable when there is none)
inline int tipset_inputs(ipcfp_ctx* ctx, uint32_t flags, uint32_t n_parents, const uint8_t* child40, const uint8_t* inline40,
                         const uint8_t* more40, TipsetCtxDev& tc, WideParents& keep) {
    if (n_parents > IPCFP_MAX_PARENTS_WIDE)
        return set_error(ctx, IPCFP_E_UNSUPPORTED, "a tipset key of %u parent blocks (the enumeration orders %u)", n_parents,
                         unsigned(IPCFP_MAX_PARENTS_WIDE));
    if (tipset_is_wide(n_parents) && !more40) return set_error(ctx, IPCFP_E_INVALID, "a tipset key of %u parent blocks without more_parents", n_parents);
    std::memset(&tc, 0, sizeof tc);
    tc.flags = flags;
    tc.n_parents = n_parents;
    if (child40) tc.child = key_from_slot(child40);
    const uint32_t n_inline = n_parents < IPCFP_MAX_PARENTS ? n_parents : uint32_t(IPCFP_MAX_PARENTS);
    for (uint32_t j = 0; j < n_inline; ++j) tc.parents[j] = key_from_slot(inline40 + size_t(j) * IPCFP_CID_SLOT);
    if (!tipset_is_wide(n_parents)) return IPCFP_OK;
    keep.host.emplace_back(new std::vector<CidKey>(n_parents));
    std::vector<CidKey>& all = *keep.host.back();
    for (uint32_t j = 0; j < n_inline; ++j) all[j] = tc.parents[j];
    for (uint32_t j = n_inline; j < n_parents; ++j) all[j] = key_from_slot(more40 + size_t(j - n_inline) * IPCFP_CID_SLOT);
    keep.bufs.emplace_back(new DevBuf<CidKey>());
    DevBuf<CidKey>& d = *keep.bufs.back();
    IPCFP_HIP(ctx, d.alloc_unpooled(n_parents));  // (outlives the pooled scratch of the call's stages)
    IPCFP_HIP(ctx, hipMemcpyAsync(d.p, all.data(), size_t(n_parents) * sizeof(CidKey), hipMemcpyHostToDevice, ctx->stream));
    tc.parents_wide = d.p;
    return IPCFP_OK;
}

inline int tipset_inputs(ipcfp_ctx* ctx, const ipcfp_tipset_ref_t& ref, TipsetCtxDev& tc, WideParents& keep) {
    return tipset_inputs(ctx, ref.flags, ref.n_parents, ref.child, &ref.parents[0][0], ref.more_parents, tc, keep);
}

// a plain list of n_parents slots (ipcfp_exec_order, the generator, the shard planner)
inline int tipset_inputs_list(ipcfp_ctx* ctx, uint32_t flags, const uint8_t* parent_cids40, uint32_t n_parents, const uint8_t* child40,
                              TipsetCtxDev& tc, WideParents& keep) {
    return tipset_inputs(ctx, flags, n_parents, child40, parent_cids40,
                         n_parents > IPCFP_MAX_PARENTS ? parent_cids40 + size_t(IPCFP_MAX_PARENTS) * IPCFP_CID_SLOT : nullptr, tc, keep);
}

// every parent key of a context on the host (the replicated part of a shard plan, a generator's touched set)
inline void tipset_parent_keys(const TipsetCtxDev& tc, const WideParents& keep, std::vector<CidKey>& out) {
    if (tc.parents_wide)
        for (const auto& h : keep.host)
            if (h->size() == tc.n_parents) {  // (one wide context per such call)
                out.insert(out.end(), h->begin(), h->end());
                return;
            }
    for (uint32_t k = 0; k < tc.n_parents && k < IPCFP_MAX_PARENTS; ++k) out.push_back(tc.parents[k]);
}

}  // namespace ipcfp
