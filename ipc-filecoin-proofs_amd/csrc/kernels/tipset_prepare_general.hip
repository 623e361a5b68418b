// csrc/kernels/tipset_prepare_general.hip — the tipset prologue with the general (global-memory) reader: the context
// kernels of the generator / shard-plan paths (k_ctx_headers, k_exec_roots) and the companion of the LDS prologue
// (tipset_prepare.hip) for blocks that do not fit its stage.  A unit of its own so that these single-wavefront
// workgroups carry 8 KB of LDS and not the line-staging area of verify_events.hip: beside the block-order event
// parse a 43 KB workgroup waits for a CU to drain.
#include <hip/hip_runtime.h>

#include <cstring>

#include "../common.h"
#include "blake2b_dev.h"
#include "claims_dev.h"
#include "amt_enum.h"
#include "event_table.h"
#include "tipset_ctx.h"
#include "types_dev.h"
#include "launch.h"

namespace ipcfp {

// ---------------------------------------------------------------------------
// context headers: one thread per context
// ---------------------------------------------------------------------------
// One wavefront (= one 64-thread workgroup) per context: the wave stages each header in LDS, lane 0 parses.
constexpr uint32_t kHeaderLds = 8192;

// Two wavefronts per context, side by side: block 2t decodes the child header, block 2t+1 the first
// parent header (a header decode is ~100 CBOR items parsed by ONE lane — tens of microseconds of pure
// latency — so the two are not done one after the other).
// `receipts_spec` (nullable, child part): where to leave the receipts AMT as an enumeration root, so that the
// enumerator can take it along with the message AMTs without the host having seen the header (amt_enum.h EnumExtra)
__device__ __forceinline__ void ctx_headers_body(const WitnessView& w, TipsetCtxDev& c, bool child_part, uint8_t* lds,
                                                 AmtRootSpec* receipts_spec = nullptr) {
    const bool lead = threadIdx.x == 0;
    const bool parsed = (c.flags & (TC_PARENTS_PARSED | TC_CHILD_PARSED)) == (TC_PARENTS_PARSED | TC_CHILD_PARSED);
    if (child_part) {
        uint32_t status = IPCFP_ST_ERR_BAD_CLAIM, match = 0;
        long long height = 0;
        if (parsed) {
            // child header (events/verifier.rs:155-158)
            const uint32_t hb = witness_find(w, c.child);  // uniform across the wave
            if (hb == kNoBlock) {
                status = IPCFP_ST_ERR_MISSING_BLOCK;
            } else {
                Rd r = open_block_staged(w, hb, lds, kHeaderLds);
                if (lead) {
                    HeaderLite h;
                    status = decode_header(r, h);
                    if (status == IPCFP_ST_TRUE) {
                        height = h.height;
                        c.receipts_root = h.parent_message_receipts;
                        // `child_hdr.parents != parent_cids` (:161): same count, same CIDs in order
                        bool same = h.n_parents == c.n_parents;
                        if (same) {
                            Rd q = r;
                            q.err = 0;
                            q.pos = h.parents_off;
                            for (uint32_t i = 0; i < c.n_parents && same; ++i) {
                                CidKey k;
                                q.read_link_key(k);
                                same = q.ok() && cid_equal(k, tipset_parent(c, i));
                            }
                        }
                        match = same ? 1u : 0u;
                    }
                }
            }
        }
        if (lead) {
            c.child_status = status;
            c.parents_match = match;
            c.child_height = height;
            if (receipts_spec) {
                AmtRootSpec rs{};
                rs.version = 0;  // Amtv0<Receipt>
                rs.kind_p1 = uint32_t(VK_RECEIPT) + 1u;
                rs.skip = status == IPCFP_ST_TRUE ? 0u : 1u;
                if (!rs.skip) rs.root = c.receipts_root;
                *receipts_spec = rs;
            }
        }
    } else {
        uint32_t status = IPCFP_ST_ERR_BAD_CLAIM;
        long long height = 0;
        if (parsed && c.n_parents > 0) {  // parent_cids[0] (:171-174)
            const uint32_t pb = witness_find(w, tipset_parent(c, 0));
            if (pb == kNoBlock) {
                status = IPCFP_ST_ERR_MISSING_BLOCK;
            } else {
                Rd r = open_block_staged(w, pb, lds, kHeaderLds);
                if (lead) {
                    HeaderLite ph;
                    status = decode_header(r, ph);
                    if (status == IPCFP_ST_TRUE) height = ph.height;
                }
            }
        }
        if (lead) {
            c.parent0_status = status;
            c.parent0_height = height;
        }
    }
}

__global__ __launch_bounds__(64) void k_ctx_headers(WitnessView w, TipsetCtxDev* __restrict__ ctxs, uint32_t n) {
    __shared__ __attribute__((aligned(16))) uint8_t lds[kHeaderLds];
    const uint32_t t = blockIdx.x >> 1;
    if (t >= n) return;
    ctx_headers_body(w, ctxs[t], (blockIdx.x & 1u) == 0, lds);
}

// ---------------------------------------------------------------------------
// execution order, stage 1 (one thread): parent headers → TxMeta → AMT roots
//   error sequence numbers: parent header b → b;  TxMeta of block b → P + 3b;
//   its BLS AMT → P + 3b + 1;  its secp AMT → P + 3b + 2   (traversal order of utils.rs)
// ---------------------------------------------------------------------------
// One lane per parent block (the per-block work is independent; the error word orders the outcomes).
// `err` must hold kNoEnumError on entry.
__device__ __forceinline__ void exec_roots_body(const WitnessView& w, const TipsetCtxDev* __restrict__ ctx,
                                                AmtRootSpec* __restrict__ roots, unsigned long long* __restrict__ err,
                                                int verify_txmeta, uint32_t b, uint8_t* lds) {
    const uint32_t P = ctx->n_parents;  // one wavefront per parent block b; lane 0 parses what the wave staged
    if (b >= P) return;
    const bool lead = threadIdx.x == 0;
    auto fail = [&](uint32_t seq, uint32_t code) { atomicMin(err, (unsigned long long)pack_enum_error(seq, 0, code)); };
    // reconstruct_execution_order (utils.rs:20-27): every parent header is decoded first
    CidKey tx[1];
    bool have_tx[1];
    {
        have_tx[0] = false;
        const uint32_t hb = witness_find(w, tipset_parent(*ctx, b));
        if (hb == kNoBlock) {
            if (lead) fail(b, IPCFP_ST_ERR_MISSING_BLOCK);
        } else {
            Rd hr = open_block_staged(w, hb, lds, kHeaderLds);
            if (lead) {
                HeaderLite h;
                const uint32_t st = decode_header(hr, h);
                have_tx[0] = st == IPCFP_ST_TRUE;
                if (have_tx[0]) tx[0] = h.messages;
                else fail(b, st);
            }
        }
    }
    if (!lead) return;  // the rest is a short chain on small blocks
    // collect_exec_list (utils.rs:56-91)
    {
        const uint32_t seq = P + 3 * b;
        AmtRootSpec bls{}, secp{};
        bls.version = secp.version = 0;
        bls.seq = seq + 1;
        secp.seq = seq + 2;
        bls.skip = secp.skip = 1;
        if (have_tx[0]) {
            const uint32_t tb = witness_find(w, tx[0]);  // :58-60
            if (tb == kNoBlock) {
                fail(seq, IPCFP_ST_ERR_MISSING_BLOCK);
            } else {
                Rd r = open_block(w, tb);
                uint32_t o0, l0, o1, l1;
                r.expect_array(2);  // (Cid, Cid)  :61
                r.read_link(o0, l0);
                r.read_link(o1, l1);
                r.finish();
                if (!r.ok()) {
                    fail(seq, IPCFP_ST_ERR_DECODE);
                } else {
                    // put_cbor(&(bls_root, secp_root), Blake2b256): canonical re-encoding, hashed (:65-72)
                    uint8_t enc[200];
                    uint32_t n = 0;
                    enc[n++] = 0x82;
                    const uint32_t offs[2] = {o0, o1}, lens[2] = {l0, l1};
                    for (int k = 0; k < 2; ++k) {
                        enc[n++] = 0xd8;
                        enc[n++] = 0x2a;
                        const uint32_t bl = lens[k] + 1;
                        if (bl < 24) enc[n++] = uint8_t(0x40 | bl);
                        else { enc[n++] = 0x58; enc[n++] = uint8_t(bl); }
                        enc[n++] = 0x00;
                        for (uint32_t i = 0; i < lens[k]; ++i) enc[n++] = uint8_t(r.at(offs[k] + i));
                    }
                    uint64_t d[4];
                    blake2b256_small(enc, n, d);
                    CidKey re;
                    re.w[0] = 0x00002002e4a07101ULL | (d[0] << 48);
                    re.w[1] = (d[0] >> 16) | (d[1] << 48);
                    re.w[2] = (d[1] >> 16) | (d[2] << 48);
                    re.w[3] = (d[2] >> 16) | (d[3] << 48);
                    re.w[4] = d[3] >> 16;
                    // verify_txmeta = false on the generation path (build_execution_order, utils.rs:44)
                    if (verify_txmeta && !cid_equal(re, tx[0])) {
                        fail(seq, IPCFP_ST_ERR_TXMETA_MISMATCH);
                    } else {
                        bls.root = r.key_any(o0, l0);
                        secp.root = r.key_any(o1, l1);
                        bls.skip = secp.skip = 0;
                    }
                }
            }
        }
        roots[2 * b] = bls;
        roots[2 * b + 1] = secp;
    }
}

__global__ __launch_bounds__(64) void k_exec_roots(WitnessView w, const TipsetCtxDev* __restrict__ ctx,
                                                   AmtRootSpec* __restrict__ roots,
                                                   unsigned long long* __restrict__ err, int verify_txmeta) {
    __shared__ __attribute__((aligned(16))) uint8_t lds[kHeaderLds];
    exec_roots_body(w, ctx, roots, err, verify_txmeta, blockIdx.x, lds);
}

// The general companion of the tipset prologue (tipset_prepare.hip): the same jobs and slots, for the slots that
// kernel left alone because a block did not fit its LDS stage (TipsetCtxDev::prologue_general).  Normally every
// workgroup leaves at once.
__global__ __launch_bounds__(64) void k_tipset_prepare_general(WitnessView w, PrepareJobs jobs, uint32_t n_jobs) {
    __shared__ __attribute__((aligned(16))) uint8_t lds[kHeaderLds];
    const uint32_t job = blockIdx.x / kPrepareSlots, slot = blockIdx.x % kPrepareSlots;
    if (job >= n_jobs) return;
    const PrepareJob jb = prepare_job(jobs, job);
    if (!((jb.ctx->prologue_general >> slot) & 1ull)) return;
    if (slot < 2) ctx_headers_body(w, *jb.ctx, slot == 0, lds, jb.roots ? jb.roots + 2u * jb.ctx->n_parents : nullptr);
    else if (jb.roots) exec_roots_body(w, jb.ctx, jb.roots, jb.err, 1, slot - 2, lds);
}

// The whole prologue of ONE context whose tipset key is wider than the inline form (TipsetCtxDev::parents_wide): workgroups
// 0 / 1 the header facts (child header with the receipts root as the enumeration's extra spec, first parent header),
// workgroup 2 + b parent block b's header → TxMeta → re-hash → message-AMT roots.  General reader throughout.
__global__ __launch_bounds__(64) void k_tipset_prepare_wide(WitnessView w, PrepareJob jb) {
    __shared__ __attribute__((aligned(16))) uint8_t lds[kHeaderLds];
    if (blockIdx.x < 2) ctx_headers_body(w, *jb.ctx, blockIdx.x == 0, lds, jb.roots ? jb.roots + 2u * jb.ctx->n_parents : nullptr);
    else if (jb.roots) exec_roots_body(w, jb.ctx, jb.roots, jb.err, 1, blockIdx.x - 2u, lds);
}

int launch_tipset_prepare_wide(ipcfp_ctx* ctx, const WitnessView& w, const void* job, uint32_t n_parents) {
    hipLaunchKernelGGL(k_tipset_prepare_wide, dim3(2u + n_parents), dim3(64), 0, ctx->stream, w, *static_cast<const PrepareJob*>(job));
    IPCFP_HIP(ctx, hipGetLastError());
    return IPCFP_OK;
}

void launch_tipset_prepare_lds(hipStream_t stream, const WitnessView& w, const PrepareJobs& jobs, uint32_t n_jobs,
                               const uint32_t* live_done, uint32_t live_total, uint32_t* anomaly, bool defer_rehash,
                               const TipsetInputs* inline_inputs);  // tipset_prepare.hip

// `jobs`: HOST array; `jobs_d`: device copy, needed (and read) only when there are more than kInlineJobs.
// `need_general`: some block of the witness may exceed the LDS stage, so the general companion has to look.
int launch_tipset_prepare(ipcfp_ctx* ctx, const WitnessView& w, const void* jobs, const void* jobs_d, uint32_t n_jobs,
                          bool need_general, const uint32_t* live_done, uint32_t live_total, uint32_t* anomaly,
                          bool defer_rehash, const void* inline_inputs) {
    if (defer_rehash && need_general) return set_error(ctx, IPCFP_E_INVALID, "deferred TxMeta re-hash: LDS slots only");
    if (live_done && (need_general || !anomaly)) return set_error(ctx, IPCFP_E_INVALID, "live prologue: LDS slots only");
    if (n_jobs == 0) return IPCFP_OK;
    PrepareJobs pj{};
    if (n_jobs <= kInlineJobs) std::memcpy(pj.inline_jobs, jobs, size_t(n_jobs) * sizeof(PrepareJob));
    else pj.more = static_cast<const PrepareJob*>(jobs_d);
    // (`inline_inputs`: the ONE context's TipsetInputs on the host — they ride in the kernel arguments and the context's
    // zeroed device copy is filled in by the launch itself)
    launch_tipset_prepare_lds(ctx->stream, w, pj, n_jobs, live_done, live_total, anomaly, defer_rehash,
                              static_cast<const TipsetInputs*>(inline_inputs));
    if (need_general)
        hipLaunchKernelGGL(k_tipset_prepare_general, dim3(n_jobs * kPrepareSlots), dim3(64), 0, ctx->stream, w, pj, n_jobs);
    IPCFP_HIP(ctx, hipGetLastError());
    return IPCFP_OK;
}

int launch_ctx_headers(ipcfp_ctx* ctx, const WitnessView& w, TipsetCtxDev* ctxs_d, uint32_t n) {
    if (n == 0) return IPCFP_OK;
    hipLaunchKernelGGL(k_ctx_headers, dim3(2 * n), dim3(64), 0, ctx->stream, w, ctxs_d, n);
    IPCFP_HIP(ctx, hipGetLastError());
    return IPCFP_OK;
}

int launch_exec_roots(ipcfp_ctx* ctx, const WitnessView& w, const TipsetCtxDev* ctx_d, AmtRootSpec* roots_d,
                      unsigned long long* err_d, int verify_txmeta, uint32_t n_parents) {
    const unsigned long long none = kNoEnumError;
    IPCFP_HIP(ctx, hipMemsetAsync(err_d, 0xff, 8, ctx->stream));  // kNoEnumError
    (void)none;
    // (one workgroup per parent block; the grid of the inline form is fixed so that the host need not know the count)
    hipLaunchKernelGGL(k_exec_roots, dim3(n_parents > IPCFP_MAX_PARENTS ? n_parents : IPCFP_MAX_PARENTS), dim3(64), 0, ctx->stream, w, ctx_d,
                       roots_d, err_d, verify_txmeta);
    IPCFP_HIP(ctx, hipGetLastError());
    return IPCFP_OK;
}

}  // namespace ipcfp
