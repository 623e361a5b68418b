// csrc/kernels/tipset_prepare.hip — the tipset prologue of the verify path, parsed out of LDS.
//
// What `verify_single_proof` derives from (parent_tipset_cids, child_block_cid) before it looks at a proof: the child
// header and the first parent header (verify_header_consistency, src/proofs/events/verifier.rs:147-181) and, per parent
// block, its header, its TxMeta and the TxMeta's re-hash (reconstruct_execution_order / collect_exec_list,
// src/proofs/events/utils.rs:16-30,48-94).  Each is a parse of ≈100 CBOR items by ONE lane — there is no data
// parallelism inside a header — so the launch is a handful of single-wavefront workgroups whose time is the length
// of one lane's instruction stream, stretched further by whatever shares the CU (K1, the block-order event parse).
// Hence this unit: the wavefront stages the block in LDS and the lane parses it with the LDS reader (cbor_dev.h
// IPCFP_RD_LDS: a byte is a ds_read_u8, a header one aligned ds_read2_b64 — no window to maintain), a third of the
// instructions of the windowed reader the same parse needs on global memory.
//
// A block that does not fit the stage is not parsed here: the slot is flagged in TipsetCtxDev::prologue_general and
// k_tipset_prepare_general (verify_events.hip), launched right behind, does that slot the general way.
#define IPCFP_RD_LDS 1
#include <hip/hip_runtime.h>

#include "../common.h"
#include "blake2b_dev.h"
#include "claims_dev.h"
#include "header_dev.h"
#include "tipset_ctx.h"

namespace ipcfp {


typedef __attribute__((address_space(3))) const uint8_t* lds_bytes_t;

// ---- lookups in an index that is still being filled ---------------------------------------------------------------------
// k_index_insert (main stream) fills the table while the prologue's wavefronts (head stream, host/verify_fast.cpp) look
// their few CIDs up: the prologue is a chain of dependent steps per slot — look the header up, stage it, decode ≈ 100
// CBOR items on one lane, look the TxMeta up, decode it, re-hash it — ≈ 130 µs that used to start when the ≈ 120 µs of
// inserts had finished.  Such a lookup
// can only err one way: a key that is not in the table YET (an entry never moves and the slots in front of it never
// empty, so a key that is in is found), or — a CID that occurs twice — an id that a later insert still raises.  So a
// miss is retried until the key shows up or every insert workgroup has counted itself done, whatever was found is
// taken as provisional, and at the end of its slot the wavefront waits for the count and looks every key up once more:
// a different answer (or a wait that ran out) raises the call's anomaly flag, and the caller does the batch again the
// general way.  Reads of the table are device-scope atomic loads (other XCDs' L2s are filling it).
constexpr uint32_t kLiveSpinMax = 1u << 21;  // ≈ 2 s of 1 µs naps: a launch that cannot finish must not hang the device
struct LiveIndex {
    const uint32_t* done;  // workgroups of the insert part that have finished; null: the index is complete (plain lookups)
    uint32_t total;
    uint32_t n_used;
    bool failed;
    CidKey k0, k1;  // (two named slots, not an array: a dynamically indexed member would live in scratch)
    uint32_t b0, b1;
};
__device__ __forceinline__ uint32_t live_probe(const WitnessView& w, const CidKey& key) {
    uint32_t s = cid_hash(key) & w.mask;
    for (;;) {
        const uint32_t b = __hip_atomic_load(&w.slots[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (b == kNoBlock) return kNoBlock;
        if (cid_equal(load_cid_slot(w.cids, b), key)) return b;
        s = (s + 1) & w.mask;
    }
}
__device__ __forceinline__ bool live_complete(const LiveIndex& li) {
    return __hip_atomic_load(li.done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= li.total;
}
// every lane of the wavefront, same key
__device__ __forceinline__ uint32_t find_block(const WitnessView& w, LiveIndex& li, const CidKey& key) {
    if (!li.done) return witness_find(w, key);
    uint32_t b = kNoBlock;
    for (uint32_t spin = 0;; ++spin) {
        const bool complete = live_complete(li);  // read BEFORE the probe: a miss after `complete` is final
        b = live_probe(w, key);
        if (b != kNoBlock || complete) break;
        if (spin >= kLiveSpinMax) {
            li.failed = true;
            break;
        }
        __builtin_amdgcn_s_sleep(32);
    }
    if (li.n_used == 0) {
        li.k0 = key;
        li.b0 = b;
    } else if (li.n_used == 1) {
        li.k1 = key;
        li.b1 = b;
    } else {
        li.failed = true;
    }
    ++li.n_used;
    return b;
}
// end of a slot: the provisional answers against the finished index
__device__ __forceinline__ void live_validate(const WitnessView& w, LiveIndex& li, uint32_t* anomaly) {
    if (!li.done) return;
    bool bad = li.failed;
    for (uint32_t spin = 0; !bad && !live_complete(li); ++spin) {
        if (spin >= kLiveSpinMax) bad = true;
        __builtin_amdgcn_s_sleep(32);
    }
    if (!bad && li.n_used > 0) bad = live_probe(w, li.k0) != li.b0;
    if (!bad && li.n_used > 1) bad = live_probe(w, li.k1) != li.b1;
    if (bad && threadIdx.x == 0) atomicOr(anomaly, 1u);
}

// every lane of the wavefront: block b → the stage; false when it does not fit
__device__ __forceinline__ bool stage_block(const WitnessView& w, uint32_t b, rd_chunk_t* stage, uint32_t& len) {
    len = w.len[b];
    if (len + 32u > kPrologueStageChunks * 16u) return false;
    const rd_chunk_t* src = reinterpret_cast<const rd_chunk_t*>(w.arena + w.off[b]);  // line-aligned, padded, + tail slack
    const uint32_t chunks = ((len + 15u) >> 4) + 2u;
    for (uint32_t i = threadIdx.x; i < chunks; i += 64u) stage[i] = src[i];
    __syncthreads();
    return true;
}

__device__ __forceinline__ void flag_general(TipsetCtxDev& c, uint32_t slot) {
    if (threadIdx.x == 0) atomicOr(&c.prologue_general, 1ull << slot);
}

// slots 0 / 1: the child header / the first parent header (ctx_headers_body in verify_events.hip is the general form)
// (`in`: the context's inputs — its own head in device memory, or the kernel argument)
__device__ __forceinline__ void headers_slot(const WitnessView& w, const TipsetInputs& in, TipsetCtxDev& c, bool child_part,
                                             rd_chunk_t* stage, AmtRootSpec* receipts_spec, LiveIndex& li) {
    const bool lead = threadIdx.x == 0;
    const bool parsed = (in.flags & (TC_PARENTS_PARSED | TC_CHILD_PARSED)) == (TC_PARENTS_PARSED | TC_CHILD_PARSED);
    uint32_t status = IPCFP_ST_ERR_BAD_CLAIM, match = 0;
    long long height = 0;
    const bool wanted = parsed && (child_part || in.n_parents > 0);
    if (wanted) {
        const uint32_t hb = find_block(w, li, child_part ? in.child : in.parents[0]);  // uniform across the wavefront
        if (hb == kNoBlock) {
            status = IPCFP_ST_ERR_MISSING_BLOCK;
        } else {
            uint32_t len;
            if (!stage_block(w, hb, stage, len)) return flag_general(c, child_part ? 0u : 1u);
            if (lead) {
                Rd r;
                r.init((lds_bytes_t)stage, len);
                HeaderLite h;
                status = decode_header(r, h);
                if (status == IPCFP_ST_TRUE) {
                    height = h.height;
                    if (child_part) {
                        c.receipts_root = h.parent_message_receipts;
                        // `child_hdr.parents != parent_cids` (:161): same count, same CIDs in order
                        bool same = h.n_parents == in.n_parents;
                        if (same) {
                            Rd q = r;
                            q.err = 0;
                            q.pos = h.parents_off;
                            for (uint32_t i = 0; i < in.n_parents && same; ++i) {
                                CidKey k;
                                q.read_link_key(k);
                                same = q.ok() && cid_equal(k, in.parents[i]);
                            }
                        }
                        match = same ? 1u : 0u;
                    }
                }
            }
        }
    }
    if (!lead) return;
    if (child_part) {
        c.child_status = status;
        c.parents_match = match;
        c.child_height = height;
        if (receipts_spec) {  // the receipts AMT as an enumeration root (amt_enum.h EnumExtra)
            AmtRootSpec rs{};
            rs.version = 0;  // Amtv0<Receipt>
            rs.kind_p1 = uint32_t(VK_RECEIPT) + 1u;
            rs.skip = status == IPCFP_ST_TRUE ? 0u : 1u;
            if (!rs.skip) rs.root = c.receipts_root;
            *receipts_spec = rs;
        }
    } else {
        c.parent0_status = status;
        c.parent0_height = height;
    }
}

// slot 2 + b: parent block b → its header, its TxMeta (re-hashed), its two message-AMT roots
// (exec_roots_body in verify_events.hip is the general form; error sequence numbers as there)
__device__ __forceinline__ void roots_slot(const WitnessView& w, const TipsetInputs& in, TipsetCtxDev& c,
                                           AmtRootSpec* __restrict__ roots, unsigned long long* __restrict__ err, uint32_t b,
                                           rd_chunk_t* stage, LiveIndex& li, bool defer_rehash) {
    __shared__ CidKey s_tx;
    __shared__ uint32_t s_have_tx;
    const uint32_t P = in.n_parents;
    if (b >= P) return;
    const bool lead = threadIdx.x == 0;
    auto fail = [&](uint32_t seq, uint32_t code) { atomicMin(err, (unsigned long long)pack_enum_error(seq, 0, code)); };
    // reconstruct_execution_order (utils.rs:20-27): the parent header
    if (lead) s_have_tx = 0;
    const uint32_t hb = find_block(w, li, in.parents[b]);
    if (hb == kNoBlock) {
        if (lead) fail(b, IPCFP_ST_ERR_MISSING_BLOCK);
    } else {
        uint32_t len;
        if (!stage_block(w, hb, stage, len)) return flag_general(c, 2u + b);
        if (lead) {
            Rd hr;
            hr.init((lds_bytes_t)stage, len);
            HeaderLite h;
            const uint32_t st = decode_header(hr, h);
            if (st == IPCFP_ST_TRUE) {
                s_tx = h.messages;
                s_have_tx = 1;
            } else {
                fail(b, st);
            }
        }
    }
    __syncthreads();  // s_tx / s_have_tx; and the header's stage is free again
    // collect_exec_list (utils.rs:56-91)
    const uint32_t seq = P + 3 * b;
    AmtRootSpec bls{}, secp{};
    bls.version = secp.version = 0;
    bls.seq = seq + 1;
    secp.seq = seq + 2;
    bls.skip = secp.skip = 1;
    if (s_have_tx) {
        const CidKey tx = s_tx;
        const uint32_t tb = find_block(w, li, tx);  // :58-60
        if (tb == kNoBlock) {
            if (lead) fail(seq, IPCFP_ST_ERR_MISSING_BLOCK);
        } else {
            uint32_t len;
            if (!stage_block(w, tb, stage, len)) return flag_general(c, 2u + b);
            if (lead) {
                Rd r;
                r.init((lds_bytes_t)stage, len);
                uint32_t o0, l0, o1, l1;
                r.expect_array(2);  // (Cid, Cid)  :61
                r.read_link(o0, l0);
                r.read_link(o1, l1);
                r.finish();
                if (!r.ok()) {
                    fail(seq, IPCFP_ST_ERR_DECODE);
                } else if (defer_rehash) {
                    // The re-hash — re-encode, one Blake2b compression on ONE lane, ≈ 3 k instructions — gates nothing: it
                    // only ever adds an error.  A launch of its own on the aux stream does it (amt_enum.hip k_txmeta_rehash), joined
                    // at the end of the call; a mismatch reaches the same error word with the same sequence number.
                    c.txmeta_block[b] = tb + 1u;
                    bls.root = r.key_any(o0, l0);
                    secp.root = r.key_any(o1, l1);
                    bls.skip = secp.skip = 0;
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
                    if (!cid_equal(re, tx)) {  // (the verify path always checks: reconstruct_execution_order)
                        fail(seq, IPCFP_ST_ERR_TXMETA_MISMATCH);
                    } else {
                        bls.root = r.key_any(o0, l0);
                        secp.root = r.key_any(o1, l1);
                        bls.skip = secp.skip = 0;
                    }
                }
            }
        }
    }
    if (lead) {
        roots[2 * b] = bls;
        roots[2 * b + 1] = secp;
    }
}

// `live_done` non-null: the CID index is being filled beside this launch (LiveIndex above); `anomaly`: the call's flag.
// INLINE: the one context's inputs are the kernel ARGUMENT `in0` (scalar loads from the kernarg segment); its device copy
// — zeroed memory, not yet written by anyone — gets them from slot 0, for the kernels behind this launch.
template <bool INLINE>
__global__ __launch_bounds__(64) void k_tipset_prepare(WitnessView w, PrepareJobs jobs, uint32_t n_jobs, const uint32_t* live_done,
                                                       uint32_t live_total, uint32_t* anomaly, int defer_rehash, TipsetInputs in0) {
    __shared__ rd_chunk_t stage[kPrologueStageChunks];
    const uint32_t job = blockIdx.x / kPrepareSlots, slot = blockIdx.x % kPrepareSlots;
    if (job >= n_jobs) return;
    const PrepareJob jb = prepare_job(jobs, job);
    const TipsetInputs& in = INLINE ? in0 : *reinterpret_cast<const TipsetInputs*>(jb.ctx);
    if (INLINE && slot == 0) {  // 688 bytes, sixteen at a time
        static_assert(sizeof(TipsetInputs) % 8 == 0, "copied as 64-bit words");
        const uint64_t* src = reinterpret_cast<const uint64_t*>(&in0);
        uint64_t* dst = reinterpret_cast<uint64_t*>(jb.ctx);
        for (uint32_t i = threadIdx.x; i < sizeof(TipsetInputs) / 8; i += 64u) dst[i] = src[i];
    }
    LiveIndex li{};
    li.done = live_done;
    li.total = live_total;
    if (slot < 2) headers_slot(w, in, *jb.ctx, slot == 0, stage, jb.roots ? jb.roots + 2u * in.n_parents : nullptr, li);
    else if (jb.roots) roots_slot(w, in, *jb.ctx, jb.roots, jb.err, slot - 2, stage, li, defer_rehash != 0);
    live_validate(w, li, anomaly);
}

void launch_tipset_prepare_lds(hipStream_t stream, const WitnessView& w, const PrepareJobs& jobs, uint32_t n_jobs,
                               const uint32_t* live_done, uint32_t live_total, uint32_t* anomaly, bool defer_rehash,
                               const TipsetInputs* inline_inputs) {
    if (inline_inputs && n_jobs == 1)
        hipLaunchKernelGGL(k_tipset_prepare<true>, dim3(kPrepareSlots), dim3(64), 0, stream, w, jobs, n_jobs, live_done, live_total,
                           anomaly, defer_rehash ? 1 : 0, *inline_inputs);
    else
        hipLaunchKernelGGL(k_tipset_prepare<false>, dim3(n_jobs * kPrepareSlots), dim3(64), 0, stream, w, jobs, n_jobs, live_done,
                           live_total, anomaly, defer_rehash ? 1 : 0, TipsetInputs{});
}

}  // namespace ipcfp
