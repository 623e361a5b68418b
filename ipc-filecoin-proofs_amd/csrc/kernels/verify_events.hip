// csrc/kernels/verify_events.hip — batch `verify_event_proof`: tipset-context preparation,
// execution-order reconstruction, and one proof per lane.
//
// Replaces src/proofs/events/verifier.rs:51-290 and src/proofs/events/utils.rs:16-30,48-94.
// Check order and every Ok(false)/Err outcome follow SURVEY.md A.10; the status byte names the
// reference line that decided.
#define IPCFP_LINE_STAGE 1  // k_verify_events parses whole blocks front to back: see cbor_dev.h
#include <hip/hip_runtime.h>

#include "../common.h"
#include "blake2b_dev.h"
#include "claims_dev.h"
#include "events_dev.h"
#include "exec_order.h"
#include "launch.h"
#include "verify_dev.h"

namespace ipcfp {

// ---------------------------------------------------------------------------
// Blake2b-256 of a short byte buffer on one lane (the TxMeta re-hash, events/utils.rs:65)
// ---------------------------------------------------------------------------
__device__ __forceinline__ void blake2b256_small(const uint8_t* buf, uint32_t len, uint64_t out[4]) {
    uint64_t h[8];
    b2b::init256(h);
    uint32_t done = 0;
    uint64_t t = 0;
    for (;;) {
        const uint32_t left = len - done;
        const bool last = left <= 128;
        const uint32_t take = last ? left : 128;
        uint64_t m[16];
#pragma unroll
        for (int w = 0; w < 16; ++w) {
            uint64_t v = 0;
            for (int b = 0; b < 8; ++b) {
                const uint32_t idx = 8u * w + b;
                if (idx < take) v |= uint64_t(buf[done + idx]) << (8 * b);
            }
            m[w] = v;
        }
        t += take;
        b2b::compress<0>(h, m, t, last);
        done += take;
        if (last) break;
    }
    out[0] = h[0];
    out[1] = h[1];
    out[2] = h[2];
    out[3] = h[3];
}

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
                                same = q.ok() && cid_equal(k, c.parents[i]);
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
            const uint32_t pb = witness_find(w, c.parents[0]);
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
        const uint32_t hb = witness_find(w, ctx->parents[b]);
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
                        bls.root = lens[0] <= 40 ? r.key_at(o0, l0) : CidKey{{~0ULL, ~0ULL, ~0ULL, ~0ULL, ~0ULL}};
                        secp.root = lens[1] <= 40 ? r.key_at(o1, l1) : CidKey{{~0ULL, ~0ULL, ~0ULL, ~0ULL, ~0ULL}};
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

// The whole tipset prologue of the verify path in ONE launch: per context, two wavefronts decode the child and
// the first parent header (ctx_headers_body) and one wavefront per parent block decodes its header, its TxMeta
// and re-hashes it (exec_roots_body).  They are independent single-lane parses of tens of microseconds each;
// launched one after the other they were the longest idle stretch of a step.
struct PrepareJob {
    TipsetCtxDev* ctx;
    AmtRootSpec* roots;          // nullptr: no execution order for this context
    unsigned long long* err;
};
constexpr uint32_t kPrepareSlots = 2 + kMaxParents;

__global__ __launch_bounds__(64) void k_tipset_prepare(WitnessView w, const PrepareJob* __restrict__ jobs, uint32_t n_jobs) {
    __shared__ __attribute__((aligned(16))) uint8_t lds[kHeaderLds];
    const uint32_t job = blockIdx.x / kPrepareSlots, slot = blockIdx.x % kPrepareSlots;
    if (job >= n_jobs) return;
    const PrepareJob jb = jobs[job];
    if (slot < 2) ctx_headers_body(w, *jb.ctx, slot == 0, lds, jb.roots ? jb.roots + 2u * jb.ctx->n_parents : nullptr);
    else if (jb.roots) exec_roots_body(w, jb.ctx, jb.roots, jb.err, 1, slot - 2, lds);
}

// stage 2: leaf values (tag-42 links, already validated) → message CID keys
__global__ __launch_bounds__(256) void k_exec_keys(WitnessView w, const LeafRef* __restrict__ leaves, uint32_t n,
                                                   CidKey* __restrict__ keys) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const LeafRef l = leaves[t];
    Rd r;
    r.init(w.arena + w.off[l.block] + l.off, l.len);
    CidKey k;
    r.read_link_key(k);
    keys[t] = k;
}

// stage 3: `seen.insert(c)` — the table keeps, per distinct CID, the SMALLEST raw position.
// A slot is one u64: fingerprint (low half of the key's 64-bit hash) in the high word, raw position in the low
// word.  A probe compares fingerprints first and reads the 40-byte key behind a slot only when they agree — at load
// 0.5 half of all inserts pass an occupied slot, and each of those used to be a random 40-byte read.

__global__ __launch_bounds__(256) void k_exec_insert(const CidKey* __restrict__ keys, uint32_t n,
                                                     unsigned long long* __restrict__ slots, uint32_t mask) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const CidKey key = keys[i];
    const uint64_t h = cid_hash64(key);
    const unsigned long long mine = ((unsigned long long)uint32_t(h) << 32) | i;
    uint32_t s = uint32_t(h >> 32) & mask;
    for (;;) {
        unsigned long long cur = slots[s];
        if (cur == kEmptySlot64) {
            cur = atomicCAS(&slots[s], kEmptySlot64, mine);
            if (cur == kEmptySlot64) return;
        }
        if (uint32_t(cur >> 32) == uint32_t(h) && cid_equal(keys[uint32_t(cur)], key)) {
            atomicMin(&slots[s], mine);  // same fingerprint: the minimum is the smaller position
            return;
        }
        s = (s + 1) & mask;
    }
}

// stage 4: first[i] = 1 iff position i is the first occurrence of its CID (`if seen.insert(*c) { out.push(*c) }`)
__global__ __launch_bounds__(256) void k_exec_first(const CidKey* __restrict__ keys, uint32_t n,
                                                    const unsigned long long* __restrict__ slots, uint32_t mask,
                                                    uint32_t* __restrict__ first) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const CidKey key = keys[i];
    const uint64_t h = cid_hash64(key);
    uint32_t s = uint32_t(h >> 32) & mask;
    uint32_t f = 0;
    for (;;) {
        const unsigned long long cur = slots[s];
        if (cur == kEmptySlot64) break;  // cannot happen after stage 3; kept as a stop
        if (uint32_t(cur >> 32) == uint32_t(h)) {
            if (uint32_t(cur) == i) {  // the slot is this position's own: no key to read
                f = 1;
                break;
            }
            if (cid_equal(keys[uint32_t(cur)], key)) break;  // an earlier position holds the same CID
        }
        s = (s + 1) & mask;
    }
    first[i] = f;
}

// the distinct CIDs in execution order (ipcfp_exec_order)
__global__ __launch_bounds__(256) void k_exec_compact(const CidKey* __restrict__ keys, uint32_t n,
                                                      const uint32_t* __restrict__ first,
                                                      const uint32_t* __restrict__ pos, CidKey* __restrict__ out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (first[i]) out[pos[i]] = keys[i];
}

// ---------------------------------------------------------------------------
// one proof per lane (the checks every route shares live in verify_dev.h)
// ---------------------------------------------------------------------------
// `where` (nullable) receives the location of the StampedEvent the claim names once the proof has reached it
// (block = 0xffffffff otherwise): a host `check_event` closure runs over those bytes (events/verifier.rs:247-251).
__device__ __forceinline__ uint32_t verify_event_one(const WitnessView& w, const EventClaimPacked& c,
                                                     const TipsetCtxDev& tc, const uint8_t* __restrict__ blob,
                                                     const ipcfp_trust_policy_t& trust, const ipcfp_event_filter_t& filter,
                                                     bool has_filter, ValueLoc* where) {
    // Steps 1-3: trust anchors, header consistency, execution order (verify_dev.h)
    const uint32_t pre = verify_event_prefix(c, tc, trust);
    if (pre != IPCFP_ST_TRUE) return pre;
    // Step 4: verify_receipt_and_event (:207-254)
    uint32_t st;
    ValueLoc rloc;
    if (tc.receipt_leaves && c.exec_index >= tc.receipt_first && c.exec_index - tc.receipt_first < tc.n_receipt_leaves) {
        // the receipts AMT was enumerated (and thereby fully validated) for this context: load + get
        // of a present index cannot fail and yields exactly this leaf                                        // :220-224
        const uint64_t slot = c.exec_index - tc.receipt_first;
        if (tc.receipt_recs) {
            // ... and its events were tabulated (event_table.h): normally k_verify_events_table has settled the claim
            bool settled;
            const uint32_t ts = verify_event_from_table(w, c, tc, blob, filter, has_filter, where, settled);
            if (settled) return ts;
        }
        const LeafRef l = tc.receipt_leaves[slot];
        rloc = ValueLoc{l.block, l.off, l.len};
    } else {
        AmtRootInfo receipts;
        st = amt_load(w, tc.receipts_root, 0, VK_RECEIPT, receipts);                                          // :220
        if (st != IPCFP_ST_TRUE) return st;
        st = amt_get(w, receipts, VK_RECEIPT, c.exec_index, rloc);                                            // :224
        if (st == IPCFP_ST_NOT_FOUND) return IPCFP_ST_FALSE_NO_RECEIPT;
        if (st != IPCFP_ST_TRUE) return st;
    }
    Rd rr;
    rr.init(w.arena + w.off[rloc.block] + rloc.off, rloc.len);
    uint32_t o, l;
    rr.expect_array(4);
    (void)rr.read_uint();
    rr.read_bytes(o, l);
    (void)rr.read_uint();
    if (rr.at_null()) return IPCFP_ST_FALSE_NO_EVENTS_ROOT;                                                   // :229
    CidKey events_root;
    rr.read_link_key(events_root);
    ValueLoc eloc;
    st = amt_load_get(w, events_root, 3, VK_STAMPED_EVENT, c.event_index, eloc);                              // :234-237
    if (st == IPCFP_ST_NOT_FOUND) return IPCFP_ST_FALSE_NO_EVENT;
    if (st != IPCFP_ST_TRUE) return st;
    if (where) *where = eloc;
    // verify_event_data_matches (:257-290)
    Rd er;
    er.init(w.arena + w.off[eloc.block] + eloc.off, eloc.len);
    uint64_t emitter;
    EvmLogLoc log;
    decode_event_log(er, emitter, log);
    if (emitter != c.emitter) return IPCFP_ST_FALSE_EMITTER;                                                  // :262
    if (!log.is_log) return IPCFP_ST_FALSE_NOT_EVM_LOG;                                                       // :267
    if (log.n_topics != c.n_topics) return IPCFP_ST_FALSE_TOPIC_COUNT;                                        // :272
    for (uint32_t i = 0; i < log.n_topics; ++i) {                                                             // :276-281
        const uint8_t* claimed = blob + c.topics_off + 33u * i;
        if (!claimed[0]) return IPCFP_ST_FALSE_TOPIC;  // the claimed string is not "0x" + 64 hex digits
        if (!er.equal32(log.topic_at(i), claimed + 1)) return IPCFP_ST_FALSE_TOPIC;
    }
    if (!(c.flags & EC_DATA_MATCHABLE) || c.data_len != log.data.len) return IPCFP_ST_FALSE_DATA;             // :284-287
    if (!er.equal_bytes(log.data.off, blob + c.data_off, c.data_len)) return IPCFP_ST_FALSE_DATA;
    if (has_filter && !log_matches(er, log, filter)) return IPCFP_ST_FALSE_FILTER;                             // :247-251
    return IPCFP_ST_TRUE;
}

__global__ __launch_bounds__(256, IPCFP_WALK_WAVES) void k_verify_events(WitnessView w, const EventClaimPacked* __restrict__ claims,
                                                       uint32_t n, const TipsetCtxDev* __restrict__ ctxs, uint32_t n_ctxs,
                                                       const uint8_t* __restrict__ blob, uint64_t blob_len,
                                                       ipcfp_trust_policy_t trust,
                                                       ipcfp_event_filter_t filter, int has_filter,
                                                       uint8_t* __restrict__ status, ValueLoc* __restrict__ where,
                                                       int pending_only) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    if (pending_only && status[t] != kStPending) return;  // settled from the event table (verify_table.hip)
    const EventClaimPacked& c = claims[t];
    ValueLoc loc{kNoBlock, 0, 0};
    uint32_t st = IPCFP_ST_ERR_BAD_CLAIM;
    if (claim_in_bounds(c, n_ctxs, blob_len))
        st = verify_event_one(w, c, ctxs[c.context], blob, trust, filter, has_filter != 0, where ? &loc : nullptr);
    status[t] = uint8_t(st);
    if (where) where[t] = loc;
}

// exec_len of a context = the total of the first-occurrence scan, copied on the device so the host
// never waits for it
// ... and the inverse of exec_pos: inv[execution index] = raw position of the message's first occurrence.  A claim
// names (exec_index, message): `exec_keys[inv[exec_index]] == message` settles `position(message) == exec_index`
// with two reads that run along with the claims instead of a hash probe (three reads somewhere in 60 MB).
__global__ __launch_bounds__(256) void k_exec_finish(TipsetCtxDev* __restrict__ c, const uint64_t* __restrict__ total,
                                                     const uint32_t* __restrict__ first, const uint32_t* __restrict__ pos,
                                                     uint32_t n, uint32_t* __restrict__ inv) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) c->exec_len = *total;
    if (i < n && first[i]) inv[pos[i]] = i;
}

// ------------------------------ launchers -----------------------------------
int launch_tipset_prepare(ipcfp_ctx* ctx, const WitnessView& w, const void* jobs_d, uint32_t n_jobs) {
    if (n_jobs == 0) return IPCFP_OK;
    hipLaunchKernelGGL(k_tipset_prepare, dim3(n_jobs * kPrepareSlots), dim3(64), 0, ctx->stream, w,
                       static_cast<const PrepareJob*>(jobs_d), n_jobs);
    IPCFP_HIP(ctx, hipGetLastError());
    return IPCFP_OK;
}

int launch_exec_finish(ipcfp_ctx* ctx, TipsetCtxDev* ctx_d, const uint64_t* total_d, const uint32_t* first_d,
                       const uint32_t* pos_d, uint32_t n, uint32_t* inv_d) {
    hipLaunchKernelGGL(k_exec_finish, dim3(n ? div_up(n, 256) : 1), dim3(256), 0, ctx->stream, ctx_d, total_d, first_d, pos_d,
                       n, inv_d);
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
                      unsigned long long* err_d, int verify_txmeta) {
    const unsigned long long none = kNoEnumError;
    IPCFP_HIP(ctx, hipMemsetAsync(err_d, 0xff, 8, ctx->stream));  // kNoEnumError
    (void)none;
    hipLaunchKernelGGL(k_exec_roots, dim3(IPCFP_MAX_PARENTS), dim3(64), 0, ctx->stream, w, ctx_d, roots_d, err_d,
                       verify_txmeta);
    IPCFP_HIP(ctx, hipGetLastError());
    return IPCFP_OK;
}

int launch_exec_dedup(ipcfp_ctx* ctx, const WitnessView& w, const LeafRef* leaves_d, uint32_t n, CidKey* keys_d,
                      unsigned long long* slots_d, uint32_t mask, uint32_t* first_d) {
    if (n == 0) return IPCFP_OK;
    const dim3 g(div_up(n, 256)), b(256);
    if (leaves_d) hipLaunchKernelGGL(k_exec_keys, g, b, 0, ctx->stream, w, leaves_d, n, keys_d);  // else: keys came with the enumeration
    hipLaunchKernelGGL(k_exec_insert, g, b, 0, ctx->stream, keys_d, n, slots_d, mask);
    hipLaunchKernelGGL(k_exec_first, g, b, 0, ctx->stream, keys_d, n, slots_d, mask, first_d);
    IPCFP_HIP(ctx, hipGetLastError());
    return IPCFP_OK;
}

int launch_exec_compact(ipcfp_ctx* ctx, const CidKey* keys_d, uint32_t n, const uint32_t* first_d,
                        const uint32_t* pos_d, CidKey* out_d) {
    if (n == 0) return IPCFP_OK;
    hipLaunchKernelGGL(k_exec_compact, dim3(div_up(n, 256)), dim3(256), 0, ctx->stream, keys_d, n, first_d, pos_d,
                       out_d);
    IPCFP_HIP(ctx, hipGetLastError());
    return IPCFP_OK;
}

int launch_verify_events(ipcfp_ctx* ctx, const WitnessView& w, const EventClaimPacked* claims_d, uint32_t n,
                         const TipsetCtxDev* ctxs_d, uint32_t n_ctxs, const uint8_t* blob_d, uint64_t blob_len,
                         const ipcfp_trust_policy_t& trust, const ipcfp_event_filter_t* filter, uint8_t* status_d,
                         void* where_d, bool tabulated) {
    if (n == 0) return IPCFP_OK;
    ipcfp_event_filter_t f{};
    if (filter) f = *filter;
    {
        ProfileScope prof(ctx, IPCFP_K_EVENT_VERIFY);
        // claims whose receipt's events are tabulated are settled by a kernel that parses nothing; the general
        // walker then only serves what that one left pending
        if (tabulated) {
            int rc = launch_verify_events_table(ctx, w, claims_d, n, ctxs_d, n_ctxs, blob_d, blob_len, trust, f, filter ? 1 : 0,
                                                status_d, where_d);
            if (rc) return rc;
        }
        hipLaunchKernelGGL(k_verify_events, dim3(div_up(n, 256)), dim3(256), 0, ctx->stream, w, claims_d, n, ctxs_d,
                           n_ctxs, blob_d, blob_len, trust, f, filter ? 1 : 0, status_d, static_cast<ValueLoc*>(where_d),
                           tabulated ? 1 : 0);
    }
    IPCFP_HIP(ctx, hipGetLastError());
    return IPCFP_OK;
}

}  // namespace ipcfp
