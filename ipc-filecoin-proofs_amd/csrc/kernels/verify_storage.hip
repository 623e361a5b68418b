// csrc/kernels/verify_storage.hip — batch `verify_storage_proof`, one proof per lane.
//
// Replaces src/proofs/storage/verifier.rs:24-63 (steps 2-6; step 1, the per-proof rebuild of the
// witness store, src/proofs/verifier.rs:19-28, is gone: the witness is resident and indexed once)
// with read_storage_slot (src/proofs/storage/decode.rs:36-97), get_actor_state / parse_evm_state
// (src/proofs/common/decode.rs:17-42,79-97) and left_pad_32 (src/proofs/common/evm.rs:91-100).
// Check order and Ok(false)/Err outcomes follow SURVEY.md A.10 exactly; the status byte says
// which line decided.
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "../common.h"
#include "claims_dev.h"
#include "launch.h"
#include "storage_dev.h"
#include "storage_runs.h"

#ifndef IPCFP_VS_STAGE
#define IPCFP_VS_STAGE 1
#endif
#ifndef IPCFP_VS_PRELOAD
#define IPCFP_VS_PRELOAD 0  // (the slot and the value in registers from the start: 27-37 µs slower — spills)
#endif

namespace ipcfp {

__device__ __forceinline__ bool trusted(const ipcfp_trust_policy_t& t, long long epoch) {
    if (t.kind == 0) return true;                       // AcceptAll
    if (t.ec_chain_empty) return false;                 // cert.rs:55-57
    return epoch >= t.min_epoch && epoch <= t.max_epoch;  // cert.rs:60-63
}

__device__ __forceinline__ uint32_t verify_storage_one(const WitnessView& w, const StorageClaimPacked& c,
                                                       const ipcfp_trust_policy_t& trust) {
    // Step 2: verify_trust_anchor (storage/verifier.rs:81-92)
    if (!(c.flags & SC_CHILD_PARSED)) return IPCFP_ST_ERR_BAD_CLAIM;             // :85
    if (!trusted(trust, c.child_epoch)) return IPCFP_ST_FALSE_UNTRUSTED_CHILD;    // :87
    // Step 3: verify_parent_state_root (:95-111)
    HeaderLite hdr;
    uint32_t hb;
    uint32_t st = load_header(w, c.child, hdr, hb);                               // :101-107
    if (st != IPCFP_ST_TRUE) return st;
    if (!((c.flags & SC_STATE_ROOT_CANON) && cid_equal(hdr.parent_state_root, c.state_root)))
        return IPCFP_ST_FALSE_STATE_ROOT;                                         // :110 string compare
    // Step 4: verify_actor_state (:114-127)
    CidKey actor_state;
    st = get_actor_state(w, c.state_root, c.actor_id, actor_state);               // :122
    if (st != IPCFP_ST_TRUE) return st;
    if (!((c.flags & SC_ACTOR_STATE_CANON) && cid_equal(actor_state, c.actor_state)))
        return IPCFP_ST_FALSE_ACTOR_STATE;                                        // :126
    // Step 5: verify_storage_root (:130-145)
    const uint32_t eb = witness_find(w, c.actor_state);                           // :136-138
    if (eb == kNoBlock) return IPCFP_ST_ERR_MISSING_BLOCK;
    CidKey contract_state;
    st = parse_evm_state(w, eb, contract_state);                                  // :141
    if (st != IPCFP_ST_TRUE) return st;
    if (!((c.flags & SC_STORAGE_ROOT_CANON) && cid_equal(contract_state, c.storage_root)))
        return IPCFP_ST_FALSE_STORAGE_ROOT;                                       // :144
    // Step 6: verify_storage_value (:148-170)
    if (!(c.flags & SC_SLOT_PARSED)) return IPCFP_ST_ERR_BAD_CLAIM;              // :155-157
    uint8_t padded[32];
    st = read_storage_slot_padded(w, c.storage_root, c.slot, padded);             // :160-165
    if (st != IPCFP_ST_TRUE) return st;
    if (!(c.flags & SC_VALUE_MATCHABLE)) return IPCFP_ST_FALSE_VALUE;            // claimed string can never equal "0x"+64 hex
    bool eq = true;
    for (int i = 0; i < 32; ++i) eq &= padded[i] == c.value[i];
    return eq ? IPCFP_ST_TRUE : IPCFP_ST_FALSE_VALUE;                             // :169
}

// WAVES = wavefronts per SIMD the register allocator must leave room for.  verify_storage_one inlines six
// layout attempts, each a Keccak + SHA-256 + HAMT walk: at 4 waves (128 VGPRs) it spills 384 bytes per
// lane to scratch, at 3 waves (168 VGPRs) it does not.  IPCFP_STORAGE_WAVES selects (default: see launch).
template <int WAVES>
__global__ __launch_bounds__(256, WAVES) void k_verify_storage(WitnessView w, const StorageClaimPacked* __restrict__ claims,
                                                               uint32_t n, ipcfp_trust_policy_t trust,
                                                               uint8_t* __restrict__ status, int pending_only) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    if (pending_only && status[t] != kStPending) return;  // settled from the node table (k_verify_storage_table)
    status[t] = uint8_t(verify_storage_one(w, claims[t], trust));
}

// ---------------------------------------------------------------------------------------------------------------------
// Runs of claims (storage_runs.h): the one-lane half — run boundaries and, per run, every fact that is a plain typed
// decode (child header, StateRoot, EVM state, the storage root's layout).  The HAMT walks go over the node table (hamt_table.h).
// ---------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ bool same_run(const StorageClaimPacked& a, const StorageClaimPacked& b) {
    return a.actor_id == b.actor_id && cid_equal(a.child, b.child) && cid_equal(a.state_root, b.state_root) &&
           cid_equal(a.actor_state, b.actor_state) && cid_equal(a.storage_root, b.storage_root);
}

__global__ __launch_bounds__(256) void k_storage_run_flags(const StorageClaimPacked* __restrict__ claims, uint32_t n,
                                                           uint32_t* __restrict__ flag) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    flag[t] = (t == 0 || !same_run(claims[t], claims[t - 1])) ? 1u : 0u;
}

// run_of[t] = index of claim t's run; the run's record gets its first claim
__global__ __launch_bounds__(256) void k_storage_run_heads(const uint32_t* __restrict__ flag, const uint32_t* __restrict__ pos,
                                                           uint32_t n, uint32_t* __restrict__ run_of, StorageRun* __restrict__ runs) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const uint32_t r = pos[t] + flag[t] - 1u;  // pos: exclusive sum of flag
    run_of[t] = r;
    if (flag[t]) runs[r].first_claim = t;
}

__global__ __launch_bounds__(256, IPCFP_WALK_WAVES) void k_storage_run_facts(WitnessView w, const StorageClaimPacked* __restrict__ claims,
                                                                             StorageRun* __restrict__ runs, uint32_t n_runs) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_runs) return;
    StorageRun run = runs[i];
    const StorageClaimPacked& c = claims[run.first_claim];
    // verify_parent_state_root (storage/verifier.rs:95-111): the child header
    HeaderLite hdr;
    uint32_t hb;
    run.hdr_status = load_header(w, c.child, hdr, hb);
    run.parent_state_root = hdr.parent_state_root;
    // get_actor_state, first half (common/decode.rs:23-26): StateRoot [version, actors, info] behind the CLAIMED state root
    {
        run.sr_status = IPCFP_ST_TRUE;
        const uint32_t b = witness_find(w, c.state_root);
        if (b == kNoBlock) {
            run.sr_status = IPCFP_ST_ERR_MISSING_BLOCK;
        } else {
            Rd r = open_block(w, b);
            CidKey info;
            r.expect_array(3);
            if (r.read_uint() > 5) r.fail();
            r.read_link_key(run.actors);
            r.read_link_key(info);
            r.finish();
            if (!r.ok()) run.sr_status = IPCFP_ST_ERR_DECODE;
        }
    }
    run.actor_status = IPCFP_ST_ERR;  // (k_storage_run_actors_table / _lane)
    // verify_storage_root (storage/verifier.rs:130-145): the EVM state behind the CLAIMED actor-state CID
    {
        const uint32_t eb = witness_find(w, c.actor_state);
        run.evm_status = eb == kNoBlock ? uint32_t(IPCFP_ST_ERR_MISSING_BLOCK) : parse_evm_state(w, eb, run.contract_state);
    }
    // read_storage_slot's layout sniff of the CLAIMED storage root (storage/decode.rs:46-96)
    run.root_kind = sniff_storage_root(w, c.storage_root, run.hamt_root, run.hamt_bw);
    runs[i] = run;
}

// get_actor_state's HAMT half for the runs k_storage_run_actors_table left undecided (a block the table does not cover)
__global__ __launch_bounds__(256, IPCFP_WALK_WAVES) void k_storage_run_actors_lane(WitnessView w, const StorageClaimPacked* __restrict__ claims,
                                                                                   StorageRun* __restrict__ runs, uint32_t n_runs,
                                                                                   uint32_t undecided) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_runs) return;
    if (runs[i].sr_status != IPCFP_ST_TRUE || runs[i].actor_status != undecided) return;
    uint8_t key[12];
    const uint32_t kl = id_address_bytes(claims[runs[i].first_claim].actor_id, key);  // common/decode.rs:34
    ValueLoc loc;
    uint32_t st = hamt_get(w, runs[i].actors, 5, VK_ACTOR_STATE, key, kl, loc);         // decode.rs:29-37
    CidKey actor_state{};
    if (st == IPCFP_ST_NOT_FOUND) st = IPCFP_ST_ERR_ACTOR_NOT_FOUND;                    // decode.rs:39
    if (st == IPCFP_ST_TRUE) {
        Rd v;
        v.init(w.arena + w.off[loc.block] + loc.off, loc.len);
        CidKey code;
        v.expect_array(5);
        v.read_link_key(code);
        v.read_link_key(actor_state);
        if (!v.ok()) st = IPCFP_ST_ERR_DECODE;
    }
    runs[i].actor_status = st;
    runs[i].actor_state = actor_state;
}

int launch_storage_run_actors_lane(ipcfp_ctx* ctx, const WitnessView& w, const void* claims_d, void* runs_d, uint32_t n_runs,
                                   uint32_t undecided) {
    if (n_runs == 0) return IPCFP_OK;
    hipLaunchKernelGGL(k_storage_run_actors_lane, dim3(div_up(n_runs, 256)), dim3(256), 0, ctx->stream, w,
                       static_cast<const StorageClaimPacked*>(claims_d), static_cast<StorageRun*>(runs_d), n_runs, undecided);
    IPCFP_HIP(ctx, hipGetLastError());
    return IPCFP_OK;
}

// get_actor_state's HAMT half over the node table (hamt_table.h), one run per lane; what the table does not cover stays
// `undecided` for k_storage_run_actors_lane
__global__ __launch_bounds__(256) void k_storage_run_actors_table(WitnessView w, const HamtNodeRec* __restrict__ table,
                                                                  const StorageClaimPacked* __restrict__ claims,
                                                                  StorageRun* __restrict__ runs, uint32_t n_runs, uint32_t undecided) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_runs) return;
    if (runs[i].sr_status != IPCFP_ST_TRUE) return;
    uint8_t key[12];
    const uint32_t kl = id_address_bytes(claims[runs[i].first_claim].actor_id, key);  // common/decode.rs:34
    ValueLoc loc;
    uint32_t st = table_hamt_get(w, table, runs[i].actors, 5, HK_ACTOR_STATE, key, kl, loc);  // decode.rs:29-37
    CidKey actor_state{};
    if (st == kTablePunt) st = undecided;
    if (st == IPCFP_ST_NOT_FOUND) st = IPCFP_ST_ERR_ACTOR_NOT_FOUND;                          // decode.rs:39
    if (st == IPCFP_ST_TRUE) {
        Rd v;
        v.init(w.arena + w.off[loc.block] + loc.off, loc.len);
        CidKey code;
        v.expect_array(5);
        v.read_link_key(code);
        v.read_link_key(actor_state);
        if (!v.ok()) st = IPCFP_ST_ERR_DECODE;
    }
    runs[i].actor_status = st;
    runs[i].actor_state = actor_state;
}

// The first step of every storage get of a run, taken ONCE for the run (round 6): the storage root's block and the blocks
// behind its 32 links.  A contract's 256 proofs all open the same root and follow one of its links: each claim read the
// link's 38 bytes, probed the index and compared a CID — three dependent random reads of its ≈ eight — for an answer its
// run's neighbours had found already.  32 lanes per run; kNoBlock where the table has no record of the root, the pointer is
// not a standard link or the block is missing (the claim's own lane then takes the long way and gives the status).
__global__ __launch_bounds__(256) void k_storage_run_children(WitnessView w, const HamtNodeRec* __restrict__ table,
                                                              const StorageRun* __restrict__ runs, uint32_t n_runs,
                                                              const StorageClaimPacked* __restrict__ claims,
                                                              uint32_t* __restrict__ root_block, uint32_t* __restrict__ run_match,
                                                              uint32_t* __restrict__ root_child) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x, i = t >> 5, p = t & 31u;
    if (i >= n_runs) return;
    if (p == 1) {
        // The three CID comparisons of steps 3-5 (storage/verifier.rs:110, :126, :144) are the same for every claim of a run — a
        // run IS "same child, state root, actor, actor state, storage root" (same_run) — so they are made here, on the run's
        // first claim; each claim still answers for its own strings' canonical form (flags).
        const StorageClaimPacked& c = claims[runs[i].first_claim];
        run_match[i] = (cid_equal(runs[i].parent_state_root, c.state_root) ? 1u : 0u) | (cid_equal(runs[i].actor_state, c.actor_state) ? 2u : 0u) |
                       (cid_equal(runs[i].contract_state, c.storage_root) ? 4u : 0u);
    }
    uint32_t rb = kNoBlock, child = kNoBlock;
    if (runs[i].root_kind == 3) {
        rb = witness_find(w, runs[i].hamt_root);  // (the 32 lanes of a run: the same probe, broadcast)
        if (rb != kNoBlock) {
            const HamtNodeRec* rec = table + rb;
            const uint32_t head = *reinterpret_cast<const uint32_t*>(rec);
            if ((head & 0xffu) != 1u) {
                rb = kNoBlock;  // not tabulated: nothing is known about its pointers
            } else if (p < ((head >> 16) & 0xffu) && ((rec->std_links >> p) & 1u)) {
                const uint8_t* g = w.arena + w.off[rb] + rec->ptr_off[p];
                CidKey link;
#pragma unroll
                for (int j = 0; j < 5; ++j) __builtin_memcpy(&link.w[j], g + 5 + 8 * j, 8);
                link.w[4] &= (1ull << 48) - 1ull;
                child = witness_find(w, link);
            }
        }
    }
    root_child[size_t(i) * 32u + p] = child;
    if (p == 0) root_block[i] = rb;
}

// left_pad_32 (src/proofs/common/evm.rs:91-100) of a serde Vec<u8> (a CBOR array of u8, type-checked by the table) as four
// little-endian words: byte i of the padded value = word i/8, bits 8·(i%8)…
__device__ __forceinline__ void left_pad_32_words(Rd& v, uint64_t out[4]) {
    out[0] = out[1] = out[2] = out[3] = 0;
    const uint64_t n = v.read_array();
    auto put = [&](uint64_t i, uint32_t x) {
        if (n >= 32 && i < n - 32) return;
        const uint32_t j = n >= 32 ? uint32_t(i - (n - 32)) : uint32_t(32 - n + i);
        const uint64_t b = uint64_t(x & 0xffu) << (8u * (j & 7u));
        const uint32_t k = j >> 3;
        out[0] |= k == 0 ? b : 0ull;
        out[1] |= k == 1 ? b : 0ull;
        out[2] |= k == 2 ? b : 0ull;
        out[3] |= k == 3 ? b : 0ull;
    };
    uint64_t i = 0;
    while (i < n && v.ok() && v.pos + 8u <= v.n) {  // eight bytes per fetch (cbor_dev.h vec_u8_step)
        const uint64_t w = v.peek64(v.pos);
        uint32_t used = 0, x;
        while (i < n && vec_u8_step(w, used, x)) put(i++, x);
        v.pos += used;
        if (i < n && used <= 6u) put(i++, uint32_t(v.read_uint()));
    }
    for (; i < n && v.ok(); ++i) put(i, uint32_t(v.read_uint()));
}

// The same from plain 8-byte reads, for the usual spelling (`8n` | `98 nn`, elements of one or two bytes): the padded value is
// what a 32-byte shift register holds after every element has been pushed in at its low end — the last 32 elements, zeros
// above a shorter one — as eight big-endian limbs, L[0] the lowest.  false: another spelling, take the reader.
__device__ __forceinline__ bool left_pad_32_raw(const uint8_t* __restrict__ p, uint32_t avail, uint32_t L[8]) {
#pragma unroll
    for (int m = 0; m < 8; ++m) L[m] = 0;
    const uint32_t hv = p[0];
    uint32_t n, pos;
    if (hv >= 0x80u && hv < 0x98u) {
        n = hv - 0x80u;
        pos = 1u;
    } else if (hv == 0x98u) {
        n = p[1];
        pos = 2u;
    } else {
        return false;
    }
    uint32_t bad = 0;
    uint32_t q = n >> 2;
    for (; q && pos < avail; --q) {  // four elements per 8-byte read: one limb
        const uint64_t w8 = raw_ld64(p + pos);
        uint32_t cur = 0;
        vec_u8_take<4>(uint32_t(w8), uint32_t(w8 >> 32), cur, pos, bad);
#pragma unroll
        for (int m = 7; m > 0; --m) L[m] = L[m - 1];
        L[0] = cur;
    }
    if (q) return false;
    const uint32_t r = n & 3u;
    if (r) {  // the last one to three: the register moves up by as many bytes
        const uint64_t w8 = raw_ld64(p + pos);
        uint32_t cur = 0;
        if (r == 1) vec_u8_take<1>(uint32_t(w8), uint32_t(w8 >> 32), cur, pos, bad);
        else if (r == 2) vec_u8_take<2>(uint32_t(w8), uint32_t(w8 >> 32), cur, pos, bad);
        else vec_u8_take<3>(uint32_t(w8), uint32_t(w8 >> 32), cur, pos, bad);
        const uint32_t down = 32u - 8u * r;
#pragma unroll
        for (int m = 7; m > 0; --m) L[m] = __builtin_amdgcn_alignbit(L[m], L[m - 1], down);  // (L[m] << 8r) | (L[m-1] >> (32 - 8r))
        L[0] = (L[0] << (8u * r)) | cur;
    }
    if (bad || pos > avail) return false;
    return true;
}

// … and out of the lane's LDS slot, where the kernel has put the 72 bytes from the value's first byte on in ONE burst of
// loads: the decode's fetches depend on each other (an element is one or two bytes), and on global memory each was a round
// trip through an L2 that the kernel's own streaming turns over every few microseconds — lines came from memory two and
// three times (FETCH_SIZE 3.2 GB for 1 GB of claims and witness; profiles/r06_experiments.md).  A value longer than the
// stage holds (more than 32 two-byte elements) returns false like any unusual spelling.
constexpr uint32_t kValueStageWords = 10;
struct ValueStage {
    uint64_t w[kValueStageWords + 1][256];  // [word][lane]: a wavefront reading the same word index is conflict-free
};
__device__ __forceinline__ uint64_t stage_ld64(const ValueStage& vs, uint32_t lane, uint32_t at) {
    const uint32_t k = at >> 3, sh = (at & 7u) * 8u;
    const uint64_t lo = vs.w[k][lane], hi = vs.w[k + 1u][lane];
    return (lo >> sh) | ((hi << 1) << (63u - sh));
}
__device__ __forceinline__ bool left_pad_32_staged(const ValueStage& vs, uint32_t lane, uint32_t avail, uint32_t L[8]) {
#pragma unroll
    for (int m = 0; m < 8; ++m) L[m] = 0;
    const uint64_t h8 = vs.w[0][lane];
    const uint32_t hv = uint32_t(h8) & 0xffu;
    uint32_t n, pos;
    if (hv >= 0x80u && hv < 0x98u) {
        n = hv - 0x80u;
        pos = 1u;
    } else if (hv == 0x98u) {
        n = uint32_t(h8 >> 8) & 0xffu;
        pos = 2u;
    } else {
        return false;
    }
    if (n > 32u) return false;  // (≤ 2 + 64 bytes: inside the stage)
    uint32_t bad = 0;
    for (uint32_t q = n >> 2; q; --q) {  // four elements per 8 bytes: one limb
        const uint64_t w8 = stage_ld64(vs, lane, pos);
        uint32_t cur = 0;
        vec_u8_take<4>(uint32_t(w8), uint32_t(w8 >> 32), cur, pos, bad);
#pragma unroll
        for (int m = 7; m > 0; --m) L[m] = L[m - 1];
        L[0] = cur;
    }
    const uint32_t r = n & 3u;
    if (r) {  // the last one to three: the register moves up by as many bytes
        const uint64_t w8 = stage_ld64(vs, lane, pos);
        uint32_t cur = 0;
        if (r == 1) vec_u8_take<1>(uint32_t(w8), uint32_t(w8 >> 32), cur, pos, bad);
        else if (r == 2) vec_u8_take<2>(uint32_t(w8), uint32_t(w8 >> 32), cur, pos, bad);
        else vec_u8_take<3>(uint32_t(w8), uint32_t(w8 >> 32), cur, pos, bad);
        const uint32_t down = 32u - 8u * r;
#pragma unroll
        for (int m = 7; m > 0; --m) L[m] = __builtin_amdgcn_alignbit(L[m], L[m - 1], down);
        L[0] = (L[0] << (8u * r)) | cur;
    }
    return !(bad || pos > avail);
}

// verify_storage_proof, steps 2-6 in the reference's order of checks (src/proofs/storage/verifier.rs:24-63), one claim per
// lane: everything its run shares comes from the run's record, `read_storage_slot`'s HAMT get (storage/decode.rs:79-96)
// walks the node table.  A claim it cannot settle (an inline small-map layout, a block the table does not cover) is left
// kStPending for k_verify_storage.
// (Seven wavefronts per SIMD asked of the allocator: 72 VGPRs and 48 more bytes of scratch against its own 83 and five.  The
// kernel is a long instruction stream with 116 loads per wavefront that depend on each other; more wavefronts in flight
// were 2.13 → 2.02 ms for configs[4]'s call — 6: 2.05, 8: 2.03; profiles/r06_experiments.md.)
__global__ __launch_bounds__(256, 7) void k_verify_storage_table(WitnessView w, const HamtNodeRec* __restrict__ table,
                                                              const StorageClaimPacked* __restrict__ claims, uint32_t n,
                                                              const uint32_t* __restrict__ run_of, const StorageRun* __restrict__ runs,
                                                              const uint32_t* __restrict__ root_block, const uint32_t* __restrict__ run_match,
                                                              const uint32_t* __restrict__ root_child, ipcfp_trust_policy_t trust, uint32_t undecided, uint8_t* __restrict__ status) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
#if IPCFP_VS_STAGE
    __shared__ ValueStage vstage;
#endif
    const StorageClaimPacked& c = claims[t];
    const uint32_t ri = run_of[t];
    const StorageRun& run = runs[ri];
    const uint32_t flags = c.flags;
    // everything of the claim that is used late — the slot (hash, bucket compare) and the value (the last compare) — now, with
    // the lines that hold the CIDs: by the time the walk is done the L2 has long dropped them
    uint64_t kw[4], cv[4];
#if IPCFP_VS_PRELOAD
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        kw[j] = reinterpret_cast<const uint64_t*>(c.slot)[j];
        cv[j] = reinterpret_cast<const uint64_t*>(c.value)[j];
    }
#endif
    uint32_t st = kStPending;
    const uint32_t match = run_match ? run_match[ri] : 8u;  // (8: nobody has compared for the run)
    do {
        // Step 2: verify_trust_anchor (storage/verifier.rs:81-92)
        if (!(flags & SC_CHILD_PARSED)) { st = IPCFP_ST_ERR_BAD_CLAIM; break; }                           // :85
        if (!trusted(trust, c.child_epoch)) { st = IPCFP_ST_FALSE_UNTRUSTED_CHILD; break; }               // :87
        // Step 3: verify_parent_state_root (:95-111)
        if (run.hdr_status != IPCFP_ST_TRUE) { st = run.hdr_status; break; }                              // :101-107
        if (!((flags & SC_STATE_ROOT_CANON) && (match & 8u ? cid_equal(run.parent_state_root, c.state_root) : (match & 1u) != 0u))) { st = IPCFP_ST_FALSE_STATE_ROOT; break; }  // :110
        // Step 4: verify_actor_state (:114-127)
        if (run.sr_status != IPCFP_ST_TRUE) { st = run.sr_status; break; }                                // decode.rs:23-26
        if (run.actor_status == undecided) break;                                                         // (pending)
        if (run.actor_status != IPCFP_ST_TRUE) { st = run.actor_status; break; }                          // :122
        if (!((flags & SC_ACTOR_STATE_CANON) && (match & 8u ? cid_equal(run.actor_state, c.actor_state) : (match & 2u) != 0u))) { st = IPCFP_ST_FALSE_ACTOR_STATE; break; }  // :126
        // Step 5: verify_storage_root (:130-145)
        if (run.evm_status != IPCFP_ST_TRUE) { st = run.evm_status; break; }                              // :136-141
        if (!((flags & SC_STORAGE_ROOT_CANON) && (match & 8u ? cid_equal(run.contract_state, c.storage_root) : (match & 4u) != 0u))) { st = IPCFP_ST_FALSE_STORAGE_ROOT; break; }  // :144
        // Step 6: verify_storage_value (:148-170)
        if (!(flags & SC_SLOT_PARSED)) { st = IPCFP_ST_ERR_BAD_CLAIM; break; }                            // :155-157
        if (run.root_kind == 4) { st = IPCFP_ST_ERR_MISSING_BLOCK; break; }                               // decode.rs:41-43
        if (run.root_kind != 3) break;  // an inline small map (A1-A3): the one-lane kernel searches it
        uint64_t padded[4] = {0, 0, 0, 0};
        ValueLoc loc;
#if !IPCFP_VS_PRELOAD
        {
            const Raw16 a = raw_ld128(c.slot), b = raw_ld128(c.slot + 16);
            kw[0] = a.lo, kw[1] = a.hi, kw[2] = b.lo, kw[3] = b.hi;
        }
#endif
        const uint32_t hs = table_hamt_get(w, table, run.hamt_root, run.hamt_bw, HK_VEC_U8, c.slot, 32, loc,  // decode.rs:79-96
                                           root_block ? root_block[ri] : kNoBlock, root_child ? root_child + size_t(ri) * 32u : nullptr, kw);
        if (hs == kTablePunt) break;
        if (hs != IPCFP_ST_NOT_FOUND) {  // unwrap_or_default(): a missing key means zero
            if (hs != IPCFP_ST_TRUE) { st = hs; break; }
            const uint8_t* vp = w.arena + w.off[loc.block] + loc.off;
            uint32_t L[8];
#if IPCFP_VS_STAGE
            {
                Raw16 tw[kValueStageWords / 2];
#pragma unroll
                for (uint32_t j = 0; j < kValueStageWords / 2; ++j) tw[j] = raw_ld128(vp + 16u * j);  // (≤ 80 bytes past a block: the arena's slack)
#pragma unroll
                for (uint32_t j = 0; j < kValueStageWords / 2; ++j) {
                    vstage.w[2 * j][threadIdx.x] = tw[j].lo;
                    vstage.w[2 * j + 1][threadIdx.x] = tw[j].hi;
                }
                vstage.w[kValueStageWords][threadIdx.x] = 0;
            }
            if (left_pad_32_staged(vstage, threadIdx.x, loc.len, L) || left_pad_32_raw(vp, loc.len, L)) {
#else
            if (left_pad_32_raw(vp, loc.len, L)) {
#endif
#pragma unroll
                for (int k = 0; k < 4; ++k)  // byte i of the value: limb (31 - i) / 4, big-endian inside it
                    padded[k] = uint64_t(__builtin_bswap32(L[7 - 2 * k])) | uint64_t(__builtin_bswap32(L[6 - 2 * k])) << 32;
            } else {
                Rd v;
                v.init(vp, loc.len);
                left_pad_32_words(v, padded);
            }
        }
        if (!(flags & SC_VALUE_MATCHABLE)) { st = IPCFP_ST_FALSE_VALUE; break; }  // can never equal "0x" + 64 hex digits
#if !IPCFP_VS_PRELOAD
        {
            const Raw16 a = raw_ld128(c.value), b = raw_ld128(c.value + 16);
            cv[0] = a.lo, cv[1] = a.hi, cv[2] = b.lo, cv[3] = b.hi;
        }
#endif
        const uint64_t diff = (padded[0] ^ cv[0]) | (padded[1] ^ cv[1]) | (padded[2] ^ cv[2]) | (padded[3] ^ cv[3]);
        st = diff == 0 ? IPCFP_ST_TRUE : IPCFP_ST_FALSE_VALUE;                                            // :169
    } while (false);
    status[t] = uint8_t(st);
}

int launch_storage_run_actors_table(ipcfp_ctx* ctx, const WitnessView& w, const void* table_d, const void* claims_d, void* runs_d,
                                    uint32_t n_runs, uint32_t undecided) {
    if (n_runs == 0) return IPCFP_OK;
    hipLaunchKernelGGL(k_storage_run_actors_table, dim3(div_up(n_runs, 256)), dim3(256), 0, ctx->stream, w,
                       static_cast<const HamtNodeRec*>(table_d), static_cast<const StorageClaimPacked*>(claims_d),
                       static_cast<StorageRun*>(runs_d), n_runs, undecided);
    IPCFP_HIP(ctx, hipGetLastError());
    return IPCFP_OK;
}
// root_children_d: n_runs × 34 words of scratch (block of every run's root, the runs' CID matches, then 32 children per run), or null: every claim resolves its own
int launch_verify_storage_table(ipcfp_ctx* ctx, const WitnessView& w, const void* table_d, const void* claims_d, uint32_t n,
                                const uint32_t* run_of_d, const void* runs_d, uint32_t n_runs, uint32_t* root_children_d,
                                const ipcfp_trust_policy_t& trust, uint32_t undecided, uint8_t* status_d) {
    uint32_t* root_block = root_children_d;
    uint32_t* run_match = root_children_d ? root_children_d + n_runs : nullptr;
    uint32_t* root_child = root_children_d ? root_children_d + 2 * size_t(n_runs) : nullptr;
    if (root_children_d && n_runs)
        hipLaunchKernelGGL(k_storage_run_children, dim3(div_up(n_runs * 32u, 256)), dim3(256), 0, ctx->stream, w,
                           static_cast<const HamtNodeRec*>(table_d), static_cast<const StorageRun*>(runs_d), n_runs,
                           static_cast<const StorageClaimPacked*>(claims_d), root_block, run_match, root_child);
    hipLaunchKernelGGL(k_verify_storage_table, dim3(div_up(n, 256)), dim3(256), 0, ctx->stream, w,
                       static_cast<const HamtNodeRec*>(table_d), static_cast<const StorageClaimPacked*>(claims_d), n, run_of_d,
                       static_cast<const StorageRun*>(runs_d), root_block, run_match, root_child, trust, undecided, status_d);
    IPCFP_HIP(ctx, hipGetLastError());
    return IPCFP_OK;
}

int launch_storage_run_flags(ipcfp_ctx* ctx, const void* claims_d, uint32_t n, uint32_t* flag_d) {
    hipLaunchKernelGGL(k_storage_run_flags, dim3(div_up(n, 256)), dim3(256), 0, ctx->stream,
                       static_cast<const StorageClaimPacked*>(claims_d), n, flag_d);
    IPCFP_HIP(ctx, hipGetLastError());
    return IPCFP_OK;
}
int launch_storage_run_heads(ipcfp_ctx* ctx, const uint32_t* flag_d, const uint32_t* pos_d, uint32_t n, uint32_t* run_of_d, void* runs_d) {
    hipLaunchKernelGGL(k_storage_run_heads, dim3(div_up(n, 256)), dim3(256), 0, ctx->stream, flag_d, pos_d, n, run_of_d,
                       static_cast<StorageRun*>(runs_d));
    IPCFP_HIP(ctx, hipGetLastError());
    return IPCFP_OK;
}
int launch_storage_run_facts(ipcfp_ctx* ctx, const WitnessView& w, const void* claims_d, void* runs_d, uint32_t n_runs) {
    if (n_runs == 0) return IPCFP_OK;
    hipLaunchKernelGGL(k_storage_run_facts, dim3(div_up(n_runs, 256)), dim3(256), 0, ctx->stream, w,
                       static_cast<const StorageClaimPacked*>(claims_d), static_cast<StorageRun*>(runs_d), n_runs);
    IPCFP_HIP(ctx, hipGetLastError());
    return IPCFP_OK;
}

// the one-lane kernel over the whole batch (pending_only = 0) or over what the run / group kernels left kStPending
int launch_verify_storage_lanes(ipcfp_ctx* ctx, const WitnessView& w, const StorageClaimPacked* claims_d, uint32_t n,
                                const ipcfp_trust_policy_t& trust, uint8_t* status_d, int pending_only) {
    static const int waves = [] {
        const char* e = std::getenv("IPCFP_STORAGE_WAVES");
        const int v = e ? std::atoi(e) : 4;
        return v >= 2 && v <= 4 ? v : 4;
    }();
    const dim3 g(div_up(n, 256)), blk(256);
    if (waves == 2) hipLaunchKernelGGL(k_verify_storage<2>, g, blk, 0, ctx->stream, w, claims_d, n, trust, status_d, pending_only);
    else if (waves == 3) hipLaunchKernelGGL(k_verify_storage<3>, g, blk, 0, ctx->stream, w, claims_d, n, trust, status_d, pending_only);
    else hipLaunchKernelGGL(k_verify_storage<4>, g, blk, 0, ctx->stream, w, claims_d, n, trust, status_d, pending_only);
    IPCFP_HIP(ctx, hipGetLastError());
    return IPCFP_OK;
}

}  // namespace ipcfp
