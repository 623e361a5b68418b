// csrc/host/verify_storage.cpp — `verify_storage_proof` batches and `create_event_filter`.
//
// Host side of src/proofs/storage/verifier.rs:24-63: parse the claim strings once
// (parse_cid → src/proofs/common/witness.rs:60-64; hex → storage/verifier.rs:155-157), upload the
// packed claims, run one kernel over the batch, download the status bytes.
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../common.h"
#include "../kernels/claims_dev.h"
#include "../kernels/launch.h"
#include "../kernels/hamt_table.h"
#include "../kernels/storage_runs.h"
#include "cidstr.h"

using namespace ipcfp;

namespace ipcfp {

CidKey key_from_slot(const uint8_t* slot40);

// Parse a CID string into a witness key.  `parsed`: Cid::try_from succeeded.  `canonical`: the
// string equals Cid::to_string() of what it parses to.  A CID longer than the 40-byte slot becomes its fold
// (cidstr.h cid_to_slot), the key the device makes of the same CID when it reads it out of a block.
void parse_cid_claim(const char* s, CidKey& key, bool& parsed, bool& canonical) {
    std::vector<uint8_t> bin;
    parsed = cid_from_string(s, bin);
    canonical = false;
    for (auto& w : key.w) w = ~0ULL;
    if (!parsed) return;
    canonical = cid_to_string(bin.data(), bin.size()) == s;
    uint8_t slot[IPCFP_CID_SLOT];
    cid_to_slot(bin.data(), bin.size(), slot);
    std::memcpy(key.w, slot, IPCFP_CID_SLOT);
}

static const ipcfp_trust_policy_t kAcceptAll = {0, 0, 0, 0};

// `verify_storage_proof` over a batch of packed claims resident in HBM (src/proofs/storage/verifier.rs:24-63, the loop of
// src/proofs/verifier.rs:19-28).  A batch that is large against the witness
//   1. cuts the claims into RUNS that agree on (child, state root, actor, actor state, storage root) and decodes what a
//      run shares once (kernels/storage_runs.h),
//   2. tabulates every block of the witness as a HAMT node (kernels/hamt_table.h: ONE parse per node instead of one per
//      proof that passes through it — a contract's storage root is decoded once, not 256 times),
//   3. settles every claim from its run's record and two or three table records (k_verify_storage_table),
// and the one-lane kernel takes what that leaves pending (an inline small-map layout, a block the table does not cover).
// A small batch — and everything, with IPCFP_HAMT_TABLE=0 — goes through the one-lane kernel alone.  The table lives
// for this call only.  One host synchronisation (the number of runs).
int launch_verify_storage(ipcfp_ctx* ctx, ipcfp_witness* wit, const StorageClaimPacked* claims_d, uint32_t n,
                          const ipcfp_trust_policy_t& trust, uint8_t* status_d) {
    if (n == 0) return IPCFP_OK;
    ProfileScope prof(ctx, IPCFP_K_STORAGE_VERIFY);
    const WitnessView w = witness_view(wit);
    const int forced = ctx->hamt_table;  // (env IPCFP_HAMT_TABLE / ipcfp_ctx_set_tuning "hamt_table")
    const bool tabled = forced == 1 || (forced != 0 && uint64_t(n) * 16u >= wit->n);
    if (!tabled) return launch_verify_storage_lanes(ctx, w, claims_d, n, trust, status_d, 0);
    constexpr uint32_t kUndecided = 0xfdu;
    DevBuf<HamtNodeRec> table;
    DevBuf<uint32_t> long_list, long_count;
    IPCFP_HIP(ctx, table.alloc(wit->n));
    static const bool ring = [] { const char* e = std::getenv("IPCFP_HAMT_TABLE_FORM"); return e && e[0] == 'r'; }();
    // Round 6: the node table (0.6 ms of one-lane parses, latency-bound) and the runs' boundary pass (0.18 ms of streaming
    // 248-byte records, bandwidth-bound) need nothing of each other: the table goes to the AUX stream, its 32-lane outline
    // of the long blocks — whose grid wants the list's size, i.e. the call's one synchronisation — to the K1 stream beside it,
    // and the main stream runs flags → scan → heads → typed decodes meanwhile and joins both before the first kernel that
    // reads a record.  IPCFP_STORAGE_SIDE=0: everything on the main stream in round 5's order.
    static const bool side_env = [] { const char* e = std::getenv("IPCFP_STORAGE_SIDE"); return !(e && std::atoi(e) == 0); }();
    const bool side = side_env && !ring && ctx->stream_aux != ctx->stream && ctx->aux_event && ctx->main_event;
    const bool side2 = side && ctx->stream_k1 != ctx->stream && ctx->stream_k1 != ctx->stream_aux;
    // (declared AFTER table / long_list / long_count: an early return drains the side streams before those buffers go back
    // to the pool — ADVICE r5)
    StreamDrainGuard aux_guard(ctx->stream_aux), k1_guard(ctx->stream_k1);
    int rc = IPCFP_OK;
    static const uint32_t table_kinds = [] {  // IPCFP_TABLE_FAST=0: the node table reads every entry item by item (round 5's way)
        const char* e = std::getenv("IPCFP_TABLE_FAST");
        return HK_ACTOR_STATE | HK_VEC_U8 | (e && std::atoi(e) == 0 ? uint32_t(HK_ITEM_BY_ITEM) : 0u);
    }();
    // (IPCFP_STORAGE_EARLY_OUTLINE=1, measured and left off: see below)
    static const bool early_env = [] { const char* e = std::getenv("IPCFP_STORAGE_EARLY_OUTLINE"); return e && std::atoi(e) == 1; }();
    const bool early_outline = side2 && early_env;
    uint32_t n_long = 0;
    hipEvent_t outline_event = nullptr;
    // the outline of the long blocks (its grid is the list's size): beside the lane kernel when that runs on the aux stream,
    // else on the aux stream beside the runs' typed decodes (round 5)
    auto queue_outline = [&]() -> int {
        if (!n_long) return IPCFP_OK;
        hipStream_t s = ctx->stream;
        if (side2) {
            s = ctx->stream_k1;
            k1_guard.armed = true;
            IPCFP_HIP(ctx, hipStreamWaitEvent(s, ctx->main_event, 0));
        } else if (ctx->stream_aux != ctx->stream && ctx->aux_event) {
            s = ctx->stream_aux;
            aux_guard.armed = true;
        }
        int rc2 = launch_hamt_outline_list(ctx, s, w, table.p, long_list.p, long_count.p, n_long);
        if (!rc2) rc2 = launch_hamt_node_table_rest(ctx, s, w, long_list.p, long_count.p, n_long, HK_ACTOR_STATE | HK_VEC_U8, table.p);
        if (rc2) return rc2;
        if (s == ctx->stream_k1) {
            if (!ctx->k1_gate_event) IPCFP_HIP(ctx, hipEventCreateWithFlags(&ctx->k1_gate_event, hipEventDisableTiming));
            outline_event = ctx->k1_gate_event;
            IPCFP_HIP(ctx, hipEventRecord(outline_event, s));
        }
        return IPCFP_OK;
    };
    if (ring) {  // (round 3's form, for A/B runs: eight lanes per block with the ring reader, every block)
        rc = launch_hamt_node_table(ctx, wit->arena.p, wit->k1_meta.p, uint32_t(wit->n), table_kinds, table.p);
    } else {
        // the long blocks (4-5 KB state-tree nodes: the head of the schedule) as a work list for the 32-lane outline …
        IPCFP_HIP(ctx, long_list.alloc(wit->n));
        IPCFP_HIP(ctx, long_count.alloc(1));
        IPCFP_HIP(ctx, hipMemsetAsync(long_count.p, 0, 4, ctx->stream));
        rc = launch_hamt_list_long(ctx, wit->k1_meta.p, uint32_t(wit->n), long_list.p, long_count.p);
        if (!rc && early_outline) {
            // The outline's grid wants the list's size.  Read behind the runs' boundary pass (the call's one synchronisation) it
            // starts 0.33 ms into the call and ends with the lane kernel; read HERE — a second synchronisation, of a stream
            // that holds 7 µs of work — it starts with the call.  Measured: 1.40 ms against 1.21 (profiles/r06_experiments.md) —
            // the synchronisation puts 60 µs in front of everything and the lane kernel, which the call waits for, shares the
            // chip with the outline from its first workgroup (546 -> 694 µs).  Off.
            IPCFP_HIP(ctx, d2h_small(ctx, &n_long, long_count.p, 4, ctx->stream));
            IPCFP_HIP(ctx, sync_stream(ctx, ctx->stream));
            IPCFP_HIP(ctx, hipEventRecord(ctx->main_event, ctx->stream));
            rc = queue_outline();
        }
        // … everything shorter: one block per lane, line-staged reader
        if (!rc && side) {
            IPCFP_HIP(ctx, hipEventRecord(ctx->main_event, ctx->stream));  // (everything that made the witness and took `table` from the pool is behind this)
            IPCFP_HIP(ctx, hipStreamWaitEvent(ctx->stream_aux, ctx->main_event, 0));
            aux_guard.armed = true;
            hipStream_t saved = ctx->stream;
            ctx->stream = ctx->stream_aux;  // (the launcher queues on the context's stream)
            rc = launch_hamt_node_table_lane(ctx, wit->arena.p, wit->k1_meta.p, uint32_t(wit->n), table_kinds, table.p);
            ctx->stream = saved;
        } else if (!rc) {
            rc = launch_hamt_node_table_lane(ctx, wit->arena.p, wit->k1_meta.p, uint32_t(wit->n), table_kinds, table.p);
        }
    }
    if (rc) return rc;
    DevBuf<uint32_t> flag, pos, run_of;
    DevBuf<uint64_t> scratch, total_d;
    IPCFP_HIP(ctx, flag.alloc(n));
    IPCFP_HIP(ctx, pos.alloc(n));
    IPCFP_HIP(ctx, run_of.alloc(n));
    IPCFP_HIP(ctx, scratch.alloc(size_t(div_up(n, 1024)) + 2));
    IPCFP_HIP(ctx, total_d.alloc(1));
    rc = launch_storage_run_flags(ctx, claims_d, n, flag.p);
    if (rc) return rc;
    rc = launch_scan_u32(ctx, flag.p, n, pos.p, total_d.p, scratch.p);
    if (rc) return rc;
    uint64_t n_runs = 0;
    IPCFP_HIP(ctx, d2h_small(ctx, &n_runs, total_d.p, 8, ctx->stream));
    if (!ring && !early_outline) IPCFP_HIP(ctx, d2h_small(ctx, &n_long, long_count.p, 4, ctx->stream));
    IPCFP_HIP(ctx, sync_stream(ctx, ctx->stream));
    if (!early_outline) {
        rc = queue_outline();
        if (rc) return rc;
    }
    if (aux_guard.armed) IPCFP_HIP(ctx, hipEventRecord(ctx->aux_event, ctx->stream_aux));
    DevBuf<StorageRun> runs;
    IPCFP_HIP(ctx, runs.alloc(n_runs));
    rc = launch_storage_run_heads(ctx, flag.p, pos.p, n, run_of.p, runs.p);
    if (rc) return rc;
    rc = launch_storage_run_facts(ctx, w, claims_d, runs.p, uint32_t(n_runs));
    if (rc) return rc;
    // the table is whole from here on
    if (aux_guard.armed) IPCFP_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->aux_event, 0));
    if (outline_event) IPCFP_HIP(ctx, hipStreamWaitEvent(ctx->stream, outline_event, 0));
    aux_guard.armed = k1_guard.armed = false;  // the main stream is ordered behind the side kernels now: pool reuse on it is safe
    rc = launch_storage_run_actors_table(ctx, w, table.p, claims_d, runs.p, uint32_t(n_runs), kUndecided);
    if (rc) return rc;
    rc = launch_storage_run_actors_lane(ctx, w, claims_d, runs.p, uint32_t(n_runs), kUndecided);
    if (rc) return rc;
    // the first step of the runs' storage gets, once per run (IPCFP_STORAGE_RUN_CHILDREN=0: every claim by itself)
    static const bool run_children = [] { const char* e = std::getenv("IPCFP_STORAGE_RUN_CHILDREN"); return !(e && std::atoi(e) == 0); }();
    DevBuf<uint32_t> root_children;
    if (run_children && n_runs) IPCFP_HIP(ctx, root_children.alloc(size_t(n_runs) * 34u));
    rc = launch_verify_storage_table(ctx, w, table.p, claims_d, n, run_of.p, runs.p, uint32_t(n_runs), root_children.p, trust, kUndecided,
                                     status_d);
    if (rc) return rc;
    return launch_verify_storage_lanes(ctx, w, claims_d, n, trust, status_d, 1);
    // (the scratch buffers go back to the pool on return; reuse is ordered on the one stream)
}

}  // namespace ipcfp

extern "C" {

int ipcfp_verify_storage_proofs(ipcfp_ctx_t* ctx, ipcfp_witness_t* w, const ipcfp_storage_proof_t* proofs,
                                uint64_t n, const ipcfp_trust_policy_t* trust, ipcfp_status_t* status) {
    if (!ctx || !w || w->ctx != ctx || (n && (!proofs || !status))) return IPCFP_E_INVALID;
    if (n >= 0xffffffffULL) return set_error(ctx, IPCFP_E_UNSUPPORTED, "batch too large");
    if (n == 0) return IPCFP_OK;
    std::vector<StorageClaimPacked> packed(n);
    for (uint64_t i = 0; i < n; ++i) {
        const ipcfp_storage_proof_t& p = proofs[i];
        StorageClaimPacked& c = packed[i];
        std::memset(&c, 0, sizeof c);
        c.child_epoch = p.child_epoch;
        c.actor_id = p.actor_id;
        bool parsed, canon;
        parse_cid_claim(p.child_block_cid, c.child, parsed, canon);
        if (parsed) c.flags |= SC_CHILD_PARSED;
        parse_cid_claim(p.parent_state_root, c.state_root, parsed, canon);
        if (parsed && canon) c.flags |= SC_STATE_ROOT_CANON;
        parse_cid_claim(p.actor_state_cid, c.actor_state, parsed, canon);
        if (parsed && canon) c.flags |= SC_ACTOR_STATE_CANON;
        parse_cid_claim(p.storage_root, c.storage_root, parsed, canon);
        if (parsed && canon) c.flags |= SC_STORAGE_ROOT_CANON;
        // slot: hex::decode_to_slice(slot.trim_start_matches("0x"), &mut [u8; 32])
        if (p.slot) {
            const char* s = p.slot;
            while (s[0] == '0' && s[1] == 'x') s += 2;
            std::vector<uint8_t> b;
            if (std::strlen(s) == 64 && hex_decode(s, 64, b)) {
                std::memcpy(c.slot, b.data(), 32);
                c.flags |= SC_SLOT_PARSED;
            }
        }
        // value: compared as `"0x" + hex(padded)` ignoring ASCII case
        if (p.value && std::strlen(p.value) == 66 && p.value[0] == '0' && (p.value[1] == 'x' || p.value[1] == 'X')) {
            std::vector<uint8_t> b;
            if (hex_decode(p.value + 2, 64, b)) {
                std::memcpy(c.value, b.data(), 32);
                c.flags |= SC_VALUE_MATCHABLE;
            }
        }
    }
    IPCFP_ENTER(ctx);
    DevBuf<StorageClaimPacked> cd;
    DevBuf<uint8_t> sd;
    IPCFP_HIP(ctx, cd.alloc(n));
    IPCFP_HIP(ctx, sd.alloc(n));
    IPCFP_HIP(ctx, hipMemcpyAsync(cd.p, packed.data(), n * sizeof(StorageClaimPacked), hipMemcpyHostToDevice, ctx->stream));
    int rc = launch_verify_storage(ctx, w, cd.p, uint32_t(n), trust ? *trust : kAcceptAll, sd.p);
    if (rc) return rc;
    IPCFP_HIP(ctx, hipMemcpyAsync(status, sd.p, n, hipMemcpyDeviceToHost, ctx->stream));
    IPCFP_HIP(ctx, sync_stream(ctx, ctx->stream));
    return IPCFP_OK;
}

int ipcfp_verify_storage_claims_device(ipcfp_ctx_t* ctx, ipcfp_witness_t* w, const void* claims_d, uint64_t n,
                                       const ipcfp_trust_policy_t* trust, void* status_d) {
    if (!ctx || !w || w->ctx != ctx || (n && (!claims_d || !status_d))) return IPCFP_E_INVALID;
    if (n >= 0xffffffffULL) return set_error(ctx, IPCFP_E_UNSUPPORTED, "batch too large");
    if (n == 0) return IPCFP_OK;
    IPCFP_ENTER(ctx);
    int rc = launch_verify_storage(ctx, w, static_cast<const StorageClaimPacked*>(claims_d), uint32_t(n),
                                   trust ? *trust : kAcceptAll, static_cast<uint8_t*>(status_d));
    if (rc) return rc;
    IPCFP_HIP(ctx, sync_stream(ctx, ctx->stream));
    return IPCFP_OK;
}

int ipcfp_cid_from_string(const char* s, uint8_t out40[IPCFP_CID_SLOT]) {
    if (!s || !out40) return IPCFP_E_INVALID;
    std::vector<uint8_t> bin;
    if (!cid_from_string(s, bin)) return IPCFP_E_PARSE;
    cid_to_slot(bin.data(), bin.size(), out40);
    return int(bin.size());
}

int ipcfp_cid_to_slot(const uint8_t* cid, uint32_t len, uint8_t out40[IPCFP_CID_SLOT]) {
    if (!cid || !out40) return IPCFP_E_INVALID;
    if (!cid_binary_ok(cid, len)) return IPCFP_E_PARSE;
    cid_to_slot(cid, len, out40);
    return int(len);
}

int ipcfp_cid_to_string(const uint8_t* cid, uint32_t len, char* out, uint32_t cap) {
    if (!cid || !out || !cid_binary_ok(cid, len)) return IPCFP_E_INVALID;
    const std::string s = cid_to_string(cid, len);
    if (s.size() + 1 > cap) return IPCFP_E_INVALID;
    std::memcpy(out, s.c_str(), s.size() + 1);
    return int(s.size());
}

int ipcfp_create_event_filter(ipcfp_ctx_t* ctx, const char* event_sig, const char* subnet_id,
                              ipcfp_event_filter_t* out) {
    if (!ctx || !event_sig || !subnet_id || !out) return IPCFP_E_INVALID;
    // topic0 = Keccak-256(signature) on the device (hash_event_signature, common/evm.rs:62-69)
    const uint64_t off = 0;
    const uint32_t len = uint32_t(std::strlen(event_sig));
    int rc = ipcfp_keccak256_batch(ctx, reinterpret_cast<const uint8_t*>(event_sig), len, &off, &len, 1, out->topic0);
    if (rc) return rc;
    // topic1 = ascii_to_bytes32(subnet_id) (common/evm.rs:72-78): right-padded with zeros, truncated at 32
    std::memset(out->topic1, 0, 32);
    const size_t sl = std::strlen(subnet_id);
    std::memcpy(out->topic1, subnet_id, sl < 32 ? sl : 32);
    return IPCFP_OK;
}

}  // extern "C"
