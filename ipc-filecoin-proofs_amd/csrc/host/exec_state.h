// csrc/host/exec_state.h — host-side handles shared by the verifier and the generator entry points.
#pragma once
#include <functional>

#include "../common.h"
#include "../kernels/amt_enum.h"
#include "../kernels/event_table.h"
#include "../kernels/tipset_ctx.h"
#include "../kernels/types_dev.h"
#include "../kernels/launch.h"

namespace ipcfp {

// device buffers of one context's execution order
struct ExecState {
    DevBuf<CidKey> keys;        // raw for_each sequence
    DevBuf<unsigned long long> slots;  // hash table: {key fingerprint, first raw position}
    DevBuf<uint32_t> first;     // 1 where the raw position is a first occurrence
    DevBuf<uint32_t> pos;       // exclusive scan of `first` → execution index
    DevBuf<uint32_t> inv;       // execution index → raw position (filled by launch_exec_finish on the verify path)
    struct Word64 { uint64_t* p = nullptr; } total;           // device: number of distinct messages (= exec_len)
    struct WordErr { unsigned long long* p = nullptr; } err;  // packed first error of the whole reconstruction
    DevBuf<uint64_t> total_own;                 // backing store when the call's control block is full
    DevBuf<unsigned long long> err_own;
    DevBuf<AmtRootSpec> roots;  // stage 1 output: the BLS / secp message AMT roots of every parent block
    uint32_t mask = 0;
    uint64_t raw_len = 0, exec_len = 0;
    uint32_t status = IPCFP_ST_ERR;
};

// reconstruct_execution_order (verify_txmeta = 1, events/utils.rs:16-30) or build_execution_order
// (verify_txmeta = 0, events/utils.rs:32-46) of the context stored at ctx_d (device).  With a recording
// view every block the traversal loads is marked.
// `host_len` = false leaves exec_len on the device only (ExecState::total) and saves a synchronisation.
// `prepared` = true: stage 1 (ex.roots / ex.err) was already run by launch_tipset_prepare.
// `extra` (with `prepared`): one more AMT — the receipts AMT, whose root launch_tipset_prepare left at
// ex.roots[2 * n_parents] — enumerated in the same launches as the message AMTs (amt_enum.h).
// `after_enum` (nullable): called once the enumeration is back (the stream is idle at that moment), before the
// execution-order hash kernels are queued — the caller's chance to start independent work on another stream.
int build_exec_order(ipcfp_ctx* ctx, const WitnessView& view, const TipsetCtxDev* ctx_d, uint32_t n_parents,
                     ExecState& ex, int verify_txmeta = 1, bool host_len = true, bool prepared = false,
                     EnumExtra* extra = nullptr, const std::function<int()>* after_enum = nullptr);
// allocate ex.roots / ex.err for a context with n_parents parent blocks and reset the error word
int exec_state_prepare(ipcfp_ctx* ctx, ExecState& ex, uint32_t n_parents);

// device-resident result of one two-pass event scan (scan_events.cpp)
struct ScanResult {
    uint32_t status = IPCFP_ST_ERR;
    // where an Err status arose (include/ipcfp.h IPCFP_SCAN_PHASE_*): the receipts enumeration comes first for the WHOLE
    // tipset, so the merge of receipt-range shards prefers an enumeration error of any shard to an events error of a lower one
    uint32_t phase = 0;
    uint64_t n_idx = 0, n_matches = 0;
    DevBuf<uint8_t> has;
    DevBuf<ipcfp_event_match_t> matches;
    // caller-owned HBM outputs (ipcfp_scan_events_device): when set and large enough PASS 2 writes there directly
    // (has_p / matches_p then point into them and `has` / `matches` stay empty) — no copy kernel behind the scan
    uint8_t* ext_has = nullptr;
    uint64_t ext_has_cap = 0;
    ipcfp_event_match_t* ext_matches = nullptr;  // capacity = the call's cap_matches
    uint8_t* has_p = nullptr;
    ipcfp_event_match_t* matches_p = nullptr;
};
// A scan that RIDES on a verify call (ipcfp_verify_and_scan_device): the route without a mid-call synchronisation
// (verify_fast.cpp) queues the scan's tail behind its verify kernel — the receipts enumeration and the event table are
// that call's own — and the scan's results come back with the call's one synchronisation.  `done` false afterwards:
// the ride did not happen (another route, an anomaly, receipts the table does not cover) and the caller scans as usual.
struct ScanRide {
    ipcfp_event_filter_t filter{};
    int has_actor = 0;
    uint64_t actor = 0;
    uint8_t* has_d = nullptr;          // caller HBM, cap_receipts bytes (nullable)
    uint64_t cap_receipts = 0;
    ipcfp_event_match_t* matches_d = nullptr;  // caller HBM, cap_matches records (nullable)
    uint64_t cap_matches = 0;
    bool done = false;
    uint32_t status = IPCFP_ST_ERR;
    uint64_t n_idx = 0, n_matches = 0;
};
ScanParams scan_params_of(const ipcfp_event_filter_t& filter, int has_actor, uint64_t actor);
// the look-back state of k_scan_tail_fused for n_tiles tiles (context-owned; zeroed when (re)allocated)
int scan_tail_scratch(ipcfp_ctx* ctx, uint32_t n_tiles, unsigned long long** out);

// `cap_matches`: how many matches the caller can take.  With a known capacity PASS 2 is launched right behind
// PASS 1 (the kernel clips its writes) and the scan has ONE synchronisation; with kAllMatches the match count is
// read back first and out.matches is sized to it.
constexpr uint64_t kAllMatches = ~0ull;
int scan_events_device(ipcfp_ctx* ctx, ipcfp_witness* w, const CidKey& root, const ipcfp_event_filter_t& filter,
                       int has_actor, uint64_t actor, uint32_t* touched_d, ScanResult& out,
                       uint64_t cap_matches = kAllMatches);

// The event table of the receipts AMT `root` (range = the witness's receipt range): the cached one, or a new one
// (k_receipt_events; the first failing receipt goes to the table's err_word).  `en` = the enumeration.
// `on_aux`: a NEW table's kernels go to the aux stream (behind the block-order parse they depend on) instead of the
// main stream, which joins them where the table is first read.  Only for a caller whose main stream is idle right now.
int event_table_get(ipcfp_ctx* ctx, ipcfp_witness* w, const CidKey& root, const EnumCached* en, const EventTableCached** out,
                    bool on_aux = false);
// the main stream waits (on the device) for what the aux stream has queued for this witness's tables
int event_table_join(ipcfp_ctx* ctx, ipcfp_witness* w);
// the table's per-receipt match counts when they were counted for exactly this filter, else null
const uint32_t* event_table_counts(const EventTableCached* t, const ipcfp_event_filter_t& filter, int has_actor, uint64_t actor);

// queue the block-order event parse of the witness on the aux stream unless its block table exists (scan_events.cpp)
int block_table_prefetch(ipcfp_ctx* ctx, ipcfp_witness* w, const ipcfp_event_filter_t* filter, int has_actor, uint64_t actor);

CidKey key_from_slot(const uint8_t* slot40);

}  // namespace ipcfp
