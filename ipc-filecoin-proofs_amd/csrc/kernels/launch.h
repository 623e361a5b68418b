// csrc/kernels/launch.h — host-callable launchers of the HIP kernels.  Every
// launcher enqueues on ctx->stream and returns without synchronising.
#pragma once
#include <cstdint>

#include "../common.h"
#include "witness_dev.h"

namespace ipcfp {

// --- blake2b_cid.hip (K1) ---
int launch_chunk_order(ipcfp_ctx* ctx, const uint32_t* len_d, uint32_t n, uint32_t* bins_d, uint32_t* order_d);
int launch_k1_layout(ipcfp_ctx* ctx, const uint32_t* len_d, uint32_t n, uint32_t* bins_d, uint32_t* order_d,
                     uint32_t* sched_len_d, uint64_t* sched_off_d, uint64_t* total_d, uint64_t* scan_scratch_d,
                     uint64_t* off_by_id_d, void* meta_d);
int launch_gather_cids(ipcfp_ctx* ctx, const uint32_t* order_d, const uint8_t* cids_d, uint32_t n, uint8_t* sched_cids_d);
int launch_blake2b256_cid(ipcfp_ctx* ctx, const uint8_t* arena, const void* meta, const uint8_t* sched_cids40, uint32_t n,
                          uint32_t* ok_bits, uint8_t* status, unsigned long long* counters);
int launch_blake2b256_raw(ipcfp_ctx* ctx, const uint8_t* arena, const void* meta, uint32_t n, uint8_t* out32);

// --- repack.hip ---
// new_off[i] = exclusive prefix sum of round_up(len[i], 16); *total_d receives the arena size.
int launch_aligned_offsets(ipcfp_ctx* ctx, const uint32_t* len_d, uint32_t n, uint64_t* new_off_d,
                           uint64_t* total_d, uint64_t* scratch_d /* >= div_up(n,1024)+1 */);
// off[i] = exclusive prefix sum of len[i] (blocks back to back); *total_d = the sum
int launch_tight_offsets(ipcfp_ctx* ctx, const uint32_t* len_d, uint32_t n, uint64_t* off_d, uint64_t* total_d,
                         uint64_t* scratch_d /* >= div_up(n,1024)+1 */);
// cids40[i] = prefix ‖ digests32[i] (zero-padded slot), then the n_esc escape slots as given (ipcfp_witness_create_packed)
int launch_expand_cids(ipcfp_ctx* ctx, const uint8_t* digests32_d, uint32_t n, const uint8_t* prefix, uint32_t prefix_len,
                       const uint32_t* esc_index_d, const uint8_t* esc_cids40_d, uint32_t n_esc, uint8_t* cids40_d);
// dst[new_off[i] .. +len[i]) = src[old_off[i] .. +len[i]); pad bytes up to the next 16 are zeroed.
int launch_repack(ipcfp_ctx* ctx, const uint8_t* src, const uint64_t* old_off, const uint32_t* len,
                  const uint64_t* new_off, uint32_t n, uint8_t* dst);
// *flag_d |= 1 if any off[i] % 16 != 0
int launch_check_aligned(ipcfp_ctx* ctx, const uint64_t* off_d, uint32_t n, uint32_t* flag_d);

// --- hash_short.hip (K2 Keccak-256, K3 SHA-256) ---
int launch_keccak256(ipcfp_ctx* ctx, const uint8_t* bytes, const uint64_t* off, const uint32_t* len, uint32_t n,
                     uint8_t* out32);
int launch_sha256(ipcfp_ctx* ctx, const uint8_t* bytes, const uint64_t* off, const uint32_t* len, uint32_t n,
                  uint8_t* out32);

// --- walk.hip (K5 amt_get, K7 hamt_get) ---
int launch_amt_get(ipcfp_ctx* ctx, const WitnessView& w, const CidKey& root, int version, int vkind,
                   const uint64_t* index_d, uint32_t n, uint8_t* status_d, void* loc_d);
int launch_hamt_get(ipcfp_ctx* ctx, const WitnessView& w, const CidKey& root, uint32_t bit_width, int vkind,
                    const uint8_t* keys_d, const uint32_t* key_off_d, const uint32_t* key_len_d, uint32_t n,
                    uint8_t* status_d, void* loc_d, int pending_only = 0);
// --- hamt_levels.hip --- K7 level by level: every visited node decoded once; what it leaves kStPending is the walker's
size_t hamt_levels_scratch_words(uint32_t n, uint32_t n_blocks, uint32_t levels);
// one lane per SHORT node of a level's work list (hamt_table_lane.hip; list entries {block, length, offset lo, offset hi})
int launch_hamt_lv_parse_lane(ipcfp_ctx* ctx, const WitnessView& w, const void* list_d, const uint32_t* count_d, uint32_t cap, uint32_t bound,
                              uint32_t kind_bit, void* recs_d, uint32_t* etab_of_d);
int launch_hamt_get_levels(ipcfp_ctx* ctx, const WitnessView& w, const CidKey& root, uint32_t bit_width, int vkind,
                           const uint8_t* keys_d, const uint32_t* key_off_d, const uint32_t* key_len_d, uint32_t n,
                           uint8_t* status_d, void* loc_d, uint32_t levels, uint32_t* scratch_d, void* recs_d, bool coop = true,
                           void* etabs_d = nullptr /* etab_cap × HamtEntryTab (hamt_table.h): the visited nodes' bucket entries */,
                           uint32_t etab_cap = 0);

// --- hamt_table.hip / walk.hip --- the HAMT node table of a witness (hamt_table.h) and K7 over it
int launch_hamt_node_table(ipcfp_ctx* ctx, const uint8_t* arena, const void* k1_meta_d, uint32_t n_blocks, uint32_t kinds, void* recs_d);
uint32_t hamt_kind_bit(int vkind);  // HK_* bit of a value kind, 0: the table does not know that kind
int launch_hamt_get_table(ipcfp_ctx* ctx, const WitnessView& w, const void* table_d, const CidKey& root, uint32_t bit_width,
                          int vkind, const uint8_t* keys_d, const uint32_t* key_off_d, const uint32_t* key_len_d, uint32_t n,
                          uint8_t* status_d, void* loc_d);

// --- verify_storage.hip ---
struct StorageClaimPacked;
int launch_verify_storage(ipcfp_ctx* ctx, ipcfp_witness* w, const StorageClaimPacked* claims_d, uint32_t n,
                          const ipcfp_trust_policy_t& trust, uint8_t* status_d);  // host/verify_storage.cpp
// the pieces of it (verify_storage.hip: runs of claims and their decoded facts; the one-lane kernel)
int launch_storage_run_flags(ipcfp_ctx* ctx, const void* claims_d, uint32_t n, uint32_t* flag_d);
int launch_storage_run_heads(ipcfp_ctx* ctx, const uint32_t* flag_d, const uint32_t* pos_d, uint32_t n, uint32_t* run_of_d, void* runs_d);
int launch_storage_run_facts(ipcfp_ctx* ctx, const WitnessView& w, const void* claims_d, void* runs_d, uint32_t n_runs);
int launch_storage_run_actors_table(ipcfp_ctx* ctx, const WitnessView& w, const void* table_d, const void* claims_d, void* runs_d,
                                    uint32_t n_runs, uint32_t undecided);
int launch_verify_storage_table(ipcfp_ctx* ctx, const WitnessView& w, const void* table_d, const void* claims_d, uint32_t n,
                                const uint32_t* run_of_d, const void* runs_d, uint32_t n_runs, uint32_t* root_children_d,
                                const ipcfp_trust_policy_t& trust, uint32_t undecided, uint8_t* status_d);
int launch_storage_run_actors_lane(ipcfp_ctx* ctx, const WitnessView& w, const void* claims_d, void* runs_d, uint32_t n_runs,
                                   uint32_t undecided);
int launch_verify_storage_lanes(ipcfp_ctx* ctx, const WitnessView& w, const StorageClaimPacked* claims_d, uint32_t n,
                                const ipcfp_trust_policy_t& trust, uint8_t* status_d, int pending_only);

// --- claims_compact.hip --- compact event claims → EventClaimPacked[n] + blob; scratch_u32: 4 n words, scan_scratch: div_up(n, 1024) + 2
// u64 whose LAST word receives the packed blob's length
int launch_expand_claims(ipcfp_ctx* ctx, const void* compact_d, uint32_t n, const ipcfp_event_claim_group_t* groups_d, uint32_t n_groups,
                         const uint8_t* cblob_d, uint64_t cblob_len, void* claims_out_d, uint8_t* blob_out_d, uint64_t cap_blob,
                         uint32_t* scratch_u32, uint64_t* scan_scratch);

// --- scan.hip ---
int launch_scan_u32(ipcfp_ctx* ctx, const uint32_t* in_d, uint32_t n, uint32_t* out_d, uint64_t* total_d,
                    uint64_t* scratch_d);

// --- verify_events.hip ---
struct TipsetCtxDev;
struct AmtRootSpec;
struct ReceiptRec;
struct EventRec;
struct EventTableView;
struct LeafRef;
struct EventClaimPacked;
int launch_ctx_headers(ipcfp_ctx* ctx, const WitnessView& w, TipsetCtxDev* ctxs_d, uint32_t n);
struct CtxFinish;
int launch_ctx_finish(ipcfp_ctx* ctx, TipsetCtxDev* ctx_d, const CtxFinish& a);
int launch_exec_finish_fused(ipcfp_ctx* ctx, TipsetCtxDev* ctx_d, const CtxFinish& a, uint32_t* first_d, uint32_t* pos_d,
                             uint64_t* tile_d, uint64_t* total_d, bool flags_ready = false);
int launch_exec_insert(ipcfp_ctx* ctx, const CidKey* keys_d, uint32_t n, unsigned long long* slots_d, uint32_t mask);
int launch_exec_insert_flags(ipcfp_ctx* ctx, const CidKey* keys_d, uint32_t n, unsigned long long* slots_d, uint32_t mask,
                             uint32_t* first_d);
int launch_scan_tiles_u64(ipcfp_ctx* ctx, uint64_t* tile_sums_d, uint32_t ntiles, uint64_t* total_d);
int launch_exec_finish(ipcfp_ctx* ctx, TipsetCtxDev* ctx_d, const uint64_t* total_d, const uint32_t* first_d,
                       const uint32_t* pos_d, uint32_t n, uint32_t* inv_d);
// jobs_d: device array of {TipsetCtxDev* ctx, AmtRootSpec* roots (nullable), unsigned long long* err}
// live_done / live_total / anomaly: the CID index is still being filled on another stream (tipset_prepare.hip LiveIndex)
int launch_tipset_prepare(ipcfp_ctx* ctx, const WitnessView& w, const void* jobs, const void* jobs_d, uint32_t n_jobs,
                          bool need_general, const uint32_t* live_done = nullptr, uint32_t live_total = 0,
                          uint32_t* anomaly = nullptr, bool defer_rehash = false,  // defer_rehash: see tipset_ctx.h txmeta_block
                          const void* inline_inputs = nullptr);  // host TipsetInputs of the one context: sent as a kernel argument
int launch_exec_roots(ipcfp_ctx* ctx, const WitnessView& w, const TipsetCtxDev* ctx_d, AmtRootSpec* roots_d,
                      unsigned long long* err_d, int verify_txmeta = 1, uint32_t n_parents = 0);
// the whole prologue of ONE context with a tipset key wider than the inline form (job: host PrepareJob)
int launch_tipset_prepare_wide(ipcfp_ctx* ctx, const WitnessView& w, const void* job, uint32_t n_parents);
int launch_exec_dedup(ipcfp_ctx* ctx, const WitnessView& w, const LeafRef* leaves_d, uint32_t n, CidKey* keys_d,
                      unsigned long long* slots_d, uint32_t mask, uint32_t* first_d);
int launch_exec_compact(ipcfp_ctx* ctx, const CidKey* keys_d, uint32_t n, const uint32_t* first_d,
                        const uint32_t* pos_d, CidKey* out_d);
int launch_verify_events(ipcfp_ctx* ctx, const WitnessView& w, const EventClaimPacked* claims_d, uint32_t n,
                         const TipsetCtxDev* ctxs_d, uint32_t n_ctxs, const uint8_t* blob_d, uint64_t blob_len,
                         const ipcfp_trust_policy_t& trust, const ipcfp_event_filter_t* filter, uint8_t* status_d,
                         void* where_d = nullptr, bool tabulated = false);
// --- verify_table.hip --- the claims the event table settles (everything else is left kStPending)
int launch_verify_events_table(ipcfp_ctx* ctx, const WitnessView& w, const EventClaimPacked* claims_d, uint32_t n,
                               const TipsetCtxDev* ctxs_d, uint32_t n_ctxs, const uint8_t* blob_d, uint64_t blob_len,
                               const ipcfp_trust_policy_t& trust, const ipcfp_event_filter_t& filter, int has_filter,
                               uint8_t* status_d, void* where_d);

// --- event_scan.hip (K6 scan, K8 replay bitmap) ---
int launch_scan_pass1(ipcfp_ctx* ctx, const WitnessView& w, const LeafRef* receipts_d, uint32_t n,
                      const ipcfp_event_filter_t& filter, int has_actor, uint64_t actor, uint32_t* counts_d,
                      unsigned long long* err_d);
int launch_scan_pass2(ipcfp_ctx* ctx, const WitnessView& w, const CidKey& receipts_root, const LeafRef* receipts_d,
                      uint32_t n, const ipcfp_event_filter_t& filter, int has_actor, uint64_t actor,
                      const uint32_t* counts_d, const uint32_t* offsets_d, void* matches_d, uint64_t matches_cap,
                      uint8_t* has_match_d, uint64_t has_cap, uint64_t has_base = 0,
                      const EventTableView* table = nullptr);
// the event table (event_table.h), step 2: receipt → its block's record (filter nullable: no match counts); the
// receipts the block table does not cover are walked
struct BlockRec;
int launch_receipt_events(ipcfp_ctx* ctx, const WitnessView& w, const LeafRef* receipts_d, uint32_t n,
                          const ipcfp_event_filter_t* filter, int has_actor, uint64_t actor, const BlockRec* brecs_d,
                          ReceiptRec* rrecs_d, uint32_t* counts_d, unsigned long long* err_d, hipStream_t stream = nullptr);
// --- block_events.hip --- step 1: every block parsed out of LDS in arena order, on `stream` (meta_d: K1Meta[n])
int launch_block_events(ipcfp_ctx* ctx, hipStream_t stream, const uint8_t* arena, const void* meta_d, uint32_t n,
                        const ipcfp_event_filter_t* filter, int has_actor, uint64_t actor, BlockRec* brecs_d,
                        EventRec* erecs_d, uint32_t cap_events, uint32_t* pool_used_d);
// the scan's tail in one launch (prefix sum by decoupled look-back + map + tabulated matches + results in the mailbox)
int launch_scan_tail_fused(ipcfp_ctx* ctx, const WitnessView& w, const LeafRef* receipts_d, uint32_t n, uint64_t dense_first,
                           const ipcfp_event_filter_t& filter, int has_actor, uint64_t actor, const EventTableView& table,
                           const uint32_t* counts_d, uint32_t* offsets_d, void* matches_d, uint64_t matches_cap, uint8_t* has_match_d,
                           uint64_t has_cap, uint64_t has_base, unsigned long long* scratch_d, unsigned long long epoch,
                           const unsigned long long* err_a_d, const unsigned long long* err_b_d, unsigned long long* mailbox_dev,
                           unsigned long long seq);
int launch_count_from_table(ipcfp_ctx* ctx, const WitnessView& w, const LeafRef* receipts_d, uint32_t n,
                            const ipcfp_event_filter_t& filter, int has_actor, uint64_t actor, const EventTableView& table,
                            uint32_t* counts_d, unsigned long long* err_d);

// --- base64.hip ---
int launch_base64_decode(ipcfp_ctx* ctx, const uint8_t* text_d, const void* spans_d, uint32_t n_blocks, uint32_t n_units,
                         const uint64_t* dst_off_d, uint8_t* arena_d, unsigned long long* first_bad_d);

int launch_parse_cid_arrays(ipcfp_ctx* ctx, const uint8_t* text_d, const void* spans_d, uint32_t n, uint8_t* cids_d,
                            unsigned long long* first_bad_d);

// --- generate.hip ---
int launch_generate_storage(ipcfp_ctx* ctx, const WitnessView& w, const CidKey& child, const void* specs_d, uint32_t n,
                            void* out_d);
int launch_mark_cids(ipcfp_ctx* ctx, const WitnessView& w, const CidKey* keys_d, uint32_t n, uint32_t* missing_d);
int launch_gather_keys(ipcfp_ctx* ctx, const CidKey* table_d, uint64_t table_len, const uint64_t* index_d, uint32_t n,
                       CidKey* out_d, uint32_t* oor_d);
int launch_gather_block_cids(ipcfp_ctx* ctx, const uint8_t* cids_d, const uint32_t* ids_d, uint32_t n, CidKey* out_d);

// --- shard.hip ---
int launch_plan_receipts(ipcfp_ctx* ctx, const WitnessView& rec, const CidKey& receipts_root, uint64_t lo, uint32_t n);
int launch_plan_receipts_all(ipcfp_ctx* ctx, const WitnessView& rec, const CidKey& receipts_root, uint32_t n,
                             const uint64_t* bounds_d, uint32_t n_shards, uint32_t words);
int launch_find_blocks(ipcfp_ctx* ctx, const WitnessView& w, const CidKey* keys_d, uint32_t n, uint32_t* ids_d);
int launch_gather_values(ipcfp_ctx* ctx, const WitnessView& w, const void* locs_d /* ipcfp_value_loc_t[n] */, uint32_t n,
                         uint8_t* out_d, uint64_t stride, unsigned long long* first_bad_d);
int launch_absolute_offsets(ipcfp_ctx* ctx, const uint64_t* off_d, uint32_t n, uint64_t base, uint64_t* out_d);
int launch_amt_root_info(ipcfp_ctx* ctx, const WitnessView& w, const CidKey& root, int version, int vkind, uint64_t* out_d);
int launch_subset_tables(ipcfp_ctx* ctx, const uint32_t* ids_d, uint32_t n, uint32_t n_src, const uint64_t* src_off,
                         const uint32_t* src_len, const uint8_t* src_cids, uint64_t* off_d, uint32_t* len_d,
                         uint8_t* cids_d, uint32_t* bad_d);

int launch_claims_window(ipcfp_ctx* ctx, const void* claims_d, uint32_t n, unsigned long long* win_d);         // claims_compact.hip
// --- the per-call HAMT node table in two kernels (hamt_table_lane.hip, hamt_levels.hip) ---
int launch_hamt_list_long(ipcfp_ctx* ctx, const void* meta_d, uint32_t n, uint32_t* work_d, uint32_t* count_d);
int launch_hamt_outline_list(ipcfp_ctx* ctx, hipStream_t stream, const WitnessView& w, void* recs_d, uint32_t* work_d, uint32_t* count_d,
                             uint32_t bound);
int launch_hamt_node_table_lane(ipcfp_ctx* ctx, const uint8_t* arena, const void* meta_d, uint32_t n, uint32_t kinds, void* recs_d);
int launch_hamt_node_table_rest(ipcfp_ctx* ctx, hipStream_t stream, const WitnessView& w, const uint32_t* work_d, const uint32_t* count_d,
                                uint32_t bound, uint32_t kinds, void* recs_d);
int launch_rebase_claims(ipcfp_ctx* ctx, void* claims_d, uint32_t n, uint64_t base, uint64_t blob_len, uint64_t full_len,
                         uint32_t* miss_d, uint32_t* order_d = nullptr, uint64_t key_lo = 0, uint64_t key_hi = ~0ull);  // claims_compact.hip
// --- shard_pull.hip (shard_pull.h) --- a rank pulls its shard out of a bundle in host memory, level by level
struct PullSeeds;
struct PullFrontier;
struct PullTables;
struct PullCtl;
int launch_pull_seed(ipcfp_ctx* ctx, const WitnessView& w, const PullSeeds& seeds, const PullFrontier& first, PullCtl* ctl_d);
int launch_pull_round(ipcfp_ctx* ctx, const WitnessView& w, const uint8_t* host_bytes_dev, const PullTables& t, const PullFrontier& cur,
                      uint32_t n_items, const PullFrontier& next, PullCtl* ctl_d, uint32_t n_shards, uint32_t shard,
                      unsigned long long* mailbox_dev, unsigned long long seq);
// --- cid_index.hip --- the CID → block table over any (cids, n): slots_d holds mask + 1 words, cleared to 0xff by the caller
int launch_index_insert(ipcfp_ctx* ctx, const uint8_t* cids_d, uint32_t n, uint32_t* slots_d, uint32_t mask);

// device view of a witness (host helper, witness.cpp)
WitnessView witness_view(const ipcfp_witness* w, uint32_t* touched_bits = nullptr);

}  // namespace ipcfp
