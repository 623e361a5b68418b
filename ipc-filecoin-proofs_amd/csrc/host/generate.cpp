// csrc/host/generate.cpp — generator-side entry points: the device does every
// block load, hash and walk of `generate_event_proof` / `generate_storage_proof`; the host only sorts
// the recorded CIDs.
//
//   ipcfp_generate_event_proofs    src/proofs/events/generator.rs:75-178  (+ :180-307 via the scan)
//   ipcfp_generate_storage_proofs  src/proofs/storage/generator.rs:29-69
//
// The reference runs against an RPC blockstore and records every block it loads in a
// `RecordingBlockStore`; here the blockstore is the HBM-resident witness (the blocks a
// caller fetched for the tipset) and the recorder is a bitmap over its blocks.  The materialised
// witness is returned as block ids in `Cid: Ord` order (`collect_witness_blocks`,
// src/proofs/common/witness.rs:34-54 iterates a BTreeSet<Cid>).
#include <algorithm>
#include <cstring>
#include <vector>

#include "../common.h"
#include "../kernels/amt_enum.h"
#include "../kernels/claims_dev.h"
#include "../kernels/launch.h"
#include "exec_state.h"
#include "tipset_wide.h"

using namespace ipcfp;

namespace {

struct CidOrd {  // the fields `#[derive(Ord)]` compares: version, codec, multihash {code, size, digest}
    uint64_t version = 0, codec = 0, mh_code = 0, mh_size = 0;
    const uint8_t* digest = nullptr;
    bool ok = false;
};

bool varint(const uint8_t* p, size_t n, size_t& pos, uint64_t& v) {
    v = 0;
    for (int i = 0; i < 9 && pos < n; ++i) {
        const uint8_t b = p[pos++];
        v |= uint64_t(b & 0x7f) << (7 * i);
        if (!(b & 0x80)) return true;
    }
    return false;
}

CidOrd cid_ord(const uint8_t* slot) {
    CidOrd o;
    if (slot[0] == 0x12 && slot[1] == 0x20) {  // CIDv0: bare sha2-256 multihash
        o.version = 0, o.codec = 0x70, o.mh_code = 0x12, o.mh_size = 32, o.digest = slot + 2, o.ok = true;
        return o;
    }
    size_t pos = 0;
    if (!varint(slot, IPCFP_CID_SLOT, pos, o.version) || !varint(slot, IPCFP_CID_SLOT, pos, o.codec) ||
        !varint(slot, IPCFP_CID_SLOT, pos, o.mh_code) || !varint(slot, IPCFP_CID_SLOT, pos, o.mh_size))
        return o;
    if (pos + o.mh_size > IPCFP_CID_SLOT) return o;
    o.digest = slot + pos;
    o.ok = true;
    return o;
}

bool cid_slot_less(const uint8_t* a, const uint8_t* b) {
    const CidOrd x = cid_ord(a), y = cid_ord(b);
    if (!x.ok || !y.ok) return std::memcmp(a, b, IPCFP_CID_SLOT) < 0;
    if (x.version != y.version) return x.version < y.version;
    if (x.codec != y.codec) return x.codec < y.codec;
    if (x.mh_code != y.mh_code) return x.mh_code < y.mh_code;
    if (x.mh_size != y.mh_size) return x.mh_size < y.mh_size;
    return std::memcmp(x.digest, y.digest, x.mh_size) < 0;
}

// recorded bitmap → block ids in `Cid: Ord` order (+ their CIDs)
int materialize(ipcfp_ctx* ctx, ipcfp_witness* w, const uint32_t* touched_d, uint32_t* ids_out, uint8_t* cids_out,
                uint64_t cap, uint64_t* n_out) {
    const uint32_t words = div_up(uint32_t(w->n), 32);
    std::vector<uint32_t> bits(words);
    IPCFP_HIP(ctx, hipMemcpyAsync(bits.data(), touched_d, size_t(words) * 4, hipMemcpyDeviceToHost, ctx->stream));
    IPCFP_HIP(ctx, sync_stream(ctx, ctx->stream));
    std::vector<uint32_t> ids;
    for (uint32_t wd = 0; wd < words; ++wd) {
        uint32_t m = bits[wd];
        while (m) {
            const int b = __builtin_ctz(m);
            m &= m - 1;
            ids.push_back(wd * 32 + uint32_t(b));
        }
    }
    const uint32_t n = uint32_t(ids.size());
    *n_out = n;
    if (n == 0) return IPCFP_OK;
    DevBuf<uint32_t> ids_d;
    DevBuf<CidKey> keys_d;
    IPCFP_HIP(ctx, ids_d.alloc(n));
    IPCFP_HIP(ctx, keys_d.alloc(n));
    IPCFP_HIP(ctx, hipMemcpyAsync(ids_d.p, ids.data(), size_t(n) * 4, hipMemcpyHostToDevice, ctx->stream));
    int rc = launch_gather_block_cids(ctx, w->cids.p, ids_d.p, n, keys_d.p);
    if (rc) return rc;
    std::vector<uint8_t> cids(size_t(n) * IPCFP_CID_SLOT);
    IPCFP_HIP(ctx, hipMemcpyAsync(cids.data(), keys_d.p, cids.size(), hipMemcpyDeviceToHost, ctx->stream));
    IPCFP_HIP(ctx, sync_stream(ctx, ctx->stream));
    std::vector<uint32_t> perm(n);
    for (uint32_t i = 0; i < n; ++i) perm[i] = i;
    // a witness may hold the same CID twice (last one wins in the index): only the indexed copy can be
    // marked, so CIDs are distinct here and the order is total
    std::sort(perm.begin(), perm.end(), [&](uint32_t a, uint32_t b) {
        return cid_slot_less(cids.data() + size_t(a) * IPCFP_CID_SLOT, cids.data() + size_t(b) * IPCFP_CID_SLOT);
    });
    const uint64_t take = n < cap ? n : cap;
    for (uint64_t i = 0; i < take; ++i) {
        if (ids_out) ids_out[i] = ids[perm[i]];
        if (cids_out) std::memcpy(cids_out + i * IPCFP_CID_SLOT, cids.data() + size_t(perm[i]) * IPCFP_CID_SLOT, IPCFP_CID_SLOT);
    }
    return IPCFP_OK;
}

struct StorageSpecHost {
    uint64_t actor_id;
    uint8_t slot[32];
};
struct StorageGenHost {
    uint8_t parent_state_root[40], actor_state[40], storage_root[40];
    uint8_t value[32];
    uint32_t status, pad;
};
static_assert(sizeof(StorageGenHost) == sizeof(ipcfp_generated_storage_t), "generated storage record layout");

}  // namespace

extern "C" {

int ipcfp_generate_event_proofs(ipcfp_ctx_t* ctx, ipcfp_witness_t* w, const uint8_t* parent_cids40, uint32_t n_parents,
                                const uint8_t* child_cid40, const ipcfp_event_filter_t* filter, int has_actor,
                                uint64_t actor, ipcfp_status_t* status_out, ipcfp_event_match_t* matches,
                                uint8_t* message_cids40, uint64_t cap_proofs, uint64_t* n_proofs,
                                uint32_t* witness_block_ids, uint8_t* witness_cids40, uint64_t cap_blocks,
                                uint64_t* n_blocks) {
    if (!ctx || !w || w->ctx != ctx || !child_cid40 || !filter || !status_out || !n_proofs || !n_blocks ||
        (n_parents && !parent_cids40))
        return IPCFP_E_INVALID;
    IPCFP_ENTER(ctx);
    *n_proofs = *n_blocks = 0;
    *status_out = IPCFP_ST_ERR;

    const uint32_t words = div_up(uint32_t(w->n), 32);
    DevBuf<uint32_t> touched;
    IPCFP_HIP(ctx, touched.alloc(words + 2));
    IPCFP_HIP(ctx, hipMemsetAsync(touched.p, 0, size_t(words + 2) * 4, ctx->stream));
    uint32_t* oor_d = touched.p + words;          // an exec_index is outside the execution order
    uint32_t* missing_d = touched.p + words + 1;  // a base CID is absent from the store
    const WitnessView rec = witness_view(w, touched.p);

    // Step 1 (generator.rs:89-95): child header → receipts root.  The context kernel also loads parent 0.
    TipsetCtxDev tc;
    WideParents wide;
    if (int rc_t = tipset_inputs_list(ctx, TC_PARENTS_PARSED | TC_CHILD_PARSED, parent_cids40, n_parents, child_cid40, tc, wide)) return rc_t;
    DevBuf<TipsetCtxDev> tc_d;
    IPCFP_HIP(ctx, tc_d.alloc(1));
    IPCFP_HIP(ctx, hipMemcpyAsync(tc_d.p, &tc, sizeof tc, hipMemcpyHostToDevice, ctx->stream));
    int rc = launch_ctx_headers(ctx, rec, tc_d.p, 1);
    if (rc) return rc;
    IPCFP_HIP(ctx, d2h_small(ctx, &tc, tc_d.p, sizeof tc, ctx->stream));
    IPCFP_HIP(ctx, sync_stream(ctx, ctx->stream));
    if (tc.child_status != IPCFP_ST_TRUE) {
        *status_out = ipcfp_status_t(tc.child_status);
        return IPCFP_OK;
    }
    // Steps 2-4 (generator.rs:97-135): parent headers, TxMeta, the message AMTs (recorded) and the
    // execution order with verify_txmeta = false.  One traversal serves record_transaction_amts and
    // build_execution_order: both load exactly the same blocks in the same order.
    ExecState ex;
    rc = build_exec_order(ctx, rec, tc_d.p, n_parents, ex, /*verify_txmeta=*/0);
    if (rc) return rc;
    if (ex.status != IPCFP_ST_TRUE) {
        *status_out = ipcfp_status_t(ex.status);
        return IPCFP_OK;
    }
    // Step 5 (generator.rs:137-150): two-pass scan, recording
    ScanResult scan;
    rc = scan_events_device(ctx, w, tc.receipts_root, *filter, has_actor, actor, touched.p, scan);
    if (rc) return rc;
    if (scan.status != IPCFP_ST_TRUE) {
        *status_out = ipcfp_status_t(scan.status);
        return IPCFP_OK;
    }
    // Step 6 (generator.rs:152-169): message CID of each match = exec_list[exec_index]
    const uint64_t nm = scan.n_matches;
    if (nm >= 0xffffffffULL) return set_error(ctx, IPCFP_E_UNSUPPORTED, "too many matches");
    DevBuf<CidKey> exec_list, msg;
    DevBuf<uint64_t> exec_idx;
    if (nm) {
        IPCFP_HIP(ctx, exec_list.alloc(ex.exec_len ? ex.exec_len : 1));
        IPCFP_HIP(ctx, msg.alloc(nm));
        rc = launch_exec_compact(ctx, ex.keys.p, uint32_t(ex.raw_len), ex.first.p, ex.pos.p, exec_list.p);
        if (rc) return rc;
        std::vector<ipcfp_event_match_t> mh(nm);
        IPCFP_HIP(ctx, hipMemcpyAsync(mh.data(), scan.matches.p, nm * sizeof(ipcfp_event_match_t), hipMemcpyDeviceToHost,
                                      ctx->stream));
        IPCFP_HIP(ctx, sync_stream(ctx, ctx->stream));
        std::vector<uint64_t> idx(nm);
        for (uint64_t i = 0; i < nm; ++i) idx[i] = mh[i].exec_index;
        IPCFP_HIP(ctx, exec_idx.alloc(nm));
        IPCFP_HIP(ctx, hipMemcpyAsync(exec_idx.p, idx.data(), nm * 8, hipMemcpyHostToDevice, ctx->stream));
        rc = launch_gather_keys(ctx, exec_list.p, ex.exec_len, exec_idx.p, uint32_t(nm), msg.p, oor_d);
        if (rc) return rc;
        const uint64_t take = nm < cap_proofs ? nm : cap_proofs;
        if (matches) std::memcpy(matches, mh.data(), take * sizeof(ipcfp_event_match_t));
        if (message_cids40)
            IPCFP_HIP(ctx, hipMemcpyAsync(message_cids40, msg.p, take * IPCFP_CID_SLOT, hipMemcpyDeviceToHost, ctx->stream));
    }
    // base witness (generator.rs:97-112): parents, child, receipts root (TxMeta CIDs were marked by the traversal)
    std::vector<CidKey> base;
    tipset_parent_keys(tc, wide, base);
    base.push_back(tc.child);
    base.push_back(tc.receipts_root);
    DevBuf<CidKey> base_d;
    IPCFP_HIP(ctx, base_d.alloc(base.size()));
    IPCFP_HIP(ctx, hipMemcpyAsync(base_d.p, base.data(), base.size() * sizeof(CidKey), hipMemcpyHostToDevice, ctx->stream));
    rc = launch_mark_cids(ctx, rec, base_d.p, uint32_t(base.size()), missing_d);
    if (rc) return rc;
    uint32_t flag[2] = {0, 0};
    IPCFP_HIP(ctx, d2h_small(ctx, flag, oor_d, 8, ctx->stream));
    IPCFP_HIP(ctx, sync_stream(ctx, ctx->stream));
    if (flag[0]) {  // "Missing message at index" (generator.rs:158-160) precedes materialisation
        *status_out = IPCFP_ST_ERR;
        return IPCFP_OK;
    }
    if (flag[1]) {  // must_get of a base CID fails (witness.rs:45-48)
        *status_out = IPCFP_ST_ERR_MISSING_BLOCK;
        return IPCFP_OK;
    }
    // Step 7 (generator.rs:171-177): materialise in BTreeSet order
    rc = materialize(ctx, w, touched.p, witness_block_ids, witness_cids40, cap_blocks, n_blocks);
    if (rc) return rc;
    *n_proofs = nm;
    *status_out = IPCFP_ST_TRUE;
    return IPCFP_OK;
}

int ipcfp_generate_storage_proofs(ipcfp_ctx_t* ctx, ipcfp_witness_t* w, const uint8_t* child_cid40,
                                  const uint64_t* actor_ids, const uint8_t* slots32, uint64_t n,
                                  ipcfp_generated_storage_t* out, uint32_t* witness_block_ids, uint8_t* witness_cids40,
                                  uint64_t cap_blocks, uint64_t* n_blocks) {
    if (!ctx || !w || w->ctx != ctx || !child_cid40 || !n_blocks || (n && (!actor_ids || !slots32 || !out)))
        return IPCFP_E_INVALID;
    if (n >= 0xffffffffULL) return set_error(ctx, IPCFP_E_UNSUPPORTED, "batch too large");
    IPCFP_ENTER(ctx);
    *n_blocks = 0;
    if (n == 0) return IPCFP_OK;
    const uint32_t words = div_up(uint32_t(w->n), 32);
    DevBuf<uint32_t> touched;
    IPCFP_HIP(ctx, touched.alloc(words));
    IPCFP_HIP(ctx, hipMemsetAsync(touched.p, 0, size_t(words) * 4, ctx->stream));
    const WitnessView rec = witness_view(w, touched.p);
    std::vector<StorageSpecHost> specs(n);
    for (uint64_t i = 0; i < n; ++i) {
        specs[i].actor_id = actor_ids[i];
        std::memcpy(specs[i].slot, slots32 + i * 32, 32);
    }
    DevBuf<StorageSpecHost> specs_d;
    DevBuf<StorageGenHost> out_d;
    IPCFP_HIP(ctx, specs_d.alloc(n));
    IPCFP_HIP(ctx, out_d.alloc(n));
    IPCFP_HIP(ctx, hipMemcpyAsync(specs_d.p, specs.data(), n * sizeof(StorageSpecHost), hipMemcpyHostToDevice, ctx->stream));
    int rc = launch_generate_storage(ctx, rec, key_from_slot(child_cid40), specs_d.p, uint32_t(n), out_d.p);
    if (rc) return rc;
    IPCFP_HIP(ctx, hipMemcpyAsync(out, out_d.p, n * sizeof(StorageGenHost), hipMemcpyDeviceToHost, ctx->stream));
    IPCFP_HIP(ctx, sync_stream(ctx, ctx->stream));
    // the bundle's witness is the union over the proofs that succeeded; a failing spec aborts
    // generate_proof_bundle (src/proofs/generator.rs:42-49), which the caller sees in out[i].status
    return materialize(ctx, w, touched.p, witness_block_ids, witness_cids40, cap_blocks, n_blocks);
}

int ipcfp_generate_proof_bundle(ipcfp_ctx_t* ctx, ipcfp_witness_t* w, const uint8_t* parent_cids40, uint32_t n_parents,
                                const uint8_t* child_cid40, const ipcfp_storage_proof_spec_t* storage_specs,
                                uint64_t n_storage, const ipcfp_event_proof_spec_t* event_specs, uint64_t n_events,
                                ipcfp_generated_storage_t* storage_out, ipcfp_status_t* event_status,
                                ipcfp_event_match_t* matches, uint8_t* message_cids40, uint32_t* match_spec,
                                uint64_t cap_proofs, uint64_t* n_proofs, uint32_t* witness_block_ids,
                                uint8_t* witness_cids40, uint64_t cap_blocks, uint64_t* n_blocks, uint64_t* first_error) {
    if (!ctx || !w || w->ctx != ctx || !child_cid40 || !n_proofs || !n_blocks || !first_error ||
        (n_storage && (!storage_specs || !storage_out)) || (n_events && (!event_specs || !event_status)))
        return IPCFP_E_INVALID;
    *n_proofs = *n_blocks = 0;
    *first_error = ~0ULL;
    std::vector<uint32_t> all_ids;  // the union, de-duplicated below (BTreeSet<(Cid, Vec<u8>)>, generator.rs:34)
    // ---- storage specs (generator.rs:42-56) ----
    if (n_storage) {
        std::vector<uint64_t> actors(n_storage);
        std::vector<uint8_t> slots(n_storage * 32);
        for (uint64_t i = 0; i < n_storage; ++i) {
            actors[i] = storage_specs[i].actor_id;
            std::memcpy(slots.data() + 32 * i, storage_specs[i].slot, 32);
        }
        std::vector<uint32_t> ids(w->n ? w->n : 1);
        uint64_t nb = 0;
        int rc = ipcfp_generate_storage_proofs(ctx, w, child_cid40, actors.data(), slots.data(), n_storage, storage_out,
                                               ids.data(), nullptr, ids.size(), &nb);
        if (rc) return rc;
        for (uint64_t i = 0; i < n_storage; ++i)
            if (storage_out[i].status != IPCFP_ST_TRUE) {
                *first_error = i;
                return IPCFP_OK;  // `generate_storage_proof(..).await?` (generator.rs:48-49)
            }
        all_ids.assign(ids.begin(), ids.begin() + nb);
    }
    // ---- event specs (generator.rs:59-80) ----
    uint64_t np = 0;
    for (uint64_t j = 0; j < n_events; ++j) {
        const ipcfp_event_proof_spec_t& sp = event_specs[j];
        ipcfp_event_filter_t filter;
        int rc = ipcfp_create_event_filter(ctx, sp.event_signature ? sp.event_signature : "", sp.topic_1 ? sp.topic_1 : "", &filter);
        if (rc) return rc;
        // size, then fill
        ipcfp_status_t st = IPCFP_ST_ERR;
        uint64_t n_p = 0, n_b = 0;
        rc = ipcfp_generate_event_proofs(ctx, w, parent_cids40, n_parents, child_cid40, &filter, sp.has_actor_id_filter ? 1 : 0,
                                         sp.actor_id_filter, &st, nullptr, nullptr, 0, &n_p, nullptr, nullptr, 0, &n_b);
        if (rc) return rc;
        event_status[j] = st;
        if (st != IPCFP_ST_TRUE) {
            *first_error = n_storage + j;
            return IPCFP_OK;  // `generate_event_proof(..).await?` (generator.rs:65-74)
        }
        std::vector<ipcfp_event_match_t> m(n_p ? n_p : 1);
        std::vector<uint8_t> mc((n_p ? n_p : 1) * IPCFP_CID_SLOT);
        std::vector<uint32_t> ids(n_b ? n_b : 1);
        rc = ipcfp_generate_event_proofs(ctx, w, parent_cids40, n_parents, child_cid40, &filter, sp.has_actor_id_filter ? 1 : 0,
                                         sp.actor_id_filter, &st, m.data(), mc.data(), n_p, &n_p, ids.data(), nullptr, n_b, &n_b);
        if (rc) return rc;
        for (uint64_t k = 0; k < n_p; ++k, ++np) {
            if (np >= cap_proofs) continue;
            if (matches) matches[np] = m[k];
            if (message_cids40) std::memcpy(message_cids40 + np * IPCFP_CID_SLOT, mc.data() + k * IPCFP_CID_SLOT, IPCFP_CID_SLOT);
            if (match_spec) match_spec[np] = uint32_t(j);
        }
        all_ids.insert(all_ids.end(), ids.begin(), ids.begin() + n_b);
    }
    *n_proofs = np;
    // ---- the union in `Cid: Ord` order, each block once (generator.rs:84-88) ----
    std::sort(all_ids.begin(), all_ids.end());
    all_ids.erase(std::unique(all_ids.begin(), all_ids.end()), all_ids.end());
    const uint32_t n = uint32_t(all_ids.size());
    *n_blocks = n;
    if (n == 0) return IPCFP_OK;
    IPCFP_ENTER(ctx);
    DevBuf<uint32_t> ids_d;
    DevBuf<CidKey> keys_d;
    IPCFP_HIP(ctx, ids_d.alloc(n));
    IPCFP_HIP(ctx, keys_d.alloc(n));
    IPCFP_HIP(ctx, hipMemcpyAsync(ids_d.p, all_ids.data(), size_t(n) * 4, hipMemcpyHostToDevice, ctx->stream));
    int rc = launch_gather_block_cids(ctx, w->cids.p, ids_d.p, n, keys_d.p);
    if (rc) return rc;
    std::vector<uint8_t> cids(size_t(n) * IPCFP_CID_SLOT);
    IPCFP_HIP(ctx, hipMemcpyAsync(cids.data(), keys_d.p, cids.size(), hipMemcpyDeviceToHost, ctx->stream));
    IPCFP_HIP(ctx, sync_stream(ctx, ctx->stream));
    std::vector<uint32_t> perm(n);
    for (uint32_t i = 0; i < n; ++i) perm[i] = i;
    std::sort(perm.begin(), perm.end(), [&](uint32_t a, uint32_t b) {
        return cid_slot_less(cids.data() + size_t(a) * IPCFP_CID_SLOT, cids.data() + size_t(b) * IPCFP_CID_SLOT);
    });
    for (uint64_t i = 0; i < n && i < cap_blocks; ++i) {
        if (witness_block_ids) witness_block_ids[i] = all_ids[perm[i]];
        if (witness_cids40) std::memcpy(witness_cids40 + i * IPCFP_CID_SLOT, cids.data() + size_t(perm[i]) * IPCFP_CID_SLOT, IPCFP_CID_SLOT);
    }
    return IPCFP_OK;
}

}  // extern "C"
