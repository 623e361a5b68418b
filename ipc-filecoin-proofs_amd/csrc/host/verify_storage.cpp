// csrc/host/verify_storage.cpp — `verify_storage_proof` batches and `create_event_filter`.
//
// Host side of src/proofs/storage/verifier.rs:24-63: parse the claim strings once
// (parse_cid → src/proofs/common/witness.rs:60-64; hex → storage/verifier.rs:155-157), upload the
// packed claims, run one kernel over the batch, download the status bytes.
#include <cstring>
#include <vector>

#include "../common.h"
#include "../kernels/claims_dev.h"
#include "../kernels/launch.h"
#include "cidstr.h"

using namespace ipcfp;

namespace ipcfp {

CidKey key_from_slot(const uint8_t* slot40);

// Parse a CID string into a witness key.  `parsed`: Cid::try_from succeeded.  `canonical`: the
// string equals Cid::to_string() of what it parses to.  CIDs longer than the 40-byte slot parse
// fine but can never be witness keys: they get the impossible key.
void parse_cid_claim(const char* s, CidKey& key, bool& parsed, bool& canonical) {
    std::vector<uint8_t> bin;
    parsed = cid_from_string(s, bin);
    canonical = false;
    for (auto& w : key.w) w = ~0ULL;
    if (!parsed) return;
    canonical = cid_to_string(bin.data(), bin.size()) == s;
    if (bin.size() <= IPCFP_CID_SLOT) {
        uint8_t slot[IPCFP_CID_SLOT] = {0};
        std::memcpy(slot, bin.data(), bin.size());
        std::memcpy(key.w, slot, IPCFP_CID_SLOT);
    }
}

static const ipcfp_trust_policy_t kAcceptAll = {0, 0, 0, 0};

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
    int rc = launch_verify_storage(ctx, witness_view(w), cd.p, uint32_t(n), trust ? *trust : kAcceptAll, sd.p);
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
    int rc = launch_verify_storage(ctx, witness_view(w), static_cast<const StorageClaimPacked*>(claims_d), uint32_t(n),
                                   trust ? *trust : kAcceptAll, static_cast<uint8_t*>(status_d));
    if (rc) return rc;
    IPCFP_HIP(ctx, sync_stream(ctx, ctx->stream));
    return IPCFP_OK;
}

int ipcfp_cid_from_string(const char* s, uint8_t out40[IPCFP_CID_SLOT]) {
    if (!s || !out40) return IPCFP_E_INVALID;
    std::vector<uint8_t> bin;
    if (!cid_from_string(s, bin)) return IPCFP_E_PARSE;
    if (bin.size() > IPCFP_CID_SLOT) return IPCFP_E_UNSUPPORTED;
    std::memset(out40, 0, IPCFP_CID_SLOT);
    std::memcpy(out40, bin.data(), bin.size());
    return int(bin.size());
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
