// csrc/host/pack_claims.h — host-side lowering of EventProof structs to the packed ABI form (pack_claims.cpp).
#pragma once
#include <string>
#include <vector>

#include "../common.h"
#include "../kernels/claims_dev.h"

namespace ipcfp {

struct PackedEvents {
    std::vector<ipcfp_tipset_ref_t> tipsets;
    std::vector<EventClaimPacked> claims;  // layout == ipcfp_event_claim_t
    std::vector<uint8_t> blob;
};

// IPCFP_OK, or IPCFP_E_UNSUPPORTED with `err` set (a proof names more than 16 parents; blob ≥ 3.75 GB)
int pack_event_claims_host(const ipcfp_event_proof_t* proofs, uint64_t n, PackedEvents& out, std::string& err);

}  // namespace ipcfp
