// csrc/host/pack_claims.h — host-side lowering of EventProof structs to the packed ABI form (pack_claims.cpp).
#pragma once
#include <memory>
#include <string>
#include <vector>

#include "../common.h"
#include "../kernels/claims_dev.h"

namespace ipcfp {

struct PackedEvents {
    std::vector<ipcfp_tipset_ref_t> tipsets;
    // parents beyond the inline IPCFP_MAX_PARENTS of the tipsets that have them (ipcfp_tipset_ref_t::more_parents points here)
    std::vector<std::unique_ptr<std::vector<uint8_t>>> more_parents;
    std::vector<EventClaimPacked> claims;  // layout == ipcfp_event_claim_t
    std::vector<uint8_t> blob;
};

// IPCFP_OK, or IPCFP_E_UNSUPPORTED with `err` set (a proof names more than IPCFP_MAX_PARENTS_WIDE parents; blob ≥ 3.75 GB)
int pack_event_claims_host(const ipcfp_event_proof_t* proofs, uint64_t n, PackedEvents& out, std::string& err);

}  // namespace ipcfp
