// csrc/host/verify_events.cpp — placeholder until the event path lands (next commit).
#include "../common.h"
using namespace ipcfp;
extern "C" int ipcfp_verify_event_proofs(ipcfp_ctx_t* ctx, ipcfp_witness_t*, const ipcfp_event_proof_t*, uint64_t,
                                         const ipcfp_trust_policy_t*, const ipcfp_event_filter_t*, ipcfp_status_t*) {
    return set_error(ctx, IPCFP_E_UNSUPPORTED, "ipcfp_verify_event_proofs: not built yet");
}
