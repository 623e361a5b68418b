// csrc/kernels/exec_order.h — device state of one tipset context: everything `verify_single_proof`
// derives from (parent_tipset_cids, child_block_cid) alone — header consistency facts
// (src/proofs/events/verifier.rs:147-181) and the reconstructed execution order
// (src/proofs/events/utils.rs:16-30,48-94) — computed ONCE per distinct pair instead of once per
// proof (the reference recomputes it for every proof: events/verifier.rs:190).
#pragma once
#include <cstdint>

#include "amt_enum.h"
#include "event_table.h"
#include "tipset_ctx.h"
#include "types_dev.h"

namespace ipcfp {

}  // namespace ipcfp
