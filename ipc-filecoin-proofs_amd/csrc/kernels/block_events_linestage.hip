// csrc/kernels/block_events_linestage.hip — k_block_events, per-lane variant, the reader staging each 128-byte line in
// the lane's LDS slot (cbor_dev.h IPCFP_LINE_STAGE; see block_events_lane.inc)
#define IPCFP_LINE_STAGE 1
#define BLOCK_EVENTS_KERNEL k_block_events_linestage
#define BLOCK_EVENTS_LAUNCH launch_block_events_linestage
#include "block_events_lane.inc"
