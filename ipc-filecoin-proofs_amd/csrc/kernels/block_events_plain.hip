// csrc/kernels/block_events_plain.hip — k_block_events, per-lane variant, plain reader (see block_events_lane.inc)
#define BLOCK_EVENTS_KERNEL k_block_events_plain
#define BLOCK_EVENTS_LAUNCH launch_block_events_plain
#include "block_events_lane.inc"
