// csrc/kernels/events_dev.h — StampedEvent decode, `extract_evm_log` and `matches_log` on the device: the walk
// primitives plus event_log_dev.h (which only needs the CBOR reader, so that a unit whose readers sit in LDS can
// include it alone).
#pragma once
#include "event_log_dev.h"
#include "walk_dev.h"
