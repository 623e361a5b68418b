// tests/native/parallel_harness.cpp — csrc/host/parallel.h run_parts, driven from tests/test_host_parallel.py.
#include <atomic>
#include <cstdint>
#include <new>
#include <stdexcept>

#include "../../ipc-filecoin-proofs_amd/csrc/host/parallel.h"

extern "C" {

// every part t of [0, parts) adds 1 << (t % 60) into sums[t % 4] and counts itself; part `throws_at` (if < parts) throws
// std::bad_alloc after counting.  Returns run_parts' verdict; *ran = parts that ran.
int parts_run(unsigned parts, unsigned throws_at, uint64_t* ran, uint64_t* mask_lo) {
    std::atomic<uint64_t> count{0}, mask{0};
    const bool ok = ipcfp::run_parts(parts, [&](unsigned t) {
        count.fetch_add(1);
        if (t < 64) mask.fetch_or(1ull << t);
        if (t == throws_at) throw std::bad_alloc();
    });
    *ran = count.load();
    *mask_lo = mask.load();
    return ok ? 1 : 0;
}

unsigned parts_max(void) { return ipcfp::kMaxParts; }

}  // extern "C"
