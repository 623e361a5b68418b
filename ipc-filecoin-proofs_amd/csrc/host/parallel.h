// csrc/host/parallel.h — a few parts of one job on threads of their own, behind a C ABI: nothing may be thrown across it.
// std::thread's constructor throws std::system_error when a thread cannot be made (a pids limit of the container), and a
// part that grows a vector can throw std::bad_alloc on a thread, where nobody catches it (std::terminate): here a part
// without a thread runs on the caller's, and a part that throws is reported, not propagated.
#pragma once
#include <atomic>
#include <thread>

namespace ipcfp {

constexpr unsigned kMaxParts = 32;

// work(t) for t in [0, parts): part 0 on the calling thread, the others on threads of their own where those can be made.
// false: a part threw (the job's output is not to be used).  parts ≤ kMaxParts (more: the rest run on the calling thread).
template <typename F>
inline bool run_parts(unsigned parts, F&& work) {
    std::atomic<bool> threw{false};
    auto guarded = [&](unsigned t) {
        try {
            work(t);
        } catch (...) {
            threw = true;
        }
    };
    std::thread pool[kMaxParts];
    unsigned started = 1;
    for (; started < parts && started < kMaxParts; ++started) {
        try {
            pool[started] = std::thread(guarded, started);
        } catch (...) {
            break;
        }
    }
    if (parts) guarded(0);
    for (unsigned t = started; t < parts; ++t) guarded(t);
    for (unsigned t = 1; t < started; ++t) pool[t].join();
    return !threw;
}

}  // namespace ipcfp
