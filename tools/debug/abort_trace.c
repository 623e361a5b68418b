/* tools/debug/abort_trace.c — a SIGABRT handler that writes the aborting thread's native stack to stderr before the default
 * action (pytest -p tools.debug.abort_trace_plugin).  Debugging aid for an abort() raised inside a runtime library without a
 * message; not part of the product. */
#include <execinfo.h>
#include <signal.h>
#include <string.h>
#include <unistd.h>
static int out_fd = 2; /* a duplicate of stderr made at install time: pytest's fd capture re-points fd 2 during a test */
static void on_abort(int sig) {
    void* frames[96];
    int n = backtrace(frames, 96);
    static const char head[] = "\n== SIGABRT: native stack of the aborting thread ==\n";
    (void)!write(out_fd, head, sizeof head - 1);
    backtrace_symbols_fd(frames, n, out_fd);
    signal(sig, SIG_DFL);
    raise(sig);
}
void ipcfp_debug_install_abort_trace(void) {
    void* warm[2];
    backtrace(warm, 2); /* (loads libgcc now, not inside the handler) */
    signal(SIGABRT, on_abort);
}
void ipcfp_debug_abort_trace_keep_stderr(void) { /* call before anything re-points fd 2 */
    int d = dup(2);
    if (d >= 0) out_fd = d;
}
