/* Test infrastructure: print the NATIVE call stack when the process receives SIGABRT, then hand over to the handler
 * that was installed before (pytest's faulthandler, which prints the Python stack).  A runtime library that calls
 * abort() without a message otherwise leaves no trace of who did.  Built and loaded by tests/conftest.py. */
#include <execinfo.h>
#include <signal.h>
#include <string.h>
#include <unistd.h>

static struct sigaction previous;
static int out_fd = 2;

static void on_abort(int sig, siginfo_t *info, void *uc) {
    static const char msg[] = "\n[abort_trace] SIGABRT -- native backtrace:\n";
    void *bt[96];
    int n;
    (void)!write(out_fd, msg, sizeof(msg) - 1);
    n = backtrace(bt, 96);
    backtrace_symbols_fd(bt, n, out_fd);
    if (previous.sa_flags & SA_SIGINFO) {
        if (previous.sa_sigaction) previous.sa_sigaction(sig, info, uc);
    } else if (previous.sa_handler && previous.sa_handler != SIG_DFL && previous.sa_handler != SIG_IGN) {
        previous.sa_handler(sig);
    }
    signal(sig, SIG_DFL);
    raise(sig);
}

/* fd: where to write (pytest redirects fd 2 while a test runs; it keeps a copy of the real one) */
void abort_trace_install(int fd) {
    struct sigaction sa;
    out_fd = fd;
    memset(&sa, 0, sizeof(sa));
    sa.sa_sigaction = on_abort;
    sa.sa_flags = SA_SIGINFO | SA_NODEFER;
    sigemptyset(&sa.sa_mask);
    sigaction(SIGABRT, &sa, &previous);
}
