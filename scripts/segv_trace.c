/* Developer aid: LD_PRELOAD this to get a C-level backtrace of the faulting thread on SIGSEGV / SIGBUS / SIGABRT
 *   gcc -shared -fPIC -o scripts/segv_trace.so scripts/segv_trace.c */
#define _GNU_SOURCE
#include <execinfo.h>
#include <signal.h>
#include <string.h>
#include <unistd.h>

static void handler(int sig, siginfo_t* si, void* uc)
{
    void* frames[64];
    const char msg[] = "\n=== segv_trace: fatal signal, backtrace of the faulting thread ===\n";
    (void)!write(2, msg, sizeof(msg) - 1);
    int n = backtrace(frames, 64);
    backtrace_symbols_fd(frames, n, 2);
    signal(sig, SIG_DFL);
    raise(sig);
}

__attribute__((constructor)) static void install(void)
{
    struct sigaction sa;
    memset(&sa, 0, sizeof(sa));
    sa.sa_sigaction = handler;
    sa.sa_flags = SA_SIGINFO | SA_ONSTACK;
    sigaction(SIGSEGV, &sa, 0);
    sigaction(SIGBUS, &sa, 0);
    sigaction(SIGABRT, &sa, 0);
}
