/* LD_PRELOAD aid for ONE diagnostic run (profiles/r06/pmc_sigsegv.txt): prints the native backtrace of a SIGSEGV / SIGBUS / SIGABRT to
 * stderr (module + offset per frame, glibc backtrace_symbols_fd: no allocation), then hands the signal back to whoever had it before.
 *   gcc -O1 -g -shared -fPIC -o scratch/libsegvbt.so tools/segv_backtrace.c
 *   LD_PRELOAD=$PWD/scratch/libsegvbt.so PYTHONFAULTHANDLER=1 rocprofv3 ... -- python3 bench.py ...
 * The handler is installed when the library is loaded AND again at the first HIP-free moment python gives us (a constructor runs before
 * the profiler's tool library installs its own handlers; the second installation is done from bench.py through segvbt_install()). */
#define _GNU_SOURCE
#include <execinfo.h>
#include <signal.h>
#include <stdio.h>
#include <string.h>
#include <unistd.h>

static struct sigaction old_segv, old_bus, old_abrt;
static char altstack[1 << 16];

static void say(const char* s) { ssize_t r = write(2, s, strlen(s)); (void)r; }

static void handler(int sig, siginfo_t* info, void* uctx) {
    (void)uctx;
    char line[128];
    snprintf(line, sizeof line, "\n=== segv_backtrace: signal %d, fault address %p ===\n", sig, info ? info->si_addr : (void*)0);
    say(line);
    void* frames[96];
    int n = backtrace(frames, 96);
    backtrace_symbols_fd(frames, n, 2);
    say("=== end of native backtrace ===\n");
    struct sigaction* old = sig == SIGSEGV ? &old_segv : (sig == SIGBUS ? &old_bus : &old_abrt);
    sigaction(sig, old, 0);                       /* previous owner (python's faulthandler, the profiler, or the default action) */
    raise(sig);
}

void segvbt_install(void) {
    stack_t ss;
    ss.ss_sp = altstack; ss.ss_size = sizeof altstack; ss.ss_flags = 0;
    sigaltstack(&ss, 0);
    struct sigaction sa;
    memset(&sa, 0, sizeof sa);
    sa.sa_sigaction = handler;
    sa.sa_flags = SA_SIGINFO | SA_ONSTACK | SA_NODEFER;
    sigemptyset(&sa.sa_mask);
    struct sigaction cur;
    sigaction(SIGSEGV, 0, &cur);
    if (cur.sa_sigaction != handler) sigaction(SIGSEGV, &sa, &old_segv);
    sigaction(SIGBUS, 0, &cur);
    if (cur.sa_sigaction != handler) sigaction(SIGBUS, &sa, &old_bus);
    sigaction(SIGABRT, 0, &cur);
    if (cur.sa_sigaction != handler) sigaction(SIGABRT, &sa, &old_abrt);
}

__attribute__((constructor)) static void at_load(void) {
    void* warm[4];
    backtrace(warm, 4);                           /* loads libgcc's unwinder now, not inside the handler */
    segvbt_install();
}
