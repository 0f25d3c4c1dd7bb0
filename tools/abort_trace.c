// LD_PRELOAD helper for GPU-box debugging: C-level backtrace on a fatal signal, on an alternate stack, and nobody else in the
// process may replace these handlers (sigaction / signal are interposed for the fatal signals).
// build: gcc -shared -fPIC -O1 -o tools/libabort_trace.so tools/abort_trace.c -ldl
#define _GNU_SOURCE
#include <dlfcn.h>
#include <execinfo.h>
#include <signal.h>
#include <stdio.h>
#include <string.h>
#include <sys/syscall.h>
#include <unistd.h>

static int fatal(int sig) { return sig == SIGABRT || sig == SIGSEGV || sig == SIGBUS || sig == SIGFPE || sig == SIGILL; }
static int (*real_sigaction)(int, const struct sigaction *, struct sigaction *);
static char altstack[1 << 16];

static void on_fatal(int sig, siginfo_t *si, void *uc)
{
    (void)uc;
    char buf[160];
    int n = snprintf(buf, sizeof buf, "\n== fatal signal %d (si_code %d, addr %p) tid %ld: C backtrace ==\n", sig, si ? si->si_code : 0,
                     si ? si->si_addr : 0, (long)syscall(SYS_gettid));
    (void)!write(2, buf, (size_t)n);
    void *bt[96];
    int d = backtrace(bt, 96);
    backtrace_symbols_fd(bt, d, 2);
    struct sigaction dfl;
    memset(&dfl, 0, sizeof dfl);
    dfl.sa_handler = SIG_DFL;
    real_sigaction(sig, &dfl, 0);
    raise(sig);
}

int sigaction(int sig, const struct sigaction *act, struct sigaction *old)
{
    if (!real_sigaction) real_sigaction = dlsym(RTLD_NEXT, "sigaction");
    if (fatal(sig) && act) { // pretend success, keep ours; say who asked
        char buf[128];
        int n = snprintf(buf, sizeof buf, "[abort_trace] sigaction(%d, handler %p) refused; caller:\n", sig, (void *)act->sa_sigaction);
        (void)!write(2, buf, (size_t)n);
        void *bt[12];
        int d = backtrace(bt, 12);
        backtrace_symbols_fd(bt, d, 2);
        if (old) memset(old, 0, sizeof *old);
        return 0;
    }
    return real_sigaction(sig, act, old);
}
sighandler_t signal(int sig, sighandler_t h)
{
    struct sigaction a, o;
    memset(&a, 0, sizeof a);
    a.sa_handler = h;
    sigemptyset(&a.sa_mask);
    a.sa_flags = SA_RESTART;
    if (sigaction(sig, &a, &o)) return SIG_ERR;
    return o.sa_handler;
}

__attribute__((constructor)) static void install(void)
{
    if (!real_sigaction) real_sigaction = dlsym(RTLD_NEXT, "sigaction");
    stack_t ss;
    ss.ss_sp = altstack;
    ss.ss_size = sizeof altstack;
    ss.ss_flags = 0;
    sigaltstack(&ss, 0); // main thread only; other threads fall back to their own stack
    struct sigaction a;
    memset(&a, 0, sizeof a);
    a.sa_sigaction = on_fatal;
    a.sa_flags = SA_SIGINFO | SA_ONSTACK;
    sigemptyset(&a.sa_mask);
    int sigs[] = {SIGABRT, SIGSEGV, SIGBUS, SIGFPE, SIGILL};
    for (unsigned i = 0; i < sizeof sigs / sizeof *sigs; i++) real_sigaction(sigs[i], &a, 0);
    void *bt[4];
    backtrace(bt, 4); // load libgcc now, not inside the handler
}
