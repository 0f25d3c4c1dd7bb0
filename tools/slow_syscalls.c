// LD_PRELOAD helper for GPU-box debugging: every ioctl / mmap / munmap / madvise that takes longer than SLOW_US microseconds (default
// 1000) is logged to stderr with its thread, start time (CLOCK_MONOTONIC, ms), duration, the ioctl request number (for /dev/kfd: the
// AMDKFD_IOC_* command in the low byte) and a short C backtrace -- to find what holds the process up during the rare multi-millisecond
// stalls of long runs.  Not product code.
// build: gcc -shared -fPIC -O1 -o tools/libslow_syscalls.so tools/slow_syscalls.c -ldl
#define _GNU_SOURCE
#include <dlfcn.h>
#include <execinfo.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <sys/mman.h>
#include <sys/syscall.h>
#include <time.h>
#include <unistd.h>

static double now_ms(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
}
static double slow_ms(void)
{
    static double v = -1;
    if (v < 0) v = getenv("SLOW_US") ? atof(getenv("SLOW_US")) * 1e-3 : 1.0;
    return v;
}
static void report(const char *what, unsigned long a, unsigned long b, double t0, double dt)
{
    char buf[256];
    int n = snprintf(buf, sizeof buf, "[slow] t=%.3f ms %8.3f ms tid %ld %s 0x%lx (nr 0x%02lx) arg 0x%lx\n", t0, dt, (long)syscall(SYS_gettid), what, a, a & 0xFF, b);
    (void)!write(2, buf, (size_t)n);
    if (getenv("SLOW_BT")) {
        void *bt[24];
        int d = backtrace(bt, 24);
        backtrace_symbols_fd(bt, d, 2);
    }
}

int ioctl(int fd, unsigned long req, ...)
{
    static int (*real)(int, unsigned long, ...);
    if (!real) real = (int (*)(int, unsigned long, ...))dlsym(RTLD_NEXT, "ioctl");
    va_list ap;
    va_start(ap, req);
    void *arg = va_arg(ap, void *);
    va_end(ap);
    const double t0 = now_ms();
    const int r = real(fd, req, arg);
    const double dt = now_ms() - t0;
    // SLOW_KFD=1: every KFD memory operation whatever it took -- ALLOC_MEMORY_OF_GPU (0x16: va, size, ..., flags), FREE (0x17), MAP (0x18),
    // UNMAP (0x19), SVM (0x20: start, size, op, nattr) -- with the sizes from the argument structs (linux/kfd_ioctl.h)
    if (getenv("SLOW_KFD") && ((req >> 8) & 0xFF) == 'K' && arg) {
        const unsigned nr = req & 0xFF;
        const unsigned long long *a = (const unsigned long long *)arg;
        char buf[256];
        int n = 0;
        if (nr == 0x16) n = snprintf(buf, sizeof buf, "[kfd] t=%.3f ms %7.3f ms tid %ld ALLOC va 0x%llx size %llu flags 0x%x\n", t0, dt, (long)syscall(SYS_gettid), a[0], a[1], ((const unsigned *)arg)[9]);
        else if (nr == 0x20) n = snprintf(buf, sizeof buf, "[kfd] t=%.3f ms %7.3f ms tid %ld SVM start 0x%llx size %llu op %u nattr %u\n", t0, dt, (long)syscall(SYS_gettid), a[0], a[1], ((const unsigned *)arg)[4], ((const unsigned *)arg)[5]);
        else if (nr == 0x17 || nr == 0x18 || nr == 0x19) n = snprintf(buf, sizeof buf, "[kfd] t=%.3f ms %7.3f ms tid %ld %s handle 0x%llx\n", t0, dt, (long)syscall(SYS_gettid), nr == 0x17 ? "FREE" : nr == 0x18 ? "MAP" : "UNMAP", a[0]);
        if (n > 0) (void)!write(2, buf, (size_t)n);
    }
    // AMDKFD_IOC_WAIT_EVENTS (nr 0x0C) blocks by design: only reported when SLOW_WAITS is set
    if (dt > slow_ms() && ((req & 0xFF) != 0x0C || getenv("SLOW_WAITS"))) report("ioctl", req, (unsigned long)fd, t0, dt);
    return r;
}
void *mmap(void *addr, size_t len, int prot, int flags, int fd, off_t off)
{
    static void *(*real)(void *, size_t, int, int, int, off_t);
    if (!real) real = (void *(*)(void *, size_t, int, int, int, off_t))dlsym(RTLD_NEXT, "mmap");
    const double t0 = now_ms();
    void *r = real(addr, len, prot, flags, fd, off);
    const double dt = now_ms() - t0;
    if (dt > slow_ms()) report("mmap len", len, (unsigned long)flags, t0, dt);
    return r;
}
int munmap(void *addr, size_t len)
{
    static int (*real)(void *, size_t);
    if (!real) real = (int (*)(void *, size_t))dlsym(RTLD_NEXT, "munmap");
    const double t0 = now_ms();
    const int r = real(addr, len);
    const double dt = now_ms() - t0;
    if (dt > slow_ms()) report("munmap len", len, 0, t0, dt);
    return r;
}
int madvise(void *addr, size_t len, int advice)
{
    static int (*real)(void *, size_t, int);
    if (!real) real = (int (*)(void *, size_t, int))dlsym(RTLD_NEXT, "madvise");
    const double t0 = now_ms();
    const int r = real(addr, len, advice);
    const double dt = now_ms() - t0;
    if (dt > slow_ms()) report("madvise len", len, (unsigned long)advice, t0, dt);
    return r;
}
