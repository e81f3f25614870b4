/*
 * pthread_fail_shim.c -- LD_PRELOADed by the tests: makes pthread_create fail with EAGAIN on demand, the way it does for a process at
 * its thread limit (RLIMIT_NPROC / the cgroup's pids.max -- what a Go host under a goroutine storm can run into).  `ulimit -u` cannot be
 * used for this: the limit is not enforced for root (this container), and it counts the whole user's threads on a shared GPU box.
 * rsn_test_fail_threads(1) switches the failure on, (0) off; RSN_TEST_FAIL_THREADS=1 in the environment starts with it on.
 */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <errno.h>
#include <execinfo.h>
#include <pthread.h>
#include <stdlib.h>
#include <unistd.h>

static volatile int g_fail = -1;
static volatile long g_refused = 0;

void rsn_test_fail_threads(int on) { g_fail = on; }
long rsn_test_threads_refused(void) { return g_refused; }

int pthread_create(pthread_t *t, const pthread_attr_t *a, void *(*fn)(void *), void *arg) {
    static int (*real)(pthread_t *, const pthread_attr_t *, void *(*)(void *), void *);
    if (!real) real = (int (*)(pthread_t *, const pthread_attr_t *, void *(*)(void *), void *))dlsym(RTLD_NEXT, "pthread_create");
    if (g_fail < 0) { const char *e = getenv("RSN_TEST_FAIL_THREADS"); g_fail = e && *e == '1'; }
    if (g_fail) { __sync_fetch_and_add(&g_refused, 1); return EAGAIN; }
    return real(t, a, fn, arg);
}

/* RSN_TEST_TRACE_THROW=1: where a C++ exception is thrown (the guard at the C boundary turns it into a code; this shows its origin) */
void __cxa_throw(void *obj, void *tinfo, void (*dest)(void *)) {
    static void (*real)(void *, void *, void (*)(void *)) __attribute__((noreturn));
    if (!real) real = (void (*)(void *, void *, void (*)(void *)))dlsym(RTLD_NEXT, "__cxa_throw");
    const char *e = getenv("RSN_TEST_TRACE_THROW");
    if (e && *e == '1') { void *bt[32]; const int k = backtrace(bt, 32); backtrace_symbols_fd(bt, k, 2); }
    real(obj, tinfo, dest);
    abort();
}
