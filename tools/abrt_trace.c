/* Diagnostic (GPU box): LD_PRELOAD this to see WHO ends the process with SIGABRT -- interposes abort() / raise() / kill() / pthread_kill() and
 * installs a SIGABRT handler; each prints a native back trace of the calling thread on stderr first.
 * gcc -shared -fPIC -o build/exp/libabrt.so tools/abrt_trace.c -ldl ; LD_PRELOAD=$PWD/build/exp/libabrt.so python -m pytest -p no:faulthandler ... */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <execinfo.h>
#include <pthread.h>
#include <signal.h>
#include <string.h>
#include <sys/types.h>
#include <unistd.h>
static int installing;
static void trace(const char *who) {
    void *b[96];
    if (write(2, "\n==== ", 6) && write(2, who, strlen(who)) && write(2, ": native back trace ====\n", 25)) {}
    int n = backtrace(b, 96);
    backtrace_symbols_fd(b, n, 2);
}
static void on_abort(int s) { trace("SIGABRT handler"); installing = 1; { struct sigaction sa; memset(&sa, 0, sizeof sa); sa.sa_handler = SIG_DFL; sigaction(s, &sa, 0); } raise(s); }
void abort(void) {
    trace("abort()");
    void (*real)(void) = (void (*)(void))dlsym(RTLD_NEXT, "abort");
    installing = 1;
    { struct sigaction sa; memset(&sa, 0, sizeof sa); sa.sa_handler = SIG_DFL; sigaction(SIGABRT, &sa, 0); }
    if (real) real();
    _exit(134);
}
int raise(int sig) {
    int (*real)(int) = (int (*)(int))dlsym(RTLD_NEXT, "raise");
    if (sig == SIGABRT) trace("raise(SIGABRT)");
    return real(sig);
}
int kill(pid_t pid, int sig) {
    int (*real)(pid_t, int) = (int (*)(pid_t, int))dlsym(RTLD_NEXT, "kill");
    if (sig == SIGABRT) trace("kill(SIGABRT)");
    return real(pid, sig);
}
int pthread_kill(pthread_t t, int sig) {
    int (*real)(pthread_t, int) = (int (*)(pthread_t, int))dlsym(RTLD_NEXT, "pthread_kill");
    if (sig == SIGABRT) trace("pthread_kill(SIGABRT)");
    return real(t, sig);
}
/* nobody takes the SIGABRT handler away again (somebody in the test process resets it to the default) */
int sigaction(int sig, const struct sigaction *act, struct sigaction *old) {
    int (*real)(int, const struct sigaction *, struct sigaction *) = (int (*)(int, const struct sigaction *, struct sigaction *))dlsym(RTLD_NEXT, "sigaction");
    if (sig == SIGABRT && act && !installing) { trace("sigaction(SIGABRT) ignored"); return real(sig, 0, old); }
    return real(sig, act, old);
}
sighandler_t signal(int sig, sighandler_t h) {
    sighandler_t (*real)(int, sighandler_t) = (sighandler_t (*)(int, sighandler_t))dlsym(RTLD_NEXT, "signal");
    if (sig == SIGABRT && !installing) { trace("signal(SIGABRT) ignored"); return SIG_DFL; }
    return real(sig, h);
}
__attribute__((constructor)) static void init(void) {
    void *b[4];
    backtrace(b, 4);            /* (loads libgcc now, not inside a handler) */
    installing = 1;
    { struct sigaction sa; memset(&sa, 0, sizeof sa); sa.sa_handler = on_abort; sigaction(SIGABRT, &sa, 0); }
    installing = 0;
}
