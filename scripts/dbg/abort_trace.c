// Debug aid: LD_PRELOAD to get a native backtrace on SIGABRT.
#define _GNU_SOURCE
#include <execinfo.h>
#include <signal.h>
#include <unistd.h>
static void on_abort(int sig) {
  void* bt[64];
  int n = backtrace(bt, 64);
  backtrace_symbols_fd(bt, n, 2);
  signal(sig, SIG_DFL);
  raise(sig);
}
__attribute__((constructor)) static void init(void) { signal(SIGABRT, on_abort); }
