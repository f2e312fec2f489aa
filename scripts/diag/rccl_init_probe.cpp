// Diagnostic: ncclCommInitRank on a 1-rank (or fork-less N-rank, same device) communicator,
// outside Python, with a SIGABRT handler that prints the aborting thread's backtrace.
//   hipcc -O1 -g rccl_init_probe.cpp -o rccl_init_probe -ldl
//   ./rccl_init_probe [local|global] [setdev_first|setdev_late]
#include <dlfcn.h>
#include <execinfo.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <signal.h>
#include <unistd.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>

static void on_abort(int sig) {
  void* bt[64];
  int n = backtrace(bt, 64);
  const char msg[] = "\n=== SIGABRT backtrace ===\n";
  (void)!write(2, msg, sizeof(msg) - 1);
  backtrace_symbols_fd(bt, n, 2);
  signal(sig, SIG_DFL);
  raise(sig);
}

int main(int argc, char** argv) {
  const bool global = argc > 1 && !strcmp(argv[1], "global");
  const bool late = argc > 2 && !strcmp(argv[2], "setdev_late");
  signal(SIGABRT, on_abort);
  signal(SIGSEGV, on_abort);
  if (!late) {
    if (hipSetDevice(0) != hipSuccess) { fprintf(stderr, "hipSetDevice failed\n"); return 2; }
    void* p = nullptr;
    if (hipMalloc(&p, 1 << 20) != hipSuccess) return 2;
    hipFree(p);
  }
  void* lib = dlopen("librccl.so.1", RTLD_NOW | (global ? RTLD_GLOBAL : RTLD_LOCAL));
  if (!lib) { fprintf(stderr, "dlopen: %s\n", dlerror()); return 3; }
  auto GetUniqueId = (ncclResult_t(*)(ncclUniqueId*))dlsym(lib, "ncclGetUniqueId");
  auto CommInitRank = (ncclResult_t(*)(ncclComm_t*, int, ncclUniqueId, int))dlsym(lib, "ncclCommInitRank");
  auto Broadcast = (ncclResult_t(*)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t))dlsym(lib, "ncclBroadcast");
  auto CommDestroy = (ncclResult_t(*)(ncclComm_t))dlsym(lib, "ncclCommDestroy");
  ncclUniqueId id;
  ncclResult_t r = GetUniqueId(&id);
  if (r != ncclSuccess) { fprintf(stderr, "GetUniqueId %d\n", (int)r); return 4; }
  if (late && hipSetDevice(0) != hipSuccess) return 2;
  ncclComm_t comm;
  r = CommInitRank(&comm, 1, id, 0);
  if (r != ncclSuccess) { fprintf(stderr, "CommInitRank %d\n", (int)r); return 5; }
  hipStream_t st;
  hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
  void* d = nullptr;
  hipMalloc(&d, 1 << 20);
  r = Broadcast(d, d, 1 << 20, ncclUint8, 0, comm, st);
  if (r != ncclSuccess) { fprintf(stderr, "Broadcast %d\n", (int)r); return 6; }
  hipStreamSynchronize(st);
  CommDestroy(comm);
  hipFree(d);
  hipStreamDestroy(st);
  printf("PROBE_OK\n");
  return 0;
}
