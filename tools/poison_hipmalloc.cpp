// LD_PRELOAD shim: every hipMalloc'ed block is filled with 0xA5 before the caller sees it.
// Fresh device memory reads as zeros on this stack, and zero is a valid action id, a valid cell code and an empty flag word —
// so a buffer that is read before it is written (a missing synchronize between a null-stream producer and a consumer on a
// non-blocking stream, a torch.empty() that should have been filled) goes unnoticed until the allocator hands out a recycled
// block.  Under this shim it shows on the first run:
//   g++ -O2 -fPIC -shared -o gpurun_out/libpoison.so tools/poison_hipmalloc.cpp -ldl
//   LD_PRELOAD=$PWD/gpurun_out/libpoison.so python3 -m pytest tests -m gpu -q
// (no HIP headers needed: the two entry points are looked up by name).  Test tooling, not product code.
#include <dlfcn.h>
#include <stddef.h>
#include <stdio.h>
#include <stdlib.h>
#include <unistd.h>

extern "C" {
typedef int (*malloc_fn)(void**, size_t);
typedef int (*memset_fn)(void*, int, size_t);
typedef int (*sync_fn)(void);

static unsigned long long blocks = 0, bytes = 0;
static void report(void) { fprintf(stderr, "poison_hipmalloc: pid %d, %llu blocks / %.1f MB poisoned\n", (int)getpid(), blocks, bytes / 1e6); }
// the runtime is usually dlopen'ed (RTLD_LOCAL) by the framework, so RTLD_NEXT does not see it: ask the library itself
static void* lookup(const char* name) {
  void* f = dlsym(RTLD_NEXT, name);
  if (!f) {
    static void* lib = dlopen("libamdhip64.so", RTLD_NOW | RTLD_GLOBAL);
    if (lib) f = dlsym(lib, name);
  }
  return f;
}

int hipMalloc(void** ptr, size_t size) {
  static malloc_fn real = (malloc_fn)lookup("hipMalloc");
  static memset_fn mset = (memset_fn)lookup("hipMemset");
  static sync_fn sync = (sync_fn)lookup("hipDeviceSynchronize");
  static int once = atexit(report);
  (void)once;
  if (!real) {
    fprintf(stderr, "poison_hipmalloc: hipMalloc not found behind the shim\n");
    abort();
  }
  const int e = real(ptr, size);
  if (e == 0 && ptr && *ptr && size && mset) {
    mset(*ptr, 0xA5, size);
    if (sync) sync();
    blocks += 1;
    bytes += size;
    if (getenv("POISON_VERBOSE") && (blocks & (blocks - 1)) == 0)
      fprintf(stderr, "poison_hipmalloc: %llu blocks, %.1f MB poisoned so far\n", blocks, bytes / 1e6);
  }
  return e;
}
}
