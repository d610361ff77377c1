// v21_jitc -- the compiler process of csrc/jit.hip.
//
// A run-time instantiation of the fused kernel is a hiprtc (LLVM) compilation of several seconds.  Inside the process
// that drives the GPU it ran in a thread at first (r4) and took the process down twice: a compilation still running
// when the process exits outlives the static objects of the compiler library it runs in (exit() destroys them in
// reverse order of loading, i.e. BEFORE the library that waits for the thread), and LLVM's global state is shared with
// whatever the HIP runtime compiles or parses itself.  So the compilation is a child process: libv21.so starts this
// program (posix_spawn), which loads the same library, calls v21_jit_prebuild -- no GPU is touched -- and leaves the code
// object in the cache directory; the parent's thread only waits for it, and kills it if the parent exits first.
//
//   v21_jitc <cache dir> <error file> <precision> <n_layers> <dims ...> <act ...>
#include <dlfcn.h>
#include <libgen.h>
#include <unistd.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

int main(int argc, char** argv) {
  if (argc < 5) { fprintf(stderr, "usage: v21_jitc <cache dir> <error file> <precision> <n_layers> <dims...> <act...>\n"); return 2; }
  const char* dir = argv[1];
  const char* errfile = argv[2];
  const int prec = atoi(argv[3]), L = atoi(argv[4]);
  if (L < 1 || L > 16 || argc != 5 + (L + 1) + L) { fprintf(stderr, "v21_jitc: bad argument count\n"); return 2; }
  std::vector<int> dims(L + 1), act(L);
  for (int i = 0; i <= L; ++i) dims[i] = atoi(argv[5 + i]);
  for (int i = 0; i < L; ++i) act[i] = atoi(argv[5 + L + 1 + i]);
  auto fail = [&](const std::string& why) {
    fprintf(stderr, "v21_jitc: %s\n", why.c_str());
    if (FILE* f = fopen(errfile, "w")) { fputs(why.c_str(), f); fclose(f); }
    return 1;
  };
  char self[4096];
  const ssize_t n = readlink("/proc/self/exe", self, sizeof self - 1);
  if (n <= 0) return fail("cannot resolve /proc/self/exe");
  self[n] = 0;
  const std::string lib = std::string(dirname(self)) + "/libv21.so";
  void* h = dlopen(lib.c_str(), RTLD_NOW | RTLD_LOCAL);
  if (!h) return fail(std::string("cannot load ") + lib + ": " + dlerror());
  auto prebuild = (int (*)(int, const int*, const int*, int, const char*))dlsym(h, "v21_jit_prebuild");
  auto last_error = (const char* (*)(void))dlsym(h, "v21_last_error");
  if (!prebuild || !last_error) return fail("libv21.so lacks v21_jit_prebuild");
  if (prebuild(L, dims.data(), act.data(), prec, dir) != 0) return fail(last_error());
  unlink(errfile);
  // (leave without running the static destructors of the compiler libraries: nothing here needs them)
  fflush(nullptr);
  _exit(0);
}
