// Host allocations of libfgo: operator new / delete of THIS library only (the symbols are hidden: -fvisibility=hidden + version
// script; the host program's own allocations never come here).  They FORWARD to the process's global operators -- the default ones
// or whatever the host program replaced them with, looked up with dlsym(RTLD_DEFAULT): memory may therefore cross between this
// library, libstdc++ and the host program in either direction exactly as if nothing were defined here -- and add one hint: an
// allocation of 4 MB or more asks for transparent huge pages.  The structure phase writes ~1.5 GB of index tables once (cfg 2) and
// frees them again: with 4 KB pages that is 390 k page faults and as many page releases, ~0.33 s of kernel time on the MI355X host
// (tools/_ab/thp.cpp: first touch 210 ms + munmap 110 ms single-threaded; with MADV_HUGEPAGE 59 + 57), and while a release is in
// flight every other address-space operation of the process waits for it -- a 6 MB pageable hipMemcpyAsync took 30-130 ms instead
// of 0.2 (tools/_ab/h2d.hip).  FGO_THP=0 switches the hint off (hosts whose memory is too fragmented to hand out 2 MB pages
// without compaction stalls).
// The replacement is safe only while it stays LOCAL to libfgo.so: the build defines FGO_LOCAL_OPERATORS next to -fvisibility=hidden and
// the version script (any other way of linking this file -- a static archive, a build without libfgo.map -- would make it a process-wide
// replacement that finds itself through dlsym), tests/test_lib_cpu.py checks the export table, and a look-up that returns these very
// functions falls back to malloc / free instead of recursing.
#ifndef FGO_LOCAL_OPERATORS
#error "host_alloc.cpp replaces operator new / delete for libfgo.so only: build with -fvisibility=hidden, csrc/libfgo.map and -DFGO_LOCAL_OPERATORS (graph_slam_amd/build.py)"
#endif
#include <dlfcn.h>
#include <sys/mman.h>
#include <cstdint>
#include <cstdlib>
#include <new>

namespace {
typedef void *(*new_fn)(std::size_t);
typedef void (*del_fn)(void *);
struct Global {
  new_fn nw, nwa;
  del_fn dl, dla;
  bool thp;
  Global() {
    nw = (new_fn)dlsym(RTLD_DEFAULT, "_Znwm"); nwa = (new_fn)dlsym(RTLD_DEFAULT, "_Znam");
    dl = (del_fn)dlsym(RTLD_DEFAULT, "_ZdlPv"); dla = (del_fn)dlsym(RTLD_DEFAULT, "_ZdaPv");
    if ((void *)nw == (void *)static_cast<void *(*)(std::size_t)>(&::operator new) || (void *)dl == (void *)static_cast<void (*)(void *) noexcept>(&::operator delete)) nw = nullptr;   // our own: never forward to ourselves
    if (!nw || !dl) { nw = nullptr; dl = nullptr; }                       // (no global pair in sight: malloc / free, as the default pair does)
    if (!nwa || !dla) { nwa = nw; dla = dl; }
    const char *e = std::getenv("FGO_THP");
    thp = !(e && e[0] == '0');
  }
};
inline const Global &global() { static const Global g; return g; }
// The hint goes to blocks that are mappings of their own only -- glibc hands those out at 16 bytes behind a page boundary (the chunk header
// of an mmapped chunk); a block carved out of the shared heap (once the dynamic mmap threshold has grown past the request) would keep the
// flag on pages other code reuses later.
inline void *hinted(void *p, std::size_t n, const Global &g) {
  if (g.thp && n >= ((std::size_t)4 << 20) && ((std::uintptr_t)p & 4095) == 16) {
    const std::uintptr_t a = ((std::uintptr_t)p + 4095) & ~(std::uintptr_t)4095, b = ((std::uintptr_t)p + n) & ~(std::uintptr_t)4095;
    if (b > a) (void)madvise((void *)a, b - a, MADV_HUGEPAGE);
  }
  return p;
}
inline void *plain(std::size_t n) { void *p = std::malloc(n ? n : 1); if (!p) throw std::bad_alloc(); return p; }
}  // namespace

void *operator new(std::size_t n) { const Global &g = global(); return hinted(g.nw ? g.nw(n) : plain(n), n, g); }
void *operator new[](std::size_t n) { const Global &g = global(); return hinted(g.nwa ? g.nwa(n) : plain(n), n, g); }
void operator delete(void *p) noexcept { const Global &g = global(); if (g.dl) g.dl(p); else std::free(p); }
void operator delete[](void *p) noexcept { const Global &g = global(); if (g.dla) g.dla(p); else std::free(p); }
void operator delete(void *p, std::size_t) noexcept { const Global &g = global(); if (g.dl) g.dl(p); else std::free(p); }
void operator delete[](void *p, std::size_t) noexcept { const Global &g = global(); if (g.dla) g.dla(p); else std::free(p); }
