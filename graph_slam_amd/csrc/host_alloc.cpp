// Host allocations of libfgo: operator new / delete of THIS library only (the symbols are hidden: -fvisibility=hidden + version
// script; the host program's own allocations never come here).  They are malloc / free like the default ones -- memory may cross to
// the default operator delete and back without harm -- plus one hint: an allocation of 4 MB or more asks for transparent huge
// pages.  The structure phase writes ~1.5 GB of index tables once (cfg 2) and frees them again: with 4 KB pages that is 390 k page
// faults and as many page releases, ~0.33 s of kernel time on the MI355X host (tools/_ab/thp.cpp: first touch 210 ms + munmap 110 ms
// single-threaded; with MADV_HUGEPAGE 59 + 57), and while a release is in flight every other address-space operation of the
// process waits for it -- a 6 MB pageable hipMemcpyAsync took 30-130 ms instead of 0.2 (tools/_ab/h2d.hip).
// FGO_THP=0 switches the hint off (hosts whose memory is too fragmented to hand out 2 MB pages without compaction stalls).
#include <sys/mman.h>
#include <cstdint>
#include <cstdlib>
#include <new>

namespace {
const bool g_thp = [] { const char *e = std::getenv("FGO_THP"); return !(e && e[0] == '0'); }();
inline void *fgo_alloc(std::size_t n) {
  void *p = std::malloc(n ? n : 1);
  if (!p) throw std::bad_alloc();
  if (g_thp && n >= ((std::size_t)4 << 20)) {
    const std::uintptr_t a = ((std::uintptr_t)p + 4095) & ~(std::uintptr_t)4095, b = ((std::uintptr_t)p + n) & ~(std::uintptr_t)4095;
    if (b > a) (void)madvise((void *)a, b - a, MADV_HUGEPAGE);
  }
  return p;
}
}  // namespace

void *operator new(std::size_t n) { return fgo_alloc(n); }
void *operator new[](std::size_t n) { return fgo_alloc(n); }
void operator delete(void *p) noexcept { std::free(p); }
void operator delete[](void *p) noexcept { std::free(p); }
void operator delete(void *p, std::size_t) noexcept { std::free(p); }
void operator delete[](void *p, std::size_t) noexcept { std::free(p); }
