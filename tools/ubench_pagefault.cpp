// Dev probe: how fast can N threads fault in a fresh 98 MB anonymous mapping (a NumPy output array), and what does
// unmapping it cost?  g++ -O2 -pthread.  MI355X box (microVM): 4 KiB pages 13 ms on one thread, 8-10 ms on 4-8 (does
// not scale; read-then-write touching is 4x worse); huge pages (what NumPy madvises) 4.5 ms on one thread, 1.4 ms on
// eight; munmap 5-12 ms either way.  Tried and dropped: touching a fresh output on helper threads while the upload
// runs - the download into it got no faster (12.5 ms fresh vs 2.0 ms reused either way), so the library does not do it;
// callers that care reuse their output (`out=`) or use page-locked buffers (mp_host_alloc).
#include <sys/mman.h>
#include <chrono>
#include <cstdio>
#include <thread>
#include <vector>
#include <atomic>
int main() {
  const size_t bytes = 98304000, blk = 2 << 20;
  for (int nt : {1, 4, 8}) {
    for (int mode = 0; mode < 6; ++mode) {
      char* p = (char*)mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
      if (mode >= 3) madvise(p, bytes, MADV_HUGEPAGE);
      std::atomic<size_t> next{0};
      auto t0 = std::chrono::steady_clock::now();
      std::vector<std::thread> th;
      for (int t = 0; t < nt; ++t)
        th.emplace_back([&] {
          for (;;) {
            size_t b = next.fetch_add(1);
            if (b * blk >= bytes) break;
            size_t hi = std::min(bytes, (b + 1) * blk);
            volatile char* q = p;
            for (size_t o = b * blk; o < hi; o += 4096) { int m = mode % 3; if (m == 1) q[o] = q[o]; else if (m == 0) q[o] = 0; else __atomic_fetch_or((char*)p + o, 0, __ATOMIC_RELAXED); }
          }
        });
      for (auto& t : th) t.join();
      double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
      auto t1 = std::chrono::steady_clock::now();
      munmap(p, bytes);
      double ms2 = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t1).count();
      printf("threads %2d mode %s: touch %.2f ms, munmap %.2f ms\n", nt, (const char*[]){"w", "rw", "atomic-or", "THP w", "THP rw", "THP atomic-or"}[mode], ms, ms2);
    }
  }
}
