// Dev microbenchmark: the c2 access pattern (three input streams, one output stream, 98.3 MB each, three rotating sets so that
// nothing is re-read from the 256 MB cache) with plain and non-temporal loads / stores.  GB/s of algorithmic bytes.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
template <int NTL, int NTS>
__global__ __launch_bounds__(256) void k(const f4* __restrict__ a, const f4* __restrict__ b, const f4* __restrict__ c, f4* __restrict__ o, long n) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  f4 x, y, z;
  if (NTL) { x = __builtin_nontemporal_load(a + i); y = __builtin_nontemporal_load(b + i); z = __builtin_nontemporal_load(c + i); }
  else { x = a[i]; y = b[i]; z = c[i]; }
  f4 r = x * y + z;
  if (NTS) __builtin_nontemporal_store(r, o + i); else o[i] = r;
}
template <int NTL, int NTS>
void run(const char* name, f4** bufs, long n) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int grid = (int)((n + 255) / 256);
  for (int w = 0; w < 30; ++w) { f4** s = bufs + 4 * (w % 3); k<NTL, NTS><<<grid, 256>>>(s[0], s[1], s[2], s[3], n); }
  (void)hipDeviceSynchronize();
  const int reps = 60;
  (void)hipEventRecord(e0);
  for (int w = 0; w < reps; ++w) { f4** s = bufs + 4 * (w % 3); k<NTL, NTS><<<grid, 256>>>(s[0], s[1], s[2], s[3], n); }
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  printf("%-28s %.4f ms per launch  %.0f GB/s\n", name, ms / reps, 4.0 * n * 16 / (ms / reps * 1e-3) / 1e9);
}
int main() {
  const long n = 4096L * 1000 * 6 / 4;  // float4 per stream: 98.3 MB
  f4* bufs[12];
  for (int i = 0; i < 12; ++i) { (void)hipMalloc(&bufs[i], n * 16); (void)hipMemset(bufs[i], 0, n * 16); }
  for (int round = 0; round < 3; ++round) {
    run<0, 0>("plain loads, plain stores", bufs, n);
    run<1, 0>("nt loads, plain stores", bufs, n);
    run<0, 1>("plain loads, nt stores", bufs, n);
    run<1, 1>("nt loads, nt stores", bufs, n);
  }
  return 0;
}
