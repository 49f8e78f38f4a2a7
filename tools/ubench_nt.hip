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
// the c2 kernel's own movement structure without its arithmetic: a wave moves ROWS 24-byte rows of each array as flat 16-byte
// chunks through LDS (non-temporal), each lane picks its row(s), adds them, the sums leave the same way
typedef unsigned u4 __attribute__((ext_vector_type(4)));
template <int ROWS>  // rows per wave: 64 (1.5 chunk instructions per array) or 128 (3 full ones, two rows per lane)
__global__ __launch_bounds__(256) void k_rows(const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ c, float* __restrict__ o, long rows) {
  constexpr int ROWB = 24, SPAN = ROWS * ROWB, NCH = SPAN / 16, NJ = (NCH + 63) / 64;
  __shared__ __attribute__((aligned(16))) char lds_all[4][3 * SPAN];
  char* lds = lds_all[threadIdx.x >> 6];
  const int lane = threadIdx.x & 63;
  const long row0 = ((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * ROWS;
  if (row0 >= rows) return;
  const float* src[3] = {a, b, c};
  u4 buf[3][NJ];
#pragma unroll
  for (int k = 0; k < 3; ++k)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
      if (j * 64 + lane < NCH) buf[k][j] = __builtin_nontemporal_load(reinterpret_cast<const u4*>(src[k] + row0 * 6) + j * 64 + lane);
#pragma unroll
  for (int k = 0; k < 3; ++k)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
      if (j * 64 + lane < NCH) *reinterpret_cast<u4*>(lds + k * SPAN + (j * 64 + lane) * 16) = buf[k][j];
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  float t[ROWS / 64][6];
#pragma unroll
  for (int r = 0; r < ROWS / 64; ++r)
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      const int off = (r * 64 + lane) * ROWB + j * 4;
      t[r][j] = *reinterpret_cast<const float*>(lds + off) + *reinterpret_cast<const float*>(lds + SPAN + off) + *reinterpret_cast<const float*>(lds + 2 * SPAN + off);
    }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
  for (int r = 0; r < ROWS / 64; ++r)
#pragma unroll
    for (int j = 0; j < 6; ++j) *reinterpret_cast<float*>(lds + (r * 64 + lane) * ROWB + j * 4) = t[r][j];
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
  for (int j = 0; j < NJ; ++j)
    if (j * 64 + lane < NCH) __builtin_nontemporal_store(*reinterpret_cast<const u4*>(lds + (j * 64 + lane) * 16), reinterpret_cast<u4*>(o + row0 * 6) + j * 64 + lane);
}
template <int ROWS>
void run_rows(const char* name, f4** bufs, long rows) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int grid = (int)((rows / ROWS + 3) / 4);
  for (int w = 0; w < 30; ++w) { f4** s = bufs + 4 * (w % 3); k_rows<ROWS><<<grid, 256>>>((float*)s[0], (float*)s[1], (float*)s[2], (float*)s[3], rows); }
  (void)hipDeviceSynchronize();
  const int reps = 60;
  (void)hipEventRecord(e0);
  for (int w = 0; w < reps; ++w) { f4** s = bufs + 4 * (w % 3); k_rows<ROWS><<<grid, 256>>>((float*)s[0], (float*)s[1], (float*)s[2], (float*)s[3], rows); }
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  printf("%-28s %.4f ms per launch  %.0f GB/s\n", name, ms / reps, 4.0 * rows * 24 / (ms / reps * 1e-3) / 1e9);
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
    run_rows<64>("24-B rows via LDS, 64/wave", bufs, 4096L * 1000);
    run_rows<128>("24-B rows via LDS, 128/wave", bufs, 4096L * 1000);
  }
  return 0;
}
