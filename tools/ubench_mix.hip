// Dev microbenchmark (not part of the product): at TWO waves per SIMD (the c5 roll-out's residency), what does a wave pay per
// RESULT when part of its float32 stream is packed?  8 independent chains; a pattern is a string of S (v_fma_f32, 1 result)
// and P (v_pk_fma_f32, 2 results).  ns per result per SIMD, lower = better; all-scalar is the shipped kernel's regime.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
#define S(i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(d[i]) : "v"(a), "v"(b));
#define P(i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p[i]) : "v"(pa), "v"(pb));

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, float fa, float fb) {
  const int tid = blockIdx.x * 256 + threadIdx.x;
  float d[8], a = fa + tid * 1e-9f, b = fb - tid * 1e-9f;
  v2f p[8], pa = {a, b}, pb = {b, a};
#pragma unroll
  for (int i = 0; i < 8; ++i) { d[i] = tid * 1e-9f + i; p[i] = (v2f){d[i], d[i] + 1}; }
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) { S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7) S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7) S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7) }   // 24 results, 24 slots
    if (MODE == 1) { P(0) P(1) P(2) P(3) P(4) P(5) P(6) P(7) P(0) P(1) P(2) P(3) }                                                            // 24 results, 12 slots
    if (MODE == 2) { S(0) S(1) P(0) S(2) S(3) P(1) S(4) S(5) P(2) S(6) S(7) P(3) S(0) S(1) P(4) S(2) S(3) P(5) }                               // 12 S + 6 P = 24 results, 18 slots
    if (MODE == 3) { S(0) P(0) S(1) P(1) S(2) P(2) S(3) P(3) S(4) P(4) S(5) P(5) S(6) P(6) S(7) P(7) }                                         // 8 S + 8 P = 24 results, 16 slots
    if (MODE == 4) { S(0) S(1) S(2) S(3) P(0) S(4) S(5) S(6) S(7) P(1) S(0) S(1) S(2) S(3) P(2) S(4) S(5) S(6) S(7) P(3) }                     // 16 S + 4 P = 24 results, 20 slots
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += d[i] + p[i].x + p[i].y;
  out[tid] = s;
}

template <int MODE>
void run(const char* name, int w) {
  const int blocks = 256 * w;
  float* d;
  (void)hipMalloc(&d, (size_t)blocks * 256 * 4);
  const int iters = 6000;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  k<MODE><<<blocks, 256>>>(d, 200, 0.5f, 0.25f);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  k<MODE><<<blocks, 256>>>(d, iters, 0.5f, 0.25f);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  printf("%-34s waves/SIMD=%d  %.3f ms  %.3f ns per result per SIMD\n", name, w, ms, ms * 1e6 / ((double)iters * 24 * w));
  (void)hipFree(d);
}

int main() {
  for (int w : {1, 2, 4}) {
    run<0>("all scalar (24 S)", w);
    run<4>("16 S + 4 P (1 in 5 packed)", w);
    run<2>("12 S + 6 P (1 in 3 packed)", w);
    run<3>("8 S + 8 P (alternating)", w);
    run<1>("all packed (12 P)", w);
  }
  return 0;
}
