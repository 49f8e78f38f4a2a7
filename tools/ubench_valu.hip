// Dev microbenchmark (not part of the product): what does one gfx950 SIMD sustain for scalar vs packed
// f32 FMA and f64 FMA?  Decides whether the row-per-thread kernels should pack two rows per lane.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a, float b) {
  const int tid = blockIdx.x * 256 + threadIdx.x;
  if (MODE == 0) {  // scalar f32 fma, 8 independent chains
    float x[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) x[i] = tid * 1e-9f + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 8; ++i) x[i] = __builtin_fmaf(x[i], a, b);
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += x[i];
    out[tid] = s;
  } else if (MODE == 1) {  // packed f32 fma, 8 independent chains of float2
    v2f x[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) x[i] = (v2f){tid * 1e-9f + i, tid * 2e-9f - i};
    const v2f va = {a, a}, vb = {b, b};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 8; ++i) x[i] = __builtin_elementwise_fma(x[i], va, vb);
    }
    v2f s = {0, 0};
#pragma unroll
    for (int i = 0; i < 8; ++i) s += x[i];
    out[tid] = s.x + s.y;
  } else {  // f64 fma
    double x[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) x[i] = tid * 1e-9 + i;
    const double da = a, db = b;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 8; ++i) x[i] = __builtin_fma(x[i], da, db);
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += x[i];
    out[tid] = (float)s;
  }
}

template <int MODE>
void run(const char* name, int waves_per_simd, double fma_per_inst) {
  const int blocks = 256 * waves_per_simd;  // 256 threads = 4 waves = one per SIMD
  float* d;
  hipMalloc(&d, (size_t)blocks * 256 * 4);
  const int iters = 20000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  k<MODE><<<blocks, 256>>>(d, 100, 0.999f, 0.001f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<MODE><<<blocks, 256>>>(d, iters, 0.999f, 0.001f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double inst_per_wave = 8.0 * iters;
  const double waves = blocks * 4.0;
  const double fmas = waves * 64 * inst_per_wave * fma_per_inst;
  // cycles per wave-instruction per SIMD at an assumed 2.4 GHz
  const double cyc = ms * 1e-3 * 2.4e9 / (inst_per_wave * waves_per_simd);
  printf("%-10s waves/SIMD=%d  %.3f ms  %.1f TFLOP/s  (%.2f cyc/wave-inst/SIMD @2.4GHz)\n", name, waves_per_simd, ms,
         2 * fmas / (ms * 1e-3) / 1e12, cyc);
  hipFree(d);
}

int main() {
  for (int w : {1, 2, 4, 8}) {
    run<0>("fma_f32", w, 1);
    run<1>("pk_fma_f32", w, 2);
    run<2>("fma_f64", w, 1);
  }
  return 0;
}
