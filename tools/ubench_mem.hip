// Dev microbenchmark (not part of the product): HBM ceiling for the row-per-lane access pattern of the
// inverse-dynamics kernels (3 input arrays + 1 output array, each lane owning a contiguous 24 / 48 byte
// run) against a perfectly coalesced float4 stream of the same bytes.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

// (a) coalesced: lane i handles float4 #i of every array
__global__ __launch_bounds__(256) void k_coalesced(const float4* a, const float4* b, const float4* c, float4* o, long n4) {
  long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  float4 x = a[i], y = b[i], z = c[i];
  o[i] = make_float4(x.x + y.x + z.x, x.y + y.y + z.y, x.z + y.z + z.z, x.w + y.w + z.w);
}
// (b) lane owns RUN consecutive float4 (RUN*16 bytes) per array: RUN=3 is the packed n=6 kernel's pattern
template <int RUN>
__global__ __launch_bounds__(256) void k_rows(const float4* a, const float4* b, const float4* c, float4* o, long nrun) {
  long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= nrun) return;
  float4 x[RUN], y[RUN], z[RUN];
#pragma unroll
  for (int k = 0; k < RUN; ++k) { x[k] = a[i * RUN + k]; y[k] = b[i * RUN + k]; z[k] = c[i * RUN + k]; }
#pragma unroll
  for (int k = 0; k < RUN; ++k)
    o[i * RUN + k] = make_float4(x[k].x + y[k].x + z[k].x, x[k].y + y[k].y + z[k].y, x[k].z + y[k].z + z[k].z, x[k].w + y[k].w + z[k].w);
}
// (c) lane owns 24 bytes (3 x float2): the scalar n=6 kernel's pattern
__global__ __launch_bounds__(256) void k_rows24(const float2* a, const float2* b, const float2* c, float2* o, long nrow) {
  long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= nrow) return;
  float2 x[3], y[3], z[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) { x[k] = a[i * 3 + k]; y[k] = b[i * 3 + k]; z[k] = c[i * 3 + k]; }
#pragma unroll
  for (int k = 0; k < 3; ++k) o[i * 3 + k] = make_float2(x[k].x + y[k].x + z[k].x, x[k].y + y[k].y + z[k].y);
}

int main() {
  for (long rows : {4096000L, 32768000L}) {
    const long floats = rows * 6, bytes = floats * 4;
    float *a, *b, *c, *o;
    hipMalloc(&a, bytes); hipMalloc(&b, bytes); hipMalloc(&c, bytes); hipMalloc(&o, bytes);
    hipMemset(a, 0, bytes); hipMemset(b, 0, bytes); hipMemset(c, 0, bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto timeit = [&](const char* name, auto launch) {
      for (int i = 0; i < 3; ++i) launch();
      hipEventRecord(e0);
      const int it = 20;
      for (int i = 0; i < it; ++i) launch();
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); ms /= it;
      printf("rows=%ld %-14s %.4f ms  %.0f GB/s\n", rows, name, ms, 4.0 * bytes / (ms * 1e-3) / 1e9);
    };
    const long n4 = floats / 4;
    timeit("coalesced", [&] { k_coalesced<<<(n4 + 255) / 256, 256>>>((float4*)a, (float4*)b, (float4*)c, (float4*)o, n4); });
    timeit("rows48B", [&] { k_rows<3><<<(n4 / 3 + 255) / 256, 256>>>((float4*)a, (float4*)b, (float4*)c, (float4*)o, n4 / 3); });
    timeit("rows24B", [&] { k_rows24<<<(rows + 255) / 256, 256>>>((float2*)a, (float2*)b, (float2*)c, (float2*)o, rows); });
    hipFree(a); hipFree(b); hipFree(c); hipFree(o);
  }
  return 0;
}
