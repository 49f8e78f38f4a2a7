// Does hipExtAnyOrderLaunch let two kernels of ONE stream overlap on gfx950 (ROCm 7.2)?  hip_ext.h says the flag is "not supported
// on AMD GFX9xx boards"; this measures it: two kernels of 8 blocks that each spin ~200 us, back to back in one stream, the second
// launched (a) normally, (b) with hipExtAnyOrderLaunch.  Serial = ~400 us, overlapped = ~200 us.
//   hipcc -O2 --offload-arch=gfx950 -o tools/ubench_anyorder tools/ubench_anyorder.hip && tools/ubench_anyorder
#include <hip/hip_ext.h>
#include <hip/hip_runtime.h>

#include <cstdio>

__global__ void spin(long long cycles, int* out) {
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < cycles) {}
  if (threadIdx.x == 0 && out) atomicAdd(out, 1);
}

int main() {
  hipStream_t s;
  hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  int* d;
  hipMalloc(&d, 4);
  hipMemset(d, 0, 4);
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  const long long cyc = 200 * 100;  // wall_clock64 ticks at 100 MHz: 200 us
  for (int mode = 0; mode < 3; ++mode) {
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(a, s);
      hipLaunchKernelGGL(spin, dim3(8), dim3(64), 0, s, cyc, d);
      if (mode == 0) hipLaunchKernelGGL(spin, dim3(8), dim3(64), 0, s, cyc, d);
      else if (mode == 1) hipExtLaunchKernelGGL(spin, dim3(8), dim3(64), 0, s, nullptr, nullptr, hipExtAnyOrderLaunch, cyc, d);
      else {  // three kernels: normal, normal (tiny), any-order: does the third overlap the second only, or the first too?
        hipLaunchKernelGGL(spin, dim3(8), dim3(64), 0, s, cyc / 4, d);
        hipExtLaunchKernelGGL(spin, dim3(8), dim3(64), 0, s, nullptr, nullptr, hipExtAnyOrderLaunch, cyc, d);
      }
      hipEventRecord(b, s);
      hipEventSynchronize(b);
      float ms = 0;
      hipEventElapsedTime(&ms, a, b);
      printf("mode %d (%s): %.1f us\n", mode, mode == 0 ? "two normal launches" : mode == 1 ? "second with hipExtAnyOrderLaunch" : "normal 200, normal 50, any-order 200",
             ms * 1e3);
    }
  }
  return 0;
}
