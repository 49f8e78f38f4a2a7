// Dev calibration (not part of the product): what does rocprofv3's FETCH_SIZE report for the roll-out's READ pattern?
// MI355X_MICROARCH.md (HBM): on gfx950 FETCH_SIZE shows half the bytes of a wide coalesced streaming read, "other access
// widths are uncalibrated: calibrate on a known byte count in your own access pattern".  Pattern A: perfectly coalesced
// float4 stream.  Pattern B: the roll-out's - lane t reads a 96-byte run (6 x dwordx4) at a 2400-byte pitch, tile after tile.
// Both read exactly `bytes` bytes once; compare FETCH_SIZE (KiB) x 1024 with that number.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));

__global__ void k_stream(const f4* __restrict__ in, float* __restrict__ out, long n4) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long stride = (long)gridDim.x * blockDim.x;
  f4 acc = {0, 0, 0, 0};
  for (long k = i; k < n4; k += stride) acc += in[k];
  out[i] = acc.x + acc.y + acc.z + acc.w;
}
// B trajectories x Nt steps x 6 floats; lane = trajectory; 4 steps (96 B) per iteration
__global__ void k_runs(const float* __restrict__ in, float* __restrict__ out, long B, long Nt) {
  const long b = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const f4* p = reinterpret_cast<const f4*>(in + b * Nt * 6);
  f4 acc = {0, 0, 0, 0};
  for (long i0 = 0; i0 < Nt; i0 += 4) {
#pragma unroll
    for (int c = 0; c < 6; ++c) acc += p[(i0 / 4) * 6 + c];
    // a little arithmetic between tiles, like the integration steps
#pragma unroll 1
    for (int r = 0; r < 64; ++r) acc = acc * 1.0001f + 0.5f;
  }
  out[b] = acc.x + acc.y + acc.z + acc.w;
}
int main() {
  const long B = 131072, Nt = 100;
  const size_t bytes = (size_t)B * Nt * 6 * 4;  // 314,572,800
  float *in, *out;
  (void)hipMalloc(&in, bytes); (void)hipMalloc(&out, B * 4);
  (void)hipMemset(in, 0, bytes);
  for (int rep = 0; rep < 3; ++rep) k_stream<<<2048, 256>>>(reinterpret_cast<const f4*>(in), out, (long)(bytes / 16));
  for (int rep = 0; rep < 3; ++rep) k_runs<<<(unsigned)(B / 64), 64>>>(in, out, B, Nt);
  (void)hipDeviceSynchronize();
  printf("bytes read per launch: %zu\n", bytes);
  return 0;
}
