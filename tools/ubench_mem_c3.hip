// Dev microbenchmark: HBM ceiling for config c3's traffic mix (per row: read 3 x 56 B, write 56 + 128 + 336 B,
// float64) in (a) the kernel's lane-owns-a-row layout and (b) a perfectly coalesced layout of the same bytes.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d2 __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(256) void k_rows(const double* q, const double* qd, const double* qdd, double* T, double* J, double* tau, long rows) {
  long r = (long)blockIdx.x * 256 + threadIdx.x;
  if (r >= rows) return;
  double a[7], b[7], c[7];
  for (int k = 0; k < 7; ++k) { a[k] = q[r * 7 + k]; b[k] = qd[r * 7 + k]; c[k] = qdd[r * 7 + k]; }
  double s = 0; for (int k = 0; k < 7; ++k) s += a[k] * b[k] + c[k];
  d2* Tp = (d2*)(T + r * 16); for (int k = 0; k < 8; ++k) Tp[k] = (d2){s + k, s - k};
  d2* Jp = (d2*)(J + r * 42); for (int k = 0; k < 21; ++k) Jp[k] = (d2){s * k, s + 2 * k};
  for (int k = 0; k < 7; ++k) tau[r * 7 + k] = s + a[k];
}
__global__ __launch_bounds__(256) void k_coal(const d2* in, d2* out, long n_in, long n_out) {
  long i = (long)blockIdx.x * 256 + threadIdx.x;
  long stride = (long)gridDim.x * 256;
  d2 acc = {0, 0};
  for (long k = i; k < n_in; k += stride) acc += in[k];
  for (long k = i; k < n_out; k += stride) out[k] = acc + (d2){(double)k, 1.0};
}
__global__ __launch_bounds__(256) void k_write(d2* out, long n_out) {
  long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i < n_out) out[i] = (d2){(double)i, 1.0};
}
int main() {
  const long rows = 32768000L;
  double *q, *qd, *qdd, *T, *J, *tau;
  hipMalloc(&q, rows * 56); hipMalloc(&qd, rows * 56); hipMalloc(&qdd, rows * 56);
  hipMalloc(&T, rows * 128); hipMalloc(&J, rows * 336); hipMalloc(&tau, rows * 56);
  hipMemset(q, 0, rows * 56); hipMemset(qd, 0, rows * 56); hipMemset(qdd, 0, rows * 56);
  // dedicated, exactly sized buffers for the coalesced variant: in = rows*168 B, out = rows*520 B
  d2 *cin, *cout;
  const long n_in = rows * 168 / 16, n_out = rows * 520 / 16;
  if (hipMalloc(&cin, n_in * 16) != hipSuccess || hipMalloc(&cout, n_out * 16) != hipSuccess) { printf("alloc failed\n"); return 1; }
  hipMemset(cin, 0, n_in * 16);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const double bytes = rows * (168.0 + 520.0);
  auto timeit = [&](const char* name, double nbytes, auto launch) {
    launch(); launch();
    hipEventRecord(e0); for (int i = 0; i < 5; ++i) launch(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    printf("%-28s %.3f ms  %.0f GB/s\n", name, ms, nbytes / (ms * 1e-3) / 1e9);
  };
  timeit("c3 pattern (lane owns row)", bytes, [&] { k_rows<<<(rows + 255) / 256, 256>>>(q, qd, qdd, T, J, tau, rows); });
  timeit("same bytes, coalesced", bytes, [&] { k_coal<<<256 * 16, 256>>>(cin, cout, n_in, n_out); });
  timeit("write only 11 GB coalesced", rows * 336.0, [&] { k_write<<<(unsigned)((rows * 21 + 255) / 256), 256>>>((d2*)J, rows * 21); });  // J holds rows*336 B = rows*21 d2
  return 0;
}
