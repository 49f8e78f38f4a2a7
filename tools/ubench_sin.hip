// Dev probe: accuracy of the hardware v_sin_f32 / v_cos_f32 (argument in revolutions) against double-precision sin / cos
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
__global__ void k(const float* x, float* s, float* c, int n) {
  int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float r = x[i] * 0.15915494309189535f;
  r = __builtin_amdgcn_fractf(r);
  s[i] = __builtin_amdgcn_sinf(r);
  c[i] = __builtin_amdgcn_cosf(r);
}
int main() {
  const int n = 1 << 22;
  std::vector<float> hx(n), hs(n), hc(n);
  for (int i = 0; i < n; ++i) hx[i] = -7.0f + 14.0f * (float)i / (float)(n - 1);
  float *dx, *ds, *dc;
  hipMalloc(&dx, n * 4); hipMalloc(&ds, n * 4); hipMalloc(&dc, n * 4);
  hipMemcpy(dx, hx.data(), n * 4, hipMemcpyHostToDevice);
  k<<<n / 256, 256>>>(dx, ds, dc, n);
  hipMemcpy(hs.data(), ds, n * 4, hipMemcpyDeviceToHost);
  hipMemcpy(hc.data(), dc, n * 4, hipMemcpyDeviceToHost);
  double es = 0, ec = 0, en = 0; int is = 0;
  for (int i = 0; i < n; ++i) {
    double a = std::fabs((double)hs[i] - std::sin((double)hx[i])), b = std::fabs((double)hc[i] - std::cos((double)hx[i]));
    if (a > es) { es = a; is = i; }
    if (b > ec) ec = b;
    double nn = std::fabs((double)hs[i] * hs[i] + (double)hc[i] * hc[i] - 1.0); if (nn > en) en = nn;
  }
  printf("max abs err sin %.3e (at x = %.6f) cos %.3e  |s^2 + c^2 - 1| %.3e\n", es, hx[is], ec, en);
  // small-angle region
  double e0 = 0;
  for (int i = 0; i < n; ++i) if (std::fabs(hx[i]) < 0.01f) { double a = std::fabs((double)hs[i] - std::sin((double)hx[i])); if (a > e0) e0 = a; }
  printf("max abs err sin for |x| < 0.01: %.3e\n", e0);
  return 0;
}
