// STUDY TOOL (not product, not test): per row, what the float32 inverse-dynamics rows' conditioning test could look at.
// Instantiates the kernels' own templates (csrc/mp_core.h) on the host, as tests/hostsim does, with a recording policy in
// the backward pass: for every row the float32 torques, the float64 torques from the same float32 inputs (what the float64
// pass stores) and the largest component of every joint's force and moment.  tools/rule_sweep.py evaluates candidate rules on that.
//   clang++ -O2 -std=c++17 -ffp-contract=fast -fopenmp -shared -fPIC -o /tmp/librule_sweep.so tools/rule_sweep.cpp \
//       manipulapy_amd/csrc/mp_model_compile.cpp
#include <cstring>

#include "../manipulapy_amd/csrc/mp_core.h"
#include "../manipulapy_amd/csrc/mp_model_compile.h"

namespace {
template <int N>
struct Recorder {
  float f[N], m[N], bm[N], bf[N], cm[N], cf[N];
  static float inf3(float a, float b, float c) { return mp_max(mp_max(mp_abs(a), mp_abs(b)), mp_abs(c)); }
  void body(int i, const float& nx, const float& ny, const float& nz, const float& fx, const float& fy, const float& fz) { bm[i] = inf3(nx, ny, nz); bf[i] = inf3(fx, fy, fz); }
  void child(int i, const float& nx, const float& ny, const float& nz, const float& fx, const float& fy, const float& fz) { cm[i] = inf3(nx, ny, nz); cf[i] = inf3(fx, fy, fz); }
  void joint(int i, const float& nx, const float& ny, const float& nz, const float& fx, const float& fy, const float& fz) {
    f[i] = mp_max(mp_max(mp_abs(fx), mp_abs(fy)), mp_abs(fz));
    m[i] = mp_max(mp_max(mp_abs(nx), mp_abs(ny)), mp_abs(nz));
  }
};

template <int N>
void run(const MpModel<double>& Md, const MpCall<double>& Cd, long rows, const float* q, const float* qd, const float* qdd, float* tau32,
         float* tau64, float* fmax, float* mmax, float* extra) {
  MpModel<float> M;
  MpCall<float> C;
  mp_model_cast(Md, &M);
  mp_call_cast(Cd, &C);
  const float tn[3] = {0, 0, 0}, tf[3] = {0, 0, 0};
#pragma omp parallel for schedule(static)
  for (long r = 0; r < rows; ++r) {
    float a[N], b[N], c[N], t[N];
    for (int j = 0; j < N; ++j) { a[j] = q[r * N + j]; b[j] = qd[r * N + j]; c[j] = qdd[r * N + j]; }
    MpJointState<float, N> js;
    mp_joint_state<float, N>(M, a, js);
    Recorder<N> rec;
    mp_rnea_impl<float, N, false>(M, C.a0, tn, tf, js, b, c, t, rec);
    double ad[N], bd[N], cd[N], td[N];
    for (int j = 0; j < N; ++j) { ad[j] = (double)a[j]; bd[j] = (double)b[j]; cd[j] = (double)c[j]; }
    MpJointState<double, N> jsd;
    mp_joint_state<double, N>(Md, ad, jsd);
    mp_rnea<double, N, false>(Md, Cd, jsd, bd, cd, td);
    for (int j = 0; j < N; ++j) { tau32[r * N + j] = t[j]; tau64[r * N + j] = (float)td[j]; fmax[r * N + j] = rec.f[j]; mmax[r * N + j] = rec.m[j];
      extra[(r * 4 + 0) * N + j] = rec.bm[j]; extra[(r * 4 + 1) * N + j] = rec.bf[j];
      extra[(r * 4 + 2) * N + j] = j ? rec.cm[j] : 0.f; extra[(r * 4 + 3) * N + j] = j ? rec.cf[j] : 0.f; }
  }
}
}  // namespace

extern "C" int rule_sweep(int n, const double* S, const double* Mcom, const double* G, const double* M_ee, const double* joint_limits,
                          const double* g, long rows, const float* q, const float* qd, const float* qdd, float* tau32, float* tau64,
                          float* fmax, float* mmax, float* extra, double* lever, char* err, long errlen) {
  MpModel<double> Md;
  int rc = mp_compile_model(n, S, Mcom, G, M_ee, joint_limits, nullptr, &Md, err, (size_t)errlen);
  if (rc) return rc;
  for (int i = 0; i < n; ++i) { lever[2 * i] = Md.j[i].a; lever[2 * i + 1] = Md.j[i].d; }
  lever[2 * n] = Md.lscale;
  MpCall<double> Cd;
  mp_make_call(Md, g, nullptr, &Cd);
  switch (n) {
#define CASE(N) case N: run<N>(Md, Cd, rows, q, qd, qdd, tau32, tau64, fmax, mmax, extra); return 0;
    CASE(1) CASE(2) CASE(3) CASE(4) CASE(5) CASE(6) CASE(7) CASE(8)
#undef CASE
  }
  return 1;
}
