// TEST INFRASTRUCTURE — host instantiation of the device math (manipulapy_amd/csrc/mp_core.h) so the
// GPU-less CI box can check the model compiler + recursion against the oracle.  Never linked into or
// called by the product library; the product path only runs these templates inside HIP kernels.
#include <cstring>

#include "../../manipulapy_amd/csrc/mp_core.h"
#include "../../manipulapy_amd/csrc/mp_ik.h"
#include "../../manipulapy_amd/csrc/mp_model_compile.h"

namespace {
template <typename T, int N>
void run_id(const MpModel<T>& M, const MpCall<T>& C, bool ftip, long rows, const double* q, const double* qd,
            const double* qdd, double* tau, double* Tout, double* Jout) {
  for (long r = 0; r < rows; ++r) {
    T a[N], b[N], c[N], t[N];
    for (int j = 0; j < N; ++j) { a[j] = (T)q[r * N + j]; b[j] = (T)qd[r * N + j]; c[j] = (T)qdd[r * N + j]; }
    MpJointState<T, N> js;
    mp_joint_state<T, N>(M, a, js);
    if (ftip) mp_rnea<T, N, true>(M, C, js, b, c, t);
    else mp_rnea<T, N, false>(M, C, js, b, c, t);
    for (int j = 0; j < N; ++j) tau[r * N + j] = (double)mp_clip(t[j], M.taumin[j], M.taumax[j]);
    if (Tout) {
      T TT[16], JJ[6 * N];
      mp_fk_jac<T, N, true>(M, js, TT, JJ);
      for (int k = 0; k < 16; ++k) Tout[r * 16 + k] = (double)TT[k];
      for (int k = 0; k < 6 * N; ++k) Jout[r * 6 * N + k] = (double)JJ[k];
    }
  }
}
#if MP_HAS_PACKED
// two rows per "lane", exactly as k_id_pk does on the device
template <int N>
void run_id_packed(const MpModel<float>& M, const MpCall<float>& C, bool ftip, long rows, const double* q,
                   const double* qd, const double* qdd, double* tau) {
  for (long r = 0; r < rows; r += 2) {
    const long r1 = (r + 1 < rows) ? r + 1 : r;
    mp_f2 a[N], b[N], c[N], t[N];
    for (int j = 0; j < N; ++j) {
      a[j] = (mp_f2){(float)q[r * N + j], (float)q[r1 * N + j]};
      b[j] = (mp_f2){(float)qd[r * N + j], (float)qd[r1 * N + j]};
      c[j] = (mp_f2){(float)qdd[r * N + j], (float)qdd[r1 * N + j]};
    }
    MpJointState<mp_f2, N> js;
    mp_joint_state<mp_f2, N>(M, a, js);
    if (ftip) mp_rnea<mp_f2, N, true>(M, C, js, b, c, t);
    else mp_rnea<mp_f2, N, false>(M, C, js, b, c, t);
    for (int j = 0; j < N; ++j) {
      const mp_f2 v = mp_clip(t[j], M.taumin[j], M.taumax[j]);
      tau[r * N + j] = (double)v.x;
      tau[r1 * N + j] = (double)((r1 == r) ? v.x : v.y);
    }
  }
}
int run_packed(const MpModel<double>& Md, const MpCall<double>& Cd, bool ftip, long rows, const double* q,
               const double* qd, const double* qdd, double* tau) {
  MpModel<float> M;
  MpCall<float> C;
  mp_model_cast(Md, &M);
  mp_call_cast(Cd, &C);
  switch (Md.n) {
#define CASE(N) case N: run_id_packed<N>(M, C, ftip, rows, q, qd, qdd, tau); return 0;
    CASE(1) CASE(2) CASE(3) CASE(4) CASE(5) CASE(6) CASE(7) CASE(8)
#undef CASE
  }
  return 1;
}
#endif

template <typename T>
int run(const MpModel<double>& Md, const MpCall<double>& Cd, bool ftip, long rows, const double* q, const double* qd,
        const double* qdd, double* tau, double* Tout, double* Jout) {
  MpModel<T> M;
  MpCall<T> C;
  mp_model_cast(Md, &M);
  mp_call_cast(Cd, &C);
  switch (Md.n) {
#define CASE(N) case N: run_id<T, N>(M, C, ftip, rows, q, qd, qdd, tau, Tout, Jout); return 0;
    CASE(1) CASE(2) CASE(3) CASE(4) CASE(5) CASE(6) CASE(7) CASE(8)
#undef CASE
  }
  return 1;
}
}  // namespace

namespace {
// forward-dynamics pieces on the host: mode 0 = mass matrix (out: rows x n x n), 1 = forward dynamics
// (out: rows x n), 2 = one trajectory roll-out like k_fd_traj (out: 3 x N x n, float32 rounded rows)
template <typename T, int N>
void run_fd(const MpModel<T>& M, const MpCall<T>& C, int mode, long rows, const double* q, const double* qd,
            const double* tau, const double* Ftipmat, double dt, int intRes, double* out) {
  if (mode == 0 || mode == 3) {  // 0: n unit-acceleration recursions, 3: composite-rigid-body algorithm
    for (long r = 0; r < rows; ++r) {
      T a[N];
      for (int j = 0; j < N; ++j) a[j] = (T)q[r * N + j];
      MpJointState<T, N> js;
      mp_joint_state<T, N>(M, a, js);
      T Mq[N][N];
      if (mode == 0) mp_mass_matrix<T, N>(M, js, Mq);
      else mp_mass_matrix_crba<T, N>(M, js, Mq);
      for (int i = 0; i < N; ++i)
        for (int j = 0; j < N; ++j) out[(r * N + i) * N + j] = (double)Mq[i][j];
    }
  } else if (mode == 1) {
    const T tn[3] = {C.F1n[0], C.F1n[1], C.F1n[2]}, tf[3] = {C.F1f[0], C.F1f[1], C.F1f[2]};
    for (long r = 0; r < rows; ++r) {
      T a[N], b[N], t[N], o[N];
      for (int j = 0; j < N; ++j) { a[j] = (T)q[r * N + j]; b[j] = (T)qd[r * N + j]; t[j] = (T)tau[r * N + j]; }
      mp_forward_dynamics<T, N, true>(M, C.a0, tn, tf, a, b, t, o);
      for (int j = 0; j < N; ++j) out[r * N + j] = (double)o[j];
    }
  } else {
    const long Nt = rows;  // rows = timesteps of ONE trajectory; q / qd hold the initial state
    T a[N], b[N];
    for (int j = 0; j < N; ++j) { a[j] = (T)q[j]; b[j] = (T)qd[j]; }
    double* pos = out; double* vel = out + Nt * N; double* acc = out + 2 * Nt * N;
    for (int j = 0; j < N; ++j) { pos[j] = (float)a[j]; vel[j] = (float)b[j]; acc[j] = 0; }
    const T h = intRes > 0 ? (T)(dt / intRes) : (T)0;
    for (long i = 1; i < Nt; ++i) {
      T t[N], tn[3] = {0, 0, 0}, tf[3] = {0, 0, 0}, last[N];
      for (int j = 0; j < N; ++j) { t[j] = (T)tau[i * N + j]; last[j] = 0; }
      if (Ftipmat) {
        T F[6];
        for (int k = 0; k < 6; ++k) F[k] = (T)Ftipmat[i * 6 + k];
        mp_wrench_to_frame1(M, F, tn, tf);
      }
      for (int s = 0; s < intRes; ++s) {
        mp_forward_dynamics<T, N, true>(M, C.a0, tn, tf, a, b, t, last);
        for (int j = 0; j < N; ++j) {
          b[j] = b[j] + last[j] * h;
          a[j] = mp_clip(a[j] + b[j] * h, M.qmin[j], M.qmax[j]);
        }
      }
      for (int j = 0; j < N; ++j) { pos[i * N + j] = (float)a[j]; vel[i * N + j] = (float)b[j]; acc[i * N + j] = (float)last[j]; }
    }
  }
}
template <typename T>
int run_fd_t(const MpModel<double>& Md, const MpCall<double>& Cd, int mode, long rows, const double* q, const double* qd,
             const double* tau, const double* Ftipmat, double dt, int intRes, double* out) {
  MpModel<T> M;
  MpCall<T> C;
  mp_model_cast(Md, &M);
  mp_call_cast(Cd, &C);
  switch (Md.n) {
#define CASE(N) case N: run_fd<T, N>(M, C, mode, rows, q, qd, tau, Ftipmat, dt, intRes, out); return 0;
    CASE(1) CASE(2) CASE(3) CASE(4) CASE(5) CASE(6) CASE(7) CASE(8)
#undef CASE
  }
  return 1;
}
}  // namespace

extern "C" int hostsim_fd(int n, const double* S, const double* Mcom, const double* G, const double* M_ee,
                          const double* joint_limits, const double* g, const double* Ftip, int mode, long rows,
                          const double* q, const double* qd, const double* tau, const double* Ftipmat, double dt, int intRes,
                          int use_f32, double* out, char* err, long errlen) {
  MpModel<double> Md;
  int rc = mp_compile_model(n, S, Mcom, G, M_ee, joint_limits, nullptr, &Md, err, (size_t)errlen);
  if (rc) return rc;
  MpCall<double> Cd;
  mp_make_call(Md, g, Ftip, &Cd);
  return use_f32 ? run_fd_t<float>(Md, Cd, mode, rows, q, qd, tau, Ftipmat, dt, intRes, out)
                 : run_fd_t<double>(Md, Cd, mode, rows, q, qd, tau, Ftipmat, dt, intRes, out);
}

extern "C" int hostsim_run(int n, const double* S, const double* Mcom, const double* G, const double* M_ee,
                           const double* joint_limits, const double* torque_limits, const double* g,
                           const double* Ftip, long rows, const double* q, const double* qd, const double* qdd,
                           int use_f32, double* tau, double* Tout, double* Jout, double* params_out, char* err,
                           long errlen) {
  MpModel<double> Md;
  int rc = mp_compile_model(n, S, Mcom, G, M_ee, joint_limits, torque_limits, &Md, err, (size_t)errlen);
  if (rc) return rc;
  if (params_out)
    for (int i = 0; i < n; ++i) std::memcpy(params_out + 16 * i, &Md.j[i], 16 * sizeof(double));
  MpCall<double> Cd;
  mp_make_call(Md, g, Ftip, &Cd);
  bool ftip = false;
  if (Ftip)
    for (int k = 0; k < 6; ++k) ftip |= (Ftip[k] != 0.0);
#if MP_HAS_PACKED
  if (use_f32 == 2) return run_packed(Md, Cd, ftip, rows, q, qd, qdd, tau);  // float32, two rows per lane
#endif
  return use_f32 ? run<float>(Md, Cd, ftip, rows, q, qd, qdd, tau, Tout, Jout)
                 : run<double>(Md, Cd, ftip, rows, q, qd, qdd, tau, Tout, Jout);
}

// Cartesian straight-line trajectory (mp_cartesian_point), N points between one pose pair
extern "C" int hostsim_cartesian(const double* Xs, const double* Xe, long N, double Tf, int method, float* pos, float* vel,
                                 float* acc, float* ori) {
  double A[16], B[16];
  for (int k = 0; k < 16; ++k) { A[k] = Xs[k]; B[k] = Xe[k]; }
  for (long i = 0; i < N; ++i) {
    float p[3], v[3], a[3], o[9];
    mp_cartesian_point(A, B, i, N, Tf, method, p, v, a, o);
    for (int k = 0; k < 3; ++k) { pos[i * 3 + k] = p[k]; vel[i * 3 + k] = v[k]; acc[i * 3 + k] = a[k]; }
    for (int k = 0; k < 9; ++k) ori[i * 9 + k] = o[k];
  }
  return 0;
}

// inverse kinematics (mp_ik_solve), `rows` independent problems:
// params = {eomg, ev, max_iterations, damping, step_cap, w_o, w_p, adaptive_tuning, backtracking}
namespace {
template <int N>
void run_ik(const MpModel<double>& M, const MpIkParams& P, long rows, const double* Td, const double* th0, double* th, int* ok,
            int* iters, int* restarts) {
  for (long r = 0; r < rows; ++r) {
    double T[16], q[N];
    for (int k = 0; k < 16; ++k) T[k] = Td[r * 16 + k];
    for (int j = 0; j < N; ++j) q[j] = th0[r * N + j];
    iters[r] = mp_ik_solve<N>(M, P, T, q, ok[r], restarts[r]);
    for (int j = 0; j < N; ++j) th[r * N + j] = q[j];
  }
}
}  // namespace
extern "C" int hostsim_ik(int n, const double* S, const double* Mcom, const double* G, const double* M_ee, const double* limits,
                          const double* params, long rows, const double* Td, const double* th0, double* th, int* ok, int* iters,
                          int* restarts, char* err, long errlen) {
  MpModel<double> Md;
  int rc = mp_compile_model(n, S, Mcom, G, M_ee, nullptr, nullptr, &Md, err, (size_t)errlen);
  if (rc) return rc;
  MpIkParams P;
  P.eomg = params[0]; P.ev = params[1]; P.max_iterations = (int)params[2]; P.damping = params[3]; P.step_cap = params[4];
  P.w_o = params[5]; P.w_p = params[6]; P.seed = 1234u;
  P.adaptive_tuning = params[7] != 0.0; P.backtracking = params[8] != 0.0;
  for (int j = 0; j < MP_MAX_DOF; ++j) { P.lo[j] = j < n ? limits[2 * j] : -HUGE_VAL; P.hi[j] = j < n ? limits[2 * j + 1] : HUGE_VAL; }
  switch (n) {
#define CASE(N) case N: run_ik<N>(Md, P, rows, Td, th0, th, ok, iters, restarts); return 0;
    CASE(1) CASE(2) CASE(3) CASE(4) CASE(5) CASE(6) CASE(7) CASE(8)
#undef CASE
  }
  return 1;
}
