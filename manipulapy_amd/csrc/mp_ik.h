// Batched damped-least-squares inverse kinematics: one lane solves one pose target.
//
// Follows the reference's iterative_inverse_kinematics (kinematics/ik.py:39-311): geometric error (:88-140), damped
// step (:142-162), step cap (:243-246), joint-limit projection (:164-180), best-solution tracking (:196-203,
// :273-280), stagnation restart (:205-213), the optional Levenberg-Marquardt style adaptive damping / step cap
// (:215-231) and the optional five-scale line search (:248-265), iteration count convention (k + 1 on convergence,
// max_iterations + 1 on exhaustion: the for / else at :267-269).
//
// The step: the reference forms  V diag(s / (s^2 + lambda^2 + 1e-12)) U^T e  from an SVD of J.  That vector equals
// J^T (J J^T + (lambda^2 + 1e-12) I)^-1 e, a 6 x 6 SPD system solved here with the in-register Cholesky of mp_core.h -
// no SVD on the device.  With lambda >= 1e-6 the system is well conditioned in float64.
//
// The stagnation restart adds 0.1 * N(0, 1) noise to the best configuration.  The reference draws it from NumPy's
// GLOBAL random stream, so its own runs are not reproducible across call orders; here the noise comes from a counter
// hash of (seed, the problem's own content: target position + initial guess, restart number, joint) - same
// distribution, reproducible, the same for a problem wherever it sits in a batch, not NumPy's numbers.
#pragma once

#include "mp_core.h"

// "no error measured yet": a large finite number, not infinity - the run-time specialised build compiles with
// -ffinite-math-only, under which comparisons against an infinite constant are not dependable
#define MP_IK_BIG 1e300

template <int CAP>
struct MpIkParamsT {
  double eomg, ev, damping, step_cap, w_o, w_p;
  int max_iterations;
  unsigned seed;
  int adaptive_tuning, backtracking;
  double lo[CAP], hi[CAP];
};
typedef MpIkParamsT<MP_MAX_DOF> MpIkParams;     // up to 8 joints: the unrolled kernels
typedef MpIkParamsT<MP_BIG_DOF> MpIkBigParams;  // 9..32 joints: the run-time-n form (MpIkLooped, csrc/mp_dyn.h)

// Where the iteration gets its joint count and its forward kinematics / Jacobian from.  MpIkUnrolled<N>: N is the joint
// count, every loop below unrolls, mp_fk_jac of mp_core.h.  MpIkLooped (csrc/mp_dyn.h): N is the CAPACITY of the per-problem
// arrays (MP_BIG_DOF), the count is the model's, the loops stay loops.
template <int N>
struct MpIkUnrolled {
  template <typename MT>
  MP_HD static constexpr int count(const MT&) { return N; }
  template <bool WANT_J, typename MT>
  MP_HD static void fk(const MT& M, const double (&theta)[N], double (&Tc)[16], double (&J)[6 * N]) {
    MpJointState<double, N> js;
    mp_joint_state<double, N>(M, theta, js);
    mp_fk_jac<double, N, WANT_J>(M, js, Tc, J);
  }
};

// 6-vector [angular, space frame; linear], rotation angle and translation norm between two poses (4x4 row-major)
MP_HD void mp_ik_error(const double (&Tc)[16], const double (&Td)[16], double (&V)[6], double& rot_err, double& trans_err) {
  const double px = Td[3] - Tc[3], py = Td[7] - Tc[7], pz = Td[11] - Tc[11];
  trans_err = sqrt(px * px + py * py + pz * pz);
  double E[9];  // Rc^T Rd
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c) E[3 * r + c] = Tc[r] * Td[c] + Tc[4 + r] * Td[4 + c] + Tc[8 + r] * Td[8 + c];
  double cs = 0.5 * (E[0] + E[4] + E[8] - 1.0);
  cs = cs < -1.0 ? -1.0 : (cs > 1.0 ? 1.0 : cs);
  const double angle = acos(cs);
  rot_err = angle;  // acos >= 0
  const double vee[3] = {E[7] - E[5], E[2] - E[6], E[3] - E[1]};
  double w[3];
  if (angle < 1e-6) {
    w[0] = 0.5 * vee[0]; w[1] = 0.5 * vee[1]; w[2] = 0.5 * vee[2];
  } else if (fabs(angle - 3.14159265358979323846) < 1e-6) {
    const int idx = (E[0] >= E[4] && E[0] >= E[8]) ? 0 : (E[4] >= E[8] ? 1 : 2);  // first maximum of the diagonal
    w[0] = idx == 0 ? angle : 0.0; w[1] = idx == 1 ? angle : 0.0; w[2] = idx == 2 ? angle : 0.0;
  } else {
    const double k = angle / (2.0 * sin(angle) + 1e-10);
    w[0] = k * vee[0]; w[1] = k * vee[1]; w[2] = k * vee[2];
  }
#pragma unroll
  for (int r = 0; r < 3; ++r) V[r] = Tc[4 * r] * w[0] + Tc[4 * r + 1] * w[1] + Tc[4 * r + 2] * w[2];
  V[3] = px; V[4] = py; V[5] = pz;
}

// standard normal from a counter (splitmix64 finaliser + Box-Muller); only the restart path uses it
MP_HD double mp_ik_normal(unsigned seed, unsigned long long key, int restart, int joint) {
  unsigned long long x = (unsigned long long)seed * 0x9E3779B97F4A7C15ull + key * 0xBF58476D1CE4E5B9ull +
                         (unsigned long long)(restart * 64 + joint) * 0x94D049BB133111EBull;
  unsigned long long u[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    x += 0x9E3779B97F4A7C15ull;
    unsigned long long z = x;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    u[i] = z ^ (z >> 31);
  }
  const double a = ((double)(u[0] >> 11) + 1.0) * (1.0 / 9007199254740993.0);  // (0, 1)
  const double b = (double)(u[1] >> 11) * (1.0 / 9007199254740992.0);
  return sqrt(-2.0 * log(a)) * cos(6.283185307179586476925 * b);
}

// Per-problem state of the iteration, so that a lane can interleave "fetch the next problem" with "advance the current
// one" (the work-queue kernel) instead of idling until the slowest problem of its wave is done.
// (The target pose is NOT part of the state: it is re-read from memory every trip - 96 bytes against ~2000
// instructions - which keeps 32 VGPRs free; the kernel is register-bound.)
template <int N>
struct MpIkState {
  double theta[N], best[N];
  double best_err, cur_err;
  double damping, step_cap, nu, prev_err;  // the adaptive-tuning state (constant when the option is off)
  int stall, k, restarts;
};
// (Nor are the success flag - it is the return value of the finishing trip - and the restart-noise key, which is
// re-hashed from memory on the rare restart: the specialised 6-DOF kernel sits exactly at the 256-VGPR line.)

// content hash of a problem (target position + initial guess): keys the restart noise
MP_HD unsigned long long mp_ik_key(int n, const double* Td, const double* theta0) {
  unsigned long long h = 0xCBF29CE484222325ull;  // FNV-1a over the bit patterns
#pragma unroll
  for (int k = 0; k < 3; ++k) h = (h ^ __builtin_bit_cast(unsigned long long, Td[4 * k + 3])) * 0x100000001B3ull;
#pragma unroll
  for (int j = 0; j < n; ++j) h = (h ^ __builtin_bit_cast(unsigned long long, theta0[j])) * 0x100000001B3ull;
  return h;
}

// S.theta (initial guess) must be set
template <int N, typename PT>
MP_HD void mp_ik_begin(MpIkState<N>& S, const PT& P) {
  S.damping = P.damping; S.step_cap = P.step_cap; S.nu = 2.0; S.prev_err = MP_IK_BIG;
#pragma unroll
  for (int j = 0; j < N; ++j) S.best[j] = S.theta[j];
  S.best_err = MP_IK_BIG;
  S.cur_err = MP_IK_BIG;
  S.stall = 0; S.k = 0; S.restarts = 0;
}

// One trip of the reference's loop (kinematics/ik.py:182-269).  Returns 0 while the problem is running, 1 when it finished
// without meeting the tolerances, 2 when it finished successfully; S.theta is then the answer and S.k + 1 the reference's
// iteration count.  Tdp = the target pose, theta0 = the problem's initial guess (both re-read from memory when needed).
template <int N, typename KIN = MpIkUnrolled<N>, typename MT, typename PT>
MP_HD int mp_ik_iterate(const MT& M, const PT& P, MpIkState<N>& S, const double* __restrict__ Tdp,
                        const double* __restrict__ theta0) {
  const int nj = KIN::count(M);  // = N (a constant: the loops unroll) for MpIkUnrolled
  double Td[16];
#pragma unroll
  for (int k = 0; k < 12; ++k) Td[k] = Tdp[k];  // the last row of a pose is never read
  Td[12] = 0.0; Td[13] = 0.0; Td[14] = 0.0; Td[15] = 1.0;
  if (S.k >= P.max_iterations) {  // exhausted (the for / else): fall back to the best configuration seen (:273-280)
    if (S.best_err < S.cur_err) {
#pragma unroll
      for (int j = 0; j < nj; ++j) S.theta[j] = S.best[j];
      double Tc[16], J[6 * N], V[6], rot, tr;
      KIN::template fk<false>(M, S.theta, Tc, J);
      mp_ik_error(Tc, Td, V, rot, tr);
      return (rot < P.eomg && tr < P.ev) ? 2 : 1;
    }
    return 1;
  }
  double Tc[16], J[6 * N], V[6], rot, tr;
  KIN::template fk<true>(M, S.theta, Tc, J);
  mp_ik_error(Tc, Td, V, rot, tr);
  S.cur_err = rot + tr;
  if (rot < P.eomg && tr < P.ev) return 2;
  if (S.cur_err < S.best_err) {
    S.best_err = S.cur_err;
    S.stall = 0;
#pragma unroll
    for (int j = 0; j < nj; ++j) S.best[j] = S.theta[j];
  } else {
    ++S.stall;
  }
  if (S.stall > 20) {
    const unsigned long long key = mp_ik_key(nj, Tdp, theta0);
#pragma unroll
    for (int j = 0; j < nj; ++j) {
      const double t = S.best[j] + 0.1 * mp_ik_normal(P.seed, key, S.restarts, j);
      S.theta[j] = t < P.lo[j] ? P.lo[j] : (t > P.hi[j] ? P.hi[j] : t);
    }
    S.stall = 0;
    S.damping = P.damping;  // the restart resets the damping and its growth factor, not the step cap (:209-212)
    S.nu = 2.0;
    ++S.restarts;
    ++S.k;
    return 0;
  }
  if (P.adaptive_tuning && S.k > 0) {
    if (S.cur_err < S.prev_err * 0.75) {  // good progress: towards Newton
      S.damping = fmax(1e-6, S.damping / 3.0);
      S.step_cap = fmin(P.step_cap * 1.5, S.step_cap * 1.2);
      S.nu = 2.0;
    } else if (S.cur_err < S.prev_err * 0.95) {
      S.damping = fmax(1e-6, S.damping / 1.5);
    } else if (S.cur_err > S.prev_err) {  // got worse: towards gradient descent
      S.damping = fmin(0.5, S.damping * S.nu);
      S.nu = fmin(S.nu * 1.5, 8.0);
      S.step_cap = fmax(0.01, S.step_cap * 0.7);
    }
  }
  S.prev_err = S.cur_err;
  const double lam = S.damping * S.damping + 1e-12;
  double A[6][6], y[6];
#pragma unroll
  for (int r = 0; r < 6; ++r) {
    y[r] = V[r] * (r < 3 ? P.w_o : P.w_p);
#pragma unroll
    for (int c = 0; c <= r; ++c) {
      double s = (r == c) ? lam : 0.0;
#pragma unroll
      for (int j = 0; j < nj; ++j) s += J[r * nj + j] * J[c * nj + j];
      A[r][c] = s;
      A[c][r] = s;
    }
  }
  mp_spd_solve<double, 6>(A, y);
  double d[N], nd = 0.0;
#pragma unroll
  for (int j = 0; j < nj; ++j) {
    double s = 0.0;
#pragma unroll
    for (int r = 0; r < 6; ++r) s += J[r * nj + j] * y[r];
    d[j] = s;
    nd += s * s;
  }
  nd = sqrt(nd);
  const double scale = nd > S.step_cap ? S.step_cap / nd : 1.0;
  if (P.backtracking) {  // try five scales of the step, keep the best pose error (:248-265)
    const double scales[5] = {1.0, 0.5, 0.25, 0.125, 0.75};
    double keep[N], keep_err = S.cur_err;
#pragma unroll
    for (int j = 0; j < nj; ++j) keep[j] = S.theta[j];
    for (int c = 0; c < 5; ++c) {
      double cand[N];
#pragma unroll
      for (int j = 0; j < nj; ++j) {
        const double t = S.theta[j] + scales[c] * (d[j] * scale);
        cand[j] = t < P.lo[j] ? P.lo[j] : (t > P.hi[j] ? P.hi[j] : t);
      }
      double Tt[16], Jt[6 * N], Vt[6], rt, tt;
      KIN::template fk<false>(M, cand, Tt, Jt);
      mp_ik_error(Tt, Td, Vt, rt, tt);
      if (rt + tt < keep_err) {
        keep_err = rt + tt;
#pragma unroll
        for (int j = 0; j < nj; ++j) keep[j] = cand[j];
      }
    }
    if (keep_err < S.cur_err * 1.1) {
#pragma unroll
      for (int j = 0; j < nj; ++j) S.theta[j] = keep[j];
    } else {  // every scale failed by more than 10 %: a small step anyway
#pragma unroll
      for (int j = 0; j < nj; ++j) {
        const double t = S.theta[j] + 0.1 * (d[j] * scale);
        S.theta[j] = t < P.lo[j] ? P.lo[j] : (t > P.hi[j] ? P.hi[j] : t);
      }
    }
  } else {
#pragma unroll
    for (int j = 0; j < nj; ++j) {
      const double t = S.theta[j] + d[j] * scale;
      S.theta[j] = t < P.lo[j] ? P.lo[j] : (t > P.hi[j] ? P.hi[j] : t);
    }
  }
  ++S.k;
  return 0;
}

// theta: in = initial guess, out = solution.  Returns the reference's iteration count; sets success / restarts.
template <int N, typename KIN = MpIkUnrolled<N>, typename MT, typename PT>
MP_HD int mp_ik_solve(const MT& M, const PT& P, const double (&Td)[16], double (&theta)[N], int& success, int& restarts) {
  MpIkState<N> S;
  double theta0[N];
#pragma unroll
  for (int j = 0; j < N; ++j) { S.theta[j] = theta[j]; theta0[j] = theta[j]; }
  mp_ik_begin(S, P);
  int done;
  while (!(done = mp_ik_iterate<N, KIN>(M, P, S, Td, theta0))) {}
#pragma unroll
  for (int j = 0; j < N; ++j) theta[j] = S.theta[j];
  success = done == 2 ? 1 : 0;
  restarts = S.restarts;
  return S.k + 1;
}
