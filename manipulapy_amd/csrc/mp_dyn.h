// Run-time-n ("looped") forms of the per-row recursions of mp_core.h, for robots with more joints than the fully
// unrolled kernels are instantiated for (MP_MAX_DOF = 8 < n <= MP_BIG_DOF = 32: the reference's Jaco arms with their
// three-finger hands have 9 and 10 actuated joints, ManipulaPy_data/__init__.py:174-189, and its algorithms loop over any n,
// dynamics/mass_matrix.py:62-96, kinematics/jacobian.py:62-73).  Same compiled link frames, same axis-aligned steps
// (mp_motion_* / mp_force_* / mp_rbi_up_* of mp_core.h), same arithmetic order per joint; the joint index is a run-time loop
// variable, so per-joint state lives in indexed arrays (scratch memory on the device) and the model is read joint by joint
// through a pointer.  Speed is not the point here - any supported robot must compute, on the GPU and on the CPU launchers.
// Header-only; compiles for gfx950 and, unchanged, for the host (csrc/mp_cpu.cpp, tests/hostsim).
#pragma once

#include "mp_core.h"

#if defined(__clang__)
#define MP_NOUNROLL _Pragma("nounroll")
#else
#define MP_NOUNROLL
#endif

// CAP: the capacity of a row's per-joint arrays - MP_MID_DOF (16) or MP_BIG_DOF (32), picked by the launchers from the joint count,
// so that the 9..16-joint robots do not pay for a 32 x 32 mass matrix of scratch memory per lane.
template <typename T, int CAP>
struct MpDynState {  // sin / cos of the joint angles and the z shifts of one row
  T s[CAP], c[CAP], d[CAP];
};

template <typename T, typename MT, int CAP>
MP_HD void mp_dyn_joint_state(const MT& M, int n, const T* q, MpDynState<T, CAP>& js) {
  MP_NOUNROLL
  for (int i = 0; i < n; ++i) {
    const auto& J = M.j[i];
    const T qr = J.rev * q[i];
    T s0, c0;  // the offset enters as a constant rotation of (sin q, cos q), never as a sum with q (MpJoint::co / so)
    mp_sincos(qr, s0, c0);
    js.s[i] = s0 * J.co + c0 * J.so;
    js.c[i] = c0 * J.co - s0 * J.so;
    js.d[i] = J.d + (q[i] - qr);
  }
}

// mp_rnea of mp_core.h with a run-time joint count.  tau is NOT clipped here.
// `scale` (optional): the size of the row's large intermediate terms, the statistic of MpRowScale (mp_core.h) with run-time indices -
// the own body moment of link J = min(1, n - 1), the moment joint J + 1 hands down, lscale x the force through joint min(2, n - 1).
template <typename T, bool HAS_FTIP, typename MT, int CAP>
MP_HD void mp_dyn_rnea(const MT& M, int n, const T (&a0)[3], const T (&tipn)[3], const T (&tipf)[3], const MpDynState<T, CAP>& js,
                       const T* qd, const T* qdd, T* tau, T* scale = nullptr) {
  const int sJ = n > 1 ? 1 : 0, sJF = n > 2 ? 2 : n - 1;
  T s_bm = 0, s_cm = 0, s_f = 0;
  T fnx[CAP], fny[CAP], fnz[CAP], ffx[CAP], ffy[CAP], ffz[CAP];
  T wx = 0, wy = 0, wz = 0, vx = 0, vy = 0, vz = 0;
  T dwx = 0, dwy = 0, dwz = 0, dvx = a0[0], dvy = a0[1], dvz = a0[2];
  T tnx = 0, tny = 0, tnz = 0, tfx = 0, tfy = 0, tfz = 0;
  if (HAS_FTIP) { tnx = tipn[0]; tny = tipn[1]; tnz = tipn[2]; tfx = tipf[0]; tfy = tipf[1]; tfz = tipf[2]; }
  MP_NOUNROLL
  for (int i = 0; i < n; ++i) {  // forward pass: twists, accelerations, body wrenches
    const auto& J = M.j[i];
    if (i > 0) {
      mp_motion_A(J.ca, J.sa, J.a, wx, wy, wz, vx, vy, vz);
      mp_motion_A(J.ca, J.sa, J.a, dwx, dwy, dwz, dvx, dvy, dvz);
      if (HAS_FTIP) mp_force_down_A(J.ca, J.sa, J.a, tnx, tny, tnz, tfx, tfy, tfz);
    }
    const T c = js.c[i], s = js.s[i], d = js.d[i];
    mp_motion_B(c, s, d, wx, wy, wz, vx, vy, vz);
    mp_motion_B(c, s, d, dwx, dwy, dwz, dvx, dvy, dvz);
    if (HAS_FTIP) mp_force_down_B(c, s, d, tnx, tny, tnz, tfx, tfy, tfz);
    const T qdr = J.rev * qd[i], qdp = qd[i] - qdr;
    const T ar = J.rev * qdd[i], ap = qdd[i] - ar;
    wz += qdr;
    vz += qdp;
    dwx += qdr * wy;
    dwy -= qdr * wx;
    dwz += ar;
    dvx += qdr * vy + qdp * wy;
    dvy -= qdr * vx + qdp * wx;
    dvz += ap;
    const T pnx = J.Ixx * wx + J.Ixy * wy + J.Ixz * wz + (J.hy * vz - J.hz * vy);
    const T pny = J.Ixy * wx + J.Iyy * wy + J.Iyz * wz + (J.hz * vx - J.hx * vz);
    const T pnz = J.Ixz * wx + J.Iyz * wy + J.Izz * wz + (J.hx * vy - J.hy * vx);
    const T pfx = J.m * vx - (J.hy * wz - J.hz * wy);
    const T pfy = J.m * vy - (J.hz * wx - J.hx * wz);
    const T pfz = J.m * vz - (J.hx * wy - J.hy * wx);
    fnx[i] = J.Ixx * dwx + J.Ixy * dwy + J.Ixz * dwz + (J.hy * dvz - J.hz * dvy) + (wy * pnz - wz * pny) + (vy * pfz - vz * pfy);
    fny[i] = J.Ixy * dwx + J.Iyy * dwy + J.Iyz * dwz + (J.hz * dvx - J.hx * dvz) + (wz * pnx - wx * pnz) + (vz * pfx - vx * pfz);
    fnz[i] = J.Ixz * dwx + J.Iyz * dwy + J.Izz * dwz + (J.hx * dvy - J.hy * dvx) + (wx * pny - wy * pnx) + (vx * pfy - vy * pfx);
    ffx[i] = J.m * dvx - (J.hy * dwz - J.hz * dwy) + (wy * pfz - wz * pfy);
    ffy[i] = J.m * dvy - (J.hz * dwx - J.hx * dwz) + (wz * pfx - wx * pfz);
    ffz[i] = J.m * dvz - (J.hx * dwy - J.hy * dwx) + (wx * pfy - wy * pfx);
    if (scale && i == sJ) s_bm = mp_max(mp_max(mp_abs(fnx[i]), mp_abs(fny[i])), mp_abs(fnz[i]));
  }
  if (HAS_FTIP) {
    fnx[n - 1] += tnx; fny[n - 1] += tny; fnz[n - 1] += tnz;
    ffx[n - 1] += tfx; ffy[n - 1] += tfy; ffz[n - 1] += tfz;
  }
  MP_NOUNROLL
  for (int i = n - 1; i >= 0; --i) {  // backward pass
    const auto& J = M.j[i];
    tau[i] = J.rev * fnz[i] + (T(1) - J.rev) * ffz[i];
    if (scale && i == sJF) s_f = mp_max(mp_max(mp_abs(ffx[i]), mp_abs(ffy[i])), mp_abs(ffz[i]));
    if (i > 0) {
      T nx = fnx[i], ny = fny[i], nz = fnz[i], fx = ffx[i], fy = ffy[i], fz = ffz[i];
      mp_force_up_B(js.c[i], js.s[i], js.d[i], nx, ny, nz, fx, fy, fz);
      mp_force_up_A(J.ca, J.sa, J.a, nx, ny, nz, fx, fy, fz);
      if (scale && i == sJ + 1) s_cm = mp_max(mp_max(mp_abs(nx), mp_abs(ny)), mp_abs(nz));
      fnx[i - 1] += nx; fny[i - 1] += ny; fnz[i - 1] += nz;
      ffx[i - 1] += fx; ffy[i - 1] += fy; ffz[i - 1] += fz;
    }
  }
  if (scale) *scale = mp_max(mp_max(s_bm, s_cm), s_f * (T)M.lscale);
}

// ---- float32 rows, adaptive precision (round 5: the run-time-n rows too).  The float64 model of the same robot travels in
// MpCall<float>::cold_model (device memory for the kernels, the handle's own copy for the CPU launchers); null = float32 throughout.
#if defined(__HIP_DEVICE_COMPILE__)
typedef const __attribute__((address_space(4))) MpBigModel<double> MpBigConstD;
#else
typedef const MpBigModel<double> MpBigConstD;
#endif
// largest |tau| of the row against the scale of its intermediate terms (mp_id_row_is_hard of mp_core.h, run-time n)
MP_HD bool mp_dyn_row_is_hard(const float* tau, int n, float scale) {
  float rowmax = 0.0f;
  for (int i = 0; i < n; ++i) rowmax = mp_max(rowmax, mp_abs(tau[i]));
  return scale > MP_HARD_ROW_K * rowmax;
}
// the row again, everything in float64 from its float32 inputs (sin / cos and model constants too); tau unclipped
template <int CAP, bool HAS_FTIP>
MP_HD void mp_dyn_row_id_f64(MpBigConstD& Md, const MpCall<float>& C, const float* q, const float* qd, const float* qdd, float* tau) {
  const int n = Md.n;
  double a[CAP], b[CAP], c[CAP], t[CAP];
  for (int j = 0; j < n; ++j) { a[j] = (double)q[j]; b[j] = (double)qd[j]; c[j] = (double)qdd[j]; }
  const double a0[3] = {(double)C.a0[0], (double)C.a0[1], (double)C.a0[2]};
  const double tn[3] = {(double)C.F1n[0], (double)C.F1n[1], (double)C.F1n[2]}, tf[3] = {(double)C.F1f[0], (double)C.F1f[1], (double)C.F1f[2]};
  MpDynState<double, CAP> js;
  mp_dyn_joint_state<double>(Md, n, a, js);
  mp_dyn_rnea<double, HAS_FTIP>(Md, n, a0, tn, tf, js, b, c, t);
  for (int j = 0; j < n; ++j) tau[j] = (float)t[j];
}

// mp_mass_matrix_crba of mp_core.h with a run-time joint count; Mq is n x n row-major with row pitch `ld`.
template <typename T, typename MT, int CAP>
MP_HD void mp_dyn_mass_matrix(const MT& M, int n, const MpDynState<T, CAP>& js, T* Mq, int ld) {
  MpRbi<T> Ic;
  Ic.m = 0; Ic.hx = 0; Ic.hy = 0; Ic.hz = 0; Ic.xx = 0; Ic.xy = 0; Ic.xz = 0; Ic.yy = 0; Ic.yz = 0; Ic.zz = 0;
  MP_NOUNROLL
  for (int i = n - 1; i >= 0; --i) {
    const auto& J = M.j[i];
    Ic.m = Ic.m + J.m; Ic.hx = Ic.hx + J.hx; Ic.hy = Ic.hy + J.hy; Ic.hz = Ic.hz + J.hz;
    Ic.xx = Ic.xx + J.Ixx; Ic.xy = Ic.xy + J.Ixy; Ic.xz = Ic.xz + J.Ixz; Ic.yy = Ic.yy + J.Iyy; Ic.yz = Ic.yz + J.Iyz;
    Ic.zz = Ic.zz + J.Izz;
    const T r = J.rev, p = T(1) - J.rev;
    T nx = r * Ic.xz + p * Ic.hy, ny = r * Ic.yz - p * Ic.hx, nz = r * Ic.zz;
    T fx = -(r * Ic.hy), fy = r * Ic.hx, fz = p * Ic.m;
    Mq[i * ld + i] = r * nz + p * fz;
    MP_NOUNROLL
    for (int k = i; k > 0; --k) {
      const auto& Jk = M.j[k];
      mp_force_up_B(js.c[k], js.s[k], js.d[k], nx, ny, nz, fx, fy, fz);
      mp_force_up_A(Jk.ca, Jk.sa, Jk.a, nx, ny, nz, fx, fy, fz);
      const T rp = M.j[k - 1].rev;
      const T v = rp * nz + (T(1) - rp) * fz;
      Mq[(k - 1) * ld + i] = v;
      Mq[i * ld + (k - 1)] = v;
    }
    if (i > 0) {
      mp_rbi_up_B(js.c[i], js.s[i], js.d[i], Ic);
      mp_rbi_up_A(J.ca, J.sa, J.a, Ic);
    }
  }
}

// mp_spd_solve of mp_core.h with a run-time size: Cholesky in place (row pitch `ld`), b overwritten by the solution
template <typename T>
MP_HD void mp_dyn_spd_solve(int n, T* A, int ld, T* b) {
  MP_NOUNROLL
  for (int j = 0; j < n; ++j) {
    T d = A[j * ld + j];
    for (int k = 0; k < j; ++k) d -= A[j * ld + k] * A[j * ld + k];
    const T inv = mp_rsqrt(d);
    A[j * ld + j] = inv;
    for (int i = j + 1; i < n; ++i) {
      T v = A[i * ld + j];
      for (int k = 0; k < j; ++k) v -= A[i * ld + k] * A[j * ld + k];
      A[i * ld + j] = v * inv;
    }
  }
  MP_NOUNROLL
  for (int i = 0; i < n; ++i) {
    T v = b[i];
    for (int k = 0; k < i; ++k) v -= A[i * ld + k] * b[k];
    b[i] = v * A[i * ld + i];
  }
  MP_NOUNROLL
  for (int i = n - 1; i >= 0; --i) {
    T v = b[i];
    for (int k = i + 1; k < n; ++k) v -= A[k * ld + i] * b[k];
    b[i] = v * A[i * ld + i];
  }
}

// qdd = M(q)^-1 (tau - bias), bias = ID(q, qd, 0, g, F)   (reference dynamics/id_fd.py:71-83)
template <int CAP, typename T, bool HAS_FTIP, typename MT>
MP_HD void mp_dyn_forward_dynamics(const MT& M, int n, const T (&a0)[3], const T (&tipn)[3], const T (&tipf)[3], const T* q,
                                   const T* qd, const T* tau, T* qdd) {
  MpDynState<T, CAP> js;
  mp_dyn_joint_state<T>(M, n, q, js);
  T zero[CAP], bias[CAP], Mq[CAP * CAP];
  for (int k = 0; k < n; ++k) zero[k] = T(0);
  mp_dyn_rnea<T, HAS_FTIP>(M, n, a0, tipn, tipf, js, qd, zero, bias);
  mp_dyn_mass_matrix<T>(M, n, js, Mq, CAP);
  for (int k = 0; k < n; ++k) qdd[k] = tau[k] - bias[k];
  mp_dyn_spd_solve<T>(n, Mq, CAP, qdd);
}

// mp_fk_jac of mp_core.h with a run-time joint count: Tout 4x4 row-major, Jout 6 x n row-major (either may be null)
template <typename T, typename MT, int CAP>
MP_HD void mp_dyn_fk_jac(const MT& M, int n, const MpDynState<T, CAP>& js, T* Tout, T* Jout) {
  T x0 = M.base_R[0], x1 = M.base_R[3], x2 = M.base_R[6];
  T y0 = M.base_R[1], y1 = M.base_R[4], y2 = M.base_R[7];
  T z0 = M.base_R[2], z1 = M.base_R[5], z2 = M.base_R[8];
  T p0 = M.base_p[0], p1 = M.base_p[1], p2 = M.base_p[2];
  MP_NOUNROLL
  for (int i = 0; i < n; ++i) {
    const auto& J = M.j[i];
    if (i > 0) {
      p0 += J.a * x0; p1 += J.a * x1; p2 += J.a * x2;
      const T a0 = y0, a1 = y1, a2 = y2;
      y0 = J.ca * a0 + J.sa * z0; y1 = J.ca * a1 + J.sa * z1; y2 = J.ca * a2 + J.sa * z2;
      z0 = J.ca * z0 - J.sa * a0; z1 = J.ca * z1 - J.sa * a1; z2 = J.ca * z2 - J.sa * a2;
    }
    if (Jout) {
      const T cx = p1 * z2 - p2 * z1, cy = p2 * z0 - p0 * z2, cz = p0 * z1 - p1 * z0;
      const T r = J.rev, pr = T(1) - J.rev;
      Jout[0 * n + i] = r * z0; Jout[1 * n + i] = r * z1; Jout[2 * n + i] = r * z2;
      Jout[3 * n + i] = r * cx + pr * z0; Jout[4 * n + i] = r * cy + pr * z1; Jout[5 * n + i] = r * cz + pr * z2;
    }
    const T c = js.c[i], s = js.s[i], d = js.d[i];
    const T b0 = x0, b1 = x1, b2 = x2;
    x0 = c * b0 + s * y0; x1 = c * b1 + s * y1; x2 = c * b2 + s * y2;
    y0 = c * y0 - s * b0; y1 = c * y1 - s * b1; y2 = c * y2 - s * b2;
    p0 += d * z0; p1 += d * z1; p2 += d * z2;
  }
  if (Tout) {
    const auto& R = M.tool_R;
    const auto& t = M.tool_p;
    Tout[0] = x0 * R[0] + y0 * R[3] + z0 * R[6]; Tout[1] = x0 * R[1] + y0 * R[4] + z0 * R[7]; Tout[2] = x0 * R[2] + y0 * R[5] + z0 * R[8];
    Tout[4] = x1 * R[0] + y1 * R[3] + z1 * R[6]; Tout[5] = x1 * R[1] + y1 * R[4] + z1 * R[7]; Tout[6] = x1 * R[2] + y1 * R[5] + z1 * R[8];
    Tout[8] = x2 * R[0] + y2 * R[3] + z2 * R[6]; Tout[9] = x2 * R[1] + y2 * R[4] + z2 * R[7]; Tout[10] = x2 * R[2] + y2 * R[5] + z2 * R[8];
    Tout[3] = p0 + x0 * t[0] + y0 * t[1] + z0 * t[2];
    Tout[7] = p1 + x1 * t[0] + y1 * t[1] + z1 * t[2];
    Tout[11] = p2 + x2 * t[0] + y2 * t[1] + z2 * t[2];
    Tout[12] = T(0); Tout[13] = T(0); Tout[14] = T(0); Tout[15] = T(1);
  }
}

// ------------------------------------------------------------------------------------------------ whole rows
// One row `r` of each operation on plain row-major arrays: what a lane of the k_dyn_* kernels and one iteration of the
// CPU launchers' loops execute.  Non-finite inputs poison the row (mp_core.h, MpBad).
template <typename T>
MP_HD bool mp_dyn_bad(const T* v, int n, MpBad<T>& bad) {
  for (int j = 0; j < n; ++j) bad.add(v[j]);
  return bad.any();
}

template <int CAP, typename T, bool HAS_FTIP, typename MT>
MP_HD void mp_dyn_row_fk_jac_id(const MT& M, const MpCall<T>& C, const T* q, const T* qd, const T* qdd, T* Tout, T* Jout, T* tau,
                                long r) {
  const int n = M.n;
  T a[CAP];
  for (int j = 0; j < n; ++j) a[j] = q[r * n + j];
  MpDynState<T, CAP> js;
  mp_dyn_joint_state<T>(M, n, a, js);
  MpBad<T> bad;
  mp_dyn_bad(a, n, bad);
  if (Tout || Jout) {
    T TT[16], JJ[6 * CAP];
    mp_dyn_fk_jac<T>(M, n, js, Tout ? TT : nullptr, Jout ? JJ : nullptr);
    const bool p = bad.any();
    if (Tout) for (int k = 0; k < 16; ++k) { T v = TT[k]; mp_poison_if(p, v); Tout[r * 16 + k] = v; }
    if (Jout) for (int k = 0; k < 6 * n; ++k) { T v = JJ[k]; mp_poison_if(p, v); Jout[r * 6 * n + k] = v; }
  }
  if (tau) {
    T b[CAP], c[CAP], t[CAP];
    for (int j = 0; j < n; ++j) { b[j] = qd[r * n + j]; c[j] = qdd[r * n + j]; }
    mp_dyn_bad(b, n, bad);
    const bool p = mp_dyn_bad(c, n, bad);
    if constexpr (MpIsF32<T>::value) {
      // (as the unrolled kernels: rows whose torques are a small difference of large terms are evaluated again in float64 - in place:
      // these kernels keep their per-joint state in indexed arrays anyway and make no speed claim)
      T scale = 0;
      mp_dyn_rnea<T, HAS_FTIP>(M, n, C.a0, C.F1n, C.F1f, js, b, c, t, C.cold_model ? &scale : nullptr);
      if (C.cold_model && !p && mp_dyn_row_is_hard(t, n, scale)) mp_dyn_row_id_f64<CAP, HAS_FTIP>(*(MpBigConstD*)C.cold_model, C, a, b, c, t);
    } else {
      mp_dyn_rnea<T, HAS_FTIP>(M, n, C.a0, C.F1n, C.F1f, js, b, c, t);
    }
    for (int j = 0; j < n; ++j) {
      T v = mp_clip(t[j], M.taumin[j], M.taumax[j]);
      mp_poison_if(p, v);
      tau[r * n + j] = v;
    }
  }
}

template <int CAP, typename T, typename MT>
MP_HD void mp_dyn_row_mass_matrix(const MT& M, const T* q, T* Mout, long r) {
  const int n = M.n;
  T a[CAP], Mq[CAP * CAP];
  for (int j = 0; j < n; ++j) a[j] = q[r * n + j];
  MpDynState<T, CAP> js;
  mp_dyn_joint_state<T>(M, n, a, js);
  mp_dyn_mass_matrix<T>(M, n, js, Mq, CAP);
  MpBad<T> bad;
  const bool p = mp_dyn_bad(a, n, bad);
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < n; ++j) { T v = Mq[i * CAP + j]; mp_poison_if(p, v); Mout[(r * n + i) * n + j] = v; }
}

template <int CAP, typename T, bool HAS_FTIP, typename MT>
MP_HD void mp_dyn_row_forward_dynamics(const MT& M, const MpCall<T>& C, const T* q, const T* qd, const T* tau, T* qdd, long r) {
  const int n = M.n;
  T a[CAP], b[CAP], t[CAP], o[CAP];
  for (int j = 0; j < n; ++j) { a[j] = q[r * n + j]; b[j] = qd[r * n + j]; t[j] = tau[r * n + j]; }
  mp_dyn_forward_dynamics<CAP, T, HAS_FTIP>(M, n, C.a0, C.F1n, C.F1f, a, b, t, o);
  MpBad<T> bad;
  mp_dyn_bad(a, n, bad); mp_dyn_bad(b, n, bad);
  const bool p = mp_dyn_bad(t, n, bad);
  for (int j = 0; j < n; ++j) { T v = o[j]; mp_poison_if(p, v); qdd[r * n + j] = v; }
}

// forward_dynamics_trajectory for trajectory `b` of B (reference planning/trajectory_dynamics.py:580-708): the loop of
// mp_body_fd_traj / mp_body_fd_traj_tm with a run-time joint count.  Row (b, i) sits at b * Nt + i (batch-major arrays) or
// i * B + b (time-major arrays).
template <int CAP, typename T, bool HAS_FTIP, typename MT>
MP_HD void mp_dyn_rollout(const MT& M, const MpCall<T>& C, const T* theta0, const T* dtheta0, const T* taumat, const T* Ftipmat,
                          long b, long B, long Nt, T h, int intRes, float* pos, float* vel, float* acc, bool time_major) {
  const int n = M.n;
  T q[CAP], qd[CAP], tau[CAP], last[CAP];
  for (int j = 0; j < n; ++j) { q[j] = theta0[b * n + j]; qd[j] = dtheta0[b * n + j]; }
  MpBad<T> bad;
  mp_dyn_bad(q, n, bad); mp_dyn_bad(qd, n, bad);
  const unsigned nanbits = 0x7fc00000u;
  for (long i = 0; i < Nt; ++i) {
    const long row = time_major ? i * B + b : b * Nt + i;
    for (int j = 0; j < n; ++j) last[j] = T(0);
    if (i > 0) {
      T tn[3] = {T(0), T(0), T(0)}, tf[3] = {T(0), T(0), T(0)};
      for (int j = 0; j < n; ++j) tau[j] = taumat[row * n + j];
      mp_dyn_bad(tau, n, bad);
      if (HAS_FTIP) {
        T F[6];
        for (int k = 0; k < 6; ++k) F[k] = Ftipmat[row * 6 + k];
        mp_dyn_bad(F, 6, bad);
        mp_wrench_to_frame1(M, F, tn, tf);
      }
      for (int k = 0; k < intRes; ++k) {
        mp_dyn_forward_dynamics<CAP, T, HAS_FTIP>(M, n, C.a0, tn, tf, q, qd, tau, last);
        for (int j = 0; j < n; ++j) {
          qd[j] = qd[j] + last[j] * h;
          q[j] = mp_clip(q[j] + qd[j] * h, M.qmin[j], M.qmax[j]);
        }
      }
      mp_dyn_bad(qd, n, bad);
    }
    const bool poison = i > 0 && bad.any();
    for (int j = 0; j < n; ++j) {
      const float nanf_ = __builtin_bit_cast(float, nanbits);
      pos[row * n + j] = poison ? nanf_ : (float)q[j];
      vel[row * n + j] = poison ? nanf_ : (float)qd[j];
      acc[row * n + j] = poison ? nanf_ : (float)last[j];
    }
  }
}

// row (b, t) of the time-scaled trajectory (the arithmetic of traj_row, mp_bodies.h) and, optionally, its torques
template <int CAP, bool HAS_FTIP, typename MT>
MP_HD void mp_dyn_row_traj(const MT& M, const MpCall<float>& C, const float* start, const float* end, long b, long t, long Nt,
                           double Tf, int method, float* pos, float* vel, float* acc, float* tau) {
  const int n = M.n;
  const double tt = (double)t * (Tf / (double)(Nt - 1));
  double s, sd, sdd;
  mp_time_scaling(method, tt / Tf, Tf, s, sd, sdd);
  float p[CAP], v[CAP], a[CAP];
  for (int j = 0; j < n; ++j) {
    const float a0 = start[b * n + j];
    const double d = (double)(end[b * n + j] - a0);  // float32 difference first, as the reference types it
    p[j] = mp_clip((float)(s * d + (double)a0), M.qmin[j], M.qmax[j]);
    v[j] = (float)(sd * d);
    a[j] = (float)(sdd * d);
  }
  const long r = b * Nt + t;
  if (pos) for (int j = 0; j < n; ++j) { pos[r * n + j] = p[j]; vel[r * n + j] = v[j]; acc[r * n + j] = a[j]; }
  if (tau) {
    MpDynState<float, CAP> js;
    mp_dyn_joint_state<float>(M, n, p, js);
    float tq[CAP], scale = 0.0f;
    mp_dyn_rnea<float, HAS_FTIP>(M, n, C.a0, C.F1n, C.F1f, js, v, a, tq, C.cold_model ? &scale : nullptr);
    MpBad<float> bad;
    mp_dyn_bad(p, n, bad); mp_dyn_bad(v, n, bad);
    const bool poison = mp_dyn_bad(a, n, bad);
    if (C.cold_model && !poison && mp_dyn_row_is_hard(tq, n, scale)) mp_dyn_row_id_f64<CAP, HAS_FTIP>(*(MpBigConstD*)C.cold_model, C, p, v, a, tq);
    for (int j = 0; j < n; ++j) {
      float x = mp_clip(tq[j], M.taumin[j], M.taumax[j]);
      mp_poison_if(poison, x);
      tau[r * n + j] = x;
    }
  }
}

// mp_pd_regulation_run of mp_core.h with a run-time joint count
template <int CAP, typename T, typename MT>
MP_HD int mp_dyn_pd_regulation_run(const MT& M, const T (&a0)[3], const T* theta0, const T* des, T Kp, T Kd, T dt, int steps, T* err) {
  const int n = M.n;
  T th[CAP], om[CAP], tau[CAP], al[CAP];
  const T z3[3] = {T(0), T(0), T(0)};
  for (int j = 0; j < n; ++j) { th[j] = theta0[j]; om[j] = T(0); }
  int done = 0;
  for (int step = 0; step < steps; ++step) {
    for (int j = 0; j < n; ++j) tau[j] = Kp * (des[j] - th[j]) - Kd * om[j];
    mp_dyn_forward_dynamics<CAP, T, false>(M, n, a0, z3, z3, th, om, tau, al);
    T e2 = T(0);
    for (int j = 0; j < n; ++j) {
      om[j] += al[j] * dt;
      th[j] += om[j] * dt;
      e2 += (th[j] - des[j]) * (th[j] - des[j]);
    }
    const T e = sqrt(e2);
    err[step] = e;
    done = step + 1;
    if (step > 10 && e > T(1e10)) break;
  }
  return done;
}

// ------------------------------------------------------------------------------- inverse kinematics, 9..32 joints
// The kinematics policy of mp_ik.h for a run-time joint count: the damped-least-squares iteration itself (error, step,
// restart, adaptive damping, line search) is the one template of mp_ik.h; only the joint count and FK + Jacobian differ.
#include "mp_ik.h"
template <int CAP>
struct MpIkLooped {
  template <typename MT>
  MP_HD static int count(const MT& M) { return M.n; }
  template <bool WANT_J, typename MT>
  MP_HD static void fk(const MT& M, const double (&theta)[CAP], double (&Tc)[16], double (&J)[6 * CAP]) {
    MpDynState<double, CAP> js;
    mp_dyn_joint_state<double>(M, M.n, theta, js);
    mp_dyn_fk_jac<double>(M, M.n, js, Tc, WANT_J ? J : nullptr);  // J: 6 x n row-major, as the iteration indexes it
  }
};
