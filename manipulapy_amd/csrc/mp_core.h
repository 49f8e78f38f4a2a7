// Per-(trajectory, timestep) rigid-body math, one call = one row.  Header-only, templated on the
// arithmetic type (float / double) and the DOF (fully unrolled: every array below lives in VGPRs,
// every model constant is an SGPR operand).  Compiles for gfx950 under hipcc and, unchanged, for the
// host under g++ (tests/hostsim uses that to check the math on a GPU-less CI box).
//
// What it replaces in the reference (per row):
//   inverse_dynamics      ManipulaPy/dynamics/id_fd.py:16-48        tau = M qdd + c + g + Js^T Ftip
//     mass_matrix         ManipulaPy/dynamics/mass_matrix.py:62-96  (x(1+2n) through the finite difference)
//     velocity_quadratic  ManipulaPy/dynamics/forces.py:45-58  + dynamics/cache.py:39-52
//     gravity_forces      ManipulaPy/dynamics/forces.py:100-133
//   forward_kinematics    ManipulaPy/kinematics/fk.py:59-70
//   jacobian (space)      ManipulaPy/kinematics/jacobian.py:62-73
//   time scaling          ManipulaPy/planning/trajectory.py:45-73
// The reference's tau is, analytically, the recursive Newton-Euler result with the base accelerated by
// -g plus Js^T Ftip; its central-difference Coriolis term carries O(1e-9) noise that
// the analytic recursion does not.
#pragma once

#include <cmath>

#include "mp_model.h"

#if defined(__HIPCC__)
#define MP_HD __host__ __device__ __forceinline__
#else
#define MP_HD inline
#endif

// ------------------------------------------------------------------------------------------- trig
// float: Cody-Waite reduction by pi/2 (two FMAs, exact enough for |x| < ~1e4) + the classic minimax
// polynomials on [-pi/4, pi/4]; ~1 ulp, branch-free, ~24 VALU instructions for BOTH results.
MP_HD void mp_sincos(float x, float& s, float& c) {
  const float k = rintf(x * 0.636619772367581343f);
  float r = fmaf(-k, 1.57079637050628662109375f, x);
  r = fmaf(-k, -4.37113900018624283e-8f, r);
  const float r2 = r * r;
  float ps = fmaf(r2, -1.9515295891e-4f, 8.3321608736e-3f);
  ps = fmaf(r2, ps, -1.6666654611e-1f);
  ps = fmaf(r * r2, ps, r);
  float pc = fmaf(r2, 2.443315711809948e-5f, -1.388731625493765e-3f);
  pc = fmaf(r2, pc, 4.166664568298827e-2f);
  pc = fmaf(r2 * r2, pc, fmaf(r2, -0.5f, 1.0f));
  const int q = (int)k;
  const float a = (q & 1) ? pc : ps;
  const float b = (q & 1) ? ps : pc;
  s = (q & 2) ? -a : a;
  c = ((q + 1) & 2) ? -b : b;
}
MP_HD void mp_sincos(double x, double& s, double& c) {
#if defined(__HIP_DEVICE_COMPILE__)
  sincos(x, &s, &c);
#else
  s = std::sin(x);
  c = std::cos(x);
#endif
}

template <typename T> MP_HD T mp_min(T a, T b) { return a < b ? a : b; }
template <typename T> MP_HD T mp_max(T a, T b) { return a > b ? a : b; }
template <typename T> MP_HD T mp_clip(T v, T lo, T hi) { return mp_min(mp_max(v, lo), hi); }  // np.clip order

// ------------------------------------------------------------------------------ axis-aligned steps
// Motion vector (w, v), parent -> child coordinates, child pose in parent = (E, r):
//     w' = E^T w,  v' = E^T (v + w x r).
// step A: E = Rx(alpha), r = (a, 0, 0);   step B: E = Rz(theta), r = (0, 0, d).
template <typename T>
MP_HD void mp_motion_A(T ca, T sa, T a, T& wx, T& wy, T& wz, T& vx, T& vy, T& vz) {
  const T ty = vy + a * wz, tz = vz - a * wy;
  vy = ca * ty + sa * tz;
  vz = ca * tz - sa * ty;
  const T uy = wy;
  wy = ca * uy + sa * wz;
  wz = ca * wz - sa * uy;
  (void)wx; (void)vx;
}
template <typename T>
MP_HD void mp_motion_B(T c, T s, T d, T& wx, T& wy, T& wz, T& vx, T& vy, T& vz) {
  const T tx = vx + d * wy, ty = vy - d * wx;
  vx = c * tx + s * ty;
  vy = c * ty - s * tx;
  const T ux = wx;
  wx = c * ux + s * wy;
  wy = c * wy - s * ux;
  (void)wz; (void)vz;
}
// Force vector (n, f), parent -> child coordinates:  f' = E^T f,  n' = E^T (n - r x f).
template <typename T>
MP_HD void mp_force_down_A(T ca, T sa, T a, T& nx, T& ny, T& nz, T& fx, T& fy, T& fz) {
  const T ty = ny + a * fz, tz = nz - a * fy;
  ny = ca * ty + sa * tz;
  nz = ca * tz - sa * ty;
  const T uy = fy;
  fy = ca * uy + sa * fz;
  fz = ca * fz - sa * uy;
  (void)nx; (void)fx;
}
template <typename T>
MP_HD void mp_force_down_B(T c, T s, T d, T& nx, T& ny, T& nz, T& fx, T& fy, T& fz) {
  const T tx = nx + d * fy, ty = ny - d * fx;
  nx = c * tx + s * ty;
  ny = c * ty - s * tx;
  const T ux = fx;
  fx = c * ux + s * fy;
  fy = c * fy - s * ux;
  (void)nz; (void)fz;
}
// Force vector (n, f), child -> parent coordinates:  f' = E f,  n' = E n + r x f'.
template <typename T>
MP_HD void mp_force_up_B(T c, T s, T d, T& nx, T& ny, T& nz, T& fx, T& fy, T& fz) {
  const T gx = c * fx - s * fy, gy = s * fx + c * fy;
  const T mx = c * nx - s * ny, my = s * nx + c * ny;
  fx = gx; fy = gy;
  nx = mx - d * gy;
  ny = my + d * gx;
  (void)nz; (void)fz;
}
template <typename T>
MP_HD void mp_force_up_A(T ca, T sa, T a, T& nx, T& ny, T& nz, T& fx, T& fy, T& fz) {
  const T gy = ca * fy - sa * fz, gz = sa * fy + ca * fz;
  const T my = ca * ny - sa * nz, mz = sa * ny + ca * nz;
  fy = gy; fz = gz;
  ny = my - a * gz;
  nz = mz + a * gy;
  (void)nx; (void)fx;
}

// ------------------------------------------------------------------------------------------- RNEA
// Per-row joint state shared between the passes: sin/cos of the joint angle and the z shift.
template <typename T, int N>
struct MpJointState {
  T s[N], c[N], d[N];
};

template <typename T, int N>
MP_HD void mp_joint_state(const MpModel<T>& M, const T (&q)[N], MpJointState<T, N>& js) {
#pragma unroll
  for (int i = 0; i < N; ++i) {
    const MpJoint<T>& J = M.j[i];
    const T qr = J.rev * q[i];
    mp_sincos(J.off + qr, js.s[i], js.c[i]);
    js.d[i] = J.d + (q[i] - qr);
  }
}

// Recursive Newton-Euler in the compiled link frames.  tau is NOT clipped here.
template <typename T, int N, bool HAS_FTIP>
MP_HD void mp_rnea(const MpModel<T>& M, const MpCall<T>& C, const MpJointState<T, N>& js, const T (&qd)[N],
                   const T (&qdd)[N], T (&tau)[N]) {
  T fnx[N], fny[N], fnz[N], ffx[N], ffy[N], ffz[N];
  T wx = 0, wy = 0, wz = 0, vx = 0, vy = 0, vz = 0;
  T dwx = 0, dwy = 0, dwz = 0, dvx = C.a0[0], dvy = C.a0[1], dvz = C.a0[2];
  T tnx = 0, tny = 0, tnz = 0, tfx = 0, tfy = 0, tfz = 0;
  if (HAS_FTIP) { tnx = C.F1n[0]; tny = C.F1n[1]; tnz = C.F1n[2]; tfx = C.F1f[0]; tfy = C.F1f[1]; tfz = C.F1f[2]; }

  // forward pass: twists, accelerations, body wrenches
#pragma unroll
  for (int i = 0; i < N; ++i) {
    const MpJoint<T>& J = M.j[i];
    if (i > 0) {
      mp_motion_A(J.ca, J.sa, J.a, wx, wy, wz, vx, vy, vz);
      mp_motion_A(J.ca, J.sa, J.a, dwx, dwy, dwz, dvx, dvy, dvz);
      if (HAS_FTIP) mp_force_down_A(J.ca, J.sa, J.a, tnx, tny, tnz, tfx, tfy, tfz);
    }
    const T c = js.c[i], s = js.s[i], d = js.d[i];
    mp_motion_B(c, s, d, wx, wy, wz, vx, vy, vz);
    mp_motion_B(c, s, d, dwx, dwy, dwz, dvx, dvy, dvz);
    if (HAS_FTIP) mp_force_down_B(c, s, d, tnx, tny, tnz, tfx, tfy, tfz);

    // joint motion: S = [z;0] (revolute) or [0;z] (prismatic)
    const T qdr = J.rev * qd[i], qdp = qd[i] - qdr;
    const T ar = J.rev * qdd[i], ap = qdd[i] - ar;
    wz += qdr;
    vz += qdp;
    // dV += S qdd + V x S qd
    dwx += qdr * wy;
    dwy -= qdr * wx;
    dwz += ar;
    dvx += qdr * vy + qdp * wy;
    dvy -= qdr * vx + qdp * wx;
    dvz += ap;

    // momentum P = G V = [Io w + h x v ; m v - h x w]
    const T pnx = J.Ixx * wx + J.Ixy * wy + J.Ixz * wz + (J.hy * vz - J.hz * vy);
    const T pny = J.Ixy * wx + J.Iyy * wy + J.Iyz * wz + (J.hz * vx - J.hx * vz);
    const T pnz = J.Ixz * wx + J.Iyz * wy + J.Izz * wz + (J.hx * vy - J.hy * vx);
    const T pfx = J.m * vx - (J.hy * wz - J.hz * wy);
    const T pfy = J.m * vy - (J.hz * wx - J.hx * wz);
    const T pfz = J.m * vz - (J.hx * wy - J.hy * wx);
    // F = G dV + [w x Pn + v x Pf ; w x Pf]
    fnx[i] = J.Ixx * dwx + J.Ixy * dwy + J.Ixz * dwz + (J.hy * dvz - J.hz * dvy) + (wy * pnz - wz * pny) + (vy * pfz - vz * pfy);
    fny[i] = J.Ixy * dwx + J.Iyy * dwy + J.Iyz * dwz + (J.hz * dvx - J.hx * dvz) + (wz * pnx - wx * pnz) + (vz * pfx - vx * pfz);
    fnz[i] = J.Ixz * dwx + J.Iyz * dwy + J.Izz * dwz + (J.hx * dvy - J.hy * dvx) + (wx * pny - wy * pnx) + (vx * pfy - vy * pfx);
    ffx[i] = J.m * dvx - (J.hy * dwz - J.hz * dwy) + (wy * pfz - wz * pfy);
    ffy[i] = J.m * dvy - (J.hz * dwx - J.hx * dwz) + (wz * pfx - wx * pfz);
    ffz[i] = J.m * dvz - (J.hx * dwy - J.hy * dwx) + (wx * pfy - wy * pfx);
  }
  if (HAS_FTIP) {  // Js^T Ftip: the space-frame wrench, now expressed in link frame N, rides the backward pass
    fnx[N - 1] += tnx; fny[N - 1] += tny; fnz[N - 1] += tnz;
    ffx[N - 1] += tfx; ffy[N - 1] += tfy; ffz[N - 1] += tfz;
  }
  // backward pass
#pragma unroll
  for (int i = N - 1; i >= 0; --i) {
    const MpJoint<T>& J = M.j[i];
    tau[i] = J.rev * fnz[i] + (T(1) - J.rev) * ffz[i];
    if (i > 0) {
      T nx = fnx[i], ny = fny[i], nz = fnz[i], fx = ffx[i], fy = ffy[i], fz = ffz[i];
      mp_force_up_B(js.c[i], js.s[i], js.d[i], nx, ny, nz, fx, fy, fz);
      mp_force_up_A(J.ca, J.sa, J.a, nx, ny, nz, fx, fy, fz);
      fnx[i - 1] += nx; fny[i - 1] += ny; fnz[i - 1] += nz;
      ffx[i - 1] += fx; ffy[i - 1] += fy; ffz[i - 1] += fz;
    }
  }
}

// ------------------------------------------------------------------------- FK + space Jacobian
// T (4x4 row-major) = prod T_{i-1,i}(q_i) . tool ;  J (6 x N row-major), column i = [z_i ; o_i x z_i]
// (revolute) or [0 ; z_i] (prismatic), z_i / o_i = axis / origin of link frame i in the space frame —
// identical to Ad(prod_{j<i} exp) S_i because link frame i's z axis IS joint axis i.
template <typename T, int N, bool WANT_J>
MP_HD void mp_fk_jac(const MpModel<T>& M, const MpJointState<T, N>& js, T* Tout, T* Jout) {
  // columns of R and origin p of the running frame
  T x0 = M.base_R[0], x1 = M.base_R[3], x2 = M.base_R[6];
  T y0 = M.base_R[1], y1 = M.base_R[4], y2 = M.base_R[7];
  T z0 = M.base_R[2], z1 = M.base_R[5], z2 = M.base_R[8];
  T p0 = M.base_p[0], p1 = M.base_p[1], p2 = M.base_p[2];
#pragma unroll
  for (int i = 0; i < N; ++i) {
    const MpJoint<T>& J = M.j[i];
    if (i > 0) {  // . Rx(alpha) Tx(a)
      p0 += J.a * x0; p1 += J.a * x1; p2 += J.a * x2;
      const T a0 = y0, a1 = y1, a2 = y2;
      y0 = J.ca * a0 + J.sa * z0; y1 = J.ca * a1 + J.sa * z1; y2 = J.ca * a2 + J.sa * z2;
      z0 = J.ca * z0 - J.sa * a0; z1 = J.ca * z1 - J.sa * a1; z2 = J.ca * z2 - J.sa * a2;
    }
    if (WANT_J) {  // the joint axis is fixed in the PARENT link: read it before the joint moves the frame
      // revolute: [z ; p x z] with p any point of the axis (the frame origin before Tz is on it)
      const T cx = p1 * z2 - p2 * z1, cy = p2 * z0 - p0 * z2, cz = p0 * z1 - p1 * z0;
      const T r = J.rev, pr = T(1) - J.rev;
      Jout[0 * N + i] = r * z0; Jout[1 * N + i] = r * z1; Jout[2 * N + i] = r * z2;
      Jout[3 * N + i] = r * cx + pr * z0; Jout[4 * N + i] = r * cy + pr * z1; Jout[5 * N + i] = r * cz + pr * z2;
    }
    // . Rz(theta) Tz(d)
    const T c = js.c[i], s = js.s[i], d = js.d[i];
    const T b0 = x0, b1 = x1, b2 = x2;
    x0 = c * b0 + s * y0; x1 = c * b1 + s * y1; x2 = c * b2 + s * y2;
    y0 = c * y0 - s * b0; y1 = c * y1 - s * b1; y2 = c * y2 - s * b2;
    p0 += d * z0; p1 += d * z1; p2 += d * z2;
  }
  const T* R = M.tool_R;
  const T* t = M.tool_p;
  Tout[0] = x0 * R[0] + y0 * R[3] + z0 * R[6]; Tout[1] = x0 * R[1] + y0 * R[4] + z0 * R[7]; Tout[2] = x0 * R[2] + y0 * R[5] + z0 * R[8];
  Tout[4] = x1 * R[0] + y1 * R[3] + z1 * R[6]; Tout[5] = x1 * R[1] + y1 * R[4] + z1 * R[7]; Tout[6] = x1 * R[2] + y1 * R[5] + z1 * R[8];
  Tout[8] = x2 * R[0] + y2 * R[3] + z2 * R[6]; Tout[9] = x2 * R[1] + y2 * R[4] + z2 * R[7]; Tout[10] = x2 * R[2] + y2 * R[5] + z2 * R[8];
  Tout[3] = p0 + x0 * t[0] + y0 * t[1] + z0 * t[2];
  Tout[7] = p1 + x1 * t[0] + y1 * t[1] + z1 * t[2];
  Tout[11] = p2 + x2 * t[0] + y2 * t[1] + z2 * t[2];
  Tout[12] = 0; Tout[13] = 0; Tout[14] = 0; Tout[15] = 1;
}

// ------------------------------------------------------------------------------------ time scaling
// Reference planning/trajectory.py:45-73 (numba semantics): float32 endpoints and difference, float64
// scalar polynomial, float32 store; cubic (3) / quintic (5), anything else -> zeros.
MP_HD void mp_time_scaling(int method, double tau, double Tf, double& s, double& sd, double& sdd) {
  if (method == 3) {
    s = 3.0 * tau * tau - 2.0 * tau * tau * tau;
    sd = 6.0 * tau * (1.0 - tau) / Tf;
    sdd = 6.0 / (Tf * Tf) * (1.0 - 2.0 * tau);
  } else if (method == 5) {
    const double t2 = tau * tau, t3 = t2 * tau, t4 = t2 * t2, t5 = t4 * tau;
    s = 10.0 * t3 - 15.0 * t4 + 6.0 * t5;
    sd = (30.0 * t2 - 60.0 * t3 + 30.0 * t4) / Tf;
    sdd = (60.0 * tau - 180.0 * t2 + 120.0 * t3) / (Tf * Tf);
  } else {
    s = sd = sdd = 0.0;
  }
}
