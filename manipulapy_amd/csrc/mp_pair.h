// Pair-native float32 forward dynamics: the per-row recursions of mp_core.h rewritten so that ONE trajectory fills both
// halves of the packed float32 instructions (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32).
//
// Why.  The roll-out (config c5) is VALU-issue-bound and runs two waves per SIMD at most (131 072 trajectories per GPU
// = 2048 waves for 1024 SIMDs): the only lever is the instruction count per step.  Packing two TRAJECTORIES per lane
// (mp_body_fd_traj_pk) halves the wave count instead, which leaves one wave per SIMD and 57 % VALU utilisation.  Here the
// two halves of a packed instruction carry the angular and the linear part of the same spatial vector:
//     motion  V = (w; v)  ->  Vx = (wx, vx), Vy = (wy, vy), Vz = (wz, vz)        [.x = low half = angular, .y = linear]
//     force   F = (n; f)  ->  Fx = (nx, fx), Fy = (ny, fy), Fz = (nz, fz)        [.x = moment, .y = force]
// Both halves of such a pair undergo the same planar rotation in every axis-aligned step of the compiled link frames, the
// shift couples the halves through an operand swizzle (the hardware's op_sel: `.xx`, `.yy`, `.yx` cost nothing), and the
// 6x6 inertia product becomes five packed FMAs per component with constant pairs such as (Ixx, m).  Same wave count,
// same registers, roughly half the arithmetic instructions of the scalar recursion.
// float32 only (there is no packed float64 arithmetic worth having); the generic float64 paths keep mp_core.h.
#pragma once

#include "mp_core.h"

#if MP_HAS_PACKED

typedef mp_f2 P2;
MP_HD P2 mp_p2(float lo, float hi) { return (P2){lo, hi}; }
MP_HD P2 mp_pfma(P2 a, P2 b, P2 c) { return __builtin_elementwise_fma(a, b, c); }
// A pair of MODEL CONSTANTS as a packed operand.  Packed instructions take no literals, so every such pair is an SGPR
// pair built by two s_mov_b32 (scalar unit, beside the vector work).  The translation units that instantiate this header
// are compiled with -mllvm -disable-machine-licm: left on, that pass hoists the ~200 materialisations of one unrolled
// step out of the time loop, the ~100 SGPRs overflow and - a 64-bit pair assembled from two moves is not rematerialisable -
// are spilled through v_writelane / v_readlane inside the loop (measured: 300 extra VALU instructions per step, which
// cancels the gain of the packed arithmetic).
MP_HD P2 mp_kp2(float lo, float hi) { return (P2){lo, hi}; }

struct MpV3P {  // a spatial vector as three (angular | moment, linear | force) pairs
  P2 x, y, z;
};

// ---- axis-aligned steps on pair vectors (cf. mp_motion_A/B, mp_force_down_A/B, mp_force_up_A/B in mp_core.h)
// parent -> child across Rz(theta) Tz(d):  linear part first picks up w x r (r = (0, 0, d)), then both halves rotate
MP_HD void mp_p_motion_B(P2 c, P2 s, float d, MpV3P& V) {
  const P2 D = mp_p2(0.0f, d);
  const P2 tx = mp_pfma(D, V.y.xx, V.x), ty = mp_pfma(-D, V.x.xx, V.y);
  V.x = mp_pfma(c, tx, s * ty);
  V.y = mp_pfma(c, ty, -(s * tx));
}
// parent -> child across Rx(alpha) Tx(a)
MP_HD void mp_p_motion_A(float ca, float sa, float a, MpV3P& V) {
  const P2 D = mp_kp2(0.0f, a), C = mp_kp2(ca, ca), S = mp_kp2(sa, sa);
  const P2 ty = mp_pfma(D, V.z.xx, V.y), tz = mp_pfma(-D, V.y.xx, V.z);
  V.y = mp_pfma(C, ty, S * tz);
  V.z = mp_pfma(C, tz, -(S * ty));
}
// force, parent -> child: the MOMENT half picks up -r x f
MP_HD void mp_p_force_down_B(P2 c, P2 s, float d, MpV3P& F) {
  const P2 D = mp_p2(d, 0.0f);
  const P2 tx = mp_pfma(D, F.y.yy, F.x), ty = mp_pfma(-D, F.x.yy, F.y);
  F.x = mp_pfma(c, tx, s * ty);
  F.y = mp_pfma(c, ty, -(s * tx));
}
MP_HD void mp_p_force_down_A(float ca, float sa, float a, MpV3P& F) {
  const P2 D = mp_kp2(a, 0.0f), C = mp_kp2(ca, ca), S = mp_kp2(sa, sa);
  const P2 ty = mp_pfma(D, F.z.yy, F.y), tz = mp_pfma(-D, F.y.yy, F.z);
  F.y = mp_pfma(C, ty, S * tz);
  F.z = mp_pfma(C, tz, -(S * ty));
}
// force, child -> parent: rotate both halves, then the moment picks up r x f'
MP_HD void mp_p_force_up_B(P2 c, P2 s, float d, MpV3P& F) {
  const P2 gx = mp_pfma(c, F.x, -(s * F.y)), gy = mp_pfma(s, F.x, c * F.y);
  const P2 D = mp_p2(d, 0.0f);
  F.x = mp_pfma(-D, gy.yy, gx);
  F.y = mp_pfma(D, gx.yy, gy);
}
MP_HD void mp_p_force_up_A(float ca, float sa, float a, MpV3P& F) {
  const P2 C = mp_kp2(ca, ca), S = mp_kp2(sa, sa), D = mp_kp2(a, 0.0f);
  const P2 gy = mp_pfma(C, F.y, -(S * F.z)), gz = mp_pfma(S, F.y, C * F.z);
  F.y = mp_pfma(-D, gz.yy, gy);
  F.z = mp_pfma(D, gy.yy, gz);
}

// (Io w + h x v ; m v - h x w) for the pair vector V = (w | v): five packed FMAs per component
template <typename JT>
MP_HD void mp_p_inertia_mul(const JT& J, const MpV3P& V, MpV3P& P) {
  P.x = mp_kp2(J.Ixx, J.m) * V.x;
  P.x = mp_pfma(mp_kp2(J.Ixy, J.hz), V.y.xx, P.x);
  P.x = mp_pfma(mp_kp2(J.Ixz, -J.hy), V.z.xx, P.x);
  P.x = mp_pfma(mp_kp2(-J.hz, 0.0f), V.y.yy, P.x);
  P.x = mp_pfma(mp_kp2(J.hy, 0.0f), V.z.yy, P.x);
  P.y = mp_kp2(J.Iyy, J.m) * V.y;
  P.y = mp_pfma(mp_kp2(J.Ixy, -J.hz), V.x.xx, P.y);
  P.y = mp_pfma(mp_kp2(J.Iyz, J.hx), V.z.xx, P.y);
  P.y = mp_pfma(mp_kp2(J.hz, 0.0f), V.x.yy, P.y);
  P.y = mp_pfma(mp_kp2(-J.hx, 0.0f), V.z.yy, P.y);
  P.z = mp_kp2(J.Izz, J.m) * V.z;
  P.z = mp_pfma(mp_kp2(J.Ixz, J.hy), V.x.xx, P.z);
  P.z = mp_pfma(mp_kp2(J.Iyz, -J.hx), V.y.xx, P.z);
  P.z = mp_pfma(mp_kp2(-J.hy, 0.0f), V.x.yy, P.z);
  P.z = mp_pfma(mp_kp2(J.hx, 0.0f), V.y.yy, P.z);
}

// sin / cos of the joint angles two joints at a time; joint i reads its values with the free half-broadcast:
// c_i = js.c[i / 2].xx or .yy
template <int N>
struct MpJointStateP {
  static constexpr int H = (N + 1) / 2;
  P2 s[H], c[H];
  float d[N];
};
template <int N> MP_HD P2 mp_p_cos(const MpJointStateP<N>& js, int i) { return (i & 1) ? js.c[i / 2].yy : js.c[i / 2].xx; }
template <int N> MP_HD P2 mp_p_sin(const MpJointStateP<N>& js, int i) { return (i & 1) ? js.s[i / 2].yy : js.s[i / 2].xx; }

template <int N, typename MT>
MP_HD void mp_p_joint_state(const MT& M, const float (&q)[N], MpJointStateP<N>& js) {
#pragma unroll
  for (int k = 0; k < MpJointStateP<N>::H; ++k) {
    const int i0 = 2 * k, i1 = (2 * k + 1 < N) ? 2 * k + 1 : 2 * k;
    const P2 qr = mp_p2(M.j[i0].rev * q[i0], M.j[i1].rev * q[i1]);
    mp_sincos(mp_p2(M.j[i0].off, M.j[i1].off) + qr, js.s[k], js.c[k]);
  }
#pragma unroll
  for (int i = 0; i < N; ++i) js.d[i] = M.j[i].d + (q[i] - M.j[i].rev * q[i]);
}

// Bias forces ID(q, qd, 0, g, Ftip): recursive Newton-Euler with zero joint accelerations (cf. mp_rnea)
template <int N, bool HAS_FTIP, typename MT>
MP_HD void mp_p_bias(const MT& M, const float (&a0)[3], const float (&tipn)[3], const float (&tipf)[3], const MpJointStateP<N>& js,
                     const float (&qd)[N], float (&tau)[N]) {
  MpV3P F[N];
  MpV3P V = {mp_p2(0.f, 0.f), mp_p2(0.f, 0.f), mp_p2(0.f, 0.f)};
  MpV3P A = {mp_p2(0.f, a0[0]), mp_p2(0.f, a0[1]), mp_p2(0.f, a0[2])};
  MpV3P W = {mp_p2(0.f, 0.f), mp_p2(0.f, 0.f), mp_p2(0.f, 0.f)};
  if (HAS_FTIP) { W.x = mp_p2(tipn[0], tipf[0]); W.y = mp_p2(tipn[1], tipf[1]); W.z = mp_p2(tipn[2], tipf[2]); }
#pragma unroll
  for (int i = 0; i < N; ++i) {
    const auto& J = M.j[i];
    if (i > 0) {
      mp_p_motion_A(J.ca, J.sa, J.a, V);
      mp_p_motion_A(J.ca, J.sa, J.a, A);
      if (HAS_FTIP) mp_p_force_down_A(J.ca, J.sa, J.a, W);
    }
    const P2 c = mp_p_cos<N>(js, i), s = mp_p_sin<N>(js, i);
    const float d = js.d[i];
    mp_p_motion_B(c, s, d, V);
    mp_p_motion_B(c, s, d, A);
    if (HAS_FTIP) mp_p_force_down_B(c, s, d, W);
    // joint motion: S = [z; 0] (revolute) or [0; z] (prismatic);  dV += V x S qd  (qdd = 0)
    const float qdr = J.rev * qd[i], qdp = qd[i] - qdr;
    V.z = V.z + mp_p2(qdr, qdp);
    const P2 R = mp_p2(qdr, qdr), Pq = mp_p2(0.0f, qdp);
    A.x = mp_pfma(R, V.y, mp_pfma(Pq, V.y.xx, A.x));
    A.y = mp_pfma(-R, V.x, mp_pfma(-Pq, V.x.xx, A.y));
    // body wrench F = G dV + [w x Pn + v x Pf ; w x Pf],  P = G V
    MpV3P P, Q;
    mp_p_inertia_mul(J, V, P);
    mp_p_inertia_mul(J, A, Q);
    Q.x = mp_pfma(V.y.xx, P.z, mp_pfma(-V.z.xx, P.y, Q.x));
    Q.y = mp_pfma(V.z.xx, P.x, mp_pfma(-V.x.xx, P.z, Q.y));
    Q.z = mp_pfma(V.x.xx, P.y, mp_pfma(-V.y.xx, P.x, Q.z));
    Q.x.x = fmaf(V.y.y, P.z.y, fmaf(-V.z.y, P.y.y, Q.x.x));  // + v x Pf on the moment half only
    Q.y.x = fmaf(V.z.y, P.x.y, fmaf(-V.x.y, P.z.y, Q.y.x));
    Q.z.x = fmaf(V.x.y, P.y.y, fmaf(-V.y.y, P.x.y, Q.z.x));
    F[i] = Q;
  }
  if (HAS_FTIP) { F[N - 1].x = F[N - 1].x + W.x; F[N - 1].y = F[N - 1].y + W.y; F[N - 1].z = F[N - 1].z + W.z; }
#pragma unroll
  for (int i = N - 1; i >= 0; --i) {
    const auto& J = M.j[i];
    tau[i] = J.rev * F[i].z.x + (1.0f - J.rev) * F[i].z.y;
    if (i > 0) {
      MpV3P U = F[i];
      mp_p_force_up_B(mp_p_cos<N>(js, i), mp_p_sin<N>(js, i), js.d[i], U);
      mp_p_force_up_A(J.ca, J.sa, J.a, U);
      F[i - 1].x = F[i - 1].x + U.x; F[i - 1].y = F[i - 1].y + U.y; F[i - 1].z = F[i - 1].z + U.z;
    }
  }
}

// ---------------------------------------------------------------------------------------- mass matrix
// Composite-rigid-body algorithm (cf. mp_mass_matrix_crba) on a composite inertia stored so that the force of a unit
// joint acceleration, Ic S_i = (Ixz, Iyz, Izz ; -hy, hx, 0) for a revolute joint, IS three of its pairs:
//     A1 = (Ixz, -hy),  A2 = (Iyz, hx),  A3 = (Izz, mass)   plus the scalars hz, Ixx, Ixy, Iyy.
// (-hy, hx) is (hx, hy) turned by 90 degrees about z, which commutes with the joint rotation, so A1 / A2 rotate as a
// pair exactly like (Ixz, Iyz); the xy block turns by the double angle in the (half difference, xy) representation.
struct MpRbiP {
  P2 A1, A2, A3;
  float hz, xx, xy, yy;
};

// child -> parent across Rz(theta) Tz(d); c / s arrive broadcast, c2 / s2 are cos / sin of twice the angle
MP_HD void mp_p_rbi_up_B(P2 c, P2 s, float c2, float s2, float d, MpRbiP& I) {
  const P2 a1 = mp_pfma(c, I.A1, -(s * I.A2)), a2 = mp_pfma(s, I.A1, c * I.A2);
  const float mean = 0.5f * (I.xx + I.yy), u = 0.5f * (I.xx - I.yy);
  const float ur = c2 * u - s2 * I.xy, xyr = s2 * u + c2 * I.xy;
  const float m = I.A3.y, hx = a2.y, nhy = a1.y;                 // rotated first moments: hx, -hy
  const float t = d * (I.hz + I.hz) + m * (d * d);                // 2 (Rh).r + m |r|^2
  I.xx = mean + ur + t; I.yy = mean - ur + t; I.xy = xyr;
  I.A1 = mp_p2(a1.x - d * hx, nhy);                                // Ixz -= d hx
  I.A2 = mp_p2(a2.x + d * nhy, hx);                                // Iyz -= d hy
  I.hz = I.hz + m * d;
}
// child -> parent across Rx(alpha) Tx(a): constants, so the products fold per robot (alpha is a multiple of 90 degrees
// for most arms); plain component arithmetic on (h, I)
MP_HD void mp_p_rbi_up_A(float ca, float sa, float a, MpRbiP& I) {
  const float hx = I.A2.y, hy0 = -I.A1.y, hz0 = I.hz, m = I.A3.y;
  const float xz0 = I.A1.x, yz0 = I.A2.x, zz0 = I.A3.x;
  const float hy = ca * hy0 - sa * hz0, hz = sa * hy0 + ca * hz0;
  const float cc = ca * ca, ss = sa * sa, sc = sa * ca;
  const float yy = cc * I.yy - (sc + sc) * yz0 + ss * zz0;
  const float zz = ss * I.yy + (sc + sc) * yz0 + cc * zz0;
  const float yz = sc * (I.yy - zz0) + (cc - ss) * yz0;
  const float xy = ca * I.xy - sa * xz0, xz = sa * I.xy + ca * xz0;
  const float t = a * (hx + hx) + m * (a * a);
  I.yy = yy + t; I.xy = xy - a * hy;
  I.A1 = mp_p2(xz - a * hz, -hy);
  I.A2 = mp_p2(yz, hx + m * a);
  I.A3 = mp_p2(zz + t, m);
  I.hz = hz;
}

// lower triangle of M(q) (row i, column j <= i); the upper triangle is NOT written (the Cholesky solve reads L only)
template <int N, typename MT>
MP_HD void mp_p_mass_matrix(const MT& M, const MpJointStateP<N>& js, const float (&c2)[N], const float (&s2)[N], float (&Mq)[N][N]) {
  MpRbiP Ic;
  Ic.A1 = mp_p2(0.f, 0.f); Ic.A2 = mp_p2(0.f, 0.f); Ic.A3 = mp_p2(0.f, 0.f);
  Ic.hz = 0.f; Ic.xx = 0.f; Ic.xy = 0.f; Ic.yy = 0.f;
#pragma unroll
  for (int i = N - 1; i >= 0; --i) {
    const auto& J = M.j[i];
    Ic.A1 = Ic.A1 + mp_kp2(J.Ixz, -J.hy); Ic.A2 = Ic.A2 + mp_kp2(J.Iyz, J.hx); Ic.A3 = Ic.A3 + mp_kp2(J.Izz, J.m);
    Ic.hz += J.hz; Ic.xx += J.Ixx; Ic.xy += J.Ixy; Ic.yy += J.Iyy;
    // F = Ic S_i:  revolute S = [z; 0] -> (Ixz, Iyz, Izz ; -hy, hx, 0);  prismatic S = [0; z] -> (hy, -hx, 0 ; 0, 0, m)
    const float r = J.rev, p = 1.0f - J.rev;
    MpV3P F;
    F.x = mp_p2(r, r) * Ic.A1 + mp_p2(-p * Ic.A1.y, 0.0f);
    F.y = mp_p2(r, r) * Ic.A2 + mp_p2(-p * Ic.A2.y, 0.0f);
    F.z = mp_p2(r * Ic.A3.x, p * Ic.A3.y);
    Mq[i][i] = r * F.z.x + p * F.z.y;
#pragma unroll
    for (int k = i; k > 0; --k) {  // carry F from frame k to frame k-1, read the component along joint k-1
      const auto& Jk = M.j[k];
      mp_p_force_up_B(mp_p_cos<N>(js, k), mp_p_sin<N>(js, k), js.d[k], F);
      mp_p_force_up_A(Jk.ca, Jk.sa, Jk.a, F);
      const auto& Jp = M.j[k - 1];
      Mq[i][k - 1] = Jp.rev * F.z.x + (1.0f - Jp.rev) * F.z.y;
    }
    if (i > 0) {
      mp_p_rbi_up_B(mp_p_cos<N>(js, i), mp_p_sin<N>(js, i), c2[i], s2[i], js.d[i], Ic);
      mp_p_rbi_up_A(J.ca, J.sa, J.a, Ic);
    }
  }
}

// Cholesky solve on the LOWER triangle only (cf. mp_spd_solve, which is handed the full symmetric matrix)
template <int N>
MP_HD void mp_p_spd_solve_lower(float (&A)[N][N], float (&b)[N]) {
#pragma unroll
  for (int j = 0; j < N; ++j) {
    float d = A[j][j];
#pragma unroll
    for (int k = 0; k < j; ++k) d -= A[j][k] * A[j][k];
    const float inv = mp_rsqrt(d);
    A[j][j] = inv;
#pragma unroll
    for (int i = j + 1; i < N; ++i) {
      float v = A[i][j];
#pragma unroll
      for (int k = 0; k < j; ++k) v -= A[i][k] * A[j][k];
      A[i][j] = v * inv;
    }
  }
#pragma unroll
  for (int i = 0; i < N; ++i) {
    float v = b[i];
#pragma unroll
    for (int k = 0; k < i; ++k) v -= A[i][k] * b[k];
    b[i] = v * A[i][i];
  }
#pragma unroll
  for (int i = N - 1; i >= 0; --i) {
    float v = b[i];
#pragma unroll
    for (int k = i + 1; k < N; ++k) v -= A[k][i] * b[k];
    b[i] = v * A[i][i];
  }
}

// qdd = M(q)^-1 (tau - bias): the pair-native counterpart of mp_forward_dynamics<float, N, HAS_FTIP>
template <int N, bool HAS_FTIP, typename MT>
MP_HD void mp_p_forward_dynamics(const MT& M, const float (&a0)[3], const float (&tipn)[3], const float (&tipf)[3],
                                 const float (&q)[N], const float (&qd)[N], const float (&tau)[N], float (&qdd)[N]) {
  MpJointStateP<N> js;
  mp_p_joint_state<N>(M, q, js);
  float bias[N];
  mp_p_bias<N, HAS_FTIP>(M, a0, tipn, tipf, js, qd, bias);
  float c2[N], s2[N];  // double-angle terms for the xy block of the composite inertia
#pragma unroll
  for (int k = 0; k < MpJointStateP<N>::H; ++k) {
    const P2 cc = js.c[k] * js.c[k], ss = js.s[k] * js.s[k], sc = js.s[k] * js.c[k];
    const P2 C2 = cc - ss, S2 = sc + sc;
    c2[2 * k] = C2.x; s2[2 * k] = S2.x;
    if (2 * k + 1 < N) { c2[2 * k + 1] = C2.y; s2[2 * k + 1] = S2.y; }
  }
  float Mq[N][N];
  mp_p_mass_matrix<N>(M, js, c2, s2, Mq);
#pragma unroll
  for (int k = 0; k < N; ++k) qdd[k] = tau[k] - bias[k];
  mp_p_spd_solve_lower<N>(Mq, qdd);
}

#endif  // MP_HAS_PACKED
