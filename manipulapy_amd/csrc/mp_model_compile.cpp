// Host-side model compiler: reference tables (S, Mcom, G, M_ee) -> modified-DH link frames + origin
// inertias (see mp_model.h).  Pure C++/fp64, no device code: it also builds with g++ for the CPU tests.
//
// Input conventions are the reference's (ManipulaPy/urdf/core.py:670-769):
//   S     (6, n) row-major, column i = space screw [w; v] of joint i at the home pose; |w| = 1
//         (revolute, v = -w x q) or w = 0, |v| = 1 (prismatic)
//   Mcom  n x (4,4)  home pose of link i's CoM frame (Mlist_per_link)
//   G     n x (6,6)  spatial inertia of link i in that CoM frame, twist order [w; v];
//         must be blockdiag(Ic, m*1) (what Inertial.spatial_inertia, urdf/types.py:202-239, produces)
//   M_ee  (4,4)  end-effector home pose
#include "mp_model_compile.h"

#include <cmath>
#include <cstdio>
#include <cstring>

namespace {

struct V3 { double x, y, z; };
inline V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 operator*(double s, V3 a) { return {s * a.x, s * a.y, s * a.z}; }
inline double dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline V3 cross(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
inline double norm(V3 a) { return std::sqrt(dot(a, a)); }
inline V3 unit(V3 a) { double n = norm(a); return (1.0 / n) * a; }

// any unit vector perpendicular to z
V3 any_perp(V3 z) {
  V3 e = (std::fabs(z.x) < 0.9) ? V3{1, 0, 0} : V3{0, 1, 0};
  return unit(e - dot(e, z) * z);
}

struct Frame { V3 x, y, z, o; };  // axes (columns of R) and origin, in the space frame

// T = A^-1 * B for frames given as (R | o); returns R (row-major 9) and p
void rel(const Frame& A, const Frame& B, double R[9], double p[3]) {
  const V3 ax[3] = {A.x, A.y, A.z}, bx[3] = {B.x, B.y, B.z};
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) R[3 * r + c] = dot(ax[r], bx[c]);
  V3 w = B.o - A.o;
  p[0] = dot(A.x, w); p[1] = dot(A.y, w); p[2] = dot(A.z, w);
}

void mat4_mul(const double* A, const double* B, double* C) {
  double t[16];
  for (int r = 0; r < 4; ++r)
    for (int c = 0; c < 4; ++c) {
      double s = 0;
      for (int k = 0; k < 4; ++k) s += A[4 * r + k] * B[4 * k + c];
      t[4 * r + c] = s;
    }
  std::memcpy(C, t, sizeof t);
}

// reference utils/se3.py:33-42 (used only by the self-check)
void exp_twist(const double* S6, double th, double* T) {
  const double wx = S6[0], wy = S6[1], wz = S6[2];
  const double W[9] = {0, -wz, wy, wz, 0, -wx, -wy, wx, 0};
  double W2[9];
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) {
      double s = 0;
      for (int k = 0; k < 3; ++k) s += W[3 * r + k] * W[3 * k + c];
      W2[3 * r + c] = s;
    }
  const double s = std::sin(th), c = std::cos(th);
  std::memset(T, 0, 16 * sizeof(double));
  for (int r = 0; r < 3; ++r) {
    double pr = 0;
    for (int k = 0; k < 3; ++k) {
      const double I = (r == k) ? 1.0 : 0.0;
      T[4 * r + k] = I + s * W[3 * r + k] + (1 - c) * W2[3 * r + k];
      pr += (I * th + (1 - c) * W[3 * r + k] + (th - s) * W2[3 * r + k]) * S6[3 + k];
    }
    T[4 * r + 3] = pr;
  }
  T[15] = 1.0;
}

void fail(char* err, size_t errlen, const char* fmt, int i, double v) {
  if (err && errlen) std::snprintf(err, errlen, fmt, i, v);
}

}  // namespace

namespace {
// FK through the compiled chain (fp64), for the self-check and for host-side consumers.
template <int CAP>
void compiled_fk(const MpModelT<double, CAP>& m, const double* q, double* T /*16*/) {
  double A[16] = {m.base_R[0], m.base_R[1], m.base_R[2], m.base_p[0], m.base_R[3], m.base_R[4], m.base_R[5], m.base_p[1],
                  m.base_R[6], m.base_R[7], m.base_R[8], m.base_p[2], 0, 0, 0, 1};
  for (int i = 0; i < m.n; ++i) {
    const MpJoint<double>& j = m.j[i];
    const double th = j.off + j.rev * q[i], d = j.d + (1.0 - j.rev) * q[i];
    const double c = std::cos(th), s = std::sin(th);
    // Rx(alpha) Tx(a) Rz(th) Tz(d)
    const double X[16] = {c, -s, 0, j.a,
                          j.ca * s, j.ca * c, -j.sa, -j.sa * d,
                          j.sa * s, j.sa * c, j.ca, j.ca * d,
                          0, 0, 0, 1};
    mat4_mul(A, X, A);
  }
  const double Tl[16] = {m.tool_R[0], m.tool_R[1], m.tool_R[2], m.tool_p[0], m.tool_R[3], m.tool_R[4], m.tool_R[5], m.tool_p[1],
                         m.tool_R[6], m.tool_R[7], m.tool_R[8], m.tool_p[2], 0, 0, 0, 1};
  mat4_mul(A, Tl, T);
}
}  // namespace

void mp_compiled_fk(const MpModel<double>& m, const double* q, double* T) { compiled_fk<MP_MAX_DOF>(m, q, T); }
void mp_compiled_fk(const MpBigModel<double>& m, const double* q, double* T) { compiled_fk<MP_BIG_DOF>(m, q, T); }

namespace {
template <int CAP>
int compile_model(int n, const double* S, const double* Mcom, const double* G, const double* M_ee,
                  const double* joint_limits, const double* torque_limits, MpModelT<double, CAP>* out, char* err,
                  size_t errlen) {
  if (n < 1 || n > CAP) { fail(err, errlen, "dof %d outside 1..%g", n, (double)CAP); return 1; }
  std::memset(out, 0, sizeof(*out));
  out->n = n;
  const double EPS = 1e-9;

  // ---- 1. joint lines at the home pose
  V3 z[CAP], c[CAP];
  bool rev[CAP];
  for (int i = 0; i < n; ++i) {
    V3 w{S[0 * n + i], S[1 * n + i], S[2 * n + i]}, v{S[3 * n + i], S[4 * n + i], S[5 * n + i]};
    const double nw = norm(w), nv = norm(v);
    if (nw > 0.5) {
      if (std::fabs(nw - 1.0) > 1e-6) { fail(err, errlen, "joint %d: |w| = %g, revolute screws must be unit", i, nw); return 2; }
      if (std::fabs(dot(w, v)) > 1e-6) { fail(err, errlen, "joint %d: screw pitch %g != 0 unsupported", i, dot(w, v)); return 2; }
      rev[i] = true;
      z[i] = unit(w);
      c[i] = cross(z[i], v);  // point of the axis closest to the space origin
    } else {
      if (nw > 1e-9 || std::fabs(nv - 1.0) > 1e-6) { fail(err, errlen, "joint %d: prismatic screw needs w = 0, |v| = 1 (|v| = %g)", i, nv); return 2; }
      rev[i] = false;
      z[i] = unit(v);
      c[i] = V3{Mcom[16 * i + 3], Mcom[16 * i + 7], Mcom[16 * i + 11]};  // a prismatic axis has no position: put it through the CoM
    }
  }

  // ---- 2. modified-DH frame assignment: x_i = common normal from axis i to axis i+1
  Frame H[CAP];
  for (int i = 0; i < n; ++i) {
    Frame& F = H[i];
    F.z = z[i];
    // point where the previous common normal meets axis i (x_{i-1} must intersect axis i)
    V3 P = c[i];
    if (i > 0) P = H[i - 1].o + dot(c[i] - H[i - 1].o, H[i - 1].x) * H[i - 1].x;
    // project P onto axis i exactly (kills rounding drift)
    P = c[i] + dot(P - c[i], z[i]) * z[i];
    if (i + 1 < n) {
      V3 w = c[i + 1] - c[i];
      V3 nn = cross(z[i], z[i + 1]);
      const double ln = norm(nn);
      if (ln > 1e-6) {  // skew or intersecting axes
        V3 nh = (1.0 / ln) * nn;
        double a = dot(w, nh);
        const double t = dot(cross(w, z[i + 1]), nn) / (ln * ln);
        F.o = c[i] + t * z[i];
        F.x = (a < -EPS) ? (-1.0) * nh : nh;
      } else {  // parallel axes: the common normal is not unique, start it where the previous one landed
        V3 wp = w - dot(w, z[i]) * z[i];
        F.o = P;
        if (norm(wp) > EPS) F.x = unit(wp);
        else if (i > 0) {  // coincident axes: keep the previous x (made perpendicular)
          V3 xp = H[i - 1].x - dot(H[i - 1].x, z[i]) * z[i];
          F.x = (norm(xp) > 1e-6) ? unit(xp) : any_perp(z[i]);
        } else F.x = any_perp(z[i]);
      }
    } else {  // last link: no successor, zero offset / zero d
      F.o = P;
      if (i > 0) {
        V3 xp = H[i - 1].x - dot(H[i - 1].x, z[i]) * z[i];
        F.x = (norm(xp) > 1e-6) ? unit(xp) : any_perp(z[i]);
      } else F.x = any_perp(z[i]);
    }
    F.x = unit(F.x - dot(F.x, F.z) * F.z);
    F.y = cross(F.z, F.x);
  }

  // ---- 3. base transform, DH parameters (extracted numerically, then verified)
  {
    const Frame& F = H[0];
    const V3 ax[3] = {F.x, F.y, F.z};
    for (int r = 0; r < 3; ++r) {
      out->base_R[3 * r + 0] = (&ax[0].x)[r];
      out->base_R[3 * r + 1] = (&ax[1].x)[r];
      out->base_R[3 * r + 2] = (&ax[2].x)[r];
    }
    out->base_p[0] = F.o.x; out->base_p[1] = F.o.y; out->base_p[2] = F.o.z;
  }
  for (int i = 0; i < n; ++i) {
    MpJoint<double>& j = out->j[i];
    j.rev = rev[i] ? 1.0 : 0.0;
    j.co = 1; j.so = 0;
    if (i == 0) { j.ca = 1; j.sa = 0; j.a = 0; j.d = 0; j.off = 0; continue; }
    double R[9], p[3];
    rel(H[i - 1], H[i], R, p);
    // R = Rx(al) Rz(th) = [[c,-s,0],[ca s, ca c,-sa],[sa s, sa c, ca]],  p = (a, -sa d, ca d)
    const double al = std::atan2(-R[5], R[8]), th = std::atan2(-R[1], R[0]);
    j.ca = std::cos(al); j.sa = std::sin(al); j.a = p[0]; j.off = th;
    j.d = -j.sa * p[1] + j.ca * p[2];
    const double c = std::cos(th), s = std::sin(th);
    const double Rr[9] = {c, -s, 0, j.ca * s, j.ca * c, -j.sa, j.sa * s, j.sa * c, j.ca};
    const double pr[3] = {j.a, -j.sa * j.d, j.ca * j.d};
    double e = 0;
    for (int k = 0; k < 9; ++k) e = std::fmax(e, std::fabs(Rr[k] - R[k]));
    for (int k = 0; k < 3; ++k) e = std::fmax(e, std::fabs(pr[k] - p[k]));
    if (e > 1e-9) { fail(err, errlen, "joint %d: link transform is not modified-DH (residual %g)", i, e); return 3; }
    // snap exact right angles so cos/sin carry no 6e-17 dust into float32
    auto snap = [](double& v) { if (std::fabs(v) < 1e-14) v = 0; else if (std::fabs(v - 1) < 1e-14) v = 1; else if (std::fabs(v + 1) < 1e-14) v = -1; };
    snap(j.ca); snap(j.sa);
    if (std::fabs(j.a) < 1e-14) j.a = 0;
    if (std::fabs(j.d) < 1e-14) j.d = 0;
    if (std::fabs(j.off) < 1e-14) j.off = 0;
    j.co = std::cos(j.off); j.so = std::sin(j.off);
    snap(j.co); snap(j.so);
  }

  // ---- 4. link inertias about the link-frame origin
  for (int i = 0; i < n; ++i) {
    const double* Gi = G + 36 * i;
    const double m = Gi[3 * 6 + 3];
    double dev = 0, scale = std::fabs(m) + 1e-12;
    for (int r = 0; r < 3; ++r)
      for (int cc = 0; cc < 3; ++cc) {
        dev = std::fmax(dev, std::fabs(Gi[r * 6 + 3 + cc]));
        dev = std::fmax(dev, std::fabs(Gi[(3 + r) * 6 + cc]));
        dev = std::fmax(dev, std::fabs(Gi[(3 + r) * 6 + 3 + cc] - (r == cc ? m : 0.0)));
        scale = std::fmax(scale, std::fabs(Gi[r * 6 + cc]));
      }
    if (dev > 1e-9 * scale) { fail(err, errlen, "link %d: G must be blockdiag(Ic, m*1) (deviation %g)", i, dev); return 4; }
    // CoM frame in link frame i
    const double* Mc = Mcom + 16 * i;
    Frame C;
    C.x = {Mc[0], Mc[4], Mc[8]}; C.y = {Mc[1], Mc[5], Mc[9]}; C.z = {Mc[2], Mc[6], Mc[10]}; C.o = {Mc[3], Mc[7], Mc[11]};
    double Rc[9], cp[3];
    rel(H[i], C, Rc, cp);
    double Ic[9], Io[9];
    for (int r = 0; r < 3; ++r)
      for (int cc = 0; cc < 3; ++cc) Ic[3 * r + cc] = 0.5 * (Gi[r * 6 + cc] + Gi[cc * 6 + r]);
    // Io = Rc Ic Rc^T + m (|c|^2 1 - c c^T)
    double t[9];
    for (int r = 0; r < 3; ++r)
      for (int cc = 0; cc < 3; ++cc) {
        double s = 0;
        for (int k = 0; k < 3; ++k) s += Rc[3 * r + k] * Ic[3 * k + cc];
        t[3 * r + cc] = s;
      }
    const double c2 = cp[0] * cp[0] + cp[1] * cp[1] + cp[2] * cp[2];
    for (int r = 0; r < 3; ++r)
      for (int cc = 0; cc < 3; ++cc) {
        double s = 0;
        for (int k = 0; k < 3; ++k) s += t[3 * r + k] * Rc[3 * cc + k];
        Io[3 * r + cc] = s + m * ((r == cc ? c2 : 0.0) - cp[r] * cp[cc]);
      }
    MpJoint<double>& j = out->j[i];
    j.m = m; j.hx = m * cp[0]; j.hy = m * cp[1]; j.hz = m * cp[2];
    j.Ixx = Io[0]; j.Ixy = 0.5 * (Io[1] + Io[3]); j.Ixz = 0.5 * (Io[2] + Io[6]);
    j.Iyy = Io[4]; j.Iyz = 0.5 * (Io[5] + Io[7]); j.Izz = Io[8];
  }

  {  // length scale for the float32 kernels' conditioning test: the longest lever a force meets on its way down - a joint-to-joint
     // offset, or a link's centre of mass seen from its joint (round 5: a short fat chain whose joints coincide still multiplies its
     // weight by the distance of its centres of mass) - never zero
    double L = 0;
    for (int i = 0; i < n; ++i) {
      const MpJoint<double>& j = out->j[i];
      L = std::fmax(L, std::fabs(j.a) + std::fabs(j.d));
      if (j.m > 0) L = std::fmax(L, std::sqrt(j.hx * j.hx + j.hy * j.hy + j.hz * j.hz) / j.m);
    }
    out->lscale = (float)std::fmax(L, 1e-3);
  }

  // ---- 5. tool frame, limits
  {
    Frame E;
    E.x = {M_ee[0], M_ee[4], M_ee[8]}; E.y = {M_ee[1], M_ee[5], M_ee[9]}; E.z = {M_ee[2], M_ee[6], M_ee[10]};
    E.o = {M_ee[3], M_ee[7], M_ee[11]};
    rel(H[n - 1], E, out->tool_R, out->tool_p);
  }
  const double INF = HUGE_VAL;
  for (int i = 0; i < CAP; ++i) {
    const bool in = i < n;
    // the planner holds limits as float32 (reference planning/trajectory_planning.py:218-223)
    out->qmin[i] = (in && joint_limits) ? (double)(float)joint_limits[2 * i] : -INF;
    out->qmax[i] = (in && joint_limits) ? (double)(float)joint_limits[2 * i + 1] : INF;
    out->taumin[i] = (in && torque_limits) ? (double)(float)torque_limits[2 * i] : -INF;
    out->taumax[i] = (in && torque_limits) ? (double)(float)torque_limits[2 * i + 1] : INF;
  }

  // ---- 6. self-check: compiled chain == product of exponentials (reference kinematics/fk.py:59-70)
  for (int trial = 0; trial < 4; ++trial) {
    double q[CAP], T[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1}, E[16], Tc[16];
    for (int i = 0; i < n; ++i) {
      q[i] = (trial == 0) ? 0.0 : std::sin(12.9898 * (i + 1) + 78.233 * trial) * (rev[i] ? 2.5 : 0.05);
      double Si[6];
      for (int k = 0; k < 6; ++k) Si[k] = S[k * n + i];
      exp_twist(Si, q[i], E);
      mat4_mul(T, E, T);
    }
    mat4_mul(T, M_ee, T);
    compiled_fk<CAP>(*out, q, Tc);
    double e = 0;
    for (int k = 0; k < 16; ++k) e = std::fmax(e, std::fabs(T[k] - Tc[k]));
    if (e > 1e-9) { fail(err, errlen, "self-check %d: compiled FK differs from PoE FK by %g", trial, e); return 5; }
  }
  return 0;
}
}  // namespace

int mp_compile_model(int n, const double* S, const double* Mcom, const double* G, const double* M_ee,
                     const double* joint_limits, const double* torque_limits, MpModel<double>* out, char* err,
                     size_t errlen) {
  return compile_model<MP_MAX_DOF>(n, S, Mcom, G, M_ee, joint_limits, torque_limits, out, err, errlen);
}
int mp_compile_model_big(int n, const double* S, const double* Mcom, const double* G, const double* M_ee,
                         const double* joint_limits, const double* torque_limits, MpBigModel<double>* out, char* err,
                         size_t errlen) {
  return compile_model<MP_BIG_DOF>(n, S, Mcom, G, M_ee, joint_limits, torque_limits, out, err, errlen);
}

// per-call constants (gravity, tip wrench) seen from the frame link 1 is attached to
void mp_make_call(const double R[9], const double p[3], const double g[3], const double Ftip[6], MpCall<double>* c) {
  c->cold_model = nullptr;
  c->hard_rows = nullptr; c->hard_ctrl = nullptr; c->hard_next = nullptr; c->hard_cap = 0; c->hard_row_base = 0;
  for (int k = 0; k < 3; ++k) c->a0[k] = -(R[0 + k] * g[0] + R[3 + k] * g[1] + R[6 + k] * g[2]);
  double n[3] = {0, 0, 0}, f[3] = {0, 0, 0};
  if (Ftip) { n[0] = Ftip[0]; n[1] = Ftip[1]; n[2] = Ftip[2]; f[0] = Ftip[3]; f[1] = Ftip[4]; f[2] = Ftip[5]; }
  // wrench parent -> child coordinates: f_c = R^T f, n_c = R^T (n - p x f)
  const double nx = n[0] - (p[1] * f[2] - p[2] * f[1]), ny = n[1] - (p[2] * f[0] - p[0] * f[2]),
               nz = n[2] - (p[0] * f[1] - p[1] * f[0]);
  for (int k = 0; k < 3; ++k) {
    c->F1n[k] = R[0 + k] * nx + R[3 + k] * ny + R[6 + k] * nz;
    c->F1f[k] = R[0 + k] * f[0] + R[3 + k] * f[1] + R[6 + k] * f[2];
  }
}
