// Host-side model compiler interface (see mp_model_compile.cpp).
#pragma once
#include <cmath>
#include <cstddef>

#include "mp_model.h"

// Returns 0 on success; otherwise a non-zero code and a message in `err`.  n <= MP_MAX_DOF / n <= MP_BIG_DOF.
int mp_compile_model(int n, const double* S, const double* Mcom, const double* G, const double* M_ee,
                     const double* joint_limits, const double* torque_limits, MpModel<double>* out, char* err,
                     size_t errlen);
int mp_compile_model_big(int n, const double* S, const double* Mcom, const double* G, const double* M_ee,
                         const double* joint_limits, const double* torque_limits, MpBigModel<double>* out, char* err,
                         size_t errlen);
void mp_compiled_fk(const MpModel<double>& m, const double* q, double* T16);
void mp_compiled_fk(const MpBigModel<double>& m, const double* q, double* T16);
void mp_make_call(const double base_R[9], const double base_p[3], const double g[3], const double Ftip[6], MpCall<double>* c);
template <int CAP>
inline void mp_make_call(const MpModelT<double, CAP>& m, const double g[3], const double Ftip[6], MpCall<double>* c) {
  mp_make_call(m.base_R, m.base_p, g, Ftip, c);
}

// precision cast; SC >= DC and the source holds at most DC joints when the capacities differ (narrowing a big model)
template <typename T, int SC, int DC>
inline void mp_model_cast(const MpModelT<double, SC>& s, MpModelT<T, DC>* d) {
  d->n = s.n;
  d->lscale = s.lscale;
  d->pad_[0] = d->pad_[1] = 0;
  for (int k = 0; k < 9; ++k) { d->base_R[k] = (T)s.base_R[k]; d->tool_R[k] = (T)s.tool_R[k]; }
  for (int k = 0; k < 3; ++k) { d->base_p[k] = (T)s.base_p[k]; d->tool_p[k] = (T)s.tool_p[k]; }
  for (int i = 0; i < DC; ++i) {
    const MpJoint<double>& a = s.j[i < SC ? i : SC - 1];
    MpJoint<T>& b = d->j[i];
    const bool in = i < SC;
    b.ca = in ? (T)a.ca : T(0); b.sa = in ? (T)a.sa : T(0); b.a = in ? (T)a.a : T(0); b.d = in ? (T)a.d : T(0);
    b.off = in ? (T)a.off : T(0); b.rev = in ? (T)a.rev : T(0);
    b.co = in ? (T)a.co : T(1); b.so = in ? (T)a.so : T(0);
    b.m = in ? (T)a.m : T(0); b.hx = in ? (T)a.hx : T(0); b.hy = in ? (T)a.hy : T(0); b.hz = in ? (T)a.hz : T(0);
    b.Ixx = in ? (T)a.Ixx : T(0); b.Ixy = in ? (T)a.Ixy : T(0); b.Ixz = in ? (T)a.Ixz : T(0);
    b.Iyy = in ? (T)a.Iyy : T(0); b.Iyz = in ? (T)a.Iyz : T(0); b.Izz = in ? (T)a.Izz : T(0);
    d->qmin[i] = in ? (T)s.qmin[i] : -(T)HUGE_VAL; d->qmax[i] = in ? (T)s.qmax[i] : (T)HUGE_VAL;
    d->taumin[i] = in ? (T)s.taumin[i] : -(T)HUGE_VAL; d->taumax[i] = in ? (T)s.taumax[i] : (T)HUGE_VAL;
  }
}

template <typename T>
inline void mp_call_cast(const MpCall<double>& s, MpCall<T>* d) {
  for (int k = 0; k < 3; ++k) { d->a0[k] = (T)s.a0[k]; d->F1n[k] = (T)s.F1n[k]; d->F1f[k] = (T)s.F1f[k]; }
  d->cold_model = nullptr;
  d->hard_rows = nullptr; d->hard_ctrl = nullptr; d->hard_next = nullptr; d->hard_cap = 0; d->hard_row_base = 0;
}
