// Host-side model compiler interface (see mp_model_compile.cpp).
#pragma once
#include <cstddef>

#include "mp_model.h"

// Returns 0 on success; otherwise a non-zero code and a message in `err`.
int mp_compile_model(int n, const double* S, const double* Mcom, const double* G, const double* M_ee,
                     const double* joint_limits, const double* torque_limits, MpModel<double>* out, char* err,
                     size_t errlen);
void mp_compiled_fk(const MpModel<double>& m, const double* q, double* T16);
void mp_make_call(const MpModel<double>& m, const double g[3], const double Ftip[6], MpCall<double>* c);

template <typename T>
inline void mp_model_cast(const MpModel<double>& s, MpModel<T>* d) {
  d->n = s.n;
  for (int k = 0; k < 9; ++k) { d->base_R[k] = (T)s.base_R[k]; d->tool_R[k] = (T)s.tool_R[k]; }
  for (int k = 0; k < 3; ++k) { d->base_p[k] = (T)s.base_p[k]; d->tool_p[k] = (T)s.tool_p[k]; }
  for (int i = 0; i < MP_MAX_DOF; ++i) {
    const MpJoint<double>& a = s.j[i];
    MpJoint<T>& b = d->j[i];
    b.ca = (T)a.ca; b.sa = (T)a.sa; b.a = (T)a.a; b.d = (T)a.d; b.off = (T)a.off; b.rev = (T)a.rev;
    b.m = (T)a.m; b.hx = (T)a.hx; b.hy = (T)a.hy; b.hz = (T)a.hz;
    b.Ixx = (T)a.Ixx; b.Ixy = (T)a.Ixy; b.Ixz = (T)a.Ixz; b.Iyy = (T)a.Iyy; b.Iyz = (T)a.Iyz; b.Izz = (T)a.Izz;
    d->qmin[i] = (T)s.qmin[i]; d->qmax[i] = (T)s.qmax[i];
    d->taumin[i] = (T)s.taumin[i]; d->taumax[i] = (T)s.taumax[i];
  }
}

template <typename T>
inline void mp_call_cast(const MpCall<double>& s, MpCall<T>* d) {
  for (int k = 0; k < 3; ++k) { d->a0[k] = (T)s.a0[k]; d->F1n[k] = (T)s.F1n[k]; d->F1f[k] = (T)s.F1f[k]; }
}
