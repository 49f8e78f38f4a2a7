/* CPU ORACLE (C restatement) — test infrastructure, NOT product code.
 *
 * The reference's NumPy-CPU algorithm for inverse dynamics, restated in plain C so that bench.py's
 * `cpu_baseline` leg can time the REFERENCE ALGORITHM (1 + 2n mass matrices per point, central-difference
 * Christoffel symbols) at native speed on all host cores instead of at NumPy-interpreter speed.  Only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may load this library; nothing under
 * manipulapy_amd/ does.
 *
 * Parity status: PINNED — tests/test_oracle_golden.py::test_c_oracle_* checks it against oracle/ref_numpy.py
 * (itself pinned to the reference's golden vectors) and against tests/golden/dynamics_*.npz.
 *
 * Every function cites the reference file:line (relative to ManipulaPy/) whose arithmetic it follows.
 * Layouts: S (6,n) row-major; Mcom n x (4,4); G n x (6,6); q/qd/qdd/tau (rows,n); all float64.
 */
#include <math.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define MAXN 8

static void mat4_mul(const double* A, const double* B, double* C) {
  double t[16];
  for (int r = 0; r < 4; ++r)
    for (int c = 0; c < 4; ++c) {
      double s = 0;
      for (int k = 0; k < 4; ++k) s += A[4 * r + k] * B[4 * k + c];
      t[4 * r + c] = s;
    }
  memcpy(C, t, sizeof t);
}

/* rigid inverse; the reference calls np.linalg.inv on the 4x4 (dynamics/mass_matrix.py:77,82) */
static void mat4_inv_rigid(const double* T, double* I) {
  double t[16] = {0};
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) t[4 * r + c] = T[4 * c + r];
  for (int r = 0; r < 3; ++r) t[4 * r + 3] = -(t[4 * r + 0] * T[3] + t[4 * r + 1] * T[7] + t[4 * r + 2] * T[11]);
  t[15] = 1.0;
  memcpy(I, t, sizeof t);
}

/* utils/se3.py:33-42 — exp of a unit (or zero) angular-velocity screw */
static void exp_twist(const double* S6, double th, double* T) {
  const double wx = S6[0], wy = S6[1], wz = S6[2];
  const double W[9] = {0, -wz, wy, wz, 0, -wx, -wy, wx, 0};
  double W2[9];
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) {
      double s = 0;
      for (int k = 0; k < 3; ++k) s += W[3 * r + k] * W[3 * k + c];
      W2[3 * r + c] = s;
    }
  const double s = sin(th), c = cos(th);
  memset(T, 0, 16 * sizeof(double));
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

/* utils/se3.py:45-52 — [[R,0],[[p]R,R]] */
static void adjoint(const double* T, double* A /*36*/) {
  const double p[3] = {T[3], T[7], T[11]};
  const double P[9] = {0, -p[2], p[1], p[2], 0, -p[0], -p[1], p[0], 0};
  memset(A, 0, 36 * sizeof(double));
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) {
      const double R = T[4 * r + c];
      A[6 * r + c] = R;
      A[6 * (r + 3) + c + 3] = R;
      double s = 0;
      for (int k = 0; k < 3; ++k) s += P[3 * r + k] * T[4 * k + c];
      A[6 * (r + 3) + c] = s;
    }
}

/* kinematics/jacobian.py:62-73 — column i = Ad(prod_{j<i} exp) S_i; also returns the prefix products
 * P_k = prod_{j<=k} exp([S_j] th_j) that kinematics/fk.py:59-70 would recompute for every truncation */
static void jacobian_and_prefix(int n, const double* S, const double* th, double* Js /*6 x n*/, double* P /*n x 16*/) {
  double T[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1}, A[36], E[16], Si[6];
  for (int i = 0; i < n; ++i) {
    for (int k = 0; k < 6; ++k) Si[k] = S[k * n + i];
    adjoint(T, A);
    for (int r = 0; r < 6; ++r) {
      double s = 0;
      for (int k = 0; k < 6; ++k) s += A[6 * r + k] * Si[k];
      Js[r * n + i] = s;
    }
    exp_twist(Si, th[i], E);
    mat4_mul(T, E, T);
    memcpy(P + 16 * i, T, sizeof T);
  }
}

/* dynamics/mass_matrix.py:62-99 (+ optionally dynamics/forces.py:100-133 with the same per-link Jacobians):
 * M = sum_k J_k^T G_k J_k, J_k[:, :k+1] = Ad(inv(T_k_com)) Js[:, :k+1], T_k_com = FK(th[:k+1]) inv(FK(0)) Mcom_k;
 * grav_i += (J_k^T [0; m_k R_k^T (-g)])_i */
static void mass_matrix_gravity(int n, const double* S, const double* Mcom, const double* G, const double* M_ee,
                                const double* th, const double* g, double* M /*n x n*/, double* grav /*n or NULL*/) {
  double Js[6 * MAXN], P[16 * MAXN], Minv[16], Tk[16], Tkcom[16], Tinv[16], A[36], Jk[6 * MAXN], GJ[6 * MAXN];
  jacobian_and_prefix(n, S, th, Js, P);
  mat4_inv_rigid(M_ee, Minv);
  memset(M, 0, (size_t)n * n * sizeof(double));
  if (grav) memset(grav, 0, (size_t)n * sizeof(double));
  for (int k = 0; k < n; ++k) {
    mat4_mul(P + 16 * k, M_ee, Tk);                 /* FK(theta[:k+1]) */
    double L[16];
    mat4_mul(Minv, Mcom + 16 * k, L);               /* inv(FK(0_{k+1})) @ Mlist_per_link[k];  FK(0) = M_ee */
    mat4_mul(Tk, L, Tkcom);
    mat4_inv_rigid(Tkcom, Tinv);
    adjoint(Tinv, A);
    for (int r = 0; r < 6; ++r)
      for (int c = 0; c <= k; ++c) {
        double s = 0;
        for (int j = 0; j < 6; ++j) s += A[6 * r + j] * Js[j * n + c];
        Jk[r * n + c] = s;
      }
    const double* Gk = G + 36 * k;
    for (int r = 0; r < 6; ++r)
      for (int c = 0; c <= k; ++c) {
        double s = 0;
        for (int j = 0; j < 6; ++j) s += Gk[6 * r + j] * Jk[j * n + c];
        GJ[r * n + c] = s;
      }
    for (int a = 0; a <= k; ++a)
      for (int b = 0; b <= k; ++b) {
        double s = 0;
        for (int j = 0; j < 6; ++j) s += Jk[j * n + a] * GJ[j * n + b];
        M[a * n + b] += s;
      }
    if (grav) {
      const double m = Gk[6 * 3 + 3];
      double f[3];
      for (int c = 0; c < 3; ++c) f[c] = -m * (Tkcom[0 + c] * g[0] + Tkcom[4 + c] * g[1] + Tkcom[8 + c] * g[2]);
      for (int a = 0; a <= k; ++a) grav[a] += Jk[3 * n + a] * f[0] + Jk[4 * n + a] * f[1] + Jk[5 * n + a] * f[2];
    }
  }
  for (int a = 0; a < n; ++a)                        /* 0.5 (M + M^T), mass_matrix.py:96 */
    for (int b = a + 1; b < n; ++b) {
      const double s = 0.5 * (M[a * n + b] + M[b * n + a]);
      M[a * n + b] = M[b * n + a] = s;
    }
}

/* dynamics/id_fd.py:37-48 with dynamics/cache.py:39-52 (eps = 1e-6 central difference) and
 * dynamics/forces.py:45-58 (Christoffel quadratic form) */
static void inverse_dynamics_row(int n, const double* S, const double* Mcom, const double* G, const double* M_ee,
                                 const double* th, const double* dth, const double* ddth, const double* g, const double* F,
                                 double* tau) {
  const double eps = 1e-6;
  double M[MAXN * MAXN], grav[MAXN], dM[MAXN * MAXN * MAXN], Mp[MAXN * MAXN], Mm[MAXN * MAXN], t[MAXN];
  mass_matrix_gravity(n, S, Mcom, G, M_ee, th, g, M, grav);
  for (int k = 0; k < n; ++k) {
    memcpy(t, th, (size_t)n * sizeof(double));
    t[k] = th[k] + eps;
    mass_matrix_gravity(n, S, Mcom, G, M_ee, t, g, Mp, 0);
    t[k] = th[k] - eps;
    mass_matrix_gravity(n, S, Mcom, G, M_ee, t, g, Mm, 0);
    for (int i = 0; i < n; ++i)
      for (int j = 0; j < n; ++j) dM[(i * n + j) * n + k] = (Mp[i * n + j] - Mm[i * n + j]) / (2.0 * eps);
  }
  double Js[6 * MAXN], P[16 * MAXN];
  jacobian_and_prefix(n, S, th, Js, P);
  for (int i = 0; i < n; ++i) {
    double c = 0;
    for (int j = 0; j < n; ++j)
      for (int k = 0; k < n; ++k)   /* Gamma_i[j,k] = 0.5 (dM[i,j,k] + dM[i,k,j] - dM[j,k,i]) */
        c += dth[j] * 0.5 * (dM[(i * n + j) * n + k] + dM[(i * n + k) * n + j] - dM[(j * n + k) * n + i]) * dth[k];
    double s = c + grav[i];
    for (int j = 0; j < n; ++j) s += M[i * n + j] * ddth[j];
    for (int r = 0; r < 6; ++r) s += Js[r * n + i] * F[r];
    tau[i] = s;
  }
}

/* planning/trajectory_dynamics.py:345-358 — the per-row loop (without the float32 cast / clip, which the
 * caller applies); rows are independent -> OpenMP over rows.  Returns the number of threads used. */
int oracle_inverse_dynamics_rows(int n, const double* S, const double* Mcom, const double* G, const double* M_ee,
                                 const double* q, const double* qd, const double* qdd, const double* g, const double* Ftip,
                                 long rows, double* tau, int nthreads) {
  if (n < 1 || n > MAXN) return -1;
  int used = 1;
#ifdef _OPENMP
  if (nthreads > 0) omp_set_num_threads(nthreads);
#pragma omp parallel
  {
#pragma omp single
    used = omp_get_num_threads();
#pragma omp for schedule(dynamic, 16)
    for (long r = 0; r < rows; ++r)
      inverse_dynamics_row(n, S, Mcom, G, M_ee, q + r * n, qd + r * n, qdd + r * n, g, Ftip, tau + r * n);
  }
#else
  (void)nthreads;
  for (long r = 0; r < rows; ++r)
    inverse_dynamics_row(n, S, Mcom, G, M_ee, q + r * n, qd + r * n, qdd + r * n, g, Ftip, tau + r * n);
#endif
  return used;
}

/* mass matrix per row (dynamics/mass_matrix.py:16-99), for the pin tests */
int oracle_mass_matrix_rows(int n, const double* S, const double* Mcom, const double* G, const double* M_ee,
                            const double* q, long rows, double* M) {
  if (n < 1 || n > MAXN) return -1;
  const double g0[3] = {0, 0, 0};
  for (long r = 0; r < rows; ++r) mass_matrix_gravity(n, S, Mcom, G, M_ee, q + r * n, g0, M + r * n * n, 0);
  return 0;
}

/* np.linalg.solve (LAPACK dgesv: LU with partial pivoting), as dynamics/id_fd.py:82 calls it */
static int lu_solve(int n, double* A /*n x n, destroyed*/, double* b /*n, in: rhs, out: x*/) {
  for (int k = 0; k < n; ++k) {
    int p = k;
    double best = fabs(A[k * n + k]);
    for (int i = k + 1; i < n; ++i)
      if (fabs(A[i * n + k]) > best) { best = fabs(A[i * n + k]); p = i; }
    if (best == 0.0) return -1;
    if (p != k) {
      for (int j = 0; j < n; ++j) { const double t = A[k * n + j]; A[k * n + j] = A[p * n + j]; A[p * n + j] = t; }
      const double t = b[k]; b[k] = b[p]; b[p] = t;
    }
    for (int i = k + 1; i < n; ++i) {
      const double f = A[i * n + k] / A[k * n + k];
      A[i * n + k] = f;
      for (int j = k + 1; j < n; ++j) A[i * n + j] -= f * A[k * n + j];
      b[i] -= f * b[k];
    }
  }
  for (int i = n - 1; i >= 0; --i) {
    double s = b[i];
    for (int j = i + 1; j < n; ++j) s -= A[i * n + j] * b[j];
    b[i] = s / A[i * n + i];
  }
  return 0;
}

/* dynamics/id_fd.py:71-83 — qdd = solve(M, tau - c - g - Js^T Ftip).  c + g + Js^T Ftip is the inverse
 * dynamics at zero acceleration (id_fd.py:37-48 with ddtheta = 0), M comes from the same mass_matrix. */
static int forward_dynamics_row(int n, const double* S, const double* Mcom, const double* G, const double* M_ee,
                                const double* th, const double* dth, const double* tau, const double* g, const double* F,
                                double* qdd) {
  double M[MAXN * MAXN], bias[MAXN], zero[MAXN] = {0};
  const double g0[3] = {0, 0, 0};
  inverse_dynamics_row(n, S, Mcom, G, M_ee, th, dth, zero, g, F, bias);
  mass_matrix_gravity(n, S, Mcom, G, M_ee, th, g0, M, 0);
  for (int i = 0; i < n; ++i) qdd[i] = tau[i] - bias[i];
  return lu_solve(n, M, qdd);
}

int oracle_forward_dynamics_rows(int n, const double* S, const double* Mcom, const double* G, const double* M_ee,
                                 const double* q, const double* qd, const double* tau, const double* g, const double* Ftip,
                                 long rows, double* qdd) {
  if (n < 1 || n > MAXN) return -1;
  for (long r = 0; r < rows; ++r)
    if (forward_dynamics_row(n, S, Mcom, G, M_ee, q + r * n, qd + r * n, tau + r * n, g, Ftip, qdd + r * n)) return -2;
  return 0;
}

/* planning/trajectory_dynamics.py:580-708 — B independent roll-outs (the reference integrates one; the batch is a
 * loop over them): semi-implicit Euler, intRes sub-steps of dt / intRes, q clipped to the float32 joint limits after
 * every sub-step, row 0 = initial state with zero acceleration, taumat[i] / Ftipmat[i] drive step i, rows stored
 * float32, the recorded acceleration is the last sub-step's.
 * state_f32 = 0: the state is float64 (what the reference does with float64 inputs — THE parity oracle, SURVEY
 *   §0.5d); state_f32 = 1: the state is rounded to float32 after every update exactly as NumPy types it with
 *   float32 inputs (dtheta: float64 sum cast back; theta: float32 product and sum), while the dynamics are still
 *   evaluated in float64 — a diagnostic that separates state rounding from arithmetic error (the reference itself
 *   would do its trigonometry and the 1e-6 finite difference in float32 there, which is numerically meaningless).
 * lim: (n, 2) float64 holding float32-representable limits.  Ftipmat may be NULL (zero wrench).
 * OpenMP over trajectories.  Returns threads used, < 0 on error. */
int oracle_fd_trajectory(int n, const double* S, const double* Mcom, const double* G, const double* M_ee, const double* lim,
                         const double* theta0, const double* dtheta0, const double* taumat, const double* g,
                         const double* Ftipmat, long B, long Nt, double dt, int intRes, int state_f32, float* pos, float* vel,
                         float* acc, int nthreads) {
  if (n < 1 || n > MAXN || Nt < 1 || intRes < 1) return -1;
  int used = 1, fail = 0;
  const double h = dt / intRes;
#ifdef _OPENMP
  if (nthreads > 0) omp_set_num_threads(nthreads);
#pragma omp parallel
  {
#pragma omp single
    used = omp_get_num_threads();
#pragma omp for schedule(dynamic, 1)
#endif
    for (long b = 0; b < B; ++b) {
      double q[MAXN], qd[MAXN], qdd[MAXN], F0[6] = {0};
      for (int j = 0; j < n; ++j) {
        q[j] = state_f32 ? (double)(float)theta0[b * n + j] : theta0[b * n + j];
        qd[j] = state_f32 ? (double)(float)dtheta0[b * n + j] : dtheta0[b * n + j];
        pos[(b * Nt) * n + j] = (float)q[j];
        vel[(b * Nt) * n + j] = (float)qd[j];
        acc[(b * Nt) * n + j] = 0.0f;
      }
      for (long i = 1; i < Nt; ++i) {
        const double* tau = taumat + (b * Nt + i) * n;
        const double* F = Ftipmat ? Ftipmat + (b * Nt + i) * 6 : F0;
        for (int j = 0; j < n; ++j) qdd[j] = 0.0;
        for (int k = 0; k < intRes; ++k) {
          if (forward_dynamics_row(n, S, Mcom, G, M_ee, q, qd, tau, g, F, qdd)) { fail = 1; break; }
          for (int j = 0; j < n; ++j) {
            if (state_f32) {
              const float v = (float)(qd[j] + qdd[j] * h);            /* float32 + float64 -> float64, cast back */
              const float hf = (float)h;
              volatile float prod = v * hf;                            /* float32 array * python float -> float32 */
              float p = (float)q[j] + prod;
              const float lo = (float)lim[2 * j], hi = (float)lim[2 * j + 1];
              p = p < lo ? lo : (p > hi ? hi : p);
              qd[j] = v; q[j] = p;
            } else {
              qd[j] = qd[j] + qdd[j] * h;
              double p = q[j] + qd[j] * h;
              p = p < lim[2 * j] ? lim[2 * j] : (p > lim[2 * j + 1] ? lim[2 * j + 1] : p);
              q[j] = p;
            }
          }
        }
        for (int j = 0; j < n; ++j) {
          pos[(b * Nt + i) * n + j] = (float)q[j];
          vel[(b * Nt + i) * n + j] = (float)qd[j];
          acc[(b * Nt + i) * n + j] = (float)qdd[j];
        }
      }
    }
#ifdef _OPENMP
  }
#endif
  return fail ? -2 : used;
}
