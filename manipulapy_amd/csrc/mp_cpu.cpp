// CPU twins of the hot-path entry points (include/manipula_hip.h, "*_cpu"): the SAME per-row templates the HIP
// kernels instantiate (mp_core.h), compiled for the host and run over the rows by a small std::thread pool.
// They are what the kernel registry's cpu_launchers call when the reference's own routing rule sends an operation to
// the CPU (NumPy backend active, or use_cuda=False): reference cuda_kernels/registry.py:85-89 picks
// `gpu_launcher if _cuda_routing_enabled() else cpu_launcher`.  They are NOT a fallback of the GPU path - a failing
// or missing GPU under the "hip" backend raises - and they are the product's own analytic recursion, not the test
// suite's restatement of the reference's 1 + 2n mass-matrix algorithm.  No HIP call is made here.
#include <sched.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <type_traits>
#include <vector>

#include "../../include/manipula_hip.h"
#include "mp_core.h"
#include "mp_dyn.h"
#include "mp_ik.h"
#include "mp_handles.h"
#include "mp_model_compile.h"

namespace {
const double kG[3] = {0.0, 0.0, -9.81};

int fail(const char* msg) { return mp_set_error(MP_ERR_INVALID, msg); }

// Threads a launcher takes when the caller names none: every core this process may run on (its affinity mask) - but a container is
// often SHOWN more cores than it is granted time on (the MI355X boxes of this project: 256 visible, a cgroup quota of 16 CPUs), and
// one thread per visible core then loses to a fraction of them (the CPU baseline of bench.py: 256 threads 0.27 M rows/s, 64: 0.68): at
// most four threads per CPU of a cgroup quota (v2 cpu.max, v1 cfs_quota_us / cfs_period_us).
int default_threads() {
  static const int n = [] {
    int have = 0;
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof set, &set) == 0) have = CPU_COUNT(&set);
    if (have <= 0) have = (int)std::thread::hardware_concurrency();
    if (have <= 0) have = 1;
    double quota = 0;
    if (FILE* f = std::fopen("/sys/fs/cgroup/cpu.max", "r")) {
      char q[32];
      long period = 0;
      if (std::fscanf(f, "%31s %ld", q, &period) == 2 && std::strcmp(q, "max") != 0 && period > 0) quota = std::atof(q) / (double)period;
      std::fclose(f);
    } else {
      long q = 0, period = 0;
      if (FILE* a = std::fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) { if (std::fscanf(a, "%ld", &q) != 1) q = 0; std::fclose(a); }
      if (FILE* b = std::fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (std::fscanf(b, "%ld", &period) != 1) period = 0; std::fclose(b); }
      if (q > 0 && period > 0) quota = (double)q / (double)period;
    }
    if (quota > 0) have = std::min(have, std::max(1, (int)std::ceil(4.0 * quota)));
    return have;
  }();
  return n;
}
int thread_count(int64_t items, int64_t grain, int nthreads) {
  int want = nthreads;
  if (want <= 0) {
    if (const char* e = getenv("MANIPULAPY_CPU_THREADS")) want = atoi(e);
    if (want <= 0) want = default_threads();
  }
  const int64_t by_work = (items + grain - 1) / grain;
  return (int)std::max<int64_t>(1, std::min<int64_t>(want, by_work));
}

// fn(lo, hi) over [0, items) in contiguous slices, one per thread; small inputs stay on the calling thread.  A thread that
// cannot be started (std::system_error under a pid / thread limit) must not cross the C ABI as an exception: its slice, and
// every later one, runs on the calling thread instead, and the threads already started are joined either way.
template <class F>
int parallel_for(int64_t items, int64_t grain, int nthreads, F fn) {
  const int T = thread_count(items, grain, nthreads);
  if (T <= 1) { fn((int64_t)0, items); return 1; }
  std::vector<std::thread> pool;
  pool.reserve(T - 1);
  const int64_t per = (items + T - 1) / T;
  int started = 1;
  for (int t = 1; t < T; ++t) {
    const int64_t lo = std::min(items, t * per), hi = std::min(items, lo + per);
    if (lo >= hi) continue;
    bool spawned = false;
    try {
      pool.emplace_back([=] { fn(lo, hi); });
      spawned = true;
      ++started;
    } catch (...) {
    }
    if (!spawned) fn(lo, hi);
  }
  fn((int64_t)0, std::min(items, per));
  for (auto& th : pool) th.join();
  return started;
}

bool any_nonzero(const double* F) {
  if (!F) return false;
  for (int k = 0; k < 6; ++k)
    if (F[k] != 0.0) return true;
  return false;
}

template <typename T> const MpModel<T>& pick(const mp_model* m);
template <> const MpModel<float>& pick<float>(const mp_model* m) { return m->f; }
template <> const MpModel<double>& pick<double>(const mp_model* m) { return m->d; }

template <typename T> const MpBigModel<T>& pick_big(const mp_model* m);
template <> const MpBigModel<float>& pick_big<float>(const mp_model* m) { return m->bf; }
template <> const MpBigModel<double>& pick_big<double>(const mp_model* m) { return m->bd; }

template <typename T>
MpCall<T> make_call(const mp_model* m, const double* g, const double* Ftip) {
  MpCall<double> cd;
  if (m->big) mp_make_call(m->bd, g ? g : kG, Ftip, &cd);
  else mp_make_call(m->d, g ? g : kG, Ftip, &cd);
  MpCall<T> c;
  mp_call_cast(cd, &c);
  // the float64 model for the re-evaluated float32 rows (mp_core.h, mp_rnea_row / mp_dyn.h, mp_dyn_row_id_f64)
  c.cold_model = m->big ? (const void*)&m->bd : (const void*)&m->d;
  return c;
}

#define MP_CPU_DISPATCH(n, ...)                                  \
  switch (n) {                                                   \
    case 1: { constexpr int N = 1; __VA_ARGS__; } break;         \
    case 2: { constexpr int N = 2; __VA_ARGS__; } break;         \
    case 3: { constexpr int N = 3; __VA_ARGS__; } break;         \
    case 4: { constexpr int N = 4; __VA_ARGS__; } break;         \
    case 5: { constexpr int N = 5; __VA_ARGS__; } break;         \
    case 6: { constexpr int N = 6; __VA_ARGS__; } break;         \
    case 7: { constexpr int N = 7; __VA_ARGS__; } break;         \
    case 8: { constexpr int N = 8; __VA_ARGS__; } break;         \
    default: return fail("dof outside 1..8 (larger models take the looped path before this dispatch)");                    \
  }

// ---- one row of FK / Jacobian / inverse dynamics: the body of k_fk_jac_id / k_id on the host
template <typename T, int N, bool F>
void rows_fk_jac_id(const MpModel<T>& M, const MpCall<T>& C, const T* q, const T* qd, const T* qdd, T* Tout, T* Jout, T* tau,
                    int64_t lo, int64_t hi) {
  for (int64_t r = lo; r < hi; ++r) {
    T a[N];
    for (int j = 0; j < N; ++j) a[j] = q[r * N + j];
    MpJointState<T, N> js;
    mp_joint_state<T, N>(M, a, js);
    MpBad<T> bad;
    bad.add(a);
    if (Tout || Jout) {
      T TT[16], JJ[6 * N];
      mp_fk_jac<T, N, true>(M, js, TT, JJ);
      mp_poison_if(bad.any(), TT);
      mp_poison_if(bad.any(), JJ);
      if (Tout) std::memcpy(Tout + r * 16, TT, sizeof TT);
      if (Jout) std::memcpy(Jout + r * 6 * N, JJ, sizeof JJ);
    }
    if (tau) {
      T b[N], c[N], t[N];
      for (int j = 0; j < N; ++j) { b[j] = qd[r * N + j]; c[j] = qdd[r * N + j]; }
      mp_rnea_row<T, N, F>(M, C, js, a, b, c, t);
      for (int j = 0; j < N; ++j) t[j] = mp_clip(t[j], M.taumin[j], M.taumax[j]);
      bad.add(b); bad.add(c);
      mp_poison_if(bad.any(), t);
      std::memcpy(tau + r * N, t, sizeof t);
    }
  }
}

template <typename T>
int fk_jac_id_cpu(const char* fn, const mp_model* model, const T* q, const T* qd, const T* qdd, int64_t rows, const double* g,
                  const double* Ftip, T* Tout, T* Jout, T* tau, int nthreads) {
  if (!model) return fail("null model");
  if (rows < 0) return fail("negative row count");
  if (rows == 0) return MP_OK;
  if (!q || !(Tout || Jout || tau) || (tau && !(qd && qdd))) return fail(fn);
  const MpModel<T>& M = pick<T>(model);
  const MpCall<T> C = make_call<T>(model, g, Ftip);
  const bool ftip = any_nonzero(Ftip);
  if (model->big) {  // 9..32 joints: the looped rows of csrc/mp_dyn.h
    const MpBigModel<T>& MB = pick_big<T>(model);
    parallel_for(rows, 128, nthreads, [&](int64_t lo, int64_t hi) {
      for (int64_t r = lo; r < hi; ++r) {
        if (ftip) mp_dyn_row_fk_jac_id<MP_BIG_DOF, T, true>(MB, C, q, qd, qdd, Tout, Jout, tau, (long)r);
        else mp_dyn_row_fk_jac_id<MP_BIG_DOF, T, false>(MB, C, q, qd, qdd, Tout, Jout, tau, (long)r);
      }
    });
    return MP_OK;
  }
  MP_CPU_DISPATCH(M.n, {
    parallel_for(rows, 256, nthreads, [&](int64_t lo, int64_t hi) {
      if (ftip) rows_fk_jac_id<T, N, true>(M, C, q, qd, qdd, Tout, Jout, tau, lo, hi);
      else rows_fk_jac_id<T, N, false>(M, C, q, qd, qdd, Tout, Jout, tau, lo, hi);
    });
  })
  return MP_OK;
}

// ---- mass matrix / forward dynamics per row
template <typename T, int N>
void rows_mass_matrix(const MpModel<T>& M, const T* q, T* out, int64_t lo, int64_t hi) {
  for (int64_t r = lo; r < hi; ++r) {
    T a[N];
    for (int j = 0; j < N; ++j) a[j] = q[r * N + j];
    MpJointState<T, N> js;
    mp_joint_state<T, N>(M, a, js);
    T Mq[N][N];
    mp_mass_matrix_crba<T, N>(M, js, Mq);
    MpBad<T> bad;
    bad.add(a);
    for (int i = 0; i < N; ++i) {
      mp_poison_if(bad.any(), Mq[i]);
      std::memcpy(out + (r * N + i) * N, Mq[i], sizeof Mq[i]);
    }
  }
}

template <typename T, int N, bool F>
void rows_forward_dynamics(const MpModel<T>& M, const MpCall<T>& C, const T* q, const T* qd, const T* tau, T* qdd, int64_t lo,
                           int64_t hi) {
  const T tn[3] = {C.F1n[0], C.F1n[1], C.F1n[2]}, tf[3] = {C.F1f[0], C.F1f[1], C.F1f[2]};
  for (int64_t r = lo; r < hi; ++r) {
    T a[N], b[N], t[N], o[N];
    for (int j = 0; j < N; ++j) { a[j] = q[r * N + j]; b[j] = qd[r * N + j]; t[j] = tau[r * N + j]; }
    mp_forward_dynamics<T, N, F>(M, C.a0, tn, tf, a, b, t, o);
    MpBad<T> bad;
    bad.add(a); bad.add(b); bad.add(t);
    mp_poison_if(bad.any(), o);
    std::memcpy(qdd + r * N, o, sizeof o);
  }
}

// ---- forward_dynamics_trajectory: the body of k_fd_traj for trajectories [lo, hi)
template <typename T, int N, bool F>
void rollouts(const MpModel<T>& M, const MpCall<T>& C, const T* theta0, const T* dtheta0, const T* taumat, const T* Ftipmat,
              int64_t Nt, T h, int intRes, float* pos, float* vel, float* acc, int64_t lo, int64_t hi) {
  const float nanf_ = __builtin_bit_cast(float, 0x7fc00000u);
  for (int64_t b = lo; b < hi; ++b) {
    T q[N], qd[N];
    for (int j = 0; j < N; ++j) { q[j] = theta0[b * N + j]; qd[j] = dtheta0[b * N + j]; }
    MpBad<T> bad;
    bad.add(q); bad.add(qd);
    for (int64_t i = 0; i < Nt; ++i) {
      T last[N];
      for (int j = 0; j < N; ++j) last[j] = T(0);
      if (i > 0) {
        T tau[N], tn[3] = {T(0), T(0), T(0)}, tf[3] = {T(0), T(0), T(0)};
        for (int j = 0; j < N; ++j) tau[j] = taumat[(b * Nt + i) * N + j];
        bad.add(tau);
        if (F) {
          T W[6];
          for (int k = 0; k < 6; ++k) W[k] = Ftipmat[(b * Nt + i) * 6 + k];
          bad.add(W);
          mp_wrench_to_frame1(M, W, tn, tf);
        }
        for (int k = 0; k < intRes; ++k) {
          mp_forward_dynamics<T, N, F>(M, C.a0, tn, tf, q, qd, tau, last);
          for (int j = 0; j < N; ++j) {
            qd[j] = qd[j] + last[j] * h;
            q[j] = mp_clip(q[j] + qd[j] * h, M.qmin[j], M.qmax[j]);
          }
        }
        bad.add(qd);
      }
      const bool poison = i > 0 && bad.any();
      for (int j = 0; j < N; ++j) {
        const int64_t o = (b * Nt + i) * N + j;
        pos[o] = poison ? nanf_ : (float)q[j];
        vel[o] = poison ? nanf_ : (float)qd[j];
        acc[o] = poison ? nanf_ : (float)last[j];
      }
    }
  }
}

template <typename T>
int fd_trajectory_cpu(const mp_model* model, const T* theta0, const T* dtheta0, const T* taumat, const T* Ftipmat, int64_t B,
                      int64_t Nt, const double* g, double dt, int intRes, float* pos, float* vel, float* acc, int nthreads) {
  if (!model) return fail("null model");
  if (B < 0 || Nt < 0) return fail("negative trajectory / step count");
  if (B == 0 || Nt == 0) return MP_OK;
  if (intRes < 1) return fail("intRes must be >= 1");
  if (!theta0 || !dtheta0 || !taumat || !pos || !vel || !acc) return fail("null pointer");
  const MpModel<T>& M = pick<T>(model);
  const MpCall<T> C = make_call<T>(model, g, nullptr);
  const T h = (T)(dt / intRes);
  if (model->big) {
    const MpBigModel<T>& MB = pick_big<T>(model);
    parallel_for(B, 1, nthreads, [&](int64_t lo, int64_t hi) {
      for (int64_t b = lo; b < hi; ++b) {
        if (Ftipmat) mp_dyn_rollout<MP_BIG_DOF, T, true>(MB, C, theta0, dtheta0, taumat, Ftipmat, (long)b, (long)B, (long)Nt, h, intRes, pos, vel, acc, false);
        else mp_dyn_rollout<MP_BIG_DOF, T, false>(MB, C, theta0, dtheta0, taumat, Ftipmat, (long)b, (long)B, (long)Nt, h, intRes, pos, vel, acc, false);
      }
    });
    return MP_OK;
  }
  MP_CPU_DISPATCH(M.n, {
    parallel_for(B, 1, nthreads, [&](int64_t lo, int64_t hi) {
      if (Ftipmat) rollouts<T, N, true>(M, C, theta0, dtheta0, taumat, Ftipmat, Nt, h, intRes, pos, vel, acc, lo, hi);
      else rollouts<T, N, false>(M, C, theta0, dtheta0, taumat, Ftipmat, Nt, h, intRes, pos, vel, acc, lo, hi);
    });
  })
  return MP_OK;
}
}  // namespace

extern "C" {

int mp_cpu_threads(int64_t items) { return thread_count(items, 1, 0); }

int mp_id_trajectory_cpu_f32(const mp_model* model, const float* q, const float* qd, const float* qdd, int64_t rows,
                             const double* g, const double* Ftip, float* tau, int nthreads) {
  if (rows > 0 && !tau) return fail("mp_id_trajectory_cpu_f32: null tau");
  return fk_jac_id_cpu<float>("mp_id_trajectory_cpu_f32: null pointer", model, q, qd, qdd, rows, g, Ftip, nullptr, nullptr, tau, nthreads);
}
int mp_id_trajectory_cpu_f64(const mp_model* model, const double* q, const double* qd, const double* qdd, int64_t rows,
                             const double* g, const double* Ftip, double* tau, int nthreads) {
  if (rows > 0 && !tau) return fail("mp_id_trajectory_cpu_f64: null tau");
  return fk_jac_id_cpu<double>("mp_id_trajectory_cpu_f64: null pointer", model, q, qd, qdd, rows, g, Ftip, nullptr, nullptr, tau, nthreads);
}
// which rows the float32 inverse-dynamics kernels evaluate in float64 (mp_core.h, mp_id_row_is_hard): 1 per such row
int mp_id_row_precision_cpu_f32(const mp_model* model, const float* q, const float* qd, const float* qdd, int64_t rows,
                                const double* g, const double* Ftip, uint8_t* in_f64, int nthreads) {
  if (!model) return fail("mp_id_row_precision_cpu_f32: null model");
  if (rows < 0) return fail("mp_id_row_precision_cpu_f32: negative row count");
  if (rows == 0) return MP_OK;
  if (!q || !qd || !qdd || !in_f64) return fail("mp_id_row_precision_cpu_f32: null pointer");
  const MpModel<float>& M = model->f;
  const MpCall<float> C = make_call<float>(model, g, Ftip);
  const bool ftip = any_nonzero(Ftip);
  if (model->big) {  // 9..32 joints: the same verdict from the looped recursion (mp_dyn.h)
    const MpBigModel<float>& MB = model->bf;
    const int n = MB.n;
    parallel_for(rows, 128, nthreads, [&](int64_t lo, int64_t hi) {
      for (int64_t r = lo; r < hi; ++r) {
        float t[MP_BIG_DOF], scale = 0.0f;
        MpDynState<float, MP_BIG_DOF> js;
        mp_dyn_joint_state<float>(MB, n, q + r * n, js);
        if (ftip) mp_dyn_rnea<float, true>(MB, n, C.a0, C.F1n, C.F1f, js, qd + r * n, qdd + r * n, t, &scale);
        else mp_dyn_rnea<float, false>(MB, n, C.a0, C.F1n, C.F1f, js, qd + r * n, qdd + r * n, t, &scale);
        in_f64[r] = mp_dyn_row_is_hard(t, n, scale) ? 1 : 0;
      }
    });
    return MP_OK;
  }
  MP_CPU_DISPATCH(M.n, {
    parallel_for(rows, 256, nthreads, [&](int64_t lo, int64_t hi) {
      const float tn[3] = {C.F1n[0], C.F1n[1], C.F1n[2]}, tf[3] = {C.F1f[0], C.F1f[1], C.F1f[2]};
      for (int64_t r = lo; r < hi; ++r) {
        float a[N], b[N], c[N], t[N];
        for (int j = 0; j < N; ++j) { a[j] = q[r * N + j]; b[j] = qd[r * N + j]; c[j] = qdd[r * N + j]; }
        MpJointState<float, N> js;
        mp_joint_state<float, N>(M, a, js);
        MpRowScale<float, N> sc;
        if (ftip) mp_rnea_impl<float, N, true>(M, C.a0, tn, tf, js, b, c, t, sc);
        else mp_rnea_impl<float, N, false>(M, C.a0, tn, tf, js, b, c, t, sc);
        in_f64[r] = mp_id_row_is_hard<N>(t, sc.scale(M.lscale)) ? 1 : 0;
      }
    });
  })
  return MP_OK;
}
int mp_fk_jac_id_cpu_f64(const mp_model* model, const double* q, const double* qd, const double* qdd, int64_t rows,
                         const double* g, const double* Ftip, double* T, double* J, double* tau, int nthreads) {
  return fk_jac_id_cpu<double>("mp_fk_jac_id_cpu_f64: null pointer / no output / tau without qd, qdd", model, q, qd, qdd, rows, g,
                               Ftip, T, J, tau, nthreads);
}

int mp_mass_matrix_cpu_f64(const mp_model* model, const double* q, int64_t rows, double* Mout, int nthreads) {
  if (!model) return fail("mp_mass_matrix_cpu_f64: null model");
  if (rows < 0) return fail("mp_mass_matrix_cpu_f64: negative row count");
  if (rows == 0) return MP_OK;
  if (!q || !Mout) return fail("mp_mass_matrix_cpu_f64: null pointer");
  const MpModel<double>& M = model->d;
  if (model->big) {
    parallel_for(rows, 128, nthreads, [&](int64_t lo, int64_t hi) {
      for (int64_t r = lo; r < hi; ++r) mp_dyn_row_mass_matrix<MP_BIG_DOF, double>(model->bd, q, Mout, (long)r);
    });
    return MP_OK;
  }
  MP_CPU_DISPATCH(M.n, { parallel_for(rows, 256, nthreads, [&](int64_t lo, int64_t hi) { rows_mass_matrix<double, N>(M, q, Mout, lo, hi); }); })
  return MP_OK;
}

int mp_forward_dynamics_cpu_f64(const mp_model* model, const double* q, const double* qd, const double* tau, int64_t rows,
                                const double* g, const double* Ftip, double* qdd, int nthreads) {
  if (!model) return fail("mp_forward_dynamics_cpu_f64: null model");
  if (rows < 0) return fail("mp_forward_dynamics_cpu_f64: negative row count");
  if (rows == 0) return MP_OK;
  if (!q || !qd || !tau || !qdd) return fail("mp_forward_dynamics_cpu_f64: null pointer");
  const MpModel<double>& M = model->d;
  const MpCall<double> C = make_call<double>(model, g, Ftip);
  const bool ftip = any_nonzero(Ftip);
  if (model->big) {
    parallel_for(rows, 64, nthreads, [&](int64_t lo, int64_t hi) {
      for (int64_t r = lo; r < hi; ++r) {
        if (ftip) mp_dyn_row_forward_dynamics<MP_BIG_DOF, double, true>(model->bd, C, q, qd, tau, qdd, (long)r);
        else mp_dyn_row_forward_dynamics<MP_BIG_DOF, double, false>(model->bd, C, q, qd, tau, qdd, (long)r);
      }
    });
    return MP_OK;
  }
  MP_CPU_DISPATCH(M.n, {
    parallel_for(rows, 128, nthreads, [&](int64_t lo, int64_t hi) {
      if (ftip) rows_forward_dynamics<double, N, true>(M, C, q, qd, tau, qdd, lo, hi);
      else rows_forward_dynamics<double, N, false>(M, C, q, qd, tau, qdd, lo, hi);
    });
  })
  return MP_OK;
}

int mp_fd_trajectory_cpu_f32(const mp_model* model, const float* theta0, const float* dtheta0, const float* taumat,
                             const float* Ftipmat, int64_t B, int64_t N, const double* g, double dt, int intRes, float* pos,
                             float* vel, float* acc, int nthreads) {
  return fd_trajectory_cpu<float>(model, theta0, dtheta0, taumat, Ftipmat, B, N, g, dt, intRes, pos, vel, acc, nthreads);
}
int mp_fd_trajectory_cpu_f64(const mp_model* model, const double* theta0, const double* dtheta0, const double* taumat,
                             const double* Ftipmat, int64_t B, int64_t N, const double* g, double dt, int intRes, float* pos,
                             float* vel, float* acc, int nthreads) {
  return fd_trajectory_cpu<double>(model, theta0, dtheta0, taumat, Ftipmat, B, N, g, dt, intRes, pos, vel, acc, nthreads);
}

int mp_inverse_kinematics_cpu_f64(const mp_model* model, const double* T_desired, const double* theta0, int64_t B,
                                  const double* joint_limits, double eomg, double ev, int max_iterations, double damping,
                                  double step_cap, double weight_orientation, double weight_position, int adaptive_tuning,
                                  int backtracking, uint32_t seed, double* theta, int32_t* success, int32_t* iterations,
                                  int32_t* restarts, int nthreads) {
  if (!model) return fail("mp_inverse_kinematics_cpu_f64: null model");
  if (B < 0) return fail("mp_inverse_kinematics_cpu_f64: negative problem count");
  if (B == 0) return MP_OK;
  if (!T_desired || !theta0 || !theta || !success || !iterations || !restarts) return fail("mp_inverse_kinematics_cpu_f64: null pointer");
  if (max_iterations < 1) return fail("mp_inverse_kinematics_cpu_f64: max_iterations must be at least 1");
  if (!(eomg > 0 && ev > 0 && damping >= 0 && step_cap > 0))
    return fail("mp_inverse_kinematics_cpu_f64: eomg, ev, step_cap must be positive and damping non-negative");
  auto fill = [&](auto& P, int cap) -> bool {
    P.eomg = eomg; P.ev = ev; P.damping = damping; P.step_cap = step_cap; P.w_o = weight_orientation; P.w_p = weight_position;
    P.max_iterations = max_iterations; P.seed = seed;
    P.adaptive_tuning = adaptive_tuning ? 1 : 0; P.backtracking = backtracking ? 1 : 0;
    for (int j = 0; j < cap; ++j) {
      P.lo[j] = (j < model->d.n && joint_limits) ? joint_limits[2 * j] : -HUGE_VAL;
      P.hi[j] = (j < model->d.n && joint_limits) ? joint_limits[2 * j + 1] : HUGE_VAL;
      if (P.lo[j] > P.hi[j]) return false;
    }
    return true;
  };
  if (model->big) {  // 9..32 joints: the body of k_dyn_ik (run-time-n kinematics of csrc/mp_dyn.h under the same iteration)
    MpIkBigParams PB;
    if (!fill(PB, MP_BIG_DOF)) return fail("mp_inverse_kinematics_cpu_f64: a joint has its lower limit above its upper limit");
    const MpBigModel<double>& MB = model->bd;
    const int n = MB.n;
    auto rows_of = [&](auto cap_tag) {   // per-problem arrays of 16 entries for 9..16 joints, 32 beyond (as the kernels: MP_DISPATCH_CAP)
      constexpr int CAP = decltype(cap_tag)::value;
      parallel_for(B, 1, nthreads, [&](int64_t lo, int64_t hi) {
        for (int64_t row = lo; row < hi; ++row) {
          MpIkState<CAP> S;
          for (int j = 0; j < CAP; ++j) S.theta[j] = j < n ? theta0[row * n + j] : 0.0;
          mp_ik_begin(S, PB);
          int done = 0;
          while (!(done = mp_ik_iterate<CAP, MpIkLooped<CAP>>(MB, PB, S, T_desired + row * 16, theta0 + row * n))) {}
          for (int j = 0; j < n; ++j) theta[row * n + j] = S.theta[j];
          success[row] = done == 2 ? 1 : 0;
          iterations[row] = S.k + 1;
          restarts[row] = S.restarts;
        }
      });
    };
    if (n <= MP_MID_DOF) rows_of(std::integral_constant<int, MP_MID_DOF>{});
    else rows_of(std::integral_constant<int, MP_BIG_DOF>{});
    return MP_OK;
  }
  MpIkParams P;
  if (!fill(P, MP_MAX_DOF)) return fail("mp_inverse_kinematics_cpu_f64: a joint has its lower limit above its upper limit");
  const MpModel<double>& M = model->d;
  // the body of k_ik (csrc/mp_kernels.hip) per problem: begin, iterate until the iteration reports done
  MP_CPU_DISPATCH(M.n, {
    parallel_for(B, 1, nthreads, [&](int64_t lo, int64_t hi) {
      for (int64_t row = lo; row < hi; ++row) {
        MpIkState<N> S;
        for (int j = 0; j < N; ++j) S.theta[j] = theta0[row * N + j];
        mp_ik_begin(S, P);
        int done = 0;
        while (!(done = mp_ik_iterate<N>(M, P, S, T_desired + row * 16, theta0 + row * N))) {}
        for (int j = 0; j < N; ++j) theta[row * N + j] = S.theta[j];
        success[row] = done == 2 ? 1 : 0;
        iterations[row] = S.k + 1;
        restarts[row] = S.restarts;
      }
    });
  })
  return MP_OK;
}

int mp_pd_regulation_cpu_f64(const mp_model* model, const double* theta0, const double* theta_des, const double* Kp, const double* Kd,
                             int64_t K, const double* g, double dt, int steps, double* errors, int32_t* count, int nthreads) {
  if (!model) return fail("mp_pd_regulation_cpu_f64: null model");
  if (K < 0 || steps < 0) return fail("mp_pd_regulation_cpu_f64: negative run or step count");
  if (K == 0) return MP_OK;
  if (!theta0 || !theta_des || !Kp || !Kd || !count || (!errors && steps > 0)) return fail("mp_pd_regulation_cpu_f64: null pointer");
  const MpCall<double> C = make_call<double>(model, g, nullptr);
  const int n = model->d.n;
  if (model->big) {  // the body of k_dyn_pd_regulation
    parallel_for(K, 1, nthreads, [&](int64_t lo, int64_t hi) {
      for (int64_t k = lo; k < hi; ++k)
        count[k] = mp_dyn_pd_regulation_run<MP_BIG_DOF, double>(model->bd, C.a0, theta0 + k * n, theta_des + k * n, Kp[k], Kd[k], dt, steps,
                                                    errors + k * steps);
    });
    return MP_OK;
  }
  const MpModel<double>& M = model->d;
  MP_CPU_DISPATCH(M.n, {  // the body of k_pd_regulation
    parallel_for(K, 1, nthreads, [&](int64_t lo, int64_t hi) {
      for (int64_t k = lo; k < hi; ++k) {
        double a[N], d[N];
        for (int j = 0; j < N; ++j) { a[j] = theta0[k * N + j]; d[j] = theta_des[k * N + j]; }
        count[k] = mp_pd_regulation_run<double, N>(M, C.a0, a, d, Kp[k], Kd[k], dt, steps, errors + k * steps);
      }
    });
  })
  return MP_OK;
}

int mp_cartesian_trajectory_cpu_f32(const double* Xstart, const double* Xend, int64_t B, int64_t N, double Tf, int method,
                                    float* pos, float* vel, float* acc, float* orient, int nthreads) {
  if (B < 0 || N < 0) return fail("mp_cartesian_trajectory_cpu_f32: negative count");
  if (B == 0 || N == 0) return MP_OK;
  if (N < 2) return fail("mp_cartesian_trajectory_cpu_f32: N must be >= 2");
  if (!Xstart || !Xend || !pos || !vel || !acc || !orient) return fail("mp_cartesian_trajectory_cpu_f32: null pointer");
  parallel_for(B * N, 512, nthreads, [&](int64_t lo, int64_t hi) {
    for (int64_t r = lo; r < hi; ++r) {
      const int64_t b = r / N, i = r - b * N;
      double A[16], E[16];
      std::memcpy(A, Xstart + b * 16, sizeof A);
      std::memcpy(E, Xend + b * 16, sizeof E);
      float p[3], v[3], a[3], o[9];
      mp_cartesian_point(A, E, (long)i, (long)N, Tf, method, p, v, a, o);
      std::memcpy(pos + r * 3, p, sizeof p); std::memcpy(vel + r * 3, v, sizeof v); std::memcpy(acc + r * 3, a, sizeof a);
      std::memcpy(orient + r * 9, o, sizeof o);
    }
  });
  return MP_OK;
}

}  // extern "C"
