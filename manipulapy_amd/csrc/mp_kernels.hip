// HIP kernels for gfx950 (MI355X): one thread per (trajectory, timestep) row.
//
// Layout in HBM: exactly the reference's API arrays — q / qd / qdd / tau are (rows, n) row-major
// ("array of rows"), T is (rows, 4, 4), J is (rows, 6, n), start/end are (B, n).  A wavefront's 64
// rows are one contiguous 64*n*sizeof(T) span per array, read/written with the widest vector access
// the row size allows (16 / 8 / 4 bytes per lane).  The robot model and the per-call constants are
// kernel ARGUMENTS (kernarg segment -> scalar loads -> SGPR operands): zero per-thread traffic.
// No MFMA: there is no dense contraction on this path (BASELINE.json north_star).
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "mp_bodies.h"
#include "mp_dyn.h"
#include "mp_ik.h"
#include "mp_kernels.h"

namespace {

constexpr int kBlock = 256;
constexpr int kPkMinWaves = 2;  // waves per SIMD asked for by the packed (two rows per lane) fused kernel: register cap = 512 / 2

// ------------------------------------------------------------------------------------- probe
__global__ void k_selftest(int* out) { out[threadIdx.x] = (int)threadIdx.x; }

// ------------------------------------------------------------------------- inverse dynamics
template <typename T, int N, bool HAS_FTIP>
__global__ __launch_bounds__(kBlock) void k_id(const MpModel<T> M, const MpCall<T> C, const T* __restrict__ q,
                                               const T* __restrict__ qd, const T* __restrict__ qdd,
                                               T* __restrict__ tau, long rows) {
  MP_COLD_BUFFER(N, kBlock, sizeof(T));
  const long r = (long)blockIdx.x * kBlock + threadIdx.x;
  if (r >= rows) return;
  mp_body_id<T, N, HAS_FTIP>(M, C, q, qd, qdd, tau, r, MP_COLD_PTR);
}

// The same with the model read through a pointer to device memory (scalar loads, K$-resident) instead of the kernel-argument
// struct: the joints' constants are then loaded joint by joint where the recursion uses them (mp_joint_of, mp_core.h), not all
// 185 dwords at the top of the kernel - which overflows the SGPR file into VGPR lanes (60 v_writelane / v_readlane of 1203
// instructions at n = 6) and keeps the wave count down.
// (five waves per SIMD asked for without a tip wrench, four with one: the float64 re-evaluation loop behind the float32 pass raised
// the unconstrained allocation from 85 to 106 VGPRs; held to 96 / 128 neither pass touches scratch)
// Round 5: the kernel's first L.blocks workgroups carry the float64 pass of an earlier launch of the same model (mp_body_id_lead, as
// the robot-specialised kernels do); the float64 path spills under this kernel's register cap - in those workgroups only.
// Round 6: ALLREV = the model holds revolute joints only (the launcher looks): the float32 rows read it as an MpModelRev, whose
// `rev` the recursion folds to 1 (csrc/mp_model.h) - 13 VALU instructions per joint fewer, no branch.
template <typename T, int N, bool HAS_FTIP, bool ALLREV = false>
__global__ __launch_bounds__(kBlock, HAS_FTIP ? 4 : 5) void k_id_dm(const MpModel<T>* __restrict__ Mdev, const MpCall<T> C, const T* __restrict__ q,
                                                  const T* __restrict__ qd, const T* __restrict__ qdd, T* __restrict__ tau, long rows,
                                                  const MpLead L) {
  MP_COLD_BUFFER(N, kBlock, sizeof(T));
  typedef const __attribute__((address_space(4))) MpModel<T> MC;
#if defined(__HIP_DEVICE_COMPILE__)
  if constexpr (sizeof(T) == 4) {
    if (blockIdx.x < L.blocks) {
      mp_body_id_lead<N, HAS_FTIP>(*(MpModelConstD*)L.C.cold_model, *(MC*)Mdev, L);
      return;
    }
  }
#endif
  const long r = (long)(blockIdx.x - L.blocks) * kBlock + threadIdx.x;
  if (r >= rows) return;
#if defined(__HIP_DEVICE_COMPILE__)
  if constexpr (ALLREV && sizeof(T) == 4) {
    mp_body_id<T, N, HAS_FTIP>(*(MpModelRevConstF*)Mdev, C, q, qd, qdd, tau, r, MP_COLD_PTR);
    return;
  }
#endif
  mp_body_id<T, N, HAS_FTIP>(*(MC*)Mdev, C, q, qd, qdd, tau, r, MP_COLD_PTR);
}

// The float64 pass over the rows k_id_dm handed over (mp_body_id_hard, csrc/mp_bodies.h): both models through device pointers.
template <int N, bool HAS_FTIP>
__global__ __launch_bounds__(64) void k_id_hard(const MpModel<float>* __restrict__ Mdev, const MpCall<float> C, const float* __restrict__ q,
                                                const float* __restrict__ qd, const float* __restrict__ qdd, float* __restrict__ tau, unsigned rows) {
#if defined(__HIP_DEVICE_COMPILE__)
  mp_body_id_hard<N, HAS_FTIP>(*(MpModelConstD*)C.cold_model, *(MpModelConstF*)Mdev, C,
                               [&](long r, float (&x)[N], float (&y)[N], float (&z)[N]) {
                                 RunIO<float, N>::load(q, r, x); RunIO<float, N>::load(qd, r, y); RunIO<float, N>::load(qdd, r, z);
                               }, tau, rows);
#endif
}

// ... up to four launches' lists in one kernel (blockIdx.y picks the launch), as the specialised programs' pass: a pass costs ~5 us of
// launch and latency however few rows it holds (round 5: the generic passes ran one kernel each until then)
template <int N, bool HAS_FTIP>
__global__ __launch_bounds__(64) void k_id_hard_batch(const MpModel<float>* __restrict__ Mdev, const MpHardBatch B) {
#if defined(__HIP_DEVICE_COMPILE__)
  const int e = blockIdx.y;
  const float* __restrict__ q = B.q[e]; const float* __restrict__ qd = B.qd[e]; const float* __restrict__ qdd = B.qdd[e];
  mp_body_id_hard<N, HAS_FTIP>(*(MpModelConstD*)B.C[e].cold_model, *(MpModelConstF*)Mdev, B.C[e],
                               [&](long r, float (&x)[N], float (&y)[N], float (&z)[N]) {
                                 RunIO<float, N>::load(q, r, x); RunIO<float, N>::load(qd, r, y); RunIO<float, N>::load(qdd, r, z);
                               }, B.tau[e], B.rows[e]);
#endif
}

// the same pass for rows the fused generic kernel handed over: a row's inputs are generated again from start / end / the time table
template <int N, bool HAS_FTIP>
__global__ __launch_bounds__(64) void k_traj_id_hard(const MpModel<float>* __restrict__ Mdev, const MpCall<float> C, const float* __restrict__ start,
                                                     const float* __restrict__ end, unsigned Nt, const double* __restrict__ tab,
                                                     float* __restrict__ tau, unsigned rows) {
#if defined(__HIP_DEVICE_COMPILE__)
  MpModelConstF& M = *(MpModelConstF*)Mdev;
  mp_body_id_hard<N, HAS_FTIP>(*(MpModelConstD*)C.cold_model, M, C,
                               [&](long r, float (&x)[N], float (&y)[N], float (&z)[N]) {
                                 const unsigned b = (unsigned)r / Nt, t = (unsigned)r - b * Nt;
                                 float a[N], e[N];
                                 RunIO<float, N>::load(start, (long)b, a);
                                 RunIO<float, N>::load(end, (long)b, e);
                                 const double u0 = tab[3 * t], u1 = tab[3 * t + 1], u2 = tab[3 * t + 2];
#pragma unroll
                                 for (int j = 0; j < N; ++j) {
                                   const double d = (double)(e[j] - a[j]);
                                   x[j] = mp_clip((float)(u0 * d + (double)a[j]), M.qmin[j], M.qmax[j]);
                                   y[j] = (float)(u1 * d);
                                   z[j] = (float)(u2 * d);
                                 }
                               }, tau, rows);
#endif
}

// -------------------------------------------------------------- trajectory generation pieces
template <int N>
__global__ __launch_bounds__(kBlock) void k_batch_traj(const MpModel<float> M, const float* __restrict__ start,
                                                       const float* __restrict__ end, long B, long Nt, double Tf,
                                                       int method, float* __restrict__ pos, float* __restrict__ vel,
                                                       float* __restrict__ acc) {
  // A pure write kernel: a full wave's 64 consecutive rows of each output leave as whole lines, non-temporal, through LDS
  // (MpRowStage, csrc/mp_bodies.h); the last, partial wave stores per lane.  (Non-temporal PER-LANE stores of these 24-byte rows
  // halve the rate: 0.053 -> 0.096 ms.)
  using ST = MpRowStage<float, N>;
  __shared__ __attribute__((aligned(16))) char lds[kBlock / 64][3 * ST::SPAN];
  const long total = B * Nt;
  const long r = (long)blockIdx.x * kBlock + threadIdx.x;
  const int lane = (int)(threadIdx.x & 63);
  const long row0 = mp_wave_uniform(r - lane);   // (every lane is still active here: the flush addresses form on the scalar unit)
  if (row0 >= total) return;
  const bool full = row0 + 64 <= total;  // wave-uniform
  if (r >= total) return;                // (only in the partial wave)
  const long b = r / Nt, t = r - b * Nt;
  float p[N], v[N], a[N];
  traj_row<N>(M, start, end, b, t, Nt, Tf, method, p, v, a);
  if (full) {
    char* w = lds[threadIdx.x >> 6];
    ST::row_out(w, lane, p);
    ST::row_out(w + ST::SPAN, lane, v);
    ST::row_out(w + 2 * ST::SPAN, lane, a);
    ST::sync();
    ST::flush(pos, row0, lane, w);
    ST::flush(vel, row0, lane, w + ST::SPAN);
    ST::flush(acc, row0, lane, w + 2 * ST::SPAN);
  } else {
    RunIO<float, N>::store(pos, r, p);
    RunIO<float, N>::store(vel, r, v);
    RunIO<float, N>::store(acc, r, a);
  }
}

// (Generic float32 forms removed in round 6, each slower than what is left - the one-row kernel k_id_dm for given rows, the table-driven
// packed kernel k_traj_id_pk_tab for generated ones: k_id<float> / k_id_pk with the model in the kernel arguments (c2 0.126 / 0.119 ms
// against 0.100), k_traj_id / k_traj_id_pk with the time scaling per row (0.0577 against 0.0525 specialised); profiles/HISTORY.md.)

// per-call table of the time scaling, three doubles per timestep: exactly traj_row's arithmetic, once per timestep
// instead of once per row
__global__ __launch_bounds__(kBlock) void k_time_table(double* __restrict__ tab, long Nt, double Tf, int method) {
  const long t = (long)blockIdx.x * kBlock + threadIdx.x;
  if (t >= Nt) return;
  const double tt = (double)t * (Tf / (double)(Nt - 1));
  double s, sd, sdd;
  mp_time_scaling(method, tt / Tf, Tf, s, sd, sdd);
  tab[3 * t] = s; tab[3 * t + 1] = sd; tab[3 * t + 2] = sdd;
}

template <int N, bool HAS_FTIP>
__global__ __launch_bounds__(kBlock, kPkMinWaves) void k_traj_id_pk_tab(const MpModel<float> M, const MpCall<float> C,
                                                              const float* __restrict__ start, const float* __restrict__ end,
                                                              long Nt, unsigned bpt, const double* __restrict__ tab,
                                                              float* __restrict__ tau) {
  MP_COLD_BUFFER(N, kBlock, 4);
  long b, t0, t1;
  bool valid1;
  if (!mp_traj_pair(blockIdx.x, threadIdx.x, kBlock, bpt, Nt, b, t0, t1, valid1)) return;
  mp_body_traj_id_pk_tab<N, HAS_FTIP>(M, C, start, end, b, t0, t1, valid1, Nt, tab, tau, MP_COLD_PTR);
}

// ------------------------------------------------------------- FK + space Jacobian + ID fused
// one wave per block: the per-wave LDS staging slice (9 KiB) then never limits residency (a 256-thread block needs
// 36 KiB, i.e. at most 4 blocks = 16 waves per CU, and blocks drain unevenly: 1.7 waves per SIMD measured)
constexpr int kFkBlock = 64;

template <typename T, int N, bool HAS_FTIP>
__global__ __launch_bounds__(kFkBlock) void k_fk_jac_id(const MpModel<T> M, const MpCall<T> C, const T* __restrict__ q,
                                                      const T* __restrict__ qd, const T* __restrict__ qdd,
                                                      T* __restrict__ Tout, T* __restrict__ Jout,
                                                      T* __restrict__ tau, long rows) {
  __shared__ __attribute__((aligned(16))) char lds[(kFkBlock / 64) * MP_WAVE_LDS_BYTES];  // one staging slice per wave
  const long r = (long)blockIdx.x * kFkBlock + threadIdx.x;
  mp_body_fk_jac_id<T, N, HAS_FTIP>(M, C, q, qd, qdd, Tout, Jout, tau, r, rows, lds + (threadIdx.x >> 6) * MP_WAVE_LDS_BYTES);
}

// ------------------------------------------------------------- mass matrix / forward dynamics
// one wave per block; the N x N rows leave through the wave-cooperative coalesced store (see mp_bodies.h)
template <typename T, int N>
__global__ __launch_bounds__(kFkBlock) void k_mass_matrix(const MpModel<T> M, const T* __restrict__ q, T* __restrict__ Mout,
                                                          long rows) {
  __shared__ __attribute__((aligned(16))) char lds[MP_WAVE_LDS_BYTES];
  const int lane = (int)threadIdx.x;
  const long row0 = (long)blockIdx.x * kFkBlock;
  if (row0 >= rows) return;
  const long left = rows - row0;
  const int nvalid = left < 64 ? (int)left : 64;
  const long r = lane < nvalid ? row0 + lane : rows - 1;  // out-of-range lanes recompute the last row, store nothing
  T a[N];
  RunIO<T, N>::load(q, r, a);
  MpJointState<T, N> js;
  mp_joint_state<T, N>(M, a, js);
  T Mq[N][N];
  mp_mass_matrix_crba<T, N>(M, js, Mq);
  T flat[N * N];
#pragma unroll
  for (int i = 0; i < N; ++i)
#pragma unroll
    for (int j = 0; j < N; ++j) flat[i * N + j] = Mq[i][j];
  MpBad<T> bad;
  bad.add(a);
  mp_poison_if(bad.any(), flat);
  mp_wave_store_auto<T, N * N>(Mout, row0, lane, nvalid, flat, lds);
}

// qdd = forward_dynamics(q, qd, tau, g, Ftip) per row; Ftip is one wrench for every row (per-call constant)
template <typename T, int N, bool HAS_FTIP>
__global__ __launch_bounds__(kBlock) void k_forward_dynamics(const MpModel<T> M, const MpCall<T> C, const T* __restrict__ q,
                                                             const T* __restrict__ qd, const T* __restrict__ tau,
                                                             T* __restrict__ qdd, long rows) {
  const long r = (long)blockIdx.x * kBlock + threadIdx.x;
  if (r >= rows) return;
  mp_body_fd<T, N, HAS_FTIP>(M, C, q, qd, tau, qdd, r);
}

// one wave per block: the roll-out's LDS tile is per wave and nothing is shared between waves
constexpr int kFdBlock = 64;
template <typename T, int N, bool HAS_FTIP>
__global__ __launch_bounds__(kFdBlock, (sizeof(T) == 4 ? 2 : 1)) void k_fd_traj(const MpModel<T> M, const MpCall<T> C, const T* __restrict__ theta0,
                                                      const T* __restrict__ dtheta0, const T* __restrict__ taumat,
                                                      const T* __restrict__ Ftipmat, long B, long Nt, T h, int intRes,
                                                      float* __restrict__ pos, float* __restrict__ vel, float* __restrict__ acc) {
  __shared__ unsigned lds[MpFdTile<T, N, HAS_FTIP>::DWORDS];
  const long b = (long)blockIdx.x * kFdBlock + threadIdx.x;  // lanes past the batch stay (wave-cooperative stores)
  mp_body_fd_traj<T, N, HAS_FTIP>(M, C, theta0, dtheta0, taumat, Ftipmat, b, B, Nt, h, intRes, pos, vel, acc, lds, (int)threadIdx.x);
}

// the roll-out on the time-major device layout (mp_body_fd_traj_tm): no LDS, 64-thread blocks so that the 2048 waves of a
// 131072-trajectory shard spread evenly over the 1024 SIMDs
template <typename T, int N, bool HAS_FTIP>
__global__ __launch_bounds__(kFdBlock, 2) void k_fd_traj_tm(const MpModel<T> M, const MpCall<T> C, const T* __restrict__ theta0,
                                                            const T* __restrict__ dtheta0, const T* __restrict__ taumat,
                                                            const T* __restrict__ Ftipmat, long B, long Nt, T h, int intRes,
                                                            float* __restrict__ pos, float* __restrict__ vel, float* __restrict__ acc) {
  const long b0 = (long)blockIdx.x * kFdBlock;  // one wave per block: its first trajectory is wave-uniform
  if (b0 + threadIdx.x >= B) return;
  mp_body_fd_traj_tm<T, N, HAS_FTIP>(M, C, theta0, dtheta0, taumat, Ftipmat, b0, (int)threadIdx.x, B, Nt, h, intRes, pos, vel, acc);
}

// (outer, inner, W dwords) -> (inner, outer, W dwords), see mp_body_transpose_rows
__global__ __launch_bounds__(kBlock) void k_transpose_rows(const unsigned* __restrict__ src, unsigned* __restrict__ dst, long outer,
                                                           long inner, int W) {
  extern __shared__ unsigned tr_lds[];
  const int TI = mp_tr_ti(W);
  const unsigned gx = (unsigned)((inner + TI - 1) / TI);  // tiles along `inner`; the grid is one-dimensional
  const unsigned ty = blockIdx.x / gx, tx = blockIdx.x - ty * gx;
  mp_body_transpose_rows(src, dst, outer, inner, W, (long)ty * MP_TR_TO, (long)tx * TI, tr_lds, (int)threadIdx.x, kBlock);
}

// ------------------------------------------------------------------- Cartesian straight-line path
// one lane per (pose pair b, timestep i); outputs float32 (B,N,3) x3 and (B,N,3,3).
// `bpt` blocks of 256 timesteps per pose pair: the pair's rotation logarithm (acos, the half-turn branches) is worked
// out once per block by lane 0 and broadcast through LDS together with the start rotation and the end points, so a lane
// neither reloads two 4x4 poses nor repeats the logarithm, and there is no row -> (pair, timestep) division
__global__ __launch_bounds__(kBlock) void k_cartesian_traj(const double* __restrict__ Xstart, const double* __restrict__ Xend,
                                                           long Nt, unsigned bpt, double Tf, int method, float* __restrict__ pos,
                                                           float* __restrict__ vel, float* __restrict__ acc,
                                                           float* __restrict__ ori) {
  __shared__ double sh[18];  // w (3), Rs (9), ps (3), pe (3)
  const unsigned bb = blockIdx.x / bpt;
  const long b = bb, i = (long)(blockIdx.x - bb * bpt) * kBlock + threadIdx.x;
  if (threadIdx.x == 0) {
    double Xs[16], Xe[16], w[3];
    RunIO<double, 16>::load(Xstart, b, Xs);
    RunIO<double, 16>::load(Xend, b, Xe);
    mp_cartesian_prepare(Xs, Xe, w);
    sh[0] = w[0]; sh[1] = w[1]; sh[2] = w[2];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
#pragma unroll
      for (int c = 0; c < 3; ++c) sh[3 + 3 * r + c] = Xs[4 * r + c];
      sh[12 + r] = Xs[4 * r + 3];
      sh[15 + r] = Xe[4 * r + 3];
    }
  }
  __syncthreads();
  if (i >= Nt) return;
  double w[3], Rs[9], ps[3], pe[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) { w[k] = sh[k]; ps[k] = sh[12 + k]; pe[k] = sh[15 + k]; }
#pragma unroll
  for (int k = 0; k < 9; ++k) Rs[k] = sh[3 + k];
  float p[3], v[3], a[3], o[9];
  mp_cartesian_eval(Rs, ps, pe, w, i, Nt, Tf, method, p, v, a, o);
  const long r = b * Nt + i;
#pragma unroll
  for (int k = 0; k < 3; ++k) { pos[r * 3 + k] = p[k]; vel[r * 3 + k] = v[k]; acc[r * 3 + k] = a[k]; }
#pragma unroll
  for (int k = 0; k < 9; ++k) ori[r * 9 + k] = o[k];
}

// ------------------------------------------------------------------- fused potential field
// Per point: U = 1/2 |p - goal|^2 + sum_obs 1/2 (1/d - 1/d0)^2 over obstacles with 0 < d < d0, and its gradient
// (reference cuda_kernels/field_kernels.py:20-104 / :113-161, float32).  One lane per point; the obstacle index is
// wave-uniform, so obstacle coordinates arrive through scalar loads.
__global__ __launch_bounds__(kBlock) void k_potential_field(const float* __restrict__ pos, float gx, float gy, float gz,
                                                            const float* __restrict__ obs, long P, long O, float inv_d0,
                                                            float d0sq, float* __restrict__ pot, float* __restrict__ grad) {
  const long i = (long)blockIdx.x * kBlock + threadIdx.x;
  if (i >= P) return;
  const float px = pos[3 * i], py = pos[3 * i + 1], pz = pos[3 * i + 2];
  float dx = px - gx, dy = py - gy, dz = pz - gz;
  float U = 0.5f * (dx * dx + dy * dy + dz * dz);
  float Gx = dx, Gy = dy, Gz = dz;
  for (long o = 0; o < O; ++o) {
    const float ox = px - obs[3 * o], oy = py - obs[3 * o + 1], oz = pz - obs[3 * o + 2];
    const float d2 = ox * ox + oy * oy + oz * oz;
    if (d2 > 0.0f && d2 < d0sq) {
      const float inv = __builtin_amdgcn_rsqf(d2);  // v_rsq_f32, 1 ulp: the divide-after-sqrt sequence it replaces is ~15 instructions of ~40
      const float t = inv - inv_d0;
      U += 0.5f * t * t;
      const float f = -t * inv * inv * inv;
      Gx += f * ox; Gy += f * oy; Gz += f * oz;
    }
  }
  pot[i] = U;
  grad[3 * i] = Gx; grad[3 * i + 1] = Gy; grad[3 * i + 2] = Gz;
}

// ------------------------------------------------------------------- batched inverse kinematics
// Work queue: a lane that finishes its pose target takes the next unsolved one from a global counter instead of idling
// until the slowest problem of its wave is done (iteration counts vary from a handful to max_iterations; with one
// fixed problem per lane nearly every wave contains a straggler).  Every trip of the loop either fetches a problem or
// advances one by a single iteration, so the lanes of a wave keep executing the same code on different problems.
// Exit: the counter passes B for every lane eventually (it only grows), so every wave drains.  float64 throughout: the
// reference's default tolerances are 1e-6 rad / 1e-6 m.
template <int N>
__global__ __launch_bounds__(kBlock) void k_ik(const MpModel<double> M, const MpIkParams P, const double* __restrict__ Tdes,
                                               const double* __restrict__ theta0, long B, double* __restrict__ theta,
                                               int* __restrict__ success, int* __restrict__ iterations,
                                               int* __restrict__ restarts, unsigned long long* __restrict__ next) {
  MpIkState<N> S;
  bool have = false;
  long row = 0;
  for (;;) {
    if (!have) {
      row = (long)atomicAdd(next, 1ull);
      if (row >= B) break;
      RunIO<double, N>::load(theta0, row, S.theta);
      mp_ik_begin(S, P);
      have = true;
    }
    if (const int done = mp_ik_iterate<N>(M, P, S, Tdes + row * 16, theta0 + row * N)) {
      RunIO<double, N>::store(theta, row, S.theta);
      success[row] = done == 2 ? 1 : 0;
      iterations[row] = S.k + 1;
      restarts[row] = S.restarts;
      have = false;
    }
  }
}

// ------------------------------------------------------------------- 9..32 joints: run-time-n kernels (csrc/mp_dyn.h)
// One lane per row (or per trajectory), the model read through a pointer to device memory, per-joint state in indexed
// arrays.  Plain per-lane accesses: these kernels exist so that every robot the reference can evaluate computes here too.
// CAP = the capacity of those arrays (MP_MID_DOF or MP_BIG_DOF; the launchers pick it from the joint count).
template <typename T> using MpBigConst = const __attribute__((address_space(4))) MpBigModel<T>;

template <int CAP, typename T, bool HAS_FTIP>
__global__ __launch_bounds__(kBlock) void k_dyn_fk_jac_id(const MpBigModel<T>* __restrict__ Mdev, const MpCall<T> C,
                                                          const T* __restrict__ q, const T* __restrict__ qd, const T* __restrict__ qdd,
                                                          T* __restrict__ Tout, T* __restrict__ Jout, T* __restrict__ tau, long rows) {
  const long r = (long)blockIdx.x * kBlock + threadIdx.x;
  if (r >= rows) return;
  mp_dyn_row_fk_jac_id<CAP, T, HAS_FTIP>(*(MpBigConst<T>*)Mdev, C, q, qd, qdd, Tout, Jout, tau, r);
}
template <int CAP, typename T>
__global__ __launch_bounds__(kBlock) void k_dyn_mass_matrix(const MpBigModel<T>* __restrict__ Mdev, const T* __restrict__ q,
                                                            T* __restrict__ Mout, long rows) {
  const long r = (long)blockIdx.x * kBlock + threadIdx.x;
  if (r >= rows) return;
  mp_dyn_row_mass_matrix<CAP, T>(*(MpBigConst<T>*)Mdev, q, Mout, r);
}
template <int CAP, typename T, bool HAS_FTIP>
__global__ __launch_bounds__(kBlock) void k_dyn_forward_dynamics(const MpBigModel<T>* __restrict__ Mdev, const MpCall<T> C,
                                                                 const T* __restrict__ q, const T* __restrict__ qd,
                                                                 const T* __restrict__ tau, T* __restrict__ qdd, long rows) {
  const long r = (long)blockIdx.x * kBlock + threadIdx.x;
  if (r >= rows) return;
  mp_dyn_row_forward_dynamics<CAP, T, HAS_FTIP>(*(MpBigConst<T>*)Mdev, C, q, qd, tau, qdd, r);
}
template <int CAP, typename T, bool HAS_FTIP>
__global__ __launch_bounds__(64) void k_dyn_fd_traj(const MpBigModel<T>* __restrict__ Mdev, const MpCall<T> C,
                                                    const T* __restrict__ theta0, const T* __restrict__ dtheta0,
                                                    const T* __restrict__ taumat, const T* __restrict__ Ftipmat, long B, long Nt, T h,
                                                    int intRes, float* __restrict__ pos, float* __restrict__ vel,
                                                    float* __restrict__ acc, int time_major) {
  const long b = (long)blockIdx.x * 64 + threadIdx.x;
  if (b >= B) return;
  mp_dyn_rollout<CAP, T, HAS_FTIP>(*(MpBigConst<T>*)Mdev, C, theta0, dtheta0, taumat, Ftipmat, b, B, Nt, h, intRes, pos, vel, acc,
                              time_major != 0);
}
template <int CAP, bool HAS_FTIP>
__global__ __launch_bounds__(kBlock) void k_dyn_traj(const MpBigModel<float>* __restrict__ Mdev, const MpCall<float> C,
                                                     const float* __restrict__ start, const float* __restrict__ end, long B, long Nt,
                                                     double Tf, int method, float* __restrict__ pos, float* __restrict__ vel,
                                                     float* __restrict__ acc, float* __restrict__ tau) {
  const long r = (long)blockIdx.x * kBlock + threadIdx.x;
  if (r >= B * Nt) return;
  const long b = r / Nt;
  mp_dyn_row_traj<CAP, HAS_FTIP>(*(MpBigConst<float>*)Mdev, C, start, end, b, r - b * Nt, Nt, Tf, method, pos, vel, acc, tau);
}

inline unsigned grid_for(long rows) { return (unsigned)((rows + kBlock - 1) / kBlock); }

#define MP_DISPATCH_N(n, ...)                                   \
  switch (n) {                                                  \
    case 1: { constexpr int N = 1; __VA_ARGS__; } break;        \
    case 2: { constexpr int N = 2; __VA_ARGS__; } break;        \
    case 3: { constexpr int N = 3; __VA_ARGS__; } break;        \
    case 4: { constexpr int N = 4; __VA_ARGS__; } break;        \
    case 5: { constexpr int N = 5; __VA_ARGS__; } break;        \
    case 6: { constexpr int N = 6; __VA_ARGS__; } break;        \
    case 7: { constexpr int N = 7; __VA_ARGS__; } break;        \
    case 8: { constexpr int N = 8; __VA_ARGS__; } break;        \
    default: return hipErrorInvalidValue;                       \
  }
// the looped kernels' array capacity for a model of n joints
#define MP_DISPATCH_CAP(n, ...)                                                       \
  if ((n) <= MP_MID_DOF) { constexpr int CAP = MP_MID_DOF; __VA_ARGS__; }             \
  else if ((n) <= MP_BIG_DOF) { constexpr int CAP = MP_BIG_DOF; __VA_ARGS__; }        \
  else return hipErrorInvalidValue;

}  // namespace

// streaming probes for the roofline: dst = a (one read per write) or dst = a + b + c (the 3 : 1 byte mix of the ID kernels)
typedef float mp_f4v __attribute__((ext_vector_type(4)));
template <int READS, bool NT>
__global__ __launch_bounds__(256) void k_stream(const mp_f4v* __restrict__ a, const mp_f4v* __restrict__ b, const mp_f4v* __restrict__ c,
                                                mp_f4v* __restrict__ d, long n4) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  mp_f4v v = NT ? __builtin_nontemporal_load(a + i) : a[i];
  if (READS == 3) v = v + (NT ? __builtin_nontemporal_load(b + i) : b[i]) + (NT ? __builtin_nontemporal_load(c + i) : c[i]);
  if (NT) __builtin_nontemporal_store(v, d + i);
  else d[i] = v;
}
hipError_t mpk_stream(hipStream_t s, int reads, bool nontemporal, const void* a, const void* b, const void* c, void* d, long n4) {
  const unsigned grid = (unsigned)((n4 + 255) / 256);
#define MP_STREAM_LAUNCH(R, NT) hipLaunchKernelGGL((k_stream<R, NT>), dim3(grid), dim3(256), 0, s, (const mp_f4v*)a, (const mp_f4v*)b, (const mp_f4v*)c, (mp_f4v*)d, n4)
  if (reads == 3) { if (nontemporal) MP_STREAM_LAUNCH(3, true); else MP_STREAM_LAUNCH(3, false); }
  else { if (nontemporal) MP_STREAM_LAUNCH(1, true); else MP_STREAM_LAUNCH(1, false); }
#undef MP_STREAM_LAUNCH
  return hipGetLastError();
}

// the same with any byte mix: every lane reads R 16-byte chunks (one from each of R arrays laid back to back in `a`) and writes W
// (into W arrays back to back in `d`) - c3's kernel writes three bytes for every one it reads, the roll-out reads two for three
template <int R, int W, bool NT>
__global__ __launch_bounds__(256) void k_stream_mix(const mp_f4v* __restrict__ a, mp_f4v* __restrict__ d, long n4) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  mp_f4v v = {1.f, 2.f, 3.f, 4.f};
#pragma unroll
  for (int r = 0; r < R; ++r) v = v + (NT ? __builtin_nontemporal_load(a + r * n4 + i) : a[r * n4 + i]);
#pragma unroll
  for (int w = 0; w < W; ++w) {
    if (NT) __builtin_nontemporal_store(v, d + w * n4 + i);
    else d[w * n4 + i] = v;
  }
}
hipError_t mpk_stream_mix(hipStream_t s, int reads, int writes, bool nontemporal, const void* a, void* d, long n4) {
  const unsigned grid = (unsigned)((n4 + 255) / 256);
#define MP_MIX(R, W)                                                                                                              \
  if (reads == R && writes == W) {                                                                                                \
    if (nontemporal) hipLaunchKernelGGL((k_stream_mix<R, W, true>), dim3(grid), dim3(256), 0, s, (const mp_f4v*)a, (mp_f4v*)d, n4); \
    else hipLaunchKernelGGL((k_stream_mix<R, W, false>), dim3(grid), dim3(256), 0, s, (const mp_f4v*)a, (mp_f4v*)d, n4);          \
    return hipGetLastError();                                                                                                     \
  }
  MP_MIX(0, 1) MP_MIX(1, 1) MP_MIX(3, 1) MP_MIX(1, 3) MP_MIX(2, 3) MP_MIX(1, 2) MP_MIX(2, 1)
#undef MP_MIX
  return hipErrorInvalidValue;
}

// Shader-clock sampler for the measurement harness (MI355X_MICROARCH.md, "DVFS give-back" (6): the in-kernel clock is
// delta s_memtime / delta s_memrealtime x 100 MHz).  One wave per block; lane 0 stamps both counters, naps `naps` x s_sleep 127
// (64 x 127 cycles each) and stamps again, `samples` times - strictly bounded, nothing to wait for.  It is launched on a stream of its
// own BESIDE the kernels whose clock is asked for and costs them one wave slot per block; out[block][sample] = {memtime, realtime}.
__global__ __launch_bounds__(64) void k_clock_sampler(unsigned long long* __restrict__ out, unsigned samples, unsigned naps) {
  if (threadIdx.x != 0) return;
  unsigned long long* o = out + (size_t)blockIdx.x * samples * 2;
  for (unsigned i = 0; i < samples; ++i) {
    o[2 * i] = __builtin_amdgcn_s_memtime();
    o[2 * i + 1] = __builtin_amdgcn_s_memrealtime();
    for (unsigned k = 0; k < naps; ++k) __builtin_amdgcn_s_sleep(127);
  }
}
hipError_t mpk_clock_sampler(hipStream_t s, unsigned long long* out, unsigned blocks, unsigned samples, unsigned naps) {
  hipLaunchKernelGGL(k_clock_sampler, dim3(blocks), dim3(64), 0, s, out, samples, naps);
  return hipGetLastError();
}

hipError_t mpk_selftest(hipStream_t s, int* d_out) {
  hipLaunchKernelGGL(k_selftest, dim3(1), dim3(64), 0, s, d_out);
  return hipGetLastError();
}

// one row per lane, model through a device pointer (k_id_dm)
hipError_t mpk_id_dm(hipStream_t s, const MpModel<float>* d_model, int n, const MpCall<float>& C, bool ftip, const float* q,
                     const float* qd, const float* qdd, float* tau, long rows, const MpLead& L, bool all_revolute) {
  if (rows <= 0) return hipSuccess;
  using T = float;
  const dim3 grid(grid_for(rows) + L.blocks), block(kBlock);
  MP_DISPATCH_N(n, {
    if (all_revolute) {
      if (ftip) hipLaunchKernelGGL((k_id_dm<T, N, true, true>), grid, block, 0, s, d_model, C, q, qd, qdd, tau, rows, L);
      else hipLaunchKernelGGL((k_id_dm<T, N, false, true>), grid, block, 0, s, d_model, C, q, qd, qdd, tau, rows, L);
    } else {
      if (ftip) hipLaunchKernelGGL((k_id_dm<T, N, true, false>), grid, block, 0, s, d_model, C, q, qd, qdd, tau, rows, L);
      else hipLaunchKernelGGL((k_id_dm<T, N, false, false>), grid, block, 0, s, d_model, C, q, qd, qdd, tau, rows, L);
    }
  })
  return hipGetLastError();
}

hipError_t mpk_id_hard(hipStream_t s, const MpModel<float>* d_model, int n, const MpCall<float>& C, bool ftip, const float* q,
                       const float* qd, const float* qdd, float* tau, unsigned rows, unsigned blocks) {
  if (blocks == 0 || !C.hard_rows || !C.cold_model) return hipSuccess;
  MP_DISPATCH_N(n, {
    if (ftip) hipLaunchKernelGGL((k_id_hard<N, true>), dim3(blocks), dim3(64), 0, s, d_model, C, q, qd, qdd, tau, rows);
    else hipLaunchKernelGGL((k_id_hard<N, false>), dim3(blocks), dim3(64), 0, s, d_model, C, q, qd, qdd, tau, rows);
  })
  return hipGetLastError();
}

hipError_t mpk_id_hard_batch(hipStream_t s, const MpModel<float>* d_model, int n, bool ftip, const MpHardBatch& B, int entries, unsigned blocks) {
  if (blocks == 0 || entries <= 0) return hipSuccess;
  MP_DISPATCH_N(n, {
    if (ftip) hipLaunchKernelGGL((k_id_hard_batch<N, true>), dim3(blocks, (unsigned)entries), dim3(64), 0, s, d_model, B);
    else hipLaunchKernelGGL((k_id_hard_batch<N, false>), dim3(blocks, (unsigned)entries), dim3(64), 0, s, d_model, B);
  })
  return hipGetLastError();
}

template <>
hipError_t mpk_id<double>(hipStream_t s, const MpModel<double>& M, const MpCall<double>& C, bool ftip, const double* q,
                          const double* qd, const double* qdd, double* tau, long rows) {
  using T = double;
  if (rows <= 0) return hipSuccess;
  MP_DISPATCH_N(M.n, {
    if (ftip) hipLaunchKernelGGL((k_id<T, N, true>), dim3(grid_for(rows)), dim3(kBlock), 0, s, M, C, q, qd, qdd, tau, rows);
    else hipLaunchKernelGGL((k_id<T, N, false>), dim3(grid_for(rows)), dim3(kBlock), 0, s, M, C, q, qd, qdd, tau, rows);
  })
  return hipGetLastError();
}

hipError_t mpk_batch_traj(hipStream_t s, const MpModel<float>& M, const float* start, const float* end, long B,
                          long Nt, double Tf, int method, float* pos, float* vel, float* acc) {
  const long rows = B * Nt;
  if (rows <= 0) return hipSuccess;
  MP_DISPATCH_N(M.n, {
    hipLaunchKernelGGL((k_batch_traj<N>), dim3(grid_for(rows)), dim3(kBlock), 0, s, M, start, end, B, Nt, Tf, method, pos, vel, acc);
  })
  return hipGetLastError();
}

hipError_t mpk_time_table(hipStream_t s, double* tab, long Nt, double Tf, int method) {
  if (Nt <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_time_table, dim3(grid_for(Nt)), dim3(kBlock), 0, s, tab, Nt, Tf, method);
  return hipGetLastError();
}

// blocks per trajectory of the table-driven fused kernels (block = 256 lanes, ceil(Nt / 2) lanes per trajectory)
unsigned mpk_traj_blocks_per_trajectory(long Nt) { return (unsigned)(((Nt + 1) / 2 + kBlock - 1) / kBlock); }

hipError_t mpk_traj_id_tab(hipStream_t s, const MpModel<float>& M, const MpCall<float>& C, bool ftip, const float* start,
                           const float* end, long B, long Nt, const double* tab, float* tau) {
  if (B <= 0 || Nt <= 0) return hipSuccess;
  const unsigned bpt = mpk_traj_blocks_per_trajectory(Nt);
  const unsigned grid = (unsigned)(B * bpt);
  MP_DISPATCH_N(M.n, {
    if (ftip) hipLaunchKernelGGL((k_traj_id_pk_tab<N, true>), dim3(grid), dim3(kBlock), 0, s, M, C, start, end, Nt, bpt, tab, tau);
    else hipLaunchKernelGGL((k_traj_id_pk_tab<N, false>), dim3(grid), dim3(kBlock), 0, s, M, C, start, end, Nt, bpt, tab, tau);
  })
  return hipGetLastError();
}

hipError_t mpk_traj_id_hard(hipStream_t s, const MpModel<float>* d_model, int n, const MpCall<float>& C, bool ftip, const float* start,
                            const float* end, unsigned Nt, const double* tab, float* tau, unsigned rows, unsigned blocks) {
  if (blocks == 0 || !C.hard_rows || !C.cold_model) return hipSuccess;
  MP_DISPATCH_N(n, {
    if (ftip) hipLaunchKernelGGL((k_traj_id_hard<N, true>), dim3(blocks), dim3(64), 0, s, d_model, C, start, end, Nt, tab, tau, rows);
    else hipLaunchKernelGGL((k_traj_id_hard<N, false>), dim3(blocks), dim3(64), 0, s, d_model, C, start, end, Nt, tab, tau, rows);
  })
  return hipGetLastError();
}

template <typename T>
hipError_t mpk_fk_jac_id(hipStream_t s, const MpModel<T>& M, const MpCall<T>& C, bool ftip, const T* q, const T* qd,
                         const T* qdd, T* Tout, T* Jout, T* tau, long rows) {
  if (rows <= 0) return hipSuccess;
  MP_DISPATCH_N(M.n, {
    const unsigned gb = (unsigned)((rows + kFkBlock - 1) / kFkBlock);
    if (ftip) hipLaunchKernelGGL((k_fk_jac_id<T, N, true>), dim3(gb), dim3(kFkBlock), 0, s, M, C, q, qd, qdd, Tout, Jout, tau, rows);
    else hipLaunchKernelGGL((k_fk_jac_id<T, N, false>), dim3(gb), dim3(kFkBlock), 0, s, M, C, q, qd, qdd, Tout, Jout, tau, rows);
  })
  return hipGetLastError();
}
template hipError_t mpk_fk_jac_id<float>(hipStream_t, const MpModel<float>&, const MpCall<float>&, bool, const float*,
                                         const float*, const float*, float*, float*, float*, long);
template hipError_t mpk_fk_jac_id<double>(hipStream_t, const MpModel<double>&, const MpCall<double>&, bool,
                                          const double*, const double*, const double*, double*, double*, double*, long);

template <typename T>
hipError_t mpk_mass_matrix(hipStream_t s, const MpModel<T>& M, const T* q, T* Mout, long rows) {
  if (rows <= 0) return hipSuccess;
  MP_DISPATCH_N(M.n, { hipLaunchKernelGGL((k_mass_matrix<T, N>), dim3((unsigned)((rows + kFkBlock - 1) / kFkBlock)), dim3(kFkBlock), 0, s, M, q, Mout, rows); })
  return hipGetLastError();
}
template hipError_t mpk_mass_matrix<float>(hipStream_t, const MpModel<float>&, const float*, float*, long);
template hipError_t mpk_mass_matrix<double>(hipStream_t, const MpModel<double>&, const double*, double*, long);

template <typename T>
hipError_t mpk_forward_dynamics(hipStream_t s, const MpModel<T>& M, const MpCall<T>& C, bool ftip, const T* q, const T* qd,
                                const T* tau, T* qdd, long rows) {
  if (rows <= 0) return hipSuccess;
  MP_DISPATCH_N(M.n, {
    if (ftip) hipLaunchKernelGGL((k_forward_dynamics<T, N, true>), dim3(grid_for(rows)), dim3(kBlock), 0, s, M, C, q, qd, tau, qdd, rows);
    else hipLaunchKernelGGL((k_forward_dynamics<T, N, false>), dim3(grid_for(rows)), dim3(kBlock), 0, s, M, C, q, qd, tau, qdd, rows);
  })
  return hipGetLastError();
}
template hipError_t mpk_forward_dynamics<float>(hipStream_t, const MpModel<float>&, const MpCall<float>&, bool, const float*,
                                                const float*, const float*, float*, long);
template hipError_t mpk_forward_dynamics<double>(hipStream_t, const MpModel<double>&, const MpCall<double>&, bool,
                                                 const double*, const double*, const double*, double*, long);

template <typename T>
hipError_t mpk_fd_traj(hipStream_t s, const MpModel<T>& M, const MpCall<T>& C, const T* theta0, const T* dtheta0,
                       const T* taumat, const T* Ftipmat, long B, long Nt, T h, int intRes, float* pos, float* vel, float* acc) {
  if (B <= 0 || Nt <= 0) return hipSuccess;
  MP_DISPATCH_N(M.n, {
    if (Ftipmat) hipLaunchKernelGGL((k_fd_traj<T, N, true>), dim3((unsigned)((B + kFdBlock - 1) / kFdBlock)), dim3(kFdBlock), 0, s, M, C, theta0, dtheta0, taumat, Ftipmat, B, Nt, h, intRes, pos, vel, acc);
    else hipLaunchKernelGGL((k_fd_traj<T, N, false>), dim3((unsigned)((B + kFdBlock - 1) / kFdBlock)), dim3(kFdBlock), 0, s, M, C, theta0, dtheta0, taumat, Ftipmat, B, Nt, h, intRes, pos, vel, acc);
  })
  return hipGetLastError();
}
template hipError_t mpk_fd_traj<float>(hipStream_t, const MpModel<float>&, const MpCall<float>&, const float*, const float*,
                                       const float*, const float*, long, long, float, int, float*, float*, float*);
template hipError_t mpk_fd_traj<double>(hipStream_t, const MpModel<double>&, const MpCall<double>&, const double*,
                                        const double*, const double*, const double*, long, long, double, int, float*, float*,
                                        float*);

template <typename T>
hipError_t mpk_fd_traj_tm(hipStream_t s, const MpModel<T>& M, const MpCall<T>& C, const T* theta0, const T* dtheta0,
                          const T* taumat, const T* Ftipmat, long B, long Nt, T h, int intRes, float* pos, float* vel, float* acc) {
  if (B <= 0 || Nt <= 0) return hipSuccess;
  const dim3 grid((unsigned)((B + kFdBlock - 1) / kFdBlock));
  MP_DISPATCH_N(M.n, {
    if (Ftipmat) hipLaunchKernelGGL((k_fd_traj_tm<T, N, true>), grid, dim3(kFdBlock), 0, s, M, C, theta0, dtheta0, taumat, Ftipmat, B, Nt, h, intRes, pos, vel, acc);
    else hipLaunchKernelGGL((k_fd_traj_tm<T, N, false>), grid, dim3(kFdBlock), 0, s, M, C, theta0, dtheta0, taumat, Ftipmat, B, Nt, h, intRes, pos, vel, acc);
  })
  return hipGetLastError();
}
template hipError_t mpk_fd_traj_tm<float>(hipStream_t, const MpModel<float>&, const MpCall<float>&, const float*, const float*,
                                          const float*, const float*, long, long, float, int, float*, float*, float*);
template hipError_t mpk_fd_traj_tm<double>(hipStream_t, const MpModel<double>&, const MpCall<double>&, const double*,
                                           const double*, const double*, const double*, long, long, double, int, float*, float*,
                                           float*);

// ---- launchers of the run-time-n kernels (d_model: MpBigModel<T> resident in device memory)
template <typename T>
hipError_t mpk_dyn_fk_jac_id(hipStream_t s, int n, const MpBigModel<T>* d_model, const MpCall<T>& C, bool ftip, const T* q, const T* qd,
                             const T* qdd, T* Tout, T* Jout, T* tau, long rows) {
  if (rows <= 0) return hipSuccess;
  MP_DISPATCH_CAP(n, {
    if (ftip) hipLaunchKernelGGL((k_dyn_fk_jac_id<CAP, T, true>), dim3(grid_for(rows)), dim3(kBlock), 0, s, d_model, C, q, qd, qdd, Tout, Jout, tau, rows);
    else hipLaunchKernelGGL((k_dyn_fk_jac_id<CAP, T, false>), dim3(grid_for(rows)), dim3(kBlock), 0, s, d_model, C, q, qd, qdd, Tout, Jout, tau, rows);
  })
  return hipGetLastError();
}
template hipError_t mpk_dyn_fk_jac_id<float>(hipStream_t, int, const MpBigModel<float>*, const MpCall<float>&, bool, const float*,
                                             const float*, const float*, float*, float*, float*, long);
template hipError_t mpk_dyn_fk_jac_id<double>(hipStream_t, int, const MpBigModel<double>*, const MpCall<double>&, bool, const double*,
                                              const double*, const double*, double*, double*, double*, long);
template <typename T>
hipError_t mpk_dyn_mass_matrix(hipStream_t s, int n, const MpBigModel<T>* d_model, const T* q, T* Mout, long rows) {
  if (rows <= 0) return hipSuccess;
  MP_DISPATCH_CAP(n, { hipLaunchKernelGGL((k_dyn_mass_matrix<CAP, T>), dim3(grid_for(rows)), dim3(kBlock), 0, s, d_model, q, Mout, rows); })
  return hipGetLastError();
}
template hipError_t mpk_dyn_mass_matrix<float>(hipStream_t, int, const MpBigModel<float>*, const float*, float*, long);
template hipError_t mpk_dyn_mass_matrix<double>(hipStream_t, int, const MpBigModel<double>*, const double*, double*, long);
template <typename T>
hipError_t mpk_dyn_forward_dynamics(hipStream_t s, int n, const MpBigModel<T>* d_model, const MpCall<T>& C, bool ftip, const T* q,
                                    const T* qd, const T* tau, T* qdd, long rows) {
  if (rows <= 0) return hipSuccess;
  MP_DISPATCH_CAP(n, {
    if (ftip) hipLaunchKernelGGL((k_dyn_forward_dynamics<CAP, T, true>), dim3(grid_for(rows)), dim3(kBlock), 0, s, d_model, C, q, qd, tau, qdd, rows);
    else hipLaunchKernelGGL((k_dyn_forward_dynamics<CAP, T, false>), dim3(grid_for(rows)), dim3(kBlock), 0, s, d_model, C, q, qd, tau, qdd, rows);
  })
  return hipGetLastError();
}
template hipError_t mpk_dyn_forward_dynamics<float>(hipStream_t, int, const MpBigModel<float>*, const MpCall<float>&, bool, const float*,
                                                    const float*, const float*, float*, long);
template hipError_t mpk_dyn_forward_dynamics<double>(hipStream_t, int, const MpBigModel<double>*, const MpCall<double>&, bool,
                                                     const double*, const double*, const double*, double*, long);
template <typename T>
hipError_t mpk_dyn_fd_traj(hipStream_t s, int n, const MpBigModel<T>* d_model, const MpCall<T>& C, const T* theta0, const T* dtheta0,
                           const T* taumat, const T* Ftipmat, long B, long Nt, T h, int intRes, float* pos, float* vel, float* acc,
                           bool time_major) {
  if (B <= 0 || Nt <= 0) return hipSuccess;
  const dim3 grid((unsigned)((B + 63) / 64));
  const int tm = time_major ? 1 : 0;
  MP_DISPATCH_CAP(n, {
    if (Ftipmat) hipLaunchKernelGGL((k_dyn_fd_traj<CAP, T, true>), grid, dim3(64), 0, s, d_model, C, theta0, dtheta0, taumat, Ftipmat, B, Nt, h, intRes, pos, vel, acc, tm);
    else hipLaunchKernelGGL((k_dyn_fd_traj<CAP, T, false>), grid, dim3(64), 0, s, d_model, C, theta0, dtheta0, taumat, Ftipmat, B, Nt, h, intRes, pos, vel, acc, tm);
  })
  return hipGetLastError();
}
template hipError_t mpk_dyn_fd_traj<float>(hipStream_t, int, const MpBigModel<float>*, const MpCall<float>&, const float*, const float*,
                                           const float*, const float*, long, long, float, int, float*, float*, float*, bool);
template hipError_t mpk_dyn_fd_traj<double>(hipStream_t, int, const MpBigModel<double>*, const MpCall<double>&, const double*,
                                            const double*, const double*, const double*, long, long, double, int, float*, float*, float*,
                                            bool);
hipError_t mpk_dyn_traj(hipStream_t s, int n, const MpBigModel<float>* d_model, const MpCall<float>& C, bool ftip, const float* start,
                        const float* end, long B, long Nt, double Tf, int method, float* pos, float* vel, float* acc, float* tau) {
  if (B <= 0 || Nt <= 0) return hipSuccess;
  MP_DISPATCH_CAP(n, {
    if (ftip) hipLaunchKernelGGL((k_dyn_traj<CAP, true>), dim3(grid_for(B * Nt)), dim3(kBlock), 0, s, d_model, C, start, end, B, Nt, Tf, method, pos, vel, acc, tau);
    else hipLaunchKernelGGL((k_dyn_traj<CAP, false>), dim3(grid_for(B * Nt)), dim3(kBlock), 0, s, d_model, C, start, end, B, Nt, Tf, method, pos, vel, acc, tau);
  })
  return hipGetLastError();
}

hipError_t mpk_transpose_rows(hipStream_t s, const void* src, void* dst, long outer, long inner, int row_dwords) {
  if (outer <= 0 || inner <= 0 || row_dwords <= 0) return hipSuccess;
  const int TI = mp_tr_ti(row_dwords);
  const long gx = (inner + TI - 1) / TI, gy = (outer + MP_TR_TO - 1) / MP_TR_TO;
  if (gx * gy > 0x7fffffffL) return hipErrorInvalidValue;
  const size_t lds = (size_t)MP_TR_TO * (TI * row_dwords + 1) * sizeof(unsigned);
  hipLaunchKernelGGL(k_transpose_rows, dim3((unsigned)(gx * gy)), dim3(kBlock), lds, s, (const unsigned*)src, (unsigned*)dst, outer,
                     inner, row_dwords);
  return hipGetLastError();
}

// K closed-loop regulation runs (mp_pd_regulation_run), one lane each: rows of theta0 / des (K, n), gains per run, errors (K, steps)
template <int N>
__global__ __launch_bounds__(64) void k_pd_regulation(const MpModel<double> M, const MpCall<double> C, const double* __restrict__ theta0,
                                                      const double* __restrict__ des, const double* __restrict__ Kp,
                                                      const double* __restrict__ Kd, long K, double dt, int steps,
                                                      double* __restrict__ err, int* __restrict__ count) {
  const long k = (long)blockIdx.x * 64 + threadIdx.x;
  if (k >= K) return;
  double a[N], d[N];
  RunIO<double, N>::load(theta0, k, a);
  RunIO<double, N>::load(des, k, d);
  count[k] = mp_pd_regulation_run<double, N>(M, C.a0, a, d, Kp[k], Kd[k], dt, steps, err + k * steps);
}
template <int CAP>
__global__ __launch_bounds__(64) void k_dyn_pd_regulation(const MpBigModel<double>* __restrict__ Mdev, const MpCall<double> C,
                                                          const double* __restrict__ theta0, const double* __restrict__ des,
                                                          const double* __restrict__ Kp, const double* __restrict__ Kd, long K, double dt,
                                                          int steps, double* __restrict__ err, int* __restrict__ count) {
  const long k = (long)blockIdx.x * 64 + threadIdx.x;
  if (k >= K) return;
  MpBigConst<double>& M = *(MpBigConst<double>*)Mdev;
  count[k] = mp_dyn_pd_regulation_run<CAP, double>(M, C.a0, theta0 + k * M.n, des + k * M.n, Kp[k], Kd[k], dt, steps, err + k * steps);
}

// inverse kinematics with a run-time joint count: the work queue of k_ik, the looped kinematics of mp_dyn.h.  CAP: the capacity of
// the per-problem arrays (16 for the reference's 9- and 10-joint Jaco arms, 32 beyond: MP_DISPATCH_CAP, as the dynamics kernels)
template <int CAP>
__global__ __launch_bounds__(kBlock) void k_dyn_ik(const MpBigModel<double>* __restrict__ Mdev, const MpIkParamsT<CAP> P,
                                                   const double* __restrict__ Tdes, const double* __restrict__ theta0, long B,
                                                   double* __restrict__ theta, int* __restrict__ success, int* __restrict__ iterations,
                                                   int* __restrict__ restarts, unsigned long long* __restrict__ next) {
  MpBigConst<double>& M = *(MpBigConst<double>*)Mdev;
  const int n = M.n;
  MpIkState<CAP> S;
  bool have = false;
  long row = 0;
  for (;;) {  // exit: the counter only grows, so every lane sees row >= B eventually
    if (!have) {
      row = (long)atomicAdd(next, 1ull);
      if (row >= B) break;
      for (int j = 0; j < CAP; ++j) S.theta[j] = j < n ? theta0[row * n + j] : 0.0;
      mp_ik_begin(S, P);
      have = true;
    }
    if (const int done = mp_ik_iterate<CAP, MpIkLooped<CAP>>(M, P, S, Tdes + row * 16, theta0 + row * n)) {
      for (int j = 0; j < n; ++j) theta[row * n + j] = S.theta[j];
      success[row] = done == 2 ? 1 : 0;
      iterations[row] = S.k + 1;
      restarts[row] = S.restarts;
      have = false;
    }
  }
}
hipError_t mpk_cartesian_traj(hipStream_t s, const double* Xstart, const double* Xend, long B, long Nt, double Tf, int method,
                              float* pos, float* vel, float* acc, float* ori) {
  if (B <= 0 || Nt <= 0) return hipSuccess;
  const unsigned bpt = (unsigned)((Nt + kBlock - 1) / kBlock);
  hipLaunchKernelGGL(k_cartesian_traj, dim3((unsigned)(B * bpt)), dim3(kBlock), 0, s, Xstart, Xend, Nt, bpt, Tf, method, pos, vel, acc, ori);
  return hipGetLastError();
}

hipError_t mpk_ik(hipStream_t s, const MpModel<double>& M, const MpIkParams& P, const double* Tdes, const double* theta0, long B,
                  double* theta, int* success, int* iterations, int* restarts, unsigned long long* queue_counter, int compute_units) {
  if (B <= 0) return hipSuccess;
  hipError_t e = hipMemsetAsync(queue_counter, 0, sizeof(unsigned long long), s);
  if (e != hipSuccess) return e;
  // resident lanes only (two 256-thread blocks per CU; fp64 IK needs > 128 VGPRs): the queue feeds them
  const long want = (B + kBlock - 1) / kBlock, cap = 2L * (compute_units > 0 ? compute_units : 256);
  const unsigned grid = (unsigned)(want < cap ? want : cap);
  MP_DISPATCH_N(M.n, { hipLaunchKernelGGL((k_ik<N>), dim3(grid), dim3(kBlock), 0, s, M, P, Tdes, theta0, B, theta, success, iterations, restarts, queue_counter); })
  return hipGetLastError();
}

hipError_t mpk_pd_regulation(hipStream_t s, const MpModel<double>& M, const MpCall<double>& C, const double* theta0, const double* des,
                             const double* Kp, const double* Kd, long K, double dt, int steps, double* err, int* count) {
  if (K <= 0) return hipSuccess;  // (steps == 0 still launches: every run reports a count of 0)
  MP_DISPATCH_N(M.n, { hipLaunchKernelGGL((k_pd_regulation<N>), dim3((unsigned)((K + 63) / 64)), dim3(64), 0, s, M, C, theta0, des, Kp, Kd, K, dt, steps, err, count); })
  return hipGetLastError();
}
hipError_t mpk_dyn_pd_regulation(hipStream_t s, int n, const MpBigModel<double>* d_model, const MpCall<double>& C, const double* theta0,
                                 const double* des, const double* Kp, const double* Kd, long K, double dt, int steps, double* err, int* count) {
  if (K <= 0) return hipSuccess;  // (steps == 0 still launches: every run reports a count of 0)
  MP_DISPATCH_CAP(n, { hipLaunchKernelGGL((k_dyn_pd_regulation<CAP>), dim3((unsigned)((K + 63) / 64)), dim3(64), 0, s, d_model, C, theta0, des, Kp, Kd, K, dt, steps, err, count); })
  return hipGetLastError();
}

template <int CAP>
static MpIkParamsT<CAP> ik_params_for(const MpIkBigParams& P) {
  MpIkParamsT<CAP> Q;
  Q.eomg = P.eomg; Q.ev = P.ev; Q.damping = P.damping; Q.step_cap = P.step_cap; Q.w_o = P.w_o; Q.w_p = P.w_p;
  Q.max_iterations = P.max_iterations; Q.seed = P.seed; Q.adaptive_tuning = P.adaptive_tuning; Q.backtracking = P.backtracking;
  for (int j = 0; j < CAP; ++j) { Q.lo[j] = P.lo[j]; Q.hi[j] = P.hi[j]; }
  return Q;
}
hipError_t mpk_dyn_ik(hipStream_t s, int n, const MpBigModel<double>* d_model, const MpIkBigParams& P, const double* Tdes, const double* theta0,
                      long B, double* theta, int* success, int* iterations, int* restarts, unsigned long long* queue_counter,
                      int compute_units) {
  if (B <= 0) return hipSuccess;
  hipError_t e = hipMemsetAsync(queue_counter, 0, sizeof(unsigned long long), s);
  if (e != hipSuccess) return e;
  const long want = (B + kBlock - 1) / kBlock, cap = 2L * (compute_units > 0 ? compute_units : 256);
  MP_DISPATCH_CAP(n, {
    const MpIkParamsT<CAP> Q = ik_params_for<CAP>(P);
    hipLaunchKernelGGL((k_dyn_ik<CAP>), dim3((unsigned)(want < cap ? want : cap)), dim3(kBlock), 0, s, d_model, Q, Tdes, theta0, B, theta, success,
                       iterations, restarts, queue_counter);
  })
  return hipGetLastError();
}

hipError_t mpk_potential_field(hipStream_t s, const float* pos, const float* goal3_host, const float* obs, long P, long O,
                               float influence, float* pot, float* grad) {
  if (P <= 0) return hipSuccess;
  const float inv = influence > 0.0f ? (float)(1.0 / (double)influence) : 0.0f;
  hipLaunchKernelGGL(k_potential_field, dim3(grid_for(P)), dim3(kBlock), 0, s, pos, goal3_host[0], goal3_host[1], goal3_host[2], obs,
                     P, O, inv, influence * influence, pot, grad);
  return hipGetLastError();
}
