// Device-side building blocks shared by the ahead-of-time kernels (mp_kernels.hip) and the run-time
// specialised kernels (mp_jit.cpp compiles thin wrappers around the same bodies with one robot's
// constants baked in).  Device code only; include after <hip/hip_runtime.h> (hipcc) or stand-alone (hiprtc).
#pragma once

#include "mp_core.h"

// ---- widest legal vector type for a run of COUNT elements of T whose start is COUNT*sizeof(T)-strided
template <typename T, int BYTES> struct VecOf;
typedef float mp_io_f4 __attribute__((ext_vector_type(4)));   // native vectors: dwordx4 / dwordx2 accesses,
typedef float mp_io_f2 __attribute__((ext_vector_type(2)));   // and legal operands of the nontemporal builtins
typedef double mp_io_d2 __attribute__((ext_vector_type(2)));
template <> struct VecOf<float, 16> { using type = mp_io_f4; static constexpr int K = 4; };
template <> struct VecOf<float, 8> { using type = mp_io_f2; static constexpr int K = 2; };
template <> struct VecOf<float, 4> { using type = float; static constexpr int K = 1; };
template <> struct VecOf<double, 16> { using type = mp_io_d2; static constexpr int K = 2; };
template <> struct VecOf<double, 8> { using type = double; static constexpr int K = 1; };

// Non-temporal global accesses for data a launch touches once.  They pay ONLY where one instruction covers whole lines (the
// wave-cooperative movers below): on the c2 pattern - three 98 MB input streams, one output stream, rotating sets - plain
// accesses reach 5.7 TB/s and non-temporal ones 6.4 (tools/ubench_nt.hip, 16 contiguous bytes per lane), but applied to the
// per-lane 24-byte rows of RunIO, which touch every line with two or three instructions and rely on the caches to merge them,
// they LOSE (c2 0.069 -> 0.085 ms, c3 3.8 -> 4.6 ms).  (The plain forms were build switches until round 5: profiles/HISTORY.md.)
typedef unsigned mp_u4 __attribute__((ext_vector_type(4)));
template <typename V>
__device__ __forceinline__ V mp_stream_load(const V* p) { return __builtin_nontemporal_load(p); }
template <typename V>
__device__ __forceinline__ void mp_stream_store(V v, V* p) { __builtin_nontemporal_store(v, p); }

// The FK + Jacobian + ID kernel moves its input rows / tau as whole lines, non-temporal, and its wave-cooperative output stores
// (whole 16-byte chunks in flat order) are non-temporal too.  c3 on a box in its fast state: 3.71 ms plain, 3.66 stores only,
// 3.52 inputs only, 3.45 both (frac 0.82); in its slow (power-capped) state 4.09 / 3.99 / 4.21 / 4.12 - the staging instructions
// cost there what the memory path gains (profiles/r03_nontemporal_ab.txt)
template <typename V>
__device__ __forceinline__ void mp_coop_store(V v, V* p) { __builtin_nontemporal_store(v, p); }

template <typename T, int COUNT>
struct RunIO {
  static constexpr int BYTES = COUNT * (int)sizeof(T);
  static constexpr int W = (BYTES % 16 == 0) ? 16 : (BYTES % 8 == 0) ? 8 : 4;
  using VO = VecOf<T, W>;
  using V = typename VO::type;
  static constexpr int K = VO::K;
  static_assert(COUNT % K == 0, "run must be a whole number of vectors");

  static __device__ __forceinline__ void load(const T* __restrict__ base, long run, T (&v)[COUNT]) {
    const V* src = reinterpret_cast<const V*>(base + run * COUNT);
#pragma unroll
    for (int k = 0; k < COUNT / K; ++k) {
      union { V vec; T e[K]; } u;
      u.vec = src[k];
#pragma unroll
      for (int j = 0; j < K; ++j) v[k * K + j] = u.e[j];
    }
  }
  static __device__ __forceinline__ void store(T* __restrict__ base, long run, const T (&v)[COUNT]) {
    V* dst = reinterpret_cast<V*>(base + run * COUNT);
#pragma unroll
    for (int k = 0; k < COUNT / K; ++k) {
      union { V vec; T e[K]; } u;
#pragma unroll
      for (int j = 0; j < K; ++j) u.e[j] = v[k * K + j];
      dst[k] = u.vec;
    }
  }
  // per-lane row stores of a kernel that reads (almost) nothing - trajectory generation, the fused generation + inverse dynamics:
  // there is no read stream for the partial lines to disturb, and non-temporal stores measure a few per cent better there
  // (c2f 0.054 -> 0.052 ms)
  static __device__ __forceinline__ void store_wo(T* __restrict__ base, long run, const T (&v)[COUNT]) {
    V* dst = reinterpret_cast<V*>(base + run * COUNT);
#pragma unroll
    for (int k = 0; k < COUNT / K; ++k) {
      union { V vec; T e[K]; } u;
#pragma unroll
      for (int j = 0; j < K; ++j) u.e[j] = v[k * K + j];
      __builtin_nontemporal_store(u.vec, dst + k);
    }
  }
};


// -------------------------------------------------------------- trajectory generation pieces
// Row (b, t) of the time-scaled point-to-point trajectory, reference planning/trajectory.py:45-73,
// with the positions clipped to the joint limits (:311-313).
template <int N, typename MT>
__device__ __forceinline__ void traj_row(const MT& M, const float* __restrict__ start,
                                         const float* __restrict__ end, long b, long t, long Nt, double Tf,
                                         int method, float (&pos)[N], float (&vel)[N], float (&acc)[N]) {
  float a[N], e[N];
  RunIO<float, N>::load(start, b, a);
  RunIO<float, N>::load(end, b, e);
  const double tt = (double)t * (Tf / (double)(Nt - 1));
  const double tau = tt / Tf;
  double s, sd, sdd;
  mp_time_scaling(method, tau, Tf, s, sd, sdd);
#pragma unroll
  for (int j = 0; j < N; ++j) {
    const double d = (double)(e[j] - a[j]);  // float32 difference first, as the reference types it
    pos[j] = mp_clip((float)(s * d + (double)a[j]), M.qmin[j], M.qmax[j]);
    vel[j] = (float)(sd * d);
    acc[j] = (float)(sdd * d);
  }
}


// Torque limits.  A robot-specialised program of a model WITHOUT torque limits (the planner's default: the reference clips only when
// it is given limits, planning/trajectory_dynamics.py:280-282) is built with MP_TAU_UNLIMITED: the clip against +-3e38 - which the
// compiler may not fold, finite-math or not - is then not emitted (n v_med3_f32 per row: 1 % of the c2 kernel, whose time follows
// its instruction count).
template <typename T, typename S>
__device__ __forceinline__ T mp_clip_tau(T v, S lo, S hi) {
#if defined(MP_TAU_UNLIMITED)
  (void)lo; (void)hi;
  return v;
#else
  return mp_clip(v, lo, hi);
#endif
}

// ------------------------------------------------------------ the float64 re-evaluation, wave by wave
// Rows whose float32 result is ill-conditioned (`hard`, mp_rnea_f32) are evaluated again by mp_rnea_cold with the per-joint state
// in LDS: `lds` is MpColdLds<N, G>::BYTES bytes that belong to this wave alone, G lanes work at a time (slot = the lane's rank among
// the wave's flagged lanes), the others wait; a wave without a flagged row leaves at the first branch.  `load` hands a lane its
// row's inputs again (from global memory, or regenerated) - the float32 pass does not keep them in registers for this.  tau is
// replaced for the flagged lanes only (unclipped).
template <int N, int G>
struct MpColdLds { static constexpr int BYTES = G * MpColdSlot<N>::BYTES; };
// the wave's earlier stores are acknowledged before a lane overwrites bytes another lane of the wave has just written (the
// whole-line flush, then the re-evaluated row).  A release FENCE is the wrong tool: at agent scope it is `buffer_wbl2` - the
// whole L2 written back, ~30 us per wave that takes the branch, c2 0.066 -> 0.275 ms - and at workgroup scope it is nothing at all.
__device__ __forceinline__ void mp_wait_stores() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
// a 64-bit value every active lane holds alike, moved to scalar registers (the first active lane's copy)
__device__ __forceinline__ long mp_wave_uniform(long v) {
#if defined(__HIP_DEVICE_COMPILE__)
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long)v);
  const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)((unsigned long)v >> 32));
  return (long)(((unsigned long)hi << 32) | (unsigned long)lo);
#else
  return v;
#endif
}
constexpr int MP_COLD_G = 8;  // slots of the kernels that carry a buffer of their own (256-thread blocks: one buffer per wave)
// that buffer, declared in a kernel of BLOCK threads (float32 rows only: W = sizeof(T)), and this wave's part of it
#define MP_COLD_BUFFER(N, BLOCK, W) \
  __shared__ __attribute__((aligned(16))) char mp_cold_lds[(BLOCK) / 64][(W) == 4 ? MpColdLds<N, MP_COLD_G>::BYTES : 16]
#define MP_COLD_PTR (mp_cold_lds[threadIdx.x >> 6])

// Hand the wave's flagged rows to the float64 pass that follows this kernel on the stream: one atomic per wave reserves the
// places, every flagged lane writes its row index.  False (nothing handed over) without a list.  A FULL list counts as handed
// over: the count then exceeds the capacity, which tells the pass to evaluate every row of the launch (mp_body_id_hard).
__device__ __forceinline__ bool mp_push_hard_rows(const MpCall<float>& C, unsigned long long mask, int rank, bool hard, long row) {
  if (C.hard_rows == nullptr) return false;                   // wave-uniform (kernel argument)
  const unsigned n = (unsigned)__builtin_popcountll(mask);
  unsigned base = 0;
  if (rank == 0 && hard) base = atomicAdd(C.hard_ctrl, n);  // the first flagged lane
  base = __builtin_amdgcn_readlane(base, (int)__builtin_ctzll(mask));
  if (base + n <= C.hard_cap && hard) C.hard_rows[base + (unsigned)rank] = C.hard_row_base + (unsigned)row;
  return true;                                                // (full: the count has overshot, the pass takes every row)
}

// True (wave-uniform) when flagged rows were re-evaluated HERE - tau of the flagged lanes then holds the float64 result; false when
// there were none or they were handed to the float64 pass (`row`: the lane's row index for that list; < 0: never hand over).
template <int N, bool HAS_FTIP, int G, typename MT, typename LoadFn>
__device__ __forceinline__ bool mp_cold_rows(const MT& M, const MpCall<float>& C, bool hard, long row, char* __restrict__ lds, LoadFn load,
                                             float (&tau)[N]) {
#if MP_ADAPTIVE_F32 && defined(__HIP_DEVICE_COMPILE__)
  const unsigned long long mask = __builtin_amdgcn_ballot_w64(hard);
  if (__builtin_expect(mask == 0ull, 1)) return false;  // wave-uniform
  const int rank = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
  // handed over only if EVERY flagged lane of the wave has a row index to hand over (wave-uniform decision)
  if (__builtin_amdgcn_ballot_w64(hard && row < 0) == 0ull && mp_push_hard_rows(C, mask, rank, hard, row)) return false;
  const int total = __builtin_popcountll(mask);
  for (int base = 0; base < total; base += G) {  // wave-uniform trip count
    const int slot = rank - base;
    if (hard && slot >= 0 && slot < G) {
      float q[N], qd[N], qdd[N];
      load(q, qd, qdd);
      MpColdMem<N> st{(MP_LDS_AS char*)(lds + slot * MpColdSlot<N>::BYTES)};
#if defined(MP_COLD_MODEL)   // robot-specialised program: the float64 literal of the same robot
      mp_rnea_cold<N, HAS_FTIP>(MP_COLD_MODEL, C, q, qd, qdd, st, tau);
#else                        // generic kernels: the float64 model in device memory (scalar loads), or the float32 one widened
      if (C.cold_model) mp_rnea_cold<N, HAS_FTIP>(*(MpModelConstD*)C.cold_model, C, q, qd, qdd, st, tau);
      else mp_rnea_cold<N, HAS_FTIP>(M, C, q, qd, qdd, st, tau);
#endif
    }
  }
  return true;
#else
  return false;
#endif
}

// The float64 pass over the rows a float32 kernel handed over: one row per lane, everything in float64 from the row's float32
// inputs (`Mc`: the float64 model; `M`: the float32 one, for the torque limits the float32 kernels clip against), unrolled - this
// kernel is register-allocated on its own (~170 VGPRs), which is the point of making it one.  `load(row, q, qd, qdd)` fetches or
// regenerates a row's inputs.  *hard_ctrl is the count, *hard_next the list's other counter.
// `first` / `stride` / `leader`: which list entries this lane takes and whether it is the one that zeroes the other counter - the
// stand-alone pass: its place in the grid; the float32 kernel's leading workgroups (mp_body_id_lead): their place among those.
template <int N, bool HAS_FTIP, typename MC, typename MF, typename LoadFn>
__device__ __forceinline__ void mp_body_id_hard(const MC& Mc, const MF& M, const MpCall<float>& C, LoadFn load, float* __restrict__ tau,
                                                unsigned rows, unsigned first, unsigned stride, bool leader) {
  // A list that overflowed (more than one row in eight AND more than 65 536 of a launch ill-conditioned: an arm balanced upright for
  // a whole trajectory) holds an arbitrary subset - which waves found room depends on their order - and the rest were re-evaluated in
  // place by rolled code whose last bits differ from this pass's.  So that the launch's result does not depend on the order of its
  // waves, the pass then evaluates EVERY row of the launch: float64 throughout, three times the float32 kernel's time, same bits
  // every time (tests/test_gpu_parity.py, test_more_ill_conditioned_rows_than_the_list_holds).
  const unsigned count = C.hard_ctrl[0];
  const bool all = count > C.hard_cap;
  const unsigned n = all ? rows : count;
  for (unsigned k = first; k < n; k += stride) {
    const long r = all ? (long)k : (long)C.hard_rows[k];
    if (r >= (long)rows) continue;  // (a list left behind by a launch whose pass never ran)
    float q[N], qd[N], qdd[N];
    load(r, q, qd, qdd);
    if (all) {
      // every row of the launch arrives here, also those the float32 kernel never listed because an input is NaN / inf: their
      // NaN rows stay as that kernel stored them (this code is compiled with -ffinite-math-only in the specialised programs, and the
      // torque clip would turn a NaN into a limit)
      MpBad<float> bad;
      bad.add(q); bad.add(qd); bad.add(qdd);
      if (bad.any()) continue;
    }
    double a[N], b[N], c[N], t[N];
#pragma unroll
    for (int i = 0; i < N; ++i) { a[i] = (double)q[i]; b[i] = (double)qd[i]; c[i] = (double)qdd[i]; }
    MpCall<double> Cd;
#pragma unroll
    for (int i = 0; i < 3; ++i) { Cd.a0[i] = (double)C.a0[i]; Cd.F1n[i] = (double)C.F1n[i]; Cd.F1f[i] = (double)C.F1f[i]; }
    MpJointState<double, N> js;
    mp_joint_state<double, N>(Mc, a, js);
    mp_rnea<double, N, HAS_FTIP>(Mc, Cd, js, b, c, t);
    float out[N];
#pragma unroll
    for (int i = 0; i < N; ++i) out[i] = mp_clip_tau((float)t[i], M.taumin[i], M.taumax[i]);
    RunIO<float, N>::store(tau, r, out);
  }
  // the OTHER counter of the list (its next user's): zeroed here, by the pass that runs between that counter's previous reader
  // and its next writers on the stream.  (Resetting this launch's own counter needs "every block has read it": one atomic per
  // block on one address - 1024 of them measured 20 us per launch.)
  if (leader) *C.hard_next = 0;
}
template <int N, bool HAS_FTIP, typename MC, typename MF, typename LoadFn>
__device__ __forceinline__ void mp_body_id_hard(const MC& Mc, const MF& M, const MpCall<float>& C, LoadFn load, float* __restrict__ tau,
                                                unsigned rows) {
  mp_body_id_hard<N, HAS_FTIP>(Mc, M, C, load, tau, rows, blockIdx.x * blockDim.x + threadIdx.x, gridDim.x * blockDim.x,
                               blockIdx.x == 0 && threadIdx.x == 0);
}
// The float64 pass of an EARLIER float32 launch, worked off by the first L.blocks workgroups of a later one (round 5): a pass of its
// own costs ~4.7 us of launch and dependent loads however few rows it holds, at the front of another kernel the same rows cost their
// arithmetic only.  The host hands a launch's list to a later launch only when that launch touches none of its arrays (mp_capi.cpp,
// launch_id / mp_traj_id_fused_f32).  The unrolled float64 recursion wants 120 - 170 VGPRs: the fused generation + inverse dynamics
// kernel, two waves per SIMD by design, has them; the given-rows kernel mp_spec_id_co is held to its five waves and lets this path
// spill to scratch instead (csrc/mp_jit.cpp) - the float32 rows keep their registers, their waves and scratch-free code.
template <int N, bool HAS_FTIP, typename MC, typename MF>
__device__ __forceinline__ void mp_body_id_lead(const MC& Mc, const MF& M, const MpLead& L) {
  const float* __restrict__ q = L.q; const float* __restrict__ qd = L.qd; const float* __restrict__ qdd = L.qdd;
  mp_body_id_hard<N, HAS_FTIP>(Mc, M, L.C, [&](long r, float (&x)[N], float (&y)[N], float (&z)[N]) {
    RunIO<float, N>::load(q, r, x); RunIO<float, N>::load(qd, r, y); RunIO<float, N>::load(qdd, r, z); }, L.tau, L.rows,
    blockIdx.x * blockDim.x + threadIdx.x, L.blocks * blockDim.x, blockIdx.x == 0 && threadIdx.x == 0);
}

// ... and of a FUSED launch (generation + inverse dynamics): the listed rows are regenerated from the earlier launch's start / end
// points (L.q / L.qd) and the context's time table (L.qdd: three doubles per timestep, L.nt timesteps per trajectory), exactly as
// that launch's float32 kernel generated them
template <int N, typename MF>
__device__ __forceinline__ void mp_regen_row(const MF& M, const float* __restrict__ start, const float* __restrict__ end,
                                             const double* __restrict__ tab, unsigned Nt, long r, float (&x)[N], float (&y)[N], float (&z)[N]) {
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
}
template <int N, bool HAS_FTIP, typename MC, typename MF>
__device__ __forceinline__ void mp_body_traj_id_lead(const MC& Mc, const MF& M, const MpLead& L) {
  const float* __restrict__ start = L.q; const float* __restrict__ end = L.qd;
  const double* __restrict__ tab = (const double*)L.qdd;
  const unsigned Nt = L.nt;
  mp_body_id_hard<N, HAS_FTIP>(Mc, M, L.C, [&](long r, float (&x)[N], float (&y)[N], float (&z)[N]) {
    mp_regen_row<N>(M, start, end, tab, Nt, r, x, y, z); }, L.tau, L.rows,
    blockIdx.x * blockDim.x + threadIdx.x, L.blocks * blockDim.x, blockIdx.x == 0 && threadIdx.x == 0);
}

// ------------------------------------------------------------------ one row per lane (float / double)
// tau for row `r`: the body of k_id.  `cold`: this wave's MpColdLds<N, MP_COLD_G> buffer (float rows only; unused for double).
template <typename T, int N, bool HAS_FTIP, typename MT>
__device__ __forceinline__ void mp_body_id(const MT& M, const MpCall<T>& C, const T* __restrict__ q, const T* __restrict__ qd,
                                           const T* __restrict__ qdd, T* __restrict__ tau, long r, char* __restrict__ cold = nullptr) {
  T a[N], b[N], c[N], t[N];
  RunIO<T, N>::load(q, r, a);
  RunIO<T, N>::load(qd, r, b);
  RunIO<T, N>::load(qdd, r, c);
  MpJointState<T, N> js;
  mp_joint_state<T, N>(M, a, js);
  MpBad<T> bad;  // a NaN / inf anywhere in the row's inputs -> a NaN row, as the reference returns (mp_core.h)
  bad.add(a); bad.add(b); bad.add(c);
  if constexpr (MpIsF32<T>::value) {
    const bool hard = mp_rnea_f32<N, HAS_FTIP>(M, C, js, b, c, t) && !bad.any();
    mp_cold_rows<N, HAS_FTIP, MP_COLD_G>(M, C, hard, r, cold, [&](float (&x)[N], float (&y)[N], float (&z)[N]) {
      RunIO<float, N>::load(q, r, x); RunIO<float, N>::load(qd, r, y); RunIO<float, N>::load(qdd, r, z);
    }, t);
  } else {
    mp_rnea<T, N, HAS_FTIP>(M, C, js, b, c, t);
  }
#pragma unroll
  for (int j = 0; j < N; ++j) t[j] = mp_clip_tau(t[j], M.taumin[j], M.taumax[j]);
  mp_poison_if(bad.any(), t);
  RunIO<T, N>::store(tau, r, t);
}

// The same rows moved as WHOLE LINES with non-temporal accesses: a wave moves its 64 consecutive rows of every array as flat
// 16-byte chunks (one contiguous 1.5 - 2 KB span per array), stages them in its LDS slice and each lane picks its row there;
// tau leaves the same way.  c2 0.0688 -> 0.0662-0.0673 ms, c4 0.140 -> 0.131-0.137 (tools/ab_co.sh); with plain accesses the
// same staging LOSES 3-4 % (0.071-0.072), as it did in round 1 - the gain is the non-temporal path, the staging is what
// makes it applicable.  FULL waves only: `rows64` is a multiple of 64 (the host launches the per-lane kernel on the last
// rows), `lds` is this wave's slice of MpRowStage<T, N>::BYTES bytes.
template <typename T, int N>
struct MpRowStage {
  static constexpr int ROWB = N * (int)sizeof(T), SPAN = 64 * ROWB, NCH = SPAN / 16, NJ = (NCH + 63) / 64;
  static constexpr int BYTES = 3 * SPAN;
  static_assert(SPAN % 16 == 0, "64 rows are whole 16-byte chunks");
  // (round 5: lanes past the span's last chunk - n = 6: lanes 32..63 of the second instruction - load and stage the LAST chunk
  // again instead of being masked off: same line, same bytes to the same LDS address, and no exec-mask branch or zero fill around
  // three loads and three LDS writes per wave.  The stores of the flush stay masked.)
  static __device__ __forceinline__ int chunk_of(int j, int lane) {
    const int c = j * 64 + lane;
    return ((j + 1) * 64 > NCH) ? (c < NCH ? c : NCH - 1) : c;
  }
  static __device__ __forceinline__ void fetch(const T* __restrict__ base, long row0, int lane, mp_u4 (&buf)[NJ]) {
    const mp_u4* g = reinterpret_cast<const mp_u4*>(base + row0 * N);
#pragma unroll
    for (int j = 0; j < NJ; ++j)
      buf[j] = mp_stream_load(g + chunk_of(j, lane));
  }
  static __device__ __forceinline__ void stage(const mp_u4 (&buf)[NJ], int lane, char* __restrict__ region) {
#pragma unroll
    for (int j = 0; j < NJ; ++j)
      *reinterpret_cast<mp_u4*>(region + chunk_of(j, lane) * 16) = buf[j];
  }
  static __device__ __forceinline__ void row_in(const char* __restrict__ region, int lane, T (&v)[N]) {
    if constexpr (sizeof(T) == 8 && ROWB % 16 != 0) {
      // 8-byte values in rows that are not whole 16-byte chunks (n = 7 float64: 56-byte rows, stride 14 dwords): left alone the
      // loads are merged into ds_read_b128, which serves 8 lanes per cycle - lanes 0 and 7 then meet in banks 2..3 (14 * 7 = 98 = 2
      // mod 32): SQ_LDS_BANK_CONFLICT 69.6 M cycles per c3 launch in round 3.  As separate 8-byte reads 16 lanes are served per
      // cycle and 14 l mod 32 is distinct for l = 0..15: none.  (volatile = do not merge; the address space is spelled out because
      // a volatile access through a generic pointer is emitted as a FLAT load)
      const volatile MP_LDS_AS T* src = (const volatile MP_LDS_AS T*)(region + lane * ROWB);
#pragma unroll
      for (int j = 0; j < N; ++j) v[j] = src[j];
    } else {
      const T* src = reinterpret_cast<const T*>(region + lane * ROWB);
#pragma unroll
      for (int j = 0; j < N; ++j) v[j] = src[j];
    }
  }
  static __device__ __forceinline__ void row_out(char* __restrict__ region, int lane, const T (&v)[N]) {
    T* dst = reinterpret_cast<T*>(region + lane * ROWB);
#pragma unroll
    for (int j = 0; j < N; ++j) dst[j] = v[j];
  }
  static __device__ __forceinline__ void flush(T* __restrict__ base, long row0, int lane, const char* __restrict__ region) {
    mp_u4* g = reinterpret_cast<mp_u4*>(base + row0 * N);
#pragma unroll
    for (int j = 0; j < NJ; ++j)
      if (j * 64 + lane < NCH) mp_stream_store(*reinterpret_cast<const mp_u4*>(region + (j * 64 + lane) * 16), g + j * 64 + lane);
  }
  static __device__ __forceinline__ void sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
};

template <typename T, int N, bool HAS_FTIP, typename MT>
__device__ __forceinline__ void mp_body_id_co(const MT& M, const MpCall<T>& C, const T* __restrict__ q, const T* __restrict__ qd,
                                              const T* __restrict__ qdd, T* __restrict__ tau, long row0, int lane, char* __restrict__ lds) {
  using ST = MpRowStage<T, N>;
  {
    mp_u4 bq[ST::NJ], bd[ST::NJ], ba[ST::NJ];
    ST::fetch(q, row0, lane, bq);
    ST::fetch(qd, row0, lane, bd);
    ST::fetch(qdd, row0, lane, ba);
    ST::stage(bq, lane, lds);
    ST::stage(bd, lane, lds + ST::SPAN);
    ST::stage(ba, lane, lds + 2 * ST::SPAN);
  }
  ST::sync();
  T a[N], b[N], c[N], t[N];
  ST::row_in(lds, lane, a);
  ST::row_in(lds + ST::SPAN, lane, b);
  ST::row_in(lds + 2 * ST::SPAN, lane, c);
  MpJointState<T, N> js;
  mp_joint_state<T, N>(M, a, js);
  MpBad<T> bad;  // the non-finite row contract of mp_body_id
  bad.add(a); bad.add(b); bad.add(c);
  const bool poison = bad.any();
  bool hard = false;
  if constexpr (MpIsF32<T>::value) hard = mp_rnea_f32<N, HAS_FTIP>(M, C, js, b, c, t) && !poison;
  else mp_rnea<T, N, HAS_FTIP>(M, C, js, b, c, t);
#pragma unroll
  for (int j = 0; j < N; ++j) t[j] = mp_clip_tau(t[j], M.taumin[j], M.taumax[j]);
  mp_poison_if(poison, t);
  ST::sync();  // every lane has read its rows: the first region is free
  ST::row_out(lds, lane, t);
  ST::sync();
  ST::flush(tau, row0, lane, lds);
  if constexpr (MpIsF32<T>::value) {
    // ill-conditioned rows again, in float64, AFTER the wave's float32 rows have left: the whole slice is free for their state
    // (12 lanes at a time), the inputs come back from memory and the result overwrites the row's 4 N bytes.  Rare (a few per
    // cent of the waves take the branch at all), so the order of the two stores to the same bytes is made explicit.
    if (__builtin_amdgcn_ballot_w64(hard) != 0ull) {
      ST::sync();
      mp_wait_stores();
      constexpr int G = ST::BYTES / MpColdSlot<N>::BYTES;
      static_assert(G >= 8, "the wave's row slice holds at least eight re-evaluation slots");
      // the row index is made opaque here: otherwise the per-lane addresses below are computed once at the top of the kernel,
      // shared with the fetch, and kept alive (or spilled) across the whole float32 pass for the sake of this branch
      long rr = row0 + lane;
      asm volatile("" : "+v"(rr));
      const bool here = mp_cold_rows<N, HAS_FTIP, G>(M, C, hard, rr, lds, [&](float (&x)[N], float (&y)[N], float (&z)[N]) {
        RunIO<float, N>::load(q, rr, x); RunIO<float, N>::load(qd, rr, y); RunIO<float, N>::load(qdd, rr, z);
      }, t);
      if (here && hard) {
#pragma unroll
        for (int j = 0; j < N; ++j) t[j] = mp_clip_tau(t[j], M.taumin[j], M.taumax[j]);
        RunIO<float, N>::store(tau, rr, t);
      }
    }
  }
}

// qdd for row `r` = forward_dynamics(q, qd, tau, g, F) with one wrench for every row: the body of k_forward_dynamics
template <typename T, int N, bool HAS_FTIP, typename MT>
__device__ __forceinline__ void mp_body_fd(const MT& M, const MpCall<T>& C, const T* __restrict__ q, const T* __restrict__ qd,
                                           const T* __restrict__ tau, T* __restrict__ qdd, long r) {
  T a[N], b[N], t[N], out[N];
  RunIO<T, N>::load(q, r, a);
  RunIO<T, N>::load(qd, r, b);
  RunIO<T, N>::load(tau, r, t);
  const T tn[3] = {C.F1n[0], C.F1n[1], C.F1n[2]}, tf[3] = {C.F1f[0], C.F1f[1], C.F1f[2]};
  mp_forward_dynamics<T, N, HAS_FTIP>(M, C.a0, tn, tf, a, b, t, out);
  MpBad<T> bad;
  bad.add(a); bad.add(b); bad.add(t);
  mp_poison_if(bad.any(), out);
  RunIO<T, N>::store(qdd, r, out);
}

// ------------------------------------------------------------- wave-cooperative coalesced stores
// A lane that owns a long output run (T: 128 B, J: 336 B at n = 7, float64) and stores it directly issues
// 16-byte stores that are 128 / 336 bytes apart across lanes: every store instruction touches 64 different
// cache lines.  Measured (tools/ubench_mem_c3.hip): that pattern tops out at 4.0 TB/s where the same bytes
// written as contiguous kilobytes reach 5.4 TB/s.  Here each wave stages PC chunks per row in its own slice of
// LDS (row pitch padded to a multiple of 128 B plus one chunk, so 8 consecutive lanes cover all 32 banks) and
// reads them back in flat order: one store instruction then writes 64 / PC row segments of PC * W contiguous
// bytes (T: eight full 128-byte lines).  Rows of one wave are consecutive, so `row0` is the wave's first row.
constexpr int MP_WAVE_LDS_BYTES = 64 * 144;  // per wave: 64 rows x (<= 128 B of payload + 16 B pad)

constexpr int mp_piece_chunks(int ch) {  // largest divisor of `ch` that is <= 8
  int best = 1;
  for (int d = 1; d <= 8; ++d)
    if (ch % d == 0) best = d;
  return best;
}

template <typename T, int COUNT>
__device__ __forceinline__ void mp_wave_store(T* __restrict__ gbase, long row0, int lane, int nvalid, const T (&v)[COUNT],
                                              char* __restrict__ lds) {
  using IO = RunIO<T, COUNT>;
  using V = typename IO::V;
  constexpr int W = IO::W, K = IO::K, CH = COUNT / K, PC = mp_piece_chunks(CH), NP = CH / PC;
  constexpr int PITCH = ((PC * W + 127) / 128) * 128 + W;
  static_assert(64 * PITCH <= MP_WAVE_LDS_BYTES, "wave staging slice too small");
  V* gout = reinterpret_cast<V*>(gbase + row0 * COUNT);
#pragma unroll
  for (int p = 0; p < NP; ++p) {
#pragma unroll
    for (int c = 0; c < PC; ++c) {
      union { V vec; T e[K]; } u;
#pragma unroll
      for (int j = 0; j < K; ++j) u.e[j] = v[(p * PC + c) * K + j];
      *reinterpret_cast<V*>(lds + lane * PITCH + c * W) = u.vec;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int j = 0; j < PC; ++j) {
      const int g = j * 64 + lane;  // flat chunk index inside this piece of the wave's 64 rows
      const int row = g / PC, col = g - row * PC;
      const V val = *reinterpret_cast<const V*>(lds + row * PITCH + col * W);
      if (row < nvalid) mp_coop_store(val, gout + ((long)row * CH + p * PC + col));
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
}

// Same idea for rows whose size is NOT a multiple of the 128-byte line (the Jacobian: 336 B at n = 7, float64):
// the wave's 64 rows are one contiguous span, so it is written in FLAT chunk order — 16 rows (a whole number of
// lines: 16 * COUNT * sizeof(T) is always a multiple of 128 here) are staged per pass by the 16 lanes that own
// them and all 64 lanes then stream them out as consecutive 16-byte chunks: every store instruction covers one
// contiguous kilobyte and no line is written twice.  (The per-piece scheme above wrote 112-byte segments at a
// 336-byte pitch: PMC showed 14 % more bytes written than produced.)
template <typename T, int COUNT>
__device__ __forceinline__ void mp_wave_store_flat(T* __restrict__ gbase, long row0, int lane, int nvalid, const T (&v)[COUNT],
                                                   char* __restrict__ lds) {
  using IO = RunIO<T, COUNT>;
  using V = typename IO::V;
  constexpr int W = IO::W, K = IO::K, CH = COUNT / K, ROWS = 16;
  // Row pitch in the staging slice.  In 16-byte chunks: the 8 lanes a ds_write_b128 serves per cycle (one row each) hit distinct
  // bank groups iff the pitch is ODD, and the 8 consecutive flat chunks a ds_read_b128 serves per cycle stay distinct across a
  // row boundary iff the padding is a multiple of 8.  An odd row (the 6 x 7 float64 Jacobian: 21 chunks) needs no padding at all;
  // round 1's rule (next multiple of 128 bytes + one chunk: 25 chunks, padding 4) was conflict-free on the writes and collided on
  // every read group that crossed a row end: SQ_LDS_BANK_CONFLICT 69.6 M cycles per c3 launch, 14 % of the LDS cycles.
  // (the A/B against round 1's padding: profiles/r04_c3_lds_ab.txt)
  constexpr int PITCH = (W == 16 && CH % 2 == 1) ? CH * W : ((CH * W + 127) / 128) * 128 + W;
  static_assert(ROWS * PITCH <= MP_WAVE_LDS_BYTES, "wave staging slice too small");
  constexpr int TOTAL = ROWS * CH, NJ = (TOTAL + 63) / 64;
  V* gout = reinterpret_cast<V*>(gbase + row0 * COUNT);
#pragma unroll
  for (int pass = 0; pass < 64 / ROWS; ++pass) {
    if ((lane >> 4) == pass) {
#pragma unroll
      for (int c = 0; c < CH; ++c) {
        union { V vec; T e[K]; } u;
#pragma unroll
        for (int j = 0; j < K; ++j) u.e[j] = v[c * K + j];
        *reinterpret_cast<V*>(lds + (lane & (ROWS - 1)) * PITCH + c * W) = u.vec;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int f = j * 64 + lane;  // flat chunk index inside this pass's 16 rows
      if (f < TOTAL) {
        const int row = f / CH, col = f - row * CH;
        const V val = *reinterpret_cast<const V*>(lds + row * PITCH + col * W);
        if (pass * ROWS + row < nvalid) mp_coop_store(val, gout + ((long)pass * TOTAL + f));
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
}

// Rows that are not even a whole number of 16-byte chunks (an odd element count: the 7 x 7 mass matrix is 49 values, 196 / 392
// bytes): chunking per row would leave 4- / 8-byte stores (256 / 512 bytes per store instruction: 4.0 TB/s measured on the
// iiwa14 mass matrix where the 6 x 6 one reaches 5.7).  16 rows always ARE a whole number of 16-byte chunks, so they are staged
// back to back (unpadded: an odd dword pitch spreads the 16 staging lanes over distinct banks by itself) and streamed out as
// 16-byte chunks that ignore the row boundaries; only the chunk that straddles the end of the valid rows is written by element.
template <typename T, int COUNT>
__device__ __forceinline__ void mp_wave_store_flat16(T* __restrict__ gbase, long row0, int lane, int nvalid, const T (&v)[COUNT],
                                                     char* __restrict__ lds) {
  constexpr int ROWS = 16, ROWB = COUNT * (int)sizeof(T), SPAN = ROWS * ROWB, NCH = SPAN / 16, NJ = (NCH + 63) / 64;
  static_assert(SPAN % 16 == 0 && SPAN <= MP_WAVE_LDS_BYTES, "16 rows must be whole 16-byte chunks and fit the wave's staging slice");
  char* gout = reinterpret_cast<char*>(gbase + row0 * COUNT);  // 64 rows per wave: a multiple of 16 bytes from a 16-byte-aligned base
  const int valid_bytes = nvalid * ROWB;
#pragma unroll
  for (int pass = 0; pass < 64 / ROWS; ++pass) {
    if ((lane >> 4) == pass) {
      T* dst = reinterpret_cast<T*>(lds + (lane & (ROWS - 1)) * ROWB);
#pragma unroll
      for (int e = 0; e < COUNT; ++e) dst[e] = v[e];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int f = j * 64 + lane;  // flat 16-byte chunk inside this pass's 16 rows
      if (f < NCH) {
        const int gofs = pass * SPAN + f * 16;
        if (gofs + 16 <= valid_bytes) {
          mp_coop_store(*reinterpret_cast<const mp_u4*>(lds + f * 16), reinterpret_cast<mp_u4*>(gout + gofs));
        } else if (gofs < valid_bytes) {  // the one chunk across the end of the valid rows (last wave only)
#pragma unroll
          for (int b = 0; b < 16; b += (int)sizeof(T))
            if (gofs + b < valid_bytes) *reinterpret_cast<T*>(gout + gofs + b) = *reinterpret_cast<const T*>(lds + f * 16 + b);
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
}

// rows that are whole lines go out per piece, rows of whole 16-byte chunks in flat chunk order, odd rows as flat 16-byte chunks
template <typename T, int COUNT>
__device__ __forceinline__ void mp_wave_store_auto(T* __restrict__ gbase, long row0, int lane, int nvalid, const T (&v)[COUNT],
                                                   char* __restrict__ lds) {
  if constexpr ((COUNT * (int)sizeof(T)) % 128 == 0) mp_wave_store<T, COUNT>(gbase, row0, lane, nvalid, v, lds);
  else if constexpr ((COUNT * (int)sizeof(T)) % 16 == 0) mp_wave_store_flat<T, COUNT>(gbase, row0, lane, nvalid, v, lds);
  else mp_wave_store_flat16<T, COUNT>(gbase, row0, lane, nvalid, v, lds);
}

// T (4x4), space Jacobian (6xN) and tau for row `r`, any output optional: the body of k_fk_jac_id.
// EVERY lane of the wave must call this (the T / J stores are wave-cooperative); `rows` bounds the valid rows,
// `lds` is this wave's MP_WAVE_LDS_BYTES staging slice.
template <typename T, int N, bool HAS_FTIP, typename MT>
__device__ __forceinline__ void mp_body_fk_jac_id(const MT& M, const MpCall<T>& C, const T* __restrict__ q,
                                                  const T* __restrict__ qd, const T* __restrict__ qdd, T* __restrict__ Tout,
                                                  T* __restrict__ Jout, T* __restrict__ tau, long r, long rows,
                                                  char* __restrict__ lds) {
  using ST = MpRowStage<T, N>;
  static_assert(ST::SPAN <= MP_WAVE_LDS_BYTES, "one array's 64 rows fit the wave's staging slice");
  const int lane = (int)(threadIdx.x & 63);
  // the wave's first row, SAID to be wave-uniform (lane 0's r: every lane of the wave is active here): the staging and flat-store
  // addresses are then formed on the scalar unit (round 5: the same change took 2.4 us off the c2 kernel, profiles/r05_ab_h.txt)
  const long row0 = mp_wave_uniform(r - lane);
  if (row0 >= rows) return;  // whole wave out of range (wave-uniform)
  const bool valid = r < rows;
  const long rr = valid ? r : rows - 1;  // out-of-range lanes recompute the last row and store nothing
  const long left = rows - row0;
  const int nvalid = left < 64 ? (int)left : 64;
  // A full wave moves its input rows and tau as whole lines, non-temporal (MpRowStage; see mp_body_id_co): all three arrays are
  // requested up front, each is staged through the slice when its values are needed.  The last, partial wave: per-lane rows.
  const bool full = nvalid == 64;
  mp_u4 bq[ST::NJ], bd[ST::NJ], ba[ST::NJ];
  if (full) {
    ST::fetch(q, row0, lane, bq);
    if (tau != nullptr) {
      ST::fetch(qd, row0, lane, bd);
      ST::fetch(qdd, row0, lane, ba);
    }
  }
  T a[N];
  if (full) {
    ST::stage(bq, lane, lds);
    ST::sync();
    ST::row_in(lds, lane, a);
    ST::sync();
  } else {
    RunIO<T, N>::load(q, rr, a);
  }
  MpJointState<T, N> js;
  mp_joint_state<T, N>(M, a, js);
  MpBad<T> bad;
  bad.add(a);
  if (Tout != nullptr || Jout != nullptr) {
    T TT[16], JJ[6 * N];
    mp_fk_jac<T, N, true>(M, js, TT, JJ);
    mp_poison_if(bad.any(), TT);
    mp_poison_if(bad.any(), JJ);
    if (Tout != nullptr) mp_wave_store_auto<T, 16>(Tout, row0, lane, nvalid, TT, lds);
    if (Jout != nullptr) mp_wave_store_auto<T, 6 * N>(Jout, row0, lane, nvalid, JJ, lds);
  }
  if (tau != nullptr) {
    T b[N], c[N], t[N];
    if (full) {
      ST::stage(bd, lane, lds);
      ST::sync();
      ST::row_in(lds, lane, b);
      ST::sync();
      ST::stage(ba, lane, lds);
      ST::sync();
      ST::row_in(lds, lane, c);
      ST::sync();
    } else {
      RunIO<T, N>::load(qd, rr, b);
      RunIO<T, N>::load(qdd, rr, c);
    }
    bad.add(b); bad.add(c);
    bool hard = false;
    if constexpr (MpIsF32<T>::value) hard = mp_rnea_f32<N, HAS_FTIP>(M, C, js, b, c, t) && !bad.any() && valid;
    else mp_rnea<T, N, HAS_FTIP>(M, C, js, b, c, t);
#pragma unroll
    for (int j = 0; j < N; ++j) t[j] = mp_clip_tau(t[j], M.taumin[j], M.taumax[j]);
    mp_poison_if(bad.any(), t);
    if (full) {
      ST::row_out(lds, lane, t);
      ST::sync();
      ST::flush(tau, row0, lane, lds);
    } else if (valid) {
      RunIO<T, N>::store(tau, r, t);
    }
    if constexpr (MpIsF32<T>::value) {  // ill-conditioned float32 rows again in float64, after the wave's rows have left (see mp_body_id_co)
      if (__builtin_amdgcn_ballot_w64(hard) != 0ull) {
        ST::sync();
        mp_wait_stores();
        constexpr int G = MP_WAVE_LDS_BYTES / MpColdSlot<N>::BYTES;
        const bool here = mp_cold_rows<N, HAS_FTIP, G>(M, C, hard, -1L, lds, [&](float (&x)[N], float (&y)[N], float (&z)[N]) {
          RunIO<float, N>::load(q, rr, x); RunIO<float, N>::load(qd, rr, y); RunIO<float, N>::load(qdd, rr, z);
        }, t);
        if (here && hard) {
#pragma unroll
          for (int j = 0; j < N; ++j) t[j] = mp_clip_tau(t[j], M.taumin[j], M.taumax[j]);
          RunIO<float, N>::store(tau, r, t);
        }
      }
    }
  }
}

// ------------------------------------------------- float32, two rows per lane (packed v_pk_* math)
// the packed recursion of a lane's two rows + the float64 re-evaluation of whichever of the two needs it (`cold`: this wave's
// MpColdLds<N, MP_COLD_G> buffer; the inputs are still in the lane's registers here)
template <int N, bool HAS_FTIP, typename MT>
__device__ __forceinline__ void mp_rnea_pk(const MT& M, const MpCall<float>& C, const MpJointState<mp_f2, N>& js, const mp_f2 (&q)[N],
                                           const mp_f2 (&qd)[N], const mp_f2 (&qdd)[N], mp_f2 (&tau)[N], const MpBad<mp_f2>& bad,
                                           char* __restrict__ cold, long rowx = -1, long rowy = -1, bool usey = true) {
#if MP_ADAPTIVE_F32
  const mp_f2 tn[3] = {(mp_f2)(C.F1n[0]), (mp_f2)(C.F1n[1]), (mp_f2)(C.F1n[2])};
  const mp_f2 tf[3] = {(mp_f2)(C.F1f[0]), (mp_f2)(C.F1f[1]), (mp_f2)(C.F1f[2])};
  MpRowScale<mp_f2, N> sc;
  mp_rnea_impl<mp_f2, N, HAS_FTIP>(M, C.a0, tn, tf, js, qd, qdd, tau, sc);
  const mp_f2 sF = sc.scale(M.lscale);
  float tx[N], ty[N];
#pragma unroll
  for (int i = 0; i < N; ++i) { tx[i] = tau[i].x; ty[i] = tau[i].y; }
  const bool hx = mp_id_row_is_hard<N>(tx, sF.x) && !bad.x.any();
  const bool hy = mp_id_row_is_hard<N>(ty, sF.y) && !bad.y.any() && usey;  // (usey false: a duplicate the caller drops)
  mp_cold_rows<N, HAS_FTIP, MP_COLD_G>(M, C, hx, rowx, cold, [&](float (&a)[N], float (&b)[N], float (&c)[N]) {
#pragma unroll
    for (int i = 0; i < N; ++i) { a[i] = q[i].x; b[i] = qd[i].x; c[i] = qdd[i].x; }
  }, tx);
  mp_cold_rows<N, HAS_FTIP, MP_COLD_G>(M, C, hy, rowy, cold, [&](float (&a)[N], float (&b)[N], float (&c)[N]) {
#pragma unroll
    for (int i = 0; i < N; ++i) { a[i] = q[i].y; b[i] = qd[i].y; c[i] = qdd[i].y; }
  }, ty);
#pragma unroll
  for (int i = 0; i < N; ++i) tau[i] = (mp_f2){tx[i], ty[i]};
#else
  mp_rnea<mp_f2, N, HAS_FTIP>(M, C, js, qd, qdd, tau);
#endif
}

// Generation fused into inverse dynamics, two rows per lane in packed arithmetic.  The time scaling comes from a per-call table
// (s, s', s'' per timestep, three doubles each, written by k_time_table with exactly the arithmetic of traj_row), and both rows of
// a lane lie in ONE trajectory: timesteps t0 and t1 = t0 + ceil(Nt / 2).  Against the first form (time scaling per row, adjacent
// rows: removed in round 6) a pair saves eleven float64 divisions and the 64-bit row -> (trajectory, timestep) division (293
// float64 + ~100 integer instructions of 1006), loads the end points once and stores tau in one-row-per-lane runs.  `valid1` is
// false for the unpaired middle row of an odd Nt.
template <int N, bool HAS_FTIP, typename MT>
__device__ __forceinline__ void mp_body_traj_id_pk_tab(const MT& M, const MpCall<float>& C, const float* __restrict__ start,
                                                       const float* __restrict__ end, long b, long t0, long t1, bool valid1,
                                                       long Nt, const double* __restrict__ tab, float* __restrict__ tau,
                                                       char* __restrict__ cold) {
  float a[N], e[N];
  RunIO<float, N>::load(start, b, a);
  RunIO<float, N>::load(end, b, e);
  const double s0 = tab[3 * t0], sd0 = tab[3 * t0 + 1], sdd0 = tab[3 * t0 + 2];
  const double s1 = tab[3 * t1], sd1 = tab[3 * t1 + 1], sdd1 = tab[3 * t1 + 2];
  mp_f2 qq[N], qd[N], qdd[N], tq[N];
#pragma unroll
  for (int j = 0; j < N; ++j) {
    const double d = (double)(e[j] - a[j]);  // float32 difference first, as the reference types it
    qq[j] = (mp_f2){mp_clip((float)(s0 * d + (double)a[j]), M.qmin[j], M.qmax[j]),
                    mp_clip((float)(s1 * d + (double)a[j]), M.qmin[j], M.qmax[j])};
    qd[j] = (mp_f2){(float)(sd0 * d), (float)(sd1 * d)};
    qdd[j] = (mp_f2){(float)(sdd0 * d), (float)(sdd1 * d)};
  }
  MpJointState<mp_f2, N> js;
  mp_joint_state<mp_f2, N>(M, qq, js);
  MpBad<mp_f2> bad;  // a non-finite end point makes the generated row non-finite
  bad.add(qq); bad.add(qd); bad.add(qdd);
  mp_rnea_pk<N, HAS_FTIP>(M, C, js, qq, qd, qdd, tq, bad, cold, b * Nt + t0, b * Nt + t1, valid1);
  float lo[N], hi[N];
#pragma unroll
  for (int j = 0; j < N; ++j) {
    const mp_f2 c = mp_clip_tau(tq[j], M.taumin[j], M.taumax[j]);
    lo[j] = c.x; hi[j] = c.y;
  }
  mp_poison_if(bad.x.any(), lo);
  mp_poison_if(bad.y.any(), hi);
  RunIO<float, N>::store_wo(tau, b * Nt + t0, lo);
  if (valid1) RunIO<float, N>::store_wo(tau, b * Nt + t1, hi);
}

// (Two other forms of this kernel were built, measured and removed in round 6: one timestep per lane in scalar arithmetic - c2f 0.063
// against 0.054 ms - and flat rows with tau as whole lines through LDS - exact traffic, 1486 issue cycles per row against 1334:
// 0.0553 - 0.0571 against 0.0533 ms, profiles/r04_c2f_ab.txt, profiles/HISTORY.md.)

// lane -> (trajectory, timestep pair) for the kernel above: `bpt` blocks of `block` lanes per trajectory
__device__ __forceinline__ bool mp_traj_pair(unsigned block_idx, unsigned lane, unsigned block, unsigned bpt, long Nt, long& b,
                                             long& t0, long& t1, bool& valid1) {
  const unsigned bb = block_idx / bpt;  // block-uniform: a scalar division
  const long half = (Nt + 1) / 2;
  t0 = (long)(block_idx - bb * bpt) * block + lane;
  if (t0 >= half) return false;
  b = bb;
  t1 = t0 + half;
  valid1 = t1 < Nt;
  if (!valid1) t1 = t0;  // the duplicate is computed and dropped
  return true;
}

// forward_dynamics_trajectory (reference planning/trajectory_dynamics.py:580-708): one lane integrates one
// trajectory — semi-implicit Euler, `intRes` sub-steps of dt/intRes per outer step, joint-limit clip after
// every sub-step, rows stored float32, the recorded acceleration is the last sub-step's, row 0 = initial
// state with zero acceleration.  Time is sequential; trajectories are independent.  Body of k_fd_traj.
//
// I/O tiling.  A lane's rows are Nt * N * 4 bytes apart from its neighbour's, so a per-step access touches 64
// different cache lines per array and uses N * 4 bytes of each; with >= 8 MB of such lines live per XCD the
// 4 MB L2 evicts every line before the next step reuses it (PMC, config c5: 5.6 GB moved for 0.94 GB of
// payload, the roll-out ran at the fabric's bandwidth).  Here each lane moves MP_FD_KS steps at a time: the
// torque rows of a tile are fetched as one contiguous run per lane and parked in the lane's column of the
// wave's LDS tile, the three output rows of every step are written to the same tile, and the tile leaves as
// contiguous KS * N * 4-byte runs (whole 32-byte sectors) per lane and array.  The tile is laid out
// [step][slot][joint][lane], so every LDS access is lane-contiguous (no bank conflicts) and a lane only ever
// touches its own column — no barriers, and lanes past the batch end may simply return.
constexpr int MP_FD_KS = 4;
// Columns of one step of the tile (each column = 64 lanes x one dword).  float32: [0,N) torque, overwritten by the
// position once the step has consumed it; [N,2N) velocity; [2N,3N) acceleration; the step's wrench (6 values, if
// any) sits in [N,N+6) and is likewise consumed before the step's velocity / acceleration are written.
// float64 inputs take two dwords per value and get their own columns after the three output slots.
template <typename T, int N, bool HAS_FTIP>
struct MpFdTile {
  static constexpr int W = 64;                     // trajectories per wave
  static constexpr int TW = (int)sizeof(T) / 4;  // dwords per input value
  static constexpr int TAU0 = (TW == 1) ? 0 : 3 * N;
  static constexpr int F0 = (TW == 1) ? N : 5 * N;
  static constexpr int COLS32 = (HAS_FTIP && N + 6 > 3 * N) ? N + 6 : 3 * N;
  static constexpr int COLS = (TW == 1) ? COLS32 : 5 * N + (HAS_FTIP ? 12 : 0);
  // Row stride in dwords.  A row holds one value of the W trajectories of the wave; the 64-wide tile pads it by one
  // dword: the flat stores read ACROSS rows (six lanes share a trajectory column and take different rows), which at a
  // stride of 64 dwords puts all six on one LDS bank (PMC: 60 % of the LDS cycles of config c5 were bank conflicts).
  static constexpr int RS = W + 1;                 // (conflict cycles 36.9 M -> 14.7 M per c5 launch; with the owner-lane flush: none)
  static constexpr int STEP = COLS * RS;           // dwords per step
  static constexpr int DWORDS = MP_FD_KS * STEP;   // per wave
};
typedef unsigned mp_io_u4 __attribute__((ext_vector_type(4)));

// dword `d` of a lane's contiguous input run (rows of E values of TW dwords) -> tile dword index, lane offset excluded
template <int E, int TW, int BASE, int STEP, int W>
__device__ __forceinline__ constexpr int mp_fd_in_slot(int d) {
  const int s = d / (E * TW), rem = d % (E * TW), e = rem / TW, w = rem % TW;
  return s * STEP + (BASE + w * E + e) * W;
}

// global -> tile: MP_FD_KS rows (or `limit` dwords of them when VW == 1) of one trajectory's run
template <int E, int TW, int BASE, int STEP, int W, int VW>
__device__ __forceinline__ void mp_fd_tile_in(const unsigned* __restrict__ g, int limit, unsigned* __restrict__ col) {
  constexpr int TOTAL = MP_FD_KS * E * TW;
#pragma unroll
  for (int d = 0; d < TOTAL; d += VW) {
    if constexpr (VW == 4) {
      const mp_io_u4 v = *reinterpret_cast<const mp_io_u4*>(g + d);
      col[mp_fd_in_slot<E, TW, BASE, STEP, W>(d)] = v.x;
      col[mp_fd_in_slot<E, TW, BASE, STEP, W>(d + 1)] = v.y;
      col[mp_fd_in_slot<E, TW, BASE, STEP, W>(d + 2)] = v.z;
      col[mp_fd_in_slot<E, TW, BASE, STEP, W>(d + 3)] = v.w;
    } else if (d < limit) {
      col[mp_fd_in_slot<E, TW, BASE, STEP, W>(d)] = g[d];
    }
  }
}

// The same in two halves: all the loads of a tile (both input arrays) are issued into registers before the first row is
// parked, so a tile's inputs cost ONE memory round trip.
template <int E, int TW>
struct MpFdPrefetch {
  static constexpr int NV = MP_FD_KS * E * TW / 4;  // 16-byte vectors per tile and lane
  mp_io_u4 v[NV];
};
template <int E, int TW>
__device__ __forceinline__ void mp_fd_tile_load(const unsigned* __restrict__ g, MpFdPrefetch<E, TW>& r) {
#pragma unroll
  for (int k = 0; k < MpFdPrefetch<E, TW>::NV; ++k) r.v[k] = *reinterpret_cast<const mp_io_u4*>(g + 4 * k);
}
template <int E, int TW, int BASE, int STEP, int W>
__device__ __forceinline__ void mp_fd_tile_park(const MpFdPrefetch<E, TW>& r, unsigned* __restrict__ col) {
#pragma unroll
  for (int k = 0; k < MpFdPrefetch<E, TW>::NV; ++k) {
    col[mp_fd_in_slot<E, TW, BASE, STEP, W>(4 * k)] = r.v[k].x;
    col[mp_fd_in_slot<E, TW, BASE, STEP, W>(4 * k + 1)] = r.v[k].y;
    col[mp_fd_in_slot<E, TW, BASE, STEP, W>(4 * k + 2)] = r.v[k].z;
    col[mp_fd_in_slot<E, TW, BASE, STEP, W>(4 * k + 3)] = r.v[k].w;
  }
}

// tile -> global: output slot `slot` (0 pos, 1 vel, 2 acc) of MP_FD_KS rows (or `limit` dwords when VW == 1)
template <int N, int STEP, int W, int VW>
__device__ __forceinline__ void mp_fd_tile_out(float* __restrict__ gdst, int slot, int limit, const unsigned* __restrict__ col) {
  constexpr int TOTAL = MP_FD_KS * N;
  unsigned* g = reinterpret_cast<unsigned*>(gdst);
#pragma unroll
  for (int d = 0; d < TOTAL; d += VW) {
    if constexpr (VW == 4) {
      mp_io_u4 v;
      v.x = col[((d) / N) * STEP + (slot * N + (d) % N) * W];
      v.y = col[((d + 1) / N) * STEP + (slot * N + (d + 1) % N) * W];
      v.z = col[((d + 2) / N) * STEP + (slot * N + (d + 2) % N) * W];
      v.w = col[((d + 3) / N) * STEP + (slot * N + (d + 3) % N) * W];
      *reinterpret_cast<mp_io_u4*>(g + d) = v;
    } else if (d < limit) {
      g[d] = col[(d / N) * STEP + (slot * N + d % N) * W];
    }
  }
}

// tile -> global, WAVE-COOPERATIVE: the three output arrays of a whole tile leave in flat chunk order.  A trajectory's
// rows of the tile are one contiguous run of MP_FD_KS * N floats per array (96 bytes at n = 6) and the runs of neighbouring
// lanes are Nt * N * 4 bytes apart; stored lane by lane (mp_fd_tile_out) every store instruction scatters 64 16-byte pieces
// over 64 different lines and each line is assembled from six instructions - measured, that output path cost 0.22 ms of
// the 0.65 ms of config c5 (a build without it: 0.37 ms; without any tile I/O: 0.30 ms).  Here chunk f = k * 64 + lane of
// the wave's 64 runs belongs to trajectory f / C, 16-byte piece f % C (C = pieces per run), so consecutive lanes write
// consecutive 16-byte pieces of one run: a store instruction covers ~11 whole runs instead of 64 fragments.  The chunk ->
// (trajectory, piece) split is the same for every tile and array: it is worked out once (`MpFdFlat`).
template <int N>
struct MpFdFlat {
  static constexpr int C = MP_FD_KS * N / 4;      // 16-byte pieces per run (MP_FD_KS = 4: N of them)
  static constexpr int NK = C;                     // store instructions per array: 64 * C chunks / 64 lanes
  int lane;
  // trajectory (lane column) and piece of this lane's chunk k - recomputed where used (a division by a constant) rather
  // than carried in 2 NK registers through the integration steps
  __device__ __forceinline__ int t(int k) const { return (int)((unsigned)(k * 64 + lane) / (unsigned)C); }
  __device__ __forceinline__ int c(int k) const { return (k * 64 + lane) - t(k) * C; }
};
// `lds` = the wave's tile base (lane offset NOT applied), `run0` = float index of (trajectory b0, step i0, joint 0),
// `pitch` = floats between the runs of neighbouring trajectories (Nt * N), `nvalid` = trajectories of this wave inside the
// batch.  An array's LDS reads are issued before its first store (every chunk's column exists: f < 64 C gives t < 64), so the
// wave waits for the LDS once per array and tile, and a whole wave stores without per-chunk branches.
// Whole blocks (MP_FD_BLOCK bytes: 128 = whole lines, the default; 64 = half lines).  A trajectory's rows of one tile are a run
// of MP_FD_KS * N * 4 bytes (96 at n = 6) that starts where the previous tile's run ended, so most runs end inside a block:
// written as it comes, the block reaches the L2 in two parts four steps (~20 us) apart, is evicted in between (one open
// block per trajectory and array is far more than the L2s hold) and goes to memory as two partial writes.
// tools/ubench_c5io.hip, the kernel's tile I/O around a stand-in for the arithmetic: 0.589 ms written as it comes, 0.472 ms
// in whole 64-byte blocks, 0.468 ms in whole lines.  So the tail of a run that ends inside a block is held back: the owner
// lane keeps it in registers (`v`: the last CP 16-byte pieces of each array's run, read out of its own tile column) and
// stores it right before the rest of the block leaves with the next tile.  Tails are multiples of 16 bytes (the vector
// path needs 16-byte aligned runs); with an even n and a row pitch that is a multiple of 32 bytes they are multiples of 32:
// up to 96 bytes at BLOCK = 128 (CP = 6 pieces = the whole run at n = 6), 0 or 32 at BLOCK = 64 (CP = 2).
constexpr int MP_FD_BLOCK = 128;  // whole lines: tails up to 96 bytes per array (72 registers at n = 6); whole 64-byte half lines measured 0.520 against 0.497 ms
template <int N>
struct MpFdCarry {
  static constexpr int BLOCK = MP_FD_BLOCK, MASK = BLOCK - 1;
  static constexpr int C = MP_FD_KS * N / 4;  // 16-byte pieces per run
  static constexpr int CP_WANT = (N % 2 == 0) ? (BLOCK - 32) / 16 : (BLOCK - 16) / 16;  // largest tail, in pieces
  static constexpr int CP = CP_WANT < C ? CP_WANT : C;                                   // (a tail is never longer than a run)
  static constexpr bool ENABLED = BLOCK == 64 ? (C > CP) : (C >= 4 && MP_FD_KS * N * 4 <= BLOCK);  // 64: not for 1 - 3 joints
  // which flush a kernel is COMPILED with (one of them: with both in one kernel the register allocator spills): the
  // owner-lane flush where runs are not whole blocks by themselves, the wave-cooperative flat stores where they are (n = 8:
  // flat 0.684 ms against 0.752 - 0.779 ms owner-lane on the Panda) and for the short rows of 1 - 3 joints
  static constexpr bool OWNER = ENABLED && (MP_FD_KS * N * 4) % BLOCK != 0;
  mp_io_u4 v[3][CP];
  int bytes;  // of this lane's trajectory: how much of the previous tile's run is still to be written
};
// `hold`: whole-block mode (see MpFdCarry): the tail of a run that ends inside a block is NOT stored (its
// owner lane keeps it and stores it right before the next tile's run, which completes the block); `pm` = bytes between
// the runs of neighbouring trajectories mod BLOCK, `e_end` = byte offset of the END of trajectory b0's run mod BLOCK.
template <int N, int STEP, int RS>
__device__ __forceinline__ void mp_fd_tile_out_flat(float* __restrict__ pos, float* __restrict__ vel, float* __restrict__ acc,
                                                    long run0, long pitch, int nvalid, const MpFdFlat<N>& F,
                                                    const unsigned* __restrict__ lds, bool hold = false, int pm = 0, int e_end = 0) {
  constexpr int NK = MpFdFlat<N>::NK;
  constexpr int RUN_BYTES = MP_FD_KS * N * 4;
  float* const arr[3] = {pos, vel, acc};
  // one array at a time: its NK x 4 LDS reads are issued together (one LDS round trip), then its NK stores; holding all
  // three arrays' chunks at once (72 registers + 18 addresses) pushed the kernel past 256 VGPRs into scratch
#pragma unroll
  for (int slot = 0; slot < 3; ++slot) {
    mp_io_u4 v[NK];
#pragma unroll
    for (int k = 0; k < NK; ++k) {
      const int t = F.t(k), c = F.c(k);
      unsigned e[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int r = 4 * c + i, s = r / N, j = r - s * N;  // run index -> (step, joint)
        e[i] = lds[s * STEP + (slot * N + j) * RS + t];
      }
      v[k].x = e[0]; v[k].y = e[1]; v[k].z = e[2]; v[k].w = e[3];
    }
    float* const base = arr[slot] + run0;
    if (hold) {
#pragma unroll
      for (int k = 0; k < NK; ++k) {
        const int t = F.t(k), c = F.c(k);
        const int tail = (t * pm + e_end) & MpFdCarry<N>::MASK;  // bytes of trajectory t's run past its last whole block
        if (t < nvalid && 16 * c < RUN_BYTES - tail) *reinterpret_cast<mp_io_u4*>(base + (long)t * pitch + 4 * c) = v[k];
      }
    } else if (nvalid == 64) {
#pragma unroll
      for (int k = 0; k < NK; ++k) {
        *reinterpret_cast<mp_io_u4*>(base + (long)F.t(k) * pitch + 4 * F.c(k)) = v[k];
      }
    } else {
#pragma unroll
      for (int k = 0; k < NK; ++k)
        if (F.t(k) < nvalid) *reinterpret_cast<mp_io_u4*>(base + (long)F.t(k) * pitch + 4 * F.c(k)) = v[k];
    }
  }
}

// the tile still holds the run: keep its last CP pieces (own column, no bank conflicts)
template <int N, int STEP, int RS>
__device__ __forceinline__ void mp_fd_carry_keep(MpFdCarry<N>& K, const unsigned* __restrict__ col) {
  constexpr int CP = MpFdCarry<N>::CP, C = MpFdCarry<N>::C;
#pragma unroll
  for (int slot = 0; slot < 3; ++slot)
#pragma unroll
    for (int q = 0; q < CP; ++q) {
      unsigned e[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int r = 4 * (C - CP + q) + i, s = r / N, j = r - s * N;
        e[i] = col[s * STEP + (slot * N + j) * RS];
      }
      K.v[slot][q].x = e[0]; K.v[slot][q].y = e[1]; K.v[slot][q].z = e[2]; K.v[slot][q].w = e[3];
    }
}
// `run` = float index of this lane's run of the CURRENT tile; the pieces kept end right before it
template <int N>
__device__ __forceinline__ void mp_fd_carry_store(const MpFdCarry<N>& K, float* __restrict__ pos, float* __restrict__ vel,
                                                  float* __restrict__ acc, long run) {
  constexpr int CP = MpFdCarry<N>::CP;
  float* const arr[3] = {pos, vel, acc};
#pragma unroll
  for (int slot = 0; slot < 3; ++slot)
#pragma unroll
    for (int q = 0; q < CP; ++q)
      if ((CP - q) * 16 <= K.bytes) *reinterpret_cast<mp_io_u4*>(arr[slot] + run - 4 * (CP - q)) = K.v[slot][q];
}

// Owner-lane flush (the default wherever runs are not whole blocks by themselves): every lane stores the rows of its OWN trajectory straight from its own tile
// column - no cross-lane reads (no wave barrier, no chunk -> (trajectory, piece) arithmetic, no bank conflicts) - and only
// what completes whole blocks: the pieces the previous tile held back, then this run's pieces up to its last whole block;
// the rest stays in `K` for the next tile.  A store instruction then touches 64 different lines, but each lane's pieces of
// one line leave within the same flush (tools/ubench_c5io.hip "whole 128-byte lines (lane = trajectory)").
template <int N, int STEP, int RS>
__device__ __forceinline__ void mp_fd_tile_out_owner(float* __restrict__ pos, float* __restrict__ vel, float* __restrict__ acc,
                                                     long run, bool in_batch, const unsigned* __restrict__ col, int tail,
                                                     MpFdCarry<N>& K) {
  constexpr int C = MpFdCarry<N>::C, CP = MpFdCarry<N>::CP, RUN_BYTES = MP_FD_KS * N * 4;
  float* const arr[3] = {pos, vel, acc};
#pragma unroll
  for (int slot = 0; slot < 3; ++slot) {
    if (in_batch) {
#pragma unroll
      for (int q = 0; q < CP; ++q)
        if ((CP - q) * 16 <= K.bytes) *reinterpret_cast<mp_io_u4*>(arr[slot] + run - 4 * (CP - q)) = K.v[slot][q];
    }
    mp_io_u4 v[C];
#pragma unroll
    for (int c = 0; c < C; ++c) {
      unsigned e[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int r = 4 * c + i, s = r / N, j = r - s * N;
        e[i] = col[s * STEP + (slot * N + j) * RS];
      }
      v[c].x = e[0]; v[c].y = e[1]; v[c].z = e[2]; v[c].w = e[3];
    }
    if (in_batch) {
#pragma unroll
      for (int c = 0; c < C; ++c)
        if (16 * c < RUN_BYTES - tail) *reinterpret_cast<mp_io_u4*>(arr[slot] + run + 4 * c) = v[c];
    }
#pragma unroll
    for (int q = 0; q < CP; ++q) K.v[slot][q] = v[C - CP + q];
  }
  K.bytes = tail;
}

template <typename T, int TW, int W>
__device__ __forceinline__ T mp_fd_tile_get(const unsigned* __restrict__ cs, int c, int E) {
  if constexpr (TW == 1) {
    return (T)__builtin_bit_cast(float, cs[c * W]);
  } else {
    const unsigned long long u = (unsigned long long)cs[c * W] | ((unsigned long long)cs[(c + E) * W] << 32);
    return (T)__builtin_bit_cast(double, u);
  }
}

// `lds` is this wave's tile (MpFdTile<T, N, HAS_FTIP>::DWORDS dwords), `lane` the lane index inside the wave
template <typename T, int N, bool HAS_FTIP, typename MT>
__device__ __forceinline__ void mp_body_fd_traj(const MT& M, const MpCall<T>& C, const T* __restrict__ theta0,
                                                const T* __restrict__ dtheta0, const T* __restrict__ taumat,
                                                const T* __restrict__ Ftipmat, long bl, long B, long Nt, T h, int intRes,
                                                float* __restrict__ pos, float* __restrict__ vel, float* __restrict__ acc,
                                                unsigned* __restrict__ lds, int lane) {
  using TL = MpFdTile<T, N, HAS_FTIP>;
  constexpr int TW = TL::TW, STEP = TL::STEP, RS = TL::RS;
  // EVERY lane of the wave runs this body (the flat stores are wave-cooperative): lanes past the batch integrate the
  // last trajectory again and store nothing
  const long b0 = mp_wave_uniform(bl - lane);      // the wave's first trajectory (wave-uniform, and said to be)
  if (b0 >= B) return;
  const bool in_batch = bl < B;
  const long b = in_batch ? bl : B - 1;
  const int nvalid = (B - b0) < 64 ? (int)(B - b0) : 64;
  unsigned* col = lds + lane;
  T q[N], qd[N];
  RunIO<T, N>::load(theta0, b, q);
  RunIO<T, N>::load(dtheta0, b, qd);
  // Sticky non-finite verdict (running maxima of the bit patterns, mp_core.h): the initial state, every torque / wrench
  // row consumed and the integrated velocity feed it; from the first step at which it trips, the trajectory's rows are
  // NaN, as in the reference, whose state stays non-finite once it is (row 0 is the initial state as given).
  MpBad<T> bad;
  bad.add(q); bad.add(qd);
  // 16-byte vector accesses need every lane's run to start on a 16-byte boundary (wave-uniform tests)
  const bool vec_tau = ((Nt * N * (long)sizeof(T)) & 15) == 0, vec_out = ((Nt * N * 4) & 15) == 0;
  const bool vec_f = ((Nt * 6 * (long)sizeof(T)) & 15) == 0;
  // whole-block output (MpFdCarry): the arrays start on a block boundary, tails come in the sizes the carry holds
  const bool whole_blocks = vec_out && MpFdCarry<N>::ENABLED &&
                       (((unsigned long long)pos | (unsigned long long)vel | (unsigned long long)acc) & (unsigned long long)MpFdCarry<N>::MASK) == 0 &&
                       (N % 2 != 0 || ((Nt * N * 4) & 31) == 0) &&
                       !((MP_FD_KS * N * 4) % MpFdCarry<N>::BLOCK == 0 && ((Nt * N * 4) & MpFdCarry<N>::MASK) == 0);  // runs that are whole blocks anyway (n = 4, 8)
  const int pitch_mod = (int)((Nt * N * 4) & MpFdCarry<N>::MASK);
  MpFdCarry<N> carry;
  carry.bytes = 0;
  // Line-exact input reads (n = 6, float32 inputs).  Time-neutral (0.499 against 0.503 ms) and
  // kept for what it does to the traffic: HBM-side bytes per launch 2.21 GB -> 1.61 GB = 1.02 x the algorithmic bytes.
  // A lane's 96-byte run of a tile starts where the previous one ended, so three of four runs straddle a 128-byte line and every line is fetched by two tiles ~20 us apart (FETCH_SIZE
  // = 2.0 x the input bytes; the second fetch is an Infinity-Cache hit, not an L2 hit).  Here a lane fetches a whole line
  // - the one that holds the LAST float of its run - only in the three of four tiles whose run reaches past what it already
  // holds, keeps the line's last 96 bytes in registers (`held_*`, 48 registers for both arrays) and assembles the run from
  // them and the new line.  With Nt % 4 == 0 the stream starts on a 32-byte boundary and everything moves in 32-byte groups:
  // phase = (tile - u0) & 3 says how the three groups of the run map onto (held, new) groups; torques and wrenches have
  // the same row size at n = 6, so one phase serves both.  Loads go through buffer descriptors over the whole arrays: a
  // lane that needs nothing this tile, or whose line would start past the array, reads nothing (offset out of range).
#if defined(MP_SPECIALISED)
  constexpr bool EXACT_OK = N == 6 && TW == 1;
#else  // the generic kernels carry the robot model in registers: 48 more for the held lines and they spill (0.80 -> 1.03 ms)
  constexpr bool EXACT_OK = false;
#endif
  const unsigned long long in_bytes = (unsigned long long)B * (unsigned long long)Nt * 24ull;
  const bool exact_in = EXACT_OK && (Nt & 3) == 0 && in_bytes < (1ull << 31) &&
                        (((unsigned long long)taumat | (HAS_FTIP ? (unsigned long long)Ftipmat : 0ull)) & 127ull) == 0;
  // (named members, literal indices: with arrays the compiler turns "select between two array elements" into a load from a
  //  selected ADDRESS and the arrays end up in scratch memory)
  struct Held { mp_io_u4 a, b, c, d, e, f; };
  const mp_io_u4 zero4 = {0u, 0u, 0u, 0u};
  Held held_tau = {zero4, zero4, zero4, zero4, zero4, zero4}, held_f = held_tau;
  int ex_s0 = 0, ex_u0 = 0;
  __amdgpu_buffer_rsrc_t ex_rt = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(taumat), 0, exact_in ? (int)in_bytes : 0, 0x00020000);
  __amdgpu_buffer_rsrc_t ex_rf = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(HAS_FTIP ? Ftipmat : taumat), 0,
                                                                   exact_in && HAS_FTIP ? (int)in_bytes : 0, 0x00020000);
  if (exact_in) {
    ex_s0 = (int)(b * Nt * 24);          // byte offset of this lane's streams (both arrays)
    ex_u0 = (ex_s0 >> 5) & 3;
    const int line0 = ex_s0 & ~127;      // the line the stream starts in: its tail is what the first tile may need
    held_tau.a = __builtin_amdgcn_raw_buffer_load_b128(ex_rt, line0 + 32, 0, 0);
    held_tau.b = __builtin_amdgcn_raw_buffer_load_b128(ex_rt, line0 + 48, 0, 0);
    held_tau.c = __builtin_amdgcn_raw_buffer_load_b128(ex_rt, line0 + 64, 0, 0);
    held_tau.d = __builtin_amdgcn_raw_buffer_load_b128(ex_rt, line0 + 80, 0, 0);
    held_tau.e = __builtin_amdgcn_raw_buffer_load_b128(ex_rt, line0 + 96, 0, 0);
    held_tau.f = __builtin_amdgcn_raw_buffer_load_b128(ex_rt, line0 + 112, 0, 0);
    if (HAS_FTIP) {
      held_f.a = __builtin_amdgcn_raw_buffer_load_b128(ex_rf, line0 + 32, 0, 0);
      held_f.b = __builtin_amdgcn_raw_buffer_load_b128(ex_rf, line0 + 48, 0, 0);
      held_f.c = __builtin_amdgcn_raw_buffer_load_b128(ex_rf, line0 + 64, 0, 0);
      held_f.d = __builtin_amdgcn_raw_buffer_load_b128(ex_rf, line0 + 80, 0, 0);
      held_f.e = __builtin_amdgcn_raw_buffer_load_b128(ex_rf, line0 + 96, 0, 0);
      held_f.f = __builtin_amdgcn_raw_buffer_load_b128(ex_rf, line0 + 112, 0, 0);
    }
  }
  // Issue priority.  The two waves of a SIMD are not served alike: the arbiter takes the older one (wave slot 0) first, which
  // then runs at the speed of a lone wave and finishes 25 % early, leaving its partner to integrate the rest alone.  The
  // favoured wave alternates tile by tile (s_setprio from tile index ^ wave slot) and the tile boundaries (loads, stores)
  // run at the highest priority, so their memory requests leave early: c5 0.554 -> 0.518 ms.
  const unsigned wave_slot = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 4) & 1u;
  unsigned tile_no = 0;
  for (long i0 = 0; i0 < Nt; i0 += MP_FD_KS) {
    const long left = Nt - i0;
    const int rows = left < MP_FD_KS ? (int)left : MP_FD_KS;
    const bool full = rows == MP_FD_KS;
    const long row0 = b * Nt + i0;
    const bool next_full = left - rows >= MP_FD_KS;  // the tile after this one is a whole tile
    bool parked = false;
    if constexpr (N == 6 && TW == 1) {
      if (exact_in && full) {
        const int tile = (int)(i0 / MP_FD_KS);
        const int phi = (tile - ex_u0) & 3;
        // (a lane that needs nothing reads past the descriptor and gets nothing; the sentinel leaves room for the + 112 below)
        const int off = phi != 3 ? ((ex_s0 + 96 * tile + 92) & ~127) : 0x7fffff00;
        const bool p0 = phi == 0, p1 = phi == 1, p2 = phi == 2;
        // the run's six 16-byte pieces: group G (pieces 2G, 2G+1) is new group G in phase 0, G-1 in phase 1, G-2 in phase 2,
        // and otherwise one of the held groups (held pieces a..f = pieces 2..7 of the previous line)
        auto sel = [&](const mp_io_u4 x0, const mp_io_u4 x1, const mp_io_u4 x2, const mp_io_u4 x3) __attribute__((always_inline)) {
          return p0 ? x0 : p1 ? x1 : p2 ? x2 : x3;
        };
        auto one = [&](const __amdgpu_buffer_rsrc_t rs, Held& h, MpFdPrefetch<6, 1>& r) __attribute__((always_inline)) {
          const mp_io_u4 n0 = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0);
          const mp_io_u4 n1 = __builtin_amdgcn_raw_buffer_load_b128(rs, off + 16, 0, 0);
          const mp_io_u4 n2 = __builtin_amdgcn_raw_buffer_load_b128(rs, off + 32, 0, 0);
          const mp_io_u4 n3 = __builtin_amdgcn_raw_buffer_load_b128(rs, off + 48, 0, 0);
          const mp_io_u4 n4 = __builtin_amdgcn_raw_buffer_load_b128(rs, off + 64, 0, 0);
          const mp_io_u4 n5 = __builtin_amdgcn_raw_buffer_load_b128(rs, off + 80, 0, 0);
          const mp_io_u4 n6 = __builtin_amdgcn_raw_buffer_load_b128(rs, off + 96, 0, 0);
          const mp_io_u4 n7 = __builtin_amdgcn_raw_buffer_load_b128(rs, off + 112, 0, 0);
          r.v[0] = sel(n0, h.e, h.c, h.a); r.v[1] = sel(n1, h.f, h.d, h.b);
          r.v[2] = sel(n2, n0, h.e, h.c);  r.v[3] = sel(n3, n1, h.f, h.d);
          r.v[4] = sel(n4, n2, n0, h.e);   r.v[5] = sel(n5, n3, n1, h.f);
          // (a lane that fetched nothing has no use for what it holds: its next run starts a new line)
          h.a = n2; h.b = n3; h.c = n4; h.d = n5; h.e = n6; h.f = n7;
        };
        MpFdPrefetch<6, 1> run_tau, run_f;
        one(ex_rt, held_tau, run_tau);
        if (HAS_FTIP) one(ex_rf, held_f, run_f);
        mp_fd_tile_park<6, 1, TL::TAU0, STEP, RS>(run_tau, col);
        if (HAS_FTIP) mp_fd_tile_park<6, 1, TL::F0, STEP, RS>(run_f, col);
        parked = true;
      }
    }
    if (parked) {
    } else
    // both input arrays in ONE round trip: all their loads are issued before the first row is parked (two separate
    // load-wait-park sequences sit in different branches, and the compiler does not hoist loads across them)
    if (full && vec_tau && (!HAS_FTIP || vec_f) && (MP_FD_KS * N * TW) % 4 == 0) {
      MpFdPrefetch<N, TW> now_tau;
      MpFdPrefetch<6, TW> now_f;
      mp_fd_tile_load<N, TW>(reinterpret_cast<const unsigned*>(taumat + row0 * N), now_tau);
      if (HAS_FTIP) mp_fd_tile_load<6, TW>(reinterpret_cast<const unsigned*>(Ftipmat + row0 * 6), now_f);
      mp_fd_tile_park<N, TW, TL::TAU0, STEP, RS>(now_tau, col);
      if (HAS_FTIP) mp_fd_tile_park<6, TW, TL::F0, STEP, RS>(now_f, col);
    } else {  // partial tiles and rows that are not 16-byte aligned: array by array, vector or dword accesses
      const unsigned* gt = reinterpret_cast<const unsigned*>(taumat + row0 * N);
      if (full && vec_tau) mp_fd_tile_in<N, TW, TL::TAU0, STEP, RS, 4>(gt, 0, col);
      else mp_fd_tile_in<N, TW, TL::TAU0, STEP, RS, 1>(gt, rows * N * TW, col);
      if (HAS_FTIP) {
        const unsigned* gf = reinterpret_cast<const unsigned*>(Ftipmat + row0 * 6);
        if (full && vec_f) mp_fd_tile_in<6, TW, TL::F0, STEP, RS, 4>(gf, 0, col);
        else mp_fd_tile_in<6, TW, TL::F0, STEP, RS, 1>(gf, rows * 6 * TW, col);
      }
    }
    if ((tile_no ^ wave_slot) & 1u) __builtin_amdgcn_s_setprio(1);
    else __builtin_amdgcn_s_setprio(0);
    ++tile_no;
    for (int s = 0; s < rows; ++s) {
      unsigned* cs = col + s * STEP;
      T last[N];
#pragma unroll
      for (int j = 0; j < N; ++j) last[j] = T(0);
      if (i0 + s > 0) {
        T tau[N], tn[3] = {T(0), T(0), T(0)}, tf[3] = {T(0), T(0), T(0)};
#pragma unroll
        for (int j = 0; j < N; ++j)
          tau[j] = mp_fd_tile_get<T, TW, RS>(cs, TL::TAU0 + j, N);
        bad.add(tau);
        if (HAS_FTIP) {
          T F[6];
#pragma unroll
          for (int k = 0; k < 6; ++k)
            F[k] = mp_fd_tile_get<T, TW, RS>(cs, TL::F0 + k, 6);
          bad.add(F);
          mp_wrench_to_frame1(M, F, tn, tf);
        }
        for (int k = 0; k < intRes; ++k) {
          mp_forward_dynamics<T, N, HAS_FTIP>(M, C.a0, tn, tf, q, qd, tau, last);
#pragma unroll
          for (int j = 0; j < N; ++j) {
            qd[j] = qd[j] + last[j] * h;
            q[j] = mp_clip(q[j] + qd[j] * h, M.qmin[j], M.qmax[j]);
          }
        }
        bad.add(qd);
      }
      const bool poison = (i0 + s > 0) && bad.any();
#pragma unroll
      for (int j = 0; j < N; ++j) {
        cs[j * RS] = poison ? 0x7fc00000u : __builtin_bit_cast(unsigned, (float)q[j]);
        cs[(N + j) * RS] = poison ? 0x7fc00000u : __builtin_bit_cast(unsigned, (float)qd[j]);
        cs[(2 * N + j) * RS] = poison ? 0x7fc00000u : __builtin_bit_cast(unsigned, (float)last[j]);
      }
    }
    __builtin_amdgcn_s_setprio(3);
    bool flushed = false;
    if constexpr (MpFdCarry<N>::OWNER) {
     if (full && vec_out) {
      const bool hold = whole_blocks && next_full;
      const int e_end = (int)(((i0 + MP_FD_KS) * N * 4) & MpFdCarry<N>::MASK);
      int my = lane;
      asm volatile("" : "+v"(my));
      mp_fd_tile_out_owner<N, STEP, RS>(pos, vel, acc, row0 * N, in_batch, col, hold ? ((my * pitch_mod + e_end) & MpFdCarry<N>::MASK) : 0,
                                        carry);
      flushed = true;
     }
    }
    if (flushed) {
    } else
    if (!MpFdCarry<N>::OWNER && full && vec_out) {
      // the tile was written column by column (each lane its own); the flat stores read across columns
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      const long run0 = (b0 * Nt + i0) * N;
      // the chunk -> (trajectory, piece) split is the same for every tile; recomputing it from an opaque copy of the lane
      // index keeps the compiler from hoisting a dozen per-lane 64-bit offsets out of the time loop (they were spilled)
      MpFdFlat<N> fl;
      fl.lane = lane;
      asm volatile("" : "+v"(fl.lane));
      const bool hold = whole_blocks && next_full;  // the next tile is a whole tile: it completes the blocks this one leaves open
      const int e_end = (int)(((i0 + MP_FD_KS) * N * 4) & MpFdCarry<N>::MASK);
      if constexpr (MpFdCarry<N>::ENABLED) {
        if (whole_blocks && in_batch && carry.bytes > 0) mp_fd_carry_store<N>(carry, pos, vel, acc, row0 * N);
      }
      mp_fd_tile_out_flat<N, STEP, RS>(pos, vel, acc, run0, Nt * N, nvalid, fl, lds, hold, pitch_mod, e_end);
      if constexpr (MpFdCarry<N>::ENABLED) {
        carry.bytes = hold ? ((fl.lane * pitch_mod + e_end) & MpFdCarry<N>::MASK) : 0;
        if (hold) mp_fd_carry_keep<N, STEP, RS>(carry, col);
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  // the next tile's inputs overwrite these columns
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    } else
    if (!in_batch) {
      // partial / unaligned tiles leave lane by lane: lanes past the batch have nothing to store
    } else if (full && vec_out) {
      mp_fd_tile_out<N, STEP, RS, 4>(pos + row0 * N, 0, 0, col);
      mp_fd_tile_out<N, STEP, RS, 4>(vel + row0 * N, 1, 0, col);
      mp_fd_tile_out<N, STEP, RS, 4>(acc + row0 * N, 2, 0, col);
    } else {
      mp_fd_tile_out<N, STEP, RS, 1>(pos + row0 * N, 0, rows * N, col);
      mp_fd_tile_out<N, STEP, RS, 1>(vel + row0 * N, 1, rows * N, col);
      mp_fd_tile_out<N, STEP, RS, 1>(acc + row0 * N, 2, rows * N, col);
    }
  }
}

// ------------------------------------------------------------------ the roll-out, TIME-MAJOR device layout
// forward_dynamics_trajectory for B trajectories whose device arrays are laid out (Nt, B, *): taumat (Nt, B, N),
// Ftipmat (Nt, B, 6), pos / vel / acc (Nt, B, N).  The reference integrates ONE (N, n) trajectory
// (planning/trajectory_dynamics.py:382-423, :580-708); the batch axis is this library's extension, so where it sits in
// device memory is the library's choice, and with time outermost the 64 trajectories of a wave are neighbours in memory
// at EVERY step: a step reads 64 x N x sizeof(T) contiguous bytes per input array and writes 64 x N x 4 contiguous bytes
// per output array (1536 B = twelve whole lines at n = 6) - the access pattern of the inverse-dynamics kernels, which
// streams at the chip's limit.  No LDS tile, no held-back tails, no tile boundary: the rows of step i + 1 are requested
// before step i is integrated (N + 6 registers) and the output rows are stored as they are produced, never waited for
// (vmcnt counts in issue order, and the only waits in the loop are for loads issued a whole step earlier).
// Same arithmetic, same clip, same sticky non-finite verdict as mp_body_fd_traj; one lane = one trajectory.
// The wave waits for the prefetched rows HERE (an empty asm that consumes every register): at the end of a step, behind the
// step's stores, where the wait is vmcnt(number of stores) - the loads are a whole step old - and the stores stay in
// flight.  Left to itself the compiler waits at the TOP of the next step, after the new loads were issued, with a count
// that also drains the previous step's stores (the loop head merges the entry path, on which the rows are still pending).
template <typename T, int N, bool HAS_FTIP>
__device__ __forceinline__ void mp_fd_rows_arrived(T (&tau)[N], T (&F)[6]) {
#pragma unroll
  for (int j = 0; j < N; ++j) asm volatile("" : "+v"(tau[j]));
  if (HAS_FTIP) {
#pragma unroll
    for (int k = 0; k < 6; ++k) asm volatile("" : "+v"(F[k]));
  }
}

// The favoured wave of a SIMD alternates step by step
// (s_setprio from step index ^ wave slot) - the arbiter serves the older wave first, which then runs ahead and leaves its
// partner alone at the end (c5: 0.391 against 0.398 ms).  Measured and not kept (profiles/r03_c5_tm_experiments.txt):
// requesting the rows two steps ahead (three register sets, time loop unrolled by three): 0.391 against 0.389 ms;
// non-temporal output stores: 0.396 against 0.398 ms.

// One row of a time-major array for this lane: `ubase` = the array, `urow` = wave-uniform element index of the wave's first
// row, `off` = this lane's element offset inside the wave's span (lane * COUNT: constant over the steps).  Written as
// (uniform 64-bit base) + (32-bit per-lane offset) so that the access takes the scalar-base form (global_load ... v_off,
// s[base:base+1]) and a step costs two scalar instructions per array instead of a per-lane 64-bit multiply-add.
template <typename T, int COUNT>
struct RowIO {
  using IO = RunIO<T, COUNT>;
  using V = typename IO::V;
  static constexpr int K = IO::K;
  static __device__ __forceinline__ void load(const T* __restrict__ ubase, long urow, unsigned off, T (&v)[COUNT]) {
    const char* p = reinterpret_cast<const char*>(ubase + urow);
#pragma unroll
    for (int k = 0; k < COUNT / K; ++k) {
      union { V vec; T e[K]; } u;
      u.vec = *reinterpret_cast<const V*>(p + (size_t)(off * (unsigned)sizeof(T) + (unsigned)(k * sizeof(V))));
#pragma unroll
      for (int j = 0; j < K; ++j) v[k * K + j] = u.e[j];
    }
  }
  static __device__ __forceinline__ void store(T* __restrict__ ubase, long urow, unsigned off, const T (&v)[COUNT]) {
    char* p = reinterpret_cast<char*>(ubase + urow);
#pragma unroll
    for (int k = 0; k < COUNT / K; ++k) {
      union { V vec; T e[K]; } u;
#pragma unroll
      for (int j = 0; j < K; ++j) u.e[j] = v[k * K + j];
      *reinterpret_cast<V*>(p + (size_t)(off * (unsigned)sizeof(T) + (unsigned)(k * sizeof(V)))) = u.vec;
    }
  }
};

template <typename T, int N>
struct MpTmRows {  // the input rows of one step
  T tau[N], F[6];
};

// `b0` = the wave's first trajectory (wave-uniform), `lane` = this lane's index in the wave; the caller has already sent
// lanes with b0 + lane >= B home.
template <typename T, int N, bool HAS_FTIP, typename MT>
__device__ __forceinline__ void mp_body_fd_traj_tm(const MT& M, const MpCall<T>& C, const T* __restrict__ theta0,
                                                   const T* __restrict__ dtheta0, const T* __restrict__ taumat,
                                                   const T* __restrict__ Ftipmat, long b0, int lane, long B, long Nt, T h,
                                                   int intRes, float* __restrict__ pos, float* __restrict__ vel,
                                                   float* __restrict__ acc) {
  using Rows = MpTmRows<T, N>;
  const unsigned offN = (unsigned)lane * N, off6 = (unsigned)lane * 6;
  T q[N], qd[N];
  RowIO<T, N>::load(theta0, b0 * N, offN, q);
  RowIO<T, N>::load(dtheta0, b0 * N, offN, qd);
  MpBad<T> bad;
  bad.add(q); bad.add(qd);
  auto request = [&](Rows& R, long step) __attribute__((always_inline)) {  // rows of `step` (clamped: the tail re-reads the last row)
    const long r = (step < Nt ? step : Nt - 1) * B + b0;
    RowIO<T, N>::load(taumat, r * N, offN, R.tau);
    if (HAS_FTIP) RowIO<T, 6>::load(Ftipmat, r * 6, off6, R.F);
  };
  auto arrived = [&](Rows& R) __attribute__((always_inline)) { mp_fd_rows_arrived<T, N, HAS_FTIP>(R.tau, R.F); };
  const unsigned wave_slot = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 4) & 1u;  // HW_ID.WAVE_ID bit 0
  // one integration step from the rows in R; its output rows are stored and never waited for
  auto step = [&](long i, const Rows& R) __attribute__((always_inline)) {
    if (((unsigned)i ^ wave_slot) & 1u) __builtin_amdgcn_s_setprio(1);
    else __builtin_amdgcn_s_setprio(0);
    T tau[N], tn[3] = {T(0), T(0), T(0)}, tf[3] = {T(0), T(0), T(0)}, last[N];
#pragma unroll
    for (int j = 0; j < N; ++j) {
      tau[j] = R.tau[j];
      last[j] = T(0);
    }
    bad.add(tau);
    if (HAS_FTIP) {
      T F[6];
#pragma unroll
      for (int k = 0; k < 6; ++k) F[k] = R.F[k];
      bad.add(F);
      mp_wrench_to_frame1(M, F, tn, tf);
    }
    for (int k = 0; k < intRes; ++k) {
      mp_forward_dynamics<T, N, HAS_FTIP>(M, C.a0, tn, tf, q, qd, tau, last);
#pragma unroll
      for (int j = 0; j < N; ++j) {
        qd[j] = qd[j] + last[j] * h;
        q[j] = mp_clip(q[j] + qd[j] * h, M.qmin[j], M.qmax[j]);
      }
    }
    bad.add(qd);
    float p[N], v[N], a[N];
#pragma unroll
    for (int j = 0; j < N; ++j) { p[j] = (float)q[j]; v[j] = (float)qd[j]; a[j] = (float)last[j]; }
    // non-finite trajectories are rare: one wave-uniform branch instead of 3 N selects per step
    if (__builtin_amdgcn_ballot_w64(bad.any()) != 0) {
      const bool poison = bad.any();
      mp_poison_if(poison, p); mp_poison_if(poison, v); mp_poison_if(poison, a);
    }
    {
      const long r = (i * B + b0) * N;
      RowIO<float, N>::store(pos, r, offN, p);
      RowIO<float, N>::store(vel, r, offN, v);
      RowIO<float, N>::store(acc, r, offN, a);
    }
  };
  Rows R0;
  request(R0, 1);
  {  // row 0 = the initial state as given, zero acceleration
    float p[N], v[N], a[N];
#pragma unroll
    for (int j = 0; j < N; ++j) { p[j] = (float)q[j]; v[j] = (float)qd[j]; a[j] = 0.0f; }
    RowIO<float, N>::store(pos, b0 * N, offN, p);
    RowIO<float, N>::store(vel, b0 * N, offN, v);
    RowIO<float, N>::store(acc, b0 * N, offN, a);
  }
  arrived(R0);
  for (long i = 1; i < Nt; ++i) {  // at the loop head R0 holds the rows of step i (arrived)
    const Rows cur = R0;
    request(R0, i + 1);
    step(i, cur);
    arrived(R0);
  }
}

// (outer, inner, W dwords) -> (inner, outer, W dwords): the conversion between the batch-major API arrays (B, N, n) and
// the time-major device layout (N, B, n), either way.  A block moves a TO x TI tile of rows through LDS: it reads TI * W
// contiguous dwords per outer index and writes TO * W contiguous dwords per inner index (768 bytes each at n = 6).
constexpr int MP_TR_TO = 32;
__device__ __host__ constexpr int mp_tr_ti(int W) { return W <= 8 ? 32 : (W <= 16 ? 16 : (W <= 32 ? 8 : 4)); }  // tile extent along `inner`: <= 33 KB of LDS up to 256-byte rows (32 float64 joints)
__device__ __forceinline__ void mp_body_transpose_rows(const unsigned* __restrict__ src, unsigned* __restrict__ dst, long outer,
                                                       long inner, int W, long o0, long i0, unsigned* __restrict__ lds, int tid,
                                                       int nthreads) {
  const int RUN = mp_tr_ti(W) * W, PITCH = RUN + 1, TOTAL = MP_TR_TO * RUN;
  for (int k = tid; k < TOTAL; k += nthreads) {
    const int o = k / RUN, r = k - o * RUN;
    if (o0 + o < outer && i0 + r / W < inner) lds[o * PITCH + r] = src[((o0 + o) * inner + i0) * W + r];
  }
  __syncthreads();
  const int ORUN = MP_TR_TO * W;
  for (int k = tid; k < TOTAL; k += nthreads) {
    const int i = k / ORUN, rr = k - i * ORUN, o = rr / W, w = rr - o * W;
    if (i0 + i < inner && o0 + o < outer) dst[((i0 + i) * outer + o0) * W + rr] = lds[o * PITCH + i * W + w];
  }
}
