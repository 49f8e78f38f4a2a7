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

#if !defined(__HIPCC_RTC__)
#include <cmath>
#endif

#include "mp_model.h"

#if defined(__HIPCC__) || defined(__HIPCC_RTC__)
#define MP_HD __host__ __device__ __forceinline__
#else
#define MP_HD inline
#endif

// ----------------------------------------------------------------------------- arithmetic types
// The per-row math is written once for an arithmetic type T and a scalar type S = MpTraits<T>::S that
// the wave-uniform model constants use:
//   T = float / double : one row per lane;
//   T = mp_f2 (2 x float): TWO rows per lane, every operation a packed v_pk_{fma,mul,add}_f32.  On gfx950 a packed
//       float32 instruction occupies the SIMD-32 for 4 cycles, a scalar v_fma_f32 / v_mul_f32 for 2
//       (tools/ubench_issue2.hip, profiles/r02_ubench_issue.txt; r01's "packing doubles the rate" was read off a loop whose
//       scalar form was not issue-limited), so packing buys no arithmetic throughput, only fewer instructions to issue:
//       it is used where that pays (odd DOF, whose 4 n-byte rows only allow dword accesses) and nowhere else.
#if defined(__clang__)
#define MP_HAS_PACKED 1
typedef float mp_f2 __attribute__((ext_vector_type(2)));
typedef int mp_i2 __attribute__((ext_vector_type(2)));
typedef unsigned mp_u2 __attribute__((ext_vector_type(2)));
#endif

template <typename T> struct MpTraits;
template <> struct MpTraits<float> {
  using S = float;
  static MP_HD float splat(float v) { return v; }
};
template <> struct MpTraits<double> {
  using S = double;
  static MP_HD double splat(double v) { return v; }
};
#if MP_HAS_PACKED
template <> struct MpTraits<mp_f2> {
  using S = float;
  static MP_HD mp_f2 splat(float v) { return (mp_f2){v, v}; }
};
#endif

// ------------------------------------------------------------------------------------------- trig
// float: Cody-Waite reduction by pi/2 (two FMAs, exact enough for |x| < ~1e4) + the classic minimax
// polynomials on [-pi/4, pi/4]; ~1 ulp, branch-free, ~24 VALU instructions for BOTH results.
MP_HD void mp_sincos(float x, float& s, float& c) {
  // x * 2/pi rounded to an integer by adding 1.5 * 2^23 inside the FMA: the sum's low mantissa bits ARE the quadrant (|x| < 6e6),
  // one instruction less than multiply + v_rndne + v_cvt (round 4: c2 -1.6 %, c5 -0.5 %, profiles/r04_ab_sincos_signs.txt)
  const float kf = fmaf(x, 0.636619772367581343f, 12582912.0f);
  const float k = kf - 12582912.0f;
  float r = fmaf(-k, 1.57079637050628662109375f, x);
  r = fmaf(-k, -4.37113900018624283e-8f, r);
  const float r2 = r * r;
  float ps = fmaf(r2, -1.9515295891e-4f, 8.3321608736e-3f);
  ps = fmaf(r2, ps, -1.6666654611e-1f);
  ps = fmaf(r * r2, ps, r);
  float pc = fmaf(r2, 2.443315711809948e-5f, -1.388731625493765e-3f);
  pc = fmaf(r2, pc, 4.166664568298827e-2f);
  pc = fmaf(r2 * r2, pc, fmaf(r2, -0.5f, 1.0f));
  const int q = __builtin_bit_cast(int, kf);
  const float a = (q & 1) ? pc : ps;
  const float b = (q & 1) ? ps : pc;
  // the signs straight from the quadrant's bits: bit 1 of q (sine) and of q + 1 (cosine) moved to bit 31 and XORed in
  const unsigned qs = (unsigned)q << 30;
  s = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, a) ^ (qs & 0x80000000u));
  c = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, b) ^ ((qs + 0x40000000u) & 0x80000000u));
}
// double: the same structure in float64 - reduction by pi/2 carried in three FMAs (pi/2 split in a 53-bit head and two
// tails; inside an FMA the product k * head is exact, so x - k pi/2 keeps full accuracy for every |x| whose own
// spacing still resolves an angle, no Payne-Hanek path needed), then the classic degree-13 / degree-12 minimax
// polynomials on [-pi/4, pi/4] (the fdlibm kernel coefficients).  ~1 ulp, branch-free, ~40 instructions for BOTH
// results; the library's sincos spends > 100 and a divergent big-argument path per call site.  The quadrant is taken
// in floating point (k mod 4), so there is no integer overflow for large |x|; NaN / inf give NaN.
MP_HD void mp_sincos(double x, double& s, double& c) {
  const double k = rint(x * 0.63661977236758134308);
  double r = fma(-k, 1.57079632679489655800e+00, x);
  r = fma(-k, 6.12323399573676603587e-17, r);
  r = fma(-k, -1.49738490485916983294e-33, r);
  const double z = r * r;
  double ps = fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08);
  ps = fma(z, ps, 2.75573137070700676789e-06);
  ps = fma(z, ps, -1.98412698298579493134e-04);
  ps = fma(z, ps, 8.33333333332248946124e-03);
  ps = fma(z, ps, -1.66666666666666324348e-01);
  ps = fma(r * z, ps, r);
  double pc = fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09);
  pc = fma(z, pc, -2.75573143513906633035e-07);
  pc = fma(z, pc, 2.48015872894767294178e-05);
  pc = fma(z, pc, -1.38888888888741095749e-03);
  pc = fma(z, pc, 4.16666666666666019037e-02);
  pc = fma(z * z, pc, fma(z, -0.5, 1.0));
  const int q = (int)(k - 4.0 * floor(k * 0.25));  // k mod 4 in {0, 1, 2, 3}
  const double a = (q & 1) ? pc : ps;
  const double b = (q & 1) ? ps : pc;
  const unsigned qs = (unsigned)q << 30;  // (as the float routine: the signs from the quadrant's bits)
  s = __builtin_bit_cast(double, __builtin_bit_cast(unsigned long long, a) ^ ((unsigned long long)(qs & 0x80000000u) << 32));
  c = __builtin_bit_cast(double, __builtin_bit_cast(unsigned long long, b) ^ ((unsigned long long)((qs + 0x40000000u) & 0x80000000u) << 32));
}
#if MP_HAS_PACKED
// the same algorithm on two rows at once (packed FMAs; rint / cvt / selects stay per component)
MP_HD void mp_sincos(mp_f2 x, mp_f2& s, mp_f2& c) {
  const mp_f2 kf = __builtin_elementwise_fma(x, (mp_f2)(0.636619772367581343f), (mp_f2)(12582912.0f));
  const mp_f2 k = kf - 12582912.0f;
  mp_f2 r = __builtin_elementwise_fma(-k, (mp_f2)(1.57079637050628662109375f), x);
  r = __builtin_elementwise_fma(-k, (mp_f2)(-4.37113900018624283e-8f), r);
  const mp_f2 r2 = r * r;
  mp_f2 ps = __builtin_elementwise_fma(r2, (mp_f2)(-1.9515295891e-4f), (mp_f2)(8.3321608736e-3f));
  ps = __builtin_elementwise_fma(r2, ps, (mp_f2)(-1.6666654611e-1f));
  ps = __builtin_elementwise_fma(r * r2, ps, r);
  mp_f2 pc = __builtin_elementwise_fma(r2, (mp_f2)(2.443315711809948e-5f), (mp_f2)(-1.388731625493765e-3f));
  pc = __builtin_elementwise_fma(r2, pc, (mp_f2)(4.166664568298827e-2f));
  pc = __builtin_elementwise_fma(r2 * r2, pc, __builtin_elementwise_fma(r2, (mp_f2)(-0.5f), (mp_f2)(1.0f)));
  const mp_i2 q = __builtin_bit_cast(mp_i2, kf);
  const mp_i2 odd = (q & 1) != 0;
  const mp_f2 a = odd ? pc : ps;
  const mp_f2 b = odd ? ps : pc;
  const mp_u2 qs = __builtin_bit_cast(mp_u2, q) << 30;  // (as the scalar routine: the signs from the quadrant's bits)
  s = __builtin_bit_cast(mp_f2, __builtin_bit_cast(mp_u2, a) ^ (qs & 0x80000000u));
  c = __builtin_bit_cast(mp_f2, __builtin_bit_cast(mp_u2, b) ^ ((qs + 0x40000000u) & 0x80000000u));
}
#endif

MP_HD float mp_min(float a, float b) { return a < b ? a : b; }
MP_HD float mp_max(float a, float b) { return a > b ? a : b; }
MP_HD double mp_min(double a, double b) { return a < b ? a : b; }
MP_HD double mp_max(double a, double b) { return a > b ? a : b; }
#if MP_HAS_PACKED
MP_HD mp_f2 mp_min(mp_f2 a, mp_f2 b) { return (a < b) ? a : b; }
MP_HD mp_f2 mp_max(mp_f2 a, mp_f2 b) { return (a > b) ? a : b; }
#endif
MP_HD float mp_abs(float a) { return __builtin_fabsf(a); }
MP_HD double mp_abs(double a) { return __builtin_fabs(a); }
#if MP_HAS_PACKED
MP_HD mp_f2 mp_abs(mp_f2 a) { return __builtin_elementwise_abs(a); }
#endif
MP_HD float mp_sqrt(float x) { return sqrtf(x); }
MP_HD double mp_sqrt(double x) { return sqrt(x); }
#if MP_HAS_PACKED
MP_HD mp_f2 mp_sqrt(mp_f2 x) { return (mp_f2){sqrtf(x.x), sqrtf(x.y)}; }
#endif
// 1 / sqrt(x) for the Cholesky pivots.  float32 on the device: v_rsq_f32 (1 ulp) plus one Newton step, 5 instructions
// where the IEEE sqrt followed by the IEEE divide expands to ~20 (v_sqrt + fix-up, v_div_scale x 2, v_rcp, four FMAs,
// v_div_fmas, v_div_fixup); the result is within 1 ulp of the correctly rounded quotient.  float64 keeps sqrt + divide.
MP_HD float mp_rsqrt(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
  const float y = __builtin_amdgcn_rsqf(x);
  const float e = fmaf(-(x * y), y, 1.0f);  // 1 - x y^2
  return fmaf(0.5f * y, e, y);
#else
  return 1.0f / sqrtf(x);
#endif
}
MP_HD double mp_rsqrt(double x) { return 1.0 / sqrt(x); }
#if MP_HAS_PACKED
MP_HD mp_f2 mp_rsqrt(mp_f2 x) { return (mp_f2){mp_rsqrt(x.x), mp_rsqrt(x.y)}; }
#endif
// np.clip order: max with the lower bound first, then min with the upper bound
template <typename T>
MP_HD T mp_clip(T v, typename MpTraits<T>::S lo, typename MpTraits<T>::S hi) {
  return mp_min(mp_max(v, MpTraits<T>::splat(lo)), MpTraits<T>::splat(hi));
}

// ------------------------------------------------------------------------------ non-finite inputs
// The reference hands back a non-finite row wherever a row's inputs hold a NaN or an infinity (NumPy raises nothing, so
// its try / except never fires; tests/golden/nonfinite.npz) and leaves every other row alone.  The robot-specialised
// kernels are compiled with -ffinite-math-only (that is what lets `x * 0` fold away, mp_jit.cpp), under which
// floating-point arithmetic may launder a NaN and a floating-point NaN test folds to false - so the guard works on the
// BIT PATTERNS with integer instructions, and the row is poisoned with integer selects on its way to memory.
// A value is non-finite iff its exponent field is all ones: for the raw (high) word b that is  b >= POS as a signed
// integer (positive values) or b >= NEG as an unsigned one (negative values) - two running maxima, which the compiler
// folds into v_max3_i32 / v_max3_u32: one instruction per value checked.
// (Under -ffinite-math-only a float value carries "never NaN / inf" facts that a future optimiser might follow through the
// bit cast; routing the word through an empty asm would stop that and costs a register copy per value - +30 instructions per
// roll-out step.  Instead mp_model_specialize CHECKS every code object it loads: a NaN row must come back NaN, or the
// specialised kernels are refused and the generic ones, built without that flag, serve - csrc/mp_capi.cpp.)
MP_HD int mp_hi_word(float x) { return __builtin_bit_cast(int, x); }
MP_HD int mp_hi_word(double x) { return (int)(__builtin_bit_cast(long long, x) >> 32); }
template <typename T> struct MpBadBits;
template <> struct MpBadBits<float> { static constexpr int POS = 0x7f800000; static constexpr unsigned NEG = 0xff800000u; };
template <> struct MpBadBits<double> { static constexpr int POS = 0x7ff00000; static constexpr unsigned NEG = 0xfff00000u; };

template <typename T>
struct MpBad {  // T = float / double
  int ms = 0;
  unsigned mu = 0;
  MP_HD void add(T v) {
    const int b = mp_hi_word(v);
    ms = b > ms ? b : ms;
    mu = (unsigned)b > mu ? (unsigned)b : mu;
  }
  template <int N> MP_HD void add(const T (&v)[N]) {
#pragma unroll
    for (int i = 0; i < N; ++i) add(v[i]);
  }
  MP_HD bool any() const { return ms >= MpBadBits<T>::POS || mu >= MpBadBits<T>::NEG; }
};
MP_HD void mp_poison_if(bool bad, float& v) {
  const unsigned b = __builtin_bit_cast(unsigned, v);
  v = __builtin_bit_cast(float, bad ? 0x7fc00000u : b);
}
MP_HD void mp_poison_if(bool bad, double& v) {
  const unsigned long long b = __builtin_bit_cast(unsigned long long, v);
  v = __builtin_bit_cast(double, bad ? 0x7ff8000000000000ull : b);
}
#if MP_HAS_PACKED
template <>
struct MpBad<mp_f2> {  // one verdict per packed row
  MpBad<float> x, y;
  MP_HD void add(mp_f2 v) { x.add(v.x); y.add(v.y); }
  template <int N> MP_HD void add(const mp_f2 (&v)[N]) {
#pragma unroll
    for (int i = 0; i < N; ++i) add(v[i]);
  }
};
#endif
template <typename T, int N>
MP_HD void mp_poison_if(bool bad, T (&v)[N]) {
#pragma unroll
  for (int i = 0; i < N; ++i) mp_poison_if(bad, v[i]);
}

// ------------------------------------------------------------------------------ model access
// Joint i of a model view.  By-value models (kernel arguments, constexpr literals, host structs): the member itself.  A
// model read through a constant-address-space POINTER (the generic one-row-per-lane kernels, MpModelConstF below) launders the pointer
// first: the 16 constants of a joint are then scalar-loaded where the joint's code uses them instead of all at once at the
// top of the kernel - 185 dwords of model do not fit the ~100 SGPRs a wave has, and what does not fit is parked in VGPR
// lanes (v_writelane / v_readlane: 60 of the 1203 instructions of the generic one-row kernel).
template <typename MT>
MP_HD const auto& mp_joint_of(const MT& M, int i) { return M.j[i]; }
#if defined(__HIP_DEVICE_COMPILE__)
typedef const __attribute__((address_space(4))) MpModel<float> MpModelConstF;
typedef const __attribute__((address_space(4))) MpModel<double> MpModelConstD;
__device__ __forceinline__ const __attribute__((address_space(4))) MpJoint<float>& mp_joint_of(MpModelConstF& M, int i) {
  MpModelConstF* p = &M;
  asm volatile("" : "+s"(p));  // opaque to CSE / hoisting: joint i's constants are loaded here
  return p->j[i];
}
__device__ __forceinline__ const __attribute__((address_space(4))) MpJoint<double>& mp_joint_of(MpModelConstD& M, int i) {
  MpModelConstD* p = &M;
  asm volatile("" : "+s"(p));
  return p->j[i];
}
typedef const __attribute__((address_space(4))) MpModelRev<float> MpModelRevConstF;   // (revolute joints only: mp_model.h)
template <> struct MpAllRevolute<MpModelRevConstF> { static constexpr bool value = true; };
template <> struct MpAllRevolute<__attribute__((address_space(4))) MpModelRev<float>> { static constexpr bool value = true; };   // (`const MT&` deduces MT without the const)
__device__ __forceinline__ const __attribute__((address_space(4))) MpJoint<float>& mp_joint_of(MpModelRevConstF& M, int i) {
  MpModelRevConstF* p = &M;
  asm volatile("" : "+s"(p));
  return p->j[i];
}
#endif

// ------------------------------------------------------------------------------ axis-aligned steps
// Motion vector (w, v), parent -> child coordinates, child pose in parent = (E, r):
//     w' = E^T w,  v' = E^T (v + w x r).
// step A: E = Rx(alpha), r = (a, 0, 0);   step B: E = Rz(theta), r = (0, 0, d).
template <typename T, typename S>
MP_HD void mp_motion_A(S ca, S sa, S a, T& wx, T& wy, T& wz, T& vx, T& vy, T& vz) {
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
template <typename T, typename S>
MP_HD void mp_force_down_A(S ca, S sa, S a, T& nx, T& ny, T& nz, T& fx, T& fy, T& fz) {
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
template <typename T, typename S>
MP_HD void mp_force_up_A(S ca, S sa, S a, T& nx, T& ny, T& nz, T& fx, T& fy, T& fz) {
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

// `MT` is any view of an MpModel<S>: the by-value kernarg struct, or the same struct behind a
// constant-address-space pointer (persistent kernels re-read it with scalar loads every iteration).
template <typename T, int N, typename MT>
MP_HD void mp_joint_state(const MT& M, const T (&q)[N], MpJointState<T, N>& js) {
  constexpr bool kAllRev = MpAllRevolute<MT>::value;   // (the model type says: revolute joints only - `rev` is 1)
#pragma unroll
  for (int i = 0; i < N; ++i) {
    const auto& J = mp_joint_of(M, i);
    const T qr = kAllRev ? q[i] : J.rev * q[i];
    // sin / cos of off + q through the constant rotation (co, so) - see MpJoint: exact for right-angle offsets (a swap / a sign,
    // folded away in the robot-specialised kernels), one rounding per product otherwise; q is never added to anything first
    T s0, c0;
    mp_sincos(qr, s0, c0);
    js.s[i] = s0 * J.co + c0 * J.so;
    js.c[i] = c0 * J.co - s0 * J.so;
    if (kAllRev) js.d[i] = MpTraits<T>::splat(J.d);
    else js.d[i] = J.d + (q[i] - qr);
  }
}

// Recursive Newton-Euler in the compiled link frames.  tau is NOT clipped here.
// a0: linear acceleration of the base in the pre-joint-1 frame (= base_R^T (-g)), wave-uniform.
// tipn / tipf: the tip wrench [moment; force] expressed in the pre-joint-1 frame, per row (ignored unless HAS_FTIP).
// `sc`: what the recursion shows of its intermediate wrenches on the way - MpNoScale for a plain recursion; MpRowScale keeps the
// size of the terms a float32 row's rounding errors scale with (mp_id_row_is_hard below compares it with the row's own torques):
//   sc.body(i, n, f)   link i's own body wrench (forward pass),
//   sc.joint(i, n, f)  the wrench joint i transmits, in link frame i (backward pass, before it is moved to the parent),
//   sc.child(i, n, f)  the same wrench as link i - 1 receives it.
struct MpNoScale {
  template <typename T> MP_HD void joint(int, const T&, const T&, const T&, const T&, const T&, const T&) {}
  template <typename T> MP_HD void body(int, const T&, const T&, const T&, const T&, const T&, const T&) {}
  template <typename T> MP_HD void child(int, const T&, const T&, const T&, const T&, const T&, const T&) {}
};
// The scale of a row's large intermediate terms, from values the recursion needs anyway (round 4 took the largest force component
// of joint 1, which a plain recursion of a robot without a lever at joints 0 and 1 never forms: 26 instructions of resurrected
// dead code on the UR5, tools/rule_sweep.py): the moments that meet at link J = 1 - the link's own and the one joint 2 hands down,
// whose SUM is what the two base torques are read from - and the force through joint 2 (times a length: the lever products
// further out are parts of it).  One v_max3_f32 each.  Models of one or two joints: what exists of the three.
template <typename T, int N>
struct MpRowScale {
  static constexpr int J = N > 1 ? 1 : 0;           // the link where the moments are looked at
  static constexpr int JF = N > 2 ? 2 : N - 1;      // the joint whose force is looked at
  T bm, cm, f;
  static MP_HD T inf3(const T& x, const T& y, const T& z) { return mp_max(mp_max(mp_abs(x), mp_abs(y)), mp_abs(z)); }
  MP_HD void body(int i, const T& nx, const T& ny, const T& nz, const T&, const T&, const T&) {
    if (i == J) bm = inf3(nx, ny, nz);
  }
  MP_HD void joint(int i, const T&, const T&, const T&, const T& fx, const T& fy, const T& fz) {
    if (i == JF) f = inf3(fx, fy, fz);
  }
  MP_HD void child(int i, const T& nx, const T& ny, const T& nz, const T&, const T&, const T&) {
    if (i == J + 1) cm = inf3(nx, ny, nz);
  }
  // lscale: the robot's longest joint-to-joint offset
  template <typename S> MP_HD T scale(S lscale) const {
    const T lf = f * MpTraits<T>::splat(lscale);
    if (N > J + 1) return mp_max(mp_max(bm, cm), lf);
    return mp_max(bm, lf);
  }
};
template <typename T, int N, bool HAS_FTIP, typename SC, typename MT>
MP_HD void mp_rnea_impl(const MT& M, const typename MpTraits<T>::S (&a0)[3], const T (&tipn)[3], const T (&tipf)[3],
                        const MpJointState<T, N>& js, const T (&qd)[N], const T (&qdd)[N], T (&tau)[N], SC& sc) {
  using S = typename MpTraits<T>::S;
  using TR = MpTraits<T>;
  const T zero = TR::splat(S(0));
  T fnx[N], fny[N], fnz[N], ffx[N], ffy[N], ffz[N];
  T wx = zero, wy = zero, wz = zero, vx = zero, vy = zero, vz = zero;
  T dwx = zero, dwy = zero, dwz = zero, dvx = TR::splat(a0[0]), dvy = TR::splat(a0[1]), dvz = TR::splat(a0[2]);
  T tnx = zero, tny = zero, tnz = zero, tfx = zero, tfy = zero, tfz = zero;
  if (HAS_FTIP) { tnx = tipn[0]; tny = tipn[1]; tnz = tipn[2]; tfx = tipf[0]; tfy = tipf[1]; tfz = tipf[2]; }

  // forward pass: twists, accelerations, body wrenches
#pragma unroll
  for (int i = 0; i < N; ++i) {
    const auto& J = mp_joint_of(M, i);
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
    if (MpAllRevolute<MT>::value) {   // (compile time: the model type says revolute joints only)
      const T qdr = qd[i], ar = qdd[i];
      wz += qdr;
      dwx += qdr * wy;
      dwy -= qdr * wx;
      dwz += ar;
      dvx += qdr * vy;
      dvy -= qdr * vx;
    } else {
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
    }

    // momentum P = G V = [Io w + h x v ; m v - h x w]
    const T pnx = J.Ixx * wx + J.Ixy * wy + J.Ixz * wz + (J.hy * vz - J.hz * vy);
    const T pny = J.Ixy * wx + J.Iyy * wy + J.Iyz * wz + (J.hz * vx - J.hx * vz);
    const T pnz = J.Ixz * wx + J.Iyz * wy + J.Izz * wz + (J.hx * vy - J.hy * vx);
    const T pfx = J.m * vx - (J.hy * wz - J.hz * wy);
    const T pfy = J.m * vy - (J.hz * wx - J.hx * wz);
    const T pfz = J.m * vz - (J.hx * wy - J.hy * wx);
    // F = G dV + [w x Pn + v x Pf ; w x Pf]
    // (Round 6 tried per-link wave-uniform branches on constants that vanish - products of inertia, the first moment: 12 + 30
    // instructions a link - in the generic kernels: the branchy code spills under their 96-VGPR cap, c2 0.103 -> 0.178 ms, c4 x 7;
    // profiles/r06_generic_ab.txt.  The robot-specialised programs fold those constants anyway.)
    fnx[i] = J.Ixx * dwx + J.Ixy * dwy + J.Ixz * dwz + (J.hy * dvz - J.hz * dvy) + (wy * pnz - wz * pny) + (vy * pfz - vz * pfy);
    fny[i] = J.Ixy * dwx + J.Iyy * dwy + J.Iyz * dwz + (J.hz * dvx - J.hx * dvz) + (wz * pnx - wx * pnz) + (vz * pfx - vx * pfz);
    fnz[i] = J.Ixz * dwx + J.Iyz * dwy + J.Izz * dwz + (J.hx * dvy - J.hy * dvx) + (wx * pny - wy * pnx) + (vx * pfy - vy * pfx);
    ffx[i] = J.m * dvx - (J.hy * dwz - J.hz * dwy) + (wy * pfz - wz * pfy);
    ffy[i] = J.m * dvy - (J.hz * dwx - J.hx * dwz) + (wz * pfx - wx * pfz);
    ffz[i] = J.m * dvz - (J.hx * dwy - J.hy * dwx) + (wx * pfy - wy * pfx);
    sc.body(i, fnx[i], fny[i], fnz[i], ffx[i], ffy[i], ffz[i]);
  }
  if (HAS_FTIP) {  // Js^T Ftip: the space-frame wrench, now expressed in link frame N, rides the backward pass
    fnx[N - 1] += tnx; fny[N - 1] += tny; fnz[N - 1] += tnz;
    ffx[N - 1] += tfx; ffy[N - 1] += tfy; ffz[N - 1] += tfz;
  }
  // backward pass
#pragma unroll
  for (int i = N - 1; i >= 0; --i) {
    const auto& J = mp_joint_of(M, i);
    if (MpAllRevolute<MT>::value) tau[i] = fnz[i];
    else tau[i] = J.rev * fnz[i] + (S(1) - J.rev) * ffz[i];
    sc.joint(i, fnx[i], fny[i], fnz[i], ffx[i], ffy[i], ffz[i]);
    if (i > 0) {
      T nx = fnx[i], ny = fny[i], nz = fnz[i], fx = ffx[i], fy = ffy[i], fz = ffz[i];
      mp_force_up_B(js.c[i], js.s[i], js.d[i], nx, ny, nz, fx, fy, fz);
      mp_force_up_A(J.ca, J.sa, J.a, nx, ny, nz, fx, fy, fz);
      sc.child(i, nx, ny, nz, fx, fy, fz);   // joint i's wrench as link i - 1 receives it
      fnx[i - 1] += nx; fny[i - 1] += ny; fnz[i - 1] += nz;
      ffx[i - 1] += fx; ffy[i - 1] += fy; ffz[i - 1] += fz;
    }
  }
}

template <typename T, int N, bool HAS_FTIP, typename MT>
MP_HD void mp_rnea(const MT& M, const typename MpTraits<T>::S (&a0)[3], const T (&tipn)[3], const T (&tipf)[3],
                   const MpJointState<T, N>& js, const T (&qd)[N], const T (&qdd)[N], T (&tau)[N]) {
  MpNoScale sc;
  mp_rnea_impl<T, N, HAS_FTIP>(M, a0, tipn, tipf, js, qd, qdd, tau, sc);
}

// Same with the wave-uniform per-call constants (gravity + one tip wrench for every row).
template <typename T, int N, bool HAS_FTIP, typename MT>
MP_HD void mp_rnea(const MT& M, const MpCall<typename MpTraits<T>::S>& C, const MpJointState<T, N>& js,
                   const T (&qd)[N], const T (&qdd)[N], T (&tau)[N]) {
  using TR = MpTraits<T>;
  const T tn[3] = {TR::splat(C.F1n[0]), TR::splat(C.F1n[1]), TR::splat(C.F1n[2])};
  const T tf[3] = {TR::splat(C.F1f[0]), TR::splat(C.F1f[1]), TR::splat(C.F1f[2])};
  mp_rnea<T, N, HAS_FTIP>(M, C.a0, tn, tf, js, qd, qdd, tau);
}

// ------------------------------------------------------------- float32 rows, adaptive precision
// A float32 recursion carries ~1 ulp of its LARGEST intermediate terms into every torque.  Almost always that is far inside the
// parity bound (1e-4 |ref| + 5e-6 max|row|); it is not on the few rows per thousand whose torques are a small difference of large
// terms - an arm swinging at 10 rad/s, or balanced near upright, whose joint wrenches reach tens or hundreds of N.m while every
// torque of the row is a fraction of one (over c2's 12.3 M rows the plain float32 kernel missed the bound on 76, by up to 4.8 x;
// with the joint offsets taken exactly - MpJoint co / so - on 1, by 1.3 x; the torque that misses is almost always the shoulder's,
// read from the sum of the upper arm's own moment and the one the forearm hands down).  Such a row announces itself: the scale of
// its intermediate terms (MpRowScale above) is more than MP_HARD_ROW_K x the row's largest torque.  Those rows - 0.6 % of
// c2-distributed rows, in runs of consecutive timesteps - are evaluated again in float64 from the same float32 inputs: by a pass of
// their own behind the float32 kernel (mp_body_id_hard, csrc/mp_bodies.h) or, where no list is attached, in place (mp_rnea_cold).
// What stays float32 sits at <= 0.36 x the bound on 12.3 M rows, the re-evaluated rows at <= 0.01 x (tools/rule_sweep.py,
// profiles/r05_rule_sweep.txt).  Deterministic: a row's precision depends on that row's values only.
#ifndef MP_HARD_ROW_K
#define MP_HARD_ROW_K 8.0f
#endif
#ifndef MP_ADAPTIVE_F32   // 0 (experiment switch): plain float32 rows, for A/B measurements of what the test and the float64 rows cost
#define MP_ADAPTIVE_F32 1
#endif
template <int N>
MP_HD bool mp_id_row_is_hard(const float (&tau)[N], float scale) {
  float rowmax = mp_abs(tau[0]);
#pragma unroll
  for (int i = 1; i < N; ++i) rowmax = mp_max(rowmax, mp_abs(tau[i]));
  return scale > MP_HARD_ROW_K * rowmax;   // (callers rule NaN rows out themselves)
}

// The float32 recursion of one row + its verdict.  tau is NOT clipped.
template <int N, bool HAS_FTIP, typename MT>
MP_HD bool mp_rnea_f32(const MT& M, const MpCall<float>& C, const MpJointState<float, N>& js, const float (&qd)[N],
                       const float (&qdd)[N], float (&tau)[N]) {
#if MP_ADAPTIVE_F32
  const float tn[3] = {C.F1n[0], C.F1n[1], C.F1n[2]}, tf[3] = {C.F1f[0], C.F1f[1], C.F1f[2]};
  MpRowScale<float, N> sc;
  mp_rnea_impl<float, N, HAS_FTIP>(M, C.a0, tn, tf, js, qd, qdd, tau, sc);
  return mp_id_row_is_hard<N>(tau, sc.scale(M.lscale));
#else
  mp_rnea<float, N, HAS_FTIP>(M, C, js, qd, qdd, tau);
  return false;
#endif
}

// Where the re-evaluation keeps its per-joint state: 8 N float64 (six body / joint wrench components, sin, cos) + 4 N float32 (the
// joint's shift, the row's q / qd / qdd, tau on the way out).
// On the device that is a slot of the wave's LDS - the re-evaluation wants ~170 VGPRs with these in registers, the kernels that
// host it are held to 80 - 110 by their launch bounds, and spilling them to scratch memory made the waves that take this path
// stragglers (54 scratch round trips each) and, past 140 MB of scratch per dispatch, every launch allocate its own: c2 +19 %, c4
// +43 % (profiles/r04_adaptive_ab.txt).  On the host plain arrays.
template <int N> struct MpColdSlot { static constexpr int BYTES = 8 * N * 8 + 4 * N * 4; };
template <int N>
struct MpColdLocal {
  double f[8 * N];
  float j[4 * N];
  MP_HD double getf(int k) const { return f[k]; }
  MP_HD void putf(int k, double v) { f[k] = v; }
  MP_HD float getj(int k) const { return j[k]; }
  MP_HD void putj(int k, float v) { j[k] = v; }
};
#if defined(__HIP_DEVICE_COMPILE__)
#define MP_LDS_AS __attribute__((address_space(3)))   // the slot is LDS: said in the pointer's type, or the accesses become FLAT ones
#else
#define MP_LDS_AS
#endif
template <int N>
struct MpColdMem {  // a slot in (LDS) memory: [8 N doubles][4 N floats].  Volatile: a lane reads back only what it wrote itself, so
  MP_LDS_AS char* p;  // the optimiser would otherwise forward every value in a register and delete the stores - the spill again
  MP_HD double getf(int k) const { return reinterpret_cast<const volatile MP_LDS_AS double*>(p)[k]; }
  MP_HD void putf(int k, double v) { reinterpret_cast<volatile MP_LDS_AS double*>(p)[k] = v; }
  MP_HD float getj(int k) const { return reinterpret_cast<const volatile MP_LDS_AS float*>(p + 8 * N * 8)[k]; }
  MP_HD void putj(int k, float v) { reinterpret_cast<volatile MP_LDS_AS float*>(p + 8 * N * 8)[k] = v; }
};

// mp_rnea_impl in float64 throughout (sin / cos included) from the row's float32 inputs; `M` is the float32 model or, in the
// robot-specialised programs and the CPU launchers, its float64 original; the per-joint state lives in `st`.  tau is NOT clipped.  The joint loops are ROLLED on purpose
// (the model is indexed at run time: scalar loads with a register offset, from the kernel arguments, the device copy or the
// specialised literal alike): unrolled, the scheduler interleaves the joints' float64 work into 150 - 200 live VGPRs, which under
// the hosting kernels' 80 is ~80 scratch round trips; rolled it is one joint's worth of state at a time, and ~170 instructions
// of code instead of ~1400.
#if defined(__clang__)
#define MP_ROLLED _Pragma("clang loop unroll(disable)")
#else
#define MP_ROLLED
#endif
// ... and inside a joint the scheduler is told not to move anything across the stage boundaries below: left alone it overlaps the
// six wrench components with the chain update for latency's sake and needs ~96 - 116 VGPRs, in program order ~70
#if defined(__HIP_DEVICE_COMPILE__)
#define MP_STAGE() __builtin_amdgcn_sched_barrier(0)
#else
#define MP_STAGE() ((void)0)
#endif
template <int N, bool HAS_FTIP, typename MT, typename ST>
MP_HD void mp_rnea_cold(const MT& M, const MpCall<float>& C, const float (&q)[N], const float (&qd)[N], const float (&qdd)[N], ST st,
                        float (&tau)[N]) {
  double wx = 0, wy = 0, wz = 0, vx = 0, vy = 0, vz = 0;
  double dwx = 0, dwy = 0, dwz = 0, dvx = (double)C.a0[0], dvy = (double)C.a0[1], dvz = (double)C.a0[2];
  double tnx = 0, tny = 0, tnz = 0, tfx = 0, tfy = 0, tfz = 0;
  if (HAS_FTIP) { tnx = (double)C.F1n[0]; tny = (double)C.F1n[1]; tnz = (double)C.F1n[2]; tfx = (double)C.F1f[0]; tfy = (double)C.F1f[1]; tfz = (double)C.F1f[2]; }
  // the row's inputs are read through `st` too (indexed by the loop variable): registers cannot be indexed at run time
#pragma unroll
  for (int i = 0; i < N; ++i) { st.putj(N + i, q[i]); st.putj(2 * N + i, qd[i]); st.putj(3 * N + i, qdd[i]); }
  MP_ROLLED
  for (int i = 0; i < N; ++i) {
    const auto& J = mp_joint_of(M, i);
    const float qi = st.getj(N + i);
    double s, c, d;
    {  // sin / cos in float64: on these rows the float32 pair's last bit is the largest single error (0.35 x the bound against 0.09)
      const double qr = (double)J.rev * (double)qi;
      double s0, c0;
      mp_sincos(qr, s0, c0);
      s = s0 * (double)J.co + c0 * (double)J.so;
      c = c0 * (double)J.co - s0 * (double)J.so;
      d = (double)J.d + ((double)qi - qr);
      st.putf(6 * N + i, s); st.putf(7 * N + i, c); st.putj(i, (float)d);
    }
    MP_STAGE();
    const double ca = (double)J.ca, sa = (double)J.sa, la = (double)J.a;
    if (i > 0) {
      mp_motion_A<double, double>(ca, sa, la, wx, wy, wz, vx, vy, vz);
      mp_motion_A<double, double>(ca, sa, la, dwx, dwy, dwz, dvx, dvy, dvz);
      if (HAS_FTIP) mp_force_down_A<double, double>(ca, sa, la, tnx, tny, tnz, tfx, tfy, tfz);
    }
    mp_motion_B<double>(c, s, d, wx, wy, wz, vx, vy, vz);
    mp_motion_B<double>(c, s, d, dwx, dwy, dwz, dvx, dvy, dvz);
    if (HAS_FTIP) mp_force_down_B<double>(c, s, d, tnx, tny, tnz, tfx, tfy, tfz);
    const double rev = (double)J.rev, qdi = (double)st.getj(2 * N + i), qddi = (double)st.getj(3 * N + i);
    const double qdr = rev * qdi, qdp = qdi - qdr;
    const double ar = rev * qddi, ap = qddi - ar;
    wz += qdr;
    vz += qdp;
    dwx += qdr * wy;
    dwy -= qdr * wx;
    dwz += ar;
    dvx += qdr * vy + qdp * wy;
    dvy -= qdr * vx + qdp * wx;
    dvz += ap;
    MP_STAGE();
    const double Ixx = (double)J.Ixx, Ixy = (double)J.Ixy, Ixz = (double)J.Ixz, Iyy = (double)J.Iyy, Iyz = (double)J.Iyz, Izz = (double)J.Izz;
    const double hx = (double)J.hx, hy = (double)J.hy, hz = (double)J.hz, m = (double)J.m;
    const double pnx = Ixx * wx + Ixy * wy + Ixz * wz + (hy * vz - hz * vy);
    const double pny = Ixy * wx + Iyy * wy + Iyz * wz + (hz * vx - hx * vz);
    const double pnz = Ixz * wx + Iyz * wy + Izz * wz + (hx * vy - hy * vx);
    const double pfx = m * vx - (hy * wz - hz * wy);
    const double pfy = m * vy - (hz * wx - hx * wz);
    const double pfz = m * vz - (hx * wy - hy * wx);
    MP_STAGE();
    st.putf(6 * i + 0, Ixx * dwx + Ixy * dwy + Ixz * dwz + (hy * dvz - hz * dvy) + (wy * pnz - wz * pny) + (vy * pfz - vz * pfy));
    MP_STAGE();
    st.putf(6 * i + 1, Ixy * dwx + Iyy * dwy + Iyz * dwz + (hz * dvx - hx * dvz) + (wz * pnx - wx * pnz) + (vz * pfx - vx * pfz));
    MP_STAGE();
    st.putf(6 * i + 2, Ixz * dwx + Iyz * dwy + Izz * dwz + (hx * dvy - hy * dvx) + (wx * pny - wy * pnx) + (vx * pfy - vy * pfx));
    MP_STAGE();
    st.putf(6 * i + 3, m * dvx - (hy * dwz - hz * dwy) + (wy * pfz - wz * pfy));
    st.putf(6 * i + 4, m * dvy - (hz * dwx - hx * dwz) + (wz * pfx - wx * pfz));
    st.putf(6 * i + 5, m * dvz - (hx * dwy - hy * dwx) + (wx * pfy - wy * pfx));
    MP_STAGE();
  }
  // the children's wrench, already in the current joint's frame (the tip wrench rides it from the start)
  double ax = tnx, ay = tny, az = tnz, bx = tfx, by = tfy, bz = tfz;
  MP_ROLLED
  for (int i = N - 1; i >= 0; --i) {
    const auto& J = mp_joint_of(M, i);
    double nx = st.getf(6 * i + 0) + ax, ny = st.getf(6 * i + 1) + ay, nz = st.getf(6 * i + 2) + az;
    double fx = st.getf(6 * i + 3) + bx, fy = st.getf(6 * i + 4) + by, fz = st.getf(6 * i + 5) + bz;
    const double rev = (double)J.rev;
    st.putj(N + i, (float)(rev * nz + (1.0 - rev) * fz));   // tau_i, in the slot q_i occupied
    if (i > 0) {
      const double s = st.getf(6 * N + i), c = st.getf(7 * N + i), d = (double)st.getj(i);
      mp_force_up_B<double>(c, s, d, nx, ny, nz, fx, fy, fz);
      mp_force_up_A<double, double>((double)J.ca, (double)J.sa, (double)J.a, nx, ny, nz, fx, fy, fz);
      ax = nx; ay = ny; az = nz; bx = fx; by = fy; bz = fz;
    }
  }
#pragma unroll
  for (int i = 0; i < N; ++i) tau[i] = st.getj(N + i);
}

// Host form (CPU launchers, tests/hostsim): tau (unclipped) of one row given its joint state - float64 rows are the recursion
// itself; float32 rows the float32 recursion and, where the row is ill-conditioned, the re-evaluation above.  The kernels do the
// same through mp_rnea_f32 + mp_cold_rows (csrc/mp_bodies.h), which runs the re-evaluation wave by wave out of LDS.
template <typename T> struct MpIsF32 { static constexpr bool value = false; };
template <> struct MpIsF32<float> { static constexpr bool value = true; };
template <typename T, int N, bool HAS_FTIP, typename MT>
MP_HD void mp_rnea_row(const MT& M, const MpCall<typename MpTraits<T>::S>& C, const MpJointState<T, N>& js, const T (&q)[N],
                       const T (&qd)[N], const T (&qdd)[N], T (&tau)[N]) {
  if constexpr (MpIsF32<T>::value) {
    if (mp_rnea_f32<N, HAS_FTIP>(M, C, js, qd, qdd, tau)) {
      MpColdLocal<N> st;
      if (C.cold_model) mp_rnea_cold<N, HAS_FTIP>(*static_cast<const MpModel<double>*>(C.cold_model), C, q, qd, qdd, st, tau);
      else mp_rnea_cold<N, HAS_FTIP>(M, C, q, qd, qdd, st, tau);
    }
  } else {
    mp_rnea<T, N, HAS_FTIP>(M, C, js, qd, qdd, tau);
  }
}

// A space-frame wrench [m; f] seen from the pre-joint-1 frame: f' = R^T f, n' = R^T (n - p x f)
// (what mp_make_call does on the host for a per-call wrench; here per row).
template <typename T, typename MT>
MP_HD void mp_wrench_to_frame1(const MT& M, const T (&F)[6], T (&tn)[3], T (&tf)[3]) {
  const auto& R = M.base_R;
  const auto& p = M.base_p;
  const T nx = F[0] - (p[1] * F[5] - p[2] * F[4]), ny = F[1] - (p[2] * F[3] - p[0] * F[5]), nz = F[2] - (p[0] * F[4] - p[1] * F[3]);
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    tn[k] = R[0 + k] * nx + R[3 + k] * ny + R[6 + k] * nz;
    tf[k] = R[0 + k] * F[3] + R[3 + k] * F[4] + R[6 + k] * F[5];
  }
}

// ------------------------------------------------------------- mass matrix / forward dynamics
// M(q) column j = ID(q, 0, e_j, g = 0, F = 0): with the loops unrolled the zero velocities and the unit
// acceleration are compile-time constants, so the velocity products and every term upstream of joint j
// fold away.  Symmetrised like the reference (dynamics/mass_matrix.py:96).
template <typename T, int N, typename MT>
MP_HD void mp_mass_matrix(const MT& M, const MpJointState<T, N>& js, T (&Mq)[N][N]) {
  using S = typename MpTraits<T>::S;
  using TR = MpTraits<T>;
  const S a0[3] = {S(0), S(0), S(0)};
  const T zero = TR::splat(S(0));
  const T z3[3] = {zero, zero, zero};
  T col[N][N];
#pragma unroll
  for (int j = 0; j < N; ++j) {
    T qd[N], qdd[N];
#pragma unroll
    for (int k = 0; k < N; ++k) { qd[k] = zero; qdd[k] = (k == j) ? TR::splat(S(1)) : zero; }
    mp_rnea<T, N, false>(M, a0, z3, z3, js, qd, qdd, col[j]);
  }
#pragma unroll
  for (int i = 0; i < N; ++i)
#pragma unroll
    for (int j = 0; j < N; ++j) Mq[i][j] = S(0.5) * (col[j][i] + col[i][j]);
}

// Composite-rigid-body form of the same matrix (Featherstone's CRBA in the compiled link frames).  A rigid-body
// inertia about a frame origin is (m, h = m c, I_o): 10 numbers, and the sum of rigid bodies is a rigid body, so
// the composite inertia of links i..n stays in that form.  Moving it from child to parent coordinates across an
// axis-aligned step costs one planar rotation of (h, I_o) plus a parallel-axis shift along one axis:
//     h' = R h + m r,   I' = R I R^T + (2 (Rh).r) 1 - (Rh) r^T - r (Rh)^T + m (|r|^2 1 - r r^T).
// Column i of M is then S_i^T of the force I^c_i S_i carried up the chain with the same force transforms the
// Newton-Euler backward pass uses.  ~750 instructions at n = 6 against ~3300 for n unit-acceleration recursions.
template <typename T>
struct MpRbi {  // rigid-body inertia about the current frame's origin, in that frame's coordinates
  T m, hx, hy, hz, xx, xy, xz, yy, yz, zz;
};

// child -> parent across Rz(theta) Tz(d):  R = Rz, r = (0, 0, d)
template <typename T>
MP_HD void mp_rbi_up_B(T c, T s, T d, MpRbi<T>& I) {
  const T hx = c * I.hx - s * I.hy, hy = s * I.hx + c * I.hy;
  const T cc = c * c, ss = s * s, sc = s * c;
  const T xx = cc * I.xx - (sc + sc) * I.xy + ss * I.yy;
  const T yy = ss * I.xx + (sc + sc) * I.xy + cc * I.yy;
  const T xy = sc * (I.xx - I.yy) + (cc - ss) * I.xy;
  const T xz = c * I.xz - s * I.yz, yz = s * I.xz + c * I.yz;
  const T t = d * (I.hz + I.hz) + I.m * d * d;  // 2 (Rh).r + m |r|^2
  I.xx = xx + t; I.yy = yy + t; I.xy = xy;
  I.xz = xz - d * hx; I.yz = yz - d * hy;       // - (Rh) r^T - r (Rh)^T off-diagonals; zz: +2 d hz - 2 d hz = 0
  I.hx = hx; I.hy = hy; I.hz = I.hz + I.m * d;
}
// child -> parent across Rx(alpha) Tx(a):  R = Rx, r = (a, 0, 0)
template <typename T, typename S>
MP_HD void mp_rbi_up_A(S ca, S sa, S a, MpRbi<T>& I) {
  const T hy = ca * I.hy - sa * I.hz, hz = sa * I.hy + ca * I.hz;
  const S cc = ca * ca, ss = sa * sa, sc = sa * ca;
  const T yy = cc * I.yy - (sc + sc) * I.yz + ss * I.zz;
  const T zz = ss * I.yy + (sc + sc) * I.yz + cc * I.zz;
  const T yz = sc * (I.yy - I.zz) + (cc - ss) * I.yz;
  const T xy = ca * I.xy - sa * I.xz, xz = sa * I.xy + ca * I.xz;
  const T t = a * (I.hx + I.hx) + I.m * (a * a);
  I.yy = yy + t; I.zz = zz + t; I.yz = yz;
  I.xy = xy - a * hy; I.xz = xz - a * hz;
  I.hy = hy; I.hz = hz; I.hx = I.hx + I.m * a;
}

template <typename T, int N, typename MT>
MP_HD void mp_mass_matrix_crba(const MT& M, const MpJointState<T, N>& js, T (&Mq)[N][N]) {
  using S = typename MpTraits<T>::S;
  using TR = MpTraits<T>;
  const T zero = TR::splat(S(0));
  MpRbi<T> Ic;  // composite inertia of links i..N-1 in frame i
  Ic.m = zero; Ic.hx = zero; Ic.hy = zero; Ic.hz = zero;
  Ic.xx = zero; Ic.xy = zero; Ic.xz = zero; Ic.yy = zero; Ic.yz = zero; Ic.zz = zero;
#pragma unroll
  for (int i = N - 1; i >= 0; --i) {
    const auto& J = mp_joint_of(M, i);
    Ic.m = Ic.m + J.m; Ic.hx = Ic.hx + J.hx; Ic.hy = Ic.hy + J.hy; Ic.hz = Ic.hz + J.hz;
    Ic.xx = Ic.xx + J.Ixx; Ic.xy = Ic.xy + J.Ixy; Ic.xz = Ic.xz + J.Ixz; Ic.yy = Ic.yy + J.Iyy; Ic.yz = Ic.yz + J.Iyz;
    Ic.zz = Ic.zz + J.Izz;
    // F = Ic S_i: S = [z;0] (revolute) -> n = Io[:, z], f = -h x z ;  S = [0;z] (prismatic) -> n = h x z, f = m z
    const S r = J.rev, p = S(1) - J.rev;
    T nx = r * Ic.xz + p * Ic.hy, ny = r * Ic.yz - p * Ic.hx, nz = r * Ic.zz;
    T fx = -(r * Ic.hy), fy = r * Ic.hx, fz = p * Ic.m;
    Mq[i][i] = r * nz + p * fz;
#pragma unroll
    for (int k = i; k > 0; --k) {  // carry F from frame k to frame k-1, read the component along joint k-1
      const auto& Jk = mp_joint_of(M, k);
      mp_force_up_B(js.c[k], js.s[k], js.d[k], nx, ny, nz, fx, fy, fz);
      mp_force_up_A(Jk.ca, Jk.sa, Jk.a, nx, ny, nz, fx, fy, fz);
      const auto& Jp = mp_joint_of(M, k - 1);
      const T v = Jp.rev * nz + (S(1) - Jp.rev) * fz;
      Mq[k - 1][i] = v;
      Mq[i][k - 1] = v;
    }
    if (i > 0) {  // move the composite into frame i-1
      mp_rbi_up_B(js.c[i], js.s[i], js.d[i], Ic);
      mp_rbi_up_A(J.ca, J.sa, J.a, Ic);
    }
  }
}

// Solve M x = b for a symmetric positive definite M held in registers (Cholesky, fully unrolled).
// The reference calls np.linalg.solve (LU with pivoting, dynamics/id_fd.py:82); for an SPD matrix both
// give the same x up to rounding.  M is overwritten by its factor, b by the solution.
template <typename T, int N>
MP_HD void mp_spd_solve(T (&A)[N][N], T (&b)[N]) {
#pragma unroll
  for (int j = 0; j < N; ++j) {
    T d = A[j][j];
#pragma unroll
    for (int k = 0; k < j; ++k) d -= A[j][k] * A[j][k];
    const T inv = mp_rsqrt(d);
    A[j][j] = inv;  // store 1 / L_jj
#pragma unroll
    for (int i = j + 1; i < N; ++i) {
      T v = A[i][j];
#pragma unroll
      for (int k = 0; k < j; ++k) v -= A[i][k] * A[j][k];
      A[i][j] = v * inv;
    }
  }
#pragma unroll
  for (int i = 0; i < N; ++i) {  // L y = b
    T v = b[i];
#pragma unroll
    for (int k = 0; k < i; ++k) v -= A[i][k] * b[k];
    b[i] = v * A[i][i];
  }
#pragma unroll
  for (int i = N - 1; i >= 0; --i) {  // L^T x = y
    T v = b[i];
#pragma unroll
    for (int k = i + 1; k < N; ++k) v -= A[k][i] * b[k];
    b[i] = v * A[i][i];
  }
}

// qdd = M(q)^-1 (tau - bias), bias = ID(q, qd, 0, g, F)   (reference dynamics/id_fd.py:71-83)
template <typename T, int N, bool HAS_FTIP, typename MT>
MP_HD void mp_forward_dynamics(const MT& M, const typename MpTraits<T>::S (&a0)[3], const T (&tipn)[3],
                               const T (&tipf)[3], const T (&q)[N], const T (&qd)[N], const T (&tau)[N], T (&qdd)[N]) {
  using S = typename MpTraits<T>::S;
  MpJointState<T, N> js;
  mp_joint_state<T, N>(M, q, js);
  T zero_acc[N], bias[N];
#pragma unroll
  for (int k = 0; k < N; ++k) zero_acc[k] = MpTraits<T>::splat(S(0));
  mp_rnea<T, N, HAS_FTIP>(M, a0, tipn, tipf, js, qd, zero_acc, bias);
  T Mq[N][N];
  mp_mass_matrix_crba<T, N>(M, js, Mq);
#pragma unroll
  for (int k = 0; k < N; ++k) qdd[k] = tau[k] - bias[k];
  mp_spd_solve<T, N>(Mq, qdd);
}

// One closed-loop regulation run under joint-space PD torque (the inner loop of the reference's gain sweep,
// control/metrics.py:316-346): tau = Kp (des - theta) - Kd omega, alpha = forward dynamics without a tip wrench,
// omega += alpha dt, theta += omega dt, err[step] = |theta - des|_2; the run stops after a step > 10 whose error
// exceeds 1e10.  Returns the number of errors recorded.
template <typename T, int N, typename MT>
MP_HD int mp_pd_regulation_run(const MT& M, const typename MpTraits<T>::S (&a0)[3], const T (&theta0)[N], const T (&des)[N], T Kp,
                               T Kd, T dt, int steps, T* err) {
  T th[N], om[N], tau[N], al[N];
  const T z3[3] = {T(0), T(0), T(0)};
#pragma unroll
  for (int j = 0; j < N; ++j) { th[j] = theta0[j]; om[j] = T(0); }
  int done = 0;
  for (int step = 0; step < steps; ++step) {
#pragma unroll
    for (int j = 0; j < N; ++j) tau[j] = Kp * (des[j] - th[j]) - Kd * om[j];
    mp_forward_dynamics<T, N, false>(M, a0, z3, z3, th, om, tau, al);
    T e2 = T(0);
#pragma unroll
    for (int j = 0; j < N; ++j) {
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

// ------------------------------------------------------------------------- FK + space Jacobian
// T (4x4 row-major) = prod T_{i-1,i}(q_i) . tool ;  J (6 x N row-major), column i = [z_i ; o_i x z_i]
// (revolute) or [0 ; z_i] (prismatic), z_i / o_i = axis / origin of link frame i in the space frame —
// identical to Ad(prod_{j<i} exp) S_i because link frame i's z axis IS joint axis i.
template <typename T, int N, bool WANT_J, typename MT>
MP_HD void mp_fk_jac(const MT& M, const MpJointState<T, N>& js, T* Tout, T* Jout) {
  using S = typename MpTraits<T>::S;
  using TR = MpTraits<T>;
  // columns of R and origin p of the running frame
  T x0 = TR::splat(M.base_R[0]), x1 = TR::splat(M.base_R[3]), x2 = TR::splat(M.base_R[6]);
  T y0 = TR::splat(M.base_R[1]), y1 = TR::splat(M.base_R[4]), y2 = TR::splat(M.base_R[7]);
  T z0 = TR::splat(M.base_R[2]), z1 = TR::splat(M.base_R[5]), z2 = TR::splat(M.base_R[8]);
  T p0 = TR::splat(M.base_p[0]), p1 = TR::splat(M.base_p[1]), p2 = TR::splat(M.base_p[2]);
#pragma unroll
  for (int i = 0; i < N; ++i) {
    const auto& J = mp_joint_of(M, i);
    if (i > 0) {  // . Rx(alpha) Tx(a)
      p0 += J.a * x0; p1 += J.a * x1; p2 += J.a * x2;
      const T a0 = y0, a1 = y1, a2 = y2;
      y0 = J.ca * a0 + J.sa * z0; y1 = J.ca * a1 + J.sa * z1; y2 = J.ca * a2 + J.sa * z2;
      z0 = J.ca * z0 - J.sa * a0; z1 = J.ca * z1 - J.sa * a1; z2 = J.ca * z2 - J.sa * a2;
    }
    if (WANT_J) {  // the joint axis is fixed in the PARENT link: read it before the joint moves the frame
      // revolute: [z ; p x z] with p any point of the axis (the frame origin before Tz is on it)
      const T cx = p1 * z2 - p2 * z1, cy = p2 * z0 - p0 * z2, cz = p0 * z1 - p1 * z0;
      const S r = J.rev, pr = S(1) - J.rev;
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
  const auto& R = M.tool_R;
  const auto& t = M.tool_p;
  Tout[0] = x0 * R[0] + y0 * R[3] + z0 * R[6]; Tout[1] = x0 * R[1] + y0 * R[4] + z0 * R[7]; Tout[2] = x0 * R[2] + y0 * R[5] + z0 * R[8];
  Tout[4] = x1 * R[0] + y1 * R[3] + z1 * R[6]; Tout[5] = x1 * R[1] + y1 * R[4] + z1 * R[7]; Tout[6] = x1 * R[2] + y1 * R[5] + z1 * R[8];
  Tout[8] = x2 * R[0] + y2 * R[3] + z2 * R[6]; Tout[9] = x2 * R[1] + y2 * R[4] + z2 * R[7]; Tout[10] = x2 * R[2] + y2 * R[5] + z2 * R[8];
  Tout[3] = p0 + x0 * t[0] + y0 * t[1] + z0 * t[2];
  Tout[7] = p1 + x1 * t[0] + y1 * t[1] + z1 * t[2];
  Tout[11] = p2 + x2 * t[0] + y2 * t[1] + z2 * t[2];
  Tout[12] = TR::splat(S(0)); Tout[13] = TR::splat(S(0)); Tout[14] = TR::splat(S(0)); Tout[15] = TR::splat(S(1));
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

// ------------------------------------------------------------------------- Cartesian straight line
// cartesian_trajectory (reference planning/trajectory.py:504-594, :676-737): straight-line position,
// orientation R_start exp(log(R_start^T R_end) s).  float64 math, float32 rows.
// log: reference utils/so3.py:172-191 (+ :36-74 half-turn axis convention, :116-160 theta / sin theta);
// exp: utils/so3.py:199-237 (Rodrigues with the theta^2 < 1e-4 Taylor branch).
MP_HD void mp_log3(const double (&R)[9], double (&w)[3]) {
  const double tr = R[0] + R[4] + R[8];
  double cs = 0.5 * (tr - 1.0);
  cs = cs < -1.0 ? -1.0 : (cs > 1.0 ? 1.0 : cs);
  const double vee[3] = {R[7] - R[5], R[2] - R[6], R[3] - R[1]};
  double vv = vee[0] * vee[0] + vee[1] * vee[1] + vee[2] * vee[2];
  vv = vv < 1e-300 ? 1e-300 : vv;
  const double sn = 0.5 * sqrt(vv);
  const double theta = atan2(sn, cs);
  if (theta > 3.14159265358979323846 - 1e-2) {  // half-turn band: axis from the symmetric part
    const double s00 = R[0] - cs, s11 = R[4] - cs, s22 = R[8] - cs;
    const double s01 = 0.5 * (R[1] + R[3]), s02 = 0.5 * (R[2] + R[6]), s12 = 0.5 * (R[5] + R[7]);
    double c0, c1, c2, sref;
    if (s22 >= 1e-6) { c0 = s02; c1 = s12; c2 = s22; sref = vee[2]; }
    else if (s11 >= 1e-6) { c0 = s01; c1 = s11; c2 = s12; sref = vee[1]; }
    else { c0 = s00; c1 = s01; c2 = s02; sref = vee[0]; }
    double nn = c0 * c0 + c1 * c1 + c2 * c2;
    nn = nn < 1e-24 ? 1e-24 : nn;
    const double k = (sref >= 0.0 ? theta : -theta) / sqrt(nn);
    w[0] = k * c0; w[1] = k * c1; w[2] = k * c2;
    return;
  }
  const double u = 1.0 - cs;
  const double coef = (cs > 1.0 - 5e-5) ? 1.0 + u / 3.0 + 4.0 * u * u / 45.0 : acos(cs) / sqrt(1.0 - cs * cs);
  w[0] = 0.5 * coef * vee[0]; w[1] = 0.5 * coef * vee[1]; w[2] = 0.5 * coef * vee[2];
}

// out = Rs . exp([w])   (row-major 3x3)
MP_HD void mp_rot_times_exp3(const double (&Rs)[9], const double (&w)[3], double (&out)[9]) {
  const double t2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
  double A, B;
  if (t2 < 1e-4) {
    A = 1.0 - t2 / 6.0 + t2 * t2 / 120.0;
    B = 0.5 - t2 / 24.0 + t2 * t2 / 720.0;
  } else {
    const double t = sqrt(t2);
    A = sin(t) / t;
    B = (1.0 - cos(t)) / t2;
  }
  // E = I + A K + B K^2,  K = [w]x,  K^2 = w w^T - |w|^2 I
  const double E[9] = {1.0 + B * (w[0] * w[0] - t2), -A * w[2] + B * w[0] * w[1], A * w[1] + B * w[0] * w[2],
                       A * w[2] + B * w[0] * w[1], 1.0 + B * (w[1] * w[1] - t2), -A * w[0] + B * w[1] * w[2],
                       -A * w[1] + B * w[0] * w[2], A * w[0] + B * w[1] * w[2], 1.0 + B * (w[2] * w[2] - t2)};
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c) out[3 * r + c] = Rs[3 * r] * E[c] + Rs[3 * r + 1] * E[3 + c] + Rs[3 * r + 2] * E[6 + c];
}

// One timestep `i` of N between the poses Xs, Xe (4x4 row-major): position, velocity, acceleration, orientation.
// Positions / orientations: cubic for method 3, QUINTIC for anything else; velocities / accelerations: cubic (3),
// quintic (5), zero otherwise — the reference's two code paths differ and both are reproduced.
// the rotation part that does not depend on the timestep: w = log(Rs^T Re)
MP_HD void mp_cartesian_prepare(const double (&Xs)[16], const double (&Xe)[16], double (&w)[3]) {
  const double Rs[9] = {Xs[0], Xs[1], Xs[2], Xs[4], Xs[5], Xs[6], Xs[8], Xs[9], Xs[10]};
  const double Re[9] = {Xe[0], Xe[1], Xe[2], Xe[4], Xe[5], Xe[6], Xe[8], Xe[9], Xe[10]};
  double D[9];  // Rs^T Re
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c) D[3 * r + c] = Rs[r] * Re[c] + Rs[3 + r] * Re[3 + c] + Rs[6 + r] * Re[6 + c];
  mp_log3(D, w);
}

// timestep `i` given the prepared rotation vector: Rs (3x3 row-major), ps / pe = start / end positions
MP_HD void mp_cartesian_eval(const double (&Rs)[9], const double (&ps)[3], const double (&pe)[3], const double (&w)[3], long i,
                             long N, double Tf, int method, float (&pos)[3], float (&vel)[3], float (&acc)[3], float (&ori)[9]) {
  const double timegap = Tf / ((double)N - 1.0);
  const double x = (timegap * (double)i) / Tf;
  const double s = (method == 3) ? 3.0 * x * x - 2.0 * x * x * x : 10.0 * x * x * x - 15.0 * x * x * x * x + 6.0 * x * x * x * x * x;
  const double ws[3] = {w[0] * s, w[1] * s, w[2] * s};
  double O[9];
  mp_rot_times_exp3(Rs, ws, O);
#pragma unroll
  for (int k = 0; k < 9; ++k) ori[k] = (float)O[k];
  const double tau = ((double)i * (Tf / (double)(N - 1))) / Tf;
  double s2, sd, sdd;
  mp_time_scaling(method, tau, Tf, s2, sd, sdd);
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    pos[k] = (float)(s * pe[k] + (1.0 - s) * ps[k]);
    vel[k] = (float)(sd * (pe[k] - ps[k]));
    acc[k] = (float)(sdd * (pe[k] - ps[k]));
  }
}

MP_HD void mp_cartesian_point(const double (&Xs)[16], const double (&Xe)[16], long i, long N, double Tf, int method,
                              float (&pos)[3], float (&vel)[3], float (&acc)[3], float (&ori)[9]) {
  double w[3];
  mp_cartesian_prepare(Xs, Xe, w);
  const double Rs[9] = {Xs[0], Xs[1], Xs[2], Xs[4], Xs[5], Xs[6], Xs[8], Xs[9], Xs[10]};
  const double ps[3] = {Xs[3], Xs[7], Xs[11]}, pe[3] = {Xe[3], Xe[7], Xe[11]};
  mp_cartesian_eval(Rs, ps, pe, w, i, N, Tf, method, pos, vel, acc, ori);
}
