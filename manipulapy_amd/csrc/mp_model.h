// Compiled robot model: the wave-uniform constant block every kernel receives BY VALUE in its
// kernarg segment (scalar loads -> SGPRs; no VGPR, LDS or HBM traffic per thread).
//
// The reference describes a robot by space screws S_i, per-link CoM home poses Mcom_i and CoM-frame
// spatial inertias G_i (reference ManipulaPy/urdf/core.py:670-769).  Evaluating its dynamics in those
// frames costs a general 3x3 rotation per transform.  The host-side model compiler
// (mp_model_compile.cpp) re-expresses the same robot in modified-DH link frames chosen so that
//      T_{i-1,i}(q_i) = Rx(alpha_i) Tx(a_i) Rz(off_i + q_i) Tz(d_i)          (revolute)
//                     = Rx(alpha_i) Tx(a_i) Rz(off_i)       Tz(d_i + q_i)    (prismatic)
// i.e. every motion/force transform is two axis-aligned rotations (4 mul + 2 add each) and two
// axis-aligned shifts (2 FMA each), and the joint motion subspace is a unit vector.  Link inertias
// are moved to the link-frame origin (mass, first moment h = m c, inertia about the origin).
// Dynamics are frame-invariant, so the joint torques are the reference's up to rounding.
#pragma once

#ifndef MP_MAX_DOF
#define MP_MAX_DOF 8    // fully unrolled kernels (model in kernel arguments / constexpr literal): 1..8 joints
#endif
#ifndef MP_BIG_DOF
#define MP_BIG_DOF 32   // looped run-time-n kernels (csrc/mp_dyn.h, model in device memory): 9..32 joints
#endif
#define MP_MID_DOF 16   // ... whose per-row arrays are instantiated for 16 and for MP_BIG_DOF joints: 9..16 keep the small ones

template <typename T>
struct MpJoint {
  T ca, sa;    // cos / sin of alpha_i  (rotation about parent x)
  T a;         // shift along parent x
  T d;         // shift along own z at q = 0
  T off;       // joint-angle offset (rotation about own z at q = 0): host-side only (self-check FK); the kernels use co / so
  T rev;       // 1 = revolute, 0 = prismatic
  T m;         // link mass
  T hx, hy, hz;                       // first moment m * c (c = CoM in link frame)
  T Ixx, Ixy, Ixz, Iyy, Iyz, Izz;     // rotational inertia about the link-frame ORIGIN
  // cos / sin of `off`, exact 0 / +-1 at right angles.  The per-row code takes sin / cos of the joint variable ALONE and turns
  // the result by this constant rotation instead of adding `off` to the angle first: a float32 sum off + q is rounded at the
  // spacing of the SUM (up to 2.4e-7 rad for |off + q| in [4, 8)), an error the link lengths and gravity torques multiply by
  // hundreds of N.m - it was the whole of the float32 kernels' distance from the 1e-4 |ref| + 5e-6 max|row| bound on rows whose
  // torque is a small difference of large terms (profiles/r04_f32_precision_study.txt); q itself is exact.
  T co, so;
};
constexpr int MP_JOINT_FIELDS = 18;

template <typename T, int CAP>
struct MpModelT {
  int n;
  float lscale;  // the robot's length scale, max_i max(|a_i| + |d_i|, |com_i|): weighs joint forces against moments in the float32 kernels'
                 // conditioning test (mp_core.h, mp_id_row_is_hard); float whatever T is
  int pad_[2];
  T base_R[9];   // pose of link frame 1 (at q1 = 0, before its own Rz/Tz) in the space frame
  T base_p[3];
  T tool_R[9];   // end-effector home pose M_ee in link frame n
  T tool_p[3];
  MpJoint<T> j[CAP];
  T qmin[CAP], qmax[CAP];       // joint limits (float32-rounded, as the planner holds them)
  T taumin[CAP], taumax[CAP];   // torque limits (+-inf by default)
};
template <typename T> using MpModel = MpModelT<T, MP_MAX_DOF>;     // what the unrolled kernels take by value
template <typename T> using MpBigModel = MpModelT<T, MP_BIG_DOF>;  // what the looped kernels read through a pointer
// The same model SAID to hold revolute joints only - every arm of the reference's database but the ones with a gripper slide.  The
// generic (not robot-specialised) float32 inverse-dynamics kernel is instantiated a second time on this type and folds `rev` to 1:
// no revolute / prismatic blend of q, qd, qdd and tau, 13 VALU instructions per joint of ~190 (c2 generic 0.107 -> see DESIGN.md).
// Same layout: a pointer to an MpModel<T> whose joints are all revolute may be read as one (the launcher checks).
template <typename T> struct MpModelRev : MpModelT<T, MP_MAX_DOF> {};
template <typename MT> struct MpAllRevolute { static constexpr bool value = false; };
template <typename T> struct MpAllRevolute<MpModelRev<T>> { static constexpr bool value = true; };
template <typename T> struct MpAllRevolute<const MpModelRev<T>> { static constexpr bool value = true; };

// Per-call constants derived on the host in fp64 (gravity / tip wrench seen from link frame 1's parent).
template <typename T>
struct MpCall {
  T a0[3];      // base_R^T * (-g): linear acceleration of the (fictitiously accelerated) base
  T F1n[3];     // Ftip moment, expressed in the pre-joint-1 frame
  T F1f[3];     // Ftip force,  expressed in the pre-joint-1 frame
  // float32 calls: the float64 model (MpModel<double>) of the same robot, in memory the callee can read (device memory for the
  // kernels, the handle's own copy for the CPU launchers), or null.  The float64 re-evaluation of ill-conditioned float32 rows
  // (mp_core.h, mp_rnea_cold) reads its constants there; null makes it widen the float32 model instead.  The robot-specialised
  // programs carry the float64 model as a literal and ignore this.
  const void* cold_model;
  // float32 device calls: where a kernel leaves the indices of its ill-conditioned rows for the float64 pass that follows it on
  // the stream (mp_capi.cpp, launch_hard_rows) instead of re-evaluating them itself: `hard_rows` takes up to `hard_cap` row
  // indices (+ hard_row_base), *hard_ctrl counts them; *hard_next is the list's other counter (the pass zeroes it for the list's
  // next user).  Null: the kernel re-evaluates in place (mp_cold_rows).
  unsigned* hard_rows;
  unsigned* hard_ctrl;
  unsigned* hard_next;
  unsigned hard_cap;
  unsigned hard_row_base;
};

// The float64 pass of ONE earlier launch carried by the next float32 launch: its first `blocks` workgroups work off that launch's
// list (csrc/mp_bodies.h, mp_body_id_lead) beside the float32 rows of the others.  blocks = 0: nothing carried.
struct MpLead {
  MpCall<float> C;       // the earlier launch's constants and its list (hard_rows / hard_ctrl / hard_next / hard_cap)
  const float* q;
  const float* qd;
  const float* qdd;
  float* tau;
  unsigned rows;
  unsigned blocks;
  unsigned nt;           // a fused launch's pass: q = start points, qd = end points, qdd = the time table, nt timesteps per trajectory
};
// Up to four launches' worth of that pass in ONE kernel (blockIdx.y picks the launch): the pass costs ~5 us of launch and memory
// latency however few rows it holds, so the host lets passes wait (mp_capi.cpp, hard_flush) and runs them together.
constexpr int MP_HARD_BATCH = 4;
struct MpHardBatch {
  MpCall<float> C[MP_HARD_BATCH];
  const float* q[MP_HARD_BATCH];
  const float* qd[MP_HARD_BATCH];
  const float* qdd[MP_HARD_BATCH];
  float* tau[MP_HARD_BATCH];
  unsigned rows[MP_HARD_BATCH];
  // passes over GENERATED rows (the fused trajectory + inverse-dynamics kernels): q = the start points, qd = the end points (B, n),
  // qdd = the per-timestep time-scaling table (doubles), nt = timesteps per trajectory; 0 for given rows
  unsigned nt[MP_HARD_BATCH];
};
