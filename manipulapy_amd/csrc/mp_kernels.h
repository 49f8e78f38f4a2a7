// Launcher prototypes (defined in mp_kernels.hip); every launch is asynchronous on `s`.
#pragma once
#include <hip/hip_runtime_api.h>

#include "mp_model.h"

hipError_t mpk_selftest(hipStream_t s, int* d_out /* 64 ints */);
hipError_t mpk_stream(hipStream_t s, int reads, bool nontemporal, const void* a, const void* b, const void* c, void* d, long n4);
hipError_t mpk_stream_mix(hipStream_t s, int reads, int writes, bool nontemporal, const void* a, void* d, long n4);
hipError_t mpk_clock_sampler(hipStream_t s, unsigned long long* out, unsigned blocks, unsigned samples, unsigned naps);

template <typename T>   // (float64 only: float32 rows take mpk_id_dm)
hipError_t mpk_id(hipStream_t s, const MpModel<T>& M, const MpCall<T>& C, bool ftip, const T* q, const T* qd,
                  const T* qdd, T* tau, long rows);

// float32, one row per lane, the model read through a pointer to a device-resident copy (scalar loads joint by joint)
// all_revolute: every joint of the model is revolute (rev == 1) - the kernel instance that folds the revolute / prismatic blend
hipError_t mpk_id_dm(hipStream_t s, const MpModel<float>* d_model, int n, const MpCall<float>& C, bool ftip, const float* q,
                     const float* qd, const float* qdd, float* tau, long rows, const MpLead& L, bool all_revolute);

// the float64 pass over the rows the kernel above handed over (C.hard_rows / hard_ctrl), `blocks` blocks of 64 lanes
hipError_t mpk_id_hard(hipStream_t s, const MpModel<float>* d_model, int n, const MpCall<float>& C, bool ftip, const float* q,
                       const float* qd, const float* qdd, float* tau, unsigned rows, unsigned blocks);
hipError_t mpk_id_hard_batch(hipStream_t s, const MpModel<float>* d_model, int n, bool ftip, const MpHardBatch& B, int entries, unsigned blocks);

hipError_t mpk_batch_traj(hipStream_t s, const MpModel<float>& M, const float* start, const float* end, long B,
                          long Nt, double Tf, int method, float* pos, float* vel, float* acc);

template <typename T>
hipError_t mpk_fk_jac_id(hipStream_t s, const MpModel<T>& M, const MpCall<T>& C, bool ftip, const T* q, const T* qd,
                         const T* qdd, T* Tout, T* Jout, T* tau, long rows);

template <typename T>
hipError_t mpk_mass_matrix(hipStream_t s, const MpModel<T>& M, const T* q, T* Mout, long rows);
template <typename T>
hipError_t mpk_forward_dynamics(hipStream_t s, const MpModel<T>& M, const MpCall<T>& C, bool ftip, const T* q, const T* qd,
                                const T* tau, T* qdd, long rows);
// Ftipmat == nullptr: no tip wrench.  h = dt / intRes.  Outputs are float32 (B, Nt, n).
template <typename T>
hipError_t mpk_fd_traj(hipStream_t s, const MpModel<T>& M, const MpCall<T>& C, const T* theta0, const T* dtheta0,
                       const T* taumat, const T* Ftipmat, long B, long Nt, T h, int intRes, float* pos, float* vel, float* acc);

// the same roll-out on the TIME-MAJOR device layout: taumat (Nt, B, n), Ftipmat (Nt, B, 6), pos / vel / acc (Nt, B, n)
template <typename T>
hipError_t mpk_fd_traj_tm(hipStream_t s, const MpModel<T>& M, const MpCall<T>& C, const T* theta0, const T* dtheta0,
                          const T* taumat, const T* Ftipmat, long B, long Nt, T h, int intRes, float* pos, float* vel, float* acc);
// (outer, inner, row_dwords x 4 bytes) -> (inner, outer, row_dwords x 4 bytes)
hipError_t mpk_transpose_rows(hipStream_t s, const void* src, void* dst, long outer, long inner, int row_dwords);

// ---- 9..32 joints (csrc/mp_dyn.h): run-time-n kernels, the model (MpBigModel<T>) resident in device memory; n = its joint count
// (picks the kernels' per-row array capacity, MP_MID_DOF or MP_BIG_DOF)
template <typename T>
hipError_t mpk_dyn_fk_jac_id(hipStream_t s, int n, const MpBigModel<T>* d_model, const MpCall<T>& C, bool ftip, const T* q, const T* qd,
                             const T* qdd, T* Tout, T* Jout, T* tau, long rows);
template <typename T>
hipError_t mpk_dyn_mass_matrix(hipStream_t s, int n, const MpBigModel<T>* d_model, const T* q, T* Mout, long rows);
template <typename T>
hipError_t mpk_dyn_forward_dynamics(hipStream_t s, int n, const MpBigModel<T>* d_model, const MpCall<T>& C, bool ftip, const T* q, const T* qd,
                                    const T* tau, T* qdd, long rows);
template <typename T>
hipError_t mpk_dyn_fd_traj(hipStream_t s, int n, const MpBigModel<T>* d_model, const MpCall<T>& C, const T* theta0, const T* dtheta0,
                           const T* taumat, const T* Ftipmat, long B, long Nt, T h, int intRes, float* pos, float* vel, float* acc,
                           bool time_major);
// pos / vel / acc (all three or none) and / or tau of the time-scaled trajectories
hipError_t mpk_dyn_traj(hipStream_t s, int n, const MpBigModel<float>* d_model, const MpCall<float>& C, bool ftip, const float* start,
                        const float* end, long B, long Nt, double Tf, int method, float* pos, float* vel, float* acc, float* tau);

// Cartesian straight-line trajectories between B pose pairs (4x4 row-major float64): float32 (B,Nt,3) x3, (B,Nt,3,3)
hipError_t mpk_cartesian_traj(hipStream_t s, const double* Xstart, const double* Xend, long B, long Nt, double Tf, int method,
                              float* pos, float* vel, float* acc, float* ori);

// Fused attractive + repulsive potential and gradient at P points against O obstacles (float32); goal on the host.
hipError_t mpk_potential_field(hipStream_t s, const float* pos, const float* goal3_host, const float* obs, long P, long O,
                               float influence, float* pot, float* grad);

// B damped-least-squares inverse-kinematics problems (csrc/mp_ik.h): Tdes (B,4,4), theta0 / theta (B,n) float64
template <int CAP> struct MpIkParamsT;
typedef MpIkParamsT<MP_MAX_DOF> MpIkParams;
typedef MpIkParamsT<MP_BIG_DOF> MpIkBigParams;
// queue_counter: 8 bytes of device memory owned by the caller (zeroed here on the stream before the launch)
hipError_t mpk_ik(hipStream_t s, const MpModel<double>& M, const MpIkParams& P, const double* Tdes, const double* theta0, long B,
                  double* theta, int* success, int* iterations, int* restarts, unsigned long long* queue_counter, int compute_units);
// K closed-loop PD regulation runs (csrc/mp_core.h mp_pd_regulation_run): theta0 / des (K,n), Kp / Kd (K), err (K,steps), count (K)
hipError_t mpk_pd_regulation(hipStream_t s, const MpModel<double>& M, const MpCall<double>& C, const double* theta0, const double* des,
                             const double* Kp, const double* Kd, long K, double dt, int steps, double* err, int* count);
hipError_t mpk_dyn_pd_regulation(hipStream_t s, int n, const MpBigModel<double>* d_model, const MpCall<double>& C, const double* theta0,
                                 const double* des, const double* Kp, const double* Kd, long K, double dt, int steps, double* err, int* count);
// the same for 9..32 joints (run-time joint count, the model resident in device memory)
hipError_t mpk_dyn_ik(hipStream_t s, int n, const MpBigModel<double>* d_model, const MpIkBigParams& P, const double* Tdes, const double* theta0,
                      long B, double* theta, int* success, int* iterations, int* restarts, unsigned long long* queue_counter,
                      int compute_units);

// table-driven fused generation + ID (float32): `tab` = 3 doubles per timestep written by mpk_time_table for the same
// (Nt, Tf, method); one lane takes timesteps t and t + ceil(Nt / 2) of one trajectory
hipError_t mpk_time_table(hipStream_t s, double* tab, long Nt, double Tf, int method);
unsigned mpk_traj_blocks_per_trajectory(long Nt);
// the float64 pass over the rows the fused generic kernel handed over (inputs regenerated)
hipError_t mpk_traj_id_hard(hipStream_t s, const MpModel<float>* d_model, int n, const MpCall<float>& C, bool ftip, const float* start,
                            const float* end, unsigned Nt, const double* tab, float* tau, unsigned rows, unsigned blocks);
hipError_t mpk_traj_id_tab(hipStream_t s, const MpModel<float>& M, const MpCall<float>& C, bool ftip, const float* start,
                           const float* end, long B, long Nt, const double* tab, float* tau);

