/* manipula_hip.h — C ABI of libmanipula_hip.so (MI355X / gfx950).
 *
 * The drop-in boundary for the batched trajectory + rigid-body-dynamics hot path of
 * boelnasr/ManipulaPy v1.4.1.  The reference has no FFI: its "GPU side" is a set of Python
 * launchers behind a kernel registry (ManipulaPy/cuda_kernels/registry.py:46-89, :828-867) and
 * mixin methods that launch Numba kernels directly (planning/trajectory_dynamics.py:248-260,
 * :524-539; planning/trajectory.py:643).  Each entry point below is what such a launcher binds
 * (through ctypes); the reference interface it replaces is cited per function.
 *
 * Conventions
 *   - every function returns int: 0 = MP_OK, otherwise an MP_ERR_* code; mp_last_error() returns
 *     the thread-local message of the last failure on the calling thread.  There is NO CPU fallback
 *     anywhere behind this ABI: without a usable GPU the compute calls that take an mp_ctx fail with MP_ERR_HIP
 *     (the *_cpu twins at the end are separate entry points a caller selects explicitly).
 *   - arrays are C-contiguous (row-major), exactly the shapes the reference's Python API uses;
 *     "d_" parameters are device pointers obtained from mp_malloc (16-byte aligned), "h_" or
 *     unprefixed pointers are host memory owned by the caller.
 *   - kernels are enqueued on the context's compute stream and return immediately; the *_host
 *     variants copy in, launch, copy out and synchronise before returning.
 *   - a context is bound to one device; one context per process per GPU (one process per GPU for
 *     multi-GPU jobs, see mp_comm_*).  Calls on one context must not race from several threads.
 */
#ifndef MANIPULA_HIP_H
#define MANIPULA_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MP_OK 0
#define MP_ERR_INVALID 1      /* bad argument (null pointer, shape, alignment, dof mismatch) */
#define MP_ERR_HIP 2          /* HIP runtime failure (no device, launch error, out of memory) */
#define MP_ERR_MODEL 3        /* robot tables rejected by the model compiler */
#define MP_ERR_UNSUPPORTED 4  /* valid request this build does not implement */
#define MP_ERR_COMM 5         /* RCCL failure */

#define MP_MAX_DOF 8      /* joints of the fully unrolled / run-time-specialisable kernels */
#define MP_BIG_DOF 32     /* joints mp_model_create accepts: 9..32 run looped run-time-n kernels (csrc/mp_dyn.h; every operation
                             incl. inverse kinematics; run-time specialisation is MP_ERR_UNSUPPORTED) */
#define MP_UNIQUE_ID_BYTES 128

typedef struct mp_ctx mp_ctx;     /* device context: device id, streams, device-buffer pool */
typedef struct mp_model mp_model; /* compiled robot model (host object; reaches the kernels by value or as a device copy) */
typedef struct mp_event mp_event; /* HIP event on the context's compute stream */
typedef struct mp_graph mp_graph; /* instantiated HIP graph captured from the context's compute stream */
typedef struct mp_comm mp_comm;   /* RCCL communicator (one rank per process) */

/* ---- library / device ----------------------------------------------------------------------- */
int mp_version(void);                 /* ABI version, currently 1 */
const char* mp_last_error(void);      /* message of the last error on this thread ("" if none) */
/* Number of visible HIP devices; 0 with MP_OK when the runtime loads but finds none.
 * Replaces the reference's CUDA probe, cuda_kernels/_runtime.py:32-73 / registry.py:92-137. */
int mp_device_count(int* count);
int mp_ctx_create(int device_id, mp_ctx** out);
int mp_ctx_destroy(mp_ctx* ctx);
int mp_ctx_synchronize(mp_ctx* ctx);  /* waits for every stream of the context */
/* The context's compute stream as a hipStream_t (returned through a void*: no HIP header is needed to include this file), for a
 * caller that orders its own HIP work - a copy, a kernel of its own - behind the library's launches.  Stream order is the whole
 * contract: every device-pointer entry point returns with everything its result needs enqueued on this stream.  (Before this call
 * the float64 pass of a float32 inverse-dynamics launch on pool arrays may stay parked - note at mp_id_trajectory_f32; the call runs
 * what is parked and switches parking OFF for the rest of the context's life: mp_malloc returns plain device pointers, so a caller
 * that holds the stream can read pool memory with its own copies and kernels.)  The reference has no streams of its
 * own to expose (its launchers return finished host arrays, cuda_kernels/trajectory_kernels.py:1043-1081). */
int mp_ctx_get_stream(mp_ctx* ctx, void** hip_stream);
/* name, CU count, total HBM bytes — replaces get_gpu_properties(), cuda_kernels/registry.py:335-356 */
int mp_ctx_properties(mp_ctx* ctx, char* name, size_t name_len, int* compute_units, uint64_t* hbm_bytes);
/* Launch a 1-block probe kernel that writes its lane ids and check the result on the host.
 * Replaces the reference's device self-test kernel, cuda_kernels/_runtime.py:127-138. */
int mp_selftest(mp_ctx* ctx);
/* Device-copy microbenchmark for the roofline ("confirm the peak with a device-copy microbenchmark in the same
 * run"): `reps` launches of dst = a (reads = 1) or dst = a + b + c (reads = 3, the 3 : 1 byte mix of the inverse-dynamics
 * kernels) over arrays of bytes_per_array bytes, 16 bytes per lane; *gb_per_s = (reads + 1) * bytes * reps / elapsed.
 * reads = 11 / 13: the same two kernels with non-temporal loads and stores (what the whole-line row movers use).
 * Nothing in the reference corresponds to it. */
int mp_stream_bandwidth(mp_ctx* ctx, size_t bytes_per_array, int reads, int reps, double* gb_per_s);
/* The same probe for any of the byte mixes the kernels have (reads : writes = 0:1, 1:1, 2:1, 3:1, 1:2, 1:3, 2:3): every lane
 * reads one 16-byte chunk from each of `reads` arrays and writes one to each of `writes` arrays of bytes_per_array bytes, plain
 * or non-temporal; *gb_per_s = (reads + writes) * bytes * reps / elapsed.  bench.py runs it with a configuration's own mix and
 * size right before its timed region: roofline.frac_of_probe says how much of what THIS box streams the kernel reaches. */
int mp_stream_bandwidth_mix(mp_ctx* ctx, size_t bytes_per_array, int reads, int writes, int nontemporal, int reps, double* gb_per_s);

/* The shader clock the GPU holds WHILE the caller's launches run (measurement plumbing, nothing in the reference corresponds to it):
 * _begin starts a bounded sampler on a stream of its own (8 one-wave blocks stamping s_memtime / s_memrealtime over about
 * duration_ms, 0 < duration_ms <= 2000), the caller launches what it wants measured, _end waits for the sampler and returns
 * *clock_hz = median over the blocks of delta s_memtime / delta s_memrealtime x 100 MHz and, optionally, the time the stamps span.
 * bench.py reports it as `clock_hz` beside every configuration's kernel time (the package's power controller runs the same code
 * object at 1.7 - 2.5 GHz depending on the kernel's mix and the box). */
int mp_clock_sample_begin(mp_ctx* ctx, double duration_ms);
int mp_clock_sample_end(mp_ctx* ctx, double* clock_hz, double* sampled_ms);

/* Profiling (replaces the reference's profile_start / profile_stop hooks, planning/trajectory_planning.py:295-296, and
 * the timing part of its performance_stats): while on, every device-pointer entry point (and so every *_host one)
 * brackets what it enqueues with a timed HIP event pair on the compute stream and an roctx range named after the
 * entry point (visible to rocprofv3 --marker-trace).  mp_ctx_profile waits for the pairs recorded since the last
 * read and returns the accumulated kernel milliseconds, the number of timed calls and the last call's milliseconds
 * (any output may be NULL); reset != 0 zeroes the accumulators afterwards.  Off by default; not active while a launch
 * graph is being captured. */
int mp_ctx_set_profiling(mp_ctx* ctx, int on);
int mp_ctx_profile(mp_ctx* ctx, double* kernel_ms_total, int64_t* timed_calls, double* kernel_ms_last, int reset);

/* ---- device memory (pooled per context; replaces _GlobalCudaMemoryPool, cuda_kernels/memory.py:55-118,
 *      and the pinned-H2D helper _h2d_pinned, cuda_kernels/memory.py:12-50) ----------------------- */
int mp_malloc(mp_ctx* ctx, size_t bytes, void** d_ptr);
int mp_free(mp_ctx* ctx, void* d_ptr);            /* returns the buffer to the pool */
int mp_pool_trim(mp_ctx* ctx);                    /* releases pooled buffers back to the driver */
int mp_memcpy_h2d(mp_ctx* ctx, void* d_dst, const void* h_src, size_t bytes); /* synchronous */
int mp_memcpy_d2h(mp_ctx* ctx, void* h_dst, const void* d_src, size_t bytes); /* synchronous */
int mp_memset(mp_ctx* ctx, void* d_dst, int value, size_t bytes);
/* page-locked host buffers (the reference's pinned staging, cuda_kernels/memory.py:12-50, handed to the caller): the
 * *_host_* entry points below accept any host pointer; on buffers from mp_host_alloc the upload, the kernels and
 * the download of a large call overlap chunk by chunk (pageable memory is staged by the runtime and serialises). */
int mp_host_alloc(mp_ctx* ctx, size_t bytes, void** h_ptr);
int mp_host_free(mp_ctx* ctx, void* h_ptr); /* ctx may be NULL: the buffer outlives the context that allocated it */

/* ---- timing on the compute stream ------------------------------------------------------------ */
int mp_event_create(mp_ctx* ctx, mp_event** out);
int mp_event_destroy(mp_event* ev);
int mp_event_record(mp_ctx* ctx, mp_event* ev);   /* on the stream the kernels are launched on */
int mp_event_elapsed_ms(mp_event* start, mp_event* stop, float* ms); /* synchronises on `stop` */

/* ---- launch graphs ----------------------------------------------------------------------------
 * Every device-pointer entry point below only enqueues kernels on the context's compute stream, so a sequence of
 * them can be captured once and replayed with a single submission (hipGraph): between mp_graph_begin and
 * mp_graph_end the calls are recorded instead of executed.  Replaying re-runs the same kernels on the same
 * device pointers with the same per-call constants; the caller refreshes the buffers' contents in between.
 * (The reference has no counterpart: its launchers pay one Numba dispatch per kernel, cuda_kernels/registry.py:828-867.)
 * Not capturable: the *_host_* entry points, mp_malloc / mp_free, mp_model_specialize, mp_comm_*. */
int mp_graph_begin(mp_ctx* ctx);
int mp_graph_end(mp_ctx* ctx, mp_graph** out);
int mp_graph_launch(mp_ctx* ctx, mp_graph* graph);   /* asynchronous, on the compute stream */
int mp_graph_destroy(mp_graph* graph);

/* ---- robot model ------------------------------------------------------------------------------
 * Inputs are the reference's constant tables (urdf/core.py:670-769; ManipulatorDynamics ctor,
 * dynamics/manipulator_dynamics.py:43-86), float64 row-major:
 *   S (6,n) space screws [w;v]; Mcom n x (4,4) = Mlist_per_link; G n x (6,6) = Glist;
 *   M_ee (4,4) = M_list; joint_limits (n,2) or NULL (= unbounded); torque_limits (n,2) or NULL
 *   (= +-inf, planning/trajectory_planning.py:219-223).  Limits are rounded to float32 as the
 *   planner stores them (:218).  No device is needed: the model is host data.  1 <= n <= MP_BIG_DOF (the reference's
 *   algorithms loop over any n, dynamics/mass_matrix.py:62-96; its database goes up to 10 actuated joints). */
int mp_model_create(int n, const double* S, const double* Mcom, const double* G, const double* M_ee,
                    const double* joint_limits, const double* torque_limits, mp_model** out);
/* Also releases what every live context holds for the model (specialised code object, device-resident copies); while a
 * launch graph captured on a context is alive, or a capture is open, they are retired instead and released with the last
 * graph / the context (a graph keeps kernel nodes and device addresses of the models it captured, no reference). */
int mp_model_destroy(mp_model* model);
int mp_model_dof(const mp_model* model, int* n);
/* Compiled per-joint parameters, 16 doubles per joint (see csrc/mp_model.h): for inspection/tests. */
int mp_model_params(const mp_model* model, double* out /* n*16 */);
/* The whole compiled model as the kernels receive it (struct MpModel<float|double>, csrc/mp_model.h):
 * *bytes = its size; out may be NULL to query the size.  Used by the kernel specialiser and by tests. */
int mp_model_blob(const mp_model* model, int use_f64, void* out, size_t* bytes);
/* End-effector pose through the compiled chain on the HOST in float64 (model self-check helper). */
int mp_model_fk_host(const mp_model* model, const double* q /* n */, double* T /* 16 */);

/* ---- run-time specialisation (new; csrc/mp_jit.cpp) -------------------------------------------------
 * Compile the float32 kernels with THIS robot's constants baked in (hiprtc, gfx950; on-disk cache in
 * MANIPULAPY_HIP_CACHE or <library dir>/jit_cache) and load them on the context's device.  Afterwards
 * mp_id_trajectory_f32 / mp_traj_id_fused_f32 / mp_fd_trajectory_f32 (+ their *_host forms) use them for
 * this (context, model) pair; every other entry point keeps using the generic kernels.  Idempotent.
 * MANIPULAPY_HIP_SPECIALIZE=0 makes the launchers ignore specialised kernels (A/B measurements). */
int mp_model_specialize(mp_ctx* ctx, const mp_model* model);
int mp_model_is_specialized(mp_ctx* ctx, const mp_model* model, int* yes);
/* Generate + compile only (needs hiprtc, no GPU): size of the two code objects together and whether both came from the cache. */
int mp_model_specialize_compile(const mp_model* model, size_t* code_bytes, int* from_cache);
/* The two generated translation units (NUL-terminated; the second - the one-row-per-lane float32 inverse dynamics, compiled with
 * another scheduling strategy - behind a "// ==== second program" line).  *len = required size incl. NUL; buf may be NULL. */
int mp_model_specialize_source(const mp_model* model, char* buf, size_t* len);

/* ---- hot path, device pointers ----------------------------------------------------------------
 * g: (3,) gravity vector or NULL (= [0,0,-9.81], planning/trajectory_dynamics.py:54);
 * Ftip: (6,) SPACE-frame wrench [m;f] or NULL (= 0) (dynamics/id_fd.py:41-47). */

/* pos/vel/acc (B,N,n) float32 for B start/end pairs (B,n): time scaling of
 * planning/trajectory.py:15-75 + clip of positions to the joint limits (:311-313, :479-481).
 * Replaces batch_trajectory_kernel, cuda_kernels/trajectory_kernels.py:763-831, and its launcher
 * optimized_batch_trajectory_generation (:1204-1290); B = 1 is joint_trajectory (registry launchers
 * "trajectory.*", cuda_kernels/registry.py:828-867).  method: 3 cubic, 5 quintic, else zeros. */
int mp_batch_trajectory_f32(mp_ctx* ctx, const mp_model* model, const float* d_start, const float* d_end,
                            int64_t B, int64_t N, double Tf, int method, float* d_pos, float* d_vel, float* d_acc);

/* tau (rows,n) = clip(inverse_dynamics(q, qd, qdd, g, Ftip), torque_limits) for `rows` independent
 * (trajectory, timestep) rows.  Replaces _inverse_dynamics_gpu / inverse_dynamics_kernel
 * (planning/trajectory_dynamics.py:92-306, cuda_kernels/trajectory_kernels.py:521-602) with the
 * arithmetic of _inverse_dynamics_cpu (:308-380) -> dynamics/id_fd.py:16-48.
 *
 * float32 rows, adaptive precision.  The float32 kernels hand the few rows per thousand whose torques are a small difference of
 * large terms (csrc/mp_core.h, MpRowScale / mp_id_row_is_hard) to a float64 pass behind them, so that EVERY row holds the
 * float32 parity bound.  When the call returns:
 *   - arrays the CALLER allocated (hipMalloc, a framework tensor - anything not from mp_malloc): kernel and pass are both
 *     enqueued on the compute stream.  Whatever the caller enqueues there next - or a wait on that stream - sees the complete
 *     torques; the arrays may be freed or reused as soon as the stream has passed the call, like after any asynchronous launch.
 *   - arrays from this context's pool (mp_malloc), all four of them: the pass may stay PARKED - to ride with the next launch of the
 *     same robot-specialised kernel on other arrays (that launch's first workgroups work it off), or to run together with the
 *     passes of up to three more launches.  It runs before anything can see the difference: every other entry point of the context
 *     (mp_memcpy_*, mp_ctx_synchronize, mp_event_record, mp_ctx_get_stream, the *_host calls, the communicator ...), a launch
 *     whose arrays overlap the parked one's, a fifth launch, mp_ctx_destroy.  Without the compute stream (mp_ctx_get_stream) pool
 *     memory is only reachable through those; once the stream has been handed out, nothing is parked any more (as for caller-owned arrays).
 * The same holds for mp_traj_id_fused_f32 (start / end / tau). */
int mp_id_trajectory_f32(mp_ctx* ctx, const mp_model* model, const float* d_q, const float* d_qd,
                         const float* d_qdd, int64_t rows, const double* g, const double* Ftip, float* d_tau);
int mp_id_trajectory_f64(mp_ctx* ctx, const mp_model* model, const double* d_q, const double* d_qd,
                         const double* d_qdd, int64_t rows, const double* g, const double* Ftip, double* d_tau);

/* joint_trajectory -> inverse_dynamics_trajectory fused: tau (B,N,n) straight from (B,n) start/end
 * pairs; the intermediate positions (clipped) / velocities / accelerations are rounded to float32
 * exactly as the two-call pipeline would store them, but never touch HBM. */
int mp_traj_id_fused_f32(mp_ctx* ctx, const mp_model* model, const float* d_start, const float* d_end,
                         int64_t B, int64_t N, double Tf, int method, const double* g, const double* Ftip,
                         float* d_tau);

/* Per row: T (4,4) = forward_kinematics(q,"space") (kinematics/fk.py:59-70), J (6,n) =
 * jacobian(q,"space") (kinematics/jacobian.py:62-73), tau as above.  Any of d_T / d_J / d_tau may
 * be NULL to skip that output (d_qd/d_qdd may then be NULL too). */
int mp_fk_jac_id_f64(mp_ctx* ctx, const mp_model* model, const double* d_q, const double* d_qd,
                     const double* d_qdd, int64_t rows, const double* g, const double* Ftip, double* d_T,
                     double* d_J, double* d_tau);
int mp_fk_jac_id_f32(mp_ctx* ctx, const mp_model* model, const float* d_q, const float* d_qd,
                     const float* d_qdd, int64_t rows, const double* g, const double* Ftip, float* d_T,
                     float* d_J, float* d_tau);

/* M (rows,n,n) = mass_matrix(q) per row (dynamics/mass_matrix.py:16-99), symmetrised like the reference. */
int mp_mass_matrix_f64(mp_ctx* ctx, const mp_model* model, const double* d_q, int64_t rows, double* d_M);
int mp_mass_matrix_f32(mp_ctx* ctx, const mp_model* model, const float* d_q, int64_t rows, float* d_M);

/* qdd (rows,n) = forward_dynamics(q, qd, tau, g, Ftip) per row = solve(M, tau - c - g - Js^T Ftip)
 * (dynamics/id_fd.py:50-83); one g / Ftip for all rows. */
int mp_forward_dynamics_f64(mp_ctx* ctx, const mp_model* model, const double* d_q, const double* d_qd,
                            const double* d_tau, int64_t rows, const double* g, const double* Ftip, double* d_qdd);
int mp_forward_dynamics_f32(mp_ctx* ctx, const mp_model* model, const float* d_q, const float* d_qd,
                            const float* d_tau, int64_t rows, const double* g, const double* Ftip, float* d_qdd);

/* forward_dynamics_trajectory for B independent trajectories (planning/trajectory_dynamics.py:382-423,
 * :580-708; replaces forward_dynamics_kernel, cuda_kernels/trajectory_kernels.py:604-705): semi-implicit
 * Euler, intRes sub-steps of dt/intRes, positions clipped to the joint limits after every sub-step, row 0 =
 * initial state.  theta0/dtheta0 (B,n), taumat (B,N,n), Ftipmat (B,N,6) or NULL (= no wrench); the state is
 * integrated in the input precision, pos/vel/acc (B,N,n) are float32 as the reference stores them. */
int mp_fd_trajectory_f32(mp_ctx* ctx, const mp_model* model, const float* d_theta0, const float* d_dtheta0,
                         const float* d_taumat, const float* d_Ftipmat, int64_t B, int64_t N, const double* g,
                         double dt, int intRes, float* d_pos, float* d_vel, float* d_acc);
int mp_fd_trajectory_f64(mp_ctx* ctx, const mp_model* model, const double* d_theta0, const double* d_dtheta0,
                         const double* d_taumat, const double* d_Ftipmat, int64_t B, int64_t N, const double* g,
                         double dt, int intRes, float* d_pos, float* d_vel, float* d_acc);

/* The same roll-out on the TIME-MAJOR device layout: d_taumat (N,B,n), d_Ftipmat (N,B,6) or NULL, d_pos / d_vel / d_acc
 * (N,B,n); d_theta0 / d_dtheta0 stay (B,n).  The reference integrates ONE trajectory (planning/trajectory_dynamics.py:
 * 382-423, :580-708); the batch axis is this library's extension, and with time outermost the trajectories of a wavefront
 * are neighbours in memory at every step (whole cache lines per step instead of 4-step LDS tiles): the faster form for
 * callers that keep their histories on the device.  Results are element for element those of mp_fd_trajectory_*. */
int mp_fd_trajectory_tm_f32(mp_ctx* ctx, const mp_model* model, const float* d_theta0, const float* d_dtheta0,
                            const float* d_taumat, const float* d_Ftipmat, int64_t B, int64_t N, const double* g,
                            double dt, int intRes, float* d_pos, float* d_vel, float* d_acc);
int mp_fd_trajectory_tm_f64(mp_ctx* ctx, const mp_model* model, const double* d_theta0, const double* d_dtheta0,
                            const double* d_taumat, const double* d_Ftipmat, int64_t B, int64_t N, const double* g,
                            double dt, int intRes, float* d_pos, float* d_vel, float* d_acc);
/* d_dst (inner, outer, row_bytes) <- d_src (outer, inner, row_bytes): converts between the batch-major API arrays
 * (B,N,n) and the time-major layout (N,B,n), either way.  row_bytes: a multiple of 4, at most 256 (32 float64 joints). */
int mp_transpose_rows(mp_ctx* ctx, const void* d_src, int64_t outer, int64_t inner, int64_t row_bytes, void* d_dst);

/* cartesian_trajectory for B pose pairs (planning/trajectory.py:504-594, :676-737; replaces
 * cartesian_trajectory_kernel, cuda_kernels/trajectory_kernels.py:707-759, and the host-side orientation loop):
 * Xstart / Xend (B,4,4) float64; pos / vel / acc (B,N,3) and orientations (B,N,3,3) float32.  N >= 2. */
int mp_cartesian_trajectory_f32(mp_ctx* ctx, const double* d_Xstart, const double* d_Xend, int64_t B, int64_t N, double Tf,
                                int method, float* d_pos, float* d_vel, float* d_acc, float* d_orient);

/* Fused potential field ("potential_field.fused", cuda_kernels/field_kernels.py:20-104, registry.py:870-897):
 * potential (P,) and gradient (P,3) at positions (P,3) for one goal (3, host) and obstacles (O,3), float32. */
int mp_potential_field_f32(mp_ctx* ctx, const float* d_positions, const float* goal, const float* d_obstacles, int64_t P,
                           int64_t O, float influence_distance, float* d_potential, float* d_gradient);

/* Batched inverse kinematics: B independent pose targets (B,4,4) from initial guesses (B,n), float64.  One lane runs
 * the reference's damped-least-squares iteration (kinematics/ik.py:39-311; adaptive_tuning / backtracking select its two options, default off: no adaptive tuning, no
 * backtracking unless asked for) for one target: geometric error, damped step through a 6x6 Cholesky (== the reference's damped
 * pseudo-inverse), step cap, joint-limit projection (joint_limits: host (n,2), +-inf = open, NULL = all open), best
 * solution tracking, stagnation restart (counter-hashed noise seeded by `seed`; the reference draws from NumPy's
 * global stream).  iterations follows the reference's count (k + 1, or max_iterations + 1 when exhausted). */
int mp_inverse_kinematics_f64(mp_ctx* ctx, const mp_model* model, const double* d_T_desired, const double* d_theta0, int64_t B,
                              const double* joint_limits, double eomg, double ev, int max_iterations, double damping,
                              double step_cap, double weight_orientation, double weight_position, int adaptive_tuning, int backtracking,
                              uint32_t seed,
                              double* d_theta, int32_t* d_success, int32_t* d_iterations, int32_t* d_restarts);

/* ---- hot path, host pointers (what a Python gpu_launcher calls): H2D, launch, D2H, synchronise - */
int mp_batch_trajectory_host_f32(mp_ctx* ctx, const mp_model* model, const float* start, const float* end,
                                 int64_t B, int64_t N, double Tf, int method, float* pos, float* vel, float* acc);
int mp_id_trajectory_host_f32(mp_ctx* ctx, const mp_model* model, const float* q, const float* qd,
                              const float* qdd, int64_t rows, const double* g, const double* Ftip, float* tau);
int mp_id_trajectory_host_f64(mp_ctx* ctx, const mp_model* model, const double* q, const double* qd,
                              const double* qdd, int64_t rows, const double* g, const double* Ftip, double* tau);
int mp_traj_id_fused_host_f32(mp_ctx* ctx, const mp_model* model, const float* start, const float* end,
                              int64_t B, int64_t N, double Tf, int method, const double* g, const double* Ftip,
                              float* tau);
int mp_fk_jac_id_host_f64(mp_ctx* ctx, const mp_model* model, const double* q, const double* qd,
                          const double* qdd, int64_t rows, const double* g, const double* Ftip, double* T,
                          double* J, double* tau);

int mp_mass_matrix_host_f64(mp_ctx* ctx, const mp_model* model, const double* q, int64_t rows, double* M);
int mp_forward_dynamics_host_f64(mp_ctx* ctx, const mp_model* model, const double* q, const double* qd,
                                 const double* tau, int64_t rows, const double* g, const double* Ftip, double* qdd);
int mp_fd_trajectory_host_f32(mp_ctx* ctx, const mp_model* model, const float* theta0, const float* dtheta0,
                              const float* taumat, const float* Ftipmat, int64_t B, int64_t N, const double* g,
                              double dt, int intRes, float* pos, float* vel, float* acc);
int mp_fd_trajectory_host_f64(mp_ctx* ctx, const mp_model* model, const double* theta0, const double* dtheta0,
                              const double* taumat, const double* Ftipmat, int64_t B, int64_t N, const double* g,
                              double dt, int intRes, float* pos, float* vel, float* acc);

int mp_cartesian_trajectory_host_f32(mp_ctx* ctx, const double* Xstart, const double* Xend, int64_t B, int64_t N, double Tf,
                                     int method, float* pos, float* vel, float* acc, float* orient);

int mp_potential_field_host_f32(mp_ctx* ctx, const float* positions, const float* goal, const float* obstacles, int64_t P,
                                int64_t O, float influence_distance, float* potential, float* gradient);

int mp_inverse_kinematics_host_f64(mp_ctx* ctx, const mp_model* model, const double* T_desired, const double* theta0, int64_t B,
                                   const double* joint_limits, double eomg, double ev, int max_iterations, double damping,
                                   double step_cap, double weight_orientation, double weight_position, int adaptive_tuning, int backtracking,
                              uint32_t seed,
                                   double* theta, int32_t* success, int32_t* iterations, int32_t* restarts);

/* K closed-loop regulation runs under joint-space PD torque, one lane each, float64 - the simulations the reference's
 * Ziegler-Nichols gain sweep runs one gain after the other (control/metrics.py:315-355: tau = Kp (des - theta) - Kd omega,
 * alpha = M^-1 (tau - c - g), omega += alpha dt, theta += omega dt, error = |theta - des|, stop after a step > 10 whose
 * error exceeds 1e10).  Host arrays: theta0 / theta_des (K,n), Kp / Kd (K); errors (K,steps) - entries past a run's
 * count are left untouched - and count (K) come back.  g: 3 doubles or NULL (0,0,-9.81). */
int mp_pd_regulation_host_f64(mp_ctx* ctx, const mp_model* model, const double* theta0, const double* theta_des, const double* Kp,
                              const double* Kd, int64_t K, const double* g, double dt, int steps, double* errors, int32_t* count);

/* ---- CPU twins (csrc/mp_cpu.cpp): what the kernel registry's cpu_launcher of each operation calls ----------------
 * The reference routes every registered operation through `gpu_launcher if _cuda_routing_enabled() else cpu_launcher`
 * (cuda_kernels/registry.py:85-89) and its planner mixins pick _*_cpu when _should_use_gpu is false
 * (planning/trajectory_dynamics.py:84-90, :414-423); its CPU launchers are NumPy code (dynamics/id_fd.py:16-83,
 * kinematics/fk.py:39-86, kinematics/jacobian.py:39-93, dynamics/mass_matrix.py:16-99,
 * planning/trajectory_dynamics.py:308-380, :580-708, planning/trajectory.py:676-737).  These entry points evaluate the
 * same per-row templates the HIP kernels instantiate, on host arrays, over `nthreads` host threads (0 = all cores, or
 * MANIPULAPY_CPU_THREADS).  No context and no GPU are involved; they are selected by the routing rule (NumPy backend
 * active / use_cuda=False), never as a fallback of a failing GPU call.  Same argument meaning as the *_host forms. */
int mp_cpu_threads(int64_t items); /* threads a call over `items` rows would use */
int mp_id_trajectory_cpu_f32(const mp_model* model, const float* q, const float* qd, const float* qdd, int64_t rows,
                             const double* g, const double* Ftip, float* tau, int nthreads);
int mp_id_trajectory_cpu_f64(const mp_model* model, const double* q, const double* qd, const double* qdd, int64_t rows,
                             const double* g, const double* Ftip, double* tau, int nthreads);
/* Diagnostic: in_f64[r] = 1 where the float32 inverse-dynamics kernels (and mp_id_trajectory_cpu_f32) evaluate row r in float64 -
 * the rows whose joint wrenches exceed 16 x their largest torque, where a float32 recursion cannot hold the parity bound
 * 1e-4 |ref| + 5e-6 max|row| (csrc/mp_core.h, mp_rnea_row).  No counterpart in the reference, whose path is float64 throughout
 * (planning/trajectory_dynamics.py:308-380). */
int mp_id_row_precision_cpu_f32(const mp_model* model, const float* q, const float* qd, const float* qdd, int64_t rows,
                                const double* g, const double* Ftip, uint8_t* in_f64, int nthreads);
int mp_fk_jac_id_cpu_f64(const mp_model* model, const double* q, const double* qd, const double* qdd, int64_t rows,
                         const double* g, const double* Ftip, double* T, double* J, double* tau, int nthreads);
int mp_mass_matrix_cpu_f64(const mp_model* model, const double* q, int64_t rows, double* M, int nthreads);
int mp_forward_dynamics_cpu_f64(const mp_model* model, const double* q, const double* qd, const double* tau, int64_t rows,
                                const double* g, const double* Ftip, double* qdd, int nthreads);
int mp_fd_trajectory_cpu_f32(const mp_model* model, const float* theta0, const float* dtheta0, const float* taumat,
                             const float* Ftipmat, int64_t B, int64_t N, const double* g, double dt, int intRes, float* pos,
                             float* vel, float* acc, int nthreads);
int mp_fd_trajectory_cpu_f64(const mp_model* model, const double* theta0, const double* dtheta0, const double* taumat,
                             const double* Ftipmat, int64_t B, int64_t N, const double* g, double dt, int intRes, float* pos,
                             float* vel, float* acc, int nthreads);
/* Batched inverse kinematics on the host: the iteration of mp_inverse_kinematics_f64 (reference kinematics/ik.py:39-311), one
 * problem after the other per thread; same arguments and results as mp_inverse_kinematics_host_f64. */
int mp_inverse_kinematics_cpu_f64(const mp_model* model, const double* T_desired, const double* theta0, int64_t B,
                                  const double* joint_limits, double eomg, double ev, int max_iterations, double damping,
                                  double step_cap, double weight_orientation, double weight_position, int adaptive_tuning,
                                  int backtracking, uint32_t seed, double* theta, int32_t* success, int32_t* iterations,
                                  int32_t* restarts, int nthreads);
int mp_cartesian_trajectory_cpu_f32(const double* Xstart, const double* Xend, int64_t B, int64_t N, double Tf, int method,
                                    float* pos, float* vel, float* acc, float* orient, int nthreads);
int mp_pd_regulation_cpu_f64(const mp_model* model, const double* theta0, const double* theta_des, const double* Kp, const double* Kd,
                             int64_t K, const double* g, double dt, int steps, double* errors, int32_t* count, int nthreads);

/* ---- multi-GPU: one process per GPU, RCCL over xGMI (new; the reference is single-device) ------
 * Trajectory batches are sharded over ranks with no exchange during compute; the only collective is
 * the all-gather that reassembles the torque history.  Rank 0 creates the id, the launcher
 * broadcasts its 128 bytes out of band (bench.py uses the torch.distributed gloo store). */
int mp_comm_unique_id(uint8_t id[MP_UNIQUE_ID_BYTES]);
int mp_comm_create(mp_ctx* ctx, const uint8_t id[MP_UNIQUE_ID_BYTES], int nranks, int rank, mp_comm** out);
int mp_comm_destroy(mp_comm* comm);
/* d_recv (nranks * bytes_per_rank) <- every rank's d_send (bytes_per_rank); enqueued on the compute
 * stream after the kernels already queued there. */
int mp_comm_allgather(mp_comm* comm, const void* d_send, void* d_recv, size_t bytes_per_rank);
/* Uneven shards (B % nranks != 0; sharding.shard_range gives the first B % nranks ranks one trajectory more): rank r
 * contributes bytes_of_rank[r] bytes and d_recv receives the shards back to back in rank order.  Grouped ncclSend / ncclRecv
 * per peer on the compute stream (ncclAllGather needs equal counts).  Every rank passes the same nranks-entry array. */
int mp_comm_allgatherv(mp_comm* comm, const void* d_send, void* d_recv, const size_t* bytes_of_rank);
/* The same reassembly, overlapped with compute.  d_all holds nranks slots of bytes_per_rank; a rank writes ITS slot
 * chunk by chunk with ordinary launches on the compute stream (output pointer = slot + offset) and, after the launches
 * of a chunk, calls mp_comm_exchange_chunk: the bytes [offset, offset + nbytes) of its slot go to every peer, every
 * peer's same range arrives in that peer's slot (one ncclSend + ncclRecv per peer in a group: xGMI is point to point),
 * on the communicator's own stream, ordered behind the compute stream's tail - so the kernels of the next chunk run
 * beside the exchange.  Every rank must issue the same sequence of chunks.  mp_comm_join makes the compute stream
 * wait for all exchanges issued so far (call it before anything reads d_all). */
int mp_comm_exchange_chunk(mp_comm* comm, void* d_all, size_t bytes_per_rank, size_t offset, size_t nbytes);
/* mp_comm_exchange_chunk for uneven shards: rank r's slot starts slot_offset[r] bytes into d_all, and this chunk is
 * [chunk_offset[r], chunk_offset[r] + chunk_bytes[r]) of it; nranks entries each, the same arrays on every rank. */
int mp_comm_exchange_chunk_v(mp_comm* comm, void* d_all, const size_t* slot_offset, const size_t* chunk_offset,
                             const size_t* chunk_bytes);
/* Buffer lifetime: mp_free hands a buffer back to the pool immediately, and the pool orders reuse with respect to the
 * context's COMPUTE stream only.  A buffer a communicator is still reading or writing on its own stream (after
 * mp_comm_exchange_chunk) must therefore not be freed before mp_comm_join (mp_comm_allgather runs on the compute
 * stream and needs no join). */
int mp_comm_join(mp_comm* comm);

#ifdef __cplusplus
}
#endif
#endif /* MANIPULA_HIP_H */
