// C ABI of libmanipula_hip.so — see include/manipula_hip.h for the contract.
// Host-only translation unit (no device code): context / streams / pooled device memory / events,
// model handles, argument validation and the host-pointer convenience wrappers.
#include <hip/hip_runtime_api.h>

#include <dlfcn.h>

#include <atomic>
#include <set>
#include <cstdarg>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <functional>
#include <map>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "../../include/manipula_hip.h"
#include "mp_jit.h"
#include "mp_ik.h"
#include "mp_kernels.h"
#include "mp_model_compile.h"
#include "mp_handles.h"

// ------------------------------------------------------------------------------------- objects
struct MpSpec {  // run-time specialised kernels of one model on one device
  hipModule_t mod = nullptr;
  hipModule_t mod_ilp = nullptr;  // second program (mp_jit part 1): id_s / id_co, compiled with the max-ILP scheduling strategy
  hipFunction_t traj_id_pk[2] = {nullptr, nullptr}, fd_traj[2] = {nullptr, nullptr};
  hipFunction_t id_d[2] = {nullptr, nullptr}, fk_jac_id_d[2] = {nullptr, nullptr};
  hipFunction_t ik = nullptr;
  hipFunction_t fd_s[2] = {nullptr, nullptr}, fd_d[2] = {nullptr, nullptr};  // forward dynamics per row, float32 / float64
  hipFunction_t id_s[2] = {nullptr, nullptr};                                  // inverse dynamics, float32, one row per lane
  hipFunction_t fd_traj_tm[2] = {nullptr, nullptr};                            // the roll-out on the time-major device layout
  hipFunction_t id_co[2] = {nullptr, nullptr};   // id_s with whole-line non-temporal row movement through LDS (full waves only)
  hipFunction_t id_hard[2] = {nullptr, nullptr}, traj_id_hard[2] = {nullptr, nullptr};  // the float64 pass over handed-over rows
};
struct mp_ctx {
  int device = -1;
  hipStream_t compute = nullptr;
  hipStream_t copy = nullptr;      // host -> device leg of the chunked host-buffer pipeline
  hipStream_t copy_out = nullptr;  // device -> host leg
  hipStream_t aux = nullptr;       // the shader-clock sampler's stream (mp_clock_sample_begin; created on first use)
  unsigned long long* clock_buf = nullptr;  // its stamps
  bool clock_sampling = false;
  std::map<size_t, std::vector<void*>> free_by_size;  // pool: exact-size free lists
  std::map<void*, size_t> live;                        // every buffer handed out -> its size
  std::map<uint64_t, void*> dev_models;                // model uid -> float32 model resident on this device
  std::map<uint64_t, void*> dev_big[2];                // model uid -> MpBigModel<float> [0] / <double> [1] resident on this device
  std::map<uint64_t, MpSpec> specs;                    // model uid -> specialised kernels (mp_model_specialize)
  int compute_units = 0;
  uint64_t uid = 0;                                    // never reused (graphs identify their context by it, not by address)
  bool capturing = false;                              // between mp_graph_begin and mp_graph_end
  bool stream_exported = false;                        // mp_ctx_get_stream has handed the compute stream out: no pass stays parked any more
  void* queue_counter = nullptr;                       // 8-byte work-queue head of the IK kernel (lazily allocated)
  // The float64 pass over the ill-conditioned rows of a float32 inverse-dynamics launch (attach_hard_list / hard_defer /
  // hard_flush).  The pass costs ~5 us of launch + memory latency however few rows it holds - 8 % of c2's kernel - so it is NOT
  // launched behind every float32 kernel: a launch parks it (`busy`), and parked passes are run, up to four in ONE kernel, before
  // anything that could see the difference - any other entry point, a float32 launch whose arrays overlap theirs, a fifth launch.
  struct HardSlot {
    unsigned* rows = nullptr;     // row indices
    unsigned* ctrl = nullptr;     // two counters used alternately: the pass zeroes the one the list's NEXT user counts in
    unsigned cap = 0, uses = 0;
    bool busy = false;
    bool orphan = false;          // attached to a launch whose pass never followed (a failed launch): its counters are reset on reuse
    unsigned long long seq = 0;   // launch order
    hipFunction_t fn = nullptr;   // the specialised pass of its model, or null: the generic one (k_id_hard_batch) on ...
    const MpModel<float>* gen_dm = nullptr;  // ... this device copy of the model,
    int gen_n = 0;                           // ... this joint count
    bool gen_ftip = false;                   // ... and with / without a tip wrench
    MpCall<float> C;
    const float *q = nullptr, *qd = nullptr, *qdd = nullptr;
    float* tau = nullptr;
    unsigned nrows = 0;
    unsigned nt = 0;              // generated rows (the fused kernels): timesteps per trajectory, q / qd = start / end points, qdd = the time table
    size_t bytes = 0;             // of the torque array
    size_t bytes_in = 0;          // of each input array the caller owns (q, qd, qdd; start, end)
  };
  static constexpr int kHardSlots = MP_HARD_BATCH;
  // The slots of one owner: the context's own (launches on the stream), or those of a launch graph being captured - a graph keeps
  // its lists for its lifetime, its passes are nodes of the graph, and its counters are zeroed by a node ahead of their first user.
  struct HardPool {
    HardSlot slot[kHardSlots];
    unsigned long long seq = 0;
    std::vector<void*> retired;  // lists outgrown during a capture: earlier nodes of the graph still point at them
  };
  HardPool own_pool;
  HardPool* hp = &own_pool;      // the pool launches attach to: own_pool, or the open capture's
  double* time_tab = nullptr;                          // per-timestep time-scaling table of the fused kernels
  long tab_cap = 0, tab_Nt = -1;                       // its capacity in timesteps / the call it currently holds
  double tab_Tf = 0;
  int tab_method = 0;
  bool tab_volatile = false;                           // a launch graph holds a table kernel: never trust the cache again
  std::vector<void*> retired_tabs;                     // outgrown tables that captured graphs may still reference
  int live_graphs = 0;                                 // launch graphs captured on this context and not yet destroyed
  std::vector<hipModule_t> retired_mods;               // code objects / device models of destroyed models that a live graph (or
  std::vector<void*> retired_bufs;                     //   an open capture) may still reference: released with the last graph
  std::recursive_mutex mu;                             // serialises the entry points of this context (CTX_ENTER)
  // profiling (mp_ctx_set_profiling): every device-pointer entry point brackets its launches with a timed HIP event pair
  // on the compute stream and an roctx range; the pairs are resolved when the figures are read (mp_ctx_profile)
  bool profiling = false;
  struct Pending { hipEvent_t a, b; };
  std::vector<Pending> prof_pending;
  double prof_total_ms = 0, prof_last_ms = 0;
  long long prof_launches = 0;
};
struct mp_event {
  hipEvent_t ev = nullptr;
  int device = -1;
};

struct mp_graph {
  hipGraph_t graph = nullptr;
  hipGraphExec_t exec = nullptr;
  int device = -1;
  mp_ctx* ctx = nullptr;  // the context it was captured on (only dereferenced while that context is still registered ...
  uint64_t ctx_uid = 0;   // ... under the SAME never-reused id: a later context may be allocated at the same address)
  mp_ctx::HardPool* pool = nullptr;  // row lists + counters of the float64 passes captured in it
};

namespace {
thread_local char g_err[512] = "";

int set_err(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  std::vsnprintf(g_err, sizeof g_err, fmt, ap);
  va_end(ap);
  return code;
}
int hip_err(hipError_t e, const char* what) {
  return set_err(MP_ERR_HIP, "%s: %s", what, hipGetErrorString(e));
}
#define HIP_TRY(expr)                                   \
  do {                                                  \
    hipError_t e_ = (expr);                             \
    if (e_ != hipSuccess) return hip_err(e_, #expr);    \
  } while (0)
#define REQUIRE(cond, ...)                                        \
  do {                                                            \
    if (!(cond)) return set_err(MP_ERR_INVALID, __VA_ARGS__);     \
  } while (0)

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
const double kG[3] = {0.0, 0.0, -9.81};  // reference planning/trajectory_dynamics.py:54

bool any_nonzero(const double* F) {
  if (!F) return false;
  for (int k = 0; k < 6; ++k)
    if (F[k] != 0.0) return true;
  return false;
}
int bind(mp_ctx* ctx) {
  HIP_TRY(hipSetDevice(ctx->device));
  return MP_OK;
}
// Every entry point that touches a context starts with CTX_ENTER: it serialises the callers of one context (the pool,
// the specialisation table and the capture flag are plain containers; ctypes releases the GIL during a call, so
// Python threads sharing a planner do arrive concurrently - the reference runs its planners from several threads in
// tests/test_trajectory_planning.py:1375) and binds the calling thread to the context's device.  Recursive: the
// host-buffer entry points call the device-pointer ones.
// ... and runs the float64 passes that float32 inverse-dynamics launches have parked (hard_flush): whatever the entry point
// enqueues next may read the torques they write.  The float32 inverse-dynamics entry points, whose point it is NOT to run one
// pass per launch, enter with CTX_ENTER_NOJOIN and flush only when their arrays overlap a parked pass's.
int hard_flush(mp_ctx* ctx);
int hard_park_or_run(mp_ctx* ctx, const void* a, const void* b, const void* c, const void* out, size_t bytes_out, size_t bytes_in);
#define CTX_ENTER_NOJOIN(ctx)                                  \
  std::lock_guard<std::recursive_mutex> ctx_lock_((ctx)->mu);  \
  if (int rc_enter_ = bind(ctx)) return rc_enter_
#define CTX_ENTER(ctx)                                         \
  CTX_ENTER_NOJOIN(ctx);                                       \
  if (int rc_join_ = hard_flush(ctx)) return rc_join_
// every live context, so that mp_model_destroy can drop the per-context state of a model (specialised code object,
// device-resident copy) instead of leaving it to mp_ctx_destroy
std::mutex g_ctxs_mu;
std::set<mp_ctx*> g_ctxs;

// roctx ranges (rocprofv3 --marker-trace shows them): libroctx64 is dlopen'd on first use only, absent = no ranges
struct Roctx {
  int (*push)(const char*) = nullptr;
  int (*pop)() = nullptr;
  Roctx() {
    for (const char* name : {"libroctx64.so.4", "libroctx64.so", "/opt/rocm/lib/libroctx64.so"}) {
      if (void* h = dlopen(name, RTLD_NOW | RTLD_LOCAL)) {
        push = reinterpret_cast<int (*)(const char*)>(dlsym(h, "roctxRangePushA"));
        pop = reinterpret_cast<int (*)()>(dlsym(h, "roctxRangePop"));
        if (push && pop) return;
        push = nullptr; pop = nullptr;
      }
    }
  }
};
const Roctx& roctx() { static const Roctx r; return r; }

// Scope of one device-pointer entry point when the context profiles: an roctx range named after the entry point and a
// timed HIP event pair around whatever it enqueues on the compute stream (the replacement of the reference's
// profile_start / CUDA profiler hooks, planning/trajectory_planning.py:295-296, cuda_kernels/_runtime.py).
struct KernelScope {
  mp_ctx* ctx;
  hipEvent_t a = nullptr, b = nullptr;
  bool ranged = false;
  KernelScope(mp_ctx* c, const char* name) : ctx(c) {
    if (!c->profiling || c->capturing) return;
    if (roctx().push) { roctx().push(name); ranged = true; }
    if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess || hipEventRecord(a, c->compute) != hipSuccess) {
      if (a) (void)hipEventDestroy(a);
      if (b) (void)hipEventDestroy(b);
      a = b = nullptr;
    }
  }
  ~KernelScope() {
    if (a && b) {
      if (hipEventRecord(b, ctx->compute) == hipSuccess) ctx->prof_pending.push_back({a, b});
      else { (void)hipEventDestroy(a); (void)hipEventDestroy(b); }
      // Nobody may ever read the figures (profiling is a flag of the shared context; other users of it never call
      // mp_ctx_profile): pairs that have completed are folded into the totals here, oldest first, once a few are pending,
      // so a long-lived process holds a bounded number of events.
      if (ctx->prof_pending.size() > 32) {
        size_t done = 0;
        while (done < ctx->prof_pending.size() && hipEventQuery(ctx->prof_pending[done].b) == hipSuccess) {
          float ms = 0.f;
          if (hipEventElapsedTime(&ms, ctx->prof_pending[done].a, ctx->prof_pending[done].b) == hipSuccess) {
            ctx->prof_total_ms += ms; ctx->prof_last_ms = ms; ctx->prof_launches += 1;
          }
          (void)hipEventDestroy(ctx->prof_pending[done].a);
          (void)hipEventDestroy(ctx->prof_pending[done].b);
          ++done;
        }
        (void)hipGetLastError();  // hipErrorNotReady of the first unfinished pair is not an error
        ctx->prof_pending.erase(ctx->prof_pending.begin(), ctx->prof_pending.begin() + (long)done);
      }
    }
    if (ranged) roctx().pop();
  }
};
#define PROFILE_SCOPE(ctx, name) KernelScope kernel_scope_((ctx), (name))

template <typename T> const MpModel<T>& pick(const mp_model* m);
template <> const MpModel<float>& pick<float>(const mp_model* m) { return m->f; }
template <> const MpModel<double>& pick<double>(const mp_model* m) { return m->d; }

template <typename T>
void make_call(const mp_model* m, const double* g, const double* Ftip, MpCall<T>* c) {
  MpCall<double> cd;
  if (m->big) mp_make_call(m->bd, g ? g : kG, Ftip, &cd);
  else mp_make_call(m->d, g ? g : kG, Ftip, &cd);
  mp_call_cast(cd, c);
}

// RAII device scratch from the pool for the *_host wrappers
struct Scratch {
  mp_ctx* ctx;
  std::vector<void*> bufs;
  explicit Scratch(mp_ctx* c) : ctx(c) {}
  ~Scratch() { for (void* p : bufs) mp_free(ctx, p); }
  int get(size_t bytes, void** p) {
    int rc = mp_malloc(ctx, bytes ? bytes : 16, p);
    if (rc == MP_OK) bufs.push_back(*p);
    return rc;
  }
};
// rows per chunk of the host-buffer pipeline (MANIPULAPY_HIP_HOST_CHUNK_ROWS, default 512 Ki rows), rounded up to a
// multiple of 4: every chunk then starts on a 16-byte boundary for any row size (4 rows x n x 4 bytes), which the
// device-pointer entry points require, and only the last chunk can end on an odd row
int64_t host_chunk_rows() {
  static const int64_t rows = [] {
    const char* e = getenv("MANIPULAPY_HIP_HOST_CHUNK_ROWS");
    long long v = e ? atoll(e) : 0;
    if (v <= 0) v = 512 * 1024;
    return (int64_t)((v + 3) & ~3LL);
  }();
  return rows;
}
// true when `p` is page-locked host memory the device can DMA directly (mp_host_alloc, hipHostMalloc, hipHostRegister)
bool is_pinned_host(const void* p) {
  hipPointerAttribute_t a;
  if (hipPointerGetAttributes(&a, p) != hipSuccess) {
    (void)hipGetLastError();  // an ordinary pageable pointer is reported as an error: clear it
    return false;
  }
  return a.type == hipMemoryTypeHost;
}

// events of one pipelined call, destroyed on every exit path
struct EventList {
  std::vector<hipEvent_t> evs;
  ~EventList() { for (hipEvent_t e : evs) (void)hipEventDestroy(e); }
  int make(hipEvent_t* out) {
    hipError_t he = hipEventCreateWithFlags(out, hipEventDisableTiming);
    if (he != hipSuccess) return hip_err(he, "hipEventCreate");
    evs.push_back(*out);
    return MP_OK;
  }
};
#define H2D(dst, src, bytes) HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->compute))
#define D2H(dst, src, bytes) HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->compute))


// MANIPULAPY_HIP_SPECIALIZE=0: ignore specialised kernels (A/B measurements)
bool specialize_enabled() {
  static const bool on = [] { const char* e = getenv("MANIPULAPY_HIP_SPECIALIZE"); return !(e && e[0] == '0'); }();
  return on;
}
const MpSpec* find_spec(mp_ctx* ctx, const mp_model* model) {
  if (!specialize_enabled()) return nullptr;
  auto it = ctx->specs.find(model->uid);
  return it == ctx->specs.end() ? nullptr : &it->second;
}
int launch_spec(mp_ctx* ctx, hipFunction_t fn, long threads, void** args, unsigned block = 256) {
  const unsigned grid = (unsigned)((threads + block - 1) / block);
  HIP_TRY(hipModuleLaunchKernel(fn, grid, 1, 1, block, 1, 1, 0, ctx->compute, args, nullptr));
  return MP_OK;
}

// While a capture is open on this thread (hipStreamCaptureModeThreadLocal) an allocation or a blocking copy is refused; the few
// that belong to a captured launch (its row list, a model's device copy on first use) run under the relaxed mode - they execute
// at once, outside the graph.
struct RelaxedCapture {
  hipStreamCaptureMode mode = hipStreamCaptureModeRelaxed;
  bool on;
  explicit RelaxedCapture(bool capturing) : on(capturing) { if (on) (void)hipThreadExchangeStreamCaptureMode(&mode); }
  ~RelaxedCapture() { if (on) (void)hipThreadExchangeStreamCaptureMode(&mode); }
};
// true when [p, p + bytes) lies inside one allocation of this context's pool (mp_malloc): memory the library owns, which no
// caller can read, free or hand to another allocator behind the library's back
bool pool_owned(mp_ctx* ctx, const void* p, size_t bytes) {
  if (!p) return true;
  auto it = ctx->live.upper_bound(const_cast<void*>(p));
  if (it == ctx->live.begin()) return false;
  --it;
  const char* base = (const char*)it->first;
  return (const char*)p >= base && (const char*)p + bytes <= base + it->second;
}
// float32 model resident in device memory (read by the *_dm kernels with scalar loads), followed in the same buffer by the
// float64 model (read by the float64 re-evaluation of ill-conditioned float32 rows in the generic kernels, MpCall::cold_model)
constexpr size_t kDevModelD = (sizeof(MpModel<float>) + 255) & ~(size_t)255;  // offset of the float64 copy
int device_model(mp_ctx* ctx, const mp_model* model, const MpModel<float>** out) {
  auto it = ctx->dev_models.find(model->uid);
  if (it == ctx->dev_models.end()) {
    void* d = nullptr;
    if (ctx->capturing) {
      // first use inside a capture: the copy is made at once, outside the graph - on the context's upload stream (not capturing;
      // a blocking hipMemcpy would tie the legacy stream to the capturing one and is refused) and waited for, so the replays'
      // kernels - and the eager launches after the capture - find it
      RelaxedCapture relaxed(true);
      const size_t bytes = (kDevModelD + sizeof(MpModel<double>) + 255) & ~size_t(255);
      HIP_TRY(hipMalloc(&d, bytes));
      ctx->live[d] = bytes;
      hipError_t he = hipMemcpyAsync(d, &model->f, sizeof(MpModel<float>), hipMemcpyHostToDevice, ctx->copy);
      if (he == hipSuccess) he = hipMemcpyAsync((char*)d + kDevModelD, &model->d, sizeof(MpModel<double>), hipMemcpyHostToDevice, ctx->copy);
      if (he == hipSuccess) he = hipStreamSynchronize(ctx->copy);
      if (he != hipSuccess) { ctx->free_by_size[bytes].push_back(d); return hip_err(he, "device copy of the model (first use inside a capture)"); }
    } else {
      if (int rc = mp_malloc(ctx, kDevModelD + sizeof(MpModel<double>), &d)) return rc;
      // (on the compute stream and waited for: the first kernel that reads the copy follows on that stream, and the streams are
      // non-blocking - nothing orders them with a copy on the null stream)
      HIP_TRY(hipMemcpyAsync(d, &model->f, sizeof(MpModel<float>), hipMemcpyHostToDevice, ctx->compute));
      HIP_TRY(hipMemcpyAsync((char*)d + kDevModelD, &model->d, sizeof(MpModel<double>), hipMemcpyHostToDevice, ctx->compute));
      HIP_TRY(hipStreamSynchronize(ctx->compute));
    }
    it = ctx->dev_models.emplace(model->uid, d).first;
  }
  *out = static_cast<const MpModel<float>*>(it->second);
  return MP_OK;
}
// a float32 call's constants + where its kernels find the float64 model.  Generic kernels only (the specialised programs carry
// the literal).
void make_call_f32(mp_ctx* ctx, const mp_model* model, const double* g, const double* Ftip, MpCall<float>* c) {
  make_call<float>(model, g, Ftip, c);
  if (model->big || find_spec(ctx, model)) return;
  auto it = ctx->dev_models.find(model->uid);
  if (it == ctx->dev_models.end()) {
    const MpModel<float>* dm = nullptr;
    if (device_model(ctx, model, &dm) != MP_OK) return;
    it = ctx->dev_models.find(model->uid);
  }
  if (it != ctx->dev_models.end()) c->cold_model = (const char*)it->second + kDevModelD;
}
// The list a float32 inverse-dynamics launch of `rows` rows leaves its ill-conditioned rows in for the float64 pass
// (csrc/mp_bodies.h, mp_push_hard_rows / mp_body_id_hard): room for one row in eight, at least 65 536 (c2-distributed rows flag 0.5 - 1.5 %; a list that
// overflows makes the pass evaluate every row of the launch).  Returns the slot (c carries its pointers), or null = no list, the
// kernels re-evaluate in place: more than 2^32 rows, or the switch.  A launch that is being CAPTURED takes its list from the
// capture's own pool (mp_graph_begin): the graph owns it, the pass is a node of the graph.
mp_ctx::HardSlot* attach_hard_list(mp_ctx* ctx, long rows, MpCall<float>* c) {
  // MANIPULAPY_HIP_HARD_PASS=0: no list - the path a launch of 2^32 rows or a failed list allocation takes (tested that way)
  static const bool on = !(getenv("MANIPULAPY_HIP_HARD_PASS") && getenv("MANIPULAPY_HIP_HARD_PASS")[0] == '0');
  if (!on || rows >= 0xffffffffL) return nullptr;
  mp_ctx::HardPool* pool = ctx->hp;
  mp_ctx::HardSlot* hs = nullptr;
  for (auto& cand : pool->slot)
    if (!cand.busy) { hs = &cand; break; }
  if (!hs) {  // all parked: run them (one kernel), then take the first
    if (hard_flush(ctx) != MP_OK) return nullptr;
    hs = &pool->slot[0];
  }
  const unsigned need = (unsigned)std::min<long>(std::max<long>(rows / 8, 1L << 16), 1L << 28);
  if (hs->cap < need) {
    RelaxedCapture relaxed(ctx->capturing);
    if (!ctx->capturing && hipStreamSynchronize(ctx->compute) != hipSuccess) return nullptr;
    // EVERY free slot of the pool gets its counters and a list of this size now, not only the one this launch takes: a slot's first
    // use in the middle of a run of launches put a stream synchronisation and two allocations between two 65 us kernels (round 5's
    // "cold" c2 figure, 0.08 - 0.12 ms against 0.065 sustained: profiles/r06_cold_before_windows.json - the kernels themselves
    // ran at their sustained duration).  A parked slot keeps its list (its pass reads it) and grows when it is next taken.
    // (zeroed ON the compute stream: a plain hipMemset of device memory is not ordered with a kernel launched on another,
    // non-blocking stream right behind it - a kernel that met the allocation's old bytes as its counter took the list for full and
    // re-evaluated in place, correct but not bit-equal to the pass: seen once, as 27 elements of a 240 000-element comparison.
    // In a capture this is the node that zeroes the graph's counters ahead of their first user, on every replay.)
    auto grow = [&](mp_ctx::HardSlot* s) -> bool {
      if (!s->ctrl && (hipMalloc((void**)&s->ctrl, 2 * sizeof(unsigned)) != hipSuccess ||
                       hipMemsetAsync(s->ctrl, 0, 2 * sizeof(unsigned), ctx->compute) != hipSuccess)) {
        s->ctrl = nullptr;
        return false;
      }
      if (s->rows && ctx->capturing) pool->retired.push_back(s->rows);  // earlier nodes of the graph still write it
      else if (s->rows) (void)hipFree(s->rows);
      s->rows = nullptr; s->cap = 0;
      if (hipMalloc((void**)&s->rows, (size_t)need * sizeof(unsigned)) != hipSuccess) { s->rows = nullptr; return false; }
      s->cap = need;
      return true;
    };
    if (!grow(hs)) return nullptr;
    for (auto& other : pool->slot)
      if (&other != hs && !other.busy && other.cap < need && !grow(&other)) break;  // (a slot left small grows when it is taken)
  }
  if (hs->orphan && hipMemsetAsync(hs->ctrl, 0, 2 * sizeof(unsigned), ctx->compute) != hipSuccess) return nullptr;
  hs->orphan = true;  // until its pass is parked (hard_defer) or launched (hard_passed)
  const unsigned turn = hs->uses++ & 1u;
  c->hard_rows = hs->rows; c->hard_ctrl = hs->ctrl + turn; c->hard_next = hs->ctrl + (turn ^ 1u); c->hard_cap = hs->cap;
  c->hard_row_base = 0;
  return hs;
}
void free_hard_pool(mp_ctx::HardPool* pool) {
  for (auto& hs : pool->slot) {
    if (hs.rows) (void)hipFree(hs.rows);
    if (hs.ctrl) (void)hipFree(hs.ctrl);
    hs.rows = nullptr; hs.ctrl = nullptr; hs.cap = 0; hs.busy = false; hs.gen_dm = nullptr;
  }
  for (void* p : pool->retired) (void)hipFree(p);
  pool->retired.clear();
}
unsigned hard_pass_blocks(long rows) {
  return (unsigned)std::min<long>((rows / 8 + 63) / 64 + 1, 1024L);
}
// the pass of the list's launch has been enqueued directly (the fused paths); returns rc for tail calls
inline int hard_passed(mp_ctx::HardSlot* hs, int rc) {
  if (rc == MP_OK) hs->orphan = false;
  return rc;
}
// park the pass of the launch just enqueued
void hard_defer(mp_ctx* ctx, mp_ctx::HardSlot* hs, hipFunction_t fn, const MpModel<float>* gen_dm, bool gen_ftip, const MpCall<float>& C,
                const float* q, const float* qd, const float* qdd, float* tau, long rows, int n) {
  hs->busy = true; hs->orphan = false; hs->seq = ++ctx->hp->seq; hs->fn = fn; hs->gen_dm = gen_dm; hs->gen_n = n; hs->gen_ftip = gen_ftip;
  hs->C = C; hs->C.hard_row_base = 0;
  hs->q = q; hs->qd = qd; hs->qdd = qdd; hs->tau = tau; hs->nrows = (unsigned)rows; hs->bytes = (size_t)rows * (size_t)n * sizeof(float);
  hs->bytes_in = hs->bytes; hs->nt = 0;
}
// ... of a fused launch: its rows are generated from B start / end points and the context's time table, which therefore must
// not be rewritten while the pass is parked (mp_traj_id_fused_f32 runs the parked passes before it touches the table)
void hard_defer_generated(mp_ctx* ctx, mp_ctx::HardSlot* hs, hipFunction_t fn, const MpCall<float>& C, const float* start, const float* end,
                          const double* tab, float* tau, long rows, int n, long B, unsigned nt) {
  hard_defer(ctx, hs, fn, nullptr, false, C, start, end, (const float*)tab, tau, rows, n);
  hs->bytes_in = (size_t)B * (size_t)n * sizeof(float); hs->nt = nt;
}
// What becomes of the pass just parked.  It STAYS parked only when every array of its launch lies in this context's pool: such
// memory is read, freed and reused through the library alone, and every entry point runs the parked passes first.  Arrays the
// caller allocated itself (hipMalloc, a framework tensor) may be read with the caller's own HIP calls on the compute stream, or
// handed back to another allocator, without the library hearing of it - their pass is enqueued at once, so that the entry point
// returns with the complete result stream-ordered behind it, as the reference's launchers return finished arrays
// (cuda_kernels/trajectory_kernels.py:1043-1081).
int hard_park_or_run(mp_ctx* ctx, const void* a, const void* b, const void* c, const void* out, size_t bytes_out, size_t bytes_in) {
  // ... and once the caller holds the compute stream (mp_ctx_get_stream) it can read pool memory too - mp_malloc returns a plain device
  // pointer - with a copy or a kernel of its own on that stream, behind the library's back: from then on nothing stays parked
  if (!ctx->stream_exported && pool_owned(ctx, a, bytes_in) && pool_owned(ctx, b, bytes_in) && pool_owned(ctx, c, bytes_in) && pool_owned(ctx, out, bytes_out))
    return MP_OK;
  return hard_flush(ctx);
}
// run every parked pass on the compute stream, in launch order; passes of one specialised program share a kernel launch
int hard_flush(mp_ctx* ctx) {
  mp_ctx::HardSlot* order[mp_ctx::kHardSlots];
  int k = 0;
  for (auto& hs : ctx->hp->slot)
    if (hs.busy) order[k++] = &hs;
  if (k == 0) return MP_OK;
  std::sort(order, order + k, [](const mp_ctx::HardSlot* a, const mp_ctx::HardSlot* b) { return a->seq < b->seq; });
  int rc = MP_OK;
  for (int i = 0; i < k;) {
    mp_ctx::HardSlot* h = order[i];
    MpHardBatch B;
    std::memset(&B, 0, sizeof B);
    mp_ctx::HardSlot* ent[MP_HARD_BATCH];
    int m = 0;
    unsigned blocks = 1;
    // passes of one program - a specialised one (fn), or the generic kernel of one (device model, joint count, wrench) - share a launch
    auto same = [&](const mp_ctx::HardSlot* o) {
      return o->fn == h->fn && (h->fn || (o->gen_dm == h->gen_dm && o->gen_n == h->gen_n && o->gen_ftip == h->gen_ftip));
    };
    while (i < k && same(order[i]) && m < MP_HARD_BATCH) {
      mp_ctx::HardSlot* e = order[i++];
      B.C[m] = e->C; B.q[m] = e->q; B.qd[m] = e->qd; B.qdd[m] = e->qdd; B.tau[m] = e->tau; B.rows[m] = e->nrows; B.nt[m] = e->nt;
      blocks = std::max(blocks, hard_pass_blocks((long)e->nrows));
      e->busy = false;
      ent[m++] = e;
    }
    if (rc == MP_OK) {
      void* args[] = {&B};
      hipError_t he = h->fn ? hipModuleLaunchKernel(h->fn, blocks, (unsigned)m, 1, 64, 1, 1, 0, ctx->compute, args, nullptr)
                            : mpk_id_hard_batch(ctx->compute, h->gen_dm, h->gen_n, h->gen_ftip, B, m, blocks);
      if (he != hipSuccess) rc = hip_err(he, "float64 pass of the ill-conditioned float32 rows");
    }
    for (int j = 0; j < m; ++j) ent[j]->orphan = rc != MP_OK;  // a pass that did not run leaves its list's counters as they are
  }
  return rc;
}
// a float32 launch about to be enqueued on [lo, lo + bytes) arrays: parked passes run first if they read or write what it writes, or
// write what it reads
int hard_flush_if_overlapping(mp_ctx* ctx, const void* const* lo, const size_t* bytes, int k) {
  for (auto& hs : ctx->hp->slot) {
    if (!hs.busy) continue;
    const char* mine[4] = {(const char*)hs.q, (const char*)hs.qd, hs.nt ? nullptr : (const char*)hs.qdd, (const char*)hs.tau};
    const size_t size[4] = {hs.bytes_in, hs.bytes_in, hs.bytes_in, hs.bytes};
    for (int i = 0; i < k; ++i) {  // lo[k - 1] is the array the launch WRITES, the others it reads: two reads do not conflict
      const char* a = (const char*)lo[i];
      if (!a) continue;
      for (int j = i == k - 1 ? 0 : 3; j < 4; ++j)
        if (mine[j] && a < mine[j] + size[j] && mine[j] < a + bytes[i]) return hard_flush(ctx);
    }
  }
  return MP_OK;
}
// A parked float64 pass rides with the next float32 launch of its program (the launch's first workgroups work it off) instead of
// waiting for a kernel of its own: both specialised kernels host the path - the fused one (two waves per SIMD by design: the float64
// recursion's registers are free) and the given-rows one (held to its five waves: the carried path spills to scratch, csrc/mp_jit.cpp)
// - and so does the generic k_id_dm.  (A/B against passes as kernels of their own: profiles/r05_ab_e.txt, r05_ab_g.txt.)
// the parked pass (given rows, specialised program `fn`) a launch may carry: the oldest; `self` is the launch's own slot
mp_ctx::HardSlot* pick_rider(mp_ctx* ctx, hipFunction_t fn, const mp_ctx::HardSlot* self, bool generated = false) {
  mp_ctx::HardSlot* best = nullptr;
  for (auto& hs : ctx->hp->slot)
    if (hs.busy && &hs != self && hs.fn == fn && fn && (hs.nt != 0) == generated && (!best || hs.seq < best->seq)) best = &hs;
  return best;
}
template <typename T> void make_call_ctx(mp_ctx* ctx, const mp_model* model, const double* g, const double* Ftip, MpCall<T>* c);
template <> void make_call_ctx<float>(mp_ctx* ctx, const mp_model* model, const double* g, const double* Ftip, MpCall<float>* c) {
  make_call_f32(ctx, model, g, Ftip, c);
}
template <> void make_call_ctx<double>(mp_ctx*, const mp_model* model, const double* g, const double* Ftip, MpCall<double>* c) {
  make_call<double>(model, g, Ftip, c);
}

// MpBigModel<T> of a 9..32-joint model resident in device memory (read by the looped k_dyn_* kernels through scalar loads)
template <typename T> const MpBigModel<T>& pick_big(const mp_model* m);
template <> const MpBigModel<float>& pick_big<float>(const mp_model* m) { return m->bf; }
template <> const MpBigModel<double>& pick_big<double>(const mp_model* m) { return m->bd; }
template <typename T>
int device_big_model(mp_ctx* ctx, const mp_model* model, const MpBigModel<T>** out) {
  auto& table = ctx->dev_big[sizeof(T) == 8 ? 1 : 0];
  auto it = table.find(model->uid);
  if (it == table.end()) {
    REQUIRE(!ctx->capturing, "a model with more than %d joints is uploaded on first use: call once before capturing a launch graph", MP_MAX_DOF);
    void* d = nullptr;
    if (int rc = mp_malloc(ctx, sizeof(MpBigModel<T>), &d)) return rc;
    HIP_TRY(hipMemcpyAsync(d, &pick_big<T>(model), sizeof(MpBigModel<T>), hipMemcpyHostToDevice, ctx->compute));  // (as device_model)
    HIP_TRY(hipStreamSynchronize(ctx->compute));
    it = table.emplace(model->uid, d).first;
  }
  *out = static_cast<const MpBigModel<T>*>(it->second);
  return MP_OK;
}
#define REQUIRE_SMALL(fn)                                                                                              \
  do {                                                                                                                 \
    if (model->big)                                                                                                    \
      return set_err(MP_ERR_UNSUPPORTED, "%s: not available for models with more than %d joints (this one has %d)", fn, \
                     MP_MAX_DOF, model->d.n);                                                                          \
  } while (0)

// float32 calls on a 9..32-joint model: where the kernels find the float64 model for their ill-conditioned rows (mp_dyn.h,
// mp_dyn_row_id_f64); float64 calls need nothing
int big_cold_model(mp_ctx* ctx, const mp_model* model, MpCall<float>* c) {
  const MpBigModel<double>* dd = nullptr;
  if (int rc = device_big_model<double>(ctx, model, &dd)) return rc;
  c->cold_model = dd;
  return MP_OK;
}
int big_cold_model(mp_ctx*, const mp_model*, MpCall<double>*) { return MP_OK; }

template <typename T>
int launch_big_fk_jac_id(mp_ctx* ctx, const mp_model* model, const MpCall<T>& c, bool ftip, const T* q, const T* qd, const T* qdd, T* Tout,
                         T* Jout, T* tau, long rows) {
  const MpBigModel<T>* dm = nullptr;
  if (int rc = device_big_model<T>(ctx, model, &dm)) return rc;
  MpCall<T> cc = c;
  if (int rc = big_cold_model(ctx, model, &cc)) return rc;
  HIP_TRY(mpk_dyn_fk_jac_id<T>(ctx->compute, model->d.n, dm, cc, ftip, q, qd, qdd, Tout, Jout, tau, rows));
  return MP_OK;
}

int launch_id(mp_ctx* ctx, const mp_model* model, const MpCall<double>& c, bool ftip, const double* q, const double* qd,
              const double* qdd, double* tau, long rows) {
  if (model->big) return launch_big_fk_jac_id<double>(ctx, model, c, ftip, q, qd, qdd, nullptr, nullptr, tau, rows);
  if (const MpSpec* sp = find_spec(ctx, model)) {
    MpCall<double> cc = c;
    void* args[] = {&cc, &q, &qd, &qdd, &tau, &rows};
    return launch_spec(ctx, sp->id_d[ftip ? 1 : 0], rows, args);
  }
  HIP_TRY(mpk_id<double>(ctx->compute, model->d, c, ftip, q, qd, qdd, tau, rows));
  return MP_OK;
}
int launch_id(mp_ctx* ctx, const mp_model* model, const MpCall<float>& c, bool ftip, const float* q, const float* qd,
              const float* qdd, float* tau, long rows) {
  if (model->big) return launch_big_fk_jac_id<float>(ctx, model, c, ftip, q, qd, qdd, nullptr, nullptr, tau, rows);
  if (const MpSpec* sp = find_spec(ctx, model)) {
    // one row per lane in scalar arithmetic (a wave64 v_fma_f32 holds its SIMD for 2 cycles, a packed one for 4: the two-rows-per-lane
    // form measured 5 - 7.5 % slower and was removed in round 6)
    MpCall<float> cc = c;
    // ill-conditioned rows go to a float64 pass of their own behind the float32 kernels (see attach_hard_list)
    mp_ctx::HardSlot* hs = attach_hard_list(ctx, rows, &cc);
    auto hard_pass = [&]() -> int {
      if (!hs) return MP_OK;
      hard_defer(ctx, hs, sp->id_hard[ftip ? 1 : 0], nullptr, false, cc, q, qd, qdd, tau, rows, model->d.n);
      return hard_park_or_run(ctx, q, qd, qdd, tau, (size_t)rows * (size_t)model->d.n * sizeof(float), (size_t)rows * (size_t)model->d.n * sizeof(float));
    };
    long done = 0;
    if (rows >= 64) {  // whole waves: rows moved as whole lines, non-temporal (mp_body_id_co)
      long rows64 = rows & ~63L;
      // One float64 pass that is still parked - an earlier launch's of the same program, none of whose arrays this launch
      // touches (the others were run by hard_flush_if_overlapping on the way in) - rides with this launch: its first workgroups
      // work the list off beside the float32 rows (mp_body_id_lead) instead of a kernel of its own doing so later.
      MpLead lead;
      std::memset(&lead, 0, sizeof lead);
      mp_ctx::HardSlot* rider = pick_rider(ctx, sp->id_hard[ftip ? 1 : 0], hs);
      if (rider) {
        lead.C = rider->C; lead.q = rider->q; lead.qd = rider->qd; lead.qdd = rider->qdd; lead.tau = rider->tau; lead.rows = rider->nrows;
        lead.blocks = std::min(hard_pass_blocks((long)rider->nrows), 512u);
      }
      const unsigned grid = (unsigned)(rows64 / MP_JIT_ID_CO_BLOCK) + lead.blocks;
      void* args[] = {&cc, &q, &qd, &qdd, &tau, &rows64, &lead};
      HIP_TRY(hipModuleLaunchKernel(sp->id_co[ftip ? 1 : 0], grid, 1, 1, MP_JIT_ID_CO_BLOCK, 1, 1, 0, ctx->compute, args, nullptr));
      if (rider) { rider->busy = false; rider->orphan = false; }   // its pass is enqueued: the slot is free for the next launch
      done = rows64;
    }
    if (done == rows) return hard_pass();
    const long off = done * model->d.n;  // the last rows (< 64): per-lane rows
    const float *q2 = q + off, *qd2 = qd + off, *qdd2 = qdd + off;
    float* tau2 = tau + off;
    long left = rows - done;
    cc.hard_row_base = (unsigned)done;
    void* args[] = {&cc, &q2, &qd2, &qdd2, &tau2, &left};
    if (int rc = launch_spec(ctx, sp->id_s[ftip ? 1 : 0], left, args)) return rc;
    return hard_pass();
  }
  {  // generic: one row per lane, device-resident model (a capture's first use uploads it outside the graph)
    const MpModel<float>* dm = nullptr;
    if (int rc = device_model(ctx, model, &dm)) return rc;
    MpCall<float> cc = c;
    mp_ctx::HardSlot* hs = cc.cold_model ? attach_hard_list(ctx, rows, &cc) : nullptr;
    // (as the specialised kernels: one parked pass of an earlier launch of this model rides with this launch's first workgroups)
    MpLead lead;
    std::memset(&lead, 0, sizeof lead);
    mp_ctx::HardSlot* rider = nullptr;
    for (auto& cand : ctx->hp->slot)
      if (cand.busy && &cand != hs && !cand.fn && cand.gen_dm == dm && cand.gen_n == model->d.n && cand.gen_ftip == ftip && cand.nt == 0 &&
          cand.C.cold_model && (!rider || cand.seq < rider->seq))
        rider = &cand;
    if (rider) {
      lead.C = rider->C; lead.q = rider->q; lead.qd = rider->qd; lead.qdd = rider->qdd; lead.tau = rider->tau; lead.rows = rider->nrows;
      lead.blocks = std::min((hard_pass_blocks((long)rider->nrows) + 3u) / 4u, 128u);   // 256-lane workgroups
    }
    bool all_rev = true;   // revolute joints only: the kernel instance without the revolute / prismatic blend (csrc/mp_model.h, MpModelRev)
    for (int i = 0; i < model->d.n; ++i) all_rev = all_rev && model->f.j[i].rev == 1.0f;
    HIP_TRY(mpk_id_dm(ctx->compute, dm, model->d.n, cc, ftip, q, qd, qdd, tau, rows, lead, all_rev));
    if (rider) { rider->busy = false; rider->orphan = false; }
    if (!hs) return MP_OK;
    const int n = model->d.n;
    hard_defer(ctx, hs, nullptr, dm, ftip, cc, q, qd, qdd, tau, rows, n);
    return hard_park_or_run(ctx, q, qd, qdd, tau, (size_t)rows * (size_t)n * sizeof(float), (size_t)rows * (size_t)n * sizeof(float));
  }
}

// template bodies shared by the f32 / f64 entry points (C++ linkage)
template <typename T>
static int id_impl(const char* fn, mp_ctx* ctx, const mp_model* model, const T* d_q, const T* d_qd, const T* d_qdd,
                   int64_t rows, const double* g, const double* Ftip, T* d_tau) {
  REQUIRE(ctx && model, "%s: null context or model", fn);
  CTX_ENTER_NOJOIN(ctx);
  {  // pending float64 passes: only those that may still write rows this launch reads or writes make the compute stream wait
    const size_t nb = (size_t)(rows > 0 ? rows : 0) * (size_t)model->d.n * sizeof(T);
    const void* lo[4] = {d_q, d_qd, d_qdd, d_tau};
    const size_t by[4] = {nb, nb, nb, nb};
    if (int rc = (sizeof(T) == 4 && !ctx->profiling) ? hard_flush_if_overlapping(ctx, lo, by, 4) : hard_flush(ctx)) return rc;
  }
  REQUIRE(rows >= 0, "%s: negative row count %lld", fn, (long long)rows);
  if (rows == 0) return MP_OK;
  REQUIRE(d_q && d_qd && d_qdd && d_tau, "%s: null device pointer", fn);
  REQUIRE(aligned16(d_q) && aligned16(d_qd) && aligned16(d_qdd) && aligned16(d_tau),
          "%s: device pointers must be 16-byte aligned", fn);
  MpCall<T> c;
  make_call_ctx<T>(ctx, model, g, Ftip, &c);
  PROFILE_SCOPE(ctx, fn);
  return launch_id(ctx, model, c, any_nonzero(Ftip), d_q, d_qd, d_qdd, d_tau, (long)rows);
}

// specialised FK + Jacobian + ID (float64 only): -1 = none available, otherwise the launch's return code
int launch_fkjid_spec(mp_ctx* ctx, const mp_model* model, const MpCall<double>& c, bool ftip, const double* q, const double* qd,
                      const double* qdd, double* T, double* J, double* tau, long rows) {
  const MpSpec* sp = find_spec(ctx, model);
  if (!sp) return -1;
  MpCall<double> cc = c;
  void* args[] = {&cc, &q, &qd, &qdd, &T, &J, &tau, &rows};
  return launch_spec(ctx, sp->fk_jac_id_d[ftip ? 1 : 0], rows, args, 64);  // one wave per block (per-wave LDS slice)
}
int launch_fkjid_spec(mp_ctx*, const mp_model*, const MpCall<float>&, bool, const float*, const float*, const float*, float*, float*,
                      float*, long) {
  return -1;
}

template <typename T>
static int fkjid_impl(const char* fn, mp_ctx* ctx, const mp_model* model, const T* d_q, const T* d_qd, const T* d_qdd,
                      int64_t rows, const double* g, const double* Ftip, T* d_T, T* d_J, T* d_tau) {
  REQUIRE(ctx && model, "%s: null context or model", fn);
  CTX_ENTER(ctx);
  REQUIRE(rows >= 0, "%s: negative row count %lld", fn, (long long)rows);
  if (rows == 0) return MP_OK;
  REQUIRE(d_q, "%s: null d_q", fn);
  REQUIRE(d_T || d_J || d_tau, "%s: at least one output is required", fn);
  REQUIRE(!d_tau || (d_qd && d_qdd), "%s: d_tau requested without d_qd / d_qdd", fn);
  REQUIRE(aligned16(d_q) && aligned16(d_qd) && aligned16(d_qdd) && aligned16(d_T) && aligned16(d_J) && aligned16(d_tau),
          "%s: device pointers must be 16-byte aligned", fn);
  MpCall<T> c;
  make_call_ctx<T>(ctx, model, g, Ftip, &c);
  PROFILE_SCOPE(ctx, fn);
  if (model->big) return launch_big_fk_jac_id<T>(ctx, model, c, any_nonzero(Ftip), d_q, d_qd, d_qdd, d_T, d_J, d_tau, (long)rows);
  const int src = launch_fkjid_spec(ctx, model, c, any_nonzero(Ftip), d_q, d_qd, d_qdd, d_T, d_J, d_tau, (long)rows);
  if (src >= 0) return src;
  HIP_TRY(mpk_fk_jac_id<T>(ctx->compute, pick<T>(model), c, any_nonzero(Ftip), d_q, d_qd, d_qdd, d_T, d_J, d_tau, (long)rows));
  return MP_OK;
}

// Chunked three-stage pipeline over `rows` rows in chunks of `chunk`: `up(r0, nr)` queues the uploads of a chunk on the
// copy stream, `run(r0, nr)` launches its kernels on the compute stream, `down(r0, nr)` queues its downloads on the
// copy-out stream; events order the stages of one chunk, so the upload of chunk k+1, the kernels of chunk k and the
// download of chunk k-1 overlap (PCIe is full duplex: with page-locked buffers a call costs about its larger
// direction).  Every stage is drained before returning, also on the error path.
template <class Up, class Run, class Down>
static int host_pipeline(mp_ctx* ctx, int64_t rows, int64_t chunk, Up up, Run run, Down down) {
  EventList ev;
  auto body = [&]() -> int {
    // The scratch blocks of this call come from the pool, whose reuse is ordered with respect to the COMPUTE stream
    // only: work still pending there (an asynchronous device-pointer call whose buffer the caller has already freed)
    // must finish before the copy streams may overwrite a recycled block.
    hipEvent_t tail = nullptr;
    if (int rc = ev.make(&tail)) return rc;
    HIP_TRY(hipEventRecord(tail, ctx->compute));
    HIP_TRY(hipStreamWaitEvent(ctx->copy, tail, 0));
    HIP_TRY(hipStreamWaitEvent(ctx->copy_out, tail, 0));
    for (int64_t r0 = 0; r0 < rows; r0 += chunk) {
      const int64_t nr = std::min(chunk, rows - r0);
      hipEvent_t uploaded = nullptr, done = nullptr;
      if (int rc = ev.make(&uploaded)) return rc;
      if (int rc = ev.make(&done)) return rc;
      if (int rc = up(r0, nr)) return rc;
      HIP_TRY(hipEventRecord(uploaded, ctx->copy));
      HIP_TRY(hipStreamWaitEvent(ctx->compute, uploaded, 0));
      if (int rc = run(r0, nr)) return rc;
      HIP_TRY(hipEventRecord(done, ctx->compute));
      HIP_TRY(hipStreamWaitEvent(ctx->copy_out, done, 0));
      if (int rc = down(r0, nr)) return rc;
    }
    return MP_OK;
  };
  const int rc = body();
  (void)hipStreamSynchronize(ctx->copy);
  (void)hipStreamSynchronize(ctx->compute);
  const hipError_t he = hipStreamSynchronize(ctx->copy_out);
  if (rc == MP_OK && he != hipSuccess) return hip_err(he, "hipStreamSynchronize");
  return rc;
}
#define UP(dst, src, bytes) HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->copy))
#define DOWN(dst, src, bytes) HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->copy_out))

template <typename T>
static int id_host_impl(const char* fn, mp_ctx* ctx, const mp_model* model, const T* q, const T* qd, const T* qdd,
                        int64_t rows, const double* g, const double* Ftip, T* tau) {
  REQUIRE(ctx && model, "%s: null context or model", fn);
  CTX_ENTER(ctx);
  REQUIRE(rows >= 0, "%s: negative row count", fn);
  if (rows == 0) return MP_OK;
  REQUIRE(q && qd && qdd && tau, "%s: null host pointer", fn);
  const size_t row_b = (size_t)model->d.n * sizeof(T), bytes = (size_t)rows * row_b;
  Scratch sc(ctx);
  void *dq, *dqd, *dqdd, *dt;
  if (int rc = sc.get(bytes, &dq)) return rc;
  if (int rc = sc.get(bytes, &dqd)) return rc;
  if (int rc = sc.get(bytes, &dqdd)) return rc;
  if (int rc = sc.get(bytes, &dt)) return rc;
  // Chunked three-stage pipeline: upload of chunk k+1 (copy stream), kernel of chunk k (compute stream) and download
  // of chunk k-1 (copy-out stream) overlap; events order the stages of one chunk.  PCIe is full duplex, so with
  // page-locked buffers the call costs about the upload alone.
  // Pageable buffers are staged by the runtime on the calling thread, which serialises the stages anyway (measured:
  // chunking then costs 8 %), so only page-locked calls are chunked.
  const int64_t chunk = host_chunk_rows();
  const bool pinned = is_pinned_host(q) && is_pinned_host(qd) && is_pinned_host(qdd) && is_pinned_host(tau);
  const int64_t nchunks = pinned ? (rows + chunk - 1) / chunk : 1;
  if (nchunks < 2) {
    H2D(dq, q, bytes);
    H2D(dqd, qd, bytes);
    H2D(dqdd, qdd, bytes);
    int rc = id_impl<T>(fn, ctx, model, (T*)dq, (T*)dqd, (T*)dqdd, rows, g, Ftip, (T*)dt);
    if (!rc) rc = hard_flush(ctx);  // the download below reads the torques: a parked float64 pass runs now
    if (rc) {
      (void)hipStreamSynchronize(ctx->compute);
      return rc;
    }
    D2H(tau, dt, bytes);
    HIP_TRY(hipStreamSynchronize(ctx->compute));
    return MP_OK;
  }
  char *cq = (char*)dq, *cqd = (char*)dqd, *cqdd = (char*)dqdd, *ct = (char*)dt;
  return host_pipeline(
      ctx, rows, chunk,
      [&](int64_t r0, int64_t nr) -> int {
        const size_t off = (size_t)r0 * row_b, nb = (size_t)nr * row_b;
        UP(cq + off, (const char*)q + off, nb);
        UP(cqd + off, (const char*)qd + off, nb);
        UP(cqdd + off, (const char*)qdd + off, nb);
        return MP_OK;
      },
      [&](int64_t r0, int64_t nr) -> int {
        const size_t off = (size_t)r0 * row_b;
        if (int rc = id_impl<T>(fn, ctx, model, (T*)(cq + off), (T*)(cqd + off), (T*)(cqdd + off), nr, g, Ftip, (T*)(ct + off))) return rc;
        return hard_flush(ctx);  // the chunk's download follows
      },
      [&](int64_t r0, int64_t nr) -> int {
        const size_t off = (size_t)r0 * row_b;
        DOWN((char*)tau + off, ct + off, (size_t)nr * row_b);
        return MP_OK;
      });
}

template <typename T>
static int mm_impl(const char* fn, mp_ctx* ctx, const mp_model* model, const T* d_q, int64_t rows, T* d_M) {
  REQUIRE(ctx && model, "%s: null context or model", fn);
  CTX_ENTER(ctx);
  REQUIRE(rows >= 0, "%s: negative row count", fn);
  if (rows == 0) return MP_OK;
  REQUIRE(d_q && d_M, "%s: null device pointer", fn);
  REQUIRE(aligned16(d_q) && aligned16(d_M), "%s: device pointers must be 16-byte aligned", fn);
  PROFILE_SCOPE(ctx, fn);
  if (model->big) {
    const MpBigModel<T>* dm = nullptr;
    if (int rc = device_big_model<T>(ctx, model, &dm)) return rc;
    HIP_TRY(mpk_dyn_mass_matrix<T>(ctx->compute, model->d.n, dm, d_q, d_M, (long)rows));
    return MP_OK;
  }
  HIP_TRY(mpk_mass_matrix<T>(ctx->compute, pick<T>(model), d_q, d_M, (long)rows));
  return MP_OK;
}

template <typename T>
static int fdyn_impl(const char* fn, mp_ctx* ctx, const mp_model* model, const T* d_q, const T* d_qd, const T* d_tau,
                     int64_t rows, const double* g, const double* Ftip, T* d_qdd) {
  REQUIRE(ctx && model, "%s: null context or model", fn);
  CTX_ENTER(ctx);
  REQUIRE(rows >= 0, "%s: negative row count", fn);
  if (rows == 0) return MP_OK;
  REQUIRE(d_q && d_qd && d_tau && d_qdd, "%s: null device pointer", fn);
  REQUIRE(aligned16(d_q) && aligned16(d_qd) && aligned16(d_tau) && aligned16(d_qdd), "%s: device pointers must be 16-byte aligned", fn);
  MpCall<T> c;
  make_call<T>(model, g, Ftip, &c);
  const bool ftip = any_nonzero(Ftip);
  PROFILE_SCOPE(ctx, fn);
  if (model->big) {
    const MpBigModel<T>* dm = nullptr;
    if (int rc = device_big_model<T>(ctx, model, &dm)) return rc;
    HIP_TRY(mpk_dyn_forward_dynamics<T>(ctx->compute, model->d.n, dm, c, ftip, d_q, d_qd, d_tau, d_qdd, (long)rows));
    return MP_OK;
  }
  if (const MpSpec* sp = find_spec(ctx, model)) {
    long nr = (long)rows;
    void* args[] = {&c, &d_q, &d_qd, &d_tau, &d_qdd, &nr};
    return launch_spec(ctx, (sizeof(T) == 4 ? sp->fd_s : sp->fd_d)[ftip ? 1 : 0], nr, args);
  }
  HIP_TRY(mpk_forward_dynamics<T>(ctx->compute, pick<T>(model), c, ftip, d_q, d_qd, d_tau, d_qdd, (long)rows));
  return MP_OK;
}

// specialised forward-dynamics roll-out (float32 only): -1 = none available, otherwise the launch's return code
int launch_fd_spec(mp_ctx* ctx, const mp_model* model, const MpCall<float>& c, const float* th0, const float* dth0,
                    const float* taumat, const float* Fm, long B, long Nt, float h, int intRes, float* pos, float* vel, float* acc,
                    bool time_major) {
  const MpSpec* sp = find_spec(ctx, model);
  if (!sp) return -1;
  MpCall<float> cc = c;
  void* args[] = {&cc, &th0, &dth0, &taumat, &Fm, &B, &Nt, &h, &intRes, &pos, &vel, &acc};
  if (time_major) return launch_spec(ctx, sp->fd_traj_tm[Fm ? 1 : 0], B, args, 64);
  return launch_spec(ctx, sp->fd_traj[Fm ? 1 : 0], B, args, 64);  // one wave per block (per-wave LDS tile)
}
int launch_fd_spec(mp_ctx*, const mp_model*, const MpCall<double>&, const double*, const double*, const double*, const double*,
                   long, long, double, int, float*, float*, float*, bool) {
  return -1;
}

template <typename T>
static int fdtraj_impl(const char* fn, mp_ctx* ctx, const mp_model* model, const T* d_theta0, const T* d_dtheta0,
                       const T* d_taumat, const T* d_Ftipmat, int64_t B, int64_t N, const double* g, double dt, int intRes,
                       float* d_pos, float* d_vel, float* d_acc, bool time_major = false) {
  REQUIRE(ctx && model, "%s: null context or model", fn);
  CTX_ENTER(ctx);
  REQUIRE(B >= 0 && N >= 0, "%s: negative B or N", fn);
  REQUIRE(intRes >= 0, "%s: negative intRes", fn);
  if (B == 0 || N == 0) return MP_OK;
  REQUIRE(d_theta0 && d_dtheta0 && d_taumat && d_pos && d_vel && d_acc, "%s: null device pointer", fn);
  REQUIRE(aligned16(d_theta0) && aligned16(d_dtheta0) && aligned16(d_taumat) && aligned16(d_Ftipmat) && aligned16(d_pos) &&
              aligned16(d_vel) && aligned16(d_acc), "%s: device pointers must be 16-byte aligned", fn);
  MpCall<T> c;
  make_call<T>(model, g, nullptr, &c);
  const T h = intRes > 0 ? (T)(dt / intRes) : (T)0;
  PROFILE_SCOPE(ctx, fn);
  if (model->big) {
    const MpBigModel<T>* dm = nullptr;
    if (int rc = device_big_model<T>(ctx, model, &dm)) return rc;
    HIP_TRY(mpk_dyn_fd_traj<T>(ctx->compute, model->d.n, dm, c, d_theta0, d_dtheta0, d_taumat, d_Ftipmat, (long)B, (long)N, h, intRes, d_pos, d_vel,
                               d_acc, time_major));
    return MP_OK;
  }
  const int src = launch_fd_spec(ctx, model, c, d_theta0, d_dtheta0, d_taumat, d_Ftipmat, (long)B, (long)N, h, intRes, d_pos, d_vel, d_acc,
                                 time_major);
  if (src >= 0) return src;  // a specialised kernel exists for this model: launched (0) or failed (error code)
  if (time_major) {
    HIP_TRY(mpk_fd_traj_tm<T>(ctx->compute, pick<T>(model), c, d_theta0, d_dtheta0, d_taumat, d_Ftipmat, (long)B, (long)N, h, intRes,
                              d_pos, d_vel, d_acc));
    return MP_OK;
  }
  HIP_TRY(mpk_fd_traj<T>(ctx->compute, pick<T>(model), c, d_theta0, d_dtheta0, d_taumat, d_Ftipmat, (long)B, (long)N, h, intRes,
                         d_pos, d_vel, d_acc));
  return MP_OK;
}

template <typename T>
static int fdtraj_host_impl(const char* fn, mp_ctx* ctx, const mp_model* model, const T* theta0, const T* dtheta0,
                            const T* taumat, const T* Ftipmat, int64_t B, int64_t N, const double* g, double dt, int intRes,
                            float* pos, float* vel, float* acc) {
  REQUIRE(ctx && model, "%s: null context or model", fn);
  CTX_ENTER(ctx);
  REQUIRE(B >= 0 && N >= 0, "%s: negative B or N", fn);
  if (B == 0 || N == 0) return MP_OK;
  REQUIRE(theta0 && dtheta0 && taumat && pos && vel && acc, "%s: null host pointer", fn);
  const size_t n = (size_t)model->d.n, sb = (size_t)B * n * sizeof(T), tb = (size_t)B * (size_t)N * n * sizeof(T);
  const size_t fb = (size_t)B * (size_t)N * 6 * sizeof(T), ob = (size_t)B * (size_t)N * n * sizeof(float);
  Scratch sc(ctx);
  void *d0, *d1, *dt_, *df = nullptr, *dp, *dv, *da;
  if (int rc = sc.get(sb, &d0)) return rc;
  if (int rc = sc.get(sb, &d1)) return rc;
  if (int rc = sc.get(tb, &dt_)) return rc;
  if (Ftipmat) if (int rc = sc.get(fb, &df)) return rc;
  if (int rc = sc.get(ob, &dp)) return rc;
  if (int rc = sc.get(ob, &dv)) return rc;
  if (int rc = sc.get(ob, &da)) return rc;
  // Page-locked arrays: the batch is cut into chunks of whole trajectories and the three stages overlap - upload of chunk k + 1,
  // roll-out of chunk k, download of chunk k - 1 (host_pipeline, as for the inverse-dynamics and FK entry points).  A roll-out
  // moves (n + 6) sizeof(T) bytes up and 12 n down per step, so with full-duplex PCIe the call costs about its larger direction.
  {
    const bool pinned = is_pinned_host(theta0) && is_pinned_host(dtheta0) && is_pinned_host(taumat) && (!Ftipmat || is_pinned_host(Ftipmat)) &&
                        is_pinned_host(pos) && is_pinned_host(vel) && is_pinned_host(acc);
    const int64_t cb = std::max<int64_t>(64, (host_chunk_rows() / std::max<int64_t>(N, 1)) & ~(int64_t)63);  // trajectories per chunk
    if (pinned && B >= 2 * cb) {
      const size_t srow = n * sizeof(T), trow = (size_t)N * n * sizeof(T), frow = (size_t)N * 6 * sizeof(T), orow = (size_t)N * n * sizeof(float);
      char *c0 = (char*)d0, *c1 = (char*)d1, *ct = (char*)dt_, *cf = (char*)df, *cp = (char*)dp, *cv = (char*)dv, *ca = (char*)da;
      return host_pipeline(
          ctx, B, cb,
          [&](int64_t b0, int64_t nb) -> int {
            UP(c0 + b0 * srow, (const char*)theta0 + b0 * srow, nb * srow);
            UP(c1 + b0 * srow, (const char*)dtheta0 + b0 * srow, nb * srow);
            UP(ct + b0 * trow, (const char*)taumat + b0 * trow, nb * trow);
            if (Ftipmat) UP(cf + b0 * frow, (const char*)Ftipmat + b0 * frow, nb * frow);
            return MP_OK;
          },
          [&](int64_t b0, int64_t nb) -> int {
            return fdtraj_impl<T>(fn, ctx, model, (T*)(c0 + b0 * srow), (T*)(c1 + b0 * srow), (T*)(ct + b0 * trow),
                                  Ftipmat ? (T*)(cf + b0 * frow) : nullptr, nb, N, g, dt, intRes, (float*)(cp + b0 * orow),
                                  (float*)(cv + b0 * orow), (float*)(ca + b0 * orow));
          },
          [&](int64_t b0, int64_t nb) -> int {
            DOWN((char*)pos + b0 * orow, cp + b0 * orow, nb * orow);
            DOWN((char*)vel + b0 * orow, cv + b0 * orow, nb * orow);
            DOWN((char*)acc + b0 * orow, ca + b0 * orow, nb * orow);
            return MP_OK;
          });
    }
  }
  H2D(d0, theta0, sb);
  H2D(d1, dtheta0, sb);
  H2D(dt_, taumat, tb);
  if (Ftipmat) H2D(df, Ftipmat, fb);
  if (int rc = fdtraj_impl<T>(fn, ctx, model, (T*)d0, (T*)d1, (T*)dt_, (T*)df, B, N, g, dt, intRes, (float*)dp, (float*)dv, (float*)da)) return rc;
  D2H(pos, dp, ob);
  D2H(vel, dv, ob);
  D2H(acc, da, ob);
  HIP_TRY(hipStreamSynchronize(ctx->compute));
  return MP_OK;
}
}  // namespace

extern "C" {

// ------------------------------------------------------------------------------ library / device
int mp_version(void) { return 1; }
const char* mp_last_error(void) { return g_err; }

int mp_device_count(int* count) {
  REQUIRE(count, "mp_device_count: null output");
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e == hipErrorNoDevice) { *count = 0; (void)hipGetLastError(); return MP_OK; }
  if (e != hipSuccess) { *count = 0; return hip_err(e, "hipGetDeviceCount"); }
  *count = n;
  return MP_OK;
}

int mp_ctx_create(int device_id, mp_ctx** out) {
  REQUIRE(out, "mp_ctx_create: null output");
  *out = nullptr;
  int n = 0;
  HIP_TRY(hipGetDeviceCount(&n));
  REQUIRE(device_id >= 0 && device_id < n, "mp_ctx_create: device %d not in [0, %d)", device_id, n);
  HIP_TRY(hipSetDevice(device_id));
  mp_ctx* c = new (std::nothrow) mp_ctx;
  REQUIRE(c, "mp_ctx_create: out of host memory");
  c->device = device_id;
  {
    static std::atomic<uint64_t> next_uid{1};
    c->uid = next_uid.fetch_add(1);
  }
  {
    hipDeviceProp_t p;
    if (hipGetDeviceProperties(&p, device_id) == hipSuccess) c->compute_units = p.multiProcessorCount;
  }
  hipError_t e = hipStreamCreateWithFlags(&c->compute, hipStreamNonBlocking);
  if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->copy, hipStreamNonBlocking);
  if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->copy_out, hipStreamNonBlocking);
  if (e != hipSuccess) { delete c; return hip_err(e, "hipStreamCreate"); }
  {
    std::lock_guard<std::mutex> lk(g_ctxs_mu);
    g_ctxs.insert(c);
  }
  *out = c;
  return MP_OK;
}

int mp_ctx_destroy(mp_ctx* ctx) {
  if (!ctx) return MP_OK;
  {
    std::lock_guard<std::mutex> lk(g_ctxs_mu);
    g_ctxs.erase(ctx);
  }
  (void)hipSetDevice(ctx->device);
  if (!ctx->capturing) (void)hard_flush(ctx);  // parked float64 passes: their launches' results are complete before the memory goes
  (void)hipDeviceSynchronize();
  for (auto& pe : ctx->prof_pending) { (void)hipEventDestroy(pe.a); (void)hipEventDestroy(pe.b); }
  for (auto& kv : ctx->specs) {
    if (kv.second.mod_ilp) (void)hipModuleUnload(kv.second.mod_ilp);
    if (kv.second.mod) (void)hipModuleUnload(kv.second.mod);
  }
  for (hipModule_t m : ctx->retired_mods) (void)hipModuleUnload(m);
  for (auto& kv : ctx->live) (void)hipFree(kv.first);
  if (ctx->queue_counter) (void)hipFree(ctx->queue_counter);
  free_hard_pool(&ctx->own_pool);
  if (ctx->hp != &ctx->own_pool) { free_hard_pool(ctx->hp); delete ctx->hp; }  // a capture left open
  if (ctx->time_tab) (void)hipFree(ctx->time_tab);
  for (void* p : ctx->retired_tabs) (void)hipFree(p);
  if (ctx->compute) (void)hipStreamDestroy(ctx->compute);
  if (ctx->copy) (void)hipStreamDestroy(ctx->copy);
  if (ctx->copy_out) (void)hipStreamDestroy(ctx->copy_out);
  if (ctx->aux) (void)hipStreamDestroy(ctx->aux);
  if (ctx->clock_buf) (void)hipFree(ctx->clock_buf);
  delete ctx;
  return MP_OK;
}

int mp_ctx_set_profiling(mp_ctx* ctx, int on) {
  REQUIRE(ctx, "mp_ctx_set_profiling: null context");
  CTX_ENTER(ctx);
  ctx->profiling = on != 0;
  return MP_OK;
}

int mp_ctx_profile(mp_ctx* ctx, double* kernel_ms_total, int64_t* timed_calls, double* kernel_ms_last, int reset) {
  REQUIRE(ctx, "mp_ctx_profile: null context");
  CTX_ENTER(ctx);
  for (auto& pe : ctx->prof_pending) {  // resolve the event pairs recorded since the last read
    float ms = 0.f;
    if (hipEventSynchronize(pe.b) == hipSuccess && hipEventElapsedTime(&ms, pe.a, pe.b) == hipSuccess) {
      ctx->prof_total_ms += ms;
      ctx->prof_last_ms = ms;
      ctx->prof_launches += 1;
    }
    (void)hipEventDestroy(pe.a);
    (void)hipEventDestroy(pe.b);
  }
  ctx->prof_pending.clear();
  if (kernel_ms_total) *kernel_ms_total = ctx->prof_total_ms;
  if (timed_calls) *timed_calls = ctx->prof_launches;
  if (kernel_ms_last) *kernel_ms_last = ctx->prof_last_ms;
  if (reset) { ctx->prof_total_ms = 0; ctx->prof_last_ms = 0; ctx->prof_launches = 0; }
  return MP_OK;
}

int mp_ctx_get_stream(mp_ctx* ctx, void** hip_stream) {
  REQUIRE(ctx && hip_stream, "mp_ctx_get_stream: null argument");
  CTX_ENTER(ctx);  // (parked float64 passes run first: what the caller enqueues behind this call sees complete results)
  ctx->stream_exported = true;  // sticky: every later float32 launch enqueues its float64 pass at once, whoever owns its arrays
  *hip_stream = (void*)ctx->compute;
  return MP_OK;
}

int mp_ctx_synchronize(mp_ctx* ctx) {
  REQUIRE(ctx, "mp_ctx_synchronize: null context");
  CTX_ENTER(ctx);
  HIP_TRY(hipStreamSynchronize(ctx->compute));
  HIP_TRY(hipStreamSynchronize(ctx->copy));
  HIP_TRY(hipStreamSynchronize(ctx->copy_out));
  return MP_OK;
}

int mp_ctx_properties(mp_ctx* ctx, char* name, size_t name_len, int* compute_units, uint64_t* hbm_bytes) {
  REQUIRE(ctx, "mp_ctx_properties: null context");
  hipDeviceProp_t p;
  HIP_TRY(hipGetDeviceProperties(&p, ctx->device));
  if (name && name_len) std::snprintf(name, name_len, "%s (%s)", p.name, p.gcnArchName);
  if (compute_units) *compute_units = p.multiProcessorCount;
  if (hbm_bytes) *hbm_bytes = (uint64_t)p.totalGlobalMem;
  return MP_OK;
}

int mp_selftest(mp_ctx* ctx) {
  REQUIRE(ctx, "mp_selftest: null context");
  CTX_ENTER(ctx);
  Scratch sc(ctx);
  void* d = nullptr;
  if (int rc = sc.get(64 * sizeof(int), &d)) return rc;
  HIP_TRY(hipMemsetAsync(d, 0xFF, 64 * sizeof(int), ctx->compute));
  HIP_TRY(mpk_selftest(ctx->compute, static_cast<int*>(d)));
  int h[64];
  HIP_TRY(hipMemcpyAsync(h, d, sizeof h, hipMemcpyDeviceToHost, ctx->compute));
  HIP_TRY(hipStreamSynchronize(ctx->compute));
  for (int i = 0; i < 64; ++i)
    if (h[i] != i) return set_err(MP_ERR_HIP, "mp_selftest: lane %d wrote %d", i, h[i]);
  return MP_OK;
}

int mp_stream_bandwidth(mp_ctx* ctx, size_t bytes_per_array, int reads, int reps, double* gb_per_s) {
  REQUIRE(ctx && gb_per_s, "mp_stream_bandwidth: null argument");
  const bool nt = reads > 10;  // 11 / 13: the same two kernels with non-temporal loads and stores
  if (nt) reads -= 10;
  REQUIRE(reads == 1 || reads == 3, "mp_stream_bandwidth: reads must be 1 or 3 (plain accesses), 11 or 13 (non-temporal)");
  REQUIRE(bytes_per_array >= 16 && reps >= 1, "mp_stream_bandwidth: nothing to move");
  CTX_ENTER(ctx);
  const size_t nb = bytes_per_array & ~(size_t)15;
  Scratch sc(ctx);
  void* buf[4] = {nullptr, nullptr, nullptr, nullptr};
  for (int k = 0; k < (reads == 3 ? 4 : 2); ++k) {
    if (int rc = sc.get(nb, &buf[k])) return rc;
    HIP_TRY(hipMemsetAsync(buf[k], 0, nb, ctx->compute));
  }
  void *a = buf[0], *d = buf[reads == 3 ? 3 : 1];
  const long n4 = (long)(nb / 16);
  for (int w = 0; w < 3; ++w) HIP_TRY(mpk_stream(ctx->compute, reads, nt, a, buf[1], buf[2], d, n4));
  hipEvent_t e0 = nullptr, e1 = nullptr;
  HIP_TRY(hipEventCreate(&e0));
  if (hipEventCreate(&e1) != hipSuccess) { (void)hipEventDestroy(e0); return set_err(MP_ERR_HIP, "mp_stream_bandwidth: hipEventCreate failed"); }
  hipError_t e = hipEventRecord(e0, ctx->compute);
  for (int r = 0; r < reps && e == hipSuccess; ++r) e = mpk_stream(ctx->compute, reads, nt, a, buf[1], buf[2], d, n4);
  if (e == hipSuccess) e = hipEventRecord(e1, ctx->compute);
  if (e == hipSuccess) e = hipEventSynchronize(e1);
  float ms = 0.f;
  if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  if (e != hipSuccess) return set_err(MP_ERR_HIP, "mp_stream_bandwidth: %s", hipGetErrorString(e));
  *gb_per_s = (double)(reads + 1) * (double)nb * (double)reps / ((double)ms * 1e-3) / 1e9;
  return MP_OK;
}

int mp_stream_bandwidth_mix(mp_ctx* ctx, size_t bytes_per_array, int reads, int writes, int nontemporal, int reps, double* gb_per_s) {
  REQUIRE(ctx && gb_per_s, "mp_stream_bandwidth_mix: null argument");
  REQUIRE(bytes_per_array >= 16 && reps >= 1 && reads >= 0 && writes >= 1, "mp_stream_bandwidth_mix: nothing to move");
  CTX_ENTER(ctx);
  const size_t nb = bytes_per_array & ~(size_t)15;
  Scratch sc(ctx);
  void *a = nullptr, *d = nullptr;
  if (int rc = sc.get(nb * (size_t)std::max(reads, 1), &a)) return rc;
  if (int rc = sc.get(nb * (size_t)writes, &d)) return rc;
  HIP_TRY(hipMemsetAsync(a, 0, nb * (size_t)std::max(reads, 1), ctx->compute));
  HIP_TRY(hipMemsetAsync(d, 0, nb * (size_t)writes, ctx->compute));
  const long n4 = (long)(nb / 16);
  hipError_t e = hipSuccess;
  for (int w = 0; w < 2 && e == hipSuccess; ++w) e = mpk_stream_mix(ctx->compute, reads, writes, nontemporal != 0, a, d, n4);
  if (e == hipErrorInvalidValue) return set_err(MP_ERR_INVALID, "mp_stream_bandwidth_mix: no probe kernel for %d reads : %d writes", reads, writes);
  if (e != hipSuccess) return hip_err(e, "mp_stream_bandwidth_mix");
  hipEvent_t e0 = nullptr, e1 = nullptr;
  HIP_TRY(hipEventCreate(&e0));
  if (hipEventCreate(&e1) != hipSuccess) { (void)hipEventDestroy(e0); return set_err(MP_ERR_HIP, "mp_stream_bandwidth_mix: hipEventCreate failed"); }
  e = hipEventRecord(e0, ctx->compute);
  for (int r = 0; r < reps && e == hipSuccess; ++r) e = mpk_stream_mix(ctx->compute, reads, writes, nontemporal != 0, a, d, n4);
  if (e == hipSuccess) e = hipEventRecord(e1, ctx->compute);
  if (e == hipSuccess) e = hipEventSynchronize(e1);
  float ms = 0.f;
  if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  if (e != hipSuccess) return set_err(MP_ERR_HIP, "mp_stream_bandwidth_mix: %s", hipGetErrorString(e));
  *gb_per_s = (double)(reads + writes) * (double)nb * (double)reps / ((double)ms * 1e-3) / 1e9;
  return MP_OK;
}

// The shader clock the chip holds WHILE the caller's launches run (bench.py: `clock_hz` beside a kernel's time, so that two boxes'
// figures can be compared at equal clock).  begin: a bounded sampler (k_clock_sampler: 8 one-wave blocks, 32 stamp pairs spread over
// about duration_ms) starts on a stream of its own; the caller then launches what it wants measured on the compute stream; end: waits
// for the sampler and returns the median over the blocks of delta s_memtime / delta s_memrealtime x 100 MHz.
constexpr unsigned kClockBlocks = 8, kClockSamples = 32;
int mp_clock_sample_begin(mp_ctx* ctx, double duration_ms) {
  REQUIRE(ctx, "mp_clock_sample_begin: null context");
  REQUIRE(duration_ms > 0 && duration_ms <= 2000.0, "mp_clock_sample_begin: duration %.3f ms outside (0, 2000]", duration_ms);
  CTX_ENTER(ctx);
  REQUIRE(!ctx->capturing, "mp_clock_sample_begin: not while a launch graph is being captured");
  REQUIRE(!ctx->clock_sampling, "mp_clock_sample_begin: a sampler is already running (call mp_clock_sample_end)");
  if (!ctx->aux) HIP_TRY(hipStreamCreateWithFlags(&ctx->aux, hipStreamNonBlocking));
  if (!ctx->clock_buf) HIP_TRY(hipMalloc((void**)&ctx->clock_buf, (size_t)kClockBlocks * kClockSamples * 2 * sizeof(unsigned long long)));
  // one nap = 64 x 127 shader cycles, ~3.9 us at 2.1 GHz
  const unsigned naps = (unsigned)std::max(1.0, duration_ms * 1e3 / ((double)(kClockSamples - 1) * 3.9));
  HIP_TRY(mpk_clock_sampler(ctx->aux, ctx->clock_buf, kClockBlocks, kClockSamples, naps));
  ctx->clock_sampling = true;
  return MP_OK;
}
int mp_clock_sample_end(mp_ctx* ctx, double* clock_hz, double* sampled_ms) {
  REQUIRE(ctx && clock_hz, "mp_clock_sample_end: null argument");
  CTX_ENTER(ctx);
  REQUIRE(ctx->clock_sampling, "mp_clock_sample_end: no sampler running");
  ctx->clock_sampling = false;
  std::vector<unsigned long long> h((size_t)kClockBlocks * kClockSamples * 2);
  HIP_TRY(hipMemcpyAsync(h.data(), ctx->clock_buf, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->aux));
  HIP_TRY(hipStreamSynchronize(ctx->aux));
  std::vector<double> hz;
  double span = 0;
  for (unsigned b = 0; b < kClockBlocks; ++b) {
    const unsigned long long* o = h.data() + (size_t)b * kClockSamples * 2;
    const double cyc = (double)(o[2 * (kClockSamples - 1)] - o[0]), ref = (double)(o[2 * (kClockSamples - 1) + 1] - o[1]);
    if (ref > 0) { hz.push_back(cyc / ref * 1e8); span = std::max(span, ref / 1e5); }
  }
  REQUIRE(!hz.empty(), "mp_clock_sample_end: the sampler recorded nothing");
  std::sort(hz.begin(), hz.end());
  *clock_hz = hz[hz.size() / 2];
  if (sampled_ms) *sampled_ms = span;
  return MP_OK;
}

// ---------------------------------------------------------------------------------- device memory
int mp_malloc(mp_ctx* ctx, size_t bytes, void** d_ptr) {
  REQUIRE(ctx && d_ptr, "mp_malloc: null argument");
  *d_ptr = nullptr;
  CTX_ENTER(ctx);
  REQUIRE(!ctx->capturing, "mp_malloc: not allowed while a launch graph is being captured (mp_graph_begin)");
  if (bytes == 0) bytes = 16;
  bytes = (bytes + 255) & ~size_t(255);
  auto it = ctx->free_by_size.find(bytes);
  if (it != ctx->free_by_size.end() && !it->second.empty()) {
    *d_ptr = it->second.back();
    it->second.pop_back();
    return MP_OK;
  }
  void* p = nullptr;
  hipError_t e = hipMalloc(&p, bytes);
  if (e == hipErrorOutOfMemory) {  // give the pool back and retry once
    (void)hipGetLastError();
    mp_pool_trim(ctx);
    e = hipMalloc(&p, bytes);
  }
  if (e != hipSuccess) return hip_err(e, "hipMalloc");
  ctx->live[p] = bytes;
  *d_ptr = p;
  return MP_OK;
}

int mp_free(mp_ctx* ctx, void* d_ptr) {
  REQUIRE(ctx, "mp_free: null context");
  if (!d_ptr) return MP_OK;
  std::lock_guard<std::recursive_mutex> lk(ctx->mu);
  auto it = ctx->live.find(d_ptr);
  REQUIRE(it != ctx->live.end(), "mp_free: pointer %p was not allocated by this context", d_ptr);
  ctx->free_by_size[it->second].push_back(d_ptr);
  return MP_OK;
}

int mp_pool_trim(mp_ctx* ctx) {
  REQUIRE(ctx, "mp_pool_trim: null context");
  CTX_ENTER(ctx);
  (void)hipStreamSynchronize(ctx->compute);
  (void)hipStreamSynchronize(ctx->copy);
  for (auto& kv : ctx->free_by_size)
    for (void* p : kv.second) {
      (void)hipFree(p);
      ctx->live.erase(p);
    }
  ctx->free_by_size.clear();
  return MP_OK;
}

// page-locked host memory: DMA reads / writes it directly, so the host-buffer entry points overlap upload, kernel
// and download on such buffers (pageable buffers are staged by the runtime and serialise).  The buffers are portable
// (usable from every device) and outlive the context that allocated them: a caller may still hold views of one
// when it destroys its context, so only mp_host_free releases them.
namespace {
std::mutex g_pinned_mu;
std::map<void*, size_t> g_pinned;
}  // namespace
int mp_host_alloc(mp_ctx* ctx, size_t bytes, void** h_ptr) {
  REQUIRE(ctx && h_ptr, "mp_host_alloc: null argument");
  *h_ptr = nullptr;
  CTX_ENTER(ctx);
  void* p = nullptr;
  HIP_TRY(hipHostMalloc(&p, bytes ? bytes : 16, hipHostMallocPortable));
  {
    std::lock_guard<std::mutex> lk(g_pinned_mu);
    g_pinned[p] = bytes;
  }
  *h_ptr = p;
  return MP_OK;
}
int mp_host_free(mp_ctx* ctx, void* h_ptr) {
  (void)ctx;  // may be null or already destroyed: the buffer does not belong to a context
  if (!h_ptr) return MP_OK;
  {
    std::lock_guard<std::mutex> lk(g_pinned_mu);
    auto it = g_pinned.find(h_ptr);
    REQUIRE(it != g_pinned.end(), "mp_host_free: pointer %p was not allocated by mp_host_alloc", h_ptr);
    g_pinned.erase(it);
  }
  HIP_TRY(hipHostFree(h_ptr));
  return MP_OK;
}

int mp_memcpy_h2d(mp_ctx* ctx, void* d_dst, const void* h_src, size_t bytes) {
  REQUIRE(ctx && (bytes == 0 || (d_dst && h_src)), "mp_memcpy_h2d: null argument");
  if (bytes == 0) return MP_OK;
  CTX_ENTER(ctx);
  HIP_TRY(hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, ctx->compute));
  HIP_TRY(hipStreamSynchronize(ctx->compute));
  return MP_OK;
}
int mp_memcpy_d2h(mp_ctx* ctx, void* h_dst, const void* d_src, size_t bytes) {
  REQUIRE(ctx && (bytes == 0 || (h_dst && d_src)), "mp_memcpy_d2h: null argument");
  if (bytes == 0) return MP_OK;
  CTX_ENTER(ctx);
  HIP_TRY(hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, ctx->compute));
  HIP_TRY(hipStreamSynchronize(ctx->compute));
  return MP_OK;
}
int mp_memset(mp_ctx* ctx, void* d_dst, int value, size_t bytes) {
  REQUIRE(ctx && (bytes == 0 || d_dst), "mp_memset: null argument");
  if (bytes == 0) return MP_OK;
  CTX_ENTER(ctx);
  HIP_TRY(hipMemsetAsync(d_dst, value, bytes, ctx->compute));
  return MP_OK;
}

// ----------------------------------------------------------------------------------------- events
int mp_event_create(mp_ctx* ctx, mp_event** out) {
  REQUIRE(ctx && out, "mp_event_create: null argument");
  CTX_ENTER(ctx);
  mp_event* e = new (std::nothrow) mp_event;
  REQUIRE(e, "mp_event_create: out of host memory");
  e->device = ctx->device;
  hipError_t he = hipEventCreate(&e->ev);
  if (he != hipSuccess) { delete e; return hip_err(he, "hipEventCreate"); }
  *out = e;
  return MP_OK;
}
int mp_event_destroy(mp_event* ev) {
  if (!ev) return MP_OK;
  (void)hipSetDevice(ev->device);
  (void)hipEventDestroy(ev->ev);
  delete ev;
  return MP_OK;
}
int mp_event_record(mp_ctx* ctx, mp_event* ev) {
  REQUIRE(ctx && ev, "mp_event_record: null argument");
  CTX_ENTER(ctx);
  HIP_TRY(hipEventRecord(ev->ev, ctx->compute));
  return MP_OK;
}
int mp_event_elapsed_ms(mp_event* start, mp_event* stop, float* ms) {
  REQUIRE(start && stop && ms, "mp_event_elapsed_ms: null argument");
  HIP_TRY(hipSetDevice(stop->device));
  HIP_TRY(hipEventSynchronize(stop->ev));
  HIP_TRY(hipEventElapsedTime(ms, start->ev, stop->ev));
  return MP_OK;
}

// ----------------------------------------------------------------------------------------- graphs
int mp_graph_begin(mp_ctx* ctx) {
  REQUIRE(ctx, "mp_graph_begin: null context");
  CTX_ENTER(ctx);
  REQUIRE(!ctx->capturing, "mp_graph_begin: a capture is already open on this context");
  // (CTX_ENTER has run the context's own parked passes on the stream.)  Captured float32 inverse-dynamics launches park their
  // float64 passes in a pool of the graph's own: lists and counters live as long as the graph, the passes become its nodes
  mp_ctx::HardPool* pool = new (std::nothrow) mp_ctx::HardPool;
  REQUIRE(pool, "mp_graph_begin: out of host memory");
  hipError_t he = hipStreamBeginCapture(ctx->compute, hipStreamCaptureModeThreadLocal);
  if (he != hipSuccess) { delete pool; return hip_err(he, "hipStreamBeginCapture"); }
  ctx->hp = pool;
  ctx->capturing = true;
  return MP_OK;
}
int mp_graph_end(mp_ctx* ctx, mp_graph** out) {
  REQUIRE(ctx && out, "mp_graph_end: null argument");
  *out = nullptr;
  CTX_ENTER(ctx);
  REQUIRE(ctx->capturing, "mp_graph_end: no capture is open on this context");
  // (CTX_ENTER has captured the passes still parked in the capture's pool: they are the graph's last nodes)
  mp_ctx::HardPool* pool = ctx->hp;
  ctx->hp = &ctx->own_pool;
  ctx->capturing = false;
  auto drop_pool = [&] { (void)hipStreamSynchronize(ctx->compute); free_hard_pool(pool); delete pool; };
  hipGraph_t g = nullptr;
  hipError_t he = hipStreamEndCapture(ctx->compute, &g);
  if (he != hipSuccess) { drop_pool(); return hip_err(he, "hipStreamEndCapture"); }
  if (!g) drop_pool();
  REQUIRE(g, "mp_graph_end: the capture was invalidated (a non-capturable call ran between begin and end)");
  hipGraphExec_t ex = nullptr;
  he = hipGraphInstantiate(&ex, g, nullptr, nullptr, 0);
  if (he != hipSuccess) { (void)hipGraphDestroy(g); drop_pool(); return hip_err(he, "hipGraphInstantiate"); }
  mp_graph* gr = new (std::nothrow) mp_graph;
  if (!gr) { (void)hipGraphExecDestroy(ex); (void)hipGraphDestroy(g); drop_pool(); }
  REQUIRE(gr, "mp_graph_end: out of host memory");
  gr->graph = g; gr->exec = ex; gr->device = ctx->device; gr->ctx = ctx; gr->ctx_uid = ctx->uid; gr->pool = pool;
  ++ctx->live_graphs;
  *out = gr;
  return MP_OK;
}
int mp_graph_launch(mp_ctx* ctx, mp_graph* graph) {
  REQUIRE(ctx && graph, "mp_graph_launch: null argument");
  REQUIRE(graph->device == ctx->device, "mp_graph_launch: graph captured on device %d, context is on %d", graph->device, ctx->device);
  CTX_ENTER(ctx);
  HIP_TRY(hipGraphLaunch(graph->exec, ctx->compute));
  return MP_OK;
}
// what destroyed models left behind for the sake of live graphs (ctx->mu held, device bound, no capture open)
static void release_retired(mp_ctx* ctx) {
  (void)hipStreamSynchronize(ctx->compute);
  for (hipModule_t m : ctx->retired_mods) (void)hipModuleUnload(m);
  ctx->retired_mods.clear();
  for (void* p : ctx->retired_bufs) (void)mp_free(ctx, p);
  ctx->retired_bufs.clear();
}

int mp_graph_destroy(mp_graph* graph) {
  if (!graph) return MP_OK;
  int prev = -1;
  (void)hipGetDevice(&prev);
  (void)hipSetDevice(graph->device);
  (void)hipDeviceSynchronize();  // a replay still running writes the graph's row lists
  (void)hipGraphExecDestroy(graph->exec);
  (void)hipGraphDestroy(graph->graph);
  if (graph->pool) { free_hard_pool(graph->pool); delete graph->pool; }
  {
    std::lock_guard<std::mutex> lk(g_ctxs_mu);
    // the context may have been destroyed first (it took the retired objects with it) - and another one created at its address
    if (graph->ctx && g_ctxs.count(graph->ctx) && graph->ctx->uid == graph->ctx_uid) {
      std::lock_guard<std::recursive_mutex> cl(graph->ctx->mu);
      if (--graph->ctx->live_graphs <= 0 && !graph->ctx->capturing) {
        graph->ctx->live_graphs = 0;
        release_retired(graph->ctx);
      }
    }
  }
  delete graph;
  if (prev >= 0) (void)hipSetDevice(prev);
  return MP_OK;
}

// ------------------------------------------------------------------------------------------ model
int mp_model_create(int n, const double* S, const double* Mcom, const double* G, const double* M_ee,
                    const double* joint_limits, const double* torque_limits, mp_model** out) {
  REQUIRE(out, "mp_model_create: null output");
  *out = nullptr;
  REQUIRE(S && Mcom && G && M_ee, "mp_model_create: S, Mcom, G and M_ee are required");
  REQUIRE(n >= 1 && n <= MP_BIG_DOF, "mp_model_create: dof %d outside 1..%d", n, MP_BIG_DOF);
  mp_model* m = new (std::nothrow) mp_model;
  REQUIRE(m, "mp_model_create: out of host memory");
  char msg[400] = "";
  int rc;
  // MANIPULAPY_HIP_LOOPED=1 (tests): build every model for the run-time-n kernels, so that the looped recursions can be
  // checked against the unrolled ones and the goldens on the 6..8-joint robots too
  const char* looped = getenv("MANIPULAPY_HIP_LOOPED");
  if (n <= MP_MAX_DOF && !(looped && looped[0] == '1')) {
    rc = mp_compile_model(n, S, Mcom, G, M_ee, joint_limits, torque_limits, &m->d, msg, sizeof msg);
    if (!rc) mp_model_cast(m->d, &m->f);
  } else {  // 9..32 joints: the looped kernels' model; d / f only carry the joint count
    m->big = true;
    rc = mp_compile_model_big(n, S, Mcom, G, M_ee, joint_limits, torque_limits, &m->bd, msg, sizeof msg);
    if (!rc) {
      mp_model_cast(m->bd, &m->bf);
      std::memset(&m->d, 0, sizeof m->d);
      std::memset(&m->f, 0, sizeof m->f);
      m->d.n = n;
      m->f.n = n;
    }
  }
  if (rc) {
    delete m;
    return set_err(MP_ERR_MODEL, "mp_model_create: %s", msg);
  }
  static std::atomic<uint64_t> next_uid{1};
  m->uid = next_uid.fetch_add(1);
  *out = m;
  return MP_OK;
}
int mp_model_destroy(mp_model* model) {
  if (!model) return MP_OK;
  // Drop what the live contexts hold for this model: its specialised code objects and its device-resident copies.  A launch
  // graph captured on a context keeps kernel nodes of those code objects and the addresses of those copies (the handles
  // hold no reference to the models they captured), and during an open capture neither a stream synchronisation nor a
  // buffer release is legal - so with a live graph or an open capture they are RETIRED (released with the last graph, or
  // with the context) instead of released here.  Called from Python's garbage collector at arbitrary times: the calling
  // thread's current device is restored.
  int prev = -1;
  {
    std::lock_guard<std::mutex> lk(g_ctxs_mu);
    if (!g_ctxs.empty()) (void)hipGetDevice(&prev);  // (no context, no HIP call: CPU-launcher processes never touch the runtime)
    for (mp_ctx* ctx : g_ctxs) {
      std::lock_guard<std::recursive_mutex> cl(ctx->mu);
      auto sp = ctx->specs.find(model->uid);
      auto dm = ctx->dev_models.find(model->uid);
      const bool has_big = ctx->dev_big[0].count(model->uid) || ctx->dev_big[1].count(model->uid);
      if (sp == ctx->specs.end() && dm == ctx->dev_models.end() && !has_big) continue;
      const bool retire = ctx->capturing || ctx->live_graphs > 0;
      if (!ctx->capturing) {  // parked float64 passes may belong to this model's code object
        (void)hipSetDevice(ctx->device);
        (void)hard_flush(ctx);
      }
      if (!retire) {
        (void)hipSetDevice(ctx->device);
        (void)hipStreamSynchronize(ctx->compute);  // a launch of this model's kernels may still be in flight
      }
      if (sp != ctx->specs.end()) {
        for (hipModule_t m : {sp->second.mod_ilp, sp->second.mod}) {
          if (!m) continue;
          if (retire) ctx->retired_mods.push_back(m);
          else (void)hipModuleUnload(m);
        }
        ctx->specs.erase(sp);
      }
      auto drop = [&](void* p) {
        if (retire) ctx->retired_bufs.push_back(p);
        else (void)mp_free(ctx, p);
      };
      if (dm != ctx->dev_models.end()) { drop(dm->second); ctx->dev_models.erase(dm); }
      for (auto& table : ctx->dev_big) {
        auto it = table.find(model->uid);
        if (it != table.end()) { drop(it->second); table.erase(it); }
      }
    }
  }
  delete model;
  if (prev >= 0) (void)hipSetDevice(prev);
  return MP_OK;
}
int mp_model_dof(const mp_model* model, int* n) {
  REQUIRE(model && n, "mp_model_dof: null argument");
  *n = model->d.n;
  return MP_OK;
}
int mp_model_params(const mp_model* model, double* out) {
  REQUIRE(model && out, "mp_model_params: null argument");
  static_assert(sizeof(MpJoint<double>) == MP_JOINT_FIELDS * sizeof(double), "MpJoint layout");  // the first 16 fields are the documented ones
  for (int i = 0; i < model->d.n; ++i) std::memcpy(out + 16 * i, model->big ? &model->bd.j[i] : &model->d.j[i], 16 * sizeof(double));
  return MP_OK;
}
int mp_model_specialize_source(const mp_model* model, char* buf, size_t* len) {
  REQUIRE(model && len, "mp_model_specialize_source: null argument");
  REQUIRE_SMALL("mp_model_specialize_source");
  // both translation units, the second behind a separator line (they are compiled separately: csrc/mp_jit.cpp)
  const std::string src = mp_jit_source(model->f, model->d) + "\n// ==== second program (max-ILP scheduling strategy) ====\n" + mp_jit_source(model->f, model->d, 1);
  if (buf) {
    REQUIRE(*len >= src.size() + 1, "mp_model_specialize_source: buffer too small");
    std::memcpy(buf, src.c_str(), src.size() + 1);
  }
  *len = src.size() + 1;
  return MP_OK;
}
int mp_model_specialize_compile(const mp_model* model, size_t* code_bytes, int* from_cache) {
  REQUIRE(model, "mp_model_specialize_compile: null model");
  REQUIRE_SMALL("mp_model_specialize_compile");
  std::vector<char> code;
  std::string err;
  bool cached = false;
  if (mp_jit_compile(model->f, model->d, &code, &cached, &err)) return set_err(MP_ERR_UNSUPPORTED, "mp_model_specialize_compile: %s", err.c_str());
  {  // the second program (the one-row float32 inverse dynamics under the max-ILP scheduling strategy): both are needed
    std::vector<char> code2;
    bool cached2 = false;
    if (mp_jit_compile(model->f, model->d, &code2, &cached2, &err, 1)) return set_err(MP_ERR_UNSUPPORTED, "mp_model_specialize_compile: %s", err.c_str());
    code.insert(code.end(), code2.begin(), code2.end());   // (the size reported is both programs')
    cached = cached && cached2;
  }
  if (code_bytes) *code_bytes = code.size();
  if (from_cache) *from_cache = cached ? 1 : 0;
  return MP_OK;
}
int mp_model_is_specialized(mp_ctx* ctx, const mp_model* model, int* yes) {
  REQUIRE(ctx && model && yes, "mp_model_is_specialized: null argument");
  std::lock_guard<std::recursive_mutex> lk(ctx->mu);
  *yes = ctx->specs.count(model->uid) ? 1 : 0;
  return MP_OK;
}
int mp_model_specialize(mp_ctx* ctx, const mp_model* model) {
  REQUIRE(ctx && model, "mp_model_specialize: null argument");
  REQUIRE_SMALL("mp_model_specialize");
  CTX_ENTER(ctx);
  if (ctx->specs.count(model->uid)) return MP_OK;
  std::vector<char> code;
  std::string err;
  if (mp_jit_compile(model->f, model->d, &code, nullptr, &err)) return set_err(MP_ERR_UNSUPPORTED, "mp_model_specialize: %s", err.c_str());
  MpSpec sp;
  HIP_TRY(hipModuleLoadData(&sp.mod, code.data()));
  // every kernel of the two programs is required: a program that lacks one is refused and the generic kernels serve
  const char* names[9][2] = {{"mp_spec_traj_id_pk_f0", "mp_spec_traj_id_pk_f1"}, {"mp_spec_fd_traj_f0", "mp_spec_fd_traj_f1"},
                            {"mp_spec_id_d_f0", "mp_spec_id_d_f1"}, {"mp_spec_fk_jac_id_d_f0", "mp_spec_fk_jac_id_d_f1"},
                            {"mp_spec_fd_s_f0", "mp_spec_fd_s_f1"}, {"mp_spec_fd_d_f0", "mp_spec_fd_d_f1"},
                            {"mp_spec_fd_traj_tm_f0", "mp_spec_fd_traj_tm_f1"}, {"mp_spec_id_hard_f0", "mp_spec_id_hard_f1"},
                            {"mp_spec_traj_id_hard_f0", "mp_spec_traj_id_hard_f1"}};
  hipFunction_t* slots[9] = {sp.traj_id_pk, sp.fd_traj, sp.id_d, sp.fk_jac_id_d, sp.fd_s, sp.fd_d, sp.fd_traj_tm, sp.id_hard, sp.traj_id_hard};
  for (int k = 0; k < 9; ++k)
    for (int f = 0; f < 2; ++f) {
      hipError_t e = hipModuleGetFunction(&slots[k][f], sp.mod, names[k][f]);
      if (e != hipSuccess) { (void)hipModuleUnload(sp.mod); return hip_err(e, names[k][f]); }
    }
  {
    hipError_t e = hipModuleGetFunction(&sp.ik, sp.mod, "mp_spec_ik");
    if (e != hipSuccess) { (void)hipModuleUnload(sp.mod); return hip_err(e, "mp_spec_ik"); }
  }
  // The second program: the float32 inverse dynamics, one row per lane - mp_spec_id_co (whole waves, rows as whole lines) and
  // mp_spec_id_s (the last < 64 rows) - compiled with LLVM's max-ILP scheduling strategy (c2 0.0704 -> 0.0695 ms, c4 0.1417 -> 0.1411;
  // as a flag for the whole program it costs the float64 and fused kernels as much as it gives: profiles/r02_d_c5_experiments.txt)
  {
    std::vector<char> code2;
    hipModule_t m2 = nullptr;
    if (mp_jit_compile(model->f, model->d, &code2, nullptr, &err, 1)) {
      (void)hipModuleUnload(sp.mod);
      return set_err(MP_ERR_UNSUPPORTED, "mp_model_specialize: %s", err.c_str());
    }
    hipError_t e = hipModuleLoadData(&m2, code2.data());
    if (e != hipSuccess) { (void)hipModuleUnload(sp.mod); return hip_err(e, "hipModuleLoadData (second program)"); }
    const char* names2[2][2] = {{"mp_spec_id_s_f0", "mp_spec_id_s_f1"}, {"mp_spec_id_co_f0", "mp_spec_id_co_f1"}};
    hipFunction_t* slots2[2] = {sp.id_s, sp.id_co};
    for (int k = 0; k < 2; ++k)
      for (int f = 0; f < 2; ++f) {
        e = hipModuleGetFunction(&slots2[k][f], m2, names2[k][f]);
        if (e != hipSuccess) { (void)hipModuleUnload(m2); (void)hipModuleUnload(sp.mod); return hip_err(e, names2[k][f]); }
      }
    sp.mod_ilp = m2;
  }
  // Self-check of THIS code object's non-finite guard.  The specialised kernels are built with -ffinite-math-only; their guard
  // works on bit patterns (csrc/mp_core.h) and today's compiler leaves it alone, but that flag would let a future one fold it.
  // A row with a NaN and a row with an infinity must come back NaN and their neighbours finite - from the per-lane float32 kernel,
  // the whole-line one (whose finite rows must carry the per-lane kernel's bits) and the float64 kernel - or the code object is
  // refused and the generic kernels (built without the flag) serve.
  {
    const int n = model->d.n;
    const long rows = 128;
    std::vector<float> h((size_t)rows * n, 0.25f), out((size_t)rows * n * 2, 0.0f);
    std::vector<double> hd((size_t)rows * n, 0.25), outd((size_t)rows * n, 0.0);
    h[3 * n] = __builtin_nanf(""); h[(size_t)70 * n + (n > 1 ? 1 : 0)] = __builtin_inff();
    hd[3 * n] = __builtin_nan(""); hd[(size_t)70 * n + (n > 1 ? 1 : 0)] = -__builtin_inf();
    const size_t fb = h.size() * sizeof(float), db = hd.size() * sizeof(double);
    Scratch sc(ctx);
    void *dq, *dz, *d0, *d1, *dqd, *dzd, *d2;
    auto unload = [&] { (void)hipModuleUnload(sp.mod_ilp); (void)hipModuleUnload(sp.mod); };
    {  // an allocation that fails (out of memory, an open graph capture) must not leak the two code objects loaded above
      void** slot[7] = {&dq, &dz, &d0, &d1, &dqd, &dzd, &d2};
      for (int k = 0; k < 7; ++k)
        if (int rc = sc.get(k < 4 ? fb : db, slot[k])) { unload(); return rc; }
    }
    hipError_t he = hipMemcpyAsync(dq, h.data(), fb, hipMemcpyHostToDevice, ctx->compute);
    if (he == hipSuccess) he = hipMemsetAsync(dz, 0, fb, ctx->compute);
    if (he == hipSuccess) he = hipMemcpyAsync(dqd, hd.data(), db, hipMemcpyHostToDevice, ctx->compute);
    if (he == hipSuccess) he = hipMemsetAsync(dzd, 0, db, ctx->compute);
    if (he != hipSuccess) { unload(); return hip_err(he, "mp_model_specialize: self-check upload"); }
    MpCall<float> cf;
    MpCall<double> cd;
    make_call<float>(model, nullptr, nullptr, &cf);
    make_call<double>(model, nullptr, nullptr, &cd);
    long nr = rows;
    const float *q = (const float*)dq, *z = (const float*)dz;
    float *o0 = (float*)d0, *o1 = (float*)d1;
    const double *qd_ = (const double*)dqd, *zd = (const double*)dzd;
    double* o2 = (double*)d2;
    MpLead no_lead;
    std::memset(&no_lead, 0, sizeof no_lead);
    void* a0[] = {&cf, &q, &z, &z, &o0, &nr};
    void* a1[] = {&cf, &q, &z, &z, &o1, &nr, &no_lead};
    void* a2[] = {&cd, &qd_, &zd, &zd, &o2, &nr};
    int rc = launch_spec(ctx, sp.id_s[0], rows, a0);
    if (!rc) rc = launch_spec(ctx, sp.id_co[0], rows, a1, MP_JIT_ID_CO_BLOCK);
    if (!rc) rc = launch_spec(ctx, sp.id_d[0], rows, a2);
    if (rc) { unload(); return rc; }
    he = hipMemcpyAsync(out.data(), d0, fb, hipMemcpyDeviceToHost, ctx->compute);
    if (he == hipSuccess) he = hipMemcpyAsync(out.data() + h.size(), d1, fb, hipMemcpyDeviceToHost, ctx->compute);
    if (he == hipSuccess) he = hipMemcpyAsync(outd.data(), d2, db, hipMemcpyDeviceToHost, ctx->compute);
    if (he == hipSuccess) he = hipStreamSynchronize(ctx->compute);
    if (he != hipSuccess) { unload(); return hip_err(he, "mp_model_specialize: self-check download"); }
    bool ok = true;
    for (int k = 0; k < 3 && ok; ++k) {  // 0: one row per lane, 1: whole-line row movement, 2: float64
      for (long r : {3L, 70L, 4L, 69L, 0L, 127L}) {
        const bool want_nan = r == 3 || r == 70;
        for (int j = 0; j < n; ++j) {
          const double v = k == 2 ? outd[(size_t)r * n + j] : (double)out[(size_t)k * h.size() + (size_t)r * n + j];
          if ((v != v) != want_nan) ok = false;
          if (k == 1 && !want_nan && out[h.size() + (size_t)r * n + j] != out[(size_t)r * n + j]) ok = false;  // = the per-lane kernel's bits
        }
      }
    }
    if (!ok) {
      unload();
      return set_err(MP_ERR_UNSUPPORTED, "mp_model_specialize: the specialised kernels of this toolchain do not keep the non-finite row "
                                         "contract (self-check failed); the generic kernels are used instead");
    }
  }
  ctx->specs[model->uid] = sp;
  return MP_OK;
}
int mp_model_blob(const mp_model* model, int use_f64, void* out, size_t* bytes) {
  REQUIRE(model && bytes, "mp_model_blob: null argument");
  const size_t need = model->big ? (use_f64 ? sizeof(MpBigModel<double>) : sizeof(MpBigModel<float>))
                                 : (use_f64 ? sizeof(MpModel<double>) : sizeof(MpModel<float>));
  if (out) {
    REQUIRE(*bytes >= need, "mp_model_blob: buffer of %zu bytes, need %zu", *bytes, need);
    const void* src = model->big ? (use_f64 ? (const void*)&model->bd : (const void*)&model->bf)
                                 : (use_f64 ? (const void*)&model->d : (const void*)&model->f);
    std::memcpy(out, src, need);
  }
  *bytes = need;
  return MP_OK;
}
int mp_model_fk_host(const mp_model* model, const double* q, double* T) {
  REQUIRE(model && q && T, "mp_model_fk_host: null argument");
  if (model->big) mp_compiled_fk(model->bd, q, T);
  else mp_compiled_fk(model->d, q, T);
  return MP_OK;
}

// ---------------------------------------------------------------------- hot path, device pointers
#define CHECK_COMMON(fn)                                              \
  REQUIRE(ctx && model, fn ": null context or model");                \
  CTX_ENTER(ctx);

int mp_batch_trajectory_f32(mp_ctx* ctx, const mp_model* model, const float* d_start, const float* d_end, int64_t B,
                            int64_t N, double Tf, int method, float* d_pos, float* d_vel, float* d_acc) {
  CHECK_COMMON("mp_batch_trajectory_f32");
  REQUIRE(B >= 0 && N >= 0, "mp_batch_trajectory_f32: negative B (%lld) or N (%lld)", (long long)B, (long long)N);
  if (B == 0 || N == 0) return MP_OK;
  REQUIRE(d_start && d_end && d_pos && d_vel && d_acc, "mp_batch_trajectory_f32: null device pointer");
  REQUIRE(aligned16(d_start) && aligned16(d_end) && aligned16(d_pos) && aligned16(d_vel) && aligned16(d_acc),
          "mp_batch_trajectory_f32: device pointers must be 16-byte aligned");
  PROFILE_SCOPE(ctx, "mp_batch_trajectory_f32");
  if (model->big) {
    const MpBigModel<float>* dm = nullptr;
    if (int rc = device_big_model<float>(ctx, model, &dm)) return rc;
    MpCall<float> c0 = {};
    HIP_TRY(mpk_dyn_traj(ctx->compute, model->d.n, dm, c0, false, d_start, d_end, (long)B, (long)N, Tf, method, d_pos, d_vel, d_acc, nullptr));
    return MP_OK;
  }
  HIP_TRY(mpk_batch_traj(ctx->compute, model->f, d_start, d_end, (long)B, (long)N, Tf, method, d_pos, d_vel, d_acc));
  return MP_OK;
}

int mp_id_trajectory_f32(mp_ctx* ctx, const mp_model* model, const float* d_q, const float* d_qd, const float* d_qdd,
                         int64_t rows, const double* g, const double* Ftip, float* d_tau) {
  return id_impl<float>("mp_id_trajectory_f32", ctx, model, d_q, d_qd, d_qdd, rows, g, Ftip, d_tau);
}
int mp_id_trajectory_f64(mp_ctx* ctx, const mp_model* model, const double* d_q, const double* d_qd,
                         const double* d_qdd, int64_t rows, const double* g, const double* Ftip, double* d_tau) {
  return id_impl<double>("mp_id_trajectory_f64", ctx, model, d_q, d_qd, d_qdd, rows, g, Ftip, d_tau);
}

int mp_traj_id_fused_f32(mp_ctx* ctx, const mp_model* model, const float* d_start, const float* d_end, int64_t B,
                         int64_t N, double Tf, int method, const double* g, const double* Ftip, float* d_tau) {
  REQUIRE(ctx && model, "mp_traj_id_fused_f32: null context or model");
  CTX_ENTER_NOJOIN(ctx);
  {  // parked float64 passes run first only where they touch this launch's arrays (as in id_impl): its own pass is parked too
    const size_t in_b = (size_t)(B > 0 ? B : 0) * (size_t)model->d.n * sizeof(float), out_b = in_b * (size_t)(N > 0 ? N : 0);
    const void* lo[3] = {d_start, d_end, d_tau};
    const size_t by[3] = {in_b, in_b, out_b};
    if (int rc = ctx->profiling ? hard_flush(ctx) : hard_flush_if_overlapping(ctx, lo, by, 3)) return rc;
  }
  REQUIRE(B >= 0 && N >= 0, "mp_traj_id_fused_f32: negative B (%lld) or N (%lld)", (long long)B, (long long)N);
  if (B == 0 || N == 0) return MP_OK;
  REQUIRE(d_start && d_end && d_tau, "mp_traj_id_fused_f32: null device pointer");
  REQUIRE(aligned16(d_start) && aligned16(d_end) && aligned16(d_tau),
          "mp_traj_id_fused_f32: device pointers must be 16-byte aligned");
  MpCall<float> c;
  make_call_f32(ctx, model, g, Ftip, &c);
  const bool ftip = any_nonzero(Ftip);
  PROFILE_SCOPE(ctx, "mp_traj_id_fused_f32");
  if (model->big) {
    const MpBigModel<float>* dm = nullptr;
    if (int rc = device_big_model<float>(ctx, model, &dm)) return rc;
    if (int rc = big_cold_model(ctx, model, &c)) return rc;
    HIP_TRY(mpk_dyn_traj(ctx->compute, model->d.n, dm, c, ftip, d_start, d_end, (long)B, (long)N, Tf, method, nullptr, nullptr, nullptr, d_tau));
    return MP_OK;
  }
  // per-timestep table of (s, s', s''): rebuilt on the stream only when (N, Tf, method) differ from the last call
  if (ctx->tab_cap < (long)N) {
    REQUIRE(!ctx->capturing, "mp_traj_id_fused_f32: the first call for this N allocates; run it once before capturing a launch graph");
    if (int rc = hard_flush(ctx)) return rc;      // parked passes of fused launches read the old table
    HIP_TRY(hipStreamSynchronize(ctx->compute));  // earlier kernels may still read the old table
    if (ctx->time_tab && ctx->tab_volatile) ctx->retired_tabs.push_back(ctx->time_tab);  // a launch graph still points at it
    else if (ctx->time_tab) (void)hipFree(ctx->time_tab);
    ctx->time_tab = nullptr; ctx->tab_cap = 0; ctx->tab_Nt = -1;
    HIP_TRY(hipMalloc((void**)&ctx->time_tab, (size_t)N * 3 * sizeof(double)));
    ctx->tab_cap = (long)N;
  }
  // A captured call RECORDS its table kernel (the graph is self-contained), and every later replay rewrites the shared
  // table behind the cache's back - so once a capture has happened on this context the cache is off for good: every
  // call then writes its own table first (one extra ~2 us kernel, stream-ordered before the kernel that reads it).
  if (ctx->capturing) ctx->tab_volatile = true;
  if (ctx->tab_volatile || ctx->tab_Nt != (long)N || ctx->tab_Tf != Tf || ctx->tab_method != method) {
    if (int rc = hard_flush(ctx)) return rc;      // (as above)
    HIP_TRY(mpk_time_table(ctx->compute, ctx->time_tab, (long)N, Tf, method));
    ctx->tab_Nt = (long)N; ctx->tab_Tf = Tf; ctx->tab_method = method;
  }
  if (const MpSpec* sp = find_spec(ctx, model)) {
    long nt = (long)N;
    const double* tab = ctx->time_tab;
    unsigned bpt = mpk_traj_blocks_per_trajectory(nt);
    const long rows_l = (long)B * (long)N;
    mp_ctx::HardSlot* hs = attach_hard_list(ctx, rows_l, &c);
    // (as launch_id: one parked pass of an earlier fused launch of this program rides with this launch's first workgroups)
    MpLead lead;
    std::memset(&lead, 0, sizeof lead);
    mp_ctx::HardSlot* rider = !ctx->tab_volatile ? pick_rider(ctx, sp->traj_id_hard[ftip ? 1 : 0], hs, true) : nullptr;
    if (rider) {
      lead.C = rider->C; lead.q = rider->q; lead.qd = rider->qd; lead.qdd = rider->qdd; lead.tau = rider->tau; lead.rows = rider->nrows;
      lead.nt = rider->nt;
      lead.blocks = std::min((hard_pass_blocks((long)rider->nrows) + 3u) / 4u, 128u);   // 256-lane workgroups
    }
    void* args[] = {&c, &d_start, &d_end, &nt, &bpt, &tab, &d_tau, &lead};
    {
      const unsigned grid = (unsigned)((long)B * bpt) + lead.blocks;
      HIP_TRY(hipModuleLaunchKernel(sp->traj_id_pk[ftip ? 1 : 0], grid, 1, 1, 256, 1, 1, 0, ctx->compute, args, nullptr));
    }
    if (rider) { rider->busy = false; rider->orphan = false; }
    if (!hs) return MP_OK;
    // the generated rows' float64 pass is parked like a given-rows launch's; it reads the time table, so a call that rewrites the
    // table runs the parked passes first (above)
    hard_defer_generated(ctx, hs, sp->traj_id_hard[ftip ? 1 : 0], c, d_start, d_end, tab, d_tau, rows_l, model->d.n, (long)B, (unsigned)N);
    if (ctx->tab_volatile) return hard_flush(ctx);  // (a table rewritten by every call cannot wait for a parked pass)
    return hard_park_or_run(ctx, d_start, d_end, nullptr, d_tau, (size_t)rows_l * (size_t)model->d.n * sizeof(float), (size_t)B * (size_t)model->d.n * sizeof(float));
  }
  {
    // generic kernels: the same hand-over (the float64 model and the float32 limits come from the device copy)
    const long rows_l = (long)B * (long)N;
    mp_ctx::HardSlot* hs = nullptr;
    const MpModel<float>* dm = nullptr;
    if (c.cold_model && device_model(ctx, model, &dm) == MP_OK) hs = attach_hard_list(ctx, rows_l, &c);
    HIP_TRY(mpk_traj_id_tab(ctx->compute, model->f, c, ftip, d_start, d_end, (long)B, (long)N, ctx->time_tab, d_tau));
    if (hs) {
      HIP_TRY(mpk_traj_id_hard(ctx->compute, dm, model->d.n, c, ftip, d_start, d_end, (unsigned)N, ctx->time_tab, d_tau, (unsigned)rows_l,
                               hard_pass_blocks(rows_l)));
      (void)hard_passed(hs, MP_OK);
    }
  }
  return MP_OK;
}

int mp_fk_jac_id_f64(mp_ctx* ctx, const mp_model* model, const double* d_q, const double* d_qd, const double* d_qdd,
                     int64_t rows, const double* g, const double* Ftip, double* d_T, double* d_J, double* d_tau) {
  return fkjid_impl<double>("mp_fk_jac_id_f64", ctx, model, d_q, d_qd, d_qdd, rows, g, Ftip, d_T, d_J, d_tau);
}
int mp_fk_jac_id_f32(mp_ctx* ctx, const mp_model* model, const float* d_q, const float* d_qd, const float* d_qdd,
                     int64_t rows, const double* g, const double* Ftip, float* d_T, float* d_J, float* d_tau) {
  return fkjid_impl<float>("mp_fk_jac_id_f32", ctx, model, d_q, d_qd, d_qdd, rows, g, Ftip, d_T, d_J, d_tau);
}

// ------------------------------------------------------------------------ hot path, host pointers
int mp_batch_trajectory_host_f32(mp_ctx* ctx, const mp_model* model, const float* start, const float* end, int64_t B,
                                 int64_t N, double Tf, int method, float* pos, float* vel, float* acc) {
  CHECK_COMMON("mp_batch_trajectory_host_f32");
  REQUIRE(B >= 0 && N >= 0, "mp_batch_trajectory_host_f32: negative B or N");
  if (B == 0 || N == 0) return MP_OK;
  REQUIRE(start && end && pos && vel && acc, "mp_batch_trajectory_host_f32: null host pointer");
  const size_t n = (size_t)model->d.n, in_b = (size_t)B * n * sizeof(float), out_b = (size_t)B * (size_t)N * n * sizeof(float);
  Scratch sc(ctx);
  void *ds, *de, *dp, *dv, *da;
  if (int rc = sc.get(in_b, &ds)) return rc;
  if (int rc = sc.get(in_b, &de)) return rc;
  if (int rc = sc.get(out_b, &dp)) return rc;
  if (int rc = sc.get(out_b, &dv)) return rc;
  if (int rc = sc.get(out_b, &da)) return rc;
  H2D(ds, start, in_b);
  H2D(de, end, in_b);
  if (int rc = mp_batch_trajectory_f32(ctx, model, (float*)ds, (float*)de, B, N, Tf, method, (float*)dp, (float*)dv, (float*)da)) return rc;
  D2H(pos, dp, out_b);
  D2H(vel, dv, out_b);
  D2H(acc, da, out_b);
  HIP_TRY(hipStreamSynchronize(ctx->compute));
  return MP_OK;
}

int mp_id_trajectory_host_f32(mp_ctx* ctx, const mp_model* model, const float* q, const float* qd, const float* qdd,
                              int64_t rows, const double* g, const double* Ftip, float* tau) {
  return id_host_impl<float>("mp_id_trajectory_host_f32", ctx, model, q, qd, qdd, rows, g, Ftip, tau);
}
int mp_id_trajectory_host_f64(mp_ctx* ctx, const mp_model* model, const double* q, const double* qd,
                              const double* qdd, int64_t rows, const double* g, const double* Ftip, double* tau) {
  return id_host_impl<double>("mp_id_trajectory_host_f64", ctx, model, q, qd, qdd, rows, g, Ftip, tau);
}

int mp_traj_id_fused_host_f32(mp_ctx* ctx, const mp_model* model, const float* start, const float* end, int64_t B,
                              int64_t N, double Tf, int method, const double* g, const double* Ftip, float* tau) {
  CHECK_COMMON("mp_traj_id_fused_host_f32");
  REQUIRE(B >= 0 && N >= 0, "mp_traj_id_fused_host_f32: negative B or N");
  if (B == 0 || N == 0) return MP_OK;
  REQUIRE(start && end && tau, "mp_traj_id_fused_host_f32: null host pointer");
  const size_t n = (size_t)model->d.n, in_b = (size_t)B * n * sizeof(float), out_b = (size_t)B * (size_t)N * n * sizeof(float);
  Scratch sc(ctx);
  void *ds, *de, *dt;
  if (int rc = sc.get(in_b, &ds)) return rc;
  if (int rc = sc.get(in_b, &de)) return rc;
  if (int rc = sc.get(out_b, &dt)) return rc;
  H2D(ds, start, in_b);
  H2D(de, end, in_b);
  if (int rc = mp_traj_id_fused_f32(ctx, model, (float*)ds, (float*)de, B, N, Tf, method, g, Ftip, (float*)dt)) return rc;
  if (int rc = hard_flush(ctx)) return rc;  // the launch's float64 pass, parked: the download reads its rows
  D2H(tau, dt, out_b);
  HIP_TRY(hipStreamSynchronize(ctx->compute));
  return MP_OK;
}

int mp_fk_jac_id_host_f64(mp_ctx* ctx, const mp_model* model, const double* q, const double* qd, const double* qdd,
                          int64_t rows, const double* g, const double* Ftip, double* T, double* J, double* tau) {
  CHECK_COMMON("mp_fk_jac_id_host_f64");
  REQUIRE(rows >= 0, "mp_fk_jac_id_host_f64: negative row count");
  if (rows == 0) return MP_OK;
  REQUIRE(q && (T || J || tau), "mp_fk_jac_id_host_f64: q and at least one output are required");
  REQUIRE(!tau || (qd && qdd), "mp_fk_jac_id_host_f64: tau requested without qd / qdd");
  const size_t n = (size_t)model->d.n, row_b = n * sizeof(double), rb = (size_t)rows * row_b;
  Scratch sc(ctx);
  void *dq = nullptr, *dqd = nullptr, *dqdd = nullptr, *dT = nullptr, *dJ = nullptr, *dt = nullptr;
  if (int rc = sc.get(rb, &dq)) return rc;
  if (tau) {
    if (int rc = sc.get(rb, &dqd)) return rc;
    if (int rc = sc.get(rb, &dqdd)) return rc;
    if (int rc = sc.get(rb, &dt)) return rc;
  }
  if (T) if (int rc = sc.get((size_t)rows * 16 * sizeof(double), &dT)) return rc;
  if (J) if (int rc = sc.get(rb * 6, &dJ)) return rc;
  // page-locked arrays throughout and more than one chunk: upload, kernels and the (much larger) download overlap
  const int64_t chunk = host_chunk_rows();
  const bool pinned = is_pinned_host(q) && (!tau || (is_pinned_host(qd) && is_pinned_host(qdd) && is_pinned_host(tau))) &&
                      (!T || is_pinned_host(T)) && (!J || is_pinned_host(J));
  if (pinned && rows > chunk) {
    char *cq = (char*)dq, *cqd = (char*)dqd, *cqdd = (char*)dqdd, *ct = (char*)dt, *cT = (char*)dT, *cJ = (char*)dJ;
    return host_pipeline(
        ctx, rows, chunk,
        [&](int64_t r0, int64_t nr) -> int {
          const size_t off = (size_t)r0 * row_b, nb = (size_t)nr * row_b;
          UP(cq + off, (const char*)q + off, nb);
          if (tau) {
            UP(cqd + off, (const char*)qd + off, nb);
            UP(cqdd + off, (const char*)qdd + off, nb);
          }
          return MP_OK;
        },
        [&](int64_t r0, int64_t nr) -> int {
          const size_t off = (size_t)r0 * row_b;
          return mp_fk_jac_id_f64(ctx, model, (double*)(cq + off), tau ? (double*)(cqd + off) : nullptr,
                                  tau ? (double*)(cqdd + off) : nullptr, nr, g, Ftip, T ? (double*)(cT + (size_t)r0 * 128) : nullptr,
                                  J ? (double*)(cJ + off * 6) : nullptr, tau ? (double*)(ct + off) : nullptr);
        },
        [&](int64_t r0, int64_t nr) -> int {
          const size_t off = (size_t)r0 * row_b, nb = (size_t)nr * row_b;
          if (T) DOWN((char*)T + (size_t)r0 * 128, cT + (size_t)r0 * 128, (size_t)nr * 128);
          if (J) DOWN((char*)J + off * 6, cJ + off * 6, nb * 6);
          if (tau) DOWN((char*)tau + off, ct + off, nb);
          return MP_OK;
        });
  }
  H2D(dq, q, rb);
  if (tau) {
    H2D(dqd, qd, rb);
    H2D(dqdd, qdd, rb);
  }
  if (int rc = mp_fk_jac_id_f64(ctx, model, (double*)dq, (double*)dqd, (double*)dqdd, rows, g, Ftip, (double*)dT,
                                (double*)dJ, (double*)dt))
    return rc;
  if (T) D2H(T, dT, (size_t)rows * 16 * sizeof(double));
  if (J) D2H(J, dJ, rb * 6);
  if (tau) D2H(tau, dt, rb);
  HIP_TRY(hipStreamSynchronize(ctx->compute));
  return MP_OK;
}

int mp_mass_matrix_f64(mp_ctx* ctx, const mp_model* model, const double* d_q, int64_t rows, double* d_M) {
  return mm_impl<double>("mp_mass_matrix_f64", ctx, model, d_q, rows, d_M);
}
int mp_mass_matrix_f32(mp_ctx* ctx, const mp_model* model, const float* d_q, int64_t rows, float* d_M) {
  return mm_impl<float>("mp_mass_matrix_f32", ctx, model, d_q, rows, d_M);
}
int mp_forward_dynamics_f64(mp_ctx* ctx, const mp_model* model, const double* d_q, const double* d_qd, const double* d_tau,
                            int64_t rows, const double* g, const double* Ftip, double* d_qdd) {
  return fdyn_impl<double>("mp_forward_dynamics_f64", ctx, model, d_q, d_qd, d_tau, rows, g, Ftip, d_qdd);
}
int mp_forward_dynamics_f32(mp_ctx* ctx, const mp_model* model, const float* d_q, const float* d_qd, const float* d_tau,
                            int64_t rows, const double* g, const double* Ftip, float* d_qdd) {
  return fdyn_impl<float>("mp_forward_dynamics_f32", ctx, model, d_q, d_qd, d_tau, rows, g, Ftip, d_qdd);
}
int mp_fd_trajectory_f32(mp_ctx* ctx, const mp_model* model, const float* d_theta0, const float* d_dtheta0,
                         const float* d_taumat, const float* d_Ftipmat, int64_t B, int64_t N, const double* g, double dt,
                         int intRes, float* d_pos, float* d_vel, float* d_acc) {
  return fdtraj_impl<float>("mp_fd_trajectory_f32", ctx, model, d_theta0, d_dtheta0, d_taumat, d_Ftipmat, B, N, g, dt, intRes,
                            d_pos, d_vel, d_acc);
}
int mp_fd_trajectory_f64(mp_ctx* ctx, const mp_model* model, const double* d_theta0, const double* d_dtheta0,
                         const double* d_taumat, const double* d_Ftipmat, int64_t B, int64_t N, const double* g, double dt,
                         int intRes, float* d_pos, float* d_vel, float* d_acc) {
  return fdtraj_impl<double>("mp_fd_trajectory_f64", ctx, model, d_theta0, d_dtheta0, d_taumat, d_Ftipmat, B, N, g, dt, intRes,
                             d_pos, d_vel, d_acc);
}
int mp_fd_trajectory_tm_f32(mp_ctx* ctx, const mp_model* model, const float* d_theta0, const float* d_dtheta0,
                            const float* d_taumat, const float* d_Ftipmat, int64_t B, int64_t N, const double* g, double dt,
                            int intRes, float* d_pos, float* d_vel, float* d_acc) {
  return fdtraj_impl<float>("mp_fd_trajectory_tm_f32", ctx, model, d_theta0, d_dtheta0, d_taumat, d_Ftipmat, B, N, g, dt, intRes,
                            d_pos, d_vel, d_acc, true);
}
int mp_fd_trajectory_tm_f64(mp_ctx* ctx, const mp_model* model, const double* d_theta0, const double* d_dtheta0,
                            const double* d_taumat, const double* d_Ftipmat, int64_t B, int64_t N, const double* g, double dt,
                            int intRes, float* d_pos, float* d_vel, float* d_acc) {
  return fdtraj_impl<double>("mp_fd_trajectory_tm_f64", ctx, model, d_theta0, d_dtheta0, d_taumat, d_Ftipmat, B, N, g, dt, intRes,
                             d_pos, d_vel, d_acc, true);
}
int mp_transpose_rows(mp_ctx* ctx, const void* d_src, int64_t outer, int64_t inner, int64_t row_bytes, void* d_dst) {
  REQUIRE(ctx, "mp_transpose_rows: null context");
  CTX_ENTER(ctx);
  REQUIRE(outer >= 0 && inner >= 0, "mp_transpose_rows: negative extent");
  REQUIRE(row_bytes > 0 && row_bytes % 4 == 0 && row_bytes <= MP_BIG_DOF * 8, "mp_transpose_rows: row_bytes %lld must be a multiple of 4 in 4..%d",
          (long long)row_bytes, MP_BIG_DOF * 8);
  if (outer == 0 || inner == 0) return MP_OK;
  REQUIRE(d_src && d_dst && d_src != d_dst, "mp_transpose_rows: null or aliased device pointer");
  PROFILE_SCOPE(ctx, "mp_transpose_rows");
  const int W = (int)(row_bytes / 4);
  HIP_TRY(mpk_transpose_rows(ctx->compute, d_src, d_dst, (long)outer, (long)inner, W));
  return MP_OK;
}
int mp_fd_trajectory_host_f32(mp_ctx* ctx, const mp_model* model, const float* theta0, const float* dtheta0,
                              const float* taumat, const float* Ftipmat, int64_t B, int64_t N, const double* g, double dt,
                              int intRes, float* pos, float* vel, float* acc) {
  return fdtraj_host_impl<float>("mp_fd_trajectory_host_f32", ctx, model, theta0, dtheta0, taumat, Ftipmat, B, N, g, dt, intRes,
                                 pos, vel, acc);
}
int mp_fd_trajectory_host_f64(mp_ctx* ctx, const mp_model* model, const double* theta0, const double* dtheta0,
                              const double* taumat, const double* Ftipmat, int64_t B, int64_t N, const double* g, double dt,
                              int intRes, float* pos, float* vel, float* acc) {
  return fdtraj_host_impl<double>("mp_fd_trajectory_host_f64", ctx, model, theta0, dtheta0, taumat, Ftipmat, B, N, g, dt,
                                  intRes, pos, vel, acc);
}
int mp_mass_matrix_host_f64(mp_ctx* ctx, const mp_model* model, const double* q, int64_t rows, double* M) {
  CHECK_COMMON("mp_mass_matrix_host_f64");
  REQUIRE(rows >= 0, "mp_mass_matrix_host_f64: negative row count");
  if (rows == 0) return MP_OK;
  REQUIRE(q && M, "mp_mass_matrix_host_f64: null host pointer");
  const size_t n = (size_t)model->d.n, qb = (size_t)rows * n * sizeof(double), mb = qb * n;
  Scratch sc(ctx);
  void *dq, *dM;
  if (int rc = sc.get(qb, &dq)) return rc;
  if (int rc = sc.get(mb, &dM)) return rc;
  H2D(dq, q, qb);
  if (int rc = mp_mass_matrix_f64(ctx, model, (double*)dq, rows, (double*)dM)) return rc;
  D2H(M, dM, mb);
  HIP_TRY(hipStreamSynchronize(ctx->compute));
  return MP_OK;
}
int mp_forward_dynamics_host_f64(mp_ctx* ctx, const mp_model* model, const double* q, const double* qd, const double* tau,
                                 int64_t rows, const double* g, const double* Ftip, double* qdd) {
  CHECK_COMMON("mp_forward_dynamics_host_f64");
  REQUIRE(rows >= 0, "mp_forward_dynamics_host_f64: negative row count");
  if (rows == 0) return MP_OK;
  REQUIRE(q && qd && tau && qdd, "mp_forward_dynamics_host_f64: null host pointer");
  const size_t bytes = (size_t)rows * (size_t)model->d.n * sizeof(double);
  Scratch sc(ctx);
  void *dq, *dqd, *dt, *dout;
  if (int rc = sc.get(bytes, &dq)) return rc;
  if (int rc = sc.get(bytes, &dqd)) return rc;
  if (int rc = sc.get(bytes, &dt)) return rc;
  if (int rc = sc.get(bytes, &dout)) return rc;
  H2D(dq, q, bytes);
  H2D(dqd, qd, bytes);
  H2D(dt, tau, bytes);
  if (int rc = mp_forward_dynamics_f64(ctx, model, (double*)dq, (double*)dqd, (double*)dt, rows, g, Ftip, (double*)dout)) return rc;
  D2H(qdd, dout, bytes);
  HIP_TRY(hipStreamSynchronize(ctx->compute));
  return MP_OK;
}

int mp_pd_regulation_host_f64(mp_ctx* ctx, const mp_model* model, const double* theta0, const double* theta_des, const double* Kp,
                              const double* Kd, int64_t K, const double* g, double dt, int steps, double* errors, int32_t* count) {
  CHECK_COMMON("mp_pd_regulation_host_f64");
  REQUIRE(K >= 0 && steps >= 0, "mp_pd_regulation_host_f64: negative run or step count");
  if (K == 0) return MP_OK;
  REQUIRE(theta0 && theta_des && Kp && Kd && count && (errors || steps == 0), "mp_pd_regulation_host_f64: null host pointer");
  REQUIRE(std::isfinite(dt), "mp_pd_regulation_host_f64: dt must be finite");
  const size_t qb = (size_t)K * (size_t)model->d.n * sizeof(double), kb = (size_t)K * sizeof(double),
               eb = (size_t)K * (size_t)steps * sizeof(double), cb = (size_t)K * sizeof(int32_t);
  Scratch sc(ctx);
  void *d0, *dd, *dkp, *dkd, *de, *dc;
  if (int rc = sc.get(qb, &d0)) return rc;
  if (int rc = sc.get(qb, &dd)) return rc;
  if (int rc = sc.get(kb, &dkp)) return rc;
  if (int rc = sc.get(kb, &dkd)) return rc;
  if (int rc = sc.get(eb, &de)) return rc;
  if (int rc = sc.get(cb, &dc)) return rc;
  H2D(d0, theta0, qb);
  H2D(dd, theta_des, qb);
  H2D(dkp, Kp, kb);
  H2D(dkd, Kd, kb);
  if (eb) H2D(de, errors, eb);  // entries past a run's count keep the caller's values
  MpCall<double> c;
  make_call<double>(model, g, nullptr, &c);
  {
    PROFILE_SCOPE(ctx, "mp_pd_regulation_host_f64");
    if (model->big) {
      const MpBigModel<double>* dm = nullptr;
      if (int rc = device_big_model<double>(ctx, model, &dm)) return rc;
      HIP_TRY(mpk_dyn_pd_regulation(ctx->compute, model->d.n, dm, c, (double*)d0, (double*)dd, (double*)dkp, (double*)dkd, (long)K, dt, steps,
                                    (double*)de, (int*)dc));
    } else {
      HIP_TRY(mpk_pd_regulation(ctx->compute, model->d, c, (double*)d0, (double*)dd, (double*)dkp, (double*)dkd, (long)K, dt, steps,
                                (double*)de, (int*)dc));
    }
  }
  if (eb) D2H(errors, de, eb);
  D2H(count, dc, cb);
  HIP_TRY(hipStreamSynchronize(ctx->compute));
  return MP_OK;
}

int mp_cartesian_trajectory_f32(mp_ctx* ctx, const double* d_Xstart, const double* d_Xend, int64_t B, int64_t N, double Tf,
                                int method, float* d_pos, float* d_vel, float* d_acc, float* d_orient) {
  REQUIRE(ctx, "mp_cartesian_trajectory_f32: null context");
  CTX_ENTER(ctx);
  REQUIRE(B >= 0 && N >= 0, "mp_cartesian_trajectory_f32: negative B or N");
  if (B == 0 || N == 0) return MP_OK;
  REQUIRE(N >= 2, "mp_cartesian_trajectory_f32: N = 1 divides by zero (Tf / (N - 1))");
  REQUIRE(d_Xstart && d_Xend && d_pos && d_vel && d_acc && d_orient, "mp_cartesian_trajectory_f32: null device pointer");
  REQUIRE(aligned16(d_Xstart) && aligned16(d_Xend), "mp_cartesian_trajectory_f32: pose pointers must be 16-byte aligned");
  HIP_TRY(mpk_cartesian_traj(ctx->compute, d_Xstart, d_Xend, (long)B, (long)N, Tf, method, d_pos, d_vel, d_acc, d_orient));
  return MP_OK;
}
int mp_cartesian_trajectory_host_f32(mp_ctx* ctx, const double* Xstart, const double* Xend, int64_t B, int64_t N, double Tf,
                                     int method, float* pos, float* vel, float* acc, float* orient) {
  REQUIRE(ctx, "mp_cartesian_trajectory_host_f32: null context");
  CTX_ENTER(ctx);
  REQUIRE(B >= 0 && N >= 0, "mp_cartesian_trajectory_host_f32: negative B or N");
  if (B == 0 || N == 0) return MP_OK;
  REQUIRE(Xstart && Xend && pos && vel && acc && orient, "mp_cartesian_trajectory_host_f32: null host pointer");
  const size_t xb = (size_t)B * 16 * sizeof(double), pb = (size_t)B * (size_t)N * 3 * sizeof(float);
  Scratch sc(ctx);
  void *ds, *de, *dp, *dv, *da, *dor;
  if (int rc = sc.get(xb, &ds)) return rc;
  if (int rc = sc.get(xb, &de)) return rc;
  if (int rc = sc.get(pb, &dp)) return rc;
  if (int rc = sc.get(pb, &dv)) return rc;
  if (int rc = sc.get(pb, &da)) return rc;
  if (int rc = sc.get(pb * 3, &dor)) return rc;
  H2D(ds, Xstart, xb);
  H2D(de, Xend, xb);
  if (int rc = mp_cartesian_trajectory_f32(ctx, (double*)ds, (double*)de, B, N, Tf, method, (float*)dp, (float*)dv, (float*)da, (float*)dor)) return rc;
  D2H(pos, dp, pb);
  D2H(vel, dv, pb);
  D2H(acc, da, pb);
  D2H(orient, dor, pb * 3);
  HIP_TRY(hipStreamSynchronize(ctx->compute));
  return MP_OK;
}

int mp_potential_field_f32(mp_ctx* ctx, const float* d_positions, const float* goal, const float* d_obstacles, int64_t P,
                           int64_t O, float influence_distance, float* d_potential, float* d_gradient) {
  REQUIRE(ctx, "mp_potential_field_f32: null context");
  CTX_ENTER(ctx);
  REQUIRE(P >= 0 && O >= 0, "mp_potential_field_f32: negative P or O");
  if (P == 0) return MP_OK;
  REQUIRE(d_positions && goal && d_potential && d_gradient && (O == 0 || d_obstacles), "mp_potential_field_f32: null pointer");
  HIP_TRY(mpk_potential_field(ctx->compute, d_positions, goal, d_obstacles, (long)P, (long)O, influence_distance, d_potential, d_gradient));
  return MP_OK;
}
int mp_potential_field_host_f32(mp_ctx* ctx, const float* positions, const float* goal, const float* obstacles, int64_t P,
                                int64_t O, float influence_distance, float* potential, float* gradient) {
  REQUIRE(ctx, "mp_potential_field_host_f32: null context");
  CTX_ENTER(ctx);
  REQUIRE(P >= 0 && O >= 0, "mp_potential_field_host_f32: negative P or O");
  if (P == 0) return MP_OK;
  REQUIRE(positions && goal && potential && gradient && (O == 0 || obstacles), "mp_potential_field_host_f32: null pointer");
  const size_t pb = (size_t)P * 3 * sizeof(float), ob = (size_t)O * 3 * sizeof(float);
  Scratch sc(ctx);
  void *dp, *dob = nullptr, *du, *dg;
  if (int rc = sc.get(pb, &dp)) return rc;
  if (O) if (int rc = sc.get(ob, &dob)) return rc;
  if (int rc = sc.get((size_t)P * sizeof(float), &du)) return rc;
  if (int rc = sc.get(pb, &dg)) return rc;
  H2D(dp, positions, pb);
  if (O) H2D(dob, obstacles, ob);
  if (int rc = mp_potential_field_f32(ctx, (float*)dp, goal, (float*)dob, P, O, influence_distance, (float*)du, (float*)dg)) return rc;
  D2H(potential, du, (size_t)P * sizeof(float));
  D2H(gradient, dg, pb);
  HIP_TRY(hipStreamSynchronize(ctx->compute));
  return MP_OK;
}

}  // extern "C"
template <typename PT, int CAP>
static int ik_params(const char* fn, const mp_model* model, const double* joint_limits, double eomg, double ev, int max_iterations,
                     double damping, double step_cap, double w_o, double w_p, int adaptive, int backtracking, uint32_t seed,
                     PT* P) {
  REQUIRE(max_iterations >= 1, "%s: max_iterations must be at least 1 (got %d)", fn, max_iterations);
  REQUIRE(eomg > 0 && ev > 0 && damping >= 0 && step_cap > 0, "%s: eomg, ev, step_cap must be positive and damping non-negative", fn);
  P->eomg = eomg; P->ev = ev; P->damping = damping; P->step_cap = step_cap; P->w_o = w_o; P->w_p = w_p;
  P->max_iterations = max_iterations; P->seed = seed;
  P->adaptive_tuning = adaptive ? 1 : 0; P->backtracking = backtracking ? 1 : 0;
  const int n = model->d.n;
  for (int j = 0; j < CAP; ++j) {
    P->lo[j] = (j < n && joint_limits) ? joint_limits[2 * j] : -HUGE_VAL;
    P->hi[j] = (j < n && joint_limits) ? joint_limits[2 * j + 1] : HUGE_VAL;
    REQUIRE(!(P->lo[j] > P->hi[j]), "%s: joint %d has lower limit above upper limit", fn, j);
  }
  return MP_OK;
}
extern "C" {

int mp_inverse_kinematics_f64(mp_ctx* ctx, const mp_model* model, const double* d_T_desired, const double* d_theta0, int64_t B,
                              const double* joint_limits, double eomg, double ev, int max_iterations, double damping,
                              double step_cap, double weight_orientation, double weight_position, int adaptive_tuning, int backtracking,
                              uint32_t seed,
                              double* d_theta, int32_t* d_success, int32_t* d_iterations, int32_t* d_restarts) {
  CHECK_COMMON("mp_inverse_kinematics_f64");
  REQUIRE(B >= 0, "mp_inverse_kinematics_f64: negative problem count");
  if (B == 0) return MP_OK;
  REQUIRE(d_T_desired && d_theta0 && d_theta && d_success && d_iterations && d_restarts, "mp_inverse_kinematics_f64: null device pointer");
  REQUIRE(aligned16(d_T_desired) && aligned16(d_theta0) && aligned16(d_theta), "mp_inverse_kinematics_f64: device pointers must be 16-byte aligned");
  if (!ctx->queue_counter) {
    REQUIRE(!ctx->capturing, "mp_inverse_kinematics_f64: first use allocates; call it once before capturing a launch graph");
    HIP_TRY(hipMalloc(&ctx->queue_counter, 256));
  }
  if (model->big) {  // 9..32 joints: the same iteration on the run-time-n kinematics (csrc/mp_dyn.h)
    MpIkBigParams PB;
    if (int rc = ik_params<MpIkBigParams, MP_BIG_DOF>("mp_inverse_kinematics_f64", model, joint_limits, eomg, ev, max_iterations, damping,
                                                      step_cap, weight_orientation, weight_position, adaptive_tuning, backtracking, seed, &PB))
      return rc;
    const MpBigModel<double>* dm = nullptr;
    if (int rc = device_big_model<double>(ctx, model, &dm)) return rc;
    HIP_TRY(mpk_dyn_ik(ctx->compute, model->d.n, dm, PB, d_T_desired, d_theta0, (long)B, d_theta, d_success, d_iterations, d_restarts,
                       (unsigned long long*)ctx->queue_counter, ctx->compute_units));
    return MP_OK;
  }
  MpIkParams P;
  if (int rc = ik_params<MpIkParams, MP_MAX_DOF>("mp_inverse_kinematics_f64", model, joint_limits, eomg, ev, max_iterations, damping, step_cap,
                                                 weight_orientation, weight_position, adaptive_tuning, backtracking, seed, &P))
    return rc;
  if (const MpSpec* sp = find_spec(ctx, model)) {  // this robot's constants baked in (mp_model_specialize)
    for (int j = 0; j < MP_MAX_DOF; ++j) {  // the specialised build assumes finite arithmetic: open limits become huge ones
      if (!(P.lo[j] > -1e300)) P.lo[j] = -1e300;
      if (!(P.hi[j] < 1e300)) P.hi[j] = 1e300;
    }
    HIP_TRY(hipMemsetAsync(ctx->queue_counter, 0, sizeof(unsigned long long), ctx->compute));
    long nb = (long)B;
    void* counter = ctx->queue_counter;
    void* args[] = {&P, &d_T_desired, &d_theta0, &nb, &d_theta, &d_success, &d_iterations, &d_restarts, &counter};
    const long want = (nb + 255) / 256, cap = 1L * (ctx->compute_units > 0 ? ctx->compute_units : 256);  // one block per CU: see mp_spec_ik
    return launch_spec(ctx, sp->ik, (want < cap ? want : cap) * 256, args);
  }
  HIP_TRY(mpk_ik(ctx->compute, model->d, P, d_T_desired, d_theta0, (long)B, d_theta, d_success, d_iterations, d_restarts,
                 (unsigned long long*)ctx->queue_counter, ctx->compute_units));
  return MP_OK;
}

int mp_inverse_kinematics_host_f64(mp_ctx* ctx, const mp_model* model, const double* T_desired, const double* theta0, int64_t B,
                                   const double* joint_limits, double eomg, double ev, int max_iterations, double damping,
                                   double step_cap, double weight_orientation, double weight_position, int adaptive_tuning, int backtracking,
                              uint32_t seed,
                                   double* theta, int32_t* success, int32_t* iterations, int32_t* restarts) {
  CHECK_COMMON("mp_inverse_kinematics_host_f64");
  REQUIRE(B >= 0, "mp_inverse_kinematics_host_f64: negative problem count");
  if (B == 0) return MP_OK;
  REQUIRE(T_desired && theta0 && theta && success && iterations && restarts, "mp_inverse_kinematics_host_f64: null host pointer");
  const size_t tb = (size_t)B * 16 * sizeof(double), qb = (size_t)B * (size_t)model->d.n * sizeof(double), ib = (size_t)B * sizeof(int32_t);
  Scratch sc(ctx);
  void *dT, *d0, *dq, *dok, *dit, *drs;
  if (int rc = sc.get(tb, &dT)) return rc;
  if (int rc = sc.get(qb, &d0)) return rc;
  if (int rc = sc.get(qb, &dq)) return rc;
  if (int rc = sc.get(ib, &dok)) return rc;
  if (int rc = sc.get(ib, &dit)) return rc;
  if (int rc = sc.get(ib, &drs)) return rc;
  H2D(dT, T_desired, tb);
  H2D(d0, theta0, qb);
  if (int rc = mp_inverse_kinematics_f64(ctx, model, (double*)dT, (double*)d0, B, joint_limits, eomg, ev, max_iterations, damping,
                                         step_cap, weight_orientation, weight_position, adaptive_tuning, backtracking, seed, (double*)dq,
                                         (int32_t*)dok,
                                         (int32_t*)dit, (int32_t*)drs))
    return rc;
  D2H(theta, dq, qb);
  D2H(success, dok, ib);
  D2H(iterations, dit, ib);
  D2H(restarts, drs, ib);
  HIP_TRY(hipStreamSynchronize(ctx->compute));
  return MP_OK;
}

}  // extern "C"


// exposed to mp_comm.cpp
// (whoever asks for the stream is about to enqueue something that may read torques: parked float64 passes run first)
hipStream_t mp_ctx_compute_stream(mp_ctx* ctx) { return ctx->compute; }
// ... after this: the communicator is about to enqueue something that reads torques, so the parked float64 passes run first - and
// a pass that cannot be launched fails the collective instead of letting it send float32-only rows
int mp_ctx_flush_parked(mp_ctx* ctx) {
  CTX_ENTER(ctx);
  return MP_OK;
}
int mp_ctx_device(mp_ctx* ctx) { return ctx->device; }
int mp_set_error(int code, const char* msg) { return set_err(code, "%s", msg); }
