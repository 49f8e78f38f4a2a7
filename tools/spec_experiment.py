#!/usr/bin/env python3
"""Experiment: how much does baking one robot's constants into the kernel buy?  Emits a stand-alone
.hip (generic packed kernel vs the same kernel with a constexpr model), builds it for gfx950 and prints
the instruction counts; run the binary on the GPU box for timings."""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import manipulapy_amd as mp  # noqa: E402

robot = sys.argv[1] if len(sys.argv) > 1 else "ur5"
sm, dyn, lim = mp.load_robot(robot)
b = dyn.hip_model(lim, None).blob(np.float32)
n = b["n"]


def snap(v):
    v = np.array(v, dtype=np.float64)
    scale = max(1.0, np.abs(v).max())
    v[np.abs(v) < 1e-7 * scale] = 0.0
    for t in (1.0, -1.0):
        v[np.abs(v - t) < 1e-7] = t
    return v


def lit(x):
    if np.isinf(x):
        return "-__builtin_inff()" if x < 0 else "__builtin_inff()"
    return repr(float(np.float32(x))) + "f"


def arr(v):
    return "{" + ", ".join(lit(x) for x in v) + "}"


J = b["joints"].copy()
for i in range(8):
    J[i, :6] = snap(J[i, :6]) if i < n else J[i, :6]
    J[i, 6:10] = snap(J[i, 6:10])
    J[i, 10:] = snap(J[i, 10:])
joints = ",\n    ".join("{" + ", ".join(lit(x) for x in J[i]) + "}" for i in range(8))
model = f"""static constexpr MpModel<float> kM = {{{n}, {{0, 0, 0}}, {arr(snap(b['base_R']))}, {arr(snap(b['base_p']))},
  {arr(snap(b['tool_R']))}, {arr(snap(b['tool_p']))},
  {{{joints}}},
  {arr(b['qmin'])}, {arr(b['qmax'])}, {arr(b['taumin'])}, {arr(b['taumax'])}}};"""

src = f"""
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "{ROOT}/manipulapy_amd/csrc/mp_core.h"
{model}
template <int N> __device__ __forceinline__ void load_pair(const float* base, long pair, mp_f2 (&v)[N]) {{
  const float4* s = reinterpret_cast<const float4*>(base + pair * 2 * N);
  float f[2 * N];
#pragma unroll
  for (int k = 0; k < 2 * N / 4; ++k) {{ float4 t = s[k]; f[4*k] = t.x; f[4*k+1] = t.y; f[4*k+2] = t.z; f[4*k+3] = t.w; }}
#pragma unroll
  for (int j = 0; j < N; ++j) v[j] = (mp_f2){{f[j], f[N + j]}};
}}
template <int N> __device__ __forceinline__ void store_pair(float* base, long pair, const mp_f2 (&v)[N]) {{
  float f[2 * N];
#pragma unroll
  for (int j = 0; j < N; ++j) {{ f[j] = v[j].x; f[N + j] = v[j].y; }}
  float4* d = reinterpret_cast<float4*>(base + pair * 2 * N);
#pragma unroll
  for (int k = 0; k < 2 * N / 4; ++k) d[k] = make_float4(f[4*k], f[4*k+1], f[4*k+2], f[4*k+3]);
}}
template <bool SPEC>
__global__ __launch_bounds__(256, 2) void k_id(const MpModel<float> Marg, const MpCall<float> C, const float* q, const float* qd,
                                               const float* qdd, float* tau, long pairs) {{
  constexpr int N = {n};
  const long p = (long)blockIdx.x * 256 + threadIdx.x;
  if (p >= pairs) return;
  mp_f2 a[N], b[N], c[N], t[N];
  load_pair<N>(q, p, a); load_pair<N>(qd, p, b); load_pair<N>(qdd, p, c);
  MpJointState<mp_f2, N> js;
  if (SPEC) {{
    mp_joint_state<mp_f2, N>(kM, a, js);
    mp_rnea<mp_f2, N, false>(kM, C, js, b, c, t);
  }} else {{
    mp_joint_state<mp_f2, N>(Marg, a, js);
    mp_rnea<mp_f2, N, false>(Marg, C, js, b, c, t);
  }}
  store_pair<N>(tau, p, t);
}}
int main() {{
  const long rows = 4096000, n = {n}, pairs = rows / 2;
  std::vector<float> h(rows * n);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) % 2000) / 1000.f - 1.f;
  float *q, *qd, *qdd, *t0, *t1;
  const size_t nb = rows * n * 4;
  hipMalloc(&q, nb); hipMalloc(&qd, nb); hipMalloc(&qdd, nb); hipMalloc(&t0, nb); hipMalloc(&t1, nb);
  hipMemcpy(q, h.data(), nb, hipMemcpyHostToDevice); hipMemcpy(qd, h.data(), nb, hipMemcpyHostToDevice); hipMemcpy(qdd, h.data(), nb, hipMemcpyHostToDevice);
  MpModel<float> M = kM; MpCall<float> C = {{{{0.f, 0.f, 9.81f}}, {{0, 0, 0}}, {{0, 0, 0}}}};
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep) for (int spec = 0; spec < 2; ++spec) {{
    float* out = spec ? t1 : t0;
    auto launch = [&] {{ if (spec) k_id<true><<<(pairs + 255) / 256, 256>>>(M, C, q, qd, qdd, out, pairs); else k_id<false><<<(pairs + 255) / 256, 256>>>(M, C, q, qd, qdd, out, pairs); }};
    for (int i = 0; i < 3; ++i) launch();
    hipEventRecord(e0); for (int i = 0; i < 20; ++i) launch(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%s: %.4f ms per launch\\n", spec ? "specialised" : "generic    ", ms / 20);
  }}
  std::vector<float> a(rows * n), b2(rows * n);
  hipMemcpy(a.data(), t0, nb, hipMemcpyDeviceToHost); hipMemcpy(b2.data(), t1, nb, hipMemcpyDeviceToHost);
  double md = 0, mx = 0; for (size_t i = 0; i < a.size(); ++i) {{ md = fmax(md, fabs((double)a[i] - b2[i])); mx = fmax(mx, fabs((double)a[i])); }}
  printf("max |generic - specialised| = %g (max |tau| %g)\\n", md, mx);
  return 0;
}}
"""
out = os.path.join(ROOT, "tools", f"spec_{robot}.hip")
open(out, "w").write(src)
flags = ["-w", "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=fast", "-fno-slp-vectorize"] + sys.argv[2:]
subprocess.run(["/opt/rocm/bin/hipcc"] + flags + ["-o", os.path.join(ROOT, "tools", f"spec_{robot}"), out], check=True)
subprocess.run(["/opt/rocm/bin/hipcc"] + flags + ["--offload-device-only", "-S", "-o", f"/tmp/spec_{robot}.s", out], check=True)
subprocess.run([sys.executable, os.path.join(ROOT, "tools/isa_stats.py"), f"/tmp/spec_{robot}.s", "k_idILb0", "k_idILb1"])
