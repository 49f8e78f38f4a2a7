// The roll-out kernel's tile I/O without its arithmetic: 2048 one-wave blocks (8 per CU by LDS, like the kernel), each
// wave moves, per 4-step tile, 12 lane-strided 16-byte loads (torques + wrenches: 96-byte runs, 2400 bytes apart) and 18
// wave-cooperative flat 16-byte stores (pos / vel / acc runs), with a stand-in for the four integration steps between
// them: nothing, s_sleep (no issue slots, no power) or a dependent FMA chain (issue slots + power).
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench_c5io tools/ubench_c5io.hip && tools/ubench_c5io
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); std::exit(1); } } while (0)
typedef unsigned u4 __attribute__((ext_vector_type(4)));
constexpr int N = 6, KS = 4, C = KS * N / 4;

// what: bit 0 loads, bit 1 stores, bit 2 = loads are issued BEFORE the stand-in and consumed after it (prefetch)
// work: 0 none, 1 s_sleep, 2 FMA chain; amount = sleep units of 64 cycles / FMA iterations; stagger: blocks whose CU_ID is odd
// start `stagger` sleep units late
template <int WHAT, int WORK>
__global__ __launch_bounds__(64) void k_io(const float* __restrict__ tau, const float* __restrict__ F, float* __restrict__ pos,
                                            float* __restrict__ vel, float* __restrict__ acc, long Nt, long B, int amount, int stagger,
                                            float* __restrict__ sink) {
  __shared__ unsigned lds[72 * 65];  // 18.7 KB: eight waves per CU, as in the kernel
  const int lane = threadIdx.x;
  const long b0 = (long)blockIdx.x * 64, b = b0 + lane;
  lds[lane] = 0;
  if (stagger) {
    const unsigned cu = __builtin_amdgcn_s_getreg((3 << 11) | (8 << 6) | 4);
    if (cu & 1)
      for (int k = 0; k < stagger; ++k) __builtin_amdgcn_s_sleep(64);
  }
  u4 keep = {0, 0, 0, 0};
  float x = (float)lane;
  for (long i0 = 0; i0 + KS <= Nt; i0 += KS) {
    u4 v[12];
    if ((WHAT & 1) && (WHAT & 128)) {
      // wave-cooperative flat loads: chunk f = k * 64 + lane of the wave's 64 runs is trajectory f / 6, piece f % 6 (a load
      // instruction covers ~11 whole runs instead of 64 fragments; the real thing would transpose through LDS)
      const long run0 = (b0 * Nt + i0) * N, pitch = Nt * N;
#pragma unroll
      for (int k = 0; k < C; ++k) {
        const int f = k * 64 + lane, t = f / C, c = f - t * C;
        keep ^= *reinterpret_cast<const u4*>(tau + run0 + (long)t * pitch + 4 * c);
        keep ^= *reinterpret_cast<const u4*>(F + run0 + (long)t * pitch + 4 * c);
      }
    } else if ((WHAT & 1) && (WHAT & 64)) {
      // line-exact reads: a lane fetches a whole 128-byte line of each input stream only when its run of this tile reaches
      // past what it has already fetched (the rest of a line waits in registers in the real thing); lanes that need nothing
      // this tile get an offset outside the descriptor (no traffic, no branch)
      const int tile = (int)(i0 / KS);
      const long s0 = b * Nt * N * 4;
      const long need = s0 + 96l * (tile + 1);
      long have = tile == 0 ? (s0 & ~127l) : ((s0 + 96l * tile + 127) & ~127l);   // fetched so far (line boundary)
      const __amdgpu_buffer_rsrc_t rt = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(tau), 0, (int)(B * Nt * N * 4), 0x00020000);
      const __amdgpu_buffer_rsrc_t rf = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(F), 0, (int)(B * Nt * 6 * 4), 0x00020000);
#pragma unroll
      for (int rep = 0; rep < 2; ++rep) {  // (the first tile of a misaligned stream needs two lines)
        const bool go = have < need && (rep == 0 || tile == 0);
        const int off = go ? (int)have : 0x7ffffff0;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          keep ^= __builtin_amdgcn_raw_buffer_load_b128(rt, off + 16 * k, 0, 0);
          keep ^= __builtin_amdgcn_raw_buffer_load_b128(rf, off + 16 * k, 0, 0);
        }
        if (go) have += 128;
        if (tile != 0) break;
      }
    } else if ((WHAT & 1) && (WHAT & 16)) {
      // whole 64-byte blocks: a run that starts on a block boundary ((b + tile) even) takes two blocks, the second one's
      // last 32 bytes are the head of the next tile's run and stay in registers; the next tile then needs one block only
      const int tile = (int)(i0 / KS);
      const bool even = ((b + tile) & 1) == 0;
      const u4* gt = reinterpret_cast<const u4*>(tau + (b * Nt + i0) * N) + (even ? 0 : 2);
      const u4* gf = reinterpret_cast<const u4*>(F + (b * Nt + i0) * 6) + (even ? 0 : 2);
      const bool last = i0 + 2 * KS > Nt;
#pragma unroll
      for (int k = 0; k < 4; ++k) { v[k] = gt[k]; v[6 + k] = gf[k]; }
      if (even && !last) {
#pragma unroll
        for (int k = 4; k < 8; ++k) { keep ^= gt[k]; keep ^= gf[k]; }
      } else if (even) {
#pragma unroll
        for (int k = 4; k < 6; ++k) { keep ^= gt[k]; keep ^= gf[k]; }
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) { keep ^= v[k]; keep ^= v[6 + k]; }
    } else if (WHAT & 1) {
      const u4* gt = reinterpret_cast<const u4*>(tau + (b * Nt + i0) * N);
      const u4* gf = reinterpret_cast<const u4*>(F + (b * Nt + i0) * 6);
#pragma unroll
      for (int k = 0; k < 6; ++k) { v[k] = gt[k]; v[6 + k] = gf[k]; }
      if (!(WHAT & 4)) {
#pragma unroll
        for (int k = 0; k < 12; ++k) keep ^= v[k];
      }
    }
    if (WORK == 1) {
      for (int k = 0; k < amount; ++k) __builtin_amdgcn_s_sleep(64);
    } else if (WORK == 2) {
      float a0 = x, a1 = x + 1, a2 = x + 2, a3 = x + 3;
      for (int k = 0; k < amount; ++k) {
        a0 = a0 * 1.0001f + 0.5f; a1 = a1 * 1.0001f + 0.5f; a2 = a2 * 1.0001f + 0.5f; a3 = a3 * 1.0001f + 0.5f;
      }
      x = a0 + a1 + a2 + a3;
    }
    if ((WHAT & 1) && (WHAT & 4)) {
#pragma unroll
      for (int k = 0; k < 12; ++k) keep ^= v[k];
    }
    if ((WHAT & 2) && (WHAT & 32)) {
      // whole 128-byte LINES only, each lane its own trajectory (an upper bound on what a line-exact flush costs: a
      // cooperative version would coalesce better): this tile the lane writes the lines its stream has completed
      const int tile = (int)(i0 / KS);
      const bool last = i0 + 2 * KS > Nt;
      float* const arr[3] = {pos, vel, acc};
      const long s0 = b * Nt * N * 4;                          // byte offset of the trajectory's stream
      const long done = tile == 0 ? s0 : ((s0 + 96l * tile) & ~127l);         // written so far (line boundary, or the start)
      const long upto = last ? s0 + 96l * (tile + 1) : ((s0 + 96l * (tile + 1)) & ~127l);
#pragma unroll
      for (int slot = 0; slot < 3; ++slot) {
        char* base = reinterpret_cast<char*>(arr[slot]);
#pragma unroll
        for (int k = 0; k < 14; ++k) {
          const long at = done + 16l * k;
          if (at < upto) *reinterpret_cast<u4*>(base + at) = keep;
        }
      }
    } else if ((WHAT & 2) && (WHAT & 8)) {
      // 64-byte-aligned variant: a trajectory's run of this tile starts on a 64-byte boundary when (b + tile) is even -
      // then its last 32 bytes (pieces 4, 5) are an incomplete block and are held back; when (b + tile) is odd the run starts
      // 32 bytes into a block: the owning lane first writes the 32 bytes held back by the previous tile, then the whole run leaves
      const long run0 = (b0 * Nt + i0) * N, pitch = Nt * N;
      const int tile = (int)(i0 / KS);
      float* const arr[3] = {pos, vel, acc};
      if (((b + tile) & 1) && tile > 0) {
#pragma unroll
        for (int slot = 0; slot < 3; ++slot) {
          u4* g = reinterpret_cast<u4*>(arr[slot] + (b * Nt + i0) * N - 8);
          g[0] = keep; g[1] = keep;
        }
      }
#pragma unroll
      for (int slot = 0; slot < 3; ++slot) {
#pragma unroll
        for (int k = 0; k < C; ++k) {
          const int f = k * 64 + lane, t = f / C, c = f - t * C;
          u4 o = keep;
          o.x += (unsigned)(slot + k);
          const bool last_tile = i0 + 2 * KS > Nt;
          if (c < 4 || ((t + tile) & 1) || last_tile) *reinterpret_cast<u4*>(arr[slot] + run0 + (long)t * pitch + 4 * c) = o;
        }
      }
    } else if (WHAT & 2) {
      const long run0 = (b0 * Nt + i0) * N, pitch = Nt * N;
      float* const arr[3] = {pos, vel, acc};
#pragma unroll
      for (int slot = 0; slot < 3; ++slot) {
#pragma unroll
        for (int k = 0; k < C; ++k) {
          const int f = k * 64 + lane, t = f / C, c = f - t * C;
          u4 o = keep;
          o.x += (unsigned)(slot + k);
          *reinterpret_cast<u4*>(arr[slot] + run0 + (long)t * pitch + 4 * c) = o;
        }
      }
    }
  }
  if (x == 12345.678f || keep.x == 0x12345678u) sink[0] = x + (float)keep.y;
}

template <int WHAT, int WORK>
float run(const char* name, int blocks, long Nt, long B, int amount, int stagger, float* tau, float* F, float* pos, float* vel, float* acc,
          float* sink, float base_ms) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int w = 0; w < 30; ++w) hipLaunchKernelGGL((k_io<WHAT, WORK>), dim3(blocks), dim3(64), 0, 0, tau, F, pos, vel, acc, Nt, B, amount, stagger, sink);
  CK(hipDeviceSynchronize());
  const int reps = 60;
  CK(hipEventRecord(a));
  for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((k_io<WHAT, WORK>), dim3(blocks), dim3(64), 0, 0, tau, F, pos, vel, acc, Nt, B, amount, stagger, sink);
  CK(hipEventRecord(b));
  CK(hipDeviceSynchronize());
  float ms = 0;
  CK(hipEventElapsedTime(&ms, a, b));
  ms /= reps;
  const double gb = ((WHAT & 1) ? 2.0 : 0.0) * B * Nt * 6 * 4 / 1e9 + ((WHAT & 2) ? 3.0 : 0.0) * B * Nt * 6 * 4 / 1e9;
  std::printf("%-58s %8.4f ms  %6.2f GB  %7.0f GB/s   over the stand-in alone: %+8.4f ms\n", name, ms, gb, gb / (ms * 1e-3), ms - base_ms);
  CK(hipEventDestroy(a)); CK(hipEventDestroy(b));
  return ms;
}

int main(int argc, char** argv) {
  const long B = argc > 1 ? std::atol(argv[1]) : 131072, Nt = argc > 2 ? std::atol(argv[2]) : 100;
  const int sleep_units = argc > 3 ? std::atoi(argv[3]) : 13, fma_iters = argc > 4 ? std::atoi(argv[4]) : 800;
  const int blocks = (int)(B / 64);
  float *tau, *F, *pos, *vel, *acc, *sink;
  const size_t nb = (size_t)B * Nt * 6 * 4;
  CK(hipMalloc(&tau, nb)); CK(hipMalloc(&F, nb)); CK(hipMalloc(&pos, nb)); CK(hipMalloc(&vel, nb)); CK(hipMalloc(&acc, nb)); CK(hipMalloc(&sink, 64));
  CK(hipMemset(tau, 0, nb)); CK(hipMemset(F, 0, nb));
  std::printf("B %ld Nt %ld: %d one-wave blocks, %ld tiles per wave; stand-ins: %d x s_sleep 64, %d FMA iterations x 4 chains\n", B, Nt, blocks, Nt / 4,
              sleep_units, fma_iters);
#define RUN(WHAT, WORK, amount, stagger, base, label) run<WHAT, WORK>(label, blocks, Nt, B, amount, stagger, tau, F, pos, vel, acc, sink, base)
  RUN(1, 0, 0, 0, 0.f, "loads only, back to back");
  RUN(2, 0, 0, 0, 0.f, "stores only, back to back");
  RUN(3, 0, 0, 0, 0.f, "loads + stores, back to back");
  RUN(10, 0, 0, 0, 0.f, "stores in whole 64-byte blocks (32-byte tails held back), back to back");
  RUN(11, 0, 0, 0, 0.f, "loads + stores in whole 64-byte blocks, back to back");
  const float s0 = RUN(0, 1, sleep_units, 0, 0.f, "s_sleep stand-in alone");
  RUN(1, 1, sleep_units, 0, s0, "loads, s_sleep between tiles");
  RUN(2, 1, sleep_units, 0, s0, "stores, s_sleep between tiles");
  RUN(3, 1, sleep_units, 0, s0, "loads + stores, s_sleep between tiles");
  RUN(7, 1, sleep_units, 0, s0, "loads (prefetched over the stand-in) + stores, s_sleep");
  RUN(3, 1, sleep_units, sleep_units / 2, s0, "loads + stores, s_sleep, odd CUs half a tile late");
  const float f0 = RUN(0, 2, fma_iters, 0, 0.f, "FMA stand-in alone");
  RUN(1, 2, fma_iters, 0, f0, "loads, FMA between tiles");
  RUN(2, 2, fma_iters, 0, f0, "stores, FMA between tiles");
  RUN(3, 2, fma_iters, 0, f0, "loads + stores, FMA between tiles");
  RUN(7, 2, fma_iters, 0, f0, "loads (prefetched over the stand-in) + stores, FMA");
  RUN(3, 2, fma_iters, sleep_units / 2, f0, "loads + stores, FMA, odd CUs half a tile late");
  RUN(10, 2, fma_iters, 0, f0, "stores in whole 64-byte blocks, FMA between tiles");
  RUN(11, 2, fma_iters, 0, f0, "loads + stores in whole 64-byte blocks, FMA between tiles");
  RUN(34, 0, 0, 0, 0.f, "stores in whole 128-byte lines (lane = trajectory), back to back");
  RUN(35, 0, 0, 0, 0.f, "loads + stores in whole 128-byte lines, back to back");
  RUN(34, 2, fma_iters, 0, f0, "stores in whole 128-byte lines, FMA between tiles");
  RUN(35, 2, fma_iters, 0, f0, "loads + stores in whole 128-byte lines, FMA between tiles");
  RUN(65, 0, 0, 0, 0.f, "loads line-exact (whole 128-byte lines, no re-reads), back to back");
  RUN(65, 2, fma_iters, 0, f0, "loads line-exact, FMA between tiles");
  RUN(75, 0, 0, 0, 0.f, "loads line-exact + stores in 64-byte blocks, back to back");
  RUN(75, 2, fma_iters, 0, f0, "loads line-exact + stores in 64-byte blocks, FMA between tiles");
  RUN(129, 0, 0, 0, 0.f, "loads wave-cooperative (flat chunk order), back to back");
  RUN(129, 2, fma_iters, 0, f0, "loads wave-cooperative, FMA between tiles");
  RUN(163, 2, fma_iters, 0, f0, "loads wave-cooperative + stores in whole 128-byte lines, FMA between tiles");
  RUN(35, 2, fma_iters, 0, f0, "loads lane-strided + stores in whole 128-byte lines, FMA between tiles (again)");
  RUN(17, 0, 0, 0, 0.f, "loads in whole 64-byte blocks only, back to back");
  RUN(17, 2, fma_iters, 0, f0, "loads in whole 64-byte blocks, FMA between tiles");
  RUN(27, 0, 0, 0, 0.f, "loads AND stores in whole 64-byte blocks, back to back");
  RUN(27, 2, fma_iters, 0, f0, "loads AND stores in whole 64-byte blocks, FMA between tiles");
  return 0;
}
