// Dev microbenchmark (not part of the product): what limits a kernel that can only put 1-2 waves on a SIMD?
// Measures cycles per INSTRUCTION of a wave for streams of (a) v_fma_f32, (b) v_fma_f32 interleaved 1:1 with s_mov_b32,
// (c) v_pk_fma_f32, (d) v_pk_fma_f32 interleaved 2:1 with s_mov_b32 (a constant pair per packed op), (e) v_pk_fma_f32 whose
// constant operand comes from s_load_dwordx2 every 4th instruction - at 1, 2, 4 and 8 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));

#define REP8(x) x x x x x x x x
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, const float* ctab) {
  const int tid = blockIdx.x * 256 + threadIdx.x;
  float a0 = tid * 1e-9f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;
  v2f p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a1, a2}, p3 = {a3, a0};
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {
      REP8(asm volatile("v_fma_f32 %0, %0, 0.5, 1.0\n v_fma_f32 %1, %1, 0.5, 1.0\n v_fma_f32 %2, %2, 0.5, 1.0\n v_fma_f32 %3, %3, 0.5, 1.0" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));)
    } else if (MODE == 1) {
      REP8(asm volatile("v_fma_f32 %0, %0, 0.5, 1.0\n s_mov_b32 s20, 0x3f000000\n v_fma_f32 %1, %1, 0.5, 1.0\n s_mov_b32 s21, 0x3f000001\n v_fma_f32 %2, %2, 0.5, 1.0\n s_mov_b32 s22, 0x3f000002\n v_fma_f32 %3, %3, 0.5, 1.0\n s_mov_b32 s23, 0x3f000003" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : : "s20", "s21", "s22", "s23");)
    } else if (MODE == 2) {
      REP8(asm volatile("v_pk_fma_f32 %0, %0, 0.5, 1.0\n v_pk_fma_f32 %1, %1, 0.5, 1.0\n v_pk_fma_f32 %2, %2, 0.5, 1.0\n v_pk_fma_f32 %3, %3, 0.5, 1.0" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3));)
    } else if (MODE == 3) {
      REP8(asm volatile("s_mov_b32 s20, 0x3f000000\n s_mov_b32 s21, 0x3f000001\n v_pk_fma_f32 %0, %0, s[20:21], 1.0\n s_mov_b32 s22, 0x3f000002\n s_mov_b32 s23, 0x3f000003\n v_pk_fma_f32 %1, %1, s[22:23], 1.0\n"
                        "s_mov_b32 s20, 0x3f000004\n s_mov_b32 s21, 0x3f000005\n v_pk_fma_f32 %2, %2, s[20:21], 1.0\n s_mov_b32 s22, 0x3f000006\n s_mov_b32 s23, 0x3f000007\n v_pk_fma_f32 %3, %3, s[22:23], 1.0"
                        : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : : "s20", "s21", "s22", "s23");)
    } else {
      REP8(asm volatile("s_load_dwordx8 s[20:27], %4, 0x0\n s_waitcnt lgkmcnt(0)\n v_pk_fma_f32 %0, %0, s[20:21], 1.0\n v_pk_fma_f32 %1, %1, s[22:23], 1.0\n v_pk_fma_f32 %2, %2, s[24:25], 1.0\n v_pk_fma_f32 %3, %3, s[26:27], 1.0"
                        : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "s"(ctab) : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");)
    }
  }
  out[tid] = a0 + a1 + a2 + a3 + p0.x + p1.y + p2.x + p3.y;
}

template <int MODE>
void run(const char* name, int waves_per_simd, int inst_per_iter, const float* ctab) {
  const int blocks = 256 * waves_per_simd;
  float* d;
  hipMalloc(&d, (size_t)blocks * 256 * 4);
  const int iters = 4000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  k<MODE><<<blocks, 256>>>(d, 100, ctab);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<MODE><<<blocks, 256>>>(d, iters, ctab);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double ns_per_inst_per_wave = ms * 1e6 / ((double)iters * inst_per_iter);
  printf("%-34s waves/SIMD=%d  %.3f ms  %.2f ns per instruction of a wave (%.2f ns per instr per SIMD)\n", name, waves_per_simd, ms,
         ns_per_inst_per_wave, ns_per_inst_per_wave / waves_per_simd);
  hipFree(d);
}

int main() {
  float* ctab;
  hipMalloc(&ctab, 256);
  hipMemset(ctab, 0, 256);
  for (int w : {1, 2, 4, 8}) {
    run<0>("v_fma x32", w, 32, ctab);
    run<1>("v_fma x32 + s_mov x32", w, 64, ctab);
    run<2>("v_pk_fma x32", w, 32, ctab);
    run<3>("v_pk_fma x32 + s_mov x64", w, 96, ctab);
    run<4>("v_pk_fma x32 + s_load_x8 x8 (+wait)", w, 48, ctab);
  }
  return 0;
}
