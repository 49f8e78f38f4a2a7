// Dev microbenchmark (not part of the product): cost of one VALU instruction by operand form on gfx950, 8 independent
// destination chains, at 1 / 2 / 4 / 8 waves per SIMD.  ns per instruction per SIMD (lower = cheaper).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
#define REP4(x) x x x x

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, float fa, float fb) {
  const int tid = blockIdx.x * 256 + threadIdx.x;
  float d[8], a = fa + tid * 1e-9f, b = fb - tid * 1e-9f;
  v2f p[8], pa = {a, b}, pb = {b, a};
#pragma unroll
  for (int i = 0; i < 8; ++i) { d[i] = tid * 1e-9f + i; p[i] = (v2f){d[i], d[i] + 1}; }
  for (int it = 0; it < iters; ++it) {
#define OPS8(fmt) REP4(asm volatile(fmt(0) fmt(1) fmt(2) fmt(3) fmt(4) fmt(5) fmt(6) fmt(7) : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]), "+v"(d[4]), "+v"(d[5]), "+v"(d[6]), "+v"(d[7]) : "v"(a), "v"(b));)
#define PKS8(fmt) REP4(asm volatile(fmt(0) fmt(1) fmt(2) fmt(3) fmt(4) fmt(5) fmt(6) fmt(7) : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]), "+v"(p[4]), "+v"(p[5]), "+v"(p[6]), "+v"(p[7]) : "v"(pa), "v"(pb));)
#define F_FMA1(i) "v_fma_f32 %" #i ", %" #i ", 0.5, 1.0\n"
#define F_FMAC(i) "v_fmac_f32 %" #i ", %8, %9\n"
#define F_FMA3(i) "v_fma_f32 %" #i ", %8, %9, %" #i "\n"
#define F_FMAMK(i) "v_fmamk_f32 %" #i ", %8, 0x3f000123, %" #i "\n"
#define F_MUL(i) "v_mul_f32 %" #i ", %8, %" #i "\n"
#define F_PK3(i) "v_pk_fma_f32 %" #i ", %8, %9, %" #i "\n"
#define F_PKSW(i) "v_pk_fma_f32 %" #i ", %8, %9, %" #i " op_sel:[1,0,0] op_sel_hi:[0,1,1]\n"
#define F_PKMUL(i) "v_pk_mul_f32 %" #i ", %8, %" #i "\n"
#define F_PKS(i) "v_pk_fma_f32 %" #i ", %8, s[20:21], %" #i "\n"
    if (MODE == 0) { OPS8(F_FMA1) }
    else if (MODE == 1) { OPS8(F_FMAC) }
    else if (MODE == 2) { OPS8(F_FMA3) }
    else if (MODE == 3) { OPS8(F_FMAMK) }
    else if (MODE == 4) { OPS8(F_MUL) }
    else if (MODE == 5) { PKS8(F_PK3) }
    else if (MODE == 6) { PKS8(F_PKSW) }
    else if (MODE == 7) { PKS8(F_PKMUL) }
    else { asm volatile("s_mov_b32 s20, 0x3f000000\n s_mov_b32 s21, 0x3f000001" ::: "s20", "s21"); PKS8(F_PKS) }
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += d[i] + p[i].x + p[i].y;
  out[tid] = s;
}

template <int MODE>
void run(const char* name, int waves_per_simd) {
  const int blocks = 256 * waves_per_simd;
  float* d;
  (void)hipMalloc(&d, (size_t)blocks * 256 * 4);
  const int iters = 4000;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  k<MODE><<<blocks, 256>>>(d, 100, 0.5f, 0.25f);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  k<MODE><<<blocks, 256>>>(d, iters, 0.5f, 0.25f);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  printf("%-44s waves/SIMD=%d  %.3f ms  %.2f ns per instruction per SIMD\n", name, waves_per_simd, ms,
         ms * 1e6 / ((double)iters * 32 * waves_per_simd));
  (void)hipFree(d);
}

int main() {
  for (int w : {1, 2, 4, 8}) {
    run<0>("v_fma_f32 d, d, 0.5, 1.0      (1 VGPR src)", w);
    run<1>("v_fmac_f32 d, a, b            (VOP2, 3 VGPR)", w);
    run<2>("v_fma_f32 d, a, b, d          (VOP3, 3 VGPR)", w);
    run<3>("v_fmamk_f32 d, a, literal, d", w);
    run<4>("v_mul_f32 d, a, d", w);
    run<5>("v_pk_fma_f32 d, a, b, d       (3 VGPR pairs)", w);
    run<6>("v_pk_fma_f32 ... op_sel swizzles", w);
    run<7>("v_pk_mul_f32 d, a, d", w);
    run<8>("v_pk_fma_f32 d, a, s[20:21], d (SGPR pair)", w);
  }
  return 0;
}
