// Micro-benchmark: sustained wave64 VALU issue rate on gfx950 (plain v_fma_f32, v_pk_fma_f32, v_exp_f32 mixes) at
// 1..8 waves per SIMD.  Diagnostic only (prices the K6/K7 VALU roofline in DESIGN.md); not part of the product.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float float2v __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ void __launch_bounds__(64) k(float* out, int iters, float s) {
  float a[8];
  float2v p[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { a[i] = threadIdx.x * 0.001f + i; p[i] = float2v{a[i], a[i] + 1.f}; }
  const float2v s2 = {s, s * 1.0001f};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 8; ++r) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (MODE == 0) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(a[i]) : "v"(s));
        if (MODE == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(p[i]) : "v"(s2));
        if (MODE == 2) { if (i == 0) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i])); else asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(a[i]) : "v"(s)); }
        if (MODE == 3) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(s));
        if (MODE == 4) asm volatile("v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(a[i]));
        if (MODE == 5) asm volatile("v_cmp_ge_f32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(s) : "vcc");
        if (MODE == 6) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(s) : "vcc");
        if (MODE == 7) asm volatile("v_cmp_ge_f32 s[20:21], %0, %1" : : "v"(a[i]), "v"(s) : "s20", "s21");
        if (MODE == 8) asm volatile("v_min_f32 %0, %0, %1" : "+v"(a[i]) : "v"(s));
        if (MODE == 9) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(s));
        if (MODE == 10) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
        if (MODE == 11) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
        if (MODE == 12) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(a[i]) : "v"(s));
        if (MODE == 13) asm volatile("v_cmp_le_u32 s[20:21], %0, %1\n\ts_and_b64 s[22:23], s[20:21], vcc" : : "v"(a[i]), "v"(s) : "s20", "s21", "s22", "s23", "scc");
        if (MODE == 14) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(a[i]), "+v"(a[(i + 1) & 7]));
        if (MODE == 15) asm volatile("v_min_u32 %0, %0, %1" : "+v"(a[i]) : "v"(s));
        if (MODE == 16) asm volatile("v_cndmask_b32 %0, %0, %1, s[20:21]" : "+v"(a[i]) : "v"(s));
        if (MODE == 17) asm volatile("v_cmp_le_i32 s[20:21], %0, %1" : : "v"(a[i]), "v"(s) : "s20", "s21");
        if (MODE == 18) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[i]) : "v"(s));
        if (MODE == 19) asm volatile("v_med3_f32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(s));
        if (MODE == 20) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(a[i]) : "v"(s));
        if (MODE == 21) asm volatile("v_and_b32 %0, %0, %1" : "+v"(a[i]) : "v"(s));
        if (MODE == 22) { if (i & 1) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i])); else asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(a[i]) : "v"(s)); }
      }
    }
  }
  float r = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) r += a[i] + p[i].x + p[i].y;
  out[blockIdx.x * 64 + threadIdx.x] = r;
}

// The instruction mix of K7's strip body (gsr_kernels.hip.h k_render_bwd<false, 4, true>, the lambda `strip` and the three
// lines in front of it), as independent register chains so that only ISSUE is measured: 2 transcendentals (v_exp, v_rcp), one
// v_min, four compares into SGPR pairs, two selects, and 23 plain multiply / add / fma -- 32 vector instructions per strip
// evaluation.  The rate this loop sustains is the ceiling of a kernel made of that mix; K7's own rate (SQ_INSTS_VALU per
// launch / duration, profiles/r06_pmc_traffic.json) is held against it in DESIGN.md section 6.
constexpr int MIX_N = 32;
__global__ void __launch_bounds__(64) kmix(float* out, int iters, float s) {
  float a[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) a[i] = threadIdx.x * 0.001f + i;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      asm volatile("v_exp_f32 %0, %0" : "+v"(a[0]));
      asm volatile("v_rcp_f32 %0, %0" : "+v"(a[1]));
      asm volatile("v_min_f32 %0, %0, %1" : "+v"(a[2]) : "v"(s));
      asm volatile("v_cmp_ge_f32 s[20:21], %0, %1" : : "v"(a[3]), "v"(s) : "s20", "s21");
      asm volatile("v_cmp_le_u32 s[22:23], %0, %1" : : "v"(a[4]), "v"(s) : "s22", "s23");
      asm volatile("v_cmp_le_f32 s[24:25], %0, %1" : : "v"(a[5]), "v"(s) : "s24", "s25");
      asm volatile("v_cmp_ge_f32 s[26:27], %0, %1" : : "v"(a[6]), "v"(s) : "s26", "s27");
      asm volatile("s_and_b64 s[20:21], s[20:21], s[22:23]\n\ts_and_b64 s[24:25], s[24:25], s[26:27]\n\ts_and_b64 s[20:21], s[20:21], s[24:25]"
                   : : : "s20", "s21", "s24", "s25", "scc");
      asm volatile("v_cndmask_b32 %0, %0, %1, s[20:21]" : "+v"(a[7]) : "v"(s));
      asm volatile("v_cndmask_b32 %0, %0, %1, s[20:21]" : "+v"(a[3]) : "v"(s));
#pragma unroll
      for (int i = 0; i < 23; ++i) {
        if (i % 3 == 0) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(a[i & 7]) : "v"(s));
        else if (i % 3 == 1) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i & 7]) : "v"(s));
        else asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i & 7]) : "v"(s));
      }
    }
  }
  float r = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) r += a[i];
  out[blockIdx.x * 64 + threadIdx.x] = r;
}

void run_mix(int wps) {
  const int waves = 256 * 4 * wps, iters = 2000;
  float* out;
  hipMalloc(&out, (size_t)waves * 64 * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  kmix<<<waves, 64>>>(out, 10, 1.0001f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  kmix<<<waves, 64>>>(out, iters, 1.0001f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double instr = (double)waves * iters * 8 * MIX_N;
  printf("%-28s waves/SIMD=%d  %.3f ms  %.1f G wave-instr/s  (%.2f cyc/instr/SIMD @2.4GHz)\n", "K7 strip-body mix (32)", wps, ms,
         instr / ms / 1e6, 1024 * 2.4e9 / (instr / (ms * 1e-3)));
  hipFree(out);
}

template <int MODE>
void run(const char* name, int wps) {
  const int waves = 256 * 4 * wps, iters = 2000;
  float* out;
  hipMalloc(&out, (size_t)waves * 64 * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  k<MODE><<<waves, 64>>>(out, 10, 1.0001f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<MODE><<<waves, 64>>>(out, iters, 1.0001f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double instr = (double)waves * iters * 64;
  printf("%-28s waves/SIMD=%d  %.3f ms  %.1f G wave-instr/s  (%.2f cyc/instr/SIMD @2.4GHz)\n", name, wps, ms,
         instr / ms / 1e6, 1024 * 2.4e9 / (instr / (ms * 1e-3)));
  hipFree(out);
}

int main() {
  for (int wps : {1, 5, 6, 8}) {
    run<0>("v_fma_f32", wps);
    run<1>("v_pk_fma_f32", wps);
    run<2>("1 v_exp + 7 v_fma", wps);
    run<3>("v_add_f32", wps);
    run<4>("v_add_f32_dpp", wps);
    run<5>("v_cmp(vcc)+v_cndmask pair", wps);
    run<6>("v_cndmask_b32", wps);
    run<7>("v_cmp_ge_f32 -> sgpr", wps);
    run<8>("v_min_f32", wps);
    run<9>("v_mul_f32", wps);
    run<10>("v_rcp_f32", wps);
    run<11>("v_exp_f32", wps);
    run<12>("v_sub_u32", wps);
    run<13>("v_cmp->sgpr + s_and", wps);
    run<14>("v_permlane32_swap", wps);
    run<15>("v_min_u32", wps);
    run<16>("v_cndmask_b32 (sgpr mask)", wps);
    run<17>("v_cmp_le_i32 -> sgpr", wps);
    run<18>("v_max_f32", wps);
    run<19>("v_med3_f32", wps);
    run<20>("v_sub_f32", wps);
    run<21>("v_and_b32", wps);
    run<22>("1 v_exp : 1 v_fma", wps);
    run_mix(wps);
  }
  return 0;
}
