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
      }
    }
  }
  float r = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) r += a[i] + p[i].x + p[i].y;
  out[blockIdx.x * 64 + threadIdx.x] = r;
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
  for (int wps : {1, 2, 4, 8}) {
    run<0>("v_fma_f32", wps);
    run<1>("v_pk_fma_f32", wps);
    run<2>("1 v_exp + 7 v_fma", wps);
    run<3>("v_add_f32", wps);
    run<4>("v_add_f32_dpp", wps);
  }
  return 0;
}
