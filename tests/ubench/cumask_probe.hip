// Diagnostic: which CUs does a hipExtStreamCreateWithCUMask stream run on?  Every wave of a grid
// large enough to touch every allowed CU records (XCC id, SE id, CU id) from the hardware registers; the host prints,
// per mask, the number of distinct CUs per XCC.  Not part of the product.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <set>
#include <vector>

__global__ void __launch_bounds__(64) probe(unsigned* out, int spin) {
  unsigned xcc, hwid;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
  float a = threadIdx.x;
  for (int i = 0; i < spin; ++i) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(a));
  if (threadIdx.x == 0) {
    out[2 * blockIdx.x] = xcc & 0xF;
    out[2 * blockIdx.x + 1] = hwid;
  }
  if (a == 12345.f) out[0] = 0;
}

int main() {
  const int nb = 8192;
  unsigned* d;
  hipMalloc(&d, 2 * nb * sizeof(unsigned));
  std::vector<unsigned> h(2 * nb);
  auto run = [&](const char* name, std::vector<uint32_t> mask) {
    hipStream_t st;
    hipError_t e = mask.empty() ? hipStreamCreate(&st) : hipExtStreamCreateWithCUMask(&st, (uint32_t)mask.size(), mask.data());
    if (e != hipSuccess) { printf("%s: stream creation failed: %s\n", name, hipGetErrorString(e)); return; }
    hipLaunchKernelGGL(probe, dim3(nb), dim3(64), 0, st, d, 20000);
    hipStreamSynchronize(st);
    hipMemcpy(h.data(), d, 2 * nb * sizeof(unsigned), hipMemcpyDeviceToHost);
    std::set<unsigned> cus[16];
    for (int b = 0; b < nb; ++b) {
      const unsigned hw = h[2 * b + 1];
      // HW_ID (gfx9): wave 3:0, simd 5:4, pipe 7:6, cu 11:8, sh 12, se 15:13 (gfx94x: se 3 bits)
      cus[h[2 * b] & 15].insert((hw >> 8) & 0xFF);
    }
    printf("%-34s", name);
    int total = 0;
    for (int x = 0; x < 8; ++x) { printf(" xcc%d:%2zu", x, cus[x].size()); total += (int)cus[x].size(); }
    printf("  total %d\n", total);
    hipStreamDestroy(st);
  };
  run("no mask", {});
  run("all ones (8 words)", std::vector<uint32_t>(8, 0xFFFFFFFFu));
  run("bits 0..63", {0xFFFFFFFFu, 0xFFFFFFFFu, 0, 0, 0, 0, 0, 0});
  run("bits 0..31", {0xFFFFFFFFu, 0, 0, 0, 0, 0, 0, 0});
  run("bits i%8==0", std::vector<uint32_t>(8, 0x01010101u));
  run("bits i%8<2", std::vector<uint32_t>(8, 0x03030303u));
  run("bits i%8<4", std::vector<uint32_t>(8, 0x0F0F0F0Fu));
  run("bits 64..255", {0, 0, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu});
  run("bits 0..127", {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0, 0, 0, 0});
  hipFree(d);
  return 0;
}
