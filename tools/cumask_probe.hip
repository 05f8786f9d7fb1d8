// Diagnostic: what does a hipExtStreamCreateWithCUMask bit select on MI355X (8 XCDs x 32 CUs)?  For a few masks, launch a
// grid on the masked stream and print which (XCC, SE, CU) the workgroups ran on.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <set>
#include <vector>
__global__ void probe(unsigned* out, int spin) {
  unsigned x, h;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(h));
  if (threadIdx.x == 0) { out[2 * blockIdx.x] = x; out[2 * blockIdx.x + 1] = h; }
  for (volatile int i = 0; i < spin; ++i) {}
}
int main() {
  const int n = 2048;
  unsigned* d; hipMalloc(&d, n * 8);
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  printf("multiProcessorCount %d\n", p.multiProcessorCount);
  struct M { const char* name; unsigned w[8]; };
  M masks[] = {
    {"bits 0..31", {0xffffffffu, 0, 0, 0, 0, 0, 0, 0}},
    {"bits 0..127", {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0, 0, 0, 0}},
    {"low 16 of every 32", {0xffffu, 0xffffu, 0xffffu, 0xffffu, 0xffffu, 0xffffu, 0xffffu, 0xffffu}},
    {"every 8th bit", {0x01010101u, 0x01010101u, 0x01010101u, 0x01010101u, 0x01010101u, 0x01010101u, 0x01010101u, 0x01010101u}},
    {"bits 0..7", {0xffu, 0, 0, 0, 0, 0, 0, 0}},
    {"all", {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu}},
  };
  for (auto& m : masks) {
    hipStream_t s;
    hipError_t e = hipExtStreamCreateWithCUMask(&s, 8, m.w);
    if (e != hipSuccess) { printf("%s: hipExtStreamCreateWithCUMask failed: %s\n", m.name, hipGetErrorString(e)); continue; }
    hipMemsetAsync(d, 0xff, n * 8, s);
    hipLaunchKernelGGL(probe, dim3(n), dim3(64), 0, s, d, 20000);
    hipStreamSynchronize(s);
    std::vector<unsigned> h(2 * n); hipMemcpy(h.data(), d, n * 8, hipMemcpyDeviceToHost);
    int per_xcc[16] = {};
    std::set<unsigned> cus;
    for (int b = 0; b < n; ++b) {
      unsigned x = h[2 * b] & 0xf, hw = h[2 * b + 1];
      unsigned cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
      per_xcc[x]++;
      cus.insert((x << 16) | (se << 8) | (sh << 4) | cu);
    }
    printf("%-20s: %zu distinct (xcc, se, sh, cu); workgroups per XCC:", m.name, cus.size());
    for (int x = 0; x < 8; ++x) printf(" %d", per_xcc[x]);
    printf("\n   CUs used per XCC:");
    for (int x = 0; x < 8; ++x) { int c = 0; for (unsigned k : cus) if ((k >> 16) == (unsigned)x) ++c; printf(" %d", c); }
    printf("\n");
    hipStreamDestroy(s);
  }
  return 0;
}
