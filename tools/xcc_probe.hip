// Diagnostic: which XCC (XCD) does workgroup b of a 1-D grid land on?  Prints the histogram of (b % 8) -> XCC_ID for a
// plain launch and for a grid launched while another kernel occupies the device.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
__global__ void probe(int* out, int spin) {
  unsigned x;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
  if (threadIdx.x == 0) out[blockIdx.x] = (int)(x & 0xf);
  for (volatile int i = 0; i < spin; ++i) {}
}
int main() {
  const int n = 2048;
  int* d; hipMalloc(&d, n * 4);
  for (int threads : {256, 512}) for (int lds : {0, 65536}) {
    hipMemset(d, 0xff, n * 4);
    hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipLaunchKernelGGL(probe, dim3(n), dim3(threads), lds, 0, d, 2000);
    hipDeviceSynchronize();
    std::vector<int> h(n); hipMemcpy(h.data(), d, n * 4, hipMemcpyDeviceToHost);
    int match = 0, first_mismatch = -1; int hist[8][8] = {};
    for (int b = 0; b < n; ++b) { if (h[b] == b % 8) ++match; else if (first_mismatch < 0) first_mismatch = b; if (h[b] >= 0 && h[b] < 8) hist[b % 8][h[b]]++; }
    printf("threads %d lds %d: %d / %d workgroups have XCC_ID == blockIdx %% 8 (first mismatch at %d)\n", threads, lds, match, n, first_mismatch);
    for (int r = 0; r < 8; ++r) { printf("  b%%8=%d:", r); for (int c = 0; c < 8; ++c) printf(" %4d", hist[r][c]); printf("\n"); }
  }
  return 0;
}
