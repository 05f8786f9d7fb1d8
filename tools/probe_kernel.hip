// Stand-in for a collective's kernel in tools/reserve_cus_probe.py: `wgs` workgroups of `threads` threads holding `lds` bytes
// of LDS each, a few microseconds of work (one pass over a small buffer).  When does it get onto the chip while a
// persistent GEMM grid runs?
#include <hip/hip_runtime.h>
__global__ void probe_kernel(float* buf, int n) {
  extern __shared__ float sm[];
  sm[threadIdx.x] = (float)blockIdx.x;
  __syncthreads();
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) buf[i] += sm[(threadIdx.x + 1) % blockDim.x];
}
extern "C" int probe_launch(void* buf, int n, int wgs, int threads, int lds, void* stream) {
  static bool attr = false;
  if (!attr) {
    if (hipFuncSetAttribute((const void*)probe_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 65536) != hipSuccess) return -1;
    attr = true;
  }
  hipLaunchKernelGGL(probe_kernel, dim3(wgs), dim3(threads), lds < threads * 4 ? threads * 4 : lds, (hipStream_t)stream, (float*)buf, n);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}
