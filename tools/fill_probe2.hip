// Diagnostic (round 5): the fill pattern of the full-row 256x384 GEMM with nothing else on the CU.  One 256-thread workgroup (4 waves) per CU,
// a stage = A 256 rows x 64 k (activation panel [M][K], HBM / Infinity Cache) + B 384 rows x 64 k (weight panel [384][K], L2) = 80 KB = 20 LDS-DMA
// requests per wave.  Modes:
//   cont : every wave keeps one stage of requests in flight (issue 20, wait for the previous 20): the ceiling of a continuously fed ring
//   burst: issue 20, wait for all of them, then idle IDLE cycles (the 2-slot ring: requests only in phase 2 of a K-step)
//   half : as cont, but every request covers 16 rows x 64 bytes (a BK = 32 stage: half cache lines) instead of 8 rows x 128
// Prints cycles per stage (issue of the first request -> all landed) and bytes per cycle per CU.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>
#define LDS_PTR(p) ((__attribute__((address_space(3))) char*)(p))
__device__ __forceinline__ void dma1(const void* p, unsigned lds) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 2\n\tglobal_load_lds_dwordx4 %2, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "s"(lds), "v"(p) : "memory", "scc");
}
template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int MODE>   // 0 cont, 1 burst, 2 half
__global__ __launch_bounds__(256) void fill2(const char* A, const char* B, int K, int tiles_m, int idle, unsigned long long* out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const unsigned lds = (unsigned)(size_t)LDS_PTR(smem);
  const int nk = K / 64;
  unsigned long long busy = 0, issue = 0;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  int stage = 0;
  for (int t = 0; t < 2; ++t) {
    const int tm = (blockIdx.x + t * gridDim.x) % tiles_m;
    const char* a0 = A + (size_t)tm * 256 * K * 2;
    for (int ks = 0; ks < nk; ++ks, ++stage) {
      const unsigned slot = lds + (stage & 1) * 81920;
      const unsigned long long s0 = __builtin_amdgcn_s_memtime();
      if (MODE == 2) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {   // two half-stages of 32 k: 16 rows x 64 B per request
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int slab = wave * 4 + i, row = slab * 16 + (lane >> 2), chunk = lane & 3;
            dma1(a0 + ((size_t)row * K + ks * 64 + h * 32 + chunk * 8) * 2, slot + h * 40960 + slab * 1024);
          }
#pragma unroll
          for (int i = 0; i < 6; ++i) {
            const int slab = wave * 6 + i, row = slab * 16 + (lane >> 2), chunk = lane & 3;
            dma1(B + ((size_t)row * K + ks * 64 + h * 32 + chunk * 8) * 2, slot + h * 40960 + 16384 + slab * 1024);
          }
        }
      } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int slab = wave * 8 + i, row = slab * 8 + (lane >> 3), chunk = lane & 7;
          dma1(a0 + ((size_t)row * K + ks * 64 + chunk * 8) * 2, slot + slab * 1024);
        }
#pragma unroll
        for (int i = 0; i < 12; ++i) {
          const int slab = wave * 12 + i, row = slab * 8 + (lane >> 3), chunk = lane & 7;
          dma1(B + ((size_t)row * K + ks * 64 + chunk * 8) * 2, slot + 32768 + slab * 1024);
        }
      }
      const unsigned long long s1 = __builtin_amdgcn_s_memtime();
      issue += s1 - s0;
      if (MODE == 1) {
        wait_vmcnt<0>();
        busy += __builtin_amdgcn_s_memtime() - s0;
        const unsigned long long w0 = __builtin_amdgcn_s_memtime();
        while (__builtin_amdgcn_s_memtime() - w0 < (unsigned long long)idle) __builtin_amdgcn_s_sleep(8);
      } else {
        wait_vmcnt<20>();
      }
    }
  }
  wait_vmcnt<0>();
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (lane == 0) {
    unsigned long long* o = out + ((size_t)blockIdx.x * 4 + wave) * 4;
    o[0] = t1 - t0; o[1] = (unsigned long long)stage; o[2] = busy; o[3] = issue;
  }
}

template <int MODE>
void run(const char* name, const char* A, const char* B, int M, int K, int grid, int idle, unsigned long long* dout) {
  hipFuncSetAttribute((const void*)fill2<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e9; double tot = 0, busy = 0, iss = 0, st = 1;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((fill2<MODE>), dim3(grid), dim3(256), 160 * 1024, 0, A, B, K, M / 256, idle, dout);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (hipGetLastError() != hipSuccess) { printf("%s: launch failed\n", name); return; }
    std::vector<unsigned long long> h(grid * 16); hipMemcpy(h.data(), dout, grid * 128, hipMemcpyDeviceToHost);
    std::vector<double> a, b, c;
    for (int w = 0; w < grid * 4; ++w) { a.push_back((double)h[4 * w]); b.push_back((double)h[4 * w + 2]); c.push_back((double)h[4 * w + 3]); st = (double)h[4 * w + 1]; }
    std::sort(a.begin(), a.end()); std::sort(b.begin(), b.end()); std::sort(c.begin(), c.end());
    if (ms < best) { best = ms; tot = a[a.size() / 2]; busy = b[b.size() / 2]; iss = c[c.size() / 2]; }
  }
  const double per = (MODE == 1 ? busy : tot) / st;
  printf("%-58s grid %3d  %7.0f cycles per 80-KB stage = %5.1f B/cycle/CU   (issue of the 20 requests %6.0f cycles)   launch %7.1f us\n", name, grid, per,
         81920.0 / per, iss / st, best * 1e3);
}

int main() {
  const int M = 50688, K = 1536;
  char *A, *B; unsigned long long* dout;
  hipMalloc(&dout, 256 * 128); hipMalloc(&B, (size_t)384 * K * 2); hipMalloc(&A, (size_t)M * K * 2);
  hipMemset(A, 1, (size_t)M * K * 2); hipMemset(B, 2, (size_t)384 * K * 2); hipDeviceSynchronize();
  printf("one 256-thread workgroup per CU, stage = A 256 x 64 k + B 384 x 64 k = 80 KB, K = %d (24 stages per tile, 2 tiles per workgroup)\n", K);
  for (int grid : {198, 256}) {
    run<0>("cont : one stage of requests always in flight", A, B, M, K, grid, 0, dout);
    run<2>("half : the same in 16-row x 64-byte requests", A, B, M, K, grid, 0, dout);
    run<1>("burst: 20 requests, wait, idle 1500 cycles", A, B, M, K, grid, 1500, dout);
    run<1>("burst: 20 requests, wait, idle 0", A, B, M, K, grid, 0, dout);
  }
  return 0;
}
