// Diagnostic: how many operand bytes per cycle reach one CU of MI355X, by path.  The persistent GEMMs' K loops run at the
// rate their LDS-DMA ring fills (~24 B/cycle/CU where the MFMA-bound rate needs 32).  Is that a limit of the
// TCP -> LDS DMA path, or of what one 160-KB ring keeps in flight -- i.e. would fetching one operand straight into VGPRs (MFMA
// fragment layout, no LDS) ADD bandwidth?  One 512-thread workgroup per CU walks GEMM-like tiles (256 rows x 64 k per stage):
//   A: activation panel [M][K] (78 MB: Infinity Cache / HBM), by LDS-DMA, or not at all
//   B: weight panel [N][K] (3.5 MB: L2), by LDS-DMA, by global_load_dwordx4 into VGPRs in fragment layout, or not at all
// one or two stages ahead, counted vmcnt, no barriers, no MFMAs.  Prints cycles per K-step and bytes per cycle per CU.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>
#include <type_traits>

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define LDS_PTR(p) ((__attribute__((address_space(3))) char*)(p))

__device__ __forceinline__ void dma1(const void* p, unsigned lds) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %1\n\t"
      "s_nop 2\n\t"
      "global_load_lds_dwordx4 %2, off\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "s"(lds), "v"(p)
      : "memory", "scc");
}
template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// A_MODE: 0 none, 1 LDS-DMA.  B_MODE: 0 none, 1 LDS-DMA, 2 VGPR fragment loads (every wave its own 64 columns: the two
// wave rows of the 2 x 4 wave grid fetch the same bytes, as a register-operand GEMM would).  The VGPR loads are plain
// loads the compiler tracks (an asm load whose destination the compiler believes already written can be copied or
// reallocated before the data lands); its own waits do not see the asm DMA, so they are a little stricter than DEPTH.
template <int A_MODE, int B_MODE, int DEPTH>
__global__ __launch_bounds__(512) void fill_kernel(const char* A, const char* B, int M, int N, int K, int ksteps_per_tile,
                                                   int tiles, unsigned long long* out, unsigned* sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), wn = wave & 3;
  const int tiles_m = M / 256, tiles_n = N / 256;
  constexpr int PER = 4;    // DMA instructions per wave per 32-KB stage
  constexpr int RING = DEPTH + 1;
  u32x4 acc = {0, 0, 0, 0};
  u32x4 fr[RING][8];
#pragma unroll
  for (int q = 0; q < RING; ++q)
#pragma unroll
    for (int f = 0; f < 8; ++f) fr[q][f] = (u32x4){0, 0, 0, 0};
  const unsigned lds_a = (unsigned)(size_t)LDS_PTR(smem), lds_b = lds_a + 3 * 32768;
  const int per_xcd = tiles_m * tiles_n / 8;   // the GEMM's order: workgroups b, b+8, ... share an XCD and walk one run of tiles
  const int total = tiles * ksteps_per_tile;   // a multiple of RING (host checks)
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int s0 = 0; s0 < total; s0 += RING) {
#pragma unroll
    for (int u = 0; u < RING; ++u) {           // ring slot u is a compile-time constant
      const int stage = s0 + u;
      const int t = stage / ksteps_per_tile, ks = stage - t * ksteps_per_tile;
      const int tile = (blockIdx.x & 7) * per_xcd + ((blockIdx.x >> 3) + t * (gridDim.x >> 3)) % per_xcd;
      const int tm = tile / tiles_n, tn = tile - tm * tiles_n;
      const char* a0 = A + (size_t)tm * 256 * K * 2;
      const char* b0 = B + (size_t)tn * 256 * K * 2;
      const int k0 = ks * 64;                  // < K (host checks ksteps_per_tile * 64 == K)
      if (B_MODE == 2) {
        // consume what this slot held (requested RING stages ago), then refill it
#pragma unroll
        for (int f = 0; f < 8; ++f) acc ^= fr[u][f];
#pragma unroll
        for (int f = 0; f < 8; ++f) {
          const int j = f & 3, kk = f >> 2;
          const int row = wn * 64 + j * 16 + (lane & 15);
          fr[u][f] = *(const u32x4*)(b0 + ((size_t)row * K + k0 + kk * 32 + (lane >> 4) * 8) * 2);
        }
      }
      if (A_MODE == 1) {
#pragma unroll
        for (int i = 0; i < PER; ++i) {
          const int slab = wave * PER + i, row = slab * 8 + (lane >> 3), chunk = lane & 7;
          dma1(a0 + ((size_t)row * K + k0 + chunk * 8) * 2, lds_a + (stage % 3) * 32768 + slab * 1024);
        }
      }
      if (B_MODE == 1) {
#pragma unroll
        for (int i = 0; i < PER; ++i) {
          const int slab = wave * PER + i, row = slab * 8 + (lane >> 3), chunk = lane & 7;
          dma1(b0 + ((size_t)row * K + k0 + chunk * 8) * 2, lds_b + (stage & 1) * 32768 + slab * 1024);
        }
      }
      // flow control of a ring: at most DEPTH stages' requests in flight
      constexpr int THIS = (A_MODE == 1 ? PER : 0) + (B_MODE == 1 ? PER : 0) + (B_MODE == 2 ? 8 : 0);
      wait_vmcnt<THIS * DEPTH>();
    }
  }
  wait_vmcnt<0>();
#pragma unroll
  for (int q = 0; q < RING; ++q)
#pragma unroll
    for (int f = 0; f < 8; ++f) acc ^= fr[q][f];
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (tid == 0) { out[2 * blockIdx.x] = t1 - t0; out[2 * blockIdx.x + 1] = (unsigned long long)total; }
  if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) sink[tid] = acc[0];
}

template <int A_MODE, int B_MODE, int DEPTH>
void run(const char* name, const char* A, const char* B, int M, int N, int K, unsigned long long* dout, unsigned* sink) {
  const int grid = 256, tiles = 12, ksteps = K / 64;
  if (ksteps * 64 != K || (tiles * ksteps) % (DEPTH + 1) != 0 || M % 256 || N % 256 || (M / 256) * (N / 256) / 8 < 1) { printf("bad shape\n"); return; }
  hipFuncSetAttribute((const void*)fill_kernel<A_MODE, B_MODE, DEPTH>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e9; double cyc = 0;
  for (int rep = 0; rep < 4; ++rep) {
    hipEventRecord(e0);
    if (rep == 0) printf("%s ...\n", name);
    hipLaunchKernelGGL((fill_kernel<A_MODE, B_MODE, DEPTH>), dim3(grid), dim3(512), 160 * 1024, 0, A, B, M, N, K, ksteps, tiles, dout, sink);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (hipGetLastError() != hipSuccess) { printf("%s: launch failed\n", name); return; }
    std::vector<unsigned long long> h(2 * grid); hipMemcpy(h.data(), dout, grid * 16, hipMemcpyDeviceToHost);
    std::vector<double> per(grid);
    for (int b = 0; b < grid; ++b) per[b] = (double)h[2 * b] / (double)h[2 * b + 1];
    std::sort(per.begin(), per.end());
    if (ms < best) { best = ms; cyc = per[grid / 2]; }
  }
  const double bytes = (A_MODE ? 32768.0 : 0.0) + (B_MODE == 1 ? 32768.0 : B_MODE == 2 ? 65536.0 : 0.0);
  const double stages = 12.0 * (K / 64);
  printf("%-44s %8.0f cycles/K-step (median workgroup)  %6.1f B/cycle/CU delivered   launch %7.1f us = %5.2f TB/s chip-wide\n", name, cyc,
         bytes / cyc, best * 1e3, bytes * stages * 256 / (best * 1e-3) / 1e12);
}

int main() {
  setvbuf(stdout, NULL, _IONBF, 0);
  const int M = 50688, N = 2304, K = 768;
  char *A, *B; unsigned long long* dout; unsigned* sink;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
  fprintf(stderr, "start\n"); if (getenv("FP_STEP") && atoi(getenv("FP_STEP")) == 0) return 0;
  CK(hipMalloc(&sink, 4096)); fprintf(stderr, "sink ok\n");
  CK(hipMalloc(&dout, 256 * 16)); fprintf(stderr, "out ok\n");
  CK(hipMalloc(&B, (size_t)N * K * 2)); fprintf(stderr, "B ok\n");
  CK(hipMalloc(&A, (size_t)M * K * 2)); fprintf(stderr, "A ok\n");
  printf("A %p B %p out %p sink %p\n", (void*)A, (void*)B, (void*)dout, (void*)sink);
  CK(hipMemset(A, 1, (size_t)M * K * 2)); CK(hipDeviceSynchronize()); printf("A set\n");
  CK(hipMemset(B, 2, (size_t)N * K * 2)); CK(hipDeviceSynchronize()); printf("B set\n");
  if (getenv("FILL_PROBE_HOST_ONLY")) return 0;
  printf("one 512-thread workgroup per CU, stage = 256 rows x 64 k (32 KB) per operand; 2048 MFMA cycles per K-step would need 32 B/cycle/CU\n");
  run<1, 0, 1>("A by LDS-DMA alone, 1 stage ahead", A, B, M, N, K, dout, sink);
  run<1, 0, 2>("A by LDS-DMA alone, 2 stages ahead", A, B, M, N, K, dout, sink);
  run<0, 1, 1>("B (L2) by LDS-DMA alone, 1 ahead", A, B, M, N, K, dout, sink);
  run<0, 1, 2>("B (L2) by LDS-DMA alone, 2 ahead", A, B, M, N, K, dout, sink);
  run<0, 2, 1>("B by VGPR fragment loads alone, 1 ahead", A, B, M, N, K, dout, sink);
  run<0, 2, 2>("B by VGPR fragment loads alone, 2 ahead", A, B, M, N, K, dout, sink);
  run<1, 1, 1>("A + B by LDS-DMA, 1 ahead", A, B, M, N, K, dout, sink);
  run<1, 1, 2>("A + B by LDS-DMA, 2 ahead", A, B, M, N, K, dout, sink);
  run<1, 2, 1>("A LDS-DMA + B VGPR fragments, 1 ahead", A, B, M, N, K, dout, sink);
  run<1, 2, 2>("A LDS-DMA + B VGPR fragments, 2 ahead", A, B, M, N, K, dout, sink);
  return 0;
}
