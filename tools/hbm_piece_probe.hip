// Diagnostic (round 5): how fast does a GEMM-like operand stream come out of HBM as a function of the contiguous bytes per row a request
// touches?  Every CU (one 256-thread workgroup) streams its own 256-row panel of a [M][K] bf16 matrix far larger than the Infinity Cache
// by LDS-DMA (1 KiB per wave-instruction), a fixed number of requests in flight per wave, in the order a tiled GEMM would: PIECE bytes of each
// of 1024 / PIECE rows per request, K-steps of PIECE bytes.  PIECE = 128 is what every BK = 64 kernel of gemm.hip does.
#include <hip/hip_runtime.h>
#include <stdio.h>
// (the 16-requests-in-flight rows of the first version, profiles/r05_a_gemm_full_row.txt, measured the same as 8)
#include <vector>
#include <algorithm>
#define LDS_PTR(p) ((__attribute__((address_space(3))) char*)(p))
__device__ __forceinline__ void dma1(const void* p, unsigned lds) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 2\n\tglobal_load_lds_dwordx4 %2, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "s"(lds), "v"(p) : "memory", "scc");
}
template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// panel = 256 rows x K.  A "stage" = 32 KB = 32 requests = 8 per wave.  PIECE bytes per row per request -> rows per request = 1024 / PIECE;
// a stage covers 256 rows x (32768 / 256 = 128 B) when PIECE = 128, or 128 rows x 256 B, 64 rows x 512 B, 32 rows x 1 KB: same bytes, the panel is
// walked so that every byte is read exactly once.  INFLIGHT requests per wave stay outstanding.
template <int PIECE, int INFLIGHT>
__global__ __launch_bounds__(256) void stream(const char* A, int K, int panels, unsigned long long* out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const unsigned lds = (unsigned)(size_t)LDS_PTR(smem);
  constexpr int RPR = 1024 / PIECE;           // rows per request
  constexpr int LPR = PIECE / 16;             // lanes per row
  const size_t rowb = (size_t)K * 2;
  const int row_groups = 256 / (RPR * 4);     // groups of (4 waves x RPR rows) in the panel
  const int ksteps = (int)(rowb / PIECE);
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  int slot = 0;
  for (int pn = blockIdx.x; pn < panels; pn += gridDim.x) {
    const char* base = A + (size_t)pn * 256 * rowb;
    for (int ks = 0; ks < ksteps; ++ks)
      for (int rg = 0; rg < row_groups; ++rg) {
        const int row = (rg * 4 + wave) * RPR + lane / LPR;
        dma1(base + (size_t)row * rowb + (size_t)ks * PIECE + (lane % LPR) * 16, lds + (unsigned)(wave * 8 + (slot & 7)) * 1024u);
        ++slot;
        wait_vmcnt<INFLIGHT - 1>();
      }
  }
  wait_vmcnt<0>();
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
}

template <int PIECE, int INFLIGHT>
void run(const char* A, size_t M, int K, char* flush, size_t flush_bytes, unsigned long long* dout, int wgs_per_cu = 1) {
  static_assert(INFLIGHT <= 8, "8 LDS slots per wave");
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e9;
  for (int rep = 0; rep < 2; ++rep) {
    hipMemset(flush, rep, flush_bytes); hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((stream<PIECE, INFLIGHT>), dim3(256 * wgs_per_cu), dim3(256), 32 * 1024, 0, A, K, (int)(M / 256), dout);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    best = std::min(best, ms);
  }
  printf("K %5d (row stride %5d B)  piece %4d B x %2d rows per request, %2d requests in flight per wave, %d workgroups of 4 waves per CU (%3d KB per CU): %7.1f us  %5.2f TB/s\n", K, K * 2, PIECE,
         1024 / PIECE, INFLIGHT, wgs_per_cu, INFLIGHT * 4 * wgs_per_cu, best * 1e3, (double)M * K * 2 / (best * 1e-3) / 1e12);
}

int main() {
  const size_t bytes = (size_t)640 << 20;     // 640 MB: 2.5 x the Infinity Cache
  char *A, *flush; unsigned long long* dout;
  hipMalloc(&A, bytes); hipMalloc(&flush, (size_t)512 << 20); hipMalloc(&dout, 256 * 8);
  hipMemset(A, 1, bytes); hipDeviceSynchronize();
  hipMalloc(&dout, 1024 * 8);
  for (int K : {384, 1536, 3072}) {
    const size_t M = bytes / ((size_t)K * 2) / 256 * 256;
    run<128, 8>(A, M, K, flush, (size_t)512 << 20, dout);
    run<256, 8>(A, M, K, flush, (size_t)512 << 20, dout);
    run<512, 8>(A, M, K, flush, (size_t)512 << 20, dout);
    if (K * 2 >= 1024) run<1024, 8>(A, M, K, flush, (size_t)512 << 20, dout);
  }
  // more waves per CU (attention runs 16): does the stream go faster?
  for (int w : {1, 2, 4}) {
    const int K = 1152;      // the qkv row of the student (2304 B): a head slice is 128 B of it
    const size_t M = bytes / ((size_t)K * 2) / 256 * 256;
    run<128, 4>(A, M, K, flush, (size_t)512 << 20, dout, w);
    run<128, 8>(A, M, K, flush, (size_t)512 << 20, dout, w);
  }
  return 0;
}
