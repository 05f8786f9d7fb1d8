// bf16 MFMA GEMM for gfx950 with fused epilogues.  C[M,N] = sum_k A(m,k) B(n,k).
//
// Tile 128x128x64, 256 threads = 4 waves (2x2), each wave a 64x64 sub-tile = 4x4 MFMA 16x16x32
// accumulators.  Operand tiles go HBM -> LDS by LDS-DMA (global_load_lds_dwordx4, 1 KiB per
// wave-instruction), double buffered (2 x 32 KiB), one barrier per K-step.  The LDS image is
// lane-linear, so the bank-conflict swizzle is applied to the per-lane SOURCE address and undone
// by the same XOR on the fragment read (cdna_hip_programming.md §5.4 rule 21):
//   k-contiguous operand  [128 rows][64 k]  (128-B rows):  chunk16 ^= (row >> 1) & 7   -> ds_read_b128
//   k-major operand       [64 k][128 cols]  (256-B rows):  chunk16 ^= h(k) << 1,
//                         h(k) = (k & 3) | ((k >> 3) & 1) << 2                          -> ds_read_b64_tr_b16
// Both are conflict-free for the MFMA 16x16x32 fragment maps (derivation in DESIGN.md §4.1).
// The accumulator tile is staged through LDS (reusing the operand buffers) so that every epilogue
// reads/writes global memory in whole 128/256-byte row segments.
#include <stdlib.h>

#include <type_traits>

#include "devit_common.h"

namespace {

constexpr int BK = 64;

struct GemmArgs {
  const __bf16* A;
  const __bf16* B;
  int lda, ldb;
  int a_group, a_skip, b_group, b_skip;
  long long a_bs, b_bs;
  int M, N, K;
  int tiles_m, tiles_n, split_k;
  int gn;  // n-tiles per L2 chunk: tiles are ordered chunk-major, then m, then n inside the chunk
  devit_epilogue ep;
};

__device__ __forceinline__ int phys_row(int r, int group, int skip) {
  return group > 0 ? r + skip * (r / group + 1) : r;
}

// Issue this wave's LDS-DMA loads (1 KiB each) for one operand tile of width W (128 or 256).
//   KM == false: operand stored [R][K]; `org` = &op[row0][0]; LDS image [W rows][64 k] (128-B rows)
//   KM == true : operand stored [K][R]; `org` = &op[0][col0]; LDS image [64 k][W cols] (2W-B rows)
// The address is split into a wave-uniform base that advances with k0 (SGPRs) and a per-lane 32-bit byte offset
// that is loop-invariant, so the K-loop issues `global_load_lds_dwordx4 voff, s[base]` with no per-step VALU math.
template <bool KM, int W, int NWAVES>
__device__ __forceinline__ unsigned lane_offset(int ld, int wave, int lane, int i) {
  constexpr int CNT = (W / 8) / NWAVES;
  const int slab = wave * CNT + i;
  if (!KM) {
    const int row = slab * 8 + (lane >> 3);
    const int chunk = (lane & 7) ^ ((row >> 1) & 7);
    return (unsigned)(row * ld + chunk * 8) * 2u;
  } else {
    constexpr int LPR = W / 8, RPS = 64 / LPR;  // lanes per k-row, k-rows per 1-KiB slab
    const int krow = slab * RPS + lane / LPR;
    const int h = (krow & 3) | (((krow >> 3) & 1) << 2);
    const int chunk = (lane % LPR) ^ (h << 1);
    return (unsigned)(krow * ld + chunk * 8) * 2u;
  }
}

template <bool KM, int W, int NWAVES>
__device__ __forceinline__ void stage_tile(const __bf16* org, int ld, int k0, int group, int skip,
                                           char* lds_tile, int wave, int lane) {
  constexpr int CNT = (W / 8) / NWAVES;
  if (KM && group > 0) {   // row-remapped reduction index (patch-embed wgrad): per-lane physical rows, generic path
#pragma unroll
    for (int i = 0; i < CNT; ++i) {
      constexpr int LPR = W / 8, RPS = 64 / LPR;
      const int slab = wave * CNT + i;
      const int krow = slab * RPS + lane / LPR;
      const int h = (krow & 3) | (((krow >> 3) & 1) << 2);
      const int chunk = (lane % LPR) ^ (h << 1);
      const __bf16* src = org + (size_t)phys_row(k0 + krow, group, skip) * ld + chunk * 8;
      __builtin_amdgcn_global_load_lds(GLB_PTR(src), LDS_PTR(lds_tile + slab * 1024), 16, 0, 0);
    }
    return;
  }
  const char* ubase = (const char*)org + (size_t)k0 * (KM ? (size_t)ld : (size_t)1) * 2;   // wave-uniform
#pragma unroll
  for (int i = 0; i < CNT; ++i) {
    const int slab = wave * CNT + i;
    const unsigned off = lane_offset<KM, W, NWAVES>(ld, wave, lane, i);
    __builtin_amdgcn_global_load_lds(GLB_PTR(ubase + off), LDS_PTR(lds_tile + slab * 1024), 16, 0, 0);
  }
}

// One MFMA operand fragment (16 rows/cols starting at t16 of the W-wide tile, k-step kk of 2).
template <bool KM, int W>
__device__ __forceinline__ bf16x8 read_frag(const char* tile, int t16, int kk, int lane) {
  if (!KM) {
    const int row = t16 + (lane & 15);
#ifdef DEVIT_GEMM_NOSWZ
    const int chunk = (kk * 4 + (lane >> 4));
#else
    const int chunk = (kk * 4 + (lane >> 4)) ^ ((row >> 1) & 7);
#endif
    return *(const bf16x8*)(tile + row * 128 + chunk * 16);
  } else {
    const int G = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
    const int h = q | ((G & 1) << 2);
    const int chunk = ((t16 >> 3) + (p >> 1)) ^ (h << 1);
    const int krow = kk * 32 + G * 8 + q;
    const char* a = tile + krow * (W * 2) + chunk * 16 + (p & 1) * 8;
    return cat8(lds_tr_read(a), lds_tr_read(a + 4 * (W * 2)));
  }
}

#ifdef DEVIT_GEMM_STAMPS
// Diagnostic build only (tools/build_stamps.sh): per-wave cycle totals of the K-loop segments.
__device__ unsigned long long devit_gemm_stamps[8];
__device__ __forceinline__ unsigned long long stamp() {
  unsigned long long t;
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  __builtin_amdgcn_sched_barrier(0);
  return t;
}
#define STAMP(v) const unsigned long long v = stamp()
#else
#define STAMP(v)
#endif

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// One 64x64 fp32 staging tile (this wave's) -> fused epilogue -> global memory.  Rows are processed in chunks of
// CH iterations, each chunk in two phases so that no load waits behind the stores of an earlier row (vmcnt counts
// loads and stores in order on gfx950): first the chunk's global inputs (residual / saved pre-activation /
// pos-embed rows) are fetched into registers, then its rows are computed and stored back-to-back.  The chunk loop
// stays rolled: the epilogue runs once per tile, so its code size is instruction-cache misses.
// bf16 outputs: 8 columns (16 B) per lane; fp32: 4 columns (16 B).
template <int KIND>
__device__ __forceinline__ void epilogue_pass(const devit_epilogue& ep, const float* cw, int lane, int mw, int nw,
                                              int m_lim, size_t ob) {
  constexpr bool BF16_OUT = KIND == DEVIT_EPI_STORE_BF16 || KIND == DEVIT_EPI_GELU_BF16 || KIND == DEVIT_EPI_DGELU_BF16;
  constexpr int COLS = BF16_OUT ? 8 : 4, LPR = 64 / COLS, RPI = 64 / LPR, ITERS = 64 / RPI, NV = COLS / 4, CH = 4;
  const int col = (lane % LPR) * COLS, rl = lane / LPR;
  const int n = nw + col;
  f32x4 bias[NV], cs[NV];
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    bias[v] = ep.bias ? *(const f32x4*)(ep.bias + n + v * 4) : (f32x4){0.f, 0.f, 0.f, 0.f};
    cs[v] = ep.colscale ? *(const f32x4*)(ep.colscale + n + v * 4) : (f32x4){1.f, 1.f, 1.f, 1.f};
  }
#pragma unroll 1
  for (int c0 = 0; c0 < ITERS; c0 += CH) {
    // ---- phase 1: global inputs of the chunk
    f32x4 gin[CH];
    float rsc[CH];
    bf16x8 pre[CH];
    size_t offs[CH];
#pragma unroll
    for (int u = 0; u < CH; ++u) {
      const int m = mw + (c0 + u) * RPI + rl;
      const bool ok = m < m_lim;
      size_t o = ob + (size_t)m * ep.ldc + n;
      if (KIND == DEVIT_EPI_PATCH_F32) {
        const int b = m / ep.patch_tokens, t = m - b * ep.patch_tokens, tok = ep.extra_tokens + t;
        o = ((size_t)b * (ep.patch_tokens + ep.extra_tokens) + tok) * ep.ldc + n;
        gin[u] = ok ? *(const f32x4*)(ep.pos + (size_t)tok * ep.ldc + n) : (f32x4){0.f, 0.f, 0.f, 0.f};
      }
      if (KIND == DEVIT_EPI_RESIDUAL_F32) {
        gin[u] = ok ? load_stream((const f32x4*)(ep.res + o)) : (f32x4){0.f, 0.f, 0.f, 0.f};
        rsc[u] = (ok && ep.rowscale) ? ep.rowscale[m / ep.rows_per_scale] : 1.0f;
      }
      if (KIND == DEVIT_EPI_DGELU_BF16) {
        const bf16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
        pre[u] = ok ? load_stream((const bf16x8*)((const __bf16*)ep.aux_in + o)) : z;
      }
      offs[u] = o;
    }
    // ---- phase 2: compute + store
#pragma unroll
    for (int u = 0; u < CH; ++u) {
      const int row = (c0 + u) * RPI + rl;
      const bool ok = mw + row < m_lim;
      const size_t o = offs[u];
      f32x4 v[NV];
#pragma unroll
      for (int w = 0; w < NV; ++w) v[w] = *(const f32x4*)(cw + row * 64 + col + w * 4) + bias[w];
      if (KIND == DEVIT_EPI_STORE_F32) {
        if (ok) *(f32x4*)((float*)ep.out + o) = v[0];
      } else if (KIND == DEVIT_EPI_PATCH_F32) {
        if (ok) *(f32x4*)((float*)ep.out + o) = v[0] + gin[u];
      } else if (KIND == DEVIT_EPI_RESIDUAL_F32) {
        if (ok) {
          if (ep.aux) {
            const bf16x4 pb = {f2bf(v[0][0]), f2bf(v[0][1]), f2bf(v[0][2]), f2bf(v[0][3])};
            *(bf16x4*)((__bf16*)ep.aux + o) = pb;
          }
          *(f32x4*)((float*)ep.out + o) = gin[u] + rsc[u] * v[0];
        }
      } else {
        float x[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) x[c] = v[c >> 2][c & 3];
        if (KIND == DEVIT_EPI_GELU_BF16) {
          if (ok && ep.aux) {
            const bf16x8 pb = {f2bf(x[0]), f2bf(x[1]), f2bf(x[2]), f2bf(x[3]), f2bf(x[4]), f2bf(x[5]), f2bf(x[6]), f2bf(x[7])};
            *(bf16x8*)((__bf16*)ep.aux + o) = pb;
          }
#pragma unroll
          for (int c = 0; c < 8; ++c) x[c] = gelu_fwd<false>(x[c]) * cs[c >> 2][c & 3];
        } else if (KIND == DEVIT_EPI_DGELU_BF16) {
#pragma unroll
          for (int c = 0; c < 8; ++c) x[c] = x[c] * cs[c >> 2][c & 3] * gelu_bwd<false>(bf2f(pre[u][c]));
        }
        if (ok) {
          const bf16x8 ob8 = {f2bf(x[0]), f2bf(x[1]), f2bf(x[2]), f2bf(x[3]), f2bf(x[4]), f2bf(x[5]), f2bf(x[6]), f2bf(x[7])};
          *(bf16x8*)((__bf16*)ep.out + o) = ob8;
        }
      }
    }
  }
}

// BM x BN x 64 tile, WAVES_M x WAVES_N waves (each (BM/WAVES_M) x (BN/WAVES_N)), NSTAGE-deep LDS ring filled
// by LDS-DMA.  One raw barrier per K-step; the DMA of the NSTAGE-2 newest stages stays in flight across it
// (counted vmcnt), cdna_hip_programming.md "Pipelining across barriers".
template <int BM, int BN, int WAVES_M, int WAVES_N, int NSTAGE, bool A_KM, bool B_KM, int KIND>
__global__ __launch_bounds__(WAVES_M * WAVES_N * 64) void gemm_kernel(const GemmArgs g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NWAVES = WAVES_M * WAVES_N;
  constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N, MI = WM / 16, NI = WN / 16;
  static_assert(WN == 64 && WM % 64 == 0, "wave tile must be (64 k) x 64");
  constexpr int A_TILE_BYTES = BM * BK * 2, B_TILE_BYTES = BN * BK * 2;
  constexpr int STAGE_BYTES = A_TILE_BYTES + B_TILE_BYTES;
  constexpr int PER = (BM / 8 + BN / 8) / NWAVES;  // LDS-DMA instructions per wave per stage
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;

  // XCD-aware bijective remap: blocks b, b+8, ... share an XCD (and its L2); give each XCD a
  // contiguous run of tiles, n-tile fastest, so one A row-panel is fetched from HBM once per XCD.
  const int nwg = gridDim.x;
  int w = blockIdx.x;
  {
    const int q = nwg >> 3, r = nwg & 7, xcd = w & 7, idx = w >> 3;
    w = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int per_z = g.tiles_n * g.tiles_m;
  const int zz = w / per_z;
  int tm, tn;
  {
    const int r0 = w - zz * per_z;
    const int chunk = r0 / (g.gn * g.tiles_m);          // full chunks come first
    const int r1 = r0 - chunk * g.gn * g.tiles_m;
    const int width = min(g.gn, g.tiles_n - chunk * g.gn);
    tm = r1 / width;
    tn = chunk * g.gn + r1 % width;
  }
  const int z = zz % g.split_k, bz = zz / g.split_k;
  const int m0 = tm * BM, n0 = tn * BN;
  const int nk_total = g.K / BK;
  const int kt0 = z * nk_total / g.split_k;
  const int kt1 = (z + 1) * nk_total / g.split_k;
  const int nk = kt1 - kt0;

  const __bf16* a_org = g.A + (size_t)bz * g.a_bs + (A_KM ? (size_t)m0 : (size_t)m0 * g.lda);
  const __bf16* b_org = g.B + (size_t)bz * g.b_bs + (B_KM ? (size_t)n0 : (size_t)n0 * g.ldb);

#ifdef DEVIT_GEMM_STAMPS
  const unsigned long long k_entry = stamp();
#endif
  f32x4 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  auto stage = [&](int t) {
    char* buf = smem + (t % NSTAGE) * STAGE_BYTES;
    stage_tile<A_KM, BM, NWAVES>(a_org, g.lda, (kt0 + t) * BK, g.a_group, g.a_skip, buf, wave, lane);
    stage_tile<B_KM, BN, NWAVES>(b_org, g.ldb, (kt0 + t) * BK, g.b_group, g.b_skip, buf + A_TILE_BYTES, wave, lane);
  };

#pragma unroll
  for (int s = 0; s < NSTAGE - 1; ++s)
    if (s < nk) stage(s);

#ifdef DEVIT_GEMM_STAMPS
  unsigned long long c_wait = 0, c_bar = 0, c_issue = 0, c_comp = 0;
  const unsigned long long k_loop0 = stamp();
#endif
  for (int t = 0; t < nk; ++t) {
    STAMP(s0);
    // stage t must have landed; the NSTAGE-2 stages issued after it may stay in flight
    if (t + NSTAGE - 2 < nk) wait_vmcnt<PER * (NSTAGE - 2)>();
    else wait_vmcnt<0>();
    STAMP(s1);
    __builtin_amdgcn_s_barrier();  // everyone's stage-t DMA landed; everyone finished reading stage t-1
    STAMP(s2);
#ifndef DEVIT_GEMM_DMA_MID
    if (t + NSTAGE - 1 < nk) stage(t + NSTAGE - 1);  // overwrites the buffer read at step t-1
#endif
    STAMP(s3);
    const char* cur = smem + (t % NSTAGE) * STAGE_BYTES;
#ifdef DEVIT_GEMM_NOCOMPUTE
    if (g.K < 0)
#endif
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
#ifdef DEVIT_GEMM_DMA_MID
      if (kk == 1) {   // issue the refill between the two MFMA bursts: the SIMD's other wave is usually mid-burst
        __builtin_amdgcn_sched_barrier(0);
        if (t + NSTAGE - 1 < nk) stage(t + NSTAGE - 1);
        __builtin_amdgcn_sched_barrier(0);
      }
#endif
      bf16x8 af[MI], bfr[NI];
#pragma unroll
      for (int j = 0; j < NI; ++j) bfr[j] = read_frag<B_KM, BN>(cur + A_TILE_BYTES, wn * WN + j * 16, kk, lane);
#pragma unroll
      for (int i = 0; i < MI; ++i) af[i] = read_frag<A_KM, BM>(cur, wm * WM + i * 16, kk, lane);
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] = mfma16(af[i], bfr[j], acc[i][j]);
    }
#ifdef DEVIT_GEMM_STAMPS
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");   // let the last MFMAs drain before the stamp
    STAMP(s4);
    c_wait += s1 - s0; c_bar += s2 - s1; c_issue += s3 - s2; c_comp += s4 - s3;
#endif
  }
#ifdef DEVIT_GEMM_STAMPS
  const unsigned long long k_loop1 = stamp();
  if (lane == 0) {
    atomicAdd(&devit_gemm_stamps[0], c_wait); atomicAdd(&devit_gemm_stamps[1], c_bar);
    atomicAdd(&devit_gemm_stamps[2], c_issue); atomicAdd(&devit_gemm_stamps[3], c_comp);
    atomicAdd(&devit_gemm_stamps[4], k_loop0 - k_entry); atomicAdd(&devit_gemm_stamps[5], k_loop1 - k_loop0);
  }
#endif
  __syncthreads();  // all fragment reads done before the ring is reused as the epilogue staging area

  // ---- epilogue: accumulators -> this wave's private 64x64 f32 LDS tile -> row-wise global I/O,
  //      one pass per 64 rows of the wave tile
  float* cw = (float*)smem + wave * 4096;
  const devit_epilogue& ep = g.ep;
  const size_t ob = (size_t)bz * ep.out_batch_stride;
  const int m_lim = ep.m_valid > 0 ? ep.m_valid : g.M;
  const int nw = n0 + wn * WN;
  auto do_pass = [&](auto pass_c) {
    constexpr int pass = decltype(pass_c)::value;

#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        cw[(i * 16 + (lane >> 4) * 4 + r) * 64 + j * 16 + (lane & 15)] = acc[pass * 4 + i][j][r];
  // (same wave wrote and reads: the compiler's lgkmcnt wait orders them; no barrier needed)
  const int mw = m0 + wm * WM + pass * 64;

  if constexpr (KIND == DEVIT_EPI_ATOMIC_F32) {
    float* out = (float*)ep.out + ob;
    for (int row = 0; row < 64; ++row) {
      const float v = cw[row * 64 + lane];
      if (mw + row < m_lim) unsafeAtomicAdd(out + (size_t)(mw + row) * ep.ldc + nw + lane, v);
    }
  } else {
    epilogue_pass<KIND>(ep, cw, lane, mw, nw, m_lim, ob);
  }
  };
  do_pass(std::integral_constant<int, 0>());
  if constexpr (MI > 4) do_pass(std::integral_constant<int, 1>());

#ifdef DEVIT_GEMM_STAMPS
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const unsigned long long k_end = stamp();
  if (lane == 0) atomicAdd(&devit_gemm_stamps[6], k_end - k_loop1);
#endif
}

#ifdef DEVIT_GEMM_STAMPS
}  // namespace
extern "C" int devit_debug_gemm_stamps(unsigned long long* out4, int reset) {
  hipMemcpyFromSymbol(out4, HIP_SYMBOL(devit_gemm_stamps), 64);
  if (reset) { unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0}; hipMemcpyToSymbol(HIP_SYMBOL(devit_gemm_stamps), z, 64); }
  return 0;
}
namespace {
#endif

bool aligned16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

}  // namespace

extern "C" int devit_gemm_bf16(const devit_operand* Aop, const devit_operand* Bop, int M, int N, int K, int batch,
                               int split_k, const devit_epilogue* ep, void* stream) {
  DEVIT_CHECK(Aop && Bop && Aop->ptr && Bop->ptr && ep && ep->out, DEVIT_ERR_ARG, "devit_gemm_bf16: null pointer");
  const void* A = Aop->ptr;
  const void* B = Bop->ptr;
  const int lda = Aop->ld, ldb = Bop->ld, a_kmajor = Aop->kmajor, b_kmajor = Bop->kmajor;
  DEVIT_CHECK(M > 0 && N > 0 && K > 0 && M % 128 == 0 && N % 128 == 0 && K % BK == 0 && batch >= 1, DEVIT_ERR_SHAPE,
              "devit_gemm_bf16: M=%d N=%d K=%d must be multiples of %d/%d/%d", M, N, K, 128, 128, BK);
  DEVIT_CHECK(lda % 8 == 0 && ldb % 8 == 0 && aligned16(A) && aligned16(B) && aligned16(ep->out) &&
                  ep->ldc % 8 == 0 && Aop->batch_stride % 8 == 0 && Bop->batch_stride % 8 == 0 &&
                  ep->out_batch_stride % 8 == 0,
              DEVIT_ERR_ARG, "devit_gemm_bf16: pointers / strides must be 16-byte aligned");
  DEVIT_CHECK(ep->kind >= DEVIT_EPI_STORE_BF16 && ep->kind <= DEVIT_EPI_STORE_F32, DEVIT_ERR_ARG,
              "devit_gemm_bf16: bad epilogue kind %d", ep->kind);
  DEVIT_CHECK(split_k >= 1 && (split_k == 1 || ep->kind == DEVIT_EPI_ATOMIC_F32) && split_k <= K / BK,
              DEVIT_ERR_ARG, "devit_gemm_bf16: split_k=%d only with ATOMIC_F32 and <= K/64", split_k);
  DEVIT_CHECK((a_kmajor ? lda >= M : lda >= K) && (b_kmajor ? ldb >= N : ldb >= K), DEVIT_ERR_ARG,
              "devit_gemm_bf16: leading dimension too small");
  DEVIT_CHECK((a_kmajor || Aop->row_group == 0) && (b_kmajor || Bop->row_group == 0), DEVIT_ERR_ARG,
              "devit_gemm_bf16: row_group/skip only for k-major operands");
  if (ep->kind == DEVIT_EPI_RESIDUAL_F32)
    DEVIT_CHECK(ep->res && (!ep->rowscale || ep->rows_per_scale > 0), DEVIT_ERR_ARG, "RESIDUAL: res / rows_per_scale");
  if (ep->kind == DEVIT_EPI_PATCH_F32)
    DEVIT_CHECK(ep->pos && ep->patch_tokens > 0 && (ep->m_valid > 0 ? ep->m_valid : M) % ep->patch_tokens == 0 && batch == 1,
                DEVIT_ERR_ARG,
                "PATCH: pos / tokens");
  if (ep->kind == DEVIT_EPI_DGELU_BF16) DEVIT_CHECK(ep->aux_in != nullptr, DEVIT_ERR_ARG, "DGELU: aux_in");
  DEVIT_CHECK(ep->exact_gelu == 0, DEVIT_ERR_ARG, "devit_gemm_bf16: exact_gelu=1 (erff) is not built; the fused GELU uses a "
              "1.5e-7-accurate erf");

  GemmArgs g;
  g.A = (const __bf16*)A; g.B = (const __bf16*)B;
  g.lda = lda; g.ldb = ldb;
  g.a_group = Aop->row_group; g.a_skip = Aop->row_skip; g.b_group = Bop->row_group; g.b_skip = Bop->row_skip;
  g.a_bs = Aop->batch_stride; g.b_bs = Bop->batch_stride;
  g.M = M; g.N = N; g.K = K;
  g.tiles_m = 0; g.tiles_n = 0; g.split_k = split_k;
  {
    // keep one chunk of B (gn * 128 rows x K) around 1 MiB so it stays in the XCD's 4 MiB L2 while A streams
    static const int gn_env = getenv("DEVIT_GEMM_GN") ? atoi(getenv("DEVIT_GEMM_GN")) : 0;
    int gn = gn_env > 0 ? gn_env : (int)((1 << 20) / ((long long)128 * K * 2));
    if (gn < 1) gn = 1;
    g.gn = gn;
  }
  g.ep = *ep;
  // tile choice: 256x256 (8 waves of 128x64, 2-deep ring) when both dims allow, else 256x128 (8 waves of
  // 64x64, 3-deep ring), else 128x128 (4 waves)
  const int variant = (a_kmajor ? 2 : 0) + (b_kmajor ? 1 : 0);
  static const int force = getenv("DEVIT_GEMM_TILE") ? atoi(getenv("DEVIT_GEMM_TILE")) : 0;  // 1: 128x128, 2: 256x128
  // Measured on the step's shapes (tools/gemm_tiles.py, M = 50688): 256x256 (one workgroup per CU) wins only when the
  // K loop is long enough to amortise its un-overlapped prologue/epilogue -- teacher qkv (K 768, plain store) 902 vs
  // 770 TFLOP/s, fc2 (K 3072) 735 vs 675 -- while 128x128 (two workgroups per CU, one's epilogue under the other's
  // MFMAs) wins every short-K or epilogue-heavy shape (student qkv 681 vs 624, fc1+GELU 391 vs 369, fc2 558 vs 496,
  // teacher fc1+GELU 659 vs 629, proj 475 vs 429, fc1 wgrad 640 vs 536).  256x128 won nowhere (kept for experiments).
  const bool light_epi = ep->kind == DEVIT_EPI_STORE_BF16 || ep->kind == DEVIT_EPI_STORE_F32;
  int cfg = 1;
  if (M % 256 == 0 && N % 256 == 0 && (K >= 1536 || (K >= 768 && light_epi)) && variant != 3) cfg = 3;
  static const int wg_cfg = getenv("DEVIT_GEMM_WGRAD_TILE") ? atoi(getenv("DEVIT_GEMM_WGRAD_TILE")) : 0;
  if (variant == 3 && wg_cfg > 0) cfg = wg_cfg;
  if (force > 0 && force < cfg) cfg = force;
  const int bm = cfg == 1 ? 128 : 256, bn = cfg == 3 ? 256 : 128;
  g.tiles_m = M / bm;
  g.tiles_n = N / bn;
  if (g.gn > g.tiles_n) g.gn = g.tiles_n;
  const long long nwg = (long long)g.tiles_m * g.tiles_n * split_k * batch;
  DEVIT_CHECK(nwg < (1ll << 31), DEVIT_ERR_SHAPE, "devit_gemm_bf16: grid too large");
  hipStream_t s = (hipStream_t)stream;
#define DEVIT_LAUNCH_ONE(BM_, BN_, WMM_, WNN_, NS_, AKM_, BKM_, KIND_)                                          \
  do {                                                                                                         \
    constexpr int ring = NS_ * (BM_ + BN_) * 128, stagebytes = WMM_ * WNN_ * 16384;                            \
    constexpr int lds = ring > stagebytes ? ring : stagebytes;                                                 \
    static bool attr = false;                                                                                  \
    if (!attr) {                                                                                               \
      hipError_t e = hipFuncSetAttribute((const void*)gemm_kernel<BM_, BN_, WMM_, WNN_, NS_, AKM_, BKM_, KIND_>, \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, lds);                     \
      DEVIT_CHECK(e == hipSuccess, DEVIT_ERR_LAUNCH, "hipFuncSetAttribute: %s", hipGetErrorString(e));         \
      attr = true;                                                                                             \
    }                                                                                                          \
    hipLaunchKernelGGL((gemm_kernel<BM_, BN_, WMM_, WNN_, NS_, AKM_, BKM_, KIND_>), dim3((unsigned)nwg),       \
                       dim3(WMM_* WNN_ * 64), lds, s, g);                                                      \
  } while (0)
  // the (layout, epilogue) pairs the DeViT path uses; anything else is DEVIT_ERR_ARG
#define DEVIT_LAUNCH_GEMM(BM_, BN_, WMM_, WNN_, NS_)                                                           \
  do {                                                                                                         \
    const int key = variant * 16 + ep->kind;                                                                   \
    switch (key) {                                                                                             \
      case 0 * 16 + DEVIT_EPI_STORE_BF16: DEVIT_LAUNCH_ONE(BM_, BN_, WMM_, WNN_, NS_, false, false, DEVIT_EPI_STORE_BF16); break;       \
      case 0 * 16 + DEVIT_EPI_STORE_F32: DEVIT_LAUNCH_ONE(BM_, BN_, WMM_, WNN_, NS_, false, false, DEVIT_EPI_STORE_F32); break;         \
      case 0 * 16 + DEVIT_EPI_GELU_BF16: DEVIT_LAUNCH_ONE(BM_, BN_, WMM_, WNN_, NS_, false, false, DEVIT_EPI_GELU_BF16); break;         \
      case 0 * 16 + DEVIT_EPI_RESIDUAL_F32: DEVIT_LAUNCH_ONE(BM_, BN_, WMM_, WNN_, NS_, false, false, DEVIT_EPI_RESIDUAL_F32); break;   \
      case 0 * 16 + DEVIT_EPI_PATCH_F32: DEVIT_LAUNCH_ONE(BM_, BN_, WMM_, WNN_, NS_, false, false, DEVIT_EPI_PATCH_F32); break;         \
      case 1 * 16 + DEVIT_EPI_STORE_BF16: DEVIT_LAUNCH_ONE(BM_, BN_, WMM_, WNN_, NS_, false, true, DEVIT_EPI_STORE_BF16); break;        \
      case 1 * 16 + DEVIT_EPI_STORE_F32: DEVIT_LAUNCH_ONE(BM_, BN_, WMM_, WNN_, NS_, false, true, DEVIT_EPI_STORE_F32); break;          \
      case 1 * 16 + DEVIT_EPI_DGELU_BF16: DEVIT_LAUNCH_ONE(BM_, BN_, WMM_, WNN_, NS_, false, true, DEVIT_EPI_DGELU_BF16); break;        \
      case 3 * 16 + DEVIT_EPI_ATOMIC_F32: DEVIT_LAUNCH_ONE(BM_, BN_, WMM_, WNN_, NS_, true, true, DEVIT_EPI_ATOMIC_F32); break;         \
      case 3 * 16 + DEVIT_EPI_STORE_F32: DEVIT_LAUNCH_ONE(BM_, BN_, WMM_, WNN_, NS_, true, true, DEVIT_EPI_STORE_F32); break;           \
      default:                                                                                                 \
        DEVIT_CHECK(false, DEVIT_ERR_ARG, "devit_gemm_bf16: layout %d with epilogue %d is not instantiated", variant, ep->kind); \
    }                                                                                                          \
  } while (0)
  if (cfg == 3) DEVIT_LAUNCH_GEMM(256, 256, 2, 4, 2);
  else if (cfg == 2) DEVIT_LAUNCH_GEMM(256, 128, 4, 2, 3);
  else DEVIT_LAUNCH_GEMM(128, 128, 2, 2, 2);
#undef DEVIT_LAUNCH_ONE
#undef DEVIT_LAUNCH_GEMM
  DEVIT_LAUNCH_CHECK();
  return DEVIT_OK;
}
