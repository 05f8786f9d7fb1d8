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
#include "devit_common.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE_BYTES = 128 * 64 * 2;  // 16 KiB, either layout
constexpr int STAGE_BYTES = 2 * TILE_BYTES;

struct GemmArgs {
  const __bf16* A;
  const __bf16* B;
  int lda, ldb;
  int a_group, a_skip, b_group, b_skip;
  long long a_bs, b_bs;
  int M, N, K;
  int tiles_m, tiles_n, split_k;
  devit_epilogue ep;
};

__device__ __forceinline__ int phys_row(int r, int group, int skip) {
  return group > 0 ? r + skip * (r / group + 1) : r;
}

// Issue the 4 LDS-DMA loads of this wave for one operand tile.
//   KM == false: operand stored [R][K]; `org` = &op[row0][0], tile = rows row0..+127, k = k0..+63
//   KM == true : operand stored [K][R]; `org` = &op[0][col0], tile = k rows k0..+63, cols col0..+127
template <bool KM>
__device__ __forceinline__ void stage_tile(const __bf16* org, int ld, int k0, int group, int skip,
                                           char* lds_tile, int wave, int lane) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int slab = wave * 4 + i;
    const __bf16* src;
    if (!KM) {
      const int row = slab * 8 + (lane >> 3);
      const int chunk = (lane & 7) ^ ((row >> 1) & 7);
      src = org + (size_t)row * ld + k0 + chunk * 8;
    } else {
      const int krow = slab * 4 + (lane >> 4);
      const int h = (krow & 3) | (((krow >> 3) & 1) << 2);
      const int chunk = (lane & 15) ^ (h << 1);
      src = org + (size_t)phys_row(k0 + krow, group, skip) * ld + chunk * 8;
    }
    __builtin_amdgcn_global_load_lds(GLB_PTR(src), LDS_PTR(lds_tile + slab * 1024), 16, 0, 0);
  }
}

// One MFMA operand fragment (16 rows/cols starting at t16 of the 128-wide tile, k-step kk of 2).
template <bool KM>
__device__ __forceinline__ bf16x8 read_frag(const char* tile, int t16, int kk, int lane) {
  if (!KM) {
    const int row = t16 + (lane & 15);
    const int chunk = (kk * 4 + (lane >> 4)) ^ ((row >> 1) & 7);
    return *(const bf16x8*)(tile + row * 128 + chunk * 16);
  } else {
    const int G = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
    const int h = q | ((G & 1) << 2);
    const int chunk = ((t16 >> 3) + (p >> 1)) ^ (h << 1);
    const int krow = kk * 32 + G * 8 + q;
    const char* a = tile + krow * 256 + chunk * 16 + (p & 1) * 8;
    return cat8(lds_tr_read(a), lds_tr_read(a + 4 * 256));
  }
}

template <bool A_KM, bool B_KM>
__global__ __launch_bounds__(256, 2) void gemm_kernel(const GemmArgs g) {
  __shared__ __attribute__((aligned(16))) char smem[2 * STAGE_BYTES];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;

  // XCD-aware bijective remap: blocks b, b+8, ... share an XCD (and its L2); give each XCD a
  // contiguous run of tiles, n-tile fastest, so one A row-panel is fetched from HBM once per XCD.
  const int nwg = gridDim.x;
  int w = blockIdx.x;
  {
    const int q = nwg >> 3, r = nwg & 7, xcd = w & 7, idx = w >> 3;
    w = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int tn = w % g.tiles_n;
  const int tm = (w / g.tiles_n) % g.tiles_m;
  const int zz = w / (g.tiles_n * g.tiles_m);
  const int z = zz % g.split_k, bz = zz / g.split_k;
  const int m0 = tm * BM, n0 = tn * BN;
  const int nk_total = g.K / BK;
  const int kt0 = z * nk_total / g.split_k;
  const int kt1 = (z + 1) * nk_total / g.split_k;
  const int nk = kt1 - kt0;

  const __bf16* a_org = g.A + (size_t)bz * g.a_bs + (A_KM ? (size_t)m0 : (size_t)m0 * g.lda);
  const __bf16* b_org = g.B + (size_t)bz * g.b_bs + (B_KM ? (size_t)n0 : (size_t)n0 * g.ldb);

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  if (nk > 0) {
    stage_tile<A_KM>(a_org, g.lda, kt0 * BK, g.a_group, g.a_skip, smem, wave, lane);
    stage_tile<B_KM>(b_org, g.ldb, kt0 * BK, g.b_group, g.b_skip, smem + TILE_BYTES, wave, lane);
  }
  __syncthreads();  // hipcc drains the LDS-DMA (vmcnt(0)) before the barrier

  for (int t = 0; t < nk; ++t) {
    char* cur = smem + (t & 1) * STAGE_BYTES;
    if (t + 1 < nk) {
      char* nxt = smem + ((t + 1) & 1) * STAGE_BYTES;
      stage_tile<A_KM>(a_org, g.lda, (kt0 + t + 1) * BK, g.a_group, g.a_skip, nxt, wave, lane);
      stage_tile<B_KM>(b_org, g.ldb, (kt0 + t + 1) * BK, g.b_group, g.b_skip, nxt + TILE_BYTES, wave, lane);
    }
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      bf16x8 af[4], bfr[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) af[i] = read_frag<A_KM>(cur, wm * 64 + i * 16, kk, lane);
#pragma unroll
      for (int j = 0; j < 4; ++j) bfr[j] = read_frag<B_KM>(cur + TILE_BYTES, wn * 64 + j * 16, kk, lane);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = mfma16(af[i], bfr[j], acc[i][j]);
    }
    __syncthreads();
  }

  // ---- epilogue: accumulators -> this wave's private 64x64 f32 LDS tile -> row-wise global I/O
  float* cw = (float*)smem + wave * 4096;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        cw[(i * 16 + (lane >> 4) * 4 + r) * 64 + j * 16 + (lane & 15)] = acc[i][j][r];
  // (same wave wrote and reads: the compiler's lgkmcnt wait orders them; no barrier needed)

  const devit_epilogue& ep = g.ep;
  const int mw = m0 + wm * 64, nw = n0 + wn * 64;
  const size_t ob = (size_t)bz * ep.out_batch_stride;
  const int m_lim = ep.m_valid > 0 ? ep.m_valid : g.M;

  if (ep.kind == DEVIT_EPI_ATOMIC_F32) {
    float* out = (float*)ep.out + ob;
    for (int row = 0; row < 64; ++row) {
      const float v = cw[row * 64 + lane];
      if (mw + row < m_lim) unsafeAtomicAdd(out + (size_t)(mw + row) * ep.ldc + nw + lane, v);
    }
    return;
  }

  const int col = (lane & 15) * 4;
  const int n = nw + col;
  f32x4 bias = {0.f, 0.f, 0.f, 0.f}, cs = {1.f, 1.f, 1.f, 1.f};
  if (ep.bias) bias = *(const f32x4*)(ep.bias + n);
  if (ep.colscale) cs = *(const f32x4*)(ep.colscale + n);

#pragma unroll 4
  for (int it = 0; it < 16; ++it) {
    const int row = it * 4 + (lane >> 4);
    const int m = mw + row;
    if (m >= m_lim) continue;
    f32x4 v = *(const f32x4*)(cw + row * 64 + col);
    v += bias;
    const size_t o = ob + (size_t)m * ep.ldc + n;
    switch (ep.kind) {
      case DEVIT_EPI_STORE_BF16: {
        bf16x4 ob = {f2bf(v[0]), f2bf(v[1]), f2bf(v[2]), f2bf(v[3])};
        *(bf16x4*)((__bf16*)ep.out + o) = ob;
      } break;
      case DEVIT_EPI_STORE_F32: {
        *(f32x4*)((float*)ep.out + o) = v;
      } break;
      case DEVIT_EPI_GELU_BF16: {
        if (ep.aux) {
          bf16x4 pb = {f2bf(v[0]), f2bf(v[1]), f2bf(v[2]), f2bf(v[3])};
          *(bf16x4*)((__bf16*)ep.aux + o) = pb;
        }
        f32x4 a;
        if (ep.exact_gelu) {
#pragma unroll
          for (int c = 0; c < 4; ++c) a[c] = gelu_fwd<true>(v[c]) * cs[c];
        } else {
#pragma unroll
          for (int c = 0; c < 4; ++c) a[c] = gelu_fwd<false>(v[c]) * cs[c];
        }
        bf16x4 ob = {f2bf(a[0]), f2bf(a[1]), f2bf(a[2]), f2bf(a[3])};
        *(bf16x4*)((__bf16*)ep.out + o) = ob;
      } break;
      case DEVIT_EPI_DGELU_BF16: {
        const bf16x4 pre = *(const bf16x4*)((const __bf16*)ep.aux_in + o);
        f32x4 a;
        if (ep.exact_gelu) {
#pragma unroll
          for (int c = 0; c < 4; ++c) a[c] = v[c] * cs[c] * gelu_bwd<true>(bf2f(pre[c]));
        } else {
#pragma unroll
          for (int c = 0; c < 4; ++c) a[c] = v[c] * cs[c] * gelu_bwd<false>(bf2f(pre[c]));
        }
        bf16x4 ob = {f2bf(a[0]), f2bf(a[1]), f2bf(a[2]), f2bf(a[3])};
        *(bf16x4*)((__bf16*)ep.out + o) = ob;
      } break;
      case DEVIT_EPI_RESIDUAL_F32: {
        if (ep.aux) {
          bf16x4 pb = {f2bf(v[0]), f2bf(v[1]), f2bf(v[2]), f2bf(v[3])};
          *(bf16x4*)((__bf16*)ep.aux + o) = pb;
        }
        const float rs = ep.rowscale ? ep.rowscale[m / ep.rows_per_scale] : 1.0f;
        const f32x4 r = *(const f32x4*)(ep.res + o);
        *(f32x4*)((float*)ep.out + o) = r + rs * v;
      } break;
      case DEVIT_EPI_PATCH_F32: {
        const int b = m / ep.patch_tokens, t = m - b * ep.patch_tokens;
        const int tok = ep.extra_tokens + t;
        const f32x4 pe = *(const f32x4*)(ep.pos + (size_t)tok * ep.ldc + n);
        const size_t orow = (size_t)b * (ep.patch_tokens + ep.extra_tokens) + tok;
        *(f32x4*)((float*)ep.out + orow * ep.ldc + n) = v + pe;
      } break;
      default:
        break;
    }
  }
}

bool aligned16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

}  // namespace

extern "C" int devit_gemm_bf16(const devit_operand* Aop, const devit_operand* Bop, int M, int N, int K, int batch,
                               int split_k, const devit_epilogue* ep, void* stream) {
  DEVIT_CHECK(Aop && Bop && Aop->ptr && Bop->ptr && ep && ep->out, DEVIT_ERR_ARG, "devit_gemm_bf16: null pointer");
  const void* A = Aop->ptr;
  const void* B = Bop->ptr;
  const int lda = Aop->ld, ldb = Bop->ld, a_kmajor = Aop->kmajor, b_kmajor = Bop->kmajor;
  DEVIT_CHECK(M > 0 && N > 0 && K > 0 && M % BM == 0 && N % BN == 0 && K % BK == 0 && batch >= 1, DEVIT_ERR_SHAPE,
              "devit_gemm_bf16: M=%d N=%d K=%d must be multiples of %d/%d/%d", M, N, K, BM, BN, BK);
  DEVIT_CHECK(lda % 8 == 0 && ldb % 8 == 0 && aligned16(A) && aligned16(B) && aligned16(ep->out) &&
                  ep->ldc % 4 == 0 && Aop->batch_stride % 8 == 0 && Bop->batch_stride % 8 == 0 &&
                  ep->out_batch_stride % 4 == 0,
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

  GemmArgs g;
  g.A = (const __bf16*)A; g.B = (const __bf16*)B;
  g.lda = lda; g.ldb = ldb;
  g.a_group = Aop->row_group; g.a_skip = Aop->row_skip; g.b_group = Bop->row_group; g.b_skip = Bop->row_skip;
  g.a_bs = Aop->batch_stride; g.b_bs = Bop->batch_stride;
  g.M = M; g.N = N; g.K = K;
  g.tiles_m = M / BM; g.tiles_n = N / BN; g.split_k = split_k;
  g.ep = *ep;
  const long long nwg = (long long)g.tiles_m * g.tiles_n * split_k * batch;
  DEVIT_CHECK(nwg < (1ll << 31), DEVIT_ERR_SHAPE, "devit_gemm_bf16: grid too large");
  dim3 grid((unsigned)nwg), block(256);
  hipStream_t s = (hipStream_t)stream;
  if (!a_kmajor && !b_kmajor)
    hipLaunchKernelGGL((gemm_kernel<false, false>), grid, block, 0, s, g);
  else if (!a_kmajor && b_kmajor)
    hipLaunchKernelGGL((gemm_kernel<false, true>), grid, block, 0, s, g);
  else if (a_kmajor && b_kmajor)
    hipLaunchKernelGGL((gemm_kernel<true, true>), grid, block, 0, s, g);
  else
    hipLaunchKernelGGL((gemm_kernel<true, false>), grid, block, 0, s, g);
  DEVIT_LAUNCH_CHECK();
  return DEVIT_OK;
}
