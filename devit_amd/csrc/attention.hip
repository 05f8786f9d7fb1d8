// Fused multi-head attention forward / backward for ViT token counts (N <= 208, head_dim = 64).
// One workgroup (4 waves) per (image, head): the whole head's K and V live in LDS, the N x N
// score matrix never reaches HBM.  Replaces models/de_vit.py:68-79 (q k^T * scale -> softmax ->
// @ v -> transpose -> head gate) and its autograd backward.
//
// qkv layout = output of the qkv GEMM: row (b, n), feature j*D + h*64 + e (j = q,k,v)
// (models/de_vit.py:67 reshape(B,N,3,H,hd)); out layout [B*N][D] with feature h*64 + e (:74,:81).
//
// LDS images are [rows][64] bf16 (128-byte rows); 16-byte chunk c of row r is stored at chunk
// c ^ ((r >> 1) & 7): conflict-free ds_read_b128 row fragments, 2-way ds_read_b64_tr_b16.
#include "devit_common.h"

namespace {

constexpr int HD = 64;            // head dim
constexpr int MAXT = 13;          // 16-row tiles: N <= 208
constexpr int KROWS = 224;        // 7 k-steps of 32
constexpr int PSTRIDE = 232;      // bf16 elements per P / dS row (464 B, 16-B multiple)
constexpr int IMG_BYTES = KROWS * HD * 2;  // 28672

__device__ __forceinline__ int img_off(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }

// MFMA fragment with 8 consecutive d (k = d) for rows r0 + (lane & 15): A of Q K^T, B = K rows, ...
__device__ __forceinline__ bf16x8 img_row_frag(const char* img, int r0, int kk, int lane) {
  return *(const bf16x8*)(img + img_off(r0 + (lane & 15), kk * 4 + (lane >> 4)));
}
// MFMA fragment whose k index is the image ROW (k0..k0+31) and whose row/col index is d (c0..c0+15).
__device__ __forceinline__ bf16x8 img_tr_frag(const char* img, int k0, int c0, int lane) {
  const int G = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
  const int r = k0 + G * 8 + q, ch = (c0 >> 3) + (p >> 1), sub = (p & 1) * 8;
  return cat8(lds_tr_read(img + img_off(r, ch) + sub), lds_tr_read(img + img_off(r + 4, ch) + sub));
}
// P / dS buffers: [rows][PSTRIDE] bf16, unswizzled.
__device__ __forceinline__ bf16x8 pbuf_row_frag(const char* buf, int r0, int ks, int lane) {
  return *(const bf16x8*)(buf + (r0 + (lane & 15)) * (PSTRIDE * 2) + (ks * 32 + (lane >> 4) * 8) * 2);
}
__device__ __forceinline__ bf16x8 pbuf_tr_frag(const char* buf, int k0, int c0, int lane) {
  const int G = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
  const char* a = buf + (k0 + G * 8 + q) * (PSTRIDE * 2) + (c0 + p * 4) * 2;
  return cat8(lds_tr_read(a), lds_tr_read(a + 4 * PSTRIDE * 2));
}

// Rows [0, KROWS) of a strided [N][64] bf16 matrix -> registers -> an LDS image (rows >= N zero).  Only the backward's dO / O
// rows still travel this way (delta needs them in registers); every other image arrives by LDS-DMA, see dma_image().
template <int NT>
struct RowRegs {
  static constexpr int ITERS = (KROWS * 8 + NT - 1) / NT;
  bf16x8 v[ITERS];
};
template <int NT>
__device__ __forceinline__ void fetch_rows(RowRegs<NT>& r, const __bf16* src, size_t row_stride, int N, int tid) {
#pragma unroll
  for (int it = 0; it < RowRegs<NT>::ITERS; ++it) {
    const int idx = tid + it * NT, row = idx >> 3, c = idx & 7;
    const bf16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
    r.v[it] = row < N ? *(const bf16x8*)(src + (size_t)row * row_stride + c * 8) : z;
  }
}
template <int NT>
__device__ __forceinline__ void put_image(char* img, const RowRegs<NT>& r, float mul, int tid) {
#pragma unroll
  for (int it = 0; it < RowRegs<NT>::ITERS; ++it) {
    const int idx = tid + it * NT, row = idx >> 3, c = idx & 7;
    bf16x8 v = r.v[it];
    if (mul != 1.0f) {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = f2bf(bf2f(v[e]) * mul);
    }
    if (row < KROWS) *(bf16x8*)(img + img_off(row, c)) = v;
  }
}

// Rows [0, KROWS) of a strided [N][64] 16-bit matrix -> an LDS image by LDS-DMA: one wave-instruction moves 8 rows x 128 B
// (1 KiB) into consecutive LDS bytes, so the image's chunk swizzle goes on the per-lane SOURCE address (lane l of slab s
// writes row 8 s + l / 8, physical chunk l % 8, which must hold logical chunk (l % 8) ^ swizzle(row)).  Rows >= N cannot
// be zero-filled by a DMA: they repeat row N - 1 (finite values; every use of a padded key or query is masked to P = 0).
template <int NWAVES>
__device__ __forceinline__ void dma_image(char* img, const __bf16* src, size_t row_stride, int N, int wave, int lane) {
#pragma unroll
  for (int it = 0; it < (KROWS / 8 + NWAVES - 1) / NWAVES; ++it) {
    const int slab = wave + it * NWAVES;
    if (slab < KROWS / 8) {
      const int row = slab * 8 + (lane >> 3), c = (lane & 7) ^ ((row >> 1) & 7);
      const __bf16* g = src + (size_t)min(row, N - 1) * row_stride + c * 8;
      __builtin_amdgcn_global_load_lds(GLB_PTR(g), LDS_PTR(img + slab * 1024), 16, 0, 0);
    }
  }
}

// Query side and key side are described separately: the packed form (devit_attn_fwd) points all three at one qkv buffer
// with NQ == N; the rows form (devit_attn_fwd_rows) reads NQ <= N query rows per image from a buffer of their own -- the
// last block of a model whose caller consumes only the class / distillation tokens (models/de_vit.py:286-288).
struct AttnFwdArgs {
  const __bf16* q;      // [B*NQ][q_rs], feature h*64 + e
  const __bf16* k;      // [B*N][kv_rs]
  const __bf16* v;
  __bf16* out;          // [B*NQ][H*64]
  float* lse;           // [B][H][NQ]
  const float* head_gate;
  int B, N, NQ, H;
  int q_rs, kv_rs;
  float scale;
};

// Forward.  S^T = K Q^T puts the QUERY on the MFMA lane (col = lane & 15) and the keys in the accumulator
// registers, so the row softmax is in-lane + two shuffles, and the bf16 P values of two key tiles are already
// the B operand of O^T = V^T P (k-slot (g, j) = key 16*t(j>>2) + 4g + (j&3); V is read transposed with the
// same key order).  P never touches LDS; each lane ends up with 4 consecutive d of its own query row.
constexpr int FWD_WAVES = 8;

template <bool F16>
__global__ __launch_bounds__(FWD_WAVES * 64, 4) void attn_fwd_kernel(const AttnFwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* k_img = smem;
  char* v_img = smem + IMG_BYTES;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int b = blockIdx.x / a.H, h = blockIdx.x % a.H;
  const int D = a.H * HD, N = a.N, NQ = a.NQ;
  const size_t rs = (size_t)a.q_rs, krs = (size_t)a.kv_rs;
  const __bf16* qbase = a.q + (size_t)b * NQ * rs + h * HD;
  const float c2 = a.scale * 1.4426950408889634f;  // scores in log2 domain
  const float gate = a.head_gate ? a.head_gate[h] : 1.0f;
  const int ntile = (NQ + 15) >> 4;                  // query tiles
  const int g = lane >> 4, lc = lane & 15;
  const int tq = (lane >> 2) & 3, tp = lane & 3;     // transposed-read row / column-quad of this lane
  constexpr int QT = (MAXT + FWD_WAVES - 1) / FWD_WAVES;   // query tiles per wave
  // Q fragments of ALL this wave's query tiles first, then the K / V images: one exposed HBM latency per workgroup
  bf16x8 qall[QT][2];
#pragma unroll
  for (int it = 0; it < QT; ++it) {
    const int q_ = (wave + it * FWD_WAVES) * 16 + lc;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const bf16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
      qall[it][kk] = (wave + it * FWD_WAVES < ntile && q_ < NQ) ? *(const bf16x8*)(qbase + (size_t)q_ * rs + kk * 32 + g * 8) : z;
    }
  }
  // K and V images by LDS-DMA: no register round trip, no ds_write pass (forward -6 ... -10 % against register staging, same
  // box, profiles/r02_l_attention_dma_prologue.txt); the Q fragments above go straight to registers in MFMA layout
  dma_image<FWD_WAVES>(k_img, a.k + (size_t)b * N * krs + h * HD, krs, N, wave, lane);
  dma_image<FWD_WAVES>(v_img, a.v + (size_t)b * N * krs + h * HD, krs, N, wave, lane);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's share has landed; the barrier covers the others'
  __syncthreads();
  const int tmask = N >> 4;                          // first key tile that contains a key >= N

#pragma unroll
  for (int it = 0; it < QT; ++it) {
    const int qt = wave + it * FWD_WAVES;
    if (qt >= ntile) break;
    const int q = qt * 16 + lc;                      // this lane's query
    f32x4 s[MAXT + 1];
#pragma unroll
    for (int t = 0; t < MAXT; ++t) {
      s[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) s[t] = mfma16t<F16>(img_row_frag(k_img, t * 16, kk, lane), qall[it][kk], s[t]);
    }
    // raw-score row max (scale > 0); only tiles >= tmask can hold padded keys
    float mx = -INFINITY;
#pragma unroll
    for (int t = 0; t < MAXT; ++t) {
      if (t >= tmask) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (t * 16 + g * 4 + r >= N) s[t][r] = -INFINITY;
      }
      mx = fmaxf(fmaxf(fmaxf(s[t][0], s[t][1]), fmaxf(s[t][2], s[t][3])), mx);
    }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float mxs = mx * c2;
    float sum = 0.f;
#pragma unroll
    for (int t = 0; t < MAXT; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        s[t][r] = __builtin_amdgcn_exp2f(fmaf(s[t][r], c2, -mxs));   // exp2(-inf) = 0 for padded keys
        sum += s[t][r];
      }
    s[MAXT] = (f32x4){0.f, 0.f, 0.f, 0.f};           // keys 208..223
    sum += __shfl_xor(sum, 16, 64);
    sum += __shfl_xor(sum, 32, 64);
    if (a.lse && g == 0 && q < NQ) a.lse[((size_t)b * a.H + h) * NQ + q] = (mxs + log2f(sum)) * 0.6931471805599453f;

    f32x4 o[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) o[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 7; ++ks) {
      const f32x4 p0 = s[2 * ks], p1 = s[2 * ks + 1];
      const bf16x8 pf = cvt8<F16>(p0, p1);
      const int r0 = ks * 32 + g * 4 + tq, r1 = r0 + 16;   // keys of tile 2ks / 2ks+1 for this lane group
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        const int ch = dt * 2 + (tp >> 1), sub = (tp & 1) * 8;
        const bf16x8 vf = cat8(lds_tr_read(v_img + img_off(r0, ch) + sub), lds_tr_read(v_img + img_off(r1, ch) + sub));
        o[dt] = mfma16t<F16>(vf, pf, o[dt]);         // O^T[d][q] += V^T[d][key] P^T[key][q]
      }
    }
    if (q < NQ) {
      const float sc = gate / sum;
      __bf16* orow = a.out + ((size_t)b * NQ + q) * D + h * HD + g * 4;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        *(bf16x4*)(orow + dt * 16) = cvt4<F16>(o[dt] * sc);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// Backward.  Recomputes P from the saved log-sum-exp.  S and dP are computed with the KEY on the MFMA lane, so
// their accumulators (two 16-query tiles = one 32-query block) are already the B operands of
// dV^T += dO^T P and dK^T += Q^T dS: P never touches LDS and each wave keeps dK/dV of its own key tiles in
// registers for the whole kernel (no cross-workgroup reduction, no atomics).  Only dS crosses LDS, once, stored
// transposed ([key][q], 8-byte writes) and double-buffered (one barrier per query block), for
// dQ^T = K^T dS^T.  Every gradient leaves as 8-byte (4 x bf16) stores along d.
// ------------------------------------------------------------------------------------------
struct AttnBwdArgs {
  const __bf16* q;      // [B*NQ][q_rs]; query side / key side split as in AttnFwdArgs
  const __bf16* k;      // [B*N][kv_rs]
  const __bf16* v;
  const __bf16* out;    // forward output (post gate)  [B*NQ][D]
  const __bf16* dout;   // gradient wrt forward output [B*NQ][D]
  const float* lse;     // [B][H][NQ]
  const float* head_gate;
  const __bf16* dq_add; // optional extra gradients added in (relation loss), laid out like dq / dk / dv
  const __bf16* dk_add;
  const __bf16* dv_add;
  __bf16* dq;           // [B*NQ][dq_rs]
  __bf16* dk;           // [B*N][dkv_rs]
  __bf16* dv;
  int B, N, NQ, H;
  int q_rs, kv_rs, dq_rs, dkv_rs;
  float scale;
};

constexpr int DST_STRIDE = 40;                       // bf16 per dS^T row: 32 queries + pad (80-B rows)
constexpr int DST_BYTES = KROWS * DST_STRIDE * 2;    // 17920

__device__ __forceinline__ void store_grad4(__bf16* dst, const __bf16* add, f32x4 v) {
  if (add) {
    const bf16x4 e = *(const bf16x4*)add;
    v += (f32x4){bf2f(e[0]), bf2f(e[1]), bf2f(e[2]), bf2f(e[3])};
  }
  const bf16x4 o = {f2bf(v[0]), f2bf(v[1]), f2bf(v[2]), f2bf(v[3])};
  *(bf16x4*)dst = o;
}

constexpr int BWD_WAVES = 8;

__global__ __launch_bounds__(BWD_WAVES * 64) void attn_bwd_kernel(const AttnBwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* q_img = smem;
  char* k_img = smem + IMG_BYTES;
  char* v_img = smem + 2 * IMG_BYTES;
  char* do_img = smem + 3 * IMG_BYTES;
  char* dst_buf = smem + 4 * IMG_BYTES;              // 2 x [224 keys][DST_STRIDE]
  float* lse2 = (float*)(dst_buf + 2 * DST_BYTES);
  float* delta = lse2 + KROWS;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int b = blockIdx.x / a.H, h = blockIdx.x % a.H;
  const int D = a.H * HD, N = a.N, NQ = a.NQ;
  const size_t rs = (size_t)a.q_rs, krs = (size_t)a.kv_rs;
  const __bf16* qbase = a.q + (size_t)b * NQ * rs + h * HD;
  const float gate = a.head_gate ? a.head_gate[h] : 1.0f;
  const __bf16* dobase = a.dout + (size_t)b * NQ * D + h * HD;
  const __bf16* obase = a.out + (size_t)b * NQ * D + h * HD;

  {
    // Q, K, V images by LDS-DMA (issued first: they are in flight while the rest of the prologue runs); dO (scaled by the
    // head gate), the O rows and lse go through registers because delta[q] = sum_d dO[q][d] O[q][d] needs them there
    constexpr int NT = BWD_WAVES * 64;
    dma_image<BWD_WAVES>(q_img, qbase, rs, NQ, wave, lane);
    dma_image<BWD_WAVES>(k_img, a.k + (size_t)b * N * krs + h * HD, krs, N, wave, lane);
    dma_image<BWD_WAVES>(v_img, a.v + (size_t)b * N * krs + h * HD, krs, N, wave, lane);
    RowRegs<NT> dr, orr;
    float ls[RowRegs<NT>::ITERS];
    fetch_rows(dr, dobase, (size_t)D, NQ, tid);
    fetch_rows(orr, obase, (size_t)D, NQ, tid);
#pragma unroll
    for (int it = 0; it < RowRegs<NT>::ITERS; ++it) {
      const int idx = tid + it * NT, row = idx >> 3, c = idx & 7;
      ls[it] = (row < NQ && c == 0) ? a.lse[((size_t)b * a.H + h) * NQ + row] * 1.4426950408889634f : 0.f;
    }
    for (int i = tid; i < 2 * DST_BYTES / 16; i += NT) ((f32x4*)dst_buf)[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int it = 0; it < RowRegs<NT>::ITERS; ++it) {
      const int idx = tid + it * NT, row = idx >> 3, c = idx & 7;
      float dl = 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) dl += bf2f(dr.v[it][e]) * bf2f(orr.v[it][e]);
      dl += __shfl_xor(dl, 1, 64);
      dl += __shfl_xor(dl, 2, 64);
      dl += __shfl_xor(dl, 4, 64);
      if (c == 0 && row < KROWS) {
        delta[row] = dl;
        lse2[row] = ls[it];
      }
    }
    put_image(do_img, dr, gate, tid);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __syncthreads();

  const float c2 = a.scale * 1.4426950408889634f;
  const int ntile = (N + 15) >> 4;
  const int g = lane >> 4, lc = lane & 15;
  const int tq = (lane >> 2) & 3, tp = lane & 3;

  constexpr int KT = (MAXT + BWD_WAVES - 1) / BWD_WAVES;   // key tiles per wave (2)
  f32x4 dv[KT][4], dk[KT][4];  // [key tile of this wave][d tile]: rows d = 4g + r, col key = lc
#pragma unroll
  for (int i = 0; i < KT; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      dv[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
      dk[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }

  const int nblk = (NQ + 31) >> 5;
  for (int qb = 0; qb < nblk; ++qb) {
    char* dst = dst_buf + (qb & 1) * DST_BYTES;
    // ---- S, dP for this wave's key tiles x the block's two query tiles; dV^T, dK^T straight from registers
    bf16x8 qf[2][2], dof[2][2];
    float l2[2][4], dl[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int q0 = qb * 32 + i * 16;
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        qf[i][kk] = img_row_frag(q_img, q0, kk, lane);
        dof[i][kk] = img_row_frag(do_img, q0, kk, lane);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        l2[i][r] = lse2[q0 + g * 4 + r];
        dl[i][r] = delta[q0 + g * 4 + r];
      }
    }
    // A operands of the dV^T / dK^T products: dO^T and Q^T with k-slot (g, j) = query 16 (j>>2) + 4g + (j&3)
    bf16x8 dot[4], qtt[4];
    {
      const int r0 = qb * 32 + g * 4 + tq, r1 = r0 + 16;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        const int ch = dt * 2 + (tp >> 1), sub = (tp & 1) * 8;
        dot[dt] = cat8(lds_tr_read(do_img + img_off(r0, ch) + sub), lds_tr_read(do_img + img_off(r1, ch) + sub));
        qtt[dt] = cat8(lds_tr_read(q_img + img_off(r0, ch) + sub), lds_tr_read(q_img + img_off(r1, ch) + sub));
      }
    }
#pragma unroll
    for (int t = 0; t < KT; ++t) {
      const int kt = wave + t * BWD_WAVES;
      if (kt < ntile) {
        bf16x8 kf[2], vf[2];
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
          kf[kk] = img_row_frag(k_img, kt * 16, kk, lane);
          vf[kk] = img_row_frag(v_img, kt * 16, kk, lane);
        }
        f32x4 pp[2], ds[2];
        const bool kok = kt * 16 + lc < N;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          f32x4 sv = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int kk = 0; kk < 2; ++kk) {
            sv = mfma16(qf[i][kk], kf[kk], sv);      // S[q][key], key on the lane
            dp = mfma16(dof[i][kk], vf[kk], dp);     // dP[q][key]
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const bool ok = kok && (qb * 32 + i * 16 + g * 4 + r < NQ);
            const float p = ok ? exp2f(sv[r] * c2 - l2[i][r]) : 0.f;
            pp[i][r] = p;
            ds[i][r] = p * (dp[r] - dl[i][r]) * a.scale;
          }
          // dS^T[key][q = 16 i + 4 g + r], 4 consecutive queries = one 8-byte store
          const bf16x4 dsb = {f2bf(ds[i][0]), f2bf(ds[i][1]), f2bf(ds[i][2]), f2bf(ds[i][3])};
          *(bf16x4*)(dst + (kt * 16 + lc) * (DST_STRIDE * 2) + (i * 16 + g * 4) * 2) = dsb;
        }
        const bf16x8 pf = {f2bf(pp[0][0]), f2bf(pp[0][1]), f2bf(pp[0][2]), f2bf(pp[0][3]),
                           f2bf(pp[1][0]), f2bf(pp[1][1]), f2bf(pp[1][2]), f2bf(pp[1][3])};
        const bf16x8 dsf = {f2bf(ds[0][0]), f2bf(ds[0][1]), f2bf(ds[0][2]), f2bf(ds[0][3]),
                            f2bf(ds[1][0]), f2bf(ds[1][1]), f2bf(ds[1][2]), f2bf(ds[1][3])};
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
          dv[t][dt] = mfma16(dot[dt], pf, dv[t][dt]);    // dV^T[d][key] += dO^T[d][q] P[q][key]
          dk[t][dt] = mfma16(qtt[dt], dsf, dk[t][dt]);   // dK^T[d][key] += Q^T[d][q] dS[q][key]
        }
      }
    }
    __syncthreads();   // dS^T of this block complete (the other buffer is free again two blocks later)
    {
      // dQ^T[d][q] = sum_key K^T[d][key] dS^T[key][q]: wave -> query tile i = wave >> 2, d tile wave & 3
      const int i = wave >> 2, dt0 = wave & 3;
      f32x4 dq[1] = {{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
      for (int ks = 0; ks < 7; ++ks) {
        const int kr = ks * 32 + g * 8 + tq;
        // B[k = key][col = q]: transposed read of the dS^T image, columns q = 16 i + 4 tp ..
        const char* pb = dst + kr * (DST_STRIDE * 2) + (i * 16 + tp * 4) * 2;
        const bf16x8 bfr = cat8(lds_tr_read(pb), lds_tr_read(pb + 4 * DST_STRIDE * 2));
#pragma unroll
        for (int u = 0; u < 1; ++u) dq[u] = mfma16(img_tr_frag(k_img, ks * 32, (dt0 + u) * 16, lane), bfr, dq[u]);
      }
      const int q = qb * 32 + i * 16 + lc;
      if (q < NQ) {
#pragma unroll
        for (int u = 0; u < 1; ++u) {
          const size_t o = ((size_t)b * NQ + q) * a.dq_rs + h * HD + (dt0 + u) * 16 + g * 4;
          store_grad4(a.dq + o, a.dq_add ? a.dq_add + o : nullptr, dq[u]);
        }
      }
    }
  }
  // ---- dK, dV of this wave's key tiles.  The accumulators hold 4 consecutive d per register quad for key = lane & 15:
  // stored directly that is 8 bytes per lane in 32-byte row segments.  The images are dead now, so each wave turns its
  // tiles around through a private LDS slab ([16 keys][64 d] fp32, padded rows) and writes -- and reads the
  // optional extra gradient -- in whole 128-byte rows, 16 bytes per lane.
  __syncthreads();                                   // every wave has finished reading the images
  {
    constexpr int SROW = 272;                        // bytes per staged key row: 64 fp32 + 16 B pad (conflict-free writes)
    char* slab = smem + wave * (2 * 16 * SROW);      // [dk | dv] x 16 rows, fp32: rounded to bf16 once, after the add
#pragma unroll
    for (int t = 0; t < KT; ++t) {
      const int kt = wave + t * BWD_WAVES;
      if (kt >= ntile) break;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        *(f32x4*)(slab + lc * SROW + (dt * 16 + g * 4) * 4) = dk[t][dt];
        *(f32x4*)(slab + 16 * SROW + lc * SROW + (dt * 16 + g * 4) * 4) = dv[t][dt];
      }
      // (same wave writes and reads: the LDS queue is in order and hipcc waits lgkmcnt before the reads' use)
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        const int row = half * 8 + (lane >> 3), c8 = lane & 7, key = kt * 16 + row;
        if (key < N) {
          const size_t oo = ((size_t)b * N + key) * a.dkv_rs + h * HD + c8 * 8;
#pragma unroll
          for (int which = 0; which < 2; ++which) {            // 0: dK, 1: dV
            const char* src = slab + which * 16 * SROW + row * SROW + c8 * 32;
            f32x4 lo = *(const f32x4*)src, hi = *(const f32x4*)(src + 16);
            const __bf16* add = which ? a.dv_add : a.dk_add;
            if (add) {
              const bf16x8 e = *(const bf16x8*)(add + oo);
              lo += (f32x4){bf2f(e[0]), bf2f(e[1]), bf2f(e[2]), bf2f(e[3])};
              hi += (f32x4){bf2f(e[4]), bf2f(e[5]), bf2f(e[6]), bf2f(e[7])};
            }
            const bf16x8 v = {f2bf(lo[0]), f2bf(lo[1]), f2bf(lo[2]), f2bf(lo[3]), f2bf(hi[0]), f2bf(hi[1]), f2bf(hi[2]), f2bf(hi[3])};
            *(bf16x8*)((which ? a.dv : a.dk) + oo) = v;
          }
        }
      }
    }
  }
}

constexpr int FWD_LDS = 2 * IMG_BYTES;                                             // 57344: 2 workgroups per CU
constexpr int BWD_LDS = 4 * IMG_BYTES + 2 * DST_BYTES + 2 * KROWS * 4;             // 152320

}  // namespace

namespace {

int launch_attn_fwd(const AttnFwdArgs& a, int dtype16, void* stream) {
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)attn_fwd_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, FWD_LDS);
    if (e == hipSuccess)
      e = hipFuncSetAttribute((const void*)attn_fwd_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, FWD_LDS);
    DEVIT_CHECK(e == hipSuccess, DEVIT_ERR_LAUNCH, "devit_attn_fwd: hipFuncSetAttribute: %s", hipGetErrorString(e));
    attr_set = true;
  }
  if (dtype16)
    hipLaunchKernelGGL(attn_fwd_kernel<true>, dim3(a.B * a.H), dim3(FWD_WAVES * 64), FWD_LDS, (hipStream_t)stream, a);
  else
    hipLaunchKernelGGL(attn_fwd_kernel<false>, dim3(a.B * a.H), dim3(FWD_WAVES * 64), FWD_LDS, (hipStream_t)stream, a);
  DEVIT_LAUNCH_CHECK();
  return DEVIT_OK;
}

int launch_attn_bwd(const AttnBwdArgs& a, void* stream) {
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)attn_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, BWD_LDS);
    DEVIT_CHECK(e == hipSuccess, DEVIT_ERR_LAUNCH, "devit_attn_bwd: hipFuncSetAttribute: %s", hipGetErrorString(e));
    attr_set = true;
  }
  hipLaunchKernelGGL(attn_bwd_kernel, dim3(a.B * a.H), dim3(BWD_WAVES * 64), BWD_LDS, (hipStream_t)stream, a);
  DEVIT_LAUNCH_CHECK();
  return DEVIT_OK;
}

bool al16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

}  // namespace

extern "C" int devit_attn_fwd(const void* qkv, void* out, float* lse, const float* head_gate, int B, int N, int H,
                              int head_dim, float scale, int dtype16, void* stream) {
  DEVIT_CHECK(qkv && out && (dtype16 == 0 || dtype16 == 1), DEVIT_ERR_ARG, "devit_attn_fwd: bad argument");
  DEVIT_CHECK(head_dim == HD && N > 0 && N <= MAXT * 16 && B > 0 && H > 0, DEVIT_ERR_SHAPE,
              "devit_attn_fwd: needs head_dim == 64 and N <= 208 (got hd=%d N=%d)", head_dim, N);
  const int D = H * HD;
  const __bf16* p = (const __bf16*)qkv;
  AttnFwdArgs a{p, p + D, p + 2 * D, (__bf16*)out, lse, head_gate, B, N, N, H, 3 * D, 3 * D, scale};
  return launch_attn_fwd(a, dtype16, stream);
}

extern "C" int devit_attn_fwd_rows(const void* q, int q_ld, const void* kv, int kv_ld, void* out, float* lse,
                                   const float* head_gate, int B, int NQ, int N, int H, int head_dim, float scale,
                                   int dtype16, void* stream) {
  DEVIT_CHECK(q && kv && out && (dtype16 == 0 || dtype16 == 1), DEVIT_ERR_ARG, "devit_attn_fwd_rows: bad argument");
  DEVIT_CHECK(head_dim == HD && N > 0 && N <= MAXT * 16 && NQ > 0 && NQ <= N && B > 0 && H > 0, DEVIT_ERR_SHAPE,
              "devit_attn_fwd_rows: needs head_dim == 64, NQ <= N <= 208 (got hd=%d NQ=%d N=%d)", head_dim, NQ, N);
  const int D = H * HD;
  DEVIT_CHECK(q_ld >= D && kv_ld >= 2 * D && q_ld % 8 == 0 && kv_ld % 8 == 0 && al16(q) && al16(kv) && al16(out),
              DEVIT_ERR_ARG, "devit_attn_fwd_rows: q_ld=%d kv_ld=%d / pointers must be 16-byte aligned and hold H*64 (2*H*64) features", q_ld, kv_ld);
  const __bf16* p = (const __bf16*)kv;
  AttnFwdArgs a{(const __bf16*)q, p, p + D, (__bf16*)out, lse, head_gate, B, N, NQ, H, q_ld, kv_ld, scale};
  return launch_attn_fwd(a, dtype16, stream);
}

extern "C" int devit_attn_bwd(const void* qkv, const void* out, const void* dout, const float* lse,
                              const float* head_gate, const void* dqkv_add, void* dqkv, int B, int N, int H,
                              int head_dim, float scale, void* stream) {
  DEVIT_CHECK(qkv && out && dout && lse && dqkv, DEVIT_ERR_ARG, "devit_attn_bwd: null pointer");
  DEVIT_CHECK(head_dim == HD && N > 0 && N <= MAXT * 16 && B > 0 && H > 0, DEVIT_ERR_SHAPE,
              "devit_attn_bwd: needs head_dim == 64 and N <= 208 (got hd=%d N=%d)", head_dim, N);
  const int D = H * HD;
  const __bf16* p = (const __bf16*)qkv;
  const __bf16* ad = (const __bf16*)dqkv_add;
  __bf16* d = (__bf16*)dqkv;
  AttnBwdArgs a{p, p + D, p + 2 * D, (const __bf16*)out, (const __bf16*)dout, lse, head_gate,
                ad, ad ? ad + D : nullptr, ad ? ad + 2 * D : nullptr, d, d + D, d + 2 * D,
                B, N, N, H, 3 * D, 3 * D, 3 * D, 3 * D, scale};
  return launch_attn_bwd(a, stream);
}

extern "C" int devit_attn_bwd_rows(const void* q, int q_ld, const void* kv, int kv_ld, const void* out, const void* dout,
                                   const float* lse, const float* head_gate, void* dq, int dq_ld, void* dkv, int dkv_ld,
                                   int B, int NQ, int N, int H, int head_dim, float scale, void* stream) {
  DEVIT_CHECK(q && kv && out && dout && lse && dq && dkv, DEVIT_ERR_ARG, "devit_attn_bwd_rows: null pointer");
  DEVIT_CHECK(head_dim == HD && N > 0 && N <= MAXT * 16 && NQ > 0 && NQ <= N && B > 0 && H > 0, DEVIT_ERR_SHAPE,
              "devit_attn_bwd_rows: needs head_dim == 64, NQ <= N <= 208 (got hd=%d NQ=%d N=%d)", head_dim, NQ, N);
  const int D = H * HD;
  DEVIT_CHECK(q_ld >= D && dq_ld >= D && kv_ld >= 2 * D && dkv_ld >= 2 * D && q_ld % 8 == 0 && kv_ld % 8 == 0 &&
                  dq_ld % 8 == 0 && dkv_ld % 8 == 0 && al16(q) && al16(kv) && al16(out) && al16(dout) && al16(dq) && al16(dkv),
              DEVIT_ERR_ARG, "devit_attn_bwd_rows: leading dimensions / pointers must be 16-byte aligned and wide enough");
  const __bf16* p = (const __bf16*)kv;
  __bf16* d = (__bf16*)dkv;
  AttnBwdArgs a{(const __bf16*)q, p, p + D, (const __bf16*)out, (const __bf16*)dout, lse, head_gate,
                nullptr, nullptr, nullptr, (__bf16*)dq, d, d + D, B, N, NQ, H, q_ld, kv_ld, dq_ld, dkv_ld, scale};
  return launch_attn_bwd(a, stream);
}
