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
#include <stdlib.h>

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
    // unconditional load from a clamped row, zeroed afterwards: a load under a per-element condition makes hipcc branch
    // around it and wait for each one in turn (one HBM round trip per iteration)
    r.v[it] = *(const bf16x8*)(src + (size_t)min(row, N - 1) * row_stride + c * 8);
  }
#pragma unroll
  for (int it = 0; it < RowRegs<NT>::ITERS; ++it) {
    const bf16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
    if ((tid + it * NT) >> 3 >= N) r.v[it] = z;
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
template <int NWAVES, int ROWS = KROWS>
__device__ __forceinline__ void dma_image(char* img, const __bf16* src, size_t row_stride, int N, int wave, int lane) {
#pragma unroll
  for (int it = 0; it < (ROWS / 8 + NWAVES - 1) / NWAVES; ++it) {
    const int slab = wave + it * NWAVES;
    if (slab < ROWS / 8) {
      const int row = slab * 8 + (lane >> 3), c = (lane & 7) ^ ((row >> 1) & 7);
      const __bf16* g = src + (size_t)min(row, N - 1) * row_stride + c * 8;
      // (default cache policy: a non-temporal DMA is 18 % faster from cold HBM -- 51 / 92 against 62 / 113 us forward -- and
      // no faster in the step, where much of qkv is still in the Infinity Cache: profiles/r03_t_attention_nt_loads.txt)
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
// -DDEVIT_ATTN_FWD_WAVES=4 (round-3 experiment, measured null): 4 waves, images of 208 rows (13 key tiles), 53 KB per
// workgroup -> THREE workgroups per CU instead of two.  Half of a workgroup's life is its prologue (in-kernel stamps,
// tools/attn_fwd_stamps.py), but a third resident workgroup only makes every prologue longer: 62.6 / 110 us against
// 62.0 / 111 us per student / teacher launch (profiles/r03_s_attention_fwd_occupancy_null.txt).  The launch runs at the rate
// the memory system delivers first-touch 128-byte row pieces (2.5-2.8 TB/s), whatever the occupancy.
#ifndef DEVIT_ATTN_FWD_WAVES
#define DEVIT_ATTN_FWD_WAVES 8
#endif
constexpr int FWD_WAVES = DEVIT_ATTN_FWD_WAVES;
#ifndef DEVIT_ATTN_OUT_ROWS
#define DEVIT_ATTN_OUT_ROWS 1        // forward output rows leave as whole 128-byte lines through a per-wave LDS slab (0: 32-byte pieces)
#endif
constexpr int OSLAB_ROW = 144;       // bytes per slab row: 128 + 16 (16 rows of one column land on 8 different bank groups)
constexpr int FWD_ROWS = FWD_WAVES == 8 ? KROWS : MAXT * 16;           // rows of the K / V images
constexpr int FWD_IMG = FWD_ROWS * HD * 2;

template <bool F16>
__global__ __launch_bounds__(FWD_WAVES * 64, FWD_WAVES == 8 ? 4 : 3) void attn_fwd_kernel(const AttnFwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* k_img = smem;
  char* v_img = smem + FWD_IMG;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int b = blockIdx.x / a.H, h = blockIdx.x % a.H;
  const int D = a.H * HD, N = a.N, NQ = a.NQ;
  const size_t rs = (size_t)a.q_rs, krs = (size_t)a.kv_rs;
  const __bf16* qbase = a.q + (size_t)b * NQ * rs + h * HD;
  const float c2 = a.scale * 1.4426950408889634f;  // scores in log2 domain
#ifdef DEVIT_ATTN_STAMP    // diagnostic build (tools/attn_stamps.py): head_gate carries a u64 stamp buffer, 8 per workgroup
  unsigned long long* stamps = (unsigned long long*)a.head_gate + (size_t)blockIdx.x * 8;
  const float gate = 1.0f;
  if (tid == 0) { stamps[0] = __builtin_amdgcn_s_memrealtime(); stamps[1] = __builtin_amdgcn_s_memtime(); }
#else
  const float gate = a.head_gate ? a.head_gate[h] : 1.0f;
#endif
  const int ntile = (NQ + 15) >> 4;                  // query tiles
  const int g = lane >> 4, lc = lane & 15;
  const int tq = (lane >> 2) & 3, tp = lane & 3;     // transposed-read row / column-quad of this lane
  constexpr int QT = (MAXT + FWD_WAVES - 1) / FWD_WAVES;   // query tiles per wave
  // Q fragments of ALL this wave's query tiles first, then the K / V images: one exposed HBM latency per workgroup
  // (fetched in MFMA layout, 16 rows x 64 bytes per instruction; whole 128-byte rows + a lane trade afterwards measured 0.04 ms
  // per step SLOWER, profiles/r03_B_attention_rows.txt)
  bf16x8 qall[QT][2];
#pragma unroll
  for (int it = 0; it < QT; ++it) {
    const bf16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
    const int q_ = (wave + it * FWD_WAVES) * 16 + lc;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
      qall[it][kk] = (wave + it * FWD_WAVES < ntile && q_ < NQ) ? *(const bf16x8*)(qbase + (size_t)q_ * rs + kk * 32 + g * 8) : z;
  }
  // K and V images by LDS-DMA: no register round trip, no ds_write pass (forward -6 ... -10 % against register staging, same
  // box, profiles/r02_l_attention_dma_prologue.txt); the Q fragments above go straight to registers in MFMA layout
  dma_image<FWD_WAVES, FWD_ROWS>(k_img, a.k + (size_t)b * N * krs + h * HD, krs, N, wave, lane);
  dma_image<FWD_WAVES, FWD_ROWS>(v_img, a.v + (size_t)b * N * krs + h * HD, krs, N, wave, lane);
#ifdef DEVIT_ATTN_STAMP
  if (tid == 0) stamps[6] = __builtin_amdgcn_s_memtime();      // loads issued
#endif
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's share has landed; the barrier covers the others'
#ifdef DEVIT_ATTN_STAMP
  if (tid == 0) stamps[7] = __builtin_amdgcn_s_memtime();      // ... landed (wave 0)
#endif
  __syncthreads();
#ifdef DEVIT_ATTN_STAMP
  if (tid == 0) stamps[2] = __builtin_amdgcn_s_memtime();      // images complete
#endif
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
      // keys of tile 2ks / 2ks+1 for this lane group; tile 13 (keys 208..223) has P = 0 and, in the 208-row image, no rows:
      // its operand is read from tile 12's rows (any finite values do)
      const int r0 = ks * 32 + g * 4 + tq, r1 = (FWD_ROWS < KROWS && ks == 6) ? r0 : r0 + 16;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        const int ch = dt * 2 + (tp >> 1), sub = (tp & 1) * 8;
        const bf16x8 vf = cat8(lds_tr_read(v_img + img_off(r0, ch) + sub), lds_tr_read(v_img + img_off(r1, ch) + sub));
        o[dt] = mfma16t<F16>(vf, pf, o[dt]);         // O^T[d][q] += V^T[d][key] P^T[key][q]
      }
    }
#if DEVIT_ATTN_OUT_ROWS
    {
      // Lane (g, lc) holds O[q = lc][d = 16 dt + 4 g + r]: stored from here, an instruction covers 16 rows x 32 bytes -- quarter
      // cache lines, four times the transactions of the bytes (ablation: the output stores cost 10 / 25 us of a 62 / 111 us
      // launch for a quarter of the bytes it reads).  Through this wave's private [16][64] LDS slab instead: two
      // instructions of eight whole 128-byte rows.
      const float sc = gate / sum;
      char* slab = smem + 2 * FWD_IMG + wave * (16 * OSLAB_ROW);
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) *(bf16x4*)(slab + lc * OSLAB_ROW + (dt * 16 + g * 4) * 2) = cvt4<F16>(o[dt] * sc);
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        const int row = half * 8 + (lane >> 3), c8 = lane & 7, qr = qt * 16 + row;
        const bf16x8 v = *(const bf16x8*)(slab + row * OSLAB_ROW + c8 * 16);     // (same wave wrote it: ordered by lgkmcnt)
        if (qr < NQ) *(bf16x8*)(a.out + ((size_t)b * NQ + qr) * D + h * HD + c8 * 8) = v;
      }
    }
#else
    if (q < NQ) {
      const float sc = gate / sum;
      __bf16* orow = a.out + ((size_t)b * NQ + q) * D + h * HD + g * 4;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        *(bf16x4*)(orow + dt * 16) = cvt4<F16>(o[dt] * sc);
      }
    }
#endif
  }
#ifdef DEVIT_ATTN_STAMP
  if (tid == 0) stamps[3] = __builtin_amdgcn_s_memtime();      // wave 0's compute + store issue done
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (tid == 0) { stamps[4] = __builtin_amdgcn_s_memtime(); stamps[5] = __builtin_amdgcn_s_memrealtime(); }
#endif
}

// ------------------------------------------------------------------------------------------
// Backward.  Recomputes P from the saved log-sum-exp.  S and dP are computed with the KEY on the MFMA lane, so their
// accumulators (two 16-query tiles = one 32-query block) are already the B operands of dV^T += dO^T P and
// dK^T += Q^T dS: P never touches LDS and each wave keeps dK / dV of its own key tiles in registers for the whole kernel
// (no cross-workgroup reduction, no atomics).  Only dS crosses LDS, once per block, stored transposed ([key][q], 8-byte
// writes), for dQ^T = K^T dS^T.
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

__device__ __forceinline__ void store_grad4(__bf16* dst, const __bf16* add, f32x4 v) {
  if (add) {
    const bf16x4 e = *(const bf16x4*)add;
    v += (f32x4){bf2f(e[0]), bf2f(e[1]), bf2f(e[2]), bf2f(e[3])};
  }
  const bf16x4 o = {f2bf(v[0]), f2bf(v[1]), f2bf(v[2]), f2bf(v[3])};
  *(bf16x4*)dst = o;
}

// ------------------------------------------------------------------------------------------
// Two workgroups per CU: 4 waves, <= 256 VGPRs, 79 KB of LDS each.  (Rounds 1-2 ran 8 waves with four [224][64] images in
// LDS, 152 KB: one workgroup per CU, so load -> compute -> store ran strictly one after the other on every CU -- 181 us per
// B = 256, H = 6 launch from cold HBM against 160 us for this form, profiles/r03_*_attention_bwd.txt.)  Only the K image
// stays in LDS (the dQ product needs every key's row); V lives in registers as the row fragments of the wave's own key
// tiles; Q and dO arrive one 32-query block at a time through a four-stage LDS ring filled by LDS-DMA up to three blocks
// ahead (counted vmcnt); the block's Q / dO fragments are re-read from LDS per key tile because 128 accumulator + 32
// V-fragment registers leave no room to hold them.  Two such workgroups share a CU and run out of phase: one loads or
// stores while the other computes.  Same MFMA shapes, operand roundings and reduction order over query blocks / key steps
// as the 8-wave kernel had, except that the head gate multiplies dP and dV in fp32 instead of a bf16 copy of dO: identical
// for the 0/1 gates of core/imp_rank.py.
// In-kernel stamps (tools/attn_stamps.py, -DDEVIT_ATTN_STAMP): a workgroup lives ~47 us = prologue 23 (140 KB of first-touch
// loads at the ~3-6 B/cycle its CU's load path gives it beside the other workgroup) + main loop 22 (bound by
// vector-instruction ISSUE, ~600 instructions per wave and block, not by MFMA or memory: halving the instruction count of
// the softmax-gradient arithmetic took it from 26 to 22) + stores 2.
// ------------------------------------------------------------------------------------------
constexpr int B4_WAVES = 4;
constexpr int B4_KT = (MAXT + 1 + B4_WAVES - 1) / B4_WAVES;      // key tiles per wave (4); tile 13 (keys 208..223) is padding
constexpr int DST4_STRIDE = 36;                                   // bf16 per dS^T row: 32 queries + pad (72-B rows)
constexpr int DST4_BYTES = KROWS * DST4_STRIDE * 2;               // 16128
constexpr int QD_STAGE = 2 * 32 * HD * 2;                         // one ring stage: Q block + dO block, 8192 B
constexpr int QD_NST = 4;                                         // ring stages: three blocks (24 KB per workgroup) in flight
constexpr int BWD4_LDS = IMG_BYTES + QD_NST * QD_STAGE + DST4_BYTES + 2 * KROWS * 4;   // 79360

// one LDS-DMA instruction (1 KiB: 8 rows x 128 B) from per-lane global addresses to LDS byte address `lds`; inline asm so
// that hipcc's waitcnt pass does not put a vmcnt(0) in front of every later ds_read (gemm.hip, dma2_perlane)
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

// rows [r0, r0 + 32) of a strided [N][64] matrix -> a [32][64] LDS block image (swizzled like the big images, local rows);
// wave w moves slab w (rows 8 w .. 8 w + 7); rows >= nrows repeat row nrows - 1
__device__ __forceinline__ void dma_block(char* blk, const __bf16* src, size_t row_stride, int r0, int nrows, int wave, int lane) {
  const int lr = wave * 8 + (lane >> 3), c = (lane & 7) ^ ((lr >> 1) & 7);
  const __bf16* g = src + (size_t)min(r0 + lr, nrows - 1) * row_stride + c * 8;
  dma1(g, (unsigned)(size_t)LDS_PTR(blk) + (unsigned)wave * 1024u);
}

__global__ __launch_bounds__(B4_WAVES * 64, 2) void attn_bwd4_kernel(const AttnBwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* k_img = smem;
  char* qd = smem + IMG_BYTES;                       // QD_NST stages x {Q block [32][64], dO block [32][64]}
  char* dst = qd + QD_NST * QD_STAGE;                // dS^T of the current block: [224 keys][DST4_STRIDE]
  float* lse2 = (float*)(dst + DST4_BYTES);
  float* delta = lse2 + KROWS;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int b = blockIdx.x / a.H, h = blockIdx.x % a.H;
  const int D = a.H * HD, N = a.N, NQ = a.NQ;
  const size_t rs = (size_t)a.q_rs, krs = (size_t)a.kv_rs;
  const __bf16* qbase = a.q + (size_t)b * NQ * rs + h * HD;
  const __bf16* kbase = a.k + (size_t)b * N * krs + h * HD;
  const __bf16* vbase = a.v + (size_t)b * N * krs + h * HD;
#ifdef DEVIT_ATTN_STAMP    // diagnostic build (tools/attn_stamps.py): head_gate carries a u64 stamp buffer, 8 per workgroup
  unsigned long long* stamps = (unsigned long long*)a.head_gate + (size_t)blockIdx.x * 8;
  const float gate = 1.0f;
  if (tid == 0) { stamps[0] = __builtin_amdgcn_s_memrealtime(); stamps[1] = __builtin_amdgcn_s_memtime(); }
#define ATTN_STAMP(i) do { if (tid == 0) stamps[i] = __builtin_amdgcn_s_memtime(); } while (0)
#else
  const float gate = a.head_gate ? a.head_gate[h] : 1.0f;
#define ATTN_STAMP(i) do { } while (0)
#endif
  const __bf16* dobase = a.dout + (size_t)b * NQ * D + h * HD;
  const __bf16* obase = a.out + (size_t)b * NQ * D + h * HD;
  const int g = lane >> 4, lc = lane & 15;
  const int tq = (lane >> 2) & 3, tp = lane & 3;
  const int ntile = (N + 15) >> 4;
  const int nblk = (NQ + 31) >> 5;

  // ---- prologue: K image and the first Q / dO block by LDS-DMA; V fragments of this wave's key tiles straight to registers;
  // delta[q] = sum_d dO[q][d] O[q][d] and lse from global rows
  dma_image<B4_WAVES>(k_img, kbase, krs, N, wave, lane);
#pragma unroll
  for (int pb = 0; pb < QD_NST - 1; ++pb)
    if (pb < nblk) {
      dma_block(qd + pb * QD_STAGE, qbase, rs, pb * 32, NQ, wave, lane);
      dma_block(qd + pb * QD_STAGE + 32 * HD * 2, dobase, (size_t)D, pb * 32, NQ, wave, lane);
    }
  bf16x8 vf[B4_KT][2];
#pragma unroll
  for (int t = 0; t < B4_KT; ++t) {
    const int key = min((wave + t * B4_WAVES) * 16 + lc, N - 1);
#pragma unroll
#ifdef DEVIT_ATTN_ABL_NOV       // ablation build (round 6): no V loads
    for (int kk = 0; kk < 2; ++kk) asm volatile("" : "=v"(vf[t][kk]));
#else
    for (int kk = 0; kk < 2; ++kk) vf[t][kk] = *(const bf16x8*)(vbase + (size_t)key * krs + kk * 32 + g * 8);
#endif
  }
  {
    constexpr int NT = B4_WAVES * 64;
    RowRegs<NT> dr, orr;
#ifdef DEVIT_ATTN_ABL_NODELTA   // ablation build (round 6): no dO / O rows for delta
#pragma unroll
    for (int it = 0; it < RowRegs<NT>::ITERS; ++it) { asm volatile("" : "=v"(dr.v[it])); asm volatile("" : "=v"(orr.v[it])); }
#else
    fetch_rows(dr, dobase, (size_t)D, NQ, tid);
    fetch_rows(orr, obase, (size_t)D, NQ, tid);
#endif
    ATTN_STAMP(6);                                     // every prologue load is issued
#ifdef DEVIT_ATTN_STAMP
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    ATTN_STAMP(7);                                     // ... and has landed
#endif
    for (int i = tid; i < DST4_BYTES / 16; i += NT) ((f32x4*)dst)[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
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
        delta[row] = dl * a.scale;                   // pre-scaled: dS = P * (dP * gate * scale - delta * scale)
        lse2[row] = row < NQ ? a.lse[((size_t)b * a.H + h) * NQ + row] * 1.4426950408889634f : 0.f;
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  ATTN_STAMP(2);

  const float c2 = a.scale * 1.4426950408889634f;
  const float gs = gate * a.scale;
  f32x4 dv[B4_KT][4], dk[B4_KT][4];  // [key tile of this wave][d tile]: rows d = 4g + r, col key = lc
#pragma unroll
  for (int i = 0; i < B4_KT; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      dv[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
      dk[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }

#ifdef DEVIT_ATTN_ABL_NOMAIN    // ablation build (round 6): prologue + final stores only
  for (int qb = 0; qb < (a.N < 0 ? nblk : 0); ++qb) {
#else
  for (int qb = 0; qb < nblk; ++qb) {
#endif
    const char* q_blk = qd + (qb & (QD_NST - 1)) * QD_STAGE;
    const char* do_blk = q_blk + 32 * HD * 2;
    // Q / dO of block qb + 3 into the stage block qb - 1 was read from (every wave is past that block's barriers).  Exactly
    // two LDS-DMA instructions per wave and block: the counted wait below relies on it.
    if (qb + QD_NST - 1 < nblk) {
      char* nq = qd + ((qb + QD_NST - 1) & (QD_NST - 1)) * QD_STAGE;
      dma_block(nq, qbase, rs, (qb + QD_NST - 1) * 32, NQ, wave, lane);
      dma_block(nq + 32 * HD * 2, dobase, (size_t)D, (qb + QD_NST - 1) * 32, NQ, wave, lane);
    }
    // ---- per key tile of this wave: S and dP against the block's two query tiles -> P, dS (registers = MFMA operands, dS^T also
    // to LDS), then dV^T += dO^T P and dK^T += Q^T dS.  The block's Q / dO fragments are read from LDS per tile, not held
    // across tiles: with 128 accumulator and 32 V-fragment registers there is no room for them (256 per wave at two
    // workgroups per CU), and LDS has the bandwidth (~70 KB per wave and block).
#pragma unroll
    for (int t = 0; t < B4_KT; ++t) {
      const int kt = wave + t * B4_WAVES;
      if (kt < ntile) {
        asm volatile("" ::: "memory");                 // keep hipcc from hoisting (and keeping alive) the loop-invariant LDS reads
        bf16x8 kf[2];
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) kf[kk] = img_row_frag(k_img, kt * 16, kk, lane);
        f32x4 pp[2], ds[2];
        const bool kok = kt * 16 + lc < N;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          f32x4 sv = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int kk = 0; kk < 2; ++kk) {
            sv = mfma16(img_row_frag(q_blk, i * 16, kk, lane), kf[kk], sv);       // S[q][key], key on the lane
            dp = mfma16(img_row_frag(do_blk, i * 16, kk, lane), vf[t][kk], dp);   // dP[q][key] (before the head gate)
          }
          const f32x4 l2 = *(const f32x4*)(lse2 + qb * 32 + i * 16 + g * 4), dl = *(const f32x4*)(delta + qb * 32 + i * 16 + g * 4);
          // The kernel is bound by vector-instruction issue (~1150 per wave and block before this form), not by MFMA or
          // memory: four instructions per element on the interior (fma, v_exp, fma, mul), the masks only where padded keys
          // (last key tile) or padded queries (last block) exist -- wave-uniform branch.
          if (kt * 16 + 16 <= N && qb * 32 + i * 16 + 16 <= NQ) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const float p = __builtin_amdgcn_exp2f(fmaf(sv[r], c2, -l2[r]));
              pp[i][r] = p;
              ds[i][r] = p * fmaf(dp[r], gs, -dl[r]);
            }
          } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const bool ok = kok && (qb * 32 + i * 16 + g * 4 + r < NQ);
              const float p = ok ? __builtin_amdgcn_exp2f(fmaf(sv[r], c2, -l2[r])) : 0.f;
              pp[i][r] = p;
              ds[i][r] = p * fmaf(dp[r], gs, -dl[r]);
            }
          }
          // dS^T[key][q = 16 i + 4 g + r], 4 consecutive queries = one 8-byte store
          const bf16x4 dsb = {f2bf(ds[i][0]), f2bf(ds[i][1]), f2bf(ds[i][2]), f2bf(ds[i][3])};
          *(bf16x4*)(dst + (kt * 16 + lc) * (DST4_STRIDE * 2) + (i * 16 + g * 4) * 2) = dsb;
        }
        const bf16x8 pf = {f2bf(pp[0][0]), f2bf(pp[0][1]), f2bf(pp[0][2]), f2bf(pp[0][3]),
                           f2bf(pp[1][0]), f2bf(pp[1][1]), f2bf(pp[1][2]), f2bf(pp[1][3])};
        const bf16x8 dsf = {f2bf(ds[0][0]), f2bf(ds[0][1]), f2bf(ds[0][2]), f2bf(ds[0][3]),
                            f2bf(ds[1][0]), f2bf(ds[1][1]), f2bf(ds[1][2]), f2bf(ds[1][3])};
        // A operands dO^T, Q^T: k-slot (g, j) = query 16 (j>>2) + 4g + (j&3), transposed reads of the block images
        const int r0 = g * 4 + tq, r1 = r0 + 16;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
          const int ch = dt * 2 + (tp >> 1), sub = (tp & 1) * 8;
          const bf16x8 dot = cat8(lds_tr_read(do_blk + img_off(r0, ch) + sub), lds_tr_read(do_blk + img_off(r1, ch) + sub));
          const bf16x8 qtt = cat8(lds_tr_read(q_blk + img_off(r0, ch) + sub), lds_tr_read(q_blk + img_off(r1, ch) + sub));
          dv[t][dt] = mfma16(dot, pf, dv[t][dt]);      // dV^T[d][key] += dO^T[d][q] P[q][key]
          dk[t][dt] = mfma16(qtt, dsf, dk[t][dt]);     // dK^T[d][key] += Q^T[d][q] dS[q][key]
        }
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                       // Y1: dS^T of this block complete
    asm volatile("" ::: "memory");                      // (s_barrier is IntrNoMem: no LDS access may be moved across it)
    // This wave's share of block qb + 1 must have landed before Y2.  vmcnt retires in order; DMA(qb + 1) was issued at the top of
    // iteration qb - 2, and newer than its two instructions are EIGHT operations: the two dq stores of block qb - 2, DMA(qb + 2)
    // x 2, the two dq stores of block qb - 1, DMA(qb + 3) x 2.  vmcnt(6) is therefore stricter than necessary by the two oldest
    // stores (issued two blocks ago: free); do NOT read the 6 as the exact count and trim the wait by it.  All of these exist for
    // every block that has a successor (a block with a successor is full: both of its dq stores are issued by every wave),
    // and without them the wait is only stricter.
    if (qb + 1 < nblk) {
      if (qb + QD_NST - 1 < nblk) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    {
      // dQ^T[d][q] = sum_key K^T[d][key] dS^T[key][q]: wave -> d tile, both query tiles of the block
      f32x4 dq[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
      for (int ks = 0; ks < 7; ++ks) {
        const int kr = ks * 32 + g * 8 + tq;
        const bf16x8 kfr = img_tr_frag(k_img, ks * 32, wave * 16, lane);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const char* pb = dst + kr * (DST4_STRIDE * 2) + (i * 16 + tp * 4) * 2;
          const bf16x8 bfr = cat8(lds_tr_read(pb), lds_tr_read(pb + 4 * DST4_STRIDE * 2));
          dq[i] = mfma16(kfr, bfr, dq[i]);
        }
      }
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int q = qb * 32 + i * 16 + lc;
        if (q < NQ) {
          const size_t o = ((size_t)b * NQ + q) * a.dq_rs + h * HD + wave * 16 + g * 4;
          store_grad4(a.dq + o, a.dq_add ? a.dq_add + o : nullptr, dq[i]);
        }
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                       // Y2: dS^T free again; every wave's share of block qb + 1 is in LDS
    asm volatile("" ::: "memory");
  }
  ATTN_STAMP(3);
  // ---- dK, dV of this wave's key tiles through a private fp32 LDS slab: whole 128-byte rows, 16 bytes per lane
  __syncthreads();                                   // every wave has finished reading the images / dS^T
  {
    constexpr int SROW = 272;
    char* slab = smem + wave * (2 * 16 * SROW);
#pragma unroll
    for (int t = 0; t < B4_KT; ++t) {
      const int kt = wave + t * B4_WAVES;
      if (kt >= ntile) break;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        *(f32x4*)(slab + lc * SROW + (dt * 16 + g * 4) * 4) = dk[t][dt];
        *(f32x4*)(slab + 16 * SROW + lc * SROW + (dt * 16 + g * 4) * 4) = dv[t][dt] * gate;
      }
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
#ifdef DEVIT_ATTN_STAMP
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (tid == 0) { stamps[4] = __builtin_amdgcn_s_memtime(); stamps[5] = __builtin_amdgcn_s_memrealtime(); }
#endif
}

constexpr int FWD_LDS = 2 * FWD_IMG + (DEVIT_ATTN_OUT_ROWS ? FWD_WAVES * 16 * OSLAB_ROW : 0);   // 57344 + 18432: two workgroups per CU

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
    hipError_t e = hipFuncSetAttribute((const void*)attn_bwd4_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, BWD4_LDS);
    DEVIT_CHECK(e == hipSuccess, DEVIT_ERR_LAUNCH, "devit_attn_bwd: hipFuncSetAttribute: %s", hipGetErrorString(e));
    attr_set = true;
  }
  hipLaunchKernelGGL(attn_bwd4_kernel, dim3(a.B * a.H), dim3(B4_WAVES * 64), BWD4_LDS, (hipStream_t)stream, a);
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
