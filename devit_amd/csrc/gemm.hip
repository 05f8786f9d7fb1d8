// bf16 MFMA GEMM for gfx950 with fused epilogues.  C[M,N] = sum_k A(m,k) B(n,k).
//
// Tiles 128x128x64 (4 waves, two workgroups per CU) or 256x256x64 (8 waves), each wave a 64x64 / 128x64 sub-tile of
// MFMA 16x16x32 accumulators.  Operand tiles go HBM -> LDS by LDS-DMA (global_load_lds_dwordx4, 1 KiB per
// wave-instruction) into a two-stage ring: the next stage is in flight while the current one is multiplied.  The
// LDS image is lane-linear, so the bank-conflict swizzle is applied to the per-lane SOURCE
// address and undone by the same XOR on the fragment read (cdna_hip_programming.md §5.4 rule 21): swz_row() for
// k-contiguous operands (ds_read_b128 fragments), swz_krow() for k-major operands (ds_read_b64_tr_b16 fragments).
// Epilogues run straight from the accumulators (epilogue_direct): the MFMAs take the weight operand on their row
// side so each lane owns consecutive output columns; only the split-K atomic epilogue stages through LDS.
// Three kernels share the images, swizzles, epilogue code and accumulation order (bit-identical results wherever two of them can run a launch):
// gemm_kernel (eight waves, 128x128 / 256x256 ping-pong), gemm4_kernel (four waves, 256x256, generated asm K loop: gemm4_kloop.inc) and
// gemmfr_kernel (four waves, full-row 256x384 for N = 384 outputs with a k-major weight, generated asm K loop: gemmfr_kloop.inc).
#include <stdlib.h>

#include <type_traits>

#include "devit_common.h"

namespace {

constexpr int BK = 64;
#ifndef DEVIT_PP_SPREAD
#define DEVIT_PP_SPREAD 0        // ping-pong schedule: LDS-DMA issue spread over all four intervals of a K-step (measured: -10 %)
#endif
#ifndef DEVIT_PP_SPLIT
#define DEVIT_PP_SPLIT 1         // ping-pong schedule: B(t+1) issued in the first read interval, A(t+2) in the second
#endif
#ifndef DEVIT_PP_MFMA_AT
#define DEVIT_PP_MFMA_AT 1       // MFMA intervals issue their DMA share after (this + 1) x 4 of their 32 MFMAs
#endif
// cache policy of the operand LDS-DMA (diagnostic switches; measured in DESIGN.md): nt = streaming / evict-first
#ifdef DEVIT_DMA_A_NT
constexpr bool DMA_A_NT = true;
#else
constexpr bool DMA_A_NT = false;
#endif
#ifdef DEVIT_DMA_B_NT
constexpr bool DMA_B_NT = true;
#else
constexpr bool DMA_B_NT = false;
#endif
#ifndef DEVIT_RAGGED_MIN_K
#define DEVIT_RAGGED_MIN_K 384   // plain-store launches take the 256x256 tile from this K on (round 4: student qkv, N 1152 K 384: 86 -> 70 us cold, 75 -> 70 in the step)
#endif

// n / d for 0 <= n < 2^31 by multiply-shift (Granlund-Montgomery round-up): three SALU ops instead of the
// float-reciprocal sequence hipcc emits for a scalar division.  Host-initialised.
struct FastDiv {
  unsigned mul, shift;
  int d;
};
__host__ inline FastDiv make_fastdiv(int d) {
  FastDiv f;
  f.d = d;
  unsigned s = 0;
  while ((1ll << s) < d) ++s;
  f.shift = 31 + s;
  f.mul = (unsigned)(((1ull << f.shift) / (unsigned long long)d) + 1ull);
  return f;
}
__device__ __forceinline__ int fdiv(int n, const FastDiv& f) {
  return (int)(((unsigned long long)(unsigned)n * f.mul) >> f.shift);
}

struct GemmArgs {
  const __bf16* A;
  const __bf16* B;
  int lda, ldb;
  int a_group, a_skip, b_group, b_skip;
  long long a_bs, b_bs;
  int M, N, K;
  int tiles_m, tiles_n, split_k, total_tiles;
  FastDiv d_per_z, d_chunk, d_gn, d_last, d_split;   // tiles per z-slice, per n-chunk; chunk widths; split_k
  int gn;  // n-tiles per L2 chunk: tiles are ordered chunk-major, then m, then n inside the chunk
  devit_epilogue ep;
};

__device__ __forceinline__ int phys_row(int r, int group, int skip) {
  return group > 0 ? r + skip * (r / group + 1) : r;
}

// LDS image swizzles (applied to the DMA's source address and again on the fragment reads; the LDS side of an LDS-DMA
// is lane-linear).  16-byte chunk index XOR:
//   row-major image [rows][64 k], 128-B rows: by row bits (1, 2^4, 3) -- conflict-free ds_read_b128 both for 16
//     consecutive rows and for the PAIRED row set {0-3, 8-11, 16-19, 24-27} (+4 for odd tiles), see tile_row();
//   k-major image [64 k][W], read by ds_read_b64_tr_b16 (lane 4q+p: k-row q, four columns): k-row bits (0,1) go to
//     chunk bits (2,3) and k-row bit 3 to chunk bit 1 -- the sixteen 8-byte pieces of a 16-lane group land on distinct
//     banks whether its four column groups are adjacent (natural) or 16 bytes apart (PAIRED); k-rows r and r+4 share it.
__device__ __forceinline__ int swz_row(int row) { return ((row >> 1) & 7) ^ (((row >> 4) & 1) << 1); }
__device__ __forceinline__ int swz_krow(int krow) { return ((krow & 3) << 2) | (((krow >> 3) & 1) << 1); }

// Issue this wave's LDS-DMA loads (1 KiB each) for one operand tile of width W (128 or 256).
//   KM == false: operand stored [R][K]; `org` = &op[row0][0]; LDS image [W rows][64 k] (128-B rows)
//   KM == true : operand stored [K][R]; `org` = &op[0][col0]; LDS image [64 k][W cols] (2W-B rows)
// The address is split into a wave-uniform base that advances with k0 (SGPRs) and a per-lane 32-bit byte offset
// that is loop-invariant, so the K-loop issues `global_load_lds_dwordx4 voff, s[base]` with no per-step VALU math.
template <bool KM, int W, int NWAVES>
__device__ __forceinline__ unsigned lane_offset(int ld, int wave, int lane, int i, int valid) {
  // `valid` (<= W, a multiple of 8): operand rows (columns if k-major) of this tile that exist; the LDS image rows past
  // them are filled from the last existing one (a ragged last n-tile: their products are never stored)
  constexpr int CNT = (W / 8) / NWAVES;
  const int slab = wave * CNT + i;
  if (!KM) {
    const int row = slab * 8 + (lane >> 3);
    const int chunk = (lane & 7) ^ swz_row(row);
    return (unsigned)(min(row, valid - 1) * ld + chunk * 8) * 2u;
  } else {
    constexpr int LPR = W / 8, RPS = 64 / LPR;  // lanes per k-row, k-rows per 1-KiB slab
    const int krow = slab * RPS + lane / LPR;
    const int chunk = (lane % LPR) ^ swz_krow(krow);
    return (unsigned)(krow * ld + min(chunk, valid / 8 - 1) * 8) * 2u;
  }
}

// Two LDS-DMA instructions (16 B per lane, 1 KiB per wave-instruction) into consecutive 1-KiB slabs at LDS byte
// address `lds`.  Inline asm on purpose: hipcc's waitcnt pass treats a builtin LDS-DMA as a pending LDS write and
// guards later ds_reads with `s_waitcnt vmcnt(0)` whenever it cannot prove the buffers distinct -- which serialises
// the ring (observed: every K-step of some instantiations, every tile boundary of all).  Hidden in asm, the DMA is
// ordered by this kernel's own vmcnt wait + barrier (advance()); hipcc's counts for its own loads/stores stay
// safe because they can only be stricter with extra operations in the queue.  M0 (the DMA's LDS base) is saved and
// restored around the statement; the padding covers SGPR-write -> VMEM-read and M0-write -> LDS-DMA wait states
// (cdna_hip_programming.md §5.7).
template <bool NT = false>
__device__ __forceinline__ void dma2_uniform(const char* ubase, unsigned off0, unsigned off1, unsigned lds) {
  unsigned keep;
  if (NT) {
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %1\n\t"
        "s_nop 2\n\t"
        "global_load_lds_dwordx4 %3, %2 nt\n\t"
        "s_add_u32 m0, %1, 0x400\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %4, %2 nt\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "s"(lds), "s"(ubase), "v"(off0), "v"(off1)
        : "memory", "scc");
    return;
  }
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %1\n\t"
      "s_nop 2\n\t"
      "global_load_lds_dwordx4 %3, %2\n\t"
      "s_add_u32 m0, %1, 0x400\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %4, %2\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "s"(lds), "s"(ubase), "v"(off0), "v"(off1)
      : "memory", "scc");
}
__device__ __forceinline__ void dma2_perlane(const void* p0, const void* p1, unsigned lds) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %1\n\t"
      "s_nop 2\n\t"
      "global_load_lds_dwordx4 %2, off\n\t"
      "s_add_u32 m0, %1, 0x400\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %3, off\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "s"(lds), "v"(p0), "v"(p1)
      : "memory", "scc");
}

// only_pair >= 0: issue just that pair of this wave's slabs (the ping-pong schedule spreads a stage's DMA instructions
// over its barrier intervals)
template <bool KM, int W, int NWAVES, bool NT = false>
__device__ __forceinline__ void stage_tile(const __bf16* org, int ld, int k0, int group, int skip,
                                           char* lds_tile, int wave, int lane, int valid = W, int only_pair = -1) {
  constexpr int CNT = (W / 8) / NWAVES;
  static_assert(CNT % 2 == 0, "slabs are issued in pairs");
  const unsigned lds0 = (unsigned)(size_t)LDS_PTR(lds_tile) + (unsigned)(wave * CNT) * 1024u;
  if (KM && group > 0) {   // row-remapped reduction index (patch-embed wgrad): per-lane physical rows, generic path
#pragma unroll
    for (int i = 0; i < CNT; i += 2) {
      const __bf16* src[2];
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        constexpr int LPR = W / 8, RPS = 64 / LPR;
        const int slab = wave * CNT + i + e;
        const int krow = slab * RPS + lane / LPR;
        const int chunk = (lane % LPR) ^ swz_krow(krow);
        src[e] = org + (size_t)phys_row(k0 + krow, group, skip) * ld + chunk * 8;
      }
      dma2_perlane(src[0], src[1], lds0 + i * 1024u);
    }
    return;
  }
  const char* ubase = (const char*)org + (size_t)k0 * (KM ? (size_t)ld : (size_t)1) * 2;   // wave-uniform
#pragma unroll
  for (int i = 0; i < CNT; i += 2)
    if (only_pair < 0 || i == 2 * only_pair)
    dma2_uniform<NT>(ubase, lane_offset<KM, W, NWAVES>(ld, wave, lane, i, valid), lane_offset<KM, W, NWAVES>(ld, wave, lane, i + 1, valid),
                     lds0 + i * 1024u);
}

// Offset, inside a 64-wide wave tile, of operand row p (0..15) of 16-row tile j.  PAIRED interleaves tiles 2q and
// 2q+1 in groups of four so that, with the operand on the MFMA's row side, lane group g = lane>>4 (which receives
// rows 4g..4g+3 of every tile) ends up with EIGHT consecutive columns of the output per tile pair: one 16-byte bf16
// store.  The natural order gives four consecutive columns per tile: one 16-byte fp32 store.
template <bool PAIRED>
__device__ __forceinline__ int tile_row(int j, int p) {
  return PAIRED ? 32 * (j >> 1) + 8 * (p >> 2) + 4 * (j & 1) + (p & 3) : 16 * j + p;
}

// One MFMA operand fragment: 16-row tile j of the wave's operand rows starting at `base` of the W-wide LDS tile,
// k-step kk of 2.
template <bool KM, int W, bool PAIRED>
__device__ __forceinline__ bf16x8 read_frag(const char* tile, int base, int j, int kk, int lane) {
  if (!KM) {
    const int row = base + tile_row<PAIRED>(j, lane & 15);
#ifdef DEVIT_GEMM_NOSWZ
    const int chunk = (kk * 4 + (lane >> 4));
#else
    const int chunk = (kk * 4 + (lane >> 4)) ^ swz_row(row);
#endif
    return *(const bf16x8*)(tile + row * 128 + chunk * 16);
  } else {
    const int G = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
    const int col0 = base + tile_row<PAIRED>(j, 4 * p);   // this lane addresses operand rows 4p..4p+3 of k-row q
    const int krow = kk * 32 + G * 8 + q;
    const int chunk = (col0 >> 3) ^ swz_krow(krow);
    const char* a = tile + krow * (W * 2) + chunk * 16 + ((col0 >> 2) & 1) * 8;
    return cat8(lds_tr_read(a), lds_tr_read(a + 4 * (W * 2)));
  }
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// Fused epilogue straight from the accumulators.  The K loop runs the MFMAs with the weight operand on the row side,
// so lane (g = lane>>4, c = lane&15) holds, for m-tile i and n-tile j, C[m = 16 i + c][n = tile_row(j, 4g + r)],
// r = 0..3: consecutive output columns in consecutive registers.  Every global access is 16 bytes per lane (8 for the
// optional bf16 copy of the RESIDUAL kind) and the four lane groups of a row cover 64 contiguous bytes.  Rows are
// handled two m-tiles at a time in two phases -- the chunk's global inputs (residual / saved pre-activation /
// pos-embed) first, then compute + stores -- so that no load queues behind the stores of an earlier row (vmcnt counts
// loads and stores in order on gfx950).  No LDS is touched: the operand ring is free while the epilogue runs.
// FULL = every row of the tile is a real row (m < m_lim): no predicates.
// Epilogue output store: plain.  Measured (tools/gemm_bench.py + bench.py A/B): write-through (sc1) stores -12 % on the
// GEMM itself; non-temporal stores +2..7 % on the bf16-output GEMMs and -1..11 % on the fp32 residual ones, and no
// change of the step time (the consumer kernels pay what the producers gain).
template <typename T>
__device__ __forceinline__ void st_out(T* p, T v) {
#ifdef DEVIT_GEMM_NOSTORE   // diagnostic build: the epilogue computes everything and stores (almost) nothing
  if (__builtin_expect(((size_t)p & 0xfffff0) == 0x7ffff0, 0))
#endif
  *p = v;
}

template <bool F16 = false>
__device__ __forceinline__ bf16x8 pack8(const float (&x)[8]) {
  if constexpr (F16) return cvt8<true>((f32x4){x[0], x[1], x[2], x[3]}, (f32x4){x[4], x[5], x[6], x[7]});
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
  bf16x2 p[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) p[e] = __builtin_convertvector((f32x2){x[2 * e], x[2 * e + 1]}, bf16x2);   // v_cvt_pk_bf16_f32
  return (bf16x8){p[0][0], p[0][1], p[1][0], p[1][1], p[2][0], p[2][1], p[3][0], p[3][1]};
}

// Per-lane column data of an epilogue: the lane's four column offsets (one per n-tile) and bias / column scale there.
template <int KIND>
__device__ __forceinline__ void load_cols(const devit_epilogue& ep, int lane, int nw, int (&noff)[4], f32x4 (&bias)[4],
                                          f32x4 (&cs)[4]) {
  constexpr bool BF16_OUT = KIND == DEVIT_EPI_STORE_BF16 || KIND == DEVIT_EPI_GELU_BF16 || KIND == DEVIT_EPI_DGELU_BF16;
  constexpr bool SCALED = KIND == DEVIT_EPI_GELU_BF16 || KIND == DEVIT_EPI_DGELU_BF16;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    noff[j] = nw + tile_row<BF16_OUT>(j, 4 * (lane >> 4));
    // (the dGELU kind is a dgrad: no bias by contract, checked on the host -- 16 registers the 255-VGPR 256x256 instantiation
    // does not have: it spilled 4 to scratch with them, and scratch reloads inside a K-step land in the counted vmcnt waits)
    if constexpr (KIND == DEVIT_EPI_DGELU_BF16) bias[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    else bias[j] = ep.bias ? *(const f32x4*)(ep.bias + noff[j]) : (f32x4){0.f, 0.f, 0.f, 0.f};
    if (SCALED) cs[j] = ep.colscale ? *(const f32x4*)(ep.colscale + noff[j]) : (f32x4){1.f, 1.f, 1.f, 1.f};
  }
}
// Make hipcc wait for the loads of load_cols() HERE (an empty asm that reads them).  Its waitcnt pass does not see
// the asm LDS-DMA: a wait it places after the next DMA issue would also wait for that DMA.
template <int KIND>
__device__ __forceinline__ void settle_cols(f32x4 (&bias)[4], f32x4 (&cs)[4]) {
  constexpr bool SCALED = KIND == DEVIT_EPI_GELU_BF16 || KIND == DEVIT_EPI_DGELU_BF16;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    if (KIND != DEVIT_EPI_DGELU_BF16) asm volatile("" : "+v"(bias[j]));
    if (SCALED) asm volatile("" : "+v"(cs[j]));
  }
}

#ifndef DEVIT_F32_FULL_ROWS
#define DEVIT_F32_FULL_ROWS 1    // fp32 epilogues move whole 128-byte rows per instruction (0: 64-byte halves, the earlier form)
#endif

// The fp32-output kinds (STORE_F32, PATCH_F32, RESIDUAL_F32) with whole 128-byte rows per memory instruction.  Lane (g, c)
// holds, per n-tile j, four consecutive floats of row c at column 16 j + 4 g: one instruction per n-tile touches 16 rows x
// 64 bytes -- half cache lines, twice the transactions of the bytes moved, on an epilogue that runs at the CU's transaction
// rate.  Here the n-tiles (2 p, 2 p + 1) of a row are one 128-byte line: lanes c and c ^ 8 trade a chunk (swap_half_rows)
// so that instruction A covers rows 0-7 and instruction B rows 8-15 of the m-tile, every row a whole line; the residual /
// pos-embed inputs are fetched in that same shape.  Same arithmetic per element as the 64-byte form.
template <int KIND, int MI, bool FULL, bool F16>
__device__ __forceinline__ void epilogue_f32_rows(const devit_epilogue& ep, f32x4 (&acc)[MI][4], const int (&noff)[4],
                                                  const f32x4 (&bias)[4], int lane, int mw, int m_lim, size_t ob) {
  constexpr int CHI = 2;
  const int c = lane & 15;
  const bool hi = c >= 8;
  const int colx = hi ? 16 : 0;                     // floats: second half of the line
  auto row_off = [&](int m, int& tok) -> size_t {   // element offset of output row m (PATCH: token remap)
    if (KIND == DEVIT_EPI_PATCH_F32) {
      const int b = m / ep.patch_tokens, t = m - b * ep.patch_tokens;
      tok = ep.extra_tokens + t;
      return ((size_t)b * (ep.patch_tokens + ep.extra_tokens) + tok) * ep.ldc;
    }
    tok = 0;
    return ob + (size_t)m * ep.ldc;
  };
#pragma unroll
  for (int i0 = 0; i0 < MI; i0 += CHI) {
    // ---- phase 1: global inputs of the chunk, in the shape they will be stored in
    f32x4 gin[CHI][2][2];          // [m-tile][line p][rows 0-7 / 8-15]
    float rsc[CHI][2];
    size_t rowo[CHI][2];
    bool okr[CHI][2];
#pragma unroll
    for (int u = 0; u < CHI; ++u)
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        const int m = mw + (i0 + u) * 16 + (c & 7) + 8 * r;
        const bool ok = FULL || m < m_lim;
        int tok;
        const size_t o = row_off(m, tok);
#pragma unroll
        for (int p = 0; p < 2; ++p) {
          const int col = noff[2 * p] + colx;
          if (KIND == DEVIT_EPI_PATCH_F32)
            gin[u][p][r] = ok ? *(const f32x4*)(ep.pos + (size_t)tok * ep.ldc + col) : (f32x4){0.f, 0.f, 0.f, 0.f};
          if (KIND == DEVIT_EPI_RESIDUAL_F32)
            gin[u][p][r] = ok ? load_stream((const f32x4*)(ep.res + o + col)) : (f32x4){0.f, 0.f, 0.f, 0.f};
        }
        if (KIND == DEVIT_EPI_RESIDUAL_F32) rsc[u][r] = (ok && ep.rowscale) ? ep.rowscale[m / ep.rows_per_scale] : 1.0f;
        rowo[u][r] = o;
        okr[u][r] = ok;
      }
    // ---- phase 2: compute + store
#pragma unroll
    for (int u = 0; u < CHI; ++u) {
      const int i = i0 + u;
      f32x4 v[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = acc[i][j] + bias[j];
      if (KIND == DEVIT_EPI_RESIDUAL_F32 && ep.aux) {   // optional bf16 copy of the branch output (output_att): row c, 8 bytes per lane
        const int m = mw + i * 16 + c;
        if (FULL || m < m_lim) {
#pragma unroll
          for (int j = 0; j < 4; ++j) st_out((bf16x4*)((__bf16*)ep.aux + ob + (size_t)m * ep.ldc + noff[j]), cvt4<F16>(v[j]));
        }
      }
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        swap_half_rows(v[2 * p], v[2 * p + 1], hi);
        const int col = noff[2 * p] + colx;
#pragma unroll
        for (int r = 0; r < 2; ++r) {
          f32x4 w = v[2 * p + r];
          if (KIND == DEVIT_EPI_PATCH_F32) w = w + gin[u][p][r];
          if (KIND == DEVIT_EPI_RESIDUAL_F32) w = gin[u][p][r] + rsc[u][r] * w;
          if (okr[u][r]) st_out((f32x4*)((float*)ep.out + rowo[u][r] + col), w);
        }
      }
    }
  }
}

template <int KIND, int MI, bool FULL, bool F16 = false>
__device__ __forceinline__ void epilogue_direct(const devit_epilogue& ep, f32x4 (&acc)[MI][4], const int (&noff)[4],
                                                const f32x4 (&bias)[4], const f32x4 (&cs)[4], int lane, int mw,
                                                int m_lim, size_t ob) {
  constexpr bool BF16_OUT = KIND == DEVIT_EPI_STORE_BF16 || KIND == DEVIT_EPI_GELU_BF16 || KIND == DEVIT_EPI_DGELU_BF16;
  if constexpr (!BF16_OUT && DEVIT_F32_FULL_ROWS) {
    epilogue_f32_rows<KIND, MI, FULL, F16>(ep, acc, noff, bias, lane, mw, m_lim, ob);
    return;
  }
  constexpr int CHI = 2;
  const int c = lane & 15;
#pragma unroll
  for (int i0 = 0; i0 < MI; i0 += CHI) {
    // ---- phase 1: global inputs of the chunk
    f32x4 gin[CHI][4];
    float rsc[CHI];
    bf16x8 pre[CHI][2];
    size_t rowo[CHI];
#pragma unroll
    for (int u = 0; u < CHI; ++u) {
      const int m = mw + (i0 + u) * 16 + c;
      const bool ok = FULL || m < m_lim;
      size_t o = ob + (size_t)m * ep.ldc;
      if (KIND == DEVIT_EPI_PATCH_F32) {
        const int b = m / ep.patch_tokens, t = m - b * ep.patch_tokens, tok = ep.extra_tokens + t;
        o = ((size_t)b * (ep.patch_tokens + ep.extra_tokens) + tok) * ep.ldc;
#pragma unroll
        for (int j = 0; j < 4; ++j)
          gin[u][j] = ok ? *(const f32x4*)(ep.pos + (size_t)tok * ep.ldc + noff[j]) : (f32x4){0.f, 0.f, 0.f, 0.f};
      }
      if (KIND == DEVIT_EPI_RESIDUAL_F32) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          gin[u][j] = ok ? load_stream((const f32x4*)(ep.res + o + noff[j])) : (f32x4){0.f, 0.f, 0.f, 0.f};
        rsc[u] = (ok && ep.rowscale) ? ep.rowscale[m / ep.rows_per_scale] : 1.0f;
      }
      if (KIND == DEVIT_EPI_DGELU_BF16) {
        const bf16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int q = 0; q < 2; ++q)
          pre[u][q] = ok ? load_stream((const bf16x8*)((const __bf16*)ep.aux_in + o + noff[2 * q])) : z;
      }
      rowo[u] = o;
    }
    // ---- phase 2: compute + store
#pragma unroll
    for (int u = 0; u < CHI; ++u) {
      const int i = i0 + u;
      const bool ok = FULL || mw + i * 16 + c < m_lim;
      const size_t o = rowo[u];
      if constexpr (!BF16_OUT) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const f32x4 v = acc[i][j] + bias[j];
          if (KIND == DEVIT_EPI_STORE_F32) {
            if (ok) st_out((f32x4*)((float*)ep.out + o + noff[j]), v);
          } else if (KIND == DEVIT_EPI_PATCH_F32) {
            if (ok) st_out((f32x4*)((float*)ep.out + o + noff[j]), v + gin[u][j]);
          } else {  // RESIDUAL_F32
            if (ok) {
              if (ep.aux) {
                const bf16x4 pb = cvt4<F16>(v);
                st_out((bf16x4*)((__bf16*)ep.aux + o + noff[j]), pb);
              }
              st_out((f32x4*)((float*)ep.out + o + noff[j]), gin[u][j] + rsc[u] * v);
            }
          }
        }
      } else {
        bf16x8 outc[2], prec[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          float x[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) x[e] = acc[i][2 * q + (e >> 2)][e & 3] + bias[2 * q + (e >> 2)][e & 3];
          if (KIND == DEVIT_EPI_GELU_BF16) {
            prec[q] = pack8<F16>(x);
#pragma unroll
            for (int e = 0; e < 8; ++e) x[e] = gelu_fwd<false>(x[e]) * cs[2 * q + (e >> 2)][e & 3];
          } else if (KIND == DEVIT_EPI_DGELU_BF16) {
#pragma unroll
            for (int e = 0; e < 8; ++e) x[e] = x[e] * cs[2 * q + (e >> 2)][e & 3] * gelu_bwd<false>(bf2f(pre[u][q][e]));
          }
          outc[q] = pack8<F16>(x);
        }
        // whole 128-byte rows per store: rows (c & 7) and 8 + (c & 7) of this m-tile, see swap_half_rows()
        const bool hi = c >= 8;
        const int mA = mw + i * 16 + (c & 7);
        const size_t oA = ob + (size_t)mA * ep.ldc + noff[0] + (hi ? 32 : 0), oB = oA + (size_t)8 * ep.ldc;
        const bool okA = FULL || mA < m_lim, okB = FULL || mA + 8 < m_lim;
        swap_half_rows(outc[0], outc[1], hi);
        if (KIND == DEVIT_EPI_GELU_BF16 && ep.aux) {
          swap_half_rows(prec[0], prec[1], hi);
          if (okA) st_out((bf16x8*)((__bf16*)ep.aux + oA), prec[0]);
          if (okB) st_out((bf16x8*)((__bf16*)ep.aux + oB), prec[1]);
        }
        if (okA) st_out((bf16x8*)((__bf16*)ep.out + oA), outc[0]);
        if (okB) st_out((bf16x8*)((__bf16*)ep.out + oB), outc[1]);
      }
    }
  }
}

// Where one output tile's operands start and which K-steps it covers (all wave-uniform).
// acc + sum of the eight bf16 values of one MFMA fragment (four v_dot2c_f32_bf16 against packed ones)
__device__ __forceinline__ float sum8_bf16(bf16x8 v, float acc) {
  typedef __bf16 bf16x2v __attribute__((ext_vector_type(2)));
  const bf16x2v one = {(__bf16)1.0f, (__bf16)1.0f};
  acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(v, v, 0, 1), one, acc, false);
  acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(v, v, 2, 3), one, acc, false);
  acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(v, v, 4, 5), one, acc, false);
  acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(v, v, 6, 7), one, acc, false);
  return acc;
}

struct TileRef {
  const __bf16* a;
  const __bf16* b;
  int m0, n0, bz, kt0, nk;
};

template <int BM, int BN, bool A_KM, bool B_KM>
__device__ __forceinline__ TileRef decode_tile(const GemmArgs& g, int w) {
  const int zz = fdiv(w, g.d_per_z);
  const int r0 = w - zz * g.d_per_z.d;
  const int chunk = fdiv(r0, g.d_chunk);                // full chunks (gn n-tiles x all m-tiles) come first
  const int r1 = r0 - chunk * g.d_chunk.d;
  const bool lastc = (chunk + 1) * g.gn > g.tiles_n;    // the last chunk may be narrower
  const int tm = lastc ? fdiv(r1, g.d_last) : fdiv(r1, g.d_gn);
  const int tn = chunk * g.gn + r1 - tm * (lastc ? g.d_last.d : g.d_gn.d);
  const int bzq = fdiv(zz, g.d_split);
  const int z = zz - bzq * g.split_k, nk_total = g.K / BK;
  TileRef t;
  t.bz = bzq;
  t.m0 = tm * BM;
  t.n0 = tn * BN;
  t.kt0 = fdiv(z * nk_total, g.d_split);
  t.nk = fdiv((z + 1) * nk_total, g.d_split) - t.kt0;
  t.a = g.A + (size_t)t.bz * g.a_bs + (A_KM ? (size_t)t.m0 : (size_t)t.m0 * g.lda);
  t.b = g.B + (size_t)t.bz * g.b_bs + (B_KM ? (size_t)t.n0 : (size_t)t.n0 * g.ldb);
  return t;
}

// Persistent BM x BN x 64 GEMM: WAVES_M x WAVES_N waves (each (BM/WAVES_M) x (BN/WAVES_N)), LDS ring (3 A + 2 B slots)
// filled by LDS-DMA.  A workgroup walks its share of the output tiles and treats their K-steps as ONE stream of ring
// stages: the refill issued during the last K-steps of a tile already belongs to the next tile, so neither the
// DMA latency of a tile's first stages nor the register-only epilogue leaves the ring empty.  One raw barrier per
// K-step (128x128) or the four-interval ping-pong schedule below (256x256).
// Tile order: workgroups b, b+8, ... share an XCD (and its L2); each XCD owns a contiguous run of tiles (n-tile
// fastest inside an L2-sized chunk of B, see decode_tile) and its workgroups walk that run side by side.
template <int BM, int BN, int WAVES_M, int WAVES_N, int NSTAGE, bool A_KM, bool B_KM, int KIND, bool F16 = false>
__global__ __launch_bounds__(WAVES_M * WAVES_N * 64) __attribute__((amdgpu_waves_per_eu(2, 2)))
void gemm_kernel(const GemmArgs g) {
  static_assert(!F16 || (KIND != DEVIT_EPI_DGELU_BF16 && KIND != DEVIT_EPI_ATOMIC_F32 && !A_KM && !B_KM),
                "f16 operands: forward layouts / epilogues only (the frozen teacher has no backward)");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NWAVES = WAVES_M * WAVES_N;
  constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N, MI = WM / 16, NI = WN / 16;
  static_assert(WN == 64 && WM % 64 == 0, "wave tile must be (64 k) x 64");
  static_assert(NSTAGE == 2, "NSTAGE is the B ring depth; uniformly deeper rings at one workgroup per CU lost everywhere");
  // Every epilogue but the split-K atomic one runs straight from the accumulators: the MFMAs then take the B operand
  // (output columns) on their row side, see epilogue_direct().  The atomic one stages through the ring's LDS, so its
  // stream stops at every tile end.
  constexpr bool DIRECT = KIND != DEVIT_EPI_ATOMIC_F32;
  constexpr bool PAIRED = KIND == DEVIT_EPI_STORE_BF16 || KIND == DEVIT_EPI_GELU_BF16 || KIND == DEVIT_EPI_DGELU_BF16;
  constexpr int A_TILE_BYTES = BM * BK * 2, B_TILE_BYTES = BN * BK * 2;
  // Ring: two B slots, THREE A slots.  A is the operand that streams from HBM (an activation; B is a weight that lives in
  // L2 -- or, in the weight-gradient GEMMs, the narrower activation), so its stages are requested one K-step earlier:
  // two A stages in flight per workgroup at unchanged occupancy (128x128: 2 x 80 KB, 256x256: 160 KB = the whole LDS).
  // Cold-HBM operands: +4...12 % on the 128x128 shapes (tools/gemm_bench.py COLD=1), -1...3 % on cache-resident ones.
  constexpr bool PP = DIRECT && BM == 256 && BN == 256 && WAVES_M == 2;   // ping-pong schedule, below
  constexpr int NA = 3;                                          // A slots (B has 2)
  constexpr int PER_A = (BM / 8) / NWAVES;                       // LDS-DMA instructions per wave per A stage
  constexpr int B_RING = NA * A_TILE_BYTES;                      // LDS offset of the B slots
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
#ifdef DEVIT_GEMM_TSTAMP
  const unsigned long long t_entry = __builtin_amdgcn_s_memtime(), rt_entry = __builtin_amdgcn_s_memrealtime();
#endif

  // this workgroup's tiles: first, first + stride, ... < last
  const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3, stride = gridDim.x >> 3;
  int first, last;
  {
    const int q = g.total_tiles >> 3, r = g.total_tiles & 7;
    const int start = xcd * q + min(xcd, r);
    first = start + idx;
    last = start + q + (xcd < r ? 1 : 0);
  }
  if (first >= last) return;

  // producer cursors: the next B stage to request (p) and, on the 3-slot A ring, the next A stage (q = p + 1 stage)
  struct Cursor {
    TileRef ref;
    int tile, t;
    bool open;
  };
  Cursor pb{decode_tile<BM, BN, A_KM, B_KM>(g, first), first, 0, true};
  Cursor pa = pb;
  int a_slot = 0, b_slot = 0;
  bool a_last = false;   // the newest vector-memory operations of this wave are the PER_A DMAs of an A stage
  auto step_cursor = [&](Cursor& c) {
    if (++c.t == c.ref.nk) {
      c.t = 0;
      c.tile += stride;
      if (DIRECT && c.tile < last) c.ref = decode_tile<BM, BN, A_KM, B_KM>(g, c.tile);
      else c.open = false;
    }
  };
  auto dma_a = [&](const Cursor& c) {
#ifdef DEVIT_GEMM_NODMA       // diagnostic build: MFMAs + LDS reads alone (operands are whatever the LDS holds)
    if (g.K < 0)
#endif
    stage_tile<A_KM, BM, NWAVES, DMA_A_NT>(c.ref.a, g.lda, (c.ref.kt0 + c.t) * BK, g.a_group, g.a_skip,
                                           smem + a_slot * A_TILE_BYTES, wave, lane);
    a_slot = a_slot + 1 == NA ? 0 : a_slot + 1;
  };
  auto dma_b = [&](const Cursor& c) {
#ifdef DEVIT_GEMM_NODMA
    if (g.K < 0)
#endif
    stage_tile<B_KM, BN, NWAVES, DMA_B_NT>(c.ref.b, g.ldb, (c.ref.kt0 + c.t) * BK, g.b_group, g.b_skip,
                                           smem + B_RING + b_slot * B_TILE_BYTES, wave, lane, min(BN, g.N - c.ref.n0));
    b_slot ^= 1;
  };
  auto issue_a = [&]() {
    a_last = pa.open;
    if (!pa.open) return;
    dma_a(pa);
    step_cursor(pa);
  };
  // One refill: B of the next stage first, then A of the stage after it.  The A request is the newest thing in the queue, so "everything but PER_A operations has completed"
  // (wait_stage) means: the stage about to be read has landed, the A stage after it may still be in flight.
  auto produce = [&]() {
    if (pb.open) {
      dma_b(pb);
      step_cursor(pb);
    }
    issue_a();
  };
  auto wait_stage = [&]() {
    if (a_last) wait_vmcnt<PER_A>();
    else wait_vmcnt<0>();
  };
  // ---------------------------------------------------------------------------------------------------------------
  // Ping-pong schedule (256x256 tile: the two waves of a SIMD are wm = 0 and wm = 1 of the SAME workgroup).  With one
  // barrier per K-step all eight waves read their fragments at the same time (96 KB through the LDS while every MFMA
  // pipe idles) and then all issue MFMAs at the same time: SQ_VALU_MFMA_BUSY_CYCLES showed the pipes 51 % busy with
  // the DMA removed.  Here a K-step is four barrier intervals -- reads(kk=0) | MFMA(0) | reads(1) | MFMA(1) -- and the
  // wm = 1 waves run ONE interval behind the wm = 0 waves (one extra barrier before a tile's first K-step, one extra
  // for wm = 0 after its last): in every interval one wave of each SIMD issues its 32 MFMAs while the other reads its next fragments.
  //   interval a: request B of stage t+1 and A of stage t+2 (the slots read in step t-1), ds_read fragments kk = 0, lgkmcnt(0)
  //   interval b: MFMA kk = 0
  //   interval c: ds_read fragments kk = 1, lgkmcnt(0), counted vmcnt (this wave's share of stage t+1 has landed)
  //   interval d: MFMA kk = 1
  // RAW: a wave reads stage t+1 after its barrier Y1(t); the lagging group passed its own vmcnt(0) before its X1(t),
  // which is the same barrier event.  WAR: stage t+1's slot was last read in interval c of step t-1, retired by the
  // lgkmcnt(0) in front of X1(t-1), at least one barrier event before any wave requests stage t+1.
  if constexpr (PP) {
    auto fence = [&]() {
      asm volatile("" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
    };
    auto bar = [&]() {
      fence();
      __builtin_amdgcn_s_barrier();
      fence();
    };
    issue_a();                         // A of stage 0, then B of stage 0 and A of stage 1
    produce();
    wait_vmcnt<0>();
    bar();
#if DEVIT_PP_SPREAD
    // Spread schedule.  A K-step is four barrier intervals per group -- reads(kk=0) | MFMA(0) | reads(1) | MFMA(1) -- and
    // the group wm = 1 runs one interval behind.  Issuing a stage's eight LDS-DMA instructions in ONE read interval
    // (as `produce()` does) makes that interval ~850 cycles against the partner's 560 cycles of MFMAs (in-kernel stamps,
    // tools/gemm_stamps.py).  Here the instructions are dealt over four ABSOLUTE slots q = (interval index) mod 4, the
    // same slot for both groups at the same time (the lagging group is in its interval q - 1):
    //   q = 0: first half of B(t+1)    q = 1: second half of B(t+1)    q = 2: first half of A(t+2)
    //   q = 3: second half of A(t+2), then the counted wait that makes stage t+1 readable (all but A(t+2) landed)
    // RAW: the leading group reads stage t+1 after the barrier that ends slot 3 -- every wave waited inside slot 3.
    // WAR: B(t+1) overwrites B(t-1), last read by the lagging group in slot 3 of the previous K-step (its reads(1)),
    //      retired by its lgkmcnt(0) in front of the barrier that ends that slot; A(t+2) overwrites A(t-1), older still.
    auto dma_half = [&](Cursor& c, bool is_a, int h) {
      if (is_a) {
        stage_tile<A_KM, BM, NWAVES, DMA_A_NT>(c.ref.a, g.lda, (c.ref.kt0 + c.t) * BK, g.a_group, g.a_skip,
                                               smem + a_slot * A_TILE_BYTES, wave, lane, BM, h);
        if (h == 1) a_slot = a_slot + 1 == NA ? 0 : a_slot + 1;
      } else {
        stage_tile<B_KM, BN, NWAVES, DMA_B_NT>(c.ref.b, g.ldb, (c.ref.kt0 + c.t) * BK, g.b_group, g.b_skip,
                                               smem + B_RING + b_slot * B_TILE_BYTES, wave, lane, min(BN, g.N - c.ref.n0), h);
        if (h == 1) b_slot ^= 1;
      }
      if (h == 1) step_cursor(c);
    };
    auto slot_q = [&](auto qc) {
      constexpr int q = decltype(qc)::value;
      if (q == 0) { if (pb.open) dma_half(pb, false, 0); }
      if (q == 1) { if (pb.open) dma_half(pb, false, 1); }
      if (q == 2) { a_last = pa.open; if (pa.open) dma_half(pa, true, 0); }
      if (q == 3) { if (pa.open) dma_half(pa, true, 1); wait_stage(); }
    };
    // called by a wave in ITS interval p (0..3): the leading group is in absolute slot p, the lagging one in slot p + 1
    auto spread_slot = [&](auto pc) {
      constexpr int p = decltype(pc)::value;
      if (wm == 0) slot_q(std::integral_constant<int, p>());
      else slot_q(std::integral_constant<int, (p + 1) & 3>());
    };
    if (wm == 1) slot_q(std::integral_constant<int, 0>());   // the lagging group's slot 0 of K-step 0 lies before its first interval
#endif
    int ca_slot = 0, cb_slot = 0;
#ifdef DEVIT_GEMM_TSTAMP   // diagnostic build: per tile {K loop start, K loop end, epilogue end}, per K-step end of the 2nd tile
    unsigned long long* tdbg = g.ep.pos ? (unsigned long long*)g.ep.pos + ((size_t)blockIdx.x * NWAVES + wave) * 48 : nullptr;
    int tcount = 0;
    if (tdbg && lane == 0) {
      tdbg[40] = t_entry;
      tdbg[42] = rt_entry;
      tdbg[44] = __builtin_amdgcn_s_memtime();      // ring primed: first K loop can start
    }
#endif
    for (int tile = first; tile < last; tile += stride) {
      const TileRef ct = decode_tile<BM, BN, A_KM, B_KM>(g, tile);
      f32x4 acc[MI][NI];
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
      const devit_epilogue& ep = g.ep;
      const int nw = ct.n0 + wn * WN;
#ifdef DEVIT_GEMM_TSTAMP
      if (tdbg && lane == 0 && tcount < 8) tdbg[tcount * 3 + 0] = __builtin_amdgcn_s_memtime();
#endif
      int noff[4];
      f32x4 bias[4], cs[4];
#ifdef DEVIT_GEMM_STAMP    // diagnostic build: s_memtime at the edges of the four intervals of K-step 3 of the first tile
      unsigned long long stamp[12];
#pragma unroll
      for (int q = 0; q < 12; ++q) stamp[q] = 0;
#define DEVIT_STAMP(q) do { if (t == 3 && tile == first) stamp[q] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define DEVIT_STAMP(q) do { } while (0)
#endif
      if (wm == 1) bar();              // the offset: this group now runs one interval behind
      for (int t = 0; t < ct.nk; ++t) {
        const char* cur_a = smem + ca_slot * A_TILE_BYTES;
        const char* cur_b = smem + B_RING + cb_slot * B_TILE_BYTES;
        ca_slot = ca_slot + 1 == NA ? 0 : ca_slot + 1;
        cb_slot ^= 1;
#if !DEVIT_PP_SPREAD && !DEVIT_PP_SPLIT
        produce();                     // B of stage t+1, A of stage t+2
#endif
        if (t == ct.nk - 1 && nw < g.N) load_cols<KIND>(ep, lane, nw, noff, bias, cs);   // under the last K-step
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
          bf16x8 af[MI], bfr[NI];
#ifdef DEVIT_GEMM_NOREAD     // diagnostic build: no fragment reads (operands are whatever the registers hold)
#pragma unroll
          for (int j = 0; j < NI; ++j) asm volatile("" : "=v"(bfr[j]));
#pragma unroll
          for (int i = 0; i < MI; ++i) asm volatile("" : "=v"(af[i]));
#else
#pragma unroll
          for (int j = 0; j < NI; ++j) bfr[j] = read_frag<B_KM, BN, PAIRED>(cur_b, wn * WN, j, kk, lane);
#pragma unroll
          for (int i = 0; i < MI; ++i) af[i] = read_frag<A_KM, BM, false>(cur_a, wm * WM, i, kk, lane);
#endif
          DEVIT_STAMP(kk * 6 + 0);     // fragment reads issued
#if DEVIT_PP_SPREAD
          fence();
          if (kk == 0) spread_slot(std::integral_constant<int, 0>());   // read interval of kk = 0: absolute slot 0 / 1
          else spread_slot(std::integral_constant<int, 2>());          // read interval of kk = 1: slot 2 / 3
          fence();
#elif DEVIT_PP_SPLIT
          // The stage's eight LDS-DMA instructions per wave split over the two read intervals (all eight in the first one
          // made it ~1050 cycles against the partner's 560 cycles of MFMAs; in-kernel stamps, tools/gemm_stamps.py): B of
          // stage t+1 behind the kk = 0 reads, A of stage t+2 behind the kk = 1 reads, then the counted wait (everything
          // but that A request has landed -> stage t+1 readable after the next barrier).
          fence();
          if (kk == 0) {
            if (pb.open) {
              dma_b(pb);
              step_cursor(pb);
            }
          } else {
            issue_a();
            wait_stage();
          }
          fence();
#else
          if (kk == 1) wait_stage();   // stage t+1 has landed; A of stage t+2 may stay in flight
#endif
          DEVIT_STAMP(kk * 6 + 1);     // DMA wait over
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          DEVIT_STAMP(kk * 6 + 2);     // fragments in registers
          bar();
          DEVIT_STAMP(kk * 6 + 3);     // through the barrier
#ifdef DEVIT_GEMM_NOMFMA     // diagnostic build: fragment reads and barriers alone
#pragma unroll
          for (int j = 0; j < NI; ++j) asm volatile("" ::"v"(bfr[j]));
#pragma unroll
          for (int i = 0; i < MI; ++i) asm volatile("" ::"v"(af[i]));
#else
#pragma unroll
          for (int i = 0; i < MI; ++i) {
#pragma unroll
            for (int j = 0; j < NI; ++j) acc[i][j] = mfma16t<F16>(bfr[j], af[i], acc[i][j]);
#if DEVIT_PP_SPREAD
            if (i == DEVIT_PP_MFMA_AT) {   // this interval's share of the DMA issue, behind the first MFMAs
              fence();
              if (kk == 0) spread_slot(std::integral_constant<int, 1>());
              else spread_slot(std::integral_constant<int, 3>());
              fence();
            }
#endif
          }
#endif
          DEVIT_STAMP(kk * 6 + 4);     // MFMAs issued
          bar();
          DEVIT_STAMP(kk * 6 + 5);     // through the barrier
        }
#ifdef DEVIT_GEMM_TSTAMP
        if (tdbg && lane == 0 && tcount == 1 && t < 16 && (g.M & 1)) tdbg[24 + t] = __builtin_amdgcn_s_memtime();
#endif
      }
#ifdef DEVIT_GEMM_TSTAMP
      if (tdbg && lane == 0 && tcount < 8) tdbg[tcount * 3 + 1] = __builtin_amdgcn_s_memtime();
#endif
      if (wm == 0) bar();              // pairs with the lagging group's last barrier: both groups run the epilogue
#ifdef DEVIT_GEMM_STAMP
      if (tile == first && lane == 0 && g.ep.pos != nullptr) {
        unsigned long long* dbg = (unsigned long long*)g.ep.pos + ((size_t)blockIdx.x * NWAVES + wave) * 12;
#pragma unroll
        for (int q = 0; q < 12; ++q) dbg[q] = stamp[q];
      }
#endif
      if (nw < g.N) {                  // (a wave whose 64 columns lie past a ragged N has nothing to store)
        settle_cols<KIND>(bias, cs);   // together (one after the other would double its MFMA-idle time)
        const size_t ob = (size_t)ct.bz * ep.out_batch_stride;
        const int m_lim = ep.m_valid > 0 ? ep.m_valid : g.M;
        if (ct.m0 + BM <= m_lim) epilogue_direct<KIND, MI, true, F16>(ep, acc, noff, bias, cs, lane, ct.m0 + wm * WM, m_lim, ob);
        else epilogue_direct<KIND, MI, false, F16>(ep, acc, noff, bias, cs, lane, ct.m0 + wm * WM, m_lim, ob);
      }
#ifdef DEVIT_GEMM_TSTAMP
      if (tdbg && lane == 0 && tcount < 8) tdbg[tcount * 3 + 2] = __builtin_amdgcn_s_memtime();
      ++tcount;
#endif
    }
#ifdef DEVIT_GEMM_TSTAMP
    if (tdbg && lane == 0) {
      tdbg[41] = __builtin_amdgcn_s_memtime();
      tdbg[43] = __builtin_amdgcn_s_memrealtime();
      tdbg[45] = tcount;
    }
#endif
    return;
  }

  issue_a();   // A of the first stage, then B of the first stage and A of the second
  produce();

  // Make the next stage readable: it must have landed, every wave must know so and must have finished reading the
  // slots the refill overwrites (the ones read a step ago).
  auto advance = [&]() {
    wait_stage();
    __builtin_amdgcn_s_barrier();
    produce();
  };

  int ca_slot = 0, cb_slot = 0;
  bool primed = false;   // the stage at c_slot is already readable (advance() ran for it before the last epilogue)
  for (int tile = first; tile < last; tile += stride) {
    const TileRef ct = decode_tile<BM, BN, A_KM, B_KM>(g, tile);
    f32x4 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NI; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // Split-K weight gradient: the row sums of A (= dY^T, i.e. the bias gradient) come from the fragments the MFMAs
    // read anyway -- one v_dot2c_f32_bf16 per two elements, on the waves that own the first 64 columns of their tile.
    // The n-tiles of one (m-tile, k-slice) see the same A rows, so they share the work: tile tn takes the K-steps
    // kt with kt % tiles_n == tn (every A element is added exactly once).
    float rsum[MI];
#pragma unroll
    for (int i = 0; i < MI; ++i) rsum[i] = 0.f;
    const bool rowsum_on = KIND == DEVIT_EPI_ATOMIC_F32 && g.ep.aux != nullptr && wn == 0;
    int rs_wait = 0;          // K-steps until this tile's next turn
    if (rowsum_on) rs_wait = (ct.n0 / BN - ct.kt0 % g.tiles_n + g.tiles_n) % g.tiles_n;

    auto kstep = [&]() {
      const char* cur_a = smem + ca_slot * A_TILE_BYTES;
      const char* cur_b = smem + B_RING + cb_slot * B_TILE_BYTES;
      ca_slot = ca_slot + 1 == NA ? 0 : ca_slot + 1;
      cb_slot ^= 1;
      const bool rs_now = KIND == DEVIT_EPI_ATOMIC_F32 && rowsum_on && rs_wait == 0;
#ifdef DEVIT_GEMM_NOCOMPUTE   // diagnostic build: the fill pipeline alone
      if (g.K < 0)
#endif
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        bf16x8 af[MI], bfr[NI];
#pragma unroll
        for (int j = 0; j < NI; ++j) bfr[j] = read_frag<B_KM, BN, PAIRED>(cur_b, wn * WN, j, kk, lane);
#pragma unroll
        for (int i = 0; i < MI; ++i) af[i] = read_frag<A_KM, BM, false>(cur_a, wm * WM, i, kk, lane);
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int j = 0; j < NI; ++j)
            acc[i][j] = DIRECT ? mfma16t<F16>(bfr[j], af[i], acc[i][j]) : mfma16t<F16>(af[i], bfr[j], acc[i][j]);
        if constexpr (KIND == DEVIT_EPI_ATOMIC_F32) {
          if (rs_now) {
#pragma unroll
            for (int i = 0; i < MI; ++i) rsum[i] = sum8_bf16(af[i], rsum[i]);
          }
        }
      }
      if constexpr (KIND == DEVIT_EPI_ATOMIC_F32) rs_wait = rs_wait == 0 ? g.tiles_n - 1 : rs_wait - 1;
    };
    const devit_epilogue& ep = g.ep;
    const int nw = ct.n0 + wn * WN;
    for (int t = 0; t < ct.nk - 1; ++t) {
      if (t > 0 || !primed) advance();
      kstep();
    }
    // last K-step of the tile: the epilogue's column data (bias, column scale) is fetched under its MFMAs
    if (ct.nk > 1 || !primed) advance();
    int noff[4];
    f32x4 bias[4], cs[4];
    if constexpr (DIRECT) load_cols<KIND>(ep, lane, nw, noff, bias, cs);
    kstep();
    if constexpr (DIRECT) settle_cols<KIND>(bias, cs);
    // The next tile's first stage is made readable BEFORE this tile's epilogue: a wait placed after the epilogue
    // would also wait for its stores (vmcnt retires in order), which a finishing workgroup never has to do.
    primed = DIRECT && tile + stride < last;
    if (primed) advance();

    const size_t ob = (size_t)ct.bz * ep.out_batch_stride;
    const int m_lim = ep.m_valid > 0 ? ep.m_valid : g.M;
    if constexpr (DIRECT) {
      // FULL: no row of the tile is padding -> straight-line code without per-row predicates (the predicated form makes
      // hipcc wait vmcnt(0) in front of every chunk: it cannot count stores across the skipped branches)
      if (ct.m0 + BM <= m_lim) epilogue_direct<KIND, MI, true, F16>(ep, acc, noff, bias, cs, lane, ct.m0 + wm * WM, m_lim, ob);
      else epilogue_direct<KIND, MI, false, F16>(ep, acc, noff, bias, cs, lane, ct.m0 + wm * WM, m_lim, ob);
    } else {
      // split-K partial sums: accumulators -> this wave's private 64x64 f32 LDS tile -> one atomic per element, 64
      // consecutive floats per instruction; one pass per 64 rows of the wave tile.  The ring is empty here
      // (the stream stops at tile ends for this kind).
      __syncthreads();  // all fragment reads done before the ring is reused as the staging area
      float* cw = (float*)smem + wave * 4096;
      float* out = (float*)ep.out + ob;
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
        const int mw = ct.m0 + wm * WM + pass * 64;
        for (int row = 0; row < 64; ++row) {
          const float v = cw[row * 64 + lane];
#if defined(DEVIT_GEMM_NOATOMIC)  // ablation build: the split-K epilogue without its atomics
          if (mw + row < m_lim && v == 1.2345e30f) out[(size_t)(mw + row) * ep.ldc + nw + lane] = v;
#else
          if (mw + row < m_lim) unsafeAtomicAdd(out + (size_t)(mw + row) * ep.ldc + nw + lane, v);
#endif
        }
      };
      do_pass(std::integral_constant<int, 0>());
      if constexpr (MI > 4) do_pass(std::integral_constant<int, 1>());
      if (rowsum_on) {
        // lane l holds the partial sum of row (l & 15) over k = 8 (l >> 4) + 0..7 of every K-step: fold the four k groups
        float* rs_out = (float*)ep.aux;
#pragma unroll
        for (int i = 0; i < MI; ++i) {
          float v = rsum[i];
          v += __shfl_xor(v, 16, 64);
          v += __shfl_xor(v, 32, 64);
          const int row = ct.m0 + wm * WM + i * 16 + lane;
          if (lane < 16 && row < m_lim) unsafeAtomicAdd(rs_out + row, v);
        }
      }
      if (pb.tile < last) {          // restart the stream on the next tile
        __syncthreads();             // every wave's staging reads done before the DMA overwrites them
        pb.ref = decode_tile<BM, BN, A_KM, B_KM>(g, pb.tile);
        pb.t = 0;
        pb.open = true;
        pa = pb;
        a_slot = ca_slot;
        b_slot = cb_slot;
        issue_a();
        produce();
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Four-wave 256x256x64 kernel (round 4): one wave per SIMD, each a 128 x 128 sub-tile with its 256 accumulator registers in
// a[0:255], the K loop one hand-scheduled inline-asm statement (gemm4_kloop.inc, generated by tools/gen_gemm4.py: register plan,
// ring protocol and operand list are documented there).  Row-major x row-major operands (the forward Linear layers), every
// DIRECT epilogue kind; same LDS images, same swizzles, same tile order and the same accumulation order per output element as
// the eight-wave ping-pong kernel above (bit-identical results).  Why: the eight-wave kernel's epilogue runs two waves per SIMD
// through one vector-issue port with the MFMA pipe idle (21-36 % of every tile), and its K-step is a serial chain of four barrier
// intervals; a lone wave with 512 registers keeps both k-halves' fragments in registers, needs one barrier per K-step and a third
// fewer LDS bytes per flop -- provided the loop is hand-placed (hipcc's schedule of it was issue-bound).
#include "gemm4_kloop.inc"

template <int KIND, bool F16>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
void gemm4_kernel(const GemmArgs g) {
  static_assert(KIND != DEVIT_EPI_ATOMIC_F32 && KIND != DEVIT_EPI_DGELU_BF16 && !F16, "forward layouts, DIRECT epilogues, bf16");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int BM = 256, BN = 256, NWAVES = 4;
  constexpr bool PAIRED = KIND == DEVIT_EPI_STORE_BF16 || KIND == DEVIT_EPI_GELU_BF16;
  constexpr int A_TILE_BYTES = BM * BK * 2, B_TILE_BYTES = BN * BK * 2, B_RING = 3 * A_TILE_BYTES;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;

  const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3, stride = gridDim.x >> 3;
  int first, last;
  {
    const int q = g.total_tiles >> 3, r = g.total_tiles & 7;
    const int start = xcd * q + min(xcd, r);
    first = start + idx;
    last = start + q + (xcd < r ? 1 : 0);
  }
  if (first >= last) return;
#ifdef DEVIT_GEMM4_STAMP
  const unsigned long long t_entry = __builtin_amdgcn_s_memtime(), rt_entry = __builtin_amdgcn_s_memrealtime();
  unsigned long long st_sum[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // tiles, K loop, epilogue, d0 (entry reads), d1 (phase 1), d2 (middle), d3 (phase 2), first tile's loop
#endif

  const unsigned lda64 = (unsigned)g.lda * 64u, ldb64 = (unsigned)g.ldb * 64u;
  const unsigned wave_lds = (unsigned)(size_t)LDS_PTR(smem) + (unsigned)wave * 8192u;

  // prologue: stages 0 and 1 of the first tile, in the order the K loop keeps (A(t), B(t), A(t + 1), B(t + 1))
  TileRef ct = decode_tile<BM, BN, false, false>(g, first);
#pragma unroll
  for (int st = 0; st < 2; ++st) {
    stage_tile<false, BM, NWAVES>(ct.a, g.lda, (ct.kt0 + st) * BK, 0, 0, smem + st * A_TILE_BYTES, wave, lane);
    stage_tile<false, BN, NWAVES>(ct.b, g.ldb, (ct.kt0 + st) * BK, 0, 0, smem + B_RING + st * B_TILE_BYTES, wave, lane);
  }
  wait_vmcnt<0>();
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  unsigned g3 = 0, g2 = B_RING;     // LDS byte offsets of the slots that hold stage 0 of the tile about to start
  const unsigned wv = (unsigned)wave;

  for (int tile = first; tile < last; tile += stride) {
    const bool has_next = tile + stride < last;
    const TileRef nt = has_next ? decode_tile<BM, BN, false, false>(g, tile + stride) : ct;
    const __bf16* a_ptr = ct.a + (size_t)ct.kt0 * BK;
    const __bf16* b_ptr = ct.b + (size_t)ct.kt0 * BK;
    const __bf16* a_next = nt.a + (size_t)nt.kt0 * BK;
    const __bf16* b_next = nt.b + (size_t)nt.kt0 * BK;
    const unsigned nk = (unsigned)ct.nk;
    const devit_epilogue& ep = g.ep;
    // per-lane constants of the K loop (byte offsets inside an LDS slot / from a tile's operand pointer).  Tile-invariant, but
    // recomputed per tile from an opaque copy of the lane index (~60 VALU instructions): kept alive across the epilogue they
    // were the values hipcc chose to spill to scratch, and their reloads are vector-memory operations in front of the loop.
    //   fragment reads: tile index x (m-tile of A, n-tile of B), k-half kk -> VAR[kk][x & 1] + 4096 (x >> 1), see read_frag()
    //   LDS-DMA source: slab i of this wave (8 rows of 128 bytes) -> dma[i & 3] (+ 32 rows for i >= 4), see lane_offset()
    unsigned dsA[4], dsB[4], dmaA[4], dmaB[4];
    {
      int lane_k;    // = lane, from nothing (v_mbcnt): even `lane` itself, kept alive across the loop, was spilled
      asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane_k));
      const int c = lane_k & 15, gq = lane_k >> 4;
  #pragma unroll
      for (int kk = 0; kk < 2; ++kk)
  #pragma unroll
        for (int par = 0; par < 2; ++par) {
          const int rowA = wm * 128 + 16 * par + c;
          dsA[kk * 2 + par] = (unsigned)(rowA * 128 + (((kk * 4 + gq) ^ swz_row(rowA)) * 16));
          const int rowB = wn * 128 + tile_row<PAIRED>(par, c);
          dsB[kk * 2 + par] = (unsigned)(rowB * 128 + (((kk * 4 + gq) ^ swz_row(rowB)) * 16));
        }
  #pragma unroll
      for (int i = 0; i < 4; ++i) {
        dmaA[i] = lane_offset<false, BM, NWAVES>(g.lda, wave, lane_k, i, BM);
        dmaB[i] = lane_offset<false, BN, NWAVES>(g.ldb, wave, lane_k, i, BN);
      }
    }
    unsigned t0, t1, t2, t3;
#ifdef DEVIT_GEMM4_STAMP   // diagnostic build (tools/gemm4_stamps.py): cycles per tile in the loop's segments, summed per wave
    unsigned d0, d1, d2, d3;
    const unsigned long long ts0 = __builtin_amdgcn_s_memtime();
    asm volatile(DEVIT_GEMM4_KLOOP_STAMPED_ASM
                 : [g3] "+s"(g3), [g2] "+s"(g2), [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2), [t3] "=&v"(t3),
                   [d0] "=&s"(d0), [d1] "=&s"(d1), [d2] "=&s"(d2), [d3] "=&s"(d3)
                 : [aptr] "s"(a_ptr), [bptr] "s"(b_ptr), [anext] "s"(a_next), [bnext] "s"(b_next), [nk] "s"(nk),
                   [lda64] "s"(lda64), [ldb64] "s"(ldb64), [wlds] "s"(wave_lds), [wv] "s"(wv),
                   [dsa0] "v"(dsA[0]), [dsa1] "v"(dsA[1]), [dsa2] "v"(dsA[2]), [dsa3] "v"(dsA[3]),
                   [dsb0] "v"(dsB[0]), [dsb1] "v"(dsB[1]), [dsb2] "v"(dsB[2]), [dsb3] "v"(dsB[3]),
                   [dmaa0] "v"(dmaA[0]), [dmaa1] "v"(dmaA[1]), [dmaa2] "v"(dmaA[2]), [dmaa3] "v"(dmaA[3]),
                   [dmab0] "v"(dmaB[0]), [dmab1] "v"(dmaB[1]), [dmab2] "v"(dmaB[2]), [dmab3] "v"(dmaB[3])
                 : DEVIT_GEMM4_KLOOP_STAMPED_CLOBBERS);
    const unsigned long long ts1 = __builtin_amdgcn_s_memtime();
    st_sum[0] += 1; st_sum[1] += ts1 - ts0; st_sum[3] += d0; st_sum[4] += d1; st_sum[5] += d2; st_sum[6] += d3;
    if (tile == first) st_sum[7] = ts1 - ts0;
#else
    asm volatile(DEVIT_GEMM4_KLOOP_ASM
                 : [g3] "+s"(g3), [g2] "+s"(g2), [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2), [t3] "=&v"(t3)
                 : [aptr] "s"(a_ptr), [bptr] "s"(b_ptr), [anext] "s"(a_next), [bnext] "s"(b_next), [nk] "s"(nk),
                   [lda64] "s"(lda64), [ldb64] "s"(ldb64), [wlds] "s"(wave_lds), [wv] "s"(wv),
                   [dsa0] "v"(dsA[0]), [dsa1] "v"(dsA[1]), [dsa2] "v"(dsA[2]), [dsa3] "v"(dsA[3]),
                   [dsb0] "v"(dsB[0]), [dsb1] "v"(dsB[1]), [dsb2] "v"(dsB[2]), [dsb3] "v"(dsB[3]),
                   [dmaa0] "v"(dmaA[0]), [dmaa1] "v"(dmaA[1]), [dmaa2] "v"(dmaA[2]), [dmaa3] "v"(dmaA[3]),
                   [dmab0] "v"(dmaB[0]), [dmab1] "v"(dmaB[1]), [dmab2] "v"(dmaB[2]), [dmab3] "v"(dmaB[3])
                 : DEVIT_GEMM4_KLOOP_CLOBBERS);
#endif
    // v1: the epilogue's column data (bias, column scale) is fetched after the loop (hipcc waits vmcnt(0) for it: the loop's last
    // requests drain with it).  Not before the loop: a counted load consumed after the K loop gets the same vmcnt(0) (hipcc does
    // not see the loop's LDS-DMA), a wait placed in front of the loop would also wait for the previous tile's epilogue stores, and
    // 64 more registers live across the loop (which owns v128-v255) spilled.
    // The epilogue's per-lane values are derived from an opaque copy of the lane index made HERE: derived from `lane` itself, hipcc
    // hoists them out of the tile loop and keeps them alive across the K loop (dozens of registers: scratch spills, whose reloads
    // are vector-memory operations in front of the loop's counted waits).
    int lane_e;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane_e));
    int noff[2][4];
    f32x4 bias[2][4], cs[2][4];
#pragma unroll
    for (int h = 0; h < 2; ++h) load_cols<KIND>(ep, lane_e, ct.n0 + wn * 128 + h * 64, noff[h], bias[h], cs[h]);
    const size_t ob = (size_t)ct.bz * ep.out_batch_stride;
    const int m_lim = ep.m_valid > 0 ? ep.m_valid : g.M;
    const bool full = ct.m0 + BM <= m_lim;
    auto chunk = [&](auto hc, auto ic) {
      constexpr int H = decltype(hc)::value, I0 = decltype(ic)::value;
      f32x4 acc[2][4];
      gemm4_read_acc<H, I0>(acc);
      const int mw = ct.m0 + wm * 128 + I0 * 16;
      if (full) epilogue_direct<KIND, 2, true, F16>(ep, acc, noff[H], bias[H], cs[H], lane_e, mw, m_lim, ob);
      else epilogue_direct<KIND, 2, false, F16>(ep, acc, noff[H], bias[H], cs[H], lane_e, mw, m_lim, ob);
    };
    auto half = [&](auto hc) {
      chunk(hc, std::integral_constant<int, 0>());
      chunk(hc, std::integral_constant<int, 2>());
      chunk(hc, std::integral_constant<int, 4>());
      chunk(hc, std::integral_constant<int, 6>());
    };
    half(std::integral_constant<int, 0>());
    half(std::integral_constant<int, 1>());
#ifdef DEVIT_GEMM4_STAMP
    st_sum[2] += __builtin_amdgcn_s_memtime() - ts1;
#endif
    ct = nt;
  }
  wait_vmcnt<0>();   // the last tile requested two stages nobody reads: they must have landed before the workgroup's LDS is released
#ifdef DEVIT_GEMM4_STAMP
  if (g.ep.pos && KIND != DEVIT_EPI_PATCH_F32 && lane == 0) {
    unsigned long long* dbg = (unsigned long long*)g.ep.pos + ((size_t)blockIdx.x * NWAVES + wave) * 16;
#pragma unroll
    for (int q = 0; q < 8; ++q) dbg[q] = st_sum[q];
    dbg[8] = t_entry; dbg[9] = __builtin_amdgcn_s_memtime(); dbg[10] = rt_entry; dbg[11] = __builtin_amdgcn_s_memrealtime();
  }
#endif
}

// ---------------------------------------------------------------------------------------------------------------------
// Full-row 256x384x64 kernel (round 5): the student's N = 384 launches -- activation operand row-major, weight operand K-MAJOR (the
// dgrads of qkv / proj / fc1 as they stand; proj / fc2 forward through a k-major copy of their weights), bf16 store or fp32 residual
// epilogue.  Four waves, one per SIMD, each a 128 x 192 sub-tile = 96 accumulator tiles: 64 in a[0:255], 32 in v[128:255] (pinned asm
// outputs); ONE fragment buffer; two A slots (32 KB) + two B slots (48 KB), the B stages shifted by half a stage so that a slot is
// released -- and 8 or 12 requests per wave leave -- in EVERY phase; the K loop is a generated inline-asm statement
// (gemmfr_kloop.inc, tools/gen_gemmfr.py: register plan, ring protocol, operand list).  Same LDS images / swizzles / MFMA operand roles /
// accumulation order per output element as the 128x128 kernels these launches ran on: bit-identical results.  Why: two 128x128
// workgroups per CU ask the CU's fill path for 64 B per cycle of matrix pipe and get ~24; this tile needs
// 26.7 and reads the activation panel once instead of three times.
#include "gemmfr_kloop.inc"

typedef float f32x32 __attribute__((ext_vector_type(32)));

// per-lane source byte offset of slab i (1 KiB, of this wave's twelve = 16 k rows) of a 384-wide k-major B stage, relative to the wave's
// first k row.  Image [64 k][384 cols]: a k-row is 48 chunks of 16 bytes, slabs cross k-rows (lane_offset<true> wants 64 % (W / 8) == 0).
__device__ __forceinline__ unsigned fr_dma_off_b(int ld, int wave, int lane, int i) {
  const int piece = (wave * 12 + i) * 64 + lane, krow = piece / 48, c = piece % 48;
  return (unsigned)((krow - 16 * wave) * ld + ((c ^ swz_krow(krow)) * 8)) * 2u;
}

// first k row of wave `wave`'s share (16 k rows) of B stage u: the stages are shifted by half a stage and cyclic in K
__device__ __forceinline__ int fr_b_row(int u, int wave, int K) {
  const int r = 64 * u - 32 + 16 * wave;
  return r < 0 ? r + K : (r >= K ? r - K : r);
}

template <int KIND>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
void gemmfr_kernel(const GemmArgs g) {
  static_assert(KIND == DEVIT_EPI_RESIDUAL_F32 || KIND == DEVIT_EPI_STORE_BF16, "the student's N = 384 launches");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int BM = 256, BN = 384, NWAVES = 4;
  constexpr bool PAIRED = KIND == DEVIT_EPI_STORE_BF16;   // column order of the n-tiles, tile_row<PAIRED>()
  constexpr int A_SLOT = BM * BK * 2, B_SLOT = BN * BK * 2, B_RING = 2 * A_SLOT;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;

  const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3, stride = gridDim.x >> 3;
  int first, last;
  {
    const int q = g.total_tiles >> 3, r = g.total_tiles & 7;
    const int start = xcd * q + min(xcd, r);
    first = start + idx;
    last = start + q + (xcd < r ? 1 : 0);
  }
  if (first >= last) return;
#ifdef DEVIT_GEMMFR_STAMP
  const unsigned long long t_entry = __builtin_amdgcn_s_memtime();
  unsigned long long st_sum[6] = {0, 0, 0, 0, 0, 0};   // tiles, K loop, epilogue, d1 (barrier to barrier), d2 (barrier waits), prologue
#endif

  const unsigned lda64 = (unsigned)g.lda * 64u, ldbs = (unsigned)g.ldb * 128u, kb = (unsigned)g.K * (unsigned)g.ldb * 2u;
  const unsigned lds_base = (unsigned)(size_t)LDS_PTR(smem);
  const unsigned wldsa = lds_base + (unsigned)wave * 8192u, wldsb = lds_base + (unsigned)wave * 12288u;   // (+ the slot's offset)

  // prologue: A stages 0, 1 of the first tile; B stages 0, 1 of the cyclic stream (one n-tile: every tile multiplies by the same B)
  TileRef ct = decode_tile<BM, BN, false, true>(g, first);
#pragma unroll
  for (int st = 0; st < 2; ++st) {
    stage_tile<false, BM, NWAVES, true>(ct.a, g.lda, (ct.kt0 + st) * BK, 0, 0, smem + st * A_SLOT, wave, lane);
    const char* ub = (const char*)(ct.b + (size_t)fr_b_row(st, wave, g.K) * g.ldb);
    const unsigned lds0 = lds_base + (unsigned)(B_RING + st * B_SLOT) + (unsigned)wave * 12288u;
#pragma unroll
    for (int i = 0; i < 12; i += 2)
      dma2_uniform<false>(ub, fr_dma_off_b(g.ldb, wave, lane, i), fr_dma_off_b(g.ldb, wave, lane, i + 1), lds0 + i * 1024u);
  }
  wait_vmcnt<0>();
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  unsigned acur = 0;                // LDS byte offset of the A slot that holds stage 0 of the tile about to start
  const unsigned wv = (unsigned)wave;
  const unsigned bv2 = (unsigned)fr_b_row(2, wave, g.K) * (unsigned)g.ldb * 2u;
  const unsigned bplo = (unsigned)(uintptr_t)ct.b, bphi = (unsigned)((uintptr_t)ct.b >> 32);
#ifdef DEVIT_GEMMFR_STAMP
  st_sum[5] = __builtin_amdgcn_s_memtime() - t_entry;
#endif

  for (int tile = first; tile < last; tile += stride) {
    const bool has_next = tile + stride < last;
    const TileRef nt = has_next ? decode_tile<BM, BN, false, true>(g, tile + stride) : ct;
    const __bf16* a_ptr = ct.a + (size_t)ct.kt0 * BK;
    const __bf16* a_next = nt.a + (size_t)nt.kt0 * BK;
    const unsigned nk = (unsigned)ct.nk, hasnext = (unsigned)__builtin_amdgcn_readfirstlane(has_next ? 1 : 0);
    const devit_epilogue& ep = g.ep;
    // per-lane constants of the K loop, recomputed per tile from an opaque copy of the lane index (see gemm4_kernel)
    unsigned dsA[4], dsB[8], dmaA[4], dmaB[12];
    {
      int lane_k;
      asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane_k));
      const int c = lane_k & 15, gq = lane_k >> 4;
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int par = 0; par < 2; ++par) {
          const int rowA = wm * 128 + 16 * par + c;
          dsA[kk * 2 + par] = (unsigned)(rowA * 128 + (((kk * 4 + gq) ^ swz_row(rowA)) * 16));
        }
      // read_frag<true, 384, PAIRED>: see b_reads() in tools/gen_gemmfr.py
      const int q4 = (lane_k >> 2) & 3, p = lane_k & 3;
#pragma unroll
      for (int x3 = 0; x3 < 4; ++x3) {
        const unsigned row = (unsigned)((gq * 8 + q4) * (BN * 2) + 64 * (x3 ^ q4));
        if constexpr (PAIRED) {
          dsB[x3] = row + (unsigned)(16 * (p ^ ((gq & 1) << 1)));
          dsB[4 + x3] = 0;
        } else {
#pragma unroll
          for (int jp = 0; jp < 2; ++jp) dsB[2 * x3 + jp] = row + (unsigned)(32 * (jp ^ (gq & 1)) + 16 * (p >> 1) + 8 * (p & 1));
        }
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) dsA[i] += lds_base;      // (the asm adds only the slot offsets: the ring need not start at LDS address 0)
#pragma unroll
      for (int i = 0; i < 8; ++i) dsB[i] += lds_base;
#pragma unroll
      for (int i = 0; i < 4; ++i) dmaA[i] = lane_offset<false, BM, NWAVES>(g.lda, wave, lane_k, i, BM);
#pragma unroll
      for (int i = 0; i < 12; ++i) dmaB[i] = fr_dma_off_b(g.ldb, wave, lane_k, i);
    }
    unsigned t0, t1, t2, t3, t4, t5, t6, t7, t8, t9;
    f32x32 c0, c1, c2, c3;
#ifdef DEVIT_GEMMFR_STAMP
    unsigned d1, d2;
    const unsigned long long ts0 = __builtin_amdgcn_s_memtime();
#define DEVIT_FR_STAMP_OUT , [d1] "=&s"(d1), [d2] "=&s"(d2)
#define DEVIT_FR_ASM(O) DEVIT_GEMMFR_KLOOP_##O##_STAMPED_ASM
#define DEVIT_FR_CLOB(O) DEVIT_GEMMFR_KLOOP_##O##_STAMPED_CLOBBERS
#else
#define DEVIT_FR_STAMP_OUT
#define DEVIT_FR_ASM(O) DEVIT_GEMMFR_KLOOP_##O##_ASM
#define DEVIT_FR_CLOB(O) DEVIT_GEMMFR_KLOOP_##O##_CLOBBERS
#endif
#define DEVIT_FR_STATEMENT(O)                                                                                                   \
    asm volatile(DEVIT_FR_ASM(O)                                                                                                \
                 : [c0] "=&{v[128:159]}"(c0), [c1] "=&{v[160:191]}"(c1), [c2] "=&{v[192:223]}"(c2), [c3] "=&{v[224:255]}"(c3),  \
                   [acur] "+s"(acur), [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2), [t3] "=&v"(t3), [t4] "=&v"(t4),           \
                   [t5] "=&v"(t5), [t6] "=&v"(t6), [t7] "=&v"(t7), [t8] "=&v"(t8), [t9] "=&v"(t9) DEVIT_FR_STAMP_OUT            \
                 : [aptr] "s"(a_ptr), [anext] "s"(a_next), [bplo] "s"(bplo), [bphi] "s"(bphi), [bv2] "s"(bv2), [kb] "s"(kb),    \
                   [nk] "s"(nk), [hasnext] "s"(hasnext), [lda64] "s"(lda64), [ldbs] "s"(ldbs), [wldsa] "s"(wldsa),              \
                   [wldsb] "s"(wldsb), [wv] "s"(wv),                                                                            \
                   [dsa0] "v"(dsA[0]), [dsa1] "v"(dsA[1]), [dsa2] "v"(dsA[2]), [dsa3] "v"(dsA[3]),                              \
                   [dsb0] "v"(dsB[0]), [dsb1] "v"(dsB[1]), [dsb2] "v"(dsB[2]), [dsb3] "v"(dsB[3]),                              \
                   [dsb4] "v"(dsB[4]), [dsb5] "v"(dsB[5]), [dsb6] "v"(dsB[6]), [dsb7] "v"(dsB[7]),                              \
                   [dmaa0] "v"(dmaA[0]), [dmaa1] "v"(dmaA[1]), [dmaa2] "v"(dmaA[2]), [dmaa3] "v"(dmaA[3]),                      \
                   [dmab0] "v"(dmaB[0]), [dmab1] "v"(dmaB[1]), [dmab2] "v"(dmaB[2]), [dmab3] "v"(dmaB[3]),                      \
                   [dmab4] "v"(dmaB[4]), [dmab5] "v"(dmaB[5]), [dmab6] "v"(dmaB[6]), [dmab7] "v"(dmaB[7]),                      \
                   [dmab8] "v"(dmaB[8]), [dmab9] "v"(dmaB[9]), [dmab10] "v"(dmaB[10]), [dmab11] "v"(dmaB[11])                   \
                 : DEVIT_FR_CLOB(O))
    if constexpr (PAIRED) DEVIT_FR_STATEMENT(PAIRED);
    else DEVIT_FR_STATEMENT(NATURAL);
#undef DEVIT_FR_STATEMENT
#undef DEVIT_FR_STAMP_OUT
#undef DEVIT_FR_ASM
#undef DEVIT_FR_CLOB
#ifdef DEVIT_GEMMFR_STAMP
    const unsigned long long ts1 = __builtin_amdgcn_s_memtime();
    st_sum[0] += 1; st_sum[1] += ts1 - ts0; st_sum[3] += d1; st_sum[4] += d2;
#endif
    // epilogue: the eight-wave kernels' register epilogue on chunks of two m-tiles x four n-tiles; the column group that lives in
    // VGPRs (n-tiles 8..11) first -- it frees the registers the other chunks' values are read out into
    int lane_e;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane_e));
    const size_t ob = (size_t)ct.bz * ep.out_batch_stride;
    const int m_lim = ep.m_valid > 0 ? ep.m_valid : g.M;
    const bool full = ct.m0 + BM <= m_lim;
    f32x4 cs[4];   // (no column scale in these kinds)
    auto run = [&](f32x4 (&acc)[2][4], const int (&noff)[4], const f32x4 (&bias)[4], int i0) {
      const int mw = ct.m0 + wm * 128 + i0 * 16;
      if (full) epilogue_direct<KIND, 2, true, false>(ep, acc, noff, bias, cs, lane_e, mw, m_lim, ob);
      else epilogue_direct<KIND, 2, false, false>(ep, acc, noff, bias, cs, lane_e, mw, m_lim, ob);
    };
    {
      int noff[4];
      f32x4 bias[4];
      load_cols<KIND>(ep, lane_e, ct.n0 + wn * 192 + 128, noff, bias, cs);
      auto from_v = [&](const f32x32& c, int i0) {
        f32x4 acc[2][4];
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            acc[u][j] = (f32x4){c[16 * u + 4 * j], c[16 * u + 4 * j + 1], c[16 * u + 4 * j + 2], c[16 * u + 4 * j + 3]};
        run(acc, noff, bias, i0);
      };
      from_v(c0, 0);
      from_v(c1, 2);
      from_v(c2, 4);
      from_v(c3, 6);
    }
    auto group = [&](auto hc) {
      constexpr int H = decltype(hc)::value;
      int noff[4];
      f32x4 bias[4];
      load_cols<KIND>(ep, lane_e, ct.n0 + wn * 192 + H * 64, noff, bias, cs);
      auto chunk = [&](auto ic) {
        constexpr int I0 = decltype(ic)::value;
        f32x4 acc[2][4];
        gemmfr_read_acc<H, I0>(acc);
        run(acc, noff, bias, I0);
      };
      chunk(std::integral_constant<int, 0>());
      chunk(std::integral_constant<int, 2>());
      chunk(std::integral_constant<int, 4>());
      chunk(std::integral_constant<int, 6>());
    };
    group(std::integral_constant<int, 0>());
    group(std::integral_constant<int, 1>());
#ifdef DEVIT_GEMMFR_STAMP
    st_sum[2] += __builtin_amdgcn_s_memtime() - ts1;
#endif
    ct = nt;
  }
  wait_vmcnt<0>();   // (requests of a next tile that does not exist are never made; this only drains the epilogue's stores)
#ifdef DEVIT_GEMMFR_STAMP
  if (g.ep.pos && lane == 0) {
    unsigned long long* dbg = (unsigned long long*)g.ep.pos + ((size_t)blockIdx.x * NWAVES + wave) * 8;
#pragma unroll
    for (int q = 0; q < 6; ++q) dbg[q] = st_sum[q];
    dbg[6] = t_entry; dbg[7] = __builtin_amdgcn_s_memtime();
  }
#endif
}

// ---------------------------------------------------------------------------------------------------------------------
// Weight gradients on the full-row tile (round 6): out (+)= A^T B over the token rows, BOTH operands k-major ([K][features], as the step holds
// dY and X), a table of up to eight jobs = the Linear layers of one block (or of several) in ONE launch, one 256 x 384 tile and one K slice per
// workgroup, all of them resident at once.  Why: the split-K 128x128 launches this replaces (four per block) asked the CU's fill path for
// 64 B per cycle of matrix pipe where it delivers ~24 and ran 14-56-step K loops in front of 64 KB of atomics each;
// this tile needs 26.7 B per cycle, reads the dY panel once, and a block's four products are 19 tiles x 13 slices = 247 workgroups with
// ~61-step K loops.  K loop: the generated asm statement DEVIT_WGRADFR_KLOOP (tools/gen_gemmfr.py, KMA variant: the ring protocol, phases and
// waits of gemmfr_kernel; A image [64 k][256], fragments by ds_read_b64_tr_b16 on both sides, PAIRED tile-row order on both sides, column
// sums of A by v_dot2c against packed ones).  Epilogue: the accumulators go through LDS 32 rows at a time and leave as fp32 atomics on whole
// 256-byte rows (128-byte columns for a transposed job).
constexpr int WGRAD_MAX_JOBS = 48;
struct WgJob {
  const __bf16* a;
  const __bf16* b;
  float* out;
  float* colsum;
  int lda, ldb, ldc, a_cols, transposed, tile0;    // tile0: index of the job's first tile in the table's tile list
};
struct WgArgs {
  WgJob job[WGRAD_MAX_JOBS];
  int njobs, tiles, split, nk_total, total;
  unsigned long long* dbg;       // stamped diagnostic build only (tools/wgradfr_stamps.py)
};

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
void wgradfr_kernel(const WgArgs g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int BM = 256, BN = 384, NWAVES = 4;
  constexpr int A_SLOT = BM * BK * 2, B_SLOT = BN * BK * 2, B_RING = 2 * A_SLOT;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;

  // workgroups b, b + 8, ... share an XCD: each XCD takes a contiguous run of (slice, tile) pairs, tile fastest -- the tiles of one job and
  // slice (2-6 of them) read the same B rows at the same time through that XCD's L2
  int L;
  {
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    const int q = g.total >> 3, r = g.total & 7;
    if (idx >= q + (xcd < r ? 1 : 0)) return;
    L = xcd * q + min(xcd, r) + idx;
  }
#ifdef DEVIT_GEMMFR_STAMP
  const unsigned long long t_entry = __builtin_amdgcn_s_memtime();
#endif
  const int z = L / g.tiles, t = L - z * g.tiles;
  int ji = 0;
  for (int j = 1; j < g.njobs; ++j)
    if (t >= g.job[j].tile0) ji = j;
  const WgJob& jb = g.job[ji];
  const int m0 = (t - jb.tile0) * BM;
  const int valid = min(BM, jb.a_cols - m0);                    // 256 or 128 columns of A exist in this tile
  const int kt0 = (int)((long long)z * g.nk_total / g.split);
  const int nk_i = (int)((long long)(z + 1) * g.nk_total / g.split) - kt0;
  const int lda = jb.lda, ldb = jb.ldb;
  const __bf16* a_tile = jb.a + (size_t)kt0 * BK * lda + m0;    // &A[k0][m0]
  const __bf16* b_sl = jb.b + (size_t)kt0 * BK * ldb;           // &B[k0][0]
  const int Ks = nk_i * BK;                                     // the slice: B's half-stage-shifted stream is cyclic in it

  const unsigned ldas = (unsigned)lda * 128u, ldbs = (unsigned)ldb * 128u, kb = (unsigned)Ks * (unsigned)ldb * 2u;
  const unsigned lds_base = (unsigned)(size_t)LDS_PTR(smem);
  const unsigned wldsa = lds_base + (unsigned)wave * 8192u, wldsb = lds_base + (unsigned)wave * 12288u;

  // prologue: stages 0, 1 of both operands
#pragma unroll
  for (int st = 0; st < 2; ++st) {
    stage_tile<true, BM, NWAVES, true>(a_tile, lda, st * BK, 0, 0, smem + st * A_SLOT, wave, lane, valid);
    const char* ub = (const char*)(b_sl + (size_t)fr_b_row(st, wave, Ks) * ldb);
    const unsigned lds0 = lds_base + (unsigned)(B_RING + st * B_SLOT) + (unsigned)wave * 12288u;
#pragma unroll
    for (int i = 0; i < 12; i += 2)
      dma2_uniform<false>(ub, fr_dma_off_b(ldb, wave, lane, i), fr_dma_off_b(ldb, wave, lane, i + 1), lds0 + i * 1024u);
  }
  wait_vmcnt<0>();
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
#ifdef DEVIT_GEMMFR_STAMP
  const unsigned long long t_primed = __builtin_amdgcn_s_memtime();
#endif
  unsigned acur = 0;
  const unsigned wv = (unsigned)wave, nk = (unsigned)nk_i;
  const unsigned bv2 = (unsigned)fr_b_row(2, wave, Ks) * (unsigned)ldb * 2u;
  const unsigned bplo = (unsigned)(uintptr_t)b_sl, bphi = (unsigned)((uintptr_t)b_sl >> 32);
  const unsigned ones = jb.colsum ? 0x3f803f80u : 0u;

  // per-lane constants of the K loop (gen_gemmfr.py, KMA variant)
  unsigned dsA[4], dsB[4], dmaA[8], dmaB[12];
  {
    int lane_k;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane_k));
    const int gq = lane_k >> 4, q4 = (lane_k >> 2) & 3, p = lane_k & 3;
#pragma unroll
    for (int x = 0; x < 4; ++x) {
      dsA[x] = lds_base + (unsigned)((gq * 8 + q4) * (BM * 2) + 256 * wm + 64 * (x ^ q4) + 16 * (p ^ ((gq & 1) << 1)));
      dsB[x] = lds_base + (unsigned)((gq * 8 + q4) * (BN * 2) + 64 * (x ^ q4) + 16 * (p ^ ((gq & 1) << 1)));
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) dmaA[i] = lane_offset<true, BM, NWAVES>(lda, wave, lane_k, i, valid);
#pragma unroll
    for (int i = 0; i < 12; ++i) dmaB[i] = fr_dma_off_b(ldb, wave, lane_k, i);
  }
  unsigned t0, t1, t2, t3, t4, t5, t6, t7;
  float rs0 = 0.f, rs1 = 0.f, rs2 = 0.f, rs3 = 0.f;
  f32x32 c0, c1, c2, c3;
#ifdef DEVIT_GEMMFR_STAMP
  unsigned d1, d2;
  const unsigned long long ts0 = __builtin_amdgcn_s_memtime();
#define DEVIT_WG_STAMP_OUT , [d1] "=&s"(d1), [d2] "=&s"(d2)
#define DEVIT_WG_ASM DEVIT_WGRADFR_KLOOP_STAMPED_ASM
#define DEVIT_WG_CLOB DEVIT_WGRADFR_KLOOP_STAMPED_CLOBBERS
#else
#define DEVIT_WG_STAMP_OUT
#define DEVIT_WG_ASM DEVIT_WGRADFR_KLOOP_ASM
#define DEVIT_WG_CLOB DEVIT_WGRADFR_KLOOP_CLOBBERS
#endif
  asm volatile(DEVIT_WG_ASM
               : [c0] "=&{v[128:159]}"(c0), [c1] "=&{v[160:191]}"(c1), [c2] "=&{v[192:223]}"(c2), [c3] "=&{v[224:255]}"(c3),
                 [acur] "+s"(acur), [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2), [t3] "=&v"(t3), [t4] "=&v"(t4),
                 [t5] "=&v"(t5), [t6] "=&v"(t6), [t7] "=&v"(t7), [rs0] "+v"(rs0), [rs1] "+v"(rs1), [rs2] "+v"(rs2), [rs3] "+v"(rs3) DEVIT_WG_STAMP_OUT
               : [aptr] "s"(a_tile), [bplo] "s"(bplo), [bphi] "s"(bphi), [bv2] "s"(bv2), [kb] "s"(kb), [nk] "s"(nk), [ldas] "s"(ldas),
                 [ldbs] "s"(ldbs), [wldsa] "s"(wldsa), [wldsb] "s"(wldsb), [wv] "s"(wv), [ones] "s"(ones),
                 [dsa0] "v"(dsA[0]), [dsa1] "v"(dsA[1]), [dsa2] "v"(dsA[2]), [dsa3] "v"(dsA[3]),
                 [dsb0] "v"(dsB[0]), [dsb1] "v"(dsB[1]), [dsb2] "v"(dsB[2]), [dsb3] "v"(dsB[3]),
                 [dmaa0] "v"(dmaA[0]), [dmaa1] "v"(dmaA[1]), [dmaa2] "v"(dmaA[2]), [dmaa3] "v"(dmaA[3]),
                 [dmaa4] "v"(dmaA[4]), [dmaa5] "v"(dmaA[5]), [dmaa6] "v"(dmaA[6]), [dmaa7] "v"(dmaA[7]),
                 [dmab0] "v"(dmaB[0]), [dmab1] "v"(dmaB[1]), [dmab2] "v"(dmaB[2]), [dmab3] "v"(dmaB[3]),
                 [dmab4] "v"(dmaB[4]), [dmab5] "v"(dmaB[5]), [dmab6] "v"(dmaB[6]), [dmab7] "v"(dmaB[7]),
                 [dmab8] "v"(dmaB[8]), [dmab9] "v"(dmaB[9]), [dmab10] "v"(dmaB[10]), [dmab11] "v"(dmaB[11])
               : DEVIT_WG_CLOB);
#undef DEVIT_WG_STAMP_OUT
#undef DEVIT_WG_ASM
#undef DEVIT_WG_CLOB
#ifdef DEVIT_GEMMFR_STAMP
  const unsigned long long ts1 = __builtin_amdgcn_s_memtime();
#endif
  // ---- epilogue.  Lane (g, c) holds, for A tile i and B tile q, register r:  C[row(i, c)][col(q, 4 g + r)] with the PAIRED tile-row order on
  // both sides: row(i, c) = 32 (i >> 1) + 8 (c >> 2) + 4 (i & 1) + (c & 3) of the wave's 128, col(q, .) = 32 (q >> 1) + 8 g + 4 (q & 1) + r of
  // its 192.  Pass k stages the A tiles (2 k, 2 k + 1) = rows 32 k .. 32 k + 31 as [32][192 (+4)] floats in the wave's own LDS region
  // (the ring is free: the K loop's last requests were waited for in its last step) and adds them to `out` a whole row piece per instruction.
  __syncthreads();             // every wave's last fragment reads are done
  int lane_e;
  asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane_e));
  constexpr int PITCH = 196;
  float* cw = (float*)smem + wave * (32 * PITCH);
  const int gq = lane_e >> 4, c = lane_e & 15;
  const bool wave_valid = wm * 128 < valid;                  // (a half tile: the rows of the wm = 1 waves do not exist)
  float* wrow = cw + (8 * (c >> 2) + (c & 3)) * PITCH + 8 * gq;
  auto stage = [&](const f32x4 (&acc)[2][4], int H) {       // n-tiles 4 H .. 4 H + 3 of A tiles (2 k, 2 k + 1)
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int j = 0; j < 4; ++j) *(f32x4*)(wrow + 4 * u * PITCH + 32 * ((4 * H + j) >> 1) + 4 * (j & 1)) = acc[u][j];
  };
  auto flush = [&](int k) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const int mb = m0 + wm * 128 + 32 * k;                   // first output row (= A column) of the pass
    if (!jb.transposed) {
      float* o = jb.out + (size_t)mb * jb.ldc + wn * 192 + lane_e;
#pragma unroll 4
      for (int row = 0; row < 32; ++row) {
        const float* src = cw + row * PITCH + lane_e;
        const float v0 = src[0], v1 = src[64], v2 = src[128];
        float* d = o + (size_t)row * jb.ldc;
        unsafeAtomicAdd(d, v0);
        unsafeAtomicAdd(d + 64, v1);
        unsafeAtomicAdd(d + 128, v2);
      }
    } else {
      const int row = lane_e & 31, nsub = lane_e >> 5;
      float* o = jb.out + (size_t)(wn * 192 + nsub) * jb.ldc + mb + row;
      const float* src = cw + row * PITCH + nsub;
#pragma unroll 4
      for (int n2 = 0; n2 < 96; ++n2) unsafeAtomicAdd(o + (size_t)(2 * n2) * jb.ldc, src[2 * n2]);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // (the region is rewritten by the next pass)
  };
  auto from_v = [&](const f32x32& cv) {
    f32x4 acc[2][4];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        acc[u][j] = (f32x4){cv[16 * u + 4 * j], cv[16 * u + 4 * j + 1], cv[16 * u + 4 * j + 2], cv[16 * u + 4 * j + 3]};
    stage(acc, 2);
  };
  auto pass = [&](auto kc, const f32x32& cv) {
    constexpr int k = decltype(kc)::value;
    if (wave_valid) {
      from_v(cv);
      f32x4 acc[2][4];
      gemmfr_read_acc<0, 2 * k>(acc);
      stage(acc, 0);
      gemmfr_read_acc<1, 2 * k>(acc);
      stage(acc, 1);
      flush(k);
    }
  };
  pass(std::integral_constant<int, 0>(), c0);
  pass(std::integral_constant<int, 1>(), c1);
  pass(std::integral_constant<int, 2>(), c2);
  pass(std::integral_constant<int, 3>(), c3);
  if (jb.colsum && wave_valid) {
    // lane (G, c) holds the sum over ITS k (8 G .. 8 G + 7 of every 32) of tile row c of A tile 2 x + wn: fold the four G
    float rs[4] = {rs0, rs1, rs2, rs3};
#pragma unroll
    for (int x = 0; x < 4; ++x) {
      float v = rs[x];
      v += __shfl_xor(v, 16, 64);
      v += __shfl_xor(v, 32, 64);
      const int i = 2 * x + wn;
      const int row = wm * 128 + 32 * (i >> 1) + 8 * (c >> 2) + 4 * (i & 1) + (c & 3);
      if (lane_e < 16) unsafeAtomicAdd(jb.colsum + m0 + row, v);
    }
  }
#ifdef DEVIT_GEMMFR_STAMP
  if (g.dbg && lane == 0) {      // per wave: K-steps, prologue, K loop, phase sum, barrier-wait sum, epilogue, entry, exit
    unsigned long long* dbg = g.dbg + ((size_t)blockIdx.x * NWAVES + wave) * 8;
    const unsigned long long t_exit = __builtin_amdgcn_s_memtime();
    dbg[0] = nk; dbg[1] = t_primed - t_entry; dbg[2] = ts1 - ts0; dbg[3] = d1; dbg[4] = d2; dbg[5] = t_exit - ts1; dbg[6] = t_entry; dbg[7] = t_exit;
  }
#endif
}

bool aligned16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

// CUs the persistent grids leave free (devit_set_reserved_cus): -1 = not set yet, take DEVIT_RESERVE_CUS from the environment
int g_reserved_cus = -1;

int reserved_cus() {
  if (g_reserved_cus < 0) {
    const char* e = getenv("DEVIT_RESERVE_CUS");
    int n = e ? atoi(e) : 0;
    g_reserved_cus = (n >= 0 && n <= 128) ? n / 8 * 8 : 0;
  }
  return g_reserved_cus;
}

}  // namespace

// The full-row 256x384 kernel (gemmfr_kernel): N == 384 exactly (one n-tile: its B stream is cyclic over the tiles), whole 256-row tiles and
// enough of them to give most CUs one (the token-row GEMMs of the lean last block stay on 128x128 tiles), K >= 3 stages.  In-step times,
// profiles/r05_*: the dgrads of qkv / proj / fc1 and fc2's forward; NOT proj's forward (6 K-steps in front of a 57 k-cycle residual epilogue on
// 198 of 256 CUs: slower than two 128x128 workgroups per CU), which therefore never passes a k-major weight.  DEVIT_GEMMFR=0 / 1 forces it
// off / on wherever it is built (read per call: tests switch it).
extern "C" int devit_gemm_full_row_selected(int M, int N, int K, int kind) {
  if (!(M > 0 && M % 256 == 0 && N == 384 && K % BK == 0 && K / BK >= 3 && (kind == DEVIT_EPI_RESIDUAL_F32 || kind == DEVIT_EPI_STORE_BF16)))
    return 0;
  const char* e = getenv("DEVIT_GEMMFR");
  return (e && *e) ? atoi(e) != 0 : M / 256 >= 64;    // (a minimum K of 1024 instead of 192 measured the same on the compacted student and 0.5 % less on the dense one)
}

#ifdef DEVIT_GEMMFR_STAMP
static unsigned long long* g_wgrad_dbg = nullptr;
extern "C" DEVIT_API void devit_wgrad_debug_buffer(void* p) { g_wgrad_dbg = (unsigned long long*)p; }   // [grid x 4 waves x 8] u64, diagnostic build only
#endif

extern "C" int devit_wgrad_grouped(const devit_wgrad_job* jobs, int njobs, int K, int split_k, void* stream) {
  DEVIT_CHECK(jobs && njobs >= 1 && njobs <= WGRAD_MAX_JOBS, DEVIT_ERR_ARG, "devit_wgrad_grouped: 1..%d jobs (host array)", WGRAD_MAX_JOBS);
  DEVIT_CHECK(K > 0 && K % BK == 0, DEVIT_ERR_SHAPE, "devit_wgrad_grouped: K=%d must be a multiple of %d", K, BK);
  WgArgs g;
  int tiles = 0;
  for (int j = 0; j < njobs; ++j) {
    const devit_wgrad_job& q = jobs[j];
    DEVIT_CHECK(q.a && q.b && q.out, DEVIT_ERR_ARG, "devit_wgrad_grouped: job %d: null pointer", j);
    DEVIT_CHECK(q.a_cols > 0 && q.a_cols % 128 == 0 && q.lda >= q.a_cols && q.ldb >= 384 && q.lda % 8 == 0 && q.ldb % 8 == 0, DEVIT_ERR_SHAPE,
                "devit_wgrad_grouped: job %d: a_cols=%d (a multiple of 128) lda=%d ldb=%d (>= 384 columns are read)", j, q.a_cols, q.lda, q.ldb);
    DEVIT_CHECK(aligned16(q.a) && aligned16(q.b) && ((uintptr_t)q.out & 3) == 0 && q.ldc >= (q.transposed ? q.a_cols : 384), DEVIT_ERR_ARG,
                "devit_wgrad_grouped: job %d: operands must be 16-byte aligned, ldc=%d too small", j, q.ldc);
    g.job[j] = WgJob{(const __bf16*)q.a, (const __bf16*)q.b, q.out, q.a_colsum, q.lda, q.ldb, q.ldc, q.a_cols, q.transposed, tiles};
    tiles += (q.a_cols + 255) / 256;
  }
  static int cus = 0;
  if (cus == 0) {
    int dev = 0, n = 0;
    DEVIT_CHECK(hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n >= 8,
                DEVIT_ERR_DEVICE, "devit_wgrad_grouped: cannot query the CU count");
    cus = n;
  }
  const int nk_total = K / BK;
  if (split_k <= 0) {
    // K slices by a two-term cost model (microseconds), both terms measured (profiles/r06_a_wgradfr_*.txt, r06_H_*): a workgroup walks a K-step in
    // ~2.1 us (the launch is bound by the LDS-DMA stream out of HBM), workgroups run one per CU in rounds of `avail`; every (tile, slice) leaves through
    // 384 KB of fp32 atomics, which the memory side retires at ~1.3 TB/s chip-wide whoever issues them (0.30 us each).  One block: 19 tiles -> 13 slices
    // (202 us modelled, 200-230 measured); eleven blocks: 209 tiles -> no split (1723 / 1720); a compacted student's 162 tiles -> 3 slices (two rounds
    // of a third of the K loop instead of one round on 162 of 256 CUs).
    const int avail = cus - reserved_cus() >= 8 ? cus - reserved_cus() : 8;
    int best = 1;
    double best_cost = 1e30;
    for (int sk = 1; sk <= 64 && nk_total / sk >= 3; ++sk) {
      const long long units = (long long)tiles * sk;
      const double rounds = (double)((units + avail - 1) / avail);
      const double cost = 2.1 * ((nk_total + sk - 1) / sk) * rounds + 0.30 * (double)units;
      if (cost < best_cost - 1e-9) {
        best_cost = cost;
        best = sk;
      }
    }
    split_k = best;
  }
  DEVIT_CHECK(split_k >= 1 && nk_total / split_k >= 3, DEVIT_ERR_SHAPE, "devit_wgrad_grouped: K=%d gives %d K-steps, fewer than 3 per slice at split_k=%d",
              K, nk_total, split_k);
  g.njobs = njobs; g.tiles = tiles; g.split = split_k; g.nk_total = nk_total; g.total = tiles * split_k;
#ifdef DEVIT_GEMMFR_STAMP
  g.dbg = g_wgrad_dbg;
#else
  g.dbg = nullptr;
#endif
  constexpr int lds = (256 + 384) * 128 * 2;
  static bool attr = false;
  if (!attr) {
    hipError_t e = hipFuncSetAttribute((const void*)wgradfr_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    DEVIT_CHECK(e == hipSuccess, DEVIT_ERR_LAUNCH, "hipFuncSetAttribute: %s", hipGetErrorString(e));
    attr = true;
  }
  hipLaunchKernelGGL(wgradfr_kernel, dim3((unsigned)((g.total + 7) / 8 * 8)), dim3(256), lds, (hipStream_t)stream, g);
  DEVIT_LAUNCH_CHECK();
  return DEVIT_OK;
}

extern "C" int devit_set_reserved_cus(int n) {
  DEVIT_CHECK(n >= 0 && n <= 128 && n % 8 == 0, DEVIT_ERR_ARG,
              "devit_set_reserved_cus: %d is not a multiple of 8 in [0, 128] (one share per XCD)", n);
  g_reserved_cus = n;
  return DEVIT_OK;
}

extern "C" int devit_get_reserved_cus(void) { return reserved_cus(); }

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
  if (ep->kind == DEVIT_EPI_DGELU_BF16)
    DEVIT_CHECK(ep->aux_in != nullptr && ep->bias == nullptr, DEVIT_ERR_ARG, "DGELU: needs aux_in, takes no bias (it is a dgrad)");
  if (ep->kind == DEVIT_EPI_ATOMIC_F32 && ep->aux)
    DEVIT_CHECK(batch == 1, DEVIT_ERR_ARG, "ATOMIC: the fused row sums of A (aux) need batch == 1");
  DEVIT_CHECK(ep->exact_gelu == 0, DEVIT_ERR_ARG, "devit_gemm_bf16: exact_gelu=1 (erff) is not built; the fused GELU is the "
              "fitted x * sigmoid(x * P(x^2)) form, |error| <= 2.6e-5 (devit_common.h); precision=\"f32\" uses erff");

  const bool f16 = ep->dtype16 != 0;
  DEVIT_CHECK(ep->dtype16 == 0 || ep->dtype16 == 1, DEVIT_ERR_ARG, "devit_gemm_bf16: dtype16 must be 0 (bf16) or 1 (f16)");
  DEVIT_CHECK(!f16 || (!a_kmajor && !b_kmajor && ep->kind != DEVIT_EPI_DGELU_BF16 && ep->kind != DEVIT_EPI_ATOMIC_F32),
              DEVIT_ERR_ARG, "devit_gemm_bf16: f16 operands are built for the forward layouts / epilogues only");
  GemmArgs g;
  g.A = (const __bf16*)A; g.B = (const __bf16*)B;
  g.lda = lda; g.ldb = ldb;
  g.a_group = Aop->row_group; g.a_skip = Aop->row_skip; g.b_group = Bop->row_group; g.b_skip = Bop->row_skip;
  g.a_bs = Aop->batch_stride; g.b_bs = Bop->batch_stride;
  g.M = M; g.N = N; g.K = K;
  g.tiles_m = 0; g.tiles_n = 0; g.split_k = split_k;
  g.ep = *ep;
  // tile choice: 128x128 (4 waves, two workgroups per CU) or 256x256 (8 waves of 128x64, ping-pong schedule, one per CU)
  const int variant = (a_kmajor ? 2 : 0) + (b_kmajor ? 1 : 0);
  // Measured on the step's shapes (tools/gemm_tiles.py + tools/gpu_tiles.sh, M = 50688, TFLOP/s 256x256 vs 128x128):
  // the ping-pong 256x256 tile wins the long-K and plain-store shapes (teacher qkv 915-950 vs 838, fc2 K 3072 786 vs
  // 778) and the VALU-heavy GELU / dGELU epilogues at any K (student fc1 484 vs 452, teacher fc1 842 vs 817, fc2
  // dgrad 499 vs 481: one wave of each SIMD pair keeps the MFMA pipe while the other is in its epilogue only with two
  // workgroups per CU, but the fused GELU now costs less than the tile's extra fill traffic); the fp32 residual
  // epilogue at K <= 768 stays on 128x128 (teacher proj 458 vs 442), as does everything whose N is not a multiple
  // of 256 and the split-K wgrads.  (A 256x128 tile with a 3-deep ring and a 128x128 tile with a 3-deep ring at one
  // workgroup per CU were built and lost on every shape, warm and cold; they are gone.)
  const bool f16_in = ep->dtype16 != 0;       // (f16 operands run the eight-wave kernels only)
  const bool light_epi = ep->kind == DEVIT_EPI_STORE_BF16 || ep->kind == DEVIT_EPI_STORE_F32;
  const bool gelu_epi = ep->kind == DEVIT_EPI_GELU_BF16 || ep->kind == DEVIT_EPI_DGELU_BF16;
  int cfg = 1;
  // N = 256 k + 128 (student qkv: 1152) runs the 256-wide tile with a half-empty last n-tile: its B rows past N are
  // filled from row N-1 and the waves that own them skip the epilogue (variant 0, bf16 / f32 store only)
  // (also the GELU / dGELU epilogues: hidden 1152 = the compacted student's MLP width at shrink_ratio 0.3)
  const bool ragged_ok = ((variant == 0 && (light_epi || ep->kind == DEVIT_EPI_GELU_BF16)) ||
                          (variant == 1 && ep->kind == DEVIT_EPI_DGELU_BF16)) && N % 256 == 128 && N >= 1024;
  // (the patch-embedding launch of a 768-wide model: 124 -> 108 us in the step on the larger tile; at N = 384 the 128x128 tile stays)
  const bool patch_wide = ep->kind == DEVIT_EPI_PATCH_F32 && K >= 768 && N % 256 == 0;
  // (the fp32 residual epilogue at K = 768, teacher proj: 125.9 us on 128x128 tiles, 120.3 on the four-wave 256x256 kernel, in the step)
  const bool resid_wide = ep->kind == DEVIT_EPI_RESIDUAL_F32 && K >= 768 && N % 256 == 0 && variant == 0 && !f16_in;
  if (M % 256 == 0 && (N % 256 == 0 || ragged_ok) && (K >= 1536 || (K >= DEVIT_RAGGED_MIN_K && light_epi) || gelu_epi || patch_wide || resid_wide) && variant != 3) cfg = 3;
  // too few 256x256 tiles to give every CU one (the token-row GEMMs of the lean last block, M = 512): 128x128 tiles
  // quarter the time of the longest workgroup; same accumulation order per output element either way
  if (cfg == 3 && (long long)(M / 256) * ((N + 255) / 256) * batch < 64) cfg = 1;
  static const int exact = getenv("DEVIT_GEMM_FORCE") ? atoi(getenv("DEVIT_GEMM_FORCE")) : 0;   // tools/gpu_tiles.sh
  if (exact == 1 || (exact == 3 && M % 256 == 0 && (N % 256 == 0 || ragged_ok) && variant != 3)) cfg = exact;
  // the full-row 256x384 kernel: the student's N = 384 launches (round 5; per-shape times inside the step: profiles/r05_*).  DEVIT_GEMMFR=0 / 1
  // forces it off / on for everything it is built for (read per call: tests switch it).
  // (the kernel addresses B densely: a row-remapped k-major B stays on the 128x128 kernel, which honours row_group / row_skip -- except with the fp32
  // residual epilogue, which exists for a k-major B on this kernel only: refused below; DEVIT_GEMM_FORCE=1 means 128x128 tiles for everything)
  const bool b_dense = Bop->row_group == 0 && Bop->row_skip == 0;
  const bool use_fr = !f16 && split_k == 1 && batch == 1 && variant == 1 && b_dense && !(exact == 1 && ep->kind != DEVIT_EPI_RESIDUAL_F32) &&
                      devit_gemm_full_row_selected(M, N, K, ep->kind);
  DEVIT_CHECK(use_fr || !(variant == 1 && ep->kind == DEVIT_EPI_RESIDUAL_F32), DEVIT_ERR_ARG,
              "devit_gemm_bf16: the fp32 residual epilogue with a k-major weight runs on the full-row kernel only (N == 384, "
              "M %% 256 == 0, >= 64 row tiles, K >= 192, DEVIT_GEMMFR != 0, no row_group / row_skip on B): M=%d N=%d K=%d row_group=%d", M, N, K,
              Bop->row_group);
  if (use_fr) cfg = 4;
  const int bm = cfg == 1 ? 128 : 256, bn = cfg == 4 ? 384 : bm;
  g.tiles_m = M / bm;
  g.tiles_n = (N + bn - 1) / bn;
  {
    // Tile order inside an XCD: n-tile fastest inside chunks of gn n-tiles, so that the W workgroups an XCD runs at
    // the same time cover about W / gn m-tiles x gn n-tiles and share their operand panels through the XCD's L2 while
    // they walk K together.  Fill bytes per K-step are ~ (W / gn) BM + gn BN: smallest near gn = sqrt(W BM / BN)
    // (8 for two 128x128 workgroups on each of the 32 CUs, 6 for one 256x256).  Measured with FETCH_SIZE on the
    // teacher fc2 GEMM (K = 3072, three 256-wide n-tiles): gn = 1 fetched the activation panel once per n-tile
    // (1137 MB per launch against 472 MB algorithmic).
    static const int gn_env = getenv("DEVIT_GEMM_GN") ? atoi(getenv("DEVIT_GEMM_GN")) : 0;
    const int per_xcd = (cfg == 1 ? 64 : 32);
    int target = 1;
    while ((target + 1) * (target + 1) * bn <= per_xcd * bm + (target + 1) * bn) ++target;   // ~ round(sqrt(W BM / BN))
    const int nchunks = (g.tiles_n + target - 1) / target;
    g.gn = gn_env > 0 ? gn_env : (g.tiles_n + nchunks - 1) / nchunks;
    if (g.gn > g.tiles_n) g.gn = g.tiles_n;
  }
  const long long tiles = (long long)g.tiles_m * g.tiles_n * split_k * batch;
  DEVIT_CHECK(tiles < (1ll << 31), DEVIT_ERR_SHAPE, "devit_gemm_bf16: too many tiles");
  g.total_tiles = (int)tiles;
  g.d_per_z = make_fastdiv(g.tiles_m * g.tiles_n);
  g.d_chunk = make_fastdiv(g.gn * g.tiles_m);
  g.d_gn = make_fastdiv(g.gn);
  g.d_last = make_fastdiv(g.tiles_n % g.gn ? g.tiles_n % g.gn : g.gn);
  g.d_split = make_fastdiv(split_k);
  // persistent grid: as many workgroups as stay resident (LDS: two 128x128 rings or one 256-wide ring per CU), a
  // multiple of 8 so that every XCD gets the same number
  static int cus = 0;
  if (cus == 0) {
    int dev = 0, n = 0;
    DEVIT_CHECK(hipGetDevice(&dev) == hipSuccess &&
                    hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n >= 8,
                DEVIT_ERR_DEVICE, "devit_gemm_bf16: cannot query the CU count");
    cus = n;
  }
  static const int occ_env = getenv("DEVIT_GEMM_OCC") ? atoi(getenv("DEVIT_GEMM_OCC")) : 0;
  const int occ = cfg == 4 ? 1 : occ_env > 0 ? occ_env : (cfg == 1 ? 2 : 1);
  // A persistent grid holds every CU it starts on (the 256x256 workgroup owns the CU's whole LDS and register file) until
  // its last tile: a collective's kernels launched meanwhile (RCCL on the exchange stream) wait for a GEMM to END, and once
  // they hold CUs the next 256-workgroup grid runs a second, nearly empty round.  With `reserved` CUs left free the grid
  // is smaller and its tiles are dealt over the workgroups that do run (devit_set_reserved_cus; 0 at world size 1).
  const int avail = cus - reserved_cus() >= 8 ? cus - reserved_cus() : 8;
  long long nwg = ((long long)avail * occ) / 8 * 8;
  if (nwg > (tiles + 7) / 8 * 8) nwg = (tiles + 7) / 8 * 8;
  hipStream_t s = (hipStream_t)stream;
#define DEVIT_LAUNCH_ONE_T(BM_, BN_, WMM_, WNN_, NS_, AKM_, BKM_, KIND_, F16_)                                   \
  do {                                                                                                         \
    constexpr int ring = (3 * BM_ + 2 * BN_) * 128, stagebytes = WMM_ * WNN_ * 16384;                             \
    constexpr int lds = (KIND_ == DEVIT_EPI_ATOMIC_F32 && stagebytes > ring) ? stagebytes : ring;              \
    static bool attr = false;                                                                                  \
    if (!attr) {                                                                                               \
      hipError_t e = hipFuncSetAttribute((const void*)gemm_kernel<BM_, BN_, WMM_, WNN_, NS_, AKM_, BKM_, KIND_, F16_>, \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, lds);                     \
      DEVIT_CHECK(e == hipSuccess, DEVIT_ERR_LAUNCH, "hipFuncSetAttribute: %s", hipGetErrorString(e));         \
      attr = true;                                                                                             \
    }                                                                                                          \
    hipLaunchKernelGGL((gemm_kernel<BM_, BN_, WMM_, WNN_, NS_, AKM_, BKM_, KIND_, F16_>), dim3((unsigned)nwg),  \
                       dim3(WMM_* WNN_ * 64), lds, s, g);                                                      \
  } while (0)
#define DEVIT_LAUNCH_ONE(BM_, BN_, WMM_, WNN_, NS_, AKM_, BKM_, KIND_) DEVIT_LAUNCH_ONE_T(BM_, BN_, WMM_, WNN_, NS_, AKM_, BKM_, KIND_, false)
  // forward layouts with f16 operands (ep->dtype16): the frozen teacher
#define DEVIT_LAUNCH_FWD16(BM_, BN_, WMM_, WNN_, NS_, KIND_)                                                   \
  do {                                                                                                         \
    if (f16) DEVIT_LAUNCH_ONE_T(BM_, BN_, WMM_, WNN_, NS_, false, false, KIND_, true);                         \
    else DEVIT_LAUNCH_ONE_T(BM_, BN_, WMM_, WNN_, NS_, false, false, KIND_, false);                            \
  } while (0)
  // the (layout, epilogue) pairs the DeViT path uses; anything else is DEVIT_ERR_ARG
#define DEVIT_LAUNCH_GEMM(BM_, BN_, WMM_, WNN_, NS_)                                                           \
  do {                                                                                                         \
    const int key = variant * 16 + ep->kind;                                                                   \
    switch (key) {                                                                                             \
      case 0 * 16 + DEVIT_EPI_STORE_BF16: DEVIT_LAUNCH_FWD16(BM_, BN_, WMM_, WNN_, NS_, DEVIT_EPI_STORE_BF16); break;       \
      case 0 * 16 + DEVIT_EPI_STORE_F32: DEVIT_LAUNCH_FWD16(BM_, BN_, WMM_, WNN_, NS_, DEVIT_EPI_STORE_F32); break;         \
      case 0 * 16 + DEVIT_EPI_GELU_BF16: DEVIT_LAUNCH_FWD16(BM_, BN_, WMM_, WNN_, NS_, DEVIT_EPI_GELU_BF16); break;         \
      case 0 * 16 + DEVIT_EPI_RESIDUAL_F32: DEVIT_LAUNCH_FWD16(BM_, BN_, WMM_, WNN_, NS_, DEVIT_EPI_RESIDUAL_F32); break;   \
      case 0 * 16 + DEVIT_EPI_PATCH_F32: DEVIT_LAUNCH_FWD16(BM_, BN_, WMM_, WNN_, NS_, DEVIT_EPI_PATCH_F32); break;         \
      case 1 * 16 + DEVIT_EPI_STORE_BF16: DEVIT_LAUNCH_ONE(BM_, BN_, WMM_, WNN_, NS_, false, true, DEVIT_EPI_STORE_BF16); break;        \
      case 1 * 16 + DEVIT_EPI_STORE_F32: DEVIT_LAUNCH_ONE(BM_, BN_, WMM_, WNN_, NS_, false, true, DEVIT_EPI_STORE_F32); break;          \
      case 1 * 16 + DEVIT_EPI_DGELU_BF16: DEVIT_LAUNCH_ONE(BM_, BN_, WMM_, WNN_, NS_, false, true, DEVIT_EPI_DGELU_BF16); break;        \
      case 3 * 16 + DEVIT_EPI_ATOMIC_F32:   /* k-major x k-major never takes the 256x256 tile (cfg 3 excludes variant 3): not */ \
      case 3 * 16 + DEVIT_EPI_STORE_F32:    /* instantiated there (the 256x256 atomic kernel needed 257 registers: 1 spill)       */ \
        if (int rc_ = [&](auto small) -> int {                                                                 \
          if constexpr (decltype(small)::value) {                                                              \
            if (ep->kind == DEVIT_EPI_ATOMIC_F32) DEVIT_LAUNCH_ONE(BM_, BN_, WMM_, WNN_, NS_, true, true, DEVIT_EPI_ATOMIC_F32);  \
            else DEVIT_LAUNCH_ONE(BM_, BN_, WMM_, WNN_, NS_, true, true, DEVIT_EPI_STORE_F32);                  \
            return 0;                                                                                          \
          } else {                                                                                             \
            devit_set_error("devit_gemm_bf16: k-major x k-major operands run on 128x128 tiles only");          \
            return DEVIT_ERR_ARG;                                                                              \
          }                                                                                                    \
        }(std::integral_constant<bool, BM_ == 128>())) return rc_;                                             \
        break;                                                                                                 \
      default:                                                                                                 \
        DEVIT_CHECK(false, DEVIT_ERR_ARG, "devit_gemm_bf16: layout %d with epilogue %d is not instantiated", variant, ep->kind); \
    }                                                                                                          \
  } while (0)
  // the four-wave kernel takes the 256x256 launches it is built for: row-major x row-major, whole 256-wide n-tiles, bf16
  // Which 256x256 launches take it (round 4, per-shape times inside the serialized step, tools/step_gemm_table.py): the two kernels
  // run their K loops at the same fill-bound rate (profiles/r04_a_gemm_four_wave.txt); the four-wave one is 2.5-3.4 % faster where the
  // epilogue is a plain bf16 store or the fp32 residual at K >= 768 (teacher qkv 172.6 -> 168.3 us, fc2 266 -> 257), and 5-10 % SLOWER
  // with the GELU epilogue (one wave per SIMD issues its vector instructions at half the rate two waves share) and on the batched
  // Gram launches.  DEVIT_GEMM4=0 / 1 forces it off / on for everything it is built for (read per call: tests switch it).
  const char* gemm4_env = getenv("DEVIT_GEMM4");
  const bool gemm4_ok = cfg == 3 && variant == 0 && !f16 && N % 256 == 0 && K / BK >= 3 && split_k == 1 &&
                        ep->kind != DEVIT_EPI_DGELU_BF16 && ep->kind != DEVIT_EPI_ATOMIC_F32;
  const bool gemm4_pays = (ep->kind == DEVIT_EPI_STORE_BF16 || ep->kind == DEVIT_EPI_RESIDUAL_F32) && K >= 768 && batch == 1;
  const bool use4 = gemm4_ok && (gemm4_env ? atoi(gemm4_env) != 0 : gemm4_pays);
#define DEVIT_LAUNCH_GEMM4(KIND_)                                                                              \
  do {                                                                                                         \
    constexpr int lds = (3 * 256 + 2 * 256) * 128;                                                             \
    static bool attr = false;                                                                                  \
    if (!attr) {                                                                                               \
      hipError_t e = hipFuncSetAttribute((const void*)gemm4_kernel<KIND_, false>,                              \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, lds);                     \
      DEVIT_CHECK(e == hipSuccess, DEVIT_ERR_LAUNCH, "hipFuncSetAttribute: %s", hipGetErrorString(e));         \
      attr = true;                                                                                             \
    }                                                                                                          \
    hipLaunchKernelGGL((gemm4_kernel<KIND_, false>), dim3((unsigned)nwg), dim3(256), lds, s, g);               \
  } while (0)
#define DEVIT_LAUNCH_GEMMFR(KIND_)                                                                              \
  do {                                                                                                         \
    constexpr int lds = (256 + 384) * 128 * 2;                                                                 \
    static bool attr = false;                                                                                  \
    if (!attr) {                                                                                               \
      hipError_t e = hipFuncSetAttribute((const void*)gemmfr_kernel<KIND_>,                                    \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, lds);                     \
      DEVIT_CHECK(e == hipSuccess, DEVIT_ERR_LAUNCH, "hipFuncSetAttribute: %s", hipGetErrorString(e));         \
      attr = true;                                                                                             \
    }                                                                                                          \
    hipLaunchKernelGGL((gemmfr_kernel<KIND_>), dim3((unsigned)nwg), dim3(256), lds, s, g);                     \
  } while (0)
  if (cfg == 4) {
    if (ep->kind == DEVIT_EPI_STORE_BF16) DEVIT_LAUNCH_GEMMFR(DEVIT_EPI_STORE_BF16);
    else DEVIT_LAUNCH_GEMMFR(DEVIT_EPI_RESIDUAL_F32);
  } else if (use4) {
    switch (ep->kind) {
      case DEVIT_EPI_STORE_BF16: DEVIT_LAUNCH_GEMM4(DEVIT_EPI_STORE_BF16); break;
      case DEVIT_EPI_STORE_F32: DEVIT_LAUNCH_GEMM4(DEVIT_EPI_STORE_F32); break;
      case DEVIT_EPI_GELU_BF16: DEVIT_LAUNCH_GEMM4(DEVIT_EPI_GELU_BF16); break;
      case DEVIT_EPI_RESIDUAL_F32: DEVIT_LAUNCH_GEMM4(DEVIT_EPI_RESIDUAL_F32); break;
      case DEVIT_EPI_PATCH_F32: DEVIT_LAUNCH_GEMM4(DEVIT_EPI_PATCH_F32); break;
      default: DEVIT_CHECK(false, DEVIT_ERR_ARG, "devit_gemm_bf16: epilogue %d has no four-wave instantiation", ep->kind);
    }
  } else if (cfg == 3) DEVIT_LAUNCH_GEMM(256, 256, 2, 4, 2);
  else DEVIT_LAUNCH_GEMM(128, 128, 2, 2, 2);
#undef DEVIT_LAUNCH_GEMM4
#undef DEVIT_LAUNCH_GEMMFR
#undef DEVIT_LAUNCH_ONE
#undef DEVIT_LAUNCH_ONE_T
#undef DEVIT_LAUNCH_FWD16
#undef DEVIT_LAUNCH_GEMM
  DEVIT_LAUNCH_CHECK();
  return DEVIT_OK;
}
