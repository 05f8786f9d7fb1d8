// Shared device helpers for the DeViT gfx950 kernels (wave = 64 lanes, MFMA 16x16x32 bf16).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/devit_hip.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) short short4v;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;

#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define GLB_PTR(p) ((const __attribute__((address_space(1))) void*)(p))

// ---- error plumbing (host) -------------------------------------------------------------
void devit_set_error(const char* fmt, ...);
#define DEVIT_CHECK(cond, code, ...)      \
  do {                                    \
    if (!(cond)) {                        \
      devit_set_error(__VA_ARGS__);       \
      return (code);                      \
    }                                     \
  } while (0)
#define DEVIT_LAUNCH_CHECK()                                            \
  do {                                                                  \
    hipError_t e__ = hipGetLastError();                                 \
    if (e__ != hipSuccess) {                                            \
      devit_set_error("%s:%d launch failed: %s", __FILE__, __LINE__,    \
                      hipGetErrorString(e__));                          \
      return DEVIT_ERR_LAUNCH;                                          \
    }                                                                   \
  } while (0)


// ---- scalar helpers ---------------------------------------------------------------------
__device__ __forceinline__ float bf2f(__bf16 v) { return (float)v; }
__device__ __forceinline__ __bf16 f2bf(float v) { return (__bf16)v; }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// Fused-epilogue GELU (EXACT == false): x * Phi(x) with Phi(x) ~= sigmoid(x * (c0 + c1 x^2 + c2 x^4)), x^2 clamped at 36.
// The three coefficients are a minimax fit to the erf form on [-8, 8]: |gelu error| <= 2.6e-5, |gelu' error| <= 1.2e-4
// (tools/fit_gelu.py), i.e. 1/150 and 1/35 of a bf16 rounding step at |y| = 1 -- the outputs are stored in bf16.  Two
// transcendentals (v_exp, v_rcp) and seven plain VALU ops per element, against two + twelve for the A&S erf form; the
// epilogue of a short-K GEMM is VALU-bound, so this is wall time.  The f32 parity kernels use erff() (EXACT).
constexpr float GELU_C0 = 1.59500523f, GELU_C1 = 7.40208836e-2f, GELU_C2 = -7.04591421e-4f;
constexpr float LOG2E = 1.44269504088896341f;
template <bool EXACT>
__device__ __forceinline__ float gelu_fwd(float x) {
  if (EXACT) return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f));
  const float x2 = fminf(x * x, 36.0f);
  const float p = fmaf(fmaf(-GELU_C2 * LOG2E, x2, -GELU_C1 * LOG2E), x2, -GELU_C0 * LOG2E);   // -log2(e) * poly
  const float t = __builtin_amdgcn_exp2f(x * p);                                               // exp(-u)
  return x * __builtin_amdgcn_rcpf(1.0f + t);
}
// d/dx [x * Phi(x)] = Phi(x) + x * phi(x); for the fused form the derivative of the fitted function:
// s + x s (1 - s) u',  s = sigmoid(u), u = x P(x^2), u' = c0 + 3 c1 x^2 + 5 c2 x^4.
template <bool EXACT>
__device__ __forceinline__ float gelu_bwd(float x) {
  if (EXACT) {
    const float pdf = 0.3989422804014327f * __expf(-0.5f * x * x);
    return 0.5f * (1.0f + erff(x * 0.70710678118654752f)) + x * pdf;
  }
  const float x2 = fminf(x * x, 36.0f);
  const float p = fmaf(fmaf(-GELU_C2 * LOG2E, x2, -GELU_C1 * LOG2E), x2, -GELU_C0 * LOG2E);
  const float q = fmaf(fmaf(5.0f * GELU_C2, x2, 3.0f * GELU_C1), x2, GELU_C0);
  const float s = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * p));
  return fmaf(x * fmaf(-s, s, s), q, s);   // s (1 - s) as s - s^2: finite for t = inf
}

// ---- streaming global accesses ------------------------------------------------------------------------
// Measured on MI355X (gemm_bench, T qkv 50688x2304x768): write-through (sc1) 8/16-byte epilogue stores cost
// -12 % (833 -> 734 TF; narrow sc1 stores are one fabric write each), so outputs use plain stores.
__device__ __forceinline__ void store16_stream(void* p, f32x4 v) { *(f32x4*)p = v; }
__device__ __forceinline__ void store8_stream(void* p, bf16x4 v) { *(bf16x4*)p = v; }
template <typename T>
__device__ __forceinline__ T load_stream(const T* p) {   // read-once data: non-temporal hint
  return __builtin_nontemporal_load(p);
}

// ---- MFMA 16x16x32 bf16 -------------------------------------------------------------------
// A frag: lane l holds A[row l&15][k = 8*(l>>4) + j], j = 0..7
// B frag: lane l holds B[k = 8*(l>>4) + j][col l&15]
// C/D   : lane l holds D[row 4*(l>>4) + r][col l&15], r = 0..3
__device__ __forceinline__ f32x4 mfma16(bf16x8 a, bf16x8 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

// ---- the second 16-bit storage type: IEEE f16 ---------------------------------------------------------
// The frozen teacher's forward can run with f16 instead of bf16 operands and stored activations (11 significand bits
// instead of 8: DeiT-B logits 1.1e-3 from the fp32 reference instead of 6.8e-3, profiles/r02_b_f16_localise.txt; same MFMA
// rate per the ISA, the step measures 1.4 % slower; no gradients, so no loss scaling).  Fragments travel through LDS and registers as raw 16-bit lanes typed bf16x8;
// only the MFMA opcode and the float -> 16-bit conversions differ.
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
template <bool F16>
__device__ __forceinline__ f32x4 mfma16t(bf16x8 a, bf16x8 b, f32x4 c) {
  if constexpr (F16)
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  else
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
// four / eight floats -> 16-bit lanes (round to nearest even in both types), returned as raw bf16-typed lanes
template <bool F16>
__device__ __forceinline__ bf16x4 cvt4(f32x4 v) {
  if constexpr (F16) {
    const f16x4 h = {(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]};
    return __builtin_bit_cast(bf16x4, h);
  } else {
    return (bf16x4){f2bf(v[0]), f2bf(v[1]), f2bf(v[2]), f2bf(v[3])};
  }
}
template <bool F16>
__device__ __forceinline__ bf16x8 cvt8(f32x4 a, f32x4 b) {
  if constexpr (F16) {
    const f16x8 h = {(_Float16)a[0], (_Float16)a[1], (_Float16)a[2], (_Float16)a[3],
                     (_Float16)b[0], (_Float16)b[1], (_Float16)b[2], (_Float16)b[3]};
    return __builtin_bit_cast(bf16x8, h);
  } else {
    return (bf16x8){f2bf(a[0]), f2bf(a[1]), f2bf(a[2]), f2bf(a[3]), f2bf(b[0]), f2bf(b[1]), f2bf(b[2]), f2bf(b[3])};
  }
}

// Lanes c and c^8 of each 16-lane row trade one 16-byte chunk so that the two stores of an m-tile each write whole
// 128-byte rows: on entry lane (g, c) holds row c's bytes [16g, 16g+16) in `a` and [64+16g, 64+16g+16) in `b`; on exit
// `a` belongs to row (c & 7) and `b` to row 8 + (c & 7), both at byte (c >= 8 ? 64 : 0) + 16g.  The exchange is its own
// inverse: applied to chunks LOADED in the second shape it yields the first (attention's Q fragments).
template <typename V>   // any 16-byte vector: bf16x8 (64 output columns = one 128-byte row) or f32x4 (two n-tiles = one)
__device__ __forceinline__ void swap_half_rows(V& a, V& b, bool hi) {
  static_assert(sizeof(V) == 16, "16-byte chunks");
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  const u32x4 ua = __builtin_bit_cast(u32x4, a), ub = __builtin_bit_cast(u32x4, b);
  u32x4 recv;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const unsigned send = hi ? ua[e] : ub[e];
    recv[e] = __builtin_amdgcn_update_dpp(0u, send, 0x128 /* row_ror:8 */, 0xf, 0xf, false);
  }
  u32x4 oa, obb;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    oa[e] = hi ? recv[e] : ua[e];
    obb[e] = hi ? ub[e] : recv[e];
  }
  a = __builtin_bit_cast(V, oa);
  b = __builtin_bit_cast(V, obb);
}

// Transposed LDS read: within each 16-lane group, lane 4q+p supplies the address of row q,
// columns 4p..4p+3 of a 4x16 block of 16-bit elements; lane i receives column i (rows 0..3).
__device__ __forceinline__ bf16x4 lds_tr_read(const void* lds_addr) {
  short4v v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4v*)(lds_addr));
  return __builtin_bit_cast(bf16x4, v);
}
__device__ __forceinline__ bf16x8 cat8(bf16x4 lo, bf16x4 hi) {
  return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}
