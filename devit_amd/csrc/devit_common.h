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

// erf with |err| < 1.5e-7 (Abramowitz & Stegun 7.1.26): enough for bf16-stored activations;
// the f32 parity kernels use erff().
__device__ __forceinline__ float fast_erf(float x) {
  const float ax = fabsf(x);
  const float t = __frcp_rn(fmaf(0.3275911f, ax, 1.0f));
  float p = fmaf(1.061405429f, t, -1.453152027f);
  p = fmaf(p, t, 1.421413741f);
  p = fmaf(p, t, -0.284496736f);
  p = fmaf(p, t, 0.254829592f);
  const float e = __expf(-ax * ax);
  const float r = fmaf(-p * t, e, 1.0f);
  return copysignf(r, x);
}
template <bool EXACT>
__device__ __forceinline__ float gelu_fwd(float x) {
  const float e = EXACT ? erff(x * 0.70710678118654752f) : fast_erf(x * 0.70710678118654752f);
  return 0.5f * x * (1.0f + e);
}
// d/dx [x * Phi(x)] = Phi(x) + x * phi(x)
template <bool EXACT>
__device__ __forceinline__ float gelu_bwd(float x) {
  const float e = EXACT ? erff(x * 0.70710678118654752f) : fast_erf(x * 0.70710678118654752f);
  const float pdf = 0.3989422804014327f * __expf(-0.5f * x * x);
  return 0.5f * (1.0f + e) + x * pdf;
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

// Transposed LDS read: within each 16-lane group, lane 4q+p supplies the address of row q,
// columns 4p..4p+3 of a 4x16 block of 16-bit elements; lane i receives column i (rows 0..3).
__device__ __forceinline__ bf16x4 lds_tr_read(const void* lds_addr) {
  short4v v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4v*)(lds_addr));
  return __builtin_bit_cast(bf16x4, v);
}
__device__ __forceinline__ bf16x8 cat8(bf16x4 lo, bf16x4 hi) {
  return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}
