// LayerNorm forward / backward for the fp32 residual stream (one wave per token row).
// HBM-bound: every row is read once (fwd) or read once + written once (bwd); statistics are fp32.
// Replaces nn.LayerNorm(D, eps=1e-6) at models/de_vit.py:113,115,286 and its autograd backward.
#include <type_traits>

#include "devit_common.h"

namespace {

constexpr int LN_MAX_NV = 8;  // D <= 1024, D % 128 == 0 ; half-wave lane holds NV float4

struct LnFwdArgs {
  const float* x;
  const float* gamma;
  const float* beta;
  __bf16* y_bf16;
  float* y_f32;
  float* mean;
  float* rstd;
  int rows, D, in_group, in_stride;
  float eps;
  int f16;              // y_bf16 holds IEEE f16 instead of bf16 (frozen-teacher forward)
};

__device__ __forceinline__ size_t ln_in_row(int r, int group, int stride) {
  return group > 0 ? (size_t)(r / group) * stride + (r % group) : (size_t)r;
}

// Half a wave (32 lanes) per row, 16-byte loads: lane l holds float4 v of the row at columns (v * 32 + l) * 4,
// NV = D / 128 of them; bf16 output leaves as 8-byte stores.  (The first version read float2 per lane with one wave
// per row: 2.6-3.1 TB/s from cold HBM, tools/ln_cold.py.)  Statistics: two-pass (mean, then centred sum of squares)
// in fp32, reduced over the 32 lanes by xor shuffles.
// RAG: D is a multiple of 4 but not of 128 (D = 192: the three narrow registered geometries, which run on the exact-fp32 path): NV = ceil(D / 128), lanes
// whose float4 lies past D hold zeros and store nothing.
template <int NV, bool RAG = false>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const LnFwdArgs a) {
  const int lane = threadIdx.x & 63, l32 = lane & 31, sub = lane >> 5;
  const int half_global = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 2 + sub;
  const int nhalves = gridDim.x * 8;
  f32x4 gm[NV], bt[NV];
  bool ok[NV];
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    ok[v] = !RAG || (v * 32 + l32) * 4 < a.D;
    gm[v] = ok[v] ? *(const f32x4*)(a.gamma + (v * 32 + l32) * 4) : (f32x4){0.f, 0.f, 0.f, 0.f};
    bt[v] = ok[v] ? *(const f32x4*)(a.beta + (v * 32 + l32) * 4) : (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  const float invD = 1.0f / (float)a.D;
  // both halves of a wave run the same number of trips (the shuffles need every lane): a half past the end redoes
  // the last row and skips its stores
  const int trips = (a.rows + nhalves - 1) / nhalves;
  for (int it = 0; it < trips; ++it) {
    const int rr = half_global + it * nhalves;
    const bool live = rr < a.rows;
    const int r = live ? rr : a.rows - 1;
    const float* xr = a.x + ln_in_row(r, a.in_group, a.in_stride) * a.D;
    f32x4 xv[NV];
    float s = 0.f;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      xv[v] = ok[v] ? load_stream((const f32x4*)(xr + (v * 32 + l32) * 4)) : (f32x4){0.f, 0.f, 0.f, 0.f};
      s += (xv[v][0] + xv[v][1]) + (xv[v][2] + xv[v][3]);
    }
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    const float mu = s * invD;
    float q = 0.f;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float d = ok[v] ? xv[v][e] - mu : 0.f;
        q = fmaf(d, d, q);
      }
    }
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) q += __shfl_xor(q, o, 64);
    const float rs = rsqrtf(q * invD + a.eps);
    if (!live) continue;
    if (l32 == 0) {
      if (a.mean) a.mean[r] = mu;
      if (a.rstd) a.rstd[r] = rs;
    }
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      const f32x4 y = (xv[v] - mu) * rs * gm[v] + bt[v];
      const size_t o = (size_t)r * a.D + (v * 32 + l32) * 4;
      if (RAG && !ok[v]) continue;
      if (a.y_bf16) *(bf16x4*)(a.y_bf16 + o) = a.f16 ? cvt4<true>(y) : cvt4<false>(y);
      if (a.y_f32) *(f32x4*)(a.y_f32 + o) = y;
    }
  }
}

struct LnBwdArgs {
  const void* dy;       // [rows][D] bf16 or f32 (dense, row r)
  const float* x;       // forward input, physical row map as in fwd
  const float* mean;
  const float* rstd;
  const float* gamma;
  const float* dres;    // [phys rows][D] f32 upstream residual-stream gradient or NULL
  float* dx;            // [phys rows][D] f32 = dres + LN'(dy)
  __bf16* dx_bf16;      // optional bf16 copy of rowscale * dx (branch gradient for the next GEMMs)
  const float* rowscale;
  int rows_per_scale;
  float* partial;       // [grid][3][D] column partial sums (dgamma, dbeta, colsum of dx_bf16)
  int rows, D, in_group, in_stride, dy_is_f32;
};

// Half a wave (32 lanes x 4*NV columns) per token row, two rows per wave pass: 16-byte loads of x / dres and 8-byte
// loads of the bf16 dy keep ~2.5 KB per wave in flight (HBM-bound kernel).
template <int NV, bool RAG = false>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const LnBwdArgs a) {
  __shared__ float red[4][3][NV * 128];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, hl = lane & 31, half = lane >> 5;
  const int slot = (blockIdx.x * 4 + wv) * 2 + half;       // half-wave id
  const int nslots = gridDim.x * 8;
  f32x4 gm[NV], dg[NV], db[NV], dsum[NV];
  bool ok[NV];
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    ok[v] = !RAG || v * 128 + hl * 4 < a.D;
    gm[v] = ok[v] ? *(const f32x4*)(a.gamma + v * 128 + hl * 4) : (f32x4){0.f, 0.f, 0.f, 0.f};
    dg[v] = (f32x4){0.f, 0.f, 0.f, 0.f};
    db[v] = (f32x4){0.f, 0.f, 0.f, 0.f};
    dsum[v] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  const float invD = 1.0f / (float)a.D;
  const int nit = (a.rows + nslots - 1) / nslots;           // wave-uniform trip count (shuffles need all lanes)
  for (int it = 0; it < nit; ++it) {
    const int r = slot + it * nslots;
    const bool live = r < a.rows;
    const size_t pr = ln_in_row(live ? r : 0, a.in_group, a.in_stride);
    const float* xr = a.x + pr * a.D;
    const float mu = live ? a.mean[r] : 0.f, rs = live ? a.rstd[r] : 0.f;
    f32x4 xh[NV], g[NV], dyv[NV];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      const int c = v * 128 + hl * 4;
      f32x4 xv = {0.f, 0.f, 0.f, 0.f};
      dyv[v] = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (live && ok[v]) {
        xv = load_stream((const f32x4*)(xr + c));
        if (a.dy_is_f32) {
          dyv[v] = load_stream((const f32x4*)((const float*)a.dy + (size_t)r * a.D + c));
        } else {
          const bf16x4 t = load_stream((const bf16x4*)((const __bf16*)a.dy + (size_t)r * a.D + c));
          dyv[v] = (f32x4){bf2f(t[0]), bf2f(t[1]), bf2f(t[2]), bf2f(t[3])};
        }
      }
      xh[v] = ok[v] ? (xv - mu) * rs : (f32x4){0.f, 0.f, 0.f, 0.f};
      g[v] = dyv[v] * gm[v];
      dg[v] += dyv[v] * xh[v];
      db[v] += dyv[v];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        s1 += g[v][e];
        s2 += g[v][e] * xh[v][e];
      }
    }
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) {
      s1 += __shfl_xor(s1, o, 64);
      s2 += __shfl_xor(s2, o, 64);
    }
    const float m1 = s1 * invD, m2 = s2 * invD;
    if (live) {
      const float rsc = (a.dx_bf16 && a.rowscale) ? a.rowscale[pr / a.rows_per_scale] : 1.0f;
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        const size_t o = pr * a.D + v * 128 + hl * 4;
        if (RAG && !ok[v]) continue;
        f32x4 d = rs * (g[v] - m1 - xh[v] * m2);
        if (a.dres) d += load_stream((const f32x4*)(a.dres + o));
        *(f32x4*)(a.dx + o) = d;
        if (a.dx_bf16) {
          const bf16x4 ob = {f2bf(d[0] * rsc), f2bf(d[1] * rsc), f2bf(d[2] * rsc), f2bf(d[3] * rsc)};
          *(bf16x4*)(a.dx_bf16 + o) = ob;
          dsum[v] += (f32x4){bf2f(ob[0]), bf2f(ob[1]), bf2f(ob[2]), bf2f(ob[3])};
        }
      }
    }
  }
  // block reduction of the column sums -> partial[block][{dgamma, dbeta, colsum(dx_bf16)}][D]
#pragma unroll
  for (int v = 0; v < NV; ++v) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {      // the two half-waves hold the same columns
      dg[v][e] += __shfl_xor(dg[v][e], 32, 64);
      db[v][e] += __shfl_xor(db[v][e], 32, 64);
      dsum[v][e] += __shfl_xor(dsum[v][e], 32, 64);
    }
    if (half == 0 && ok[v]) {
      *(f32x4*)&red[wv][0][v * 128 + hl * 4] = dg[v];
      *(f32x4*)&red[wv][1][v * 128 + hl * 4] = db[v];
      *(f32x4*)&red[wv][2][v * 128 + hl * 4] = dsum[v];
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 3 * a.D; i += 256) {
    const int which = i / a.D, c = i - which * a.D;
    a.partial[((size_t)blockIdx.x * 3 + which) * a.D + c] =
        red[0][which][c] + red[1][which][c] + red[2][which][c] + red[3][which][c];
  }
}

// out_k[c] (+)= sum_p partial[p][k][c], k = 0..2  (deterministic order; 32 columns x 32 part-groups per block)
__global__ __launch_bounds__(1024) void colsum_partials_kernel(const float* partial, int nparts, int D, float* out0,
                                                               float* out1, float* out2, int accumulate) {
  __shared__ float red[32][33];
  const int ncols = 3 * D;
  const int cl = threadIdx.x & 31, grp = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + cl;
  float s = 0.f;
  if (c < ncols) {
#pragma unroll 4
    for (int p = grp; p < nparts; p += 32) s += partial[(size_t)p * ncols + c];
  }
  red[grp][cl] = s;
  __syncthreads();
  if (grp == 0 && c < ncols) {
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < 32; ++i) t += red[i][cl];
    float* dst = c < D ? out0 + c : (c < 2 * D ? out1 + (c - D) : (out2 ? out2 + (c - 2 * D) : nullptr));
    if (dst) *dst = accumulate ? *dst + t : t;
  }
}

template <typename Args, typename F>
int dispatch_nv(int D, F&& f) {
  if (D % 128) {         // narrow geometries (D = 192): ceil(D / 128) vectors per lane, the last one ragged
    switch ((D + 127) / 128) {
      case 1: f(std::integral_constant<int, 1>(), std::true_type()); return 0;
      case 2: f(std::integral_constant<int, 2>(), std::true_type()); return 0;
      case 3: f(std::integral_constant<int, 3>(), std::true_type()); return 0;
      default: return -1;
    }
  }
  switch (D / 128) {
    case 1: f(std::integral_constant<int, 1>(), std::false_type()); return 0;
    case 2: f(std::integral_constant<int, 2>(), std::false_type()); return 0;
    case 3: f(std::integral_constant<int, 3>(), std::false_type()); return 0;
    case 4: f(std::integral_constant<int, 4>(), std::false_type()); return 0;
    case 6: f(std::integral_constant<int, 6>(), std::false_type()); return 0;
    case 8: f(std::integral_constant<int, 8>(), std::false_type()); return 0;
    default: return -1;
  }
}

}  // namespace

extern "C" int devit_layernorm_fwd(const float* x, int rows, int D, int in_group, int in_stride,
                                   const float* gamma, const float* beta, float eps, void* y_bf16, float* y_f32,
                                   float* mean, float* rstd, int dtype16, void* stream) {
  DEVIT_CHECK(x && gamma && beta && (y_bf16 || y_f32), DEVIT_ERR_ARG, "devit_layernorm_fwd: null pointer");
  DEVIT_CHECK(dtype16 == 0 || dtype16 == 1, DEVIT_ERR_ARG, "devit_layernorm_fwd: dtype16 must be 0 (bf16) or 1 (f16)");
  DEVIT_CHECK(rows > 0 && D > 0 && D % 64 == 0 && D <= LN_MAX_NV * 128, DEVIT_ERR_SHAPE,
              "devit_layernorm_fwd: D=%d must be a multiple of 64 (of 128 above 384), <= 1024", D);
  LnFwdArgs a{x, gamma, beta, (__bf16*)y_bf16, y_f32, mean, rstd, rows, D, in_group, in_stride, eps, dtype16};
  const int grid = rows < 8 * 2048 ? (rows + 7) / 8 : 2048;   // 8 half-waves (rows in flight) per 256-thread block
  int rc = dispatch_nv<LnFwdArgs>(D, [&](auto nv, auto rag) {
    hipLaunchKernelGGL((ln_fwd_kernel<decltype(nv)::value, decltype(rag)::value>), dim3(grid), dim3(256), 0, (hipStream_t)stream, a);
  });
  DEVIT_CHECK(rc == 0, DEVIT_ERR_SHAPE, "devit_layernorm_fwd: unsupported D=%d", D);
  DEVIT_LAUNCH_CHECK();
  return DEVIT_OK;
}

static int ln_bwd_grid(int rows) { return rows < 8 * 1024 ? (rows + 7) / 8 : 1024; }

extern "C" size_t devit_layernorm_bwd_workspace(int rows, int D) {
  return (size_t)ln_bwd_grid(rows) * 3 * D * sizeof(float);
}

extern "C" int devit_layernorm_bwd(const void* dy, int dy_is_f32, const float* x, int rows, int D, int in_group,
                                   int in_stride, const float* mean, const float* rstd, const float* gamma,
                                   const float* dres, float* dx, void* dx_bf16, const float* rowscale,
                                   int rows_per_scale, float* dgamma, float* dbeta, float* dx_bf16_colsum,
                                   int accumulate, void* workspace, size_t workspace_bytes, void* stream) {
  DEVIT_CHECK(dy && x && mean && rstd && gamma && dx && dgamma && dbeta && workspace, DEVIT_ERR_ARG,
              "devit_layernorm_bwd: null pointer");
  DEVIT_CHECK(rows > 0 && D > 0 && D % 64 == 0 && D <= LN_MAX_NV * 128, DEVIT_ERR_SHAPE, "devit_layernorm_bwd: D=%d", D);
  DEVIT_CHECK(workspace_bytes >= devit_layernorm_bwd_workspace(rows, D), DEVIT_ERR_ARG,
              "devit_layernorm_bwd: workspace too small");
  DEVIT_CHECK(!rowscale || rows_per_scale > 0, DEVIT_ERR_ARG, "devit_layernorm_bwd: rows_per_scale");
  DEVIT_CHECK(!dx_bf16_colsum || dx_bf16, DEVIT_ERR_ARG, "devit_layernorm_bwd: dx_bf16_colsum needs dx_bf16");
  const int grid = ln_bwd_grid(rows);
  LnBwdArgs a{dy, x, mean, rstd, gamma, dres, dx, (__bf16*)dx_bf16, rowscale, rows_per_scale,
              (float*)workspace, rows, D, in_group, in_stride, dy_is_f32};
  int rc = dispatch_nv<LnBwdArgs>(D, [&](auto nv, auto rag) {
    hipLaunchKernelGGL((ln_bwd_kernel<decltype(nv)::value, decltype(rag)::value>), dim3(grid), dim3(256), 0, (hipStream_t)stream, a);
  });
  DEVIT_CHECK(rc == 0, DEVIT_ERR_SHAPE, "devit_layernorm_bwd: unsupported D=%d", D);
  DEVIT_LAUNCH_CHECK();
  hipLaunchKernelGGL(colsum_partials_kernel, dim3((3 * D + 31) / 32), dim3(1024), 0, (hipStream_t)stream,
                     (const float*)workspace, grid, D, dgamma, dbeta, dx_bf16_colsum, accumulate);
  DEVIT_LAUNCH_CHECK();
  return DEVIT_OK;
}
