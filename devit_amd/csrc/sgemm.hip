// Exact-fp32 companion path ("parity mode"): a generic strided fp32 GEMM with the same fused epilogues as the bf16
// MFMA GEMM, plus the row-softmax kernels that turn it into attention.  Every product is an fmaf chain in k order,
// every activation stays fp32, GELU uses erff(): results track the reference's fp32 CPU path to ~1e-6.  Built for
// correctness, not speed (a few TFLOP/s): BASELINE.json's "logits within 1e-3 rel" is asserted through this path
// (tests/test_gpu_model.py, precision="f32"); the bf16 path is the one that is benchmarked.
#include "devit_common.h"

namespace {

struct SgemmArgs {
  const float* A; long long sam, sak, a_bo, a_bi;   // element (m, k) at A[m*sam + k*sak], batch offsets outer/inner
  const float* B; long long sbn, sbk, b_bo, b_bi;
  long long c_bo, c_bi;
  int M, N, K, batch_inner;
  int k_group, k_skip;                                // physical k of A only = k + skip * (k / group + 1)
  float alpha;
  const float* batch_scale;                           // [batch_inner] multiplies alpha (head gate) or NULL
  int accumulate;                                     // STORE_F32: out += result
  devit_epilogue ep;
};

template <int KIND>
__global__ __launch_bounds__(256) void sgemm_kernel(const SgemmArgs g) {
  __shared__ float As[16][65], Bs[16][65];
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const int z = blockIdx.z, zo = z / g.batch_inner, zi = z % g.batch_inner;
  const float* A = g.A + zo * g.a_bo + zi * g.a_bi;
  const float* B = g.B + zo * g.b_bo + zi * g.b_bi;
  const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
  float acc[4][4] = {};
  for (int k0 = 0; k0 < g.K; k0 += 16) {
    for (int i = threadIdx.x; i < 16 * 64; i += 256) {
      const int kk = i >> 6, r = i & 63;
      const int k = k0 + kk;
      const long long pk = g.k_group > 0 ? (long long)k + (long long)g.k_skip * (k / g.k_group + 1) : k;
      As[kk][r] = (k < g.K && m0 + r < g.M) ? A[(long long)(m0 + r) * g.sam + pk * g.sak] : 0.f;
      Bs[kk][r] = (k < g.K && n0 + r < g.N) ? B[(long long)(n0 + r) * g.sbn + (long long)k * g.sbk] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) {
      float a[4], b[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) { a[i] = As[kk][ty * 4 + i]; b[i] = Bs[kk][tx * 4 + i]; }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(a[i], b[j], acc[i][j]);
    }
    __syncthreads();
  }
  const devit_epilogue& ep = g.ep;
  const float alpha = g.alpha * (g.batch_scale ? g.batch_scale[zi] : 1.0f);
  const long long cb = zo * g.c_bo + zi * g.c_bi;
  const int m_lim = ep.m_valid > 0 ? ep.m_valid : g.M;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = m0 + ty * 4 + i;
    if (m >= m_lim) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = n0 + tx * 4 + j;
      if (n >= g.N) continue;
      float v = acc[i][j] * alpha + (ep.bias ? ep.bias[n] : 0.f);
      const float cs = ep.colscale ? ep.colscale[n] : 1.0f;
      long long o = cb + (long long)m * ep.ldc + n;
      if (KIND == DEVIT_EPI_STORE_F32) {
        float* out = (float*)ep.out;
        out[o] = g.accumulate ? out[o] + v : v;
      } else if (KIND == DEVIT_EPI_GELU_BF16) {           // fp32 flavour: out = gelu(v) * gate, aux = v
        if (ep.aux) ((float*)ep.aux)[o] = v;
        ((float*)ep.out)[o] = gelu_fwd<true>(v) * cs;
      } else if (KIND == DEVIT_EPI_DGELU_BF16) {
        ((float*)ep.out)[o] = v * cs * gelu_bwd<true>(((const float*)ep.aux_in)[o]);
      } else if (KIND == DEVIT_EPI_RESIDUAL_F32) {
        if (ep.aux) ((float*)ep.aux)[o] = v;
        const float rs = ep.rowscale ? ep.rowscale[m / ep.rows_per_scale] : 1.0f;
        ((float*)ep.out)[o] = ep.res[o] + rs * v;
      } else if (KIND == DEVIT_EPI_PATCH_F32) {
        const int b = m / ep.patch_tokens, t = m - b * ep.patch_tokens, tok = ep.extra_tokens + t;
        o = ((long long)b * (ep.patch_tokens + ep.extra_tokens) + tok) * ep.ldc + n;
        ((float*)ep.out)[o] = v + ep.pos[(long long)tok * ep.ldc + n];
      }
    }
  }
}

// P[r][j] = softmax_j(scale * S[r][j]) in place; lse[r] = log sum exp (natural).  One wave per row.
__global__ __launch_bounds__(256) void softmax_rows_kernel(float* S, int rows, int ncols, int ld, float scale, float* lse) {
  const int lane = threadIdx.x & 63, r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  float* row = S + (size_t)r * ld;
  float mx = -INFINITY;
  for (int j = lane; j < ncols; j += 64) mx = fmaxf(mx, row[j] * scale);
  mx = wave_max(mx);
  float sum = 0.f;
  for (int j = lane; j < ncols; j += 64) sum += expf(row[j] * scale - mx);
  sum = wave_sum(sum);
  for (int j = lane; j < ncols; j += 64) row[j] = expf(row[j] * scale - mx) / sum;
  if (lse && lane == 0) lse[r] = mx + logf(sum);
}
// dS[r][j] = scale * P[r][j] * (dP[r][j] - sum_j P dP)   (softmax backward), in place over dP
__global__ __launch_bounds__(256) void softmax_bwd_rows_kernel(const float* P, float* dP, int rows, int ncols, int ld, float scale) {
  const int lane = threadIdx.x & 63, r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  const float* p = P + (size_t)r * ld;
  float* d = dP + (size_t)r * ld;
  float dot = 0.f;
  for (int j = lane; j < ncols; j += 64) dot += p[j] * d[j];
  dot = wave_sum(dot);
  for (int j = lane; j < ncols; j += 64) d[j] = scale * p[j] * (d[j] - dot);
}

// fp32 helpers of the parity path
__global__ __launch_bounds__(256) void im2row_f32_kernel(const float* img, float* rows, int B) {
  const int total = B * 196 * 768;
  for (int idx = blockIdx.x * 256 + threadIdx.x; idx < total; idx += gridDim.x * 256) {
    const int k = idx % 768, row = idx / 768;
    const int b = row / 196, t = row % 196, py = t / 14, px = t % 14, c = k >> 8, kh = (k >> 4) & 15, kw = k & 15;
    rows[idx] = img[(((size_t)b * 3 + c) * 224 + py * 16 + kh) * 224 + px * 16 + kw];
  }
}
__global__ __launch_bounds__(256) void scale_rows_f32_kernel(const float* src, float* dst, const float* rowscale,
                                                             int rows_per_scale, int M, int D) {
  const size_t total = (size_t)M * D;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256)
    dst[i] = src[i] * (rowscale ? rowscale[(i / D) / rows_per_scale] : 1.0f);
}
__global__ __launch_bounds__(256) void colsum_f32_kernel(const float* y, int M, int N, int ld, float* out, int accumulate) {
  const int n = blockIdx.x * 256 + threadIdx.x;
  if (n >= N) return;
  float s = 0.f;
  for (int m = 0; m < M; ++m) s += y[(size_t)m * ld + n];
  out[n] = accumulate ? out[n] + s : s;
}

}  // namespace

extern "C" int devit_gemm_f32(const float* A, long long sam, long long sak, long long a_bs_outer, long long a_bs_inner,
                              const float* B, long long sbn, long long sbk, long long b_bs_outer, long long b_bs_inner,
                              int M, int N, int K, int batch, int batch_inner, long long c_bs_outer, long long c_bs_inner,
                              int k_group, int k_skip, float alpha, const float* batch_scale, int accumulate,
                              const devit_epilogue* ep, void* stream) {
  DEVIT_CHECK(A && B && ep && ep->out && M > 0 && N > 0 && K > 0 && batch >= 1 && batch_inner >= 1 && batch % batch_inner == 0,
              DEVIT_ERR_ARG, "devit_gemm_f32: bad argument");
  SgemmArgs g{A, sam, sak, a_bs_outer, a_bs_inner, B, sbn, sbk, b_bs_outer, b_bs_inner, c_bs_outer, c_bs_inner,
              M, N, K, batch_inner, k_group, k_skip, alpha, batch_scale, accumulate, *ep};
  dim3 grid((N + 63) / 64, (M + 63) / 64, batch), block(256);
  hipStream_t s = (hipStream_t)stream;
  switch (ep->kind) {
    case DEVIT_EPI_STORE_F32: hipLaunchKernelGGL(sgemm_kernel<DEVIT_EPI_STORE_F32>, grid, block, 0, s, g); break;
    case DEVIT_EPI_GELU_BF16: hipLaunchKernelGGL(sgemm_kernel<DEVIT_EPI_GELU_BF16>, grid, block, 0, s, g); break;
    case DEVIT_EPI_DGELU_BF16:
      DEVIT_CHECK(ep->aux_in, DEVIT_ERR_ARG, "devit_gemm_f32: DGELU needs aux_in");
      hipLaunchKernelGGL(sgemm_kernel<DEVIT_EPI_DGELU_BF16>, grid, block, 0, s, g); break;
    case DEVIT_EPI_RESIDUAL_F32:
      DEVIT_CHECK(ep->res, DEVIT_ERR_ARG, "devit_gemm_f32: RESIDUAL needs res");
      hipLaunchKernelGGL(sgemm_kernel<DEVIT_EPI_RESIDUAL_F32>, grid, block, 0, s, g); break;
    case DEVIT_EPI_PATCH_F32:
      DEVIT_CHECK(ep->pos && ep->patch_tokens > 0, DEVIT_ERR_ARG, "devit_gemm_f32: PATCH needs pos");
      hipLaunchKernelGGL(sgemm_kernel<DEVIT_EPI_PATCH_F32>, grid, block, 0, s, g); break;
    default: DEVIT_CHECK(false, DEVIT_ERR_ARG, "devit_gemm_f32: epilogue %d not available in fp32", ep->kind);
  }
  DEVIT_LAUNCH_CHECK();
  return DEVIT_OK;
}

extern "C" int devit_softmax_rows_f32(float* S, int rows, int ncols, int ld, float scale, float* lse, void* stream) {
  DEVIT_CHECK(S && rows > 0 && ncols > 0 && ld >= ncols, DEVIT_ERR_ARG, "devit_softmax_rows_f32: bad argument");
  hipLaunchKernelGGL(softmax_rows_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, S, rows, ncols, ld, scale, lse);
  DEVIT_LAUNCH_CHECK();
  return DEVIT_OK;
}

extern "C" int devit_softmax_bwd_rows_f32(const float* P, float* dP, int rows, int ncols, int ld, float scale, void* stream) {
  DEVIT_CHECK(P && dP && rows > 0 && ncols > 0 && ld >= ncols, DEVIT_ERR_ARG, "devit_softmax_bwd_rows_f32: bad argument");
  hipLaunchKernelGGL(softmax_bwd_rows_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, P, dP, rows, ncols, ld, scale);
  DEVIT_LAUNCH_CHECK();
  return DEVIT_OK;
}

extern "C" int devit_im2row_f32(const float* img, float* rows, int B, void* stream) {
  DEVIT_CHECK(img && rows && B > 0, DEVIT_ERR_ARG, "devit_im2row_f32: bad argument");
  hipLaunchKernelGGL(im2row_f32_kernel, dim3(2048), dim3(256), 0, (hipStream_t)stream, img, rows, B);
  DEVIT_LAUNCH_CHECK();
  return DEVIT_OK;
}

extern "C" int devit_scale_rows_f32(const float* src, float* dst, const float* rowscale, int rows_per_scale, int M, int D,
                                    void* stream) {
  DEVIT_CHECK(src && dst && (!rowscale || rows_per_scale > 0), DEVIT_ERR_ARG, "devit_scale_rows_f32: bad argument");
  hipLaunchKernelGGL(scale_rows_f32_kernel, dim3(2048), dim3(256), 0, (hipStream_t)stream, src, dst, rowscale, rows_per_scale, M, D);
  DEVIT_LAUNCH_CHECK();
  return DEVIT_OK;
}

extern "C" int devit_colsum_f32(const float* y, int M, int N, int ld, float* out, int accumulate, void* stream) {
  DEVIT_CHECK(y && out && M > 0 && N > 0, DEVIT_ERR_ARG, "devit_colsum_f32: bad argument");
  hipLaunchKernelGGL(colsum_f32_kernel, dim3((N + 255) / 256), dim3(256), 0, (hipStream_t)stream, y, M, N, ld, out, accumulate);
  DEVIT_LAUNCH_CHECK();
  return DEVIT_OK;
}
