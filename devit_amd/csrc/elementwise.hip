// HBM-bound helper kernels of the DeViT path: patch im2row, token assembly, casts, column sums
// (bias gradients), embedding backward, small strided f32 GEMM (classifier heads), fused AdamW+EMA.
#include "devit_common.h"

namespace {

// ---- im2row: image f32 [B,3,224,224] -> bf16 rows [B*196][768], k = c*256 + kh*16 + kw -------------
// (timm PatchEmbed Conv2d(3,D,16,16) as a GEMM operand; models/de_vit.py:166-168,258)
__global__ __launch_bounds__(256) void im2row_kernel(const float* img, __bf16* rows, int B, int f16) {
  const int total = B * 196 * 96;  // 16-byte (8-element) output chunks
  for (int idx = blockIdx.x * 256 + threadIdx.x; idx < total; idx += gridDim.x * 256) {
    const int k8 = idx % 96, row = idx / 96;
    const int b = row / 196, t = row % 196, py = t / 14, px = t % 14;
    const int c = k8 >> 5, kh = (k8 >> 1) & 15, kw0 = (k8 & 1) * 8;
    const float* src = img + (((size_t)b * 3 + c) * 224 + py * 16 + kh) * 224 + px * 16 + kw0;
    const f32x4 v0 = *(const f32x4*)src, v1 = *(const f32x4*)(src + 4);
    *(bf16x8*)(rows + (size_t)idx * 8) = f16 ? cvt8<true>(v0, v1) : cvt8<false>(v0, v1);
  }
}

// ---- Mixup / CutMix fused into im2row (engine.py:65-66 -> timm Mixup(mode='batch') -> patch_embed, de_vit.py:258) ---
// rows[b] = patches of   mode 1: lam * img[b] + (1 - lam) * img[B-1-b]        (x.mul_(lam).add_(x.flip(0) * (1 - lam)))
//                        mode 2: img[b] with the box [y0,y1) x [x0,x1) taken from img[B-1-b]   (x[:, :, yl:yh, xl:xh] = ...)
//                        mode 0: img[b]
// One read of the fp32 batch (each image twice), bf16 patch rows out; the mixed fp32 batch never exists.  The products
// and the sum are rounded separately (no FMA contraction), as the reference's tensor ops round them.
struct MixArgs {
  const float* img;
  __bf16* rows;         // bf16 patch rows (student) or NULL
  __bf16* rows_f16;     // the same values as IEEE f16 (f16 teacher) or NULL
  int B, mode, y0, y1, x0, x1;
  float lam, oml;      // f32(lam), f32(1 - lam) with the subtraction done in double on the host, as torch does for a python scalar
};
__global__ __launch_bounds__(256) void mix_im2row_kernel(const MixArgs a) {
#pragma clang fp contract(off)   // x * lam + flip * (1 - lam) as three rounded operations (hipcc would fuse a multiply-add)
  const int total = a.B * 196 * 96;
  const float lam = a.lam, oml = a.oml;
  for (int idx = blockIdx.x * 256 + threadIdx.x; idx < total; idx += gridDim.x * 256) {
    const int k8 = idx % 96, row = idx / 96;
    const int b = row / 196, t = row % 196, py = t / 14, px = t % 14;
    const int c = k8 >> 5, kh = (k8 >> 1) & 15, kw0 = (k8 & 1) * 8;
    const int y = py * 16 + kh, x = px * 16 + kw0;
    const size_t off = ((size_t)c * 224 + y) * 224 + x;
    const float* src = a.img + (size_t)b * 3 * 224 * 224 + off;
    const float* flp = a.img + (size_t)(a.B - 1 - b) * 3 * 224 * 224 + off;
    float v[8];
    *(f32x4*)v = *(const f32x4*)src;
    *(f32x4*)(v + 4) = *(const f32x4*)(src + 4);
    if (a.mode == 1) {
      float w[8];
      *(f32x4*)w = *(const f32x4*)flp;
      *(f32x4*)(w + 4) = *(const f32x4*)(flp + 4);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float p0 = v[e] * lam, p1 = w[e] * oml;
        v[e] = p0 + p1;
      }
    } else if (a.mode == 2 && y >= a.y0 && y < a.y1 && x + 8 > a.x0 && x < a.x1) {
      float w[8];
      *(f32x4*)w = *(const f32x4*)flp;
      *(f32x4*)(w + 4) = *(const f32x4*)(flp + 4);
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = (x + e >= a.x0 && x + e < a.x1) ? w[e] : v[e];
    }
    const f32x4 lo = {v[0], v[1], v[2], v[3]}, hi = {v[4], v[5], v[6], v[7]};
    if (a.rows) *(bf16x8*)(a.rows + (size_t)idx * 8) = cvt8<false>(lo, hi);
    if (a.rows_f16) *(bf16x8*)(a.rows_f16 + (size_t)idx * 8) = cvt8<true>(lo, hi);
  }
}

// targets[b][c] = lam * smooth(y[b])[c] + (1 - lam) * smooth(y[B-1-b])[c],  smooth(y)[c] = eps / C + (c == y) * (1 - eps)
// (timm mixup_target / one_hot; distill_sub.py:315-318)
__global__ __launch_bounds__(256) void mix_targets_kernel(const long long* y, float* out, int B, int C, float lam, float oml,
                                                          float off, float on) {
#pragma clang fp contract(off)
  const int total = B * C;
  for (int idx = blockIdx.x * 256 + threadIdx.x; idx < total; idx += gridDim.x * 256) {
    const int b = idx / C, c = idx % C;
    const float t1 = (int)y[b] == c ? on : off, t2 = (int)y[B - 1 - b] == c ? on : off;
    const float p0 = t1 * lam, p1 = t2 * oml;
    out[idx] = p0 + p1;
  }
}

// ---- x[b, t] = token_t + pos[t] for the extra (cls / dist) tokens: models/de_vit.py:259-264 ----------
__global__ __launch_bounds__(256) void embed_tokens_kernel(const float* cls, const float* dist, const float* pos,
                                                           float* x, int B, int T, int D, int ntok) {
  const int total = B * ntok * D;
  for (int idx = blockIdx.x * 256 + threadIdx.x; idx < total; idx += gridDim.x * 256) {
    const int d = idx % D, t = (idx / D) % ntok, b = idx / (D * ntok);
    const float tok = t == 0 ? cls[d] : dist[d];
    x[((size_t)b * T + t) * D + d] = tok + pos[(size_t)t * D + d];
  }
}

// ---- f32 -> bf16 cast (weights, once per optimizer step) ----------------------------------------------
__global__ __launch_bounds__(256) void cast_bf16_kernel(const float* src, __bf16* dst, size_t n, int f16) {
  const size_t n8 = n / 8;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (size_t)gridDim.x * 256) {
    const f32x4 a = *(const f32x4*)(src + i * 8), b = *(const f32x4*)(src + i * 8 + 4);
    *(bf16x8*)(dst + i * 8) = f16 ? cvt8<true>(a, b) : cvt8<false>(a, b);
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 7)) {
    const float v = src[n8 * 8 + threadIdx.x];
    if (f16) ((_Float16*)dst)[n8 * 8 + threadIdx.x] = (_Float16)v;
    else dst[n8 * 8 + threadIdx.x] = f2bf(v);
  }
}

// ---- f32 [M][D] -> bf16 with an optional per-sample row scale (DropPath) -------------------------------
__global__ __launch_bounds__(256) void scale_cast_kernel(const float* src, __bf16* dst, const float* rowscale,
                                                         int rows_per_scale, int M, int D) {
  const size_t total = (size_t)M * D / 4;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int row = (int)(i * 4 / D);
    const float sc = rowscale ? rowscale[row / rows_per_scale] : 1.0f;
    const f32x4 v = *(const f32x4*)(src + i * 4) * sc;
    bf16x4 o = {f2bf(v[0]), f2bf(v[1]), f2bf(v[2]), f2bf(v[3])};
    *(bf16x4*)(dst + i * 4) = o;
  }
}

// ---- column sums of a bf16 [M][ld] matrix (bias gradients): partial[chunk][N] -------------------------
__global__ __launch_bounds__(256) void colsum_bf16_kernel(const __bf16* y, int M, int N, int ld, int row_group,
                                                          int row_skip, float* partial) {
  __shared__ float red[8][256];
  const int cg = threadIdx.x & 31, rl = threadIdx.x >> 5;
  const int col = (blockIdx.x * 32 + cg) * 8;
  const int chunk = blockIdx.y, nchunk = gridDim.y;
  const int r0 = (int)((long long)M * chunk / nchunk), r1 = (int)((long long)M * (chunk + 1) / nchunk);
  float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (col < N) {
    auto prow = [&](int r) { return row_group > 0 ? (size_t)r + row_skip * (r / row_group + 1) : (size_t)r; };
    int r = r0 + rl;
    for (; r + 24 < r1; r += 32) {      // four independent 16-byte loads in flight per thread
      bf16x8 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = load_stream((const bf16x8*)(y + prow(r + 8 * u) * ld + col));
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] += bf2f(v[u][e]);
    }
    for (; r < r1; r += 8) {
      const bf16x8 v = *(const bf16x8*)(y + prow(r) * ld + col);
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] += bf2f(v[e]);
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) red[rl][cg * 8 + e] = acc[e];
  __syncthreads();
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c < N) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += red[i][threadIdx.x];
    partial[(size_t)chunk * N + c] = s;
  }
}

__global__ __launch_bounds__(256) void sum_partials_kernel(const float* partial, int nparts, int ncols, float* out,
                                                           int accumulate) {
  __shared__ float red[8][32];
  const int cl = threadIdx.x & 31, grp = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + cl;
  float s = 0.f;
  if (c < ncols)
    for (int p = grp; p < nparts; p += 8) s += partial[(size_t)p * ncols + c];
  red[grp][cl] = s;
  __syncthreads();
  if (grp == 0 && c < ncols) {
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) t += red[i][cl];
    out[c] = accumulate ? out[c] + t : t;
  }
}

// ---- embedding backward: dpos[t][d] = sum_b dx[b][t][d]; bf16 copy of dx for the patch wgrad ------------
// gridDim.y slices of the batch per (t, d4) column, four independent 16-byte loads in flight, partial sums combined by
// fp32 atomics into a zeroed / accumulating dpos (8 adds per address).  (One thread per column walking all 256 images
// with one load in flight: 74 workgroups, 1.1 TB/s.)
__global__ __launch_bounds__(256) void embed_bwd_kernel(const float* dx, int B, int T, int D, float* dpos,
                                                        __bf16* dx_bf16) {
  const int total = T * D / 4;
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int d = (idx % (D / 4)) * 4, t = idx / (D / 4);
  const int b0 = (int)((long long)B * blockIdx.y / gridDim.y), b1 = (int)((long long)B * (blockIdx.y + 1) / gridDim.y);
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  auto one = [&](int b, f32x4& v) { v = load_stream((const f32x4*)(dx + ((size_t)b * T + t) * D + d)); };
  auto put = [&](int b, const f32x4& v) {
    s += v;
    if (dx_bf16) {
      const bf16x4 ob = {f2bf(v[0]), f2bf(v[1]), f2bf(v[2]), f2bf(v[3])};
      *(bf16x4*)(dx_bf16 + ((size_t)b * T + t) * D + d) = ob;
    }
  };
  int b = b0;
  for (; b + 4 <= b1; b += 4) {
    f32x4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) one(b + u, v[u]);
#pragma unroll
    for (int u = 0; u < 4; ++u) put(b + u, v[u]);
  }
  for (; b < b1; ++b) {
    f32x4 v;
    one(b, v);
    put(b, v);
  }
  float* dst = dpos + (size_t)t * D + d;
#pragma unroll
  for (int e = 0; e < 4; ++e) unsafeAtomicAdd(dst + e, s[e]);
}
// dcls / ddist / dbias from dpos (tiny)
__global__ __launch_bounds__(256) void embed_bwd_tail_kernel(const float* dpos, int T, int D, int ntok, float* dcls,
                                                             float* ddist, float* dbias, int accumulate) {
  const int d = blockIdx.x * 256 + threadIdx.x;
  if (d >= D) return;
  float s = 0.f;
  for (int t = ntok; t < T; ++t) s += dpos[(size_t)t * D + d];
  // NOTE: dpos already holds the accumulated value when accumulate != 0, so the tail always overwrites
  dbias[d] = s;
  dcls[d] = dpos[d];
  if (ddist) ddist[d] = dpos[(size_t)D + d];
  (void)accumulate;
}

// ---- small strided f32 GEMM: C[m][n] = sum_k A[m*sam + k*sak] * B[n*sbn + k*sbk] (+ bias[n]) ------------
// classifier heads (models/de_vit.py:317) and their backward.  Eight lanes per output element, lane p taking
// k = p, p + 8, ... (adjacent lanes read adjacent k: 32-byte pieces of both operand rows), eight independent loads in
// flight per operand and trip, fp32 fmaf partial sums combined by three xor-shuffles.  (One thread per output with a
// serial K loop was a 48 us latency chain at 6400 outputs x K = 384: 25 workgroups on a 256-CU device.)
__global__ __launch_bounds__(256) void sgemm_small_kernel(const float* A, long long sam, long long sak, const float* B,
                                                          long long sbn, long long sbk, const float* bias, float* C,
                                                          int ldc, int M, int N, int K, float alpha, int accumulate) {
  const long long total = (long long)M * N;
  const int part = threadIdx.x & 7;
  for (long long idx = ((long long)blockIdx.x * 256 + threadIdx.x) >> 3; idx < ((total + 31) & ~31ll);
       idx += ((long long)gridDim.x * 256) >> 3) {
    const bool live = idx < total;                       // whole waves stay in the loop: the shuffles need every lane
    const long long id = live ? idx : total - 1;
    const int n = (int)(id % N), m = (int)(id / N);
    const float* a = A + m * sam;
    const float* b = B + n * sbn;
    float s = 0.f;
    int k = part;
    for (; k + 56 < K; k += 64) {
      float av[8], bv[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) { av[u] = a[(k + 8 * u) * sak]; bv[u] = b[(k + 8 * u) * sbk]; }
#pragma unroll
      for (int u = 0; u < 8; ++u) s = fmaf(av[u], bv[u], s);
    }
    for (; k < K; k += 8) s = fmaf(a[k * sak], b[k * sbk], s);
    s += __shfl_xor(s, 1, 64);
    s += __shfl_xor(s, 2, 64);
    s += __shfl_xor(s, 4, 64);
    if (live && part == 0) {
      s = s * alpha + (bias ? bias[n] : 0.f);
      float* c = C + (size_t)m * ldc + n;
      *c = accumulate ? *c + s : s;
    }
  }
}

// ---- fused AdamW (+ global-norm clip) + EMA + bf16 re-cast over a flat parameter buffer -----------------
// torch.optim.AdamW + timm NativeScaler clip_grad_norm_ + timm ModelEma.update (engine.py:127,131-132)
struct AdamArgs {
  float* p; const float* g; float* m; float* v; float* ema; __bf16* p_bf16;
  const float* gnorm_sq;  // device scalar: sum of squared grads (NULL = no clipping)
  const float* dyn;       // device [3]: lr, 1 - beta1^step, 1 - beta2^step (changes every step; graph-safe)
  const unsigned char* no_decay4;   // one byte per 4-element granule: != 0 -> no weight decay there (NULL: decay all)
  size_t n;
  float beta1, beta2, eps, wd, max_norm, ema_decay, grad_scale;
};
__global__ __launch_bounds__(256) void adamw_kernel(const AdamArgs a) {
  float clip = a.grad_scale;
  const float lr = a.dyn[0], bc1 = a.dyn[1], bc2 = a.dyn[2];
  if (a.gnorm_sq) {
    const float nrm = sqrtf(*a.gnorm_sq) * a.grad_scale;
    const float c = a.max_norm / (nrm + 1e-6f);   // torch.nn.utils.clip_grad_norm_
    clip *= c < 1.0f ? c : 1.0f;
  }
  const size_t n4 = a.n / 4;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    f32x4 p = *(const f32x4*)(a.p + i * 4);
    const f32x4 g = *(const f32x4*)(a.g + i * 4) * clip;
    f32x4 m = *(const f32x4*)(a.m + i * 4), v = *(const f32x4*)(a.v + i * 4);
    if (!(a.no_decay4 && a.no_decay4[i])) p *= (1.0f - lr * a.wd);
    m = a.beta1 * m + (1.0f - a.beta1) * g;
    v = a.beta2 * v + (1.0f - a.beta2) * g * g;
#pragma unroll
    for (int e = 0; e < 4; ++e) p[e] -= (lr / bc1) * m[e] / (sqrtf(v[e]) / sqrtf(bc2) + a.eps);
    *(f32x4*)(a.p + i * 4) = p;
    *(f32x4*)(a.m + i * 4) = m;
    *(f32x4*)(a.v + i * 4) = v;
    if (a.ema) {
      f32x4 e = *(const f32x4*)(a.ema + i * 4);
      e = e * a.ema_decay + (1.0f - a.ema_decay) * p;
      *(f32x4*)(a.ema + i * 4) = e;
    }
    if (a.p_bf16) {
      bf16x4 ob = {f2bf(p[0]), f2bf(p[1]), f2bf(p[2]), f2bf(p[3])};
      *(bf16x4*)(a.p_bf16 + i * 4) = ob;
    }
  }
}

// sum of squares of a flat f32 buffer -> out[0] (two-stage, deterministic)
__global__ __launch_bounds__(256) void sumsq_stage1(const float* g, size_t n, float* partial) {
  __shared__ float red[4];
  float s = 0.f;
  const size_t n4 = n / 4;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    const f32x4 v = *(const f32x4*)(g + i * 4);
    s += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}
__global__ __launch_bounds__(256) void sumsq_stage2(const float* partial, int n, float* out) {
  __shared__ float red[4];
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) s += partial[i];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) out[0] = red[0] + red[1] + red[2] + red[3];
}

inline int grid_for(size_t work_items, int cap = 2048) {
  size_t g = (work_items + 255) / 256;
  return (int)(g < 1 ? 1 : (g > (size_t)cap ? cap : g));
}

}  // namespace

extern "C" int devit_im2row_bf16(const float* img, void* rows, int B, int C, int H, int W, int patch, int dtype16,
                                 void* stream) {
  DEVIT_CHECK(img && rows && (dtype16 == 0 || dtype16 == 1), DEVIT_ERR_ARG, "devit_im2row_bf16: bad argument");
  DEVIT_CHECK(C == 3 && H == 224 && W == 224 && patch == 16 && B > 0, DEVIT_ERR_SHAPE,
              "devit_im2row_bf16: only 3x224x224 / patch 16 (got %dx%dx%d / %d)", C, H, W, patch);
  hipLaunchKernelGGL(im2row_kernel, dim3(grid_for((size_t)B * 196 * 96, 4096)), dim3(256), 0, (hipStream_t)stream, img,
                     (__bf16*)rows, B, dtype16);
  DEVIT_LAUNCH_CHECK();
  return DEVIT_OK;
}

extern "C" int devit_mix_im2row_bf16(const float* img, void* rows, void* rows_f16, int B, int mode, double lam, int y0,
                                     int y1, int x0, int x1, void* stream) {
  DEVIT_CHECK(img && (rows || rows_f16) && B > 0, DEVIT_ERR_ARG, "devit_mix_im2row_bf16: bad argument");
  DEVIT_CHECK(mode >= 0 && mode <= 2, DEVIT_ERR_ARG, "devit_mix_im2row_bf16: mode %d (0 none, 1 mixup, 2 cutmix)", mode);
  DEVIT_CHECK(mode != 2 || (0 <= y0 && y0 <= y1 && y1 <= 224 && 0 <= x0 && x0 <= x1 && x1 <= 224), DEVIT_ERR_ARG,
              "devit_mix_im2row_bf16: box [%d,%d) x [%d,%d) outside 224x224", y0, y1, x0, x1);
  MixArgs a{img, (__bf16*)rows, (__bf16*)rows_f16, B, mode, y0, y1, x0, x1, (float)lam, (float)(1.0 - lam)};
  hipLaunchKernelGGL(mix_im2row_kernel, dim3(grid_for((size_t)B * 196 * 96, 4096)), dim3(256), 0, (hipStream_t)stream, a);
  DEVIT_LAUNCH_CHECK();
  return DEVIT_OK;
}

extern "C" int devit_mix_targets(const long long* labels, float* targets, int B, int C, double lam, double smoothing,
                                 void* stream) {
  DEVIT_CHECK(labels && targets && B > 0 && C > 0, DEVIT_ERR_ARG, "devit_mix_targets: bad argument");
  const double off = smoothing / C, on = 1.0 - smoothing + off;      // timm mixup_target: off / on values in double
  hipLaunchKernelGGL(mix_targets_kernel, dim3(grid_for((size_t)B * C)), dim3(256), 0, (hipStream_t)stream, labels, targets, B,
                     C, (float)lam, (float)(1.0 - lam), (float)off, (float)on);
  DEVIT_LAUNCH_CHECK();
  return DEVIT_OK;
}

extern "C" int devit_embed_tokens(const float* cls, const float* dist, const float* pos, float* x, int B, int T, int D,
                                  void* stream) {
  DEVIT_CHECK(cls && pos && x, DEVIT_ERR_ARG, "devit_embed_tokens: null pointer");
  const int ntok = dist ? 2 : 1;
  hipLaunchKernelGGL(embed_tokens_kernel, dim3(grid_for((size_t)B * ntok * D)), dim3(256), 0, (hipStream_t)stream, cls,
                     dist, pos, x, B, T, D, ntok);
  DEVIT_LAUNCH_CHECK();
  return DEVIT_OK;
}

extern "C" int devit_cast_bf16(const float* src, void* dst, size_t n, int dtype16, void* stream) {
  DEVIT_CHECK(src && dst && (dtype16 == 0 || dtype16 == 1), DEVIT_ERR_ARG, "devit_cast_bf16: bad argument");
  DEVIT_CHECK((((uintptr_t)src) & 15) == 0 && (((uintptr_t)dst) & 15) == 0, DEVIT_ERR_ARG, "devit_cast_bf16: alignment");
  hipLaunchKernelGGL(cast_bf16_kernel, dim3(grid_for(n / 8 + 1)), dim3(256), 0, (hipStream_t)stream, src, (__bf16*)dst, n, dtype16);
  DEVIT_LAUNCH_CHECK();
  return DEVIT_OK;
}

extern "C" int devit_scale_cast_bf16(const float* src, void* dst, const float* rowscale, int rows_per_scale, int M,
                                     int D, void* stream) {
  DEVIT_CHECK(src && dst && (!rowscale || rows_per_scale > 0), DEVIT_ERR_ARG, "devit_scale_cast_bf16: bad argument");
  DEVIT_CHECK(D % 4 == 0 && M > 0, DEVIT_ERR_SHAPE, "devit_scale_cast_bf16: D %% 4");
  hipLaunchKernelGGL(scale_cast_kernel, dim3(grid_for((size_t)M * D / 4, 4096)), dim3(256), 0, (hipStream_t)stream, src,
                     (__bf16*)dst, rowscale, rows_per_scale, M, D);
  DEVIT_LAUNCH_CHECK();
  return DEVIT_OK;
}

extern "C" size_t devit_colsum_workspace(int M, int N) {
  (void)M;
  return (size_t)64 * N * sizeof(float);
}

extern "C" int devit_colsum_bf16(const void* y, int M, int N, int ld, int row_group, int row_skip, float* out,
                                 int accumulate, void* workspace, size_t workspace_bytes, void* stream) {
  DEVIT_CHECK(y && out && workspace, DEVIT_ERR_ARG, "devit_colsum_bf16: null pointer");
  DEVIT_CHECK(N % 8 == 0 && ld % 8 == 0 && M > 0, DEVIT_ERR_SHAPE, "devit_colsum_bf16: N, ld must be multiples of 8");
  const int nchunk = M >= 64 * 8 ? 64 : 1;
  DEVIT_CHECK(workspace_bytes >= (size_t)nchunk * N * sizeof(float), DEVIT_ERR_ARG, "devit_colsum_bf16: workspace too small");
  hipLaunchKernelGGL(colsum_bf16_kernel, dim3((N + 255) / 256, nchunk), dim3(256), 0, (hipStream_t)stream,
                     (const __bf16*)y, M, N, ld, row_group, row_skip, (float*)workspace);
  DEVIT_LAUNCH_CHECK();
  hipLaunchKernelGGL(sum_partials_kernel, dim3((N + 31) / 32), dim3(256), 0, (hipStream_t)stream,
                     (const float*)workspace, nchunk, N, out, accumulate);
  DEVIT_LAUNCH_CHECK();
  return DEVIT_OK;
}

extern "C" int devit_embed_bwd(const float* dx, int B, int T, int D, int ntok, float* dpos, float* dcls, float* ddist,
                               float* dbias, void* dx_bf16, int accumulate, void* stream) {
  DEVIT_CHECK(dx && dpos && dcls && dbias, DEVIT_ERR_ARG, "devit_embed_bwd: null pointer");
  DEVIT_CHECK(D % 4 == 0 && (ntok == 1 || (ntok == 2 && ddist)), DEVIT_ERR_SHAPE, "devit_embed_bwd: D %% 4, ntok");
  DEVIT_CHECK(accumulate == 0, DEVIT_ERR_ARG, "devit_embed_bwd: accumulate is not supported");
  hipError_t me = hipMemsetAsync(dpos, 0, (size_t)T * D * sizeof(float), (hipStream_t)stream);
  DEVIT_CHECK(me == hipSuccess, DEVIT_ERR_LAUNCH, "devit_embed_bwd: hipMemsetAsync: %s", hipGetErrorString(me));
  const int slices = B >= 64 ? 8 : 1;
  hipLaunchKernelGGL(embed_bwd_kernel, dim3((T * D / 4 + 255) / 256, slices), dim3(256), 0, (hipStream_t)stream, dx, B,
                     T, D, dpos, (__bf16*)dx_bf16);
  DEVIT_LAUNCH_CHECK();
  hipLaunchKernelGGL(embed_bwd_tail_kernel, dim3((D + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                     (const float*)dpos, T, D, ntok, dcls, ddist, dbias, accumulate);
  DEVIT_LAUNCH_CHECK();
  return DEVIT_OK;
}

extern "C" int devit_sgemm_small(const float* A, long long sam, long long sak, const float* B, long long sbn,
                                 long long sbk, const float* bias, float* C, int ldc, int M, int N, int K, float alpha,
                                 int accumulate, void* stream) {
  DEVIT_CHECK(A && B && C && M > 0 && N > 0 && K > 0, DEVIT_ERR_ARG, "devit_sgemm_small: bad argument");
  hipLaunchKernelGGL(sgemm_small_kernel, dim3(grid_for((size_t)M * N * 8, 8192)), dim3(256), 0, (hipStream_t)stream, A, sam,
                     sak, B, sbn, sbk, bias, C, ldc, M, N, K, alpha, accumulate);
  DEVIT_LAUNCH_CHECK();
  return DEVIT_OK;
}

extern "C" size_t devit_sumsq_workspace(void) { return 1024 * sizeof(float); }

extern "C" int devit_sumsq_f32(const float* g, size_t n, float* out, void* workspace, size_t workspace_bytes,
                               void* stream) {
  DEVIT_CHECK(g && out && workspace && workspace_bytes >= 1024 * sizeof(float), DEVIT_ERR_ARG, "devit_sumsq_f32: bad argument");
  DEVIT_CHECK(n % 4 == 0, DEVIT_ERR_SHAPE, "devit_sumsq_f32: n must be a multiple of 4 (pad the flat buffer)");
  hipLaunchKernelGGL(sumsq_stage1, dim3(1024), dim3(256), 0, (hipStream_t)stream, g, n, (float*)workspace);
  DEVIT_LAUNCH_CHECK();
  hipLaunchKernelGGL(sumsq_stage2, dim3(1), dim3(256), 0, (hipStream_t)stream, (const float*)workspace, 1024, out);
  DEVIT_LAUNCH_CHECK();
  return DEVIT_OK;
}

extern "C" int devit_adamw_step(float* p, const float* g, float* m, float* v, float* ema, void* p_bf16,
                                const unsigned char* no_decay4, const float* gnorm_sq, const float* dyn, size_t n,
                                float beta1, float beta2, float eps,
                                float weight_decay, float max_norm, float ema_decay, float grad_scale,
                                void* stream) {
  DEVIT_CHECK(p && g && m && v && dyn, DEVIT_ERR_ARG, "devit_adamw_step: bad argument");
  DEVIT_CHECK(n % 4 == 0, DEVIT_ERR_SHAPE, "devit_adamw_step: n must be a multiple of 4 (pad the flat buffer)");
  AdamArgs a;
  a.p = p; a.g = g; a.m = m; a.v = v; a.ema = ema; a.p_bf16 = (__bf16*)p_bf16; a.gnorm_sq = gnorm_sq; a.n = n;
  a.no_decay4 = no_decay4;
  a.dyn = dyn; a.beta1 = beta1; a.beta2 = beta2; a.eps = eps; a.wd = weight_decay;
  a.max_norm = max_norm; a.ema_decay = ema_decay; a.grad_scale = grad_scale;
  hipLaunchKernelGGL(adamw_kernel, dim3(grid_for(n / 4, 4096)), dim3(256), 0, (hipStream_t)stream, a);
  DEVIT_LAUNCH_CHECK();
  return DEVIT_OK;
}
