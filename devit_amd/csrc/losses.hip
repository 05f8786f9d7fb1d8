// DEKD loss kernels: fused logit-distillation loss (+ its gradient) and the row-softmax / KL part of
// the q/k/v feature-relation loss (the Gram products run on the MFMA GEMM, see relation_* in ops.py).
//   cls loss   : utils/losses.py:135-177 (DistillLoss) + timm SoftTargetCrossEntropy
//   relation   : utils/losses.py:307-328 (feature_relation_loss), closed form in SURVEY.md App. A
#include "devit_common.h"

namespace {

constexpr int MAXC_PER_LANE = 16;  // classes <= 1024

struct ClsArgs {
  const float* lo;      // student cls-head logits [B][C]
  const float* lk;      // student dist-head logits [B][C]
  const float* lt;      // teacher logits [B][C]
  const float* y;       // soft targets [B][C]
  float* loss;          // [3]: total, base, distill
  float* dlo;           // [B][C] d total / d lo
  float* dlk;
  int B, C, kind;       // kind: 0 none, 1 soft, 2 hard
  float alpha, tau;
};

__device__ __forceinline__ void row_lse(const float (&x)[MAXC_PER_LANE], int n, float inv_t, float& mx, float& lse) {
  mx = -INFINITY;
#pragma unroll
  for (int i = 0; i < MAXC_PER_LANE; ++i)
    if (i < n) mx = fmaxf(mx, x[i] * inv_t);
  mx = wave_max(mx);
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < MAXC_PER_LANE; ++i)
    if (i < n) s += expf(x[i] * inv_t - mx);
  s = wave_sum(s);
  lse = mx + logf(s);
}

// One wave per row, 16 rows per workgroup, B / 16 workgroups; each workgroup adds its weighted share to the three loss
// scalars (zeroed by the launcher) with fp32 atomics -- 16 adds per scalar at B = 256.  (One workgroup walking all rows
// was a 135 us chain of dependent row reductions.)
__global__ __launch_bounds__(1024) void cls_loss_kernel(const ClsArgs a) {
  __shared__ float red[16][2];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int C = a.C;
  float base_acc = 0.f, dist_acc = 0.f;
  for (int b = blockIdx.x * 16 + wv; b < a.B; b += 16 * gridDim.x) {
    float xo[MAXC_PER_LANE], xk[MAXC_PER_LANE], xt[MAXC_PER_LANE], yy[MAXC_PER_LANE];
    int n = 0;
#pragma unroll
    for (int i = 0; i < MAXC_PER_LANE; ++i) {
      const int c = lane + i * 64;
      if (c < C) {
        xo[i] = a.lo[(size_t)b * C + c];
        xk[i] = a.lk[(size_t)b * C + c];
        xt[i] = a.lt ? a.lt[(size_t)b * C + c] : 0.f;
        yy[i] = a.y[(size_t)b * C + c];
        n = i + 1;
      } else {
        xo[i] = xk[i] = xt[i] = -INFINITY;
        yy[i] = 0.f;
      }
    }
    // n differs between lanes only in the last partial group: make the loop bound wave-uniform and
    // rely on the -inf / 0 padding above
    n = (C + 63) / 64;
    float mo, lseo;
    row_lse(xo, n, 1.0f, mo, lseo);
    float ysum = 0.f, bl = 0.f;
#pragma unroll
    for (int i = 0; i < MAXC_PER_LANE; ++i)
      if (i < n && lane + i * 64 < C) {
        ysum += yy[i];
        bl -= yy[i] * (xo[i] - lseo);
      }
    ysum = wave_sum(ysum);
    base_acc += wave_sum(bl);
    const float wbase = (a.kind == 0 ? 1.0f : 1.0f - a.alpha) / (float)a.B;
#pragma unroll
    for (int i = 0; i < MAXC_PER_LANE; ++i) {
      const int c = lane + i * 64;
      if (i < n && c < C) a.dlo[(size_t)b * C + c] = (expf(xo[i] - lseo) * ysum - yy[i]) * wbase;
    }
    if (a.kind == 2) {  // hard: CE(kd, argmax teacher)   utils/losses.py:153
      float best = -INFINITY;
      int bi = 0x7fffffff;
#pragma unroll
      for (int i = 0; i < MAXC_PER_LANE; ++i) {
        const int c = lane + i * 64;
        if (i < n && c < C && xt[i] > best) { best = xt[i]; bi = c; }
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {   // max value, ties -> lowest index (torch.argmax on CPU)
        const float ob = __shfl_xor(best, o, 64);
        const int oi = __shfl_xor(bi, o, 64);
        if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
      }
      float mk, lsek;
      row_lse(xk, n, 1.0f, mk, lsek);
      float xsel = 0.f;
#pragma unroll
      for (int i = 0; i < MAXC_PER_LANE; ++i)
        if (i < n && lane + i * 64 == bi) xsel = xk[i];
      xsel = wave_sum(xsel);
      dist_acc += lsek - xsel;
      const float wd = a.alpha / (float)a.B;
#pragma unroll
      for (int i = 0; i < MAXC_PER_LANE; ++i) {
        const int c = lane + i * 64;
        if (i < n && c < C) a.dlk[(size_t)b * C + c] = (expf(xk[i] - lsek) - (c == bi ? 1.f : 0.f)) * wd;
      }
    } else if (a.kind == 1) {  // soft: KL(log_softmax(kd/T) || log_softmax(tea/T)) T^2 / numel   :140-148
      const float it = 1.0f / a.tau;
      float mk, lsek, mt, lset;
      row_lse(xk, n, it, mk, lsek);
      row_lse(xt, n, it, mt, lset);
      float kl = 0.f;
      const float wd = a.alpha * a.tau / ((float)a.B * (float)C);
#pragma unroll
      for (int i = 0; i < MAXC_PER_LANE; ++i) {
        const int c = lane + i * 64;
        if (i < n && c < C) {
          const float la = xk[i] * it - lsek, lb = xt[i] * it - lset;
          kl += expf(lb) * (lb - la);
          a.dlk[(size_t)b * C + c] = (expf(la) - expf(lb)) * wd;
        }
      }
      dist_acc += wave_sum(kl);
    } else {
#pragma unroll
      for (int i = 0; i < MAXC_PER_LANE; ++i) {
        const int c = lane + i * 64;
        if (i < n && c < C) a.dlk[(size_t)b * C + c] = 0.f;
      }
    }
  }
  if (lane == 0) { red[wv][0] = base_acc; red[wv][1] = dist_acc; }
  __syncthreads();
  if (threadIdx.x == 0) {
    float bs = 0.f, ds = 0.f;
    for (int i = 0; i < 16; ++i) { bs += red[i][0]; ds += red[i][1]; }
    bs /= (float)a.B;
    if (a.kind == 2) ds /= (float)a.B;
    else if (a.kind == 1) ds *= a.tau * a.tau / ((float)a.B * (float)C);
    unsafeAtomicAdd(a.loss + 1, bs);
    unsafeAtomicAdd(a.loss + 2, ds);
    unsafeAtomicAdd(a.loss + 0, a.kind == 0 ? bs : bs * (1.0f - a.alpha) + ds * a.alpha);
  }
}

// ---- relation loss, stage 1: per-row log-sum-exp of R/sqrt(hd) for teacher and student Grams, and the
//      row's KL contribution sum_j e^{t_ij}(t_ij - s_ij).  R_* are [B][ldr][ldr] f32 (padded Grams).
struct RelStatArgs {
  const float* rt; const float* rs;
  float* lse_t; float* lse_s;   // [B][N]
  float* row_kl;                // [B][N]
  int B, N, ldr;
  float inv_sqrt_t, inv_sqrt_s;
};
__global__ __launch_bounds__(256) void rel_stats_kernel(const RelStatArgs a) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= a.B * a.N) return;
  const int b = row / a.N, i = row % a.N;
  // one 16-byte load per lane and matrix: columns 4 lane .. 4 lane + 3 of the 256-wide padded Gram row
  const f32x4 vt = lane * 4 < a.ldr ? *(const f32x4*)(a.rt + ((size_t)b * a.ldr + i) * a.ldr + lane * 4) : (f32x4){0.f, 0.f, 0.f, 0.f};
  const f32x4 vs = lane * 4 < a.ldr ? *(const f32x4*)(a.rs + ((size_t)b * a.ldr + i) * a.ldr + lane * 4) : (f32x4){0.f, 0.f, 0.f, 0.f};
  float t[4], s[4];
  float mt = -INFINITY, ms = -INFINITY;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int j = lane * 4 + k;
    t[k] = j < a.N ? vt[k] * a.inv_sqrt_t : -INFINITY;
    s[k] = j < a.N ? vs[k] * a.inv_sqrt_s : -INFINITY;
    mt = fmaxf(mt, t[k]);
    ms = fmaxf(ms, s[k]);
  }
  mt = wave_max(mt);
  ms = wave_max(ms);
  float st = 0.f, ss = 0.f;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    st += expf(t[k] - mt);
    ss += expf(s[k] - ms);
  }
  const float lt = mt + logf(wave_sum(st)), ls = ms + logf(wave_sum(ss));
  float kl = 0.f;
#pragma unroll
  for (int k = 0; k < 4; ++k)
    if (lane * 4 + k < a.N) {
      const float a_t = t[k] - lt, a_s = s[k] - ls;
      kl += expf(a_t) * (a_t - a_s);
    }
  kl = wave_sum(kl);
  if (lane == 0) {
    a.lse_t[row] = lt;
    a.lse_s[row] = ls;
    a.row_kl[row] = kl;
  }
}
// loss = sum(row_kl) / B   (KLDivLoss batchmean, utils/losses.py:309)
// One block of 1024 threads, 8 independent partial sums per thread (50688 values: six 16-byte-strided trips instead
// of 198 dependent ones); fixed summation order, so the result is run-to-run deterministic.
__global__ __launch_bounds__(1024) void rel_reduce_kernel(const float* row_kl, int n, float inv_b, float* loss) {
  __shared__ float red[16];
  float p[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int i0 = threadIdx.x; i0 < n; i0 += 8 * 1024) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = i0 + u * 1024;
      p[u] += i < n ? row_kl[i] : 0.f;
    }
  }
  float s = ((p[0] + p[1]) + (p[2] + p[3])) + ((p[4] + p[5]) + (p[6] + p[7]));
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < 16; ++w) t += red[w];
    loss[0] = t * inv_b;
  }
}
// stage 2 (backward): S = G + G^T with G = (softmax(R_s) - softmax(R_t)) * upstream / (B sqrt(hd_s)); the Gram
// is symmetric, so S_ij = e^{rs_ij}(e^{-ls_i} + e^{-ls_j}) - e^{rt_ij}(e^{-lt_i} + e^{-lt_j}).  bf16 [B][ldr][ldr],
// zero outside N x N; dF_student = S F_student runs on the MFMA GEMM.
struct RelGradArgs {
  const float* rt; const float* rs; const float* lse_t; const float* lse_s;
  const float* upstream;  // device scalar (dL/dloss) or NULL (= 1)
  void* S;
  int B, N, ldr, out_f32;
  float inv_sqrt_t, inv_sqrt_s, coef;  // coef = 1 / (B * sqrt(hd_s))
};
__global__ __launch_bounds__(256) void rel_grad_kernel(const RelGradArgs a) {
  // one wave per Gram row, four columns per lane: 16-byte loads of both Grams, one 8-byte (bf16) or 16-byte store
  const int b = blockIdx.y, i = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int j0 = (threadIdx.x & 63) * 4;  // ldr == 256
  const size_t o = ((size_t)b * a.ldr + i) * a.ldr + j0;
  f32x4 v = {0.f, 0.f, 0.f, 0.f};
  if (i < a.N) {
    const float up = (a.upstream ? *a.upstream : 1.0f) * a.coef;
    const f32x4 rsv = *(const f32x4*)(a.rs + o), rtv = *(const f32x4*)(a.rt + o);
    const float lsi = a.lse_s[b * a.N + i], lti = a.lse_t[b * a.N + i];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int j = j0 + e;
      if (j < a.N) {
        const float rs = rsv[e] * a.inv_sqrt_s, rt = rtv[e] * a.inv_sqrt_t;
        const float lsj = a.lse_s[b * a.N + j], ltj = a.lse_t[b * a.N + j];
        v[e] = (expf(rs - lsi) + expf(rs - lsj) - expf(rt - lti) - expf(rt - ltj)) * up;
      }
    }
  }
  if (a.out_f32) *(f32x4*)((float*)a.S + o) = v;
  else {
    const bf16x4 ob = {f2bf(v[0]), f2bf(v[1]), f2bf(v[2]), f2bf(v[3])};
    *(bf16x4*)((__bf16*)a.S + o) = ob;
  }
}

// ---- token MSE (nn.MSELoss, utils/losses.py:194,228,241-242): loss (+)= mean((a-b)^2), da = 2 (a-b) / n ---------
__global__ __launch_bounds__(1024) void mse_kernel(const float* a, const float* b, size_t n, float* loss, float* da,
                                                   int accumulate) {
  __shared__ float red[16];
  float s = 0.f;
  const float inv = 1.0f / (float)n;
  for (size_t i = threadIdx.x; i < n; i += 1024) {
    const float d = a[i] - b[i];
    s += d * d;
    if (da) da[i] = 2.0f * d * inv;
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
    for (int i = 0; i < 16; ++i) t += red[i];
    loss[0] = (accumulate ? loss[0] : 0.f) + t * inv;
  }
}

}  // namespace

extern "C" int devit_token_mse(const float* a, const float* b, size_t n, float* loss, float* da, int accumulate,
                               void* stream) {
  DEVIT_CHECK(a && b && loss && n > 0, DEVIT_ERR_ARG, "devit_token_mse: bad argument");
  hipLaunchKernelGGL(mse_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, a, b, n, loss, da, accumulate);
  DEVIT_LAUNCH_CHECK();
  return DEVIT_OK;
}

extern "C" int devit_cls_distill_loss(const float* logits, const float* logits_kd, const float* teacher_logits,
                                      const float* soft_targets, int B, int C, int kind, float alpha, float tau,
                                      float* loss3, float* dlogits, float* dlogits_kd, void* stream) {
  DEVIT_CHECK(logits && logits_kd && soft_targets && loss3 && dlogits && dlogits_kd, DEVIT_ERR_ARG,
              "devit_cls_distill_loss: null pointer");
  DEVIT_CHECK(kind == 0 || teacher_logits, DEVIT_ERR_ARG, "devit_cls_distill_loss: teacher logits required");
  DEVIT_CHECK(B > 0 && C > 0 && C <= 64 * MAXC_PER_LANE && kind >= 0 && kind <= 2, DEVIT_ERR_SHAPE,
              "devit_cls_distill_loss: C=%d must be <= 1024", C);
  ClsArgs a{logits, logits_kd, teacher_logits, soft_targets, loss3, dlogits, dlogits_kd, B, C, kind, alpha, tau};
  hipError_t me = hipMemsetAsync(loss3, 0, 3 * sizeof(float), (hipStream_t)stream);
  DEVIT_CHECK(me == hipSuccess, DEVIT_ERR_LAUNCH, "devit_cls_distill_loss: hipMemsetAsync: %s", hipGetErrorString(me));
  hipLaunchKernelGGL(cls_loss_kernel, dim3((B + 15) / 16), dim3(1024), 0, (hipStream_t)stream, a);
  DEVIT_LAUNCH_CHECK();
  return DEVIT_OK;
}

extern "C" int devit_relation_stats(const float* gram_t, const float* gram_s, int B, int N, int ldr, int head_dim_t,
                                    int head_dim_s, float* lse_t, float* lse_s, float* row_kl, float* loss,
                                    void* stream) {
  DEVIT_CHECK(gram_t && gram_s && lse_t && lse_s && row_kl && loss, DEVIT_ERR_ARG, "devit_relation_stats: null pointer");
  DEVIT_CHECK(N > 0 && N <= 256 && ldr >= N && ldr <= 256 && ldr % 4 == 0, DEVIT_ERR_SHAPE,
              "devit_relation_stats: N=%d must be <= ldr=%d <= 256, ldr a multiple of 4", N, ldr);
  RelStatArgs a{gram_t, gram_s, lse_t, lse_s, row_kl, B, N, ldr, 1.0f / sqrtf((float)head_dim_t),
                1.0f / sqrtf((float)head_dim_s)};
  hipLaunchKernelGGL(rel_stats_kernel, dim3((B * N + 3) / 4), dim3(256), 0, (hipStream_t)stream, a);
  DEVIT_LAUNCH_CHECK();
  hipLaunchKernelGGL(rel_reduce_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, (const float*)row_kl, B * N,
                     1.0f / (float)B, loss);
  DEVIT_LAUNCH_CHECK();
  return DEVIT_OK;
}

extern "C" int devit_relation_grad(const float* gram_t, const float* gram_s, const float* lse_t, const float* lse_s,
                                   const float* upstream, int B, int N, int ldr, int head_dim_t, int head_dim_s,
                                   void* S_out, int out_is_f32, void* stream) {
  DEVIT_CHECK(gram_t && gram_s && lse_t && lse_s && S_out, DEVIT_ERR_ARG, "devit_relation_grad: null pointer");
  DEVIT_CHECK(ldr == 256 && N <= 256, DEVIT_ERR_SHAPE, "devit_relation_grad: padded Gram must be 256 wide");
  RelGradArgs a{gram_t, gram_s, lse_t, lse_s, upstream, S_out, B, N, ldr, out_is_f32,
                1.0f / sqrtf((float)head_dim_t), 1.0f / sqrtf((float)head_dim_s),
                1.0f / ((float)B * sqrtf((float)head_dim_s))};
  hipLaunchKernelGGL(rel_grad_kernel, dim3(ldr / 4, B), dim3(256), 0, (hipStream_t)stream, a);
  DEVIT_LAUNCH_CHECK();
  return DEVIT_OK;
}
