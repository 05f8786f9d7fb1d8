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

// Copy rows [0, KROWS) of a strided [N][64] bf16 matrix into an LDS image, zero rows >= N.
// All 7 loads of a thread are issued before the first LDS write (one HBM latency, not seven).
__device__ __forceinline__ void load_image(char* img, const __bf16* src, size_t row_stride, int N, float mul, int tid) {
  constexpr int ITERS = KROWS * 8 / 256;
  bf16x8 v[ITERS];
#pragma unroll
  for (int it = 0; it < ITERS; ++it) {
    const int idx = tid + it * 256, row = idx >> 3, c = idx & 7;
    const bf16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
    v[it] = row < N ? *(const bf16x8*)(src + (size_t)row * row_stride + c * 8) : z;
  }
#pragma unroll
  for (int it = 0; it < ITERS; ++it) {
    const int idx = tid + it * 256, row = idx >> 3, c = idx & 7;
    if (mul != 1.0f) {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[it][e] = f2bf(bf2f(v[it][e]) * mul);
    }
    *(bf16x8*)(img + img_off(row, c)) = v[it];
  }
}

struct AttnFwdArgs {
  const __bf16* qkv;
  __bf16* out;
  float* lse;
  const float* head_gate;
  int B, N, H;
  float scale;
};

__global__ __launch_bounds__(256) void attn_fwd_kernel(const AttnFwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* k_img = smem;
  char* v_img = smem + IMG_BYTES;
  char* p_all = smem + 2 * IMG_BYTES;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int b = blockIdx.x / a.H, h = blockIdx.x % a.H;
  const int D = a.H * HD, N = a.N;
  const size_t rs = (size_t)3 * D;
  const __bf16* qbase = a.qkv + (size_t)b * N * rs + h * HD;
  load_image(k_img, qbase + D, rs, N, 1.0f, tid);
  load_image(v_img, qbase + 2 * D, rs, N, 1.0f, tid);
  __syncthreads();

  char* pbuf = p_all + wave * (16 * PSTRIDE * 2);
  const float c2 = a.scale * 1.4426950408889634f;  // scores in log2 domain
  const float gate = a.head_gate ? a.head_gate[h] : 1.0f;
  const int ntile = (N + 15) >> 4;
  const int lr = lane >> 4, lc = lane & 15;

  for (int qt = wave; qt < ntile; qt += 4) {
    const int q0 = qt * 16;
    // Q fragments straight from global (row-major, d contiguous)
    bf16x8 qf[2];
    {
      const int row = q0 + lc;
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        bf16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
        qf[kk] = row < N ? *(const bf16x8*)(qbase + (size_t)row * rs + kk * 32 + lr * 8) : z;
      }
    }
    f32x4 s[MAXT];
#pragma unroll
    for (int j = 0; j < MAXT; ++j) {
      s[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) s[j] = mfma16(qf[kk], img_row_frag(k_img, j * 16, kk, lane), s[j]);
    }
    float mx[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
    for (int j = 0; j < MAXT; ++j) {
      const bool ok = j * 16 + lc < N;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        s[j][r] = ok ? s[j][r] * c2 : -INFINITY;
        mx[r] = fmaxf(mx[r], s[j][r]);
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
#pragma unroll
      for (int o = 1; o < 16; o <<= 1) mx[r] = fmaxf(mx[r], __shfl_xor(mx[r], o, 64));
    }
    float sum[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < MAXT; ++j) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float p = exp2f(s[j][r] - mx[r]);
        sum[r] += p;
        *(__bf16*)(pbuf + (lr * 4 + r) * (PSTRIDE * 2) + (j * 16 + lc) * 2) = f2bf(p);
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      *(__bf16*)(pbuf + (lr * 4 + r) * (PSTRIDE * 2) + (MAXT * 16 + lc) * 2) = f2bf(0.f);  // keys 208..223
#pragma unroll
      for (int o = 1; o < 16; o <<= 1) sum[r] += __shfl_xor(sum[r], o, 64);
    }
    if (a.lse && lc == 0) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int q = q0 + lr * 4 + r;
        if (q < N) a.lse[((size_t)b * a.H + h) * N + q] = (mx[r] + log2f(sum[r])) * 0.6931471805599453f;
      }
    }
    // O = P V
    f32x4 o[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) o[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 7; ++ks) {
      const bf16x8 pf = pbuf_row_frag(pbuf, 0, ks, lane);
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) o[dt] = mfma16(pf, img_tr_frag(v_img, ks * 32, dt * 16, lane), o[dt]);
    }
    // stage the 16 x 64 output tile in this wave's P buffer (128-B rows) and store whole rows
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        *(__bf16*)(pbuf + (lr * 4 + r) * 128 + (dt * 16 + lc) * 2) = f2bf(o[dt][r] * (gate / sum[r]));
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int idx = lane + i * 64, row = idx >> 3, c = idx & 7;
      const bf16x8 v = *(const bf16x8*)(pbuf + row * 128 + c * 16);
      if (q0 + row < N) *(bf16x8*)(a.out + ((size_t)b * N + q0 + row) * D + h * HD + c * 8) = v;
    }
  }
}

// ------------------------------------------------------------------------------------------
// Backward.  Recomputes P from the saved log-sum-exp.  Per 32-query block: phase A computes
// S and dP (key tiles split over waves) and writes P, dS (bf16) to LDS; phase B accumulates
// dV += P^T dO, dK += dS^T Q in registers (each wave owns its key tiles for the whole kernel,
// so no cross-workgroup reduction) and computes dQ = dS K for the block.
// ------------------------------------------------------------------------------------------
struct AttnBwdArgs {
  const __bf16* qkv;
  const __bf16* out;    // forward output (post gate)  [B*N][D]
  const __bf16* dout;   // gradient wrt forward output [B*N][D]
  const float* lse;     // [B][H][N]
  const float* head_gate;
  const __bf16* dqkv_add;  // optional extra gradient added into dqkv (relation loss), same layout
  __bf16* dqkv;         // [B*N][3D]
  int B, N, H;
  float scale;
};

__global__ __launch_bounds__(256) void attn_bwd_kernel(const AttnBwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* q_img = smem;
  char* k_img = smem + IMG_BYTES;
  char* v_img = smem + 2 * IMG_BYTES;
  char* do_img = smem + 3 * IMG_BYTES;
  char* p_buf = smem + 4 * IMG_BYTES;
  char* ds_buf = p_buf + 32 * PSTRIDE * 2;
  float* lse2 = (float*)(ds_buf + 32 * PSTRIDE * 2);
  float* delta = lse2 + KROWS;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int b = blockIdx.x / a.H, h = blockIdx.x % a.H;
  const int D = a.H * HD, N = a.N;
  const size_t rs = (size_t)3 * D;
  const __bf16* qbase = a.qkv + (size_t)b * N * rs + h * HD;
  const float gate = a.head_gate ? a.head_gate[h] : 1.0f;
  const __bf16* dobase = a.dout + (size_t)b * N * D + h * HD;
  const __bf16* obase = a.out + (size_t)b * N * D + h * HD;

  load_image(q_img, qbase, rs, N, 1.0f, tid);
  load_image(k_img, qbase + D, rs, N, 1.0f, tid);
  load_image(v_img, qbase + 2 * D, rs, N, 1.0f, tid);
  load_image(do_img, dobase, (size_t)D, N, gate, tid);
  // zero the P / dS buffers once: key columns of tile 13 (208..223) stay zero for the dQ k-steps
  for (int i = tid; i < (2 * 32 * PSTRIDE * 2) / 16; i += 256) ((f32x4*)p_buf)[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  if (tid < KROWS) {
    float l2 = 0.f, dl = 0.f;
    if (tid < N) {
      l2 = a.lse[((size_t)b * a.H + h) * N + tid] * 1.4426950408889634f;
      const __bf16* dr = dobase + (size_t)tid * D;
      const __bf16* orow = obase + (size_t)tid * D;
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        const bf16x8 x = *(const bf16x8*)(dr + c * 8), y = *(const bf16x8*)(orow + c * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) dl += bf2f(x[e]) * bf2f(y[e]);
      }
    }
    lse2[tid] = l2;
    delta[tid] = dl;
  }
  __syncthreads();

  const float c2 = a.scale * 1.4426950408889634f;
  const int ntile = (N + 15) >> 4;  // key tiles / q tiles that contain real tokens
  const int lr = lane >> 4, lc = lane & 15;

  f32x4 dv[4][4], dk[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      dv[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
      dk[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }

  const int nblk = (N + 31) >> 5;
  for (int qb = 0; qb < nblk; ++qb) {
    // ---------------- phase A: P and dS for query rows [32 qb, 32 qb + 32)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int q0 = qb * 32 + i * 16;
      bf16x8 qf[2], dof[2];
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        qf[kk] = img_row_frag(q_img, q0, kk, lane);
        dof[kk] = img_row_frag(do_img, q0, kk, lane);
      }
      float l2[4], dl[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        l2[r] = lse2[q0 + lr * 4 + r];
        dl[r] = delta[q0 + lr * 4 + r];
      }
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int kt = wave + t * 4;
        if (kt < ntile) {
          f32x4 s = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int kk = 0; kk < 2; ++kk) {
            s = mfma16(qf[kk], img_row_frag(k_img, kt * 16, kk, lane), s);
            dp = mfma16(dof[kk], img_row_frag(v_img, kt * 16, kk, lane), dp);
          }
          const bool kok = kt * 16 + lc < N;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const bool ok = kok && (q0 + lr * 4 + r < N);
            const float p = ok ? exp2f(s[r] * c2 - l2[r]) : 0.f;
            const float d = p * (dp[r] - dl[r]) * a.scale;
            const int off = (i * 16 + lr * 4 + r) * (PSTRIDE * 2) + (kt * 16 + lc) * 2;
            *(__bf16*)(p_buf + off) = f2bf(p);
            *(__bf16*)(ds_buf + off) = f2bf(d);
          }
        }
      }
    }
    __syncthreads();
    // ---------------- phase B
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int kt = wave + t * 4;
      if (kt < ntile) {
        const bf16x8 pt = pbuf_tr_frag(p_buf, 0, kt * 16, lane);    // A[m = key][k = q]
        const bf16x8 dst = pbuf_tr_frag(ds_buf, 0, kt * 16, lane);
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
          dv[t][dt] = mfma16(pt, img_tr_frag(do_img, qb * 32, dt * 16, lane), dv[t][dt]);
          dk[t][dt] = mfma16(dst, img_tr_frag(q_img, qb * 32, dt * 16, lane), dk[t][dt]);
        }
      }
    }
    {
      // dQ: wave -> q tile i = wave >> 1, d tiles {2 (wave & 1), +1}
      const int i = wave >> 1, dt0 = (wave & 1) * 2;
      f32x4 dq[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
      for (int ks = 0; ks < 7; ++ks) {
        const bf16x8 dsf = pbuf_row_frag(ds_buf, i * 16, ks, lane);
#pragma unroll
        for (int u = 0; u < 2; ++u) dq[u] = mfma16(dsf, img_tr_frag(k_img, ks * 32, (dt0 + u) * 16, lane), dq[u]);
      }
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int q = qb * 32 + i * 16 + lr * 4 + r;
          if (q < N) {
            const size_t o = ((size_t)b * N + q) * rs + h * HD + (dt0 + u) * 16 + lc;
            float v = dq[u][r];
            if (a.dqkv_add) v += bf2f(a.dqkv_add[o]);
            a.dqkv[o] = f2bf(v);
          }
        }
    }
    __syncthreads();
  }
  // ---------------- dK, dV of this wave's key tiles
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int kt = wave + t * 4;
    if (kt < ntile) {
#pragma unroll
      for (int dt = 0; dt < 4; ++dt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int key = kt * 16 + lr * 4 + r;
          if (key < N) {
            const size_t o = ((size_t)b * N + key) * rs + h * HD + dt * 16 + lc;
            float vk = dk[t][dt][r], vv = dv[t][dt][r];
            if (a.dqkv_add) {
              vk += bf2f(a.dqkv_add[o + D]);
              vv += bf2f(a.dqkv_add[o + 2 * D]);
            }
            a.dqkv[o + D] = f2bf(vk);
            a.dqkv[o + 2 * D] = f2bf(vv);
          }
        }
    }
  }
}

constexpr int FWD_LDS = 2 * IMG_BYTES + 4 * 16 * PSTRIDE * 2;                     // 87040
constexpr int BWD_LDS = 4 * IMG_BYTES + 2 * 32 * PSTRIDE * 2 + 2 * KROWS * 4;     // 146176

}  // namespace

extern "C" int devit_attn_fwd(const void* qkv, void* out, float* lse, const float* head_gate, int B, int N, int H,
                              int head_dim, float scale, void* stream) {
  DEVIT_CHECK(qkv && out, DEVIT_ERR_ARG, "devit_attn_fwd: null pointer");
  DEVIT_CHECK(head_dim == HD && N > 0 && N <= MAXT * 16 && B > 0 && H > 0, DEVIT_ERR_SHAPE,
              "devit_attn_fwd: needs head_dim == 64 and N <= 208 (got hd=%d N=%d)", head_dim, N);
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)attn_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, FWD_LDS);
    DEVIT_CHECK(e == hipSuccess, DEVIT_ERR_LAUNCH, "devit_attn_fwd: hipFuncSetAttribute: %s", hipGetErrorString(e));
    attr_set = true;
  }
  AttnFwdArgs a{(const __bf16*)qkv, (__bf16*)out, lse, head_gate, B, N, H, scale};
  hipLaunchKernelGGL(attn_fwd_kernel, dim3(B * H), dim3(256), FWD_LDS, (hipStream_t)stream, a);
  DEVIT_LAUNCH_CHECK();
  return DEVIT_OK;
}

extern "C" int devit_attn_bwd(const void* qkv, const void* out, const void* dout, const float* lse,
                              const float* head_gate, const void* dqkv_add, void* dqkv, int B, int N, int H,
                              int head_dim, float scale, void* stream) {
  DEVIT_CHECK(qkv && out && dout && lse && dqkv, DEVIT_ERR_ARG, "devit_attn_bwd: null pointer");
  DEVIT_CHECK(head_dim == HD && N > 0 && N <= MAXT * 16 && B > 0 && H > 0, DEVIT_ERR_SHAPE,
              "devit_attn_bwd: needs head_dim == 64 and N <= 208 (got hd=%d N=%d)", head_dim, N);
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)attn_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, BWD_LDS);
    DEVIT_CHECK(e == hipSuccess, DEVIT_ERR_LAUNCH, "devit_attn_bwd: hipFuncSetAttribute: %s", hipGetErrorString(e));
    attr_set = true;
  }
  AttnBwdArgs a{(const __bf16*)qkv, (const __bf16*)out, (const __bf16*)dout, lse, head_gate,
                (const __bf16*)dqkv_add, (__bf16*)dqkv, B, N, H, scale};
  hipLaunchKernelGGL(attn_bwd_kernel, dim3(B * H), dim3(256), BWD_LDS, (hipStream_t)stream, a);
  DEVIT_LAUNCH_CHECK();
  return DEVIT_OK;
}
