/* devit_hip.h -- C ABI of libdevit_hip.so: hand-written gfx950 (MI355X) kernels for the
 * DeViT hot path (ViT block forward/backward + DEKD distillation losses).
 *
 * The reference (falcon-xu/DeViT) has no FFI: its boundary is the PyTorch module surface
 * (SURVEY.md §8b).  Each entry point below names the reference op sequence it replaces
 * (file:line in the reference tree).  INTEGRATION.md shows the ctypes binding.
 *
 * Conventions (all entry points):
 *   - plain pointers + sizes, no torch types; every pointer is a DEVICE pointer unless noted
 *   - returns 0 on success, a negative DEVIT_ERR_* otherwise; devit_last_error() has the text
 *   - never allocates/frees device memory, never synchronises, enqueues only on `stream`
 *     (a hipStream_t passed as void*; NULL = default stream); safe to capture in a hipGraph
 *   - bf16 tensors are raw uint16 storage; "f32" is IEEE float
 *   - rows = tokens: M = B * N (N = 198 tokens for distilled 224x224 models)
 */
#ifndef DEVIT_HIP_H
#define DEVIT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2 (round 6): devit_block_weights grew (fc2_w16t), devit_index_copy mode 4, devit_wgrad_grouped, devit_abi_struct_size, exported symbols
 * only (the library is built with -fvisibility=hidden).  A binding asks devit_version() AND checks its struct sizes against
 * devit_abi_struct_size() before it passes arrays of structs (devit_encoder_fwd strides by sizeof(devit_block_weights)). */
#define DEVIT_ABI_VERSION 2
#define DEVIT_API __attribute__((visibility("default")))

enum {
  DEVIT_OK = 0,
  DEVIT_ERR_SHAPE = -1,   /* dimension not supported by the kernel's tiling */
  DEVIT_ERR_ARG = -2,     /* null pointer / bad enum / misaligned pointer */
  DEVIT_ERR_LAUNCH = -3,  /* HIP reported a launch error */
  DEVIT_ERR_DEVICE = -4   /* not a gfx950 device / no device */
};

DEVIT_API int devit_version(void);
DEVIT_API const char* devit_last_error(void); /* host string, thread-local, valid until next call */
/* 0 if device `dev` is gfx950; DEVIT_ERR_DEVICE otherwise (product path refuses to run). */
DEVIT_API int devit_check_device(int dev);
/* sizeof() of the ABI's structs as THIS library was compiled (which: 0 devit_epilogue, 1 devit_operand, 2 devit_block_weights,
 * 3 devit_block_wgrads, 4 devit_block_acts, 5 devit_block_bwd_io, 6 devit_index_job, 7 devit_wgrad_job; anything else: 0).  A binding
 * compares them with its own mirrors once at load time: a stale mirror of a struct that is passed as an ARRAY would otherwise be misread
 * from its second element on, with no error. */
DEVIT_API size_t devit_abi_struct_size(int which);

/* ------------------------------------------------------------------------------------------
 * GEMM  C[M,N] = sum_k A(m,k) * B(n,k), bf16 operands, fp32 accumulate (MFMA 16x16x32).
 * Replaces every nn.Linear on the path: qkv models/de_vit.py:67, proj :82, fc1 :36, fc2 :45,
 * PatchEmbed conv-as-GEMM :258 and their autograd backward (dgrad / wgrad).
 *
 *   Operands are described by devit_operand:
 *   kmajor = 0: stored [rows][K] (k contiguous), row stride ld elements       (fwd: X[M][K], W[N][K])
 *   kmajor = 1: stored [K][rows] (row index contiguous), row stride ld         (dgrad: W as B; wgrad: both)
 *   row_group/row_skip (k-major only): physical row of reduction row r is r + skip * (r / group + 1)
 *   (0/0 = dense).  Lets the patch-embed wgrad read token rows 2..197 of every image straight from the
 *   [B,198,D] gradient.
 *   batch > 1: operand z uses ptr + z * batch_stride (elements); output uses out + z * out_batch_stride.
 *   Requirements: M % 128 == 0, N % 128 == 0, K % 64 == 0, 16-byte aligned pointers / strides.
 *   Instantiated (layout, epilogue) pairs: A,B k-contiguous: STORE_BF16/F32, GELU, RESIDUAL, PATCH; B k-major
 *   (dgrad): STORE_BF16/F32, DGELU; A and B k-major (wgrad): ATOMIC_F32, STORE_F32.  Others: DEVIT_ERR_ARG.
 *   split_k > 1 is only legal with DEVIT_EPI_ATOMIC_F32.  ep->m_valid > 0 stores only rows m < m_valid
 *   of each batch (padded per-image Grams of the relation loss).
 * ---------------------------------------------------------------------------------------- */
typedef enum {
  DEVIT_EPI_STORE_BF16 = 0, /* out_bf16 = acc + bias                                           */
  DEVIT_EPI_GELU_BF16 = 1,  /* out_bf16 = gelu(acc + bias) * colscale; aux_bf16 = acc + bias    */
                            /*   (Mlp.forward de_vit.py:36-43: fc1, exact-erf GELU, neuron gate) */
  DEVIT_EPI_RESIDUAL_F32 = 2, /* out_f32 = res_f32 + rowscale[m / rows_per_scale] * (acc + bias) */
                            /*   (Block.forward :114-115 residual + DropPath); aux_bf16 optional  */
                            /*   copy of (acc + bias) (the 'attention' output, :119)              */
  DEVIT_EPI_PATCH_F32 = 3,  /* row m=(b,t): out_f32[b*(T+tok)+tok+t] = acc + bias + pos[tok+t]    */
                            /*   (forward_features :258-264; cls/dist rows written by            */
                            /*   devit_embed_tokens)                                             */
  DEVIT_EPI_DGELU_BF16 = 4, /* out_bf16 = acc * colscale * gelu'(aux_in_bf16)  (fc2 dgrad -> dfc1) */
  DEVIT_EPI_ATOMIC_F32 = 5, /* out_f32 += acc   (atomicAdd; wgrad with split-K); aux_f32[m] += sum_k A[m][k]       */
                            /*   when aux != NULL (the bias gradient of the same Linear: A = dY^T), batch == 1   */
  DEVIT_EPI_STORE_F32 = 6   /* out_f32 = acc + bias                                             */
} devit_epilogue_kind;

typedef struct {
  int kind;                /* devit_epilogue_kind */
  void* out;               /* [M][ldc] bf16 or f32 depending on kind */
  int ldc;
  const float* bias;       /* [N] or NULL */
  const float* colscale;   /* [N] gate (GELU / DGELU) or NULL (= 1) */
  void* aux;               /* GELU: pre-activation out (bf16, ld = ldc) or NULL; RESIDUAL: bf16 copy or NULL;
                              ATOMIC: f32 [M] row sums of A, accumulated, or NULL */
  const void* aux_in;      /* DGELU: saved pre-activation [M][ldc] bf16 */
  const float* res;        /* RESIDUAL: [M][ldc] f32 input stream (may alias out) */
  const float* rowscale;   /* RESIDUAL: [M / rows_per_scale] per-sample DropPath scale or NULL */
  int rows_per_scale;      /* RESIDUAL: tokens per sample (198) */
  const float* pos;        /* PATCH: [tok + T][N] f32 position embedding */
  int patch_tokens;        /* PATCH: T = 196 */
  int extra_tokens;        /* PATCH: tok = 2 (cls + dist) or 1 */
  int exact_gelu;          /* must be 0: the fused GELU is a fitted form, |err| <= 2.6e-5 (erff() variant not built) */
  long long out_batch_stride; /* elements between consecutive batch outputs (out, aux, aux_in, res) */
  int m_valid;             /* > 0: rows m >= m_valid of each batch are not stored */
  int dtype16;             /* 16-bit type of A, B and of any 16-bit output of the launch: 0 = bf16, 1 = IEEE f16 (forward
                              layouts and epilogues only: the frozen teacher's forward, 11 significand bits instead of 8) */
} devit_epilogue;

typedef struct {
  const void* ptr;         /* bf16 */
  int ld;                  /* elements */
  int kmajor;
  int row_group, row_skip;
  long long batch_stride;  /* elements */
} devit_operand;

DEVIT_API int devit_gemm_bf16(const devit_operand* A, const devit_operand* B, int M, int N, int K, int batch, int split_k,
                    const devit_epilogue* ep, void* stream);
/* The GEMM grids are persistent: one (256x256 tile) or two (128x128) workgroups per CU that walk their share of the tiles.
 * devit_set_reserved_cus(n) makes every later launch leave n CUs (a multiple of 8: one share per XCD) free for kernels of
 * other streams -- the RCCL all-reduce of the data-parallel step (distill_sub.py:333), whose kernels could otherwise start
 * only when a GEMM ends.  Default: the DEVIT_RESERVE_CUS environment variable, else 0.  Process-wide. */
/* Would devit_gemm_bf16 run (row-major A) x (K-MAJOR B) with N outputs and this epilogue kind on the full-row 256x384 kernel (N == 384, whole
 * 256-row tiles and >= 64 of them, K >= 192; DEVIT_GEMMFR=0 / 1 in the environment forces it off / on)?  1 / 0.  The fp32 residual epilogue
 * with a k-major weight exists on that kernel only, so a caller that holds a k-major copy of a forward weight (devit_block_weights.fc2_w16t)
 * asks first. */
DEVIT_API int devit_gemm_full_row_selected(int M, int N, int K, int kind);
DEVIT_API int devit_set_reserved_cus(int n);
DEVIT_API int devit_get_reserved_cus(void);

/* ------------------------------------------------------------------------------------------
 * Weight gradients of several nn.Linear layers as ONE launch (autograd's dW = dY^T X of models/de_vit.py:67,82,36,45, i.e. what
 * engine.py:123-127's backward() leaves in .grad): a table of jobs, each
 *     out[i][j] += sum_k a[k][i] * b[k][j]      i < a_cols, j < 384           (transposed == 0: out is [a_cols][ldc])
 *     out[j][i] += ...                                                         (transposed != 0: out is [384][ldc])
 *     a_colsum[i] += sum_k a[k][i]              (optional: the bias gradient when a = dY)
 * with both operands K-MAJOR as the step holds them ([K token rows][features] bf16, rows >= the real row count zero).  One operand must be
 * exactly 384 columns wide (the student's D; fc2's gradient is taken transposed, a = the GELU output, b = dY); a_cols % 128 == 0.
 * Kernel: 256 x 384 output tiles on the full-row tile of devit_gemm_bf16 (k-major x k-major variant, one tile and one K slice per
 * workgroup, fp32 atomics through LDS in whole 256-byte rows); every workgroup of the table runs at once, so split_k slices x tiles
 * should fill the device: split_k == 0 picks floor(CUs / tiles).  K % 64 == 0, K / 64 / split_k >= 3, njobs <= 48 (the four products of
 * up to twelve blocks: devit_block_bwd_io.defer_jobs collects them).
 * jobs is a HOST array (copied into the kernel arguments).
 * ---------------------------------------------------------------------------------------- */
typedef struct {
  const void* a;           /* bf16 [K][lda]; a_cols columns are read (tiles of 256; a last half tile is allowed) */
  int lda, a_cols;
  const void* b;           /* bf16 [K][ldb]; columns 0..383 are read */
  int ldb;
  float* out;              /* fp32, accumulated */
  int ldc, transposed;
  float* a_colsum;         /* fp32 [a_cols], accumulated, or NULL */
} devit_wgrad_job;
DEVIT_API int devit_wgrad_grouped(const devit_wgrad_job* jobs, int njobs, int K, int split_k, void* stream);

/* ------------------------------------------------------------------------------------------
 * LayerNorm over the fp32 residual stream.  Replaces nn.LayerNorm(D, eps=1e-6) at
 * models/de_vit.py:113 (norm1), :115 (norm2), :286 (final norm) and its backward.
 *   physical input row of logical row r: in_group > 0 ? (r / in_group) * in_stride + r % in_group : r
 *   (final norm: only the cls/dist rows are consumed, de_vit.py:288 -> in_group = 2, in_stride = 198)
 *   y_bf16 / y_f32: either may be NULL.  mean / rstd: [rows] saved for backward (may be NULL).
 *   D % 128 == 0, D <= 1024.
 * bwd: dx[phys row] = (dres ? dres[phys row] : 0) + LN'(dy[r]); dx_bf16 (optional) = rowscale * dx as the
 *   bf16 branch gradient consumed by the previous sub-block's dgrad / wgrad GEMMs;
 *   dgamma / dbeta [D] (accumulate != 0 adds); dx_bf16_colsum (optional) [D] (+)= column sums of dx_bf16 = the
 *   bias gradient of the Linear layer that produced the branch (saves a pass over the matrix).
 *   workspace >= devit_layernorm_bwd_workspace(rows, D).
 * ---------------------------------------------------------------------------------------- */
DEVIT_API int devit_layernorm_fwd(const float* x, int rows, int D, int in_group, int in_stride, const float* gamma,
                        const float* beta, float eps, void* y_bf16, float* y_f32, float* mean, float* rstd,
                        int dtype16 /* type of y_bf16: 0 bf16, 1 f16 */, void* stream);
DEVIT_API size_t devit_layernorm_bwd_workspace(int rows, int D);
DEVIT_API int devit_layernorm_bwd(const void* dy, int dy_is_f32, const float* x, int rows, int D, int in_group, int in_stride,
                        const float* mean, const float* rstd, const float* gamma, const float* dres, float* dx,
                        void* dx_bf16, const float* rowscale, int rows_per_scale, float* dgamma, float* dbeta,
                        float* dx_bf16_colsum, int accumulate, void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------
 * Fused attention core.  Replaces models/de_vit.py:68-79 (unbind q,k,v; q k^T * scale; softmax;
 * @ v; transpose; head-gate mul_) and its backward.
 *   qkv  : bf16 [B*N][3*H*hd], feature index j*H*hd + h*hd + e (output of the qkv GEMM, de_vit.py:67)
 *   out  : bf16 [B*N][H*hd] (post head gate == Attention.head_output, de_vit.py:77-79)
 *   lse  : f32 [B][H][N] natural-log sum-exp of the scaled scores (NULL when no backward is needed)
 *   head_gate: f32 [H] or NULL.   head_dim must be 64, N <= 208.
 * bwd: dqkv (same layout as qkv) from dout; dqkv_add (optional, same layout) is added in.
 * ---------------------------------------------------------------------------------------- */
DEVIT_API int devit_attn_fwd(const void* qkv, void* out, float* lse, const float* head_gate, int B, int N, int H, int head_dim,
                   float scale, int dtype16 /* type of qkv and out: 0 bf16, 1 f16 */, void* stream);
DEVIT_API int devit_attn_bwd(const void* qkv, const void* out, const void* dout, const float* lse, const float* head_gate,
                   const void* dqkv_add, void* dqkv, int B, int N, int H, int head_dim, float scale, void* stream);
/* The same kernels with the query side apart from the key side: NQ <= N query rows per image against all N keys.
 * Used for the LAST block when the caller consumes only the class / distillation tokens (models/de_vit.py:286-288 reads
 * x[:, 0], x[:, 1] after the final norm; engine.py:91-92 reads q/k/v of the middle block only): the other 196 query
 * rows of that block are never read, so their attention (and projection, MLP) is not computed -- same arithmetic per
 * computed row, bit-identical outputs.
 *   q   : bf16 [B*NQ][q_ld], feature h*hd + e        kv  : bf16 [B*N][kv_ld], K at feature h*hd + e, V at H*hd + h*hd + e
 *   out / dout : bf16 [B*NQ][H*hd]                    lse : f32 [B][H][NQ]
 *   dq  : bf16 [B*NQ][dq_ld]                          dkv : bf16 [B*N][dkv_ld] (dK | dV), every key row written */
DEVIT_API int devit_attn_fwd_rows(const void* q, int q_ld, const void* kv, int kv_ld, void* out, float* lse, const float* head_gate,
                        int B, int NQ, int N, int H, int head_dim, float scale, int dtype16, void* stream);
DEVIT_API int devit_attn_bwd_rows(const void* q, int q_ld, const void* kv, int kv_ld, const void* out, const void* dout,
                        const float* lse, const float* head_gate, void* dq, int dq_ld, void* dkv, int dkv_ld, int B, int NQ,
                        int N, int H, int head_dim, float scale, void* stream);

/* ------------------------------------------------------------------------------------------
 * Whole encoder blocks per call.  Replaces Block.forward (models/de_vit.py:103-121: norm1 -> Attention :65-87 ->
 * DropPath + residual -> norm2 -> Mlp :35-47 -> DropPath + residual) and its autograd backward as ONE host call that
 * enqueues the block's 8 (forward) / 14 (backward) kernels -- the same kernels, arguments and order as the
 * single-kernel entry points above, so results are bit-identical to calling those one by one.  It exists to take the
 * host out of the step: a DEKD step is ~650 launches, and a Python/ctypes round trip per launch costs more than the
 * kernels take (19 ms of host time for a 28 ms step).  All memory is the caller's; nothing is allocated here.
 *
 *   devit_block_weights : device pointers of one block's parameters (bf16 GEMM copies + fp32 biases / LN params / gates)
 *   devit_block_acts    : one block forward's activations; buffer sizes from devit_block_acts_sizes (index = DEVIT_ACT_*).
 *                         bf16 row buffers have pad_rows(M) = ceil(M / 256) * 256 rows (qkv: + 128 when flagged) and
 *                         their rows >= M are zeroed by the forward (GEMM tiles read them, the weight-gradient GEMMs
 *                         reduce over them).  Pointers of buffers whose size is reported 0 may be NULL.
 *   flags               : DEVIT_BLK_SAVE   keep what backward needs (mean/rstd, lse, fc1 pre-activation)
 *                         DEVIT_BLK_QKV_PAD 128 zeroed overhang rows behind qkv (read by the relation-loss Gram windows)
 *                         DEVIT_BLK_ATT    also store the attention-branch output (pre-residual, de_vit.py:119) as bf16
 *   devit_encoder_fwd   : blocks 0..nblocks-1 in sequence; acts[i].x2 feeds acts[i+1].x (the caller points them at the
 *                         same memory); flags per block.
 *   devit_block_bwd     : gradient of one block.  io->dx = d loss / d x2 (fp32), io->g2 = bf16(dp2 * dx) with zero pad
 *                         rows; produces io->dx_in and (optional) io->g_prev = bf16(prev_dp2 * dx_in) for the block
 *                         below together with that block's fc2 bias gradient (column sums, io->prev_fc2_b_grad);
 *                         weight gradients are ACCUMULATED into devit_block_wgrads (fp32).  io->dqkv_add: gradient
 *                         arriving at the packed qkv output from outside the block (relation loss) or NULL.
 *                         The seven transient buffers are sized by devit_block_bwd_sizes and may be reused by the next
 *                         call; their pad rows are zeroed here.
 * ---------------------------------------------------------------------------------------- */
typedef struct {
  const float *n1w, *n1b, *qkv_b, *proj_b, *n2w, *n2b, *fc1_b, *fc2_b;
  const void *qkv_w16, *proj_w16, *fc1_w16, *fc2_w16;   /* bf16 [out][in] */
  const float *head_gate, *neuron_gate;                 /* [heads] / [hidden] or NULL (= 1) */
  int num_heads, attn_width, hidden;                    /* attn_width = heads * 64 (== D unless compacted) */
  int dtype16;                                          /* 16-bit type of the weights and of every stored activation: 0 bf16,
                                                           1 f16 (forward without DEVIT_BLK_SAVE only) */
  const void* fc2_w16t;                                 /* bf16 [hidden][D]: a k-major copy of fc2_w16 (devit_index_copy mode 4), or NULL.
                                                           With it, D == 384 and >= 64 row tiles of 256 the forward's fc2 launch runs on the
                                                           full-row 256x384 GEMM (same products, same accumulation order: bit-identical) */
} devit_block_weights;

typedef struct {
  float *n1w, *n1b, *qkv_w, *qkv_b, *proj_w, *proj_b, *n2w, *n2b, *fc1_w, *fc1_b, *fc2_w, *fc2_b;
} devit_block_wgrads;

enum { DEVIT_ACT_LN1 = 0, DEVIT_ACT_MEAN1, DEVIT_ACT_RSTD1, DEVIT_ACT_QKV, DEVIT_ACT_ATTN_O, DEVIT_ACT_LSE, DEVIT_ACT_X1,
       DEVIT_ACT_ATT, DEVIT_ACT_LN2, DEVIT_ACT_MEAN2, DEVIT_ACT_RSTD2, DEVIT_ACT_H, DEVIT_ACT_H_PRE, DEVIT_ACT_X2,
       DEVIT_ACT_COUNT };
enum { DEVIT_BLK_SAVE = 1, DEVIT_BLK_QKV_PAD = 2, DEVIT_BLK_ATT = 4 };

typedef struct {
  const float* x;          /* in: fp32 [M][D] residual stream */
  void* buf[DEVIT_ACT_COUNT];
  const float *dp1, *dp2;  /* per-sample DropPath scales [B] of the two branches or NULL */
  int flags;
} devit_block_acts;

enum { DEVIT_BWD_DH_PRE = 0, DEVIT_BWD_DLN2, DEVIT_BWD_DX1, DEVIT_BWD_G1, DEVIT_BWD_DATTN, DEVIT_BWD_DQKV, DEVIT_BWD_DLN1,
       DEVIT_BWD_LNWS, DEVIT_BWD_COUNT };

typedef struct {
  const float* dx;         /* in: fp32 [M][D] */
  const void* g2;          /* in: bf16 [pad_rows(M)][D] */
  float* dx_in;            /* out: fp32 [M][D] */
  void* g_prev;            /* out: bf16 [pad_rows(M)][D] or NULL */
  const float* prev_dp2;   /* [B] or NULL */
  float* prev_fc2_b_grad;  /* [D] accumulated, or NULL */
  int g2_bias_done;        /* != 0: this block's fc2 bias gradient was already produced by the block above */
  const void* dqkv_add;    /* bf16, qkv layout, or NULL */
  void* ws[DEVIT_BWD_COUNT];
  size_t lnws_bytes;
  devit_wgrad_job* defer_jobs; /* HOST array of >= 4 entries or NULL.  Not NULL: the block's weight gradients that can run on the full-row
                                  weight-gradient kernel are NOT launched; they are written here as jobs for a later devit_wgrad_grouped call
                                  (several blocks in one launch: fewer K slices, fewer atomics).  The caller then keeps io->g2 and the
                                  DH_PRE / G1 / DQKV transients (and the forward's activations) alive until that launch has been enqueued */
  int* defer_count;            /* out: jobs written */
} devit_block_bwd_io;

DEVIT_API int devit_block_acts_sizes(int B, int N, int D, int attn_width, int hidden, int flags, size_t* sizes /* [DEVIT_ACT_COUNT] */);
DEVIT_API int devit_block_bwd_sizes(int B, int N, int D, int attn_width, int hidden, size_t* sizes /* [DEVIT_BWD_COUNT] */);
DEVIT_API int devit_encoder_fwd(int nblocks, const devit_block_weights* w, const devit_block_acts* acts, int B, int N, int D,
                      float eps, void* stream);
/* devit_block_bwd is the ONE entry point that does not keep to "enqueue on the caller's stream only": its four weight-gradient GEMMs go to a
 * stream the LIBRARY owns -- one non-blocking stream + five events per device, created at the first call on that device (thread-safe), never
 * destroyed: process lifetime -- each behind an event of the kernel on `stream` that produces its operand, and `stream` waits for the last of them
 * before the call returns (also when it returns an error after the first fork).  Callers therefore still see one stream: everything the call
 * enqueued is ordered before whatever is enqueued on `stream` next, the transient buffers may be reused, the accumulators in `grads` are complete
 * for the next kernel on `stream`.  What a binding must know: (1) the call is not capturable into a hipGraph of `stream` alone before
 * the side stream exists (first call outside capture); (2) a stream-ordered allocator must treat the buffers of `acts` / `io` / `grads` as in use
 * until work enqueued on `stream` AFTER the call has run (they are read by another stream meanwhile); (3) DEVIT_WGRAD_STREAM=0 in the environment
 * (read per call) keeps every launch on `stream`; a device index >= 16 does the same.  Worth +0.9 % on the DEKD step (profiles/r04_h_*). */
DEVIT_API int devit_block_bwd(const devit_block_weights* w, const devit_block_acts* acts, const devit_block_wgrads* grads,
                    const devit_block_bwd_io* io, int B, int N, int D, float eps, void* stream);

/* ------------------------------------------------------------------------------------------
 * Index copies of physically shrunk (compacted) blocks that are being TRAINED (distill_sub.py:384-401 trains the gated
 * student; core/imp_rank.py:65-71,147-153 only mask).  A table of jobs in device memory, one launch:
 *   mode 0  gather rows     dst[i][c]      = src[idx[i]][c]        (16-bit or f32 elements)
 *   mode 1  gather columns  dst[r][j]      = src[r][idx[j]]
 *   mode 2  add rows        dst[idx[i]][c] += src[i][c]            (f32: compact weight gradients into the masters'; the
 *   mode 3  add columns     dst[r][idx[j]] += src[r][j]             source is an accumulator and is ZEROED by the call)
 *   mode 4  transpose       dst[c][r]      = src[r][c]             (16-bit elements, idx unused: the k-major copy of a Linear
 *                                                                    weight the full-row GEMM reads, devit_block_weights.fc2_w16t)
 * rows x cols is the extent of the COMPACT (dense) side; idx entries < 0 mark padding units and are skipped; kept
 * indices are distinct, so the adds need no atomics.  Jobs of one call must not write the same memory.
 * ---------------------------------------------------------------------------------------- */
typedef struct {
  const void* src;
  void* dst;
  const int* idx;        /* device, rows entries (modes 0, 2) or cols entries (modes 1, 3) */
  int rows, cols;
  int src_ld, dst_ld;    /* elements */
  int mode;
  int elem;              /* bytes per element: 2 or 4 (modes 2, 3: 4) */
} devit_index_job;
DEVIT_API int devit_index_copy(const devit_index_job* jobs_device, int njobs, int blocks_per_job, void* stream);

/* ------------------------------------------------------------------------------------------
 * Patch embedding helpers (timm PatchEmbed used at models/de_vit.py:166-168,258 and token assembly
 * :259-264).  im2row: f32 [B,3,224,224] -> bf16 [B*196][768] with k = c*256 + kh*16 + kw; the
 * projection itself is devit_gemm_bf16 with DEVIT_EPI_PATCH_F32.  embed_tokens writes
 * x[b, t, :] = (t == 0 ? cls : dist) + pos[t] for the 1-2 extra tokens (dist == NULL: cls only).
 * embed_bwd: dpos[T][D] = sum_b dx[b]; dcls = dpos[0]; ddist = dpos[1]; dbias = sum_{t>=ntok} dpos[t];
 *   dx_bf16 (optional) = bf16 copy of dx for the patch-projection wgrad.
 * ---------------------------------------------------------------------------------------- */
DEVIT_API int devit_im2row_bf16(const float* img, void* rows, int B, int C, int H, int W, int patch, int dtype16, void* stream);
/* On-device input stage (engine.py:65-66: timm Mixup(mode='batch') on the fp32 batch, then patch_embed): the mixed batch is
 * produced directly as the bf16 patch rows both models' patch-embedding GEMMs read -- one pass over the images.
 *   mode 0: plain im2row; 1: mixup  lam * x + (1 - lam) * x.flip(0);  2: cutmix  x[:, :, y0:y1, x0:x1] = x.flip(0)[...]
 *   (lam, the box and the mixup / cutmix draw are host-side numpy RNG in timm; they are arguments here).
 * devit_mix_targets: [B][C] f32 = lam * smooth_one_hot(y) + (1 - lam) * smooth_one_hot(y.flip(0)), int64 labels. */
DEVIT_API int devit_mix_im2row_bf16(const float* img, void* rows /* bf16 or NULL */, void* rows_f16 /* f16 or NULL */, int B, int mode,
                          double lam, int y0, int y1, int x0, int x1, void* stream);
DEVIT_API int devit_mix_targets(const long long* labels, float* targets, int B, int C, double lam, double smoothing, void* stream);
DEVIT_API int devit_embed_tokens(const float* cls, const float* dist, const float* pos, float* x, int B, int T, int D,
                       void* stream);
DEVIT_API int devit_embed_bwd(const float* dx, int B, int T, int D, int ntok, float* dpos, float* dcls, float* ddist,
                    float* dbias, void* dx_bf16, int accumulate, void* stream);

/* f32 -> bf16 cast of a flat buffer (weights, once per optimizer step). */
DEVIT_API int devit_cast_bf16(const float* src, void* dst, size_t n, int dtype16, void* stream);

/* dst_bf16[m][d] = bf16(src[m][d] * (rowscale ? rowscale[m / rows_per_scale] : 1)): turns the fp32
 * residual-stream gradient into the bf16 branch gradient (DropPath scale folded, de_vit.py:114-115). */
DEVIT_API int devit_scale_cast_bf16(const float* src, void* dst, const float* rowscale, int rows_per_scale, int M, int D,
                          void* stream);

/* Column sums of a bf16 [M][ld] matrix: out[n] (+)= sum_m y[m][n]  (bias gradients of nn.Linear).
 * row_group/row_skip as in devit_operand.  workspace >= devit_colsum_workspace(M, N). */
DEVIT_API size_t devit_colsum_workspace(int M, int N);
DEVIT_API int devit_colsum_bf16(const void* y, int M, int N, int ld, int row_group, int row_skip, float* out, int accumulate,
                      void* workspace, size_t workspace_bytes, void* stream);

/* Small strided f32 GEMM  C[m][n] (+)= alpha * sum_k A[m*sam + k*sak] * B[n*sbn + k*sbk] + bias[n].
 * Classifier heads (models/de_vit.py:317) and their backward; exact fp32 FMA chain. */
DEVIT_API int devit_sgemm_small(const float* A, long long sam, long long sak, const float* B, long long sbn, long long sbk,
                      const float* bias, float* C, int ldc, int M, int N, int K, float alpha, int accumulate,
                      void* stream);

/* ------------------------------------------------------------------------------------------
 * Optimizer tail over FLAT fp32 buffers (engine.py:127 NativeScaler -> clip_grad_norm_ -> AdamW.step,
 * engine.py:131-132 ModelEma.update): sum of squares (global grad norm), then one fused pass
 *   g *= grad_scale * min(1, max_norm / (||g|| + 1e-6));  AdamW;  ema = ema*d + (1-d)*p;  p_bf16 = bf16(p)
 * dyn = device float[3] {lr, 1 - beta1^step, 1 - beta2^step} (device-resident so a captured graph can
 * be replayed while the schedule advances).  gnorm_sq == NULL: no clipping.  n % 4 == 0.
 * no_decay4: one byte per 4-element granule of the flat buffer, non-zero = exempt from weight decay (timm
 * create_optimizer's second parameter group: 1-D tensors, biases, model.no_weight_decay(); distill_sub.py:340); NULL =
 * decay everywhere.  grad_scale: the 1 / world_size of the data-parallel mean (the buckets are all-reduced as sums).
 * ---------------------------------------------------------------------------------------- */
DEVIT_API size_t devit_sumsq_workspace(void);
DEVIT_API int devit_sumsq_f32(const float* g, size_t n, float* out, void* workspace, size_t workspace_bytes, void* stream);
DEVIT_API int devit_adamw_step(float* p, const float* g, float* m, float* v, float* ema, void* p_bf16,
                     const unsigned char* no_decay4, const float* gnorm_sq, const float* dyn, size_t n, float beta1,
                     float beta2, float eps, float weight_decay, float max_norm, float ema_decay, float grad_scale,
                     void* stream);

/* ------------------------------------------------------------------------------------------
 * DEKD logit loss + gradient in one launch.  Replaces DistillLoss.forward (utils/losses.py:156-177)
 * with a timm SoftTargetCrossEntropy base criterion (distill_sub.py:348) and its backward.
 *   kind: 0 none, 1 soft (KL, tau), 2 hard (CE vs argmax of teacher logits, ties -> lowest index)
 *   loss3 = {total, base, distill};  dlogits / dlogits_kd = d total / d logits (upstream grad 1).
 * ---------------------------------------------------------------------------------------- */
DEVIT_API int devit_cls_distill_loss(const float* logits, const float* logits_kd, const float* teacher_logits,
                           const float* soft_targets, int B, int C, int kind, float alpha, float tau, float* loss3,
                           float* dlogits, float* dlogits_kd, void* stream);

/* Token feature-matching loss of the ensemble stage (EnsLoss, utils/losses.py:194,228,241-242: nn.MSELoss):
 * loss[0] (+)= mean((a - b)^2); da (optional) = 2 (a - b) / n. */
DEVIT_API int devit_token_mse(const float* a, const float* b, size_t n, float* loss, float* da, int accumulate, void* stream);

/* ------------------------------------------------------------------------------------------
 * q/k/v feature-relation loss (utils/losses.py:307-328).  The per-image Grams F F^T run on
 * devit_gemm_bf16 (batched, padded to 256x256, DEVIT_EPI_STORE_F32); these two kernels do the rest:
 * stats: row log-sum-exp of gram / sqrt(head_dim) for teacher and student, per-row KL, and
 *        loss = sum_{b,i,j} e^{t}(t - s) / B   (KLDivLoss batchmean, log_target)
 * grad : S = G + G^T, G = (softmax_s - softmax_t) * upstream / (B sqrt(hd_s)), bf16 [B][256][256],
 *        zero outside N x N; then dF_student = S F_student on devit_gemm_bf16.
 * ---------------------------------------------------------------------------------------- */
DEVIT_API int devit_relation_stats(const float* gram_t, const float* gram_s, int B, int N, int ldr, int head_dim_t,
                         int head_dim_s, float* lse_t, float* lse_s, float* row_kl, float* loss, void* stream);
DEVIT_API int devit_relation_grad(const float* gram_t, const float* gram_s, const float* lse_t, const float* lse_s,
                        const float* upstream, int B, int N, int ldr, int head_dim_t, int head_dim_s, void* S_out,
                        int out_is_f32, void* stream);

/* ------------------------------------------------------------------------------------------
 * Exact-fp32 companion path ("parity mode", csrc/sgemm.hip): the same op sequences with fp32 activations and
 * k-ordered fmaf accumulation, for asserting BASELINE.json's 1e-3 logit bar against the reference's fp32 CPU
 * path.  Not tuned (a few TFLOP/s); never used by bench.py.
 *   devit_gemm_f32: C[z][m][n] = alpha * (batch_scale ? batch_scale[z % batch_inner] : 1) *
 *                                sum_k A[z][m*sam + pk(k)*sak] * B[z][n*sbn + k*sbk]  (+ epilogue, fp32 flavours of
 *     STORE_F32 / GELU / DGELU / RESIDUAL / PATCH; accumulate != 0 adds into out for STORE_F32)
 *     batch z = zo * batch_inner + zi uses ptr + zo * bs_outer + zi * bs_inner (image / head);
 *     pk(k) = k_group > 0 ? k + k_skip * (k / k_group + 1) : k   (A only; patch-embed wgrad)
 *   devit_softmax_rows_f32 / devit_softmax_bwd_rows_f32: in-place row softmax of scale*S (+ natural-log LSE) and
 *     its backward dS = scale * P * (dP - sum_j P dP): with the GEMM above they restate de_vit.py:70-74.
 * ---------------------------------------------------------------------------------------- */
DEVIT_API int devit_gemm_f32(const float* A, long long sam, long long sak, long long a_bs_outer, long long a_bs_inner,
                   const float* B, long long sbn, long long sbk, long long b_bs_outer, long long b_bs_inner, int M, int N,
                   int K, int batch, int batch_inner, long long c_bs_outer, long long c_bs_inner, int k_group, int k_skip,
                   float alpha, const float* batch_scale, int accumulate, const devit_epilogue* ep, void* stream);
DEVIT_API int devit_softmax_rows_f32(float* S, int rows, int ncols, int ld, float scale, float* lse, void* stream);
DEVIT_API int devit_softmax_bwd_rows_f32(const float* P, float* dP, int rows, int ncols, int ld, float scale, void* stream);
DEVIT_API int devit_im2row_f32(const float* img, float* rows, int B, void* stream);
DEVIT_API int devit_scale_rows_f32(const float* src, float* dst, const float* rowscale, int rows_per_scale, int M, int D,
                         void* stream);
DEVIT_API int devit_colsum_f32(const float* y, int M, int N, int ld, float* out, int accumulate, void* stream);

/* ------------------------------------------------------------------------------------------
 * Gradient exchange of the data-parallel step (distill_sub.py:333: DistributedDataParallel's reducer):
 * RCCL all-reduce over xGMI, one communicator per process (= per GPU).  RCCL is bound at run time (dlopen; a copy the
 * process already holds, e.g. PyTorch's, is reused), so the library has no link-time dependency on it.
 *   devit_comm_unique_id: rank 0 creates the 128-byte rendezvous id and hands it to the other ranks by any host
 *     channel (the launcher's store, a file, torch.distributed's object broadcast).
 *   devit_comm_init: collective over all ranks; uses the calling thread's current HIP device.
 *   devit_comm_allreduce_f32: in-place SUM of buf[count] on `stream` (asynchronous, like every other entry point);
 *     the caller scales by 1/world (devit_adamw_step's grad_scale) -- one bucket of the flat gradient buffer per call.
 * ---------------------------------------------------------------------------------------- */
#define DEVIT_COMM_ID_BYTES 128
DEVIT_API int devit_comm_unique_id(void* id /* [DEVIT_COMM_ID_BYTES] host */);
DEVIT_API int devit_comm_init(const void* id, int rank, int world, void** comm);
DEVIT_API int devit_comm_allreduce_f32(void* comm, float* buf, size_t count, void* stream);
DEVIT_API int devit_comm_destroy(void* comm);

#ifdef __cplusplus
}
#endif
#endif /* DEVIT_HIP_H */
