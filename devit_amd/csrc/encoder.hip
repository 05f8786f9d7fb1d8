// Whole encoder blocks per host call (devit_encoder_fwd / devit_block_bwd): the block's kernels are enqueued here, in C++,
// through the same single-kernel entry points the granular path uses -- identical kernels, arguments and order, so the
// two paths agree bit for bit (tests/test_gpu_model.py::test_block_calls_match_granular_path; the weight gradients are
// split-K atomics either way).  devit_block_bwd puts its four weight-gradient launches on a side stream of its own and joins
// it before it returns (round 4, see there).  Host code only.
#include <mutex>

#include "devit_common.h"

namespace {

inline int pad_rows(int m) { return (m + 255) / 256 * 256; }
inline size_t align256(size_t b) { return (b + 255) / 256 * 256; }

// split-K factor of a weight-gradient GEMM: one round of resident 128x128 workgroups (two per CU), see ops.split_k_for
inline int split_k_for(int out_rows, int out_cols, int ksteps) {
  const int tiles = (out_rows / 128) * (out_cols / 128);
  int s = 512 / (tiles > 0 ? tiles : 1);
  if (s > ksteps) s = ksteps;
  return s < 1 ? 1 : s;
}

struct Ctx {
  int M, Mp, B, N, D;
  float eps;
  void* stream;
};

int zero_pad(const Ctx& c, void* buf, int cols, size_t elem, int extra_rows = 0) {
  if (!buf) return DEVIT_OK;
  const int rows = c.Mp + extra_rows - c.M;
  if (rows <= 0) return DEVIT_OK;
  hipError_t e = hipMemsetAsync((char*)buf + (size_t)c.M * cols * elem, 0, (size_t)rows * cols * elem, (hipStream_t)c.stream);
  DEVIT_CHECK(e == hipSuccess, DEVIT_ERR_LAUNCH, "hipMemsetAsync: %s", hipGetErrorString(e));
  return DEVIT_OK;
}

devit_epilogue make_ep(int kind, void* out, int ldc, int m_valid, int dtype16 = 0) {
  devit_epilogue ep = {};
  ep.kind = kind;
  ep.out = out;
  ep.ldc = ldc;
  ep.m_valid = m_valid;
  ep.dtype16 = dtype16;
  return ep;
}

// out[M][Nout] = x[Mp][K] @ w[Nout][K]^T (+ epilogue)
int linear_fwd(const Ctx& c, const void* x, const void* w, int Nout, int K, devit_epilogue ep) {
  devit_operand A = {x, K, 0, 0, 0, 0}, Bo = {w, K, 0, 0, 0, 0};
  return devit_gemm_bf16(&A, &Bo, c.Mp, Nout, K, 1, 1, &ep, c.stream);
}
// out[M][K] = dy[Mp][Nw] @ w[Nw][K]   (w read k-major)
int linear_dgrad(const Ctx& c, const void* dy, const void* w, int Nw, int K, devit_epilogue ep) {
  devit_operand A = {dy, Nw, 0, 0, 0, 0}, Bo = {w, K, 1, 0, 0, 0};
  return devit_gemm_bf16(&A, &Bo, c.Mp, K, Nw, 1, 1, &ep, c.stream);
}
// w_grad[Nw][K] += dy[Mp][Nw]^T @ x[Mp][K]; b_grad[Nw] += column sums of dy (same launch)
int linear_wgrad(const Ctx& c, const void* dy, const void* x, float* w_grad, float* b_grad, int Nw, int K) {
  devit_operand A = {dy, Nw, 1, 0, 0, 0}, Bo = {x, K, 1, 0, 0, 0};
  devit_epilogue ep = make_ep(DEVIT_EPI_ATOMIC_F32, w_grad, K, 0);
  ep.aux = b_grad;
  return devit_gemm_bf16(&A, &Bo, Nw, K, c.Mp, 1, split_k_for(Nw, K, c.Mp / 64), &ep, c.stream);
}

// The block's weight gradients as jobs of ONE devit_wgrad_grouped launch (the full-row weight-gradient kernel, gemm.hip): possible when
// one side of every product is exactly 384 features wide (D == 384: the student) and the other a multiple of 128.
//   dW[Nw][K] += dy^T x:   K == 384 -> tiles over dy's features (column sums of dy = the bias gradient from the same launch);
//                          Nw == 384 -> the product transposed, tiles over x's features (no bias gradient: the caller has it from elsewhere)
bool wgrad_job(devit_wgrad_job* j, const void* dy, const void* x, float* w_grad, float* b_grad, int Nw, int K) {
  if (K == 384 && Nw % 128 == 0) {
    *j = devit_wgrad_job{dy, Nw, Nw, x, K, w_grad, K, 0, b_grad};
    return true;
  }
  if (Nw == 384 && K % 128 == 0 && b_grad == nullptr) {
    *j = devit_wgrad_job{x, K, K, dy, Nw, w_grad, K, 1, nullptr};
    return true;
  }
  return false;
}
// DEVIT_WGRADFR (read per call: tests switch it): 0 = four split-K launches on 128x128 tiles (rounds 1-5); else (default) the grouped launch
// (two launches per block -- the MLP pair behind the fc1 dgrad, the attention pair behind the qkv dgrad, for the Infinity Cache -- measured
// 4 % slower on the step than one: twice the atomics)
int wgrad_mode() {
  const char* e = getenv("DEVIT_WGRADFR");
  return (e && *e) ? atoi(e) != 0 : 1;
}

#define TRY(x)                 \
  do {                         \
    int rc__ = (x);            \
    if (rc__ != DEVIT_OK) return rc__; \
  } while (0)

// devit_block_bwd's side stream for the weight-gradient launches: one per device, created at the first call that wants it (once, under
// std::call_once: two host threads making their first call together see one stream), owned by the library for the life of the process
// (declared at devit_block_bwd in include/devit_hip.h).  A device index past the table runs without one (= on the caller's stream).
struct WgradSide {
  hipStream_t stream;
  hipEvent_t ev[5];
  bool ok = false;
};
constexpr int MAX_DEVICES = 16;
int wgrad_side(WgradSide** out) {
  const char* env = getenv("DEVIT_WGRAD_STREAM");      // (read per call: tests switch it)
  const bool on = !(env && atoi(env) == 0);
  static WgradSide per_dev[MAX_DEVICES];
  static std::once_flag once[MAX_DEVICES];
  *out = nullptr;
  if (!on) return DEVIT_OK;
  int dev = 0;
  DEVIT_CHECK(hipGetDevice(&dev) == hipSuccess && dev >= 0, DEVIT_ERR_DEVICE, "devit_block_bwd: hipGetDevice");
  if (dev >= MAX_DEVICES) return DEVIT_OK;
  std::call_once(once[dev], [dev]() {
    WgradSide& s = per_dev[dev];
    if (hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking) != hipSuccess) return;
    for (auto& e : s.ev)
      if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return;
    s.ok = true;
  });
  DEVIT_CHECK(per_dev[dev].ok, DEVIT_ERR_DEVICE, "devit_block_bwd: cannot create the weight-gradient stream / its events");
  *out = &per_dev[dev];
  return DEVIT_OK;
}

int check_dims(int B, int N, int D, int Da, int Hd) {
  DEVIT_CHECK(B > 0 && N > 0 && N <= 208 && D > 0 && D % 128 == 0 && Da > 0 && Da % 128 == 0 && Hd > 0 && Hd % 128 == 0,
              DEVIT_ERR_SHAPE, "block: B=%d N=%d D=%d attn_width=%d hidden=%d not supported", B, N, D, Da, Hd);
  return DEVIT_OK;
}

int block_fwd(const Ctx& c, const devit_block_weights& w, const devit_block_acts& a) {
  const int D = c.D, Da = w.attn_width, Hd = w.hidden, H = w.num_heads;
  TRY(check_dims(c.B, c.N, D, Da, Hd));
  DEVIT_CHECK(H > 0 && Da == H * 64, DEVIT_ERR_SHAPE, "block: attn_width %d != heads %d * 64", Da, H);
  const bool save = a.flags & DEVIT_BLK_SAVE;
  const int t16 = w.dtype16;
  DEVIT_CHECK(t16 == 0 || (t16 == 1 && !save), DEVIT_ERR_ARG,
              "devit_encoder_fwd: dtype16 = %d; f16 is the frozen-teacher forward (no DEVIT_BLK_SAVE: the backward kernels are bf16)", t16);
  void* const* b = a.buf;
  DEVIT_CHECK(a.x && b[DEVIT_ACT_LN1] && b[DEVIT_ACT_QKV] && b[DEVIT_ACT_ATTN_O] && b[DEVIT_ACT_X1] && b[DEVIT_ACT_LN2] &&
                  b[DEVIT_ACT_H] && b[DEVIT_ACT_X2],
              DEVIT_ERR_ARG, "devit_encoder_fwd: null activation buffer");
  DEVIT_CHECK(!save || (b[DEVIT_ACT_MEAN1] && b[DEVIT_ACT_RSTD1] && b[DEVIT_ACT_LSE] && b[DEVIT_ACT_MEAN2] &&
                        b[DEVIT_ACT_RSTD2] && b[DEVIT_ACT_H_PRE]),
              DEVIT_ERR_ARG, "devit_encoder_fwd: DEVIT_BLK_SAVE needs the mean/rstd/lse/h_pre buffers");
  DEVIT_CHECK(!(a.flags & DEVIT_BLK_ATT) || b[DEVIT_ACT_ATT], DEVIT_ERR_ARG, "devit_encoder_fwd: DEVIT_BLK_ATT needs att");
  TRY(zero_pad(c, b[DEVIT_ACT_LN1], D, 2));
  TRY(zero_pad(c, b[DEVIT_ACT_QKV], 3 * Da, 2, (a.flags & DEVIT_BLK_QKV_PAD) ? 128 : 0));
  TRY(zero_pad(c, b[DEVIT_ACT_ATTN_O], Da, 2));
  TRY(zero_pad(c, b[DEVIT_ACT_LN2], D, 2));
  TRY(zero_pad(c, b[DEVIT_ACT_H], Hd, 2));
  if (save) TRY(zero_pad(c, b[DEVIT_ACT_H_PRE], Hd, 2));
  // ---- x1 = x + dp1 * proj(gate * attn(qkv(ln1(x))))
  TRY(devit_layernorm_fwd(a.x, c.M, D, 0, 0, w.n1w, w.n1b, c.eps, b[DEVIT_ACT_LN1], nullptr,
                          save ? (float*)b[DEVIT_ACT_MEAN1] : nullptr, save ? (float*)b[DEVIT_ACT_RSTD1] : nullptr, t16, c.stream));
  {
    devit_epilogue ep = make_ep(DEVIT_EPI_STORE_BF16, b[DEVIT_ACT_QKV], 3 * Da, c.M, t16);
    ep.bias = w.qkv_b;
    TRY(linear_fwd(c, b[DEVIT_ACT_LN1], w.qkv_w16, 3 * Da, D, ep));
  }
  TRY(devit_attn_fwd(b[DEVIT_ACT_QKV], b[DEVIT_ACT_ATTN_O], save ? (float*)b[DEVIT_ACT_LSE] : nullptr, w.head_gate, c.B, c.N, H,
                     64, 0.125f, t16, c.stream));
  {
    devit_epilogue ep = make_ep(DEVIT_EPI_RESIDUAL_F32, b[DEVIT_ACT_X1], D, c.M, t16);
    ep.bias = w.proj_b;
    ep.res = a.x;
    ep.rowscale = a.dp1;
    ep.rows_per_scale = c.N;
    ep.aux = (a.flags & DEVIT_BLK_ATT) ? b[DEVIT_ACT_ATT] : nullptr;
    TRY(linear_fwd(c, b[DEVIT_ACT_ATTN_O], w.proj_w16, D, Da, ep));
  }
  // ---- x2 = x1 + dp2 * fc2(gate * gelu(fc1(ln2(x1))))
  TRY(devit_layernorm_fwd((const float*)b[DEVIT_ACT_X1], c.M, D, 0, 0, w.n2w, w.n2b, c.eps, b[DEVIT_ACT_LN2], nullptr,
                          save ? (float*)b[DEVIT_ACT_MEAN2] : nullptr, save ? (float*)b[DEVIT_ACT_RSTD2] : nullptr, t16, c.stream));
  {
    devit_epilogue ep = make_ep(DEVIT_EPI_GELU_BF16, b[DEVIT_ACT_H], Hd, c.M, t16);
    ep.bias = w.fc1_b;
    ep.colscale = w.neuron_gate;
    ep.aux = save ? b[DEVIT_ACT_H_PRE] : nullptr;
    TRY(linear_fwd(c, b[DEVIT_ACT_LN2], w.fc1_w16, Hd, D, ep));
  }
  {
    devit_epilogue ep = make_ep(DEVIT_EPI_RESIDUAL_F32, b[DEVIT_ACT_X2], D, c.M, t16);
    ep.bias = w.fc2_b;
    ep.res = (const float*)b[DEVIT_ACT_X1];
    ep.rowscale = a.dp2;
    ep.rows_per_scale = c.N;
    // with a k-major copy of the weight the launch can take the full-row 256x384 kernel (gemm.hip: bit-identical, 126 -> ~95 us in the step)
    if (w.fc2_w16t && !t16 && devit_gemm_full_row_selected(c.Mp, D, Hd, DEVIT_EPI_RESIDUAL_F32)) {
      devit_operand A = {b[DEVIT_ACT_H], Hd, 0, 0, 0, 0}, Bo = {w.fc2_w16t, D, 1, 0, 0, 0};
      TRY(devit_gemm_bf16(&A, &Bo, c.Mp, D, Hd, 1, 1, &ep, c.stream));
    } else {
      TRY(linear_fwd(c, b[DEVIT_ACT_H], w.fc2_w16, D, Hd, ep));
    }
  }
  return DEVIT_OK;
}

}  // namespace

extern "C" int devit_block_acts_sizes(int B, int N, int D, int Da, int Hd, int flags, size_t* sizes) {
  DEVIT_CHECK(sizes != nullptr, DEVIT_ERR_ARG, "devit_block_acts_sizes: null");
  TRY(check_dims(B, N, D, Da, Hd));
  const size_t M = (size_t)B * N, Mp = pad_rows((int)M);
  const bool save = flags & DEVIT_BLK_SAVE;
  sizes[DEVIT_ACT_LN1] = Mp * D * 2;
  sizes[DEVIT_ACT_MEAN1] = sizes[DEVIT_ACT_RSTD1] = sizes[DEVIT_ACT_MEAN2] = sizes[DEVIT_ACT_RSTD2] = save ? M * 4 : 0;
  sizes[DEVIT_ACT_QKV] = (Mp + ((flags & DEVIT_BLK_QKV_PAD) ? 128 : 0)) * 3 * (size_t)Da * 2;
  sizes[DEVIT_ACT_ATTN_O] = Mp * Da * 2;
  sizes[DEVIT_ACT_LSE] = save ? (size_t)B * (Da / 64) * N * 4 : 0;
  sizes[DEVIT_ACT_X1] = M * D * 4;
  sizes[DEVIT_ACT_ATT] = (flags & DEVIT_BLK_ATT) ? M * D * 2 : 0;
  sizes[DEVIT_ACT_LN2] = Mp * D * 2;
  sizes[DEVIT_ACT_H] = Mp * Hd * 2;
  sizes[DEVIT_ACT_H_PRE] = save ? Mp * Hd * 2 : 0;
  sizes[DEVIT_ACT_X2] = M * D * 4;
  for (int i = 0; i < DEVIT_ACT_COUNT; ++i) sizes[i] = align256(sizes[i]);
  return DEVIT_OK;
}

extern "C" int devit_block_bwd_sizes(int B, int N, int D, int Da, int Hd, size_t* sizes) {
  DEVIT_CHECK(sizes != nullptr, DEVIT_ERR_ARG, "devit_block_bwd_sizes: null");
  TRY(check_dims(B, N, D, Da, Hd));
  const size_t M = (size_t)B * N, Mp = pad_rows((int)M);
  sizes[DEVIT_BWD_DH_PRE] = Mp * Hd * 2;
  sizes[DEVIT_BWD_DLN2] = sizes[DEVIT_BWD_G1] = sizes[DEVIT_BWD_DLN1] = Mp * D * 2;
  sizes[DEVIT_BWD_DATTN] = Mp * (size_t)Da * 2;
  sizes[DEVIT_BWD_DX1] = M * D * 4;
  sizes[DEVIT_BWD_DQKV] = Mp * 3 * (size_t)Da * 2;
  sizes[DEVIT_BWD_LNWS] = devit_layernorm_bwd_workspace((int)M, D);
  for (int i = 0; i < DEVIT_BWD_COUNT; ++i) sizes[i] = align256(sizes[i]);
  return DEVIT_OK;
}

extern "C" int devit_encoder_fwd(int nblocks, const devit_block_weights* w, const devit_block_acts* acts, int B, int N, int D,
                                 float eps, void* stream) {
  DEVIT_CHECK(nblocks > 0 && w && acts, DEVIT_ERR_ARG, "devit_encoder_fwd: bad argument");
  Ctx c{B * N, pad_rows(B * N), B, N, D, eps, stream};
  for (int i = 0; i < nblocks; ++i) {
    DEVIT_CHECK(i == 0 || acts[i].x == (const float*)acts[i - 1].buf[DEVIT_ACT_X2], DEVIT_ERR_ARG,
                "devit_encoder_fwd: block %d does not read block %d's output", i, i - 1);
    TRY(block_fwd(c, w[i], acts[i]));
  }
  return DEVIT_OK;
}

extern "C" int devit_block_bwd(const devit_block_weights* wp, const devit_block_acts* ap, const devit_block_wgrads* gp,
                               const devit_block_bwd_io* io, int B, int N, int D, float eps, void* stream) {
  DEVIT_CHECK(wp && ap && gp && io, DEVIT_ERR_ARG, "devit_block_bwd: null argument");
  const devit_block_weights& w = *wp;
  const devit_block_acts& a = *ap;
  const devit_block_wgrads& g = *gp;
  const int Hd = w.hidden, H = w.num_heads, Da = w.attn_width;   // Da = H * 64 (< D when heads were compacted away)
  TRY(check_dims(B, N, D, Da, Hd));
  DEVIT_CHECK(H > 0 && Da == H * 64, DEVIT_ERR_SHAPE, "devit_block_bwd: attn_width %d != heads %d * 64", Da, H);
  DEVIT_CHECK(a.flags & DEVIT_BLK_SAVE, DEVIT_ERR_ARG, "devit_block_bwd: the forward ran without DEVIT_BLK_SAVE");
  DEVIT_CHECK(w.dtype16 == 0, DEVIT_ERR_ARG, "devit_block_bwd: f16 blocks have no backward");
  DEVIT_CHECK(io->dx && io->g2 && io->dx_in, DEVIT_ERR_ARG, "devit_block_bwd: dx / g2 / dx_in");
  for (int i = 0; i < DEVIT_BWD_COUNT; ++i) DEVIT_CHECK(io->ws[i], DEVIT_ERR_ARG, "devit_block_bwd: workspace %d is null", i);
  DEVIT_CHECK(g.n1w && g.n1b && g.qkv_w && g.qkv_b && g.proj_w && g.proj_b && g.n2w && g.n2b && g.fc1_w && g.fc1_b &&
                  g.fc2_w && g.fc2_b, DEVIT_ERR_ARG, "devit_block_bwd: null gradient accumulator");
  Ctx c{B * N, pad_rows(B * N), B, N, D, eps, stream};
  // The four weight-gradient launches go to a side stream of the library's own, each behind an event of the kernel that produces
  // its operand; `stream` waits for them before this call returns its last kernel's successor (so callers see ONE stream, as
  // before).  They depend on nothing downstream, and beside the dgrad chain their workgroups fill its kernels' partial last
  // rounds (the N = 384 dgrads run 2.32 rounds of 512 workgroups): 10233 -> 10323 img/s, three interleaved pairs on one box
  // (profiles/r04_h_wgrad_side_stream_ab.txt).  DEVIT_WGRAD_STREAM=0 keeps them on `stream`.
  WgradSide* sd = nullptr;
  TRY(wgrad_side(&sd));
  Ctx cs = c;
  bool forked = false;
  auto fork = [&](int i) -> int {
    if (!sd) return DEVIT_OK;
    DEVIT_CHECK(hipEventRecord(sd->ev[i], (hipStream_t)stream) == hipSuccess && hipStreamWaitEvent(sd->stream, sd->ev[i], 0) == hipSuccess,
                DEVIT_ERR_LAUNCH, "devit_block_bwd: cannot fork the weight-gradient stream");
    cs.stream = sd->stream;
    forked = true;
    return DEVIT_OK;
  };
  // (the body is a lambda so that an error return after a fork still joins the side stream below: the caller's stream never runs ahead
  // of launches it does not know about)
  const int rc = [&]() -> int {
  void* const* b = a.buf;
  void* dh_pre = io->ws[DEVIT_BWD_DH_PRE];
  void* dln2 = io->ws[DEVIT_BWD_DLN2];
  float* dx1 = (float*)io->ws[DEVIT_BWD_DX1];
  void* g1 = io->ws[DEVIT_BWD_G1];
  void* dattn = io->ws[DEVIT_BWD_DATTN];
  void* dqkv = io->ws[DEVIT_BWD_DQKV];
  void* dln1 = io->ws[DEVIT_BWD_DLN1];
  TRY(zero_pad(c, dh_pre, Hd, 2));
  TRY(zero_pad(c, dln2, D, 2));
  TRY(zero_pad(c, g1, D, 2));
  TRY(zero_pad(c, dattn, Da, 2));
  TRY(zero_pad(c, dqkv, 3 * Da, 2));
  TRY(zero_pad(c, dln1, D, 2));
  TRY(zero_pad(c, io->g_prev, D, 2));
  // ---- MLP branch: x2 = x1 + dp2 * fc2(gate * gelu(fc1(ln2))).  The weight gradient that only needs g2 first, then
  // dh_pre's producer and its consumers back to back (dh_pre is 156 MB at B = 256: keep it in the Infinity Cache)
  // Weight gradients: jobs of the grouped full-row launch where the shapes allow (D == 384), else -- and for a product whose bias gradient
  // must come from its B side (the top block's fc2) -- the split-K launch on 128x128 tiles, behind the kernel that produces its operand.
  const int wmode = (D == 384 && c.Mp / 64 >= 3) ? wgrad_mode() : 0;
  if (io->defer_count) *io->defer_count = 0;
  devit_wgrad_job jobs[4];
  int nj = 0;
  auto wgrad = [&](int ev, const void* dy, const void* x, float* w_grad, float* b_grad, int Nw, int K) -> int {
    if (wmode && wgrad_job(&jobs[nj], dy, x, w_grad, b_grad, Nw, K)) {
      ++nj;
      return DEVIT_OK;
    }
    TRY(fork(ev));
    return linear_wgrad(cs, dy, x, w_grad, b_grad, Nw, K);
  };
  auto flush = [&](int ev) -> int {
    if (io->defer_jobs) {           // the caller launches them later, with other blocks' (devit_block_bwd_io.defer_jobs)
      for (int i = 0; i < nj; ++i) io->defer_jobs[i] = jobs[i];
      if (io->defer_count) *io->defer_count = nj;
      nj = 0;
      return DEVIT_OK;
    }
    if (nj == 0) return DEVIT_OK;
    TRY(fork(ev));
    const int n = nj;
    nj = 0;
    return devit_wgrad_grouped(jobs, n, c.Mp, 0, cs.stream);
  };
  // fc2's product is taken transposed (its 384 outputs are the B side): its bias gradient cannot come out of the grouped kernel.  Every block but the
  // topmost got it from the LN1 backward of the block above; the topmost takes a column-sum pass over g2 (the LayerNorm workspace is idle here).
  bool fc2_bias_here = !io->g2_bias_done;
  if (fc2_bias_here && wmode && Hd % 128 == 0 && io->lnws_bytes >= devit_colsum_workspace(c.Mp, D)) {
    TRY(devit_colsum_bf16(io->g2, c.Mp, D, D, 0, 0, g.fc2_b, 1, io->ws[DEVIT_BWD_LNWS], io->lnws_bytes, stream));
    fc2_bias_here = false;
  }
  TRY(wgrad(0, io->g2, b[DEVIT_ACT_H], g.fc2_w, fc2_bias_here ? g.fc2_b : nullptr, D, Hd));
  {
    devit_epilogue ep = make_ep(DEVIT_EPI_DGELU_BF16, dh_pre, Hd, c.M);
    ep.colscale = w.neuron_gate;
    ep.aux_in = b[DEVIT_ACT_H_PRE];
    TRY(linear_dgrad(c, io->g2, w.fc2_w16, D, Hd, ep));
  }
  TRY(linear_dgrad(c, dh_pre, w.fc1_w16, Hd, D, make_ep(DEVIT_EPI_STORE_BF16, dln2, D, c.M)));
  TRY(wgrad(1, dh_pre, b[DEVIT_ACT_LN2], g.fc1_w, g.fc1_b, Hd, D));
  // LN2 backward: dx1 = dx + LN'(dln2); g1 = bf16(dp1 * dx1); its column sums = the proj bias gradient
  TRY(devit_layernorm_bwd(dln2, 0, (const float*)b[DEVIT_ACT_X1], c.M, D, 0, 0, (const float*)b[DEVIT_ACT_MEAN2],
                          (const float*)b[DEVIT_ACT_RSTD2], w.n2w, io->dx, dx1, g1, a.dp1, N, g.n2w, g.n2b, g.proj_b, 1,
                          io->ws[DEVIT_BWD_LNWS], io->lnws_bytes, stream));
  // ---- attention branch: x1 = x + dp1 * proj(gate * attn(qkv(ln1)))
  TRY(linear_dgrad(c, g1, w.proj_w16, D, Da, make_ep(DEVIT_EPI_STORE_BF16, dattn, Da, c.M)));
  TRY(wgrad(2, g1, b[DEVIT_ACT_ATTN_O], g.proj_w, nullptr, D, Da));
  TRY(devit_attn_bwd(b[DEVIT_ACT_QKV], b[DEVIT_ACT_ATTN_O], dattn, (const float*)b[DEVIT_ACT_LSE], w.head_gate, io->dqkv_add,
                     dqkv, B, N, H, 64, 0.125f, stream));
  TRY(linear_dgrad(c, dqkv, w.qkv_w16, 3 * Da, D, make_ep(DEVIT_EPI_STORE_BF16, dln1, D, c.M)));
  TRY(wgrad(3, dqkv, b[DEVIT_ACT_LN1], g.qkv_w, g.qkv_b, 3 * Da, D));
  TRY(flush(3));
  // LN1 backward: dx_in = dx1 + LN'(dln1); g_prev = bf16(prev_dp2 * dx_in) (+ the block below's fc2 bias gradient)
  TRY(devit_layernorm_bwd(dln1, 0, a.x, c.M, D, 0, 0, (const float*)b[DEVIT_ACT_MEAN1], (const float*)b[DEVIT_ACT_RSTD1],
                          w.n1w, dx1, io->dx_in, io->g_prev, io->prev_dp2, N, g.n1w, g.n1b,
                          io->g_prev ? io->prev_fc2_b_grad : nullptr, 1, io->ws[DEVIT_BWD_LNWS], io->lnws_bytes, stream));
  return DEVIT_OK;
  }();
  if (sd && forked) {
    const bool joined = hipEventRecord(sd->ev[4], sd->stream) == hipSuccess && hipStreamWaitEvent((hipStream_t)stream, sd->ev[4], 0) == hipSuccess;
    if (rc == DEVIT_OK) DEVIT_CHECK(joined, DEVIT_ERR_LAUNCH, "devit_block_bwd: cannot join the weight-gradient stream");
  }
  return rc;
}
