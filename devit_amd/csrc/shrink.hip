// Index copies for TRAINING through compacted blocks (devit_amd/shrink.py, shrink.compact(trainable=True)): every step the
// kept rows / columns of the masters' bf16 copies are re-gathered into the compact GEMM weights, and after each block's
// backward the compact weight gradients are added into the kept rows / columns of the masters' gradients
// (distill_sub.py:384-401 trains the gated student; core/imp_rank.py only masks).  A model is ~70 such small copies per
// step: one launch works through a table of them.
#include "devit_common.h"

namespace {

// The compact side of a job is dense [rows][cols]; idx maps a compact row (modes 0, 2) or column (modes 1, 3) to the
// master's row / column, or -1 for a padding unit (compacted widths are padded to the GEMM granules with all-zero units).
__global__ __launch_bounds__(256) void index_copy_kernel(const devit_index_job* jobs) {
  const devit_index_job j = jobs[blockIdx.y];
  if (j.mode == 4) {   // 16-bit transpose dst[c][r] = src[r][c]: the k-major copy of a Linear weight (full-row GEMM, gemm.hip); 64 x 64 tiles through LDS
    __shared__ unsigned short tile[64][66];
    const int tr = (j.rows + 63) / 64, tc = (j.cols + 63) / 64;
    const unsigned short* src = (const unsigned short*)j.src;
    unsigned short* dst = (unsigned short*)j.dst;
    const bool pairs = ((j.rows | j.cols | j.src_ld | j.dst_ld) & 1) == 0 && (((size_t)j.src | (size_t)j.dst) & 3) == 0;   // 4-byte accesses
    for (int t = blockIdx.x; t < tr * tc; t += gridDim.x) {
      const int r0 = (t / tc) * 64, c0 = (t % tc) * 64;
      if (pairs) {
        for (int e = threadIdx.x; e < 2048; e += 256) {
          const int r = e >> 5, c = (e & 31) * 2;
          if (r0 + r < j.rows && c0 + c < j.cols) {
            const unsigned v = *(const unsigned*)(src + (long long)(r0 + r) * j.src_ld + c0 + c);
            tile[r][c] = (unsigned short)v;
            tile[r][c + 1] = (unsigned short)(v >> 16);
          }
        }
        __syncthreads();
        for (int e = threadIdx.x; e < 2048; e += 256) {
          const int c = e >> 5, r = (e & 31) * 2;
          if (r0 + r < j.rows && c0 + c < j.cols)
            *(unsigned*)(dst + (long long)(c0 + c) * j.dst_ld + r0 + r) = (unsigned)tile[r][c] | ((unsigned)tile[r + 1][c] << 16);
        }
        __syncthreads();
        continue;
      }
      for (int e = threadIdx.x; e < 4096; e += 256) {
        const int r = e >> 6, c = e & 63;
        if (r0 + r < j.rows && c0 + c < j.cols) tile[r][c] = src[(long long)(r0 + r) * j.src_ld + c0 + c];
      }
      __syncthreads();
      for (int e = threadIdx.x; e < 4096; e += 256) {
        const int c = e >> 6, r = e & 63;
        if (r0 + r < j.rows && c0 + c < j.cols) dst[(long long)(c0 + c) * j.dst_ld + r0 + r] = tile[r][c];
      }
      __syncthreads();
    }
    return;
  }
  const long long total = (long long)j.rows * j.cols;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
    const int r = (int)(e / j.cols), c = (int)(e - (long long)r * j.cols);
    const int k = j.idx[(j.mode & 1) ? c : r];
    const long long compact = (long long)r * ((j.mode & 2) ? j.src_ld : j.dst_ld) + c;
    if (j.mode & 2) {                              // add, and leave the compact accumulator zeroed for the next backward
      float* acc = (float*)j.src + compact;
      const float v = *acc;
      *acc = 0.f;
      if (k >= 0) ((float*)j.dst)[(j.mode & 1) ? (long long)r * j.dst_ld + k : (long long)k * j.dst_ld + c] += v;   // kept indices are distinct
      continue;
    }
    if (k < 0) continue;
    const long long master = (j.mode & 1) ? (long long)r * j.src_ld + k : (long long)k * j.src_ld + c;
    if (j.elem == 2) {
      ((unsigned short*)j.dst)[compact] = ((const unsigned short*)j.src)[master];
    } else {
      ((float*)j.dst)[compact] = ((const float*)j.src)[master];
    }
  }
}

}  // namespace

extern "C" int devit_index_copy(const devit_index_job* jobs_device, int njobs, int blocks_per_job, void* stream) {
  DEVIT_CHECK(jobs_device && njobs > 0 && njobs <= 65535 && blocks_per_job > 0 && blocks_per_job <= 4096, DEVIT_ERR_ARG,
              "devit_index_copy: bad argument (njobs %d, blocks_per_job %d)", njobs, blocks_per_job);
  hipLaunchKernelGGL(index_copy_kernel, dim3(blocks_per_job, njobs), dim3(256), 0, (hipStream_t)stream, jobs_device);
  DEVIT_LAUNCH_CHECK();
  return DEVIT_OK;
}
