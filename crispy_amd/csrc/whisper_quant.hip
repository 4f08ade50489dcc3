// whisper_quant.hip -- de-quantisation of resident ggml blocks at the point of use (asr_quant.h).
#include "asr_quant.h"

namespace crispy {
namespace {

typedef _Float16 q_half8 __attribute__((ext_vector_type(8)));
typedef float q_f4 __attribute__((ext_vector_type(4)));

// One thread per block of 32 weights: consecutive threads read consecutive blocks (18 - 34 bytes each, contiguous) and
// write consecutive 64- or 128-byte runs.  The pass is a stream: q bytes in, 2 or 4 bytes per weight out.
template <int TT, bool F16OUT, bool GAMMA>
__global__ __launch_bounds__(256) void dequant_kernel(const unsigned char* __restrict__ q, long n_blocks, int blocks_per_row,
                                                      void* __restrict__ dst, const float* __restrict__ gamma) {
#pragma clang fp contract(off)
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n_blocks) return;
  float y[32];
  q_block<TT>(q + i * quant_block_bytes(TT), y);
  if (GAMMA) {
    const float* g = gamma + (i % blocks_per_row) * 32;
#pragma unroll
    for (int j = 0; j < 32; ++j) y[j] = y[j] * g[j];
  }
  if (F16OUT) {
    q_half8* o = reinterpret_cast<q_half8*>(reinterpret_cast<_Float16*>(dst) + i * 32);
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      q_half8 v;
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = (_Float16)y[8 * w + e];
      o[w] = v;
    }
  } else {
    q_f4* o = reinterpret_cast<q_f4*>(reinterpret_cast<float*>(dst) + i * 32);
#pragma unroll
    for (int w = 0; w < 8; ++w) o[w] = q_f4{y[4 * w], y[4 * w + 1], y[4 * w + 2], y[4 * w + 3]};
  }
}

__global__ __launch_bounds__(256) void embed_q_kernel(const int* __restrict__ tokens, const unsigned char* __restrict__ q_emb,
                                                      int ttype, const float* __restrict__ pos_emb, int pos,
                                                      const int* __restrict__ pos_dev, float* __restrict__ x, int D, int rpc,
                                                      const int* __restrict__ row_off) {
  const int b = blockIdx.x;
  const int tok = tokens[b];
  if (pos_dev) pos = *pos_dev;
  pos += b % rpc;
  if (row_off) pos = max(pos - row_off[b / rpc], 0);
  for (int c = threadIdx.x; c < D; c += 256) x[(long)b * D + c] = q_elem(q_emb, ttype, (long)tok * D + c) + pos_emb[(long)pos * D + c];
}

template <int TT>
hipError_t dq_launch(const void* q, long n_blocks, int cols, void* dst, int dst_f16, const float* gamma, hipStream_t s) {
  const dim3 grid((unsigned)((n_blocks + 255) / 256)), block(256);
  const unsigned char* qb = reinterpret_cast<const unsigned char*>(q);
  const int bpr = cols / 32;
  if (dst_f16) {
    if (gamma) return hipErrorInvalidValue;      // the folded projections stay f32 (NOTEBOOK.md section 4)
    hipLaunchKernelGGL((dequant_kernel<TT, true, false>), grid, block, 0, s, qb, n_blocks, bpr, dst, gamma);
  } else if (gamma) {
    hipLaunchKernelGGL((dequant_kernel<TT, false, true>), grid, block, 0, s, qb, n_blocks, bpr, dst, gamma);
  } else {
    hipLaunchKernelGGL((dequant_kernel<TT, false, false>), grid, block, 0, s, qb, n_blocks, bpr, dst, gamma);
  }
  return hipGetLastError();
}

}  // namespace

hipError_t dequant_blocks(const void* q, int ttype, long n_blocks, int cols, void* dst, int dst_f16, const float* gamma,
                          hipStream_t s) {
  if (n_blocks <= 0) return hipSuccess;
  if (cols <= 0 || cols % 32) return hipErrorInvalidValue;
  switch (ttype) {
    case QT_F32: return dq_launch<QT_F32>(q, n_blocks, cols, dst, dst_f16, gamma, s);
    case QT_Q4_0: return dq_launch<QT_Q4_0>(q, n_blocks, cols, dst, dst_f16, gamma, s);
    case QT_Q4_1: return dq_launch<QT_Q4_1>(q, n_blocks, cols, dst, dst_f16, gamma, s);
    case QT_Q5_0: return dq_launch<QT_Q5_0>(q, n_blocks, cols, dst, dst_f16, gamma, s);
    case QT_Q5_1: return dq_launch<QT_Q5_1>(q, n_blocks, cols, dst, dst_f16, gamma, s);
    case QT_Q8_0: return dq_launch<QT_Q8_0>(q, n_blocks, cols, dst, dst_f16, gamma, s);
    default: return hipErrorInvalidValue;
  }
}

hipError_t embed_tokens_q(const int* tokens, const void* q_emb, int ttype, const float* pos_emb, int pos, const int* pos_dev,
                          float* x, int B, int D, hipStream_t s, int rows_per_clip, const int* row_off) {
  hipLaunchKernelGGL(embed_q_kernel, dim3(B), dim3(256), 0, s, tokens, reinterpret_cast<const unsigned char*>(q_emb), ttype,
                     pos_emb, pos, pos_dev, x, D, rows_per_clip < 1 ? 1 : rows_per_clip, row_off);
  return hipGetLastError();
}

}  // namespace crispy
