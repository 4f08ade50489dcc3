// asr_quant.h -- ggml block-quantised tensors read in place on the device.
//
// The reference's catalog ships quantised Whisper files (src-tauri/src/managers/model.rs:99 whisper-medium-q4_1.bin,
// :137 ggml-large-v3-q5_0.bin).  `crispy_asr_load_resident` keeps their 2-D tensors in HBM exactly as the file holds
// them -- rows of 32-weight blocks [UPSTREAM-RECALL ggml-quants: q4_0 {f16 d, 16 B nibbles}, q4_1 {f16 d, f16 m, 16 B},
// q5_0 {f16 d, u32 high bits, 16 B}, q5_1 {f16 d, f16 m, u32, 16 B}, q8_0 {f16 d, 32 x int8}; low nibbles are elements
// 0..15 of a block, high nibbles 16..31] -- and de-quantises at the point of use.  Every value is computed with the
// loader's operations in the loader's order (int -> float, one multiply, one add, each rounded on its own), so a
// resident model equals the model built from the de-quantised weights bit for bit.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace crispy {

constexpr int QT_F32 = 0, QT_Q4_0 = 2, QT_Q4_1 = 3, QT_Q5_0 = 6, QT_Q5_1 = 7, QT_Q8_0 = 8;

__host__ __device__ constexpr int quant_block_bytes(int ttype) {
  return ttype == QT_Q4_0 ? 18 : ttype == QT_Q4_1 ? 20 : ttype == QT_Q5_0 ? 22 : ttype == QT_Q5_1 ? 24 : ttype == QT_Q8_0 ? 34 : 128;
}

__device__ __forceinline__ float q_f16(const unsigned char* p) {       // blocks are only 2-byte aligned
  unsigned short v;
  __builtin_memcpy(&v, p, 2);
  _Float16 h;
  __builtin_memcpy(&h, &v, 2);
  return (float)h;
}
__device__ __forceinline__ unsigned q_u32(const unsigned char* p) {
  unsigned v;
  __builtin_memcpy(&v, p, 4);                                          // global memory takes unaligned dword loads
  return v;
}

// the 32 weights of block `b` (TT compile-time), y[j] for j < 16 from the low nibbles, y[16 + j] from the high ones
template <int TT>
__device__ __forceinline__ void q_block(const unsigned char* b, float (&y)[32]) {
#pragma clang fp contract(off)
  if (TT == QT_F32) {
#pragma unroll
    for (int j = 0; j < 32; ++j) y[j] = reinterpret_cast<const float*>(b)[j];
    return;
  }
  const float d = q_f16(b);
  if (TT == QT_Q8_0) {
#pragma unroll
    for (int w = 0; w < 8; ++w) {
      const unsigned v = q_u32(b + 2 + 4 * w);
#pragma unroll
      for (int e = 0; e < 4; ++e) y[4 * w + e] = (float)(int)(signed char)((v >> (8 * e)) & 0xff) * d;
    }
    return;
  }
  const bool has_m = TT == QT_Q4_1 || TT == QT_Q5_1, has_h = TT == QT_Q5_0 || TT == QT_Q5_1;
  const float m = has_m ? q_f16(b + 2) : 0.f;
  const int off_h = has_m ? 4 : 2;
  const unsigned qh = has_h ? q_u32(b + off_h) : 0u;
  const unsigned char* qs = b + off_h + (has_h ? 4 : 0);
#pragma unroll
  for (int w = 0; w < 4; ++w) {
    const unsigned v = q_u32(qs + 4 * w);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int j = 4 * w + e;
      int x0 = (int)((v >> (8 * e)) & 0x0f), x1 = (int)((v >> (8 * e + 4)) & 0x0f);
      if (has_h) { x0 |= (int)((qh >> j) & 1u) << 4; x1 |= (int)((qh >> (j + 16)) & 1u) << 4; }
      if (TT == QT_Q4_0) { y[j] = (float)(x0 - 8) * d; y[j + 16] = (float)(x1 - 8) * d; }
      else if (TT == QT_Q5_0) { y[j] = (float)(x0 - 16) * d; y[j + 16] = (float)(x1 - 16) * d; }
      else { y[j] = (float)x0 * d + m; y[j + 16] = (float)x1 * d + m; }
    }
  }
}

// one element of a row-major quantised tensor (embedding gathers: a few hundred values per token)
__device__ __forceinline__ float q_elem(const unsigned char* base, int ttype, long idx) {
#pragma clang fp contract(off)
  if (ttype == QT_F32) return reinterpret_cast<const float*>(base)[idx];
  const unsigned char* b = base + (idx >> 5) * quant_block_bytes(ttype);
  const int j = (int)(idx & 31);
  const float d = q_f16(b);
  if (ttype == QT_Q8_0) return (float)(int)(signed char)b[2 + j] * d;
  const bool has_m = ttype == QT_Q4_1 || ttype == QT_Q5_1, has_h = ttype == QT_Q5_0 || ttype == QT_Q5_1;
  const int off_h = has_m ? 4 : 2;
  const unsigned char* qs = b + off_h + (has_h ? 4 : 0);
  int x = j < 16 ? (qs[j] & 0x0f) : (qs[j - 16] >> 4);
  if (has_h) x |= (int)((q_u32(b + off_h) >> j) & 1u) << 4;
  if (ttype == QT_Q4_0) return (float)(x - 8) * d;
  if (ttype == QT_Q5_0) return (float)(x - 16) * d;
  return (float)x * d + q_f16(b + 2);
}

// dst[i * 32 + j] (f16 or f32, row-major, contiguous) = block i of q (optionally times gamma[k], k = column of the
// element in rows of `cols` weights: W' = W . diag(gamma), the LayerNorm fold of the decode-step projections)
hipError_t dequant_blocks(const void* q, int ttype, long n_blocks, int cols, void* dst, int dst_f16, const float* gamma,
                          hipStream_t s);
// x[b][:] = E[tokens[b]][:] + pos_emb[pos][:] with E a resident quantised tensor
hipError_t embed_tokens_q(const int* tokens, const void* q_emb, int ttype, const float* pos_emb, int pos, const int* pos_dev,
                          float* x, int B, int D, hipStream_t s, int rows_per_clip = 1, const int* row_off = nullptr);

}  // namespace crispy
