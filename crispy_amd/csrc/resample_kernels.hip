// resample_kernels.hip -- 48 kHz -> 16 kHz between the denoiser and the ASR front end.
//
// Replaces rubato `FftFixedIn::<f32>::new(48000, 16000, 1024, 1, 1)` as driven by
// src-tauri/src/commands/transcription.rs:198-208, 314-357, and optionally the s16 WAV hand-off
// (recording.rs:101-118 write, commands/transcription.rs:306-313 read) that sits in front of it.
// [UPSTREAM-RECALL] semantics: oracle/resample_oracle.py.
//
// The FFT resampler is a fixed linear map per 1026-sample block: y[n] = sum_j g[(3n - j) mod 2052] x[j],
// n = 0..683 (first 342 samples + 342 of overlap for the next block).  Instead of radix-19 FFTs the map
// is applied as a GEMM against its circulant matrix on the f32 matrix cores:
//   rs_prep_kernel   blocks the audio into rows of 1040 floats (zero padded), applying the scale and
//                    the optional s16 truncation of the WAV hand-off
//   gemm_f32_nt      [blocks x 1040] . W[684 x 1040]^T
//   rs_ola_kernel    overlap-add of consecutive blocks into the 16 kHz stream
#include "asr_common.h"

namespace crispy {
namespace {

__global__ __launch_bounds__(256) void rs_prep_kernel(const float* __restrict__ in, long in_stride, long n_in,
                                                      float scale, int wav_s16, float* __restrict__ A, int n_blk) {
  const long row = blockIdx.x;            // b * n_blk + blk
  const int b = (int)(row / n_blk), blk = (int)(row % n_blk);
  const float* x = in + (long)b * in_stride + (long)blk * RS_FFT_IN;
  float* a = A + row * RS_K;
  for (int j = threadIdx.x; j < RS_K; j += 256) {
    float v = 0.f;
    if (j < RS_FFT_IN && (long)blk * RS_FFT_IN + j < n_in) {
      v = x[j] * scale;
      if (wav_s16 >= 1) v = fminf(1.f, fmaxf(-1.f, v));          // adapter clamp (audio.rs:272)
      if (wav_s16 >= 2) v = truncf(v * 32767.f) / 32768.f;        // WavWriter s16 truncation, read back /32768
    }
    a[j] = v;
  }
}

__global__ __launch_bounds__(256) void rs_ola_kernel(const float* __restrict__ Y, float* __restrict__ out,
                                                     long out_stride, int n_blk) {
  const long row = blockIdx.x;
  const int b = (int)(row / n_blk), blk = (int)(row % n_blk);
  const float* y = Y + row * RS_N;
  float* o = out + (long)b * out_stride + (long)blk * RS_FFT_OUT;
  for (int n = threadIdx.x; n < RS_FFT_OUT; n += 256) {
    float v = y[n];
    if (blk > 0) v += y[n - RS_N + RS_FFT_OUT];   // second half of the previous block's row
    o[n] = v;
  }
}

// hi | lo planes of the scaled input (rs_prep_kernel's value, split): plane p, stream b, block blk + 1 (block 0 stays zero),
// RS_PITCH halves per block with the 30 behind the 1026 samples zero.  x_lo = f16(x - f32(x_hi)): exact difference, rounded
// once (absolute error <= 2^-25 where it is a subnormal) -- the pair carries x to ~3e-8 of full scale.
__global__ __launch_bounds__(256) void rs_prep_split_kernel(const float* __restrict__ in, long in_stride, long n_in, float scale,
                                                            int wav_s16, _Float16* __restrict__ planes, long plane_stride, int n_blk,
                                                            long n_slots) {
  // a thread: two consecutive samples of one block slot -> one 4-byte store per plane (consecutive lanes, consecutive
  // addresses on both sides; eight samples per thread made every load a 32-byte-strided gather and was no faster than
  // one sample per thread)
  typedef _Float16 half2v __attribute__((ext_vector_type(2)));
  constexpr int PER = RS_PITCH / 2;       // 528 threads' worth per slot
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  const long row = idx / PER;             // b * (n_blk + 1) + block slot
  if (row >= n_slots) return;
  const int j0 = (int)(idx % PER) * 2;
  const int b = (int)(row / (n_blk + 1)), blk = (int)(row % (n_blk + 1)) - 1;
  const float* x = in + (long)b * in_stride + (long)blk * RS_FFT_IN;
  half2v hi, lo;
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    const int j = j0 + e;
    float v = 0.f;
    if (blk >= 0 && j < RS_FFT_IN && (long)blk * RS_FFT_IN + j < n_in) {
      v = x[j] * scale;
      if (wav_s16 >= 1) v = fminf(1.f, fmaxf(-1.f, v));
      if (wav_s16 >= 2) v = truncf(v * 32767.f) / 32768.f;
    }
    const _Float16 h = (_Float16)v;
    hi[e] = h;
    lo[e] = (_Float16)(v - (float)h);
  }
  *reinterpret_cast<half2v*>(planes + row * RS_PITCH + j0) = hi;
  *reinterpret_cast<half2v*>(planes + plane_stride + row * RS_PITCH + j0) = lo;
}

}  // namespace

hipError_t rs_prep_split(const float* in, long in_stride, long n_in, float scale, int wav_s16, void* planes, int batch, int n_blk,
                         hipStream_t s) {
  const long plane_stride = (long)batch * (n_blk + 1) * RS_PITCH;
  const long n_slots = (long)batch * (n_blk + 1), n_thr = n_slots * (RS_PITCH / 2);
  hipLaunchKernelGGL(rs_prep_split_kernel, dim3((unsigned)((n_thr + 255) / 256)), dim3(256), 0, s, in, in_stride, n_in, scale,
                     wav_s16, reinterpret_cast<_Float16*>(planes), plane_stride, n_blk, n_slots);
  return hipGetLastError();
}

hipError_t rs_prep(const float* in, long in_stride, long n_in, float scale, int wav_s16, float* A, int batch,
                   int n_blk, hipStream_t s) {
  hipLaunchKernelGGL(rs_prep_kernel, dim3((unsigned)((long)batch * n_blk)), dim3(256), 0, s, in, in_stride, n_in, scale,
                     wav_s16, A, n_blk);
  return hipGetLastError();
}
hipError_t rs_ola(const float* Y, float* out, long out_stride, int batch, int n_blk, hipStream_t s) {
  hipLaunchKernelGGL(rs_ola_kernel, dim3((unsigned)((long)batch * n_blk)), dim3(256), 0, s, Y, out, out_stride, n_blk);
  return hipGetLastError();
}

}  // namespace crispy
