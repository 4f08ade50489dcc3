// mel_kernels.hip -- Whisper log-mel front end for MI355X (gfx950).
//
// Replaces whisper.cpp `log_mel_spectrogram` behind transcribe_rs::SpeechModel::transcribe
// (reference call sites src-tauri/src/managers/transcription.rs:183-185, 213-215).
// Semantics: SURVEY.md Appendix B.1 / oracle/logmel_oracle.c.
//
// HBM-bound stage: per 30 s clip 1.92 MB of PCM in, 0.96 MB of mel out.
//   mel_frames_kernel  grid (47 tiles, batch); 4 waves per workgroup, each wave transforms 16
//                      consecutive frames: Hann window, 400-point real FFT as a 200-point complex
//                      Stockham FFT in LDS (radices 4.2.5.5), power spectrum, sparse triangular mel
//                      filters (double accumulation as upstream), log10.  The [n_mel][64] tile is
//                      staged in LDS and written as 256-byte rows; the per-clip maximum is folded
//                      into one atomicMax per wave.
//   mel_finish_kernel  clamp to (clip max - 8), (x + 4) / 4; also emits the frame-major, zero
//                      padded copy [3002][n_mel] the encoder's first convolution reads as a GEMM.
#include "asr_common.h"
#include "fft_lds.h"

namespace crispy {
namespace {

using namespace fftx;

__device__ __forceinline__ int float_order_key(float f) {
  const int b = __float_as_int(f);
  return b >= 0 ? b : (b ^ 0x7fffffff);
}
__device__ __forceinline__ float float_from_key(int k) {
  return __int_as_float(k >= 0 ? k : (k ^ 0x7fffffff));
}

__global__ __launch_bounds__(256) void mel_frames_kernel(MelArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int b = blockIdx.y;
  const int t0 = blockIdx.x * MEL_TILE;
  float2* buf = reinterpret_cast<float2*>(smem) + wave * 208;                       // 201 used
  float* pw = reinterpret_cast<float*>(smem + 4 * 208 * sizeof(float2)) + wave * 208;  // 201 used
  float* tile = reinterpret_cast<float*>(smem + 4 * 208 * (sizeof(float2) + sizeof(float)));  // [n_mel][65]
  const float* __restrict__ x = a.pcm + (long)b * a.pcm_stride;
  const int n = a.n_samples[b];
  const MelTables* __restrict__ tab = a.tab;
  float wmax = -1e30f;

  for (int fi = 0; fi < 16; ++fi) {
    const int f = wave * 16 + fi;
    const int t = t0 + f;
    // window + pack: z[m] = (h[2m] x[2m], h[2m+1] x[2m+1]); reflect at the start, zeros past the end
    for (int m = lane; m < 200; m += 64) {
      float v[2];
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        int s = t * 160 + 2 * m + q - 200;
        if (s < 0) s = -s;
        v[q] = s < n ? x[s] * tab->hann[2 * m + q] : 0.f;
      }
      buf[m] = make_float2(v[0], v[1]);
    }
    wave_lds_sync();
    pass<200, 4, 1, 400>(buf, tab->w400, lane);
    pass<200, 2, 4, 400>(buf, tab->w400, lane);
    pass<200, 5, 8, 400>(buf, tab->w400, lane);
    pass<200, 5, 40, 400>(buf, tab->w400, lane);
    // real post-processing of the pair (k, 200-k) and power spectrum
    for (int k = lane; k <= 100; k += 64) {
      const float2 zk = buf[k];
      const float2 zn = (k == 0) ? zk : buf[200 - k];
      float2 zc = cconj(zn);
      float2 fe = make_float2(0.5f * (zk.x + zc.x), 0.5f * (zk.y + zc.y));
      float2 d = make_float2(0.5f * (zk.x - zc.x), 0.5f * (zk.y - zc.y));
      float2 tt = cmul(tab->w400[k], make_float2(d.y, -d.x));
      const float2 xk = cadd(fe, tt);
      zc = cconj(zk);
      fe = make_float2(0.5f * (zn.x + zc.x), 0.5f * (zn.y + zc.y));
      d = make_float2(0.5f * (zn.x - zc.x), 0.5f * (zn.y - zc.y));
      tt = cmul(tab->w400[200 - k], make_float2(d.y, -d.x));
      const float2 xn = cadd(fe, tt);
      pw[k] = xk.x * xk.x + xk.y * xk.y;
      pw[200 - k] = xn.x * xn.x + xn.y * xn.y;
    }
    wave_lds_sync();
    // sparse triangular filters, double accumulation, log10
    for (int m = lane; m < a.n_mel; m += 64) {
      const int k0 = tab->f_start[m], len = tab->f_len[m];
      const float* __restrict__ w = tab->f_w + tab->f_off[m];
      double sum = 0.0;
      for (int q = 0; q < len; ++q) sum += (double)pw[k0 + q] * (double)w[q];
      const float lv = (float)log10(sum > 1e-10 ? sum : 1e-10);
      tile[m * 65 + f] = lv;
      wmax = fmaxf(wmax, lv);
    }
    wave_lds_sync();
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) wmax = fmaxf(wmax, __shfl_xor(wmax, off, 64));
  if (lane == 0) atomicMax(a.clip_max + b, float_order_key(wmax));
  __syncthreads();
  // coalesced store of the tile: rows of 64 frames
  for (int idx = threadIdx.x; idx < a.n_mel * MEL_TILE; idx += 256) {
    const int m = idx >> 6, f = idx & 63;
    const int t = t0 + f;
    a.raw[((long)b * a.n_mel + m) * MEL_RAW_FRAMES + t] = tile[m * 65 + f];
  }
}

__global__ __launch_bounds__(256) void mel_finish_kernel(MelArgs a) {
  const int k = blockIdx.y;                        // output index
  const int b = a.clip_idx ? a.clip_idx[k] : k;    // clip whose raw frames / maximum are used
  const int seek = a.seek ? a.seek[k] : 0;
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  const long total = (long)a.n_mel * MEL_FRAMES;
  if (idx >= total) return;
  const int m = (int)(idx / MEL_FRAMES), t = (int)(idx % MEL_FRAMES);
  const double mmax = (double)float_from_key(a.clip_max[b]) - 8.0;
  const int src = seek + t;
  // frames >= 3002 are zeros only (clips are <= 30 s): log10(1e-10)
  double v = src < MEL_RAW_FRAMES ? (double)a.raw[((long)b * a.n_mel + m) * MEL_RAW_FRAMES + src] : -10.0;
  if (v < mmax) v = mmax;
  const float r = (float)((v + 4.0) / 4.0);
  if (a.out) a.out[(long)k * total + idx] = r;
  if (a.out_t) a.out_t[((long)k * (MEL_FRAMES + 2) + t + 1) * a.n_mel + m] = r;
}

}  // namespace

hipError_t mel_launch(const MelArgs& a, int batch, hipStream_t s) {
  hipError_t e = hipMemsetAsync(a.clip_max, 0x80, sizeof(int) * batch, s);  // 0x80808080: below any key
  if (e != hipSuccess) return e;
  const size_t smem = 4 * 208 * (sizeof(float2) + sizeof(float)) + (size_t)a.n_mel * 65 * sizeof(float);
  hipLaunchKernelGGL(mel_frames_kernel, dim3(MEL_TILES, batch), dim3(256), smem, s, a);
  e = hipGetLastError();
  if (e != hipSuccess) return e;
  const long total = (long)a.n_mel * MEL_FRAMES;
  hipLaunchKernelGGL(mel_finish_kernel, dim3((unsigned)((total + 255) / 256), batch), dim3(256), 0, s, a);
  return hipGetLastError();
}

hipError_t mel_window_launch(const MelArgs& a, int n, hipStream_t s) {
  const long total = (long)a.n_mel * MEL_FRAMES;
  hipLaunchKernelGGL(mel_finish_kernel, dim3((unsigned)((total + 255) / 256), n), dim3(256), 0, s, a);
  return hipGetLastError();
}

}  // namespace crispy
