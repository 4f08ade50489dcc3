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
//                      filters (double accumulation as upstream), log10.  The tables of the stage
//                      (window, twiddles, filter taps: 9.4 KB) are copied into LDS once per
//                      workgroup, the power spectrum overwrites the transform buffer in place
//                      (36 KB of LDS per workgroup: four per CU); a lane stores the four frames
//                      of its mel bin as one 16-byte piece; the per-clip maximum is folded into
//                      one atomicMax per wave.
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

constexpr int MEL_NF = 4;        // frames a wave transforms at once
constexpr int MEL_BS = 208;      // complex numbers between two frames' buffers (201 used)
constexpr size_t MEL_LDS_BUF = 4 * MEL_NF * MEL_BS * sizeof(float2);
constexpr size_t MEL_LDS = MEL_LDS_BUF + 400 * sizeof(float2) + 400 * sizeof(float) + MEL_FW_MAX * sizeof(float) + MEL_MAX_MELS * sizeof(int);

// (float) log10(s) for a normal double s: log10(2) e + log10(e) ln m with s = m 2^e, m in [sqrt(1/2), sqrt(2)), and
// ln m = 2 atanh z, z = (m - 1) / (m + 1) <= 0.172, as its odd series to z^13 (the z^15 term is 4e-13).  The absolute
// error is ~1e-12 against the 4.8e-7 ulp of the float the value is rounded to: the float differs from the correctly
// rounded one about once in 10^5 values.  ~25 double operations where the library's log10 took ~300 (it was a fifth
// of the kernel's instruction issue).
__device__ __forceinline__ float log10_to_float(double s) {
  const long long bits = __double_as_longlong(s);
  int e = (int)((bits >> 52) & 0x7ff) - 1023;
  double m = __longlong_as_double((bits & 0x000fffffffffffffLL) | 0x3ff0000000000000LL);
  if (m > 1.4142135623730951) { m *= 0.5; e += 1; }
  const double den = m + 1.0;
  double r = (double)__builtin_amdgcn_rcpf((float)den);
  r = r * (2.0 - den * r);
  r = r * (2.0 - den * r);
  const double z = (m - 1.0) * r, z2 = z * z;
  double p = 1.0 / 13.0;
  p = p * z2 + 1.0 / 11.0;
  p = p * z2 + 1.0 / 9.0;
  p = p * z2 + 1.0 / 7.0;
  p = p * z2 + 1.0 / 5.0;
  p = p * z2 + 1.0 / 3.0;
  const double ln_m = 2.0 * z + 2.0 * z * (z2 * p);
  return (float)((double)e * 0.30102999566398120 + ln_m * 0.43429448190325176);
}
__global__ __launch_bounds__(256) void mel_frames_kernel(MelArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int b = blockIdx.y;
  const int t0 = blockIdx.x * MEL_TILE;
  // per wave: MEL_NF transform buffers (the power spectrum of bin k replaces the real part of element k); per workgroup: the tables
  float2* buf = reinterpret_cast<float2*>(smem) + wave * (MEL_NF * MEL_BS);
  float* pw = reinterpret_cast<float*>(buf);
  float2* l_w400 = reinterpret_cast<float2*>(smem + MEL_LDS_BUF);
  float* l_hann = reinterpret_cast<float*>(l_w400 + 400);
  float* l_fw = l_hann + 400;
  int* l_meta = reinterpret_cast<int*>(l_fw + MEL_FW_MAX);
  const float* __restrict__ x = a.pcm + (long)b * a.pcm_stride;
  const int n = a.n_samples[b];
  {
    const MelTables* __restrict__ tab = a.tab;
    for (int i = threadIdx.x; i < 400; i += 256) { l_w400[i] = tab->w400[i]; l_hann[i] = tab->hann[i]; }
    for (int i = threadIdx.x; i < MEL_FW_MAX; i += 256) l_fw[i] = tab->f_w[i];
    if (threadIdx.x < MEL_MAX_MELS) l_meta[threadIdx.x] = tab->f_meta[threadIdx.x];
    __syncthreads();
  }
  float wmax = -1e30f;

  // A wave owns 16 consecutive frames and takes them MEL_NF at a time: every stage below runs over the MEL_NF frames at
  // once (frame-major work items over the 64 lanes), so the radix-5 passes with their 40 butterflies per frame, the
  // 101 spectrum pairs and the 80 mel bins fill the lanes, and a wave synchronisation is paid once per MEL_NF frames.
  for (int fb = 0; fb < 16; fb += MEL_NF) {
    const int f0 = wave * 16 + fb;
    // window + pack: z[m] = (h[2m] x[2m], h[2m+1] x[2m+1]); reflect at the start, zeros past the end
    {
      constexpr int TOT = MEL_NF * 200, NT = (TOT + 63) / 64;
      float2 xv[NT], hv[NT];
#pragma unroll
      for (int it = 0; it < NT; ++it) {          // all requests of the stage first
        const int idx = min(lane + 64 * it, TOT - 1);
        const int f = idx / 200, m = idx - f * 200;
        const int s0 = (t0 + f0 + f) * 160 + 2 * m - 200;
        int sa = s0 < 0 ? -s0 : s0, sb = s0 + 1 < 0 ? -(s0 + 1) : s0 + 1;
        xv[it].x = sa < n ? x[sa] : 0.f;
        xv[it].y = sb < n ? x[sb] : 0.f;
        hv[it] = *reinterpret_cast<const float2*>(l_hann + 2 * m);
      }
#pragma unroll
      for (int it = 0; it < NT; ++it) {
        const int idx = lane + 64 * it;
        if (idx < TOT) {
          const int f = idx / 200, m = idx - f * 200;
          buf[f * MEL_BS + m] = make_float2(xv[it].x * hv[it].x, xv[it].y * hv[it].y);
        }
      }
    }
    wave_lds_sync();
    pass_batched<200, 4, 1, 400, MEL_NF, MEL_BS>(buf, l_w400, lane);
    pass_batched<200, 2, 4, 400, MEL_NF, MEL_BS>(buf, l_w400, lane);
    pass_batched<200, 5, 8, 400, MEL_NF, MEL_BS>(buf, l_w400, lane);
    pass_batched<200, 5, 40, 400, MEL_NF, MEL_BS>(buf, l_w400, lane);
    // real post-processing of the pair (k, 200-k) and power spectrum
    {
      constexpr int TOT = MEL_NF * 101, NT = (TOT + 63) / 64;
#pragma unroll
      for (int it = 0; it < NT; ++it) {
        const int idx = lane + 64 * it;
        if (idx < TOT) {
          const int f = idx / 101, k = idx - f * 101;
          const float2* bf = buf + f * MEL_BS;
          const float2 zk = bf[k];
          const float2 zn = (k == 0) ? zk : bf[200 - k];
          float2 zc = cconj(zn);
          float2 fe = make_float2(0.5f * (zk.x + zc.x), 0.5f * (zk.y + zc.y));
          float2 d = make_float2(0.5f * (zk.x - zc.x), 0.5f * (zk.y - zc.y));
          float2 tt = cmul(l_w400[k], make_float2(d.y, -d.x));
          const float2 xk = cadd(fe, tt);
          zc = cconj(zk);
          fe = make_float2(0.5f * (zn.x + zc.x), 0.5f * (zn.y + zc.y));
          d = make_float2(0.5f * (zn.x - zc.x), 0.5f * (zn.y - zc.y));
          tt = cmul(l_w400[200 - k], make_float2(d.y, -d.x));
          const float2 xn = cadd(fe, tt);
          // in place: elements k and 200 - k belong to this work item alone, and both were read above
          pw[2 * (f * MEL_BS + k)] = xk.x * xk.x + xk.y * xk.y;
          pw[2 * (f * MEL_BS + 200 - k)] = xn.x * xn.x + xn.y * xn.y;
        }
      }
    }
    wave_lds_sync();
    // sparse triangular filters, double accumulation, log10: a lane takes one mel bin for the MEL_NF frames of the batch
    // (one pass over the filter weights, MEL_NF independent accumulation chains) and stores the MEL_NF consecutive
    // frames of its bin as one 16-byte piece -- no [n_mel][64] staging tile, which had cost half the resident waves
    for (int m = lane; m < a.n_mel; m += 64) {
      const int meta = l_meta[m];
      const int k0 = meta & 0xff, len = (meta >> 8) & 0xff;
      const float* w = l_fw + (meta >> 16);
      double sum[MEL_NF];
#pragma unroll
      for (int f = 0; f < MEL_NF; ++f) sum[f] = 0.0;
      for (int q = 0; q < len; ++q) {
        const double wq = (double)w[q];
#pragma unroll
        for (int f = 0; f < MEL_NF; ++f) sum[f] += (double)pw[2 * (f * MEL_BS + k0 + q)] * wq;
      }
      float lv[MEL_NF];
#pragma unroll
      for (int f = 0; f < MEL_NF; ++f) {
        lv[f] = sum[f] > 1e-10 ? log10_to_float(sum[f]) : -10.f;
        wmax = fmaxf(wmax, lv[f]);
      }
      static_assert(MEL_NF == 4, "one float4 per mel bin and batch");
      *reinterpret_cast<float4*>(a.raw + ((long)b * a.n_mel + m) * MEL_RAW_FRAMES + t0 + f0) = make_float4(lv[0], lv[1], lv[2], lv[3]);
    }
    wave_lds_sync();
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) wmax = fmaxf(wmax, __shfl_xor(wmax, off, 64));
  if (lane == 0) atomicMax(a.clip_max + b, float_order_key(wmax));
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
  constexpr size_t smem = MEL_LDS;     // 36 KB whatever the number of mel bins
  static_assert(smem <= 64 * 1024, "above the default dynamic-LDS limit: the launch would need hipFuncSetAttribute per device");
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
