/*
 * logmel_oracle.c -- TEST INFRASTRUCTURE ONLY (parity oracle).
 *
 * CPU restatement of the log-mel front end behind
 *   transcribe_rs::SpeechModel::transcribe  (whisper_cpp::WhisperEngine)
 *   reference call sites: src-tauri/src/managers/transcription.rs:183-185, 213-215
 * i.e. whisper.cpp's `log_mel_spectrogram` (whisper-rs-sys 0.15.0, Cargo.lock:6235-6245; source
 * not vendored).  [UPSTREAM-RECALL] SURVEY.md Appendix B.1.
 *
 * PARITY UNPINNED against the reference itself (it cannot be built here); this file is pinned
 * instead against HuggingFace `WhisperFeatureExtractor` golden vectors (a different
 * implementation of the same published front end) in tests/test_oracle_logmel.py -- the two
 * agree except on the last frames of a full 30 s clip, where whisper.cpp zero-pads and
 * OpenAI/HF reflect-pad.
 *
 * Semantics restated:
 *   - 16 kHz f32 PCM in +-1; N_FFT 400, hop 160, periodic Hann;
 *   - signal = reflect-pad 200 at the start | samples | 30 s of zeros (+200);
 *   - frame i starts at padded[i*160]; power spectrum of 201 bins;
 *   - mel = filters[n_mel][201] . power, accumulated in double; log10(max(., 1e-10));
 *   - clamp to (global max over all frames) - 8, then (x + 4) / 4;
 *   - the encoder consumes frames [0, 3000): that window is what this oracle returns, [n_mel][3000].
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define WL_SAMPLE_RATE 16000
#define WL_N_FFT 400
#define WL_HOP 160
#define WL_N_BINS 201
#define WL_FRAMES 3000

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

static float g_sin[WL_N_FFT], g_cos[WL_N_FFT], g_hann[WL_N_FFT];
static int g_ready = 0;

static void init(void) {
  if (g_ready) return;
  for (int i = 0; i < WL_N_FFT; i++) {
    double th = 2.0 * M_PI * i / WL_N_FFT;
    g_sin[i] = (float)sin(th);
    g_cos[i] = (float)cos(th);
    g_hann[i] = (float)(0.5 * (1.0 - cos(th)));
  }
  g_ready = 1;
}

/* naive DFT of N real inputs (N odd at the bottom of the recursion), interleaved complex output */
static void dft(const float *in, int N, float *out) {
  const int step = WL_N_FFT / N;
  for (int k = 0; k < N; k++) {
    float re = 0, im = 0;
    for (int n = 0; n < N; n++) {
      int idx = (k * n * step) % WL_N_FFT;
      re += in[n] * g_cos[idx];
      im -= in[n] * g_sin[idx];
    }
    out[2 * k] = re;
    out[2 * k + 1] = im;
  }
}

/* radix-2 decimation in time on real input, as whisper.cpp does (float throughout) */
static void fft(float *in, int N, float *out) {
  if (N == 1) { out[0] = in[0]; out[1] = 0; return; }
  const int half = N / 2;
  if (N - half * 2 == 1) { dft(in, N, out); return; }
  float *even = in + N;
  for (int i = 0; i < half; i++) even[i] = in[2 * i];
  float *even_fft = out + 2 * N;
  fft(even, half, even_fft);
  float *odd = even;
  for (int i = 0; i < half; i++) odd[i] = in[2 * i + 1];
  float *odd_fft = even_fft + N;
  fft(odd, half, odd_fft);
  const int step = WL_N_FFT / N;
  for (int k = 0; k < half; k++) {
    int idx = k * step;
    float re = g_cos[idx], im = -g_sin[idx];
    float re_odd = odd_fft[2 * k], im_odd = odd_fft[2 * k + 1];
    out[2 * k] = even_fft[2 * k] + re * re_odd - im * im_odd;
    out[2 * k + 1] = even_fft[2 * k + 1] + re * im_odd + im * re_odd;
    out[2 * (k + half)] = even_fft[2 * k] - re * re_odd + im * im_odd;
    out[2 * (k + half) + 1] = even_fft[2 * k + 1] - re * im_odd - im * re_odd;
  }
}

/* out: [n_mel][3000] = frames [seek, seek + 3000) of the clip's log-mel (whisper_full's later windows read the
 * same normalised spectrogram at an offset); returns 0, or -1 on bad arguments */
int wlo_logmel_window(const float *samples, int n_samples, const float *filters, int n_mel, int seek, float *out) {
  if (!samples || n_samples <= 0 || n_samples > 30 * WL_SAMPLE_RATE || !filters || n_mel <= 0 || !out) return -1;
  if (seek < 0 || seek > WL_FRAMES) return -1;
  init();
  const long pad1 = 30L * WL_SAMPLE_RATE, pad2 = WL_N_FFT / 2;
  const long n_pad = n_samples + pad1 + 2 * pad2;
  float *x = (float *)calloc((size_t)n_pad, sizeof(float));
  if (!x) return -1;
  memcpy(x + pad2, samples, sizeof(float) * (size_t)n_samples);
  for (long i = 0; i < pad2 && i + 1 < n_samples; i++) x[pad2 - 1 - i] = samples[1 + i]; /* reflect, edge not repeated */
  const long n_len = (n_pad - WL_N_FFT) / WL_HOP;
  float *mel = (float *)malloc(sizeof(float) * (size_t)n_mel * (size_t)n_len);
  if (!mel) { free(x); return -1; }
  float fft_in[2 * WL_N_FFT], fft_out[8 * WL_N_FFT];
  long i = 0;
  const long n_live = (n_pad / WL_HOP + 1) < n_len ? (n_pad / WL_HOP + 1) : n_len;
  for (; i < n_live; i++) {
    const long off = i * WL_HOP;
    for (int j = 0; j < WL_N_FFT; j++) fft_in[j] = (off + j < n_pad) ? g_hann[j] * x[off + j] : 0.f;
    fft(fft_in, WL_N_FFT, fft_out);
    for (int j = 0; j < WL_N_BINS; j++)
      fft_out[j] = fft_out[2 * j] * fft_out[2 * j] + fft_out[2 * j + 1] * fft_out[2 * j + 1];
    for (int j = 0; j < n_mel; j++) {
      double sum = 0.0;
      for (int k = 0; k < WL_N_BINS; k++) sum += fft_out[k] * filters[j * WL_N_BINS + k];
      sum = log10(sum > 1e-10 ? sum : 1e-10);
      mel[j * n_len + i] = (float)sum;
    }
  }
  for (; i < n_len; i++)
    for (int j = 0; j < n_mel; j++) mel[j * n_len + i] = (float)log10(1e-10);
  double mmax = -1e20;
  for (long q = 0; q < (long)n_mel * n_len; q++) if (mel[q] > mmax) mmax = mel[q];
  mmax -= 8.0;
  for (int j = 0; j < n_mel; j++)
    for (int t = 0; t < WL_FRAMES; t++) {
      double v = seek + t < n_len ? mel[j * n_len + seek + t] : mmax;
      if (v < mmax) v = mmax;
      out[j * WL_FRAMES + t] = (float)((v + 4.0) / 4.0);
    }
  free(mel);
  free(x);
  return 0;
}

/* the window the encoder consumes first: frames [0, 3000) */
int wlo_logmel(const float *samples, int n_samples, const float *filters, int n_mel, float *out) {
  return wlo_logmel_window(samples, n_samples, filters, n_mel, 0, out);
}
