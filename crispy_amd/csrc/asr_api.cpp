// asr_api.cpp -- extern "C" entry points of the log-mel / Whisper path (include/crispy_hip.h).
#include "../../include/crispy_hip.h"
#include "api_util.h"
#include "asr_common.h"

#include <cmath>
#include <cstring>
#include <new>
#include <vector>

using namespace crispy;

struct crispy_mel {
  int device = 0;
  int n_mel = 0;
  hipStream_t stream = nullptr;
  MelTables* d_tab = nullptr;
  // workspace, grown on demand
  int cap_batch = 0;
  long cap_stride = 0;
  float* d_pcm = nullptr;
  int* d_n = nullptr;
  float* d_raw = nullptr;
  int* d_max = nullptr;
  float* d_out = nullptr;
  int* d_idx = nullptr;     // [2][cap_batch]: clip index | seek of a window call
  int last_batch = 0;       // clips whose raw frames d_raw / d_max currently hold
};

namespace {

int mel_reserve(crispy_mel* h, int batch, long stride, bool need_pcm, bool need_out) {
  if (batch > h->cap_batch) {
    for (void* p : {(void*)h->d_n, (void*)h->d_raw, (void*)h->d_max, (void*)h->d_out, (void*)h->d_idx})
      if (p) (void)hipFree(p);
    h->d_n = nullptr; h->d_raw = nullptr; h->d_max = nullptr; h->d_out = nullptr; h->d_idx = nullptr;
    h->last_batch = 0;
    if (h->d_pcm) { (void)hipFree(h->d_pcm); h->d_pcm = nullptr; h->cap_stride = 0; }
    h->cap_batch = 0;
    const size_t elems = (size_t)batch * h->n_mel * MEL_RAW_FRAMES;
    HIP_TRY(hipMalloc(&h->d_n, sizeof(int) * batch));
    HIP_TRY(hipMalloc(&h->d_idx, sizeof(int) * 2 * batch));
    HIP_TRY(hipMalloc(&h->d_max, sizeof(int) * batch));
    HIP_TRY(hipMalloc(&h->d_raw, sizeof(float) * elems));
    h->cap_batch = batch;
  }
  if (need_out && !h->d_out)
    HIP_TRY(hipMalloc(&h->d_out, sizeof(float) * (size_t)h->cap_batch * h->n_mel * MEL_FRAMES));
  if (need_pcm && (!h->d_pcm || stride > h->cap_stride)) {
    if (h->d_pcm) (void)hipFree(h->d_pcm);
    h->d_pcm = nullptr;
    HIP_TRY(hipMalloc(&h->d_pcm, sizeof(float) * (size_t)h->cap_batch * stride));
    h->cap_stride = stride;
  }
  return CRISPY_OK;
}

int mel_check_lengths(const int* n_samples, int batch, long stride) {
  for (int b = 0; b < batch; ++b)
    if (n_samples[b] <= 0 || n_samples[b] > 480000 || n_samples[b] > stride)
      return fail(CRISPY_ERR_INVALID_ARG, "crispy_mel: clip %d has %d samples (1..480000, <= stride %ld)", b,
                  n_samples[b], stride);
  return CRISPY_OK;
}

}  // namespace

extern "C" {

int crispy_mel_create(const float* filters, int n_mel, int device, crispy_mel** out) try {
  if (!out) return fail(CRISPY_ERR_INVALID_ARG, "crispy_mel_create: out is NULL");
  *out = nullptr;
  if (!filters || n_mel <= 0 || n_mel > MEL_MAX_MELS)
    return fail(CRISPY_ERR_BAD_MODEL, "crispy_mel_create: filters must be [n_mel<=%d][201]", MEL_MAX_MELS);
  int rc = check_device(device, "crispy_mel_create");
  if (rc != CRISPY_OK) return rc;
  MelTables* tab = new (std::nothrow) MelTables();
  if (!tab) return fail(CRISPY_ERR_OOM, "crispy_mel_create: host allocation failed");
  std::memset(tab, 0, sizeof(*tab));
  const double pi = 3.14159265358979323846;
  for (int i = 0; i < 400; ++i) {
    const double th = 2.0 * pi * i / 400.0;
    tab->hann[i] = (float)(0.5 * (1.0 - std::cos(th)));
    tab->w400[i].x = (float)std::cos(-th);
    tab->w400[i].y = (float)std::sin(-th);
  }
  int off = 0;
  for (int m = 0; m < n_mel; ++m) {
    int k0 = -1, k1 = -1;
    for (int k = 0; k < MEL_BINS; ++k)
      if (filters[m * MEL_BINS + k] != 0.f) { if (k0 < 0) k0 = k; k1 = k; }
    const int len = k0 < 0 ? 0 : k1 - k0 + 1;
    if (len > 64 || off + len > MEL_FW_MAX) {
      delete tab;
      return fail(CRISPY_ERR_BAD_MODEL, "crispy_mel_create: filter bank is not triangular-sparse (a filter spans %d bins of at most 64, "
                  "the bank %d of at most %d)", len, off + len, MEL_FW_MAX);
    }
    tab->f_meta[m] = (k0 < 0 ? 0 : k0) | len << 8 | off << 16;
    for (int q = 0; q < len; ++q) tab->f_w[off + q] = filters[m * MEL_BINS + k0 + q];
    off += len;
  }
  crispy_mel* h = new (std::nothrow) crispy_mel();
  if (!h) { delete tab; return fail(CRISPY_ERR_OOM, "crispy_mel_create: host allocation failed"); }
  h->device = device;
  h->n_mel = n_mel;
  auto body = [&]() -> int {
    HIP_TRY(hipSetDevice(device));
    HIP_TRY(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
    HIP_TRY(hipMalloc(&h->d_tab, sizeof(MelTables)));
    HIP_TRY(hipMemcpy(h->d_tab, tab, sizeof(MelTables), hipMemcpyHostToDevice));
    return CRISPY_OK;
  };
  rc = body();
  delete tab;
  if (rc != CRISPY_OK) { crispy_mel_destroy(h); return rc; }
  *out = h;
  return CRISPY_OK;
} CRISPY_CATCH_RET("crispy_mel_create")

void crispy_mel_destroy(crispy_mel* h) try {
  if (!h) return;
  (void)hipSetDevice(h->device);
  if (h->stream) (void)hipStreamSynchronize(h->stream);
  for (void* p : {(void*)h->d_tab, (void*)h->d_pcm, (void*)h->d_n, (void*)h->d_raw, (void*)h->d_max, (void*)h->d_out, (void*)h->d_idx})
    if (p) (void)hipFree(p);
  if (h->stream) (void)hipStreamDestroy(h->stream);
  delete h;
} CRISPY_CATCH_VOID("crispy_mel_destroy")

int crispy_mel_compute_device(crispy_mel* h, const float* d_pcm, long pcm_stride, const int* n_samples,
                              int batch, float* d_out, float* d_out_t, void* hip_stream) try {
  if (!h) return fail(CRISPY_ERR_INVALID_ARG, "crispy_mel_compute_device: NULL handle");
  if (batch < 0) return fail(CRISPY_ERR_INVALID_ARG, "crispy_mel_compute_device: batch < 0");
  if (batch == 0) return CRISPY_OK;
  if (!d_pcm || !n_samples || (!d_out && !d_out_t))
    return fail(CRISPY_ERR_INVALID_ARG, "crispy_mel_compute_device: NULL argument");
  int rc = mel_check_lengths(n_samples, batch, pcm_stride);
  if (rc != CRISPY_OK) return rc;
  HIP_TRY(hipSetDevice(h->device));
  hipStream_t s = hip_stream ? (hipStream_t)hip_stream : h->stream;
  rc = mel_reserve(h, batch, pcm_stride, false, false);
  if (rc != CRISPY_OK) return rc;
  HIP_TRY(hipMemcpyAsync(h->d_n, n_samples, sizeof(int) * batch, hipMemcpyHostToDevice, s));
  MelArgs a{};
  a.pcm = d_pcm;
  a.pcm_stride = pcm_stride;
  a.n_samples = h->d_n;
  a.n_mel = h->n_mel;
  a.tab = h->d_tab;
  a.raw = h->d_raw;
  a.clip_max = h->d_max;
  a.out = d_out;
  a.out_t = d_out_t;
  HIP_TRY(mel_launch(a, batch, s));
  h->last_batch = batch;
  return CRISPY_OK;
} CRISPY_CATCH_RET("crispy_mel_compute_device")

int crispy_mel_compute(crispy_mel* h, const float* pcm, long pcm_stride, const int* n_samples, int batch,
                       float* out) try {
  if (!h) return fail(CRISPY_ERR_INVALID_ARG, "crispy_mel_compute: NULL handle");
  if (batch < 0) return fail(CRISPY_ERR_INVALID_ARG, "crispy_mel_compute: batch < 0");
  if (batch == 0) return CRISPY_OK;
  if (!pcm || !n_samples || !out) return fail(CRISPY_ERR_INVALID_ARG, "crispy_mel_compute: NULL argument");
  int rc = mel_check_lengths(n_samples, batch, pcm_stride);
  if (rc != CRISPY_OK) return rc;
  HIP_TRY(hipSetDevice(h->device));
  rc = mel_reserve(h, batch, pcm_stride, true, true);
  if (rc != CRISPY_OK) return rc;
  HIP_TRY(hipMemcpyAsync(h->d_pcm, pcm, sizeof(float) * (size_t)batch * pcm_stride, hipMemcpyHostToDevice, h->stream));
  rc = crispy_mel_compute_device(h, h->d_pcm, pcm_stride, n_samples, batch, h->d_out, nullptr, nullptr);
  if (rc != CRISPY_OK) return rc;
  HIP_TRY(hipMemcpyAsync(out, h->d_out, sizeof(float) * (size_t)batch * h->n_mel * MEL_FRAMES, hipMemcpyDeviceToHost,
                         h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  return CRISPY_OK;
} CRISPY_CATCH_RET("crispy_mel_compute")

int crispy_mel_window_device(crispy_mel* h, const int* clip_idx, const int* seek, int n, float* d_out,
                             float* d_out_t, void* hip_stream) try {
  if (!h) return fail(CRISPY_ERR_INVALID_ARG, "crispy_mel_window_device: NULL handle");
  if (n < 0) return fail(CRISPY_ERR_INVALID_ARG, "crispy_mel_window_device: n < 0");
  if (n == 0) return CRISPY_OK;
  if (!clip_idx || !seek || (!d_out && !d_out_t)) return fail(CRISPY_ERR_INVALID_ARG, "crispy_mel_window_device: NULL argument");
  if (n > h->cap_batch || h->last_batch == 0)
    return fail(CRISPY_ERR_INVALID_ARG, "crispy_mel_window_device: %d windows, but the last crispy_mel_compute_device call held %d clips",
                n, h->last_batch);
  for (int k = 0; k < n; ++k)
    if (clip_idx[k] < 0 || clip_idx[k] >= h->last_batch || seek[k] < 0 || seek[k] > MEL_FRAMES)
      return fail(CRISPY_ERR_INVALID_ARG, "crispy_mel_window_device: window %d = (clip %d, seek %d) out of range", k, clip_idx[k], seek[k]);
  HIP_TRY(hipSetDevice(h->device));
  hipStream_t s = hip_stream ? (hipStream_t)hip_stream : h->stream;
  HIP_TRY(hipMemcpyAsync(h->d_idx, clip_idx, sizeof(int) * n, hipMemcpyHostToDevice, s));
  HIP_TRY(hipMemcpyAsync(h->d_idx + h->cap_batch, seek, sizeof(int) * n, hipMemcpyHostToDevice, s));
  HIP_TRY(hipStreamSynchronize(s));     // the host arrays may be reused by the caller
  MelArgs a{};
  a.n_mel = h->n_mel;
  a.raw = h->d_raw;
  a.clip_max = h->d_max;
  a.out = d_out;
  a.out_t = d_out_t;
  a.clip_idx = h->d_idx;
  a.seek = h->d_idx + h->cap_batch;
  HIP_TRY(mel_window_launch(a, n, s));
  return CRISPY_OK;
} CRISPY_CATCH_RET("crispy_mel_window_device")

int crispy_mel_synchronize(crispy_mel* h) try {
  if (!h) return fail(CRISPY_ERR_INVALID_ARG, "crispy_mel_synchronize: NULL handle");
  HIP_TRY(hipSetDevice(h->device));
  HIP_TRY(hipStreamSynchronize(h->stream));
  return CRISPY_OK;
} CRISPY_CATCH_RET("crispy_mel_synchronize")

}  // extern "C"

// ---------------------------------------------------------------------------------------------
// 48 -> 16 kHz resampler
// ---------------------------------------------------------------------------------------------
struct crispy_resampler {
  int device = 0;
  hipStream_t stream = nullptr;
  float* d_w = nullptr;     // [684][1040] circulant operator
  float* d_a = nullptr;     // [rows][1040]
  float* d_y = nullptr;     // [rows][684]
  long cap_rows = 0;
  // the f16-pair form (asr_common.h: rs_prep_split): W' [342][3 x 2 x RS_PITCH] f16 = W_hi | W_hi | W_lo over the window
  // [previous block | block]; planes [2][streams][(n_blk + 1) x RS_PITCH] f16
  void* d_w16 = nullptr;
  void* d_planes = nullptr;
  long cap_plane_blocks = 0;
};

// f32 -> f16 bits, round to nearest even (host: this file is also built by g++ without _Float16)
uint16_t f16_bits(float f) {
  uint32_t x;
  std::memcpy(&x, &f, 4);
  const uint32_t sign = (x >> 16) & 0x8000u;
  x &= 0x7fffffffu;
  if (x >= 0x7f800000u) return (uint16_t)(sign | 0x7c00u | (x > 0x7f800000u ? 0x200u : 0u));
  if (x >= 0x477ff000u) return (uint16_t)(sign | 0x7c00u);                 // >= 65520: rounds to infinity
  if (x < 0x38800000u) {                                                   // below 2^-14: a subnormal half, spacing 2^-24
    const float a = std::fabs(f) * 16777216.0f;
    return (uint16_t)(sign | (uint32_t)std::nearbyint(a));
  }
  uint32_t h = (((x >> 23) - 112u) << 10) | ((x & 0x7fffffu) >> 13);
  const uint32_t rem = x & 0x1fffu;
  if (rem > 0x1000u || (rem == 0x1000u && (h & 1u))) ++h;
  return (uint16_t)(sign | h);
}
double f16_value(uint16_t h) {
  const int e = (h >> 10) & 31, m = h & 1023;
  const double v = e == 0 ? std::ldexp((double)m, -24) : std::ldexp((double)(1024 + m), e - 25);
  return (h & 0x8000) ? -v : v;
}

extern "C" {

long crispy_resampler_out_len(long n_in) try {
  if (n_in <= 0) return 0;
  const long n_pad = (n_in + RS_CHUNK - 1) / RS_CHUNK * RS_CHUNK;   // last chunk zero-padded (transcription.rs:347-351)
  return n_pad / RS_FFT_IN * RS_FFT_OUT;
} CRISPY_CATCH_RET("crispy_resampler_out_len")

int crispy_resampler_create(int device, crispy_resampler** out) try {
  if (!out) return fail(CRISPY_ERR_INVALID_ARG, "crispy_resampler_create: out is NULL");
  *out = nullptr;
  int rc = check_device(device, "crispy_resampler_create");
  if (rc != CRISPY_OK) return rc;
  // windowed-sinc anti-aliasing filter and its band-limited circular impulse response g[2052], in double
  const double pi = 3.14159265358979323846;
  const int NI = RS_FFT_IN, NO = RS_FFT_OUT, N2 = 2 * RS_FFT_IN;
  std::vector<double> ft(NI);
  {
    const double cutoff = std::pow(0.4, 16.0 / NI) * NO / NI;
    double sum = 0.0;
    for (int n = 0; n < NI; ++n) {
      const double x = (double)n / NI;
      double w = 0.35875 - 0.48829 * std::cos(2 * pi * x) + 0.14128 * std::cos(4 * pi * x) - 0.01168 * std::cos(6 * pi * x);
      w *= w;
      const double t = (n - NI / 2) * cutoff;
      const double sinc = t == 0.0 ? 1.0 : std::sin(pi * t) / (pi * t);
      ft[n] = w * sinc;
      sum += ft[n];
    }
    for (int n = 0; n < NI; ++n) ft[n] = ft[n] / sum / (double)N2;
  }
  std::vector<double> Fr(NO), Fi(NO);
  for (int k = 0; k < NO; ++k) {
    double re = 0, im = 0;
    for (int n = 0; n < NI; ++n) {
      const double th = -2.0 * pi * (double)((long)k * n % N2) / N2;
      re += ft[n] * std::cos(th);
      im += ft[n] * std::sin(th);
    }
    Fr[k] = re; Fi[k] = im;
  }
  std::vector<double> g(N2);
  for (int m = 0; m < N2; ++m) {
    double acc = Fr[0];
    for (int k = 1; k < NO; ++k) {
      const double th = 2.0 * pi * (double)((long)k * m % N2) / N2;
      acc += 2.0 * (Fr[k] * std::cos(th) - Fi[k] * std::sin(th));
    }
    g[m] = acc;
  }
  std::vector<float> W((size_t)RS_N * RS_K, 0.f);
  for (int n = 0; n < RS_N; ++n)
    for (int j = 0; j < NI; ++j) W[(size_t)n * RS_K + j] = (float)g[((3 * n - j) % N2 + N2) % N2];
  // the f16-pair operator over the window [previous block | block]: output n of a block = row n of the block's own
  // product + row 342 + n of the previous block's (the overlap-add, folded into the operator)
  const int KW = 2 * RS_PITCH;
  std::vector<uint16_t> W16((size_t)NO * 3 * KW, 0);
  for (int n = 0; n < NO; ++n)
    for (int half = 0; half < 2; ++half)
      for (int j = 0; j < NI; ++j) {
        const int m = half == 0 ? NO + n : n;
        const double w = g[((3 * m - j) % N2 + N2) % N2];
        const uint16_t hi = f16_bits((float)w);
        const uint16_t lo = f16_bits((float)(w - f16_value(hi)));
        uint16_t* row = W16.data() + (size_t)n * 3 * KW;
        const int k = half * RS_PITCH + j;
        row[k] = hi; row[KW + k] = hi; row[2 * KW + k] = lo;
      }
  crispy_resampler* h = new (std::nothrow) crispy_resampler();
  if (!h) return fail(CRISPY_ERR_OOM, "crispy_resampler_create: host allocation failed");
  h->device = device;
  auto body = [&]() -> int {
    HIP_TRY(hipSetDevice(device));
    HIP_TRY(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
    HIP_TRY(hipMalloc(&h->d_w, W.size() * sizeof(float)));
    HIP_TRY(hipMemcpy(h->d_w, W.data(), W.size() * sizeof(float), hipMemcpyHostToDevice));
    HIP_TRY(hipMalloc(&h->d_w16, W16.size() * sizeof(uint16_t)));
    HIP_TRY(hipMemcpy(h->d_w16, W16.data(), W16.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
    return CRISPY_OK;
  };
  rc = body();
  if (rc != CRISPY_OK) { crispy_resampler_destroy(h); return rc; }
  *out = h;
  return CRISPY_OK;
} CRISPY_CATCH_RET("crispy_resampler_create")

void crispy_resampler_destroy(crispy_resampler* h) try {
  if (!h) return;
  (void)hipSetDevice(h->device);
  if (h->stream) (void)hipStreamSynchronize(h->stream);
  for (void* p : {(void*)h->d_w, (void*)h->d_a, (void*)h->d_y, h->d_w16, h->d_planes})
    if (p) (void)hipFree(p);
  if (h->stream) (void)hipStreamDestroy(h->stream);
  delete h;
} CRISPY_CATCH_VOID("crispy_resampler_destroy")

int crispy_resampler_process_device(crispy_resampler* h, const float* d_in, long in_stride, long n_in, int batch,
                                    float scale, int wav_s16, float* d_out, long out_stride, void* hip_stream) try {
  if (!h) return fail(CRISPY_ERR_INVALID_ARG, "crispy_resampler_process_device: NULL handle");
  if (batch < 0 || n_in < 0) return fail(CRISPY_ERR_INVALID_ARG, "crispy_resampler_process_device: negative size");
  const long n_out = crispy_resampler_out_len(n_in);
  if (batch == 0 || n_out == 0) return CRISPY_OK;
  if (!d_in || !d_out || in_stride < n_in || out_stride < n_out)
    return fail(CRISPY_ERR_INVALID_ARG, "crispy_resampler_process_device: bad pointer or stride");
  HIP_TRY(hipSetDevice(h->device));
  hipStream_t s = hip_stream ? (hipStream_t)hip_stream : h->stream;
  const int n_blk = (int)(n_out / RS_FFT_OUT);
  // The f16-pair form on the f16 matrix cores (rows of the output must take 8-byte stores); CRISPY_RS_GEMM=f32 (developer
  // build) keeps the f32 form below for A/B.
  static const bool pair_form = [] { const char* e = dev_env("CRISPY_RS_GEMM"); return !(e && e[0] == 'f'); }();
  if (pair_form && out_stride % 2 == 0 && ((size_t)d_out & 7) == 0) {
    const long max_blocks = 1L << 18;   // 262144 block slots: 1.1 GB of planes
    int group = (int)(max_blocks / (n_blk + 1));
    if (group < 1) group = 1;
    if (group > batch) group = batch;
    const long need = (long)group * (n_blk + 1);
    if (need > h->cap_plane_blocks) {
      if (h->d_planes) (void)hipFree(h->d_planes);
      h->d_planes = nullptr;
      h->cap_plane_blocks = 0;
      HIP_TRY(hipMalloc(&h->d_planes, (size_t)need * RS_PITCH * 2 * sizeof(uint16_t)));
      h->cap_plane_blocks = need;
    }
    for (int b0 = 0; b0 < batch; b0 += group) {
      const int nb = (batch - b0) < group ? (batch - b0) : group;
      HIP_TRY(rs_prep_split(d_in + (long)b0 * in_stride, in_stride, n_in, scale, wav_s16, h->d_planes, nb, n_blk, s));
      // the streams of the group as ONE row dimension (their block slots are contiguous in the planes): row m is window
      // m % (n_blk + 1) of stream m / (n_blk + 1); the last window of a stream [last block | the next stream's zero slot] is
      // not an output block and is dropped by the epilogue (as separate matrices of 1404 rows a tenth of the tiles was padding)
      HGemmArgs g{};
      g.A = reinterpret_cast<const _Float16*>(h->d_planes); g.lda = RS_PITCH;
      g.W = reinterpret_cast<const _Float16*>(h->d_w16); g.ldw = 3L * 2 * RS_PITCH;
      g.C = d_out + (long)b0 * out_stride; g.ldc = RS_FFT_OUT;
      g.M = nb * (n_blk + 1) - 1; g.N = RS_FFT_OUT; g.K = 3 * 2 * RS_PITCH;      // (- 1: the very last window would read past the planes)
      g.c_group_rows = n_blk + 1; g.c_group_valid = n_blk; g.c_group_stride = out_stride;
      g.k_seg = 2 * RS_PITCH;
      g.a_seg_off[0] = 0; g.a_seg_off[1] = (long)nb * (n_blk + 1) * RS_PITCH; g.a_seg_off[2] = 0;      // x_hi | x_lo | x_hi
      g.xcd_swizzle = 1;
      HIP_TRY(gemm_hh(g, HGEMM_F32, 1, s));
    }
    return CRISPY_OK;
  }
  // bounded workspace: process the streams in groups
  const long max_rows = 1L << 17;   // 131072 rows: 545 MB of A + 359 MB of Y
  int group = (int)(max_rows / n_blk);
  if (group < 1) group = 1;
  if (group > batch) group = batch;
  const long rows_cap = (long)group * n_blk;
  if (rows_cap > h->cap_rows) {
    if (h->d_a) (void)hipFree(h->d_a);
    if (h->d_y) (void)hipFree(h->d_y);
    h->d_a = h->d_y = nullptr;
    h->cap_rows = 0;
    HIP_TRY(hipMalloc(&h->d_a, (size_t)rows_cap * RS_K * sizeof(float)));
    HIP_TRY(hipMalloc(&h->d_y, (size_t)rows_cap * RS_N * sizeof(float)));
    h->cap_rows = rows_cap;
  }
  for (int b0 = 0; b0 < batch; b0 += group) {
    const int nb = (batch - b0) < group ? (batch - b0) : group;
    const long rows = (long)nb * n_blk;
    HIP_TRY(rs_prep(d_in + (long)b0 * in_stride, in_stride, n_in, scale, wav_s16, h->d_a, nb, n_blk, s));
    GemmArgs g{};
    g.A = h->d_a; g.lda = RS_K; g.W = h->d_w; g.ldw = RS_K; g.C = h->d_y; g.ldc = RS_N;
    g.M = (int)rows; g.N = RS_N; g.K = RS_K;
    HIP_TRY(gemm_f32_nt(g, 1, s));
    HIP_TRY(rs_ola(h->d_y, d_out + (long)b0 * out_stride, out_stride, nb, n_blk, s));
  }
  return CRISPY_OK;
} CRISPY_CATCH_RET("crispy_resampler_process_device")

int crispy_resampler_synchronize(crispy_resampler* h) try {
  if (!h) return fail(CRISPY_ERR_INVALID_ARG, "crispy_resampler_synchronize: NULL handle");
  HIP_TRY(hipSetDevice(h->device));
  HIP_TRY(hipStreamSynchronize(h->stream));
  return CRISPY_OK;
} CRISPY_CATCH_RET("crispy_resampler_synchronize")

}  // extern "C"
