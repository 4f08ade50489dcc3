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
};

namespace {

int mel_reserve(crispy_mel* h, int batch, long stride, bool need_pcm, bool need_out) {
  if (batch > h->cap_batch) {
    for (void* p : {(void*)h->d_n, (void*)h->d_raw, (void*)h->d_max, (void*)h->d_out})
      if (p) (void)hipFree(p);
    h->d_n = nullptr; h->d_raw = nullptr; h->d_max = nullptr; h->d_out = nullptr;
    if (h->d_pcm) { (void)hipFree(h->d_pcm); h->d_pcm = nullptr; h->cap_stride = 0; }
    h->cap_batch = 0;
    const size_t elems = (size_t)batch * h->n_mel * MEL_FRAMES;
    HIP_TRY(hipMalloc(&h->d_n, sizeof(int) * batch));
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

int crispy_mel_create(const float* filters, int n_mel, int device, crispy_mel** out) {
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
    if (off + len > MEL_MAX_MELS * 64) {
      delete tab;
      return fail(CRISPY_ERR_BAD_MODEL, "crispy_mel_create: filter bank is not triangular-sparse");
    }
    tab->f_start[m] = k0 < 0 ? 0 : k0;
    tab->f_len[m] = len;
    tab->f_off[m] = off;
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
}

void crispy_mel_destroy(crispy_mel* h) {
  if (!h) return;
  (void)hipSetDevice(h->device);
  if (h->stream) (void)hipStreamSynchronize(h->stream);
  for (void* p : {(void*)h->d_tab, (void*)h->d_pcm, (void*)h->d_n, (void*)h->d_raw, (void*)h->d_max, (void*)h->d_out})
    if (p) (void)hipFree(p);
  if (h->stream) (void)hipStreamDestroy(h->stream);
  delete h;
}

int crispy_mel_compute_device(crispy_mel* h, const float* d_pcm, long pcm_stride, const int* n_samples,
                              int batch, float* d_out, float* d_out_t, void* hip_stream) {
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
  return CRISPY_OK;
}

int crispy_mel_compute(crispy_mel* h, const float* pcm, long pcm_stride, const int* n_samples, int batch,
                       float* out) {
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
}

int crispy_mel_synchronize(crispy_mel* h) {
  if (!h) return fail(CRISPY_ERR_INVALID_ARG, "crispy_mel_synchronize: NULL handle");
  HIP_TRY(hipSetDevice(h->device));
  HIP_TRY(hipStreamSynchronize(h->stream));
  return CRISPY_OK;
}

}  // extern "C"
