/*
 * process_frame_dropin.c -- the literal drop-in call, from plain C99 against include/crispy_hip.h (no Python, no
 * ctypes prototypes in between): one stream, one 480-sample frame per call, host slices in and out -- exactly what
 * RnnNoiseProcessor::push_sample does with `self.denoise.process_frame(&mut out[..], &in[..])` inside the 10 ms audio
 * callback (/root/reference/src-tauri/src/audio.rs:260-268).
 *
 *   process_frame_dropin <model.txt> <in.f32> <out.f32> <n_frames> <timed_calls>
 *
 * model.txt: rnnoise-nu text model (crispy_rn_create_from_file); in.f32: n_frames * 480 raw floats (int16 range);
 * out.f32 receives the denoised frames + one float of VAD per frame appended (n_frames * 481 floats).
 * Then `timed_calls` more process_frame calls on the last frame are timed one by one (copy-in + high-pass + frame
 * kernel + copy-out + synchronisation each) and one JSON line with the percentiles in microseconds goes to stdout.
 * Used by tests/test_gpu_c_dropin.py (parity of the C path with the oracle) and by bench.py (latency_us).
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "crispy_hip.h"

static double now_us(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return ts.tv_sec * 1e6 + ts.tv_nsec * 1e-3;
}
static int cmp_double(const void *a, const void *b) {
  const double x = *(const double *)a, y = *(const double *)b;
  return (x > y) - (x < y);
}

int main(int argc, char **argv) {
  if (argc != 6) {
    fprintf(stderr, "usage: %s model.txt in.f32 out.f32 n_frames timed_calls\n", argv[0]);
    return 2;
  }
  const int n_frames = atoi(argv[4]), timed = atoi(argv[5]);
  if (n_frames <= 0 || timed < 0) return 2;
  crispy_rn *h = NULL;
  if (crispy_rn_create_from_file(argv[1], 1, 0, &h) != CRISPY_OK) {
    fprintf(stderr, "create failed: %s\n", crispy_last_error());
    return 1;
  }
  if (crispy_rn_n_streams(h) != 1) return 1;
  float *in = (float *)malloc(sizeof(float) * CRISPY_RN_FRAME_SIZE * (size_t)n_frames);
  float *out = (float *)malloc(sizeof(float) * (CRISPY_RN_FRAME_SIZE + 1) * (size_t)n_frames);
  FILE *f = fopen(argv[2], "rb");
  if (!in || !out || !f || fread(in, sizeof(float) * CRISPY_RN_FRAME_SIZE, (size_t)n_frames, f) != (size_t)n_frames) {
    fprintf(stderr, "cannot read %s\n", argv[2]);
    return 1;
  }
  fclose(f);
  float *vad = out + (size_t)CRISPY_RN_FRAME_SIZE * n_frames;
  for (int t = 0; t < n_frames; ++t) {   /* one frame per call, as the audio callback does */
    float frame_out[CRISPY_RN_FRAME_SIZE];
    const int rc = crispy_rn_process(h, in + (size_t)t * CRISPY_RN_FRAME_SIZE, frame_out, &vad[t], 1,
                                     CRISPY_RN_LAYOUT_TBF);
    if (rc != CRISPY_OK) {
      fprintf(stderr, "process failed (%d): %s\n", rc, crispy_last_error());
      return 1;
    }
    memcpy(out + (size_t)t * CRISPY_RN_FRAME_SIZE, frame_out, sizeof(frame_out));
  }
  f = fopen(argv[3], "wb");
  if (!f || fwrite(out, sizeof(float), (size_t)(CRISPY_RN_FRAME_SIZE + 1) * n_frames, f) !=
                (size_t)(CRISPY_RN_FRAME_SIZE + 1) * n_frames) {
    fprintf(stderr, "cannot write %s\n", argv[3]);
    return 1;
  }
  fclose(f);
  /* error behaviour through the header's types */
  if (crispy_rn_process(h, NULL, out, NULL, 1, CRISPY_RN_LAYOUT_TBF) != CRISPY_ERR_INVALID_ARG) return 1;
  if (crispy_rn_reset(h, 5) == CRISPY_OK) return 1;   /* stream index out of range */
  if (timed > 0) {
    double *us = (double *)malloc(sizeof(double) * (size_t)timed);
    float frame_out[CRISPY_RN_FRAME_SIZE], v;
    const float *last = in + (size_t)(n_frames - 1) * CRISPY_RN_FRAME_SIZE;
    for (int i = 0; i < 200; ++i) crispy_rn_process(h, last, frame_out, &v, 1, CRISPY_RN_LAYOUT_TBF);
    for (int i = 0; i < timed; ++i) {
      const double t0 = now_us();
      if (crispy_rn_process(h, last, frame_out, &v, 1, CRISPY_RN_LAYOUT_TBF) != CRISPY_OK) return 1;
      us[i] = now_us() - t0;
    }
    qsort(us, (size_t)timed, sizeof(double), cmp_double);
    double mean = 0;
    for (int i = 0; i < timed; ++i) mean += us[i];
    printf("{\"calls\": %d, \"p50\": %.1f, \"p90\": %.1f, \"p99\": %.1f, \"max\": %.1f, \"mean\": %.1f, "
           "\"budget_us\": 10000, \"what\": \"crispy_rn_process(h, in, out, &vad, 1, TBF), n_streams = 1: host slice in, "
           "480 samples out, synchronous\"}\n",
           timed, us[timed / 2], us[(int)(timed * 0.9)], us[(int)(timed * 0.99)], us[timed - 1], mean / timed);
    free(us);
  }
  crispy_rn_destroy(h);
  free(in);
  free(out);
  return 0;
}
