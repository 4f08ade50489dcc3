/*
 * multi_gpu.c -- the stream-sharded multi-GPU split (independent streams, static block partition by
 * stream id, no data-path collective) from plain C against include/crispy_hip.h: ONE process, one crispy_rn handle per
 * shard, one host thread per shard, every shard on device (shard % crispy_device_count()).  What a Rust host gets from
 * the C ABI without Python or torch.distributed (bench.py --gpus N is the one-process-per-GPU form of the same split).
 *
 *   multi_gpu <model.txt> <in.f32> <out.f32> <n_streams> <n_frames> <n_shards>
 *
 * in.f32: [n_frames][n_streams][480] raw floats (int16 range, CRISPY_RN_LAYOUT_TBF); out.f32 receives the denoised
 * audio in the same layout.  Shard r owns a contiguous block of streams (shard_range): it copies its columns into a contiguous
 * [n_frames][own][480] block, runs crispy_rn_process on its own handle and thread, and scatters the result back.
 * One JSON line on stdout: devices, shards, streams per shard, a checksum of the output (sum of every sample as a
 * double) and the wall time.  tests/test_gpu_c_dropin.py runs it with 1 shard and with several and compares both the
 * bytes and the checksum (the shards are independent: the split must not change a single sample).
 */
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "crispy_hip.h"

typedef struct shard {
  const char *model;
  const float *in;   /* whole batch */
  float *out;        /* whole batch */
  int n_streams, n_frames, lo, hi, device;
  int rc;
  char err[256];
} shard;

static void *run_shard(void *p) {
  shard *s = (shard *)p;
  const int own = s->hi - s->lo;
  const size_t n = (size_t)s->n_frames * own * CRISPY_RN_FRAME_SIZE;
  crispy_rn *h = NULL;
  float *x = (float *)malloc(n * sizeof(float)), *y = (float *)malloc(n * sizeof(float));
  s->rc = CRISPY_ERR_OOM;
  if (x && y) {
    for (int t = 0; t < s->n_frames; ++t)
      memcpy(x + (size_t)t * own * CRISPY_RN_FRAME_SIZE,
             s->in + ((size_t)t * s->n_streams + s->lo) * CRISPY_RN_FRAME_SIZE, (size_t)own * CRISPY_RN_FRAME_SIZE * sizeof(float));
    s->rc = crispy_rn_create_from_file(s->model, own, s->device, &h);
    if (s->rc == CRISPY_OK) s->rc = crispy_rn_process(h, x, y, NULL, s->n_frames, CRISPY_RN_LAYOUT_TBF);
    if (s->rc != CRISPY_OK) {        /* crispy_last_error is per thread: copy it out here */
      strncpy(s->err, crispy_last_error(), sizeof(s->err) - 1);
    } else {
      for (int t = 0; t < s->n_frames; ++t)
        memcpy(s->out + ((size_t)t * s->n_streams + s->lo) * CRISPY_RN_FRAME_SIZE,
               y + (size_t)t * own * CRISPY_RN_FRAME_SIZE, (size_t)own * CRISPY_RN_FRAME_SIZE * sizeof(float));
    }
    crispy_rn_destroy(h);
  }
  free(x);
  free(y);
  return NULL;
}

int main(int argc, char **argv) {
  if (argc != 7) {
    fprintf(stderr, "usage: %s model.txt in.f32 out.f32 n_streams n_frames n_shards\n", argv[0]);
    return 2;
  }
  const int B = atoi(argv[4]), T = atoi(argv[5]), R = atoi(argv[6]);
  if (B <= 0 || T <= 0 || R <= 0 || R > B || R > 64) return 2;
  const int n_dev = crispy_device_count();
  if (n_dev < 1) {
    fprintf(stderr, "no gfx950 device\n");
    return 3;
  }
  const size_t n = (size_t)T * B * CRISPY_RN_FRAME_SIZE;
  float *in = (float *)malloc(n * sizeof(float)), *out = (float *)calloc(n, sizeof(float));
  if (!in || !out) return 4;
  FILE *f = fopen(argv[2], "rb");
  if (!f || fread(in, sizeof(float), n, f) != n) {
    fprintf(stderr, "cannot read %zu floats from %s\n", n, argv[2]);
    return 4;
  }
  fclose(f);
  shard sh[64];
  pthread_t th[64];
  struct timespec t0, t1;
  clock_gettime(CLOCK_MONOTONIC, &t0);
  for (int r = 0; r < R; ++r) {
    memset(&sh[r], 0, sizeof(sh[r]));
    sh[r].model = argv[1]; sh[r].in = in; sh[r].out = out; sh[r].n_streams = B; sh[r].n_frames = T;
    sh[r].lo = r * (B / R) + (r < B % R ? r : B % R);     /* crispy_amd/sharding.py: shard_range (the first B % R shards get one more) */
    sh[r].hi = sh[r].lo + B / R + (r < B % R ? 1 : 0);
    sh[r].device = r % n_dev;
    if (pthread_create(&th[r], NULL, run_shard, &sh[r]) != 0) return 5;
  }
  int bad = 0;
  for (int r = 0; r < R; ++r) {
    pthread_join(th[r], NULL);
    if (sh[r].rc != CRISPY_OK) {
      fprintf(stderr, "shard %d (streams %d..%d, device %d): status %d: %s\n", r, sh[r].lo, sh[r].hi, sh[r].device, sh[r].rc, sh[r].err);
      bad = 1;
    }
  }
  clock_gettime(CLOCK_MONOTONIC, &t1);
  if (bad) return 6;
  double sum = 0.0;
  for (size_t i = 0; i < n; ++i) sum += (double)out[i];
  f = fopen(argv[3], "wb");
  if (!f || fwrite(out, sizeof(float), n, f) != n) return 4;
  fclose(f);
  printf("{\"devices\": %d, \"shards\": %d, \"streams\": %d, \"frames\": %d, \"checksum\": %.17g, \"wall_ms\": %.3f}\n", n_dev, R, B, T,
         sum, (t1.tv_sec - t0.tv_sec) * 1e3 + (t1.tv_nsec - t0.tv_nsec) * 1e-6);
  free(in);
  free(out);
  return 0;
}
