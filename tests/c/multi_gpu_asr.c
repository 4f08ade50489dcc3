/*
 * multi_gpu_asr.c -- BASELINE configs[4]'s split (Whisper transcription of many independent 30 s chunks, sharded over the
 * GPUs of a node: SURVEY.md 8e; caller managers/transcription.rs:27,178 -- one engine behind a Mutex per process) from plain
 * C against include/crispy_hip.h: ONE process, one crispy_asr engine per shard, one host thread per shard, every shard
 * on device (shard % crispy_device_count()).  Chunks are dealt in blocks: shard r owns a contiguous block of chunks
 * (crispy_amd/sharding.py: shard_range), transcribes them with ONE crispy_asr_transcribe_batch call on its own engine,
 * and writes the token ids of every chunk into its slot of the common table.  No data-path collective: chunks are
 * independent (no prompt carry-over between chunks, managers/transcription.rs:184).
 *
 *   multi_gpu_asr <model.bin> <pcm.f32> <n_chunks> <samples_per_chunk> <n_shards> <max_new_tokens>
 *
 * pcm.f32: [n_chunks][samples_per_chunk] raw 16 kHz floats.  One JSON line on stdout: devices, shards, and per chunk
 * its language token and token ids.  tests/test_gpu_c_dropin.py runs it with 1 and with 3 shards and compares the ids
 * with each other and with the Python binding's single-engine run (the split must not change a token).
 */
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "crispy_hip.h"

#define MAX_TOK 448

typedef struct chunk_out {
  int lang, n;
  int tok[MAX_TOK];
} chunk_out;

typedef struct shard {
  const char *model;
  const float *pcm;
  size_t samples;
  int lo, hi, device, max_new;
  chunk_out *out; /* whole table */
  int rc;
  char err[256];
} shard;

static void *run_shard(void *p) {
  shard *s = (shard *)p;
  const int own = s->hi - s->lo;
  crispy_asr *h = NULL;
  const float **ptr = (const float **)malloc((size_t)own * sizeof(*ptr));
  size_t *len = (size_t *)malloc((size_t)own * sizeof(*len));
  crispy_asr_result **res = (crispy_asr_result **)calloc((size_t)own, sizeof(*res));
  s->rc = CRISPY_ERR_OOM;
  if (ptr && len && res) {
    for (int i = 0; i < own; ++i) {
      ptr[i] = s->pcm + (size_t)(s->lo + i) * s->samples;
      len[i] = s->samples;
    }
    /* what bindings/rust/.../GpuWhisperEngine::load does: resident load, then whisper.cpp's precision */
    s->rc = crispy_asr_load_resident(s->model, s->device, &h);
    if (s->rc == CRISPY_OK) s->rc = crispy_asr_set_precision(h, 1);
    if (s->rc == CRISPY_OK) {
      crispy_asr_opts o;
      memset(&o, 0, sizeof(o));
      o.max_new_tokens = s->max_new;
      o.temperature_inc = -1.0f; /* one greedy pass per window: north_star's "greedy transcript" */
      s->rc = crispy_asr_transcribe_batch(h, ptr, len, own, &o, res);
    }
    if (s->rc != CRISPY_OK) {
      strncpy(s->err, crispy_last_error(), sizeof(s->err) - 1); /* per thread: copy it out here */
    } else {
      for (int i = 0; i < own; ++i) {
        chunk_out *c = &s->out[s->lo + i];
        c->lang = res[i]->language_token;
        c->n = res[i]->n_tokens < MAX_TOK ? res[i]->n_tokens : MAX_TOK;
        memcpy(c->tok, res[i]->tokens, (size_t)c->n * sizeof(int));
        crispy_asr_free_result(res[i]);
      }
    }
    crispy_asr_free(h);
  }
  free(ptr);
  free(len);
  free(res);
  return NULL;
}

int main(int argc, char **argv) {
  if (argc != 7) {
    fprintf(stderr, "usage: %s model.bin pcm.f32 n_chunks samples_per_chunk n_shards max_new_tokens\n", argv[0]);
    return 2;
  }
  const int N = atoi(argv[3]), R = atoi(argv[5]), max_new = atoi(argv[6]);
  const long S = atol(argv[4]);
  if (N <= 0 || S <= 0 || S > 480000 || R <= 0 || R > N || R > 64 || max_new < 0) return 2;
  if (crispy_abi_version() != CRISPY_ABI_VERSION) {
    fprintf(stderr, "libcrispy_hip ABI %d, header %d\n", crispy_abi_version(), CRISPY_ABI_VERSION);
    return 2;
  }
  const int n_dev = crispy_device_count();
  if (n_dev < 1) {
    fprintf(stderr, "no gfx950 device\n");
    return 3;
  }
  const size_t n = (size_t)N * (size_t)S;
  float *pcm = (float *)malloc(n * sizeof(float));
  chunk_out *out = (chunk_out *)calloc((size_t)N, sizeof(*out));
  if (!pcm || !out) return 4;
  FILE *f = fopen(argv[2], "rb");
  if (!f || fread(pcm, sizeof(float), n, f) != n) {
    fprintf(stderr, "cannot read %zu floats from %s\n", n, argv[2]);
    return 4;
  }
  fclose(f);
  shard sh[64];
  pthread_t th[64];
  for (int r = 0; r < R; ++r) {
    memset(&sh[r], 0, sizeof(sh[r]));
    sh[r].model = argv[1]; sh[r].pcm = pcm; sh[r].samples = (size_t)S; sh[r].max_new = max_new; sh[r].out = out;
    /* crispy_amd/sharding.py: shard_range -- the first N % R shards get one chunk more */
    sh[r].lo = r * (N / R) + (r < N % R ? r : N % R);
    sh[r].hi = sh[r].lo + N / R + (r < N % R ? 1 : 0);
    sh[r].device = r % n_dev;
    if (pthread_create(&th[r], NULL, run_shard, &sh[r]) != 0) return 5;
  }
  int bad = 0;
  for (int r = 0; r < R; ++r) {
    pthread_join(th[r], NULL);
    if (sh[r].rc != CRISPY_OK) {
      fprintf(stderr, "shard %d (chunks %d..%d, device %d): status %d: %s\n", r, sh[r].lo, sh[r].hi, sh[r].device, sh[r].rc, sh[r].err);
      bad = 1;
    }
  }
  if (bad) return 6;
  printf("{\"devices\": %d, \"shards\": %d, \"chunks\": [", n_dev, R);
  for (int i = 0; i < N; ++i) {
    printf("%s{\"lang\": %d, \"tokens\": [", i ? ", " : "", out[i].lang);
    for (int k = 0; k < out[i].n; ++k) printf("%s%d", k ? ", " : "", out[i].tok[k]);
    printf("]}");
  }
  printf("]}\n");
  free(pcm);
  free(out);
  return 0;
}
