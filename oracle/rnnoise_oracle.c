/*
 * rnnoise_oracle.c -- TEST INFRASTRUCTURE ONLY (parity oracle), see rnnoise_oracle.h.
 *
 * PARITY UNPINNED (D1, D5, D7): restates the published RNNoise
 * algorithm behind nnnoiseless 0.5.2 `DenoiseState::process_frame`
 * (reference call site src-tauri/src/audio.rs:268; ctor audio.rs:229).
 *
 * Every function below names the step of SURVEY.md Appendix A it follows.  Single
 * precision throughout, strictly sequential sums, built with -ffp-contract=off so that a
 * multiply-add is two roundings (Rust never contracts).  The biquad keeps xiph's double
 * intermediates (Appendix A.3 step 1).
 */
#include "rnnoise_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#define FRAME_SIZE RNO_FRAME_SIZE
#define WINDOW_SIZE RNO_WINDOW_SIZE
#define FREQ_SIZE RNO_FREQ_SIZE
#define NB_BANDS RNO_NB_BANDS
#define NB_FEATURES RNO_NB_FEATURES
#define CEPS_MEM 8
#define NB_DELTA_CEPS 6
#define PITCH_MIN_PERIOD 60
#define PITCH_MAX_PERIOD 768
#define PITCH_FRAME_SIZE 960
#define PITCH_BUF_SIZE RNO_PITCH_BUF_SIZE
#define FRAME_SIZE_SHIFT 2

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

/* Appendix A.1: Opus band edges in units of 4 FFT bins. */
static const int eband5ms[NB_BANDS] = {0,  1,  2,  3,  4,  5,  6,  7,  8,  10, 12,
                                       14, 16, 20, 24, 28, 34, 40, 48, 60, 78, 100};

/* ------------------------------------------------------------------------------------ */
/* tables                                                                               */
/* ------------------------------------------------------------------------------------ */
static int g_tables_ready = 0;
static float g_half_window[FRAME_SIZE];
static float g_dct_table[NB_BANDS * NB_BANDS];
static float g_tansig_table[201];
static float g_w960_re[WINDOW_SIZE], g_w960_im[WINDOW_SIZE]; /* exp(-2*pi*i*k/960) */

static void init_tables(void) {
  if (g_tables_ready) return;
  for (int i = 0; i < FRAME_SIZE; i++) {
    double s = sin(.5 * M_PI * (i + .5) / FRAME_SIZE);
    g_half_window[i] = (float)sin(.5 * M_PI * s * s);
  }
  for (int i = 0; i < NB_BANDS; i++)
    for (int j = 0; j < NB_BANDS; j++) {
      double v = cos((i + .5) * j * M_PI / NB_BANDS);
      if (j == 0) v *= sqrt(.5);
      g_dct_table[i * NB_BANDS + j] = (float)v;
    }
  /* Appendix A.3 step 6: 201-entry table of tanh(0.04 i) printed with 6 decimals. */
  for (int i = 0; i <= 200; i++) g_tansig_table[i] = (float)(floor(tanh(0.04 * i) * 1e6 + 0.5) / 1e6);
  for (int k = 0; k < WINDOW_SIZE; k++) {
    g_w960_re[k] = (float)cos(-2.0 * M_PI * k / WINDOW_SIZE);
    g_w960_im[k] = (float)sin(-2.0 * M_PI * k / WINDOW_SIZE);
  }
  g_tables_ready = 1;
}

void rno_half_window(float *w480) {
  init_tables();
  memcpy(w480, g_half_window, sizeof(g_half_window));
}

/* ------------------------------------------------------------------------------------ */
/* FFT: 960-point real transform through a 480-point complex Stockham FFT.              */
/* Convention (Appendix A.1): forward = DFT/960, inverse = unscaled inverse DFT.        */
/* ------------------------------------------------------------------------------------ */
#define NC 480
static const int fft_radices[5] = {4, 4, 2, 3, 5};

typedef struct { float re, im; } cpx;

static inline cpx cmul(cpx a, cpx b) {
  cpx r = {a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re};
  return r;
}
static inline cpx tw480(int idx) { /* exp(-2 pi i idx / 480) */
  idx %= NC;
  cpx r = {g_w960_re[2 * idx], g_w960_im[2 * idx]};
  return r;
}

/* in-order forward 480-point complex DFT, unscaled; `a` is overwritten with the result */
static void fft480(cpx *a) {
  cpx b[NC];
  cpx *src = a, *dst = b;
  int Ns = 1;
  for (int p = 0; p < 5; p++) {
    const int R = fft_radices[p];
    const int M = NC / R;
    for (int j = 0; j < M; j++) {
      const int k = j % Ns;
      cpx v[5], o[5];
      for (int r = 0; r < R; r++) {
        cpx x = src[j + r * M];
        v[r] = (k == 0 || r == 0) ? x : cmul(x, tw480(k * r * (NC / (Ns * R))));
      }
      if (R == 2) {
        o[0].re = v[0].re + v[1].re; o[0].im = v[0].im + v[1].im;
        o[1].re = v[0].re - v[1].re; o[1].im = v[0].im - v[1].im;
      } else if (R == 4) {
        cpx s0 = {v[0].re + v[2].re, v[0].im + v[2].im};
        cpx d0 = {v[0].re - v[2].re, v[0].im - v[2].im};
        cpx s1 = {v[1].re + v[3].re, v[1].im + v[3].im};
        cpx d1 = {v[1].re - v[3].re, v[1].im - v[3].im};
        o[0].re = s0.re + s1.re; o[0].im = s0.im + s1.im;
        o[2].re = s0.re - s1.re; o[2].im = s0.im - s1.im;
        /* -i * d1 = (d1.im, -d1.re) */
        o[1].re = d0.re + d1.im; o[1].im = d0.im - d1.re;
        o[3].re = d0.re - d1.im; o[3].im = d0.im + d1.re;
      } else {
        for (int q = 0; q < R; q++) {
          cpx acc = v[0];
          for (int r = 1; r < R; r++) {
            cpx t = cmul(v[r], tw480(((q * r) % R) * (NC / R)));
            acc.re += t.re; acc.im += t.im;
          }
          o[q] = acc;
        }
      }
      const int j0 = (j / Ns) * Ns * R + k;
      for (int r = 0; r < R; r++) dst[j0 + r * Ns] = o[r];
    }
    Ns *= R;
    cpx *t = src; src = dst; dst = t;
  }
  if (src != a) memcpy(a, src, sizeof(cpx) * NC);
}

/* forward_transform of denoise.c: X[k] = DFT(x)[k] / 960 for k = 0..480 */
void rno_forward_transform(float *out_re, float *out_im, const float *in) {
  init_tables();
  cpx z[NC];
  for (int n = 0; n < NC; n++) { z[n].re = in[2 * n]; z[n].im = in[2 * n + 1]; }
  fft480(z);
  const float scale = 1.0f / WINDOW_SIZE;
  for (int k = 0; k <= NC; k++) {
    cpx zk = z[k % NC];
    cpx zc = z[(NC - k) % NC]; zc.im = -zc.im;
    float fe_re = 0.5f * (zk.re + zc.re), fe_im = 0.5f * (zk.im + zc.im);
    /* Fo = -i (zk - zc) / 2 */
    float dr = 0.5f * (zk.re - zc.re), di = 0.5f * (zk.im - zc.im);
    cpx fo = {di, -dr};
    cpx w = {g_w960_re[k], g_w960_im[k]};
    cpx t = cmul(w, fo);
    out_re[k] = (fe_re + t.re) * scale;
    out_im[k] = (fe_im + t.im) * scale;
  }
}

/* inverse_transform of denoise.c: x[n] = sum over the Hermitian-completed spectrum, unscaled */
void rno_inverse_transform(float *out, const float *in_re, const float *in_im) {
  init_tables();
  cpx z[NC];
  for (int k = 0; k < NC; k++) {
    cpx a = {in_re[k], in_im[k]};
    cpx b = {in_re[NC - k], -in_im[NC - k]};
    cpx fe = {a.re + b.re, a.im + b.im};
    cpx d = {a.re - b.re, a.im - b.im};
    cpx w = {g_w960_re[k], -g_w960_im[k]}; /* exp(+2 pi i k / 960) */
    cpx fo = cmul(d, w);
    /* Z = Fe + i Fo; conjugate so that a forward FFT performs the inverse */
    z[k].re = fe.re - fo.im;
    z[k].im = -(fe.im + fo.re);
  }
  fft480(z);
  for (int n = 0; n < NC; n++) { out[2 * n] = z[n].re; out[2 * n + 1] = -z[n].im; }
}

/* ------------------------------------------------------------------------------------ */
/* denoise.c helpers                                                                    */
/* ------------------------------------------------------------------------------------ */
/* Appendix A.3 step 1: DF-II transposed high-pass, products in double as in xiph's C. */
void rno_biquad(float *y, float mem[2], const float *x, int n) {
  static const float a_hp[2] = {-1.99599f, 0.99600f};
  static const float b_hp[2] = {-2.f, 1.f};
  for (int i = 0; i < n; i++) {
    float xi = x[i];
    float yi = x[i] + mem[0];
    mem[0] = (float)(mem[1] + (b_hp[0] * (double)xi - a_hp[0] * (double)yi));
    mem[1] = (float)(b_hp[1] * (double)xi - a_hp[1] * (double)yi);
    y[i] = yi;
  }
}

static void apply_window(float *x) {
  for (int i = 0; i < FRAME_SIZE; i++) {
    x[i] *= g_half_window[i];
    x[WINDOW_SIZE - 1 - i] *= g_half_window[i];
  }
}

/* Appendix A.3 step 2: triangular Opus-band energies */
void rno_band_energy(float *bandE, const float *re, const float *im) {
  float sum[NB_BANDS] = {0};
  for (int i = 0; i < NB_BANDS - 1; i++) {
    int band_size = (eband5ms[i + 1] - eband5ms[i]) << FRAME_SIZE_SHIFT;
    for (int j = 0; j < band_size; j++) {
      float frac = (float)j / band_size;
      int b = (eband5ms[i] << FRAME_SIZE_SHIFT) + j;
      float tmp = re[b] * re[b];
      tmp += im[b] * im[b];
      sum[i] += (1 - frac) * tmp;
      sum[i + 1] += frac * tmp;
    }
  }
  sum[0] *= 2;
  sum[NB_BANDS - 1] *= 2;
  memcpy(bandE, sum, sizeof(sum));
}

static void band_corr(float *bandE, const float *xr, const float *xi, const float *pr,
                      const float *pi) {
  float sum[NB_BANDS] = {0};
  for (int i = 0; i < NB_BANDS - 1; i++) {
    int band_size = (eband5ms[i + 1] - eband5ms[i]) << FRAME_SIZE_SHIFT;
    for (int j = 0; j < band_size; j++) {
      float frac = (float)j / band_size;
      int b = (eband5ms[i] << FRAME_SIZE_SHIFT) + j;
      float tmp = xr[b] * pr[b];
      tmp += xi[b] * pi[b];
      sum[i] += (1 - frac) * tmp;
      sum[i + 1] += frac * tmp;
    }
  }
  sum[0] *= 2;
  sum[NB_BANDS - 1] *= 2;
  memcpy(bandE, sum, sizeof(sum));
}

/* Appendix A.3 step 7: 22 band gains -> 481 bin gains, zero above bin 400 */
void rno_interp_band_gain(float *g, const float *bandE) {
  memset(g, 0, sizeof(float) * FREQ_SIZE);
  for (int i = 0; i < NB_BANDS - 1; i++) {
    int band_size = (eband5ms[i + 1] - eband5ms[i]) << FRAME_SIZE_SHIFT;
    for (int j = 0; j < band_size; j++) {
      float frac = (float)j / band_size;
      g[(eband5ms[i] << FRAME_SIZE_SHIFT) + j] = (1 - frac) * bandE[i] + frac * bandE[i + 1];
    }
  }
}

void rno_dct(float *out, const float *in) {
  init_tables();
  const float norm = (float)sqrt(2. / 22);
  for (int i = 0; i < NB_BANDS; i++) {
    float sum = 0;
    for (int j = 0; j < NB_BANDS; j++) sum += in[j] * g_dct_table[j * NB_BANDS + i];
    out[i] = sum * norm;
  }
}

/* ------------------------------------------------------------------------------------ */
/* pitch.c / celt_lpc.c  (Appendix A.3 step 3)                                          */
/* ------------------------------------------------------------------------------------ */
static float inner_prod(const float *x, const float *y, int n) {
  float s = 0;
  for (int i = 0; i < n; i++) s += x[i] * y[i];
  return s;
}

static void pitch_xcorr(const float *x, const float *y, float *xcorr, int len, int max_pitch) {
  for (int i = 0; i < max_pitch; i++) xcorr[i] = inner_prod(x, y + i, len);
}

static void celt_lpc4(float *lpc, const float *ac) {
  const int p = 4;
  float error = ac[0];
  for (int i = 0; i < p; i++) lpc[i] = 0;
  if (ac[0] != 0) {
    for (int i = 0; i < p; i++) {
      float rr = 0;
      for (int j = 0; j < i; j++) rr += lpc[j] * ac[i - j];
      rr += ac[i + 1];
      float r = -rr / error;
      lpc[i] = r;
      for (int j = 0; j < (i + 1) >> 1; j++) {
        float tmp1 = lpc[j];
        float tmp2 = lpc[i - 1 - j];
        lpc[j] = tmp1 + r * tmp2;
        lpc[i - 1 - j] = tmp2 + r * tmp1;
      }
      error = error - r * r * error;
      if (error < .001f * ac[0]) break;
    }
  }
}

void rno_pitch_downsample(const float *x, float *x_lp) {
  const int len = PITCH_BUF_SIZE;
  const int n = len >> 1;
  float ac[5];
  float lpc[4], lpc2[5];
  float tmp = 1.f;
  const float c1 = .8f;
  for (int i = 1; i < n; i++) x_lp[i] = .5f * (.5f * (x[2 * i - 1] + x[2 * i + 1]) + x[2 * i]);
  x_lp[0] = .5f * (.5f * x[1] + x[0]);

  /* _celt_autocorr with lag 4, no window */
  {
    const int lag = 4, fastN = n - lag;
    pitch_xcorr(x_lp, x_lp, ac, fastN, lag + 1);
    for (int k = 0; k <= lag; k++) {
      float d = 0;
      for (int i = k + fastN; i < n; i++) d += x_lp[i] * x_lp[i - k];
      ac[k] += d;
    }
  }
  ac[0] *= 1.0001f;
  for (int i = 1; i <= 4; i++) ac[i] -= ac[i] * (.008f * i) * (.008f * i);
  celt_lpc4(lpc, ac);
  for (int i = 0; i < 4; i++) {
    tmp = .9f * tmp;
    lpc[i] = lpc[i] * tmp;
  }
  lpc2[0] = lpc[0] + .8f;
  lpc2[1] = lpc[1] + c1 * lpc[0];
  lpc2[2] = lpc[2] + c1 * lpc[1];
  lpc2[3] = lpc[3] + c1 * lpc[2];
  lpc2[4] = c1 * lpc[3];
  /* celt_fir5 in place */
  {
    float mem0 = 0, mem1 = 0, mem2 = 0, mem3 = 0, mem4 = 0;
    for (int i = 0; i < n; i++) {
      float xi = x_lp[i];
      float sum = xi;
      sum += lpc2[0] * mem0;
      sum += lpc2[1] * mem1;
      sum += lpc2[2] * mem2;
      sum += lpc2[3] * mem3;
      sum += lpc2[4] * mem4;
      mem4 = mem3; mem3 = mem2; mem2 = mem1; mem1 = mem0; mem0 = xi;
      x_lp[i] = sum;
    }
  }
}

/* ---- decision margin of the pitch analysis (test instrumentation, no effect on any result) -------------------------
 * The pitch index is an integer picked by comparisons between correlations.  Another correct f32 implementation adds the
 * same products in another order, so its correlations differ by ~1e-6 of their Cauchy-Schwarz scale sqrt(Sxx Syy) -- and
 * where two compared quantities sit closer than that, it may legitimately pick the other side.  g_pitch_margin is, for
 * the frame being analysed, the SMALLEST gap at any comparison that decides the index, each in units in which such an
 * error is O(1e-6): cos^2 = xcorr^2 / (Sxx Syy) for the rankings of find_best_pitch (gap between the last candidate kept
 * and the best one dropped), correlation / sqrt(Sxx Syy) for the two interpolation tests, the pitch gain (already a
 * cosine) for remove_doubling's `g1 > thresh`.  Thread-local: the oracle is also timed on many threads (bench.py).
 * tests/test_gpu_rnnoise.py: the HIP kernel's pitch index must EQUAL the oracle's on every frame whose margin exceeds 1e-5. */
static __thread float g_pitch_margin;
static __thread int g_pitch_margin_site;      /* which comparison it was: 1 coarse ranking, 2 fine ranking, 3 sign test, 4 / 6 interpolation, 5 g1 > thresh */
static void margin_note_at(float gap, int site) {
  if (gap < 0) gap = -gap;
  if (gap < g_pitch_margin) { g_pitch_margin = gap; g_pitch_margin_site = site; }
}
/* gap protecting the `keep` best candidates of a find_best_pitch pass: cos^2 of the keep-th best minus the next one */
static void margin_ranking(const float *xcorr, const float *y, int len, int max_pitch, float Sxx, int keep) {
  float top[3] = {0, 0, 0};
  float Syy0 = 1, Syy;
  if (!(Sxx > 0)) return;                       /* an all-zero frame: every sum is exactly zero in any order */
  for (int j = 0; j < len; j++) Syy0 += y[j] * y[j];
  Syy = Syy0;
  for (int i = 0; i < max_pitch; i++) {
    if (xcorr[i] > 0) {
      const float c = (xcorr[i] / Sxx) * (xcorr[i] / Syy);
      if (c > top[0]) { top[2] = top[1]; top[1] = top[0]; top[0] = c; }
      else if (c > top[1]) { top[2] = top[1]; top[1] = c; }
      else if (c > top[2]) top[2] = c;
    }
    Syy += y[i + len] * y[i + len] - y[i] * y[i];
    if (Syy < 1) Syy = 1;
  }
  /* the sign test `xcorr > 0` is a decision too -- for a lag that would rank among the kept ones if it passed */
  Syy = Syy0;
  for (int i = 0; i < max_pitch; i++) {
    if (!(xcorr[i] > 0)) {
      const float c = (xcorr[i] / Sxx) * (xcorr[i] / Syy);
      if (c > top[keep - 1]) margin_note_at((float)sqrt(c), 3);
    }
    Syy += y[i + len] * y[i + len] - y[i] * y[i];
    if (Syy < 1) Syy = 1;
  }
  if (top[keep - 1] > 0) margin_note_at(top[keep - 1] - top[keep], keep == 2 ? 1 : 2);   /* fewer positive lags than kept: the sign tests are the decisions */
}
static void margin_interp(float a, float b, float c, float scale, int site) {
  if (!(scale > 0)) return;
  margin_note_at(((c - a) - .7f * (b - a)) / scale, site);
  margin_note_at(((a - c) - .7f * (b - c)) / scale, site);
}

static void find_best_pitch(const float *xcorr, const float *y, int len, int max_pitch,
                            int *best_pitch) {
  float Syy = 1;
  float best_num[2] = {-1, -1};
  float best_den[2] = {0, 0};
  best_pitch[0] = 0;
  best_pitch[1] = 1;
  for (int j = 0; j < len; j++) Syy += y[j] * y[j];
  for (int i = 0; i < max_pitch; i++) {
    if (xcorr[i] > 0) {
      float xcorr16 = xcorr[i];
      xcorr16 *= 1e-12f;
      float num = xcorr16 * xcorr16;
      if (num * best_den[1] > best_num[1] * Syy) {
        if (num * best_den[0] > best_num[0] * Syy) {
          best_num[1] = best_num[0]; best_den[1] = best_den[0]; best_pitch[1] = best_pitch[0];
          best_num[0] = num; best_den[0] = Syy; best_pitch[0] = i;
        } else {
          best_num[1] = num; best_den[1] = Syy; best_pitch[1] = i;
        }
      }
    }
    Syy += y[i + len] * y[i + len] - y[i] * y[i];
    if (Syy < 1) Syy = 1;
  }
}

int rno_pitch_search(const float *x_lp, const float *y, int len, int max_pitch) {
  const int lag = len + max_pitch;
  int best_pitch[2] = {0, 0};
  int offset;
  float x_lp4[PITCH_FRAME_SIZE >> 2];
  float y_lp4[(PITCH_FRAME_SIZE + PITCH_MAX_PERIOD) >> 2];
  float xcorr[PITCH_MAX_PERIOD >> 1] = {0};   /* pitch_xcorr fills the first max_pitch >> 2 entries */
  for (int j = 0; j < len >> 2; j++) x_lp4[j] = x_lp[2 * j];
  for (int j = 0; j < lag >> 2; j++) y_lp4[j] = y[2 * j];
  pitch_xcorr(x_lp4, y_lp4, xcorr, len >> 2, max_pitch >> 2);
  find_best_pitch(xcorr, y_lp4, len >> 2, max_pitch >> 2, best_pitch);
  margin_ranking(xcorr, y_lp4, len >> 2, max_pitch >> 2, inner_prod(x_lp4, x_lp4, len >> 2), 2);   /* both of the top two are used */
  for (int i = 0; i < max_pitch >> 1; i++) {
    xcorr[i] = 0;
    if (abs(i - 2 * best_pitch[0]) > 2 && abs(i - 2 * best_pitch[1]) > 2) continue;
    float sum = inner_prod(x_lp, y + i, len >> 1);
    xcorr[i] = sum < -1 ? -1 : sum;
  }
  find_best_pitch(xcorr, y, len >> 1, max_pitch >> 1, best_pitch);
  {
    const float Sxx = inner_prod(x_lp, x_lp, len >> 1);
    margin_ranking(xcorr, y, len >> 1, max_pitch >> 1, Sxx, 1);                                       /* only the best is used */
    if (best_pitch[0] > 0 && best_pitch[0] < (max_pitch >> 1) - 1)
      margin_interp(xcorr[best_pitch[0] - 1], xcorr[best_pitch[0]], xcorr[best_pitch[0] + 1],
                    (float)sqrt(Sxx * (1 + inner_prod(y + best_pitch[0], y + best_pitch[0], len >> 1))), 4);
  }
  if (best_pitch[0] > 0 && best_pitch[0] < (max_pitch >> 1) - 1) {
    float a = xcorr[best_pitch[0] - 1];
    float b = xcorr[best_pitch[0]];
    float c = xcorr[best_pitch[0] + 1];
    if ((c - a) > .7f * (b - a)) offset = 1;
    else if ((a - c) > .7f * (b - c)) offset = -1;
    else offset = 0;
  } else {
    offset = 0;
  }
  return 2 * best_pitch[0] - offset;
}

static float compute_pitch_gain(float xy, float xx, float yy) {
  return xy / (float)sqrt(1 + xx * yy);
}

static const int second_check[16] = {0, 0, 3, 2, 3, 2, 5, 2, 3, 2, 3, 2, 5, 2, 3, 2};

float rno_remove_doubling(const float *x, int maxperiod, int minperiod, int N, int *T0_,
                          int prev_period, float prev_gain) {
  int k, i, T, T0;
  float g, g0, pg;
  float xy, xx, yy, xy2;
  float xcorr[3];
  float best_xy, best_yy;
  int offset;
  const int minperiod0 = minperiod;
  float yy_lookup[(PITCH_MAX_PERIOD >> 1) + 1];
  maxperiod /= 2;
  minperiod /= 2;
  *T0_ /= 2;
  prev_period /= 2;
  N /= 2;
  x += maxperiod;
  if (*T0_ >= maxperiod) *T0_ = maxperiod - 1;
  T = T0 = *T0_;
  xx = 0; xy = 0;
  for (i = 0; i < N; i++) { xx += x[i] * x[i]; xy += x[i] * x[i - T0]; }
  yy_lookup[0] = xx;
  yy = xx;
  for (i = 1; i <= maxperiod; i++) {
    yy = yy + x[-i] * x[-i] - x[N - i] * x[N - i];
    yy_lookup[i] = yy < 0 ? 0 : yy;
  }
  yy = yy_lookup[T0];
  best_xy = xy;
  best_yy = yy;
  g = g0 = compute_pitch_gain(xy, xx, yy);
  for (k = 2; k <= 15; k++) {
    int T1, T1b;
    float g1, cont, thresh;
    T1 = (2 * T0 + k) / (2 * k);
    if (T1 < minperiod) break;
    if (k == 2) {
      if (T1 + T0 > maxperiod) T1b = T0;
      else T1b = T0 + T1;
    } else {
      T1b = (2 * second_check[k] * T0 + k) / (2 * k);
    }
    xy = 0; xy2 = 0;
    for (i = 0; i < N; i++) { xy += x[i] * x[i - T1]; xy2 += x[i] * x[i - T1b]; }
    xy = .5f * (xy + xy2);
    yy = .5f * (yy_lookup[T1] + yy_lookup[T1b]);
    g1 = compute_pitch_gain(xy, xx, yy);
    if (abs(T1 - prev_period) <= 1) cont = prev_gain;
    else if (abs(T1 - prev_period) <= 2 && 5 * k * k < T0) cont = .5f * prev_gain;
    else cont = 0;
    thresh = .7f * g0 - cont; if (thresh < .3f) thresh = .3f;
    if (T1 < 3 * minperiod) { thresh = .85f * g0 - cont; if (thresh < .4f) thresh = .4f; }
    else if (T1 < 2 * minperiod) { thresh = .9f * g0 - cont; if (thresh < .5f) thresh = .5f; }
    margin_note_at(g1 - thresh, 5);
    if (g1 > thresh) { best_xy = xy; best_yy = yy; T = T1; g = g1; }
  }
  if (best_xy < 0) best_xy = 0;
  if (best_yy <= best_xy) pg = 1.f;
  else pg = best_xy / (best_yy + 1);
  for (k = 0; k < 3; k++) xcorr[k] = inner_prod(x, x - (T + k - 1), N);
  margin_interp(xcorr[0], xcorr[1], xcorr[2], (float)sqrt(xx * (1 + yy_lookup[T])), 6);
  if ((xcorr[2] - xcorr[0]) > .7f * (xcorr[1] - xcorr[0])) offset = 1;
  else if ((xcorr[0] - xcorr[2]) > .7f * (xcorr[1] - xcorr[2])) offset = -1;
  else offset = 0;
  if (pg > g) pg = g;
  *T0_ = 2 * T + offset;
  if (*T0_ < minperiod0) *T0_ = minperiod0;
  return pg;
}

/* ------------------------------------------------------------------------------------ */
/* rnn.c (Appendix A.3 step 6)                                                          */
/* ------------------------------------------------------------------------------------ */
#define WEIGHTS_SCALE (1.f / 256)
enum { ACT_TANH = 0, ACT_SIGMOID = 1, ACT_RELU = 2 };

float rno_tansig_approx(float x) {
  init_tables();
  float sign = 1, y, dy;
  int i;
  if (!(x < 8)) return 1;
  if (!(x > -8)) return -1;
  if (x != x) return 0;
  if (x < 0) { x = -x; sign = -1; }
  i = (int)floorf(.5f + 25 * x);
  x -= .04f * i;
  y = g_tansig_table[i];
  dy = 1 - y * y;
  y = y + x * dy * (1 - y * x);
  return sign * y;
}

float rno_sigmoid_approx(float x) { return .5f + .5f * rno_tansig_approx(.5f * x); }

static float act(int a, float x) {
  if (a == ACT_SIGMOID) return rno_sigmoid_approx(x);
  if (a == ACT_TANH) return rno_tansig_approx(x);
  return x < 0 ? 0 : x;
}

/* layout of the flat blob (Appendix A.5) */
enum {
  OFF_ID_W = 0, OFF_ID_B = OFF_ID_W + 42 * 24,
  OFF_VG_W = OFF_ID_B + 24, OFF_VG_R = OFF_VG_W + 24 * 72, OFF_VG_B = OFF_VG_R + 24 * 72,
  OFF_VO_W = OFF_VG_B + 72, OFF_VO_B = OFF_VO_W + 24,
  OFF_NG_W = OFF_VO_B + 1, OFF_NG_R = OFF_NG_W + 90 * 144, OFF_NG_B = OFF_NG_R + 48 * 144,
  OFF_DG_W = OFF_NG_B + 144, OFF_DG_R = OFF_DG_W + 114 * 288, OFF_DG_B = OFF_DG_R + 96 * 288,
  OFF_DO_W = OFF_DG_B + 288, OFF_DO_B = OFF_DO_W + 96 * 22,
  OFF_END = OFF_DO_B + 22
};
typedef char blob_size_check[(OFF_END == RNO_WEIGHTS_BYTES) ? 1 : -1];

static void compute_dense(const int8_t *W, const int8_t *bias, int M, int N, int activation,
                          float *output, const float *input) {
  const int stride = N;
  for (int i = 0; i < N; i++) {
    float sum = bias[i];
    for (int j = 0; j < M; j++) sum += W[j * stride + i] * input[j];
    output[i] = act(activation, WEIGHTS_SCALE * sum);
  }
}

static void compute_gru(const int8_t *W, const int8_t *U, const int8_t *bias, int M, int N,
                        int activation, float *state, const float *input) {
  float z[128], r[128], h[128];
  const int stride = 3 * N;
  for (int i = 0; i < N; i++) {
    float sum = bias[i];
    for (int j = 0; j < M; j++) sum += W[j * stride + i] * input[j];
    for (int j = 0; j < N; j++) sum += U[j * stride + i] * state[j];
    z[i] = rno_sigmoid_approx(WEIGHTS_SCALE * sum);
  }
  for (int i = 0; i < N; i++) {
    float sum = bias[N + i];
    for (int j = 0; j < M; j++) sum += W[N + j * stride + i] * input[j];
    for (int j = 0; j < N; j++) sum += U[N + j * stride + i] * state[j];
    r[i] = rno_sigmoid_approx(WEIGHTS_SCALE * sum);
  }
  for (int i = 0; i < N; i++) {
    float sum = bias[2 * N + i];
    for (int j = 0; j < M; j++) sum += W[2 * N + j * stride + i] * input[j];
    for (int j = 0; j < N; j++) sum += U[2 * N + j * stride + i] * state[j] * r[j];
    sum = act(activation, WEIGHTS_SCALE * sum);
    h[i] = z[i] * state[i] + (1 - z[i]) * sum;
  }
  for (int i = 0; i < N; i++) state[i] = h[i];
}

void rno_compute_rnn(const int8_t *w, float *state, float *gains, float *vad, const float *input) {
  float dense_out[24];
  float noise_input[90];
  float denoise_input[114];
  float *vad_state = state, *noise_state = state + 24, *denoise_state = state + 72;
  compute_dense(w + OFF_ID_W, w + OFF_ID_B, 42, 24, ACT_TANH, dense_out, input);
  compute_gru(w + OFF_VG_W, w + OFF_VG_R, w + OFF_VG_B, 24, 24, ACT_RELU, vad_state, dense_out);
  compute_dense(w + OFF_VO_W, w + OFF_VO_B, 24, 1, ACT_SIGMOID, vad, vad_state);
  for (int i = 0; i < 24; i++) noise_input[i] = dense_out[i];
  for (int i = 0; i < 24; i++) noise_input[24 + i] = vad_state[i];
  for (int i = 0; i < 42; i++) noise_input[48 + i] = input[i];
  compute_gru(w + OFF_NG_W, w + OFF_NG_R, w + OFF_NG_B, 90, 48, ACT_RELU, noise_state, noise_input);
  for (int i = 0; i < 24; i++) denoise_input[i] = vad_state[i];
  for (int i = 0; i < 48; i++) denoise_input[24 + i] = noise_state[i];
  for (int i = 0; i < 42; i++) denoise_input[72 + i] = input[i];
  compute_gru(w + OFF_DG_W, w + OFF_DG_R, w + OFF_DG_B, 114, 96, ACT_RELU, denoise_state,
              denoise_input);
  compute_dense(w + OFF_DO_W, w + OFF_DO_B, 96, 22, ACT_SIGMOID, gains, denoise_state);
}

/* ------------------------------------------------------------------------------------ */
/* DenoiseState (Appendix A.2)                                                          */
/* ------------------------------------------------------------------------------------ */
struct rno_state {
  int8_t weights[RNO_WEIGHTS_BYTES];
  float analysis_mem[FRAME_SIZE];
  float cepstral_mem[CEPS_MEM][NB_BANDS];
  int memid;
  float synthesis_mem[FRAME_SIZE];
  float pitch_buf[PITCH_BUF_SIZE];
  float last_gain;
  int last_period;
  float mem_hp_x[2];
  float lastg[NB_BANDS];
  float rnn_state[168];
  float taps[RNO_TAPS];
  float dbg[RNO_DBG_FLOATS];
  float pitch_margin;
  int pitch_margin_site;
};

rno_state *rno_create(const int8_t *weights, size_t nbytes) {
  if (!weights || nbytes != RNO_WEIGHTS_BYTES) return NULL;
  init_tables();
  rno_state *st = (rno_state *)calloc(1, sizeof(rno_state));
  if (!st) return NULL;
  memcpy(st->weights, weights, RNO_WEIGHTS_BYTES);
  return st;
}

void rno_destroy(rno_state *st) { free(st); }

void rno_reset(rno_state *st) {
  memset(st->analysis_mem, 0, sizeof(rno_state) - offsetof(rno_state, analysis_mem));
}

void rno_last_taps(const rno_state *st, float *taps) { memcpy(taps, st->taps, sizeof(st->taps)); }
float rno_last_pitch_margin(const rno_state *st) { return st->pitch_margin; }
int rno_last_pitch_margin_site(const rno_state *st) { return st->pitch_margin_site; }
void rno_last_debug(const rno_state *st, float *dbg) { memcpy(dbg, st->dbg, sizeof(st->dbg)); }

/* Appendix A.3 steps 2-5; returns 1 on the silence branch */
static int compute_frame_features(rno_state *st, float *Xr, float *Xi, float *Pr, float *Pi,
                                  float *Ex, float *Ep, float *Exp, float *features,
                                  const float *in) {
  float E = 0;
  float spec_variability = 0;
  float Ly[NB_BANDS];
  float p[WINDOW_SIZE];
  float pitch_buf[PITCH_BUF_SIZE >> 1];
  float tmp[NB_BANDS];
  float follow, logMax;
  int pitch_index;
  float gain;
  /* frame_analysis */
  {
    float x[WINDOW_SIZE];
    memcpy(x, st->analysis_mem, sizeof(float) * FRAME_SIZE);
    memcpy(x + FRAME_SIZE, in, sizeof(float) * FRAME_SIZE);
    memcpy(st->analysis_mem, in, sizeof(float) * FRAME_SIZE);
    apply_window(x);
    rno_forward_transform(Xr, Xi, x);
    rno_band_energy(Ex, Xr, Xi);
    for (int i = 0; i < FREQ_SIZE; i++) { st->dbg[RNO_DBG_X + 2 * i] = Xr[i]; st->dbg[RNO_DBG_X + 2 * i + 1] = Xi[i]; }
    memcpy(st->dbg + RNO_DBG_EX, Ex, sizeof(float) * NB_BANDS);
  }
  memmove(st->pitch_buf, &st->pitch_buf[FRAME_SIZE], (PITCH_BUF_SIZE - FRAME_SIZE) * sizeof(float));
  memcpy(&st->pitch_buf[PITCH_BUF_SIZE - FRAME_SIZE], in, FRAME_SIZE * sizeof(float));
  rno_pitch_downsample(st->pitch_buf, pitch_buf);
  memcpy(st->dbg + RNO_DBG_LP, pitch_buf, sizeof(pitch_buf));
  g_pitch_margin = 1e30f;
  pitch_index = rno_pitch_search(pitch_buf + (PITCH_MAX_PERIOD >> 1), pitch_buf, PITCH_FRAME_SIZE,
                                 PITCH_MAX_PERIOD - 3 * PITCH_MIN_PERIOD);
  pitch_index = PITCH_MAX_PERIOD - pitch_index;
  st->dbg[RNO_DBG_MISC + 0] = (float)pitch_index;
  gain = rno_remove_doubling(pitch_buf, PITCH_MAX_PERIOD, PITCH_MIN_PERIOD, PITCH_FRAME_SIZE,
                             &pitch_index, st->last_period, st->last_gain);
  st->last_period = pitch_index;
  st->last_gain = gain;
  st->taps[64] = (float)pitch_index;
  st->taps[65] = gain;
  st->pitch_margin_site = g_pitch_margin_site;
  st->pitch_margin = g_pitch_margin;     /* smallest gap at a comparison that decided the index (see margin_note) */
  for (int i = 0; i < WINDOW_SIZE; i++)
    p[i] = st->pitch_buf[PITCH_BUF_SIZE - WINDOW_SIZE - pitch_index + i];
  apply_window(p);
  rno_forward_transform(Pr, Pi, p);
  rno_band_energy(Ep, Pr, Pi);
  band_corr(Exp, Xr, Xi, Pr, Pi);
  for (int i = 0; i < NB_BANDS; i++) Exp[i] = Exp[i] / (float)sqrt(.001f + Ex[i] * Ep[i]);
  for (int i = 0; i < FREQ_SIZE; i++) { st->dbg[RNO_DBG_P + 2 * i] = Pr[i]; st->dbg[RNO_DBG_P + 2 * i + 1] = Pi[i]; }
  memcpy(st->dbg + RNO_DBG_EP, Ep, sizeof(float) * NB_BANDS);
  memcpy(st->dbg + RNO_DBG_EXP, Exp, sizeof(float) * NB_BANDS);
  rno_dct(tmp, Exp);
  for (int i = 0; i < NB_DELTA_CEPS; i++) features[NB_BANDS + 2 * NB_DELTA_CEPS + i] = tmp[i];
  features[NB_BANDS + 2 * NB_DELTA_CEPS] -= 1.3f;
  features[NB_BANDS + 2 * NB_DELTA_CEPS + 1] -= 0.9f;
  features[NB_BANDS + 3 * NB_DELTA_CEPS] = .01f * (pitch_index - 300);
  logMax = -2;
  follow = -2;
  for (int i = 0; i < NB_BANDS; i++) {
    Ly[i] = log10f(1e-2f + Ex[i]);
    float t = follow - 1.5f > Ly[i] ? follow - 1.5f : Ly[i];
    Ly[i] = logMax - 7 > t ? logMax - 7 : t;
    logMax = logMax > Ly[i] ? logMax : Ly[i];
    follow = follow - 1.5f > Ly[i] ? follow - 1.5f : Ly[i];
    E += Ex[i];
  }
  if (E < 0.04f) {
    memset(features, 0, NB_FEATURES * sizeof(float));
    return 1;
  }
  rno_dct(features, Ly);
  features[0] -= 12;
  features[1] -= 4;
  float *ceps_0 = st->cepstral_mem[st->memid];
  float *ceps_1 = (st->memid < 1) ? st->cepstral_mem[CEPS_MEM + st->memid - 1]
                                  : st->cepstral_mem[st->memid - 1];
  float *ceps_2 = (st->memid < 2) ? st->cepstral_mem[CEPS_MEM + st->memid - 2]
                                  : st->cepstral_mem[st->memid - 2];
  for (int i = 0; i < NB_BANDS; i++) ceps_0[i] = features[i];
  st->memid++;
  for (int i = 0; i < NB_DELTA_CEPS; i++) {
    features[i] = ceps_0[i] + ceps_1[i] + ceps_2[i];
    features[NB_BANDS + i] = ceps_0[i] - ceps_2[i];
    features[NB_BANDS + NB_DELTA_CEPS + i] = ceps_0[i] - 2 * ceps_1[i] + ceps_2[i];
  }
  if (st->memid == CEPS_MEM) st->memid = 0;
  for (int i = 0; i < CEPS_MEM; i++) {
    float mindist = 1e15f;
    for (int j = 0; j < CEPS_MEM; j++) {
      float dist = 0;
      for (int k = 0; k < NB_BANDS; k++) {
        float t = st->cepstral_mem[i][k] - st->cepstral_mem[j][k];
        dist += t * t;
      }
      if (j != i) mindist = mindist < dist ? mindist : dist;
    }
    spec_variability += mindist;
  }
  features[NB_BANDS + 3 * NB_DELTA_CEPS + 1] = spec_variability / CEPS_MEM - 2.1f;
  return 0;
}

/* Appendix A.3 step 7 */
static void pitch_filter(float *Xr, float *Xi, const float *Pr, const float *Pi, const float *Ex,
                         const float *Ep, const float *Exp, const float *g) {
  float r[NB_BANDS];
  float rf[FREQ_SIZE];
  float newE[NB_BANDS];
  float norm[NB_BANDS];
  float normf[FREQ_SIZE];
  for (int i = 0; i < NB_BANDS; i++) {
    if (Exp[i] > g[i]) r[i] = 1;
    else
      r[i] = (Exp[i] * Exp[i]) * (1 - g[i] * g[i]) /
             (.001f + (g[i] * g[i]) * (1 - Exp[i] * Exp[i]));
    float c = r[i] < 0 ? 0 : r[i];
    c = c > 1 ? 1 : c;
    r[i] = (float)sqrt(c);
    r[i] *= (float)sqrt(Ex[i] / (1e-8f + Ep[i]));
  }
  rno_interp_band_gain(rf, r);
  for (int i = 0; i < FREQ_SIZE; i++) {
    Xr[i] += rf[i] * Pr[i];
    Xi[i] += rf[i] * Pi[i];
  }
  rno_band_energy(newE, Xr, Xi);
  for (int i = 0; i < NB_BANDS; i++) norm[i] = (float)sqrt(Ex[i] / (1e-8f + newE[i]));
  rno_interp_band_gain(normf, norm);
  for (int i = 0; i < FREQ_SIZE; i++) {
    Xr[i] *= normf[i];
    Xi[i] *= normf[i];
  }
}

float rno_process_frame(rno_state *st, float *out, const float *in) {
  float Xr[FREQ_SIZE], Xi[FREQ_SIZE], Pr[FREQ_SIZE], Pi[FREQ_SIZE];
  float x[FRAME_SIZE];
  float Ex[NB_BANDS], Ep[NB_BANDS], Exp[NB_BANDS];
  float features[NB_FEATURES];
  float g[NB_BANDS];
  float gf[FREQ_SIZE];
  float vad_prob = 0;
  int silence;
  memset(g, 0, sizeof(g));
  rno_biquad(x, st->mem_hp_x, in, FRAME_SIZE);
  silence = compute_frame_features(st, Xr, Xi, Pr, Pi, Ex, Ep, Exp, features, x);
  if (!silence) {
    rno_compute_rnn(st->weights, st->rnn_state, g, &vad_prob, features);
    pitch_filter(Xr, Xi, Pr, Pi, Ex, Ep, Exp, g);
    for (int i = 0; i < NB_BANDS; i++) {
      const float alpha = .6f;
      g[i] = g[i] > alpha * st->lastg[i] ? g[i] : alpha * st->lastg[i];
      st->lastg[i] = g[i];
    }
    rno_interp_band_gain(gf, g);
    for (int i = 0; i < FREQ_SIZE; i++) {
      Xr[i] *= gf[i];
      Xi[i] *= gf[i];
    }
  }
  for (int i = 0; i < FREQ_SIZE; i++) { st->dbg[RNO_DBG_XOUT + 2 * i] = Xr[i]; st->dbg[RNO_DBG_XOUT + 2 * i + 1] = Xi[i]; }
  memcpy(st->dbg + RNO_DBG_HP, x, sizeof(x));
  /* frame_synthesis (Appendix A.3 step 8) */
  {
    float y[WINDOW_SIZE];
    rno_inverse_transform(y, Xr, Xi);
    apply_window(y);
    for (int i = 0; i < FRAME_SIZE; i++) out[i] = y[i] + st->synthesis_mem[i];
    memcpy(st->synthesis_mem, &y[FRAME_SIZE], FRAME_SIZE * sizeof(float));
  }
  memcpy(st->taps, features, sizeof(features));
  memcpy(st->taps + 42, g, sizeof(g));
  st->taps[66] = vad_prob;
  st->taps[67] = (float)silence;
  return vad_prob;
}

void rno_process_frames(rno_state *st, float *out, const float *in, int n_frames, float *vad) {
  for (int t = 0; t < n_frames; t++) {
    float v = rno_process_frame(st, out + (size_t)t * FRAME_SIZE, in + (size_t)t * FRAME_SIZE);
    if (vad) vad[t] = v;
  }
}
