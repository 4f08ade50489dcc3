/*
 * rnnoise_oracle.h -- TEST INFRASTRUCTURE ONLY (parity oracle).
 *
 * CPU restatement of the RNNoise frame algorithm that crispy reaches through
 *   nnnoiseless::DenoiseState::process_frame   (call site: src-tauri/src/audio.rs:268,
 *   ctor audio.rs:229, crate pinned at nnnoiseless 0.5.2 in Cargo.lock:2825-2838).
 *
 * PARITY UNPINNED: the crate source is not vendored under /root/reference, there is no
 * Rust toolchain in this image and the reference has no test/golden vector for this
 * call.  This file restates the published algorithm
 * (xiph/rnnoise denoise.c / pitch.c / celt_lpc.c / rnn.c, which nnnoiseless ports) and is
 * anchored only on the reference's call-site contract: 480-sample f32 frames in int16
 * range, returns the VAD probability, state owned by the object.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this.
 * The product (libcrispy_hip.so) never links or calls it.
 */
#ifndef RNNOISE_ORACLE_H
#define RNNOISE_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RNO_FRAME_SIZE 480
#define RNO_WINDOW_SIZE 960
#define RNO_FREQ_SIZE 481
#define RNO_NB_BANDS 22
#define RNO_NB_FEATURES 42
#define RNO_PITCH_BUF_SIZE 1728
#define RNO_WEIGHTS_BYTES 87503
#define RNO_TAPS 72
/* debug capture of the most recent frame (floats): */
#define RNO_DBG_X 0        /* analysis spectrum, 481 interleaved re/im */
#define RNO_DBG_EX 962     /* 22 */
#define RNO_DBG_LP 984     /* 864 whitened half-rate pitch buffer */
#define RNO_DBG_MISC 1848  /* [0] pitch_index before remove_doubling */
#define RNO_DBG_P 1856     /* pitch-frame spectrum, 481 interleaved */
#define RNO_DBG_EP 2818    /* 22 */
#define RNO_DBG_EXP 2840   /* 22 */
#define RNO_DBG_XOUT 2862  /* spectrum handed to synthesis, 481 interleaved */
#define RNO_DBG_HP 3824    /* 480 high-passed input */
#define RNO_DBG_FLOATS 4304

typedef struct rno_state rno_state;

/* weights: flat int8 blob, layer order input_dense, vad_gru, vad_output, noise_gru,
 * denoise_gru, denoise_output; inside a layer: input weights [in][out] (GRU: [in][3N],
 * gate order z,r,h), recurrent weights [N][3N], bias.   */
rno_state *rno_create(const int8_t *weights, size_t nbytes);
void rno_destroy(rno_state *st);
void rno_reset(rno_state *st);

/* DenoiseState::process_frame: returns the VAD probability. */
float rno_process_frame(rno_state *st, float *out, const float *in);

/* Convenience loop used by the cpu_baseline timer: n_frames consecutive frames. */
void rno_process_frames(rno_state *st, float *out, const float *in, int n_frames, float *vad);

/* taps of the most recent frame: [0..41] features, [42..63] gains (after the 0.6 decay max),
 * [64] pitch_index, [65] pitch gain, [66] vad, [67] silence flag, [68..71] reserved. */
void rno_last_taps(const rno_state *st, float *taps);
/* smallest gap at any comparison that decided the most recent frame's pitch index, in units in which the difference
 * between two correct f32 implementations is O(1e-6) (rnnoise_oracle.c: margin_note): test instrumentation */
float rno_last_pitch_margin(const rno_state *st);
int rno_last_pitch_margin_site(const rno_state *st);
void rno_last_debug(const rno_state *st, float *dbg);

/* ---- stage entry points (unit tests pin these against numpy/scipy) ---- */
void rno_forward_transform(float *out_re, float *out_im, const float *in960);
void rno_inverse_transform(float *out960, const float *in_re, const float *in_im);
void rno_biquad(float *y, float mem[2], const float *x, int n);
void rno_band_energy(float *bandE, const float *re, const float *im);
void rno_interp_band_gain(float *g481, const float *bandE);
void rno_dct(float *out22, const float *in22);
void rno_half_window(float *w480);
float rno_tansig_approx(float x);
float rno_sigmoid_approx(float x);
void rno_pitch_downsample(const float *x1728, float *x_lp864);
int rno_pitch_search(const float *x_lp, const float *y, int len, int max_pitch);
float rno_remove_doubling(const float *x, int maxperiod, int minperiod, int N, int *T0,
                          int prev_period, float prev_gain);
/* one RNN step on explicit state: state168 = vad(24) | noise(48) | denoise(96) */
void rno_compute_rnn(const int8_t *weights, float *state168, float *gains22, float *vad,
                     const float *features42);

#ifdef __cplusplus
}
#endif
#endif
