// asr_common.h -- shared declarations of the log-mel / Whisper path.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace crispy {

constexpr int MEL_FRAMES = 3000;   // frames the encoder consumes (30 s)
constexpr int MEL_TILE = 64;
constexpr int MEL_TILES = 47;      // 3008 frames computed: the clip maximum also sees the tail frames
constexpr int MEL_BINS = 201;
constexpr int MEL_MAX_MELS = 128;

struct MelTables {
  float hann[400];
  float2 w400[400];                // exp(-2 pi i k / 400)
  int f_start[MEL_MAX_MELS], f_len[MEL_MAX_MELS], f_off[MEL_MAX_MELS];
  float f_w[MEL_MAX_MELS * 64];    // concatenated non-zero filter weights
};

struct MelArgs {
  const float* pcm;        // [batch][pcm_stride] 16 kHz f32
  long pcm_stride;
  const int* n_samples;    // [batch] (device)
  int n_mel;
  const MelTables* tab;
  float* raw;              // [batch][n_mel][3000] log10 values before normalisation
  int* clip_max;           // [batch] order-preserving int key of the clip maximum
  float* out;              // [batch][n_mel][3000] or null
  float* out_t;            // [batch][3002][n_mel] zero padded frame-major copy or null
};

hipError_t mel_launch(const MelArgs& a, int batch, hipStream_t s);

}  // namespace crispy
