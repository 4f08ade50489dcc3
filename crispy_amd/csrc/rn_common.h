// rn_common.h -- shared host/device declarations for the batched RNNoise path (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace crispy {

constexpr int RN_FRAME = 480;
constexpr int RN_WINDOW = 960;
constexpr int RN_NFREQ = 481;
constexpr int RN_NB = 22;
constexpr int RN_NFEAT = 42;
constexpr int RN_PITCH_BUF = 1728;
constexpr int RN_HIST = 1920;          // 4 frames of high-passed history kept per stream
constexpr int RN_HIST_FRAMES = 4;
constexpr int RN_TAPS = 72;
constexpr int RN_WEIGHT_BYTES = 87503;
constexpr int RN_DBG_FLOATS = 4304;    // mirrors oracle RNO_DBG_*

// Device tables (built on the host in double precision, one copy per handle).
struct RnTables {
  float half_window[RN_FRAME];
  float dct[RN_NB * RN_NB];   // dct[j*22+i] = cos((j+.5) i pi/22), column 0 scaled by sqrt(.5)
  float tansig[208];          // tanh(0.04 i) rounded to 6 decimals, i = 0..200
  float2 w960[RN_WINDOW];     // exp(-2 pi i k / 960)
  float bin_frac[400];        // position of bin inside its Opus band, j / band_size
  int bin_band[400];          // band index of every bin below 20 kHz
  int eband[24];              // band edges in 4-bin chunks (22 used)
  // band_sum (rn_kernels.hip, round 3): the 42 non-empty half-bands -- rising half of band b = chunk partials
  // part_hi[e(b-1) .. e(b)), falling half = part_lo[e(b) .. e(b+1)) -- cut into pieces of <= 6 chunks, one piece per lane,
  // the pieces of a half on neighbouring lanes of one 16-lane row.  band_piece[lane] packs
  //   bits 0-6 first chunk | 7-9 chunks (0: idle lane) | 10 part_lo (else part_hi) | 11 add lane + 1 | 12 then add lane + 2
  //   | 13-18 lane holding the total of the rising half of band `lane` | 19-24 ... of the falling half (lanes < 22)
  int band_piece[64];
};

// Flat blob offsets.
struct RnBlob {
  static constexpr int ID_W = 0, ID_B = ID_W + 42 * 24;
  static constexpr int VG_W = ID_B + 24, VG_R = VG_W + 24 * 72, VG_B = VG_R + 24 * 72;
  static constexpr int VO_W = VG_B + 72, VO_B = VO_W + 24;
  static constexpr int NG_W = VO_B + 1, NG_R = NG_W + 90 * 144, NG_B = NG_R + 48 * 144;
  static constexpr int DG_W = NG_B + 144, DG_R = DG_W + 114 * 288, DG_B = DG_R + 96 * 288;
  static constexpr int DO_W = DG_B + 288, DO_B = DO_W + 96 * 22;
  static constexpr int END = DO_B + 22;
};
static_assert(RnBlob::END == RN_WEIGHT_BYTES, "blob layout");

// Repacked weights: every matrix [K][rows] int8 becomes f16 (int8 values are exact in f16) in units of
// 8 halfs = 16 bytes, laid out [ceil(K/8)][rows][8]: lane == row reads one coalesced 16-byte vector per
// 8 MACs and each MAC is a single v_fma_mix_f32.  Biases are widened to float.
// Matrix offsets are in 16-byte units from the pack base, bias offsets in floats from the pack base.
constexpr int rn_k8(int k) { return (k + 7) / 8; }
struct RnPack {
  static constexpr int ID_W = 0;                               // K=42  rows=24
  static constexpr int VG_W = ID_W + rn_k8(42) * 24;           // K=24  rows=72
  static constexpr int VG_R = VG_W + rn_k8(24) * 72;           // K=24  rows=72
  static constexpr int VO_W = VG_R + rn_k8(24) * 72;           // K=24  rows=1
  static constexpr int NG_W = VO_W + rn_k8(24) * 1;            // K=90  rows=144
  static constexpr int NG_R = NG_W + rn_k8(90) * 144;          // K=48  rows=144
  static constexpr int DG_W = NG_R + rn_k8(48) * 144;          // K=114 rows=288
  static constexpr int DG_R = DG_W + rn_k8(114) * 288;         // K=96  rows=288
  static constexpr int DO_W = DG_R + rn_k8(96) * 288;          // K=96  rows=22
  static constexpr int MAT_END = DO_W + rn_k8(96) * 22;        // 16-byte units
  // float biases (offsets in floats)
  static constexpr int ID_B = MAT_END * 4;
  static constexpr int VG_B = ID_B + 24;
  static constexpr int VO_B = VG_B + 72;
  static constexpr int NG_B = VO_B + 1;
  static constexpr int DG_B = NG_B + 144;
  static constexpr int DO_B = DG_B + 288;
  static constexpr int END = DO_B + 22;                        // total size in dwords
};

// Second copy of the matrices for the int8 MFMA form of the gain network (rn_kernels.hip, RN_GRU_MFMA == 2):
// the int8 values as they are, [ceil(K/16)][rows][16] -- lane == row reads one 16-byte vector per 16 MACs, half
// the bytes of the f16 pack through the CU's vector L1, which is what bounds that stage.  Appended to the same
// buffer; offsets in 16-byte units from the pack base.
constexpr int rn_k16(int k) { return (k + 15) / 16; }
struct RnPack8 {
  static constexpr int ID_W = (RnPack::END + 3) / 4;             // K=42  rows=24
  static constexpr int VG_W = ID_W + rn_k16(42) * 24;            // K=24  rows=72
  static constexpr int VG_R = VG_W + rn_k16(24) * 72;            // K=24  rows=72
  static constexpr int NG_W = VG_R + rn_k16(24) * 72;            // K=90  rows=144
  static constexpr int NG_R = NG_W + rn_k16(90) * 144;           // K=48  rows=144
  static constexpr int DG_W = NG_R + rn_k16(48) * 144;           // K=114 rows=288
  static constexpr int DG_R = DG_W + rn_k16(114) * 288;          // K=96  rows=288
  static constexpr int DO_W = DG_R + rn_k16(96) * 288;           // K=96  rows=22
  static constexpr int MAT_END = DO_W + rn_k16(96) * 22;         // 16-byte units
  static constexpr int END = MAT_END * 4;                        // total size of the buffer in dwords
};

// Kernel arguments of one enqueue (chunk of T frames for all B streams).
struct RnArgs {
  // audio
  const float* in;      // caller layout
  float* out;           // caller layout
  long stride_t, stride_b;  // element strides of (frame, stream) in `in`/`out`
  // int16 sample transport (crispy_rn_process_s16*): `in` / `out` point at int16_t samples (same ELEMENT strides).
  // in_s16: the high-pass reads (float)s -- what the reference's i16 capture path feeds process_frame: s / 32768 (audio.rs:814)
  // x 32768 (audio.rs:264), both exact.  out_s16: the frame kernel stores trunc(clamp(y / 32768, -1, 1) x 32767) -- the
  // adapter's / 32768 + clamp (audio.rs:270-273) and the WAV writer's quantisation (recording.rs:109-110).
  int in_s16, out_s16;
  float* vad;           // [T][B] or null
  float* taps;          // [T][B][72] or null
  float* dbg;           // [B][RN_DBG_FLOATS] or null (last frame of the call)
  int T, B;
  // workspace: high-passed signal, per stream contiguous: [B][xhp_stride], first RN_HIST = history
  float* xhp;
  long xhp_stride;
  // persistent per-stream state (HBM)
  float* hp_mem;        // [B][2]
  float* synth;         // [B][480] overlap-add tails, read at the start of a launch and written at its end
  float* ceps;          // [B][8*22]
  float* lastg;         // [B][22]
  float* rnn;           // [B][168]
  float* last_gain;     // [B]
  int* last_period;     // [B]
  int* memid;           // [B]
  // constants
  const RnTables* tab;
  const uint32_t* wpack;
};

hipError_t rn_launch_highpass(const RnArgs& a, hipStream_t s, bool deep = false);   // deep: 32 samples requested ahead (few streams)
hipError_t rn_launch_frames(const RnArgs& a, hipStream_t s, int waves_per_stream = 1);   // the frame: analysis + gain network + synthesis
hipError_t rn_launch_roll_history(const RnArgs& a, hipStream_t s);
hipError_t rn_launch_tansig(const RnTables* tab, const float* x, float* y, long n, int sigmoid, hipStream_t s);

}  // namespace crispy
