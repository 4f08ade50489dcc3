// crispy_api.cpp -- extern "C" boundary of libcrispy_hip.so (see include/crispy_hip.h).
//
// Host side of the batched RNNoise path: owns the per-stream state tensors in HBM, the device
// tables, the repacked weights and the workspace; enqueues high-pass -> frame -> history-roll
// kernels per chunk of frames.  No CPU compute path exists here: without a gfx950 device every
// constructor fails.
#include "../../include/crispy_hip.h"
#include "api_util.h"
#include "rn_common.h"

#include <cmath>
#include <cstdarg>
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <atomic>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

using namespace crispy;

namespace {

constexpr int kChunkFrames = 250;   // workspace bound: frames of high-passed signal kept per call segment
// Pipeline grain: a frame-kernel launch covers one sub-chunk and is gated on that sub-chunk's high-pass, which runs
// ahead on the helper stream.  Measured per 100-frame step with the final kernels (4096 streams): 8 -> 8.2 ms,
// 10 -> 8.0, 12 -> 7.68, 16 -> 7.75, 20 -> 7.85, 30 -> 8.03; a longer ramp (3, 8, 20, 48) with 48-frame sub-chunks
// shortens the frame-kernel sum by 1.5 % (five launch tails instead of ten) but waits 0.8 ms for the high-pass.
#ifndef RN_SUB_FRAMES
#define RN_SUB_FRAMES 12
#endif
constexpr int kSubFrames = RN_SUB_FRAMES;
// Up to this many streams a handle runs the stage-pipelined frame kernel (three waves per stream, 26.7 KB of LDS and 128
// VGPRs: five such workgroups per CU = 1280 streams on 256 CUs); above it one wave per stream.  Measured
// (tools/rn_small_batch.py, 300 frames per call): frame kernels alone 20.6 against 34.7 us per frame at 1024 streams,
// 17.4 against 31.6 at 256; whole calls 9.4 against 11.0 ms and 7.9 against 10.0 ms -- the sequential high-pass chain of a
// stream (~26 us per frame: ten f64-path instructions per sample on one lane) is what bounds a call from there on.
constexpr int kStagedMaxStreams = 1280;

// Frames of sub-chunk `index` of a segment with `remaining` frames left.  The high-pass recurrence of the first
// sub-chunk cannot overlap anything (the previous call still reads xhp), so the segment starts with a short
// sub-chunk and ramps up: the high-pass runs ~2.9x faster per frame than the frame kernel, so every next sub-chunk
// (<= 2.7x the previous one) is filtered while the previous one is being processed.  3, 8, 12, 12, ... (rounds 1 - 2)
// Round 3: beside the frame kernel the high-pass manages ~38 us per frame against the frame kernel's ~57, not the 2.9x it
// has alone, so a sub-chunk may be at most ~1.5x its predecessor: behind 3, 8 the frame kernels of sub-chunks 1 and 2
// waited 107 + 55 us for their input (tools/step_timeline.py).  CRISPY_RN_RAMP="3,4,6,9" overrides the ramp 3, 4, 5, 7, 10 (developer
// knob; entries above the sub-chunk size are clipped).
inline const std::vector<int>& ramp_frames() {
  static const std::vector<int> ramp = [] {
    std::vector<int> r;
    if (const char* env = dev_env("CRISPY_RN_RAMP")) {
      for (const char* p = env; *p;) {
        char* end = nullptr;
        const long v = std::strtol(p, &end, 10);
        if (end == p) break;
        if (v > 0) r.push_back(v > kSubFrames ? kSubFrames : (int)v);
        p = *end ? end + 1 : end;
      }
    }
    if (r.empty()) r = {3, 4, 5, 7, 10};
    return r;
  }();
  return ramp;
}
inline int sub_frames(int index, int remaining) {
  const std::vector<int>& ramp = ramp_frames();
  const int want = index < (int)ramp.size() ? ramp[index] : kSubFrames;
  return remaining < want ? remaining : want;
}
inline int count_subs(int T) {
  int n = 0;
  for (int done = 0; done < T; ++n) done += sub_frames(n, T - done);
  return n;
}

void build_tables(RnTables* t) {
  const double pi = 3.14159265358979323846;
  for (int i = 0; i < RN_FRAME; ++i) {
    const double s = std::sin(.5 * pi * (i + .5) / RN_FRAME);
    t->half_window[i] = (float)std::sin(.5 * pi * s * s);
  }
  for (int i = 0; i < RN_NB; ++i)
    for (int j = 0; j < RN_NB; ++j) {
      double v = std::cos((i + .5) * j * pi / RN_NB);
      if (j == 0) v *= std::sqrt(.5);
      t->dct[i * RN_NB + j] = (float)v;
    }
  for (int i = 0; i < 208; ++i)
    t->tansig[i] = i <= 200 ? (float)(std::floor(std::tanh(0.04 * i) * 1e6 + 0.5) / 1e6) : 1.f;
  for (int k = 0; k < RN_WINDOW; ++k) {
    t->w960[k].x = (float)std::cos(-2.0 * pi * k / RN_WINDOW);
    t->w960[k].y = (float)std::sin(-2.0 * pi * k / RN_WINDOW);
  }
  static const int eband[RN_NB] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 10, 12, 14, 16, 20, 24, 28, 34, 40, 48, 60, 78, 100};
  for (int i = 0; i < 24; ++i) t->eband[i] = i < RN_NB ? eband[i] : 100;
  for (int i = 0; i < RN_NB - 1; ++i) {
    const int bs = (eband[i + 1] - eband[i]) * 4;
    for (int j = 0; j < bs; ++j) {
      t->bin_band[eband[i] * 4 + j] = i;
      t->bin_frac[eband[i] * 4 + j] = (float)j / (float)bs;   // same f32 division as the reference
    }
  }
  // band_piece (rn_common.h): lane 0 stays idle = the "zero" source of the two half-bands that do not exist (rising
  // half of band 0, falling half of band 21)
  {
    int piece[64] = {0}, head[2][RN_NB];
    for (int b = 0; b < RN_NB; ++b) head[0][b] = head[1][b] = 0;
    int lane = 1;
    for (int side = 0; side < 2; ++side)          // 0: rising halves (part_hi), 1: falling halves (part_lo)
      for (int b = 0; b < RN_NB; ++b) {
        const int c0 = side == 0 ? (b > 0 ? eband[b - 1] : 0) : eband[b];
        const int c1 = side == 0 ? eband[b] : (b < RN_NB - 1 ? eband[b + 1] : eband[b]);
        const int w = (side == 0 && b == 0) ? 0 : c1 - c0;
        if (w <= 0) continue;
        const int np = (w + 5) / 6;                                  // <= 4 pieces (w <= 22)
        if ((lane & 15) + np > 16) lane = (lane + 15) & ~15;         // a half's pieces stay inside one DPP row
        if (np > 4 || lane + np > 64) std::abort();                  // (cannot happen with the Opus band table)
        head[side][b] = lane;
        for (int p = 0, c = c0; p < np; ++p) {
          const int n = (w - (c - c0) + (np - p) - 1) / (np - p);    // balanced: 22 -> 6, 6, 5, 5
          piece[lane + p] = c | (n << 7) | (side << 10) | ((p + 1 < np ? 1 : 0) << 11) | ((p == 0 && np > 2 ? 1 : 0) << 12);
          c += n;
        }
        lane += np;
      }
    for (int l = 0; l < 64; ++l) {
      const int b = l < RN_NB ? l : 0;
      t->band_piece[l] = piece[l] | (head[0][b] << 13) | (head[1][b] << 19);
    }
  }
}

// [K][rows] int8 (row stride `stride`) -> f16 [ceil(K/8)][rows][8]; dst in 16-byte units (4 dwords)
void pack_matrix(uint32_t* dst, const int8_t* src, int K, int rows, int stride) {
  const int k8n = (K + 7) / 8;
  uint16_t* h = reinterpret_cast<uint16_t*>(dst);
  for (int k8 = 0; k8 < k8n; ++k8)
    for (int r = 0; r < rows; ++r)
      for (int q = 0; q < 8; ++q) {
        const int k = 8 * k8 + q;
        const _Float16 v = (_Float16)(float)(k < K ? src[(size_t)k * stride + r] : 0);
        uint16_t bits;
        std::memcpy(&bits, &v, 2);
        h[((size_t)k8 * rows + r) * 8 + q] = bits;
      }
}

// [K][rows] int8 -> int8 [ceil(K/16)][rows][16] (zero padded in K); dst in 16-byte units
void pack_matrix8(uint32_t* dst, const int8_t* src, int K, int rows, int stride) {
  const int k16n = (K + 15) / 16;
  int8_t* b = reinterpret_cast<int8_t*>(dst);
  for (int k16 = 0; k16 < k16n; ++k16)
    for (int r = 0; r < rows; ++r)
      for (int q = 0; q < 16; ++q) {
        const int k = 16 * k16 + q;
        b[((size_t)k16 * rows + r) * 16 + q] = k < K ? src[(size_t)k * stride + r] : 0;
      }
}

void pack_bias(uint32_t* dst, const int8_t* src, int n) {
  for (int i = 0; i < n; ++i) {
    const float f = (float)src[i];
    std::memcpy(&dst[i], &f, 4);
  }
}

void pack_weights(std::vector<uint32_t>& out, const int8_t* w) {
  out.assign(RnPack8::END, 0);
  uint32_t* p = out.data();
  pack_matrix(p + 4 * RnPack::ID_W, w + RnBlob::ID_W, 42, 24, 24);
  pack_matrix(p + 4 * RnPack::VG_W, w + RnBlob::VG_W, 24, 72, 72);
  pack_matrix(p + 4 * RnPack::VG_R, w + RnBlob::VG_R, 24, 72, 72);
  pack_matrix(p + 4 * RnPack::VO_W, w + RnBlob::VO_W, 24, 1, 1);
  pack_matrix(p + 4 * RnPack::NG_W, w + RnBlob::NG_W, 90, 144, 144);
  pack_matrix(p + 4 * RnPack::NG_R, w + RnBlob::NG_R, 48, 144, 144);
  pack_matrix(p + 4 * RnPack::DG_W, w + RnBlob::DG_W, 114, 288, 288);
  pack_matrix(p + 4 * RnPack::DG_R, w + RnBlob::DG_R, 96, 288, 288);
  pack_matrix(p + 4 * RnPack::DO_W, w + RnBlob::DO_W, 96, 22, 22);
  pack_bias(p + RnPack::ID_B, w + RnBlob::ID_B, 24);
  pack_bias(p + RnPack::VG_B, w + RnBlob::VG_B, 72);
  pack_bias(p + RnPack::VO_B, w + RnBlob::VO_B, 1);
  pack_bias(p + RnPack::NG_B, w + RnBlob::NG_B, 144);
  pack_bias(p + RnPack::DG_B, w + RnBlob::DG_B, 288);
  pack_bias(p + RnPack::DO_B, w + RnBlob::DO_B, 22);
  pack_matrix8(p + 4 * RnPack8::ID_W, w + RnBlob::ID_W, 42, 24, 24);
  pack_matrix8(p + 4 * RnPack8::VG_W, w + RnBlob::VG_W, 24, 72, 72);
  pack_matrix8(p + 4 * RnPack8::VG_R, w + RnBlob::VG_R, 24, 72, 72);
  pack_matrix8(p + 4 * RnPack8::NG_W, w + RnBlob::NG_W, 90, 144, 144);
  pack_matrix8(p + 4 * RnPack8::NG_R, w + RnBlob::NG_R, 48, 144, 144);
  pack_matrix8(p + 4 * RnPack8::DG_W, w + RnBlob::DG_W, 114, 288, 288);
  pack_matrix8(p + 4 * RnPack8::DG_R, w + RnBlob::DG_R, 96, 288, 288);
  pack_matrix8(p + 4 * RnPack8::DO_W, w + RnBlob::DO_W, 96, 22, 22);
}


}  // namespace

struct crispy_rn {
  int device = 0;
  int B = 0;
  hipStream_t stream = nullptr;
  hipStream_t hp_stream = nullptr;   // helper stream: the latency-bound high-pass runs beside the frame kernel
  hipEvent_t ev_begin = nullptr;
  std::vector<hipEvent_t> ev_hp;     // one per sub-chunk: high-pass done
  // constants
  RnTables* d_tab = nullptr;
  uint32_t* d_wpack = nullptr;
  // state
  float* d_hp_mem = nullptr;
  float* d_synth = nullptr;       // overlap-add tails [B][480]
  float* d_ceps = nullptr;
  float* d_lastg = nullptr;
  float* d_rnn = nullptr;
  float* d_last_gain = nullptr;
  int* d_last_period = nullptr;
  int* d_memid = nullptr;
  // workspace
  float* d_xhp = nullptr;
  long xhp_stride = 0;
  // Frames per high-pass launch.  A high-pass wave keeps the VALU of its SIMD ~35 % busy (nine dependent f64
  // operations per sample) and a frame-kernel launch lasts as long as its slowest wave, so a sub-chunk's high-pass as
  // one 0.3 ms kernel delays the four frame waves that share its SIMD by ~0.08 ms per launch (0.8 ms per 100-frame
  // step, measured with CRISPY_RN_HP=upfront).  As kernels of two frames the 64 waves land on other SIMDs every
  // ~50 us and the delay spreads: 8.17 -> 7.68 ms per step (1 frame: 7.83, 3: 8.07, 4: 8.15, 6: 8.0, whole: 8.17).
  int hp_split = 2;
  // Waves per stream of the frame kernel: 1 = one wave runs the whole frame (every pipe of the chip is busy from ~4 waves per
  // SIMD = 4096 streams up); 3 = the frame's three stages on three waves, a frame apart (rn_frame3_kernel: a stream
  // advances a frame per ~10 k quad-cycles instead of ~26 k -- what counts while there are fewer waves than SIMD slots).
  // Chosen at create time from the stream count; CRISPY_RN_WAVES=1|3 overrides (tests run both forms).
  int waves = 1;
  int hp_ahead = 0;          // > 0: the high-pass runs at most this many sub-chunks in front of the frame kernels
  std::vector<hipEvent_t> ev_fr;   // one per sub-chunk: frame kernel done (only used with hp_ahead)
  bool hp_deep = false;      // the high-pass requests 32 samples ahead (80 registers): set where a wave of it fits beside the frame waves
  bool hp_upfront = false;   // diagnostic (CRISPY_RN_HP=upfront): every high-pass of a call segment first, on the main stream
  float* d_dbg = nullptr;
  // host-pointer staging
  float* d_stage_in = nullptr;
  float* d_stage_out = nullptr;
  float* d_stage_vad = nullptr;
  size_t stage_frames = 0;
  // pipelined host path: copy-in / compute / copy-out streams and per-piece events
  hipStream_t h2d_stream = nullptr;
  hipStream_t d2h_stream = nullptr;
  std::vector<hipEvent_t> ev_in, ev_done;
  // timing
  bool timing = false;
  std::vector<hipEvent_t> ev;  // per segment: begin, (frame_begin, frame_end) x sub-chunks, end
  size_t ev_used = 0;
  std::vector<int> seg_subs;   // sub-chunks of every timed segment
};

namespace {

void free_all(crispy_rn* h) {
  if (!h) return;
  (void)hipSetDevice(h->device);
  if (h->stream) (void)hipStreamSynchronize(h->stream);
  void* ptrs[] = {h->d_tab, h->d_wpack, h->d_hp_mem, h->d_synth, h->d_ceps, h->d_lastg, h->d_rnn,
                  h->d_last_gain, h->d_last_period, h->d_memid, h->d_xhp, h->d_dbg, h->d_stage_in,
                  h->d_stage_out, h->d_stage_vad};
  for (void* p : ptrs)
    if (p) (void)hipFree(p);
  for (hipEvent_t e : h->ev) (void)hipEventDestroy(e);
  for (hipEvent_t e : h->ev_hp) (void)hipEventDestroy(e);
  for (hipEvent_t e : h->ev_fr) (void)hipEventDestroy(e);
  if (h->ev_begin) (void)hipEventDestroy(h->ev_begin);
  for (hipEvent_t e : h->ev_in) (void)hipEventDestroy(e);
  for (hipEvent_t e : h->ev_done) (void)hipEventDestroy(e);
  if (h->h2d_stream) { (void)hipStreamSynchronize(h->h2d_stream); (void)hipStreamDestroy(h->h2d_stream); }
  if (h->d2h_stream) { (void)hipStreamSynchronize(h->d2h_stream); (void)hipStreamDestroy(h->d2h_stream); }
  if (h->hp_stream) { (void)hipStreamSynchronize(h->hp_stream); (void)hipStreamDestroy(h->hp_stream); }
  if (h->stream) (void)hipStreamDestroy(h->stream);
  delete h;
}

int zero_state(crispy_rn* h, int stream) {
  const int b0 = stream < 0 ? 0 : stream;
  const size_t n = stream < 0 ? (size_t)h->B : 1;
  hipStream_t s = h->stream;
  HIP_TRY(hipMemsetAsync(h->d_hp_mem + (size_t)b0 * 2, 0, n * 2 * sizeof(float), s));
  HIP_TRY(hipMemsetAsync(h->d_synth + (size_t)b0 * 480, 0, n * 480 * sizeof(float), s));
  HIP_TRY(hipMemsetAsync(h->d_ceps + (size_t)b0 * 176, 0, n * 176 * sizeof(float), s));
  HIP_TRY(hipMemsetAsync(h->d_lastg + (size_t)b0 * RN_NB, 0, n * RN_NB * sizeof(float), s));
  HIP_TRY(hipMemsetAsync(h->d_rnn + (size_t)b0 * 168, 0, n * 168 * sizeof(float), s));
  HIP_TRY(hipMemsetAsync(h->d_last_gain + b0, 0, n * sizeof(float), s));
  HIP_TRY(hipMemsetAsync(h->d_last_period + b0, 0, n * sizeof(int), s));
  HIP_TRY(hipMemsetAsync(h->d_memid + b0, 0, n * sizeof(int), s));
  if (stream < 0) {
    HIP_TRY(hipMemsetAsync(h->d_xhp, 0, (size_t)h->B * h->xhp_stride * sizeof(float), s));
  } else {
    HIP_TRY(hipMemsetAsync(h->d_xhp + (size_t)b0 * h->xhp_stride, 0, RN_HIST * sizeof(float), s));
  }
  return CRISPY_OK;
}

int process_device_impl(crispy_rn* h, const void* d_in, void* d_out, float* d_vad, float* d_taps, int n_frames,
                        long stride_t, long stride_b, hipStream_t s, bool s16 = false);
int process_host_impl(crispy_rn* h, const void* in, void* out, float* vad, int n_frames, crispy_rn_layout layout, bool s16);

}  // namespace

extern "C" {

const char* crispy_last_error(void) { return last_error_cstr(); }

const char* crispy_version(void) { return "crispy_hip 0.3.0 gfx950"; }
int crispy_abi_version(void) { return CRISPY_ABI_VERSION; }

int crispy_device_count(void) try {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  int ok = 0;
  for (int d = 0; d < n; ++d)
    if (device_is_gfx950(d)) ++ok;
  return ok;
} CRISPY_CATCH_RET("crispy_device_count")

// ---- rnnoise-nu text model files (RnnModel::from_read upstream [UPSTREAM-RECALL, SURVEY Appendix A.7]) ----
// "rnnoise-nu model file version 1\n", then whitespace-separated decimal integers: per layer `n_in n_out activation`
// followed by its arrays; file order input_dense, vad_gru, noise_gru, denoise_gru, denoise_output, vad_output;
// activation ids 0 tanh, 1 sigmoid, 2 ReLU.  The topology is fixed, so sizes and activations are checked,
// not obeyed.  Output: the flat blob crispy_rn_create takes (layer order input_dense, vad_gru, vad_output,
// noise_gru, denoise_gru, denoise_output).
int crispy_rn_weights_from_file(const char* path, int8_t* blob, size_t blob_bytes) try {
  if (!path || !blob) return fail(CRISPY_ERR_INVALID_ARG, "crispy_rn_weights_from_file: NULL argument");
  if (blob_bytes != (size_t)RN_WEIGHT_BYTES)
    return fail(CRISPY_ERR_INVALID_ARG, "crispy_rn_weights_from_file: blob must hold %d bytes (got %zu)", RN_WEIGHT_BYTES,
                blob_bytes);
  FILE* f = std::fopen(path, "rb");
  if (!f) return fail(CRISPY_ERR_BAD_MODEL, "crispy_rn_weights_from_file: cannot open '%s'", path);
  struct Closer { FILE* f; ~Closer() { std::fclose(f); } } closer{f};
  char header[64] = {0};
  if (!std::fgets(header, sizeof(header), f)) return fail(CRISPY_ERR_BAD_MODEL, "crispy_rn_weights_from_file: empty file");
  {
    size_t n = std::strlen(header);
    while (n && (header[n - 1] == '\n' || header[n - 1] == '\r' || header[n - 1] == ' ')) header[--n] = 0;
  }
  if (std::strcmp(header, "rnnoise-nu model file version 1") != 0)
    return fail(CRISPY_ERR_BAD_MODEL, "crispy_rn_weights_from_file: not an rnnoise-nu model file (header '%.40s')", header);
  struct Layer { const char* name; bool gru; int n_in, n_out, act; size_t off; };
  // blob offsets follow the blob's layer order, the table follows the FILE's layer order
  size_t off[6];
  {
    const int kind[6] = {0, 1, 0, 1, 1, 0}, nin[6] = {42, 24, 24, 90, 114, 96}, nout[6] = {24, 24, 1, 48, 96, 22};
    size_t o = 0;
    for (int i = 0; i < 6; ++i) {
      off[i] = o;
      o += kind[i] ? (size_t)nin[i] * 3 * nout[i] + (size_t)nout[i] * 3 * nout[i] + 3 * nout[i]
                   : (size_t)nin[i] * nout[i] + nout[i];
    }
    if (o != (size_t)RN_WEIGHT_BYTES) return fail(CRISPY_ERR_HIP, "crispy_rn_weights_from_file: layout table is wrong");
  }
  const Layer layers[6] = {
      {"input_dense", false, 42, 24, 0, off[0]},   {"vad_gru", true, 24, 24, 2, off[1]},
      {"noise_gru", true, 90, 48, 2, off[3]},      {"denoise_gru", true, 114, 96, 2, off[4]},
      {"denoise_output", false, 96, 22, 1, off[5]}, {"vad_output", false, 24, 1, 1, off[2]},
  };
  auto next_int = [&](long* v) -> bool {
    int c;
    do { c = std::fgetc(f); } while (c == ' ' || c == '\n' || c == '\r' || c == '\t');
    if (c == EOF) return false;
    bool neg = false;
    if (c == '-' || c == '+') { neg = c == '-'; c = std::fgetc(f); }
    if (c < '0' || c > '9') return false;
    long x = 0;
    int digits = 0;
    while (c >= '0' && c <= '9') {
      if (++digits > 9) return false;
      x = x * 10 + (c - '0');
      c = std::fgetc(f);
    }
    if (c != EOF && c != ' ' && c != '\n' && c != '\r' && c != '\t') return false;   // "12x", "1.5": not an integer
    *v = neg ? -x : x;
    return true;
  };
  for (const Layer& L : layers) {
    long hdr[3];
    for (long& v : hdr)
      if (!next_int(&v)) return fail(CRISPY_ERR_BAD_MODEL, "crispy_rn_weights_from_file: truncated model file (%s header)", L.name);
    if (hdr[0] != L.n_in || hdr[1] != L.n_out)
      return fail(CRISPY_ERR_BAD_MODEL, "crispy_rn_weights_from_file: %s: expected %dx%d, file has %ldx%ld", L.name, L.n_in,
                  L.n_out, hdr[0], hdr[1]);
    if (hdr[2] != L.act)
      return fail(CRISPY_ERR_BAD_MODEL, "crispy_rn_weights_from_file: %s: unsupported activation id %ld", L.name, hdr[2]);
    const size_t count = L.gru ? (size_t)L.n_in * 3 * L.n_out + (size_t)L.n_out * 3 * L.n_out + 3 * L.n_out
                               : (size_t)L.n_in * L.n_out + L.n_out;
    for (size_t i = 0; i < count; ++i) {
      long v;
      if (!next_int(&v)) return fail(CRISPY_ERR_BAD_MODEL, "crispy_rn_weights_from_file: truncated model file (%s)", L.name);
      if (v < -128 || v > 127)
        return fail(CRISPY_ERR_BAD_MODEL, "crispy_rn_weights_from_file: %s: weight %ld out of int8 range", L.name, v);
      blob[L.off + i] = (int8_t)v;
    }
  }
  return CRISPY_OK;
} CRISPY_CATCH_RET("crispy_rn_weights_from_file")

int crispy_rn_create_from_file(const char* path, int n_streams, int device, crispy_rn** out) try {
  if (!out) return fail(CRISPY_ERR_INVALID_ARG, "crispy_rn_create_from_file: out is NULL");
  *out = nullptr;
  std::vector<int8_t> blob(RN_WEIGHT_BYTES);
  const int rc = crispy_rn_weights_from_file(path, blob.data(), blob.size());
  if (rc != CRISPY_OK) return rc;
  return crispy_rn_create(blob.data(), blob.size(), n_streams, device, out);
} CRISPY_CATCH_RET("crispy_rn_create_from_file")

// Self-test of the exception guard every entry point is wrapped in (tests/test_abi_and_host.py): throws the
// requested kind inside a guarded body and reports what the guard turned it into.
int crispy_selftest_exception_guard(int kind) try {
  switch (kind) {
    case 1: throw std::bad_alloc();
    case 2: throw std::length_error("vector::_M_default_append");
    case 3: throw std::runtime_error("selftest");
    case 4: throw 42;
    case 5: {   // a real allocation failure, not a simulated one: more elements than any allocator can give
      std::vector<double> v;
      v.resize(v.max_size());
      return (int)v.size();
    }
    default: return CRISPY_OK;
  }
} CRISPY_CATCH_RET("crispy_selftest_exception_guard")

int crispy_rn_create(const int8_t* weights, size_t nbytes, int n_streams, int device, crispy_rn** out) try {
  if (!out) return fail(CRISPY_ERR_INVALID_ARG, "crispy_rn_create: out is NULL");
  *out = nullptr;
  if (!weights || nbytes != (size_t)RN_WEIGHT_BYTES)
    return fail(CRISPY_ERR_BAD_MODEL, "crispy_rn_create: weights must be a %d-byte blob (got %zu)",
                RN_WEIGHT_BYTES, weights ? nbytes : (size_t)0);
  if (n_streams <= 0) return fail(CRISPY_ERR_INVALID_ARG, "crispy_rn_create: n_streams must be > 0");
  {
    const int rc0 = check_device(device, "crispy_rn_create");
    if (rc0 != CRISPY_OK) return rc0;
  }

  crispy_rn* h = new (std::nothrow) crispy_rn();
  if (!h) return fail(CRISPY_ERR_OOM, "crispy_rn_create: host allocation failed");
  h->device = device;
  h->B = n_streams;
  h->xhp_stride = (long)(kChunkFrames + RN_HIST_FRAMES) * RN_FRAME;
  int rc = CRISPY_OK;
  auto body = [&]() -> int {
    HIP_TRY(hipSetDevice(device));
    HIP_TRY(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
    {
      // The helper stream must not share a hardware queue with the main stream (the high-pass of the next
      // sub-chunks has to run *beside* the frame kernel): HIP multiplexes streams of one priority over a small pool
      // of hardware queues in creation order, and with other libraries' streams in the process (RCCL: measured 9.2 ms
      // per step instead of 7.7) the two can land on the same one.  Queues are pooled per priority, so the helper
      // stream gets the highest one -- which also suits it: its kernels are short and gate the frame kernels.
      int least = 0, greatest = 0;
      HIP_TRY(hipDeviceGetStreamPriorityRange(&least, &greatest));
      HIP_TRY(hipStreamCreateWithPriority(&h->hp_stream, hipStreamNonBlocking, greatest));
    }
    HIP_TRY(hipEventCreateWithFlags(&h->ev_begin, hipEventDisableTiming));
    const size_t B = (size_t)n_streams;
    HIP_TRY(hipMalloc(&h->d_tab, sizeof(RnTables)));
    HIP_TRY(hipMalloc(&h->d_wpack, sizeof(uint32_t) * RnPack8::END));
    HIP_TRY(hipMalloc(&h->d_hp_mem, B * 2 * sizeof(float)));
    HIP_TRY(hipMalloc(&h->d_synth, B * 480 * sizeof(float)));
    HIP_TRY(hipMalloc(&h->d_ceps, B * 176 * sizeof(float)));
    HIP_TRY(hipMalloc(&h->d_lastg, B * RN_NB * sizeof(float)));
    HIP_TRY(hipMalloc(&h->d_rnn, B * 168 * sizeof(float)));
    HIP_TRY(hipMalloc(&h->d_last_gain, B * sizeof(float)));
    HIP_TRY(hipMalloc(&h->d_last_period, B * sizeof(int)));
    HIP_TRY(hipMalloc(&h->d_memid, B * sizeof(int)));
    HIP_TRY(hipMalloc(&h->d_xhp, B * h->xhp_stride * sizeof(float)));
    {
      const char* hp = dev_env("CRISPY_RN_HP");
      h->hp_upfront = hp && std::strcmp(hp, "upfront") == 0;
      const char* wv = test_env("CRISPY_RN_WAVES");
      h->waves = wv ? (std::atoi(wv) == 3 ? 3 : 1) : (n_streams <= kStagedMaxStreams ? 3 : 1);
      // with the stage-pipelined form, i.e. <= 1280 streams: from 1536 streams on a call is as long as its frame kernels
      // with either request depth (sweep in NOTEBOOK 8.6)
      const char* hd = dev_env("CRISPY_RN_HP_DEEP");
      h->hp_deep = hd ? std::atoi(hd) != 0 : h->waves == 3;
      const char* sp = dev_env("CRISPY_RN_HP_SPLIT");
      if (sp) h->hp_split = std::atoi(sp);   // 0: one kernel per sub-chunk
      const char* ah = dev_env("CRISPY_RN_HP_AHEAD");
      h->hp_ahead = ah ? std::atoi(ah) : 0;
    }
    RnTables* tab = new RnTables();
    build_tables(tab);
    hipError_t e = hipMemcpy(h->d_tab, tab, sizeof(RnTables), hipMemcpyHostToDevice);
    delete tab;
    HIP_TRY(e);
    std::vector<uint32_t> pack;
    pack_weights(pack, weights);
    HIP_TRY(hipMemcpy(h->d_wpack, pack.data(), sizeof(uint32_t) * pack.size(), hipMemcpyHostToDevice));
    int z = zero_state(h, -1);
    if (z != CRISPY_OK) return z;
    HIP_TRY(hipStreamSynchronize(h->stream));
    return CRISPY_OK;
  };
  rc = body();
  if (rc != CRISPY_OK) {
    const std::string keep = last_error_cstr();
    free_all(h);
    return fail(rc, "%s", keep.c_str());
  }
  *out = h;
  return CRISPY_OK;
} CRISPY_CATCH_RET("crispy_rn_create")

void crispy_rn_destroy(crispy_rn* h) { free_all(h); }

int crispy_rn_n_streams(const crispy_rn* h) { return h ? h->B : 0; }

int crispy_rn_frames_per_launch(void) { return kSubFrames; }
int crispy_rn_n_launches(int n_frames) try {
  int n = 0;
  for (int t0 = 0; t0 < n_frames; t0 += kChunkFrames) n += count_subs(n_frames - t0 < kChunkFrames ? n_frames - t0 : kChunkFrames);
  return n;
} CRISPY_CATCH_RET("crispy_rn_n_launches")

int crispy_rn_reset(crispy_rn* h, int stream) try {
  if (!h) return fail(CRISPY_ERR_INVALID_ARG, "crispy_rn_reset: NULL handle");
  if (stream < -1 || stream >= h->B)
    return fail(CRISPY_ERR_INVALID_ARG, "crispy_rn_reset: stream %d out of range", stream);
  HIP_TRY(hipSetDevice(h->device));
  int rc = zero_state(h, stream);
  if (rc != CRISPY_OK) return rc;
  HIP_TRY(hipStreamSynchronize(h->stream));
  return CRISPY_OK;
} CRISPY_CATCH_RET("crispy_rn_reset")

int crispy_rn_process_device(crispy_rn* h, const float* d_in, float* d_out, float* d_vad, float* d_taps,
                             int n_frames, crispy_rn_layout layout, void* hip_stream) try {
  if (!h) return fail(CRISPY_ERR_INVALID_ARG, "crispy_rn_process_device: NULL handle");
  if (n_frames < 0) return fail(CRISPY_ERR_INVALID_ARG, "crispy_rn_process_device: n_frames < 0");
  if (n_frames == 0) return CRISPY_OK;
  if (!d_in || !d_out) return fail(CRISPY_ERR_INVALID_ARG, "crispy_rn_process_device: NULL audio pointer");
  if (layout != CRISPY_RN_LAYOUT_TBF && layout != CRISPY_RN_LAYOUT_BTF)
    return fail(CRISPY_ERR_INVALID_ARG, "crispy_rn_process_device: unknown layout %d", (int)layout);
  if (((uintptr_t)d_in | (uintptr_t)d_out) & 15)
    return fail(CRISPY_ERR_INVALID_ARG, "crispy_rn_process_device: audio pointers must be 16-byte aligned");
  HIP_TRY(hipSetDevice(h->device));
  hipStream_t s = hip_stream ? (hipStream_t)hip_stream : h->stream;
  const long stride_t = layout == CRISPY_RN_LAYOUT_TBF ? (long)h->B * RN_FRAME : (long)RN_FRAME;
  const long stride_b = layout == CRISPY_RN_LAYOUT_TBF ? (long)RN_FRAME : (long)n_frames * RN_FRAME;
  return process_device_impl(h, d_in, d_out, d_vad, d_taps, n_frames, stride_t, stride_b, s);
} CRISPY_CATCH_RET("crispy_rn_process_device")

int crispy_rn_process_s16_device(crispy_rn* h, const int16_t* d_in, int16_t* d_out, float* d_vad, int n_frames,
                                 crispy_rn_layout layout, void* hip_stream) try {
  if (!h) return fail(CRISPY_ERR_INVALID_ARG, "crispy_rn_process_s16_device: NULL handle");
  if (n_frames < 0) return fail(CRISPY_ERR_INVALID_ARG, "crispy_rn_process_s16_device: n_frames < 0");
  if (n_frames == 0) return CRISPY_OK;
  if (!d_in || !d_out) return fail(CRISPY_ERR_INVALID_ARG, "crispy_rn_process_s16_device: NULL audio pointer");
  if (layout != CRISPY_RN_LAYOUT_TBF && layout != CRISPY_RN_LAYOUT_BTF)
    return fail(CRISPY_ERR_INVALID_ARG, "crispy_rn_process_s16_device: unknown layout %d", (int)layout);
  if (((uintptr_t)d_in | (uintptr_t)d_out) & 15)
    return fail(CRISPY_ERR_INVALID_ARG, "crispy_rn_process_s16_device: audio pointers must be 16-byte aligned");
  HIP_TRY(hipSetDevice(h->device));
  hipStream_t s = hip_stream ? (hipStream_t)hip_stream : h->stream;
  const long stride_t = layout == CRISPY_RN_LAYOUT_TBF ? (long)h->B * RN_FRAME : (long)RN_FRAME;
  const long stride_b = layout == CRISPY_RN_LAYOUT_TBF ? (long)RN_FRAME : (long)n_frames * RN_FRAME;
  return process_device_impl(h, d_in, d_out, d_vad, nullptr, n_frames, stride_t, stride_b, s, true);
} CRISPY_CATCH_RET("crispy_rn_process_s16_device")

}  // extern "C"

namespace {
// n_frames frames of every stream with explicit element strides of (frame, stream): what the public entry point
// derives from its layout argument, and what the pipelined host path calls per piece of a larger BTF tensor
int process_device_impl(crispy_rn* h, const void* d_in_v, void* d_out_v, float* d_vad, float* d_taps, int n_frames,
                        long stride_t, long stride_b, hipStream_t s, bool s16) {
  // int16 transport (crispy_rn_process_s16*): the same element strides over 2-byte samples; the kernels cast back
  const long esz = s16 ? 2 : 4;
  auto in_at = [&](long elems) { return reinterpret_cast<const float*>(static_cast<const char*>(d_in_v) + elems * esz); };
  auto out_at = [&](long elems) { return reinterpret_cast<float*>(static_cast<char*>(d_out_v) + elems * esz); };
  RnArgs a{};
  a.in_s16 = a.out_s16 = s16 ? 1 : 0;
  a.B = h->B;
  a.stride_t = stride_t;
  a.stride_b = stride_b;
  a.xhp = h->d_xhp;
  a.xhp_stride = h->xhp_stride;
  a.hp_mem = h->d_hp_mem;
  a.synth = h->d_synth;
  a.ceps = h->d_ceps;
  a.lastg = h->d_lastg;
  a.rnn = h->d_rnn;
  a.last_gain = h->d_last_gain;
  a.last_period = h->d_last_period;
  a.memid = h->d_memid;
  a.tab = h->d_tab;
  a.wpack = h->d_wpack;

  if (h->timing) { h->ev_used = 0; h->seg_subs.clear(); }
  for (int t0 = 0; t0 < n_frames; t0 += kChunkFrames) {
    const int T = (n_frames - t0) < kChunkFrames ? (n_frames - t0) : kChunkFrames;
    // The high-pass of this segment may start once everything already enqueued on `s` is done
    // (producer of d_in, previous segment's frame kernels and history roll).
    HIP_TRY(hipEventRecord(h->ev_begin, s));
    HIP_TRY(hipStreamWaitEvent(h->hp_stream, h->ev_begin, 0));
    const int n_sub = count_subs(T);
    while ((int)h->ev_hp.size() < n_sub) {
      hipEvent_t ne;
      HIP_TRY(hipEventCreateWithFlags(&ne, hipEventDisableTiming));
      h->ev_hp.push_back(ne);
    }
    hipEvent_t* e = nullptr;
    if (h->timing) {
      while (h->ev.size() < h->ev_used + 2 + 2 * (size_t)n_sub) {
        hipEvent_t ne;
        HIP_TRY(hipEventCreate(&ne));
        h->ev.push_back(ne);
      }
      e = &h->ev[h->ev_used];
      h->ev_used += 2 + 2 * (size_t)n_sub;
      HIP_TRY(hipEventRecord(e[0], s));
    }
    // high-pass sub-chunks on the helper stream, enqueued up to `hp_ahead` sub-chunks in front of the frame kernels
    // (0: all of them back to back at once)
    int hp_next = 0, hp_ts = 0;
    while ((int)h->ev_fr.size() < n_sub) {
      hipEvent_t ne;
      HIP_TRY(hipEventCreateWithFlags(&ne, hipEventDisableTiming));
      h->ev_fr.push_back(ne);
    }
    auto enqueue_hp_until = [&](int last) -> int {
      for (; hp_next <= last && hp_next < n_sub; ++hp_next) {
        const int i = hp_next, ts = hp_ts;
        RnArgs sa = a;
        sa.T = sub_frames(i, T - ts);
        sa.in = in_at((long)(t0 + ts) * a.stride_t);
        sa.xhp = h->d_xhp + (long)ts * RN_FRAME;   // row pointer shifted by the frames already filtered
        hipStream_t hs = h->hp_upfront ? s : h->hp_stream;
        if (h->hp_ahead > 0 && i >= h->hp_ahead) HIP_TRY(hipStreamWaitEvent(hs, h->ev_fr[i - h->hp_ahead], 0));
        const int per = h->hp_split > 0 ? h->hp_split : sa.T;
        for (int f0 = 0; f0 < sa.T; f0 += per) {
          RnArgs pa = sa;
          pa.T = (sa.T - f0) < per ? (sa.T - f0) : per;
          pa.in = in_at((long)(t0 + ts + f0) * a.stride_t);
          pa.xhp = sa.xhp + (long)f0 * RN_FRAME;
          HIP_TRY(rn_launch_highpass(pa, hs, h->hp_deep));
        }
        HIP_TRY(hipEventRecord(h->ev_hp[i], hs));
        hp_ts += sa.T;
      }
      return CRISPY_OK;
    };
    if (h->hp_ahead <= 0) { const int rc_hp = enqueue_hp_until(n_sub - 1); if (rc_hp != CRISPY_OK) return rc_hp; }
    // frame kernels on the caller's stream, each gated on its own sub-chunk's high-pass
    for (int i = 0, ts = 0; i < n_sub; ++i) {
      RnArgs sa = a;
      sa.T = sub_frames(i, T - ts);
      sa.out = out_at((long)(t0 + ts) * a.stride_t);
      sa.vad = d_vad ? d_vad + (long)(t0 + ts) * h->B : nullptr;
      sa.taps = d_taps ? d_taps + (long)(t0 + ts) * h->B * RN_TAPS : nullptr;
      sa.dbg = (t0 + ts + sa.T == n_frames) ? h->d_dbg : nullptr;
      sa.xhp = h->d_xhp + (long)ts * RN_FRAME;
      if (h->hp_ahead > 0) { const int rc_hp = enqueue_hp_until(i + h->hp_ahead - 1); if (rc_hp != CRISPY_OK) return rc_hp; }
      HIP_TRY(hipStreamWaitEvent(s, h->ev_hp[i], 0));
      if (e) HIP_TRY(hipEventRecord(e[1 + 2 * i], s));
      HIP_TRY(rn_launch_frames(sa, s, h->waves));
      if (e) HIP_TRY(hipEventRecord(e[2 + 2 * i], s));
      if (h->hp_ahead > 0) HIP_TRY(hipEventRecord(h->ev_fr[i], s));
      ts += sa.T;
    }
    a.T = T;
    HIP_TRY(rn_launch_roll_history(a, s));
    if (e) HIP_TRY(hipEventRecord(e[1 + 2 * n_sub], s));
    if (h->timing) h->seg_subs.push_back(n_sub);
  }
  return CRISPY_OK;
}
}  // namespace

extern "C" {

int crispy_host_register(void* p, size_t bytes) try {
  if (!p || bytes == 0) return fail(CRISPY_ERR_INVALID_ARG, "crispy_host_register: NULL pointer or zero size");
  HIP_TRY(hipHostRegister(p, bytes, hipHostRegisterDefault));
  return CRISPY_OK;
} CRISPY_CATCH_RET("crispy_host_register")

int crispy_host_unregister(void* p) try {
  if (!p) return fail(CRISPY_ERR_INVALID_ARG, "crispy_host_unregister: NULL pointer");
  HIP_TRY(hipHostUnregister(p));
  return CRISPY_OK;
} CRISPY_CATCH_RET("crispy_host_unregister")

int crispy_rn_process(crispy_rn* h, const float* in, float* out, float* vad, int n_frames,
                      crispy_rn_layout layout) try {
  return process_host_impl(h, in, out, vad, n_frames, layout, false);
} CRISPY_CATCH_RET("crispy_rn_process")

int crispy_rn_process_s16(crispy_rn* h, const int16_t* in, int16_t* out, float* vad, int n_frames,
                          crispy_rn_layout layout) try {
  return process_host_impl(h, in, out, vad, n_frames, layout, true);
} CRISPY_CATCH_RET("crispy_rn_process_s16")

}  // extern "C"

namespace {
// crispy_rn_process / crispy_rn_process_s16: host tensors of f32 or int16 samples (esz bytes each) through the staging
// buffers -- sized for f32, an int16 call uses half of them
int process_host_impl(crispy_rn* h, const void* in_v, void* out_v, float* vad, int n_frames, crispy_rn_layout layout, bool s16) {
  if (!h) return fail(CRISPY_ERR_INVALID_ARG, "crispy_rn_process: NULL handle");
  if (n_frames < 0) return fail(CRISPY_ERR_INVALID_ARG, "crispy_rn_process: n_frames < 0");
  if (n_frames == 0) return CRISPY_OK;
  if (!in_v || !out_v) return fail(CRISPY_ERR_INVALID_ARG, "crispy_rn_process: NULL audio pointer");
  if (layout != CRISPY_RN_LAYOUT_TBF && layout != CRISPY_RN_LAYOUT_BTF)
    return fail(CRISPY_ERR_INVALID_ARG, "crispy_rn_process: unknown layout %d", (int)layout);
  const size_t esz = s16 ? sizeof(int16_t) : sizeof(float);
  const char* in = static_cast<const char*>(in_v);
  char* out = static_cast<char*>(out_v);
  HIP_TRY(hipSetDevice(h->device));
  const size_t n = (size_t)n_frames * h->B * RN_FRAME;
  if (h->stage_frames < (size_t)n_frames) {
    if (h->d_stage_in) (void)hipFree(h->d_stage_in);
    if (h->d_stage_out) (void)hipFree(h->d_stage_out);
    if (h->d_stage_vad) (void)hipFree(h->d_stage_vad);
    h->d_stage_in = h->d_stage_out = h->d_stage_vad = nullptr;
    h->stage_frames = 0;
    HIP_TRY(hipMalloc(&h->d_stage_in, n * sizeof(float)));
    HIP_TRY(hipMalloc(&h->d_stage_out, n * sizeof(float)));
    HIP_TRY(hipMalloc(&h->d_stage_vad, (size_t)n_frames * h->B * sizeof(float)));
    h->stage_frames = (size_t)n_frames;
  }
  const size_t B = (size_t)h->B;
  const size_t frame_bytes = B * RN_FRAME * esz;          // one frame of every stream
  if (n * esz < (size_t)(8u << 20)) {
    // small calls (the single-stream process_frame drop-in): one copy in, one call, one copy out
    HIP_TRY(hipMemcpyAsync(h->d_stage_in, in, n * esz, hipMemcpyHostToDevice, h->stream));
    int rc = process_device_impl(h, h->d_stage_in, h->d_stage_out, vad ? h->d_stage_vad : nullptr, nullptr, n_frames,
                                 layout == CRISPY_RN_LAYOUT_TBF ? (long)h->B * RN_FRAME : (long)RN_FRAME,
                                 layout == CRISPY_RN_LAYOUT_TBF ? (long)RN_FRAME : (long)n_frames * RN_FRAME, h->stream, s16);
    if (rc != CRISPY_OK) return rc;
    HIP_TRY(hipMemcpyAsync(out, h->d_stage_out, n * esz, hipMemcpyDeviceToHost, h->stream));
    if (vad)
      HIP_TRY(hipMemcpyAsync(vad, h->d_stage_vad, (size_t)n_frames * h->B * sizeof(float),
                             hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return CRISPY_OK;
  }
  // Large calls are cut into pieces of frames (~64 MB each, at least the 3 + 8 ramp of a call) that flow through
  // three streams: copy-in of piece i + 1, frame kernels of piece i and copy-out of piece i - 1 overlap, so a call
  // costs about one direction of PCIe traffic instead of in + compute + out.  From pageable memory a copy call blocks
  // its host thread while the runtime stages it, so the copy-out side runs on its own thread; from registered memory
  // (crispy_host_register) both directions are plain DMA.
  // (pieces of the same FRAME count for both sample widths: ~64 MB of f32, ~32 MB of int16 -- the pipeline's fill and drain
  // are one piece each, and an int16 call of twice the frames per piece measured 36 GB/s each way against 41.5)
  int P = (int)(((size_t)(64u << 20) * esz / sizeof(float)) / frame_bytes);
  if (P < 11) P = 11;
  if (P > n_frames) P = n_frames;
  const int n_pieces = (n_frames + P - 1) / P;
  if (!h->h2d_stream || !h->d2h_stream) {
    // copy streams in the lowest-priority queue pool: never on the hardware queue of the main or the helper stream
    int least = 0, greatest = 0;
    HIP_TRY(hipDeviceGetStreamPriorityRange(&least, &greatest));
    if (!h->h2d_stream) HIP_TRY(hipStreamCreateWithPriority(&h->h2d_stream, hipStreamNonBlocking, least));
    if (!h->d2h_stream) HIP_TRY(hipStreamCreateWithPriority(&h->d2h_stream, hipStreamNonBlocking, least));
  }
  while ((int)h->ev_in.size() < n_pieces) {
    hipEvent_t e0, e1;
    HIP_TRY(hipEventCreateWithFlags(&e0, hipEventDisableTiming));
    h->ev_in.push_back(e0);
    HIP_TRY(hipEventCreateWithFlags(&e1, hipEventDisableTiming));
    h->ev_done.push_back(e1);
  }
  const bool tbf = layout == CRISPY_RN_LAYOUT_TBF;
  const long stride_t = tbf ? (long)B * RN_FRAME : (long)RN_FRAME;
  const long stride_b = tbf ? (long)RN_FRAME : (long)n_frames * RN_FRAME;
  // piece [t0, t0 + T): contiguous in TBF, B rows of T * 480 floats at a pitch of n_frames * 480 in BTF
  auto copy_piece = [&](void* dst_v, const void* src_v, int t0, int T, hipMemcpyKind kind, hipStream_t st) -> hipError_t {
    char* dst = static_cast<char*>(dst_v);
    const char* src = static_cast<const char*>(src_v);
    if (tbf)
      return hipMemcpyAsync(dst + (size_t)t0 * stride_t * esz, src + (size_t)t0 * stride_t * esz, (size_t)T * frame_bytes, kind, st);
    const size_t pitch = (size_t)n_frames * RN_FRAME * esz;
    return hipMemcpy2DAsync(dst + (size_t)t0 * RN_FRAME * esz, pitch, src + (size_t)t0 * RN_FRAME * esz, pitch,
                            (size_t)T * RN_FRAME * esz, B, kind, st);
  };
  std::atomic<int> recorded{0};          // pieces whose "frame kernels done" event has been recorded by this call
  std::atomic<bool> abort_flag{false};
  hipError_t drain_err = hipSuccess;
  std::thread drain([&]() noexcept {
    if (hipSetDevice(h->device) != hipSuccess) { drain_err = hipErrorInvalidDevice; return; }
    for (int i = 0; i < n_pieces; ++i) {
      while (recorded.load(std::memory_order_acquire) <= i) {
        if (abort_flag.load(std::memory_order_acquire)) return;
        std::this_thread::yield();
      }
      const int t0 = i * P, T = (n_frames - t0) < P ? (n_frames - t0) : P;
      hipError_t e = hipStreamWaitEvent(h->d2h_stream, h->ev_done[i], 0);
      if (e == hipSuccess) e = copy_piece(out, h->d_stage_out, t0, T, hipMemcpyDeviceToHost, h->d2h_stream);
      if (e == hipSuccess && vad)
        e = hipMemcpyAsync(vad + (size_t)t0 * B, h->d_stage_vad + (size_t)t0 * B, (size_t)T * B * sizeof(float),
                           hipMemcpyDeviceToHost, h->d2h_stream);
      if (e != hipSuccess) { drain_err = e; return; }
    }
    drain_err = hipStreamSynchronize(h->d2h_stream);
  });
  // an exception between here and the join below (the thread constructor itself throws std::system_error before this
  // line) must not destroy a joinable std::thread -- that is std::terminate, i.e. an abort of the Rust host
  struct JoinOnUnwind {
    std::thread& t;
    std::atomic<bool>& stop;
    ~JoinOnUnwind() {
      if (t.joinable()) { stop.store(true, std::memory_order_release); t.join(); }
    }
  } join_on_unwind{drain, abort_flag};
  int rc = CRISPY_OK;
  hipError_t feed_err = hipSuccess;
  for (int i = 0; i < n_pieces && rc == CRISPY_OK && feed_err == hipSuccess; ++i) {
    const int t0 = i * P, T = (n_frames - t0) < P ? (n_frames - t0) : P;
    feed_err = copy_piece(h->d_stage_in, in, t0, T, hipMemcpyHostToDevice, h->h2d_stream);
    if (feed_err == hipSuccess) feed_err = hipEventRecord(h->ev_in[i], h->h2d_stream);
    if (feed_err == hipSuccess) feed_err = hipStreamWaitEvent(h->stream, h->ev_in[i], 0);
    if (feed_err != hipSuccess) break;
    rc = process_device_impl(h, reinterpret_cast<const char*>(h->d_stage_in) + (size_t)t0 * stride_t * esz,
                             reinterpret_cast<char*>(h->d_stage_out) + (size_t)t0 * stride_t * esz,
                             vad ? h->d_stage_vad + (size_t)t0 * B : nullptr, nullptr, T, stride_t, stride_b, h->stream, s16);
    if (rc != CRISPY_OK) break;
    feed_err = hipEventRecord(h->ev_done[i], h->stream);
    if (feed_err == hipSuccess) recorded.store(i + 1, std::memory_order_release);
  }
  if (rc != CRISPY_OK || feed_err != hipSuccess) abort_flag.store(true, std::memory_order_release);
  drain.join();
  (void)hipStreamSynchronize(h->h2d_stream);
  (void)hipStreamSynchronize(h->stream);
  if (rc != CRISPY_OK) return rc;
  if (feed_err != hipSuccess)
    return fail(CRISPY_ERR_HIP, "crispy_rn_process: copy-in / launch failed: %s", hipGetErrorString(feed_err));
  if (drain_err != hipSuccess)
    return fail(CRISPY_ERR_HIP, "crispy_rn_process: copy-out failed: %s", hipGetErrorString(drain_err));
  return CRISPY_OK;
}
}  // namespace

extern "C" {

int crispy_rn_synchronize(crispy_rn* h) try {
  if (!h) return fail(CRISPY_ERR_INVALID_ARG, "crispy_rn_synchronize: NULL handle");
  HIP_TRY(hipSetDevice(h->device));
  HIP_TRY(hipStreamSynchronize(h->stream));
  return CRISPY_OK;
} CRISPY_CATCH_RET("crispy_rn_synchronize")

int crispy_rn_set_timing(crispy_rn* h, int enable) try {
  if (!h) return fail(CRISPY_ERR_INVALID_ARG, "crispy_rn_set_timing: NULL handle");
  h->timing = enable != 0;
  h->ev_used = 0;
  h->seg_subs.clear();
  return CRISPY_OK;
} CRISPY_CATCH_RET("crispy_rn_set_timing")

int crispy_rn_last_kernel_ms(crispy_rn* h, float* frame_kernel_ms, float* total_ms) try {
  if (!h) return fail(CRISPY_ERR_INVALID_ARG, "crispy_rn_last_kernel_ms: NULL handle");
  if (!h->timing || h->ev_used == 0)
    return fail(CRISPY_ERR_INVALID_ARG, "crispy_rn_last_kernel_ms: no timed call recorded");
  HIP_TRY(hipSetDevice(h->device));
  float fk = 0.f, tot = 0.f;
  size_t i = 0;
  const bool timeline = dev_env("CRISPY_RN_TIMELINE") != nullptr;   // developer aid: where a step's time goes (stderr)
  for (int n_sub : h->seg_subs) {
    const size_t last = i + 1 + 2 * (size_t)n_sub;
    HIP_TRY(hipEventSynchronize(h->ev[last]));
    float ms = 0.f;
    for (int k = 0; k < n_sub; ++k) {
      HIP_TRY(hipEventElapsedTime(&ms, h->ev[i + 1 + 2 * k], h->ev[i + 2 + 2 * k]));
      fk += ms;
      if (timeline) {
        float t0 = 0.f, gap = 0.f;
        HIP_TRY(hipEventElapsedTime(&t0, h->ev[i], h->ev[i + 1 + 2 * k]));
        HIP_TRY(hipEventElapsedTime(&gap, h->ev[i + 2 * k], h->ev[i + 1 + 2 * k]));   // since the previous frame kernel ended (k = 0: since the segment began)
        std::fprintf(stderr, "[rn timeline] sub-chunk %d: starts at %.3f ms (%.3f ms after the previous one ended), frame kernel %.3f ms\n", k, t0, gap, ms);
      }
    }
    HIP_TRY(hipEventElapsedTime(&ms, h->ev[i], h->ev[last]));
    tot += ms;
    i = last + 1;
  }
  if (frame_kernel_ms) *frame_kernel_ms = fk;
  if (total_ms) *total_ms = tot;
  return CRISPY_OK;
} CRISPY_CATCH_RET("crispy_rn_last_kernel_ms")

int crispy_rn_stage_tansig_device(crispy_rn* h, const float* d_x, float* d_y, size_t n, int sigmoid, void* hip_stream) try {
  if (!h) return fail(CRISPY_ERR_INVALID_ARG, "crispy_rn_stage_tansig_device: NULL handle");
  if (n == 0) return CRISPY_OK;
  if (!d_x || !d_y) return fail(CRISPY_ERR_INVALID_ARG, "crispy_rn_stage_tansig_device: NULL pointer");
  if (n > ((size_t)1 << 40)) return fail(CRISPY_ERR_INVALID_ARG, "crispy_rn_stage_tansig_device: n too large");
  HIP_TRY(hipSetDevice(h->device));
  HIP_TRY(rn_launch_tansig(h->d_tab, d_x, d_y, (long)n, sigmoid != 0, hip_stream ? (hipStream_t)hip_stream : h->stream));
  return CRISPY_OK;
} CRISPY_CATCH_RET("crispy_rn_stage_tansig_device")

int crispy_rn_debug_capture(crispy_rn* h, int enable) try {
  if (!h) return fail(CRISPY_ERR_INVALID_ARG, "crispy_rn_debug_capture: NULL handle");
  HIP_TRY(hipSetDevice(h->device));
  if (enable && !h->d_dbg) {
    HIP_TRY(hipMalloc(&h->d_dbg, (size_t)h->B * RN_DBG_FLOATS * sizeof(float)));
    HIP_TRY(hipMemset(h->d_dbg, 0, (size_t)h->B * RN_DBG_FLOATS * sizeof(float)));
    HIP_TRY(hipDeviceSynchronize());        // a NULL-stream memset is not ordered against the handle's non-blocking stream
  } else if (!enable && h->d_dbg) {
    HIP_TRY(hipStreamSynchronize(h->stream));
    HIP_TRY(hipFree(h->d_dbg));
    h->d_dbg = nullptr;
  }
  return CRISPY_OK;
} CRISPY_CATCH_RET("crispy_rn_debug_capture")

int crispy_rn_debug_read(crispy_rn* h, int stream, float* dst, size_t n_floats) try {
  if (!h || !dst) return fail(CRISPY_ERR_INVALID_ARG, "crispy_rn_debug_read: NULL argument");
  if (!h->d_dbg) return fail(CRISPY_ERR_INVALID_ARG, "crispy_rn_debug_read: capture not enabled");
  if (stream < 0 || stream >= h->B || n_floats > (size_t)RN_DBG_FLOATS)
    return fail(CRISPY_ERR_INVALID_ARG, "crispy_rn_debug_read: bad stream/size");
  HIP_TRY(hipSetDevice(h->device));
  HIP_TRY(hipStreamSynchronize(h->stream));
  HIP_TRY(hipMemcpy(dst, h->d_dbg + (size_t)stream * RN_DBG_FLOATS, n_floats * sizeof(float),
                    hipMemcpyDeviceToHost));
  return CRISPY_OK;
} CRISPY_CATCH_RET("crispy_rn_debug_read")

}  // extern "C"
