// rn_kernels.hip -- batched RNNoise frame pipeline for MI355X (gfx950, wave64).
//
// Replaces nnnoiseless::DenoiseState::process_frame (reference call site
// src-tauri/src/audio.rs:268) for B independent streams.  Algorithm: SURVEY.md Appendix A.
//
// Mapping
//   rn_highpass_kernel   (rn_highpass.hip) one LANE per stream: the biquad is a strict 480-step recurrence
//                        with f32 state rounding, so it is run sequentially per stream, 64 streams per
//                        wave, and its output is written to a per-stream contiguous history
//                        (xhp) that the frame kernel reads coalesced.
//   rn_frame_kernel      one WAVE per stream, persistent over the T frames of a call.  The wave
//                        keeps its stream's spectra / pitch buffers / GRU state in LDS; HBM sees
//                        only the 480 in + 480 out samples per frame plus L2-resident re-reads of
//                        the high-passed history.  64-thread workgroups: every barrier is a
//                        single-wave barrier.
//   rn_roll_history      (rn_highpass.hip) keeps the last 4 high-passed frames for the next call.
//
// The analysis window [x_prev, x_cur] and the 1728-sample pitch buffer of the reference are both
// windows of the same high-passed signal, so neither is stored as state: they are views of xhp.
#include <type_traits>
#include "rn_common.h"
#include "rn_wave_sums.h"

namespace crispy {
namespace {

constexpr int WAVE = 64;

// In-kernel stage stamps (diagnostic build only: make PROFILE=1 -> libcrispy_hip_prof.so).
#ifdef RN_PROFILE
// Stage stamps accumulate straight into the debug buffer (one lane, global read-modify-write): 24 counters in
// registers pushed the frame loop into scratch, which this compiler does not handle safely (see dotn_h).
#define RN_PROF_DECL long long tprev_ = clock64(); \
  float* profp_ = (DBG && a.dbg) ? a.dbg + (long)blockIdx.x * RN_DBG_FLOATS + 3824 : nullptr;
#define STAMP(k) { const long long tn_ = clock64(); if (profp_ && threadIdx.x == 0) profp_[k] += (float)(tn_ - tprev_); tprev_ = tn_; }
#elif defined(RN_STOP_AFTER)
// Diagnostic builds (tools/lds_by_stage.sh): the frame ends at stamp RN_STOP_AFTER, so that the difference of a hardware
// counter between two such builds is what one stage contributes.  Results are meaningless; never shipped.
#define RN_PROF_DECL
#define STAMP(k) { if ((k) == RN_STOP_AFTER) { __syncthreads(); continue; } }
#else
#define RN_PROF_DECL
#define STAMP(k)
#endif

__device__ __constant__ int c_second_check[16] = {0, 0, 3, 2, 3, 2, 5, 2, 3, 2, 3, 2, 5, 2, 3, 2};

// (A hand-scheduled two-instruction form -- v_pk_mul_f32 + v_pk_fma_f32 with op_sel / neg_lo -- removes 230 VALU
// instructions per frame and measured 4 % SLOWER, round 2: the frame kernel is bound by dependent chains, and a packed
// result has to be waited for before its first reader.)
__device__ __forceinline__ float2 cmul(float2 a, float2 b) {
  return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ float2 cconj(float2 a) { return make_float2(a.x, -a.y); }

// DPP lane permutations (gfx9 family): no LDS crossbar round trip, one VALU op per step.
// (`old` = 0: passing the value itself -- legal for the quad / mirror permutations, whose every lane has a valid source --
// ties the destination to the source register and cost 83 more copies than the zero-initialising moves it removed.)
template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ float dpp_mov(float v) {
  const int iv = __float_as_int(v);
  return __int_as_float(__builtin_amdgcn_update_dpp(0, iv, CTRL, ROW_MASK, 0xf, false));
}
// v + (v of the permuted lane) as ONE v_add_f32_dpp: the DPP combiner only folds a v_mov_b32_dpp into its user inside one
// basic block, and the compiler likes to sink the add into a following `if (lane ...) store` -- then the move, its
// zero-initialised destination and the add are three instructions.  Pinning the sum (an empty asm that "uses" it) keeps
// the add where the move is.
template <int CTRL>
__device__ __forceinline__ float dpp_add(float v) {
  float r = v + dpp_mov<CTRL>(v);
  asm volatile("" : "+v"(r));
  return r;
}
// sum over the 64 lanes, result uniform (read from lane 63 into an SGPR)
__device__ __forceinline__ float wave_sum(float v) {
  v += dpp_mov<0xB1>(v);        // quad_perm [1,0,3,2]
  v += dpp_mov<0x4E>(v);        // quad_perm [2,3,0,1]
  v += dpp_mov<0x141>(v);       // row_half_mirror
  v += dpp_mov<0x140>(v);       // row_mirror: every lane of a 16-lane row holds the row sum
  v += dpp_mov<0x142, 0xa>(v);  // row_bcast:15 into rows 1 and 3
  v += dpp_mov<0x143, 0xc>(v);  // row_bcast:31 into rows 2 and 3
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

// ---------------------------------------------------------------------------------------------
// 480-point complex FFT, in place in LDS, one wave.  Stockham passes with register staging:
// every lane reads the inputs of its butterflies, the wave synchronises, every lane writes.
// ---------------------------------------------------------------------------------------------
// Synchronisation of the lanes that work on one stream.  NW = 1 (the default kernel): the workgroup IS the wave, and
// rn_sync<NW>() is the cheapest statement of "this wave's LDS traffic has landed".  NW = 3 (the stage-pipelined kernel
// for small stream counts): the workgroup holds three waves that are at DIFFERENT points of different frames between
// two ticks, so inside a stage only the wave itself may be waited for; the workgroup barrier stands at the tick.
template <int NW>
__device__ __forceinline__ void rn_sync() {
  if constexpr (NW == 1) {
    __syncthreads();
  } else {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_s_waitcnt(0xc07f);      // lgkmcnt(0)
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
}

template <int R>
__device__ __forceinline__ void butterfly(const float2 (&v)[R], float2 (&o)[R]);

template <>
__device__ __forceinline__ void butterfly<4>(const float2 (&v)[4], float2 (&o)[4]) {
  const float2 s0 = cadd(v[0], v[2]), d0 = csub(v[0], v[2]);
  const float2 s1 = cadd(v[1], v[3]), d1 = csub(v[1], v[3]);
  o[0] = cadd(s0, s1);
  o[2] = csub(s0, s1);
  o[1] = make_float2(d0.x + d1.y, d0.y - d1.x);  // d0 - i d1
  o[3] = make_float2(d0.x - d1.y, d0.y + d1.x);  // d0 + i d1
}
template <>
__device__ __forceinline__ void butterfly<3>(const float2 (&v)[3], float2 (&o)[3]) {
  const float2 t = cadd(v[1], v[2]), d = csub(v[1], v[2]);
  o[0] = cadd(v[0], t);
  const float2 a = make_float2(v[0].x - 0.5f * t.x, v[0].y - 0.5f * t.y);
  const float s = 0.86602540378443864676f;
  const float2 b = make_float2(s * d.x, s * d.y);
  o[1] = make_float2(a.x + b.y, a.y - b.x);  // a - i b
  o[2] = make_float2(a.x - b.y, a.y + b.x);  // a + i b
}
template <>
__device__ __forceinline__ void butterfly<5>(const float2 (&v)[5], float2 (&o)[5]) {
  const float c1 = 0.30901699437494742410f, s1 = 0.95105651629515357212f;
  const float c2 = -0.80901699437494742410f, s2 = 0.58778525229247312917f;
  const float2 t1 = cadd(v[1], v[4]), t2 = cadd(v[2], v[3]);
  const float2 t3 = csub(v[1], v[4]), t4 = csub(v[2], v[3]);
  o[0] = make_float2(v[0].x + t1.x + t2.x, v[0].y + t1.y + t2.y);
  const float2 a1 = make_float2(v[0].x + c1 * t1.x + c2 * t2.x, v[0].y + c1 * t1.y + c2 * t2.y);
  const float2 a2 = make_float2(v[0].x + c2 * t1.x + c1 * t2.x, v[0].y + c2 * t1.y + c1 * t2.y);
  const float2 b1 = make_float2(s1 * t3.x + s2 * t4.x, s1 * t3.y + s2 * t4.y);
  const float2 b2 = make_float2(s2 * t3.x - s1 * t4.x, s2 * t3.y - s1 * t4.y);
  o[1] = make_float2(a1.x + b1.y, a1.y - b1.x);
  o[4] = make_float2(a1.x - b1.y, a1.y + b1.x);
  o[2] = make_float2(a2.x + b2.y, a2.y - b2.x);
  o[3] = make_float2(a2.x - b2.y, a2.y + b2.x);
}

template <>
__device__ __forceinline__ void butterfly<8>(const float2 (&v)[8], float2 (&o)[8]) {
  // two radix-4 transforms of the even / odd inputs, then the w8^k twiddles (1, (1-i)/sqrt2, -i, (-1-i)/sqrt2)
  const float2 ev[4] = {v[0], v[2], v[4], v[6]}, od[4] = {v[1], v[3], v[5], v[7]};
  float2 e[4], d[4];
  butterfly<4>(ev, e);
  butterfly<4>(od, d);
  const float h = 0.70710678118654752f;
  const float2 t0 = d[0];
  const float2 t1 = make_float2((d[1].x + d[1].y) * h, (d[1].y - d[1].x) * h);
  const float2 t2 = make_float2(d[2].y, -d[2].x);
  const float2 t3 = make_float2((d[3].y - d[3].x) * h, -(d[3].x + d[3].y) * h);
  o[0] = cadd(e[0], t0); o[4] = csub(e[0], t0);
  o[1] = cadd(e[1], t1); o[5] = csub(e[1], t1);
  o[2] = cadd(e[2], t2); o[6] = csub(e[2], t2);
  o[3] = cadd(e[3], t3); o[7] = csub(e[3], t3);
}

// Where element i of the Stockham sequence lives between two passes.  In natural order the stores of the first two
// passes collide: a store is serviced in groups of 16 lanes over 32 banks, and `buf[4 j + r]` (radix 4, stride 1) puts
// lanes j and j + 4, `buf[32 (j / 4) + (j % 4) + 4 r]` (radix 8, stride 4) all lanes with equal j % 4, on one bank
// -- 4-way, 16 LDS cycles per store instead of 4.  Both hand-offs are private to two passes, so they use layouts
// in which the store *and* the matching load are conflict-free:
//   pass 1 -> 2: element i at (i % 4) * 120 + i / 4  (the store becomes r * 120 + j: consecutive lanes; the load of
//                butterfly j reads (j % 4) * 120 + j / 4 + 15 r: four 16-bank windows 0 / 48 / 32 / 16 apart)
//   pass 2 -> 3: element i = 32 a + 4 r + m at i ^ ((a & 3) << 2): the four a of a 16-lane group spread over the
//                four bank quarters; a 32-aligned run of the radix-3 load stays a permutation of one 32-block
// Register caps that let a high-pass wave run *beside* four frame waves of a SIMD instead of waiting for one of them
// to finish: the frame kernel at 120 of its 128 registers (no spill store in the frame loop, same speed), the
// high-pass kernel (rn_highpass.hip) at 32 (two float4 blocks of input in flight instead of eight).
// amdgpu_num_vgpr counts the unified VGPR + AGPR file on gfx950, so the attribute wants half the number (120 / 32
// given directly are silently ignored).  Measured: 7.71 -> 7.585 ms per 100-frame step.
#define RN_VGPR_CAP __attribute__((amdgpu_num_vgpr(60)))
template <int R, int NS>
__device__ __forceinline__ int fft_rd(int j, int r) {
  constexpr int M = 480 / R;
  if (R == 8 && NS == 4) return (j & 3) * 120 + (j >> 2) + 15 * r;
  if (R == 3 && NS == 32) {
    const int i = j + r * M;
    return i ^ (((i >> 5) & 3) << 2);
  }
  return j + r * M;
}
template <int R, int NS>
__device__ __forceinline__ int fft_wr(int j, int r) {
  if (R == 4 && NS == 1) return r * 120 + j;
  if (R == 8 && NS == 4) {
    const int a = j >> 2, m = j & 3;
    return 32 * a + m + 4 * (r ^ (a & 3));
  }
  const int k = j % NS;
  return (j / NS) * NS * R + k + r * NS;
}
template <int NW, int R, int NS>
__device__ __forceinline__ void fft_pass(float2* buf, const float2* __restrict__ w960, int lane) {
  constexpr int M = 480 / R;
  constexpr int NBF = (M + WAVE - 1) / WAVE;
  float2 o[NBF][R];
  // twiddles of every trip first (clamped butterfly index): one exposed table round trip per pass, not one per trip
  float2 tw[NBF][R];
  if (NS > 1) {
#pragma unroll
    for (int nb = 0; nb < NBF; ++nb) {
      const int k = min(lane + WAVE * nb, M - 1) % NS;
#pragma unroll
      for (int r = 1; r < R; ++r) tw[nb][r] = w960[k * r * (960 / (NS * R))];
    }
  }
#pragma unroll
  for (int nb = 0; nb < NBF; ++nb) {
    const int j = lane + WAVE * nb;
    if (j < M) {
      float2 v[R];
#pragma unroll
      for (int r = 0; r < R; ++r) {
        float2 x = buf[fft_rd<R, NS>(j, r)];
        if (NS > 1 && r > 0) x = cmul(x, tw[nb][r]);
        v[r] = x;
      }
      butterfly<R>(v, o[nb]);
    }
  }
  rn_sync<NW>();
#pragma unroll
  for (int nb = 0; nb < NBF; ++nb) {
    const int j = lane + WAVE * nb;
    if (j < M) {
#pragma unroll
      for (int r = 0; r < R; ++r) buf[fft_wr<R, NS>(j, r)] = o[nb][r];
    }
  }
  rn_sync<NW>();
}

// forward DFT of the 480 complex points in buf (unscaled, natural order); caller synchronised
template <int NW>
__device__ __forceinline__ void fft480(float2* buf, const float2* __restrict__ w960, int lane) {
  fft_pass<NW, 4, 1>(buf, w960, lane);
  fft_pass<NW, 8, 4>(buf, w960, lane);      // 480 = 4 . 8 . 3 . 5: four LDS round trips instead of five (4 . 4 . 2 . 3 . 5)
  fft_pass<NW, 3, 32>(buf, w960, lane);
  fft_pass<NW, 5, 96>(buf, w960, lane);
}

// The same transform with the first pass (radix 4, stride 1) fed by a loader instead of LDS: in(j, r) returns point
// j + 120 r.  The analysis transforms window their input straight from global memory this way, which saves the
// separate "window -> LDS -> barrier -> read back" phase.  buf must be free (caller synchronised).
template <int NW, class In>
__device__ __forceinline__ void fft480_from(float2* buf, In in, const float2* __restrict__ w960, int lane) {
  {
    float2 o[2][4];
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
      const int j = lane + WAVE * nb;
      if (j < 120) {
        float2 v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = in(j, r);
        butterfly<4>(v, o[nb]);
      }
    }
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
      const int j = lane + WAVE * nb;
      if (j < 120) {
#pragma unroll
        for (int r = 0; r < 4; ++r) buf[fft_wr<4, 1>(j, r)] = o[nb][r];
      }
    }
    rn_sync<NW>();
  }
  fft_pass<NW, 8, 4>(buf, w960, lane);
  fft_pass<NW, 3, 32>(buf, w960, lane);
  fft_pass<NW, 5, 96>(buf, w960, lane);
}

// buf holds Z = FFT480(x[2n] + i x[2n+1]); turn it into X[0..480] = DFT960(x)/960 in place.
template <int NW>
__device__ __forceinline__ void real_fwd_post(float2* buf, const float2* __restrict__ w960, int lane) {
  const float scale = 1.0f / 960.0f;
  // table loads of all four trips first (clamped index): inside the `k <= 240` bodies each of them was a load - wait
  float2 wk[4], wn[4];
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    const int k = min(lane + WAVE * m, 240);
    wk[m] = w960[k];
    wn[m] = w960[480 - k];
  }
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    const int k = lane + WAVE * m;
    if (k <= 240) {
      const float2 zk = buf[k];
      const float2 zn = (k == 0) ? zk : buf[480 - k];
      // X[k]
      float2 zc = cconj(zn);
      float2 fe = make_float2(0.5f * (zk.x + zc.x), 0.5f * (zk.y + zc.y));
      float2 d = make_float2(0.5f * (zk.x - zc.x), 0.5f * (zk.y - zc.y));
      float2 t = cmul(wk[m], make_float2(d.y, -d.x));
      const float2 xk = make_float2((fe.x + t.x) * scale, (fe.y + t.y) * scale);
      // X[480-k]
      zc = cconj(zk);
      fe = make_float2(0.5f * (zn.x + zc.x), 0.5f * (zn.y + zc.y));
      d = make_float2(0.5f * (zn.x - zc.x), 0.5f * (zn.y - zc.y));
      t = cmul(wn[m], make_float2(d.y, -d.x));
      const float2 xn = make_float2((fe.x + t.x) * scale, (fe.y + t.y) * scale);
      buf[k] = xk;
      buf[480 - k] = xn;
    }
  }
  rn_sync<NW>();
}

// buf holds X[0..480]; replace it by conj(Z) with Z[k] = (X[k]+conj X[480-k]) + i w^-k (X[k]-conj X[480-k])
// so that a forward FFT yields conj of the interleaved time signal.
template <int NW>
__device__ __forceinline__ void real_inv_pre(float2* buf, const float2* __restrict__ w960, int lane) {
  float2 wk[4], wn[4];
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    const int k = min(lane + WAVE * m, 240);
    wk[m] = w960[k];
    wn[m] = w960[480 - k];
  }
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    const int k = lane + WAVE * m;
    if (k <= 240) {
      const float2 a = buf[k];
      const float2 bn = buf[480 - k];
      float2 b = cconj(bn);
      float2 fe = cadd(a, b);
      float2 fo = cmul(csub(a, b), cconj(wk[m]));
      const float2 zk = make_float2(fe.x - fo.y, -(fe.y + fo.x));
      b = cconj(a);
      fe = cadd(bn, b);
      fo = cmul(csub(bn, b), cconj(wn[m]));
      const float2 zn = make_float2(fe.x - fo.y, -(fe.y + fo.x));
      buf[k] = zk;
      if (k > 0) buf[480 - k] = zn;
    }
  }
  rn_sync<NW>();
}

// ---------------------------------------------------------------------------------------------
// Opus-band helpers (Appendix A.3 step 2).  `part` is 200 floats of scratch; e0/e1/em1 are this
// lane's band edges (band = lane) in 4-bin chunks, loaded once per kernel.
// ---------------------------------------------------------------------------------------------
struct BandEdges {
  int piece;           // RnTables::band_piece[lane]
};

// Sum of one band's chunk partials: part_hi[c] for c in [em1, e0) -- the rising half of the previous interval -- plus
// part_lo[c] for c in [e0, e1), doubled at the two edge bands.  ALL 64 lanes call it; the result is valid in lanes
// < RN_NB (lane == band).  Round 3 form: lanes 0..21 add up the rising halves, lanes 32..53 the falling halves of band
// (lane - 32), one ds_bpermute joins them -- 24 select-and-add steps per call instead of 48 -- and the `k < n` tests
// stay INLINE (`n` is laundered): they are invariant across frames, and hoisted out of the frame loop the compiler
// turned them into 48 lane masks = 96 SGPRs, spilled into two VGPRs' lanes and fetched back with two v_readlane per
// use: 416 of the kernel's 7 800 VALU instructions per frame were those v_readlane (ISA census, NOTEBOOK.md section 4 v8).
// The partials are *read* eight at a time (as plain `for (c = ...) sum += part[c]` loops every chunk was its own
// LDS round trip); reads past a lane's range stay inside the workgroup's LDS and are replaced by 0.f.
__device__ __forceinline__ float band_sum(const float* part_lo, const float* part_hi, const BandEdges& be, int lane) {
  // Round 3, second form.  A band is up to 22 + 22 chunk partials but most are 1 - 4: with one or two lanes per band the
  // wave ran the longest band's 24 select-and-add steps (3 instructions and an LDS read each) for every band.  Here the 42
  // half-bands are cut into 56 pieces of <= 6 chunks, one per lane (RnTables::band_piece, built on the host): 6 steps, then
  // the <= 4 pieces of a half -- neighbouring lanes of one DPP row -- are joined by two shifted adds, and lane == band
  // fetches its two halves with two ds_bpermute.  ~40 instructions and 8 LDS operations per call instead of ~84 and 25.
  int bd = be.piece;
  asm volatile("" : "+v"(bd));      // (laundered: the tests below are invariant across frames, and hoisted out of the frame
                                    // loop they become lane masks in SGPRs that spill -- see the comment above)
  const int n = (bd >> 7) & 7;
  const float* p = ((bd >> 10) & 1 ? part_lo : part_hi) + (bd & 127);
  float v[6];
#pragma unroll
  for (int k = 0; k < 6; ++k) v[k] = p[k];          // (up to five floats past the piece: still this workgroup's LDS)
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < 6; ++k) s += k < n ? v[k] : 0.f;
  {
    const float t1 = dpp_mov<0x101>(s);             // row_shl:1 -- the next lane's piece (0 past the row's end)
    s += (bd >> 11) & 1 ? t1 : 0.f;
    const float t2 = dpp_mov<0x102>(s);             // row_shl:2
    s += (bd >> 12) & 1 ? t2 : 0.f;
  }
  const float rise = __int_as_float(__builtin_amdgcn_ds_bpermute(((bd >> 13) & 63) << 2, __float_as_int(s)));
  const float fall = __int_as_float(__builtin_amdgcn_ds_bpermute(((bd >> 19) & 63) << 2, __float_as_int(s)));
  float tot = rise + fall;                           // lanes >= RN_NB: band 0's value, never used
  if (lane == 0 || lane == RN_NB - 1) tot *= 2.f;
  return tot;
}

// Band energies in the pair layout of the comb-filter stage: lane handles bins (2p, 2p+1), p = lane + 64 m, with
// one ds_read_b128 per spectrum; the two pairs of a 4-bin chunk are neighbouring lanes (one DPP add), even lanes
// write the chunk partials, 22 lanes add them up (deterministic, no atomics).  With CORR the same pass also yields
// the band correlation Re(X P*) against a second spectrum and parks S in global memory from the registers.
//   E[band]  of S            -> Eout
//   C[band]  of (Xc, S)      -> Cout   (CORR)
template <int NW, bool CORR>
__device__ __forceinline__ void band_pairs(const float2* S, const float2* Xc, float* part, float* Eout, float* Cout,
                                           float2* park, const RnTables* __restrict__ tab, const BandEdges& be,
                                           int lane) {
  float clo[4], chi[4];
  float2 fr[4];
#pragma unroll
  for (int m = 0; m < 4; ++m) fr[m] = *reinterpret_cast<const float2*>(tab->bin_frac + 2 * min(lane + WAVE * m, 199));
  // the spectrum pairs of all four trips first (clamped index): inside the `pidx < 200` bodies every trip was an LDS
  // read - wait - use round trip
  float4 svq[4];
#pragma unroll
  for (int m = 0; m < 4; ++m) svq[m] = *reinterpret_cast<const float4*>(S + 2 * min(lane + WAVE * m, 199));
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    const int pidx = lane + WAVE * m;
    clo[m] = 0.f; chi[m] = 0.f;
    if (pidx < 200) {
      const float2 f = fr[m];
      const float4 sv = svq[m];
      float e0 = sv.x * sv.x; e0 += sv.y * sv.y;
      float e1 = sv.z * sv.z; e1 += sv.w * sv.w;
      float lo = (1.f - f.x) * e0 + (1.f - f.y) * e1;
      float hi = f.x * e0 + f.y * e1;
      lo = dpp_add<0xB1>(lo);
      hi = dpp_add<0xB1>(hi);
      if ((lane & 1) == 0) { part[pidx >> 1] = lo; part[100 + (pidx >> 1)] = hi; }
      if (CORR) {
        const float4 xv = *reinterpret_cast<const float4*>(Xc + 2 * pidx);
        float c0 = xv.x * sv.x; c0 += xv.y * sv.y;
        float c1 = xv.z * sv.z; c1 += xv.w * sv.w;
        float l2 = (1.f - f.x) * c0 + (1.f - f.y) * c1;
        float h2 = f.x * c0 + f.y * c1;
        clo[m] = dpp_add<0xB1>(l2);
        chi[m] = dpp_add<0xB1>(h2);
        if (park) *reinterpret_cast<float4*>(park + 2 * pidx) = sv;
      }
    }
  }
  auto band_total = [&](float* out) {
    rn_sync<NW>();
    const float bsum = band_sum(part, part + 100, be, lane);
    if (lane < RN_NB) out[lane] = bsum;
    rn_sync<NW>();
  };
  band_total(Eout);
  if (CORR) {
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const int pidx = lane + WAVE * m;
      if (pidx < 200 && (lane & 1) == 0) { part[pidx >> 1] = clo[m]; part[100 + (pidx >> 1)] = chi[m]; }
    }
    band_total(Cout);
  }
}

// ---------------------------------------------------------------------------------------------
// RNN (Appendix A.3 step 6): lane == output row, weights as packed int8 dwords from L2.
// ---------------------------------------------------------------------------------------------
// The 201-entry tanh table lives in four registers per lane (lane l holds T[l], T[64+l], T[128+l], T[192+l]) for the
// duration of the gain network; a lookup is four ds_bpermute (LDS crossbar, no memory access) and a select instead
// of a dependent global load (~1 us at this occupancy, six of them per frame on the critical path).
// ds_bpermute returns 0 for source lanes that are masked off, so activations are evaluated with ALL lanes active and
// only the stores are predicated.
struct TansigTab {
  float t[4];
  __device__ __forceinline__ void load(const float* __restrict__ table, int lane) {
#pragma unroll
    for (int q = 0; q < 4; ++q) t[q] = table[min(64 * q + lane, 200)];
  }
  __device__ __forceinline__ float at(int i) const {
    const int addr = (i & 63) << 2;
    const float v0 = __int_as_float(__builtin_amdgcn_ds_bpermute(addr, __float_as_int(t[0])));
    const float v1 = __int_as_float(__builtin_amdgcn_ds_bpermute(addr, __float_as_int(t[1])));
    const float v2 = __int_as_float(__builtin_amdgcn_ds_bpermute(addr, __float_as_int(t[2])));
    const float v3 = __int_as_float(__builtin_amdgcn_ds_bpermute(addr, __float_as_int(t[3])));
    const int q = i >> 6;
    return q == 0 ? v0 : (q == 1 ? v1 : (q == 2 ? v2 : v3));
  }
};
// Every product and sum is rounded on its own (`fp contract(off)`: hipcc contracts a * b + c into an FMA by default,
// and HIP's __fmul_rn / __fadd_rn are plain operators that contract too), in the reference's order: the Rust crate does
// not fuse multiply-adds, and with this the function is bit-identical to the oracle on every table cell and both
// clamps (tests/test_gpu_rnnoise.py::test_tansig_and_sigmoid_every_table_cell...).
__device__ __forceinline__ float tansig_approx(float x, const TansigTab& table) {
#pragma clang fp contract(off)
  const float x0 = x;
  float sign = 1.f;
  if (x < 0.f) { x = -x; sign = -1.f; }
  x = fminf(x, 8.f);                       // keeps the index in range; the clamp result is selected below
  const int i = (int)floorf(.5f + 25.f * x);
  x = x - .04f * (float)i;
  float y = table.at(i);
  const float dy = 1.f - y * y;
  y = y + (x * dy) * (1.f - y * x);
  y = sign * y;
  if (!(x0 > -8.f)) y = -1.f;
  if (!(x0 < 8.f)) y = 1.f;                // tested last: NaN takes this branch, as in the reference's `!(x < 8)` first
  return y;
}
__device__ __forceinline__ float sigmoid_approx(float x, const TansigTab& table) {
  return .5f + .5f * tansig_approx(.5f * x, table);
}

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
// After the lane id has been laundered its range is unknown, every `table[lane + 64 k]` index is sign-extended and
// the load takes a 64-bit VGPR address (ashr + 64-bit shift-add per load).  Stating the range lets the sign extension
// fold away: SGPR base + 32-bit lane offset + immediate.
#define RN_LANE_RANGE(site) __builtin_assume((unsigned)lane < (unsigned)WAVE)
// 16-byte weight loads per row in flight ahead of their use, for 3 / 2 / 1 rows per lane: the largest that keep
// the frame loop free of spill stores (tests/test_build_resources.py)
constexpr int RN_BLK3 = 1, RN_BLK2 = 1, RN_BLK1 = 4;

typedef unsigned int rn_u4 __attribute__((ext_vector_type(4)));
// The one product that stays on the vector ALU: vad_output, 24 -> 1, on lane 0.  Its row of the f16 weight pack is
// three 16-byte buffer loads (one resource descriptor for the whole pack, the k-row offset as an immediate); even k
// into one accumulator, odd k into a second one (dependent v_fma_mix_f32 need a wait state in between).
__device__ __forceinline__ float dot_vad(__amdgpu_buffer_rsrc_t rs, int w_off, const float* x, float acc0) {
  float acc = acc0, acc1 = 0.f;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const h8 w = __builtin_bit_cast(h8, __builtin_amdgcn_raw_buffer_load_b128(rs, 0, (w_off + k) * 16, 0));
#pragma unroll
    for (int e = 0; e < 8; e += 2) {
      acc = fmaf((float)w[e], x[8 * k + e], acc);
      acc1 = fmaf((float)w[e + 1], x[8 * k + e + 1], acc1);
    }
  }
  return acc + acc1;
}

// One GRU layer (ReLU candidate).  in_vec[M] and state[N] in LDS, zero padded to multiples of 8;
// zbuf / hr: N floats of scratch each.  w_off / u_off: 16-byte offsets of the two matrices in the weight pack.
// The stage is bound by the weight stream through the CU's vector L1 (16 waves x 176 KB of f16 per frame = 71 B per
// clock against 64), not by arithmetic, so the weights stay int8 in memory (RnPack8: 16 MACs per 16-byte load) and
// the activations become *fixed point*: per vector one power-of-two scale 2^s with max|x| 2^s <= 2^30, q = rint(x 2^s),
// and q's four signed base-256 digits ((q + 0x00808080) ^ 0x00808080, byte t = digit t) are the four "A" rows of
// v_mfma_i32_4x4x4_16B_i8.  Register t of lane == row then holds sum_k digit_t(x_k) W[k][row] exactly (int32), and
// sum_t 256^t 2^-s (float)D_t is the dot product to 2^-30 of the largest activation -- finer than the rounding of an
// f32 accumulation of the same length.  The input part and the recurrent part of a GRU row have different scales,
// so the accumulators are folded into the float result where the k loop crosses from one to the other.
typedef int rn_i4 __attribute__((ext_vector_type(4)));
constexpr int RN_IMG8_LD = 144;   // bytes per digit image (K <= 128), +16: the four 16-byte reads hit distinct banks
// Wave maximum of NON-NEGATIVE floats (magnitudes), as a maximum of their bit patterns: for values >= 0 the unsigned
// integer order IS the float order, v_max_u32 needs no canonicalisation of its operands (every fmaxf compiled to a
// `v_max_f32 x, x, x` quieting step in front of the real v_max under the IEEE mode -- 170 of them per frame), and NaN
// bit patterns simply compare as large numbers (scale_of clamps the exponent).
__device__ __forceinline__ float wave_max(float v) {
  unsigned u = __float_as_uint(v);
  auto dm = [](unsigned x, auto ctrl) {
    // old = 0 with all rows and banks enabled: the combiner folds move + maximum into one v_max_u32_dpp (with old = x
    // it cannot -- x is not the identity of the operation -- and each step was a copy, a DPP move and the maximum)
    return (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, decltype(ctrl)::value, 0xf, 0xf, false);
  };
  u = max(u, dm(u, std::integral_constant<int, 0xB1>{}));
  u = max(u, dm(u, std::integral_constant<int, 0x4E>{}));
  u = max(u, dm(u, std::integral_constant<int, 0x141>{}));
  u = max(u, dm(u, std::integral_constant<int, 0x140>{}));      // every lane of a 16-lane row holds the row maximum
  unsigned m = (unsigned)__builtin_amdgcn_readlane((int)u, 0);
  m = max(m, (unsigned)__builtin_amdgcn_readlane((int)u, 16));
  m = max(m, (unsigned)__builtin_amdgcn_readlane((int)u, 32));
  m = max(m, (unsigned)__builtin_amdgcn_readlane((int)u, 48));   // scalar maxima (s_max_u32): no VALU
  return __uint_as_float(m);
}
// scale pair of a vector whose largest magnitude is m (wave-uniform): up = 2^s, dn = 2^-s
struct RnScale { float up, dn; };
__device__ __forceinline__ RnScale scale_of(float m) {
  int eb = (__float_as_int(m) >> 23) & 0xff;       // m < 2^(eb - 126)
  eb = min(max(eb, 64), 250);
  RnScale r;
  r.up = __int_as_float((283 - eb) << 23);         // 2^(30 - (eb - 126))
  r.dn = __int_as_float((eb - 29) << 23);
  return r;
}
__device__ __forceinline__ void digits_store(signed char* img, int i, float v, float up) {
  const int q = __float2int_rn(v * up);
  const unsigned r = ((unsigned)q + 0x00808080u) ^ 0x00808080u;
  img[i] = (signed char)(r & 0xff);
  img[RN_IMG8_LD + i] = (signed char)((r >> 8) & 0xff);
  img[2 * RN_IMG8_LD + i] = (signed char)((r >> 16) & 0xff);
  img[3 * RN_IMG8_LD + i] = (signed char)(r >> 24);
}
// image of the NPAD-long vector f(0..NPAD-1) (NPAD <= 128, a multiple of 16; f returns 0 in the padding)
template <int NPAD, class F>
__device__ __forceinline__ RnScale image_i8(signed char* img, int lane, F f) {
  constexpr int NT = (NPAD + WAVE - 1) / WAVE;
  float v[NT];
  float m = 0.f;
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int i = lane + WAVE * j;
    v[j] = i < NPAD ? f(min(i, NPAD - 1)) : 0.f;
    m = __uint_as_float(max(__float_as_uint(m), __float_as_uint(v[j]) & 0x7fffffffu));     // max |v| on the bit patterns
  }
  const RnScale sc = scale_of(wave_max(m));
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int i = lane + WAVE * j;
    if (i < NPAD) digits_store(img, i, v[j], sc.up);
  }
  return sc;
}
__device__ __forceinline__ rn_u4 wload8(__amdgpu_buffer_rsrc_t rs, int row16, int off16) {
  return __builtin_amdgcn_raw_buffer_load_b128(rs, row16, off16 * 16, 0);
}
// acc[r] += sa sum_k W[k][row[r]] qa[k] + sb sum_k U[k][row[r]] qb[k]; toff = (lane & 3) * RN_IMG8_LD
template <int MK16, int NK16, int ROWS, int NR>
__device__ __forceinline__ void dotn_i(__amdgpu_buffer_rsrc_t rs, int w_off, int u_off, const int (&row)[NR],
                                       const signed char* xa, const signed char* xb, int toff, float sa, float sb,
                                       float (&acc)[NR]) {
  constexpr int K16 = MK16 + NK16;
  constexpr int BLK = NR >= 3 ? RN_BLK3 : (NR == 2 ? RN_BLK2 : RN_BLK1);
  constexpr int NBLK = (K16 + BLK - 1) / BLK;
  constexpr int NC = NR >= 3 ? 1 : 2;
  int row16[NR];
#pragma unroll
  for (int r = 0; r < NR; ++r) row16[r] = row[r] * 16;
  asm volatile("" : "+v"(toff));
  const rn_u4* xa4 = reinterpret_cast<const rn_u4*>(xa + toff);
  const rn_u4* xb4 = reinterpret_cast<const rn_u4*>(xb + toff);
  rn_u4 w[2][BLK][NR];
  rn_u4 xc = MK16 > 0 ? xa4[0] : xb4[0], xn = xc;
#pragma unroll
  for (int q = 0; q < BLK; ++q)
    if (q < K16) {
#pragma unroll
      for (int r = 0; r < NR; ++r)
        w[0][q][r] = wload8(rs, row16[r], q < MK16 ? w_off + q * ROWS : u_off + (q - MK16) * ROWS);
    }
  rn_i4 a4[NR][NC];
#pragma unroll
  for (int r = 0; r < NR; ++r)
#pragma unroll
    for (int c = 0; c < NC; ++c) a4[r][c] = rn_i4{0, 0, 0, 0};
  auto fold = [&](float s) {
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      rn_i4 d = a4[r][0];
      if (NC == 2) d += a4[r][1];
      float f = (float)d[0] * s;
      f = fmaf((float)d[1], s * 256.f, f);
      f = fmaf((float)d[2], s * 65536.f, f);
      f = fmaf((float)d[3], s * 16777216.f, f);
      acc[r] += f;
#pragma unroll
      for (int c = 0; c < NC; ++c) a4[r][c] = rn_i4{0, 0, 0, 0};
    }
  };
#pragma unroll
  for (int blk = 0; blk < NBLK; ++blk) {
    if (blk + 1 < NBLK) {
#pragma unroll
      for (int q = 0; q < BLK; ++q) {
        const int k = (blk + 1) * BLK + q;
        if (k < K16) {
#pragma unroll
          for (int r = 0; r < NR; ++r)
            w[(blk + 1) & 1][q][r] = wload8(rs, row16[r], k < MK16 ? w_off + k * ROWS : u_off + (k - MK16) * ROWS);
        }
      }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int q = 0; q < BLK; ++q) {
      const int k = blk * BLK + q;
      if (k < K16) {
        if (k + 1 < K16) xn = k + 1 < MK16 ? xa4[k + 1] : xb4[k + 1 - MK16];
        if (MK16 > 0 && NK16 > 0 && k == MK16) fold(sa);
        // (not in front of the first block of a chain: its accumulators are still the constant 0, which the MFMA takes
        // as an inline operand -- laundering them there made the compiler zero 4 registers per accumulator first)
        if (!(k == 0 || (MK16 > 0 && NK16 > 0 && k == MK16))) {
#pragma unroll
          for (int r = 0; r < NR; ++r)
#pragma unroll
            for (int c = 0; c < NC; ++c) asm volatile("" : "+v"(a4[r][c]));
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
#pragma unroll
          for (int r = 0; r < NR; ++r)
            a4[r][e % NC] = __builtin_amdgcn_mfma_i32_4x4x4i8((int)xc[e], (int)w[blk & 1][q][r][e], a4[r][e % NC], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        xc = xn;
      }
    }
  }
  fold(NK16 > 0 ? sb : sa);
}
template <int K16, int ROWS>
__device__ __forceinline__ float dot_i(__amdgpu_buffer_rsrc_t rs, int w_off, int row, const signed char* x, int toff,
                                       float s, float acc0) {
  const int rows[1] = {row};
  float acc[1] = {acc0};
  dotn_i<K16, 0, ROWS, 1>(rs, w_off, w_off, rows, x, x, toff, s, s, acc);
  return acc[0];
}
// in_img / sin: digit image and 2^-s of the layer input (caller); st_img: the state's image, then h*r's at the same
// scale (|h r| <= |h|).
template <int NW, int M, int N>
__device__ __forceinline__ void gru_layer_i(__amdgpu_buffer_rsrc_t rs, int w_off, int u_off,
                                            const float* __restrict__ bias, const signed char* in_img, float sin,
                                            float* state, float* zbuf, signed char* st_img, int toff,
                                            const TansigTab& tansig, int lane) {
  constexpr int ROWS = 3 * N;
  constexpr int MK16 = (M + 15) / 16, NK16 = (N + 15) / 16;
  constexpr int NRZ = (2 * N + WAVE - 1) / WAVE, NRC = (N + WAVE - 1) / WAVE;
  const float S = 1.f / 256.f;
  const RnScale sst = image_i8<NK16 * 16>(st_img, lane, [&](int i) { return i < N ? state[min(i, N - 1)] : 0.f; });
  rn_sync<NW>();
  {
    int rows[NRZ];
    float acc[NRZ];
#pragma unroll
    for (int r = 0; r < NRZ; ++r) {
      rows[r] = min(lane + WAVE * r, 2 * N - 1);
      acc[r] = bias[rows[r]];
    }
    dotn_i<MK16, NK16, ROWS, NRZ>(rs, w_off, u_off, rows, in_img, st_img, toff, sin, sst.dn, acc);
#pragma unroll
    for (int r = 0; r < NRZ; ++r) {
      const int row = lane + WAVE * r;
      const float s = sigmoid_approx(S * acc[r], tansig);
      if (row < 2 * N) {
        if (row < N) zbuf[row] = s;
        else digits_store(st_img, row - N, state[row - N] * s, sst.up);
      }
    }
  }
  rn_sync<NW>();
  {
    int rows[NRC];
    float acc[NRC];
#pragma unroll
    for (int r = 0; r < NRC; ++r) {
      rows[r] = 2 * N + min(lane + WAVE * r, N - 1);
      acc[r] = bias[rows[r]];
    }
    dotn_i<MK16, NK16, ROWS, NRC>(rs, w_off, u_off, rows, in_img, st_img, toff, sin, sst.dn, acc);
#pragma unroll
    for (int r = 0; r < NRC; ++r) {
      const int i = lane + WAVE * r;
      if (i < N) {
        float c = S * acc[r];
        c = c < 0.f ? 0.f : c;
        const float z = zbuf[i];
        state[i] = z * state[i] + (1.f - z) * c;
      }
    }
  }
  rn_sync<NW>();
}

// ---------------------------------------------------------------------------------------------
// LDS of one stream.  RnSlot is what one frame occupies from its spectra onwards:
//   A   analysis spectrum X / synthesis                (NW = 1 also: pitch-phase scratch -- x4|y4, fine xcorr, yy_lookup)
//   Bb  pitch spectrum P, RNN vectors, band partials   (NW = 1 also: lp[864], the whitened half-rate buffer)
//   U   band partial sums while A and Bb both hold spectra | Ly, tmp22, g, r
// NW = 1 (one wave per stream): one slot + the cepstral ring + the GRU state = 10 192 bytes, 16 streams per CU.
// NW = 3 (stage pipeline, small stream counts): TWO slots (the spectra wave fills one while the synthesis wave empties
// the other), the pitch wave's own scratch, and the pitch results of two frames in flight: 26.7 KB, 6 streams per CU.
// ---------------------------------------------------------------------------------------------
struct alignas(16) RnSlot {
  float2 A[482];
  float2 Bb[482];
  float U[200];
  float Ex[24], Ep[24], Exp[24];
  int pitch_index;       // NW = 3: handed from the spectra stage to the synthesis stage with the frame
  float pitch_gain;
  int silence;
  int pad;
};
template <int NW> struct RnLdsT;
template <> struct alignas(16) RnLdsT<1> {
  RnSlot s[1];
  float ceps[8 * 22];
  float rnn_state[168];  // vad 24 | noise 48 | denoise 96
};
template <> struct alignas(16) RnLdsT<3> {
  RnSlot s[2];           // by frame parity
  float2 pa[482];        // pitch stage: scratch
  float2 pb[482];        // pitch stage: lp[864]
  float ceps[8 * 22];
  float rnn_state[168];
  int pitch_index[2];    // pitch stage -> spectra stage, by frame parity
  float pitch_gain[2];
};
static_assert(sizeof(RnLdsT<1>) <= 10240, "16 workgroups per CU need <= 10 KB of LDS each");
static_assert(6 * sizeof(RnLdsT<3>) <= 160 * 1024, "six stage-pipelined streams per CU");

// offsets inside U outside band_sums (Ly is dead once the features exist: U[48, 192) is free for the gain network)
constexpr int U_G = 0, U_R = 24, U_VAD = 46, U_LY = 48;
// ---- pitch spectrum kept in LDS (fused kernel, int8 gain network) ----
// P used to be parked in global memory between its band sums and the comb filter (3.2 KB written and read back per
// frame, 1.6 MB of L2 per XCD that the history window of the streams wants: profiles/r01q_pmc.json showed 15 KB of
// L2-miss traffic per stream-frame against 3.84 KB algorithmic).  The comb filter needs bins < 400 only, so P stays
// where its transform left it, Bb[0, 800) floats, and the gain network's 432 floats of workspace move into what is
// dead at that point: the bins >= 400 of both spectra (never read after the band energies; X's are zeroed before the
// inverse transform anyway) and the part of U behind the band values.
//   Bb[800, 842) features   Bb[842, 938) z gates   Bb[938, 962) dense layer
//   A [800, 944) digit image of the recurrent operand (4 x 144 B), then the rising-half band partials of the comb filter
//   U [ 48, 192) digit image of the layer input (4 x 144 B; Ly before it), then the falling-half band partials
constexpr int PL_FEAT = 800, PL_Z = 842, PL_DENSE = 938, PL_A_FREE = 800, PL_U_FREE = 48;

// top-2 bookkeeping of find_best_pitch as an ordering on (num, den, idx)
struct Cand {
  float num, den;
  int idx;
};
__device__ __forceinline__ bool cand_better(const Cand& a, const Cand& b) {
  const float l = a.num * b.den, r = b.num * a.den;
  return (l > r) || (!(r > l) && a.idx < b.idx);
}
__device__ __forceinline__ Cand wave_best(Cand c) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    Cand o;
    o.num = __shfl_xor(c.num, off, WAVE);
    o.den = __shfl_xor(c.den, off, WAVE);
    o.idx = __shfl_xor(c.idx, off, WAVE);
    if (cand_better(o, c)) c = o;
  }
  c.num = __shfl(c.num, 0, WAVE);
  c.den = __shfl(c.den, 0, WAVE);
  c.idx = __shfl(c.idx, 0, WAVE);
  return c;
}

// Inner products of x[0, 480) with NL lagged windows of the same LDS buffer, reduced over the wave:
// sxy[q] = sum_j x[j] y_q[j], and with SQ also syy[q] = sum_j y_q[j]^2.  xr / yr[q] already include the lane
// offset, so every read is one ds_read_b32 with an immediate offset (64 m); the NL (or 2 NL) accumulation chains
// and their DPP reductions are independent, which is what hides the LDS and cross-lane latencies -- the loops
// these replace did one dependent wave reduction per lag.
typedef float rn_f2 __attribute__((ext_vector_type(2)));
template <int NL, bool SQ>
__device__ __forceinline__ void lag_dots(const float* xr, const float* const (&yr)[NL], int lane, float (&sxy)[NL],
                                         float (&syy)[NL]) {
  float xv[8];
#pragma unroll
  for (int m = 0; m < 7; ++m) xv[m] = xr[WAVE * m];
#pragma unroll
  for (int q = 0; q < NL; ++q) { sxy[q] = 0.f; syy[q] = 0.f; }
#pragma unroll
  for (int q = 0; q < NL; ++q) {
#pragma unroll
    for (int m = 0; m < 7; ++m) {
      const float y = yr[q][WAVE * m];
      sxy[q] = fmaf(xv[m], y, sxy[q]);
      if (SQ) syy[q] = fmaf(y, y, syy[q]);
    }
  }
  if (lane < 32) {   // j = 448 + lane < 480
    xv[7] = xr[WAVE * 7];
#pragma unroll
    for (int q = 0; q < NL; ++q) {
      const float y = yr[q][WAVE * 7];
      sxy[q] = fmaf(xv[7], y, sxy[q]);
      if (SQ) syy[q] = fmaf(y, y, syy[q]);
    }
  }
  // all NL (2 NL) totals reduced together: rn_wave_sums.h (four sums for ten VALU instructions instead of 4 x 12)
  wave_sums<NL>(sxy);
  if (SQ) wave_sums<NL>(syy);
}

// =============================================================================================
// frame kernel: one wave per stream, loops over the T frames of the call
// =============================================================================================
// The fused frame: analysis, in-wave gain network, synthesis.
// DBG: the per-stage debug capture of the last frame (crispy_rn_debug_capture) and the per-frame taps (features, gains,
// pitch: parity tests) are a separate instantiation: the six
// `a.dbg && t == a.T - 1` tests kept two more kernel arguments live across the frame loop, where the scalar registers
// are already spilled into VGPR lanes (v_writelane / v_readlane are VALU instructions).
#ifndef RN_POISON_LDS
#define RN_POISON_LDS 0
#endif
[[maybe_unused]] __device__ __forceinline__ void rn_poison_lds(uint32_t* p, int words) {
  for (int i = threadIdx.x; i < words; i += WAVE) p[i] = 0x7fc0dead;
  __syncthreads();
}

// NW = 1: one wave per stream, the whole frame in sequence (4096 streams = 16 waves per CU: every pipe is busy).
// NW = 3: three waves per stream, one per STAGE, each a frame behind the one before it -- wave 0 the pitch analysis of
// frame k, wave 1 the spectra and features of frame k - 1, wave 2 gain network + synthesis of frame k - 2 -- with one
// workgroup barrier per tick.  A frame is a chain of ~26 k dependent quad-cycles whatever the stream count; with 1024
// streams there is one wave per SIMD and nothing to hide that chain behind, so the chain is cut in three (9.8 k / 6.5 k /
// 10.2 k quad-cycles) and the stream advances a frame per ~10 k instead.  Same code per stage, same arithmetic.
__device__ __forceinline__ float2* rn_lds_lp(RnLdsT<1>& L) { return L.s[0].Bb; }
__device__ __forceinline__ float2* rn_lds_sa(RnLdsT<1>& L) { return L.s[0].A; }
__device__ __forceinline__ float2* rn_lds_lp(RnLdsT<3>& L) { return L.pb; }
__device__ __forceinline__ float2* rn_lds_sa(RnLdsT<3>& L) { return L.pa; }

template <bool DBG, int NW>
__device__ __forceinline__ void rn_frame_body(const RnArgs& a) {
  __shared__ RnLdsT<NW> L;
  const int wave = NW == 1 ? 0 : (int)(threadIdx.x >> 6);
  const int lane0 = threadIdx.x & (WAVE - 1);
  const int b = blockIdx.x;
  if (b >= a.B) return;
  int lane = lane0;
  const RnTables* tab = a.tab;
  const float* xs = a.xhp + (long)b * a.xhp_stride;
  if constexpr (NW == 1) {
    // Different issue priorities per wave slot, so that the four waves of a SIMD drift out of phase without anyone
    // sleeping: -0.8 % per step over three alternating 30-step runs (by SIMD, or by slot + SIMD parity: no change).
    unsigned hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    const unsigned pr = hw & 3u;
    if (pr == 0) __builtin_amdgcn_s_setprio(0);
    else if (pr == 1) __builtin_amdgcn_s_setprio(1);
    else if (pr == 2) __builtin_amdgcn_s_setprio(2);
    else __builtin_amdgcn_s_setprio(3);
  }

#if RN_POISON_LDS
  // Checker build (`make variants` -> libcrispy_hip_poison.so, tests/test_gpu_rnnoise.py): LDS is not cleared between
  // workgroups, so a read of a word this workgroup has not written yet sees whatever the previous kernel on the CU left
  // there -- right or wrong depending on the history of the process.  Filling the allocation with NaNs first turns
  // every such read into a NaN in the output.
  rn_poison_lds(reinterpret_cast<uint32_t*>(&L), sizeof(L) / 4);
#endif

  // ---- load per-stream state (NW = 3: each stage's wave owns its part) ----
  const bool own_a = NW == 1 || wave == 0, own_b = NW == 1 || wave == 1, own_c = NW == 1 || wave == 2;
  if (own_b)
    for (int i = lane; i < 176; i += WAVE) L.ceps[i] = a.ceps[(long)b * 176 + i];
  if (own_c)
    for (int i = lane; i < 168; i += WAVE) L.rnn_state[i] = a.rnn[(long)b * 168 + i];
  // Overlap-add tail, samples (2n, 2n+1) for n = lane + 64 m, in registers across the frames of the launch
  float2 synth[4];
  float* synth_g = a.synth + (long)b * 480;
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    const int n = lane + WAVE * m;
    synth[m] = n < 240 ? *reinterpret_cast<const float2*>(synth_g + 2 * n) : make_float2(0.f, 0.f);
  }
  float lastg = lane < RN_NB ? a.lastg[(long)b * RN_NB + lane] : 0.f;
  BandEdges be;
  {
    be.piece = tab->band_piece[lane0];
  }
  int memid = a.memid[b];
  int last_period = a.last_period[b];
  float last_gain = a.last_gain[b];
  __syncthreads();

  constexpr int KB_FEAT = PL_FEAT;

  RN_PROF_DECL
  // NW = 3: tick k runs frame k on wave 0, frame k - 1 on wave 1, frame k - 2 on wave 2, then the workgroup barrier
  for (int tick = 0; tick < a.T + (NW - 1); ++tick) {
    const int t = tick - wave;
    if (NW == 1 || (t >= 0 && t < a.T)) {
    RnSlot& SL = L.s[NW == 1 ? 0 : (t & 1)];
    float* lp = reinterpret_cast<float*>(rn_lds_lp(L));   // 864 floats during the pitch phase
    float* Sa = reinterpret_cast<float*>(rn_lds_sa(L));   // pitch-phase scratch
    float* Rb = reinterpret_cast<float*>(SL.Bb);           // RNN vectors, band partials
    float* Xf = reinterpret_cast<float*>(SL.A);
    // Launder the lane id and the table / weight base pointers once per frame: every per-lane table
    // address is loop-invariant, and without this the compiler hoists ~250 of them out of the frame
    // loop into registers (429 VGPR+AGPR, one wave per SIMD).  Opaque values keep them per-frame.
    lane = lane0;
    asm volatile("" : "+v"(lane));
    RN_LANE_RANGE(0);
    // The laundered values are typed as address-space-1 pointers: laundering a generic pointer hides that it is
    // global, and every table / weight access then becomes a FLAT load (counts on lgkmcnt too, so it serialises
    // with the LDS traffic, and cannot use the SGPR-base + lane-offset form).
    typedef const __attribute__((address_space(1))) RnTables* GTabPtr;
    typedef const __attribute__((address_space(1))) uint32_t* GWordPtr;
    GTabPtr tabg = (GTabPtr)a.tab;
    GWordPtr wpg = (GWordPtr)a.wpack;
    asm volatile("" : "+s"(tabg), "+s"(wpg)::"memory");
    const RnTables* tabv = (const RnTables*)tabg;
    const uint32_t* wpraw = (const uint32_t*)wpg;
    // buffer resource over the weight pack (f16 matrices first, f32 biases behind them)
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(wpraw), 0, RnPack8::END * 4, 0x00020000);
    const RnTables* __restrict__ tab = tabv;
    const float2* __restrict__ w960 = tab->w960;
    const float* __restrict__ hw = tab->half_window;
    const float* __restrict__ wpf = reinterpret_cast<const float*>(wpraw);
    const float* xw = xs + (long)(t + 3) * RN_FRAME;  // [x_prev, x_cur]
    const float* pb = xs + (long)t * RN_FRAME + 672;  // 1728-sample pitch buffer ending at x_cur

    int pitch_index = 0;
    float pitch_gain = 0.f;
    bool silence = false;
    float vad_prob = 0.f;
    if (own_a) {      // ======== stage 1: pitch analysis ========
      // ---- 1. pitch: half-rate, LPC whitening (lp in Bb) ----
      {
        // all 14 x 2 window loads of a lane are requested before the first one is used (a `for (i = lane; ...)` loop
        // with the bound test in it compiled to load - wait - load - wait per trip: 27 exposed round trips per frame)
        float2 v[14];
        float x0[14];
  #pragma unroll
        for (int k = 0; k < 14; ++k) {
          const int i = min(lane + WAVE * k, 863);
          v[k] = *reinterpret_cast<const float2*>(pb + 2 * i);
          x0[k] = pb[2 * i - 1];                      // i = 0 reads the (valid) sample before the buffer, zeroed below
        }
  #pragma unroll
        for (int k = 0; k < 14; ++k) {
          const int i = lane + WAVE * k;
          const float xm = i > 0 ? x0[k] : 0.f;
          if (i < 864) lp[i] = .5f * (.5f * (xm + v[k].y) + v[k].x);
        }
      }
      rn_sync<NW>();
      // Each lane owns 14 consecutive half-rate samples (plus a 5-sample halo) in registers: the same window feeds
      // the autocorrelation partial sums (lags 0..4) and, once the LPC is known, the 5-tap FIR -- no second pass of
      // LDS reads.  Samples outside [0, 864) are zero, which is also what the reference's `i >= k` guard amounts to.
      float lpc2[5];
      const int base = lane * 14;
      float w[19];
  #pragma unroll
      for (int q = 0; q < 19; ++q) {
        const int idx = base - 5 + q;
        w[q] = (idx >= 0 && idx < 864) ? lp[idx] : 0.f;
      }
      {
        float ac[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
  #pragma unroll
        for (int q = 0; q < 14; ++q) {
  #pragma unroll
          for (int k = 0; k <= 4; ++k) ac[k] = fmaf(w[q + 5], w[q + 5 - k], ac[k]);
        }
        wave_sums<5>(ac);
        ac[0] *= 1.0001f;
  #pragma unroll
        for (int k = 1; k <= 4; ++k) ac[k] -= ac[k] * (.008f * k) * (.008f * k);
        float lpc[4] = {0.f, 0.f, 0.f, 0.f};
        float error = ac[0];
        if (ac[0] != 0.f) {
          bool done = false;
  #pragma unroll
          for (int i = 0; i < 4; ++i) {
            if (!done) {
              float rr = 0.f;
  #pragma unroll
              for (int j = 0; j < i; ++j) rr += lpc[j] * ac[i - j];
              rr += ac[i + 1];
              const float r = -rr / error;
              lpc[i] = r;
  #pragma unroll
              for (int j = 0; j < (i + 1) >> 1; ++j) {
                const float t1 = lpc[j], t2 = lpc[i - 1 - j];
                lpc[j] = t1 + r * t2;
                lpc[i - 1 - j] = t2 + r * t1;
              }
              error = error - r * r * error;
              if (error < .001f * ac[0]) done = true;
            }
          }
        }
        float tmp = 1.f;
  #pragma unroll
        for (int i = 0; i < 4; ++i) {
          tmp = .9f * tmp;
          lpc[i] = lpc[i] * tmp;
        }
        lpc2[0] = lpc[0] + .8f;
        lpc2[1] = lpc[1] + .8f * lpc[0];
        lpc2[2] = lpc[2] + .8f * lpc[1];
        lpc2[3] = lpc[3] + .8f * lpc[2];
        lpc2[4] = .8f * lpc[3];
      }
      {
        // 5-tap FIR in place from the register window
        float y[14];
  #pragma unroll
        for (int q = 0; q < 14; ++q) {
          float sum = w[q + 5];
          sum = fmaf(lpc2[0], w[q + 4], sum);
          sum = fmaf(lpc2[1], w[q + 3], sum);
          sum = fmaf(lpc2[2], w[q + 2], sum);
          sum = fmaf(lpc2[3], w[q + 1], sum);
          sum = fmaf(lpc2[4], w[q], sum);
          y[q] = sum;
        }
        rn_sync<NW>();
  #pragma unroll
        for (int q = 0; q < 14; ++q)
          if (base + q < 864) lp[base + q] = y[q];
        rn_sync<NW>();
      }
      STAMP(0)
      lane = lane0;
      asm volatile("" : "+v"(lane));
      RN_LANE_RANGE(4);

      // ---- 2. pitch_search: 4x-decimated coarse search over 147 lags (scratch in A) ----
      float* x4 = Sa;        // 240
      float* y4 = Sa + 240;  // 387 (+ guard to 392)
      float* pre = Sa;       // 392: exclusive prefix sums of y4^2, written after x4/y4 are dead (aliases them)
      {
        // reads of all trips first (clamped indices): as `for (j = lane; ...) x4[j] = lp[...]` loops every trip was an
        // LDS read - wait - write round trip, eleven in a row
        float vx[4], vy[7];
  #pragma unroll
        for (int m = 0; m < 4; ++m) vx[m] = lp[384 + 2 * min(lane + WAVE * m, 239)];
  #pragma unroll
        for (int m = 0; m < 7; ++m) vy[m] = lp[2 * min(lane + WAVE * m, 386)];
  #pragma unroll
        for (int m = 0; m < 4; ++m) {
          const int j = lane + WAVE * m;
          if (j < 240) x4[j] = vx[m];
        }
  #pragma unroll
        for (int m = 0; m < 7; ++m) {
          const int j = lane + WAVE * m;
          if (j < 392) y4[j] = j < 387 ? vy[m] : 0.f;
        }
      }
      rn_sync<NW>();
      int best0, best1;
      {
        // lane owns the three consecutive lags 3*lane + {0,1,2} (lanes 0..48): the y window slides through registers,
        // one new LDS read per step instead of three (lane stride 3 floats: conflict-free), x broadcast as float4
        float xc[3] = {0.f, 0.f, 0.f};
        {
          const float4* xv4 = reinterpret_cast<const float4*>(x4);
          const float* yp = y4 + 3 * min(lane, 48);      // reads reach y4[3*48 + 2 + 239] = y4[385] < 387
          float y0 = yp[0], y1 = yp[1];
  #pragma unroll 4
          for (int j4 = 0; j4 < 60; ++j4) {
            const float4 xv = xv4[j4];
            const int j = 4 * j4;
            const float y2 = yp[j + 2], y3 = yp[j + 3], y4v = yp[j + 4], y5 = yp[j + 5];
            xc[0] = fmaf(xv.x, y0, xc[0]);
            xc[1] = fmaf(xv.x, y1, xc[1]);
            xc[2] = fmaf(xv.x, y2, xc[2]);
            xc[0] = fmaf(xv.y, y1, xc[0]);
            xc[1] = fmaf(xv.y, y2, xc[1]);
            xc[2] = fmaf(xv.y, y3, xc[2]);
            xc[0] = fmaf(xv.z, y2, xc[0]);
            xc[1] = fmaf(xv.z, y3, xc[1]);
            xc[2] = fmaf(xv.z, y4v, xc[2]);
            xc[0] = fmaf(xv.w, y3, xc[0]);
            xc[1] = fmaf(xv.w, y4v, xc[1]);
            xc[2] = fmaf(xv.w, y5, xc[2]);
            y0 = y4v;
            y1 = y5;
          }
        }
        STAMP(1)
        // Syy of find_best_pitch: Syy(lag) = 1 + sum_{j=lag}^{lag+239} y4[j]^2 from a wave prefix sum.
        // (the reference's running update is the same quantity; its max(1, .) clamp only guards rounding)
        {
          float loc[7];
          float run = 0.f;
  #pragma unroll
          for (int q = 0; q < 7; ++q) {
            const int j = 7 * lane + q;
            const float v = j < 392 ? y4[j] : 0.f;
            loc[q] = run;          // exclusive
            run = fmaf(v, v, run);
          }
          float incl = run;
  #pragma unroll
          for (int off = 1; off < WAVE; off <<= 1) {
            const float o = __shfl_up(incl, off, WAVE);
            if (lane >= off) incl += o;
          }
          const float excl = incl - run;
          rn_sync<NW>();  // every lane has read its y4 values before pre overwrites the region
  #pragma unroll
          for (int q = 0; q < 7; ++q)
            if (7 * lane + q < 392) pre[7 * lane + q] = excl + loc[q];
        }
        rn_sync<NW>();
        // this lane's best two candidates, in lag order
        Cand c0 = {-1.f, 0.f, 1 << 20}, c1 = {-1.f, 0.f, 1 << 20};
        int nvalid = 0;
  #pragma unroll
        for (int rr = 0; rr < 3; ++rr) {
          const int lag = 3 * lane + rr;
          if (lag < 147 && xc[rr] > 0.f) {
            const float syy = fmaxf(1.f, 1.f + (pre[lag + 240] - pre[lag]));
            const float x16 = xc[rr] * 1e-12f;
            const Cand c = {x16 * x16, syy, lag};
            ++nvalid;
            if (cand_better(c, c0)) { c1 = c0; c0 = c; }
            else if (cand_better(c, c1)) c1 = c;
          }
        }
        const int total_valid = __popcll(__ballot(nvalid > 0)) == 0 ? 0 : (int)wave_sum((float)nvalid);
        const Cand w0 = wave_best(c0);
        const Cand mine = (c0.idx == w0.idx) ? c1 : c0;   // the winner's lane offers its runner-up
        const Cand w1 = wave_best(mine);
        if (total_valid == 0) { best0 = 0; best1 = 1; }
        else if (total_valid == 1) { best0 = w0.idx; best1 = 0; }
        else { best0 = w0.idx; best1 = w1.idx; }
      }
      rn_sync<NW>();
      STAMP(2)

      // ---- 3. fine search at half rate around the two coarse candidates ----
      float* fine = Sa;  // 294 (+2 guard) correlation values, zero where not evaluated
      for (int i = lane; i < 296; i += WAVE) fine[i] = 0.f;
      rn_sync<NW>();
      {
        const int ca = 2 * min(best0, best1), cb = 2 * max(best0, best1);
        Cand bestc = {-1.f, 0.f, 0};
        bool any = false;
        // all (up to) ten lags ca-2..ca+2, cb-2..cb+2 in one pass; skipped ones are computed at lag 0 and ignored
        int lagi[10];
        bool use[10];
        const float* yr[10];
  #pragma unroll
        for (int q = 0; q < 10; ++q) {
          const int pass = q / 5, d = q % 5 - 2;
          const int i = (pass == 0 ? ca : cb) + d;
          use[q] = !(i < 0 || i >= 294) && !(pass == 1 && abs(i - ca) <= 2);
          lagi[q] = use[q] ? i : 0;
          yr[q] = lp + lagi[q] + lane;
        }
        float sxy[10], syy[10];
        lag_dots<10, true>(lp + 384 + lane, yr, lane, sxy, syy);
  #pragma unroll
        for (int q = 0; q < 10; ++q) {
          if (use[q]) {
            const float syq = fmaxf(1.f, 1.f + syy[q]);
            const float xv = fmaxf(-1.f, sxy[q]);
            if (lane == 0) fine[lagi[q]] = xv;
            if (xv > 0.f) {
              const float x16 = xv * 1e-12f;
              const Cand c = {x16 * x16, syq, lagi[q]};
              if (!any || cand_better(c, bestc)) { bestc = c; any = true; }
            }
          }
        }
        rn_sync<NW>();
        const int bp = any ? bestc.idx : 0;
        int offset = 0;
        if (bp > 0 && bp < 293) {
          const float fa = fine[bp - 1], fb = fine[bp], fc = fine[bp + 1];
          if ((fc - fa) > .7f * (fb - fa)) offset = 1;
          else if ((fa - fc) > .7f * (fb - fc)) offset = -1;
        }
        pitch_index = 768 - (2 * bp - offset);
      }
      rn_sync<NW>();
      STAMP(3)
      if (DBG && a.dbg && t == a.T - 1) {
        float* D = a.dbg + (long)b * RN_DBG_FLOATS;
        for (int i = lane; i < 864; i += WAVE) D[984 + i] = lp[i];
        if (lane == 0) D[1848] = (float)pitch_index;
  #ifndef RN_PROFILE
        for (int i = lane; i < 480; i += WAVE) D[3824 + i] = xw[480 + i];
  #endif
      }

      // ---- 4. remove_doubling at half rate (maxperiod 384, minperiod 30, N 480) ----
      {
        const float* x = lp + 384;
        int T0 = pitch_index / 2;
        const int prev_period = last_period / 2;
        if (T0 >= 384) T0 = 383;
        int T = T0;
        float xx, xy;
        {
          const float* yr[2] = {x + lane, x - T0 + lane};
          float sa[2], sb[2];
          lag_dots<2, false>(x + lane, yr, lane, sa, sb);
          xx = sa[0];
          xy = sa[1];
        }
        // yy_lookup[m] = max(0, xx + sum_{q<=m} (x[-q]^2 - x[480-q]^2)) via a wave prefix sum
        float* yyl = Sa + 296;  // 385 entries
        {
          float loc[6];
          float run = 0.f;
  #pragma unroll
          for (int q = 0; q < 6; ++q) {
            const int m = 6 * lane + 1 + q;
            const float u = x[-m], v = x[480 - m];
            run += u * u - v * v;
            loc[q] = run;
          }
          float incl = run;
  #pragma unroll
          for (int off = 1; off < WAVE; off <<= 1) {
            const float o = __shfl_up(incl, off, WAVE);
            if (lane >= off) incl += o;
          }
          const float excl = incl - run;
  #pragma unroll
          for (int q = 0; q < 6; ++q) yyl[6 * lane + 1 + q] = fmaxf(0.f, xx + (excl + loc[q]));
          if (lane == 0) yyl[0] = xx;
        }
        rn_sync<NW>();
        float yy = yyl[T0];
        float best_xy = xy, best_yy = yy;
        const float g0 = xy / sqrtf(1.f + xx * yy);
        float g = g0;
        // The candidate lags depend on T0 only, so the (up to) 28 inner products are taken in chunks of four k
        // (eight lags) with independent accumulation chains and reductions; T1 falls with k, so the reference's
        // `break` at T1 < 30 is a prefix: a chunk is skipped when its first k is already out, and the sequential
        // threshold logic below runs on the stored sums.
        float xyk_[14], yyk_[14];
        int T1_[14];
  #pragma unroll
        for (int c = 0; c < 4; ++c) {
          const int k0 = 2 + 4 * c;
          if ((2 * T0 + k0) / (2 * k0) >= 30) {
            constexpr int NKC = 4;
            const float* yr[2 * NKC];
            int t1[NKC], t1b[NKC];
  #pragma unroll
            for (int q = 0; q < NKC; ++q) {
              const int k = k0 + q;
              t1[q] = 0; t1b[q] = 0;
              if (k <= 15) {
                const int T1 = (2 * T0 + k) / (2 * k);
                if (T1 >= 30) {
                  t1[q] = T1;
                  if (k == 2) t1b[q] = (T1 + T0 > 384) ? T0 : T0 + T1;
                  else t1b[q] = (2 * c_second_check[k] * T0 + k) / (2 * k);
                }
              }
              yr[2 * q] = x - t1[q] + lane;
              yr[2 * q + 1] = x - t1b[q] + lane;
            }
            float sa[2 * NKC], sb[2 * NKC];
            lag_dots<2 * NKC, false>(x + lane, yr, lane, sa, sb);
  #pragma unroll
            for (int q = 0; q < NKC; ++q) {
              if (k0 + q <= 15) {
                T1_[k0 + q - 2] = t1[q];
                xyk_[k0 + q - 2] = .5f * (sa[2 * q] + sa[2 * q + 1]);
                yyk_[k0 + q - 2] = .5f * (yyl[t1[q]] + yyl[t1b[q]]);
              }
            }
          } else {
  #pragma unroll
            for (int q = 0; q < 4; ++q)
              if (k0 + q <= 15) { T1_[k0 + q - 2] = 0; xyk_[k0 + q - 2] = 0.f; yyk_[k0 + q - 2] = 0.f; }
          }
        }
        // The threshold test of candidate k depends on k, T0, g0 and the previous frame only, never on the running best,
        // and the reference's loop keeps the *last* k that passes: so lane k - 2 evaluates candidate k (one square root
        // and one division per lane instead of fourteen of each in every lane, one after the other) and the highest
        // passing lane wins.  Same operations per candidate, bit-identical decisions.
        {
          int T1 = 0;
          float xyk = 0.f, yyk = 0.f;
  #pragma unroll
          for (int k = 2; k <= 15; ++k) {
            const bool mine = lane == k - 2;
            T1 = mine ? T1_[k - 2] : T1;
            xyk = mine ? xyk_[k - 2] : xyk;
            yyk = mine ? yyk_[k - 2] : yyk;
          }
          const int k = lane + 2;
          const float g1 = xyk / sqrtf(1.f + xx * yyk);
          float cont;
          if (abs(T1 - prev_period) <= 1) cont = last_gain;
          else if (abs(T1 - prev_period) <= 2 && 5 * k * k < T0) cont = .5f * last_gain;
          else cont = 0.f;
          float thresh = fmaxf(.3f, .7f * g0 - cont);
          if (T1 < 90) thresh = fmaxf(.4f, .85f * g0 - cont);
          else if (T1 < 60) thresh = fmaxf(.5f, .9f * g0 - cont);
          const unsigned long long pass = __ballot(lane < 14 && T1 >= 30 && g1 > thresh);
          if (pass != 0ull) {
            const int w = 63 - __builtin_clzll(pass);   // wave-uniform: the largest k that passes
            best_xy = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(xyk), w));
            best_yy = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(yyk), w));
            T = __builtin_amdgcn_readlane(T1, w);
            g = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(g1), w));
          }
        }
        best_xy = fmaxf(0.f, best_xy);
        float pgv = (best_yy <= best_xy) ? 1.f : best_xy / (best_yy + 1.f);
        float xc3[3];
        {
          const float* yr[3] = {x - (T - 1) + lane, x - T + lane, x - (T + 1) + lane};
          float sb[3];
          lag_dots<3, false>(x + lane, yr, lane, xc3, sb);
        }
        int offset = 0;
        if ((xc3[2] - xc3[0]) > .7f * (xc3[1] - xc3[0])) offset = 1;
        else if ((xc3[0] - xc3[2]) > .7f * (xc3[1] - xc3[2])) offset = -1;
        if (pgv > g) pgv = g;
        pitch_index = 2 * T + offset;
        if (pitch_index < 60) pitch_index = 60;
        pitch_gain = pgv;
        last_period = pitch_index;
        last_gain = pgv;
      }
      rn_sync<NW>();
      STAMP(4)
      if constexpr (NW > 1) {
        if (lane == 0) { L.pitch_index[t & 1] = pitch_index; L.pitch_gain[t & 1] = pitch_gain; }
      }
    }
    if (own_b) {      // ======== stage 2: spectra, band energies, features ========
      if constexpr (NW > 1) {
        pitch_index = __builtin_amdgcn_readfirstlane(L.pitch_index[t & 1]);
        pitch_gain = L.pitch_gain[t & 1];
      }
      lane = lane0;
      asm volatile("" : "+v"(lane));
      RN_LANE_RANGE(5);

      // ---- 5. frame_analysis: window, 960-point real FFT (in A), band energies (partials in Bb) ----
      // complex point n = (x[2n], x[2n+1]) times the window; points >= 240 sit in the mirrored half of the window
      fft480_from<NW>(SL.A, [&](int j, int r) {
        const int n = j + 120 * r;
        const float2 v = *reinterpret_cast<const float2*>(xw + 2 * n);
        const float w0 = r < 2 ? hw[2 * n] : hw[959 - 2 * n];
        const float w1 = r < 2 ? hw[2 * n + 1] : hw[958 - 2 * n];
        return make_float2(v.x * w0, v.y * w1);
      }, w960, lane);
      real_fwd_post<NW>(SL.A, w960, lane);
      STAMP(5)
      band_pairs<NW, false>(SL.A, nullptr, Rb, SL.Ex, nullptr, nullptr, tab, be, lane);
      STAMP(6)
      lane = lane0;
      asm volatile("" : "+v"(lane));
      RN_LANE_RANGE(6);
      if (DBG && a.dbg && t == a.T - 1) {
        float* D = a.dbg + (long)b * RN_DBG_FLOATS;
        for (int i = lane; i < 962; i += WAVE) D[0 + i] = Xf[i];
        if (lane < RN_NB) D[962 + lane] = SL.Ex[lane];
      }

      // ---- 6. pitch frame: window, FFT (in Bb), band energy / correlation (partials in U) ----
      {
        const float* pp = pb + (768 - pitch_index);
        fft480_from<NW>(SL.Bb, [&](int j, int r) {
          const int n = j + 120 * r;
          const float w0 = r < 2 ? hw[2 * n] : hw[959 - 2 * n];
          const float w1 = r < 2 ? hw[2 * n + 1] : hw[958 - 2 * n];
          return make_float2(pp[2 * n] * w0, pp[2 * n + 1] * w1);
        }, w960, lane);
      }
      real_fwd_post<NW>(SL.Bb, w960, lane);
      STAMP(7)
      // band energy of P, band correlation with X, and P parked in L2 from the same registers (read back by the comb
      // filter, which needs bins < 400 only); Bb becomes the RNN workspace
      band_pairs<NW, true>(SL.Bb, SL.A, SL.U, SL.Ep, SL.Exp, nullptr, tab, be, lane);
      if (DBG && a.dbg && t == a.T - 1) {
        float* D = a.dbg + (long)b * RN_DBG_FLOATS;
        const float* Pf = reinterpret_cast<const float*>(SL.Bb);
        for (int i = lane; i < 962; i += WAVE) D[1856 + i] = Pf[i];
      }
      rn_sync<NW>();
      STAMP(8)
      lane = lane0;
      asm volatile("" : "+v"(lane));
      RN_LANE_RANGE(7);

      // ---- 7. features (Appendix A.3 step 5) ----
      // this lane's column of the DCT table serves both transforms; issued first so that the L2 round trip
      // overlaps the band normalisation below
      float dctc[RN_NB];
      {
        const float* __restrict__ dcol = tab->dct + min(lane, RN_NB - 1);
  #pragma unroll
        for (int j = 0; j < RN_NB; ++j) dctc[j] = dcol[j * RN_NB];
      }
      if (lane < RN_NB) {
        SL.Exp[lane] = SL.Exp[lane] / sqrtf(.001f + SL.Ex[lane] * SL.Ep[lane]);
        SL.U[U_LY + lane] = log10f(1e-2f + SL.Ex[lane]);
      }
      rn_sync<NW>();
      if (DBG && a.dbg && t == a.T - 1 && lane < RN_NB) {
        float* D = a.dbg + (long)b * RN_DBG_FLOATS;
        D[2818 + lane] = SL.Ep[lane];
        D[2840 + lane] = SL.Exp[lane];
      }
      const float dct_norm = 0.30151134457776363f;  // sqrt(2/22)
      float E = 0.f;
      float se = 0.f, sl = 0.f;
      {
        // The band follower is a 22-step recurrence: every lane runs it on broadcast reads and feeds its own DCT
        // sums on the way -- of the followed log energies (22 coefficients) and of the band correlation (6), which
        // share this lane's table column.
        float logMax = -2.f, follow = -2.f;
  #pragma unroll
        for (int i = 0; i < RN_NB; ++i) {
          float ly = SL.U[U_LY + i];
          ly = fmaxf(logMax - 7.f, fmaxf(follow - 1.5f, ly));
          logMax = fmaxf(logMax, ly);
          follow = fmaxf(follow - 1.5f, ly);
          E += SL.Ex[i];
          sl = fmaf(ly, dctc[i], sl);
          se = fmaf(SL.Exp[i], dctc[i], se);
          if (i % 4 == 3) {   // bound the live ranges: all 66 broadcast reads hoisted to the top would spill
            asm volatile("" : "+v"(sl), "+v"(se), "+v"(E), "+v"(logMax), "+v"(follow));
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      }
      {
        if (lane < 6) {
          float v = se * dct_norm;
          if (lane == 0) v -= 1.3f;
          if (lane == 1) v -= 0.9f;
          Rb[KB_FEAT + 34 + lane] = v;
        }
        if (lane == 6) Rb[KB_FEAT + 40] = .01f * (float)(pitch_index - 300);
        silence = E < 0.04f;
        if (!silence && lane < RN_NB) {
          float v = sl * dct_norm;
          if (lane == 0) v -= 12.f;
          if (lane == 1) v -= 4.f;
          Rb[KB_FEAT + lane] = v;
          L.ceps[memid * RN_NB + lane] = v;
        }
      }
      rn_sync<NW>();
      if (silence) {
        if (lane < RN_NFEAT) Rb[KB_FEAT + lane] = 0.f;
        if (lane < RN_NB) SL.U[U_G + lane] = 0.f;
        rn_sync<NW>();
      } else {
        {
          const int m1 = (memid < 1) ? 8 + memid - 1 : memid - 1;
          const int m2 = (memid < 2) ? 8 + memid - 2 : memid - 2;
          if (lane < 6) {
            const float c0 = L.ceps[memid * RN_NB + lane];
            const float c1 = L.ceps[m1 * RN_NB + lane];
            const float c2 = L.ceps[m2 * RN_NB + lane];
            Rb[KB_FEAT + lane] = c0 + c1 + c2;
            Rb[KB_FEAT + RN_NB + lane] = c0 - c2;
            Rb[KB_FEAT + RN_NB + 6 + lane] = c0 - 2.f * c1 + c2;
          }
          memid = (memid + 1 == 8) ? 0 : memid + 1;
        }
        {
          // spectral variability: lane = 8*i + j holds ||ceps_i - ceps_j||^2; min over j within each group of 8
          // lanes and the sum over the 8 groups by DPP (every lane of a group ends up with the group minimum, so
          // the wave sum counts each minimum eight times)
          const int ci = lane >> 3, cj = lane & 7;
          float dist = 0.f;
  #pragma unroll
          for (int k = 0; k < RN_NB; ++k) {
            const float d = L.ceps[ci * RN_NB + k] - L.ceps[cj * RN_NB + k];
            dist = fmaf(d, d, dist);
          }
          float md = (ci == cj) ? 1e15f : dist;
          md = fminf(md, dpp_mov<0xB1>(md));     // quad_perm [1,0,3,2]
          md = fminf(md, dpp_mov<0x4E>(md));     // quad_perm [2,3,0,1]
          md = fminf(md, dpp_mov<0x141>(md));    // row_half_mirror: the other quad of the group of 8
          const float sv = wave_sum(md) * .125f;
          if (lane == 0) Rb[KB_FEAT + 41] = sv / 8.f - 2.1f;
        }
        rn_sync<NW>();
        STAMP(9)
      }
      if constexpr (NW > 1) {
        if (lane == 0) { SL.pitch_index = pitch_index; SL.pitch_gain = pitch_gain; SL.silence = silence ? 1 : 0; }
      }
    }
    if (own_c) {      // ======== stage 3: gain network, pitch filter, synthesis ========
      if constexpr (NW > 1) {
        pitch_index = __builtin_amdgcn_readfirstlane(SL.pitch_index);
        pitch_gain = SL.pitch_gain;
        silence = __builtin_amdgcn_readfirstlane(SL.silence) != 0;
      }
      if (!silence) {
        lane = lane0;
        asm volatile("" : "+v"(lane));   // (lane re-laundered: per-lane addresses of later stages are otherwise computed early / shared with earlier
        RN_LANE_RANGE(1);
        // stages and stay live across the gain network, which is where registers are scarcest)
        // ---- 8. RNN (vectors in Bb) ----
        const float S = 1.f / 256.f;
        float* feat = Rb + KB_FEAT;
        float* dense = Rb + PL_DENSE;
        float* zbuf = Rb + PL_Z;
        TansigTab tansig;
        tansig.load(tab->tansig, lane);
        // signed-digit int8 images of the layer inputs (in_img) and of the recurrent operand (st_img)
        signed char* in_img = reinterpret_cast<signed char*>(SL.U + PL_U_FREE);
        signed char* st_img = reinterpret_cast<signed char*>(Xf + PL_A_FREE);
        const int toff = (lane & 3) * RN_IMG8_LD;
        RnScale sc = image_i8<48>(in_img, lane, [&](int i) { return i < RN_NFEAT ? feat[i] : 0.f; });
        rn_sync<NW>();
        {
          const int row = min(lane, 23);
          const float acc = dot_i<rn_k16(42), 24>(wrs, RnPack8::ID_W, row, in_img, toff, sc.dn, wpf[RnPack::ID_B + row]);
          const float d = tansig_approx(S * acc, tansig);
          if (lane < 24) dense[lane] = d;
        }
        rn_sync<NW>();
        sc = image_i8<32>(in_img, lane, [&](int i) { return i < 24 ? dense[i] : 0.f; });
        gru_layer_i<NW, 24, 24>(wrs, RnPack8::VG_W, RnPack8::VG_R, wpf + RnPack::VG_B, in_img, sc.dn,
                            L.rnn_state, zbuf, st_img, toff, tansig, lane);
        {
          float acc = 0.f;
          if (lane == 0) {
            acc = wpf[RnPack::VO_B];
            acc = dot_vad(wrs, RnPack::VO_W, L.rnn_state, acc);
          }
          const float v = sigmoid_approx(S * acc, tansig);
          if (lane == 0) SL.U[U_VAD] = v;
        }
        sc = image_i8<96>(in_img, lane, [&](int i) {
          return i < 24 ? dense[i] : (i < 48 ? L.rnn_state[i - 24] : (i < 90 ? feat[i - 48] : 0.f)); });
        rn_sync<NW>();
        vad_prob = SL.U[U_VAD];
        STAMP(10)
        gru_layer_i<NW, 90, 48>(wrs, RnPack8::NG_W, RnPack8::NG_R, wpf + RnPack::NG_B, in_img, sc.dn,
                            L.rnn_state + 24, zbuf, st_img, toff, tansig, lane);
        STAMP(11)
        sc = image_i8<128>(in_img, lane, [&](int i) { return i < 72 ? L.rnn_state[i] : (i < 114 ? feat[i - 72] : 0.f); });
        rn_sync<NW>();
        gru_layer_i<NW, 114, 96>(wrs, RnPack8::DG_W, RnPack8::DG_R, wpf + RnPack::DG_B, in_img, sc.dn,
                             L.rnn_state + 72, zbuf, st_img, toff, tansig, lane);
        sc = image_i8<96>(st_img, lane, [&](int i) { return L.rnn_state[72 + i]; });
        rn_sync<NW>();
        {
          const int row = min(lane, RN_NB - 1);
          const float acc = dot_i<rn_k16(96), RN_NB>(wrs, RnPack8::DO_W, row, st_img, toff, sc.dn, wpf[RnPack::DO_B + row]);
          const float gv = sigmoid_approx(S * acc, tansig);
          if (lane < RN_NB) SL.U[U_G + lane] = gv;
        }
        rn_sync<NW>();
        STAMP(12)

        lane = lane0;
        asm volatile("" : "+v"(lane));
        RN_LANE_RANGE(2);
        // ---- 9. pitch_filter + gain application (Appendix A.3 step 7) ----
        // Pair layout: lane handles bins (2p, 2p+1), p = lane + 64 m -- one ds_read_b128 / one 16-byte global load
        // per pair at a 16-byte lane stride (conflict-free, coalesced), and a pair never straddles a band (edges are
        // multiples of 4 bins).  The new band energies are summed from the registers of the comb-filter pass.
        // (Keeping X in registers up to the final gains as well costs 16 VGPRs across two barriers and spills.)
        float2 fq[4];
        int bq[4];
        {
          float4 pq[4];
  #pragma unroll
          for (int m = 0; m < 4; ++m) {       // parked pitch spectrum and the per-bin tables: issued before r is ready
            const int pidx = min(lane + WAVE * m, 199);
            pq[m] = *reinterpret_cast<const float4*>(SL.Bb + 2 * pidx);
            fq[m] = *reinterpret_cast<const float2*>(tab->bin_frac + 2 * pidx);
            bq[m] = tab->bin_band[2 * pidx];
          }
          if (lane < RN_NB) {
            const float ex = SL.Exp[lane], gg = SL.U[U_G + lane];
            float r;
            if (ex > gg) r = 1.f;
            else r = (ex * ex) * (1.f - gg * gg) / (.001f + (gg * gg) * (1.f - ex * ex));
            r = sqrtf(fminf(1.f, fmaxf(0.f, r)));
            r *= sqrtf(SL.Ex[lane] / (1e-8f + SL.Ep[lane]));
            SL.U[U_R + lane] = r;
          }
          rn_sync<NW>();
          float* part = Xf + PL_A_FREE;
          float* part_hi = SL.U + PL_U_FREE;
  #pragma unroll
          for (int m = 0; m < 4; ++m) {
            const int pidx = lane + WAVE * m;
            if (pidx < 200) {
              const float r0 = SL.U[U_R + bq[m]], r1 = SL.U[U_R + bq[m] + 1];
              const float rf0 = (1.f - fq[m].x) * r0 + fq[m].x * r1;
              const float rf1 = (1.f - fq[m].y) * r0 + fq[m].y * r1;
              float4 x = *reinterpret_cast<const float4*>(SL.A + 2 * pidx);
              x.x = fmaf(rf0, pq[m].x, x.x);
              x.y = fmaf(rf0, pq[m].y, x.y);
              x.z = fmaf(rf1, pq[m].z, x.z);
              x.w = fmaf(rf1, pq[m].w, x.w);
              *reinterpret_cast<float4*>(SL.A + 2 * pidx) = x;
              float e0 = x.x * x.x; e0 += x.y * x.y;
              float e1 = x.z * x.z; e1 += x.w * x.w;
              float lo = (1.f - fq[m].x) * e0 + (1.f - fq[m].y) * e1;
              float hi = fq[m].x * e0 + fq[m].y * e1;
              lo = dpp_add<0xB1>(lo);        // the other pair of this 4-bin chunk sits in the neighbouring lane
              hi = dpp_add<0xB1>(hi);
              if ((lane & 1) == 0) { part[pidx >> 1] = lo; part_hi[pidx >> 1] = hi; }
            }
          }
        }
        rn_sync<NW>();
        float sum;
        {                          // new band energies (Ep is dead): every lane takes part (band_sum splits a band over two lanes)
          const float* part = Xf + PL_A_FREE;
          const float* part_hi = SL.U + PL_U_FREE;
          sum = band_sum(part, part_hi, be, lane);
        }
        if (lane < RN_NB) {        // the renormalisation and the smoothed gains
          SL.U[U_R + lane] = sqrtf(SL.Ex[lane] / (1e-8f + sum));  // norm
          const float gg = fmaxf(SL.U[U_G + lane], .6f * lastg);
          SL.U[U_G + lane] = gg;
          lastg = gg;
        }
        rn_sync<NW>();
        // band values and spectrum pairs of all four trips first (bq / indices are always valid: clamped where they
        // were made), so that the trips do not each wait for their own LDS reads
        float nq[4][2], gq[4][2];
        float4 xq[4];
  #pragma unroll
        for (int m = 0; m < 4; ++m) {
          nq[m][0] = SL.U[U_R + bq[m]]; nq[m][1] = SL.U[U_R + bq[m] + 1];
          gq[m][0] = SL.U[U_G + bq[m]]; gq[m][1] = SL.U[U_G + bq[m] + 1];
          xq[m] = *reinterpret_cast<const float4*>(SL.A + 2 * min(lane + WAVE * m, 199));
        }
  #pragma unroll
        for (int m = 0; m < 4; ++m) {
          const int pidx = lane + WAVE * m;
          if (pidx < 200) {
            const float n0 = nq[m][0], n1 = nq[m][1];
            const float g0 = gq[m][0], g1 = gq[m][1];
            const float nf0 = (1.f - fq[m].x) * n0 + fq[m].x * n1, nf1 = (1.f - fq[m].y) * n0 + fq[m].y * n1;
            const float gf0 = (1.f - fq[m].x) * g0 + fq[m].x * g1, gf1 = (1.f - fq[m].y) * g0 + fq[m].y * g1;
            float4 x = xq[m];
            x.x *= nf0; x.y *= nf0; x.z *= nf1; x.w *= nf1;
            x.x *= gf0; x.y *= gf0; x.z *= gf1; x.w *= gf1;
            *reinterpret_cast<float4*>(SL.A + 2 * pidx) = x;
          }
        }
        for (int i = 400 + lane; i < RN_NFREQ; i += WAVE) SL.A[i] = make_float2(0.f, 0.f);   // above 20 kHz
        rn_sync<NW>();
      }
      STAMP(13)

      // ---- taps / debug ----
      if (DBG && a.taps) {
        float* tp = a.taps + ((long)t * a.B + b) * RN_TAPS;
        if (lane < RN_NFEAT) tp[lane] = Rb[KB_FEAT + lane];
        if (lane < RN_NB) tp[42 + lane] = SL.U[U_G + lane];
        if (lane == 0) {
          tp[64] = (float)pitch_index;
          tp[65] = pitch_gain;
          tp[66] = vad_prob;
          tp[67] = silence ? 1.f : 0.f;
          tp[68] = tp[69] = tp[70] = tp[71] = 0.f;
        }
      }
      if (a.vad && lane == 0) a.vad[(long)t * a.B + b] = vad_prob;
      if (DBG && a.dbg && t == a.T - 1) {
        float* D = a.dbg + (long)b * RN_DBG_FLOATS;
        for (int i = lane; i < 962; i += WAVE) D[2862 + i] = Xf[i];
      }
      rn_sync<NW>();

      lane = lane0;
      asm volatile("" : "+v"(lane));
      RN_LANE_RANGE(3);
      // ---- 10. frame_synthesis: inverse FFT, window, overlap-add ----
      real_inv_pre<NW>(SL.A, w960, lane);
      fft480<NW>(SL.A, w960, lane);
      STAMP(14)
      {
        float* o = a.out + (long)t * a.stride_t + (long)b * a.stride_b;
        float2 hwa[4], hwb[4];                  // window values of all four trips, requested together
  #pragma unroll
        for (int m = 0; m < 4; ++m) {
          const int n = min(lane + WAVE * m, 239);
          hwa[m] = *reinterpret_cast<const float2*>(hw + 2 * n);          // hw[i0], hw[i1]
          hwb[m] = *reinterpret_cast<const float2*>(hw + 478 - 2 * n);    // hw[479 - i1], hw[479 - i0]
        }
  #pragma unroll
        for (int m = 0; m < 4; ++m) {
          const int n = lane + WAVE * m;
          if (n < 240) {
            const float2 z = SL.A[n];
            const float2 z2 = SL.A[240 + n];
            const int i0 = 2 * n;
            float2 ov;
            ov.x = fmaf(z.x, hwa[m].x, synth[m].x);
            ov.y = fmaf(-z.y, hwa[m].y, synth[m].y);
            // the output is never read back here: non-temporal stores keep it from allocating in L2 next to the history
            // window and the parked spectrum (-1.1 KB of fetches per stream-frame, time unchanged)
            if (a.out_s16) {
              // int16 transport (RnArgs::out_s16): the adapter's / 32768 and clamp (audio.rs:270-273), the WAV writer's
              // x 32767 truncated toward zero (recording.rs:109-110); two samples = one dword of the caller's int16 frame
              const float q0 = truncf(fminf(fmaxf(ov.x * (1.f / 32768.f), -1.f), 1.f) * 32767.f);
              const float q1 = truncf(fminf(fmaxf(ov.y * (1.f / 32768.f), -1.f), 1.f) * 32767.f);
              const unsigned pk = ((unsigned)(int)q0 & 0xffffu) | ((unsigned)(int)q1 << 16);
              __builtin_nontemporal_store(pk, reinterpret_cast<unsigned*>(reinterpret_cast<int16_t*>(a.out) + (long)t * a.stride_t + (long)b * a.stride_b + i0));
            } else {
              __builtin_nontemporal_store(ov.x, o + i0); __builtin_nontemporal_store(ov.y, o + i0 + 1);
            }
            synth[m].x = z2.x * hwb[m].y;
            synth[m].y = -z2.y * hwb[m].x;
          }
        }
      }
      rn_sync<NW>();
      STAMP(15)
    }
    }      // this wave has a frame in this tick
    if constexpr (NW > 1) __syncthreads();
  }

  // ---- store per-stream state (NW = 3: by the wave that owns it) ----
  if (own_c) {
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const int n = lane + WAVE * m;
      if (n < 240) *reinterpret_cast<float2*>(synth_g + 2 * n) = synth[m];
    }
    for (int i = lane; i < 168; i += WAVE) a.rnn[(long)b * 168 + i] = L.rnn_state[i];
    if (lane < RN_NB) a.lastg[(long)b * RN_NB + lane] = lastg;
  }
  if (own_b) {
    for (int i = lane; i < 176; i += WAVE) a.ceps[(long)b * 176 + i] = L.ceps[i];
    if (lane == 0) a.memid[b] = memid;
  }
  if (own_a && lane == 0) {
    a.last_period[b] = last_period;
    a.last_gain[b] = last_gain;
  }
}

template <bool DBG>
__global__ __launch_bounds__(WAVE) __attribute__((amdgpu_waves_per_eu(4))) RN_VGPR_CAP void rn_frame_kernel(RnArgs a) {
  rn_frame_body<DBG, 1>(a);
}
// Four waves per SIMD (<= 128 registers, what the one-wave form lives in too): five of these workgroups per CU, so that
// 1024 streams are resident at once on 256 CUs with room to spare (six fit by their LDS).
template <bool DBG>
__global__ __launch_bounds__(3 * WAVE) __attribute__((amdgpu_waves_per_eu(4))) void rn_frame3_kernel(RnArgs a) {
  rn_frame_body<DBG, 3>(a);
}

// Stage entry point for parity tests: the frame kernel's own activation code (TansigTab in four registers per lane,
// ds_bpermute lookups, the +-8 clamps) applied to n arbitrary arguments, so that every table cell and both clamps can
// be bit-compared with the oracle's tansig_approx / sigmoid_approx.  All 64 lanes stay active (bpermute reads 0 from
// masked lanes): the index is clamped, the store predicated.
__global__ __launch_bounds__(WAVE) void rn_tansig_kernel(const RnTables* tab, const float* x, float* y, long n, int sigmoid) {
  const int lane = threadIdx.x;
  TansigTab tansig;
  tansig.load(tab->tansig, lane);
  for (long base = (long)blockIdx.x * WAVE; base < n; base += (long)gridDim.x * WAVE) {
    const long i = base + lane;
    const float v = x[i < n ? i : n - 1];
    const float r = sigmoid ? sigmoid_approx(v, tansig) : tansig_approx(v, tansig);
    if (i < n) y[i] = r;
  }
}

}  // namespace

hipError_t rn_launch_frames(const RnArgs& a, hipStream_t s, int waves_per_stream) {
  if (waves_per_stream == 3) {
    if (a.dbg || a.taps) hipLaunchKernelGGL((rn_frame3_kernel<true>), dim3(a.B), dim3(3 * WAVE), 0, s, a);
    else hipLaunchKernelGGL((rn_frame3_kernel<false>), dim3(a.B), dim3(3 * WAVE), 0, s, a);
    return hipGetLastError();
  }
  if (a.dbg || a.taps) hipLaunchKernelGGL((rn_frame_kernel<true>), dim3(a.B), dim3(WAVE), 0, s, a);
  else hipLaunchKernelGGL((rn_frame_kernel<false>), dim3(a.B), dim3(WAVE), 0, s, a);
  return hipGetLastError();
}
hipError_t rn_launch_tansig(const RnTables* tab, const float* x, float* y, long n, int sigmoid, hipStream_t s) {
  const long blocks = (n + WAVE - 1) / WAVE;
  hipLaunchKernelGGL(rn_tansig_kernel, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(WAVE), 0, s, tab, x, y, n, sigmoid);
  return hipGetLastError();
}
}  // namespace crispy
