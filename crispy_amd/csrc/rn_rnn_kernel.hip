// rn_rnn_kernel.hip -- the RNNoise gain network (dense + 3 GRUs + 2 output layers, SURVEY.md Appendix
// A.3 step 6) batched over streams on the CDNA4 matrix cores.
//
// One workgroup = 16 streams (the M dimension of v_mfma_f32_16x16x32_bf16), 8 waves, persistent over the
// T frames of a launch.  Time is strictly serial (GRU state), streams are the batch.
//   * weights: int8 values are exact in bf16; every (layer phase, 16-column tile, 32-deep k step) is one
//     B fragment of 8 bf16 per lane, loaded ONCE per launch and kept in registers (41 fragments per wave);
//   * activations: f32 values are split into three bf16 terms (hi + lo + lo2 = 24 mantissa bits, exact), so
//     three MFMAs per fragment reproduce f32 products exactly and accumulate in f32 -- same accuracy class as
//     the f32 FMA chain of the reference, ~8x fewer issue slots;
//   * the split is done once by the lane that produces a value, straight into the per-layer concatenated
//     input images in LDS (row stride padded by 16 B: conflict-free ds_read_b128 A-fragment reads).
// Silent frames (E < 0.04 upstream) leave the stream's state untouched, as the reference does.
#include "rn_common.h"

namespace crispy {
namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// padded row lengths (bf16 elements) of the LDS input images
constexpr int LD_XD = 64 + 8, LD_XV = 64 + 8, LD_XVO = 32 + 8, LD_XN = 160 + 8, LD_XDN = 224 + 8, LD_XO = 96 + 8;

struct alignas(16) RnnLds {
  __bf16 XD[3][16][LD_XD];     // dense in : feat(42)
  __bf16 XV[3][16][LD_XV];     // vad GRU  : dense(24) | vad_state or h*r (24)
  __bf16 XVO[3][16][LD_XVO];   // vad out  : vad_state(24)
  __bf16 XN[3][16][LD_XN];     // noise GRU: dense(24) | vad_state(24) | feat(42) | noise_state or h*r (48)
  __bf16 XDN[3][16][LD_XDN];   // den GRU  : vad_state(24) | noise_state(48) | feat(42) | den_state or h*r (96)
  __bf16 XO[3][16][LD_XO];     // gains out: den_state(96)
  float vad_state[16][24], noise_state[16][48], den_state[16][96];
  float zbuf[16][96];
  float lastg[16][24];
  float tansig[208];
  int silent[16];
};

__device__ __forceinline__ float tansig_lds(float x, const float* table) {
#pragma clang fp contract(off)
  if (!(x < 8.f)) return 1.f;
  if (!(x > -8.f)) return -1.f;
  float sign = 1.f;
  if (x < 0.f) { x = -x; sign = -1.f; }
  // every product and sum rounded on its own, in the reference's order (see tansig_approx in rn_kernels.hip)
  const int i = (int)floorf(.5f + 25.f * x);
  x = x - .04f * (float)i;
  float y = table[i];
  const float dy = 1.f - y * y;
  y = y + (x * dy) * (1.f - y * x);
  return sign * y;
}
__device__ __forceinline__ float sigmoid_lds(float x, const float* table) { return .5f + .5f * tansig_lds(.5f * x, table); }

// write v as hi + lo + lo2 into column p of row `row` of a three-plane bf16 image with row length LD
template <int LD>
__device__ __forceinline__ void put3(__bf16 (*X)[16][LD], int row, int p, float v) {
  const __bf16 hi = (__bf16)v;
  const float r1 = v - (float)hi;
  const __bf16 lo = (__bf16)r1;
  const float r2 = r1 - (float)lo;
  X[0][row][p] = hi;
  X[1][row][p] = lo;
  X[2][row][p] = (__bf16)r2;
}

// acc += X[rows 0..15][ks*32 .. +32] . Wfrag  for the three bf16 planes
template <int LD, int KS>
__device__ __forceinline__ f32x4 tile_mma(const __bf16 (*X)[16][LD], const bf16x8 (&w)[KS], int lane, f32x4 acc) {
  const int row = lane & 15, kg = lane >> 4;
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
    for (int s = 0; s < 3; ++s) {
      const bf16x8 a = *reinterpret_cast<const bf16x8*>(&X[s][row][ks * 32 + 8 * kg]);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, w[ks], acc, 0, 0, 0);
    }
  }
  return acc;
}

template <int KS>
__device__ __forceinline__ void load_frags(bf16x8 (&w)[KS], const bf16x8* __restrict__ base, int first_frag, int tile,
                                           int n_tiles, int lane) {
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    bf16x8 z;
#pragma unroll
    for (int j = 0; j < 8; ++j) z[j] = (__bf16)0.f;
    w[ks] = tile < n_tiles ? base[(long)(first_frag + tile * KS + ks) * 64 + lane] : z;
  }
}

__global__ __launch_bounds__(512) void rn_rnn_kernel(RnnArgs a) {
  __shared__ RnnLds L;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int b0 = blockIdx.x * 16;
  const int col = lane & 15, rg = lane >> 4;   // accumulator: neuron column, row group (rows 4*rg .. 4*rg+3)
  const bf16x8* __restrict__ frags = reinterpret_cast<const bf16x8*>(a.frags);
  const float* __restrict__ bias = a.bias;
  const float S = 1.f / 256.f;

  // ---- weight fragments of this wave, resident for the whole launch ----
  bf16x8 w_dense[2], w_vzr[2], w_vh[2], w_vo[1], w_nzr[5], w_nh[5], w_dzr0[7], w_dzr1[7], w_dh[7], w_out[3];
  load_frags<2>(w_dense, frags, RnnPack::F_DENSE, wave, 2, lane);
  load_frags<2>(w_vzr, frags, RnnPack::F_VZR, wave, 3, lane);
  load_frags<2>(w_vh, frags, RnnPack::F_VH, wave, 2, lane);
  load_frags<1>(w_vo, frags, RnnPack::F_VO, wave == 7 ? 0 : 99, 1, lane);   // vad output rides on wave 7
  load_frags<5>(w_nzr, frags, RnnPack::F_NZR, wave, 6, lane);
  load_frags<5>(w_nh, frags, RnnPack::F_NH, wave, 3, lane);
  load_frags<7>(w_dzr0, frags, RnnPack::F_DZR, wave, 12, lane);
  load_frags<7>(w_dzr1, frags, RnnPack::F_DZR, wave + 8, 12, lane);
  load_frags<7>(w_dh, frags, RnnPack::F_DH, wave, 6, lane);
  load_frags<3>(w_out, frags, RnnPack::F_OUT, wave, 2, lane);

#if defined(RN_POISON_LDS) && RN_POISON_LDS
  // checker build (see rn_kernels.hip): NaNs in every word of LDS this workgroup has not written itself
  for (int i = tid; i < (int)(sizeof(L) / 4); i += 512) reinterpret_cast<uint32_t*>(&L)[i] = 0x7fc0dead;
  __syncthreads();
#endif
  // ---- state in, LDS images zeroed ----
  {
    __bf16* z = &L.XD[0][0][0];
    const int n_bf = (int)((reinterpret_cast<char*>(&L.vad_state[0][0]) - reinterpret_cast<char*>(z)) / 2);
    for (int i = tid; i < n_bf; i += 512) z[i] = (__bf16)0.f;
    for (int i = tid; i < 208; i += 512) L.tansig[i] = a.tansig[i];
  }
  __syncthreads();
  for (int i = tid; i < 16 * 168; i += 512) {
    const int row = i / 168, k = i % 168;
    const int b = b0 + row;
    const float v = b < a.B ? a.rnn[(long)b * 168 + k] : 0.f;
    if (k < 24) {
      L.vad_state[row][k] = v;
      put3<LD_XV>(L.XV, row, 24 + k, v); put3<LD_XVO>(L.XVO, row, k, v);
      put3<LD_XN>(L.XN, row, 24 + k, v); put3<LD_XDN>(L.XDN, row, k, v);
    } else if (k < 72) {
      L.noise_state[row][k - 24] = v;
      put3<LD_XN>(L.XN, row, 90 + k - 24, v); put3<LD_XDN>(L.XDN, row, 24 + k - 24, v);
    } else {
      L.den_state[row][k - 72] = v;
      put3<LD_XDN>(L.XDN, row, 114 + k - 72, v); put3<LD_XO>(L.XO, row, k - 72, v);
    }
  }
  for (int i = tid; i < 16 * 24; i += 512) {
    const int row = i / 24, k = i % 24, b = b0 + row;
    L.lastg[row][k] = (b < a.B && k < RN_NB) ? a.lastg[(long)b * RN_NB + k] : 0.f;
  }
  __syncthreads();

  for (int t = 0; t < a.T; ++t) {
    // ---- features of this frame -> dense / noise / denoise input images ----
    for (int i = tid; i < 16 * 42; i += 512) {
      const int row = i / 42, k = i % 42, b = b0 + row;
      const float f = b < a.B ? a.feat[((long)t * a.B + b) * RNN_FEAT_LD + k] : 0.f;
      put3<LD_XD>(L.XD, row, k, f);
      put3<LD_XN>(L.XN, row, 48 + k, f);
      put3<LD_XDN>(L.XDN, row, 72 + k, f);
    }
    if (tid < 16) L.silent[tid] = (b0 + tid < a.B) ? (int)a.silent[(long)t * a.B + b0 + tid] : 1;
    __syncthreads();

    // ---- input dense 42 -> 24, tanh ----
    if (wave < 2) {
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      acc = tile_mma<LD_XD, 2>(L.XD, w_dense, lane, acc);
      const int i = wave * 16 + col;
      if (i < 24) {
        const float bi = bias[RnnPack::B_DENSE + i];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = 4 * rg + r;
          const float v = tansig_lds(S * (acc[r] + bi), L.tansig);
          put3<LD_XV>(L.XV, row, i, v);
          put3<LD_XN>(L.XN, row, i, v);
        }
      }
    }
    __syncthreads();

    // ---- vad GRU (24): z, r ----
    {
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      if (wave < 3) acc = tile_mma<LD_XV, 2>(L.XV, w_vzr, lane, acc);
      __syncthreads();   // every tile has read the state columns before h*r overwrites them
      const int c = wave * 16 + col;
      if (wave < 3 && c < 48) {
        const float bi = bias[RnnPack::B_VG + c];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = 4 * rg + r;
          const float sg = sigmoid_lds(S * (acc[r] + bi), L.tansig);
          if (c < 24) L.zbuf[row][c] = sg;
          else put3<LD_XV>(L.XV, row, c, L.vad_state[row][c - 24] * sg);
        }
      }
    }
    __syncthreads();
    // ---- vad GRU: candidate + state update ----
    {
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      if (wave < 2) acc = tile_mma<LD_XV, 2>(L.XV, w_vh, lane, acc);
      __syncthreads();
      const int i = wave * 16 + col;
      if (wave < 2 && i < 24) {
        const float bi = bias[RnnPack::B_VG + 48 + i];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = 4 * rg + r;
          float cnd = S * (acc[r] + bi);
          cnd = cnd < 0.f ? 0.f : cnd;
          const float z = L.zbuf[row][i], ho = L.vad_state[row][i];
          const float hn = L.silent[row] ? ho : z * ho + (1.f - z) * cnd;
          L.vad_state[row][i] = hn;
          put3<LD_XV>(L.XV, row, 24 + i, hn); put3<LD_XVO>(L.XVO, row, i, hn);
          put3<LD_XN>(L.XN, row, 24 + i, hn); put3<LD_XDN>(L.XDN, row, i, hn);
        }
      }
    }
    __syncthreads();

    // ---- noise GRU (48): z, r  (+ vad output on wave 7) ----
    {
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      if (wave < 6) acc = tile_mma<LD_XN, 5>(L.XN, w_nzr, lane, acc);
      if (wave == 7) {
        f32x4 av = {0.f, 0.f, 0.f, 0.f};
        av = tile_mma<LD_XVO, 1>(L.XVO, w_vo, lane, av);
        if (col == 0 && a.vad) {
          const float bi = bias[RnnPack::B_VO];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int row = 4 * rg + r, b = b0 + row;
            if (b < a.B) a.vad[(long)t * a.B + b] = L.silent[row] ? 0.f : sigmoid_lds(S * (av[r] + bi), L.tansig);
          }
        }
      }
      __syncthreads();
      const int c = wave * 16 + col;
      if (wave < 6) {
        const float bi = bias[RnnPack::B_NG + c];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = 4 * rg + r;
          const float sg = sigmoid_lds(S * (acc[r] + bi), L.tansig);
          if (c < 48) L.zbuf[row][c] = sg;
          else put3<LD_XN>(L.XN, row, 90 + c - 48, L.noise_state[row][c - 48] * sg);
        }
      }
    }
    __syncthreads();
    {
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      if (wave < 3) acc = tile_mma<LD_XN, 5>(L.XN, w_nh, lane, acc);
      __syncthreads();
      const int i = wave * 16 + col;
      if (wave < 3) {
        const float bi = bias[RnnPack::B_NG + 96 + i];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = 4 * rg + r;
          float cnd = S * (acc[r] + bi);
          cnd = cnd < 0.f ? 0.f : cnd;
          const float z = L.zbuf[row][i], ho = L.noise_state[row][i];
          const float hn = L.silent[row] ? ho : z * ho + (1.f - z) * cnd;
          L.noise_state[row][i] = hn;
          put3<LD_XN>(L.XN, row, 90 + i, hn); put3<LD_XDN>(L.XDN, row, 24 + i, hn);
        }
      }
    }
    __syncthreads();

    // ---- denoise GRU (96): z, r on 12 tiles (waves 0-3 take two) ----
    {
      f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
      acc0 = tile_mma<LD_XDN, 7>(L.XDN, w_dzr0, lane, acc0);
      if (wave < 4) acc1 = tile_mma<LD_XDN, 7>(L.XDN, w_dzr1, lane, acc1);
      __syncthreads();
#pragma unroll
      for (int slot = 0; slot < 2; ++slot) {
        if (slot == 1 && wave >= 4) break;
        const int c = (wave + 8 * slot) * 16 + col;
        const float bi = bias[RnnPack::B_DG + c];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = 4 * rg + r;
          const float sg = sigmoid_lds(S * ((slot ? acc1[r] : acc0[r]) + bi), L.tansig);
          if (c < 96) L.zbuf[row][c] = sg;
          else put3<LD_XDN>(L.XDN, row, 114 + c - 96, L.den_state[row][c - 96] * sg);
        }
      }
    }
    __syncthreads();
    {
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      if (wave < 6) acc = tile_mma<LD_XDN, 7>(L.XDN, w_dh, lane, acc);
      __syncthreads();
      const int i = wave * 16 + col;
      if (wave < 6) {
        const float bi = bias[RnnPack::B_DG + 192 + i];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = 4 * rg + r;
          float cnd = S * (acc[r] + bi);
          cnd = cnd < 0.f ? 0.f : cnd;
          const float z = L.zbuf[row][i], ho = L.den_state[row][i];
          const float hn = L.silent[row] ? ho : z * ho + (1.f - z) * cnd;
          L.den_state[row][i] = hn;
          put3<LD_XDN>(L.XDN, row, 114 + i, hn); put3<LD_XO>(L.XO, row, i, hn);
        }
      }
    }
    __syncthreads();

    // ---- gains 96 -> 22, sigmoid; smoothing g = max(g, 0.6 lastg) ----
    if (wave < 2) {
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      acc = tile_mma<LD_XO, 3>(L.XO, w_out, lane, acc);
      const int i = wave * 16 + col;
      if (i < RN_NB) {
        const float bi = bias[RnnPack::B_OUT + i];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = 4 * rg + r, b = b0 + row;
          float g = 0.f, gs = 0.f;
          if (!L.silent[row]) {
            g = sigmoid_lds(S * (acc[r] + bi), L.tansig);
            gs = fmaxf(g, .6f * L.lastg[row][i]);
            L.lastg[row][i] = gs;
          }
          if (b < a.B) {
            a.g_raw[((long)t * a.B + b) * RNN_GAIN_LD + i] = g;
            a.g_smooth[((long)t * a.B + b) * RNN_GAIN_LD + i] = gs;
          }
        }
      }
    }
    __syncthreads();
  }

  // ---- state out ----
  for (int i = tid; i < 16 * 168; i += 512) {
    const int row = i / 168, k = i % 168, b = b0 + row;
    if (b < a.B) {
      const float v = k < 24 ? L.vad_state[row][k] : (k < 72 ? L.noise_state[row][k - 24] : L.den_state[row][k - 72]);
      a.rnn[(long)b * 168 + k] = v;
    }
  }
  for (int i = tid; i < 16 * RN_NB; i += 512) {
    const int row = i / RN_NB, k = i % RN_NB, b = b0 + row;
    if (b < a.B) a.lastg[(long)b * RN_NB + k] = L.lastg[row][k];
  }
}

}  // namespace

hipError_t rn_launch_rnn(const RnnArgs& a, hipStream_t s) {
  hipLaunchKernelGGL(rn_rnn_kernel, dim3((a.B + 15) / 16), dim3(512), 0, s, a);
  return hipGetLastError();
}

}  // namespace crispy
