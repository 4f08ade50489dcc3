// whisper_dec_f16.hip -- decode-step kernels of precision mode 1 (f16 operands, f32 accumulation: ggml's numerics).
//
// Vocabulary projection of a decode step: logits[M][V] = f16(LayerNorm(x))[M][K] . f16(E)[V][K]^T, V ~ 51 865.
// whisper.cpp keeps the token embedding in f16 and ggml's mul_mat rounds the f32 activations to f16 in front of the
// dot products (f32 accumulation), so this is the reference's arithmetic, not an approximation of the f32 mode.
// Reference path: transcribe_rs::SpeechModel::transcribe -> whisper.cpp decoder graph (called from
// src-tauri/src/managers/transcription.rs:183-185, 213-215).
//
// In f32 the projection is bound by the f32 matrix pipe (45 us for 64 Whisper-tiny clips);
// with f16 operands the 32x32x16 MFMA is 16 x faster per k and the kernel is the 40 MB stream of E.  So E is stored
// PACKED in MFMA operand order when the mode is switched on -- tile of 32 vocabulary rows, 16-wide K chunk, lane:
// 16 bytes -- and a wave's load instruction is one contiguous KB.  256 (or 512) persistent workgroups walk over
// the 1621 tiles; the four waves split K four ways (x slice in registers for the whole launch), the next tile's E is
// requested before the current tile's MFMAs, the four partial tiles meet in a double-buffered LDS image: one barrier
// per tile.
#include "asr_common.h"
#include "fd_ln.h"

namespace crispy {
namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ int acc_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

// dst[((t * KC + kc) * 64 + lane) * 8 + e] = f16(E[32 t + (lane & 31)][16 kc + 8 (lane >> 5) + e]), rows >= V zero
__global__ __launch_bounds__(256) void pack_vocab_kernel(const float* __restrict__ E, _Float16* __restrict__ dst, int V, int K,
                                                         long pieces) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= pieces) return;
  const int KC = K / 16;
  const int lane = (int)(idx & 63);
  const long tc = idx >> 6;
  const int kc = (int)(tc % KC);
  const long t = tc / KC;
  const long v = 32 * t + (lane & 31);
  const int k0 = 16 * kc + 8 * (lane >> 5);
  half8 o;
#pragma unroll
  for (int e = 0; e < 8; ++e) o[e] = v < V ? (_Float16)E[v * K + k0 + e] : (_Float16)0.f;
  *reinterpret_cast<half8*>(dst + idx * 8) = o;
}

// FUSE (steps of <= VOCAB_FUSE_ROWS rows on the fused decode path): x is not read -- every workgroup assembles the step's
// final residual stream from the last MLP block's partial rows and normalises it itself (fd_ln.h: the instructions of
// fused_finish_kernel, row r on wave r), which takes the final-LayerNorm launch out of a step that is a chain of launches.
template <int KCH, bool FUSE>     // 16-wide K chunks per wave: K = 64 KCH
__global__ __launch_bounds__(256) void vocab_f16_kernel(const _Float16* __restrict__ x, long ldx, const _Float16* __restrict__ Ep,
                                                        float* __restrict__ C, long ldc, int M, int N, int ntiles, FusedIn in) {
  __shared__ float red[2][4][64 * 33];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int li = lane & 31, g = lane >> 5;
  const int mb = blockIdx.y * 64;
  constexpr int KC = 4 * KCH;
  const half8* ep = reinterpret_cast<const half8*>(Ep) + (long)wave * KCH * 64 + lane;
  int t = blockIdx.x;
  half8 w[KCH], wn[KCH];
  if (t < ntiles) {        // (requested first: in flight while the rows are normalised)
#pragma unroll
    for (int c = 0; c < KCH; ++c) w[c] = ep[((long)t * KC + c) * 64];
  }
  // this wave's K slice of the 64 rows of x, in operand order (rows >= M: a clamped row, never stored)
  half8 xa[2][KCH];
  if constexpr (FUSE) {
    constexpr int D = 64 * KCH, NP = D / 32;
    float* xs = &red[0][0][0];                                   // [VOCAB_FUSE_ROWS][D] f32, then gamma | beta [2 D]
    float* gb = xs + VOCAB_FUSE_ROWS * D;
    _Float16* xn = reinterpret_cast<_Float16*>(&red[1][0][0]);   // [VOCAB_FUSE_ROWS][D] f16
    // a thread's two columns (tid, tid + 256; the second clamped where D ends), everything of a row requested at once:
    // one round trip per row, like the final-LayerNorm kernel this replaces
    constexpr int CPT = (D + 255) / 256;
    int col[CPT];
    float bias[CPT];
#pragma unroll
    for (int k = 0; k < CPT; ++k) {
      col[k] = min(tid + 256 * k, D - 1);
      gb[col[k]] = in.ln_g[col[k]];
      gb[D + col[k]] = in.ln_b[col[k]];
      bias[k] = in.bias[col[k]];
    }
#pragma unroll 1
    for (int r = 0; r < M; ++r) {
      float v[CPT], pv[CPT][NP];
#pragma unroll
      for (int k = 0; k < CPT; ++k) {
        v[k] = in.x_in[(long)r * D + col[k]];
#pragma unroll
        for (int p = 0; p < NP; ++p) pv[k][p] = in.part[((long)p * M + r) * D + col[k]];
      }
#pragma unroll
      for (int k = 0; k < CPT; ++k) xs[r * D + col[k]] = fd_assemble<NP>(v[k], bias[k], pv[k]);      // (a clamped duplicate writes the same value)
    }
    __syncthreads();
    if (wave < M) fd_layernorm_wave<D>(xs + wave * D, gb, xn + wave * D, lane);
    __syncthreads();
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) {
      const _Float16* xr = xn + min(32 * rb + li, M - 1) * D + 16 * (wave * KCH) + 8 * g;
#pragma unroll
      for (int c = 0; c < KCH; ++c) xa[rb][c] = *reinterpret_cast<const half8*>(xr + 16 * c);
    }
    __syncthreads();                                             // red[] is the tiles' from here on
  } else {
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) {
      const _Float16* xr = x + (long)min(mb + 32 * rb + li, M - 1) * ldx + 16 * (wave * KCH) + 8 * g;
#pragma unroll
      for (int c = 0; c < KCH; ++c) xa[rb][c] = *reinterpret_cast<const half8*>(xr + 16 * c);
    }
  }
  int buf = 0;
  for (; t < ntiles; t += gridDim.x) {
    const int tn = min(t + (int)gridDim.x, ntiles - 1);        // past the end: a valid tile, requested and dropped
#pragma unroll
    for (int c = 0; c < KCH; ++c) wn[c] = ep[((long)tn * KC + c) * 64];
    f32x16 a0, a1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { a0[r] = 0.f; a1[r] = 0.f; }
#pragma unroll
    for (int c = 0; c < KCH; ++c) {
      a0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(xa[0][c], w[c], a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(xa[1][c], w[c], a1, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      red[buf][wave][acc_row(r, lane) * 33 + li] = a0[r];
      red[buf][wave][(32 + acc_row(r, lane)) * 33 + li] = a1[r];
    }
    __syncthreads();        // the other buffer is free again once every wave has passed this barrier of the NEXT tile
    const int col = 32 * t + (tid & 31);
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int row = (tid >> 5) + 8 * q;
      const int o = row * 33 + (tid & 31);
      const float v = (red[buf][0][o] + red[buf][1][o]) + (red[buf][2][o] + red[buf][3][o]);
      if (mb + row < M && col < N) C[(long)(mb + row) * ldc + col] = v;
    }
    buf ^= 1;
#pragma unroll
    for (int c = 0; c < KCH; ++c) w[c] = wn[c];
  }
}

}  // namespace

size_t vocab_f16_packed_bytes(int V, int K) { return (size_t)((V + 31) / 32) * 32 * K * sizeof(_Float16); }

hipError_t pack_vocab_f16(const float* E, void* dst, int V, int K, hipStream_t s) {
  if (K % 64 != 0) return hipErrorInvalidValue;
  const long pieces = (long)((V + 31) / 32) * (K / 16) * 64;
  hipLaunchKernelGGL(pack_vocab_kernel, dim3((unsigned)((pieces + 255) / 256)), dim3(256), 0, s, E,
                     reinterpret_cast<_Float16*>(dst), V, K, pieces);
  return hipGetLastError();
}

hipError_t vocab_f16_fused(const FusedIn& in, const void* Ep, float* C, long ldc, int M, int N, int K, hipStream_t s) {
  if (M < 1 || M > VOCAB_FUSE_ROWS || !in.x_in || !in.bias || !in.part || !in.ln_g || !in.ln_b) return hipErrorInvalidValue;
  const int ntiles = (N + 31) / 32;
  const _Float16* eh = reinterpret_cast<const _Float16*>(Ep);
  const dim3 grid((unsigned)min(ntiles, 512), 1), block(256);
  if (K == 384) hipLaunchKernelGGL((vocab_f16_kernel<6, true>), grid, block, 0, s, nullptr, 0L, eh, C, ldc, M, N, ntiles, in);
  else if (K == 512) hipLaunchKernelGGL((vocab_f16_kernel<8, true>), grid, block, 0, s, nullptr, 0L, eh, C, ldc, M, N, ntiles, in);
  else return hipErrorInvalidValue;
  return hipGetLastError();
}

hipError_t vocab_f16(const void* x, long ldx, const void* Ep, float* C, long ldc, int M, int N, int K, hipStream_t s) {
  const int ntiles = (N + 31) / 32;
  const _Float16* xh = reinterpret_cast<const _Float16*>(x);
  const _Float16* eh = reinterpret_cast<const _Float16*>(Ep);
  const dim3 block(256);
  const unsigned my = (unsigned)((M + 63) / 64);
#define CRISPY_VOCAB(KCH, WGS)                                                                              \
  hipLaunchKernelGGL((vocab_f16_kernel<KCH, false>), dim3((unsigned)min(ntiles, WGS), my), block, 0, s, xh, ldx, eh, C, ldc, M, N, ntiles, FusedIn{})
  switch (K) {      // the widths of the Whisper family
    case 384: CRISPY_VOCAB(6, 512); break;
    case 512: CRISPY_VOCAB(8, 512); break;
    case 768: CRISPY_VOCAB(12, 512); break;
    case 1024: CRISPY_VOCAB(16, 256); break;
    case 1280: CRISPY_VOCAB(20, 256); break;
    default: return hipErrorInvalidValue;
  }
#undef CRISPY_VOCAB
  return hipGetLastError();
}

}  // namespace crispy
