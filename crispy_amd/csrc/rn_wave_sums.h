// rn_wave_sums.h -- several wave-wide sums for the price of one (gfx950, wave64).
//
// A lone wave sum is six dependent cross-lane steps (rn_kernels.hip: wave_sum) and the frame kernel takes ~62 of them per
// frame (the lag inner products of the pitch search, the LPC autocorrelation): ~560 VALU instructions, 7 % of the frame.
// Here N values are reduced TOGETHER: v_permlane32_swap puts the lower halves of two values into one register and their
// upper halves into another -- one add folds both values to 32 lanes --, v_permlane16_swap does the same with the odd /
// even 16-lane rows of two such registers, and four DPP steps finish the four rows at once.  Four sums: 3 swaps + 3 adds +
// 4 DPP adds + 4 v_readlane instead of 4 x (6 steps + v_readlane).  The association of the additions differs from the
// single form (lanes l and l + 32 first); both are trees, neither is the reference's sequential order.
// Lane semantics checked on the hardware by tools/micro/wave_sum4_test.hip.
#pragma once
#include <hip/hip_runtime.h>

namespace crispy {

template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ float ws_dpp(float v) {
  const int iv = __float_as_int(v);
  return __int_as_float(__builtin_amdgcn_update_dpp(0, iv, CTRL, ROW_MASK, 0xf, false));
}
// first := [first.lanes 0-31 | second.lanes 0-31], second := [first.lanes 32-63 | second.lanes 32-63]
__device__ __forceinline__ void ws_swap32(float& a, float& b) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
  a = __uint_as_float(r[0]);
  b = __uint_as_float(r[1]);
}
// rows of 16 lanes: first := [f0, s0, f2, s2], second := [f1, s1, f3, s3]
__device__ __forceinline__ void ws_swap16(float& a, float& b) {
  const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
  a = __uint_as_float(r[0]);
  b = __uint_as_float(r[1]);
}
__device__ __forceinline__ float ws_rows(float q) {     // every lane of a 16-lane row ends up with the row's sum
  q += ws_dpp<0xB1>(q);         // quad_perm [1,0,3,2]
  q += ws_dpp<0x4E>(q);         // quad_perm [2,3,0,1]
  q += ws_dpp<0x141>(q);        // row_half_mirror
  q += ws_dpp<0x140>(q);        // row_mirror
  return q;
}
__device__ __forceinline__ float ws_lane(float v, int l) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l));
}
__device__ __forceinline__ void wave_sum4(float& a, float& b, float& c, float& d) {
  ws_swap32(a, b);
  float ab = a + b;             // lanes 0-31: a folded to 32 lanes, lanes 32-63: b
  ws_swap32(c, d);
  float cd = c + d;
  ws_swap16(ab, cd);
  const float q = ws_rows(ab + cd);     // row 0: a, row 1: c, row 2: b, row 3: d
  a = ws_lane(q, 0);
  c = ws_lane(q, 16);
  b = ws_lane(q, 32);
  d = ws_lane(q, 48);
}
__device__ __forceinline__ void wave_sum2(float& a, float& b) {
  ws_swap32(a, b);
  float q = ws_rows(a + b);             // rows 0, 1: a; rows 2, 3: b
  q += ws_dpp<0x142, 0xa>(q);           // row_bcast:15 into rows 1 and 3
  a = ws_lane(q, 31);
  b = ws_lane(q, 63);
}
__device__ __forceinline__ float wave_sum1(float v) {
  v = ws_rows(v);
  v += ws_dpp<0x142, 0xa>(v);           // row_bcast:15 into rows 1 and 3
  v += ws_dpp<0x143, 0xc>(v);           // row_bcast:31 into rows 2 and 3
  return ws_lane(v, 63);
}
// v[i] := sum over the 64 lanes of v[i], wave-uniform, for all i < N
template <int N>
__device__ __forceinline__ void wave_sums(float (&v)[N]) {
  int i = 0;
#pragma unroll
  for (; i + 4 <= N; i += 4) wave_sum4(v[i], v[i + 1], v[i + 2], v[i + 3]);
  if (N - i == 3) {
    float z = 0.f;
    wave_sum4(v[i], v[i + 1], v[i + 2], z);
  } else if (N - i == 2) {
    wave_sum2(v[i], v[i + 1]);
  } else if (N - i == 1) {
    v[i] = wave_sum1(v[i]);
  }
}

}  // namespace crispy
