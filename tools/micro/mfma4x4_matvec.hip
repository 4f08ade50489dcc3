// Developer micro-test: one-wave matrix-vector product on v_mfma_f32_4x4x4_16B_f16 with lane = output row.
// 16 blocks of 4x4x4: the "A" operand carries the activation vector split into three f16 terms (row i of a block =
// term i, the same in all 16 blocks), the "B" operand carries 4 consecutive-k f16 weights of row = lane, so D[i][j]
// of block b lands in lane 4b+j, register i = partial sum of term i for row `lane`.  Checks the register layout
// against a host reference and measures the issue rate beside v_fma_mix_f32.
//   hipcc --offload-arch=gfx950 -O3 -o gpurun_out/mfma4x4 tools/micro/mfma4x4_matvec.hip && gpurun_out/mfma4x4
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));

// W: [K/8][64][8] f16 (int8 values), x: [K] f32; y[row] = sum_k W[k][row] x[k]
template <int K8>
__global__ __launch_bounds__(64) void matvec_mfma(const h8* __restrict__ W, const float* __restrict__ x, float* y) {
  __shared__ __attribute__((aligned(16))) _Float16 img[3][K8 * 8];
  const int lane = threadIdx.x;
  for (int i = lane; i < K8 * 8; i += 64) {
    const float v = x[i];
    const _Float16 hi = (_Float16)v;
    const float r1 = v - (float)hi;
    const _Float16 lo = (_Float16)r1;
    const _Float16 lo2 = (_Float16)(r1 - (float)lo);
    img[0][i] = hi; img[1][i] = lo; img[2][i] = lo2;
  }
  __syncthreads();
  const int t = min(lane & 3, 2);
  f4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int k = 0; k < K8; ++k) {
    const h8 w = W[k * 64 + lane];
    const h8 a = *reinterpret_cast<const h8*>(&img[t][k * 8]);
    acc = __builtin_amdgcn_mfma_f32_4x4x4f16(h4{a[0], a[1], a[2], a[3]}, h4{w[0], w[1], w[2], w[3]}, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_4x4x4f16(h4{a[4], a[5], a[6], a[7]}, h4{w[4], w[5], w[6], w[7]}, acc, 0, 0, 0);
  }
  y[lane] = acc[0] + acc[1] + acc[2];
}

// rate: ITER x (CH chains x 2 MFMA) per wave, vs ITER x CH x 8 v_fma_mix
template <int CH>
__global__ __launch_bounds__(64) void rate_mfma(float* out, int iters, h4 a, h4 b) {
  f4 acc[CH];
#pragma unroll
  for (int c = 0; c < CH; ++c) acc[c] = f4{0.f, 0.f, 0.f, 0.f};
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int c = 0; c < CH; ++c) acc[c] = __builtin_amdgcn_mfma_f32_4x4x4f16(a, b, acc[c], 0, 0, 0);
#pragma unroll
    for (int c = 0; c < CH; ++c) acc[c] = __builtin_amdgcn_mfma_f32_4x4x4f16(b, a, acc[c], 0, 0, 0);
  }
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < CH; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
  out[blockIdx.x * 64 + threadIdx.x] = s;
}
template <int CH>
__global__ __launch_bounds__(64) void rate_mix(float* out, int iters, h8 w, float x) {
  float acc[CH][2];
#pragma unroll
  for (int c = 0; c < CH; ++c) acc[c][0] = acc[c][1] = 0.f;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int e = 0; e < 8; e += 2)
#pragma unroll
      for (int c = 0; c < CH; ++c) {
        acc[c][0] = fmaf((float)w[e], x, acc[c][0]);
        acc[c][1] = fmaf((float)w[e + 1], x, acc[c][1]);
      }
    asm volatile("" : "+v"(x));
  }
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < CH; ++c) s += acc[c][0] + acc[c][1];
  out[blockIdx.x * 64 + threadIdx.x] = s;
}
// both at once in one wave: does the matrix pipe run beside the VALU of the same SIMD?
template <int CH>
__global__ __launch_bounds__(64) void rate_both(float* out, int iters, h4 a, h4 b, h8 w, float x) {
  f4 acc[CH];
  float av[CH][2];
#pragma unroll
  for (int c = 0; c < CH; ++c) { acc[c] = f4{0.f, 0.f, 0.f, 0.f}; av[c][0] = av[c][1] = 0.f; }
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      acc[c] = __builtin_amdgcn_mfma_f32_4x4x4f16(a, b, acc[c], 0, 0, 0);
      av[c][0] = fmaf((float)w[0], x, av[c][0]);
      av[c][1] = fmaf((float)w[1], x, av[c][1]);
      acc[c] = __builtin_amdgcn_mfma_f32_4x4x4f16(b, a, acc[c], 0, 0, 0);
      av[c][0] = fmaf((float)w[2], x, av[c][0]);
      av[c][1] = fmaf((float)w[3], x, av[c][1]);
    }
    asm volatile("" : "+v"(x));
  }
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < CH; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3] + av[c][0] + av[c][1];
  out[blockIdx.x * 64 + threadIdx.x] = s;
}

template <class F>
static float timed(F f) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  f(100);
  hipEventRecord(e0);
  f(20000);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms;
}

int main() {
  constexpr int K8 = 15, K = K8 * 8;
  std::vector<_Float16> W(K8 * 64 * 8);
  std::vector<float> x(K), yref(64, 0.f);
  srand(1);
  for (int k = 0; k < K; ++k) x[k] = (float)((rand() % 20001) - 10000) / 9973.f * (k % 7 == 0 ? 13.f : 1.f);
  for (int k8 = 0; k8 < K8; ++k8)
    for (int r = 0; r < 64; ++r)
      for (int e = 0; e < 8; ++e) W[(k8 * 64 + r) * 8 + e] = (_Float16)(float)((rand() % 255) - 127);
  std::vector<double> yd(64, 0.0);
  for (int r = 0; r < 64; ++r)
    for (int k = 0; k < K; ++k) yd[r] += (double)(float)W[((k / 8) * 64 + r) * 8 + (k % 8)] * (double)x[k];
  h8* dW; float *dx, *dy;
  hipMalloc(&dW, W.size() * 2); hipMalloc(&dx, K * 4); hipMalloc(&dy, 64 * 4);
  hipMemcpy(dW, W.data(), W.size() * 2, hipMemcpyHostToDevice);
  hipMemcpy(dx, x.data(), K * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(matvec_mfma<K8>, dim3(1), dim3(64), 0, 0, dW, dx, dy);
  std::vector<float> y(64);
  hipMemcpy(y.data(), dy, 64 * 4, hipMemcpyDeviceToHost);
  double worst = 0, scale = 0;
  for (int r = 0; r < 64; ++r) { worst = fmax(worst, fabs(y[r] - yd[r])); scale = fmax(scale, fabs(yd[r])); }
  printf("layout check: max |y - ref| = %.3e (scale %.3e) -> %s\n", worst, scale, worst < 1e-5 * scale ? "OK" : "MISMATCH");
  printf("  y[0..3] = %g %g %g %g   ref = %g %g %g %g\n", y[0], y[1], y[2], y[3], yd[0], yd[1], yd[2], yd[3]);

  float* out; hipMalloc(&out, 4 * 64 * 4096 * 4);
  const h4 a = {1, 2, 3, 4}, b = {1, 1, 1, 1};
  const h8 w8 = {1, 2, 3, 4, 5, 6, 7, 8};
  for (int waves = 1; waves <= 4; waves *= 2) {
    const int blocks = 256 * 4 * waves;   // `waves` waves per SIMD
    const float t1 = timed([&](int it) { hipLaunchKernelGGL(rate_mfma<3>, dim3(blocks), dim3(64), 0, 0, out, it, a, b); });
    const float t2 = timed([&](int it) { hipLaunchKernelGGL(rate_mix<3>, dim3(blocks), dim3(64), 0, 0, out, it, w8, 1.f); });
    const float t3 = timed([&](int it) { hipLaunchKernelGGL(rate_both<3>, dim3(blocks), dim3(64), 0, 0, out, it, a, b, w8, 1.f); });
    // per iteration and chain: 8 MACs per lane either way
    printf("%d waves/SIMD, 3 chains, 20000 x 8 MACs per lane and chain: mfma4x4x4 %.3f ms, fma_mix %.3f ms, "
           "both interleaved (mfma 8 MACs + 4 fma_mix) %.3f ms\n", waves, t1, t2, t3);
  }
  return 0;
}
