// Developer micro-benchmark: how much does doubling the resident waves buy for a representative stage of the RNNoise
// frame kernel?  The stage is the real code (windowed 960-point analysis transform from global memory, real
// post-processing, Opus band energies) from rn_kernels.hip, one wave per stream, looped over T frames.
//   config A: 10 KB of LDS per wave, <= 128 VGPRs  -> 16 waves per CU  (what 4096 streams give the frame kernel)
//   config B:  5 KB of LDS per wave, <=  64 VGPRs  -> 32 waves per CU  (twice the streams, i.e. what a design with
//              twice the resident waves could approach)
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I crispy_amd/csrc -o tools/micro/fft_occupancy.bin tools/micro/fft_occupancy.hip
#include "../../crispy_amd/csrc/rn_kernels.hip"
#include <cmath>
#include <cstdio>
#include <vector>

namespace crispy {
namespace {

template <int LDS_BYTES, int WAVES_PER_EU>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(WAVES_PER_EU, WAVES_PER_EU)))
void fft_stage_kernel(const float* __restrict__ x, float* __restrict__ out, const RnTables* __restrict__ tab_in, int T) {
  __shared__ __attribute__((aligned(16))) float2 A[482];
  __shared__ float part[200];
  __shared__ float E[24];
  __shared__ char pad[LDS_BYTES - 482 * 8 - 200 * 4 - 24 * 4];
  const int lane0 = threadIdx.x, b = blockIdx.x;
  if (lane0 == 0) pad[0] = 0;
  BandEdges be;
  {
    const int i = min(lane0, RN_NB - 1);
    be.e0 = tab_in->eband[i]; be.e1 = tab_in->eband[i + 1]; be.em1 = tab_in->eband[max(i - 1, 0)];
  }
  float accE = 0.f;
  const float* xs = x + (long)b * (T + 1) * RN_FRAME;
  for (int t = 0; t < T; ++t) {
    int lane = lane0;
    asm volatile("" : "+v"(lane));
    typedef const __attribute__((address_space(1))) RnTables* GTabPtr;
    GTabPtr tg = (GTabPtr)tab_in;
    asm volatile("" : "+s"(tg)::"memory");
    const RnTables* __restrict__ tab = (const RnTables*)tg;
    const float2* __restrict__ w960 = tab->w960;
    const float* __restrict__ hw = tab->half_window;
    const float* xw = xs + (long)t * RN_FRAME;
    fft480_from(A, [&](int j, int r) {
      const int n = j + 120 * r;
      const float2 v = *reinterpret_cast<const float2*>(xw + 2 * n);
      const float w0 = r < 2 ? hw[2 * n] : hw[959 - 2 * n];
      const float w1 = r < 2 ? hw[2 * n + 1] : hw[958 - 2 * n];
      return make_float2(v.x * w0, v.y * w1);
    }, w960, lane);
    real_fwd_post(A, w960, lane);
    band_pairs<false>(A, nullptr, part, E, nullptr, nullptr, tab, be, lane);
    if (lane < RN_NB) accE += E[lane];
    __syncthreads();
  }
  out[(long)b * 64 + lane0] = accE + (float)pad[0];
}

}  // namespace
}  // namespace crispy

int main() {
  using namespace crispy;
  RnTables* tab = new RnTables();
  const double pi = 3.14159265358979323846;
  for (int i = 0; i < RN_FRAME; ++i) { const double s = std::sin(.5 * pi * (i + .5) / RN_FRAME); tab->half_window[i] = (float)std::sin(.5 * pi * s * s); }
  for (int k = 0; k < RN_WINDOW; ++k) { tab->w960[k].x = (float)std::cos(-2.0 * pi * k / RN_WINDOW); tab->w960[k].y = (float)std::sin(-2.0 * pi * k / RN_WINDOW); }
  static const int eband[RN_NB] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 10, 12, 14, 16, 20, 24, 28, 34, 40, 48, 60, 78, 100};
  for (int i = 0; i < 24; ++i) tab->eband[i] = i < RN_NB ? eband[i] : 100;
  for (int i = 0; i < RN_NB - 1; ++i) { const int bs = (eband[i + 1] - eband[i]) * 4; for (int j = 0; j < bs; ++j) { tab->bin_band[eband[i] * 4 + j] = i; tab->bin_frac[eband[i] * 4 + j] = (float)j / (float)bs; } }
  RnTables* d_tab; (void)hipMalloc(&d_tab, sizeof(RnTables)); (void)hipMemcpy(d_tab, tab, sizeof(RnTables), hipMemcpyHostToDevice);
  const int T = 50, BMAX = 16384;
  std::vector<float> hx((size_t)BMAX * (T + 1) * RN_FRAME);
  for (size_t i = 0; i < hx.size(); ++i) hx[i] = (float)((i * 2654435761u) % 2001) - 1000.f;
  float *d_x, *d_out; (void)hipMalloc(&d_x, hx.size() * 4); (void)hipMalloc(&d_out, (size_t)BMAX * 64 * 4);
  (void)hipMemcpy(d_x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  auto time = [&](auto launch, const char* name, int B) {
    launch(B);
    (void)hipEventRecord(e0);
    launch(B);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-44s B=%5d: %7.3f ms  -> %7.1f M stage-frames/s\n", name, B, ms, (double)B * T / ms / 1e3);
  };
  for (int B : {4096, 8192, 16384}) {
    time([&](int b) { hipLaunchKernelGGL((fft_stage_kernel<10176, 4>), dim3(b), dim3(64), 0, 0, d_x, d_out, d_tab, T); }, "A: 10 KB LDS, 128 VGPR (16 waves/CU)", B);
    time([&](int b) { hipLaunchKernelGGL((fft_stage_kernel<5056, 8>), dim3(b), dim3(64), 0, 0, d_x, d_out, d_tab, T); }, "B:  5 KB LDS,  64 VGPR (32 waves/CU)", B);
  }
  return 0;
}
