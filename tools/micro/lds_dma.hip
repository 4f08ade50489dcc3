// Developer micro-test: where does global_load_lds_dwordx4 put each lane's 16 bytes, and does vmcnt cover it?
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/lds_dma tools/micro/lds_dma.hip && /tmp/lds_dma
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(const uint4* g, uint4* out) {
  __shared__ __attribute__((aligned(16))) uint4 s[256];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  // lane L of wave w fetches global chunk (w * 64 + (63 - L)): a permutation, so that position != source is visible
  __builtin_amdgcn_global_load_lds(g + wave * 64 + (63 - lane), (__attribute__((address_space(3))) void*)(s + wave * 64), 16, 0, 0);
  __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0), lgkmcnt / expcnt untouched
  __syncthreads();
  out[tid] = s[tid];
}
int main() {
  std::vector<uint4> h(256), o(256);
  for (int i = 0; i < 256; ++i) h[i] = make_uint4(i, i + 1000, i + 2000, i + 3000);
  uint4 *dg, *dout;
  hipMalloc(&dg, 4096); hipMalloc(&dout, 4096);
  hipMemcpy(dg, h.data(), 4096, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(256), 0, 0, dg, dout);
  hipMemcpy(o.data(), dout, 4096, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int t = 0; t < 256; ++t) {
    const int w = t >> 6, l = t & 63, want = w * 64 + (63 - l);   // LDS position base + lane * 16 holds what lane fetched
    if ((int)o[t].x != want || (int)o[t].w != want + 3000) { if (bad < 5) printf("pos %d: got %u want %d\n", t, o[t].x, want); ++bad; }
  }
  printf("global_load_lds_dwordx4: lane L -> LDS base + 16 L : %s (%d mismatches)\n", bad ? "NO" : "yes", bad);
  return bad != 0;
}
