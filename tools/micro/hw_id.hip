#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* out) {
  unsigned hw, xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  if (threadIdx.x == 0) { out[2 * blockIdx.x] = hw; out[2 * blockIdx.x + 1] = xcc; }
}
int main() {
  unsigned* d; hipMalloc(&d, 8 * 4096); hipMemset(d, 0, 8 * 4096);
  hipLaunchKernelGGL(k, dim3(2048), dim3(256), 0, 0, d);
  static unsigned h[2 * 2048]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int i = 0; i < 24; ++i) printf("wg %d: hw_id %08x xcc %x  (cu %u sh %u se %u simd %u wave %u)\n", i, h[2*i], h[2*i+1], (h[2*i]>>8)&15, (h[2*i]>>12)&1, (h[2*i]>>13)&7, (h[2*i]>>4)&3, h[2*i]&15);
  // distinct (xcc, se, sh, cu) keys
  int seen[1 << 12] = {0}, n = 0;
  for (int i = 0; i < 2048; ++i) { unsigned key = ((h[2*i+1] & 15) << 8) | ((h[2*i] >> 8) & 0xff); if (!seen[key]++) ++n; }
  printf("distinct keys %d\n", n);
  return 0;
}
