// Developer micro-test: what does a wave-wide 16-byte-per-lane global store cost, by the shape of what it covers?
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/store_rate tools/micro/store_rate.hip && /tmp/store_rate
// Each workgroup (256 threads) writes a [rows][row_bytes] tile of an [M][ld] matrix, one wave-store = 64 lanes x 16 B:
//   lanes_per_row = 64: 1 row x 1024 B;  8: 8 rows x 128 B (the GEMM epilogue's f16 stores);  16: 4 rows x 256 B (f32);
//   4: 16 rows x 64 B.  Prints bytes per clock per CU at 2.4 GHz.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ __launch_bounds__(256) void k(uint4* out, long ld16 /* row stride in 16-B units */, int lanes_per_row, int stores_per_wave,
                                         int tiles_per_wg) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int rows_per_store = 64 / lanes_per_row;
  const uint4 v = make_uint4(lane, wave, blockIdx.x, 7);
  for (int t = 0; t < tiles_per_wg; ++t) {
    const long tile = (long)blockIdx.x * tiles_per_wg + t;
    // a tile = 4 waves x stores_per_wave stores; rows are consecutive matrix rows, the tile's column block is fixed
    const long row0 = (tile * 4 + wave) * (long)(stores_per_wave * rows_per_store);
    for (int sidx = 0; sidx < stores_per_wave; ++sidx) {
      const long row = row0 + (long)sidx * rows_per_store + lane / lanes_per_row;
      out[row * ld16 + (lane % lanes_per_row)] = v;
    }
  }
}
int main() {
  const size_t bytes = 40ull << 30;
  uint4* d; hipMalloc(&d, bytes);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  struct Cfg { const char* name; int lpr; long ld16; } cfgs[] = {
    {"1 row x 1024 B, dense rows      ", 64, 64}, {"8 rows x 128 B, ld = 128 B (dense)", 8, 8}, {"8 rows x 128 B, ld = 4096 B      ", 8, 256},
    {"8 rows x 128 B, ld = 2048 B      ", 8, 128}, {"4 rows x 256 B, ld = 2048 B      ", 16, 128}, {"16 rows x 64 B, ld = 4096 B      ", 4, 256},
    {"8 rows x 128 B, ld = 1024 B      ", 8, 64}};
  for (auto& c : cfgs) {
   for (int per_cu : {8, 1, -16}) {      // workgroups of 4 waves per CU that store at the same time; -16: 16 workgroups on the whole chip
    const int stores_per_wave = 16, wgs = per_cu > 0 ? 256 * per_cu : -per_cu, tiles = per_cu > 0 ? 64 / per_cu : 64;
    const long rows = (long)wgs * tiles * 4 * stores_per_wave * (64 / c.lpr);
    if ((size_t)rows * c.ld16 * 16 > bytes) { printf("%s: too large\n", c.name); continue; }
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(e0);
      hipLaunchKernelGGL(k, dim3(wgs), dim3(256), 0, 0, d, c.ld16, c.lpr, stores_per_wave, tiles);
      hipEventRecord(e1); hipEventSynchronize(e1);
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double b = (double)wgs * tiles * 4 * stores_per_wave * 1024.0;
    printf("%s %d WG/CU %4.0f MB: %.1f us, %.2f TB/s, %.1f B/clk/CU\n", c.name, per_cu, b / 1e6, ms * 1e3, b / ms / 1e9, b / (ms * 1e-3) / 2.4e9 / (per_cu > 0 ? 256 : -per_cu));
   }
  }
  return 0;
}
