// Ablation micro-benchmark of conv1d_k9_p16_kernel.  hipcc --offload-arch=gfx950 -O3 -std=c++17 -I orca_amd/csrc ...
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "conv_p16.h"
template <int CT, int MW, int NW, int WM, int ABL>
static void run(ConvP16Args a, const char* what) {
  constexpr int MT = WM * MW * 32;
  int per_cu = 1;
  (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, conv1d_k9_p16_kernel<CT, MW, NW, WM, ABL>, WM * 64, 0);
  a.tiles_per_row = (a.n + MT - 1) / MT;
  long ntiles = a.tiles_per_row * (a.cout / CT), grid = 256L * per_cu; if (grid > ntiles) grid = ntiles;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e9;
  for (int r = 0; r < 4; ++r) {
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((conv1d_k9_p16_kernel<CT, MW, NW, WM, ABL>), dim3((unsigned)grid), dim3(WM * 64), 0, 0, a);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); if (r > 0 && ms < best) best = ms;
  }
  double fl = 2.0 * 9 * a.nchunks * 16 * a.cout * (double)a.n;
  printf("CT=%d MT=%d cin=%d cout=%d n=%ld ABL=%2d (%s) occ=%d: %.3f ms  %.1f TFLOP/s-eq  [%s]\n", CT, MT, a.nchunks * 16, a.cout, a.n, ABL, what, per_cu, best, fl / best / 1e9, hipGetErrorString(hipGetLastError()));
}
int main(int argc, char** argv) {
  long n = argc > 1 ? atol(argv[1]) : 8000000;
  const long plen = ((n + 511) / 512) * 512 + 8;
  f32x4 *x, *y, *w; float* bias;
  hipMalloc(&x, (size_t)128 * plen * 4); hipMalloc(&y, (size_t)128 * plen * 4); hipMalloc(&w, (size_t)8 * 2 * 9 * 2 * 128 * 16); hipMalloc(&bias, 512);
  if (argc > 2) {   // random fp16 content (hi ~ U(-1,1), lo tiny) instead of a constant fill: realistic switching activity
    std::vector<unsigned short> hx((size_t)64 * plen * 2);
    unsigned s = 1234567u;
    for (auto& v : hx) { s = s * 1664525u + 1013904223u; v = (unsigned short)(((s >> 9) & 0x8fff) | 0x3000); }
    hipMemcpy(x, hx.data(), hx.size() * 2, hipMemcpyHostToDevice);
    printf("random activations\n");
  } else
  hipMemset(x, 0x2c, (size_t)128 * plen * 4); hipMemset(w, 0x2c, (size_t)8 * 2 * 9 * 2 * 128 * 16); hipMemset(bias, 0, 512);
  ConvP16Args a{}; a.x = x; a.w = w; a.bias = bias; a.y = y; a.r1 = nullptr; a.x_plen = plen; a.y_plen = plen; a.n = n; a.nchunks = 4; a.cout = 64; a.relu = 1; a.out_mode = 0; a.flag = nullptr;
  for (int round = 0; round < 3; ++round) {
    printf("-- round %d\n", round);
    a.out_mode = 0;
    run<64, 2, 2, 8, 0>(a, "8 waves, 64x64 wave tile (library)");
    run<64, 1, 2, 16, 0>(a, "16 waves, 32x64 wave tile");
    run<64, 1, 2, 8, 0>(a, "8 waves, 32x64, MT=256");
    run<64, 4, 2, 4, 0>(a, "4 waves, 128x64 wave tile");
  }
  return 0;
}
