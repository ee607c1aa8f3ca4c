// Cost split of the 96-cout planar conv (stage 2 of the Encoder: the dominant kernel instantiation of round 3) by timing-only ablations:
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -I orca_amd/csrc tools/microbench_p16_96.hip -o tools/microbench_p16_96
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "conv_p16.h"
template <int CT, int MW, int NW, int WM, int OM, bool R1, int ABL>
static void run(ConvP16Args a, const char* what) {
  constexpr int MT = WM * MW * 32;
  a.tiles_per_row = (a.n + MT - 1) / MT; a.out_mode = OM;
  long ntiles = a.tiles_per_row * (a.cout / CT), grid = 256; if (grid > ntiles) grid = ntiles;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e9;
  for (int r = 0; r < 4; ++r) {
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((conv1d_k9_p16_kernel<CT, MW, NW, WM, OM, R1, ABL, false>), dim3((unsigned)grid), dim3(WM * 64), 0, 0, a);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); if (r > 0 && ms < best) best = ms;
  }
  double fl = 2.0 * 9 * a.nchunks * 16 * a.cout * (double)a.n;
  printf("CT=%d MT=%d cin=%d cout=%d n=%ld ABL=%5d (%s): %.3f ms  %.1f TFLOP/s-eq  [%s]\n", CT, MT, a.nchunks * 16, a.cout, a.n, ABL, what, best, fl / best / 1e9, hipGetErrorString(hipGetLastError()));
}
int main(int argc, char** argv) {
  setvbuf(stdout, nullptr, _IONBF, 0);
  long n = argc > 1 ? atol(argv[1]) : 8000000;
  const long plen = ((n + 512) / 512) * 512 + 32;
  f32x4 *x, *y, *w; float* bias;
  hipMalloc(&x, (size_t)32 * plen * 16); hipMalloc(&y, (size_t)32 * plen * 16); hipMalloc(&w, (size_t)16 * 2 * 9 * 2 * 128 * 16); hipMalloc(&bias, 512);
  std::vector<unsigned short> hx((size_t)32 * plen * 8);
  unsigned s = 1234567u;
  for (auto& v : hx) { s = s * 1664525u + 1013904223u; v = (unsigned short)(((s >> 9) & 0x8fff) | 0x3000); }
  hipMemcpy(x, hx.data(), hx.size() * 2, hipMemcpyHostToDevice);
  std::vector<unsigned short> hw((size_t)16 * 2 * 9 * 2 * 128 * 8);
  for (auto& v : hw) { s = s * 1664525u + 1013904223u; v = (unsigned short)(((s >> 9) & 0x8fff) | 0x2800); }
  hipMemcpy(w, hw.data(), hw.size() * 2, hipMemcpyHostToDevice); hipMemset(bias, 0, 512);
  ConvP16Args a{}; a.x = x; a.w = w; a.bias = bias; a.y = y; a.x_plen = plen; a.y_plen = plen; a.n = n; a.nchunks = 6; a.cout = 96; a.relu = 1;
  run<96, 1, 3, 8, 0, false, 0>(a, "warm");
  run<96, 1, 3, 8, 0, false, 0>(a, "96 -> 96 plain");
  run<96, 1, 3, 8, 0, false, 8>(a, "LDS operands read once");
  run<96, 1, 3, 8, 0, false, 1>(a, "no DMA after the first step");
  run<96, 1, 3, 8, 0, false, 16>(a, "no stores");
  run<96, 1, 3, 8, 0, false, 1 + 16>(a, "no DMA, no stores");
  run<96, 1, 3, 8, 0, false, 1 + 16 + 8>(a, "no DMA, no stores, LDS once");
  run<96, 1, 3, 8, 0, false, 16384>(a, "6 of 9 taps' MFMAs");
  { ConvP16Args b = a; b.cout = 64; b.nchunks = 6; run<64, 2, 2, 8, 0, false, 0>(b, "96 -> 64 on the 64-cout tile (same K)"); run<64, 2, 2, 8, 0, false, 8>(b, "... LDS once"); }
  { ConvP16Args b = a; b.cout = 128; b.nchunks = 6; run<64, 2, 2, 8, 0, false, 0>(b, "96 -> 128 on the 64-cout tile"); }
  return 0;
}
