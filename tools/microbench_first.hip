// Ablation micro-benchmark of conv1d_first_mfma_p16_kernel.  hipcc --offload-arch=gfx950 -O3 -std=c++17 -I orca_amd/csrc ...
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "conv_p16.h"
template <int ABL>
static void run(FirstMfmaArgs a, int grid, const char* what) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e9;
  for (int r = 0; r < 4; ++r) {
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(conv1d_first_mfma_p16_kernel<ABL>, dim3(grid), dim3(256), 0, 0, a);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); if (r > 0 && ms < best) best = ms;
  }
  printf("n=%ld grid=%d ABL=%2d (%s): %.3f ms  %.2f TB/s written  [%s]\n", a.n, grid, ABL, what, best, a.n * 256.0 / best / 1e9, hipGetErrorString(hipGetLastError()));
}
int main(int argc, char** argv) {
  long n = argc > 1 ? atol(argv[1]) : 32000000;
  const long plen = ((n + 512) / 512) * 512 + 8 + (argc > 2 ? atol(argv[2]) : 0);
  unsigned char* codes; f32x4 *y, *w; float* bias;
  hipMalloc(&codes, n); hipMalloc(&y, (size_t)16 * plen * 16 + ((size_t)9 << 30)); hipMalloc(&w, 768 * 16); hipMalloc(&bias, 256);
  std::vector<unsigned char> hc(n); unsigned s = 12345u;
  for (auto& c : hc) { s = s * 1664525u + 1013904223u; c = (s >> 24) & 3; }
  hipMemcpy(codes, hc.data(), n, hipMemcpyHostToDevice); hipMemset(w, 0x2c, 768 * 16); hipMemset(bias, 0, 256);
  FirstMfmaArgs a{}; a.x = nullptr; a.codes = codes; a.codes_L = n; a.codes_off = 0; a.reverse = 0; a.n = n; a.w = w; a.bias = bias; a.y = y; a.y_plen = plen; a.flag = nullptr;
  for (int grid : {768, 1536, 2048, 4096}) run<0>(a, grid, "full");
  run<64>(a, 2048, "tile-major stores");
  run<64 + (4 << 8)>(a, 2048, "blocked planar, 4 tiles (16 KB runs)");
  run<64 + (16 << 8)>(a, 2048, "blocked planar, 16 tiles (64 KB runs)");
  run<64 + (256 << 8)>(a, 2048, "blocked planar, 256 tiles (1 MB runs)");
  run<64 + (4096 << 8)>(a, 2048, "blocked planar, 4096 tiles (16 MB runs)");
  run<64 + (16384 << 8)>(a, 2048, "blocked planar, 16384 tiles (64 MB runs)");
  run<64 + (32768 << 8)>(a, 2048, "blocked planar, 32768 tiles (128 MB runs)");
  run<64 + (65536 << 8)>(a, 2048, "blocked planar, 65536 tiles (256 MB runs)");
  run<16>(a, 2048, "no stores");
  run<1>(a, 2048, "no MFMA");
  run<1 + 32>(a, 2048, "no MFMA, no input fetch");
  run<1 + 16 + 32>(a, 2048, "sync + split only");
  return 0;
}
