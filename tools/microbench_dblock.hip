// Ablation micro-benchmark of conv2d_dblock_kernel (one Decoder residual block per launch).
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -w -I orca_amd/csrc -I include -I tools tools/microbench_dblock.hip -o tools/microbench_dblock
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "parked/conv2d_dblock2.h"
template <int NS, int DT, int ABL>
static void run(DBlockArgs a, int B, const char* what) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e9;
  for (int r = 0; r < 6; ++r) {
    hipEventRecord(e0, 0);
    for (int k = 0; k < 10; ++k) hipLaunchKernelGGL((conv2d_dblock_kernel<NS, DT, ABL>), dim3(256, B), dim3(512), 0, 0, a);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); if (r > 0 && ms < best) best = ms;
  }
  printf("NS=%d d=%2d B=%d ABL=%2d (%s): %.1f us per launch  [%s]\n", NS, a.dil, B, ABL, what, best * 100.f, hipGetErrorString(hipGetLastError()));
}
template <int NS, int DT, int PX, int ABL>
static void run2(DBlockArgs a, int B, const char* what) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e9;
  for (int r = 0; r < 6; ++r) {
    hipEventRecord(e0, 0);
    for (int k = 0; k < 10; ++k) hipLaunchKernelGGL((conv2d_dblock2_kernel<NS, DT, PX, ABL>), dim3(65536 / PX, B), dim3(2 * PX), 0, 0, a);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); if (r > 0 && ms < best) best = ms;
  }
  printf("v2 PX=%d NS=%d d=%2d B=%d ABL=%2d (%s): %.1f us per launch  [%s]\n", PX, NS, a.dil, B, ABL, what, best * 100.f, hipGetErrorString(hipGetLastError()));
}
// one launch of each kernel from the same input: the maps must be bit-identical
template <int NS, int DT, int PX>
static void same(DBlockArgs a, int B, const std::vector<unsigned short>& h, size_t map_units) {
  const size_t bytes = map_units * 16 * B;
  std::vector<unsigned short> o1(bytes / 2), o2(bytes / 2);
  hipMemcpy(a.cur, h.data(), bytes, hipMemcpyHostToDevice);
  hipLaunchKernelGGL((conv2d_dblock_kernel<NS, DT, 0>), dim3(256, B), dim3(512), 0, 0, a);
  hipMemcpy(o1.data(), a.cur, bytes, hipMemcpyDeviceToHost);
  hipMemcpy(a.cur, h.data(), bytes, hipMemcpyHostToDevice);
  hipLaunchKernelGGL((conv2d_dblock2_kernel<NS, DT, PX, 0>), dim3(65536 / PX, B), dim3(2 * PX), 0, 0, a);
  hipMemcpy(o2.data(), a.cur, bytes, hipMemcpyDeviceToHost);
  size_t diff = 0, nz = 0;
  for (size_t i = 0; i < o1.size(); ++i) { diff += o1[i] != o2[i]; nz += o1[i] != 0; }
  printf("same? PX=%d NS=%d DT=%d d=%2d B=%d: %zu of %zu halfwords differ (%zu nonzero)  [%s]\n", PX, NS, DT, a.dil, B, diff, o1.size(), nz, hipGetErrorString(hipGetLastError()));
  hipMemcpy(a.cur, h.data(), bytes, hipMemcpyHostToDevice);
}
int main() {
  const int n = 250;
  const size_t map = (size_t)8 * 2 * n * 256;           // units of a 64-channel M16 map (NS = 2; the bf16 runs use half of it)
  f32x4* cur; hipMalloc(&cur, map * 16 * 8);
  std::vector<unsigned short> h(map * 8 * 8); unsigned s = 12345u;
  for (auto& v : h) { s = s * 1664525u + 1013904223u; v = (unsigned short)(((s >> 9) & 0x83ff) | 0x3000); }
  hipMemcpy(cur, h.data(), h.size() * 2, hipMemcpyHostToDevice);
  void* w[4]; float* b[4];
  for (int k = 0; k < 4; ++k) {
    const size_t units = (size_t)4 * 2 * 9 * 2 * 64;
    hipMalloc(&w[k], units * 16); hipMalloc(&b[k], 256);
    std::vector<unsigned short> hw(units * 8);
    for (auto& v : hw) { s = s * 1664525u + 1013904223u; v = (unsigned short)(((s >> 9) & 0x83ff) | 0x2000); }
    hipMemcpy(w[k], hw.data(), hw.size() * 2, hipMemcpyHostToDevice); hipMemset(b[k], 0, 256);
  }
  DBlockArgs a{}; a.cur = cur; a.bs = map; a.H = n; a.W = n; a.flag = nullptr;
  for (int k = 0; k < 4; ++k) { a.w[k] = w[k]; a.bias[k] = b[k]; }
  a.bs = map;
  for (int d : {32, 64}) { a.dil = d; same<2, 1, 128>(a, 2, h, map); same<1, 1, 128>(a, 2, h, map); same<1, 0, 128>(a, 2, h, map); }
  for (int d : {16, 32, 64}) { a.dil = d; same<2, 1, 256>(a, 2, h, map); same<1, 0, 256>(a, 2, h, map); }
  for (int d : {32, 64}) {
    a.dil = d;
    for (int B : {1, 2, 4, 8}) { run<2, 1, 0>(a, B, "v1"); run2<2, 1, 128, 0>(a, B, "v2"); run2<2, 1, 256, 0>(a, B, "v2"); }
    for (int B : {2, 8}) { run<1, 1, 0>(a, B, "v1 f16"); run2<1, 1, 128, 0>(a, B, "v2 f16"); run2<1, 1, 256, 0>(a, B, "v2 f16"); }
  }
  a.dil = 16;
  for (int B : {1, 2, 4, 8}) { run<2, 1, 0>(a, B, "v1"); run2<2, 1, 256, 0>(a, B, "v2"); }
  for (int B : {2, 8}) { run<1, 1, 0>(a, B, "v1 f16"); run2<1, 1, 256, 0>(a, B, "v2 f16"); }
  if (getenv("DBLOCK_V1_ABL")) {
    for (int d : {16, 32, 64}) {
      a.dil = d;
      for (int rep = 0; rep < 2; ++rep) {
        run<2, 1, 0>(a, 2, "v1 full");
        run<2, 1, 1>(a, 2, "no MFMA");
        run<2, 1, 2>(a, 2, "no W DMA in the loop");
        run<2, 1, 4>(a, 2, "no gather");
        run<2, 1, 8>(a, 2, "no operand reads");
        run<2, 1, 16>(a, 2, "one barrier per layer");
        run<2, 1, 1 + 8>(a, 2, "no MFMA, no operand reads");
        run<2, 1, 1 + 2 + 8>(a, 2, "no MFMA, no reads, no DMA");
        run<2, 1, 1 + 2 + 4 + 8>(a, 2, "only barriers + epilogues");
        run<2, 1, 1 + 2 + 4 + 8 + 16>(a, 2, "only epilogues");
      }
    }
    return 0;
  }
  if (getenv("DBLOCK_V2_ABL")) {
    for (int d : {32, 64}) {
      a.dil = d;
      for (int rep = 0; rep < 2; ++rep) {
      run<2, 1, 0>(a, 2, "v1 full");
      run2<2, 1, 128, 0>(a, 2, "full");
      run2<2, 1, 128, 1>(a, 2, "no MFMA");
      run2<2, 1, 128, 2>(a, 2, "no W DMA in the loop");
      run2<2, 1, 128, 4>(a, 2, "no gather");
      run2<2, 1, 128, 8>(a, 2, "no operand reads");
      run2<2, 1, 128, 16>(a, 2, "one barrier per layer");
      run2<2, 1, 128, 2 + 16>(a, 2, "no W DMA, one barrier per layer");
      run2<2, 1, 128, 1 + 8>(a, 2, "no MFMA, no operand reads");
      run2<2, 1, 128, 1 + 2 + 8>(a, 2, "no MFMA, no reads, no DMA");
      run2<2, 1, 128, 1 + 2 + 4 + 8>(a, 2, "only barriers + epilogues");
      run2<2, 1, 128, 1 + 2 + 4 + 8 + 16>(a, 2, "only epilogues");
      }
    }
    return 0;
  }
  if (getenv("DBLOCK_V2_ONLY")) return 0;
  for (int d : {16, 32, 64}) {
    a.dil = d;
    run<2, 1, 0>(a, 1, "full");
    run<2, 1, 0>(a, 2, "full");
    run<2, 1, 0>(a, 8, "full");
  }
  a.dil = 16;
  run<2, 1, 1>(a, 1, "no MFMA");
  run<2, 1, 2>(a, 1, "no W DMA in the loop");
  run<2, 1, 4>(a, 1, "no gather");
  run<2, 1, 64>(a, 1, "no XCD grouping of the sub-images");
  run<2, 1, 8>(a, 1, "no operand reads");
  run<2, 1, 16>(a, 1, "one barrier per layer");
  run<2, 1, 1 + 8>(a, 1, "no MFMA, no operand reads");
  run<2, 1, 1 + 2 + 8>(a, 1, "no MFMA, no reads, no DMA");
  run<2, 1, 1 + 2 + 4 + 8>(a, 1, "only barriers + epilogues");
  run<1, 0, 0>(a, 1, "bf16 full");
  run<1, 0, 0>(a, 8, "bf16 full");
  return 0;
}
