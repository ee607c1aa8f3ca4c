// F1 cost split (timing only): hipcc --offload-arch=gfx950 -O3 -std=c++17 -I orca_amd/csrc tools/microbench_f1.hip -o tools/microbench_f1
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "conv_p16.h"
template <int ABL, bool F1>
static void run(ConvP16Args a, const char* what) {
  int per_cu = 1;
  (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, conv1d_k9_p16_kernel<64, 2, 2, 8, 0, false, ABL, F1>, 512, 0);
  const long ntiles = a.tiles_per_row * (a.cout / 64);
  long grid = 256L * per_cu; if (grid > ntiles) grid = ntiles;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e9f;
  for (int r = 0; r < 4; ++r) {
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((conv1d_k9_p16_kernel<64, 2, 2, 8, 0, false, ABL, F1>), dim3((unsigned)grid), dim3(512), 0, 0, a);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); if (r && ms < best) best = ms;
  }
  printf("%-60s %.3f ms  [%s]\n", what, best, hipGetErrorString(hipGetLastError()));
}
int main() {
  const long n = 32000000, plen = ((n + 512) / 512) * 512 + 8;
  ConvP16Args a{};
  f32x4 *x, *y, *w; float* bias; unsigned* flag;
  hipMalloc(&x, 16L * plen * 16); hipMalloc(&y, 16L * plen * 16); hipMemset(x, 0, 16L * plen * 16);
  hipMalloc(&w, 4L * 2 * 9 * 2 * 64 * 16); hipMemset(w, 0, 4L * 2 * 9 * 2 * 64 * 16);
  hipMalloc(&bias, 256); hipMemset(bias, 0, 256); hipMalloc(&flag, 4);
  a.x = x; a.w = w; a.bias = bias; a.y = y; a.r1 = nullptr; a.x_plen = plen; a.y_plen = plen; a.n = n; a.tiles_per_row = (n + 511) / 512;
  a.nchunks = 4; a.cout = 64; a.relu = 1; a.out_mode = 0; a.flag = flag; a.stamps = nullptr;
  unsigned char* codes; float *tab, *b1;
  hipMalloc(&codes, n); hipMalloc(&tab, 9 * 6 * 64 * 4); hipMalloc(&b1, 256);
  std::vector<unsigned char> hc(n); unsigned sd = 99u;
  for (auto& c : hc) { sd = sd * 1664525u + 1013904223u; c = (sd >> 24) & 3; }
  std::vector<float> ht(9 * 6 * 64);
  for (auto& v : ht) { sd = sd * 1664525u + 1013904223u; v = ((sd >> 8) & 0xffff) / 65536.f - 0.5f; }
  hipMemcpy(codes, hc.data(), n, hipMemcpyHostToDevice); hipMemcpy(tab, ht.data(), ht.size() * 4, hipMemcpyHostToDevice); hipMemset(b1, 0, 256);
  ConvP16Args f = a; f.f1_codes = codes; f.f1_codes_L = n; f.f1_codes_off = 0; f.f1_reverse = 0; f.f1_table = tab; f.f1_bias = b1;
  for (int rep = 0; rep < 2; ++rep) {
    run<0, false>(a, "plain");
    run<256, false>(a, "plain, no X DMA after the first step");
    run<0, true>(f, "fused");
    run<1024, true>(f, "fused, no halo tail");
    run<2048, true>(f, "fused, no image writes");
    run<4096, true>(f, "fused, no table reads");
    run<4096 + 2048, true>(f, "fused, no table reads, no image writes");
    run<4096 + 2048 + 1024, true>(f, "fused, no reads, no writes, no tail");
    run<8192, true>(f, "fused, no in-loop producer at all");
  }
  return 0;
}
