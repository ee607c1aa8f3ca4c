// Stand-alone ablation micro-benchmark of conv1d_k9_bf16s_kernel (compile-time masks).
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -I orca_amd/csrc tools/microbench_b16.hip -o /tmp/mb && /tmp/mb
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "conv_bf16s.h"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int COUT, int MW, int NW, int WM, int WN, int NS, int DT, int ABL>
static float run(ConvB16Args a, int B, int reps) {
  constexpr int MT = WM * MW * 32;
  int per_cu = 1;
  (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, conv1d_k9_bf16s_kernel<COUT, MW, NW, WM, WN, NS, DT, ABL>, WM * WN * 64, 0);
  a.tiles_per_row = (a.n + MT - 1) / MT; a.batch = B;
  long ntiles = a.tiles_per_row * B;
  long grid = 256L * per_cu; if (grid > ntiles) grid = ntiles;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e9;
  for (int r = 0; r < reps + 1; ++r) {
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((conv1d_k9_bf16s_kernel<COUT, MW, NW, WM, WN, NS, DT, ABL>), dim3((unsigned)grid), dim3(WM * WN * 64), 0, 0, a);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); if (r > 0 && ms < best) best = ms;
  }
  double fl = 2.0 * 9 * a.cin * COUT * (double)a.n * B;
  printf("cout=%d cin=%d n=%ld NS=%d ABL=%2d occ/CU=%d grid=%ld : %.3f ms  %.1f TFLOP/s-eq\n", COUT, a.cin, a.n, NS, ABL, per_cu, grid, best, fl / best / 1e9);
  return best;
}

int main(int argc, char** argv) {
  long n = argc > 1 ? atol(argv[1]) : 8000000;
  const int cin = 64, cout = 64;
  std::vector<float> hx((size_t)n * cin);
  unsigned s = 12345; for (auto& v : hx) { s = s * 1664525u + 1013904223u; v = ((s >> 8) & 0xffff) / 65536.f - 0.5f; }
  float *x, *y, *bias; void* w;
  CK(hipMalloc(&x, hx.size() * 4)); CK(hipMalloc(&y, (size_t)n * 128 * 4)); CK(hipMalloc(&bias, 128 * 4));
  size_t wbytes = (size_t)(128 / 16) * 3 * 9 * 2 * 128 * 8 * 2; CK(hipMalloc(&w, wbytes));
  CK(hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice)); CK(hipMemset(w, 0x3c, wbytes)); CK(hipMemset(bias, 0, 512));
  ConvB16Args a{}; a.x = x; a.w = w; a.bias = bias; a.y = y; a.r1 = nullptr; a.x_bs = 0; a.y_bs = 0; a.n = n; a.cin = cin; a.nchunks = cin / 16; a.relu = 1;
  run<64, 2, 2, 4, 1, 3, 0, 0>(a, 1, 3); run<64, 2, 2, 4, 1, 2, 1, 0>(a, 1, 3);
  for (int st = 1; st <= 4; ++st) { a.stagger = st; printf("stagger %d (upper half): ", st); run<64, 2, 2, 4, 1, 3, 0, 32>(a, 1, 3); }
  for (int st = 1; st <= 4; ++st) { a.stagger = st; printf("stagger %d (odd): ", st); run<64, 2, 2, 4, 1, 2, 1, 32>(a, 1, 3); }
  return 0;
}
