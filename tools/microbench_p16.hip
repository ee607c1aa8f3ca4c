// Ablation + per-wave s_memtime stamp micro-benchmark of conv1d_k9_p16_kernel.  hipcc --offload-arch=gfx950 -O3 -std=c++17 -I orca_amd/csrc ...
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#include "conv_p16.h"
template <int CT, int MW, int NW, int WM, int OM, bool R1, int ABL, bool F1 = false>
static void run(ConvP16Args a, const char* what) {
  constexpr int MT = WM * MW * 32;
  int per_cu = 1;
  (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, conv1d_k9_p16_kernel<CT, MW, NW, WM, OM, R1, ABL, F1>, WM * 64, 0);
  a.tiles_per_row = (a.n + MT - 1) / MT; a.out_mode = OM;
  long ntiles = a.tiles_per_row * (a.cout / CT), grid = 256L * per_cu; if (grid > ntiles) grid = ntiles;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e9;
  for (int r = 0; r < 4; ++r) {
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((conv1d_k9_p16_kernel<CT, MW, NW, WM, OM, R1, ABL, F1>), dim3((unsigned)grid), dim3(WM * 64), 0, 0, a);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); if (r > 0 && ms < best) best = ms;
  }
  double fl = 2.0 * 9 * a.nchunks * 16 * a.cout * (double)a.n;
  printf("CT=%d MT=%d cin=%d cout=%d n=%ld OM=%d R1=%d ABL=%3d (%s) occ=%d: %.3f ms  %.1f TFLOP/s-eq  [%s]\n", CT, MT, a.nchunks * 16, a.cout, a.n, OM, (int)R1, ABL, what, per_cu, best, fl / best / 1e9, hipGetErrorString(hipGetLastError()));
}
static void report(unsigned long long* st) {
  std::vector<unsigned long long> h(8192); hipMemcpy(h.data(), st, 8192 * 8, hipMemcpyDeviceToHost);
  printf("  shader clock during the kernel: %.0f MHz\n", (double)h[8190] / ((double)h[8191] * 0.01));
  for (int w = 0; w < 2; ++w) { printf("  step 250, wave %d: cycles per tap:", w * 4); for (int t = 1; t < 9; ++t) printf(" %llu", h[8100 + w * 16 + t] - h[8100 + w * 16 + t - 1]); printf("\n"); }
  // per wave, relative to the step's earliest start: averages over plain steps and over epilogue steps
  for (int kind = 0; kind < 2; ++kind) {
    double sum[8][5] = {}; long cnt = 0; double steplen = 0;
    for (int st_ = 0; st_ + 1 < 200; ++st_) {
      unsigned long long t0 = ~0ull, t0n = ~0ull;
      for (int w = 0; w < 8; ++w) { t0 = std::min(t0, h[(st_ * 8 + w) * 5]); t0n = std::min(t0n, h[((st_ + 1) * 8 + w) * 5]); }
      const bool epi = (h[(st_ * 8) * 5 + 1] - h[(st_ * 8) * 5]) > 400;
      if ((int)epi != kind) continue;
      ++cnt; steplen += (double)(t0n - t0);
      for (int w = 0; w < 8; ++w) for (int k = 0; k < 5; ++k) if (k != 2) sum[w][k] += (double)(h[(st_ * 8 + w) * 5 + k] - t0);
    }
    printf("  %s steps (n=%ld), avg length %.0f cycles; per wave: start | epilogue done | MFMA block done | vmcnt(0)\n", kind ? "epilogue" : "plain", cnt, cnt ? steplen / cnt : 0.0);
    for (int w = 0; w < 8 && cnt; ++w) printf("    wave %d: %6.0f %6.0f %6.0f %6.0f\n", w, sum[w][0] / cnt, sum[w][1] / cnt, sum[w][3] / cnt, sum[w][4] / cnt);
  }
}
int main(int argc, char** argv) {
  long n = argc > 1 ? atol(argv[1]) : 8000000;
  const long plen = ((n + 512) / 512) * 512 + 8;
  f32x4 *x, *y, *w; float* bias;
  hipMalloc(&x, (size_t)256 * plen * 4); hipMalloc(&y, (size_t)128 * plen * 4); hipMalloc(&w, (size_t)16 * 2 * 9 * 2 * 128 * 16); hipMalloc(&bias, 512);
  if (argc > 2) {   // random fp16 content (hi ~ U(-1,1), lo tiny) instead of a constant fill: realistic switching activity
    std::vector<unsigned short> hx((size_t)256 * plen * 2);   // every channel random: constant planes run measurably faster (less switching power)
    unsigned s = 1234567u;
    for (auto& v : hx) { s = s * 1664525u + 1013904223u; v = (unsigned short)(((s >> 9) & 0x8fff) | 0x3000); }
    hipMemcpy(x, hx.data(), hx.size() * 2, hipMemcpyHostToDevice);
    printf("random activations\n");
  } else
  hipMemset(x, 0x2c, (size_t)128 * plen * 4); hipMemset(w, 0x2c, (size_t)16 * 2 * 9 * 2 * 128 * 16); hipMemset(bias, 0, 512);
  ConvP16Args a{}; a.x = x; a.w = w; a.bias = bias; a.y = y; a.r1 = nullptr; a.x_plen = plen; a.y_plen = plen; a.n = n; a.nchunks = 4; a.cout = 64; a.relu = 1; a.out_mode = 0; a.flag = nullptr;
  unsigned long long* st; hipMalloc(&st, 8192 * 8); hipMemset(st, 0, 8192 * 8); a.stamps = st;
  a.out_mode = 0;
  run<64, 2, 2, 8, 0, false, 0>(a, "warm");
  run<64, 2, 2, 8, 0, false, 0>(a, "plain");
  run<64, 2, 2, 8, 0, false, 128>(a, "plain, stamped"); report(st);
  run<64, 2, 2, 8, 0, false, 16384>(a, "6 of 9 taps' MFMAs");
  run<64, 2, 2, 8, 0, false, 16384 + 128>(a, "6 of 9 taps' MFMAs, stamped"); report(st);
  if (argc > 3) return 0;
  run<64, 2, 2, 8, 0, false, 512>(a, "epilogues of waves 4-7 half a tile late (timing only)");
  run<64, 2, 2, 8, 0, false, 0>(a, "plain");
  run<64, 2, 2, 8, 0, false, 512>(a, "epilogues of waves 4-7 half a tile late (timing only)");
  run<64, 2, 2, 8, 0, false, 16>(a, "no stores");
  run<64, 2, 2, 8, 0, false, 32>(a, "stores folded into an L2-resident window");
  run<64, 2, 2, 8, 0, false, 1>(a, "no DMA after the first step");
  run<64, 2, 2, 8, 0, false, 64>(a, "no W DMA after the first step");
  run<64, 2, 2, 8, 0, false, 256>(a, "no X DMA after the first step");
  run<64, 2, 2, 8, 0, false, 1 + 16>(a, "no DMA, no stores");
  a.r1 = x;
  run<64, 2, 2, 8, 0, true, 0>(a, "r1");
  run<64, 2, 2, 8, 1, true, 0>(a, "r1 pool");
  run<64, 2, 2, 8, 2, true, 0>(a, "r1 f32 out");
  a.r1 = nullptr;
  { ConvP16Args b = a; b.cout = 96; b.nchunks = 6; run<96, 1, 3, 8, 0, false, 0>(b, "96"); }
  { ConvP16Args b = a; b.cout = 128; b.nchunks = 8; run<64, 2, 2, 8, 0, false, 0>(b, "128 -> 128"); }
  { ConvP16Args b = a; b.cout = 64; b.nchunks = 8; run<64, 2, 2, 8, 0, false, 0>(b, "128 -> 64"); }
  { ConvP16Args b = a; b.cout = 64; b.nchunks = 16; run<64, 2, 2, 8, 0, false, 0>(b, "256 -> 64"); }
  { ConvP16Args b = a; b.cout = 64; b.nchunks = 2; run<64, 2, 2, 8, 0, false, 0>(b, "32 -> 64"); }
  { ConvP16Args b = a; b.cout = 128; b.nchunks = 4; run<64, 2, 2, 8, 0, false, 0>(b, "64 -> 128"); }
  run<64, 2, 2, 8, 0, false, 128>(a, "stamped");
  report(st);
  {   // fused first layer: the input image is produced from packed bases
    unsigned char* codes; float *tab, *b1;
    hipMalloc(&codes, n); hipMalloc(&tab, 9 * 6 * 64 * 4); hipMalloc(&b1, 256);
    std::vector<unsigned char> hc(n); unsigned sd = 99u;
    for (auto& c : hc) { sd = sd * 1664525u + 1013904223u; c = (sd >> 24) & 3; }
    std::vector<float> ht(9 * 6 * 64);
    for (auto& v : ht) { sd = sd * 1664525u + 1013904223u; v = ((sd >> 8) & 0xffff) / 65536.f - 0.5f; }
    hipMemcpy(codes, hc.data(), n, hipMemcpyHostToDevice); hipMemcpy(tab, ht.data(), ht.size() * 4, hipMemcpyHostToDevice); hipMemset(b1, 0, 256);
    ConvP16Args f = a; f.f1_codes = codes; f.f1_codes_L = n; f.f1_codes_off = 0; f.f1_reverse = 0; f.f1_table = tab; f.f1_bias = b1;
    run<64, 2, 2, 8, 0, false, 0, true>(f, "fused first layer");
    run<64, 2, 2, 8, 0, false, 0, false>(a, "plain");
    run<64, 2, 2, 8, 0, false, 0, true>(f, "fused first layer");
    run<64, 2, 2, 8, 0, false, 128, true>(f, "fused first layer, stamped");
    report(st);
  }
  run<64, 2, 2, 8, 0, false, 128 + 1 + 16>(a, "stamped, no DMA, no stores");
  report(st);
  run<64, 2, 2, 8, 0, false, 128 + 1 + 16 + 8>(a, "stamped, no DMA, no stores, LDS once");
  report(st);
  return 0;
}
