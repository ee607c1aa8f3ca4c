// Ablation + per-wave stamp micro-benchmark of conv1d_k9_p16_kernel in the B16 format (FMT = 1: single bf16 plane, one
// product, 32 input channels per step).  hipcc --offload-arch=gfx950 -O3 -std=c++17 -I orca_amd/csrc -I include ...
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#include "conv_p16.h"
#include "conv_ws.h"
#ifndef BFMT
#define BFMT 1
#endif
template <int CT, int MW, int NW, int WM, int OM, bool R1, int ABL>
static void run(ConvP16Args a, const char* what) {
  constexpr int MT = WM * MW * 32;
  int per_cu = 1;
  (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, conv1d_k9_p16_kernel<CT, MW, NW, WM, OM, R1, ABL, false, BFMT>, WM * 64, 0);
  a.tiles_per_row = (a.n + MT - 1) / MT; a.out_mode = OM;
  long ntiles = a.tiles_per_row * (a.cout / CT), grid = 256L * per_cu; if (grid > ntiles) grid = ntiles;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e9;
  for (int r = 0; r < 4; ++r) {
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((conv1d_k9_p16_kernel<CT, MW, NW, WM, OM, R1, ABL, false, BFMT>), dim3((unsigned)grid), dim3(WM * 64), 0, 0, a);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); if (r > 0 && ms < best) best = ms;
  }
  const int kc = BFMT ? 32 : 16;
  double fl = 2.0 * 9 * a.nchunks * kc * a.cout * (double)a.n;
  printf("FMT=%d CT=%d MT=%d cin=%d cout=%d n=%ld OM=%d R1=%d ABL=%3d (%s): %.3f ms  %.1f TFLOP/s  [%s]\n", BFMT, CT, MT, a.nchunks * kc, a.cout, a.n, OM, (int)R1, ABL, what, best, fl / best / 1e9, hipGetErrorString(hipGetLastError()));
}
template <int CT, int MW, int NW, int OM, bool R1, int ABL>
static void run_ws(ConvP16Args a, const char* what) {
  a.out_mode = OM;
  const int ncb = a.cout / CT;
  int grid = (256 / (8 * ncb)) * (8 * ncb);
  if (grid < 248) grid = (256 / ncb) * ncb;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e9;
  for (int r = 0; r < 4; ++r) {
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((conv1d_k9_ws_kernel<BFMT, 64, CT, MW, NW, OM, R1, ABL>), dim3((unsigned)grid), dim3(512), 0, 0, a);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); if (r > 0 && ms < best) best = ms;
  }
  double fl = 2.0 * 9 * 64 * a.cout * (double)a.n;
  printf("WS FMT=%d CT=%d cin=64 cout=%d n=%ld OM=%d R1=%d ABL=%3d (%s): %.3f ms  %.1f TFLOP/s  [%s]\n", BFMT, CT, a.cout, a.n, OM, (int)R1, ABL, what, best, fl / best / 1e9, hipGetErrorString(hipGetLastError()));
}
static void report_ws(unsigned long long* st) {
  std::vector<unsigned long long> h(8192); hipMemcpy(h.data(), st, 8192 * 8, hipMemcpyDeviceToHost);
  printf("  shader clock during the kernel: %.0f MHz (wave 0 of workgroup 0 ran %.3f ms)\n", (double)h[8190] / ((double)h[8191] * 0.01), (double)h[8191] * 1e-5);
  // per wave: step = [vmcnt wait | epilogue (every NCH-th step) | MFMA block]
  for (int w = 0; w < 8; ++w) {
    double len[2] = {0, 0}, wait[2] = {0, 0}, epi[2] = {0, 0}, blk[2] = {0, 0}; long cnt[2] = {0, 0};
    for (int s_ = 0; s_ + 1 < 199; ++s_) {
      const unsigned long long t0 = h[(s_ * 8 + w) * 5], t1 = h[(s_ * 8 + w) * 5 + 1], t2 = h[(s_ * 8 + w) * 5 + 2], t3 = h[(s_ * 8 + w) * 5 + 3], tn = h[((s_ + 1) * 8 + w) * 5];
      if (!t0 || !tn) continue;
      const int k = (t1 - t2) > 300;
      len[k] += (double)(tn - t0); wait[k] += (double)(t2 - t0); epi[k] += (double)(t1 - t2); blk[k] += (double)(t3 - t1); ++cnt[k];
    }
    for (int k = 0; k < 2; ++k) if (cnt[k]) printf("    wave %d %s step (n=%ld): length %6.0f = vmcnt wait %5.0f + epilogue/prefetch %5.0f + MFMA block %5.0f\n", w, k ? "epilogue" : "plain   ", cnt[k], len[k] / cnt[k], wait[k] / cnt[k], epi[k] / cnt[k], blk[k] / cnt[k]);
  }
}
static void report(unsigned long long* st) {
  std::vector<unsigned long long> h(8192); hipMemcpy(h.data(), st, 8192 * 8, hipMemcpyDeviceToHost);
  printf("  shader clock during the kernel: %.0f MHz\n", (double)h[8190] / ((double)h[8191] * 0.01));
  for (int w = 0; w < 2; ++w) { printf("  step 250, wave %d: cycles per tap:", w * 4); for (int t = 1; t < 9; ++t) printf(" %llu", h[8100 + w * 16 + t] - h[8100 + w * 16 + t - 1]); printf("\n"); }
  for (int kind = 0; kind < 2; ++kind) {
    double sum[8][5] = {}; long cnt = 0; double steplen = 0;
    for (int st_ = 0; st_ + 1 < 200; ++st_) {
      unsigned long long t0 = ~0ull, t0n = ~0ull;
      for (int w = 0; w < 8; ++w) { t0 = std::min(t0, h[(st_ * 8 + w) * 5]); t0n = std::min(t0n, h[((st_ + 1) * 8 + w) * 5]); }
      const bool epi = (h[(st_ * 8) * 5 + 1] - h[(st_ * 8) * 5]) > 300;
      if ((int)epi != kind) continue;
      ++cnt; steplen += (double)(t0n - t0);
      for (int w = 0; w < 8; ++w) for (int k = 0; k < 5; ++k) if (k != 2) sum[w][k] += (double)(h[(st_ * 8 + w) * 5 + k] - t0);
    }
    printf("  %s steps (n=%ld), avg length %.0f cycles; per wave: start | epilogue done | MFMA block done | vmcnt(0)\n", kind ? "epilogue" : "plain", cnt, cnt ? steplen / cnt : 0.0);
    for (int w = 0; w < 8 && cnt; ++w) printf("    wave %d: %6.0f %6.0f %6.0f %6.0f\n", w, sum[w][0] / cnt, sum[w][1] / cnt, sum[w][3] / cnt, sum[w][4] / cnt);
  }
}
int main(int argc, char** argv) {
  long n = argc > 1 ? atol(argv[1]) : 32000000;
  const long plen = ((n + 512) / 512) * 512 + 8;
  f32x4 *x, *y, *w; float* bias;
  const size_t xunits = (size_t)(BFMT ? 16 : 32) * plen;   // 128 channels
  hipMalloc(&x, xunits * 16); hipMalloc(&y, xunits * 16); hipMalloc(&w, (size_t)16 * 2 * 9 * 2 * 128 * 16); hipMalloc(&bias, 512);
  {   // random 16-bit content, |v| in [0.125, 2): realistic switching activity
    std::vector<unsigned short> hx(xunits * 8);
    unsigned s = 1234567u;
    for (auto& v : hx) { s = s * 1664525u + 1013904223u; v = BFMT ? (unsigned short)(((s >> 9) & 0x81ff) | 0x3e00) : (unsigned short)(((s >> 9) & 0x8fff) | 0x3000); }
    hipMemcpy(x, hx.data(), hx.size() * 2, hipMemcpyHostToDevice);
    std::vector<unsigned short> hw((size_t)16 * 2 * 9 * 2 * 128 * 8);
    for (auto& v : hw) { s = s * 1664525u + 1013904223u; v = BFMT ? (unsigned short)(((s >> 9) & 0x80ff) | 0x3c00) : (unsigned short)(((s >> 9) & 0x83ff) | 0x2400); }
    hipMemcpy(w, hw.data(), hw.size() * 2, hipMemcpyHostToDevice);
  }
  hipMemset(bias, 0, 512);
  const int CH = BFMT ? 2 : 4;   // chunks of a 64-channel input
  ConvP16Args a{}; a.x = x; a.w = w; a.bias = bias; a.y = y; a.r1 = nullptr; a.x_plen = plen; a.y_plen = plen; a.n = n; a.nchunks = CH; a.cout = 64; a.relu = 1; a.out_mode = 0; a.flag = nullptr;
  unsigned long long* st; hipMalloc(&st, 8192 * 8); hipMemset(st, 0, 8192 * 8); a.stamps = st;
  run<64, 2, 2, 8, 0, false, 0>(a, "warm");
  run<64, 2, 2, 8, 0, false, 0>(a, "plain");
  run<64, 2, 2, 8, 0, false, 16>(a, "no stores");
  run<64, 2, 2, 8, 0, false, 1>(a, "no DMA after the first step");
  run<64, 2, 2, 8, 0, false, 64>(a, "no W DMA after the first step");
  run<64, 2, 2, 8, 0, false, 256>(a, "no X DMA after the first step");
  run<64, 2, 2, 8, 0, false, 1 + 16>(a, "no DMA, no stores");
  run<64, 2, 2, 8, 0, false, 1 + 16 + 8>(a, "no DMA, no stores, LDS once");
  run<64, 2, 2, 8, 0, false, 0>(a, "plain");
  a.r1 = x;
  run<64, 2, 2, 8, 0, true, 0>(a, "r1");
  run<64, 2, 2, 8, 1, true, 0>(a, "r1 pool");
  a.r1 = nullptr;
  { ConvP16Args b = a; b.cout = 96; b.nchunks = CH * 3 / 2; b.n = n / 4; run<96, 1, 3, 8, 0, false, 0>(b, "96 -> 96, n/4"); }
  { ConvP16Args b = a; b.cout = 128; b.nchunks = CH * 2; b.n = n / 16; run<64, 2, 2, 8, 0, false, 0>(b, "128 -> 128, n/16"); }
  run<64, 2, 2, 8, 0, false, 128>(a, "stamped");
  report(st);
  {
    constexpr int WCT = BFMT ? 64 : 32, WNW = BFMT ? 2 : 1;
    run_ws<WCT, 2, WNW, 0, false, 0>(a, "ws plain");
    run_ws<WCT, 2, WNW, 0, false, 0>(a, "ws plain");
    run_ws<WCT, 2, WNW, 0, false, 16>(a, "ws no stores");
    run_ws<WCT, 2, WNW, 0, false, 1>(a, "ws no DMA after the first step");
    run_ws<WCT, 2, WNW, 0, false, 17>(a, "ws no DMA no stores");
    run_ws<WCT, 2, WNW, 0, false, 4>(a, "ws no MFMA (data movement only)");
    run_ws<WCT, 2, WNW, 0, false, 4 + 16>(a, "ws no MFMA, no stores");
    run_ws<WCT, 2, WNW, 0, false, 4 + 1>(a, "ws no MFMA, no DMA");
    if (BFMT) {
      run_ws<WCT, 2, WNW, 0, false, 4 + 1 + 32>(a, "ws no MFMA, no DMA, tile-major stores");
      run_ws<WCT, 2, WNW, 0, false, 32>(a, "ws tile-major stores");
    }
    a.r1 = x;
    run_ws<WCT, 2, WNW, 0, true, 0>(a, "ws r1");
    run_ws<WCT, 2, WNW, 1, true, 0>(a, "ws r1 pool");
    a.r1 = nullptr;
    if (!BFMT) { ConvP16Args b = a; b.cout = 96; b.n = n / 4; run_ws<32, 2, 1, 0, false, 0>(b, "ws 64 -> 96, n/4"); run<96, 1, 3, 8, 0, false, 0>(b, "64 -> 96, n/4"); }
    hipMemset(st, 0, 8192 * 8);
    run_ws<WCT, 2, WNW, 0, false, 128>(a, "ws stamped");
    report_ws(st);
    hipMemset(st, 0, 8192 * 8);
    run_ws<WCT, 2, WNW, 0, false, 128 + 1 + 16>(a, "ws stamped, no DMA, no stores");
    report_ws(st);
  }
  run<64, 2, 2, 8, 0, false, 128 + 1 + 16>(a, "stamped, no DMA, no stores");
  report(st);
  return 0;
}
