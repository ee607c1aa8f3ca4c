// Timeline + ablation micro-benchmark of conv2d_3x3_m16q_kernel (one Decoder layer per launch, dependent chain as in a Decoder).
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -w -DM16Q_STAMPS=100 [-DM16Q_ABL=..] -I orca_amd/csrc -I include tools/microbench_m16q.hip -o tools/microbench_m16q
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#include "conv_kernels.h"
#include "conv_bf16s.h"
#include "conv_p16.h"
#include "conv2d_m16q.h"
template <int COUT>
static void run(ConvM16QArgs a, ConvM16QArgs b2, int B, int gy, const char* what) {
  a.nb = b2.nb = B;       // gy < B: a workgroup walks the maps b, b + gy, ... of its tile (one resident round)
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e9;
#ifdef M16Q_TWO_QUEUES   // timing only: consecutive launches on alternating streams with NO dependency between them - the upper bound of what
                         // overlapping a layer's ramp with its predecessor's tail could give (results are wrong: races)
  static hipStream_t qs[2] = {nullptr, nullptr};
  if (!qs[0]) { hipStreamCreateWithFlags(&qs[0], hipStreamNonBlocking); hipStreamCreateWithFlags(&qs[1], hipStreamNonBlocking); }
  for (int r = 0; r < 6; ++r) {
    hipDeviceSynchronize();
    hipEventRecord(e0, qs[0]);
    for (int k = 0; k < 20; ++k)
      hipLaunchKernelGGL((conv2d_3x3_m16q_kernel<COUT, 2, 1>), dim3((a.ngroups * 2 + 7) / 8 * 8, gy), dim3(512), 0, qs[k & 1], (k & 1) ? b2 : a);
    hipStreamSynchronize(qs[1]);
    hipEventRecord(e1, qs[0]); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms *= 0.5f; if (r > 0 && ms < best) best = ms;
  }
#else
  for (int r = 0; r < 6; ++r) {
    hipEventRecord(e0, 0);
    for (int k = 0; k < 10; ++k)       // ping-pong: every launch reads what the previous one wrote
      hipLaunchKernelGGL((conv2d_3x3_m16q_kernel<COUT, 2, 1>), dim3((a.ngroups * 2 + 7) / 8 * 8, gy), dim3(512), 0, 0, (k & 1) ? b2 : a);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); if (r > 0 && ms < best) best = ms;
  }
#endif
  printf("cout=%d cin=%d d=%d B=%d grid.y=%d res=%d (%s): %.2f us per launch  [%s]\n", COUT, a.c.nchunks * 16, a.c.dil, B, gy, a.c.r != nullptr, what, best * 100.f, hipGetErrorString(hipGetLastError()));
#ifdef M16Q_STAMPS
  unsigned long long h[128]; hipMemcpyFromSymbol(h, HIP_SYMBOL(m16q_stamp_buf), sizeof h);
  unsigned long long t0 = ~0ull; for (int w = 0; w < 8; ++w) t0 = std::min(t0, h[w * 16]);
  const int np = a.c.nchunks * (COUT / 32);
  printf("   wave: start | DMA issued |");
  for (int i = 0; i < np; ++i) printf(" p%d landed, barrier |", i);
  printf(" MFMAs issued | stores issued   (shader cycles from the workgroup's first stamp)\n");
  for (int w = 0; w < 8; w += 3) {
    printf("   %d: %5llu %5llu |", w, h[w * 16] - t0, h[w * 16 + 1] - t0);
    for (int i = 0; i < np; ++i) printf(" %5llu %5llu |", h[w * 16 + 2 + 2 * i] - t0, h[w * 16 + 3 + 2 * i] - t0);
    printf(" %5llu | %5llu   [%.2f us by the 100 MHz clock -> %.2f GHz]\n", h[w * 16 + 10] - t0, h[w * 16 + 11] - t0, (h[w * 16 + 13] - h[w * 16 + 12]) * 0.01, (h[w * 16 + 11] - h[w * 16]) / ((h[w * 16 + 13] - h[w * 16 + 12]) * 10.0));
  }
#endif
}
int main() {
  const int n = 250, BMAX = 8;
  const size_t map = (size_t)8 * 2 * n * 256;           // units of a 64-channel f16x2 M16 map
  f32x4 *m0, *m1, *m2, *zero; hipMalloc(&m0, BMAX * map * 16); hipMalloc(&m1, BMAX * map * 16); hipMalloc(&m2, BMAX * map * 16); hipMalloc(&zero, 256); hipMemset(zero, 0, 256);
  std::vector<unsigned short> h(map * 8); unsigned s = 12345u;
  for (auto& v : h) { s = s * 1664525u + 1013904223u; v = (unsigned short)(((s >> 9) & 0x83ff) | 0x2800); }
  for (int b = 0; b < BMAX; ++b) { hipMemcpy(m0 + b * map, h.data(), h.size() * 2, hipMemcpyHostToDevice); hipMemcpy(m1 + b * map, h.data(), h.size() * 2, hipMemcpyHostToDevice); hipMemcpy(m2 + b * map, h.data(), h.size() * 2, hipMemcpyHostToDevice); }
  void* w; float* bias; const size_t wunits = (size_t)4 * 2 * 9 * 2 * 64;
  hipMalloc(&w, wunits * 16); hipMalloc(&bias, 256); hipMemset(bias, 0, 256);
  std::vector<unsigned short> hw(wunits * 8);
  for (auto& v : hw) { s = s * 1664525u + 1013904223u; v = (unsigned short)(((s >> 9) & 0x83ff) | 0x1c00); }
  hipMemcpy(w, hw.data(), hw.size() * 2, hipMemcpyHostToDevice);
  ConvM16QArgs a{}; a.c.w = w; a.c.bias = bias; a.c.x_bs = a.c.y_bs = a.c.r_bs = (long)map; a.c.H = n; a.c.W = n; a.c.relu = 0; a.c.flag = nullptr; a.zero = zero;
  for (int B : {1, 2, 4, 8})
   for (int gy : {B, 2})
    if (gy <= B && (gy == B || B > 2))
    for (int d : {1, 8}) {
      a.c.dil = d; a.ngroups = ((n + 4 * d - 1) / (4 * d)) * d;
      { ConvM16QArgs x = a, y = a; x.c.nchunks = y.c.nchunks = 4; x.c.x = m0; x.c.y = m1; y.c.x = m1; y.c.y = m0; run<32>(x, y, B, gy, "64 -> 32"); }
      { ConvM16QArgs x = a, y = a; x.c.nchunks = y.c.nchunks = 2; x.c.x = m0; x.c.y = m1; x.c.r = m2; y.c.x = m1; y.c.y = m0; y.c.r = m2; run<64>(x, y, B, gy, "32 -> 64 + residual"); }
      { ConvM16QArgs x = a, y = a; x.c.nchunks = y.c.nchunks = 2; x.c.x = m0; x.c.y = m1; x.c.r = m2; y.c.x = m1; y.c.y = m0; y.c.r = m2; x.c.relu = y.c.relu = 1; run<64>(x, y, B, gy, "32 -> 64, ReLU, + residual"); }
    }
  return 0;
}
