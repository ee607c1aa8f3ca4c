// Host cost of a kernel launch on this platform (what bounds paths made of many short launches: the SV screen's local encodes, 650
// launches per variant).  hipcc --offload-arch=gfx950 -O3 -o tools/microbench_launch tools/microbench_launch.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
struct Big { long a[28]; };   // 224 bytes of kernel arguments (ConvP16Args is about that size)
__global__ void k_empty() {}
__global__ void k_big(Big b, float* p) { if (b.a[0] == 12345 && p) p[0] = 1.f; }
__global__ void k_work(float* p, int n) { for (int i = threadIdx.x + blockIdx.x * blockDim.x; i < n; i += blockDim.x * gridDim.x) p[i] = p[i] * 1.0001f + 1.f; }
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
  hipStream_t s;
  hipStreamCreate(&s);
  float* d;
  hipMalloc(&d, 64 << 20);
  const int N = 2000;
  Big b{};
  for (int mode = 0; mode < 6; ++mode) {
    for (int rep = 0; rep < 2; ++rep) {
      hipStreamSynchronize(s);
      const double t0 = now();
      for (int i = 0; i < N; ++i) {
        switch (mode) {
          case 0: hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, s); break;
          case 1: hipLaunchKernelGGL(k_big, dim3(1), dim3(64), 0, s, b, d); break;
          case 2: hipLaunchKernelGGL(k_big, dim3(256), dim3(512), 0, s, b, d); (void)hipGetLastError(); break;
          case 3: hipLaunchKernelGGL(k_big, dim3(256), dim3(512), 0, s, b, d); (void)hipGetLastError(); for (int j = 0; j < 4; ++j) if (getenv("ORCA_NO_SUCH_SWITCH")) b.a[1]++; break;
          case 4: hipLaunchKernelGGL(k_work, dim3(256), dim3(256), 0, s, d, 1 << 20); break;      // ~4 MB in + out: a few us of GPU time
          default: hipLaunchKernelGGL(k_work, dim3(1024), dim3(256), 0, s, d, 16 << 20); break;    // 64 MB: ~25 us of GPU time (GPU-bound)
        }
      }
      const double t1 = now();
      hipStreamSynchronize(s);
      const double t2 = now();
      if (rep) printf("mode %d: host %.2f us per launch, all done after %.2f us per launch\n", mode, (t1 - t0) / N * 1e6, (t2 - t0) / N * 1e6);
    }
  }
  return 0;
}
