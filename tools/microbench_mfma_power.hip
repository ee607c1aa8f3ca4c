// Sustained f16 MFMA rate under the package power cap, registers only (no LDS, no memory): v_mfma_f32_32x32x16_f16 against
// v_mfma_f32_16x16x32_f16 (half the accumulator traffic per FLOP, twice the operand traffic), random and half-zero operands.
// hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/microbench_mfma_power.hip -o tools/microbench_mfma_power
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f16x8 mk(unsigned seed, int zero_every) {
  f16x8 v;
  for (int i = 0; i < 8; ++i) {
    seed = seed * 1664525u + 1013904223u;
    const float f = (float)((seed >> 9) & 0xffff) * (1.f / 65536.f) - 0.5f;
    v[i] = (zero_every && ((seed >> 27) % zero_every) == 0) ? (_Float16)0.f : (_Float16)f;
  }
  return v;
}

template <int KIND, int LDS>
__global__ __launch_bounds__(512) void k(int iters, int zero_every, float* sink) {
  __shared__ f16x8 frag[LDS ? 4096 : 1];   // 64 KB: operand fragments re-read at the rate of the real kernels (10 x 1 KB per 18 / 36 MFMAs)
  const unsigned t = blockIdx.x * 512 + threadIdx.x;
  f16x8 a[4], b[4];
  for (int i = 0; i < 4; ++i) { a[i] = mk(t * 8 + i, zero_every); b[i] = mk(t * 8 + 4 + i, zero_every); }
  if (LDS) {
    for (int i = threadIdx.x; i < 4096; i += 512) frag[i] = mk(t * 64 + i, zero_every);
    __syncthreads();
  }
  float out = 0.f;
  if (KIND == 0) {
    f32x16 acc[6];
    for (int i = 0; i < 6; ++i) acc[i] = (f32x16)(0.f);
    for (int it = 0; it < iters; ++it) {
      if (LDS) {   // 24 MFMAs of 32 KFLOP per iteration: 13 fragment reads (0.56 per MFMA)
#pragma unroll
        for (int i = 0; i < 4; ++i) { a[i] = frag[(threadIdx.x + 64 * ((it + i) & 31)) & 4095]; b[i] = frag[(threadIdx.x + 64 * ((it + i + 7) & 31) + 2048) & 4095]; }
#pragma unroll
        for (int i = 0; i < 3; ++i) a[i] = frag[(threadIdx.x + 64 * ((it + i + 3) & 31) + 1024) & 4095];
#pragma unroll
        for (int i = 0; i < 2; ++i) b[i] = frag[(threadIdx.x + 64 * ((it + i + 11) & 31) + 3072) & 4095];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int i = 0; i < 6; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[(i + u) & 3], b[u], acc[i], 0, 0, 0);
    }
    for (int i = 0; i < 6; ++i) out += acc[i][0] + acc[i][7];
  } else {
    f32x4 acc[12];
    for (int i = 0; i < 12; ++i) acc[i] = (f32x4)(0.f);
    for (int it = 0; it < iters; ++it) {
      if (LDS) {   // 48 MFMAs of 16 KFLOP per iteration: the same 13 fragment reads per 24 x 32 KFLOP
#pragma unroll
        for (int i = 0; i < 4; ++i) { a[i] = frag[(threadIdx.x + 64 * ((it + i) & 31)) & 4095]; b[i] = frag[(threadIdx.x + 64 * ((it + i + 7) & 31) + 2048) & 4095]; }
#pragma unroll
        for (int i = 0; i < 3; ++i) a[i] = frag[(threadIdx.x + 64 * ((it + i + 3) & 31) + 1024) & 4095];
#pragma unroll
        for (int i = 0; i < 2; ++i) b[i] = frag[(threadIdx.x + 64 * ((it + i + 11) & 31) + 3072) & 4095];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int i = 0; i < 12; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[(i + u) & 3], b[u], acc[i], 0, 0, 0);
    }
    for (int i = 0; i < 12; ++i) out += acc[i][0] + acc[i][3];
  }
  if (out == 12345.678f) sink[0] = out;
}

template <int KIND, int LDS = 0>
static void run(int zero_every, const char* what, float* sink) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int iters = 400000;    // ~0.3 s per launch: long enough for the power management to settle
  float best = 1e9;
  for (int r = 0; r < 3; ++r) {
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL((k<KIND, LDS>), dim3(256 * 1), dim3(512), 0, 0, iters, zero_every, sink);
    (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (r > 0 && ms < best) best = ms;
  }
  const double flop = (KIND == 0 ? 24.0 * 32 * 32 * 16 * 2 : 48.0 * 16 * 16 * 32 * 2) * iters * 256.0 * 8;
  printf("%-44s %8.2f ms  %7.1f TFLOP/s  [%s]\n", what, best, flop / best / 1e9, hipGetErrorString(hipGetLastError()));
}
int main() {
  setvbuf(stdout, nullptr, _IONBF, 0);
  float* sink; (void)hipMalloc(&sink, 4);
  run<0>(0, "32x32x16 f16, random operands", sink);
  run<1>(0, "16x16x32 f16, random operands", sink);
  run<0>(2, "32x32x16 f16, half of the operands zero", sink);
  run<1>(2, "16x16x32 f16, half of the operands zero", sink);
  run<0, 1>(0, "32x32x16 f16, random, + LDS fragment reads", sink);
  run<1, 1>(0, "16x16x32 f16, random, + LDS fragment reads", sink);
  run<0, 1>(2, "32x32x16 f16, half zero, + LDS fragment reads", sink);
  run<1, 1>(2, "16x16x32 f16, half zero, + LDS fragment reads", sink);
  run<0>(1, "32x32x16 f16, all-zero operands", sink);
  run<1>(1, "16x16x32 f16, all-zero operands", sink);
  return 0;
}
