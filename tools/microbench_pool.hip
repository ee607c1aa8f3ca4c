// Bandwidth check of maxpool1d_nlc_kernel<5> (channel-last [n][128] fp32) in isolation.
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -w -I orca_amd/csrc -I include tools/microbench_pool.hip -o tools/microbench_pool
#include <hip/hip_runtime.h>
#include <cstdio>
#include "conv_p16.h"
#include "conv_bf16s.h"
__global__ void stream_read_kernel(const f32x4* __restrict__ x, float* __restrict__ y, long n4) {
  f32x4 acc = (f32x4)(0.f);
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) acc += x[i];
  if (acc.x == 12345.f) y[0] = acc.y;
}
int main() {
  const long n = 500000, C = 128, n_out = n / 5;
  float *x, *y; hipMalloc(&x, n * C * 4); hipMalloc(&y, n_out * C * 4); hipMemset(x, 0x3c, n * C * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const long per_block = 2 * (256 / (C / 4));
  for (int r = 0; r < 3; ++r) {
    hipEventRecord(e0, 0);
    for (int k = 0; k < 10; ++k) hipLaunchKernelGGL((maxpool1d_nlc_kernel<5>), dim3((unsigned)((n_out + per_block - 1) / per_block)), dim3(256), 0, 0, x, y, n_out, (int)C);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("maxpool<5> n=%ld: %.1f us per launch = %.2f TB/s (read + write)  [%s]\n", n, ms * 100.f, (n * C * 4 + n_out * C * 4) / (ms * 1e-4) / 1e12, hipGetErrorString(hipGetLastError()));
  }
  for (int g : {1024, 4096, 16384}) {
    hipEventRecord(e0, 0);
    for (int k = 0; k < 10; ++k) hipLaunchKernelGGL(stream_read_kernel, dim3(g), dim3(256), 0, 0, reinterpret_cast<const f32x4*>(x), y, n * C / 4);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("grid-stride 16-byte read, %d blocks: %.1f us = %.2f TB/s\n", g, ms * 100.f, (n * C * 4) / (ms * 1e-4) / 1e12);
  }
  return 0;
}
