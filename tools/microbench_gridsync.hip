// What a grid-wide barrier + cross-XCD visibility costs on MI355X inside ONE persistent kernel (the price a layer-fused Decoder would
// pay per layer instead of a kernel boundary): 256 workgroups x 512 threads, per iteration each workgroup writes `wkb` KB of its own
// slab, releases (agent-scope fence), joins a counter barrier, acquires, and reads `rkb` KB of OTHER workgroups' slabs.
// hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/microbench_gridsync.hip -o tools/microbench_gridsync
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void grid_barrier(unsigned* counter, unsigned target) {
  __syncthreads();
  if (threadIdx.x == 0) {
    __atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE);                 // agent scope (HIP default for __atomic on global)
    while (__atomic_load_n(counter, __ATOMIC_ACQUIRE) < target) __builtin_amdgcn_s_sleep(1);
  }
  __syncthreads();
}

template <int MODE>   // 0: barrier only; 1: + fences; 2: + write/read traffic
__global__ __launch_bounds__(512) void k(f32x4* buf, unsigned* counter, int iters, int wunits, int runits, float* sink) {
  const int nwg = gridDim.x, wg = blockIdx.x, tid = threadIdx.x;
  const long slab = 65536 / 16 * 4;   // 256 KB per workgroup in 16-byte units
  f32x4 acc = (f32x4)(0.f);
  for (int it = 0; it < iters; ++it) {
    f32x4* mine = buf + ((long)(it & 1) * nwg + wg) * slab;
    if (MODE >= 2) for (int i = tid; i < wunits; i += 512) mine[i] = (f32x4)((float)(it + i));
    if (MODE >= 1) __threadfence();
    grid_barrier(counter, (unsigned)(nwg * (it + 1)));
    if (MODE >= 1) __threadfence();
    if (MODE >= 2) {
      const f32x4* a = buf + ((long)(it & 1) * nwg + (wg + 1) % nwg) * slab;
      const f32x4* b = buf + ((long)(it & 1) * nwg + (wg + 37) % nwg) * slab;
      for (int i = tid; i < runits / 2; i += 512) { acc += __builtin_nontemporal_load(a + i); acc += __builtin_nontemporal_load(b + i); }
    }
  }
  if (acc.x == 12345.678f) sink[0] = acc.y;
}

template <int MODE>
static void run(f32x4* buf, unsigned* counter, float* sink, int iters, int wkb, int rkb, const char* what) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e9;
  for (int r = 0; r < 3; ++r) {
    hipMemset(counter, 0, 4);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 0, 0, buf, counter, iters, wkb * 64, rkb * 64, sink);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
  }
  printf("%-58s %7.2f us per iteration  [%s]\n", what, best * 1e3 / iters, hipGetErrorString(hipGetLastError()));
}
int main() {
  setvbuf(stdout, nullptr, _IONBF, 0);
  f32x4* buf; unsigned* counter; float* sink;
  hipMalloc(&buf, (size_t)2 * 256 * 262144); hipMalloc(&counter, 4); hipMalloc(&sink, 4);
  hipMemset(buf, 0, (size_t)2 * 256 * 262144);
  const int iters = 200;
  run<0>(buf, counter, sink, iters, 0, 0, "counter barrier only");
  run<1>(buf, counter, sink, iters, 0, 0, "+ agent-scope fences on both sides");
  run<2>(buf, counter, sink, iters, 64, 192, "+ 64 KB written, 192 KB of two other slabs read");
  run<2>(buf, counter, sink, iters, 32, 96, "+ 32 KB written, 96 KB read");
  return 0;
}
