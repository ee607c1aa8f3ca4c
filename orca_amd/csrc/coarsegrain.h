// coarsegrain.h - adaptive coarse-graining of an observed Hi-C matrix (the reference's `adaptive_coarsegrain_gpu`,
// selene_utils2.py:274-463: what fills output["experiments"] from real data).  HBM-bound pyramid of elementwise passes:
//   init      : pad to the next power of two N; invalid (non-finite) pixels -> value 0, count 0, mask 0
//   coarsen   : 2x2 sums of (value, raw count, number of valid pixels), level by level while the side is > min_shape
//   refine    : from the coarsest level down: where the min raw count of a 2x2 block (invalid pixels count as 0) is below
//               `cutoff`, the block's pixels take (coarse value / coarse valid count) * own valid count; invalid -> 0
//   finish    : invalid pixels of the finest level -> NaN, crop to n x n
// The float32 operation order is the reference's ((row pair) + (row pair), then the two columns; one division, one
// multiplication), so the result is bit-identical to it.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

static __global__ void cg_init_kernel(const float* __restrict__ ar, const float* __restrict__ cnt, long ld_in, int n, int N, float* __restrict__ v,
                               float* __restrict__ c, int* __restrict__ m) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)N * N) return;
  const int i = (int)(idx / N), j = (int)(idx - (long)i * N);
  float a = __builtin_nanf("");
  float k = 0.f;
  if (i < n && j < n) { a = ar[(long)i * ld_in + j]; k = cnt[(long)i * ld_in + j]; }
  const bool ok = __builtin_isfinite(a);
  v[idx] = ok ? a : 0.f;
  c[idx] = ok ? k : 0.f;
  m[idx] = ok ? 1 : 0;
}

// fine side F = 2M; (row 2i + row 2i+1) first, then the two columns - the order of torch.sum(axis=1) then (axis=2)
static __global__ void cg_coarsen_kernel(const float* __restrict__ v, const float* __restrict__ c, const int* __restrict__ m, int M, float* __restrict__ vo,
                                  float* __restrict__ co, int* __restrict__ mo) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)M * M) return;
  const int i = (int)(idx / M), j = (int)(idx - (long)i * M);
  const long F = 2L * M, p = 2L * i * F + 2 * j;
  vo[idx] = (v[p] + v[p + F]) + (v[p + 1] + v[p + F + 1]);
  co[idx] = (c[p] + c[p + F]) + (c[p + 1] + c[p + F + 1]);
  mo[idx] = (m[p] + m[p + F]) + (m[p + 1] + m[p + F + 1]);
}

// one thread per COARSE pixel: updates its 2x2 block of the finer level in place
static __global__ void cg_refine_kernel(const float* __restrict__ v_cur, const int* __restrict__ m_cur, int M, float cutoff, float* __restrict__ v_next,
                                 const float* __restrict__ c_next, const int* __restrict__ m_next) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)M * M) return;
  const int i = (int)(idx / M), j = (int)(idx - (long)i * M);
  const long F = 2L * M, p = 2L * i * F + 2 * j;
  const long q[4] = {p, p + 1, p + F, p + F + 1};
  const float cmin = fminf(fminf(c_next[q[0]], c_next[q[1]]), fminf(c_next[q[2]], c_next[q[3]]));
  const float val = v_cur[idx] / (float)m_cur[idx];      // NaN when the coarse pixel has no valid fine pixel
  const bool repl = cmin < cutoff;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int mk = m_next[q[e]];
    float x = v_next[q[e]];
    if (repl) x = val * (float)mk;
    if (mk == 0) x = 0.f;
    v_next[q[e]] = x;
  }
}

static __global__ void cg_finish_kernel(const float* __restrict__ v, const int* __restrict__ m, int N, int n, float* __restrict__ out, long ld_out) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)n * n) return;
  const int i = (int)(idx / n), j = (int)(idx - (long)i * n);
  const long p = (long)i * N + j;
  out[(long)i * ld_out + j] = m[p] ? v[p] : __builtin_nanf("");
}
