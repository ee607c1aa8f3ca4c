// p16_planes.h - moving pieces of P16 sequence tensors (conv_p16.h: planes of 16-byte units, one unit = 8 channels of one position as fp16 hi or lo
// parts; plane of octet o, split s at base + (o * 2 + s) * plen, positions behind P16_GUARD units) between a stage-3 cache of a chromosome and the
// stage-4 input of a window (orca_encoder_stage3_planes / orca_p16_pool5_into / orca_encoder_front_snippet / orca_encoder_back, orca_encoder.hip).
#pragma once
#include "conv2d_m16.h"   // m16_unpack8 / m16_pack8

// nn.MaxPool1d(5, 5) (orca_modules.py:853) from positions x_pos0 + 5 m .. + 4 of x into position y_pos0 + m of y, m < count: the pooled value is the
// max of the stored (hi + lo) values, re-split - the split rounds monotonically, so this equals splitting the max of the fp32 values, which is what
// the fused pool of conv_p16p5.h stores.  grid (ceil(count / 256), 16 octets of 128 channels), block 256.
static __global__ void p16_pool5_into_kernel(const f32x4* __restrict__ x, long x_plen, long x_pos0, f32x4* __restrict__ y, long y_plen, long y_pos0, long count) {
  const long m = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int o = blockIdx.y;
  if (m >= count) return;
  const u32x4_t* xp = reinterpret_cast<const u32x4_t*>(x) + (long)o * 2 * x_plen + P16_GUARD + x_pos0 + m * 5;
  float best[8];
#pragma unroll
  for (int j = 0; j < 5; ++j) {
    u32x4_t u[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) u[s] = xp[(long)s * x_plen + j];
    float v[8];
    m16_unpack8<2, 1>(u, v);
#pragma unroll
    for (int e = 0; e < 8; ++e) best[e] = j == 0 ? v[e] : fmaxf(best[e], v[e]);
  }
  f32x4 a, b;
  a.x = best[0]; a.y = best[1]; a.z = best[2]; a.w = best[3]; b.x = best[4]; b.y = best[5]; b.z = best[6]; b.w = best[7];
  u32x4_t out[2];
  bool ovf = false;
  m16_pack8<2, 1>(a, b, out, ovf);
#pragma unroll
  for (int s = 0; s < 2; ++s) reinterpret_cast<u32x4_t*>(y)[((long)o * 2 + s) * y_plen + P16_GUARD + y_pos0 + m] = out[s];
}

// units [x_pos0, x_pos0 + count) of every plane of x -> units [y_pos0, ..) of y.  grid (ceil(count / 256), planes), block 256.
static __global__ void p16_copy_units_kernel(const f32x4* __restrict__ x, long x_plen, long x_pos0, f32x4* __restrict__ y, long y_plen, long y_pos0, long count) {
  const long m = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= count) return;
  y[(long)blockIdx.y * y_plen + P16_GUARD + y_pos0 + m] = x[(long)blockIdx.y * x_plen + P16_GUARD + x_pos0 + m];
}

// the same pool on fp32 channel-last rows [n][128] (stage 4's output -> stage 5's input): rows src_pos0 + 5 m .. + 4 of x -> row y_pos0 + m of y.
// One thread = 4 channels of one output row; grid ceil(count * 32 / 256), block 256.
static __global__ void rows_pool5_into_kernel(const f32x4* __restrict__ x, long x_pos0, f32x4* __restrict__ y, long y_pos0, long count) {
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long m = t >> 5;
  const int q = (int)(t & 31);
  if (m >= count) return;
  const f32x4* xp = x + (x_pos0 + 5 * m) * 32 + q;
  f32x4 v = xp[0];
#pragma unroll
  for (int j = 1; j < 5; ++j) {
    const f32x4 u = xp[j * 32];
    v.x = fmaxf(v.x, u.x); v.y = fmaxf(v.y, u.y); v.z = fmaxf(v.z, u.z); v.w = fmaxf(v.w, u.w);
  }
  y[(y_pos0 + m) * 32 + q] = v;
}
