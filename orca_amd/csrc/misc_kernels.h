// misc_kernels.h - the small HBM-bound kernels around the convolutions:
// input staging, max-pool, nearest/bilinear upsampling, the pairwise outer sum,
// the 1x1 "final" head with symmetrisation, and the strand merge.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "conv_kernels.h"

#ifndef ORCA_MAX_TARGETS
#define ORCA_MAX_TARGETS 8   // num_2d of the multi-target decoders (orca_leukemia.py:512-990); hidden width F = max(5, T)
#endif

// x[b] viewed as [4][L] with element strides (sc, sl)  ->  y [4][ld] channel-major.
// The reference feeds `seq.transpose(1,2)` of a [B,L,4] array (orca_predict.py:334),
// i.e. sc=1, sl=4: one float4 per position.
static __global__ void seq_to_channel_major_kernel(const float* __restrict__ x, long sc, long sl, long L, float* __restrict__ y, long ld) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= L) return;
  float v0, v1, v2, v3;
  if (sc == 1 && sl == 4 && ((reinterpret_cast<uintptr_t>(x) & 15) == 0)) {
    const float4 q = reinterpret_cast<const float4*>(x)[i];
    v0 = q.x; v1 = q.y; v2 = q.z; v3 = q.w;
  } else {
    v0 = x[i * sl]; v1 = x[i * sl + sc]; v2 = x[i * sl + 2 * sc]; v3 = x[i * sl + 3 * sc];
  }
  y[i] = v0; y[ld + i] = v1; y[2 * ld + i] = v2; y[3 * ld + i] = v3;
}

// x viewed as [4][L] with element strides (sc, sl)  ->  y [L][4] contiguous rows (input of the first-layer MFMA kernel)
static __global__ void seq_to_rows_kernel(const float* __restrict__ x, long sc, long sl, long L, float* __restrict__ y) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= L) return;
  float4 q;
  q.x = x[i * sl]; q.y = x[i * sl + sc]; q.z = x[i * sl + 2 * sc]; q.w = x[i * sl + 3 * sc];
  reinterpret_cast<float4*>(y)[i] = q;
}

// 2-bit + N-mask genome window -> 1-byte base codes (one thread = 4 consecutive output bases)
static __global__ void genome_unpack_2bit_kernel(const unsigned char* __restrict__ two, const unsigned char* __restrict__ nmask, long start, long n,
                                          unsigned char* __restrict__ codes) {
  const long q = (long)blockIdx.x * blockDim.x + threadIdx.x;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const long o = 4 * q + e;
    if (o >= n) return;
    const long i = start + o;
    const unsigned c = (two[i >> 2] >> ((i & 3) * 2)) & 3u;
    const unsigned isn = (nmask[i >> 3] >> (i & 7)) & 1u;
    codes[o] = (unsigned char)(isn ? 4u : c);
  }
}

// nn.MaxPool1d(k, k): y[r][m] = max_j x[r][k*m+j]
template <int K>
__global__ void maxpool1d_kernel(const float* __restrict__ x, long ldx, float* __restrict__ y, long ldy, long n_out) {
  const long m = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long r = blockIdx.y;
  if (m >= n_out) return;
  const float* p = x + r * ldx + (long)K * m;
  float v;
  if (K == 4 && (ldx & 3) == 0 && ((reinterpret_cast<uintptr_t>(x) & 15) == 0)) {
    const float4 q = *reinterpret_cast<const float4*>(p);
    v = fmaxf(fmaxf(q.x, q.y), fmaxf(q.z, q.w));
  } else if (K == 2 && (ldx & 1) == 0 && ((reinterpret_cast<uintptr_t>(x) & 7) == 0)) {
    const float2 q = *reinterpret_cast<const float2*>(p);
    v = fmaxf(q.x, q.y);
  } else {
    v = p[0];
#pragma unroll
    for (int j = 1; j < K; ++j) v = fmaxf(v, p[j]);
  }
  y[r * ldy + m] = v;
}

// nn.Upsample(scale_factor=2) (nearest, 1-D): y[r][m] = x[r][m>>1]
static __global__ void upsample1d_x2_kernel(const float* __restrict__ x, long ldx, float* __restrict__ y, long ldy, long n_out) {
  const long m = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long r = blockIdx.y;
  if (m >= n_out) return;
  y[r * ldy + m] = x[r * ldx + (m >> 1)];
}

// nn.Upsample(scale_factor=2) (nearest) on channel-last rows: y[m][:] = x[m >> 1][:]; one thread = 4 channels.  grid ceil(n_out*C/4 / 256)
static __global__ void upsample1d_x2_nlc_kernel(const float* __restrict__ x, float* __restrict__ y, long n_out, int C) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int c4n = C / 4;
  if (idx >= n_out * c4n) return;
  const long m = idx / c4n;
  const int c4 = (int)(idx - m * c4n);
  reinterpret_cast<f32x4*>(y + m * C)[c4] = reinterpret_cast<const f32x4*>(x + (m >> 1) * C)[c4];
}

// [rows][cols] <-> its transpose through a 32 x 33 LDS tile, both sides coalesced: dst[c * ldd + r] = src[r * lds_ + c]
// (channel-last [n][128] <-> channel-major [128][n] hand-overs of the U-net encoders).  grid (ceil(cols/32), ceil(rows/32), batch), block (32, 8)
static __global__ void transpose2d_kernel(const float* __restrict__ src, long lds_, long src_bs, float* __restrict__ dst, long ldd, long dst_bs,
                                   long rows, long cols) {
  __shared__ float tile[32][33];
  src += (long)blockIdx.z * src_bs;
  dst += (long)blockIdx.z * dst_bs;
  const long r0 = (long)blockIdx.y * 32, c0 = (long)blockIdx.x * 32;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const long r = r0 + threadIdx.y + 8 * k, c = c0 + threadIdx.x;
    if (r < rows && c < cols) tile[threadIdx.y + 8 * k][threadIdx.x] = src[r * lds_ + c];
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const long c = c0 + threadIdx.y + 8 * k, r = r0 + threadIdx.x;
    if (r < rows && c < cols) dst[c * ldd + r] = tile[threadIdx.x][threadIdx.y + 8 * k];
  }
}

// generic strided 2-D copy: dst[r*ldd + c] = src[r*lds_ + c*scol]
static __global__ void copy2d_kernel(const float* __restrict__ src, long lds_, long scol, float* __restrict__ dst, long ldd, long cols) {
  const long c = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long r = blockIdx.y;
  if (c >= cols) return;
  dst[r * ldd + c] = src[r * lds_ + c * scol];
}

// Decoder input (orca_modules.py:462-463): mat[c][i][j] = x[c][i] + x[c][j] (c<128),
// channel 128 = distenc[i][j] (if given), remaining pad channels = 0.
// out layout [cpad][n][256]; one thread writes a float4 of columns.
static __global__ void outer_sum_kernel(const float* __restrict__ x, long sx_c, long sx_l, const float* __restrict__ de, long sd_c, long sd_h,
                                 long sd_w, int nt, float* __restrict__ out, int n, int cpad) {
  const int j4 = threadIdx.x;  // 0..63 -> columns 4*j4..4*j4+3
  const int i = blockIdx.x;
  const int c = blockIdx.y;
  float v[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int j = 4 * j4 + e;
    float t = 0.f;
    if (j < n) {
      if (c < 128) t = x[c * sx_c + i * sx_l] + x[c * sx_c + j * sx_l];
      else if (c < 128 + nt && de) t = de[(c - 128) * sd_c + i * sd_h + j * sd_w];
    }
    v[e] = t;
  }
  *reinterpret_cast<float4*>(out + ((long)c * n + i) * ORCA_LDW + 4 * j4) = make_float4(v[0], v[1], v[2], v[3]);
}

// nn.Upsample(scale_factor=(2,2), mode) of y [n/2][n/2] written into ONE channel plane
// [n][256] (orca_modules.py:430,468); the 7 pad planes behind it are zeroed.
// bilinear = PyTorch align_corners=False: src = max((dst+0.5)/2-0.5, 0).
static __global__ void upsample2d_x2_kernel(const float* __restrict__ y, long sy_c, long sy_h, long sy_w, int nt, float* __restrict__ out, int n,
                                     int bilinear, int nplanes) {
  const int j = threadIdx.x;
  const int i = blockIdx.x;
  const int p = blockIdx.y;
  float v = 0.f;
  const int h = n / 2;
  if (p < nt && j < n) {
    y += p * sy_c;
    if (!bilinear) {
      v = y[(i >> 1) * sy_h + (j >> 1) * sy_w];
    } else {
      const float fy = fmaxf(0.5f * (i + 0.5f) - 0.5f, 0.f), fx = fmaxf(0.5f * (j + 0.5f) - 0.5f, 0.f);
      const int y0 = (int)fy, x0 = (int)fx;
      const int y1 = y0 + (y0 < h - 1 ? 1 : 0), x1 = x0 + (x0 < h - 1 ? 1 : 0);
      const float ly = fy - y0, lx = fx - x0;
      const float hy = 1.f - ly, hx = 1.f - lx;
      v = hy * (hx * y[y0 * sy_h + x0 * sy_w] + lx * y[y0 * sy_h + x1 * sy_w]) +
          ly * (hx * y[y1 * sy_h + x0 * sy_w] + lx * y[y1 * sy_h + x1 * sy_w]);
    }
  }
  out[((long)p * n + i) * ORCA_LDW + j] = v;
}

// `final` head (orca_modules.py:423-428) + symmetrisation (:488):
// f(i,j) = w2 . relu(W1 cur[:,i,j] + b1) + b2 ; out[i][j] = 0.5 f(i,j) + 0.5 f(j,i) (+= if accumulate)
struct FinalArgs {
  const float* cur;  // [64][n][256]
  const float* w1;   // [F][64] (BN folded)
  const float* b1;   // [F]
  const float* w2;   // [T][F]
  const float* b2;   // [T]
  float* out;        // [T][n][n]
  long cur_bs, out_bs;
  int n;
  int accumulate;
  int F, T;
};

// shared tail of the two `final` kernels: h1 / h2 = W1 cur[:, i, j] + b1 and the same at (j, i)
__device__ __forceinline__ void final_head_store(const FinalArgs& a, const float* h1, const float* h2, const float* w2s, const float* b2s,
                                                  int b, int i, int j) {
  const int n = a.n;
  for (int t = 0; t < a.T; ++t) {
    float f1 = b2s[t], f2 = b2s[t];
#pragma unroll
    for (int o = 0; o < ORCA_MAX_TARGETS; ++o)
      if (o < a.F) { f1 = fmaf(w2s[t * ORCA_MAX_TARGETS + o], fmaxf(h1[o], 0.f), f1); f2 = fmaf(w2s[t * ORCA_MAX_TARGETS + o], fmaxf(h2[o], 0.f), f2); }
    float* op = a.out + (long)b * a.out_bs + ((long)t * n + i) * n + j;
    const float r = 0.5f * f1 + 0.5f * f2;
    *op = a.accumulate ? (*op + r) : r;
  }
}

#define ORCA_FINAL_LOAD_HEAD()                                                                              \
  __shared__ float w1s[ORCA_MAX_TARGETS * 64], b1s[ORCA_MAX_TARGETS], w2s[ORCA_MAX_TARGETS * ORCA_MAX_TARGETS], b2s[ORCA_MAX_TARGETS]; \
  for (int t = threadIdx.x; t < ORCA_MAX_TARGETS * 64; t += blockDim.x) w1s[t] = t < a.F * 64 ? a.w1[t] : 0.f; \
  if (threadIdx.x < ORCA_MAX_TARGETS) {                                                                     \
    b1s[threadIdx.x] = threadIdx.x < a.F ? a.b1[threadIdx.x] : 0.f;                                         \
    b2s[threadIdx.x] = threadIdx.x < a.T ? a.b2[threadIdx.x] : 0.f;                                         \
  }                                                                                                         \
  if (threadIdx.x < ORCA_MAX_TARGETS * ORCA_MAX_TARGETS) {                                                  \
    const int t_ = threadIdx.x / ORCA_MAX_TARGETS, o_ = threadIdx.x % ORCA_MAX_TARGETS;                     \
    w2s[threadIdx.x] = (t_ < a.T && o_ < a.F) ? a.w2[t_ * a.F + o_] : 0.f;                                  \
  }                                                                                                         \
  __syncthreads();

static __global__ void final_sym_kernel(FinalArgs a) {
  ORCA_FINAL_LOAD_HEAD();
  const int j = threadIdx.x, i = blockIdx.x, b = blockIdx.y, n = a.n;
  if (j >= n) return;
  const float* cur = a.cur + (long)b * a.cur_bs;
  const long cs = (long)n * ORCA_LDW;
  float h1[ORCA_MAX_TARGETS], h2[ORCA_MAX_TARGETS];
#pragma unroll
  for (int o = 0; o < ORCA_MAX_TARGETS; ++o) { h1[o] = b1s[o]; h2[o] = b1s[o]; }
  for (int c = 0; c < 64; ++c) {
    const float u = cur[c * cs + (long)i * ORCA_LDW + j];
    const float v = cur[c * cs + (long)j * ORCA_LDW + i];
#pragma unroll
    for (int o = 0; o < ORCA_MAX_TARGETS; ++o)
      if (o < a.F) { h1[o] = fmaf(w1s[o * 64 + c], u, h1[o]); h2[o] = fmaf(w1s[o * 64 + c], v, h2[o]); }
  }
  final_head_store(a, h1, h2, w2s, b2s, b, i, j);
}

// strand merge (orca_predict.py:514-523): out = 0.5*fwd + 0.5*rev[::-1, ::-1]
static __global__ void strand_merge_kernel(const float* __restrict__ fwd, const float* __restrict__ rev, float* __restrict__ out, int n) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n * n) return;
  out[idx] = fwd[idx] * 0.5f + rev[n * n - 1 - idx] * 0.5f;
}

// [B,C,n,n] contiguous <-> padded [B,C,n,256] (single-layer conv2d entry point only)
static __global__ void pad_rows_kernel(const float* __restrict__ src, float* __restrict__ dst, int n, int to_padded) {
  const int j = threadIdx.x;
  const long row = blockIdx.x;  // over B*C*n
  if (to_padded) dst[row * ORCA_LDW + j] = (j < n) ? src[row * n + j] : 0.f;
  else if (j < n) dst[row * n + j] = src[row * ORCA_LDW + j];
}

// y[b][co][m] = act(bias[co] + sum_ci w[co][ci] * x[b][ci][m])  - the kernel-size-1 Conv1d layers of Net.final_1d.
// grid (ceil(n/256), cout, B); x reads are coalesced along m, w/bias are wave-uniform (scalar loads).
static __global__ void pointwise1d_kernel(const float* __restrict__ w, const float* __restrict__ bias, int cin, const float* __restrict__ x,
                                   long x_bs, long ldx, float* __restrict__ y, long y_bs, long ldy, long n, int act) {
  const long m = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int co = blockIdx.y, b = blockIdx.z;
  if (m >= n) return;
  const float* xp = x + (long)b * x_bs + m;
  const float* wp = w + (long)co * cin;
  float acc = bias[co];
  for (int ci = 0; ci < cin; ++ci) acc = fmaf(wp[ci], xp[(long)ci * ldx], acc);
  if (act == 1) acc = fmaxf(acc, 0.f);
  else if (act == 2) acc = 1.f / (1.f + expf(-acc));
  y[(long)b * y_bs + (long)co * ldy + m] = acc;
}

// ---- block means of the 256 Mb model's background (orca_predict.py:724-737) -------------------------------------------
// normmat_r = nanmean(nanmean(reshape(normmat[s : s + 250 nb, s : s + 250 nb], (250, nb, 250, nb)), axis=3), axis=1) in float64,
// then distenc = log(float32(normmat_r)) [flipped in both axes on the reverse strand, :703].  One thread per output pixel,
// numpy's own operation order so that the float64 means are BIT-IDENTICAL to the reference's:
//   inner axis (contiguous, n = nb): numpy's pairwise sum - n < 8 sequential from -0.0, else 8 running sums r[k] += a[i+k]
//   combined ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)) - divided by the count; outer axis: sequential sum of those nb means, divided by the count.
// NaN entries count as missing (nanmean); an all-NaN group gives NaN.
static __global__ void block_mean_f64_kernel(const double* __restrict__ mat, long ld, long r0, long c0, int nb, int npix, double* __restrict__ mean_out,
                                      float* __restrict__ log_out, int flip) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x, i = blockIdx.y;
  if (j >= npix) return;
  double acc = 0.0;
  int cnt2 = 0;
  for (int a = 0; a < nb; ++a) {
    const double* row = mat + (r0 + (long)i * nb + a) * ld + c0 + (long)j * nb;
    double s;
    int cnt = 0;
    if (nb < 8) {
      s = -0.0;
      for (int k = 0; k < nb; ++k) { const double v = row[k]; const bool ok = v == v; cnt += ok; s = s + (ok ? v : 0.0); }
    } else {
      double r[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) { const double v = row[k]; const bool ok = v == v; cnt += ok; r[k] = ok ? v : 0.0; }
      int q = 8;
      for (; q < nb - (nb % 8); q += 8) {
#pragma unroll
        for (int k = 0; k < 8; ++k) { const double v = row[q + k]; const bool ok = v == v; cnt += ok; r[k] = r[k] + (ok ? v : 0.0); }
      }
      s = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
      for (; q < nb; ++q) { const double v = row[q]; const bool ok = v == v; cnt += ok; s = s + (ok ? v : 0.0); }
    }
    const double m1 = s / (double)cnt;          // 0/0 = NaN for an all-NaN group, as numpy
    const bool ok1 = m1 == m1;
    cnt2 += ok1;
    acc = a == 0 ? (ok1 ? m1 : 0.0) : acc + (ok1 ? m1 : 0.0);
  }
  const double m = acc / (double)cnt2;
  if (mean_out) mean_out[(long)i * npix + j] = m;
  if (log_out) {
    const long o = flip ? (long)(npix - 1 - i) * npix + (npix - 1 - j) : (long)i * npix + j;
    log_out[o] = logf((float)m);
  }
}
