// conv_bf16s.h - Conv1d k=9 on the bf16 matrix cores with SPLIT fp32 operands.
//
// gfx950 has no TF32; its f32-input MFMA runs at the vector rate (157 TFLOP/s) while
// v_mfma_f32_32x32x16_bf16 runs 16x faster (2.5 PFLOP/s) with fp32 accumulation.  An
// fp32 value is the exact sum of three bf16 values x = x1 + x2 + x3 (8 significant bits
// each, same exponent range as fp32 - no overflow/underflow hazard), so
//     x*w = x1w1 + x1w2 + x2w1 + x1w3 + x2w2 + x3w1  + O(2^-24 |xw|)
// i.e. SIX bf16 MFMAs reproduce an fp32 product to fp32-rounding accuracy
// (end-to-end emulation: max-abs 2.3e-6 through the 28-conv Encoder vs fp32) at
// 16/6 = 2.67x the fp32-MFMA rate.  NS = 3 is that mode; NS = 2 keeps the first three
// products (~2^-17 relative), NS = 1 is plain bf16 (throughput mode).
//
// Layout: activations channel-LAST [pos][C] fp32 in HBM (a lane's MFMA operand is 8
// consecutive channels of one position = one 16-byte LDS read after splitting).
//   A (lane l) = X[pos0 + (l&31) + tap][ci0 + 8*(l>>5) .. +7]    32 positions x 16 channels
//   B (lane l) = W[tap][ci0 + 8*(l>>5) .. +7][n0 + (l&31)]        16 channels x 32 couts
//   D = B^T-role swap: issued as mfma(W-frag, X-frag) so that a lane owns one position and
//                4 consecutive couts per register group -> float4 epilogue on channel-last rows
// K is walked in chunks of 16 input channels x 9 taps.  Per chunk the LDS holds
//   X image [split][g=ci/8][MT+8 positions][8 ch] bf16   (split on the fly while staging)
//   W image [split][tap][g][COUT][8 ch]          bf16   (pre-split on the host)
// Chunk c+1 is prefetched global->registers during the MFMA block of chunk c.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "conv_kernels.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

// 16-bit operand type of the MFMA: DT = 0 bf16 (8 significant bits, fp32 exponent range: splits are
// exact for every finite fp32), DT = 1 fp16 (11 significant bits: TWO parts carry 22 bits, so three
// products give ~2^-22 relative error at half the MFMA count of bf16x3 - but |x| must stay < 65504).
template <int DT> struct Op16;
template <> struct Op16<0> {
  typedef bf16x8 vec;
  static __device__ __forceinline__ f32x16 mfma(vec a, vec b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
};
template <> struct Op16<1> {
  typedef f16x8 vec;
  static __device__ __forceinline__ f32x16 mfma(vec a, vec b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
};

struct ConvB16Args {
  const float* x;   // [B][n][cin]   channel-last fp32
  const void* w;    // packed bf16 [nchunks][3 splits][9][2][COUT][8] (a kernel reads the first NS splits)
  const float* bias;
  float* y;         // [B][n][COUT]
  const float* r1;  // optional residual, layout of y
  const float* r2;  // optional second residual (the U-net's skip connection), same layout and batch stride
  long x_bs, y_bs;  // batch strides (elements)
  long r_bs;        // batch stride of r1
  long n;
  long tiles_per_row;  // ceil(n / MT)
  int batch;
  int cin;          // row stride of x (elements)
  int nchunks;      // cin / 16
  int relu;
  int stagger;      // units of 4096 cycles by which half of the resident workgroups start late
  unsigned* flag;   // fp16 mode: set to 1 if an activation exceeds the fp16 range (result then invalid)
  int pool4;        // fuse nn.MaxPool1d(4,4) into the epilogue: y is [n/4][COUT], y_bs its batch stride
  int cout;         // output channels of the layer (conv_small.h: its workgroups take 32-cout blocks; the kernel below has it as COUT)
};

__device__ __forceinline__ unsigned cvt_pk_bf16(float lo, float hi) {
  unsigned r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));  // RNE, lo -> bits 0..15
  return r;
}
__device__ __forceinline__ float bf16lo_f32(unsigned pk) { return __uint_as_float(pk << 16); }
__device__ __forceinline__ float bf16hi_f32(unsigned pk) { return __uint_as_float(pk & 0xffff0000u); }

// 4 fp32 -> NS x (4 halves packed in 8 bytes): successive round-to-nearest residual splits
template <int NS, int DT>
__device__ __forceinline__ void split4(f32x4 v, u32x2 (&out)[NS], bool& overflow) {
  if (DT == 1) {
    overflow |= (fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))) > 65504.f);
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      const f16x2 h0 = {(_Float16)v.x, (_Float16)v.y}, h1 = {(_Float16)v.z, (_Float16)v.w};
      out[s].x = __builtin_bit_cast(unsigned, h0);
      out[s].y = __builtin_bit_cast(unsigned, h1);
      if (s + 1 < NS) {
        v.x -= (float)h0.x; v.y -= (float)h0.y; v.z -= (float)h1.x; v.w -= (float)h1.y;
      }
    }
    return;
  }
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    const unsigned p0 = cvt_pk_bf16(v.x, v.y), p1 = cvt_pk_bf16(v.z, v.w);
    out[s].x = p0;
    out[s].y = p1;
    if (s + 1 < NS) {
      v.x -= bf16lo_f32(p0); v.y -= bf16hi_f32(p0);
      v.z -= bf16lo_f32(p1); v.w -= bf16hi_f32(p1);
    }
  }
}

// PERSISTENT kernel: the grid is (CUs x resident workgroups per CU); each workgroup walks tiles
// t = blockIdx.x, blockIdx.x + gridDim.x, ... and treats (tile, chunk) as ONE continuous stream, so the
// first chunk of the next tile is prefetched under the MFMA block of the current tile's last chunk and
// the epilogue stores overlap the in-flight prefetch.  (A one-workgroup-per-tile launch of this kernel
// is workgroup-DISPATCH bound: an empty 80 KB-LDS workgroup costs ~83 ns of dispatch, 2.6 ms per 31 250
// tiles - more than the 1.6 ms of MFMA work in them.)
// ABL: compile-time ablation mask for tools/microbench_b16.hip (0 in the library): 1 = no global prefetch,
// 2 = no LDS restage + barriers, 4 = no MFMA, 8 = no LDS operand reads, 16 = no epilogue stores.
template <int COUT, int MW, int NW, int WM, int WN, int NS, int DT = 0, int ABL = 0>
__global__ __launch_bounds__(WM* WN * 64, 2) void conv1d_k9_bf16s_kernel(ConvB16Args a) {
  static_assert(WN * NW * 32 == COUT, "cout tiling");
  constexpr int NT = WM * WN * 64;
  constexpr int MT = WM * MW * 32;
  constexpr int XROW = MT + 8;
  constexpr int XU = NS * 2 * XROW;        // 16-byte units of the X image
  constexpr int WU = NS * 9 * 2 * COUT;    // 16-byte units of the W image
  constexpr int XF4 = XROW * 4;            // float4 loads per chunk (16 channels = 4 quads per position)
  constexpr int XIT = (XF4 + NT - 1) / NT;
  constexpr int WIT = (WU + NT - 1) / NT;
  constexpr int NPROD = NS == 3 ? 6 : (NS == 2 ? 3 : 1);

  __shared__ f32x4 smem[XU + WU + COUT / 4];   // + bias (read with ds_read in the epilogue: a global load there
                                               // would make the epilogue wait for the whole in-flight prefetch)

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int l31 = lane & 31, g = lane >> 5;
  float* bias_s = reinterpret_cast<float*>(smem + XU + WU);
  if (tid < COUT) bias_s[tid] = a.bias[tid];   // visible after the first barrier
  const f32x4* wg = reinterpret_cast<const f32x4*>(a.w);
  const long ntiles = a.tiles_per_row * a.batch;

  long tile = blockIdx.x;
  if (tile >= ntiles) return;
  // De-phase the workgroups that share a CU: identical workgroups started together stay in lockstep
  // (both in their MFMA block, then both in their staging block, matrix pipe idle).  ABL bits 32/64 pick
  // the stagger rule in the micro-benchmark.
  if ((ABL == 0 || (ABL & 32)) && gridDim.x >= 512 && blockIdx.x >= gridDim.x / 2) { for (int k = 0; k < a.stagger; ++k) __builtin_amdgcn_s_sleep(64); }
  if ((ABL & 64) && (blockIdx.x & 1)) { for (int k = 0; k < a.stagger; ++k) __builtin_amdgcn_s_sleep(64); }

  f32x16 acc[MW][NW];
#pragma unroll
  for (int i = 0; i < MW; ++i)
#pragma unroll
    for (int j = 0; j < NW; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  bool overflow = false;
  // ---- staging registers (thread-constant geometry: which float4 of the X window, where it lands) ----
  f32x4 xr[XIT], wr[WIT];
  int xprel[XIT], xq4[XIT], xdst[XIT];
#pragma unroll
  for (int it = 0; it < XIT; ++it) {
    int u = tid + it * NT;
    u = u < XF4 ? u : XF4 - 1;
    const int prel = u >> 2, q = u & 3;
    xprel[it] = prel;
    xq4[it] = 4 * q;
    xdst[it] = ((q >> 1) * XROW + prel) * 16 + (q & 1) * 8;
  }

  // global -> registers for (tile t, chunk c)
#define B16_LOAD_CHUNK(t, c)                                                                 \
  {                                                                                          \
    const long tb_ = (t) / a.tiles_per_row;                                                  \
    const long tm0_ = ((t) - tb_ * a.tiles_per_row) * MT;                                    \
    const float* xb_ = a.x + tb_ * a.x_bs + 16 * (c);                                        \
    _Pragma("unroll") for (int it = 0; it < XIT; ++it) {                                     \
      const long pos = tm0_ - 4 + xprel[it];                                                 \
      f32x4 v = (f32x4)(0.f);                                                                \
      if (pos >= 0 && pos < a.n) v = *reinterpret_cast<const f32x4*>(xb_ + pos * (long)a.cin + xq4[it]); \
      xr[it] = v;                                                                            \
    }                                                                                        \
    const f32x4* wc = wg + (long)(c) * ((DT == 1 ? 2 : 3) * 9 * 2 * COUT); /* pack: 3 bf16 / 2 fp16 splits */ \
    _Pragma("unroll") for (int it = 0; it < WIT; ++it) {                                     \
      int idx = tid + it * NT;                                                               \
      idx = idx < WU ? idx : WU - 1;                                                         \
      wr[it] = wc[idx];                                                                      \
    }                                                                                        \
  }
  // registers -> LDS (fp32 -> NS bf16 splits on the way)
#define B16_STORE_CHUNK()                                                                    \
  {                                                                                          \
    char* xs = reinterpret_cast<char*>(smem);                                                \
    _Pragma("unroll") for (int it = 0; it < XIT; ++it) {                                     \
      if (tid + it * NT < XF4) {                                                             \
        u32x2 sp[NS];                                                                        \
        split4<NS, DT>(xr[it], sp, overflow);                                                              \
        _Pragma("unroll") for (int s = 0; s < NS; ++s)                                       \
            *reinterpret_cast<u32x2*>(xs + s * (2 * XROW * 16) + xdst[it]) = sp[s];          \
      }                                                                                      \
    }                                                                                        \
    _Pragma("unroll") for (int it = 0; it < WIT; ++it) {                                     \
      const int idx = tid + it * NT;                                                         \
      if (idx < WU) smem[XU + idx] = wr[it];                                                 \
    }                                                                                        \
  }

  B16_LOAD_CHUNK(tile, 0);
  B16_STORE_CHUNK();
  __syncthreads();

  const f32x4* xa0 = smem + g * XROW + wm * (MW * 32) + l31;           // + s*2*XROW + mi*32 + tap
  const f32x4* wb0 = smem + XU + g * COUT + wn * (NW * 32) + l31;      // + ((s*9+tap)*2)*COUT + ni*32

  int c = 0;
  while (true) {
    // what comes after (tile, c) in the stream
    const bool last_chunk = (c + 1 == a.nchunks);
    const long ntile = last_chunk ? tile + gridDim.x : tile;
    const int nc = last_chunk ? 0 : c + 1;
    const bool more = ntile < ntiles;
    if (more && !(ABL & 1)) B16_LOAD_CHUNK(ntile, nc);

    typename Op16<DT>::vec av[NS][MW], bv[NS][NW];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      if (!(ABL & 8) || tap == 0)
#pragma unroll
      for (int s = 0; s < NS; ++s) {
#pragma unroll
        for (int i = 0; i < MW; ++i) av[s][i] = __builtin_bit_cast(typename Op16<DT>::vec, xa0[s * 2 * XROW + i * 32 + tap]);
#pragma unroll
        for (int j = 0; j < NW; ++j) bv[s][j] = __builtin_bit_cast(typename Op16<DT>::vec, wb0[((s * 9 + tap) * 2) * COUT + j * 32]);
      }
      if (ABL & 4) {
#pragma unroll
        for (int s = 0; s < NS; ++s) {
#pragma unroll
          for (int i = 0; i < MW; ++i) asm volatile("" ::"v"(av[s][i]));
#pragma unroll
          for (int j = 0; j < NW; ++j) asm volatile("" ::"v"(bv[s][j]));
        }
      } else
#pragma unroll
      for (int p = 0; p < NPROD; ++p) {
        // product list ordered small -> large so the dominant x1*w1 term is added last
        constexpr int PA3[6] = {2, 1, 0, 1, 0, 0}, PB3[6] = {0, 1, 2, 0, 1, 0};
        constexpr int PA2[3] = {1, 0, 0}, PB2[3] = {0, 1, 0};
        const int sa = NS == 3 ? PA3[p] : (NS == 2 ? PA2[p] : 0);
        const int sb = NS == 3 ? PB3[p] : (NS == 2 ? PB2[p] : 0);
#pragma unroll
        for (int i = 0; i < MW; ++i)
#pragma unroll
          for (int j = 0; j < NW; ++j)
            acc[i][j] = Op16<DT>::mfma(bv[sb][j], av[sa][i], acc[i][j]);  // D[cout][pos]
      }
    }

    if (last_chunk) {
      // ---- epilogue.  The MFMA was issued as D = W-tile (rows = cout) x X-tile (cols = pos), so a lane owns
      // ONE position (l&31) and, per register group q, 4 CONSECUTIVE couts 8q+4g..+3 of each 32-cout
      // subtile: bias / ReLU / residual / store are float4 along the channel axis of the channel-last output.
      const long tb = tile / a.tiles_per_row;
      const long m0 = (tile - tb * a.tiles_per_row) * MT;
      float* yb = a.y + tb * a.y_bs;
      const float* rb = a.r1 ? a.r1 + tb * a.r_bs : nullptr;
      const float* rb2 = a.r2 ? a.r2 + tb * a.r_bs : nullptr;
#pragma unroll
      for (int i = 0; i < MW; ++i) {
        const long pos = m0 + wm * (MW * 32) + i * 32 + l31;
#pragma unroll
        for (int j = 0; j < NW; ++j) {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int co = wn * (NW * 32) + j * 32 + 8 * q + 4 * g;
            const f32x4 bias = *reinterpret_cast<const f32x4*>(bias_s + co);
            f32x4 v;
            v.x = acc[i][j][4 * q + 0] + bias.x;
            v.y = acc[i][j][4 * q + 1] + bias.y;
            v.z = acc[i][j][4 * q + 2] + bias.z;
            v.w = acc[i][j][4 * q + 3] + bias.w;
            acc[i][j][4 * q + 0] = 0.f; acc[i][j][4 * q + 1] = 0.f; acc[i][j][4 * q + 2] = 0.f; acc[i][j][4 * q + 3] = 0.f;
            if (a.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
            if (!a.pool4) {
              if (pos < a.n) {
                const long o = pos * COUT + co;
                if (rb) v += *reinterpret_cast<const f32x4*>(rb + o);
                if (rb2) v += *reinterpret_cast<const f32x4*>(rb2 + o);     // may alias y: read before this lane's own store
                if (!(ABL & 16) || v.x == 12345.678f) *reinterpret_cast<f32x4*>(yb + o) = v;
              }
            } else {
              // residual add, then max over the 4 consecutive positions held by lanes 4k..4k+3 (quad
              // butterflies), lane 4k stores the pooled row: the next stage's MaxPool1d(4) never runs.
              if (rb && pos < a.n) v += *reinterpret_cast<const f32x4*>(rb + pos * COUT + co);
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                float t = v[e];
                t = fmaxf(t, __shfl_xor(t, 1));
                t = fmaxf(t, __shfl_xor(t, 2));
                v[e] = t;
              }
              if ((l31 & 3) == 0 && pos + 3 < a.n) *reinterpret_cast<f32x4*>(yb + (pos >> 2) * COUT + co) = v;
            }
          }
        }
      }
    }

    if (!more) break;
    if (!(ABL & 2)) {
      __syncthreads();
      B16_STORE_CHUNK();
      __syncthreads();
    }
    tile = ntile;
    c = nc;
  }
#undef B16_LOAD_CHUNK
#undef B16_STORE_CHUNK
  if (DT == 1 && overflow && a.flag) *a.flag = 1u;
}

// nn.MaxPool1d(k,k) on channel-last data: y[m][c] = max_j x[k*m+j][c]; one thread = 4 channels of TWO outputs (2K independent
// 16-byte loads in flight per thread), 32-bit index arithmetic (C / 4 divides the block: grid = ceil(n_out / (2 * 256 / (C/4)))).
template <int K>
__global__ void maxpool1d_nlc_kernel(const float* __restrict__ x, float* __restrict__ y, long n_out, int C) {
  const int c4n = C / 4, per = 256 / c4n;
  const int c4 = threadIdx.x % c4n, r = threadIdx.x / c4n;
  if (r >= per) return;
  const long m0 = ((long)blockIdx.x * 2) * per + r, m1 = m0 + per;
  if (m0 >= n_out) return;
  const bool two = m1 < n_out;
  const f32x4* p = reinterpret_cast<const f32x4*>(x + ((long)K * m0) * C) + c4;
  const f32x4* q = reinterpret_cast<const f32x4*>(x + ((long)K * (two ? m1 : m0)) * C) + c4;
  f32x4 a[K], b[K];
#pragma unroll
  for (int j = 0; j < K; ++j) { a[j] = p[(long)j * c4n]; b[j] = q[(long)j * c4n]; }
  f32x4 v = a[0], w = b[0];
#pragma unroll
  for (int j = 1; j < K; ++j) {
    v.x = fmaxf(v.x, a[j].x); v.y = fmaxf(v.y, a[j].y); v.z = fmaxf(v.z, a[j].z); v.w = fmaxf(v.w, a[j].w);
    w.x = fmaxf(w.x, b[j].x); w.y = fmaxf(w.y, b[j].y); w.z = fmaxf(w.z, b[j].z); w.w = fmaxf(w.w, b[j].w);
  }
  reinterpret_cast<f32x4*>(y + m0 * C)[c4] = v;
  if (two) reinterpret_cast<f32x4*>(y + m1 * C)[c4] = w;
}
