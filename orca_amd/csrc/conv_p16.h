// conv_p16.h - the Encoder's hot convolutions (stages 1-3, 96 % of its FLOPs) on "P16" activations.
//
// P16 = planar 2-way-split fp16 storage of an fp32 activation tensor [n][C]:
//     value(pos, ch) = hi + lo,   hi = fp16(v), lo = fp16(v - hi)            (22 significant bits)
//     plane (p = ch/8, s in {hi,lo}) is a contiguous array of 16-byte units, one unit per position holding the
//     8 channels 8p..8p+7;  unit(p, s, pos) at  base + ((2p + s) * PLEN + 4 + pos) * 16 bytes.
//   Same 4 bytes per element as fp32, but it IS the MFMA operand image: a conv's input tile is a set of
//   contiguous 16-byte runs that `global_load_lds` (LDS-DMA) drops straight into the LDS operand image - no
//   staging registers, no VALU conversion, no ds_write pass.  Each plane carries 4 guard units on the left and
//   >= 4 on the right that are kept ZERO (p16_zero_pads_kernel + producers), so the conv's zero padding and the
//   ragged last tile need no predication at all.
//
// Kernel: persistent workgroups, 8 waves, (tile, chunk) stream as in conv_bf16s.h, but with TWO LDS buffers:
// the DMA of step s+1 is issued before the MFMA block of step s and lands underneath it; one barrier per step.
// (Tried and dropped: a de-phased two-group variant - waves 0-3 / 4-7 on different tiles, group B walking its
// K-chunks in rotated order so both share the W chunk - was correct but 15 % slower: with one barrier per step
// the step time becomes max(group in epilogue, group in MFMA) twice per tile instead of once.)
// Arithmetic: 3 fp16 MFMA products per fp32 product (hi*hi + hi*lo + lo*hi), fp32 accumulate.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "conv_bf16s.h"

#define P16_GUARD 4

struct ConvP16Args {
  const f32x4* x;      // P16 input, cin channels
  const f32x4* w;      // fp16 pack [cin/16][2][9][2][cout][8]  (units of 16 B)
  const float* bias;
  void* y;             // output: P16 (out_mode 0/1) or fp32 channel-last [n][cout] (out_mode 2)
  const f32x4* r1;     // optional residual, P16 with cout channels and the INPUT's plane length
  long x_plen, y_plen; // plane lengths (16-byte units) of x (and r1) / y
  long n;              // valid positions of x
  long tiles_per_row;  // ceil(n / MT)
  int nchunks;         // cin / 16
  int cout;            // total output channels (multiple of CT)
  int relu;
  int out_mode;        // 0: P16 same length; 1: P16 with MaxPool1d(4) fused (length n/4); 2: fp32 [n][cout]
  unsigned* flag;      // raised when a value written to P16 leaves the fp16 range
};

__device__ __forceinline__ void p16_split_store(char* plane_hi, long plen_bytes, f32x4 v, bool valid, bool& ovf) {
  // 4 consecutive channels of one position -> 8 bytes in the hi plane and 8 bytes in the lo plane
  u32x2 sp[2];
  if (!valid) v = (f32x4)(0.f);
  split4<2, 1>(v, sp, ovf);
  *reinterpret_cast<u32x2*>(plane_hi) = sp[0];
  *reinterpret_cast<u32x2*>(plane_hi + plen_bytes) = sp[1];
}

typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4 p16_load4(const char* plane_hi, long plen_bytes) {
  const f16x4 h = *reinterpret_cast<const f16x4*>(plane_hi), l = *reinterpret_cast<const f16x4*>(plane_hi + plen_bytes);
  f32x4 v;
  v.x = (float)h[0] + (float)l[0];
  v.y = (float)h[1] + (float)l[1];
  v.z = (float)h[2] + (float)l[2];
  v.w = (float)h[3] + (float)l[3];
  return v;
}

// 16-byte LDS-DMA: each lane supplies its own global address, the LDS destination is wave-uniform base +
// lane*16.  (The builtin only exists in the device pass; the host pass just needs the kernel to parse.)
__device__ __forceinline__ void p16_glds16(const f32x4* gsrc, f32x4* lds_wave_base) {
#if defined(__HIP_DEVICE_COMPILE__)
  __builtin_amdgcn_global_load_lds(gsrc, lds_wave_base, 16, 0, 0);
#else
  (void)gsrc; (void)lds_wave_base;
#endif
}

// CT = couts per workgroup tile (cout blocks of CT are separate tiles), wave tile = MW x NW subtiles of 32x32.
// ABL (micro-benchmark only, 0 in the library): 1 = no DMA after the first step, 4 = no MFMA, 8 = LDS operands
// read once, 16 = no epilogue stores, 32 = no epilogue at all.
template <int CT, int MW, int NW, int WM, int ABL = 0>
__global__ __launch_bounds__(WM * 64, 2) void conv1d_k9_p16_kernel(ConvP16Args a) {
  static_assert(NW * 32 == CT, "one wave covers all couts of the tile");
  constexpr int NT = WM * 64;
  constexpr int MT = WM * MW * 32;
  constexpr int XROW = MT + 8;
  constexpr int XU = 2 * 2 * XROW;      // X image units  [s][g][XROW]
  constexpr int WU = 2 * 9 * 2 * CT;    // W image units  [s][tap][g][CT]
  constexpr int BU = XU + WU;           // one buffer
  constexpr int NIT = (BU + NT - 1) / NT;
  __shared__ f32x4 smem[2 * BU + 32];   // + bias of all couts (<= 128 floats), read with ds_read in the epilogue

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, g = lane >> 5;
  const int ncb = a.cout / CT;                       // cout blocks
  const long ntiles = a.tiles_per_row * ncb;
  long tile = blockIdx.x;
  if (tile >= ntiles) return;
  bool overflow = false;

  f32x16 acc[MW][NW];
#pragma unroll
  for (int i = 0; i < MW; ++i)
#pragma unroll
    for (int j = 0; j < NW; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  float* bias_s = reinterpret_cast<float*>(smem + 2 * BU);
  if (tid < a.cout) bias_s[tid] = a.bias[tid];   // visible after the first barrier

  // thread-constant DMA geometry: unit i = tid + it*NT of the buffer image
  long xrel[NIT];   // X: (g*2 + s) * x_plen + col        W: grp * cout + cc     (in 16-byte units)
  bool isx[NIT], act[NIT];
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int i = tid + it * NT;
    act[it] = i < BU;
    isx[it] = i < XU;
    if (isx[it]) {
      const int row = i / XROW, col = i - row * XROW;      // row = s*2 + g
      const int s = row >> 1, gg = row & 1;
      xrel[it] = (long)(gg * 2 + s) * a.x_plen + col;
    } else {
      const int u = (i < BU ? i : BU - 1) - XU;
      const int grp = u / CT, cc = u - grp * CT;           // grp = (s*9 + tap)*2 + g
      xrel[it] = (long)grp * a.cout + cc;
    }
  }
  const long wchunk = (long)2 * 9 * 2 * a.cout;            // units per K-chunk in the weight pack

  // LDS-DMA of (tile t, chunk c) into buffer `buf`
#define P16_DMA(t, c, buf)                                                                         \
  {                                                                                                \
    const long tcb_ = (t) / a.tiles_per_row;                                                       \
    const long tm0_ = ((t) - tcb_ * a.tiles_per_row) * MT;                                         \
    const f32x4* xsrc_ = a.x + (long)(c) * 4 * a.x_plen + tm0_; /* (pos - 4 + GUARD) = tm0 + col */ \
    const f32x4* wsrc_ = a.w + (long)(c) * wchunk + tcb_ * CT;                                     \
    _Pragma("unroll") for (int it = 0; it < NIT; ++it) {                                           \
      if (act[it]) {                                                                               \
        const f32x4* src_ = (isx[it] ? xsrc_ : wsrc_) + xrel[it];                                  \
        p16_glds16(src_, smem + (buf) * BU + it * NT + wave * 64);                                 \
      }                                                                                            \
    }                                                                                              \
  }

  long epi_tile = -1;   // finished tile whose accumulators still await their epilogue (-1: none)

  // Epilogue of the finished tile: a lane owns ONE position and 4 consecutive couts per register group q.
  // It runs at the START of the next step (after the barrier that drained this step's DMA), so its stores -
  // the output is as large as the input, this kernel sits at the HBM/MFMA ridge - drain underneath that step's
  // DMA + MFMA block and are retired by the step's closing barrier.
#define P16_EPILOGUE()                                                                                           \
  {                                                                                                              \
    const long tcb = epi_tile / a.tiles_per_row;                                                                 \
    const long m0 = (epi_tile - tcb * a.tiles_per_row) * MT;                                                     \
    const long xpl = a.x_plen * 16, ypl = a.y_plen * 16;                                                         \
    _Pragma("unroll") for (int i = 0; i < MW; ++i) _Pragma("unroll") for (int j = 0; j < NW; ++j) {              \
      const long pos = m0 + wave * (MW * 32) + i * 32 + l31;                                                     \
      _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                                            \
        const int co = (int)tcb * CT + j * 32 + 8 * q + 4 * g;                                                   \
        const f32x4 bias = *reinterpret_cast<const f32x4*>(bias_s + co);                                         \
        f32x4 v;                                                                                                 \
        v.x = acc[i][j][4 * q + 0] + bias.x;                                                                     \
        v.y = acc[i][j][4 * q + 1] + bias.y;                                                                     \
        v.z = acc[i][j][4 * q + 2] + bias.z;                                                                     \
        v.w = acc[i][j][4 * q + 3] + bias.w;                                                                     \
        acc[i][j][4 * q + 0] = 0.f; acc[i][j][4 * q + 1] = 0.f; acc[i][j][4 * q + 2] = 0.f; acc[i][j][4 * q + 3] = 0.f; \
        if (a.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); } \
        const bool valid = pos < a.n;                                                                            \
        const long pl = (long)(co >> 3) * 2;   /* hi plane index of this cout octet */                           \
        if (a.r1 && valid)                                                                                       \
          v += p16_load4(reinterpret_cast<const char*>(a.r1) + pl * xpl + (P16_GUARD + pos) * 16 + g * 8, xpl);  \
        if (ABL & 16) {                                                                                          \
          asm volatile("" ::"v"(v));                                                                             \
        } else if (a.out_mode == 0) {                                                                            \
          p16_split_store(reinterpret_cast<char*>(a.y) + pl * ypl + (P16_GUARD + pos) * 16 + g * 8, ypl, v, valid, overflow); \
        } else if (a.out_mode == 1) {                                                                            \
          if (!valid) v = (f32x4)(-3.0e38f);                                                                     \
          _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                                        \
            float t = v[e];                                                                                      \
            t = fmaxf(t, __shfl_xor(t, 1));                                                                      \
            t = fmaxf(t, __shfl_xor(t, 2));                                                                      \
            v[e] = t;                                                                                            \
          }                                                                                                      \
          if ((l31 & 3) == 0)                                                                                    \
            p16_split_store(reinterpret_cast<char*>(a.y) + pl * ypl + (P16_GUARD + (pos >> 2)) * 16 + g * 8, ypl, v, \
                            pos + 3 < a.n, overflow);                                                            \
        } else if (valid) {                                                                                      \
          *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(a.y) + pos * a.cout + co) = v;                      \
        }                                                                                                        \
      }                                                                                                          \
    }                                                                                                            \
  }

  P16_DMA(tile, 0, 0);
  __syncthreads();   // drains vmcnt (the DMA) and joins the waves

  int c = 0, cur = 0;
  while (true) {
    const bool last_chunk = (c + 1 == a.nchunks);
    const long ntile = last_chunk ? tile + gridDim.x : tile;
    const int nc = last_chunk ? 0 : c + 1;
    const bool more = ntile < ntiles;
    if (epi_tile >= 0) {
      if (!(ABL & 32)) P16_EPILOGUE();
      epi_tile = -1;
    }
    if (more && !(ABL & 1)) P16_DMA(ntile, nc, cur ^ 1);

    const f32x4* xa0 = smem + cur * BU + g * XROW + wave * (MW * 32) + l31;   // + s*2*XROW + i*32 + tap
    const f32x4* wb0 = smem + cur * BU + XU + g * CT + l31;                    // + ((s*9+tap)*2)*CT + j*32
    // operand fragments are double-buffered across taps: tap t+1 is read from LDS while tap t feeds the MFMAs
    f16x8 av[2][2][MW], bv[2][2][NW];
#define P16_READ_FRAGS(buf_, tap_)                                                                                 \
  _Pragma("unroll") for (int s = 0; s < 2; ++s) {                                                                  \
    _Pragma("unroll") for (int i = 0; i < MW; ++i) av[buf_][s][i] = __builtin_bit_cast(f16x8, xa0[s * 2 * XROW + i * 32 + (tap_)]); \
    _Pragma("unroll") for (int j = 0; j < NW; ++j) bv[buf_][s][j] = __builtin_bit_cast(f16x8, wb0[((s * 9 + (tap_)) * 2) * CT + j * 32]); \
  }
    P16_READ_FRAGS(0, 0);
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int fb = tap & 1;
      if (tap + 1 < 9 && !(ABL & 8)) P16_READ_FRAGS(fb ^ 1, tap + 1);
      if (ABL & 4) {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
#pragma unroll
          for (int i = 0; i < MW; ++i) asm volatile("" ::"v"(av[fb][s][i]));
#pragma unroll
          for (int j = 0; j < NW; ++j) asm volatile("" ::"v"(bv[fb][s][j]));
        }
      } else
#pragma unroll
      for (int p = 0; p < 3; ++p) {
        constexpr int PA[3] = {1, 0, 0}, PB[3] = {0, 1, 0};   // lo*hi, hi*lo, hi*hi (largest last)
#pragma unroll
        for (int i = 0; i < MW; ++i)
#pragma unroll
          for (int j = 0; j < NW; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bv[fb][PB[p]][j], av[fb][PA[p]][i], acc[i][j], 0, 0, 0);  // D[cout][pos]
      }
    }
#undef P16_READ_FRAGS

    if (last_chunk) epi_tile = tile;

    if (!more) break;
    __syncthreads();   // next buffer has landed (vmcnt drained), everyone is done reading the current one
    tile = ntile;
    c = nc;
    cur ^= 1;
  }
  if (epi_tile >= 0 && !(ABL & 32)) P16_EPILOGUE();   // the last tile
#undef P16_EPILOGUE
#undef P16_DMA
  if (overflow && a.flag) *a.flag = 1u;
}

// zero the guard / tail units of every plane: [0,4) and [4 + n_valid, plen)
__global__ void p16_zero_pads_kernel(f32x4* __restrict__ base, long plen, long n_valid) {
  f32x4* pl = base + (long)blockIdx.x * plen;
  const long tail0 = P16_GUARD + n_valid;
  for (long i = threadIdx.x; i < P16_GUARD + (plen - tail0); i += blockDim.x) {
    const long u = i < P16_GUARD ? i : tail0 + (i - P16_GUARD);
    pl[u] = (f32x4)(0.f);
  }
}

// ---- first layer: Conv1d(4,64,k9,p4)+BN straight from the [L][4] float sequence to P16 ------------------
// K = 36 only (1 % of the Encoder FLOPs): plain fp32 FMAs.  One thread = one position x one cout octet.
struct FirstP16Args {
  const float* x;   // element strides sc (channel), sl (position)
  long sc, sl, n;
  const float* w;   // [64][4][9] folded
  const float* bias;
  f32x4* y;         // P16, 64 channels
  long y_plen;
  unsigned* flag;
};

__global__ __launch_bounds__(256) void conv1d_first_p16_kernel(FirstP16Args a) {
  __shared__ float ws[36 * 8];   // [tap*4+ci][8 couts of this octet]
  __shared__ float bs[8];
  const int oct = blockIdx.y;
  for (int t = threadIdx.x; t < 288; t += 256) {
    const int k = t >> 3, e = t & 7;               // k = tap*4 + ci
    ws[t] = a.w[((oct * 8 + e) * 4 + (k & 3)) * 9 + (k >> 2)];
  }
  if (threadIdx.x < 8) bs[threadIdx.x] = a.bias[oct * 8 + threadIdx.x];
  __syncthreads();
  const long pos = (long)blockIdx.x * 256 + threadIdx.x;
  if (pos >= a.n) return;
  float xin[36];
  const bool fast = (a.sc == 1 && a.sl == 4);
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    const long p = pos + t - 4;
    const bool ok = (p >= 0 && p < a.n);
    if (fast) {
      f32x4 v = (f32x4)(0.f);
      if (ok) v = *reinterpret_cast<const f32x4*>(a.x + p * 4);
      xin[4 * t + 0] = v.x; xin[4 * t + 1] = v.y; xin[4 * t + 2] = v.z; xin[4 * t + 3] = v.w;
    } else {
#pragma unroll
      for (int ci = 0; ci < 4; ++ci) xin[4 * t + ci] = ok ? a.x[p * a.sl + ci * a.sc] : 0.f;
    }
  }
  float o[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) o[e] = bs[e];
#pragma unroll
  for (int k = 0; k < 36; ++k) {
    const f32x4 w0 = *reinterpret_cast<const f32x4*>(ws + k * 8), w1 = *reinterpret_cast<const f32x4*>(ws + k * 8 + 4);
    o[0] = fmaf(w0.x, xin[k], o[0]); o[1] = fmaf(w0.y, xin[k], o[1]); o[2] = fmaf(w0.z, xin[k], o[2]); o[3] = fmaf(w0.w, xin[k], o[3]);
    o[4] = fmaf(w1.x, xin[k], o[4]); o[5] = fmaf(w1.y, xin[k], o[5]); o[6] = fmaf(w1.z, xin[k], o[6]); o[7] = fmaf(w1.w, xin[k], o[7]);
  }
  char* pl = reinterpret_cast<char*>(a.y) + (long)oct * 2 * a.y_plen * 16 + (P16_GUARD + pos) * 16;
  f32x4 lo4, hi4;
  lo4.x = o[0]; lo4.y = o[1]; lo4.z = o[2]; lo4.w = o[3];
  hi4.x = o[4]; hi4.y = o[5]; hi4.z = o[6]; hi4.w = o[7];
  bool ovf = false;
  p16_split_store(pl, a.y_plen * 16, lo4, true, ovf);
  p16_split_store(pl + 8, a.y_plen * 16, hi4, true, ovf);
  if (ovf && a.flag) *a.flag = 1u;
}

// ---- first layer on the matrix cores ----------------------------------------------------------------------
// With the sequence stored [L][4] the 9-tap x 4-channel window of a position is 36 CONTIGUOUS floats, so
// Conv1d(4,64,k9) is a GEMM with K = tap*4+ci = 36 (padded to 48 = three k16 steps; the pad weights are zero).
// X operand of lane (pos, g) for k-step kk = the 8 halves at flat index 4*(pos-4) + 16*kk + 8*g of the split
// window image (8-byte aligned: two ds_read_b64).  Persistent; W stays in LDS; the next tile's window is
// prefetched to registers under the current tile's MFMAs + epilogue.  HBM-write bound (8 GB P16 output).
struct FirstMfmaArgs {
  const float* x;     // [n][4] contiguous fp32, or NULL when `codes` is given
  // packed input: 1 byte per base (0..3 = A,C,G,T one-hot rows, 4 = N = 0.25 x 4, other = zero row).  The chunk's
  // position p is strand position off+p; on the reverse strand that is base L-1-(off+p), complemented (3-code).
  const unsigned char* codes;
  long codes_L, codes_off;
  int reverse;
  long n;
  const f32x4* w;     // fp16 pack [2 splits][3 ksteps][2 g][64 couts][8]  (units of 16 B), k >= 36 zero
  const float* bias;
  f32x4* y;           // P16, 64 channels
  long y_plen;
  unsigned* flag;
};

__global__ __launch_bounds__(256, 2) void conv1d_first_mfma_p16_kernel(FirstMfmaArgs a) {
  constexpr int MT = 256, WIN = MT + 12;          // positions m0-4 .. m0+MT+7 (9-tap window + k padding overrun)
  constexpr int WU = 2 * 3 * 2 * 64;              // 768 units
  __shared__ f32x4 wsm[WU];
  __shared__ u32x2 xsm[2][WIN];                   // [split][position] -> 4 halves (the 4 channels)
  __shared__ float bias_s[64];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, g = lane >> 5;
  const long ntiles = (a.n + MT - 1) / MT;
  for (int i = tid; i < WU; i += 256) wsm[i] = a.w[i];
  if (tid < 64) bias_s[tid] = a.bias[tid];
  bool overflow = false;

  f32x4 xr[2];
  auto fetch = [&](long p) -> f32x4 {
    if (p < 0 || p >= a.n) return (f32x4)(0.f);
    if (!a.codes) return *reinterpret_cast<const f32x4*>(a.x + p * 4);
    const long P = a.codes_off + p;
    int c = a.reverse ? a.codes[a.codes_L - 1 - P] : a.codes[P];
    if (a.reverse && c < 4) c = 3 - c;
    f32x4 v = (f32x4)(c == 4 ? 0.25f : 0.f);   // LDS-staged one-hot expansion of the packed base
    if (c < 4) v[c] = 1.f;
    return v;
  };
  auto load_win = [&](long t, f32x4& r0, f32x4& r1) {
    const long p0 = t * MT - 4 + tid;
    r0 = fetch(p0);
    r1 = (tid < WIN - 256) ? fetch(p0 + 256) : (f32x4)(0.f);
  };
  long tile = blockIdx.x;
  if (tile >= ntiles) return;
  load_win(tile, xr[0], xr[1]);
  for (; tile < ntiles; tile += gridDim.x) {
    __syncthreads();   // previous tile's LDS reads are done
    {
      u32x2 sp[2];
      split4<2, 1>(xr[0], sp, overflow);
      xsm[0][tid] = sp[0]; xsm[1][tid] = sp[1];
      if (tid < WIN - 256) {
        split4<2, 1>(xr[1], sp, overflow);
        xsm[0][256 + tid] = sp[0]; xsm[1][256 + tid] = sp[1];
      }
    }
    __syncthreads();
    if (tile + gridDim.x < ntiles) load_win(tile + gridDim.x, xr[0], xr[1]);

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
#pragma unroll
    for (int kk = 0; kk < 3; ++kk) {
      f16x8 xv[2][2], wv[2][2];
#pragma unroll
      for (int s = 0; s < 2; ++s) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int pr = wave * 64 + i * 32 + l31 + 4 * kk + 2 * g;   // window position holding k = 16kk + 8g
          const u32x2 lo = xsm[s][pr], hi = xsm[s][pr + 1];
          u32x4_t q = {lo.x, lo.y, hi.x, hi.y};
          xv[s][i] = __builtin_bit_cast(f16x8, q);
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) wv[s][j] = __builtin_bit_cast(f16x8, wsm[((s * 3 + kk) * 2 + g) * 64 + j * 32 + l31]);
      }
#pragma unroll
      for (int p = 0; p < 3; ++p) {
        constexpr int PA[3] = {1, 0, 0}, PB[3] = {0, 1, 0};
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wv[PB[p]][j], xv[PA[p]][i], acc[i][j], 0, 0, 0);
      }
    }
    const long ypl = a.y_plen * 16;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const long pos = tile * MT + wave * 64 + i * 32 + l31;
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int co = j * 32 + 8 * q + 4 * g;
          const f32x4 bias = *reinterpret_cast<const f32x4*>(bias_s + co);
          f32x4 v;
          v.x = acc[i][j][4 * q + 0] + bias.x; v.y = acc[i][j][4 * q + 1] + bias.y;
          v.z = acc[i][j][4 * q + 2] + bias.z; v.w = acc[i][j][4 * q + 3] + bias.w;
          if (pos < a.n)
            p16_split_store(reinterpret_cast<char*>(a.y) + (long)(co >> 3) * 2 * ypl + (P16_GUARD + pos) * 16 + g * 8, ypl, v, true, overflow);
        }
    }
  }
  if (overflow && a.flag) *a.flag = 1u;
}

// ---- packed sequence helpers --------------------------------------------------------------------------------
// [L][4] float rows (element strides sc, sl) -> 1-byte codes; *bad is raised for rows that are neither one-hot nor N
__global__ void pack_sequence_kernel(const float* __restrict__ x, long sc, long sl, long L, unsigned char* __restrict__ codes,
                                     unsigned* __restrict__ bad) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= L) return;
  float v[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) v[c] = x[i * sl + c * sc];
  int code = 5;
  if (v[0] == 0.25f && v[1] == 0.25f && v[2] == 0.25f && v[3] == 0.25f) code = 4;
  else {
    int ones = 0, zeros = 0, which = 0;
#pragma unroll
    for (int c = 0; c < 4; ++c) { if (v[c] == 1.f) { ++ones; which = c; } else if (v[c] == 0.f) ++zeros; }
    if (ones == 1 && zeros == 3) code = which;
  }
  if (code == 5) *bad = 1u;
  codes[i] = (unsigned char)code;
}
// codes -> [n][4] fp32 rows of strand positions off .. off+n-1 (used by the non-P16 arithmetic modes)
__global__ void expand_codes_kernel(const unsigned char* __restrict__ codes, long L, long off, int reverse, long n, float* __restrict__ y) {
  const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  const long P = off + p;
  int c = reverse ? codes[L - 1 - P] : codes[P];
  if (reverse && c < 4) c = 3 - c;
  f32x4 v = (f32x4)(c == 4 ? 0.25f : 0.f);
  if (c < 4) v[c] = 1.f;
  reinterpret_cast<f32x4*>(y)[p] = v;
}

// ---- converters (tests, and the stage 3 -> 4 hand-over) -------------------------------------------------
// fp32 channel-last [n][C] -> P16
__global__ void nlc_to_p16_kernel(const float* __restrict__ x, f32x4* __restrict__ y, long n, int C, long plen) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int c4n = C / 4;
  if (idx >= n * c4n) return;
  const long pos = idx / c4n;
  const int c4 = (int)(idx - pos * c4n);
  const f32x4 v = *reinterpret_cast<const f32x4*>(x + pos * C + 4 * c4);
  bool ovf = false;
  p16_split_store(reinterpret_cast<char*>(y) + (long)(c4 >> 1) * 2 * plen * 16 + (P16_GUARD + pos) * 16 + (c4 & 1) * 8, plen * 16, v, true, ovf);
}
// P16 -> fp32 channel-last [n][C]
__global__ void p16_to_nlc_kernel(const f32x4* __restrict__ x, float* __restrict__ y, long n, int C, long plen) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int c4n = C / 4;
  if (idx >= n * c4n) return;
  const long pos = idx / c4n;
  const int c4 = (int)(idx - pos * c4n);
  *reinterpret_cast<f32x4*>(y + pos * C + 4 * c4) =
      p16_load4(reinterpret_cast<const char*>(x) + (long)(c4 >> 1) * 2 * plen * 16 + (P16_GUARD + pos) * 16 + (c4 & 1) * 8, plen * 16);
}
