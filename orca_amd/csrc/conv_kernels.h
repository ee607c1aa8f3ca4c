// conv_kernels.h - fp32 MFMA implicit-GEMM convolutions for gfx950 (MI355X).
//
// Both kernels compute  D[pos][cout] = sum_{tap,ci} X[ci][src(pos,tap)] * W[tap][ci][cout]
// with v_mfma_f32_32x32x2_f32 (exact fp32 FMA chain, 64 FLOP/clk/SIMD):
//   A operand (lane l) = X[ci0 + (l>>5)][pos0 + (l&31) (+tap shift)]   -> 32 positions x 2 channels
//   B operand (lane l) = W[tap][ci0 + (l>>5)][n0 + (l&31)]             ->  2 channels x 32 couts
//   D (16 regs/lane)   : cout = n0 + (l&31), pos = pos0 + (r&3) + 8*(r>>2) + 4*(l>>5)
// so every lane ends up with 4 groups of 4 CONSECUTIVE positions of one output
// channel: the epilogue (bias, ReLU, residual adds) works on float4 and stores
// float4 along the sequence axis of the channel-major [C][L] activation layout.
//
// Operands are staged through LDS in K-chunks of KC input channels x all taps;
// the next chunk is prefetched into registers while the current one feeds the
// matrix pipe (global->reg issue before the MFMA block, reg->LDS after it).
// The fp32 MFMA is slow enough (64 cycles per 32x32x2) that one ds_read_b32 per
// operand per MFMA leaves the LDS pipe >90% idle; the design goal is simply to
// keep >=2 waves per SIMD resident with MFMA work so the chunk hand-over of one
// workgroup is covered by another.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));  // native vector: stays in VGPRs (HIP's float4 struct arrays went to scratch)

__device__ __forceinline__ f32x4 ld4_guard(const float* __restrict__ row, long pos, long n, bool vec_ok) {
  // 4 floats row[pos..pos+3] with zero fill outside [0,n)
  if (vec_ok && pos >= 0 && pos + 3 < n) return *reinterpret_cast<const f32x4*>(row + pos);
  f32x4 v;
  v.x = (pos + 0 >= 0 && pos + 0 < n) ? row[pos + 0] : 0.f;
  v.y = (pos + 1 >= 0 && pos + 1 < n) ? row[pos + 1] : 0.f;
  v.z = (pos + 2 >= 0 && pos + 2 < n) ? row[pos + 2] : 0.f;
  v.w = (pos + 3 >= 0 && pos + 3 < n) ? row[pos + 3] : 0.f;
  return v;
}

// ---------------------------------------------------------------------------
// Conv1d, kernel 9, padding 4, stride 1 (every Conv1d of Encoder/Encoder2/Encoder3,
// orca_modules.py:811-927, :991-1149, :1286-1386) + folded BN + ReLU + residuals.
// ---------------------------------------------------------------------------
struct Conv1dArgs {
  const float* x;   // [B][cin][ldx]
  const float* w;   // packed [nchunks][9][KC][COUT]
  const float* bias;
  float* y;         // [B][COUT][ldy]
  const float* r1;  // optional residuals, same layout as y
  const float* r2;
  long x_bs, y_bs;  // batch strides (elements)
  long ldx, ldy;
  long n;           // valid positions (input length == output length)
  int nchunks;      // cin / KC
  int relu;
  int x_vec_ok;     // x rows are 16-byte aligned at multiples of 4 positions
  int y_vec_ok;     // same for y / r1 / r2
  int y_nlc;        // write y (and read r1/r2) channel-LAST: y[pos*COUT + cout] (feeds conv_bf16s.h)
};

template <int COUT, int MW, int NW, int WM, int WN, int KC>
__global__ __launch_bounds__(WM* WN * 64) void conv1d_k9_kernel(Conv1dArgs a) {
  static_assert(WN * NW * 32 == COUT, "cout tiling");
  constexpr int NT = WM * WN * 64;
  constexpr int MT = WM * MW * 32;      // positions per workgroup
  constexpr int XS = MT + 8;            // LDS row: positions m0-4 .. m0+MT+3
  constexpr int XV = XS / 4;            // float4 per row
  constexpr int XN4 = KC * XV;          // float4 in one X chunk
  constexpr int WN4 = 9 * KC * COUT / 4;
  constexpr int XIT = (XN4 + NT - 1) / NT;
  constexpr int WIT = (WN4 + NT - 1) / NT;

  __shared__ f32x4 smem[XN4 + WN4];
  float* Xs = reinterpret_cast<float*>(smem);
  float* Ws = Xs + KC * XS;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int l31 = lane & 31, kh = lane >> 5;
  const long m0 = (long)blockIdx.x * MT;
  const int b = blockIdx.y;

  const float* xb = a.x + (long)b * a.x_bs;
  const f32x4* wg = reinterpret_cast<const f32x4*>(a.w);

  f32x16 acc[MW][NW];
#pragma unroll
  for (int i = 0; i < MW; ++i)
#pragma unroll
    for (int j = 0; j < NW; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // Register staging of the next chunk.  Loads are unconditional (indices are
  // clamped) so the arrays stay in VGPRs; only the LDS store is predicated.
  f32x4 xr[XIT], wr[WIT];
  const float* xrow[XIT];
  long xpos[XIT];
#pragma unroll
  for (int it = 0; it < XIT; ++it) {
    int idx = tid + it * NT;
    idx = idx < XN4 ? idx : XN4 - 1;
    const int row = idx / XV, v4 = idx - row * XV;
    xrow[it] = xb + (long)row * a.ldx;
    xpos[it] = m0 - 4 + 4 * v4;
  }
  const bool xvec = a.x_vec_ok != 0;
  const long chunk_stride = (long)KC * a.ldx;

#define CONV1D_LOAD_CHUNK(c)                                                              \
  {                                                                                       \
    _Pragma("unroll") for (int it = 0; it < XIT; ++it)                                    \
        xr[it] = ld4_guard(xrow[it] + (long)(c) * chunk_stride, xpos[it], a.n, xvec);     \
    const f32x4* wc = wg + (long)(c) * WN4;                                              \
    _Pragma("unroll") for (int it = 0; it < WIT; ++it) {                                  \
      int idx = tid + it * NT;                                                            \
      idx = idx < WN4 ? idx : WN4 - 1;                                                    \
      wr[it] = wc[idx];                                                                   \
    }                                                                                     \
  }
#define CONV1D_STORE_CHUNK()                                                              \
  {                                                                                       \
    _Pragma("unroll") for (int it = 0; it < XIT; ++it) {                                  \
      const int idx = tid + it * NT;                                                      \
      if (idx < XN4) smem[idx] = xr[it];                                                  \
    }                                                                                     \
    _Pragma("unroll") for (int it = 0; it < WIT; ++it) {                                  \
      const int idx = tid + it * NT;                                                      \
      if (idx < WN4) smem[XN4 + idx] = wr[it];                                            \
    }                                                                                     \
  }

  CONV1D_LOAD_CHUNK(0);
  CONV1D_STORE_CHUNK();
  __syncthreads();

  const float* xa0 = Xs + kh * XS + wm * (MW * 32) + l31;
  const float* wb0 = Ws + kh * COUT + wn * (NW * 32) + l31;

  for (int c = 0; c < a.nchunks; ++c) {
    const bool more = (c + 1 < a.nchunks);
    if (more) CONV1D_LOAD_CHUNK(c + 1);
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
#pragma unroll
      for (int cp = 0; cp < KC / 2; ++cp) {
        float av[MW], bv[NW];
#pragma unroll
        for (int i = 0; i < MW; ++i) av[i] = xa0[(2 * cp) * XS + i * 32 + tap];
#pragma unroll
        for (int j = 0; j < NW; ++j) bv[j] = wb0[(tap * KC + 2 * cp) * COUT + j * 32];
#pragma unroll
        for (int i = 0; i < MW; ++i)
#pragma unroll
          for (int j = 0; j < NW; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[j], acc[i][j], 0, 0, 0);
      }
    }
    __syncthreads();
    if (more) CONV1D_STORE_CHUNK();
    __syncthreads();
  }
#undef CONV1D_LOAD_CHUNK
#undef CONV1D_STORE_CHUNK

  // ---- epilogue: bias, ReLU, residual adds, float4 stores along the sequence
  float* yb = a.y + (long)b * a.y_bs;
  const float* r1b = a.r1 ? a.r1 + (long)b * a.y_bs : nullptr;
  const float* r2b = a.r2 ? a.r2 + (long)b * a.y_bs : nullptr;
#pragma unroll
  for (int j = 0; j < NW; ++j) {
    const int co = wn * (NW * 32) + j * 32 + l31;
    const float bias = a.bias[co];
    const long rowoff = (long)co * a.ldy;
#pragma unroll
    for (int i = 0; i < MW; ++i) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const long pos = m0 + wm * (MW * 32) + i * 32 + 8 * g + 4 * kh;
        if (pos >= a.n) continue;
        float4 v;
        v.x = acc[i][j][4 * g + 0] + bias;
        v.y = acc[i][j][4 * g + 1] + bias;
        v.z = acc[i][j][4 * g + 2] + bias;
        v.w = acc[i][j][4 * g + 3] + bias;
        if (a.relu) {
          v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
        }
        if (a.y_nlc) {
          const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            if (pos + e < a.n) {
              const long o = (pos + e) * COUT + co;
              float t = vv[e];
              if (r1b) t += r1b[o];
              if (r2b) t += r2b[o];
              yb[o] = t;
            }
          }
        } else if (a.y_vec_ok && pos + 3 < a.n) {
          if (r1b) { const float4 q = *reinterpret_cast<const float4*>(r1b + rowoff + pos); v.x += q.x; v.y += q.y; v.z += q.z; v.w += q.w; }
          if (r2b) { const float4 q = *reinterpret_cast<const float4*>(r2b + rowoff + pos); v.x += q.x; v.y += q.y; v.z += q.z; v.w += q.w; }
          *reinterpret_cast<float4*>(yb + rowoff + pos) = v;
        } else {
          const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            if (pos + e < a.n) {
              float t = vv[e];
              if (r1b) t += r1b[rowoff + pos + e];
              if (r2b) t += r2b[rowoff + pos + e];
              yb[rowoff + pos + e] = t;
            }
          }
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------
// Conv2d 3x3, dilation d, padding d (every 3x3 Conv2d of Decoder / Decoder_1m,
// orca_modules.py:22-459, :499-780) + folded BN + ReLU + residual.
// Feature maps live as [C][H][LDW=256] (row stride padded 250->256 so that rows
// are float4-aligned).  One workgroup = one output row (all W<=256 columns, all
// COUT channels); wave w owns columns [32w, 32w+32).  Zero padding is handled
// by predication: out-of-range source rows are staged as zeros (and their taps
// skipped, workgroup-uniform), out-of-range source columns are masked on the
// A-operand read, and a wave skips a tap column entirely when all of its 32
// source columns fall outside the map (wave-uniform; ~1/3 of the taps of the
// edge waves at dilation 64).
// ---------------------------------------------------------------------------
#define ORCA_LDW 256

struct Conv2dArgs {
  const float* x;  // [B][cin_pad][H][256]
  const float* w;  // packed [nchunks][9][8][COUT]
  const float* bias;
  float* y;        // [B][*][H][256], written at channel offset 0
  const float* r;  // optional residual, layout/batch stride like y
  long x_bs, y_bs, r_bs;
  int H, W;
  int dil;
  int nchunks;     // cin_pad / 8
  int relu;
};

template <int COUT>
__global__ __launch_bounds__(512) void conv2d_3x3_kernel(Conv2dArgs a) {
  constexpr int KC = 8;
  constexpr int NW = COUT / 32;
  constexpr int NT = 512;
  constexpr int XN4 = KC * 3 * (ORCA_LDW / 4);  // 1536
  constexpr int WN4 = 9 * KC * COUT / 4;
  constexpr int XIT = XN4 / NT;                 // 3
  constexpr int WIT = (WN4 + NT - 1) / NT;

  __shared__ f32x4 smem[XN4 + WN4];
  float* Xs = reinterpret_cast<float*>(smem);
  float* Ws = Xs + XN4 * 4;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, kh = lane >> 5;
  const int y0 = blockIdx.x;
  const int b = blockIdx.y;
  const int H = a.H, W = a.W, d = a.dil;

  const float* xb = a.x + (long)b * a.x_bs;
  const f32x4* wg = reinterpret_cast<const f32x4*>(a.w);
  const long cstride = (long)H * ORCA_LDW;

  bool rowok[3];
#pragma unroll
  for (int ky = 0; ky < 3; ++ky) {
    const int ys = y0 + (ky - 1) * d;
    rowok[ky] = (ys >= 0 && ys < H);
  }
  // per-kx source column of this lane, validity and wave-level skip
  int xsrc[3];
  bool xok[3], wskip[3];
#pragma unroll
  for (int kx = 0; kx < 3; ++kx) {
    const int x0 = wave * 32 + (kx - 1) * d;
    const int xs = x0 + l31;
    xok[kx] = (xs >= 0 && xs < W);
    xsrc[kx] = xs < 0 ? 0 : (xs > ORCA_LDW - 1 ? ORCA_LDW - 1 : xs);
    wskip[kx] = (x0 + 31 < 0) || (x0 >= W);
  }

  f32x16 acc[NW];
#pragma unroll
  for (int j = 0; j < NW; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

  f32x4 xr[XIT], wr[WIT];
  long xoff[XIT];   // element offset of this thread's float4 inside a chunk (clamped to a valid row)
  bool xrowok[XIT];
#pragma unroll
  for (int it = 0; it < XIT; ++it) {
    const int idx = tid + it * NT;
    const int row = idx >> 6, v4 = idx & 63;  // row = ci*3 + ky
    const int ci = row / 3, ky = row - ci * 3;
    const int ys = y0 + (ky - 1) * d;
    xrowok[it] = (ys >= 0 && ys < H);
    xoff[it] = (long)ci * cstride + (long)(xrowok[it] ? ys : y0) * ORCA_LDW + 4 * v4;
  }
#define CONV2D_LOAD_CHUNK(c)                                                               \
  {                                                                                        \
    const float* xc = xb + (long)(c) * KC * cstride;                                       \
    _Pragma("unroll") for (int it = 0; it < XIT; ++it) {                                   \
      f32x4 v = *reinterpret_cast<const f32x4*>(xc + xoff[it]);                            \
      if (!xrowok[it]) v = (f32x4)(0.f);                                                   \
      xr[it] = v;                                                                          \
    }                                                                                      \
    const f32x4* wc = wg + (long)(c) * WN4;                                               \
    _Pragma("unroll") for (int it = 0; it < WIT; ++it) {                                   \
      int idx = tid + it * NT;                                                             \
      idx = idx < WN4 ? idx : WN4 - 1;                                                     \
      wr[it] = wc[idx];                                                                    \
    }                                                                                      \
  }
#define CONV2D_STORE_CHUNK()                                                               \
  {                                                                                        \
    _Pragma("unroll") for (int it = 0; it < XIT; ++it) smem[tid + it * NT] = xr[it];       \
    _Pragma("unroll") for (int it = 0; it < WIT; ++it) {                                   \
      const int idx = tid + it * NT;                                                       \
      if (idx < WN4) smem[XN4 + idx] = wr[it];                                             \
    }                                                                                      \
  }

  CONV2D_LOAD_CHUNK(0);
  CONV2D_STORE_CHUNK();
  __syncthreads();

  const float* wb0 = Ws + kh * COUT + l31;

  for (int c = 0; c < a.nchunks; ++c) {
    const bool more = (c + 1 < a.nchunks);
    if (more) CONV2D_LOAD_CHUNK(c + 1);
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      if (!rowok[ky]) continue;  // workgroup-uniform
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        if (wskip[kx]) continue;  // wave-uniform
        const int tap = ky * 3 + kx;
#pragma unroll
        for (int cp = 0; cp < KC / 2; ++cp) {
          float av = Xs[((2 * cp + kh) * 3 + ky) * ORCA_LDW + xsrc[kx]];
          av = xok[kx] ? av : 0.f;
          float bv[NW];
#pragma unroll
          for (int j = 0; j < NW; ++j) bv[j] = wb0[(tap * KC + 2 * cp) * COUT + j * 32];
#pragma unroll
          for (int j = 0; j < NW; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv[j], acc[j], 0, 0, 0);
        }
      }
    }
    __syncthreads();
    if (more) CONV2D_STORE_CHUNK();
    __syncthreads();
  }
#undef CONV2D_LOAD_CHUNK
#undef CONV2D_STORE_CHUNK

  float* yb = a.y + (long)b * a.y_bs;
  const float* rb = a.r ? a.r + (long)b * a.r_bs : nullptr;
#pragma unroll
  for (int j = 0; j < NW; ++j) {
    const int co = j * 32 + l31;
    const float bias = a.bias[co];
    const long base = (long)co * cstride + (long)y0 * ORCA_LDW;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int xo = wave * 32 + 8 * g + 4 * kh;
      if (xo >= W) continue;
      float4 v;
      v.x = acc[j][4 * g + 0] + bias;
      v.y = acc[j][4 * g + 1] + bias;
      v.z = acc[j][4 * g + 2] + bias;
      v.w = acc[j][4 * g + 3] + bias;
      if (a.relu) {
        v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
      }
      if (rb) {
        const float4 q = *reinterpret_cast<const float4*>(rb + base + xo);
        v.x += q.x; v.y += q.y; v.z += q.z; v.w += q.w;
      }
      *reinterpret_cast<float4*>(yb + base + xo) = v;  // columns >= W are padding, never read unmasked
    }
  }
}
