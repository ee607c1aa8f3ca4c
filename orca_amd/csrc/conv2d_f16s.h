// conv2d_f16s.h - dilated 3x3 Conv2d of the Decoders on the fp16 matrix cores with 2-way split fp32
// operands (3 MFMA products per fp32 product, ~2^-22 relative error; see conv_bf16s.h).
//
// Feature maps are CHUNK-PLANAR: [C/16][row][256 px][16 ch] fp32 (row pitch padded 250 -> 256 px; pad pixels are
// kept at ZERO by every producer so that reads past the right edge need no mask): the 16 channels of one K-chunk of a
// pixel are 64 contiguous bytes and consecutive pixels are adjacent, so every wave-level load of the K loop and every
// store of the epilogue covers whole 128-byte lines (a pixel-major [row][px][C] map has them half used; measured
// 24.3 -> 23.3 / 33.5 -> 33.1 us per launch for COUT 32 / 64, 105.7 -> 104.0 ms per step).  One workgroup = one
// output row; wave w owns pixels [32w, 32w+32) and all COUT channels.  K walks chunks of 16 input
// channels x 9 taps; per chunk the LDS holds
//   X image [split][g][3 source rows][256 px][8 ch] fp16     (split while staging)
//   W image [split][tap][g][COUT][8 ch]             fp16     (pre-split on the host)
// MFMA is issued as D = W-frag x X-frag, so a lane owns ONE pixel and 4 consecutive couts per register
// group: the epilogue (bias, ReLU, residual) is float4 along the channel axis.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "conv_bf16s.h"
#include "misc_kernels.h"

struct Conv2dF16Args {
  const float* x;    // [B][xc/16][H][256][16]
  const void* w;     // packed fp16 [nchunks][2][9][2][COUT][8]
  const float* bias;
  float* y;          // [B][yc/16][H][256][16]   (channels 0..COUT-1 written)
  const float* r;    // optional residual, same layout
  long x_bs, y_bs, r_bs;
  long x_cs, y_cs, r_cs;   // chunk strides (H * 256 * 16 floats of the respective buffer)
  int H, W, dil, nchunks, relu;
  int banded;        // grid.x = 8 * ceil(H/8), XCD-banded row order (see the kernel)
  unsigned* flag;
  unsigned long long* stamps;   // tools/microbench_c2.hip only (NULL in the library): s_memtime stamps of one workgroup
};

// NS / DT: operand splits and 16-bit type (conv_bf16s.h).  <2, 1> = fp32-class f16x2 (3 products); <1, 0> = plain bf16
// operands, ONE product (the throughput mode of BASELINE config 3; a.w is then the bf16 pack [nchunks][9][2][COUT][8]).
template <int COUT, int NS = 2, int DT = 1>
__global__ __launch_bounds__(512, (NS == 1 && COUT == 64) ? 4 : 2) void conv2d_3x3_f16s_kernel(Conv2dF16Args a) {   // bf16, 64 couts: 70 KB of LDS - two workgroups per CU need <= 128 VGPRs
  static_assert((NS == 2 && DT == 1) || (NS == 1 && DT == 0), "f16x2 or plain bf16");
  constexpr int NT = 512, NW = COUT / 32, PX = 256;
  constexpr int XU = NS * 2 * 3 * PX;     // 16-byte units of the X image (6144)
  constexpr int WU = NS * 9 * 2 * COUT;   // 16-byte units of the W image
  constexpr int XF4 = 3 * PX * 4;         // float4 loads per chunk (3 rows x 256 px x 4 channel quads)
  constexpr int XIT = XF4 / NT;           // 6
  constexpr int WIT = (WU + NT - 1) / NT;
  constexpr int TU = 8 * 32 * (COUT + 4) / 4;   // units of the epilogue's transposition tiles (8 waves x 32 px x (COUT + 4) floats)
  constexpr int SU = (XU + WU) > TU ? (XU + WU) : TU;
  __shared__ f32x4 smem[SU + COUT / 4];   // + bias

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, g = lane >> 5;
  const int b = blockIdx.y, H = a.H, W = a.W, d = a.dil;
  // Every source row is read by the three workgroups y-d, y, y+d.  Workgroups go to the 8 XCDs (each with its own
  // 4 MB L2) round-robin by linear id: for d >= 8 (all multiples of 8) the three readers share an XCD already; for
  // d = 1, 2, 4 each XCD takes a contiguous band of ceil(H/8) rows instead (grid.x = 8 * band), so that the re-reads
  // hit its L2 - measured 33 -> 24 us (COUT 32) and 42 -> 37 us (COUT 64) per launch at d = 1, 2.
  int y0 = blockIdx.x;
  if (a.banded) {
    y0 = (int)(blockIdx.x & 7) * ((H + 7) >> 3) + (int)(blockIdx.x >> 3);
    if (y0 >= H) return;
  }
  const float* xb = a.x + (long)b * a.x_bs;
  const f32x4* wg = reinterpret_cast<const f32x4*>(a.w);
  float* bias_s = reinterpret_cast<float*>(smem + SU);
  if (tid < COUT) bias_s[tid] = a.bias[tid];   // visible after the first barrier

  bool rowok[3];
#pragma unroll
  for (int ky = 0; ky < 3; ++ky) {
    const int ys = y0 + (ky - 1) * d;
    rowok[ky] = (ys >= 0 && ys < H);
  }
  // this lane's source pixel per kx (pad pixels [W,256) are zero in memory; only <0 / >=256 need a mask)
  int xsrc[3];
  bool xok[3], wskip[3];
#pragma unroll
  for (int kx = 0; kx < 3; ++kx) {
    const int x0 = wave * 32 + (kx - 1) * d, xs = x0 + l31;
    xok[kx] = (xs >= 0 && xs < PX);
    xsrc[kx] = xs < 0 ? 0 : (xs > PX - 1 ? PX - 1 : xs);
    wskip[kx] = (x0 + 31 < 0) || (x0 >= W);
  }

  int nst = 0;
#define C2_STAMP() if (a.stamps && blockIdx.x == 100 && blockIdx.y == 0 && lane == 0 && nst < 40) a.stamps[wave * 40 + nst++] = __builtin_readcyclecounter();
  C2_STAMP();
  f32x16 acc[NW];
#pragma unroll
  for (int j = 0; j < NW; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

  bool overflow = false;
  f32x4 xr[XIT], wr[WIT];
  long xoff[XIT];
  int xdst[XIT];
  bool xvalid[XIT];
#pragma unroll
  for (int it = 0; it < XIT; ++it) {
    const int u = tid + it * NT;            // u = (ky*256 + px)*4 + q
    const int q = u & 3, px = (u >> 2) & (PX - 1), ky = u >> 10;
    const int ys = y0 + (ky - 1) * d;
    xvalid[it] = (ys >= 0 && ys < H);
    xoff[it] = (long)(xvalid[it] ? ys : y0) * (PX * 16) + (long)px * 16 + 4 * q;
    xdst[it] = ((((q >> 1) * 3 + ky) * PX) + px) * 16 + (q & 1) * 8;
  }
#define C2_LOAD_CHUNK(c)                                                                     \
  {                                                                                          \
    _Pragma("unroll") for (int it = 0; it < XIT; ++it) {                                     \
      f32x4 v = *reinterpret_cast<const f32x4*>(xb + xoff[it] + (long)(c) * a.x_cs);           \
      if (!xvalid[it]) v = (f32x4)(0.f);                                                     \
      xr[it] = v;                                                                            \
    }                                                                                        \
    const f32x4* wc = wg + (long)(c) * WU;                                                   \
    _Pragma("unroll") for (int it = 0; it < WIT; ++it) {                                     \
      int idx = tid + it * NT;                                                               \
      idx = idx < WU ? idx : WU - 1;                                                         \
      wr[it] = wc[idx];                                                                      \
    }                                                                                        \
  }
#define C2_STORE_CHUNK()                                                                     \
  {                                                                                          \
    char* xs_ = reinterpret_cast<char*>(smem);                                               \
    _Pragma("unroll") for (int it = 0; it < XIT; ++it) {                                     \
      u32x2 sp[NS];                                                                          \
      split4<NS, DT>(xr[it], sp, overflow);                                                  \
      _Pragma("unroll") for (int s = 0; s < NS; ++s)                                         \
          *reinterpret_cast<u32x2*>(xs_ + s * (2 * 3 * PX * 16) + xdst[it]) = sp[s];         \
    }                                                                                        \
    _Pragma("unroll") for (int it = 0; it < WIT; ++it) {                                     \
      const int idx = tid + it * NT;                                                         \
      if (idx < WU) smem[XU + idx] = wr[it];                                                 \
    }                                                                                        \
  }

  C2_LOAD_CHUNK(0);
  C2_STAMP();   // 1: first loads issued
  C2_STORE_CHUNK();
  C2_STAMP();   // 2: first chunk landed, split, in LDS
  __syncthreads();
  C2_STAMP();   // 3

  const f32x4* xa0 = smem + g * (3 * PX);                  // + s*2*3*PX + ky*PX + px
  const f32x4* wb0 = smem + XU + g * COUT + l31;           // + ((s*9+tap)*2)*COUT + j*32

  for (int c = 0; c < a.nchunks; ++c) {
    const bool more = (c + 1 < a.nchunks);
    if (more) C2_LOAD_CHUNK(c + 1);
    C2_STAMP();   // +0: prefetch issued
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      if (!rowok[ky]) continue;   // workgroup-uniform: the whole source row is zero padding
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        if (wskip[kx]) continue;  // wave-uniform: all 32 source pixels of this wave are padding
        const int tap = ky * 3 + kx;
        typename Op16<DT>::vec xv[NS], wv[NS][NW];
#pragma unroll
        for (int s = 0; s < NS; ++s) {
          f32x4 t = xa0[s * (2 * 3 * PX) + ky * PX + xsrc[kx]];
          if (!xok[kx]) t = (f32x4)(0.f);
          xv[s] = __builtin_bit_cast(typename Op16<DT>::vec, t);
#pragma unroll
          for (int j = 0; j < NW; ++j) wv[s][j] = __builtin_bit_cast(typename Op16<DT>::vec, wb0[((s * 9 + tap) * 2) * COUT + j * 32]);
        }
#pragma unroll
        for (int j = 0; j < NW; ++j) {
          if constexpr (NS == 2) {
            acc[j] = Op16<DT>::mfma(wv[0][j], xv[1], acc[j]);
            acc[j] = Op16<DT>::mfma(wv[1][j], xv[0], acc[j]);
          }
          acc[j] = Op16<DT>::mfma(wv[0][j], xv[0], acc[j]);
        }
      }
    }
    C2_STAMP();   // +1: MFMA block issued
    if (!more) break;
    __syncthreads();
    C2_STAMP();   // +2: everyone done with the LDS image
    C2_STORE_CHUNK();
    C2_STAMP();   // +3: prefetch landed, split, stored
    __syncthreads();
    C2_STAMP();   // +4
  }
#undef C2_LOAD_CHUNK
#undef C2_STORE_CHUNK

  // Epilogue.  A lane owns ONE pixel and 4-cout groups, and a pixel's channels are 128-256 contiguous bytes of the
  // channel-last map: written straight from the accumulators, every store (and residual load) instruction would touch
  // 64 different lines, 16 bytes each - stamps put that at 12-24 000 cycles for COUT 64, a third of the workgroup's
  // life.  Instead each wave transposes its 32 px x COUT tile through its own slice of the (now idle) operand LDS
  // (row pitch + 16 B: conflict-free both ways) and moves whole pixel rows: 1 KB contiguous per instruction.
  constexpr int PITCH = COUT + 4;        // floats
  constexpr int LPR = COUT / 4;          // lanes per pixel row
  constexpr int RPI = 64 / LPR;          // pixel rows per wave instruction
  constexpr int NI = 32 / RPI;
  static_assert(8 * 32 * PITCH * 4 <= SU * 16, "tile transposition needs the operand images' LDS");
  const int lr = lane / LPR, lc = (lane % LPR) * 4;
  const long pix0 = (long)y0 * PX + wave * 32;
  f32x4 rres[NI];
  if (a.r) {   // residual rows, fetched before the transposition
    const float* rp = a.r + (long)b * a.r_bs + (long)(lc >> 4) * a.r_cs + pix0 * 16 + (lc & 15);
#pragma unroll
    for (int k = 0; k < NI; ++k) rres[k] = *reinterpret_cast<const f32x4*>(rp + (long)(k * RPI + lr) * 16);
  }
  __syncthreads();                       // every wave is done reading the operand images
  float* tile = reinterpret_cast<float*>(smem) + wave * (32 * PITCH);
#pragma unroll
  for (int j = 0; j < NW; ++j) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int co = j * 32 + 8 * q + 4 * g;
      const f32x4 bias = *reinterpret_cast<const f32x4*>(bias_s + co);
      f32x4 v;
      v.x = acc[j][4 * q + 0] + bias.x;
      v.y = acc[j][4 * q + 1] + bias.y;
      v.z = acc[j][4 * q + 2] + bias.z;
      v.w = acc[j][4 * q + 3] + bias.w;
      if (a.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
      *reinterpret_cast<f32x4*>(tile + l31 * PITCH + co) = v;
    }
  }
  // same-wave LDS writes and reads complete in order: no barrier needed for the wave-private tile
  float* yp = a.y + (long)b * a.y_bs + (long)(lc >> 4) * a.y_cs + pix0 * 16 + (lc & 15);
#pragma unroll
  for (int k = 0; k < NI; ++k) {
    const int row = k * RPI + lr;
    f32x4 v = *reinterpret_cast<const f32x4*>(tile + row * PITCH + lc);
    if (a.r) v += rres[k];
    if (wave * 32 + row >= W) v = (f32x4)(0.f);   // keep the pad pixels of every feature map at zero
    *reinterpret_cast<f32x4*>(yp + (long)row * 16) = v;
  }
  C2_STAMP();   // last: epilogue issued
  if (DT == 1 && overflow && a.flag) *a.flag = 1u;
}

// ---- channel-last helpers of the Decoder --------------------------------------------------------------
// mat[i][j][c] = x[c][i] + x[c][j] (c<128); channels 128..128+nt-1 = distenc[t][i][j] (if given); other pad channels and
// pad pixels = 0.  out [cp/16][n][256][16].  block = (cp/4 threads x 8 pixels), grid = (32, n)
__global__ void outer_sum_nhwc_kernel(const float* __restrict__ x, long sx_c, long sx_l, const float* __restrict__ de, long sd_c, long sd_h,
                                      long sd_w, int nt, float* __restrict__ out, int n, int cp) {
  const int c4 = threadIdx.x, j = blockIdx.x * blockDim.y + threadIdx.y, i = blockIdx.y;
  f32x4 v = (f32x4)(0.f);
  if (j < n) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int c = 4 * c4 + e;
      float t = 0.f;
      if (c < 128) t = x[c * sx_c + i * sx_l] + x[c * sx_c + j * sx_l];
      else if (c < 128 + nt && de) t = de[(c - 128) * sd_c + i * sd_h + j * sd_w];
      v[e] = t;
    }
  }
  *reinterpret_cast<f32x4*>(out + (long)(c4 >> 2) * ((long)n * 256 * 16) + ((long)i * 256 + j) * 16 + 4 * (c4 & 3)) = v;
}

// bilinear / nearest x2 upsample of y [nt][n/2][n/2] into channels [c0, c0+16) (one chunk) of a [cp/16][n][256][16] map
// (channels c0..c0+nt-1 = values, the rest = 0; pad pixels 0).  block 256 (pixels), grid n
__global__ void upsample2d_nhwc_kernel(const float* __restrict__ y, long sy_c, long sy_h, long sy_w, int nt, float* __restrict__ out, int n,
                                       int cp, int c0, int bilinear) {
  const int j = threadIdx.x, i = blockIdx.x, h = n / 2;
  float v[16];
#pragma unroll
  for (int t = 0; t < 16; ++t) v[t] = 0.f;
  if (j < n) {
    const float fy = fmaxf(0.5f * (i + 0.5f) - 0.5f, 0.f), fx = fmaxf(0.5f * (j + 0.5f) - 0.5f, 0.f);
    const int y0 = (int)fy, x0 = (int)fx;
    const int y1 = y0 + (y0 < h - 1 ? 1 : 0), x1 = x0 + (x0 < h - 1 ? 1 : 0);
    const float ly = fy - y0, lx = fx - x0, hy = 1.f - ly, hx = 1.f - lx;
#pragma unroll
    for (int t = 0; t < ORCA_MAX_TARGETS; ++t) {
      if (t >= nt) break;
      const float* yt = y + t * sy_c;
      if (!bilinear) v[t] = yt[(i >> 1) * sy_h + (j >> 1) * sy_w];
      else
        v[t] = hy * (hx * yt[y0 * sy_h + x0 * sy_w] + lx * yt[y0 * sy_h + x1 * sy_w]) +
               ly * (hx * yt[y1 * sy_h + x0 * sy_w] + lx * yt[y1 * sy_h + x1 * sy_w]);
    }
  }
  f32x4* o = reinterpret_cast<f32x4*>(out + (long)(c0 >> 4) * ((long)n * 256 * 16) + ((long)i * 256 + j) * 16);   // c0 % 16 == 0
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    f32x4 f;
    f.x = v[4 * q]; f.y = v[4 * q + 1]; f.z = v[4 * q + 2]; f.w = v[4 * q + 3];
    o[q] = f;
  }
}

// `final` head + symmetrisation on a chunk-planar [4][n][256][16] map (see final_sym_kernel)
__global__ void final_sym_nhwc_kernel(FinalArgs a) {
  ORCA_FINAL_LOAD_HEAD();
  const int j = threadIdx.x, i = blockIdx.x, b = blockIdx.y, n = a.n;
  if (j >= n) return;
  const float* cur = a.cur + (long)b * a.cur_bs;
  const long cs4 = (long)n * 256 * 4;   // chunk stride in float4 units
  const f32x4* pu = reinterpret_cast<const f32x4*>(cur + ((long)i * 256 + j) * 16);
  const f32x4* pv = reinterpret_cast<const f32x4*>(cur + ((long)j * 256 + i) * 16);
  float h1[ORCA_MAX_TARGETS], h2[ORCA_MAX_TARGETS];
#pragma unroll
  for (int o = 0; o < ORCA_MAX_TARGETS; ++o) { h1[o] = b1s[o]; h2[o] = b1s[o]; }
  for (int c4 = 0; c4 < 16; ++c4) {
    const f32x4 u = pu[(c4 >> 2) * cs4 + (c4 & 3)], v = pv[(c4 >> 2) * cs4 + (c4 & 3)];
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int o = 0; o < ORCA_MAX_TARGETS; ++o)
        if (o < a.F) { h1[o] = fmaf(w1s[o * 64 + 4 * c4 + e], u[e], h1[o]); h2[o] = fmaf(w1s[o * 64 + 4 * c4 + e], v[e], h2[o]); }
  }
  final_head_store(a, h1, h2, w2s, b2s, b, i, j);
}

// Tried and dropped (round 1): a tap-paired variant with 8-channel chunks (45 KB LDS, three 4-wave workgroups per CU
// instead of one 8-wave one) was correct but 30 % slower per Decoder (5.2 vs 4.0 ms at B=2): these kernels are bound by
// the per-row latency chain global -> registers -> split -> LDS -> MFMA over only 2-4 chunks, and halving the chunk
// doubled the number of links.  Several rows per workgroup is slower still, so it is not dispatch-bound either.
// Per-wave s_memtime stamps (tools/microbench_c2.hip; COUT 32, 4 K-chunks, ~36 000 cycles per workgroup, two workgroups
// resident per CU): per chunk ~8 200 cycles = issuing the next chunk's loads 2 200-2 700 (each wave-instruction touches
// 16 half-used 128-B lines of the then pixel-major [px][64 ch] rows: TA-bound) + MFMA block 2 500 (the matrix pipe is shared with the
// other resident workgroup) + barrier 1 200 + wait/split/ds_write 1 200 + barrier 1 000.  Dispatch is not the limit
// (an empty 500-workgroup launch with the same LDS: 12 ns per workgroup).  The chunk-planar layout adopted since
// halves the load-issue cost; LDS-DMA from pre-split maps needs a double-buffered X image (96 KB) and loses the
// second resident workgroup.
// Also measured and dropped: one accumulator per MFMA product (acc 48 -> 144 registers for COUT 64): the block is
// pipe-bound, not dependency-bound, and the registers cost occupancy (5.0 vs 3.7 ms); running the 64-cout layers as two 32-cout workgroups per row (67 KB LDS each, two resident
// per CU): no change (3.72 vs 3.73 ms per Decoder); keeping TWO K-chunks of global loads in flight (register slots
// 0/1): 166 VGPRs for COUT 32 cost the second resident workgroup, 4.0-4.4 ms, and 5.8 ms when capped at 128 VGPRs.
// Two output rows (y, y + d) per 16-wave workgroup from the FOUR source rows y-d .. y+2d and one W image (33 % less X
// and 50 % less W traffic per output row; parity green): COUT 32 26.0 vs 24.3 us per launch, COUT 64 (128-VGPR cap at
// 1024 threads: spills) 47.8 vs 33.5 us - the K loop is bound by its latency chain, not by L2 bandwidth, and the
// bigger workgroup costs the second resident one.  Dropped.
// A wave-specialised persistent variant (waves 0-7 MFMA + epilogue only, waves 8-11 global -> registers two steps
// ahead -> split -> the other of two 66 KB LDS images; stream of (map, row, 32-cout half, chunk) steps, one barrier
// per step; parity green): 3.79 vs 3.42 ms per Decoder.  With one 12-wave workgroup per CU the bytes in flight per CU
// drop from 2 x 67 KB to ~85 KB and the loop becomes latency-bound (4.3 TB/s out of the L2s instead of 8.8): what
// these layers need is more bytes in flight per CU (LDS-DMA from pre-split, chunk-planar maps), not a different
// split of the work between waves.  Dropped.
