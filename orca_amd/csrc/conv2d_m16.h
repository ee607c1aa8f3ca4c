// conv2d_m16.h - the Decoders' feature maps in "M16" and the dilated 3x3 Conv2d on them (dilations 1-8; the blocks with
// dilation 16-64 run in conv2d_dblock.h).
//
// M16 = the 2-D analogue of the Encoder's P16 / B16: a map [C][H][W] is stored as planes of 16-byte units, one unit = the 8
// channels 8o..8o+7 of one pixel in 16-bit form; plane (octet o, split s) = [H rows][256 pixels] units (row pitch padded
// 250 -> 256, pad pixels kept ZERO by every producer):
//     unit(o, s, y, x) at base + (((o * NS + s) * H + y) * 256 + x) * 16 bytes
//   NS = 2, fp16: value = hi + lo (22 significant bits, the fp32-class default: 4 bytes per element as fp32);
//   NS = 1: one 16-bit plane, 2 bytes per element - bf16 (the throughput mode of BASELINE config 3) or fp16 ("f16": the same
//           rate and traffic, 11 instead of 8 significant bits, fp16 range guard as in the default mode).
// A plane row IS the MFMA operand image of that row: a conv's K-chunk (16 channels x 3 source rows x 256 pixels) is
// 12 (6) contiguous 4 KB runs that LDS-DMA drops into LDS - no staging registers, no split in the consumer (the per-layer
// kernel of round 1 re-split every fp32 row three times, once per reader, behind a load -> split -> ds_write -> barrier
// chain per chunk: 18-20 us per conv against ~8 us in the fused block, which had no such chain).  Producers split once,
// in their epilogue, with the P16 recipe (v_cvt_pk + fma_mix, v_permlane32_swap): every lane stores whole 16-byte units,
// 512 contiguous bytes per half wave - the LDS transposition of the fp32 layout is gone too.
//
// conv2d_3x3_m16_kernel: one workgroup = one output row (8 waves x 32 pixels, all couts); the work is a stream of pieces
// (K-chunk k, 32-cout half h): X image of chunk k double-buffered ([split][2][3 rows][8 + 256 + 8] units, zero margins
// for the taps that leave the row), weight pieces [split][9][2][32] in a 3-deep ring, both by LDS-DMA issued one / two
// pieces ahead and retired by counted s_waitcnt vmcnt; one barrier per piece; operand reads in inline asm with counted
// lgkmcnt waits (as conv_p16.h).
#pragma once
#include "conv_p16.h"
#include "misc_kernels.h"

#define M16_PX 256

__device__ __forceinline__ long m16_plane(int o, int s, int NS, int H) { return (long)(o * NS + s) * H * M16_PX; }   // in units

// Store of one whole 16-byte unit of a map by the conv epilogues: WRITE-THROUGH (sc0 sc1).  A layer's output then leaves the XCD's L2 while the
// kernel still runs instead of sitting there dirty until the end-of-kernel write-back (8-33 MB per launch, MI355X_MICROARCH.md "boundary":
// + B / 6 TB/s per dependent launch); round 5, tools/microbench_m16q.hip, dependent chain at B = 2 / 4: 64 -> 32 19.5-20.4 -> 18.6-19.2 us,
// 32 -> 64 + residual 23.3 -> 23.3, B = 4 32.0 / 43.3 -> 30.6 / 40.8; `nt` instead: slower.  M16_STORE_WT=0 (compile time): plain stores.
#ifndef M16_STORE_WT
#define M16_STORE_WT 1
#endif
__device__ __forceinline__ void m16_store_unit(u32x4_t* p, const u32x4_t& v) {
#if M16_STORE_WT == 1
  asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
#elif M16_STORE_WT == 2
  asm volatile("global_store_dwordx4 %0, %1, off nt\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
#else
  *p = v;
#endif
}

// 8 consecutive channels of one pixel -> NS units
template <int NS, int DT>
__device__ __forceinline__ void m16_pack8(const f32x4 lo4, const f32x4 hi4, u32x4_t (&out)[NS], bool& ovf) {
  u32x2 a[NS], b[NS];
  split4<NS, DT>(lo4, a, ovf);
  split4<NS, DT>(hi4, b, ovf);
#pragma unroll
  for (int s = 0; s < NS; ++s) { out[s].x = a[s].x; out[s].y = a[s].y; out[s].z = b[s].x; out[s].w = b[s].y; }
}
// two floats -> packed 16-bit pair (round to nearest even)
template <int DT>
__device__ __forceinline__ unsigned m16_pk2(float a, float b) {
  if (DT == 1) { const f16x2 h = {(_Float16)a, (_Float16)b}; return __builtin_bit_cast(unsigned, h); }
  return cvt_pk_bf16(a, b);
}
// two packed 16-bit values -> floats
template <int DT>
__device__ __forceinline__ void m16_pair(unsigned pk, float& a, float& b) {
  if (DT == 1) { const f16x2 h = __builtin_bit_cast(f16x2, pk); a = (float)h.x; b = (float)h.y; }
  else { a = bf16lo_f32(pk); b = bf16hi_f32(pk); }
}
template <int NS, int DT>
__device__ __forceinline__ void m16_unpack8(const u32x4_t (&u)[NS], float (&v)[8]) {
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    float a, b;
    m16_pair<DT>(u[0][e], a, b);
    if (NS == 2) { float c, d; m16_pair<DT>(u[NS - 1][e], c, d); a += c; b += d; }
    v[2 * e] = a; v[2 * e + 1] = b;
  }
}

// ---- producers / consumers at the ends of a Decoder ---------------------------------------------------------------
// mat[c][i][j] = x[c][i] + x[c][j] (c < 128); channels 128..128+nt-1 = distenc[t][i][j]; other channels and pad pixels 0.
// `o0`: first channel octet produced (the Decoder only materialises octets 16, 17 = distenc + padding; octet o lands in plane o - o0).
// (role of decoder_head_m16_kernel) workgroup = (row i, octet), block 256 (pixel j): 4 500 independent workgroups per map (one row of ALL octets per workgroup was a serial
// chain of 18 load -> pack -> store rounds on 1 000 waves: 59 us per map).
template <int NS, int DT>
__device__ __forceinline__ void outer_sum_m16_body(const float* __restrict__ x, long sx_c, long sx_l, const float* __restrict__ de, long sd_c, long sd_h,
                                                   long sd_w, int nt, f32x4* __restrict__ out, int n, int o0, unsigned* flag, int i, int oy) {
  const int j = threadIdx.x, o = oy + o0;
  bool ovf = false;
  float v[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int c = 8 * o + e;
    float t = 0.f;
    if (j < n) {
      if (c < 128) t = x[c * sx_c + i * sx_l] + x[c * sx_c + j * sx_l];
      else if (c < 128 + nt && de) t = de[(c - 128) * sd_c + i * sd_h + j * sd_w];
    }
    v[e] = t;
  }
  f32x4 a, b;
  a.x = v[0]; a.y = v[1]; a.z = v[2]; a.w = v[3]; b.x = v[4]; b.y = v[5]; b.z = v[6]; b.w = v[7];
  u32x4_t u[NS];
  m16_pack8<NS, DT>(a, b, u, ovf);
#pragma unroll
  for (int s = 0; s < NS; ++s) reinterpret_cast<u32x4_t*>(out)[m16_plane(o - o0, s, NS, n) + (long)i * M16_PX + j] = u[s];
  if (DT == 1 && ovf && flag) *flag = 1u;
}

// The separable part of the Decoder's first conv (lcombinerD.a on mat = x_i + x_j, orca_modules.py:462-465), exact fp32:
//   tab[which][cls][pos][co] = sum_k sum_c wsep[which][cls][k][c][co] * x[c][pos + k - 1]       (taps inside [0, n) only)
// which = 0: the row term, indexed by the pixel's row, one table per COLUMN class (0: first column, 1: interior, 2: last column - the
// classes differ in which kx taps were summed into wsep); which = 1: the column term per ROW class.
#define SEP_TILE 16
struct SepSmem { float xs[128][SEP_TILE + 4]; float part[4][SEP_TILE][64]; };
__device__ __forceinline__ void sep_tables_body(const float* __restrict__ x, long sx_c, long sx_l, const float* __restrict__ wsep,
                                                float* __restrict__ tab, int n, int tile, int wc, SepSmem& sm) {
  // One workgroup = weight class wc x 16 positions; block 256 = 64 couts x 4 groups of 32 input channels (partials reduced through LDS).  A thread
  // loads each of its 96 weights ONCE and uses it for the 16 positions (x tile in LDS, read as broadcast float4s): 96 workgroups per map read
  // 9.4 MB of weights from L2.  Until round 5 a workgroup was ONE position (grid (n, 6)): 150 MB of L2 reads and 11 us per map (measured then:
  // ten positions per workgroup with serial loads 28 us; one-wave workgroups without LDS 22 us).
  const int t0 = tile * SEP_TILE;
  if (t0 >= n) return;
  const int co = threadIdx.x & 63, grp = threadIdx.x >> 6;
  for (int e = threadIdx.x; e < 128 * (SEP_TILE + 2); e += 256) {
    const int c = e / (SEP_TILE + 2), q = e % (SEP_TILE + 2), p = t0 - 1 + q;
    sm.xs[c][q] = (p >= 0 && p < n) ? x[(long)c * sx_c + (long)p * sx_l] : 0.f;     // taps outside [0, n) contribute 0
  }
  __syncthreads();
  float acc[SEP_TILE];
#pragma unroll
  for (int p = 0; p < SEP_TILE; ++p) acc[p] = 0.f;
  const float* w = wsep + (size_t)wc * 3 * 128 * 64 + co;
#pragma unroll 4
  for (int cc = 0; cc < 32; ++cc) {
    const int c = grp * 32 + cc;
    const float w0 = w[(size_t)(0 * 128 + c) * 64], w1 = w[(size_t)(1 * 128 + c) * 64], w2 = w[(size_t)(2 * 128 + c) * 64];
    float xr[SEP_TILE + 4];
#pragma unroll
    for (int q4 = 0; q4 < (SEP_TILE + 4) / 4; ++q4) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(&sm.xs[c][4 * q4]);
      xr[4 * q4] = v.x; xr[4 * q4 + 1] = v.y; xr[4 * q4 + 2] = v.z; xr[4 * q4 + 3] = v.w;
    }
#pragma unroll
    for (int p = 0; p < SEP_TILE; ++p) acc[p] = fmaf(w2, xr[p + 2], fmaf(w1, xr[p + 1], fmaf(w0, xr[p], acc[p])));
  }
#pragma unroll
  for (int p = 0; p < SEP_TILE; ++p) sm.part[grp][p][co] = acc[p];
  __syncthreads();
  for (int p = grp; p < SEP_TILE; p += 4)
    if (t0 + p < n) tab[((size_t)wc * n + t0 + p) * 64 + co] = (sm.part[0][p][co] + sm.part[1][p][co]) + (sm.part[2][p][co] + sm.part[3][p][co]);
}

// bilinear / nearest x2 upsample of y [nt][n/2][n/2] into octet o0 (channels 8*o0 .. +nt-1; the rest of the octet and octet
// o0 + 1 = 0) of an M16 map.  (role of decoder_head_m16_kernel) workgroup = row i, block 256
template <int NS, int DT>
__device__ __forceinline__ void upsample2d_m16_body(const float* __restrict__ y, long sy_c, long sy_h, long sy_w, int nt, f32x4* __restrict__ out, int n,
                                                    int o0, int bilinear, unsigned* flag, int i) {
  const int j = threadIdx.x, h = n / 2;
  float v[8];
#pragma unroll
  for (int t = 0; t < 8; ++t) v[t] = 0.f;
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
  bool ovf = false;
  f32x4 a, b;
  a.x = v[0]; a.y = v[1]; a.z = v[2]; a.w = v[3]; b.x = v[4]; b.y = v[5]; b.z = v[6]; b.w = v[7];
  u32x4_t u[NS];
  m16_pack8<NS, DT>(a, b, u, ovf);
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    reinterpret_cast<u32x4_t*>(out)[m16_plane(o0, s, NS, n) + (long)i * M16_PX + j] = u[s];
    reinterpret_cast<u32x4_t*>(out)[m16_plane(o0 + 1, s, NS, n) + (long)i * M16_PX + j] = (u32x4_t)(0u);
  }
  if (DT == 1 && ovf && flag) *flag = 1u;
}

// Everything a Decoder forward computes from its INPUTS alone, for up to 8 maps of the batch, in ONE launch (VERDICT r4 #1 iii: until round 5 three
// launches per map - 44 us + five launch gaps per forward at B = 2): grid (n, roles, maps), block 256; role = blockIdx.y:
//   [0, noct)          outer_sum_m16_body: octet role of IN (the Decoder's distenc chunk / Decoder_1m's 16 outer-sum octets)
//   [noct, noct + 6)   sep_tables_body: weight class role - noct of the separable tables, blockIdx.x = tile of 16 positions (nsep = 6, Decoder only)
//   noct + nsep        upsample2d_m16_body: the coarse prediction into octets 8, 9 of A                    (when y is given)
// The roles write disjoint buffers (IN, TAB, octets 8-9 of A; the convs that fill A's octets 0-7 come later and do not touch 8-9).
struct M16HeadArgs {
  const float* x[8];
  const float* de[8];
  const float* y[8];
  long sx_c, sx_l, sd_c, sd_h, sd_w, sy_c, sy_h, sy_w;
  f32x4* in; long in_bs;      // IN maps (units per map)
  float* tab; long tab_bs;    // tables (floats per map)
  f32x4* a; long a_bs;        // A maps
  const float* wsep;
  int nt, n, o0, noct, nsep, bilinear;
  unsigned* flag;
};
template <int NS, int DT>
__global__ __launch_bounds__(256) void decoder_head_m16_kernel(M16HeadArgs a) {
  __shared__ SepSmem sm;
  const int i = blockIdx.x, role = blockIdx.y, b = blockIdx.z;
  if (role < a.noct)
    outer_sum_m16_body<NS, DT>(a.x[b], a.sx_c, a.sx_l, a.de[b], a.sd_c, a.sd_h, a.sd_w, a.nt, a.in + b * a.in_bs, a.n, a.o0, a.flag, i, role);
  else if (role < a.noct + a.nsep)
    sep_tables_body(a.x[b], a.sx_c, a.sx_l, a.wsep, a.tab + b * a.tab_bs, a.n, i, role - a.noct, sm);   // i = tile of 16 positions (the rest exit)
  else
    upsample2d_m16_body<NS, DT>(a.y[b], a.sy_c, a.sy_h, a.sy_w, a.nt, a.a + b * a.a_bs, a.n, 8, a.bilinear, a.flag, i);
}

// `final` head + symmetrisation on a 64-channel M16 map (see final_sym_kernel); a.cur = the map's units, a.cur_bs in units.
// One workgroup per PAIR of 16 x 16 pixel tiles (I, J), (J, I), I <= J: grid (136, B), block 256 = one pixel of each tile per thread.
// Both tiles are read along rows (16 px = 256 contiguous bytes per octet and split); the transposed partner of a pixel is taken from
// LDS (a thread per output row reading column i of the map, 64 lines per wave instruction and 16 of those in a row, was 33 us per map).
template <int NS, int DT>
__global__ __launch_bounds__(256) void final_sym_m16_kernel(FinalArgs a) {
  ORCA_FINAL_LOAD_HEAD();
  __shared__ float ft[2][ORCA_MAX_TARGETS][16][17];
  const int n = a.n, b = blockIdx.y, r = threadIdx.x >> 4, c = threadIdx.x & 15;
  int I = 0, p = blockIdx.x;                       // pair index -> (I, J), rows of the upper triangle have 16 - I entries
  while (p >= 16 - I) { p -= 16 - I; ++I; }
  const int J = I + p;
  const u32x4_t* cur = reinterpret_cast<const u32x4_t*>(a.cur) + (long)b * a.cur_bs;
  const int i1 = I * 16 + r, j1 = J * 16 + c;      // this thread's pixel of tile (I, J)
  const int i2 = J * 16 + r, j2 = I * 16 + c;      // ... and of tile (J, I)
  const bool in1 = i1 < n && j1 < n, in2 = i2 < n && j2 < n;
  float h1[ORCA_MAX_TARGETS], h2[ORCA_MAX_TARGETS];
#pragma unroll
  for (int o = 0; o < ORCA_MAX_TARGETS; ++o) { h1[o] = b1s[o]; h2[o] = b1s[o]; }
  // all 16 x NS units of the two pixels are requested before the first is used: one memory round trip per thread instead of eight
  u32x4_t ua[8][NS], va[8][NS];
#pragma unroll
  for (int oc = 0; oc < 8; ++oc)
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      ua[oc][s] = in1 ? cur[m16_plane(oc, s, NS, n) + (long)i1 * M16_PX + j1] : (u32x4_t)(0u);
      va[oc][s] = in2 ? cur[m16_plane(oc, s, NS, n) + (long)i2 * M16_PX + j2] : (u32x4_t)(0u);
    }
#pragma unroll
  for (int oc = 0; oc < 8; ++oc) {
    float u[8], v[8];
    m16_unpack8<NS, DT>(ua[oc], u);
    m16_unpack8<NS, DT>(va[oc], v);
#pragma unroll
    for (int e = 0; e < 8; ++e)
#pragma unroll
      for (int o = 0; o < ORCA_MAX_TARGETS; ++o)
        if (o < a.F) { h1[o] = fmaf(w1s[o * 64 + 8 * oc + e], u[e], h1[o]); h2[o] = fmaf(w1s[o * 64 + 8 * oc + e], v[e], h2[o]); }
  }
  for (int t = 0; t < a.T; ++t) {
    float f1 = b2s[t], f2 = b2s[t];
#pragma unroll
    for (int o = 0; o < ORCA_MAX_TARGETS; ++o)
      if (o < a.F) { f1 = fmaf(w2s[t * ORCA_MAX_TARGETS + o], fmaxf(h1[o], 0.f), f1); f2 = fmaf(w2s[t * ORCA_MAX_TARGETS + o], fmaxf(h2[o], 0.f), f2); }
    ft[0][t][r][c] = f1;
    ft[1][t][r][c] = f2;
  }
  __syncthreads();
  for (int t = 0; t < a.T; ++t) {
    // out(i1, j1) = (f(i1, j1) + f(j1, i1)) / 2, f(j1, i1) = tile (J, I) at (c, r); the (J, I) pixel likewise from tile (I, J)
    if (in1) {
      float* op = a.out + (long)b * a.out_bs + ((long)t * n + i1) * n + j1;
      const float v1 = 0.5f * ft[0][t][r][c] + 0.5f * ft[1][t][c][r];
      *op = a.accumulate ? (*op + v1) : v1;
    }
    if (in2 && I != J) {
      float* op = a.out + (long)b * a.out_bs + ((long)t * n + i2) * n + j2;
      const float v2 = 0.5f * ft[1][t][r][c] + 0.5f * ft[0][t][c][r];
      *op = a.accumulate ? (*op + v2) : v2;
    }
  }
}

// converters for the single-layer test entry point: [C][n][n] fp32 <-> M16 (`noct` octets; channels >= C and pad pixels zero)
template <int NS, int DT>
__global__ void nchw_to_m16_kernel(const float* __restrict__ x, int C, int n, f32x4* __restrict__ out, int noct) {
  const int j = threadIdx.x, i = blockIdx.x;
  bool ovf = false;
  for (int o = 0; o < noct; ++o) {
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { const int c = 8 * o + e; v[e] = (j < n && c < C) ? x[((long)c * n + i) * n + j] : 0.f; }
    f32x4 a, b;
    a.x = v[0]; a.y = v[1]; a.z = v[2]; a.w = v[3]; b.x = v[4]; b.y = v[5]; b.z = v[6]; b.w = v[7];
    u32x4_t u[NS];
    m16_pack8<NS, DT>(a, b, u, ovf);
#pragma unroll
    for (int s = 0; s < NS; ++s) reinterpret_cast<u32x4_t*>(out)[m16_plane(o, s, NS, n) + (long)i * M16_PX + j] = u[s];
  }
}
template <int NS, int DT>
__global__ void m16_to_nchw_kernel(const f32x4* __restrict__ in, int C, int n, float* __restrict__ y) {
  const int j = threadIdx.x, i = blockIdx.x;
  if (j >= n) return;
  for (int o = 0; o < C / 8; ++o) {
    u32x4_t u[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) u[s] = reinterpret_cast<const u32x4_t*>(in)[m16_plane(o, s, NS, n) + (long)i * M16_PX + j];
    float v[8];
    m16_unpack8<NS, DT>(u, v);
#pragma unroll
    for (int e = 0; e < 8; ++e) y[((long)(8 * o + e) * n + i) * n + j] = v[e];
  }
}

// ---- the conv ---------------------------------------------------------------------------------------------------------
struct ConvM16Args {
  const f32x4* x;     // M16 map with >= 2*nchunks octets
  const void* w;      // pack [nchunks][NS][9][2][cout][8]
  const float* bias;
  f32x4* y;           // M16 map, octets 0 .. cout/8-1 written
  const f32x4* r;     // optional residual map (64 channels)
  long x_bs, y_bs, r_bs;   // batch strides in units
  int H, W, dil, nchunks, relu;
  int banded;
  unsigned* flag;
  const float* tab;   // optional [2][3][H][64] fp32 added to the conv (sep_tables_kernel; 64-cout layers only)
  long tab_bs;        // floats per map
};

template <int N, int NS>
__device__ __forceinline__ void m16_wait(f16x8 (&x)[NS], f16x8 (&w)[NS]) {
  if constexpr (NS == 2) asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(x[0]), "+v"(x[1]), "+v"(w[0]), "+v"(w[1]) : "n"(N));
  else asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(x[0]), "+v"(w[0]) : "n"(N));
}

#ifndef M16_RAW_BARRIER
#define M16_RAW_BARRIER 1
#endif
#if M16_RAW_BARRIER
#define M16_BARRIER() asm volatile("s_barrier" ::: "memory")
#else
#define M16_BARRIER() __syncthreads()
#endif
#ifndef M16_XKEEP
#define M16_XKEEP 1
#endif
#ifndef M16_DMA_SPREAD
#define M16_DMA_SPREAD 0   // 1: one transfer per tap between the MFMA groups instead of a block at the top of the piece.  Measured: no gain for
#endif                     // f16x2 (pieces stay ~3 800 cycles), 8-15 % slower single-plane: a piece waits for its DATA
#ifndef M16_ABL
#define M16_ABL 0   // timing-only ablations (tools/microbench_m16.hip, tools/time_decoder.py): 1 no MFMA, 2 no fragment reads, 4 no epilogue loads / stores, 8 / 16 X / W DMA of the first pieces only, 32 two distinct source rows
#endif
#ifdef M16_STAMPS   // tools/microbench_m16.hip: s_memtime stamps of every wave of ONE workgroup [wave][16]
__device__ unsigned long long m16_stamp_buf[8 * 16];
#define M16_STAMP(n_) { if (blockIdx.x == M16_STAMPS && lane == 0) m16_stamp_buf[wave * 16 + (n_)] = __builtin_readcyclecounter(); }
#else
#define M16_STAMP(n_)
#endif
template <int COUT, int NS, int DT>
__global__ __launch_bounds__(512, 2) void conv2d_3x3_m16_kernel(ConvM16Args a) {
  static_assert(NS == 1 || DT == 1, "f16x2, plain fp16 or plain bf16");
  constexpr int WNS = DT == 1 ? 2 : 1;              // splits in the weight pack (the fp16 pack always carries hi and lo)
  constexpr int NT = 512, NH = COUT / 32, ROWP = 8 + M16_PX + 8;
  constexpr int XROWS = NS * 2 * 3;                 // rows of an X image: [s][g][ky]
  constexpr int XB = XROWS * ROWP;                  // units per X buffer
  constexpr int WP = NS * 9 * 2 * 32;               // units per weight piece
  constexpr int XIT = XROWS * M16_PX / NT;          // X DMA instructions per thread and chunk (6 / 3): the same for every wave
  constexpr int WIT = (WP + NT - 1) / NT;
  constexpr int EXTRA = (WP - (WIT - 1) * NT + 63) / 64;   // waves that issue WIT weight DMAs per piece (the others WIT - 1)
  static_assert(XROWS * M16_PX % NT == 0, "X image rows per DMA round");
#ifndef M16_LDS_PAD_UNITS
#define M16_LDS_PAD_UNITS 0   // experiment hook: extra LDS per workgroup (forces one workgroup per CU for the single-plane kernels)
#endif
  __shared__ f32x4 smem[2 * XB + 3 * WP + COUT / 4 + (NS == 1 ? M16_LDS_PAD_UNITS : 0)];
  f32x4* const Xs = smem;
  f32x4* const Ws = smem + 2 * XB;
  float* const bias_s = reinterpret_cast<float*>(smem + 2 * XB + 3 * WP);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, g = lane >> 5;
  const int b = blockIdx.y, H = a.H, W = a.W, d = a.dil;
  int y0 = blockIdx.x;
  if (a.banded) {   // XCD-banded row order for small dilations: the three readers of a source row share an XCD's L2
    y0 = (int)(blockIdx.x & 7) * ((H + 7) >> 3) + (int)(blockIdx.x >> 3);
    if (y0 >= H) return;
  }
  const int NP = a.nchunks * NH;

  M16_STAMP(0);
  // zero margins of both X buffers - BEFORE the first DMA (a visible LDS store behind a pending DMA makes the compiler wait for it)
  for (int i = tid; i < 2 * XROWS * 16; i += NT) {
    const int row = i >> 4, m = i & 15;
    Xs[row * ROWP + (m < 8 ? m : M16_PX + m)] = (f32x4)(0.f);
  }

  bool rowok[3];
  int ysrc[3];
#pragma unroll
  for (int ky = 0; ky < 3; ++ky) {
    const int ys = y0 + (ky - 1) * d;
    rowok[ky] = ys >= 0 && ys < H;
    ysrc[ky] = rowok[ky] ? ys : y0;      // out-of-map rows are fetched from a valid row (uniform DMA counts) and never read
    if ((M16_ABL & 32) && ky == 2) ysrc[ky] = y0;   // timing only: two distinct source rows instead of three
  }
  const f32x4* const xb = a.x + (long)b * a.x_bs;
  auto issue_x1 = [&](int k, int buf, int it) {      // one of the XIT transfers of chunk k's X image
    const int row = 2 * it + (wave >> 2), px = tid & 255;       // row = (s*2 + g)*3 + ky: one wave = 64 pixels of one row
    const int s = row / 6, gg = (row / 3) & 1, ky = row % 3;
    p16_glds16(xb + m16_plane(2 * k + gg, s, NS, H) + (long)ysrc[ky] * M16_PX + px, Xs + buf * XB + row * ROWP + 8 + (wave & 3) * 64);
  };
  auto issue_w1 = [&](int i, int it) {               // one of the WIT transfers of weight piece i = (k, h) = (i / NH, i % NH)
    const int k = i / NH, h = i - k * NH;
    const int u = tid + it * NT;
    if (u < WP) p16_glds16(reinterpret_cast<const f32x4*>(a.w) + ((long)k * (WNS * 9 * 2) + (u >> 5)) * COUT + h * 32 + (u & 31), Ws + (i % 3) * WP + it * NT + wave * 64);
  };
  auto issue_x = [&](int k, int buf) {
#pragma unroll
    for (int it = 0; it < XIT; ++it) issue_x1(k, buf, it);
  };
  auto issue_w = [&](int i) {
#pragma unroll
    for (int it = 0; it < WIT; ++it) issue_w1(i, it);
  };
  __syncthreads();                     // margins are in before anything else touches LDS
  // the bias goes in by DMA as well, ahead of the first pieces (a register-staged copy put a global-load round trip and a
  // barrier in front of the first DMA of every workgroup); it is this wave's OLDEST transfer, so every counted wait below covers it
  if (wave == 0 && lane < COUT / 4) p16_glds16(reinterpret_cast<const f32x4*>(a.bias) + lane, smem + 2 * XB + 3 * WP);
  issue_x(0, 0);
  issue_w(0);
  if (NP > 1) issue_w(1);
  M16_STAMP(1);

  f32x16 acc[NH];
#pragma unroll
  for (int h = 0; h < NH; ++h)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[h][r] = 0.f;
  const bool allrows = rowok[0] && rowok[2];
  const unsigned ws_lds = p16_lds_addr(Ws + g * 32 + l31);
  const unsigned xs_lds = p16_lds_addr(Xs + g * 3 * ROWP + 8 + wave * 32 + l31);
  const unsigned bias_lds = p16_lds_addr(bias_s + 4 * g);

  // taps in kernel-column-major order (kx, then ky) and accumulators that start from the bias: the summation order of conv2d_m16q.h, so
  // that a map comes out bit-identical whichever of the two kernels a batch size selects (tap t = ky * 3 + kx of the weight pack)
#define M16_TAP(n_) (((n_) % 3) * 3 + (n_) / 3)
#define M16_READ(buf_, t_)                                                                                          \
  if constexpr (!(M16_ABL & 2)) _Pragma("unroll") for (int s = 0; s < NS; ++s) {                                                                  \
    xv[buf_][s] = p16_lds_read16(xcol[(t_) % 3], (s * 6 + (t_) / 3) * ROWP * 16);                                   \
    wv[buf_][s] = p16_lds_read16(wrow, ((s * 9 + (t_)) * 2) * 32 * 16);                                             \
  }
#define M16_READ_XK(t_)                                                                                             \
  if constexpr (!(M16_ABL & 2)) _Pragma("unroll") for (int s = 0; s < NS; ++s) xk[t_][s] = p16_lds_read16(xcol[(t_) % 3], (s * 6 + (t_) / 3) * ROWP * 16);
#define M16_READ_W(buf_, t_)                                                                                        \
  if constexpr (!(M16_ABL & 2)) _Pragma("unroll") for (int s = 0; s < NS; ++s) wv[buf_][s] = p16_lds_read16(wrow, ((s * 9 + (t_)) * 2) * 32 * 16);
#define M16_MFMA_XK(t_, fb_, h_)                                                                                    \
  if constexpr (!(M16_ABL & 1)) {                                                                                   \
    typedef typename Op16<DT>::vec V_;                                                                              \
    if constexpr (NS == 2) {                                                                                        \
      acc[h_] = Op16<DT>::mfma(__builtin_bit_cast(V_, wv[fb_][0]), __builtin_bit_cast(V_, xk[t_][NS - 1]), acc[h_]); \
      acc[h_] = Op16<DT>::mfma(__builtin_bit_cast(V_, wv[fb_][NS - 1]), __builtin_bit_cast(V_, xk[t_][0]), acc[h_]); \
    }                                                                                                               \
    acc[h_] = Op16<DT>::mfma(__builtin_bit_cast(V_, wv[fb_][0]), __builtin_bit_cast(V_, xk[t_][0]), acc[h_]);     \
  }
#define M16_MFMA(fb_, h_)                                                                                           \
  if constexpr (!(M16_ABL & 1)) {                                                                                   \
    typedef typename Op16<DT>::vec V_;                                                                              \
    if constexpr (NS == 2) {                                                                                        \
      acc[h_] = Op16<DT>::mfma(__builtin_bit_cast(V_, wv[fb_][0]), __builtin_bit_cast(V_, xv[fb_][NS - 1]), acc[h_]); \
      acc[h_] = Op16<DT>::mfma(__builtin_bit_cast(V_, wv[fb_][NS - 1]), __builtin_bit_cast(V_, xv[fb_][0]), acc[h_]); \
    }                                                                                                               \
    acc[h_] = Op16<DT>::mfma(__builtin_bit_cast(V_, wv[fb_][0]), __builtin_bit_cast(V_, xv[fb_][0]), acc[h_]);     \
  }

  // residual units of this lane (see the epilogue), requested when the LAST piece starts: their round trip runs under its MFMAs
  const int px = wave * 32 + l31;
  const long rowoff = (long)y0 * M16_PX + px;
  const f32x4* const rb = a.r ? a.r + (long)b * a.r_bs : nullptr;
  // (f16x2 only: one workgroup per CU there anyway; the single-plane <64> kernel sits exactly on the 128 VGPRs that two workgroups per CU allow)
  constexpr int NRU = NS == 2 ? NH * 4 : 1;
  u32x4_t ru[NRU];
  for (int k = 0; k < a.nchunks; ++k) {
    f16x8 xk[NH == 2 && NS == 2 ? 9 : 1][NS];     // see M16_XKEEP below (f16x2 only: the single-plane kernels must stay under 128 VGPRs for two workgroups per CU)
    if constexpr (M16_ABL & 2) { _Pragma("unroll") for (int t = 0; t < (NH == 2 && NS == 2 ? 9 : 1); ++t) _Pragma("unroll") for (int s = 0; s < NS; ++s) xk[t][s] = (f16x8)(0); }
#pragma unroll
    for (int h = 0; h < NH; ++h) {
      const int i = k * NH + h;
      // what was issued at the top of piece i-1 may stay in flight: weight piece i+1 and - 64 couts, second half of a chunk -
      // the next chunk's X image; everything older (this piece's weights, this chunk's X image) has landed
      if (i + 1 >= NP) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      else if (NH == 2 && h == 1 && k + 1 < a.nchunks) {
        if (wave < EXTRA) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WIT + XIT) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WIT - 1 + XIT) : "memory");
      } else {
        if (wave < EXTRA) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WIT) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WIT - 1) : "memory");
      }
      M16_STAMP(2 + 2 * i);            // this piece's transfers have landed (this wave)
      // ... for every wave; everyone is done with the buffers the next DMAs go into.  A bare s_barrier: __syncthreads() is a fence, and the
      // compiler implements it as s_waitcnt vmcnt(0) lgkmcnt(0) - which retired EVERY transfer in flight at every piece and turned the two
      // pieces of weight lookahead (and the counted waits above) into one.  All LDS traffic of this loop is inline asm or DMA whose
      // completion the counted waits establish, so the barrier itself is all that is needed.
      M16_BARRIER();
      M16_STAMP(3 + 2 * i);
      if (i == 0) {                    // the bias has landed with the first piece: the accumulators start from it (register group q = couts 8q + 4g ..)
#pragma unroll
        for (int hh = 0; hh < NH; ++hh)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const f32x4 b4 = p16_lds_read16f(bias_lds, (hh * 32 + 8 * q) * 4);
            acc[hh][4 * q + 0] = b4.x; acc[hh][4 * q + 1] = b4.y; acc[hh][4 * q + 2] = b4.z; acc[hh][4 * q + 3] = b4.w;
          }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
      if constexpr (NS == 2) {
        if (i + 1 == NP && rb && !(M16_ABL & 4)) {
#pragma unroll
          for (int j = 0; j < NRU; ++j) ru[j] = reinterpret_cast<const u32x4_t*>(rb)[m16_plane(j, g, NS, H) + rowoff];      // j = octet h*4 + q
        }
      }
      // the next chunk's X image and weight piece i + 2: in a block here, or (M16_DMA_SPREAD) one transfer per tap between the MFMA groups
      const bool dox = h == 0 && k + 1 < a.nchunks && !(M16_ABL & 8), dow = i + 2 < NP && !(M16_ABL & 16);
      static_assert(XIT + WIT <= 9, "one transfer per tap");
#define M16_DMA_SLOT(t_)                                                                       \
  {                                                                                            \
    if ((t_) < XIT) { if (dox) issue_x1(k + 1, (k + 1) & 1, (t_)); }                           \
    else if ((t_) - XIT < WIT) { if (dow) issue_w1(i + 2, (t_) - XIT); }                       \
  }
      if (!allrows || !M16_DMA_SPREAD) {      // edge rows skip taps: everything up front
        if (dox) issue_x(k + 1, (k + 1) & 1);
        if (dow) issue_w(i + 2);
      }
      const unsigned wrow = ws_lds + (unsigned)((i % 3) * WP * 16);
      unsigned xcol[3];
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) xcol[kx] = xs_lds + (unsigned)(((k & 1) * XB + (kx - 1) * d) * 16);
      f16x8 xv[2][NS], wv[2][NS];
      if constexpr (M16_ABL & 2) { _Pragma("unroll") for (int q = 0; q < 2; ++q) _Pragma("unroll") for (int s = 0; s < NS; ++s) { xv[q][s] = (f16x8)(0); wv[q][s] = (f16x8)(0); } }
      if (allrows && NH == 2 && NS == 2 && M16_XKEEP) {
        // 64 couts, interior rows: the X fragments of a chunk are the same for both cout halves - read for the first half, kept in
        // registers (9 taps x NS x 4 VGPRs) for the second, which then issues weight reads only.  The workgroup is LDS-read bound
        // (4 ds_read_b128 per 3 MFMAs and wave, 8 waves on one 128 B/clk port): 54 instead of 72 reads per chunk and wave.
        if (h == 0) {
          M16_READ_XK(0); M16_READ_W(0, 0);
#pragma unroll
          for (int n = 0; n < 9; ++n) {
            const int fb = n & 1, t = M16_TAP(n), t1 = M16_TAP(n + 1);
            if (M16_DMA_SPREAD) M16_DMA_SLOT(n);
            if (n + 1 < 9) { M16_READ_XK(t1); M16_READ_W(fb ^ 1, t1); m16_wait<2 * NS, NS>(xk[t], wv[fb]); }
            else m16_wait<0, NS>(xk[t], wv[fb]);
            M16_MFMA_XK(t, fb, h);
            __builtin_amdgcn_sched_barrier(0);
          }
        } else {
          M16_READ_W(0, 0);
#pragma unroll
          for (int n = 0; n < 9; ++n) {
            const int fb = n & 1, t = M16_TAP(n), t1 = M16_TAP(n + 1);
            if (M16_DMA_SPREAD) M16_DMA_SLOT(n);
            if (n + 1 < 9) { M16_READ_W(fb ^ 1, t1); m16_wait<NS, NS>(xk[t], wv[fb]); }
            else m16_wait<0, NS>(xk[t], wv[fb]);
            M16_MFMA_XK(t, fb, h);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      } else if (allrows) {            // interior rows: all nine taps, fragments double-buffered across taps
        M16_READ(0, 0);
#pragma unroll
        for (int n = 0; n < 9; ++n) {
          const int fb = n & 1, t1 = M16_TAP(n + 1);
          if (M16_DMA_SPREAD) M16_DMA_SLOT(n);
          if (n + 1 < 9) { M16_READ(fb ^ 1, t1); m16_wait<2 * NS, NS>(xv[fb], wv[fb]); }
          else m16_wait<0, NS>(xv[fb], wv[fb]);
          M16_MFMA(fb, h);
          __builtin_amdgcn_sched_barrier(0);
        }
      } else {                         // rows near the top / bottom edge: taps of out-of-map source rows are skipped
#pragma unroll
        for (int n = 0; n < 9; ++n) {
          const int t = M16_TAP(n);
          if (!rowok[t / 3]) continue;
          M16_READ(0, t);
          m16_wait<0, NS>(xv[0], wv[0]);
          M16_MFMA(0, h);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
  }
#undef M16_READ
#undef M16_TAP
#undef M16_MFMA
#undef M16_DMA_SLOT
#undef M16_READ_XK
#undef M16_READ_W
#undef M16_MFMA_XK

  // ---- epilogue: bias, ReLU, residual, back to M16 units (the P16 / B16 recipe: after v_permlane32_swap every lane holds
  // one whole 16-byte unit - g = 0 the hi (or even-octet) one, g = 1 the lo (or odd-octet) one; 512 contiguous bytes per half wave)
  M16_STAMP(10);                       // MFMAs of the last piece issued
  const bool pxok = px < W;
  f32x4* const yb = a.y + (long)b * a.y_bs;
  // separable-part tables: row term [column class of px][y0][co], column term [row class of y0][px][co]
  const float* tabr = nullptr;
  const float* tabc = nullptr;
  if (a.tab && pxok) {
    const float* tb = a.tab + (long)b * a.tab_bs;
    const int xc = px == 0 ? 0 : (px == W - 1 ? 2 : 1), yc = y0 == 0 ? 0 : (y0 == H - 1 ? 2 : 1);
    tabr = tb + ((long)xc * H + y0) * 64;
    tabc = tb + ((long)(3 + yc) * H + px) * 64;
  }
  float vmax = 0.f;
  if constexpr (NS == 2) {
#pragma unroll
    for (int h = 0; h < NH; ++h)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int o = h * 4 + q;
        f32x4 v;
        v.x = acc[h][4 * q + 0]; v.y = acc[h][4 * q + 1]; v.z = acc[h][4 * q + 2]; v.w = acc[h][4 * q + 3];
        if (tabr) {
          const f32x4 tr = *reinterpret_cast<const f32x4*>(tabr + h * 32 + 8 * q + 4 * g), tc = *reinterpret_cast<const f32x4*>(tabc + h * 32 + 8 * q + 4 * g);
          v.x += tr.x + tc.x; v.y += tr.y + tc.y; v.z += tr.z + tc.z; v.w += tr.w + tc.w;
        }
        if (a.relu) { v.x = p16_vmax(v.x, 0.f); v.y = p16_vmax(v.y, 0.f); v.z = p16_vmax(v.z, 0.f); v.w = p16_vmax(v.w, 0.f); }
        if (rb && !(M16_ABL & 4)) {
          const u32x4_t u_ = ru[o];                                                                    // g = 0: the hi unit, g = 1: the lo unit
          unsigned ux_ = u_.x, uy_ = u_.y, uz_ = u_.z, uw_ = u_.w;
          p16_swap32(ux_, uz_);
          p16_swap32(uy_, uw_);
          const f16x2 h0_ = __builtin_bit_cast(f16x2, ux_), h1_ = __builtin_bit_cast(f16x2, uy_);
          const f16x2 l0_ = __builtin_bit_cast(f16x2, uz_), l1_ = __builtin_bit_cast(f16x2, uw_);
          v.x += (float)h0_.x + (float)l0_.x; v.y += (float)h0_.y + (float)l0_.y;
          v.z += (float)h1_.x + (float)l1_.x; v.w += (float)h1_.y + (float)l1_.y;
        }
        if (!pxok) v = (f32x4)(0.f);
        vmax = p16_vmax3_abs(p16_vmax3_abs(vmax, v.x, v.y), v.z, v.w);
        unsigned h0_, h1_, l0_, l1_;
        p16_split_hl(v, h0_, h1_, l0_, l1_);
        p16_swap32(h0_, l0_);
        p16_swap32(h1_, l1_);
        u32x4_t unit_;
        unit_.x = h0_; unit_.y = h1_; unit_.z = l0_; unit_.w = l1_;
        if (!(M16_ABL & 4) || unit_.x == 0x12345u) m16_store_unit(reinterpret_cast<u32x4_t*>(yb) + m16_plane(o, g, NS, H) + rowoff, unit_);
      }
    if (vmax > 65504.f && a.flag) *a.flag = 1u;
  } else {
#pragma unroll
    for (int h = 0; h < NH; ++h)
#pragma unroll
      for (int qp = 0; qp < 2; ++qp) {
        const int o = h * 4 + 2 * qp;                 // octets o (q0 = 2 qp) and o + 1
        f32x4 v0, v1;
        v0.x = acc[h][8 * qp + 0]; v0.y = acc[h][8 * qp + 1]; v0.z = acc[h][8 * qp + 2]; v0.w = acc[h][8 * qp + 3];
        v1.x = acc[h][8 * qp + 4]; v1.y = acc[h][8 * qp + 5]; v1.z = acc[h][8 * qp + 6]; v1.w = acc[h][8 * qp + 7];
        if (tabr) {
          const int co_ = h * 32 + 16 * qp + 4 * g;
          const f32x4 r0 = *reinterpret_cast<const f32x4*>(tabr + co_), c0 = *reinterpret_cast<const f32x4*>(tabc + co_);
          const f32x4 r1 = *reinterpret_cast<const f32x4*>(tabr + co_ + 8), c1 = *reinterpret_cast<const f32x4*>(tabc + co_ + 8);
          v0.x += r0.x + c0.x; v0.y += r0.y + c0.y; v0.z += r0.z + c0.z; v0.w += r0.w + c0.w;
          v1.x += r1.x + c1.x; v1.y += r1.y + c1.y; v1.z += r1.z + c1.z; v1.w += r1.w + c1.w;
        }
        if (a.relu) {
          v0.x = p16_vmax(v0.x, 0.f); v0.y = p16_vmax(v0.y, 0.f); v0.z = p16_vmax(v0.z, 0.f); v0.w = p16_vmax(v0.w, 0.f);
          v1.x = p16_vmax(v1.x, 0.f); v1.y = p16_vmax(v1.y, 0.f); v1.z = p16_vmax(v1.z, 0.f); v1.w = p16_vmax(v1.w, 0.f);
        }
        if (rb) {                                     // the lane loads the whole unit of octet o + g
          const u32x4_t u_ = reinterpret_cast<const u32x4_t*>(rb)[m16_plane(o + g, 0, NS, H) + rowoff];
          unsigned ux_ = u_.x, uy_ = u_.y, uz_ = u_.z, uw_ = u_.w;
          p16_swap32(ux_, uz_);
          p16_swap32(uy_, uw_);
          float t0, t1;
          m16_pair<DT>(ux_, t0, t1); v0.x += t0; v0.y += t1;
          m16_pair<DT>(uy_, t0, t1); v0.z += t0; v0.w += t1;
          m16_pair<DT>(uz_, t0, t1); v1.x += t0; v1.y += t1;
          m16_pair<DT>(uw_, t0, t1); v1.z += t0; v1.w += t1;
        }
        if (!pxok) { v0 = (f32x4)(0.f); v1 = (f32x4)(0.f); }
        if (DT == 1) vmax = p16_vmax3_abs(p16_vmax3_abs(p16_vmax3_abs(p16_vmax3_abs(vmax, v0.x, v0.y), v0.z, v0.w), v1.x, v1.y), v1.z, v1.w);
        unsigned a0_ = m16_pk2<DT>(v0.x, v0.y), a1_ = m16_pk2<DT>(v0.z, v0.w), b0_ = m16_pk2<DT>(v1.x, v1.y), b1_ = m16_pk2<DT>(v1.z, v1.w);
        p16_swap32(a0_, b0_);
        p16_swap32(a1_, b1_);
        u32x4_t unit_;
        unit_.x = a0_; unit_.y = a1_; unit_.z = b0_; unit_.w = b1_;
        m16_store_unit(reinterpret_cast<u32x4_t*>(yb) + m16_plane(o + g, 0, NS, H) + rowoff, unit_);
      }
    if (DT == 1 && vmax > 65504.f && a.flag) *a.flag = 1u;
  }
  M16_STAMP(11);                       // epilogue stores issued
}
