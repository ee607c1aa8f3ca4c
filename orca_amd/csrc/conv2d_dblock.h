// conv2d_dblock.h - one whole dilated residual block of the Decoders in ONE launch, for dilations >= 16.
//
// A Decoder is 28 (Decoder_1m: 19) blocks   oth = lm(cur) + cur;  cur = m(oth) + oth   of four 3x3 convolutions
// 64 -> 32 -> 64 -> 32 -> 64 that all share one dilation d (orca_modules.py:22-422, :477-486; dilations 1..64 cycling).
// A 3x3 conv with dilation d only ever combines pixels whose rows AND columns are congruent mod d: the 250 x 250 map is
// d*d independent sub-images of ceil(250/d)^2 pixels (16x16 at d = 16, 8x8 at 32, 4x4 at 64) on each of which the block
// is four PLAIN 3x3 convs with zero padding at the sub-image border (= the map border).  A workgroup takes 256 pixels'
// worth of sub-images (1 / 4 / 16 of them), gathers them, and runs the four convs back to back out of LDS: no halo, no
// recompute, no intermediate map in HBM, the result written IN PLACE - 1 launch and 1 read + 1 write of the 16 MB map
// instead of 4 launches and 4 reads (each 3x) + 4 writes.  The per-layer kernel (conv2d_f16s.h) is bound by a latency
// chain per workgroup (load 3 rows -> split -> LDS -> barrier, 2-4 times for ~1 us of MFMA work); here a workgroup has
// 16 x 27 MFMAs per wave between its gather and its store and only the 18 KB weight pieces stream (LDS-DMA, 3-deep ring).
//
// LDS: A = 64-channel operand image [split][8 channel octets][256 px + zero unit], B = the 32-channel one, W ring of
// three pieces [split][9 taps][2][32 couts]; 154 KB in f16x2, 77 KB in bf16.  A lane's operand for tap (dy, dx) is the
// unit of pixel p + dy*S + dx of its sub-image (S = side) or the zero unit when that leaves the sub-image: the padding
// costs one address select per read.  Arithmetic and rounding points are those of the per-layer kernels: operands split
// to 2 x fp16 (3 products) or rounded to bf16 (1 product), fp32 accumulate, fp32 residual stream (cur / oth stay in
// registers in fp32 and go back to HBM in fp32).
#pragma once
#include "conv2d_f16s.h"
#include "conv_p16.h"

struct DBlockArgs {
  float* cur;              // [B][4 chunks][H][256][16] fp32, updated in place
  const void* w[4];        // lm.a (64->32), lm.b (32->64), m.a (64->32), m.b (32->64): packs [cin/16][NS][9][2][cout][8]
  const float* bias[4];
  long bs, cs;             // batch / chunk strides (floats)
  int H, W, dil;
  unsigned* flag;
};

// counted LDS wait that pins the fragments it guards (see p16_lds_wait)
template <int N, int NS>
__device__ __forceinline__ void dblock_wait(f16x8 (&x)[NS], f16x8 (&w)[NS]) {
  if constexpr (NS == 2) asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(x[0]), "+v"(x[1]), "+v"(w[0]), "+v"(w[1]) : "n"(N));
  else asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(x[0]), "+v"(w[0]) : "n"(N));
}

// ABL (tools/microbench_dblock.hip only, 0 in the library): 1 = no MFMAs, 2 = weight pieces fetched once (no DMA in the loop),
// 4 = no gather / residual loads, 8 = no operand reads, 16 = no piece barriers (timing only: races)
template <int NS, int DT, int ABL = 0>
__global__ __launch_bounds__(512, NS == 1 ? 4 : 2) void conv2d_dblock_kernel(DBlockArgs a) {   // bf16: 77 KB of LDS, two workgroups per CU if <= 128 VGPRs
  constexpr int NT = 512, PXW = 257;                 // 256 pixels + the zero unit
  constexpr int AU = NS * 8 * PXW, BU = NS * 4 * PXW; // operand images (16-byte units)
  constexpr int WP = NS * 9 * 2 * 32;                // one weight piece: 16 input channels x 32 couts
  constexpr int NPIECE = 16;
  constexpr int WIT = (WP + NT - 1) / NT;
  __shared__ f32x4 smem[AU + BU + 3 * WP + 48];
  f32x4* const As = smem;
  f32x4* const Bs = smem + AU;
  f32x4* const Ws = smem + AU + BU;
  float* const bias_s = reinterpret_cast<float*>(smem + AU + BU + 3 * WP);   // [32 | 64 | 32 | 64]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, g = lane >> 5;
  const int d = a.dil, H = a.H, W = a.W;                // d = 16 / 32 / 64
  const int ld = 31 - __builtin_clz(d);                 // log2(d)
  const int lS = 8 - ld, S = 1 << lS;                   // sub-image side 16 / 8 / 4
  const int lG = 2 * ld - 8, G = 1 << lG;               // sub-images per workgroup 1 / 4 / 16 (all in one row of sub-images)
  float* const cur = a.cur + (long)blockIdx.y * a.bs;

  // pixel p of the workgroup -> (sub-image, r, c) -> map (row, col)
  // Workgroups go to the XCDs round-robin (b % 8).  At d = 16 a workgroup is ONE sub-image and the sub-images (i0, 2m) and
  // (i0, 2m + 1) interleave their pixels inside every 128-byte line: blocks b and b + 8 take such a pair, so that the line is
  // fetched into one XCD's L2 only.  (At d = 32 / 64 a workgroup holds 4 / 16 consecutive j0 itself.)
  const int bx = (G == 1 && !(ABL & 64)) ? (int)((blockIdx.x & ~15u) | ((blockIdx.x & 7u) << 1) | ((blockIdx.x >> 3) & 1u)) : (int)blockIdx.x;
  auto pix = [&](int p, int& row, int& col, int& r, int& c) {
    const int sid = (bx << lG) + (p >> (2 * lS)), q = p & ((1 << (2 * lS)) - 1);
    r = q >> lS; c = q & (S - 1);
    row = (sid >> ld) + (r << ld);
    col = (sid & (d - 1)) + (c << ld);
  };
  {   // workgroup-uniform early exit (small maps): its sub-images start at row i0, columns j0 .. j0 + G - 1
    const int sid0 = bx << lG;
    if ((sid0 >> ld) >= H || (sid0 & (d - 1)) >= W) return;
  }

  // ---- biases, zero units -------------------------------------------------------------------------------------
  if (tid < 192) {
    const int L = tid < 32 ? 0 : (tid < 96 ? 1 : (tid < 128 ? 2 : 3)), o = tid - (tid < 32 ? 0 : (tid < 96 ? 32 : (tid < 128 ? 96 : 128)));
    bias_s[tid] = a.bias[L][o];
  }
  if (tid < NS * 8) As[tid * PXW + 256] = (f32x4)(0.f);
  if (tid >= 64 && tid < 64 + NS * 4) Bs[(tid - 64) * PXW + 256] = (f32x4)(0.f);

  // ---- weight pieces: piece i = (layer, K-chunk k, cout half h) ------------------------------------------------
  // layers 0 / 2 (64 -> 32): pieces k = 0..3;  layers 1 / 3 (32 -> 64): h = 0: k = 0, 1;  h = 1: k = 0, 1
  auto piece_src = [&](int i, int u) -> const f32x4* {
    const int L = i >> 2, j = i & 3;
    const bool wide = L & 1;                       // 32 -> 64
    const int k = wide ? (j & 1) : j, h = wide ? (j >> 1) : 0, cout = wide ? 64 : 32;
    const int grp = u >> 5, co = u & 31;           // grp = (s*9 + tap)*2 + g
    return reinterpret_cast<const f32x4*>(a.w[L]) + ((long)k * (NS * 9 * 2) + grp) * cout + h * 32 + co;
  };
  auto issue_piece = [&](int i) {
    f32x4* dst = Ws + (i % 3) * WP;
#pragma unroll
    for (int it = 0; it < WIT; ++it) {
      const int u = tid + it * NT;
      if (u < WP) p16_glds16(piece_src(i, u), dst + it * NT + wave * 64);
    }
  };
  issue_piece(0);
  issue_piece(1);

  // ---- this lane's pixel in the MFMA layout, its residual (fp32, registers) and its neighbour units -------------
  const int p = wave * 32 + l31;
  int prow, pcol, pr, pc;
  pix(p, prow, pcol, pr, pc);
  const bool pvalid = prow < H && pcol < W;
  unsigned nb16[9];                                 // byte offset of the tap's source unit within a plane (256 = the zero unit)
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    const int dy = t / 3 - 1, dx = t % 3 - 1;
    const bool in = (unsigned)(pr + dy) < (unsigned)S && (unsigned)(pc + dx) < (unsigned)S;
    nb16[t] = (in ? (unsigned)(p + dy * S + dx) : 256u) * 16u;
  }

  // ---- gather cur into A.  A pixel's 16 channels of one chunk are 64 contiguous bytes, pixels of a sub-image are d columns
  // (>= 1 KB) apart: FOUR consecutive lanes fetch the four 16-byte pieces of one (pixel, chunk) segment, so a wave-level load
  // touches 16 segments instead of 64 (measured: the gather was 15 of the kernel's 38 us with one lane per pixel).
  bool overflow = false;
  const int piece4 = tid & 3, gpl = tid >> 2;         // 128 pixels per round
  f32x4 gv[8];                                        // cur, later oth: the fp32 residual stream, (pixel gpx[rnd], channels 16k + 4 piece4 .. +3)
  int gpx[2];
  bool gok[2];
  long goff[2];
#pragma unroll
  for (int rnd = 0; rnd < 2; ++rnd) {
    gpx[rnd] = rnd * 128 + gpl;
    int row, col, r_, c_;
    pix(gpx[rnd], row, col, r_, c_);
    gok[rnd] = row < H && col < W;
    goff[rnd] = ((long)row * 256 + col) * 16 + 4 * piece4;
#pragma unroll
    for (int k = 0; k < 4; ++k) gv[rnd * 4 + k] = (gok[rnd] && !(ABL & 4)) ? *reinterpret_cast<const f32x4*>(cur + goff[rnd] + (long)k * a.cs) : (f32x4)(0.f);
  }
  // (pixel, channels 16k + 4 piece4 .. +3) -> octet 2k + piece4/2 of the 64-channel operand image, 8-byte half piece4 & 1
  auto to_A = [&]() {
#pragma unroll
    for (int rnd = 0; rnd < 2; ++rnd)
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        u32x2 sp[NS];
        split4<NS, DT>(gv[rnd * 4 + k], sp, overflow);
#pragma unroll
        for (int s = 0; s < NS; ++s)
          *reinterpret_cast<u32x2*>(reinterpret_cast<char*>(As + (s * 8 + 2 * k + (piece4 >> 1)) * PXW + gpx[rnd]) + 8 * (piece4 & 1)) = sp[s];
      }
  };
  to_A();
  // Epilogue of a 64-cout layer: the accumulators (MFMA layout: lane = pixel, 4 couts per register group) go through a
  // 64 KB fp32 staging tile in the (then idle) A region back to the GATHER layout, where the residual stream lives in
  // registers and a (pixel, chunk) segment is 4 consecutive lanes: no second, scattered read of the map for the residual
  // (it cost 9 of 34 us) and 64-byte-coalesced final stores.  16-byte granule c of pixel p sits at granule c ^ f(p),
  // f(p) = 4 (p & 3) + ((p >> 2) & 3): conflict-free both for the writes (16 consecutive pixels, one granule) and for the
  // reads (4 consecutive pixels x 4 granules).
  float* const stage = reinterpret_cast<float*>(As);
  const int fsw_p = 4 * (p & 3) + ((p >> 2) & 3);
  f32x16 acc[2];
  int piece = 0;
  constexpr int EXTRA = (WP - (WIT - 1) * NT + 63) / 64;     // waves that issue WIT (the others WIT - 1) DMA instructions per piece
  const unsigned as_lds = p16_lds_addr(As), bs_lds = p16_lds_addr(Bs), ws_lds = p16_lds_addr(Ws + g * 32 + l31);
  // One layer = four weight pieces.  WIDE: 32 -> 64 (pieces (h, k) = (0,0) (0,1) (1,0) (1,1)), else 64 -> 32 (k = 0..3).
  // The operand reads are inline asm with counted waits, as in conv_p16.h: with an LDS-DMA in flight the compiler would
  // put s_waitcnt vmcnt(0) in front of every LDS read it can see.
#define DB_READ(buf_, t_)                                                                                         \
  _Pragma("unroll") for (int s = 0; s < NS; ++s) {                                                                \
    xv[buf_][s] = p16_lds_read16(xrow + nb16[t_], s * XG * PXW * 16);                                             \
    wv[buf_][s] = p16_lds_read16(wrow, ((s * 9 + (t_)) * 2) * 32 * 16);                                           \
  }
#define DB_LAYER(XLDS, XG_, WIDE)                                                                                 \
  {                                                                                                               \
    constexpr int XG = XG_;                                                                                       \
    _Pragma("unroll") for (int j = 0; j < 4; ++j, ++piece) {                                                      \
      constexpr bool wide_ = WIDE;                                                                                \
      const int k = wide_ ? (j & 1) : j;                                                                          \
      const int h = wide_ ? (j >> 1) : 0;                                                                         \
      /* this piece's weights have landed (issued two pieces ago; the next piece's may still be in flight) */     \
      if (piece + 1 < NPIECE) {                                                                                   \
        if (wave < EXTRA) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WIT) : "memory");                             \
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WIT - 1) : "memory");                                      \
      } else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                     \
      if (!(ABL & 16) || j == 0) __syncthreads();   /* ... for every wave, and everyone is done with the buffer piece + 2 goes into */ \
      if (piece + 2 < NPIECE && !(ABL & 2)) issue_piece(piece + 2);                                               \
      const unsigned wrow = ws_lds + (unsigned)((piece % 3) * WP * 16);                                           \
      const unsigned xrow = (XLDS) + (unsigned)((2 * k + g) * PXW * 16);                                          \
      f16x8 xv[2][NS], wv[2][NS];                                                                                 \
      DB_READ(0, 0);                                                                                              \
      _Pragma("unroll") for (int t = 0; t < 9; ++t) {                                                             \
        const int fb = t & 1;                                                                                     \
        if (t + 1 < 9) { if (!(ABL & 8)) { DB_READ(fb ^ 1, t + 1); dblock_wait<2 * NS, NS>(xv[fb], wv[fb]); } }   \
        else dblock_wait<0, NS>(xv[fb], wv[fb]);                                                                  \
        typedef typename Op16<DT>::vec V_;                                                                        \
        if constexpr ((ABL & 1) != 0) { asm volatile("" ::"v"(xv[fb][0]), "v"(wv[fb][0])); } else                \
        if constexpr (NS == 2) {                                                                                  \
          acc[h] = Op16<DT>::mfma(__builtin_bit_cast(V_, wv[fb][0]), __builtin_bit_cast(V_, xv[fb][NS - 1]), acc[h]); \
          acc[h] = Op16<DT>::mfma(__builtin_bit_cast(V_, wv[fb][NS - 1]), __builtin_bit_cast(V_, xv[fb][0]), acc[h]); \
        }                                                                                                         \
        if constexpr ((ABL & 1) == 0) acc[h] = Op16<DT>::mfma(__builtin_bit_cast(V_, wv[fb][0]), __builtin_bit_cast(V_, xv[fb][0]), acc[h]); \
        __builtin_amdgcn_sched_barrier(0);                                                                        \
      }                                                                                                           \
    }                                                                                                             \
  }
  auto acc_to_gather = [&](int boff, bool relu, f32x4 (&out)[8]) {
    if constexpr (NS == 2) {   // the A region (65.8 KB) holds the whole 64 KB tile
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int gr = (h * 32 + 8 * q + 4 * g) >> 2;
          f32x4 v;
          v.x = acc[h][4 * q + 0]; v.y = acc[h][4 * q + 1]; v.z = acc[h][4 * q + 2]; v.w = acc[h][4 * q + 3];
          *reinterpret_cast<f32x4*>(stage + p * 64 + ((gr ^ fsw_p) << 2)) = v;
        }
      __syncthreads();
#pragma unroll
      for (int rnd = 0; rnd < 2; ++rnd) {
        const int px = gpx[rnd], fsw = 4 * (px & 3) + ((px >> 2) & 3);
#pragma unroll
        for (int k = 0; k < 4; ++k) out[rnd * 4 + k] = *reinterpret_cast<const f32x4*>(stage + px * 64 + (((4 * k + piece4) ^ fsw) << 2));
      }
    } else {                   // bf16: the A region is 32.9 KB - one 32-cout half (32 KB) at a time; granule c of pixel p at
                               // c ^ f8(p), f8(p) = 4 ((p >> 1) & 1) + ((p >> 2) & 3), pixel pitch 128 bytes
      const int f8_p = 4 * ((p >> 1) & 1) + ((p >> 2) & 3);
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        if (h) __syncthreads();                    // the first half has been read
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          f32x4 v;
          v.x = acc[h][4 * q + 0]; v.y = acc[h][4 * q + 1]; v.z = acc[h][4 * q + 2]; v.w = acc[h][4 * q + 3];
          *reinterpret_cast<f32x4*>(stage + p * 32 + (((2 * q + g) ^ f8_p) << 2)) = v;
        }
        __syncthreads();
#pragma unroll
        for (int rnd = 0; rnd < 2; ++rnd) {
          const int px = gpx[rnd], f8 = 4 * ((px >> 1) & 1) + ((px >> 2) & 3);
#pragma unroll
          for (int kk = 0; kk < 2; ++kk) out[rnd * 4 + 2 * h + kk] = *reinterpret_cast<const f32x4*>(stage + px * 32 + (((4 * kk + piece4) ^ f8) << 2));
        }
      }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const f32x4 b = *reinterpret_cast<const f32x4*>(bias_s + boff + 4 * (4 * (i & 3) + piece4));
      f32x4 v = out[i] + b;
      if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
      out[i] = v;
    }
  };
  // epilogue of a 32-cout layer: bias (+ReLU), invalid pixels -> 0, split, into B
  auto to_B = [&](int boff, bool relu) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 b = *reinterpret_cast<const f32x4*>(bias_s + boff + 8 * q + 4 * g);
      f32x4 v;
      v.x = acc[0][4 * q + 0] + b.x; v.y = acc[0][4 * q + 1] + b.y; v.z = acc[0][4 * q + 2] + b.z; v.w = acc[0][4 * q + 3] + b.w;
      if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
      if (!pvalid) v = (f32x4)(0.f);
      u32x2 sp[NS];
      split4<NS, DT>(v, sp, overflow);
#pragma unroll
      for (int s = 0; s < NS; ++s) *reinterpret_cast<u32x2*>(reinterpret_cast<char*>(Bs + (s * 4 + q) * PXW + p) + 8 * g) = sp[s];
    }
  };
  auto zero_acc = [&]() {
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[h][r] = 0.f;
  };

  // ---- lm.a: 64 -> 32, linear --------------------------------------------------------------------------------
  zero_acc();
  DB_LAYER(as_lds, 8, false);
  to_B(0, false);
  // ---- lm.b: 32 -> 64, linear, + cur -> oth (registers) -> A ----------------------------------------------------
  zero_acc();
  DB_LAYER(bs_lds, 4, true);
  {
    f32x4 t[8];
    acc_to_gather(32, false, t);
#pragma unroll
    for (int i = 0; i < 8; ++i) gv[i] = gok[i >> 2] ? (f32x4)(t[i] + gv[i]) : (f32x4)(0.f);      // oth (invalid pixels stay zero)
    __syncthreads();                              // every thread has read the staging tile: A becomes the operand image of oth
    if (tid < NS * 8) As[tid * PXW + 256] = (f32x4)(0.f);   // the staging tile covered the zero units
    to_A();
  }
  // ---- m.a: 64 -> 32, ReLU --------------------------------------------------------------------------------------
  zero_acc();
  DB_LAYER(as_lds, 8, false);
  to_B(96, true);
  // ---- m.b: 32 -> 64, ReLU, + oth -> cur (HBM, in place) --------------------------------------------------------
  zero_acc();
  DB_LAYER(bs_lds, 4, true);
#undef DB_LAYER
#undef DB_READ
  {
    f32x4 t[8];
    acc_to_gather(128, true, t);
#pragma unroll
    for (int rnd = 0; rnd < 2; ++rnd)
      if (gok[rnd]) {
#pragma unroll
        for (int k = 0; k < 4; ++k) *reinterpret_cast<f32x4*>(cur + goff[rnd] + (long)k * a.cs) = t[rnd * 4 + k] + gv[rnd * 4 + k];
      }
  }
  if (DT == 1 && overflow && a.flag) *a.flag = 1u;
}
