// conv2d_dblock.h - one whole dilated residual block of the Decoders in ONE launch, for dilations >= 16, on M16 maps
// (conv2d_m16.h).
//
// A Decoder is 28 (Decoder_1m: 19) blocks   oth = lm(cur) + cur;  cur = m(oth) + oth   of four 3x3 convolutions
// 64 -> 32 -> 64 -> 32 -> 64 that all share one dilation d (orca_modules.py:22-422, :477-486; dilations 1..64 cycling).
// A 3x3 conv with dilation d only ever combines pixels whose rows AND columns are congruent mod d: the 250 x 250 map is
// d*d independent sub-images of ceil(250/d)^2 pixels (16x16 at d = 16, 8x8 at 32, 4x4 at 64) on each of which the block
// is four PLAIN 3x3 convs with zero padding at the sub-image border (= the map border).  A workgroup takes 256 pixels'
// worth of sub-images (1 / 4 / 16 of them), gathers their units, and runs the four convs back to back out of LDS: no halo,
// no recompute, no intermediate map in HBM, the result written IN PLACE - 1 launch and 1 read + 1 write of the map
// instead of 4 launches and 4 reads (each 3x) + 4 writes.
//
// LDS: A = 64-channel operand image [split][8 octets][256 px + 16 zero units], B = the 32-channel one, W ring of three pieces
// [split][9 taps][2][32 couts]; 157 KB in f16x2, 79 KB in bf16.  The gather is LDS-DMA straight from the M16 planes (the
// units ARE the operand image; a pixel outside the map is fetched from a zero pad pixel).  A lane's operand for tap (dy, dx)
// is the unit of pixel p + dy*S + dx of its sub-image (S = side) or the zero unit when that leaves the sub-image: the padding
// costs one address select per read.  The residuals (cur for lm, oth for m) are read back from the A image in the MFMA
// layout (hi + lo), the result goes out as whole 16-byte units per lane (P16 recipe).  Weights stream as 18 KB pieces
// through the ring, issued two pieces ahead, counted waits, one barrier per piece.
#pragma once
#include "conv2d_m16.h"

struct DBlockArgs {
  f32x4* cur;              // 64-channel M16 map, updated in place
  const void* w[4];        // lm.a (64->32), lm.b (32->64), m.a (64->32), m.b (32->64): packs [cin/16][NS][9][2][cout][8]
  const float* bias[4];
  long bs;                 // batch stride (units)
  int H, W, dil;
  unsigned* flag;
};

// The block's result goes out with PLAIN (write-back) stores, unlike the per-layer kernels' write-through ones (m16_store_unit): a workgroup's units are
// scattered - dilation 16: 16 bytes of every 128-byte line, 32: 64 bytes - and the 8 workgroups that share a line sit on one XCD, so its L2 merges them
// into whole lines before they leave; written through, every unit went out as its own partial line.  tools/microbench_dblock.hip, us per launch at
// B = 2, write-through -> write-back: d = 16 66.8 -> 55.5, d = 32 74.5 -> 61.7, d = 64 52.8 -> 48.0; single fp16 plane, B = 8: 106 -> 86, 108 -> 85,
// 84 -> 77 (round 5; DBLOCK_STORE_WT=1 at compile time: the write-through form).
#ifndef DBLOCK_STORE_WT
#define DBLOCK_STORE_WT 0
#endif
__device__ __forceinline__ void db_store_unit(u32x4_t* p, const u32x4_t& v) {
#if DBLOCK_STORE_WT
  m16_store_unit(p, v);
#else
  *p = v;
#endif
}

// ABL (tools/microbench_dblock.hip only, 0 in the library): 1 = no MFMAs, 2 = weight pieces fetched once (no DMA in the loop),
// 4 = no gather, 8 = no operand reads, 16 = no piece barriers (timing only: races)
template <int NS, int DT, int ABL = 0>
__global__ __launch_bounds__(512, NS == 1 ? 4 : 2) void conv2d_dblock_kernel(DBlockArgs a) {   // bf16: 77 KB of LDS, two workgroups per CU if <= 128 VGPRs
  constexpr int WNS = DT == 1 ? 2 : 1;               // splits in the weight pack (the fp16 pack always carries hi and lo)
  constexpr int NT = 512, PXW = 256 + 16;            // 256 pixels + 16 zero units (one per bank slot of a 16-lane read group, see nb16)
  constexpr int AU = NS * 8 * PXW, BU = NS * 4 * PXW; // operand images (16-byte units)
  constexpr int WP = NS * 9 * 2 * 32;                // one weight piece: 16 input channels x 32 couts
  constexpr int NPIECE = 16;
  constexpr int WIT = (WP + NT - 1) / NT;
  static_assert(NS * 8 * 256 / NT == 4 * NS, "gather: NS DMA instructions per thread and K-chunk of the first layer (the same for every wave)");
  constexpr int EXTRA = (WP - (WIT - 1) * NT + 63) / 64;     // waves that issue WIT (the others WIT - 1) DMA instructions per piece
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
  f32x4* const cur = a.cur + (long)blockIdx.y * a.bs;

  // Workgroups go to the XCDs round-robin (b % 8); 8 consecutive pixels of a map row (one 128-byte line of a plane) belong to
  // 8 consecutive sub-image columns j0: the workgroups holding them are placed on ONE XCD (32 per XCD = 4 runs of 8 columns at
  // d = 16, 2 workgroups x 4 columns at d = 32), so that a line is fetched into one L2 only.
  int bx = (int)blockIdx.x;
  if (!(ABL & 64)) {
    const int xcd = bx & 7, t = bx >> 3;                // t = 0..31 on this XCD
    if (G == 1) { const int run = xcd * 4 + (t >> 3); bx = (run >> 1) * 16 + (run & 1) * 8 + (t & 7); }       // sid = i0*16 + j0
    else if (G == 4) { const int pair = xcd * 16 + (t >> 1); bx = pair * 2 + (t & 1); }                        // workgroup = 4 columns: pairs
  }
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

  // ---- zero units: BEFORE the first DMA (a visible LDS store behind a pending DMA makes the compiler wait for it) ----
  if (tid < NS * 8 * 16) As[(tid >> 4) * PXW + 256 + (tid & 15)] = (f32x4)(0.f);
  if (tid >= 256 && tid < 256 + NS * 4 * 16) Bs[((tid - 256) >> 4) * PXW + 256 + (tid & 15)] = (f32x4)(0.f);
  __syncthreads();
  // ---- the four biases [32 | 64 | 32 | 64] by ONE DMA instruction of wave 0 (48 lanes x 16 bytes, per-lane sources): a register-staged
  // copy put a global-load round trip in front of the gather.  It is the wave's oldest transfer: every counted wait below covers it ----
  if (wave == 0 && lane < 48) {
    const float* src = lane < 8 ? a.bias[0] + 4 * lane : (lane < 24 ? a.bias[1] + 4 * (lane - 8) : (lane < 32 ? a.bias[2] + 4 * (lane - 24) : a.bias[3] + 4 * (lane - 32)));
    p16_glds16(reinterpret_cast<const f32x4*>(src), smem + AU + BU + 3 * WP);
  }

  // ---- weight pieces: piece i = (layer, K-chunk k, cout half h) ------------------------------------------------
  // layers 0 / 2 (64 -> 32): pieces k = 0..3;  layers 1 / 3 (32 -> 64): h = 0: k = 0, 1;  h = 1: k = 0, 1
  auto piece_src = [&](int i, int u) -> const f32x4* {
    const int L = i >> 2, j = i & 3;
    const bool wide = L & 1;                       // 32 -> 64
    const int k = wide ? (j & 1) : j, h = wide ? (j >> 1) : 0, cout = wide ? 64 : 32;
    const int grp = u >> 5, co = u & 31;           // grp = (s*9 + tap)*2 + g
    return reinterpret_cast<const f32x4*>(a.w[L]) + ((long)k * (WNS * 9 * 2) + grp) * cout + h * 32 + co;
  };
  auto issue_piece = [&](int i) {
    f32x4* dst = Ws + (i % 3) * WP;
#pragma unroll
    for (int it = 0; it < WIT; ++it) {
      const int u = tid + it * NT;
      if (u < WP) p16_glds16(piece_src(i, u), dst + it * NT + wave * 64);
    }
  };
  // ---- gather: the units of the workgroup's 256 pixels, all 8 octets x NS planes, by LDS-DMA; a pixel outside the map is
  // fetched from pad pixel 255 of row 0 (zero; maps with W = 256 have no outside pixels).  Issued K-CHUNK BY K-CHUNK of the first layer
  // (octets 2k, 2k + 1 of every split), the first three weight pieces in between: piece k of layer 0 waits for chunk k alone (counted
  // waits in DB_LAYER), so three quarters of the cold gather run under the first pieces' MFMAs ----
  {
    int row, col, r_, c_;
    pix(tid & 255, row, col, r_, c_);
    const bool ok = row < H && col < W;
    const long off = ok ? (long)row * M16_PX + col : 255;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
#pragma unroll
      for (int s_ = 0; s_ < NS; ++s_) {
        const int plane = s_ * 8 + 2 * k + (wave >> 2);            // = s*8 + o
        if (!(ABL & 4)) p16_glds16(cur + m16_plane(plane & 7, s_, NS, H) + off, As + plane * PXW + (wave & 3) * 64);
      }
      if (k < 3) issue_piece(k);
    }
  }

  // ---- this lane's pixel in the MFMA layout and its neighbour units -----------------------------------------------
  const int p = wave * 32 + l31;
  int prow, pcol, pr, pc;
  pix(p, prow, pcol, pr, pc);
  const bool pvalid = prow < H && pcol < W;
  const long poff = (long)prow * M16_PX + pcol;
  // byte offset of the tap's source unit within a plane.  Outside the sub-image: a zero unit - the one in the SAME bank slot as the unit the
  // lane would have read (256 + (q & 15)): the 16 lanes of a ds_read_b128 group read 16 consecutive units = all 64 banks, so a single zero
  // unit collided with one of them whenever a group mixed inside and outside lanes (SQ_LDS_BANK_CONFLICT 18.7 % of SQ_LDS_IDX_ACTIVE, round 4)
  unsigned nb16[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    const int dy = t / 3 - 1, dx = t % 3 - 1;
    const bool in = (unsigned)(pr + dy) < (unsigned)S && (unsigned)(pc + dx) < (unsigned)S;
    nb16[t] = (in ? (unsigned)(p + dy * S + dx) : 256u + ((unsigned)(p + dy * S + dx) & 15u)) * 16u;
  }

  f32x16 acc[2];
  int piece = 0;
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
      /* this piece's weights (and, pieces 0-2, its K-chunk of the gather) have landed; what was issued behind them may still be  */ \
      /* in flight: G1 W1 G2 W2 G3 behind piece 0, G2 W2 G3 behind piece 1, G3 W3 behind piece 2, then the next piece's weights    */ \
      if (piece == 0) {                                                                                           \
        if (wave < EXTRA) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * NS + 2 * WIT) : "memory");                \
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * NS + 2 * (WIT - 1)) : "memory");                       \
      } else if (piece == 1) {                                                                                    \
        if (wave < EXTRA) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NS + WIT) : "memory");                    \
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NS + WIT - 1) : "memory");                             \
      } else if (piece == 2) {                                                                                    \
        if (wave < EXTRA) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NS + WIT) : "memory");                        \
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NS + WIT - 1) : "memory");                                 \
      } else if (piece + 1 < NPIECE) {                                                                            \
        if (wave < EXTRA) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WIT) : "memory");                             \
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WIT - 1) : "memory");                                      \
      } else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                     \
      /* ... for every wave, and everyone is done with the buffer piece + 2 goes into.  A bare barrier behind the wave's own LDS   */ \
      /* traffic (the `put`s of the previous layer): __syncthreads() compiles to s_waitcnt vmcnt(0) lgkmcnt(0) + s_barrier and   */ \
      /* retired the weight piece in flight at every piece (conv2d_m16.h)                                                          */ \
      if (!(ABL & 16) || j == 0) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");                                   \
      if (piece >= 1 && piece + 2 < NPIECE && !(ABL & 2)) issue_piece(piece + 2);   /* (pieces 0-2 went out with the gather) */ \
      const unsigned wrow = ws_lds + (unsigned)((piece % 3) * WP * 16);                                           \
      const unsigned xrow = (XLDS) + (unsigned)((2 * k + g) * PXW * 16);                                          \
      f16x8 xv[2][NS], wv[2][NS];                                                                                 \
      DB_READ(0, 0);                                                                                              \
      _Pragma("unroll") for (int t = 0; t < 9; ++t) {                                                             \
        const int fb = t & 1;                                                                                     \
        if (t + 1 < 9) { if (!(ABL & 8)) { DB_READ(fb ^ 1, t + 1); m16_wait<2 * NS, NS>(xv[fb], wv[fb]); } }      \
        else m16_wait<0, NS>(xv[fb], wv[fb]);                                                                     \
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
  bool overflow = false;
  // 4 consecutive couts of the lane's pixel -> the 8-byte half g of the units of octet `oct` in image X (XG octets per split)
  auto put = [&](f32x4* Xs, int XG, int oct, f32x4 v) {
    if (!pvalid) v = (f32x4)(0.f);
    u32x2 sp[NS];
    split4<NS, DT>(v, sp, overflow);
#pragma unroll
    for (int s = 0; s < NS; ++s) *reinterpret_cast<u32x2*>(reinterpret_cast<char*>(Xs + (s * XG + oct) * PXW + p) + 8 * g) = sp[s];
  };
  // the same 4 channels of the 64-channel image A as fp32 (hi + lo): the block's residual stream
  auto get_A = [&](int oct) -> f32x4 {
    float f[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      const u32x2 u = *reinterpret_cast<const u32x2*>(reinterpret_cast<const char*>(As + (s * 8 + oct) * PXW + p) + 8 * g);
      float a0, a1, a2, a3;
      m16_pair<DT>(u.x, a0, a1);
      m16_pair<DT>(u.y, a2, a3);
      f[0] += a0; f[1] += a1; f[2] += a2; f[3] += a3;
    }
    f32x4 r;
    r.x = f[0]; r.y = f[1]; r.z = f[2]; r.w = f[3];
    return r;
  };
  auto acc4 = [&](int h, int q, int boff, bool relu) -> f32x4 {
    const f32x4 b = *reinterpret_cast<const f32x4*>(bias_s + boff + 8 * q + 4 * g);
    f32x4 v;
    v.x = acc[h][4 * q + 0] + b.x; v.y = acc[h][4 * q + 1] + b.y; v.z = acc[h][4 * q + 2] + b.z; v.w = acc[h][4 * q + 3] + b.w;
    if (relu) { v.x = p16_vmax(v.x, 0.f); v.y = p16_vmax(v.y, 0.f); v.z = p16_vmax(v.z, 0.f); v.w = p16_vmax(v.w, 0.f); }
    return v;
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
#pragma unroll
  for (int q = 0; q < 4; ++q) put(Bs, 4, q, acc4(0, q, 0, false));
  // ---- lm.b: 32 -> 64, linear, + cur (read back from A) -> oth -> A (a lane rewrites exactly the bytes it has read) ----
  zero_acc();
  DB_LAYER(bs_lds, 4, true);
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int q = 0; q < 4; ++q) put(As, 8, h * 4 + q, acc4(h, q, 32 + h * 32, false) + get_A(h * 4 + q));
  // ---- m.a: 64 -> 32, ReLU --------------------------------------------------------------------------------------
  zero_acc();
  DB_LAYER(as_lds, 8, false);
#pragma unroll
  for (int q = 0; q < 4; ++q) put(Bs, 4, q, acc4(0, q, 96, true));
  // ---- m.b: 32 -> 64, ReLU, + oth (from A) -> cur (HBM, in place, whole 16-byte units per lane) --------------------
  zero_acc();
  DB_LAYER(bs_lds, 4, true);
#undef DB_LAYER
#undef DB_READ
  float vmax = 0.f;
  if constexpr (NS == 2) {
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 v = acc4(h, q, 128 + h * 32, true) + get_A(h * 4 + q);
        vmax = p16_vmax3_abs(p16_vmax3_abs(vmax, v.x, v.y), v.z, v.w);
        unsigned h0_, h1_, l0_, l1_;
        p16_split_hl(v, h0_, h1_, l0_, l1_);
        p16_swap32(h0_, l0_);          // g = 0: the hi unit of octet h*4 + q, g = 1: its lo unit
        p16_swap32(h1_, l1_);
        u32x4_t unit_;
        unit_.x = h0_; unit_.y = h1_; unit_.z = l0_; unit_.w = l1_;
        if (pvalid) db_store_unit(reinterpret_cast<u32x4_t*>(cur) + m16_plane(h * 4 + q, g, NS, H) + poff, unit_);
      }
    if (pvalid && vmax > 65504.f) overflow = true;
  } else {
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int qp = 0; qp < 2; ++qp) {
        const f32x4 v0 = acc4(h, 2 * qp, 128 + h * 32, true) + get_A(h * 4 + 2 * qp);
        const f32x4 v1 = acc4(h, 2 * qp + 1, 128 + h * 32, true) + get_A(h * 4 + 2 * qp + 1);
        if (DT == 1) vmax = p16_vmax3_abs(p16_vmax3_abs(p16_vmax3_abs(p16_vmax3_abs(vmax, v0.x, v0.y), v0.z, v0.w), v1.x, v1.y), v1.z, v1.w);
        unsigned a0_ = m16_pk2<DT>(v0.x, v0.y), a1_ = m16_pk2<DT>(v0.z, v0.w), b0_ = m16_pk2<DT>(v1.x, v1.y), b1_ = m16_pk2<DT>(v1.z, v1.w);
        p16_swap32(a0_, b0_);          // g = 0: the unit of octet h*4 + 2 qp, g = 1: of the next octet
        p16_swap32(a1_, b1_);
        u32x4_t unit_;
        unit_.x = a0_; unit_.y = a1_; unit_.z = b0_; unit_.w = b1_;
        if (pvalid) db_store_unit(reinterpret_cast<u32x4_t*>(cur) + m16_plane(h * 4 + 2 * qp + g, 0, NS, H) + poff, unit_);
      }
    if (DT == 1 && pvalid && vmax > 65504.f) overflow = true;
  }
  if (DT == 1 && overflow && a.flag) *a.flag = 1u;
}
