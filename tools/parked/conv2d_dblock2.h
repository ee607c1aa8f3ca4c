// conv2d_dblock2.h - the fused residual block of conv2d_dblock.h (one launch = oth = lm(cur) + cur; cur = m(oth) + oth, four 3x3 convs on the
// d*d independent sub-images of a map, in place) re-cut so that TWO workgroups share a CU in the default arithmetic:
//
//   * a workgroup takes PX = 128 pixels' worth of sub-images (dilation 32: two 8x8 sub-images, 64: eight 4x4 ones) with 4 waves instead of
//     256 pixels with 8: the operand images A (64 channels) and B (32) are 36.9 + 18.4 KB in f16x2;
//   * the weights stream in SUB-PIECES of one tap row (3 taps x 16 input channels x 32 couts = 6 KB in f16x2, a third of conv2d_dblock.h's
//     pieces) through a ring of four slots, issued three ahead: 24.6 KB instead of 55.3 KB.  A + B + ring + biases = 80 640 bytes <= 80 KB.
//     conv2d_dblock.h's workgroup (157 KB) has the CU to itself, so its gather, its sixteen pieces and its stores are serial phases and
//     every workgroup of a launch is in the same phase (HISTORY.md section 8: 12 us of matrix work inside 36 us); here one workgroup's
//     gather / epilogues / stores run under the other's matrix instructions;
//   * in the 32 -> 64 layers the sub-pieces go (K-chunk k, tap row r, cout half h) with h innermost: the X fragments of (k, r) - 3 taps x NS
//     reads, 24 registers - stay in registers for the second cout half (VERDICT r4 #1 ii): 0.67 instead of 1.33 ds_read_b128 per matrix
//     instruction on half of those layers' sub-pieces.
// PX = 256 (8 waves; dilation 16, whose sub-image IS 256 pixels) is the same kernel with one workgroup per CU (130 KB).
// Arithmetic, summation order per accumulator (K-chunk, tap row, tap) and epilogues are those of conv2d_dblock.h: the K-chunks of a cout half are
// accumulated in the same order, so the results are bit-identical to it.
#pragma once
#include "conv2d_dblock.h"

template <int NS, int DT, int PX, int ABL = 0>
__global__ __launch_bounds__(2 * PX, NS == 1 ? 4 : 2) void conv2d_dblock2_kernel(DBlockArgs a) {
  constexpr int WNS = DT == 1 ? 2 : 1;               // splits in the weight pack (the fp16 pack always carries hi and lo)
  constexpr int NT = 2 * PX, NW = NT / 64, PXW = PX + 16, LPX = PX == 128 ? 7 : 8;
  constexpr int AU = NS * 8 * PXW, BU = NS * 4 * PXW; // operand images (16-byte units)
  constexpr int SWP = NS * 3 * 2 * 32;               // one weight sub-piece: one tap row x 16 input channels x 32 couts
  constexpr int NSUB = 48, NSLOT = 4;
  constexpr int WIT = (SWP + NT - 1) / NT;
  constexpr int EXTRA = (SWP - (WIT - 1) * NT + 63) / 64;     // waves that issue WIT (the others WIT - 1) DMA instructions per sub-piece
  static_assert(PX == 128 || PX == 256, "workgroup = 128 or 256 pixels");
  static_assert(SWP % 64 == 0 && EXTRA <= NW, "sub-piece DMA geometry");
  __shared__ f32x4 smem[AU + BU + NSLOT * SWP + 48];
  f32x4* const As = smem;
  f32x4* const Bs = smem + AU;
  f32x4* const Ws = smem + AU + BU;
  float* const bias_s = reinterpret_cast<float*>(smem + AU + BU + NSLOT * SWP);   // [32 | 64 | 32 | 64]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, g = lane >> 5;
  const int d = a.dil, H = a.H, W = a.W;                // PX = 256: d = 16 / 32 / 64;  PX = 128: 32 / 64
  const int ld = 31 - __builtin_clz(d);                 // log2(d)
  const int lS = 8 - ld, S = 1 << lS;                   // sub-image side 16 / 8 / 4
  const int lG = LPX + 2 * ld - 16, G = 1 << lG;        // sub-images per workgroup (all in one row of sub-images: G <= d)
  f32x4* const cur = a.cur + (long)blockIdx.y * a.bs;

  // Workgroups go to the XCDs round-robin (b % 8); 8 consecutive pixels of a map row (one 128-byte line of a plane) belong to 8 consecutive
  // sub-image columns j0 = 8 / G workgroups: those are placed on ONE XCD, so that a line is fetched into one L2 only (conv2d_dblock.h).
  int bx = (int)blockIdx.x;
  if (G < 8 && !(ABL & 64)) {
    const int m = 8 >> lG, lm_ = 3 - lG;               // workgroups per line
    const int per_xcd = (65536 / PX) / 8;
    const int xcd = bx & 7, t = bx >> 3;               // t = 0 .. per_xcd - 1 on this XCD
    bx = ((xcd * (per_xcd >> lm_) + (t >> lm_)) << lm_) + (t & (m - 1));
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
  for (int i = tid; i < NS * 12 * 16; i += NT) {
    if (i < NS * 8 * 16) As[(i >> 4) * PXW + PX + (i & 15)] = (f32x4)(0.f);
    else Bs[((i - NS * 8 * 16) >> 4) * PXW + PX + (i & 15)] = (f32x4)(0.f);
  }
  __syncthreads();
  // ---- the four biases [32 | 64 | 32 | 64] by ONE DMA instruction of wave 0 (its oldest transfer: every counted wait below covers it) ----
  if (wave == 0 && lane < 48) {
    const float* src = lane < 8 ? a.bias[0] + 4 * lane : (lane < 24 ? a.bias[1] + 4 * (lane - 8) : (lane < 32 ? a.bias[2] + 4 * (lane - 24) : a.bias[3] + 4 * (lane - 32)));
    p16_glds16(reinterpret_cast<const f32x4*>(src), smem + AU + BU + NSLOT * SWP);
  }

  // ---- weight sub-pieces: i = layer * 12 + j;  64 -> 32 layers: j = k * 3 + r;  32 -> 64 layers: j = (k * 3 + r) * 2 + h ----------------
  auto issue_sub = [&](int i) {
    const int L = i / 12, j = i % 12;
    const bool wide = L & 1;
    const int k = wide ? j / 6 : j / 3, r = wide ? (j % 6) >> 1 : j % 3, h = wide ? (j & 1) : 0, cout = wide ? 64 : 32;
    f32x4* dst = Ws + (i % NSLOT) * SWP;
#pragma unroll
    for (int it = 0; it < WIT; ++it) {
      const int u = tid + it * NT;                   // unit (s, tr, g, co) of the slot
      if (u < SWP) {
        const int grp = u >> 5, co = u & 31;         // grp = (s*3 + tr)*2 + g
        const int s_ = grp / 6, tr = (grp % 6) >> 1, gg = grp & 1;
        p16_glds16(reinterpret_cast<const f32x4*>(a.w[L]) + ((long)k * (WNS * 9 * 2) + (s_ * 9 + 3 * r + tr) * 2 + gg) * cout + h * 32 + co,
                   dst + it * NT + wave * 64);
      }
    }
  };
  // ---- gather: the units of the workgroup's PX pixels, all 8 octets x NS planes, by LDS-DMA (a pixel outside the map is fetched from pad
  // pixel 255 of row 0: zero).  K-chunk 0 of the first layer, then the first three sub-pieces, then the rest: sub-pieces 0-2 wait for chunk 0 alone ----
  {
    int row, col, r_, c_;
    pix(tid & (PX - 1), row, col, r_, c_);
    const bool ok = row < H && col < W;
    const long off = ok ? (long)row * M16_PX + col : 255;
    const int par = wave / (PX / 64), wq = wave % (PX / 64);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
#pragma unroll
      for (int s_ = 0; s_ < NS; ++s_) {
        const int plane = s_ * 8 + 2 * k + par;                      // = s*8 + o
        if (!(ABL & 4)) p16_glds16(cur + m16_plane(plane & 7, s_, NS, H) + off, As + plane * PXW + wq * 64);
      }
      if (k == 0) { issue_sub(0); issue_sub(1); issue_sub(2); }
    }
  }

  // ---- this lane's pixel in the MFMA layout and its neighbour units -----------------------------------------------
  const int p = wave * 32 + l31;
  int prow, pcol, pr, pc;
  pix(p, prow, pcol, pr, pc);
  const bool pvalid = prow < H && pcol < W;
  const long poff = (long)prow * M16_PX + pcol;
  // byte offset of the tap's source unit within a plane; outside the sub-image: the zero unit in the bank slot of the unit the lane would have read
  unsigned nb16[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    const int dy = t / 3 - 1, dx = t % 3 - 1;
    const bool in = (unsigned)(pr + dy) < (unsigned)S && (unsigned)(pc + dx) < (unsigned)S;
    nb16[t] = (in ? (unsigned)(p + dy * S + dx) : (unsigned)PX + ((unsigned)(p + dy * S + dx) & 15u)) * 16u;
  }

  f32x16 acc[2];
  f16x8 xk[3][NS];                                  // X fragments of the current (K-chunk, tap row): kept across the two cout halves
  int sub = 0;
  const unsigned as_lds = p16_lds_addr(As), bs_lds = p16_lds_addr(Bs), ws_lds = p16_lds_addr(Ws + g * 32 + l31);
#define DB2_WAITV(n_)                                                                                            \
  {                                                                                                               \
    if (wave < EXTRA) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((n_) * WIT) : "memory");                           \
    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"((n_) * (WIT - 1)) : "memory");                                  \
  }
#define DB2_WAITV_G(n_)   /* + the three K-chunks of the gather issued behind sub-pieces 0-2 */                  \
  {                                                                                                               \
    if (wave < EXTRA) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((n_) * WIT + 3 * NS) : "memory");                  \
    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"((n_) * (WIT - 1) + 3 * NS) : "memory");                         \
  }
  // One layer = twelve sub-pieces.  WIDE: 32 -> 64, sub-pieces (k, r, h) with h innermost; else 64 -> 32, (k, r).
#define DB2_LAYER(XLDS, XG_, WIDE)                                                                                \
  {                                                                                                               \
    constexpr int XG = XG_;                                                                                       \
    _Pragma("unroll") for (int j = 0; j < 12; ++j, ++sub) {                                                       \
      constexpr bool wide_ = WIDE;                                                                                \
      const int k = wide_ ? j / 6 : j / 3;                                                                        \
      const int r = wide_ ? (j % 6) >> 1 : j % 3;                                                                 \
      const int h = wide_ ? (j & 1) : 0;                                                                          \
      /* this sub-piece's weights have landed (sub-pieces 0-2: and K-chunk 0 of the gather; from 3 on: the whole gather, issued in front of */ \
      /* sub-piece 3); the two sub-pieces issued behind it may still be in flight                                                         */ \
      if (sub < 3) { DB2_WAITV_G(2) }                                                                             \
      else if (sub + 2 < NSUB) { DB2_WAITV(2) }                                                                   \
      else if (sub + 1 < NSUB) { DB2_WAITV(1) }                                                                   \
      else { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }                                                   \
      /* ... for every wave, and everyone is done with the slot sub-piece sub + 3 goes into (that of sub - 1) */  \
      if (!(ABL & 16) || j == 0) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");                  \
      if (sub + 3 < NSUB && !(ABL & 2)) issue_sub(sub + 3);                                                       \
      const unsigned wrow = ws_lds + (unsigned)((sub % NSLOT) * SWP * 16);                                        \
      const unsigned xrow = (XLDS) + (unsigned)((2 * k + g) * PXW * 16);                                          \
      f16x8 wv[3][NS];                                                                                            \
      const bool newx = !wide_ || h == 0;                                                                         \
      if (!(ABL & 8)) {                                                                                           \
        _Pragma("unroll") for (int tr = 0; tr < 3; ++tr) {                                                        \
          _Pragma("unroll") for (int s = 0; s < NS; ++s) {                                                        \
            if (newx) xk[tr][s] = p16_lds_read16(xrow + nb16[3 * r + tr], s * XG * PXW * 16);                     \
            wv[tr][s] = p16_lds_read16(wrow, ((s * 3 + tr) * 2) * 32 * 16);                                       \
          }                                                                                                       \
        }                                                                                                         \
      }                                                                                                           \
      _Pragma("unroll") for (int tr = 0; tr < 3; ++tr) {                                                          \
        if (!(ABL & 8)) {                                                                                         \
          if (newx) { if (tr == 0) m16_wait<4 * NS, NS>(xk[0], wv[0]); else if (tr == 1) m16_wait<2 * NS, NS>(xk[1], wv[1]); else m16_wait<0, NS>(xk[2], wv[2]); } \
          else { if (tr == 0) m16_wait<2 * NS, NS>(xk[0], wv[0]); else if (tr == 1) m16_wait<NS, NS>(xk[1], wv[1]); else m16_wait<0, NS>(xk[2], wv[2]); } \
        }                                                                                                         \
        typedef typename Op16<DT>::vec V_;                                                                        \
        if constexpr ((ABL & 1) != 0) { asm volatile("" ::"v"(xk[tr][0]), "v"(wv[tr][0])); } else                 \
        if constexpr (NS == 2) {                                                                                  \
          acc[h] = Op16<DT>::mfma(__builtin_bit_cast(V_, wv[tr][0]), __builtin_bit_cast(V_, xk[tr][NS - 1]), acc[h]); \
          acc[h] = Op16<DT>::mfma(__builtin_bit_cast(V_, wv[tr][NS - 1]), __builtin_bit_cast(V_, xk[tr][0]), acc[h]); \
        }                                                                                                         \
        if constexpr ((ABL & 1) == 0) acc[h] = Op16<DT>::mfma(__builtin_bit_cast(V_, wv[tr][0]), __builtin_bit_cast(V_, xk[tr][0]), acc[h]); \
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
    if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
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
  DB2_LAYER(as_lds, 8, false);
#pragma unroll
  for (int q = 0; q < 4; ++q) put(Bs, 4, q, acc4(0, q, 0, false));
  // ---- lm.b: 32 -> 64, linear, + cur (read back from A) -> oth -> A (a lane rewrites exactly the bytes it has read) ----
  zero_acc();
  DB2_LAYER(bs_lds, 4, true);
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int q = 0; q < 4; ++q) put(As, 8, h * 4 + q, acc4(h, q, 32 + h * 32, false) + get_A(h * 4 + q));
  // ---- m.a: 64 -> 32, ReLU --------------------------------------------------------------------------------------
  zero_acc();
  DB2_LAYER(as_lds, 8, false);
#pragma unroll
  for (int q = 0; q < 4; ++q) put(Bs, 4, q, acc4(0, q, 96, true));
  // ---- m.b: 32 -> 64, ReLU, + oth (from A) -> cur (HBM, in place, whole 16-byte units per lane) --------------------
  zero_acc();
  DB2_LAYER(bs_lds, 4, true);
#undef DB2_LAYER
#undef DB2_WAITV
#undef DB2_WAITV_G
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
