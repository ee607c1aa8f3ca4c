// conv2d_m16q.h - the Decoders' dilated 3x3 Conv2d on M16 maps (dilations 1-8), round 4 form: tiles of FOUR output rows
// (y, y + d, y + 2d, y + 3d) x 128 pixels, operand fragments shared inside the wave.
//
// Why (conv2d_m16.h's one-row kernel, measured in rounds 2-3: 11.8 / 13.3 us per 250 x 250 map against 2.9 us of matrix work):
// a launch is bound by the fill of LDS - every piece (16-channel K-chunk, 32-cout half) of a one-row workgroup pulls 70 KB
// (three source rows + an 18 KB weight piece) for 1 728 matrix-pipe cycles - and by the LDS reads of its 32 x 32 wave tile
// (4 ds_read_b128 per 3 MFMAs).  A chain of four output rows spaced by the dilation reads SIX source rows (1.5 per output row
// instead of 3) and ONE weight piece for four rows: 73.7 KB per piece for 3 456 matrix-pipe cycles - 2.1x fewer L2 -> LDS bytes
// per MFMA, so the transfer issued at the top of a piece lands under that piece's MFMAs.  Half-width tiles (128 + 2 x 8 halo
// pixels) keep the X image of a chunk at 55.3 KB: two of them and two weight pieces are 147.5 KB of LDS.
//   * wave tile = 2 output rows x 32 pixels x 32 couts: per kernel column kx the wave reads FOUR X fragments (source rows
//     r .. r+3 of its row pair) and three W fragments for 2 x 3 taps - 14 ds_read_b128 per 18 MFMAs (0.78 per MFMA; 1.33 in the
//     one-row kernel): the W fragment of a tap serves both rows, the X fragment of source row r+1 (r+2) serves tap ky = 1 (2) of
//     the first row and ky = 0 (1) of the second.
//   * a 250 x 250 map is 63-64 row groups x 2 halves = 126-128 workgroups: both strands of a level are ONE round on 256 CUs
//     (the one-row kernel: 500 workgroups of 160 KB, two rounds, each with its own cold prologue).
//   * out-of-map source rows, halo pixels left of pixel 0 / right of the row pitch come from a zero unit by per-lane address
//     select in the DMA: no margin is zeroed, no tap is skipped, every group runs the same instruction stream.
//   * batches of more than one round (the SV screen's four strands, config 3's eight): the grid holds ONE round and a workgroup walks
//     the maps b, b + gridDim.y, ... of its tile - the weights' pieces and the X geometry are the same for each - requesting the next
//     map's first weight piece and X image under the LAST piece of the current one (ring slot and buffer 0 are free then: the piece
//     count and the chunk count of such a launch are even), so only a workgroup's first map pays the cold first fill.
// Every layer shape of a Decoder goes through it (16- to 144-channel inputs, 32 / 64 couts, the table-fed first conv).
#pragma once
#include <type_traits>
#include "conv2d_m16.h"

struct ConvM16QArgs {
  ConvM16Args c;
  const f32x4* zero;   // one 16-byte unit of zeros in global memory
  int ngroups;         // row groups per map = ceil(H / 4d) * d (group p = (q, r) = (p / d, p % d) holds rows 4dq + r + {0, d, 2d, 3d})
  int nb;              // maps of the batch: the workgroup at blockIdx.y takes maps blockIdx.y, blockIdx.y + gridDim.y, ... (gridDim.y < nb only with an even nchunks)
};

// fp16 pair (hi part, lo part) of a stored value -> fp32, exactly: hi + lo has at most 22 significant bits (one v_fma_mix_f32 per value)
__device__ __forceinline__ float m16_hl_lo(unsigned h, unsigned l) {
  float r;
  asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel_hi:[1,0,1]" : "=v"(r) : "v"(h), "v"(l));
  return r;
}
__device__ __forceinline__ float m16_hl_hi(unsigned h, unsigned l) {
  float r;
  asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel:[1,0,1] op_sel_hi:[1,0,1]" : "=v"(r) : "v"(h), "v"(l));
  return r;
}


// epilogue of one 32-pixel x COUT accumulator tile (the accumulators started from the bias): separable-part tables, ReLU, residual, back to
// M16 units (the P16 / B16 recipe: after v_permlane32_swap every lane holds one whole 16-byte unit; 512 contiguous bytes per half wave)
// (HB, NH: the cout halves [HB, HB + NH) of the layer are in acc[0 .. NH) / ru[0 .. 4 NH) - the whole layer by default, one half for the
// half-by-half kernel below)
template <int COUT, int NS, int DT, int HB = 0, int NH = COUT / 32>
__device__ __forceinline__ void m16_tile_epilogue(const ConvM16Args& a, f32x16 (&acc)[NH], int b, int yr, int px, int g,
                                                  const u32x4_t* ru, bool res, float& vmax) {
  const int H = a.H, W = a.W;
  const bool pxok = px < W;
  const long rowoff = (long)yr * M16_PX + px;
  f32x4* const yb = a.y + (long)b * a.y_bs;
  const f32x4* const rb = res ? a.r + (long)b * a.r_bs : nullptr;
  // separable-part tables: row term [column class of px][yr][co], column term [row class of yr][px][co]
  const float* tabr = nullptr;
  const float* tabc = nullptr;
  if (a.tab && pxok) {
    const float* tb = a.tab + (long)b * a.tab_bs;
    const int xc = px == 0 ? 0 : (px == W - 1 ? 2 : 1), yc = yr == 0 ? 0 : (yr == H - 1 ? 2 : 1);
    tabr = tb + ((long)xc * H + yr) * 64;
    tabc = tb + ((long)(3 + yc) * H + px) * 64;
  }
  if constexpr (NS == 2) {
#pragma unroll
    for (int h = 0; h < NH; ++h)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int o = (HB + h) * 4 + q;
        f32x4 v;
        v.x = acc[h][4 * q + 0]; v.y = acc[h][4 * q + 1]; v.z = acc[h][4 * q + 2]; v.w = acc[h][4 * q + 3];
        if (tabr) {
          const f32x4 tr = *reinterpret_cast<const f32x4*>(tabr + (HB + h) * 32 + 8 * q + 4 * g), tc = *reinterpret_cast<const f32x4*>(tabc + (HB + h) * 32 + 8 * q + 4 * g);
          v.x += tr.x + tc.x; v.y += tr.y + tc.y; v.z += tr.z + tc.z; v.w += tr.w + tc.w;
        }
        if (a.relu) { v.x = p16_vmax(v.x, 0.f); v.y = p16_vmax(v.y, 0.f); v.z = p16_vmax(v.z, 0.f); v.w = p16_vmax(v.w, 0.f); }
        if (rb) {
          const u32x4_t u_ = ru[h * 4 + q];                                                            // g = 0: the hi unit, g = 1: the lo unit
          unsigned ux_ = u_.x, uy_ = u_.y, uz_ = u_.z, uw_ = u_.w;
          p16_swap32(ux_, uz_);
          p16_swap32(uy_, uw_);
          v.x += m16_hl_lo(ux_, uz_); v.y += m16_hl_hi(ux_, uz_);     // ux = hi pair 0, uz = lo pair 0, uy / uw = pair 1 of the lane's 4 couts
          v.z += m16_hl_lo(uy_, uw_); v.w += m16_hl_hi(uy_, uw_);
        }
        if (!pxok) v = (f32x4)(0.f);
        vmax = p16_vmax3_abs(p16_vmax3_abs(vmax, v.x, v.y), v.z, v.w);
        unsigned h0_, h1_, l0_, l1_;
        p16_split_hl(v, h0_, h1_, l0_, l1_);
        p16_swap32(h0_, l0_);
        p16_swap32(h1_, l1_);
        u32x4_t unit_;
        unit_.x = h0_; unit_.y = h1_; unit_.z = l0_; unit_.w = l1_;
        m16_store_unit(reinterpret_cast<u32x4_t*>(yb) + m16_plane(o, g, NS, H) + rowoff, unit_);
      }
  } else {
#pragma unroll
    for (int h = 0; h < NH; ++h)
#pragma unroll
      for (int qp = 0; qp < 2; ++qp) {
        const int o = (HB + h) * 4 + 2 * qp;          // octets o (q0 = 2 qp) and o + 1
        f32x4 v0, v1;
        v0.x = acc[h][8 * qp + 0]; v0.y = acc[h][8 * qp + 1]; v0.z = acc[h][8 * qp + 2]; v0.w = acc[h][8 * qp + 3];
        v1.x = acc[h][8 * qp + 4]; v1.y = acc[h][8 * qp + 5]; v1.z = acc[h][8 * qp + 6]; v1.w = acc[h][8 * qp + 7];
        if (tabr) {
          const int co_ = (HB + h) * 32 + 16 * qp + 4 * g;
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
  }
}

// counted LDS wait that pins the fragments consumed next: W fragment `w` and (XR = 1) the four X fragments of a kernel column
template <int N, int NS>
__device__ __forceinline__ void m16q_wait_w(f16x8 (&w)[NS]) {
  if constexpr (NS == 2) asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(w[0]), "+v"(w[1]) : "n"(N));
  else asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(w[0]) : "n"(N));
}
template <int N, int NS>
__device__ __forceinline__ void m16q_wait_xw(f16x8 (&x)[4][NS], f16x8 (&w)[NS]) {
  if constexpr (NS == 2)
    asm volatile("s_waitcnt lgkmcnt(%10)"
                 : "+v"(x[0][0]), "+v"(x[0][1]), "+v"(x[1][0]), "+v"(x[1][1]), "+v"(x[2][0]), "+v"(x[2][1]), "+v"(x[3][0]), "+v"(x[3][1]), "+v"(w[0]), "+v"(w[1])
                 : "n"(N));
  else asm volatile("s_waitcnt lgkmcnt(%5)" : "+v"(x[0][0]), "+v"(x[1][0]), "+v"(x[2][0]), "+v"(x[3][0]), "+v"(w[0]) : "n"(N));
}
template <int N>
__device__ __forceinline__ void m16q_vmwait() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

#ifndef M16Q_SPREAD
#define M16Q_SPREAD 1
#endif
#ifndef M16Q_ABL
#define M16Q_ABL 0   // timing-only ablations (tools/microbench_m16q.hip): 1 no MFMA, 2 no fragment reads, 4 no epilogue, 8 / 16 X / W DMA of the first piece only, 32 no wait for a piece's first fragments
#endif
#ifdef M16Q_STAMPS   // tools/microbench_m16q.hip: s_memtime stamps of every wave of ONE workgroup [wave][16]
__device__ unsigned long long m16q_stamp_buf[8 * 16];
#define M16Q_STAMP(n_) { if (blockIdx.x == M16Q_STAMPS && blockIdx.y == 0 && lane == 0) { m16q_stamp_buf[wave * 16 + (n_)] = __builtin_readcyclecounter(); if ((n_) == 0 || (n_) == 11) m16q_stamp_buf[wave * 16 + 12 + ((n_) != 0)] = __builtin_amdgcn_s_memrealtime(); } }
#else
#define M16Q_STAMP(n_)
#endif
template <int COUT, int NS, int DT>
__global__ __launch_bounds__(512, 1) void conv2d_3x3_m16q_kernel(ConvM16QArgs aq) {
  static_assert(NS == 1 || DT == 1, "f16x2, plain fp16 or plain bf16");
  const ConvM16Args& a = aq.c;
  constexpr int WNS = DT == 1 ? 2 : 1;              // splits in the weight pack (the fp16 pack always carries hi and lo)
  constexpr int NT = 512, NH = COUT / 32, TW = 128, ROWP = 8 + TW + 8, SR = 6;
  constexpr int XROWS = NS * 2 * SR;                // rows of an X image: [s][g][6 source rows]
  constexpr int XB = XROWS * ROWP;                  // units per X buffer (3 456 / 1 728)
  constexpr int WP = NS * 9 * 2 * 32;               // units per weight piece (1 152 / 576)
  constexpr int XIT = (XB + NT - 1) / NT;           // X DMA rounds per chunk; the last one is issued by the first XFULL waves only
  constexpr int XFULL = (XB - (XIT - 1) * NT) / 64;
  constexpr int WIT = (WP + NT - 1) / NT;
  constexpr int WFULL = (WP - (WIT - 1) * NT) / 64;
  static_assert(XB % 64 == 0 && WP % 64 == 0, "whole-wave DMA rounds");
  __shared__ f32x4 smem[2 * XB + 2 * WP];
  f32x4* const Xs = smem;
  f32x4* const Ws = smem + 2 * XB;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, g = lane >> 5;
  const int H = a.H, d = a.dil;
  int b = blockIdx.y;
  const int nb = aq.nb, bstep = gridDim.y;
  // Workgroups go to the XCDs round-robin (blockIdx.x % 8; the grid's x extent is a multiple of 8).  Tiles are ordered residue class r,
  // chain position q, half, and XCD j takes a CONTIGUOUS run of them: neighbouring groups of a chain share two of their six source rows,
  // the halves of a group 16 pixels of each - those are then fetched into ONE L2 (d = 8: XCD r holds class r; d = 1: XCD j rows 32j ..)
  const int per_xcd = (int)gridDim.x >> 3, tix = ((int)blockIdx.x & 7) * per_xcd + ((int)blockIdx.x >> 3);
  if (tix >= aq.ngroups * 2) return;
  const int nq = aq.ngroups / d, cls = (tix >> 1) / nq, q4 = (tix >> 1) - cls * nq, px0 = (tix & 1) * TW;
  const int y0 = q4 * 4 * d + cls;                  // first output row of the group
  if (y0 >= H) return;
  const int r0 = (wave >> 2) * 2;                   // the wave's row pair inside the group: output rows y0 + (r0 + {0, 1}) d
  const int wpx = (wave & 3) * 32;                  // its 32 pixels inside the tile
  M16Q_STAMP(0);

  // X DMA geometry of this thread: unit u = it * NT + tid of an X image = (row = (s*2 + gg)*6 + rr, col); global unit offset relative
  // to the chunk's first plane, or -1 = the zero unit (source row outside the map, halo pixel outside the row)
  int xoff[XIT];
#pragma unroll
  for (int it = 0; it < XIT; ++it) {
    const int u = it * NT + tid;
    const int uu = u < XB ? u : 0;
    const int row = uu / ROWP, col = uu - row * ROWP;
    const int s = row / (2 * SR), gg = (row / SR) & 1, rr = row % SR;
    const int ys = y0 + (rr - 1) * d, px = px0 - 8 + col;
    const bool ok = ys >= 0 && ys < H && px >= 0 && px < M16_PX;
    xoff[it] = ok ? (int)(m16_plane(gg, s, NS, H) + (long)ys * M16_PX + px) : -1;
  }
  const f32x4* xb = a.x + (long)b * a.x_bs;
  auto issue_x1 = [&](const f32x4* xm, int k, int buf, int it) {      // one of the XIT transfers of chunk k's X image of the map at xm
    if (it + 1 < XIT || wave < XFULL) p16_glds16(xoff[it] >= 0 ? xm + m16_plane(2 * k, 0, NS, H) + xoff[it] : aq.zero, Xs + buf * XB + it * NT + wave * 64);
  };
  auto issue_w1 = [&](int i, int slot, int it) {     // one of the WIT transfers of weight piece i = (k, h) -> ring slot `slot`
    const int k = i / NH, h = i - k * NH;
    const int u = tid + it * NT;
    if (it + 1 < WIT || wave < WFULL)
      p16_glds16(reinterpret_cast<const f32x4*>(a.w) + ((long)k * (WNS * 9 * 2) + (u >> 5)) * COUT + h * 32 + (u & 31), Ws + slot * WP + it * NT + wave * 64);
  };
  auto issue_x = [&](const f32x4* xm, int k, int buf) {
#pragma unroll
    for (int it = 0; it < XIT; ++it) issue_x1(xm, k, buf, it);
  };
  auto issue_w = [&](int i, int slot) {
#pragma unroll
    for (int it = 0; it < WIT; ++it) issue_w1(i, slot, it);
  };

  issue_w(0, 0);
  issue_x(xb, 0, 0);
  M16Q_STAMP(1);
  const unsigned ws_lds = p16_lds_addr(Ws + g * 32 + l31);
  const unsigned xs_lds = p16_lds_addr(Xs + (g * SR + r0) * ROWP + 8 + wpx + l31);
  const int px = px0 + wpx + l31;
  const int yr0 = y0 + r0 * d, yr1 = yr0 + d;
  float vmax = 0.f;

  while (true) {       // the maps of this workgroup: b, b + bstep, ...
  const bool has_next = b + bstep < nb;
  const f32x4* const xb_next = xb + (long)bstep * a.x_bs;

  // accumulators [row of the pair][cout half] start from the bias: register group q of a lane = couts 8q + 4g .. + 3 of the half
  // (the loads' round trip runs under the first piece's transfer)
  f32x16 acc[2][NH];
  int gb_ = g;                     // (opaque per map: the bias loads are invariant over the map loop - hoisted, their 16 / 32 registers would stay live)
  asm volatile("" : "+v"(gb_));
#pragma unroll
  for (int h = 0; h < NH; ++h)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 b4 = *reinterpret_cast<const f32x4*>(a.bias + h * 32 + 8 * q + 4 * gb_);
#pragma unroll
      for (int j = 0; j < 2; ++j) { acc[j][h][4 * q + 0] = b4.x; acc[j][h][4 * q + 1] = b4.y; acc[j][h][4 * q + 2] = b4.z; acc[j][h][4 * q + 3] = b4.w; }
    }

  // residual units (f16x2) of the pair's FIRST row are requested when the last piece starts (their round trip runs under its MFMAs), the
  // second row's under the last kernel column of that piece (the other X fragment buffer is dead by then).  (Measured alternative for the
  // layers without ReLU - accumulators started from the residual, 128 KB per workgroup requested with the first piece: the first barrier
  // moves from 5 400 to 15 400 cycles, the launch from 22.3 to 23.4 us: the residual costs transfer time, and the start is where the
  // matrix pipe has nothing to cover it.)
  constexpr int NRU = NS == 2 ? NH * 4 : 1;
  u32x4_t ru[NRU], ru1[NRU];
  const f32x4* const rb = a.r ? a.r + (long)b * a.r_bs : nullptr;
  const bool res_late = rb != nullptr;

#define M16Q_READ_X(dst_, kx_)                                            \
  if constexpr (!(M16Q_ABL & 2)) _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_)                         \
      _Pragma("unroll") for (int s_ = 0; s_ < NS; ++s_) dst_[j_][s_] = p16_lds_read16(xcol[kx_], ((s_ * 2 * SR) + j_) * ROWP * 16);
#define M16Q_READ_W(dst_, t_) if constexpr (!(M16Q_ABL & 2)) _Pragma("unroll") for (int s_ = 0; s_ < NS; ++s_) dst_[s_] = p16_lds_read16(wrow, ((s_ * 9 + (t_)) * 2) * 32 * 16);
#define M16Q_MFMA(w_, x_, j_, h_)                                                                                       \
  if constexpr (!(M16Q_ABL & 1)) {                                                                                      \
    typedef typename Op16<DT>::vec V_;                                                                                  \
    if constexpr (NS == 2) {                                                                                            \
      acc[j_][h_] = Op16<DT>::mfma(__builtin_bit_cast(V_, w_[0]), __builtin_bit_cast(V_, x_[NS - 1]), acc[j_][h_]);     \
      acc[j_][h_] = Op16<DT>::mfma(__builtin_bit_cast(V_, w_[NS - 1]), __builtin_bit_cast(V_, x_[0]), acc[j_][h_]);     \
    }                                                                                                                   \
    acc[j_][h_] = Op16<DT>::mfma(__builtin_bit_cast(V_, w_[0]), __builtin_bit_cast(V_, x_[0]), acc[j_][h_]);           \
  }

  // one piece = (K-chunk k, cout half H_): LASTP = the layer's last piece (peeled in the source: the residual units requested under it must
  // not be live across the piece loop)
  auto piece = [&](const int k, auto hc, auto lastc) __attribute__((always_inline)) {
    constexpr int h = decltype(hc)::value;
    constexpr bool LASTP = decltype(lastc)::value;
    const int i = k * NH + h;
    // this piece's weights and X image have landed; with 64 couts the next chunk's X image (issued under the first half, after
    // this half's weights) may stay in flight through the second half
    if (!LASTP && NH == 2 && h == 1) {
      if (wave < XFULL) m16q_vmwait<XIT>();
      else m16q_vmwait<XIT - 1>();
    } else m16q_vmwait<0>();
    M16Q_STAMP(2 + 2 * (i < 4 ? i : 3));
    M16_BARRIER();                   // ... for every wave, and everyone is done with the buffers the next transfers go into
    M16Q_STAMP(3 + 2 * (i < 4 ? i : 3));
    if constexpr (NS == 2 && LASTP) {
      if (res_late && yr0 < H) {
        // (the lane's coordinates from opaque copies, here and in the epilogue: per-lane addresses that are invariant over the map loop
        // would be hoisted out of it and sit in ~60 registers through every piece - the kernel then spills)
        int px_ = px, g_ = g;
        asm volatile("" : "+v"(px_), "+v"(g_));
#pragma unroll
        for (int j = 0; j < NRU; ++j) ru[j] = reinterpret_cast<const u32x4_t*>(rb)[m16_plane(j, g_, NS, H) + (long)yr0 * M16_PX + px_];
      } else {      // defined on every path: a value left from the workgroup's previous map would otherwise be live through the whole map loop
#pragma unroll
        for (int j = 0; j < NRU; ++j) ru[j] = (u32x4_t)(0u);
      }
    }
    // the next piece's weights, then (first half of a chunk) the next chunk's X image: in a block here, or (M16Q_SPREAD) one transfer per tap
    // between the MFMA groups - a block of 10 transfers per wave stalls the issuing waves until the load path has taken them (~1 400 cycles
    // per piece with the matrix pipe idle); spread, each costs its issue slot under the other wave's MFMAs
    // (the workgroup's NEXT map: its weight piece 0 is requested under the LAST piece (slot 0), its first X image (buffer 0) under the last
    // piece too with 32 couts, under the first half of the last chunk with 64 - that half has no X transfer of its own, and the last piece
    // of a 64-cout layer holds the residual units: the X offsets must not be live beside them)
    const bool lastk = k + 1 == a.nchunks;
    const bool dow = (LASTP ? has_next : true) && !(M16Q_ABL & 16);
    const bool dox = (LASTP ? (NH == 1 && has_next) : (h == 0 && (lastk ? (NH == 2 && has_next) : true))) && !(M16Q_ABL & 8);
    const int i_nx = LASTP ? 0 : i + 1, k_nx = (LASTP || lastk) ? 0 : k + 1;
    const f32x4* const x_nx = (LASTP || lastk) ? xb_next : xb;
    static_assert(WIT + XIT <= 10, "transfer slots of a piece");
#define M16Q_DMA_SLOT(n_)                                                                              \
  {                                                                                                    \
    const int first_ = (n_) == 0 ? 0 : (n_) + 1, cnt_ = (n_) == 0 ? 2 : 1;                              \
    _Pragma("unroll") for (int q_ = first_; q_ < first_ + cnt_; ++q_) {                                 \
      if (q_ < WIT) { if (dow) issue_w1(i_nx, (i + 1) & 1, q_); }                                       \
      else if (q_ - WIT < XIT) { if (dox) issue_x1(x_nx, k_nx, (k + 1) & 1, q_ - WIT); }                \
    }                                                                                                  \
  }
    if (!M16Q_SPREAD) {
      if (dow) issue_w(i_nx, (i + 1) & 1);
      if (dox) issue_x(x_nx, k_nx, (k + 1) & 1);
    }
    const unsigned wrow = ws_lds + (unsigned)((i & 1) * WP * 16);
    unsigned xcol[3];
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) xcol[kx] = xs_lds + (unsigned)(((k & 1) * XB + (kx - 1) * d) * 16);
    f16x8 xr[2][4][NS], wv[2][NS];
    if constexpr (M16Q_ABL & 2) { _Pragma("unroll") for (int q_ = 0; q_ < 2; ++q_) _Pragma("unroll") for (int s_ = 0; s_ < NS; ++s_) { wv[q_][s_] = (f16x8)(0); _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_) xr[q_][j_][s_] = (f16x8)(0); } }
    M16Q_READ_X(xr[0], 0);
    M16Q_READ_W(wv[0], 0);
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) {
        const int n = kx * 3 + ky, fb = n & 1, xq = kx & 1;
        if (M16Q_SPREAD) M16Q_DMA_SLOT(n);
        if constexpr (NS == 2 && LASTP) {       // the second row's residual units: from here on the other X fragment buffer is dead
          if (kx == 2 && ky == 0 && res_late && yr1 < H) {
            int px_ = px, g_ = g;
            asm volatile("" : "+v"(px_), "+v"(g_));
#pragma unroll
            for (int j = 0; j < NRU; ++j) ru1[j] = reinterpret_cast<const u32x4_t*>(rb)[m16_plane(j, g_, NS, H) + (long)yr1 * M16_PX + px_];
          } else if (kx == 2 && ky == 0) {
#pragma unroll
            for (int j = 0; j < NRU; ++j) ru1[j] = (u32x4_t)(0u);
          }
        }
        if (n < 8) { const int n1 = n + 1, t1 = (n1 % 3) * 3 + n1 / 3; M16Q_READ_W(wv[fb ^ 1], t1); }
        if (ky == 0 && kx < 2) M16Q_READ_X(xr[xq ^ 1], kx + 1);
        if (ky == 0) {
          if (kx == 0 && (M16Q_ABL & 32)) m16q_wait_xw<15, NS>(xr[xq], wv[fb]);      // timing only: the piece's first fragments as if prefetched
          else if (kx < 2) m16q_wait_xw<5 * NS, NS>(xr[xq], wv[fb]);
          else m16q_wait_xw<NS, NS>(xr[xq], wv[fb]);
        } else {
          if (n < 8) m16q_wait_w<NS, NS>(wv[fb]);
          else m16q_wait_w<0, NS>(wv[fb]);
        }
        M16Q_MFMA(wv[fb], xr[xq][ky], 0, h);
        M16Q_MFMA(wv[fb], xr[xq][ky + 1], 1, h);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  };
  typedef std::integral_constant<int, 0> H0_;
  typedef std::integral_constant<int, 1> H1_;
  for (int k = 0; k + 1 < a.nchunks; ++k) {
    piece(k, H0_(), std::false_type());
    if constexpr (NH == 2) piece(k, H1_(), std::false_type());
  }
  if constexpr (NH == 2) {
    piece(a.nchunks - 1, H0_(), std::false_type());
    piece(a.nchunks - 1, H1_(), std::true_type());
  } else piece(a.nchunks - 1, H0_(), std::true_type());

  // ---- epilogue: the pair's two rows
  M16Q_STAMP(10);
  if constexpr (M16Q_ABL & 4) { if (acc[0][0][0] != 0.12345f) return; }
  {
    int px_ = px, g_ = g;
    asm volatile("" : "+v"(px_), "+v"(g_));
    if (yr0 < H) m16_tile_epilogue<COUT, NS, DT>(a, acc[0], b, yr0, px_, g_, ru, res_late, vmax);
    if (yr1 < H) m16_tile_epilogue<COUT, NS, DT>(a, acc[1], b, yr1, px_, g_, ru1, res_late, vmax);
  }
  M16Q_STAMP(11);
  if (!has_next) break;
  b += bstep;
  xb = xb_next;
  }   // maps
#undef M16Q_READ_X
#undef M16Q_READ_W
#undef M16Q_MFMA
#undef M16Q_DMA_SLOT
  if (DT == 1 && vmax > 65504.f && a.flag) *a.flag = 1u;
}
