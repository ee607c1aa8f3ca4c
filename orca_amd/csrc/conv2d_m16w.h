// conv2d_m16w.h - the Decoders' dilated 3x3 Conv2d on M16 maps, second form: PERSISTENT workgroups, the layer's weights
// RESIDENT in LDS, tiles of TWO output rows (y, y + d) x 128 pixels.
//
// Why (conv2d_m16.h's one-row kernel, measured in rounds 2-3): a launch is bound by the fill of LDS - every piece (K-chunk, 32-cout
// half) pulls 70 KB through the CU's 64 B/clk load path (52 KB of X rows + an 18 KB weight piece) for 1 728 matrix-pipe cycles, with
// exactly one chunk of lookahead in 160 KB of LDS, and every workgroup pays a ~4 300-cycle cold prologue for one row of output.
//   * an output row pair (y, y + d) reads the four source rows y - d, y, y + d, y + 2d: 2 rows per output row instead of 3;
//   * half-width tiles (128 + 2 x 8 halo pixels) keep the X image of a chunk at 36.9 KB, so its double buffer (73.7 KB) fits NEXT TO
//     all weight pieces of a 64 -> 32 or 32 -> 64 layer (4 x 18.4 KB = 73.7 KB): the weights are fetched ONCE per workgroup, piece by
//     piece alongside the first tile's chunks, and every later tile of the persistent workgroup fills 36.9 KB per piece - 47 % less;
//   * the chunk stream runs ACROSS tiles: the next tile's first X image is in flight under the current tile's last piece and
//     epilogue, so only a workgroup's first tile pays the prologue.
// A launch covers the whole batch (both strands: ~500 tiles on 256 workgroups); out-of-map source rows / halo pixels are fetched from
// a zero unit (per-lane address select), so no tap is ever skipped and no margin has to be zeroed.  Layers whose pieces do not fit
// (64 -> 64, the 80- and 16-channel inputs: 7 of a Decoder's 79 per-layer convs) and the table-fed first conv stay on the one-row kernel.
#pragma once
#include "conv2d_m16.h"

struct ConvM16WArgs {
  ConvM16Args c;
  const f32x4* zero;   // one 16-byte unit of zeros in global memory
  int B;               // maps in the launch
  int npairs;          // row pairs per map = ceil(H / 2d) * d (pairs whose first row is >= H are empty)
};

template <int COUT, int NS, int DT>
__device__ __forceinline__ void m16w_epilogue(const ConvM16Args& a, f32x16 (&acc)[COUT / 32], const float* bias_s, int b, int yr, int px, int g,
                                              const u32x4_t* ru, float& vmax) {
  constexpr int NH = COUT / 32;
  const int H = a.H;
  const bool pxok = px < a.W;
  const long rowoff = (long)yr * M16_PX + px;
  f32x4* const yb = a.y + (long)b * a.y_bs;
  const f32x4* const rb = a.r ? a.r + (long)b * a.r_bs : nullptr;
  if constexpr (NS == 2) {
#pragma unroll
    for (int h = 0; h < NH; ++h)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int o = h * 4 + q;
        const f32x4 bs4 = *reinterpret_cast<const f32x4*>(bias_s + h * 32 + 8 * q + 4 * g);
        f32x4 v;
        v.x = acc[h][4 * q + 0] + bs4.x; v.y = acc[h][4 * q + 1] + bs4.y; v.z = acc[h][4 * q + 2] + bs4.z; v.w = acc[h][4 * q + 3] + bs4.w;
        if (a.relu) { v.x = p16_vmax(v.x, 0.f); v.y = p16_vmax(v.y, 0.f); v.z = p16_vmax(v.z, 0.f); v.w = p16_vmax(v.w, 0.f); }
        if (rb) {
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
        reinterpret_cast<u32x4_t*>(yb)[m16_plane(o, g, NS, H) + rowoff] = unit_;
      }
  } else {
#pragma unroll
    for (int h = 0; h < NH; ++h)
#pragma unroll
      for (int qp = 0; qp < 2; ++qp) {
        const int o = h * 4 + 2 * qp;                 // octets o (q0 = 2 qp) and o + 1
        const f32x4 b0 = *reinterpret_cast<const f32x4*>(bias_s + h * 32 + 16 * qp + 4 * g);
        const f32x4 b1 = *reinterpret_cast<const f32x4*>(bias_s + h * 32 + 16 * qp + 8 + 4 * g);
        f32x4 v0, v1;
        v0.x = acc[h][8 * qp + 0] + b0.x; v0.y = acc[h][8 * qp + 1] + b0.y; v0.z = acc[h][8 * qp + 2] + b0.z; v0.w = acc[h][8 * qp + 3] + b0.w;
        v1.x = acc[h][8 * qp + 4] + b1.x; v1.y = acc[h][8 * qp + 5] + b1.y; v1.z = acc[h][8 * qp + 6] + b1.z; v1.w = acc[h][8 * qp + 7] + b1.w;
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
        reinterpret_cast<u32x4_t*>(yb)[m16_plane(o + g, 0, NS, H) + rowoff] = unit_;
      }
  }
}

template <int COUT, int NS, int DT>
__global__ __launch_bounds__(512, 1) void conv2d_3x3_m16w_kernel(ConvM16WArgs aw) {
  static_assert(NS == 1 || DT == 1, "f16x2, plain fp16 or plain bf16");
  const ConvM16Args& a = aw.c;
  constexpr int WNS = DT == 1 ? 2 : 1;              // splits in the weight pack (the fp16 pack always carries hi and lo)
  constexpr int NT = 512, NH = COUT / 32, TW = 128, ROWP = 8 + TW + 8;
  constexpr int XROWS = NS * 2 * 4;                 // rows of an X image: [s][g][4 source rows]
  constexpr int XB = XROWS * ROWP;                  // units per X buffer
  constexpr int WP = NS * 9 * 2 * 32;               // units per weight piece
  constexpr int MAXNP = 4;                          // resident pieces: 64 -> 32 (4 chunks) or 32 -> 64 (2 chunks x 2 halves)
  constexpr int XIT = (XB + NT - 1) / NT;           // X DMA rounds per chunk (the last one partial)
  constexpr int WIT = (WP + NT - 1) / NT;
  __shared__ f32x4 smem[2 * XB + MAXNP * WP + COUT / 4];
  f32x4* const Xs = smem;
  f32x4* const Ws = smem + 2 * XB;
  float* const bias_s = reinterpret_cast<float*>(smem + 2 * XB + MAXNP * WP);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, g = lane >> 5;
  const int H = a.H, d = a.dil;
  const int rslot = wave >> 2;                      // 0: output row y0, 1: output row y0 + d
  const int wpx = (wave & 3) * 32;                  // the wave's 32 pixels inside the tile
  const int tiles_per_map = aw.npairs * 2;
  const long ntiles = (long)tiles_per_map * aw.B;

  // tile t -> (map b, first row y0, first pixel px0); pair p -> rows (2d q + r, 2d q + r + d), p = q d + r
  auto tile_geo = [&](long t, int& b, int& y0, int& px0) {
    b = (int)(t / tiles_per_map);
    const int u = (int)(t - (long)b * tiles_per_map);
    const int p = u >> 1;
    px0 = (u & 1) * TW;
    y0 = (p / d) * 2 * d + (p % d);
  };

  // thread-constant part of the X DMA geometry: unit u = it * NT + tid of an X image = (row = (s*2 + g)*4 + j, col)
  int xrow_s[XIT], xrow_g[XIT], xrow_j[XIT], xcol[XIT];
  bool xact[XIT];
#pragma unroll
  for (int it = 0; it < XIT; ++it) {
    const int u = it * NT + tid;
    xact[it] = u < XB;
    const int uu = xact[it] ? u : 0;
    const int row = uu / ROWP;
    xcol[it] = uu - row * ROWP;
    xrow_s[it] = row / 8; xrow_g[it] = (row / 4) & 1; xrow_j[it] = row & 3;
  }
  // per-tile part: global unit offsets relative to the chunk's plane pair (k-dependent part added at issue), or -1 = the zero unit
  long xoff[XIT];
  const f32x4* xb = nullptr;
  auto tile_src = [&](int b, int y0, int px0) {
    xb = a.x + (long)b * a.x_bs;
#pragma unroll
    for (int it = 0; it < XIT; ++it) {
      const int ys = y0 + (xrow_j[it] - 1) * d, px = px0 - 8 + xcol[it];
      const bool ok = ys >= 0 && ys < H && px >= 0 && px < M16_PX;
      xoff[it] = ok ? m16_plane(xrow_g[it], xrow_s[it], NS, H) + (long)ys * M16_PX + px : -1;
    }
  };
  auto issue_x = [&](int k, int buf) {               // the X image of chunk k of the tile whose geometry is in xoff
    const long kbase = m16_plane(2 * k, 0, NS, H);
#pragma unroll
    for (int it = 0; it < XIT; ++it)
      if (xact[it]) p16_glds16(xoff[it] >= 0 ? xb + kbase + xoff[it] : aw.zero, Xs + buf * XB + it * NT + wave * 64);
  };
  auto issue_w = [&](int i) {                        // weight piece i = (k, h) -> its resident slot
    const int k = i / NH, h = i - k * NH;
#pragma unroll
    for (int it = 0; it < WIT; ++it) {
      const int u = tid + it * NT;
      if (u < WP) p16_glds16(reinterpret_cast<const f32x4*>(a.w) + ((long)k * (WNS * 9 * 2) + (u >> 5)) * COUT + h * 32 + (u & 31), Ws + i * WP + it * NT + wave * 64);
    }
  };

  long tile = blockIdx.x;
  if (tile >= ntiles) return;
  int tb, ty0, tpx0;
  tile_geo(tile, tb, ty0, tpx0);
  tile_src(tb, ty0, tpx0);
  if (wave == 0 && lane < COUT / 4) p16_glds16(reinterpret_cast<const f32x4*>(a.bias) + lane, smem + 2 * XB + MAXNP * WP);
  issue_x(0, 0);
#pragma unroll
  for (int h = 0; h < NH; ++h) issue_w(h);           // the pieces of chunk 0

  f32x16 acc[NH];
#pragma unroll
  for (int h = 0; h < NH; ++h)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[h][r] = 0.f;
  const unsigned ws_lds = p16_lds_addr(Ws + g * 32 + l31);
  const unsigned xs_lds = p16_lds_addr(Xs + (g * 4 + rslot) * ROWP + 8 + wpx + l31);
  float vmax = 0.f;
  bool first_tile = true;                            // the weights are still arriving piece by piece
  int cbuf = 0;                                      // X buffer of the current chunk
  constexpr int NRU = NS == 2 ? NH * 4 : 1;
  u32x4_t ru[NRU];

#define M16W_READ_X(dst_, t_) _Pragma("unroll") for (int s = 0; s < NS; ++s) dst_[s] = p16_lds_read16(xcolb[(t_) % 3], (s * 8 + (t_) / 3) * ROWP * 16);
#define M16W_READ_W(dst_, t_) _Pragma("unroll") for (int s = 0; s < NS; ++s) dst_[s] = p16_lds_read16(wrow, ((s * 9 + (t_)) * 2) * 32 * 16);
#define M16W_MFMA(w_, x_, h_)                                                                                      \
  {                                                                                                                \
    typedef typename Op16<DT>::vec V_;                                                                              \
    if constexpr (NS == 2) {                                                                                        \
      acc[h_] = Op16<DT>::mfma(__builtin_bit_cast(V_, w_[0]), __builtin_bit_cast(V_, x_[NS - 1]), acc[h_]);         \
      acc[h_] = Op16<DT>::mfma(__builtin_bit_cast(V_, w_[NS - 1]), __builtin_bit_cast(V_, x_[0]), acc[h_]);         \
    }                                                                                                               \
    acc[h_] = Op16<DT>::mfma(__builtin_bit_cast(V_, w_[0]), __builtin_bit_cast(V_, x_[0]), acc[h_]);               \
  }

  while (true) {
    const long ntile = tile + gridDim.x;
    const bool more = ntile < ntiles;
    int nb = 0, ny0 = 0, npx0 = 0;
    if (more) tile_geo(ntile, nb, ny0, npx0);
    const int yr = ty0 + rslot * d;                  // this wave's output row
    const bool rowvalid = yr < H;
    for (int k = 0; k < a.nchunks; ++k) {
      const bool lastk = k + 1 == a.nchunks;
      // everything this wave has in flight - this chunk's X image (and, on the first tile, its weight pieces), the previous tile's
      // stores - is retired; the barrier then makes it true for every wave and frees the other X buffer
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      M16_BARRIER();
      if (NS == 2 && lastk && a.r && rowvalid) {     // residual units of this lane: their round trip runs under the last chunk's MFMAs
        const f32x4* const rb = a.r + (long)tb * a.r_bs;
        const long rowoff = (long)yr * M16_PX + tpx0 + wpx + l31;
#pragma unroll
        for (int j = 0; j < NRU; ++j) ru[j] = reinterpret_cast<const u32x4_t*>(rb)[m16_plane(j, g, NS, H) + rowoff];
      }
      // the next chunk's X image - of this tile or, under the last chunk, of the next tile - and the next chunk's weight pieces
      if (!lastk) {
        issue_x(k + 1, cbuf ^ 1);
        if (first_tile) {
#pragma unroll
          for (int h = 0; h < NH; ++h) issue_w((k + 1) * NH + h);
        }
      } else if (more) {
        tile_src(nb, ny0, npx0);
        issue_x(0, cbuf ^ 1);
      }
      unsigned xcolb[3];
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) xcolb[kx] = xs_lds + (unsigned)((cbuf * XB + (kx - 1) * d) * 16);
      f16x8 xk[9][NS];
#pragma unroll
      for (int h = 0; h < NH; ++h) {
        const unsigned wrow = ws_lds + (unsigned)(((k * NH + h) * WP) * 16);
        f16x8 wv[2][NS];
        if (h == 0) {
          M16W_READ_X(xk[0], 0); M16W_READ_W(wv[0], 0);
#pragma unroll
          for (int t = 0; t < 9; ++t) {
            const int fb = t & 1;
            if (t + 1 < 9) { M16W_READ_X(xk[t + 1], t + 1); M16W_READ_W(wv[fb ^ 1], t + 1); m16_wait<2 * NS, NS>(xk[t], wv[fb]); }
            else m16_wait<0, NS>(xk[t], wv[fb]);
            M16W_MFMA(wv[fb], xk[t], h);
            __builtin_amdgcn_sched_barrier(0);
          }
        } else {
          M16W_READ_W(wv[0], 0);
#pragma unroll
          for (int t = 0; t < 9; ++t) {
            const int fb = t & 1;
            if (t + 1 < 9) { M16W_READ_W(wv[fb ^ 1], t + 1); m16_wait<NS, NS>(xk[t], wv[fb]); }
            else m16_wait<0, NS>(xk[t], wv[fb]);
            M16W_MFMA(wv[fb], xk[t], h);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      }
      cbuf ^= 1;
    }
    // ---- epilogue of the tile (its stores are retired by the next chunk's vmcnt(0))
    if (rowvalid) m16w_epilogue<COUT, NS, DT>(a, acc, bias_s, tb, yr, tpx0 + wpx + l31, g, ru, vmax);
#pragma unroll
    for (int h = 0; h < NH; ++h)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[h][r] = 0.f;
    first_tile = false;
    if (!more) break;
    tile = ntile; tb = nb; ty0 = ny0; tpx0 = npx0;
  }
#undef M16W_READ_X
#undef M16W_READ_W
#undef M16W_MFMA
  if (DT == 1 && vmax > 65504.f && a.flag) *a.flag = 1u;
}
