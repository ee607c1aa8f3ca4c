// PARKED (round 5; not part of the library): the 32 -> 64 (+ residual) Decoder layers cout half by cout half.
// Built, bit-identical to conv2d_3x3_m16q_kernel (tests/test_gpu_nets.py ran green with it as the default), and NOT faster:
// tools/microbench_m16q (B = 2, d = 1 / 8): 21.4-21.7 us per launch either way; Decoder forward 2.39 vs 2.38 ms.  The s_memtime stamps say why:
// the final epilogue shrinks from 13 000 to 3 600-6 000 cycles, but the residual units requested at the start of a pass take > 6 000 cycles
// to land (the wait in front of piece 1 / 3 grows by 2 400 cycles each) and the first half's epilogue + the second pass take 9 800 cycles
// instead of 4 600: a workgroup moves 446 KB (X 110, weights 74, residual 131 in, 131 out) = 114 MB per launch at B = 2 in ~19 us = 6 TB/s -
// the launch is bound by what crosses L2 / the fabric, not by when it is asked for.  Appended to conv2d_m16q.h it compiles as it stands
// (it uses that file's m16_tile_epilogue<COUT, NS, DT, HB, NH>, M16Q_STAMP and wait helpers).

// ---------------------------------------------------------------------------------------------------------------------------------
// conv2d_3x3_m16q_hm_kernel - the 32 -> 64 layers (+ residual) of the residual blocks, COUT-HALF BY COUT-HALF (round 5).
//
// The four-row kernel above walks its pieces K-chunk major - (k0,h0) (k0,h1) (k1,h0) (k1,h1) - so all 64 couts of a tile finish with the
// last piece, and the 64-channel epilogue (16 residual units in, 16 units out per lane; every workgroup of the round at the same moment:
// 33 MB of residual and 33 MB of output cross the fabric at once) is 13 000 of a workgroup's 38 000 cycles at B = 2 (s_memtime stamps,
// tools/microbench_m16q.hip) - a third of the launch with the matrix pipe idle.  A 32-channel input is two K-chunks: BOTH X images fit the
// two X buffers.  So: pieces (k0,h0) (k1,h0) | (k0,h1) (k1,h1) - the second pass reads the X images that are still in LDS (no transfer
// but its two weight pieces) - and
//   * the first half's epilogue runs right behind the barrier of piece 2: its stores drain under the second pass's MFMAs;
//   * only 32 accumulators and 8 residual units per lane are live at a time, so a half's residual units are requested at the START of
//     its pass (two pieces ahead) instead of under the last kernel column: they have landed when the epilogue wants them;
//   * what is left at the end is half an epilogue: 8 units in registers already, 8 stores.
// Same products, same order of accumulation per cout as the kernel above (K-chunks ascending, kernel-column-major taps, bias first):
// bit-identical maps (tests/test_gpu_kernels.py).  f16x2 only (the single-plane modes load their residual inside the epilogue).
template <int NS, int DT>
__global__ __launch_bounds__(512, 1) void conv2d_3x3_m16q_hm_kernel(ConvM16QArgs aq) {
  static_assert(NS == 2 && DT == 1, "f16x2");
  const ConvM16Args& a = aq.c;
  constexpr int COUT = 64, WNS = 2;
  constexpr int NT = 512, TW = 128, ROWP = 8 + TW + 8, SR = 6;
  constexpr int XROWS = NS * 2 * SR, XB = XROWS * ROWP, WP = NS * 9 * 2 * 32;
  constexpr int XIT = (XB + NT - 1) / NT, XFULL = (XB - (XIT - 1) * NT) / 64;
  constexpr int WIT = (WP + NT - 1) / NT, WFULL = (WP - (WIT - 1) * NT) / 64;
  __shared__ f32x4 smem[2 * XB + 2 * WP + 16];
  f32x4* const Xs = smem;
  f32x4* const Ws = smem + 2 * XB;
  const float* const bias_s = reinterpret_cast<const float*>(smem + 2 * XB + 2 * WP);      // the 64 biases (one DMA instruction of wave 0, below)

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, g = lane >> 5;
  const int H = a.H, d = a.dil;
  int b = blockIdx.y;
  const int nb = aq.nb, bstep = gridDim.y;
  const int per_xcd = (int)gridDim.x >> 3, tix = ((int)blockIdx.x & 7) * per_xcd + ((int)blockIdx.x >> 3);   // XCD-contiguous tiles (see above)
  if (tix >= aq.ngroups * 2) return;
  const int nq = aq.ngroups / d, cls = (tix >> 1) / nq, q4 = (tix >> 1) - cls * nq, px0 = (tix & 1) * TW;
  const int y0 = q4 * 4 * d + cls;
  if (y0 >= H) return;
  const int r0 = (wave >> 2) * 2, wpx = (wave & 3) * 32;
  M16Q_STAMP(0);

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
  auto issue_x1 = [&](const f32x4* xm, int k, int it) {             // chunk k's X image -> buffer k
    if (it + 1 < XIT || wave < XFULL) p16_glds16(xoff[it] >= 0 ? xm + m16_plane(2 * k, 0, NS, H) + xoff[it] : aq.zero, Xs + k * XB + it * NT + wave * 64);
  };
  auto issue_w1 = [&](int i, int it) {                               // weight piece i = (k = i & 1, h = i >> 1) -> ring slot i & 1
    const int k = i & 1, h = i >> 1;
    const int u = tid + it * NT;
    if (it + 1 < WIT || wave < WFULL)
      p16_glds16(reinterpret_cast<const f32x4*>(a.w) + ((long)k * (WNS * 9 * 2) + (u >> 5)) * COUT + h * 32 + (u & 31), Ws + (i & 1) * WP + it * NT + wave * 64);
  };
  // the biases by ONE LDS-DMA instruction (16 lanes x 16 bytes) in front of the first piece: a pass starts its accumulators from LDS - a
  // global load behind the pass's barrier put its round trip (~3 000 cycles under the launch's first fill) in front of the pass's MFMAs
  if (wave == 0 && lane < 16) p16_glds16(reinterpret_cast<const f32x4*>(a.bias) + lane, smem + 2 * XB + 2 * WP);
#pragma unroll
  for (int it = 0; it < WIT; ++it) issue_w1(0, it);
#pragma unroll
  for (int it = 0; it < XIT; ++it) issue_x1(xb, 0, it);
  M16Q_STAMP(1);
  const unsigned ws_lds = p16_lds_addr(Ws + g * 32 + l31);
  const unsigned xs_lds = p16_lds_addr(Xs + (g * SR + r0) * ROWP + 8 + wpx + l31);
  const int px = px0 + wpx + l31;
  const int yr0 = y0 + r0 * d, yr1 = yr0 + d;
  float vmax = 0.f;

  while (true) {       // the maps of this workgroup: b, b + bstep, ...
    const bool has_next = b + bstep < nb;
    const f32x4* const xb_next = xb + (long)bstep * a.x_bs;
    const f32x4* const rb = a.r ? a.r + (long)b * a.r_bs : nullptr;
    f32x16 acc[2][1];
    u32x4_t ru[4], ru1[4];

#define HM_READ_X(dst_, kx_) _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_) _Pragma("unroll") for (int s_ = 0; s_ < NS; ++s_) dst_[j_][s_] = p16_lds_read16(xcol[kx_], ((s_ * 2 * SR) + j_) * ROWP * 16);
#define HM_READ_W(dst_, t_) _Pragma("unroll") for (int s_ = 0; s_ < NS; ++s_) dst_[s_] = p16_lds_read16(wrow, ((s_ * 9 + (t_)) * 2) * 32 * 16);
#define HM_MFMA(w_, x_, j_)                                                                                   \
  {                                                                                                           \
    typedef typename Op16<DT>::vec V_;                                                                        \
    acc[j_][0] = Op16<DT>::mfma(__builtin_bit_cast(V_, w_[0]), __builtin_bit_cast(V_, x_[1]), acc[j_][0]);    \
    acc[j_][0] = Op16<DT>::mfma(__builtin_bit_cast(V_, w_[1]), __builtin_bit_cast(V_, x_[0]), acc[j_][0]);    \
    acc[j_][0] = Op16<DT>::mfma(__builtin_bit_cast(V_, w_[0]), __builtin_bit_cast(V_, x_[0]), acc[j_][0]);    \
  }
    auto piece = [&](auto ic) __attribute__((always_inline)) {
      constexpr int i = decltype(ic)::value;
      constexpr int k = i & 1, h = i >> 1;
      M16Q_STAMP(2 + 2 * i);
      m16q_vmwait<0>();                // this piece's weights (piece 1: and X image) have landed - and, pieces 1 / 3, the half's residual units
      M16_BARRIER();
      M16Q_STAMP(3 + 2 * i);
      if constexpr (i == 2) {          // the first half is complete: its epilogue here, its stores drain under the second pass
        int px_ = px, g_ = g;
        asm volatile("" : "+v"(px_), "+v"(g_));
        if (yr0 < H) m16_tile_epilogue<COUT, NS, DT, 0, 1>(a, acc[0], b, yr0, px_, g_, ru, rb != nullptr, vmax);
        if (yr1 < H) m16_tile_epilogue<COUT, NS, DT, 0, 1>(a, acc[1], b, yr1, px_, g_, ru1, rb != nullptr, vmax);
      }
      if constexpr (k == 0) {          // start of a pass: accumulators from the bias, the half's residual units requested (8 per lane)
        int gb_ = g, px_ = px;
        asm volatile("" : "+v"(gb_), "+v"(px_));
        {
          const unsigned bl = p16_lds_addr(bias_s) + (unsigned)(16 * gb_);     // (asm reads: a visible LDS read costs a compiler-placed vmcnt(0))
          f32x4 b4[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) b4[q] = p16_lds_read16f(bl, (h * 32 + 8 * q) * 4);
          asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(b4[0]), "+v"(b4[1]), "+v"(b4[2]), "+v"(b4[3]));
#pragma unroll
          for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int j = 0; j < 2; ++j) { acc[j][0][4 * q + 0] = b4[q].x; acc[j][0][4 * q + 1] = b4[q].y; acc[j][0][4 * q + 2] = b4[q].z; acc[j][0][4 * q + 3] = b4[q].w; }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          ru[q] = (rb && yr0 < H) ? reinterpret_cast<const u32x4_t*>(rb)[m16_plane(h * 4 + q, gb_, NS, H) + (long)yr0 * M16_PX + px_] : (u32x4_t)(0u);
          ru1[q] = (rb && yr1 < H) ? reinterpret_cast<const u32x4_t*>(rb)[m16_plane(h * 4 + q, gb_, NS, H) + (long)yr1 * M16_PX + px_] : (u32x4_t)(0u);
        }
      }
      // transfers under this piece: the next weight piece; piece 0: chunk 1's X image; piece 3: the workgroup's next map (weights 0 -> slot 0,
      // X image of chunk 0 -> buffer 0: both free since the barrier above)
      const bool dow = i < 3 || has_next;
      const bool dox = i == 0 || (i == 3 && has_next);
      const f32x4* const x_nx = i == 3 ? xb_next : xb;
#define HM_DMA_SLOT(n_)                                                                                 \
  {                                                                                                     \
    const int first_ = (n_) == 0 ? 0 : (n_) + 1, cnt_ = (n_) == 0 ? 2 : 1;                               \
    _Pragma("unroll") for (int q_ = first_; q_ < first_ + cnt_; ++q_) {                                  \
      if (q_ < WIT) { if (dow) issue_w1((i + 1) & 3, q_); }                                              \
      else if (q_ - WIT < XIT) { if (dox) issue_x1(x_nx, i == 0 ? 1 : 0, q_ - WIT); }                    \
    }                                                                                                   \
  }
      const unsigned wrow = ws_lds + (unsigned)((i & 1) * WP * 16);
      unsigned xcol[3];
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) xcol[kx] = xs_lds + (unsigned)((k * XB + (kx - 1) * d) * 16);
      f16x8 xr[2][4][NS], wv[2][NS];
      HM_READ_X(xr[0], 0);
      HM_READ_W(wv[0], 0);
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
          const int n = kx * 3 + ky, fb = n & 1, xq = kx & 1;
          HM_DMA_SLOT(n);
          if (n < 8) { const int n1 = n + 1, t1 = (n1 % 3) * 3 + n1 / 3; HM_READ_W(wv[fb ^ 1], t1); }
          if (ky == 0 && kx < 2) HM_READ_X(xr[xq ^ 1], kx + 1);
          if (ky == 0) {
            if (kx < 2) m16q_wait_xw<5 * NS, NS>(xr[xq], wv[fb]);
            else m16q_wait_xw<NS, NS>(xr[xq], wv[fb]);
          } else {
            if (n < 8) m16q_wait_w<NS, NS>(wv[fb]);
            else m16q_wait_w<0, NS>(wv[fb]);
          }
          HM_MFMA(wv[fb], xr[xq][ky], 0);
          HM_MFMA(wv[fb], xr[xq][ky + 1], 1);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    };
    piece(std::integral_constant<int, 0>());
    piece(std::integral_constant<int, 1>());
    piece(std::integral_constant<int, 2>());
    piece(std::integral_constant<int, 3>());
    M16Q_STAMP(10);
    {
      int px_ = px, g_ = g;
      asm volatile("" : "+v"(px_), "+v"(g_));
      if (yr0 < H) m16_tile_epilogue<COUT, NS, DT, 1, 1>(a, acc[0], b, yr0, px_, g_, ru, rb != nullptr, vmax);
      if (yr1 < H) m16_tile_epilogue<COUT, NS, DT, 1, 1>(a, acc[1], b, yr1, px_, g_, ru1, rb != nullptr, vmax);
    }
    M16Q_STAMP(11);
    if (!has_next) break;
    b += bstep;
    xb = xb_next;
  }
#undef HM_READ_X
#undef HM_READ_W
#undef HM_MFMA
#undef HM_DMA_SLOT
  if (vmax > 65504.f && a.flag) *a.flag = 1u;
}
