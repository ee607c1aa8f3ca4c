// conv_p16p5.h - the planar 16-bit Conv1d k9 (conv_p16.h) of a 128-cout layer with ReLU, residual and MaxPool1d(5) fused: the last conv
// of the Encoder's stage 3 (orca_modules.py:846-879: `out3 + lout3`, then the `MaxPool1d(5)` that opens stage 4).  SURVEY 2b K3.
//
// A pooling window of 5 does not fit the accumulator layout of the other kernels (lane = position inside a 32-wide tile: a window would
// straddle lanes, tiles and waves).  Here the POSITIONS are dealt to the lanes with stride 5: accumulator tile r (0..4) of a wave holds, in
// lane l, position 5 l + r of the wave's 160 - the five members of pooling window l sit in the SAME lane of five accumulator tiles, and the
// pool is a register-local max.  The B operand (input) of tile r at tap t is image column 5 l + r + t: a lane stride of 80 bytes, which the 16
// lanes of a ds_read_b128 group spread over 16 distinct 4-bank slots (80 B = 20 banks; 20 k mod 64, k = 0..15, are 16 different multiples of 4):
// conflict-free.  And the fragment only depends on r + t, so the 45 (tile, tap) pairs of a step need 13 input fragments per split plane, one
// new one per tap: 0.33 ds_read_b128 per MFMA.
// Workgroup tile: 320 positions (64 pooled outputs) x ALL 128 couts - 8 waves = 2 position groups (160 positions) x 4 cout groups (32 couts),
// so the input is read once (the 64-cout tiles of conv_p16.h read it once per cout block).  LDS 116 KB: X image double-buffered (2 x 21 KB),
// ONE 73.7 KB weight buffer refilled half by half behind the waves that are done with it (conv_p16w1.h: two plain-vmcnt(0) barriers per step).
// Epilogue per lane: bias (accumulator init), ReLU, + residual (whole 16-byte units of the lane's five positions, halves traded with lane
// l +- 32), max over the five tiles, hi / lo split, one 16-byte store of the pooled unit; stores beyond n / 5 are masked (the ragged last
// window is dropped as MaxPool1d does).  The argument block is ConvP16Args (out_mode 3; y_plen = plane length of the POOLED output).
// (Round 3 also ran the UNPOOLED 128-cout layers of stages 3-4 on this geometry: measured equal to the 64-cout tiles - 24.66 vs 24.69 ms per
// strand - and removed in round 4; it pays only where it removes a pass.  profiles/HISTORY.md.)
#pragma once
#include "conv_p16.h"

template <int NX>
__device__ __forceinline__ void p16p5_wait_first(f16x8 (&x)[2][13], f16x8 (&w)[2][2]) {   // X fragments 0..4 and the first W pair
  asm volatile("s_waitcnt lgkmcnt(%12)"
               : "+v"(x[0][0]), "+v"(x[1][0]), "+v"(x[0][1]), "+v"(x[1][1]), "+v"(x[0][2]), "+v"(x[1][2]), "+v"(x[0][3]), "+v"(x[1][3]), "+v"(x[0][4]),
                 "+v"(x[1][4]), "+v"(w[0][0]), "+v"(w[1][0])
               : "n"(NX));
}
template <int N>
__device__ __forceinline__ void p16p5_wait(f16x8& x0, f16x8& x1, f16x8& w0, f16x8& w1) {
  asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(x0), "+v"(x1), "+v"(w0), "+v"(w1) : "n"(N));
}

// FMT = 0: P16 (2 x fp16 split planes, 3 products, 16 input channels per step); FMT = 1: B16 (one bf16 plane per channel octet, 1 product, 32
// input channels per step: the split index s of the images becomes the k-pair index, conv_p16.h)
template <bool R1, int FMT = 0>
__global__ __launch_bounds__(512, 2) void conv1d_k9_p16p5_kernel(ConvP16Args a) {
  constexpr int CT = 128, NT = 512, GP = 160, MT = 2 * GP;      // 2 position groups of 160 = 32 windows of 5
  constexpr int XROW = MT + 8;
  constexpr int XU = 2 * 2 * XROW;          // X image units [s][g][XROW]
  constexpr int WU = 2 * 9 * 2 * CT;        // W image units [s][tap][g][CT]
  constexpr int WSPLIT = 5;
  constexpr int WH0 = 2 * WSPLIT * 2 * CT, WH1 = WU - WH0;
  constexpr int XIT = (XU + NT - 1) / NT, W0IT = (WH0 + NT - 1) / NT, W1IT = (WH1 + NT - 1) / NT;
  static_assert((WSPLIT * 2 * CT) % 64 == 0 && ((9 - WSPLIT) * 2 * CT) % 64 == 0, "a wave's 64 DMA lanes never straddle the split planes");
  __shared__ f32x4 smem[2 * XU + WU + 32];
  f32x4* const Wl = smem + 2 * XU;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int pg = wave >> 2, cg = wave & 3;  // (waves w, w + 4 share a SIMD: same couts, the two position groups)
  const int l31 = lane & 31, g = lane >> 5;
  const long ntiles = a.tiles_per_row;
  long tile = blockIdx.x;
  if (tile >= ntiles) return;
  float* bias_s = reinterpret_cast<float*>(smem + 2 * XU + WU);
  if (tid < CT) bias_s[tid] = a.bias[tid];

  int xrel[XIT];
  bool xact[XIT];
#pragma unroll
  for (int it = 0; it < XIT; ++it) {
    const int i = tid + it * NT;
    xact[it] = i < XU;
    const int ii = xact[it] ? i : 0;
    const int row = ii / XROW, col = ii - row * XROW;      // row = s*2 + g
    const int s = row >> 1, gg = row & 1;
    xrel[it] = (int)((FMT == 1 ? (s * 2 + gg) : (gg * 2 + s)) * a.x_plen) + col;
  }
  auto w0_unit = [](int k) { return k < WSPLIT * 2 * CT ? k : k + (9 - WSPLIT) * 2 * CT; };
  auto w1_unit = [](int k) { return k < (9 - WSPLIT) * 2 * CT ? k + WSPLIT * 2 * CT : k + 2 * WSPLIT * 2 * CT; };
  const f32x4 *xsrc = nullptr, *wsrc = nullptr;
  auto set_src = [&](long pos, int c) {
    const int cx = a.k17 ? (c >> 1) : c;
    const int xo = a.k17 ? ((c & 1) ? 9 : 0) : (P16_GUARD - P16_HALO);
    xsrc = a.x + (long)cx * 4 * a.x_plen + pos * MT + xo;
    wsrc = a.w + (long)c * WU;
  };
  auto issue_x = [&](int buf) {
#pragma unroll
    for (int it = 0; it < XIT; ++it)
      if (xact[it]) p16_glds16(xsrc + xrel[it], smem + buf * XU + it * NT + wave * 64);
  };
  auto issue_w0 = [&]() {
#pragma unroll
    for (int it = 0; it < W0IT; ++it) {
      const int k0 = it * NT + wave * 64;
      if (k0 < WH0) p16_glds16(wsrc + w0_unit(k0) + lane, Wl + w0_unit(k0));
    }
  };
  auto issue_w1 = [&]() {
#pragma unroll
    for (int it = 0; it < W1IT; ++it) {
      const int k0 = it * NT + wave * 64;
      if (k0 < WH1) p16_glds16(wsrc + w1_unit(k0) + lane, Wl + w1_unit(k0));
    }
  };

  f32x16 acc[5];
  auto acc_init = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 b = *reinterpret_cast<const f32x4*>(bias_s + cg * 32 + 8 * q + 4 * g);
#pragma unroll
      for (int r = 0; r < 5; ++r) { acc[r][4 * q + 0] = b.x; acc[r][4 * q + 1] = b.y; acc[r][4 * q + 2] = b.z; acc[r][4 * q + 3] = b.w; }
    }
  };
  const long xpl16 = a.x_plen * 16, ypl16 = a.y_plen * 16;
  const long npool = a.n / 5;
  float vmax = 0.f;
  long epi_pos = -1;

  auto epilogue_p16 = [&](long pos) __attribute__((always_inline)) {   // (not inlined, the accumulators live in scratch)
    const long p0 = pos * MT + pg * GP + 5 * l31;           // + r
    const long P = pos * (MT / 5) + pg * 32 + l31;          // pooled position of the lane
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int oct = cg * 4 + q;                           // channel octet (8 couts): planes 2 oct (hi), 2 oct + 1 (lo)
      u32x4_t rr[R1 ? 5 : 1];
      if (R1) {
        const char* rb = reinterpret_cast<const char*>(a.r1) + (long)oct * 2 * xpl16 + (g ? xpl16 : 0) + (P16_GUARD + p0) * 16;
#pragma unroll
        for (int r = 0; r < 5; ++r) rr[r] = *reinterpret_cast<const u32x4_t*>(rb + r * 16);
      }
      f32x4 m;
#pragma unroll
      for (int r = 0; r < 5; ++r) {
        f32x4 v;
        v.x = acc[r][4 * q + 0]; v.y = acc[r][4 * q + 1]; v.z = acc[r][4 * q + 2]; v.w = acc[r][4 * q + 3];
        if (a.relu) { v.x = p16_vmax(v.x, 0.f); v.y = p16_vmax(v.y, 0.f); v.z = p16_vmax(v.z, 0.f); v.w = p16_vmax(v.w, 0.f); }
        if (R1) {   // the lane loaded the whole unit of plane (hi | lo by g): trade halves with lane l +- 32 (conv_p16.h)
          unsigned ux = rr[r].x, uy = rr[r].y, uz = rr[r].z, uw = rr[r].w;
          p16_swap32(ux, uz);
          p16_swap32(uy, uw);
          const f16x2 h0 = __builtin_bit_cast(f16x2, ux), h1 = __builtin_bit_cast(f16x2, uy);
          const f16x2 l0 = __builtin_bit_cast(f16x2, uz), l1 = __builtin_bit_cast(f16x2, uw);
          v.x += (float)h0.x + (float)l0.x; v.y += (float)h0.y + (float)l0.y;
          v.z += (float)h1.x + (float)l1.x; v.w += (float)h1.y + (float)l1.y;
        }
        if (r == 0) m = v;
        else { m.x = p16_vmax(m.x, v.x); m.y = p16_vmax(m.y, v.y); m.z = p16_vmax(m.z, v.z); m.w = p16_vmax(m.w, v.w); }
      }
      {
        vmax = p16_vmax3_abs(p16_vmax3_abs(vmax, m.x, m.y), m.z, m.w);
        unsigned h0, h1, l0, l1;
        p16_split_hl(m, h0, h1, l0, l1);
        p16_swap32(h0, l0);    // g=0: {h, l} = hi halves of couts 0-3 | 4-7 of the octet;  g=1: the lo halves
        p16_swap32(h1, l1);
        u32x4_t unit;
        unit.x = h0; unit.y = h1; unit.z = l0; unit.w = l1;
        if (P < npool)
          *reinterpret_cast<u32x4_t*>(reinterpret_cast<char*>(a.y) + (long)oct * 2 * ypl16 + (g ? ypl16 : 0) + (P16_GUARD + P) * 16) = unit;
      }
    }
  };
  auto epilogue_b16 = [&](long pos) __attribute__((always_inline)) {   // (conv_p16.h, P16_EPILOGUE_B16: the lane handles 4 couts of TWO octets per register octet)
    const long p0 = pos * MT + pg * GP + 5 * l31;
    const long P = pos * (MT / 5) + pg * 32 + l31;
#pragma unroll
    for (int qp = 0; qp < 2; ++qp) {
      const int oct = cg * 4 + 2 * qp;                      // planes oct (q0) and oct + 1 (q1); the lane loads / stores plane oct + g
      u32x4_t rr[R1 ? 5 : 1];
      if (R1) {
        const char* rb = reinterpret_cast<const char*>(a.r1) + (long)(oct + g) * xpl16 + (P16_GUARD + p0) * 16;
#pragma unroll
        for (int r = 0; r < 5; ++r) rr[r] = *reinterpret_cast<const u32x4_t*>(rb + r * 16);
      }
      f32x4 m0, m1;
#pragma unroll
      for (int r = 0; r < 5; ++r) {
        f32x4 v0, v1;
        v0.x = acc[r][8 * qp + 0]; v0.y = acc[r][8 * qp + 1]; v0.z = acc[r][8 * qp + 2]; v0.w = acc[r][8 * qp + 3];
        v1.x = acc[r][8 * qp + 4]; v1.y = acc[r][8 * qp + 5]; v1.z = acc[r][8 * qp + 6]; v1.w = acc[r][8 * qp + 7];
        if (a.relu) {
          v0.x = p16_vmax(v0.x, 0.f); v0.y = p16_vmax(v0.y, 0.f); v0.z = p16_vmax(v0.z, 0.f); v0.w = p16_vmax(v0.w, 0.f);
          v1.x = p16_vmax(v1.x, 0.f); v1.y = p16_vmax(v1.y, 0.f); v1.z = p16_vmax(v1.z, 0.f); v1.w = p16_vmax(v1.w, 0.f);
        }
        if (R1) {
          unsigned ux = rr[r].x, uy = rr[r].y, uz = rr[r].z, uw = rr[r].w;
          p16_swap32(ux, uz);   // ux, uy = couts 4g..4g+3 of q0;  uz, uw = the same of q1
          p16_swap32(uy, uw);
          v0.x += bf16lo_f32(ux); v0.y += bf16hi_f32(ux); v0.z += bf16lo_f32(uy); v0.w += bf16hi_f32(uy);
          v1.x += bf16lo_f32(uz); v1.y += bf16hi_f32(uz); v1.z += bf16lo_f32(uw); v1.w += bf16hi_f32(uw);
        }
        if (r == 0) { m0 = v0; m1 = v1; }
        else {
          m0.x = p16_vmax(m0.x, v0.x); m0.y = p16_vmax(m0.y, v0.y); m0.z = p16_vmax(m0.z, v0.z); m0.w = p16_vmax(m0.w, v0.w);
          m1.x = p16_vmax(m1.x, v1.x); m1.y = p16_vmax(m1.y, v1.y); m1.z = p16_vmax(m1.z, v1.z); m1.w = p16_vmax(m1.w, v1.w);
        }
      }
      unsigned a0 = cvt_pk_bf16(m0.x, m0.y), a1 = cvt_pk_bf16(m0.z, m0.w);
      unsigned b0 = cvt_pk_bf16(m1.x, m1.y), b1 = cvt_pk_bf16(m1.z, m1.w);
      p16_swap32(a0, b0);   // g=0: {a, b} = couts 0-3 | 4-7 of plane oct;  g=1: of plane oct + 1
      p16_swap32(a1, b1);
      u32x4_t unit;
      unit.x = a0; unit.y = a1; unit.z = b0; unit.w = b1;
      if (P < npool) *reinterpret_cast<u32x4_t*>(reinterpret_cast<char*>(a.y) + (long)(oct + g) * ypl16 + (P16_GUARD + P) * 16) = unit;
    }
  };
  auto epilogue = [&](long pos) __attribute__((always_inline)) {
    if constexpr (FMT == 1) epilogue_b16(pos); else epilogue_p16(pos);
  };

  set_src(tile, 0);
  issue_x(0);
  issue_w0();
  __syncthreads();            // X(0), weight taps 0-4 of the first step, the bias
  acc_init();

  int c = 0, cur = 0;
  while (true) {
    const bool last_chunk = (c + 1 == a.nchunks);
    const long ntile = last_chunk ? tile + gridDim.x : tile;
    const int nc = last_chunk ? 0 : c + 1;
    const bool more = ntile < ntiles;
    // ---- phase A: everyone is done with weight taps 5-8 of the previous step.  The finished tile's epilogue first (nothing is in flight)
    if (epi_pos >= 0) {
      __builtin_amdgcn_s_setprio(3);
      epilogue(epi_pos);
      acc_init();
      __builtin_amdgcn_s_setprio(0);
      epi_pos = -1;
    }
    issue_w1();
    const bool skip_tap8 = a.k17 && (c & 1);
    const unsigned xa0 = p16_lds_addr(smem + cur * XU + g * XROW + pg * GP + 5 * l31);   // + (s*2*XROW + r + tap) * 16
    const unsigned wb0 = p16_lds_addr(Wl + g * CT + cg * 32 + l31);                     // + (tap * 2 * CT) * 16; split plane s at wbs[s]
    const unsigned wbs[2] = {wb0, wb0 + 9 * 2 * CT * 16};                               // (the s = 1 offsets exceed ds_read's 16-bit immediate)
    f16x8 xf[2][13], wf[2][2];   // input fragment u = r + tap (both split planes); weight fragments double-buffered across taps
#pragma unroll
    for (int u = 0; u < 5; ++u)
#pragma unroll
      for (int s = 0; s < 2; ++s) xf[s][u] = p16_lds_read16(xa0, (s * 2 * XROW + u) * 16);
#pragma unroll
    for (int s = 0; s < 2; ++s) wf[s][0] = p16_lds_read16(wbs[s], 0);
#define P5_TAP(tap)                                                                                                    \
  {                                                                                                                    \
    const bool skip_ = (tap) == 8 && skip_tap8;   /* (uniform) the 18th tap of a 17-tap conv */                         \
    constexpr int fb = (tap) & 1;                                                                                      \
    constexpr bool NEWX = (tap) < 8, NEWW = (tap) < 8 && (tap) != WSPLIT - 1;   /* weight taps 5-8 are only read behind barrier B */ \
    if (NEWX) {                                                                                                        \
      _Pragma("unroll") for (int s = 0; s < 2; ++s) xf[s][(tap) + 5 < 13 ? (tap) + 5 : 12] = p16_lds_read16(xa0, (s * 2 * XROW + (tap) + 5) * 16); \
    }                                                                                                                  \
    if (NEWW) {                                                                                                        \
      _Pragma("unroll") for (int s = 0; s < 2; ++s) wf[s][fb ^ 1] = p16_lds_read16(wbs[s], (((tap) + 1) * 2 * CT) * 16); \
    }                                                                                                                  \
    if ((tap) == 0) p16p5_wait_first<4>(xf, wf);                                                                       \
    else p16p5_wait<2 * NEWX + 2 * NEWW>(xf[0][(tap) + 4], xf[1][(tap) + 4], wf[0][fb], wf[1][fb]);                     \
    if (skip_) {                                                                                                       \
    } else if constexpr (FMT == 1) {                                                                                   \
      _Pragma("unroll") for (int p = 0; p < 2; ++p) _Pragma("unroll") for (int r = 0; r < 5; ++r)                      \
        acc[r] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wf[p][fb]), __builtin_bit_cast(bf16x8, xf[p][r + (tap)]), acc[r], 0, 0, 0); \
    } else {                                                                                                           \
      _Pragma("unroll") for (int p = 0; p < 3; ++p) {                                                                  \
        constexpr int PA[3] = {1, 0, 0}, PB[3] = {0, 1, 0};                                                            \
        _Pragma("unroll") for (int r = 0; r < 5; ++r)                                                                  \
          acc[r] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[PB[p]][fb], xf[PA[p]][r + (tap)], acc[r], 0, 0, 0);       \
      }                                                                                                                \
    }                                                                                                                  \
    __builtin_amdgcn_sched_barrier(0);                                                                                 \
  }
    P5_TAP(0) P5_TAP(1) P5_TAP(2) P5_TAP(3) P5_TAP(4)
    // ---- phase B: taps 5-8 have landed; everyone is done with taps 0-4 of this step
    __syncthreads();
    if (more) {
      set_src(ntile, nc);
      issue_x(cur ^ 1);
      issue_w0();
    }
#pragma unroll
    for (int s = 0; s < 2; ++s) wf[s][1] = p16_lds_read16(wbs[s], (5 * 2 * CT) * 16);
    P5_TAP(5) P5_TAP(6) P5_TAP(7) P5_TAP(8)
#undef P5_TAP
    if (last_chunk) epi_pos = tile;
    if (!more) break;
    __syncthreads();          // X and weight taps 0-4 of the next step have landed; everyone is done with this step's X and taps 5-8
    tile = ntile;
    c = nc;
    cur ^= 1;
  }
  if (epi_pos >= 0) epilogue(epi_pos);
  if (FMT == 0 && vmax > 65504.f && a.flag) *a.flag = 1u;
}
