// conv_p16x.h - the 96-cout planar Conv1d k9 of conv_p16w1.h on the 16 x 16 x 32 matrix instruction.
//
// Why: the Encoder's conv kernels run against the package power cap, and the cap is set by what an MFMA moves through the register file.
// tools/microbench_mfma_power.hip (all CUs issuing f16 MFMAs, operand fragments re-read from LDS at this kernel's rate): 1 594 TFLOP/s with
// v_mfma_f32_32x32x16_f16 against 1 865 with v_mfma_f32_16x16x32_f16 on random operands (+17 %; +14 % with half of them zero) - the 16 x 16
// shape reads and writes a quarter of the accumulator registers per instruction for half the FLOPs.
//
// K = 32 of one instruction = TWO TAPS of the step's 16 input channels: lane group g = lane / 16 supplies channel octet g & 1 of tap
// 2 j + (g >> 1), for the weights (A operand, rows = 16 couts) and for the input (B operand, columns = 16 positions: the second tap is the
// image column one further) - pure addressing, the LDS images, the DMA and the weight pack are those of conv_p16w1.h.  The ninth tap has no
// partner: its hi x hi product runs with the upper half of K zeroed (the lanes g >= 2 hold a zero weight fragment), and its two cross
// products SHARE one instruction - weights [hi | lo] against input [lo | hi] sum to Whi Xlo + Wlo Xhi: 14 instructions per 16 x 16 tile
// and step (4 tap pairs x 3 products + 2) against the 13.5 the arithmetic needs.
// Workgroup: 512 positions x 96 couts, 16 waves (8 position groups of 64 x 2 cout groups of 48): wave tile 4 x 3 tiles of 16 x 16 = 48
// accumulator registers; per tap pair the three products run as Whi Xlo, Whi Xhi, Wlo Xhi on 11 + 3 single-buffered fragments (Wlo is
// fetched while Whi Xhi runs); four waves per SIMD cover the LDS latency; 118-126 VGPRs.  Measured (rocprofv3, same boxes): 3.59 / 2.95 /
// 3.37 ms for stage 2's three launches against 3.72-3.81 / 3.11-3.14 / 3.35-3.39 on conv_p16w1.h: -0.3 ms per strand.  The 8-wave form
// (NC = 6: 64 x 96 wave tiles, two waves per SIMD) measured 3.72 / 3.02 / 3.51 with the same single-buffered schedule and 3.49 / 2.94 / 3.39
// with every fragment group fetched one product ahead (counted lgkmcnt waits): no better than 16 waves.  Two things keep the register
// count under the caps: the DMA offsets are 32-bit (saddr + voffset), and the epilogue takes the lane id from an opaque copy - its
// per-lane addresses are invariant over the tile loop, and hoisted they sat in ~20 registers through the MFMA blocks and pushed the DMA
// offsets into scratch, whose reloads then wait on vmcnt in the middle of the step (+8 %).
// Accumulator tile: lane (p = lane & 15, g) holds couts 4 g .. 4 g + 3 of position p, so the rows g = 2k / 2k + 1 hold the two halves of
// channel octet k: v_permlane16_swap pairs them exactly as v_permlane32_swap pairs lanes l / l + 32 in conv_p16.h, and every lane stores -
// and loads, for the residual - whole 16-byte units.  Steps, barriers and the half-by-half weight refill: conv_p16w1.h with the split after
// tap 3 (a tap pair may not straddle it).
#pragma once
#include "conv_p16.h"

__device__ __forceinline__ void p16_swap16(unsigned& a, unsigned& b) {   // v_permlane16_swap: rows 1, 3 of `a` <-> rows 0, 2 of `b`
#if defined(__HIP_DEVICE_COMPILE__)
  const u32x2 r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
  a = r.x; b = r.y;
#endif
}
template <int NC>
__device__ __forceinline__ void p16x_wait11(f16x8 (&x0)[4], f16x8 (&x1)[4], f16x8 (&w)[NC]) {
  static_assert(NC == 2 || NC == 3 || NC == 6, "operand lists below");
  if constexpr (NC == 2)
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(x0[0]), "+v"(x0[1]), "+v"(x0[2]), "+v"(x0[3]), "+v"(x1[0]), "+v"(x1[1]), "+v"(x1[2]), "+v"(x1[3]), "+v"(w[0]), "+v"(w[1]));
  else if constexpr (NC == 3)
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(x0[0]), "+v"(x0[1]), "+v"(x0[2]), "+v"(x0[3]), "+v"(x1[0]), "+v"(x1[1]), "+v"(x1[2]), "+v"(x1[3]), "+v"(w[0]), "+v"(w[1]), "+v"(w[2]));
  else
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(x0[0]), "+v"(x0[1]), "+v"(x0[2]), "+v"(x0[3]), "+v"(x1[0]), "+v"(x1[1]), "+v"(x1[2]), "+v"(x1[3]), "+v"(w[0]), "+v"(w[1]), "+v"(w[2]),
                   "+v"(w[3]), "+v"(w[4]), "+v"(w[5]));
}
template <int NC>
__device__ __forceinline__ void p16x_wait3(f16x8 (&w)[NC]) {
  if constexpr (NC == 2) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(w[0]), "+v"(w[1]));
  else if constexpr (NC == 3) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(w[0]), "+v"(w[1]), "+v"(w[2]));
  else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(w[0]), "+v"(w[1]), "+v"(w[2]), "+v"(w[3]), "+v"(w[4]), "+v"(w[5]));
}

// NC = cout tiles (of 16) per wave: 3 -> 16 waves (2 cout groups), 6 -> 8 waves
// NC = cout tiles (of 16) per wave: 3 -> 16 waves (2 cout groups), 6 -> 8 waves
// (Also measured and dropped: with 64-cout tiles two accumulator sets fit, so the finished tile's epilogue can run from a parked copy inside
// the next tile's first step, at a different tap pair for each of a SIMD's four waves - correct, and no faster (26.61 vs 26.52 ms per strand
// in the stored-residual form of stage 1): under the power cap an idle matrix pipe is not the loss, the energy per output is.)
// CT = couts per workgroup tile (96: stage 2; 64: the cout blocks of the 64- / 128-cout layers - round 3's opt-in ORCA_P16X_64, removed in round 4: measured equal to the
// 32 x 32 x 16 tiles of conv_p16.h there: 24.65 vs 24.56 ms per strand), NC = cout tiles (of 16) per wave:
// CT / 16 / NC cout groups x 8 position groups of waves (96 / 3 and 64 / 2: 16 waves)
template <int OM, bool R1, int NC = 3, int CT = 96>
__global__ __launch_bounds__(8 * (CT / 16 / NC) * 64) void conv1d_k9_p16x_kernel(ConvP16Args a) {
  constexpr int CGS = CT / 16 / NC, NT = 8 * CGS * 64, MT = 512;
  static_assert(CGS * NC * 16 == CT && (CT == 96 || CT == 64), "cout groups; weight rows of 64 or all 96 couts");
  constexpr int XROW = MT + 8;
  constexpr int XU = 2 * 2 * XROW;          // X image units [s][octet][XROW]
  constexpr int WU = 2 * 9 * 2 * CT;        // W image units [s][tap][octet][CT]
  constexpr int WSPLIT = 4;                 // taps 0 .. 3 | 4 .. 8
  constexpr int WH0 = 2 * WSPLIT * 2 * CT, WH1 = WU - WH0;
  constexpr int XIT = (XU + NT - 1) / NT, W0IT = (WH0 + NT - 1) / NT, W1IT = (WH1 + NT - 1) / NT;
  static_assert((WSPLIT * 2 * CT) % 64 == 0 && ((9 - WSPLIT) * 2 * CT) % 64 == 0, "a wave's 64 DMA lanes never straddle the split planes");
  __shared__ f32x4 smem[2 * XU + WU + 32];
  f32x4* const Wl = smem + 2 * XU;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int pgrp = wave / CGS, cgrp = wave % CGS;     // 8 position groups of 64, CGS cout groups of NC * 16
  const int l15 = lane & 15, g = lane >> 4;
  const long ntiles = CT == 96 ? a.tiles_per_row : a.tiles_per_row * (a.cout / CT);   // cout block cb = tile / tiles_per_row
  long tile = blockIdx.x;
  if (tile >= ntiles) return;
  float* bias_s = reinterpret_cast<float*>(smem + 2 * XU + WU);
  if (tid < a.cout) bias_s[tid] = a.bias[tid];

  // byte offsets of the thread's X units from the step's source (< 4 GB: saddr + 32-bit voffset addressing, one VGPR per round)
  unsigned xrel[XIT];
  bool xact[XIT];
#pragma unroll
  for (int it = 0; it < XIT; ++it) {
    const int i = tid + it * NT;
    xact[it] = i < XU;
    const int ii = xact[it] ? i : 0;
    const int row = ii / XROW, col = ii - row * XROW;      // row = s*2 + octet
    const int s = row >> 1, gg = row & 1;
    xrel[it] = (unsigned)(((long)(gg * 2 + s) * a.x_plen + col) * 16);
  }
  auto w0_unit = [](int k) { return k < WSPLIT * 2 * CT ? k : k + (9 - WSPLIT) * 2 * CT; };                      // s = 0: taps 0..3 | s = 1
  auto w1_unit = [](int k) { return k < (9 - WSPLIT) * 2 * CT ? k + WSPLIT * 2 * CT : k + 2 * WSPLIT * 2 * CT; };
  const f32x4 *xsrc = nullptr, *wsrc = nullptr;
  long wrow = 0;             // source stride of a weight row [s][tap][octet]: all couts of the layer
  auto set_src = [&](long t, int c) {
    long pos = t;
    int cb = 0;
    if constexpr (CT != 96) while (pos >= a.tiles_per_row) { pos -= a.tiles_per_row; ++cb; }
    const int cx = a.k17 ? (c >> 1) : c;
    const int xo = a.k17 ? ((c & 1) ? 9 : 0) : (P16_GUARD - P16_HALO);
    xsrc = a.x + (long)cx * 4 * a.x_plen + pos * MT + xo;
    if constexpr (CT == 96) {
      wsrc = a.w + (long)c * WU;
    } else {
      wrow = a.cout;
      wsrc = a.w + (long)c * (2 * 9 * 2) * wrow + cb * CT;
    }
  };
  auto issue_x = [&](int buf) {
#pragma unroll
    for (int it = 0; it < XIT; ++it)
      if (xact[it]) p16_glds16(reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(xsrc) + xrel[it]), smem + buf * XU + it * NT + wave * 64);
  };
  auto issue_w0 = [&]() {
#pragma unroll
    for (int it = 0; it < W0IT; ++it) {
      const int k0 = it * NT + wave * 64;        // wave-uniform
      if constexpr (CT == 96) {           // the layer has 96 couts: image rows = source rows
        if (k0 < WH0) p16_glds16(wsrc + w0_unit(k0) + lane, Wl + w0_unit(k0));
      } else {                            // CT = 64: a wave's 64 lanes are one image row [s][tap][octet] = 64 of the source row's couts
        if (k0 < WH0) p16_glds16(wsrc + (long)(w0_unit(k0) / CT) * wrow + lane, Wl + w0_unit(k0));
      }
    }
  };
  auto issue_w1 = [&]() {
#pragma unroll
    for (int it = 0; it < W1IT; ++it) {
      const int k0 = it * NT + wave * 64;
      if constexpr (CT == 96) {
        if (k0 < WH1) p16_glds16(wsrc + w1_unit(k0) + lane, Wl + w1_unit(k0));
      } else {
        if (k0 < WH1) p16_glds16(wsrc + (long)(w1_unit(k0) / CT) * wrow + lane, Wl + w1_unit(k0));
      }
    }
  };

  f32x4 acc[4][NC];      // [position tile][cout tile]: couts cgrp*48 + n*16 + 4 g + r of position pgrp*64 + i*16 + l15
  auto acc_init = [&](long t) __attribute__((always_inline)) {
    int cb = 0;
    if constexpr (CT != 96) while (t >= a.tiles_per_row) { t -= a.tiles_per_row; ++cb; }
#pragma unroll
    for (int n = 0; n < NC; ++n) {
      const f32x4 b = *reinterpret_cast<const f32x4*>(bias_s + cb * CT + cgrp * (NC * 16) + n * 16 + 4 * g);
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i][n] = b;
    }
  };
  const long xpl16 = a.x_plen * 16, ypl16 = a.y_plen * 16;
  float vmax = 0.f;
  long epi_pos = -1;

  auto epilogue = [&](long pos) __attribute__((always_inline)) {
    int cb = 0;
    if constexpr (CT != 96) while (pos >= a.tiles_per_row) { pos -= a.tiles_per_row; ++cb; }
    // the lane's coordinates from an opaque copy: every per-lane address below is loop-invariant, and hoisted out of the tile loop they
    // would sit in ~20 registers through the MFMA blocks (the pooled + residual form then spills its DMA offsets)
    unsigned lane_ = (unsigned)lane;
    asm volatile("" : "+v"(lane_));
    const int l15 = (int)(lane_ & 15), g = (int)(lane_ >> 4);
    const long m0 = pos * MT + pgrp * 64;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const long p = m0 + i * 16 + l15;
      // residual: the lane loads whole units of plane (hi | lo by g & 1) - the three of a position tile up front - and trades halves
      // with the lane 16 further / back
      u32x4_t rr[R1 ? NC : 1];
      if (R1) {
#pragma unroll
        for (int n = 0; n < NC; ++n)
          rr[n] = *reinterpret_cast<const u32x4_t*>(reinterpret_cast<const char*>(a.r1) + ((long)(cb * (CT / 8) + cgrp * (NC * 2) + n * 2 + (g >> 1)) * 2 + (g & 1)) * xpl16 + (P16_GUARD + p) * 16);
      }
#pragma unroll
      for (int n = 0; n < NC; ++n) {
        const int oct = cb * (CT / 8) + cgrp * (NC * 2) + n * 2 + (g >> 1);      // channel octet of the lane's row pair; plane 2 oct + (g & 1) after the swap
        f32x4 v = acc[i][n];
        if (a.relu) { v.x = p16_vmax(v.x, 0.f); v.y = p16_vmax(v.y, 0.f); v.z = p16_vmax(v.z, 0.f); v.w = p16_vmax(v.w, 0.f); }
        if (R1) {
          const u32x4_t u = rr[n];
          unsigned ux = u.x, uy = u.y, uz = u.z, uw = u.w;
          p16_swap16(ux, uz);
          p16_swap16(uy, uw);
          const f16x2 h0 = __builtin_bit_cast(f16x2, ux), h1 = __builtin_bit_cast(f16x2, uy);
          const f16x2 l0 = __builtin_bit_cast(f16x2, uz), l1 = __builtin_bit_cast(f16x2, uw);
          v.x += (float)h0.x + (float)l0.x; v.y += (float)h0.y + (float)l0.y;
          v.z += (float)h1.x + (float)l1.x; v.w += (float)h1.y + (float)l1.y;
        }
        if (OM == 2) {
          if (p < a.n) *reinterpret_cast<f32x4*>(reinterpret_cast<char*>(a.y) + (p * a.cout + cb * CT + cgrp * (NC * 16) + n * 16 + 4 * g) * 4) = v;
        } else {
          if (OM == 1) { v.x = p16_dpp_quad_max(v.x); v.y = p16_dpp_quad_max(v.y); v.z = p16_dpp_quad_max(v.z); v.w = p16_dpp_quad_max(v.w); }
          vmax = p16_vmax3_abs(p16_vmax3_abs(vmax, v.x, v.y), v.z, v.w);
          unsigned h0, h1, l0, l1;
          p16_split_hl(v, h0, h1, l0, l1);
          p16_swap16(h0, l0);    // rows 0 / 2: {h, l} = hi halves of couts 0-3 | 4-7 of the octet;  rows 1 / 3: the lo halves
          p16_swap16(h1, l1);
          u32x4_t unit;
          unit.x = h0; unit.y = h1; unit.z = l0; unit.w = l1;
          char* yb = reinterpret_cast<char*>(a.y) + ((long)oct * 2 + (g & 1)) * ypl16;
          if (OM == 0) {
            *reinterpret_cast<u32x4_t*>(yb + (P16_GUARD + p) * 16) = unit;
          } else {               // MaxPool1d(4): the 4 lanes of a quad store one dword each of the pooled unit
            const unsigned d01 = (l15 & 1) ? unit.y : unit.x, d23 = (l15 & 1) ? unit.w : unit.z;
            *reinterpret_cast<unsigned*>(yb + (P16_GUARD + ((m0 + i * 16) >> 2) + (l15 >> 2)) * 16 + (l15 & 3) * 4) = (l15 & 2) ? d23 : d01;
          }
        }
      }
    }
  };

  set_src(tile, 0);
  issue_x(0);
  issue_w0();
  __syncthreads();            // X(0), weight taps 0-3 of the first step, the bias
  acc_init(tile);

  // thread-constant fragment addresses (bytes in LDS)
  const unsigned x_lane = (unsigned)(((g & 1) * XROW + (g >> 1) + pgrp * 64 + l15) * 16);   // tap pair: + (s*2*XROW + i*16 + 2 j) * 16
  const unsigned w_lane = p16_lds_addr(Wl + g * CT + cgrp * (NC * 16) + l15);                                  // tap pair: + ((s*9 + 2 j) * 2 * CT + n*16) * 16
#define P16X_MM(W_, X_)                                                                                                \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) _Pragma("unroll") for (int n = 0; n < NC; ++n)                       \
      acc[i][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(W_[n], X_[i], acc[i][n], 0, 0, 0);
#define P16X_LDW(D_, base_, uoff_) _Pragma("unroll") for (int n = 0; n < NC; ++n) D_[n] = p16_lds_read16(base_, ((uoff_) + n * 16) * 16);
#define P16X_LDX(D_, base_, uoff_) _Pragma("unroll") for (int i = 0; i < 4; ++i) D_[i] = p16_lds_read16(xb + base_, ((uoff_) + i * 16) * 16);
  // one tap pair: Whi, Xlo, Xhi (11 fragments) -> Whi Xlo;  Wlo is fetched while Whi Xhi runs -> Wlo Xhi
#define P16X_PAIR(j)                                                                                                   \
  {                                                                                                                    \
    f16x8 xlo[4], xhi[4], whi[NC], wlo[NC];                                                                              \
    P16X_LDW(whi, w_lane, (2 * (j)) * 2 * CT) P16X_LDX(xlo, x_lane, 2 * XROW + 2 * (j)) P16X_LDX(xhi, x_lane, 2 * (j)) \
    p16x_wait11<NC>(xlo, xhi, whi);                                                                                        \
    P16X_MM(whi, xlo)                                                                                                  \
    P16X_LDW(wlo, w_lane, (9 + 2 * (j)) * 2 * CT)                                                                      \
    P16X_MM(whi, xhi)                                                                                                  \
    p16x_wait3<NC>(wlo);                                                                                                   \
    P16X_MM(wlo, xhi)                                                                                                  \
    __builtin_amdgcn_sched_barrier(0);                                                                                 \
  }

  int c = 0, cur = 0;
  while (true) {
    const bool last_chunk = (c + 1 == a.nchunks);
    const long ntile = last_chunk ? tile + gridDim.x : tile;
    const int nc = last_chunk ? 0 : c + 1;
    const bool more = ntile < ntiles;
    // ---- phase A: everyone is done with weight taps 4-8 of the previous step.  The finished tile's epilogue first (nothing is in flight)
    if (epi_pos >= 0) {
      epilogue(epi_pos);
      acc_init(tile);
      epi_pos = -1;
    }
    issue_w1();
    const bool skip_tap8 = a.k17 && (c & 1);
    const unsigned xb = p16_lds_addr(smem + cur * XU);
    P16X_PAIR(0) P16X_PAIR(1)
    // ---- phase B: taps 4-8 have landed; everyone is done with taps 0-3 of this step
    __syncthreads();
    if (more) {
      set_src(ntile, nc);
      issue_x(cur ^ 1);
      issue_w0();
    }
    P16X_PAIR(2) P16X_PAIR(3)
    if (!skip_tap8) {
      // tap 8 (no partner): hi x hi with the upper half of K zeroed, then the two cross products in one instruction.  Input: hi plane for
      // every lane group | lo (g < 2), hi (g >= 2); weights: hi (g >= 2 zeroed) | hi (g < 2), lo (g >= 2).  The four lane addresses are
      // derived here from opaque copies (loop-invariant, the compiler would otherwise carry them in registers through the whole loop)
      unsigned xl_ = x_lane, wl_ = w_lane;
      asm volatile("" : "+v"(xl_), "+v"(wl_));
      const unsigned x_lane8 = xl_ - (unsigned)((g >> 1) * 16);
      const unsigned x_lane8m = x_lane8 + (g < 2 ? (unsigned)(2 * XROW * 16) : 0u);
      const unsigned w_lane8 = wl_ - (unsigned)((g >> 1) * 2 * CT * 16);
      const unsigned w_lane8m = w_lane8 + (g < 2 ? 0u : (unsigned)(9 * 2 * CT * 16));
      f16x8 x8[4], x8m[4], w8[NC], w8m[NC];
      P16X_LDW(w8, w_lane8, 8 * 2 * CT) P16X_LDX(x8, x_lane8, 8) P16X_LDX(x8m, x_lane8m, 8)
      p16x_wait11<NC>(x8, x8m, w8);
      if (g >= 2) {
#pragma unroll
        for (int n = 0; n < NC; ++n) w8[n] = (f16x8)(_Float16)0.f;
      }
      P16X_LDW(w8m, w_lane8m, 8 * 2 * CT)
      P16X_MM(w8, x8)
      p16x_wait3<NC>(w8m);
      P16X_MM(w8m, x8m)
    }
    if (last_chunk) epi_pos = tile;
    if (!more) break;
    __syncthreads();          // X and weight taps 0-3 of the next step have landed; everyone is done with this step's X and taps 4-8
    tile = ntile;
    c = nc;
    cur ^= 1;
  }
#undef P16X_PAIR
#undef P16X_LDW
#undef P16X_LDX
#undef P16X_MM
  if (epi_pos >= 0) epilogue(epi_pos);
  if (OM != 2 && vmax > 65504.f && a.flag) *a.flag = 1u;
}
