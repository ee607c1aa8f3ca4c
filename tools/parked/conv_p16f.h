// conv_p16f.h - the planar 16-bit Conv1d k9 (conv_p16.h) as a 2-parallel fast FIR: 14 instead of 18 tap products per pair of outputs.
//
// A 9-tap correlation y[n] = sum_t h[t] x[n + t - 4] over the even / odd phases X0[m] = x[2m], X1[m] = x[2m+1] of the input, with the
// even / odd taps H0[j] = h[2j] (5), H1[j] = h[2j+1] (4):
//     y[2m]   = sum_j H0[j] X0[m-2+j] + sum_j H1[j] X1[m-2+j]
//     y[2m+1] = sum_j H0[j] X1[m-2+j] + sum_j H1[j] X0[m-1+j]
// is four half-rate filters (5 + 4 + 5 + 4 = 18 taps per output pair).  With Hs = H0 + H1 and the input-side differences
//     U0[m] = X0[m] - X1[m],      U2[m] = X0[m+1] - X1[m]
// it is three:   V0 = H0 * U0 (5 taps),  V1 = Hs * X1 (5 taps),  V2 = H1 * U2 (4 taps),   y[2m] = V0[m] + V1[m],   y[2m+1] = V1[m] + V2[m]
// (the transposed form of the 2-parallel FFA: every delay sits on the INPUT side, so the three accumulator sets of a wave are combined
// lane by lane in the epilogue - no shifted accumulators).  22 % fewer MFMAs for the same result; the additions are exact in fp32 and the
// differences are re-split into the same 2 x fp16 operand form (22 bits) the planes themselves have.  Under the package power cap fewer
// matrix instructions is the lever that is left.
//
// Workgroup tile: 256 output PAIRS (512 positions) x CT couts, 8 waves, wave tile 32 pairs x CT: 3 accumulator sets x CT/32 tiles.
// LDS (CT = 96: 160 KB): raw rows X0 | X1 (the step's 16 input channels, de-interleaved by the LDS-DMA: lane k fetches position
// 2k + parity), computed rows U0 | U2, and one buffer per weight part.  A step (16 input channels) runs as three sub-steps, each between
// two bare s_barriers, each with its own weight part [s][tap][g][CT]; every part is fetched TWO sub-steps ahead and retired by a counted
// s_waitcnt vmcnt (the DMA instructions a wave issues per sub-step are wave-uniform constants):
//     B: V1 += Hs * X1   while the waves also compute U0 / U2 of the step from the raw rows (VALU + LDS, hidden from the compiler; the
//                        lower half of the waves before their MFMAs, the upper half after: each SIMD's pair overlaps it with the other's MFMAs)
//                        issues: weight part C of this step
//     A: V0 += H0 * U0   issues: weight part B of the NEXT step, first rows of its raw X (the raw rows are free after B)
//     C: V2 += H1 * U2   issues: the rest of the next step's raw X, its weight part A
// Weight pack ("F14", make_pack_f14 in orca_hip.hip): [cout block][cin/16][part B | A | C][s][tap][g][CT][8] fp16.
// Epilogues (P16 out, MaxPool1d(4) fused, fp32 channel-last; ReLU, residual, fp16 range guard) follow conv_p16.h with position
// 2 * lane + {0, 1} instead of lane.
#pragma once
#include "conv_p16.h"

template <int N, int NW>
__device__ __forceinline__ void p16f_wait(f16x8 (&a)[2], f16x8 (&b)[2][NW]) {
  static_assert(NW == 2 || NW == 3, "operand list below");
  if constexpr (NW == 3)
    asm volatile("s_waitcnt lgkmcnt(%8)" : "+v"(a[0]), "+v"(a[1]), "+v"(b[0][0]), "+v"(b[0][1]), "+v"(b[0][2]), "+v"(b[1][0]), "+v"(b[1][1]), "+v"(b[1][2]) : "n"(N));
  else
    asm volatile("s_waitcnt lgkmcnt(%6)" : "+v"(a[0]), "+v"(a[1]), "+v"(b[0][0]), "+v"(b[0][1]), "+v"(b[1][0]), "+v"(b[1][1]) : "n"(N));
}
__device__ __forceinline__ f16x8 p16f_lds_read16m(unsigned addr, const int off) {
  f16x8 r;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "i"(off) : "memory");
  return r;
}
__device__ __forceinline__ void p16f_lds_write16(unsigned addr, u32x4_t v, const int off) {
  asm volatile("ds_write_b128 %0, %1 offset:%2" : : "v"(addr), "v"(v), "i"(off) : "memory");
}
__device__ __forceinline__ void p16f_wait6(f16x8& a, f16x8& b, f16x8& c, f16x8& d, f16x8& e, f16x8& f) {
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f));
}
__device__ __forceinline__ float p16f_dpp_pair_max(float v) {   // max with the other lane of the pair (2i, 2i+1)
#if defined(__HIP_DEVICE_COMPILE__)
  const float t = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, true));   // quad_perm [1,0,3,2]
  v = p16_vmax(v, t);
#endif
  return v;
}

// ABL (micro-benchmark only, tools/microbench_p16f.hip): 1 = no DMA after the prologue, 2 = no U0 / U2 producer, 4 = no epilogue stores,
// 8 = one barrier per step instead of three (results wrong, timing only)
template <int CT, int OM, bool R1, int ABL = 0>
__global__ __launch_bounds__(512, 2) void conv1d_k9_p16f_kernel(ConvP16Args a) {
  constexpr int NW = CT / 32, NT = 512, PT = 256, MT = 2 * PT;
  constexpr int XR = PT + 8;                 // row pitch: pairs m0-2 .. m0+261 (260 / 261 used)
  constexpr int RS = 4 * XR;                 // one row set [s][g][XR]
  constexpr int NU = PT + 4;                 // U0 / U2 entries computed per g
  constexpr int WP5 = 5 * 4 * CT, WP4 = 4 * 4 * CT;   // units of a 5-tap / 4-tap weight part [s][tap][g][CT]
  constexpr int WCH = 2 * WP5 + WP4;         // units of a step's three parts (B | A | C)
  constexpr int XIT = (2 * RS + NT - 1) / NT;
  constexpr int XSPLIT = 3;                  // raw-X DMA rounds issued in sub-step A (the rest in C)
  static_assert(XIT == 5 && 4 * NT < 2 * RS && 2 * RS - 4 * NT == 64, "raw-X DMA: 4 full rounds + one wave");
  constexpr int W5R = (WP5 + NT - 1) / NT, W5X = (WP5 - (W5R - 1) * NT) / 64;   // rounds of a 5-tap part; waves active in the last one
  constexpr int W4R = WP4 / NT;
  static_assert(WP4 % NT == 0 && WP5 % 64 == 0, "4-tap part: whole rounds");
  __shared__ f32x4 smem[2 * RS + 2 * RS + 3 * WP5 + 32];
  f32x4* const Xraw = smem;                  // X0 rows | X1 rows
  f32x4* const Urow = smem + 2 * RS;         // U0 rows | U2 rows
  f32x4* const Wb = smem + 4 * RS;           // part buffers: B | A | C
  float* const bias_s = reinterpret_cast<float*>(smem + 4 * RS + 3 * WP5);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, g = lane >> 5;
  const int ncb = a.cout / CT;
  const long ntiles = a.tiles_per_row * ncb;
  long tile = blockIdx.x;
  if (tile >= ntiles) return;
  if (tid < a.cout) bias_s[tid] = a.bias[tid];

  // ---- thread-constant DMA geometry of the raw rows: image unit i = (parity*4 + s*2 + gg) * XR + col  <-  plane gg*2+s, position 2 col + parity
  int xrel[XIT];
#pragma unroll
  for (int it = 0; it < XIT; ++it) {
    const int i = tid + it * NT;
    const int ii = i < 2 * RS ? i : 0;       // (round 4: wave 0 only, see issue_x)
    const int row = ii / XR, col = ii - row * XR;
    const int par = row >> 2, s = (row >> 1) & 1, gg = row & 1;
    xrel[it] = (ABL & 16) ? (int)((gg * 2 + s) * a.x_plen) + col + par * XR : (int)((gg * 2 + s) * a.x_plen) + 2 * col + par;   // (16: timing of a contiguous fetch)
  }
  const f32x4 *xsrc = nullptr, *wsrc = nullptr;
  auto set_src = [&](long t, int c) {
    long pos = t;
    int cb = 0;
    while (pos >= a.tiles_per_row) { pos -= a.tiles_per_row; ++cb; }
    xsrc = a.x + (long)c * 4 * a.x_plen + pos * MT + (P16_GUARD - 4);          // pair m0 - 2 = position m0*2 - 4
    wsrc = a.w + ((long)cb * a.nchunks + c) * WCH;
  };
  auto issue_x = [&](int it0, int it1) {
#pragma unroll
    for (int it = 0; it < XIT; ++it)
      if (!(ABL & 1) && it >= it0 && it < it1 && (it < 4 || wave == 0)) p16_glds16(xsrc + xrel[it], Xraw + it * NT + wave * 64);
  };
  auto issue_w = [&](int part_off, int units, int buf) {   // (constants after inlining)
#pragma unroll
    for (int it = 0; it < (WP5 + NT - 1) / NT; ++it) {
      const int k0 = it * NT + wave * 64;                 // wave-uniform
      if (!(ABL & 1) && k0 < units) p16_glds16(wsrc + part_off + k0 + lane, Wb + buf * WP5 + k0);   // wave-uniform
    }
  };

  // ---- thread-constant geometry of the U0 / U2 producer: item i = gg * NU + k
  unsigned u_src[2], u_dst[2];
  bool u_act[2];
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int i = tid + it * NT;
    u_act[it] = i < 2 * NU;
    const int ii = u_act[it] ? i : 0;
    const int gg = ii / NU, k = ii - gg * NU;
    u_src[it] = p16_lds_addr(Xraw + gg * XR + k);        // + (s*2*XR [+ RS for X1] [+ 1 for the next pair]) * 16
    u_dst[it] = p16_lds_addr(Urow + gg * XR + k);        // + (s*2*XR [+ RS for U2]) * 16
  }

  f32x16 V0[NW], V1[NW], V2[NW];
  float vmax = 0.f, umax = 0.f;
  const long xpl16 = a.x_plen * 16, ypl16 = a.y_plen * 16;
  long epi_tile = -1;

  const unsigned bias_lds = p16_lds_addr(bias_s + 4 * g);
  auto acc_init = [&](long t) {   // V1 = bias (it is part of both phases), V0 = V2 = 0.  asm reads: a DMA is in flight here
    int cb = 0;
    long pos = t;
    while (pos >= a.tiles_per_row) { pos -= a.tiles_per_row; ++cb; }
    const unsigned b0 = bias_lds + (unsigned)(cb * CT * 4);
#pragma unroll
    for (int j = 0; j < NW; ++j) {
      f32x4 b[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) b[q] = p16_lds_read16f(b0, (j * 32 + 8 * q) * 4);
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]));
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        V1[j][4 * q + 0] = b[q].x; V1[j][4 * q + 1] = b[q].y; V1[j][4 * q + 2] = b[q].z; V1[j][4 * q + 3] = b[q].w;
#pragma unroll
        for (int r = 0; r < 4; ++r) { V0[j][4 * q + r] = 0.f; V2[j][4 * q + r] = 0.f; }
      }
    }
  };

  auto epilogue = [&](long t) {
    int cb = 0;
    long pos = t;
    while (pos >= a.tiles_per_row) { pos -= a.tiles_per_row; ++cb; }
    const long p0 = pos * MT + wave * 64;                  // + 2 l31 + e per lane
    // residual: the lane loads whole units of plane (hi | lo by g) and trades halves with lane l +- 32 (conv_p16.h); the loads of cout
    // tile j + 1 are in flight while tile j is finished
    u32x4_t rr[2][R1 ? 8 : 1];
    auto load_res = [&](int j) {
      if (R1) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            const int co = cb * CT + j * 32 + 8 * q;
            const char* rb = reinterpret_cast<const char*>(a.r1) + (long)(co >> 3) * 2 * xpl16 + (P16_GUARD + p0 + 2 * l31 + e) * 16 + (g ? xpl16 : 0);
            rr[j & 1][q * 2 + e] = *reinterpret_cast<const u32x4_t*>(rb);
          }
      }
    };
    load_res(0);
#pragma unroll
    for (int j = 0; j < NW; ++j) {
      if (j + 1 < NW) load_res(j + 1);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int co = cb * CT + j * 32 + 8 * q;            // + 4 g per lane
        f32x4 v[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const f32x16& S = e == 0 ? V0[j] : V2[j];
          v[e].x = S[4 * q + 0] + V1[j][4 * q + 0]; v[e].y = S[4 * q + 1] + V1[j][4 * q + 1];
          v[e].z = S[4 * q + 2] + V1[j][4 * q + 2]; v[e].w = S[4 * q + 3] + V1[j][4 * q + 3];
          if (a.relu) { v[e].x = p16_vmax(v[e].x, 0.f); v[e].y = p16_vmax(v[e].y, 0.f); v[e].z = p16_vmax(v[e].z, 0.f); v[e].w = p16_vmax(v[e].w, 0.f); }
          if (R1) {
            const u32x4_t u = rr[j & 1][q * 2 + e];
            unsigned ux = u.x, uy = u.y, uz = u.z, uw = u.w;
            p16_swap32(ux, uz);
            p16_swap32(uy, uw);
            const f16x2 h0 = __builtin_bit_cast(f16x2, ux), h1 = __builtin_bit_cast(f16x2, uy);
            const f16x2 l0 = __builtin_bit_cast(f16x2, uz), l1 = __builtin_bit_cast(f16x2, uw);
            v[e].x += (float)h0.x + (float)l0.x; v[e].y += (float)h0.y + (float)l0.y;
            v[e].z += (float)h1.x + (float)l1.x; v[e].w += (float)h1.y + (float)l1.y;
          }
        }
        if (OM == 2) {
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            const long p = p0 + 2 * l31 + e;
            if (p < a.n) *reinterpret_cast<f32x4*>(reinterpret_cast<char*>(a.y) + (p * a.cout + co) * 4 + g * 16) = v[e];
          }
        } else if (OM == 1) {   // MaxPool1d(4): positions 4i .. 4i+3 = both phases of the lane pair (2i, 2i+1)
          f32x4 m;
          m.x = p16f_dpp_pair_max(p16_vmax(v[0].x, v[1].x)); m.y = p16f_dpp_pair_max(p16_vmax(v[0].y, v[1].y));
          m.z = p16f_dpp_pair_max(p16_vmax(v[0].z, v[1].z)); m.w = p16f_dpp_pair_max(p16_vmax(v[0].w, v[1].w));
          vmax = p16_vmax3_abs(p16_vmax3_abs(vmax, m.x, m.y), m.z, m.w);
          unsigned h0, h1, l0, l1;
          p16_split_hl(m, h0, h1, l0, l1);
          p16_swap32(h0, l0);
          p16_swap32(h1, l1);
          char* yb = reinterpret_cast<char*>(a.y) + (long)(co >> 3) * 2 * ypl16 + (g ? ypl16 : 0) + (P16_GUARD + (p0 >> 2) + (l31 >> 1)) * 16;
          u32x2 d;
          d.x = (l31 & 1) ? l0 : h0;     // the pair's lanes store 8 bytes each of the pooled unit {h0, h1, l0, l1} = couts 0-3 | 4-7
          d.y = (l31 & 1) ? l1 : h1;
          *reinterpret_cast<u32x2*>(yb + (l31 & 1) * 8) = d;
        } else {
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            vmax = p16_vmax3_abs(p16_vmax3_abs(vmax, v[e].x, v[e].y), v[e].z, v[e].w);
            unsigned h0, h1, l0, l1;
            p16_split_hl(v[e], h0, h1, l0, l1);
            p16_swap32(h0, l0);    // g=0: {h, l} = hi halves of couts 0-3 | 4-7;  g=1: the lo halves
            p16_swap32(h1, l1);
            u32x4_t unit;
            unit.x = h0; unit.y = h1; unit.z = l0; unit.w = l1;
            char* yb = reinterpret_cast<char*>(a.y) + (long)(co >> 3) * 2 * ypl16 + (g ? ypl16 : 0);
            if (ABL & 4) asm volatile("" ::"v"(unit)); else
            *reinterpret_cast<u32x4_t*>(yb + (P16_GUARD + p0 + 2 * l31 + e) * 16) = unit;
          }
        }
      }
    }
  };

  // U0 = X0 - X1, U2 = X0(next pair) - X1 of the step's raw rows, exact in fp32, re-split hi / lo.  All LDS traffic from inline asm
  // (an LDS access the compiler can see while an LDS-DMA is in flight is guarded with vmcnt(0), conv_p16.h)
  auto produce_u = [&]() {
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      if (!u_act[it]) continue;
      f16x8 ah = p16f_lds_read16m(u_src[it], 0), al = p16f_lds_read16m(u_src[it], 2 * XR * 16);
      f16x8 bh = p16f_lds_read16m(u_src[it], RS * 16), bl = p16f_lds_read16m(u_src[it], (RS + 2 * XR) * 16);
      f16x8 ch = p16f_lds_read16m(u_src[it], 16), cl = p16f_lds_read16m(u_src[it], (2 * XR + 1) * 16);
      p16f_wait6(ah, al, bh, bl, ch, cl);
      u32x4_t o[4];   // U0 hi, U0 lo, U2 hi, U2 lo
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {
        f32x4 u0, u2;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float x0 = (float)ah[4 * hf + e] + (float)al[4 * hf + e];
          const float x1 = (float)bh[4 * hf + e] + (float)bl[4 * hf + e];
          const float xn = (float)ch[4 * hf + e] + (float)cl[4 * hf + e];
          u0[e] = x0 - x1;
          u2[e] = xn - x1;
        }
        umax = p16_vmax3_abs(p16_vmax3_abs(umax, u0.x, u0.y), u0.z, u0.w);
        umax = p16_vmax3_abs(p16_vmax3_abs(umax, u2.x, u2.y), u2.z, u2.w);
        unsigned h0, h1, l0, l1;
        p16_split_hl(u0, h0, h1, l0, l1);
        o[0][2 * hf] = h0; o[0][2 * hf + 1] = h1; o[1][2 * hf] = l0; o[1][2 * hf + 1] = l1;
        p16_split_hl(u2, h0, h1, l0, l1);
        o[2][2 * hf] = h0; o[2][2 * hf + 1] = h1; o[3][2 * hf] = l0; o[3][2 * hf + 1] = l1;
      }
      p16f_lds_write16(u_dst[it], o[0], 0);
      p16f_lds_write16(u_dst[it], o[1], 2 * XR * 16);
      p16f_lds_write16(u_dst[it], o[2], RS * 16);
      p16f_lds_write16(u_dst[it], o[3], (RS + 2 * XR) * 16);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  };

  // one sub-step: NTAP taps of the part in weight buffer `buf_` on the row set at `rows_` into accumulator set V_
#define P16F_SUBSTEP(rows_, NTAP_, buf_, V_)                                                                           \
  {                                                                                                                    \
    const unsigned xa0 = p16_lds_addr((rows_) + g * XR + wave * 32 + l31);             /* + (s*2*XR + tap) * 16 */         \
    const unsigned wb0 = p16_lds_addr(Wb + (buf_) * WP5 + g * CT + l31);               /* + (((s*NTAP + tap)*2)*CT + j*32) * 16 */ \
    f16x8 av[2][2], bv[2][2][NW];                                                                                      \
    _Pragma("unroll") for (int s = 0; s < 2; ++s) {                                                                    \
      av[0][s] = p16_lds_read16(xa0, (s * 2 * XR) * 16);                                                               \
      _Pragma("unroll") for (int j = 0; j < NW; ++j) bv[0][s][j] = p16_lds_read16(wb0, (((s * NTAP_) * 2) * CT + j * 32) * 16); \
    }                                                                                                                  \
    _Pragma("unroll") for (int tap = 0; tap < NTAP_; ++tap) {                                                          \
      const int fb = tap & 1;                                                                                          \
      if (tap + 1 < NTAP_) {                                                                                           \
        _Pragma("unroll") for (int s = 0; s < 2; ++s) {                                                                \
          av[fb ^ 1][s] = p16_lds_read16(xa0, (s * 2 * XR + tap + 1) * 16);                                            \
          _Pragma("unroll") for (int j = 0; j < NW; ++j) bv[fb ^ 1][s][j] = p16_lds_read16(wb0, (((s * NTAP_ + tap + 1) * 2) * CT + j * 32) * 16); \
        }                                                                                                              \
        p16f_wait<2 * (1 + NW), NW>(av[fb], bv[fb]);                                                                   \
      } else {                                                                                                         \
        p16f_wait<0, NW>(av[fb], bv[fb]);                                                                              \
      }                                                                                                                \
      _Pragma("unroll") for (int p = 0; p < 3; ++p) {                                                                  \
        constexpr int PA[3] = {1, 0, 0}, PB[3] = {0, 1, 0};                                                            \
        _Pragma("unroll") for (int j = 0; j < NW; ++j)                                                                 \
          V_[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bv[fb][PB[p]][j], av[fb][PA[p]], V_[j], 0, 0, 0);             \
      }                                                                                                                \
      __builtin_amdgcn_sched_barrier(0);                                                                               \
    }                                                                                                                  \
  }

  // bare barrier + explicit waits: __syncthreads() is a fence the compiler implements as vmcnt(0) lgkmcnt(0), which would retire the
  // parts fetched two sub-steps ahead at every barrier (conv2d_m16.h)
#define P16F_BARRIER() asm volatile("s_barrier" ::: "memory")
#define P16F_WAIT_VM(n_) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(n_) : "memory")
  // ---- prologue: raw X of the first step, weight parts B and A
  set_src(tile, 0);
  issue_x(0, XIT);
  issue_w(0, WP5, 0);
  issue_w(WP5, WP5, 1);
  P16F_WAIT_VM(0);
  P16F_BARRIER();
  acc_init(tile);

  int c = 0;
  while (true) {
    const bool last_chunk = (c + 1 == a.nchunks);
    const long ntile = last_chunk ? tile + gridDim.x : tile;
    const int nc = last_chunk ? 0 : c + 1;
    const bool more = ntile < ntiles;
    // ---- sub-step B.  The finished tile's epilogue first
    if (epi_tile >= 0) {
      __builtin_amdgcn_s_setprio(3);
      epilogue(epi_tile);
      acc_init(tile);
      __builtin_amdgcn_s_setprio(0);
      epi_tile = -1;
    }
    issue_w(2 * WP5, WP4, 2);                    // part C of this step: W4R instructions per wave
    if (!(ABL & 2) && wave < 4) produce_u();
    P16F_SUBSTEP(Xraw + RS, 5, 0, V1);
    if (!(ABL & 2) && wave >= 4) produce_u();
    if (ABL & 1) P16F_WAIT_VM(0); else P16F_WAIT_VM(W4R);   // part A landed (and the epilogue's stores); own U0 / U2 entries written
    if (!(ABL & 8)) P16F_BARRIER();              // everyone is done with the raw rows and with part B
    // ---- sub-step A
    if (more) {
      set_src(ntile, nc);
      issue_w(0, WP5, 0);                        // part B of the next step: W5R (waves < W5X) or W5R - 1 instructions
      issue_x(0, XSPLIT);                        // XSPLIT instructions
    }
    P16F_SUBSTEP(Urow, 5, 1, V0);
    if (!more || (ABL & 1)) P16F_WAIT_VM(0);     // part C landed
    else if (wave < W5X) P16F_WAIT_VM(W5R + XSPLIT);
    else P16F_WAIT_VM(W5R - 1 + XSPLIT);
    if (!(ABL & 8)) P16F_BARRIER();              // everyone is done with part A
    // ---- sub-step C
    if (more) {
      issue_x(XSPLIT, XIT);
      issue_w(WP5, WP5, 1);                      // part A of the next step
    }
    P16F_SUBSTEP(Urow + RS, 4, 2, V2);
    if (last_chunk) epi_tile = tile;
    if (!more) break;
    if (ABL & 1) P16F_WAIT_VM(0);                // the next step's raw rows and part B landed
    else if (wave < W5X) P16F_WAIT_VM(W5R);
    else P16F_WAIT_VM(W5R - 1);
    P16F_BARRIER();                              // everyone is done with U0 / U2 and part C
    tile = ntile;
    c = nc;
  }
#undef P16F_BARRIER
#undef P16F_WAIT_VM
#undef P16F_SUBSTEP
  if (epi_tile >= 0) epilogue(epi_tile);
  if ((vmax > 65504.f || umax > 65504.f) && a.flag) *a.flag = 1u;   // (vmax stays 0 with fp32 output)
}
