// conv_ws.h - W-STATIONARY, BARRIER-FREE form of the planar 16-bit Conv1d k9 (conv_p16.h) for layers whose weight block
// fits in LDS next to the input rings: cin = 64 with 64 couts per workgroup in B16 (73.7 KB) or 32 couts in P16 (73.7 KB).
//
// Why: in conv1d_k9_p16_kernel the eight waves of a workgroup share one (X, W) image per step, so every step ends in
// s_waitcnt vmcnt(0) + s_barrier.  s_memtime stamps (tools/microbench_b16p.hip, B16 64 -> 64, n = 32 M): a plain step
// lasts 7 270 cycles of which the matrix pipe needs 4 608 - the SIMD's older wave leaves its MFMA block at 5 000, the
// younger at 6 300 (alone on the SIMD it runs at 73 %: the read / DMA issue slots of one in-order wave are not covered),
// then DMA drain + barrier + the next step's first fragment reads; and the tile's epilogue (1 400 - 1 600 cycles of VALU
// and stores) runs with the pipe idle on BOTH waves of each SIMD at once.
// Here the weights of ALL K-chunks of the workgroup's cout block are loaded ONCE per (persistent) workgroup and every
// WAVE owns its tiles of MW*32 positions outright: a private two-slot LDS ring for its input slices (4 planes x (MW*32 + 8)
// units per step, LDS-DMA'd by the wave itself, halo included) and its own accumulators.  After the single barrier
// behind the weight load there is NO workgroup-level synchronisation: the two waves of a SIMD drift apart, one wave's
// epilogue, DMA issue and LDS waits are covered by the other's MFMAs, and a slice's DMA is retired by the wave's own
// s_waitcnt vmcnt(0) a whole step after it was issued.
// Measured (tools/microbench_b16p.hip, B16 64 -> 64 at n = 32 M, random operands): 2.09 ms against 2.26 for the barrier
// kernel - the matrix pipe is busy 68 % instead of 57 % of the cycles, but the part then clocks at 1.47-1.50 GHz instead
// of 1.75-1.77 (s_memtime / s_memrealtime inside the kernel): both kernels sit on the package power limit, ~1.1 PFLOP/s of
// MFMA work on random data next to 4 TB/s of HBM and 36 TB/s of LDS operand traffic.  Data movement alone (MFMAs
// ablated) takes 1.79 ms (reads 0.70 + stores 1.06: 1 KB store segments in 8 planes 512 MB apart run at 3.8 TB/s; the same
// stores in tile-major order 0.68 ms - but the full kernel does not get faster with them).  Tried and dropped: a dummy
// 40-lane global_load_dword per step that pulls the slice two steps ahead into L2 (+4 %: VMEM issue slots cost more
// than the shorter vmcnt wait returns).  In P16 (32 couts per workgroup: the input is read once per cout block and the
// wave tile is 64 x 32) this form is 3 % SLOWER than the barrier kernel (6.25 vs 6.10 ms) - the library uses it for B16 only.
// Cost: the halo columns are fetched per wave (72 / 64 units) and, with cout blocks, once per block (L2 / MALL hits:
// the workgroups of the blocks of one position range sit on the same XCD and run in step).
#pragma once
#include "conv_p16.h"

template <int FMT, int CIN, int CT, int MW, int NW, int OM, bool R1, int ABL = 0>
__global__ __launch_bounds__(512, 2) void conv1d_k9_ws_kernel(ConvP16Args a) {
  static_assert(NW * 32 == CT, "one wave covers all couts of the workgroup's block");
  constexpr int WM = 8, NT = WM * 64;
  constexpr int KC = FMT == 1 ? 32 : 16;       // input channels per step
  constexpr int NCH = CIN / KC;                // steps per tile
  constexpr int MTW = MW * 32;                 // positions per wave tile
  constexpr int XW = MTW + 8;                  // columns per plane of a slice
  constexpr int SLOT = 4 * XW;                 // units per ring slot: [s][g][XW]
  constexpr int NP = (SLOT + 63) / 64;         // DMA pieces per slice
  constexpr int WU = NCH * 2 * 9 * 2 * CT;     // resident weight block: [c][s][tap][g][CT]
  constexpr int NG = MW * NW * 4;
  static_assert(NP <= 9, "one DMA piece per tap");
  static_assert((WU + WM * 2 * SLOT) * 16 + 512 <= 160 * 1024, "LDS budget");
  __shared__ f32x4 smem[WU + WM * 2 * SLOT + 32];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, g = lane >> 5;
  // cout block and wave-tile stream of this workgroup (see the launcher: blocks of one position range share an XCD)
  const int ncb = a.cout / CT;
  const int G = gridDim.x, b = blockIdx.x;
  int cb, sb;
  if (G % (8 * ncb) == 0) { cb = (b >> 3) % ncb; sb = ((b >> 3) / ncb) * 8 + (b & 7); }
  else { cb = b % ncb; sb = b / ncb; }
  const long TW = (long)(G / ncb) * WM;        // waves per cout block = stride of a wave's tile sequence
  const long ntw = (a.n + MTW - 1) / MTW;      // wave tiles per cout block
  long tile = (long)sb * WM + wave;

  // ---- resident weights + bias of the cout block ----
  float* bias_s = reinterpret_cast<float*>(smem + WU + WM * 2 * SLOT);
  for (int i = tid; i < WU; i += NT) {
    const int grp = i / CT, co = i - grp * CT;
    smem[i] = a.w[(long)grp * a.cout + cb * CT + co];
  }
  if (tid < CT) bias_s[tid] = a.bias[cb * CT + tid];
  __syncthreads();                             // the only barrier of the kernel
  if (tile >= ntw) return;

  f32x4* const ring = smem + WU + wave * (2 * SLOT);
  // thread-constant DMA geometry: unit u = p*64 + lane of a slice = (plane, col)
  int xrel[NP];
  bool act[NP];
#pragma unroll
  for (int p = 0; p < NP; ++p) {
    const int u = p * 64 + lane;
    act[p] = u < SLOT;
    const int uu = act[p] ? u : 0;
    const int row = uu / XW, col = uu - row * XW;       // row = s*2 + g of the LDS image
    const int s = row >> 1, gg = row & 1;
    xrel[p] = (int)((FMT == 1 ? (s * 2 + gg) : (gg * 2 + s)) * a.x_plen) + col;
  }
  const f32x4* xsrc = nullptr;
#define WS_SRC(t, c) { xsrc = a.x + (long)(c) * 4 * a.x_plen + (t) * MTW + (P16_GUARD - P16_HALO); }
#define WS_DMA_ONE(p, slot) if (act[p]) p16_glds16(xsrc + xrel[p], ring + (slot) * SLOT + (p) * 64);

  f32x16 acc[MW][NW];
  const int quad_r = l31 & 3;
  const long xpl16 = a.x_plen * 16, ypl16 = a.y_plen * 16;
  const unsigned lane_unit = (unsigned)(l31 * 16) + (g ? (unsigned)ypl16 : 0u);
  const unsigned lane_res = (unsigned)(l31 * 16) + (g ? (unsigned)xpl16 : 0u);
  const unsigned lane_pool = (unsigned)((l31 >> 2) * 16 + quad_r * 4) + (g ? (unsigned)ypl16 : 0u);
  const unsigned lane_f32 = (unsigned)(l31 * a.cout * 4 + g * 16);
  float vmax = 0.f;
  long epi_tile = -1;
#define P16_EPI_CB cb
#define P16_EPI_HOOK()
#define P16_EPI_M0 (epi_tile * MTW)
  // the macros of conv_p16.h index the bias and the cout block through `tcb * CT`: bias_s holds only this block's couts
#define WS_ACC_INIT()                                                                                         \
  {                                                                                                           \
    _Pragma("unroll") for (int j = 0; j < NW; ++j) _Pragma("unroll") for (int q = 0; q < 4; ++q) {            \
      const f32x4 b_ = *reinterpret_cast<const f32x4*>(bias_s + 4 * g + j * 32 + 8 * q);                      \
      _Pragma("unroll") for (int i = 0; i < MW; ++i) {                                                        \
        acc[i][j][4 * q + 0] = b_.x; acc[i][j][4 * q + 1] = b_.y; acc[i][j][4 * q + 2] = b_.z; acc[i][j][4 * q + 3] = b_.w; \
      }                                                                                                       \
    }                                                                                                         \
  }

  WS_SRC(tile, 0);
#pragma unroll
  for (int p = 0; p < NP; ++p) WS_DMA_ONE(p, 0);
  WS_ACC_INIT();

  int c = 0, cur = 0;
  int nstamp = 0;
  unsigned long long clk0 = 0, rt0 = 0;
  if (ABL & 128) { clk0 = __builtin_readcyclecounter(); rt0 = __builtin_amdgcn_s_memrealtime(); }
#define WS_STAMP(k) if ((ABL & 128) && blockIdx.x == 0 && lane == 0 && nstamp >= 200 && nstamp < 400) a.stamps[((nstamp - 200) * 8 + wave) * 5 + (k)] = __builtin_readcyclecounter();
  while (true) {
    const bool last_chunk = (c + 1 == NCH);
    const long ntile = last_chunk ? tile + TW : tile;
    const int nc = last_chunk ? 0 : c + 1;
    const bool more = ntile < ntw;
    WS_STAMP(0);
    // this step's slice (DMA issued a step ago) and the stores of the last epilogue are retired; nothing else of this
    // wave is in flight
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    WS_STAMP(2);
    if (epi_tile >= 0) {
      __builtin_amdgcn_s_setprio(3);
      P16_EPILOGUE();
      WS_ACC_INIT();
      __builtin_amdgcn_s_setprio(0);
      epi_tile = -1;
    }
    WS_STAMP(1);
    if (more) WS_SRC(ntile, nc);

    const unsigned xa0 = p16_lds_addr(ring + cur * SLOT + g * XW + l31);     // + (s*2*XW + i*32 + tap)*16
    const unsigned wb0 = p16_lds_addr(smem + (c * 2 * 9 * 2 + g) * CT + l31);  // + (((s*9+tap)*2)*CT + j*32)*16
    f16x8 av[2][2][MW], bv[2][2][NW];
#define WS_READ_FRAGS(buf_, tap_)                                                                                 \
  _Pragma("unroll") for (int s = 0; s < 2; ++s) {                                                                 \
    _Pragma("unroll") for (int i = 0; i < MW; ++i) av[buf_][s][i] = p16_lds_read16(xa0, (s * 2 * XW + i * 32 + (tap_)) * 16); \
    _Pragma("unroll") for (int j = 0; j < NW; ++j) bv[buf_][s][j] = p16_lds_read16(wb0, (((s * 9 + (tap_)) * 2) * CT + j * 32) * 16); \
  }
    WS_READ_FRAGS(0, 0);
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int fb = tap & 1;
      if (tap + 1 < 9 && !(ABL & 8)) WS_READ_FRAGS(fb ^ 1, tap + 1);
      if (more && tap < NP && !(ABL & 1)) WS_DMA_ONE(tap, cur ^ 1);    // the next step's slice, one piece per tap
      if (tap + 1 < 9) p16_lds_wait<2 * (MW + NW), MW, NW>(av[fb], bv[fb]);
      else p16_lds_wait<0, MW, NW>(av[fb], bv[fb]);
      if constexpr ((ABL & 4) != 0) {   // micro-benchmark: data movement only
#pragma unroll
        for (int s = 0; s < 2; ++s) {
#pragma unroll
          for (int i = 0; i < MW; ++i) asm volatile("" ::"v"(av[fb][s][i]));
#pragma unroll
          for (int j = 0; j < NW; ++j) asm volatile("" ::"v"(bv[fb][s][j]));
        }
      } else if constexpr (FMT == 1) {
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
          for (int i = 0; i < MW; ++i)
#pragma unroll
            for (int j = 0; j < NW; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, bv[fb][p][j]), __builtin_bit_cast(bf16x8, av[fb][p][i]), acc[i][j], 0, 0, 0);
      } else {
#pragma unroll
        for (int p = 0; p < 3; ++p) {
          constexpr int PA[3] = {1, 0, 0}, PB[3] = {0, 1, 0};   // lo*hi, hi*lo, hi*hi (largest last)
#pragma unroll
          for (int i = 0; i < MW; ++i)
#pragma unroll
            for (int j = 0; j < NW; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bv[fb][PB[p]][j], av[fb][PA[p]][i], acc[i][j], 0, 0, 0);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
#undef WS_READ_FRAGS
    WS_STAMP(3);
    if (last_chunk) epi_tile = tile;
    if (!more) break;
    if (ABL & 128) ++nstamp;
    tile = ntile;
    c = nc;
    cur ^= 1;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (epi_tile >= 0) P16_EPILOGUE();
  if ((ABL & 128) && blockIdx.x == 0 && tid == 0) {
    a.stamps[8190] = __builtin_readcyclecounter() - clk0;
    a.stamps[8191] = __builtin_amdgcn_s_memrealtime() - rt0;
  }
#undef WS_STAMP
#undef WS_ACC_INIT
#undef WS_DMA_ONE
#undef WS_SRC
#undef P16_EPI_CB
#undef P16_EPI_HOOK
#undef P16_EPI_M0
  if (FMT == 0 && OM != 2 && vmax > 65504.f && a.flag) *a.flag = 1u;
}
