// conv_p16w1.h - the planar 16-bit Conv1d k9 (conv_p16.h) for the 96-cout layers of the Encoder's stage 2 on 512-POSITION tiles.
//
// Why: conv1d_k9_p16_kernel double-buffers the whole (X, W) image of a step; with 96 couts the weight image of a 16-channel step is
// 55 KB, so only 256-position tiles fit (2 x (17 + 55) = 144 KB) - a 32 x 96 wave tile (0.89 LDS operand reads per MFMA) and the weight
// image re-streamed from L2 for every 256 positions.  Timing ablations of that kernel (tools/microbench_p16_96.hip, 96 -> 96 at n = 8 M):
// 3.47 ms, of which the DMA is 17 %, the stores 10 %, the LDS operand reads 4 %; the same K on the 64-cout tile (512 positions) runs 7 %
// faster per FLOP.
// Here: 512-position tiles (64 x 96 wave tile: 0.56 reads per MFMA), the X image double-buffered (2 x 33 KB) and ONE weight buffer (55 KB)
// that is refilled half by half behind the waves that are done with it - two barriers per step (which is twice as long, so as often as before):
//   step s:  [barrier A]  issue DMA of the weight taps 5-8 of step s           (everyone is done with taps 5-8 of step s-1)
//            taps 0-4     (X(s), weight taps 0-4 of step s)
//            [barrier B]  vmcnt(0): taps 5-8 have landed; issue DMA of X(s+1) and of the weight taps 0-4 of step s+1 (everyone is done with taps 0-4)
//            taps 5-8
//            [barrier A]  vmcnt(0): X(s+1) and taps 0-4 of step s+1 have landed ...
// Both waits are plain vmcnt(0): whatever is in flight at a barrier is exactly what the next phase needs.  DMA bytes per matrix-pipe cycle:
// 88 KB per 10 368 cycles against 72 KB per 5 184.  Epilogues, formats (P16 / B16), the 17-tap form (k17) and the argument block are those of
// conv_p16.h.
#pragma once
#include "conv_p16.h"

template <int N>
__device__ __forceinline__ void p16w1_wait(f16x8 (&a)[2][2], f16x8 (&b)[2][3]) {
  asm volatile("s_waitcnt lgkmcnt(%10)"
               : "+v"(a[0][0]), "+v"(a[0][1]), "+v"(a[1][0]), "+v"(a[1][1]), "+v"(b[0][0]), "+v"(b[0][1]), "+v"(b[0][2]), "+v"(b[1][0]), "+v"(b[1][1]), "+v"(b[1][2])
               : "n"(N));
}

template <int OM, bool R1, int FMT>
__global__ __launch_bounds__(512, 2) void conv1d_k9_p16w1_kernel(ConvP16Args a) {
  constexpr int CT = 96, MW = 2, NW = 3, WM = 8;
  constexpr int NT = WM * 64, MT = WM * MW * 32;
  constexpr int XROW = MT + 8;
  constexpr int XU = 2 * 2 * XROW;          // X image units [s][g][XROW]
  constexpr int WU = 2 * 9 * 2 * CT;        // W image units [s][tap][g][CT]
  constexpr int WSPLIT = 5;                 // taps 0 .. 4 | 5 .. 8
  constexpr int WH0 = 2 * WSPLIT * 2 * CT;  // 1920 units
  constexpr int WH1 = WU - WH0;             // 1536 units
  constexpr int XIT = (XU + NT - 1) / NT, W0IT = (WH0 + NT - 1) / NT, W1IT = (WH1 + NT - 1) / NT;
  constexpr int NG = MW * NW * 4;
  constexpr bool F1 = false, RL = false;    // (names the shared epilogue macros look at)
  constexpr int ABL = 0;
  static_assert((WSPLIT * 2 * CT) % 64 == 0 && ((9 - WSPLIT) * 2 * CT) % 64 == 0, "a wave's 64 DMA lanes never straddle the split planes");
  __shared__ f32x4 smem[2 * XU + WU + 32];
  f32x4* const Wl = smem + 2 * XU;
  (void)F1; (void)RL; (void)ABL;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, g = lane >> 5;
  const long ntiles = a.tiles_per_row;      // one cout block
  long tile = blockIdx.x;
  if (tile >= ntiles) return;
  float* bias_s = reinterpret_cast<float*>(smem + 2 * XU + WU);
  if (tid < a.cout) bias_s[tid] = a.bias[tid];

  // thread-constant DMA geometry
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
  // weight halves: dense index k of a half -> unit u of the image (= its offset in the pack's K-chunk: [s][tap][g][cout] with cout = CT)
  auto w0_unit = [](int k) { return k < WSPLIT * 2 * CT ? k : k + (9 - WSPLIT) * 2 * CT; };                      // s = 0: taps 0..4 | s = 1
  auto w1_unit = [](int k) { return k < (9 - WSPLIT) * 2 * CT ? k + WSPLIT * 2 * CT : k + 2 * WSPLIT * 2 * CT; };
  const long wchunk = (long)WU;
  const f32x4 *xsrc = nullptr, *wsrc = nullptr;
  auto set_src = [&](long pos, int c) {
    const int cx = a.k17 ? (c >> 1) : c;
    const int xo = a.k17 ? ((c & 1) ? 9 : 0) : (P16_GUARD - P16_HALO);
    xsrc = a.x + (long)cx * 4 * a.x_plen + pos * MT + xo;
    wsrc = a.w + (long)c * wchunk;
  };
  auto issue_x = [&](int buf) {
#pragma unroll
    for (int it = 0; it < XIT; ++it)
      if (xact[it]) p16_glds16(xsrc + xrel[it], smem + buf * XU + it * NT + wave * 64);
  };
  auto issue_w0 = [&]() {
#pragma unroll
    for (int it = 0; it < W0IT; ++it) {
      const int k0 = it * NT + wave * 64;        // wave-uniform
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

  f32x16 acc[MW][NW];
#define P16_EPI_CB 0
#define P16_EPI_M0 (epi_pos * MT + wave * (MW * 32))
#define P16_EPI_HOOK()
  const int quad_r = l31 & 3;
  const long xpl16 = a.x_plen * 16, ypl16 = a.y_plen * 16;
  const unsigned lane_unit = (unsigned)(l31 * 16) + (g ? (unsigned)ypl16 : 0u);
  const unsigned lane_res = (unsigned)(l31 * 16) + (g ? (unsigned)xpl16 : 0u);
  const unsigned lane_pool = (unsigned)((l31 >> 2) * 16 + quad_r * 4) + (g ? (unsigned)ypl16 : 0u);
  const unsigned lane_f32 = (unsigned)(l31 * a.cout * 4 + g * 16);
  float vmax = 0.f;
  long epi_pos = -1;

  set_src(tile, 0);
  issue_x(0);
  issue_w0();
  __syncthreads();            // X(0), weight taps 0-4 of the first step, the bias
  P16_ACC_INIT(0);

  int c = 0, cur = 0;
  while (true) {
    const bool last_chunk = (c + 1 == a.nchunks);
    const long ntile = last_chunk ? tile + gridDim.x : tile;
    const int nc = last_chunk ? 0 : c + 1;
    const bool more = ntile < ntiles;
    // ---- phase A: everyone is done with weight taps 5-8 of the previous step.  The finished tile's epilogue first (no DMA in flight:
    // the compiler guards the LDS reads it can see - the bias - with vmcnt(0) otherwise), then taps 5-8 of THIS step (wsrc points at it)
    if (epi_pos >= 0) {
      __builtin_amdgcn_s_setprio(3);
      P16_EPILOGUE();
      P16_ACC_INIT(0);
      __builtin_amdgcn_s_setprio(0);
      epi_pos = -1;
    }
    issue_w1();
    const bool skip_tap8 = a.k17 && (c & 1);
    const unsigned xa0 = p16_lds_addr(smem + cur * XU + g * XROW + wave * (MW * 32) + l31);   // + (s*2*XROW + i*32 + tap)*16
    const unsigned wb0 = p16_lds_addr(Wl + g * CT + l31);                                      // + (((s*9+tap)*2)*CT + j*32)*16
    f16x8 av[2][2][MW], bv[2][2][NW];
#define W1_READ_FRAGS(buf_, tap_)                                                                                  \
  _Pragma("unroll") for (int s = 0; s < 2; ++s) {                                                                  \
    _Pragma("unroll") for (int i = 0; i < MW; ++i) av[buf_][s][i] = p16_lds_read16(xa0, (s * 2 * XROW + i * 32 + (tap_)) * 16); \
    _Pragma("unroll") for (int j = 0; j < NW; ++j) bv[buf_][s][j] = p16_lds_read16(wb0, (((s * 9 + (tap_)) * 2) * CT + j * 32) * 16); \
  }
#define W1_MFMAS(fb_)                                                                                              \
  if constexpr (FMT == 1) {                                                                                        \
    _Pragma("unroll") for (int p = 0; p < 2; ++p) _Pragma("unroll") for (int i = 0; i < MW; ++i) _Pragma("unroll") for (int j = 0; j < NW; ++j) \
      acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, bv[fb_][p][j]), __builtin_bit_cast(bf16x8, av[fb_][p][i]), acc[i][j], 0, 0, 0); \
  } else {                                                                                                         \
    _Pragma("unroll") for (int p = 0; p < 3; ++p) {                                                                \
      constexpr int PA[3] = {1, 0, 0}, PB[3] = {0, 1, 0};                                                          \
      _Pragma("unroll") for (int i = 0; i < MW; ++i) _Pragma("unroll") for (int j = 0; j < NW; ++j)                \
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bv[fb_][PB[p]][j], av[fb_][PA[p]][i], acc[i][j], 0, 0, 0); \
    }                                                                                                              \
  }
    W1_READ_FRAGS(0, 0);
#pragma unroll
    for (int tap = 0; tap < WSPLIT; ++tap) {
      const int fb = tap & 1;
      if (tap + 1 < WSPLIT) { W1_READ_FRAGS(fb ^ 1, tap + 1); p16w1_wait<2 * (MW + NW)>(av[fb], bv[fb]); }
      else p16w1_wait<0>(av[fb], bv[fb]);
      W1_MFMAS(fb);
      __builtin_amdgcn_sched_barrier(0);
    }
    // ---- phase B: taps 5-8 have landed (and the epilogue's stores are retired); everyone is done with taps 0-4 of this step
    __syncthreads();
    if (more) {
      set_src(ntile, nc);
      issue_x(cur ^ 1);
      issue_w0();
    }
    W1_READ_FRAGS(1, WSPLIT);     // (WSPLIT = 5 is odd: buffer 1 continues the alternation)
#pragma unroll
    for (int tap = WSPLIT; tap < 9; ++tap) {
      const int fb = tap & 1;
      if (tap + 1 < 9) { W1_READ_FRAGS(fb ^ 1, tap + 1); p16w1_wait<2 * (MW + NW)>(av[fb], bv[fb]); }
      else p16w1_wait<0>(av[fb], bv[fb]);
      if (!(tap == 8 && skip_tap8)) { W1_MFMAS(fb); }
      __builtin_amdgcn_sched_barrier(0);
    }
#undef W1_READ_FRAGS
#undef W1_MFMAS
    if (last_chunk) epi_pos = tile;
    if (!more) break;
    __syncthreads();          // X and weight taps 0-4 of the next step have landed; everyone is done with this step's X and taps 5-8
    tile = ntile;
    c = nc;
    cur ^= 1;
  }
  if (epi_pos >= 0) P16_EPILOGUE();
#undef P16_EPI_CB
#undef P16_EPI_M0
#undef P16_EPI_HOOK
  if (FMT == 0 && OM != 2 && vmax > 65504.f && a.flag) *a.flag = 1u;
}
