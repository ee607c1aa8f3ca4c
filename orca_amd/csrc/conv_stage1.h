// conv_stage1.h - stage 1 of the Encoder in ONE kernel, straight from the bases, for the throughput mode (B16 planes, config 3) - round 5.
//
// What it replaces (orca_modules.py:811-826): with the linear groups composed, stage 1 of a strand is two launches at
// n = 32 M - `conv1.a o lconv1` as a 25-tap conv from the bases + ReLU (conv1d_first_mfma_p16_kernel<.,1,25>: writes the 64-channel tensor
// a1, 4.1 GB in bf16) and conv1.b (64 -> 64, k9, ReLU, + lout1 computed from the bases in the epilogue, MaxPool1d(4): re-reads a1).  In
// bf16 that pair is bound by those two passes over HBM: conv1.b ran 2.70 ms against 0.94 ms of matrix work.
// Here a1 never exists in HBM.  The kernel is the W-STATIONARY, BARRIER-FREE form of conv_ws.h - conv1.b's 73.7 KB of bf16 weights stay in
// LDS, every WAVE owns 64-position tiles outright - with the LDS-DMA of a wave's input slice replaced by a PRODUCER that computes it:
//   * the wave fetches the 123 bases its tile looks at (1 byte each; reverse complement = index / code flip) into a private 128-byte window;
//   * a1 at the tile's 64 new positions (the 8 halo columns roll over from the wave's previous tile) = a K = 112 GEMM of the composed
//     25-tap weights (fp16 hi plane, 14 KB of LDS) with one-hot operand units
//     built from two window bytes each (K index = tap * 4 + channel) - 28 matrix instructions per tile beside conv1.b's 144 -, bias, ReLU,
//     rounded to bf16 and written into the wave's slice in the B16 operand layout (v_permlane32_swap pairs the two half-units of a lane pair,
//     as every B16 epilogue does);
//   * conv1.b's 2 x 9 taps run out of that slice exactly as in conv_ws.h; its epilogue adds lout1 (K = 80 GEMM from the same window, the
//     17-tap pack's hi plane read through L1), pools by 4 and stores.
// No workgroup barrier after the weight load: while one wave of a SIMD builds operand units (VALU, LDS) the other multiplies.
// What a single convolution cannot express - PyTorch's zero padding of the INTERMEDIATE tensors at the two ends of a chunk - is patched as
// before: the edge-fix chain (lconv_edge_layer_kernel) writes a1's first / last 8 positions into `a1_edge`, the producer takes those units
// from there; positions outside the chunk are conv1.b's own zero padding.
// Arithmetic vs the two-launch form: the 25-tap and 17-tap weights enter with their fp16 hi part, ONE product each (there: hi + lo, two
// products) - a1 and the stage's output are rounded to bf16 right after either way (measured vs exact fp32: the same error); the mode's parity is the one stated for config 3 (tests/test_gpu_config3.py, bench.py `config3.parity`).
#pragma once
#include <type_traits>
#include "conv_ws.h"

struct Stage1Args {
  ConvP16Args c;          // conv1.b: w (bf16 pack), bias, y / y_plen (pooled B16 output), n, cout = 64, relu = 1; the bases in f1_codes ..;
                          // rl_w / f1_bias = the composed 17-tap lconv1 (fp16 split pack, bias) of the residual
  const f32x4* w25;       // fp16 split pack [2 splits][7 k-steps][2 g][64 couts][8] of conv1.a o lconv1 (K = tap * 4 + channel, 100 -> 112)
  const float* b25;       // its bias [64]
  const f32x4* a1_edge;   // planar B16 tensor (plane length a1_plen units) whose first / last 8 positions hold the edge-fixed a1
  long a1_plen;
};

template <int OM>
__global__ __launch_bounds__(512, 2) void conv1d_stage1_b16_kernel(Stage1Args sa) {
  const ConvP16Args& a = sa.c;
  constexpr int FMT = 1, CT = 64, MW = 2, NW = 2, WM = 8, NT = WM * 64, ABL = 0;
  constexpr bool R1 = false;
  constexpr int NCH = 2, MTW = MW * 32, XW = MTW + 8, SLOT = 4 * XW;    // a step's slice: [k-pair s][octet g][XW] units
  constexpr int WU = NCH * 2 * 9 * 2 * CT;                             // conv1.b: [c][s][tap][g][64]
  constexpr int KS = 7, W25U = KS * 2 * 64;                            // 25-tap pack, fp16 hi plane: [k-step][g][64]
  constexpr int NG = MW * NW * 4;
  constexpr int WINB = 128;                                            // bases m0 - 16 .. m0 + 111 of a wave tile at m0
  constexpr int XTRA = (WM * WINB + 3 * 64 * 4 + 8 * 8) / 16;          // windows, three biases, the one-hot table
  static_assert((WU + WM * NCH * SLOT + W25U + XTRA) * 16 <= 160 * 1024, "LDS budget");
  __shared__ f32x4 smem[WU + WM * NCH * SLOT + W25U + XTRA];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, g = lane >> 5;
  // every wave takes a CONTIGUOUS run of wave tiles: the last 8 columns of a tile's slice are the first 8 of the next tile's (the conv's
  // halo), so only a run's first tile produces 72 + positions - the others copy 8 columns and produce 64 (two position tiles instead of three)
  const long ntw = (a.n + MTW - 1) / MTW;
  const long nwv = (long)gridDim.x * WM;
  const long run = (ntw + nwv - 1) / nwv;
  long tile = ((long)blockIdx.x * WM + wave) * run;
  const long tile_end = tile + run < ntw ? tile + run : ntw;

  f32x4* const w25s = smem + WU + WM * NCH * SLOT;
  unsigned char* const xbase = reinterpret_cast<unsigned char*>(smem + WU + WM * NCH * SLOT + W25U);
  unsigned char* const win = xbase + wave * WINB;
  float* const bias_s = reinterpret_cast<float*>(xbase + WM * WINB);   // [0,64) conv1.b | [64,128) 25-tap group | [128,192) lout1
  u32x2* const oh16 = reinterpret_cast<u32x2*>(xbase + WM * WINB + 3 * 64 * 4);   // fp16 one-hot row per base code (4 = N = 0.25 x 4, 5.. = zero)

  // ---- resident weights, biases, tables ---------------------------------------------------------------------------------------------
  for (int i = tid; i < WU; i += NT) smem[i] = a.w[i];
  for (int i = tid; i < W25U; i += NT) w25s[i] = sa.w25[i];       // the hi plane of the fp16 split pack (11 significant bits: below a1's bf16 rounding)
  if (tid < 64) { bias_s[tid] = a.bias[tid]; bias_s[64 + tid] = sa.b25[tid]; bias_s[128 + tid] = a.f1_bias[tid]; }
  if (tid < 8) {
    u32x2 v;
    v.x = tid == 0 ? 0x3C00u : tid == 1 ? 0x3C000000u : tid == 4 ? 0x34003400u : 0u;
    v.y = tid == 2 ? 0x3C00u : tid == 3 ? 0x3C000000u : tid == 4 ? 0x34003400u : 0u;
    oh16[tid] = v;
  }
  __syncthreads();                             // the only barrier of the kernel
  if (tile >= tile_end) return;

  f32x4* const slice = smem + WU + wave * (NCH * SLOT);
  // base code at chunk position p (5 = outside the chunk: the composed first layer's zero padding), reverse complement applied
  auto base_at = [&](long p) -> unsigned char {
    if (p < 0 || p >= a.n) return (unsigned char)5;
    const long P = a.f1_codes_off + p;
    int cc = p16_base_at(a.f1_codes, a.f1_nmask, a.f1_origin, a.f1_reverse ? a.f1_codes_L - 1 - P : P);
    if (a.f1_reverse && cc < 4) cc = 3 - cc;
    return (unsigned char)(cc > 4 ? 5 : cc);
  };

  f32x16 acc[MW][NW];
  const int quad_r = l31 & 3;
  const long xpl16 = a.x_plen * 16, ypl16 = a.y_plen * 16;
  const unsigned lane_unit = (unsigned)(l31 * 16) + (g ? (unsigned)ypl16 : 0u);
  const unsigned lane_res = (unsigned)(l31 * 16) + (g ? (unsigned)xpl16 : 0u);
  const unsigned lane_pool = (unsigned)((l31 >> 2) * 16 + quad_r * 4) + (g ? (unsigned)ypl16 : 0u);
  const unsigned lane_f32 = (unsigned)(l31 * a.cout * 4 + g * 16);
  float vmax = 0.f;
  (void)lane_res; (void)lane_unit; (void)lane_f32; (void)vmax; (void)xpl16;
  long epi_tile = -1;
#define P16_EPI_CB 0
#define P16_EPI_M0 (epi_tile * MTW)
  // + lout1 of the tile's positions, straight from the bases (conv_p16.h, RL): K = 80 GEMM of the composed 17-tap pack - its fp16 hi plane,
  // ONE product (11 significant bits: the sum is rounded to bf16 right after), read through L1 (10 KB that do not fit beside the slices) - with
  // fp16 one-hot units from the wave's window (base of tap t of position p: p + t - 8)
#define P16_EPI_HOOK()                                                                                           \
  {                                                                                                              \
    _Pragma("unroll") for (int i = 0; i < MW; ++i) _Pragma("unroll") for (int j = 0; j < NW; ++j) _Pragma("unroll") for (int q = 0; q < 4; ++q) { \
      const f32x4 b_ = *reinterpret_cast<const f32x4*>(bias_s + 128 + j * 32 + 8 * q + 4 * g);                   \
      acc[i][j][4 * q + 0] += b_.x; acc[i][j][4 * q + 1] += b_.y; acc[i][j][4 * q + 2] += b_.z; acc[i][j][4 * q + 3] += b_.w; \
    }                                                                                                            \
    const unsigned char* win_ = win + 8 + l31 + 2 * g;                                                           \
    f32x4 wa_[3][NW];          /* the pack's units (hi plane) for k-step kk, requested two k-steps ahead (L1 / L2 round trips) */ \
    _Pragma("unroll") for (int k0 = 0; k0 < 2; ++k0) _Pragma("unroll") for (int j = 0; j < NW; ++j)              \
      wa_[k0][j] = a.rl_w[(k0 * 2 + g) * 64 + j * 32 + l31];                                                     \
    _Pragma("unroll") for (int kk = 0; kk < 5; ++kk) {                                                           \
      if (kk + 2 < 5) { _Pragma("unroll") for (int j = 0; j < NW; ++j) wa_[(kk + 2) % 3][j] = a.rl_w[((kk + 2) * 2 + g) * 64 + j * 32 + l31]; } \
      f16x8 xf_[MW];                                                                                             \
      _Pragma("unroll") for (int i = 0; i < MW; ++i) {                                                           \
        const u32x2 o0_ = oh16[win_[i * 32 + 4 * kk]], o1_ = oh16[win_[i * 32 + 4 * kk + 1]];                    \
        u32x4_t u_;                                                                                              \
        u_.x = o0_.x; u_.y = o0_.y; u_.z = o1_.x; u_.w = o1_.y;                                                  \
        xf_[i] = __builtin_bit_cast(f16x8, u_);                                                                  \
      }                                                                                                          \
      _Pragma("unroll") for (int i = 0; i < MW; ++i) _Pragma("unroll") for (int j = 0; j < NW; ++j)              \
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, wa_[kk % 3][j]), xf_[i], acc[i][j], 0, 0, 0); \
    }                                                                                                            \
  }

  // the producer of NPI position tiles: columns col0 + 32 pi + l31 of the slice = a1 at positions m0 - 4 + column
  auto produce = [&](const long m0, auto npi_c, auto col0_c) __attribute__((always_inline)) {
    constexpr int NPI = decltype(npi_c)::value, COL0 = decltype(col0_c)::value;
    f32x16 pacc[2][NPI];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 b_ = *reinterpret_cast<const f32x4*>(bias_s + 64 + j * 32 + 8 * q + 4 * g);
#pragma unroll
        for (int pi = 0; pi < NPI; ++pi) { pacc[j][pi][4 * q + 0] = b_.x; pacc[j][pi][4 * q + 1] = b_.y; pacc[j][pi][4 * q + 2] = b_.z; pacc[j][pi][4 * q + 3] = b_.w; }
      }
    const unsigned char* wp = win + COL0 + l31 + 2 * g;        // base of tap t of a column: window index column + t (t = 4 kk + 2 g, + 1)
#pragma unroll
    for (int kk = 0; kk < KS; ++kk) {
      f16x8 xb[NPI];
#pragma unroll
      for (int pi = 0; pi < NPI; ++pi) {
        const u32x2 o0 = oh16[wp[32 * pi + 4 * kk]], o1 = oh16[wp[32 * pi + 4 * kk + 1]];
        u32x4_t u;
        u.x = o0.x; u.y = o0.y; u.z = o1.x; u.w = o1.y;
        xb[pi] = __builtin_bit_cast(f16x8, u);
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const f16x8 wv = __builtin_bit_cast(f16x8, w25s[(kk * 2 + g) * 64 + j * 32 + l31]);
#pragma unroll
        for (int pi = 0; pi < NPI; ++pi) pacc[j][pi] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wv, xb[pi], pacc[j][pi], 0, 0, 0);
      }
    }
    const bool edge = (m0 - 4 < 8) || (m0 - 4 + 104 > a.n - 8);    // wave-uniform: the tile touches an end of the chunk
#pragma unroll
    for (int pi = 0; pi < NPI; ++pi) {
      const int col = COL0 + 32 * pi + l31;
      const long p = m0 - 4 + col;
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int qp = 0; qp < 2; ++qp) {
          f32x4 v0, v1;
          // (fmaxf, not the inline-asm v_max of the other epilogues: these reads follow the MFMA chain directly, and the compiler only
          // inserts the MFMA -> VALU wait states in front of instructions it can see - with asm reads the first unit of a tile came out
          // wrong in a quarter of the lanes)
          v0.x = fmaxf(pacc[j][pi][8 * qp + 0], 0.f); v0.y = fmaxf(pacc[j][pi][8 * qp + 1], 0.f);
          v0.z = fmaxf(pacc[j][pi][8 * qp + 2], 0.f); v0.w = fmaxf(pacc[j][pi][8 * qp + 3], 0.f);
          v1.x = fmaxf(pacc[j][pi][8 * qp + 4], 0.f); v1.y = fmaxf(pacc[j][pi][8 * qp + 5], 0.f);
          v1.z = fmaxf(pacc[j][pi][8 * qp + 6], 0.f); v1.w = fmaxf(pacc[j][pi][8 * qp + 7], 0.f);
          unsigned a0 = cvt_pk_bf16(v0.x, v0.y), a1 = cvt_pk_bf16(v0.z, v0.w), b0 = cvt_pk_bf16(v1.x, v1.y), b1 = cvt_pk_bf16(v1.z, v1.w);
          asm volatile("s_nop 1" : "+v"(a0), "+v"(a1), "+v"(b0), "+v"(b1));     // v_permlane32_swap: two wait states behind the (inline-asm) conversions
          p16_swap32(a0, b0);     // g = 0: the unit of plane j*4 + 2 qp, g = 1: of plane j*4 + 2 qp + 1 (8 consecutive channels of position p)
          p16_swap32(a1, b1);
          u32x4_t unit;
          unit.x = a0; unit.y = a1; unit.z = b0; unit.w = b1;
          const int P = j * 4 + 2 * qp + g;
          if (edge) {
            if (p < 0 || p >= a.n) unit = (u32x4_t)(0u);                       // conv1.b's own zero padding
            else if (p < 8 || p >= a.n - 8)                                    // the edge-fixed a1 (intermediates zero-padded as PyTorch does)
              unit = *reinterpret_cast<const u32x4_t*>(reinterpret_cast<const char*>(sa.a1_edge) + ((long)P * sa.a1_plen + P16_GUARD + p) * 16);
          }
          if (col < XW) slice[(P >> 2) * SLOT + (P & 3) * XW + col] = __builtin_bit_cast(f32x4, unit);
        }
    }
  };

  // the bases of the wave's NEXT tile travel in two registers through the current one (their round trip used to open every tile)
  unsigned char nb0 = base_at(tile * MTW - 16 + lane), nb1 = base_at(tile * MTW - 16 + 64 + lane);
  bool first = true;
  // the two waves of a SIMD (w, w + 4) start half a tile apart: one builds operand units / runs its epilogue (VALU, LDS) while the other
  // multiplies; started together they spend their first tiles in step - both on the VALU, then both queueing for the matrix pipe
  if (wave >= 4) { __builtin_amdgcn_s_sleep(60); __builtin_amdgcn_s_sleep(60); }
  for (; tile < tile_end; ++tile) {
    const long m0 = tile * MTW;
    // ---- 1. the tile's bases -> the wave's window (the previous tile's epilogue has read its window: same wave, program order) ----
    if (!first) {   // the halo: columns 64 .. 71 of the previous tile's slice are columns 0 .. 7 of this one (8 planes x 8 columns = 64 units)
      const int pl = lane >> 3, cc = lane & 7;
      const f32x4 u = slice[(pl >> 2) * SLOT + (pl & 3) * XW + 64 + cc];
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      slice[(pl >> 2) * SLOT + (pl & 3) * XW + cc] = u;
    }
    win[lane] = nb0;
    win[64 + lane] = nb1;
    if (tile + 1 < tile_end) { nb0 = base_at(m0 + MTW - 16 + lane); nb1 = base_at(m0 + MTW - 16 + 64 + lane); }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    // ---- 2. producer: a1 = relu(25-tap conv of the bases) -> the slice ----
    if (first) produce(m0, std::integral_constant<int, 3>(), std::integral_constant<int, 0>());
    else produce(m0, std::integral_constant<int, 2>(), std::integral_constant<int, 8>());
    first = false;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the slice is written (same wave: LDS operations retire in order)
    // ---- 3. conv1.b's accumulators start from its bias ----
#pragma unroll
    for (int j = 0; j < NW; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 b_ = *reinterpret_cast<const f32x4*>(bias_s + 4 * g + j * 32 + 8 * q);
#pragma unroll
        for (int i = 0; i < MW; ++i) { acc[i][j][4 * q + 0] = b_.x; acc[i][j][4 * q + 1] = b_.y; acc[i][j][4 * q + 2] = b_.z; acc[i][j][4 * q + 3] = b_.w; }
      }
    // ---- 4. conv1.b: 2 steps x 9 taps out of the slice (conv_ws.h's block; fragments double-buffered across taps) ----
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const unsigned xa0 = p16_lds_addr(slice + c * SLOT + g * XW + l31);          // + (s*2*XW + i*32 + tap)*16
      const unsigned wb0 = p16_lds_addr(smem + (c * 2 * 9 * 2 + g) * CT + l31);     // + (((s*9+tap)*2)*CT + j*32)*16
      f16x8 av[2][2][MW], bv[2][2][NW];
#define S1_READ_FRAGS(buf_, tap_)                                                                                 \
  _Pragma("unroll") for (int s = 0; s < 2; ++s) {                                                                 \
    _Pragma("unroll") for (int i = 0; i < MW; ++i) av[buf_][s][i] = p16_lds_read16(xa0, (s * 2 * XW + i * 32 + (tap_)) * 16); \
    _Pragma("unroll") for (int j = 0; j < NW; ++j) bv[buf_][s][j] = p16_lds_read16(wb0, (((s * 9 + (tap_)) * 2) * CT + j * 32) * 16); \
  }
      S1_READ_FRAGS(0, 0);
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        const int fb = tap & 1;
        if (tap + 1 < 9) S1_READ_FRAGS(fb ^ 1, tap + 1);
        if (tap + 1 < 9) p16_lds_wait<2 * (MW + NW), MW, NW>(av[fb], bv[fb]);
        else p16_lds_wait<0, MW, NW>(av[fb], bv[fb]);
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
          for (int i = 0; i < MW; ++i)
#pragma unroll
            for (int j = 0; j < NW; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, bv[fb][p][j]), __builtin_bit_cast(bf16x8, av[fb][p][i]), acc[i][j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
#undef S1_READ_FRAGS
    }
    // ---- 5. epilogue: ReLU, + lout1 from the bases, MaxPool1d(4), bf16 units ----
    epi_tile = tile;
    P16_EPILOGUE();
    epi_tile = -1;
  }
#undef P16_EPI_CB
#undef P16_EPI_HOOK
#undef P16_EPI_M0
}
