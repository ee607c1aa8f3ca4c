// conv_small.h - Conv1d k = 9 on SHORT channel-last rows (n <= 2048 positions): the Encoder's stages 5-7 of a local re-encode (55-550 positions
// per layer: a 220-440 kb piece of an SV allele window, orca_amd/sv.py) and of one- to three-bin inputs.
//
// Why: conv1d_k9_bf16s_kernel walks the K-chunks of a tile one after the other - per chunk a global load, a split, an LDS store and two
// barriers in front of 27-54 MFMAs - and a workgroup covers all couts of the layer (74 KB of weights per chunk): on 1-5 tiles that is a
// latency chain of 8 x ~3 us on 1-5 CUs, 23-27 us per launch whatever n is (11.5 % of the kernel time of a 1 024-variant SV screen).
// Here the K-chunks of a tile run SIDE BY SIDE: a workgroup is 32 positions x 32 couts, wave w takes chunk w (, w + 8, ...) - its 16 input
// channels over 40 positions through a wave-private LDS image (split on the fly, no workgroup barrier), its weight fragments straight
// from global memory into registers (a wave is the only reader of its 18 KB: lane (cout, k-group) loads the unit the MFMA wants) -, the
// eight partial accumulators meet in LDS and waves 0-3 add them in a fixed order (chunk 0 first) and run the epilogue.  One global round
// trip, 27 MFMAs per wave, one barrier: ~6 us per launch, ceil(n / 32) x cout / 32 workgroups per row.
// Same operand splits and product order per tap as conv_bf16s.h (NS / DT); the fp32 accumulation order differs (per-chunk partials),
// so results agree with it to fp32 rounding, not bit for bit - which kernel runs is a function of n alone, never of the batch.
#pragma once
#include "conv_bf16s.h"

template <int NS, int DT>
__global__ __launch_bounds__(512, 2) void conv1d_k9_small_kernel(ConvB16Args a) {
  constexpr int XP = 40;                                   // staged positions per wave: 32 + 8 halo
  constexpr int SPL = DT == 1 ? 2 : 3;                     // splits in the weight pack
  constexpr int NPROD = NS == 3 ? 6 : (NS == 2 ? 3 : 1);
  __shared__ f32x4 xs[8][NS * 2 * XP];                     // per wave: [split][k-group g][40 positions] 16-byte units
  __shared__ float part[8][16][64];                        // per wave: partial accumulator [register][lane]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, g = lane >> 5;
  const int cout = a.cout, ncb = cout >> 5;
  const int cb = (int)(blockIdx.x % ncb);
  const long m0 = (long)(blockIdx.x / ncb) * 32;
  const int b = blockIdx.y;
  const float* xb = a.x + (long)b * a.x_bs;
  const f32x4* wg = reinterpret_cast<const f32x4*>(a.w);
  bool overflow = false;

  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  char* const xw = reinterpret_cast<char*>(xs[wave]);
  for (int c = wave; c < a.nchunks; c += 8) {
    // weight fragments of chunk c, cout block cb: unit ((s*9 + tap)*2 + g) * cout + cb*32 + l31 of the chunk's pack
    typename Op16<DT>::vec bv[NS][9];
    const f32x4* wc = wg + (long)c * (SPL * 9 * 2) * cout + cb * 32 + l31;
#pragma unroll
    for (int s = 0; s < NS; ++s)
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) bv[s][tap] = __builtin_bit_cast(typename Op16<DT>::vec, wc[(long)((s * 9 + tap) * 2 + g) * cout]);
    // X image of the chunk: 40 positions x 16 channels = 160 float4, split into NS parts on the way into LDS
#pragma unroll
    for (int it = 0; it < 3; ++it) {
      const int u = lane + it * 64;
      if (u < 4 * XP) {
        const int prel = u >> 2, q = u & 3;
        const long pos = m0 - 4 + prel;
        f32x4 v = (f32x4)(0.f);
        if (pos >= 0 && pos < a.n) v = *reinterpret_cast<const f32x4*>(xb + pos * (long)a.cin + 16 * c + 4 * q);
        u32x2 sp[NS];
        split4<NS, DT>(v, sp, overflow);
#pragma unroll
        for (int s = 0; s < NS; ++s) *reinterpret_cast<u32x2*>(xw + ((s * 2 + (q >> 1)) * XP + prel) * 16 + (q & 1) * 8) = sp[s];
      }
    }
    // the image is private to the wave and LDS executes a wave's instructions in order: no barrier, only the compiler must not move the
    // fragment reads above the stores (nor, on the next chunk of this wave, the stores above these reads)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      typename Op16<DT>::vec av[NS];
#pragma unroll
      for (int s = 0; s < NS; ++s) av[s] = __builtin_bit_cast(typename Op16<DT>::vec, xs[wave][(s * 2 + g) * XP + l31 + tap]);
#pragma unroll
      for (int p = 0; p < NPROD; ++p) {        // small -> large, as conv_bf16s.h
        constexpr int PA3[6] = {2, 1, 0, 1, 0, 0}, PB3[6] = {0, 1, 2, 0, 1, 0};
        constexpr int PA2[3] = {1, 0, 0}, PB2[3] = {0, 1, 0};
        const int sa = NS == 3 ? PA3[p] : (NS == 2 ? PA2[p] : 0);
        const int sb = NS == 3 ? PB3[p] : (NS == 2 ? PB2[p] : 0);
        acc = Op16<DT>::mfma(bv[sb][tap], av[sa], acc);          // D[cout][pos]
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) part[wave][r][lane] = acc[r];
  __syncthreads();
  if (wave < 4) {
    // register group q = wave of every partial tile: couts 8q + 4g .. + 3 of position l31 (the MFMA's D layout, conv_bf16s.h)
    const int q = wave;
    const long pos = m0 + l31;
    const int co = cb * 32 + 8 * q + 4 * g;
    f32x4 v = *reinterpret_cast<const f32x4*>(a.bias + co);
#pragma unroll
    for (int w = 0; w < 8; ++w) {
      v.x += part[w][4 * q + 0][lane]; v.y += part[w][4 * q + 1][lane]; v.z += part[w][4 * q + 2][lane]; v.w += part[w][4 * q + 3][lane];
    }
    if (a.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
    if (pos < a.n) {
      const long o = pos * cout + co;
      if (a.r1) v += *reinterpret_cast<const f32x4*>(a.r1 + (long)b * a.r_bs + o);
      if (a.r2) v += *reinterpret_cast<const f32x4*>(a.r2 + (long)b * a.r_bs + o);     // may alias y: read before this lane's own store
      *reinterpret_cast<f32x4*>(a.y + (long)b * a.y_bs + o) = v;
    }
  }
  if (DT == 1 && overflow && a.flag) *a.flag = 1u;
}
