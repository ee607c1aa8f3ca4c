// orca_decoder.hip - Conv2d launchers, Decoder / Decoder_1m (orca_modules.py:461-488, :782-800), strand merge, 256 Mb background block means, the observed-data smoother, the 2-bit genome expander
// Part of liborca_hip.so (include/orca_hip.h is the ABI; orca_internal.h what the units share).
#include "orca_internal.h"

#include "conv2d_m16.h"
#include "conv2d_m16q.h"
#include "conv2d_dblock.h"
#include "misc_kernels.h"
#include "coarsegrain.h"

// ---------------------------------------------------------------------------
// kernel launch helpers
// ---------------------------------------------------------------------------
int launch_conv2d(orca_ctx* ctx, const ConvLayer& L, const float* x, long x_bs, float* y, long y_bs,
                         const float* r, long r_bs, int B, int n, int relu) {
  if (L.ksize != 3) return fail(ORCA_EINVAL, "launch_conv2d on a non-3x3 layer");
  Conv2dArgs a;
  a.x = x; a.w = L.d_w; a.bias = L.d_bias; a.y = y; a.r = r;
  a.x_bs = x_bs; a.y_bs = y_bs; a.r_bs = r_bs; a.H = n; a.W = n; a.dil = L.dil; a.nchunks = L.nchunks; a.relu = relu;
  dim3 grid((unsigned)n, (unsigned)B);
  if (L.cout == 64) hipLaunchKernelGGL((conv2d_3x3_kernel<64>), grid, dim3(512), 0, ctx->stream, a);
  else hipLaunchKernelGGL((conv2d_3x3_kernel<32>), grid, dim3(512), 0, ctx->stream, a);
  LAUNCHCHECK("conv2d_3x3_kernel");
  return ORCA_OK;
}

// dilated 3x3 conv on M16 maps (conv2d_m16.h); maps are unit arrays [octets][NS][n][256]; strides in units
// mode: 0 = f16x2 (two fp16 planes, 3 products), 1 = bf16 (one plane, 1 product), 2 = f16 (one fp16 plane, 1 product)
// chunk0 / nchunks_: a sub-range of the layer's 16-channel input chunks (x then starts at channel octet 0 of THAT range); tab: per-map
// tables [2][3][n][64] added in the epilogue (row term by column class, column term by row class - see sep_tables_kernel)
int launch_conv2d_m16(orca_ctx* ctx, const ConvLayer& L, const f32x4* x, long x_bs, int x_oct, f32x4* y, long y_bs, int y_oct,
                             const f32x4* r, long r_bs, int B, int n, int relu, int mode, int chunk0, int nchunks_,
                             const float* tab, long tab_bs) {
  const bool bf16 = mode == 1;
  if (L.ksize != 3 || !L.d_wf16 || !L.d_wb16p) return fail(ORCA_EINVAL, "launch_conv2d_m16 on a layer without a 16-bit pack");
  if (!bf16 && !L.f16_ok) return fail(ORCA_EINVAL, "layer weights exceed the fp16 range");
  if (L.dil > 8) return fail(ORCA_EINVAL, "conv2d_3x3_m16_kernel handles dilations 1-8 (got %d); larger ones run as fused blocks", L.dil);
  ConvM16Args a;
  a.x = x; a.w = bf16 ? L.d_wb16p : L.d_wf16; a.bias = L.d_bias; a.y = y; a.r = r; a.x_bs = x_bs; a.y_bs = y_bs; a.r_bs = r_bs;
  a.H = n; a.W = n; a.dil = L.dil; a.nchunks = nchunks_ > 0 ? nchunks_ : (L.cin + 15) / 16; a.relu = relu; a.flag = ctx->d_flag;
  a.tab = tab; a.tab_bs = tab_bs;
  if (chunk0 > 0) a.w = static_cast<const char*>(a.w) + (size_t)chunk0 * (bf16 ? 1 : 2) * 9 * 2 * L.cout * 8 * 2;   // pack [chunk][splits][9][2][cout][8] halves
  if (a.nchunks * 2 > x_oct) return fail(ORCA_EINVAL, "conv2d_m16: input map has %d channel octets, layer needs %d", x_oct, a.nchunks * 2);
  if (L.cout / 8 > y_oct) return fail(ORCA_EINVAL, "conv2d_m16: output map narrower than the layer");
  // batches (conv2d_m16q.h): tiles of four output rows (y .. y + 3d) x 128 pixels, one launch for the whole batch - both strands of a level
  // are one round of 252-256 workgroups (Decoder forward at B = 2: 2.32 against 2.43 ms, same box).  A single map is 126-128 such workgroups,
  // half the chip: it stays on the one-row kernel (1.40 against 1.62 ms).  Both kernels sum in one order: a map is bit-identical whichever one
  // its batch size selects (tests/test_gpu_nets.py compares B = 1 with rows of B = 2 / 4 / 8).
  if (B >= 2) {
    ConvM16QArgs aq;
    aq.c = a; aq.c.banded = 0; aq.zero = reinterpret_cast<const f32x4*>(ctx->d_zero);
    aq.ngroups = ((n + 4 * L.dil - 1) / (4 * L.dil)) * L.dil;
    aq.nb = B;
    // batches of more than one round (SV screen: 4 strands, config 3: 8): the grid is ONE round, a workgroup walks the maps b, b + grid.y, ...
    // of its tile and requests the next map's first piece under the last piece of the current one (needs an even chunk count: the heads'
    // 16- / 80- / 144-channel layers keep one workgroup per map and tile)
    static int ncu = [] { int dev = 0, v = 256; (void)hipGetDevice(&dev); (void)hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev); return v; }();
    const int gx = (aq.ngroups * 2 + 7) / 8 * 8;
    int gy = B;
    // (single-plane modes, 32 couts: 73.7 KB of LDS and 111 VGPRs - TWO workgroups fit a CU, one's transfers and epilogue under the other's
    // MFMAs: the resident round is twice as large)
    const int res = (mode != 0 && L.cout == 32) ? 2 * ncu : ncu;
    if (a.nchunks % 2 == 0 && gx * B > res) gy = res / gx > 1 ? res / gx : 1;
    if (gy > B) gy = B;
    dim3 gridq((unsigned)gx, (unsigned)gy);
    if (bf16) {
      if (L.cout == 64) hipLaunchKernelGGL((conv2d_3x3_m16q_kernel<64, 1, 0>), gridq, dim3(512), 0, ctx->stream, aq);
      else hipLaunchKernelGGL((conv2d_3x3_m16q_kernel<32, 1, 0>), gridq, dim3(512), 0, ctx->stream, aq);
    } else if (mode == 2) {
      if (L.cout == 64) hipLaunchKernelGGL((conv2d_3x3_m16q_kernel<64, 1, 1>), gridq, dim3(512), 0, ctx->stream, aq);
      else hipLaunchKernelGGL((conv2d_3x3_m16q_kernel<32, 1, 1>), gridq, dim3(512), 0, ctx->stream, aq);
    } else {
      if (L.cout == 64) hipLaunchKernelGGL((conv2d_3x3_m16q_kernel<64, 2, 1>), gridq, dim3(512), 0, ctx->stream, aq);
      else hipLaunchKernelGGL((conv2d_3x3_m16q_kernel<32, 2, 1>), gridq, dim3(512), 0, ctx->stream, aq);
    }
    LAUNCHCHECK("conv2d_3x3_m16q_kernel");
    return ORCA_OK;
  }
  a.banded = n >= 64 ? 1 : 0;
  dim3 grid((unsigned)(a.banded ? 8 * ((n + 7) / 8) : n), (unsigned)B);
  if (bf16) {
    if (L.cout == 64) hipLaunchKernelGGL((conv2d_3x3_m16_kernel<64, 1, 0>), grid, dim3(512), 0, ctx->stream, a);
    else hipLaunchKernelGGL((conv2d_3x3_m16_kernel<32, 1, 0>), grid, dim3(512), 0, ctx->stream, a);
  } else if (mode == 2) {
    if (L.cout == 64) hipLaunchKernelGGL((conv2d_3x3_m16_kernel<64, 1, 1>), grid, dim3(512), 0, ctx->stream, a);
    else hipLaunchKernelGGL((conv2d_3x3_m16_kernel<32, 1, 1>), grid, dim3(512), 0, ctx->stream, a);
  } else {
    if (L.cout == 64) hipLaunchKernelGGL((conv2d_3x3_m16_kernel<64, 2, 1>), grid, dim3(512), 0, ctx->stream, a);
    else hipLaunchKernelGGL((conv2d_3x3_m16_kernel<32, 2, 1>), grid, dim3(512), 0, ctx->stream, a);
  }
  LAUNCHCHECK("conv2d_3x3_m16_kernel");
  return ORCA_OK;
}

// ---------------------------------------------------------------------------
// Decoder / Decoder_1m (orca_modules.py:461-488, :782-800)
// ---------------------------------------------------------------------------
// batch rows of a Decoder input: slices of one strided tensor (base + b*bs) or one device pointer per row
struct RowSrc {
  const float* base = nullptr;
  long bs = 0;
  const float* const* rows = nullptr;
  const float* at(int b) const { return rows ? rows[b] : (base ? base + (long)b * bs : nullptr); }
  explicit operator bool() const { return rows || base; }
};

static int launch_final(orca_ctx* ctx, orca_net* net, const float* cur, long cur_bs, float* out, int B, int n, int accumulate) {
  const ConvLayer& fa = net->convs[net->convs.size() - 2];
  const ConvLayer& fb = net->convs[net->convs.size() - 1];
  FinalArgs a;
  a.cur = cur; a.w1 = fa.d_w; a.b1 = fa.d_bias; a.w2 = fb.d_w; a.b2 = fb.d_bias; a.out = out;
  a.cur_bs = cur_bs; a.out_bs = (long)net->num_2d * n * n; a.n = n; a.accumulate = accumulate;
  a.T = net->num_2d; a.F = fa.cout;
  hipLaunchKernelGGL(final_sym_kernel, dim3((unsigned)n, (unsigned)B), dim3(256), 0, ctx->stream, a);
  LAUNCHCHECK("final_sym_kernel");
  return ORCA_OK;
}

// Decoder / Decoder_1m on the 16-bit matrix cores, feature maps in M16 (conv2d_m16.h)
template <int NS, int DT>
static int decoder_m16(orca_ctx* ctx, orca_net* net, const RowSrc& x, long sx_c, long sx_l, const RowSrc& de,
                       long sd_c, long sd_h, long sd_w, const RowSrc& y, long sy_c, long sy_h, long sy_w, int B, int n,
                       float* out, int accumulate) {
  const int nt2 = net->num_2d;
  const bool is1m = net->kind == ORCA_NET_DECODER_1M, bf16 = DT == 0;
  const int mode = DT == 0 ? 1 : (NS == 1 ? 2 : 0);
  // channel octets: Decoder_1m 128 channels of outer sum; Decoder: ONLY the distenc chunk (16 channels) - the 128 outer-sum channels of
  // lcombinerD.a never exist as a map (separable, see orca_net_create); A: 80 (64 + coarse prediction)
  const int oIN = is1m ? 16 : 2, oA = 10;
  const size_t tabsz = (size_t)2 * 3 * n * 64;               // floats per map
  const size_t upo = (size_t)NS * n * ORCA_LDW;              // units per octet and map
  const size_t szIN = upo * oIN, szA = upo * oA, sz64 = upo * 8, sz32 = upo * 4;   // units
  const size_t need = ru256(B * szIN * 16) + ru256(B * szA * 16) + 3 * ru256(B * sz64 * 16) + ru256(B * sz32 * 16) + ru256(B * tabsz * 4);
  ORCA_TRY(ws_ensure(ctx, need));
  auto take = [&](size_t units) { return reinterpret_cast<f32x4*>(ws_take(ctx, units * 4)); };
  float* const TAB0 = is1m ? nullptr : ws_take(ctx, B * tabsz);
  f32x4* const IN0 = take(B * szIN);
  f32x4* const A0 = take(B * szA);
  f32x4* const Bf0 = take(B * sz64);
  f32x4* const Cf0 = take(B * sz64);
  f32x4* const Df0 = take(B * sz64);
  f32x4* const T0 = take(B * sz32);
  // maps [b0, b0 + nb) of the batch, on ctx->stream
  auto run = [&](int b0, int nb) -> int {
    f32x4* IN = IN0 + b0 * szIN;
    f32x4* A = A0 + b0 * szA;
    f32x4* Bf = Bf0 + b0 * sz64;
    f32x4* Cf = Cf0 + b0 * sz64;
    f32x4* Df = Df0 + b0 * sz64;
    f32x4* T = T0 + b0 * sz32;
    hipStream_t s = ctx->stream;
    float* TAB = is1m ? nullptr : TAB0 + b0 * tabsz;
    // everything computed from the inputs alone - IN (outer sum / distenc chunk), the separable tables, the upsampled coarse prediction - in one
    // launch per 8 maps (decoder_head_m16_kernel)
    for (int c0 = 0; c0 < nb; c0 += 8) {
      const int nc = nb - c0 < 8 ? nb - c0 : 8;
      M16HeadArgs ha{};
      for (int b = 0; b < nc; ++b) { ha.x[b] = x.at(b0 + c0 + b); ha.de[b] = de.at(b0 + c0 + b); ha.y[b] = (!is1m && y) ? y.at(b0 + c0 + b) : nullptr; }
      ha.sx_c = sx_c; ha.sx_l = sx_l; ha.sd_c = sd_c; ha.sd_h = sd_h; ha.sd_w = sd_w; ha.sy_c = sy_c; ha.sy_h = sy_h; ha.sy_w = sy_w;
      ha.in = IN + c0 * szIN; ha.in_bs = (long)szIN;
      ha.tab = is1m ? nullptr : TAB + c0 * tabsz; ha.tab_bs = (long)tabsz;
      ha.a = A + c0 * szA; ha.a_bs = (long)szA;
      ha.wsep = net->d_sep; ha.nt = nt2; ha.n = n; ha.o0 = is1m ? 0 : 16; ha.noct = oIN; ha.nsep = is1m ? 0 : 6;
      ha.bilinear = net->upsample_mode == ORCA_UPSAMPLE_BILINEAR ? 1 : 0; ha.flag = ctx->d_flag;
      const unsigned roles = (unsigned)(oIN + ha.nsep + ((!is1m && y) ? 1 : 0));
      hipLaunchKernelGGL((decoder_head_m16_kernel<NS, DT>), dim3((unsigned)n, roles, (unsigned)nc), dim3(256), 0, s, ha);
      LAUNCHCHECK("decoder_head_m16_kernel");
    }
    const ConvLayer* L = net->convs.data();
    const ConvLayer* pairs;
    int npairs;
#define C2(layer, src, sbs, so, dst, dbs, dso, res, rbs, relu) \
  ORCA_TRY(launch_conv2d_m16(ctx, layer, src, sbs, so, dst, dbs, dso, res, rbs, nb, n, relu, mode))
    if (!is1m) {
      // lcombinerD.a = (MFMA conv over the distenc chunk) + (separable outer-sum part from the tables, added in the epilogue)
      ORCA_TRY(launch_conv2d_m16(ctx, L[0], IN, szIN, oIN, Bf, sz64, 8, nullptr, 0, nb, n, 0, mode, 8, 1, TAB, (long)tabsz));
      C2(L[1], Bf, sz64, 8, Cf, sz64, 8, nullptr, 0, 0);
      C2(L[2], Cf, sz64, 8, Bf, sz64, 8, nullptr, 0, 1);
      C2(L[3], Bf, sz64, 8, A, szA, oA, Cf, sz64, 1);           // A[octets 0..7] = combinerD(.) + .
      pairs = L + 8; npairs = 28;
      if (y) {
        // (octets 8, 9 of A - the upsampled coarse prediction - were written by the head launch)
        C2(L[4], A, szA, oA, Bf, sz64, 8, nullptr, 0, 0);
        C2(L[5], Bf, sz64, 8, Cf, sz64, 8, nullptr, 0, 0);
        C2(L[6], Cf, sz64, 8, Bf, sz64, 8, nullptr, 0, 1);
        C2(L[7], Bf, sz64, 8, Df, sz64, 8, Cf, sz64, 1);
      } else {
        C2(pairs[0], A, szA, oA, T, sz32, 4, nullptr, 0, 0);
        C2(pairs[1], T, sz32, 4, Cf, sz64, 8, nullptr, 0, 0);
        C2(pairs[2], Cf, sz64, 8, T, sz32, 4, nullptr, 0, 1);
        C2(pairs[3], T, sz32, 4, Df, sz64, 8, Cf, sz64, 1);
      }
    } else {
      pairs = L; npairs = 19;
      C2(pairs[0], IN, szIN, oIN, T, sz32, 4, nullptr, 0, 0);
      C2(pairs[1], T, sz32, 4, Cf, sz64, 8, nullptr, 0, 0);
      C2(pairs[2], Cf, sz64, 8, T, sz32, 4, nullptr, 0, 1);
      C2(pairs[3], T, sz32, 4, Df, sz64, 8, Cf, sz64, 1);
    }
    f32x4* cur = Df;
    f32x4* oth = Cf;
    for (int i = 1; i < npairs; ++i) {
      const ConvLayer* p = pairs + 4 * i;
      const int dil = p[0].dil;
      if (dil >= 16) {
        // the whole block (oth = lm(cur) + cur; cur = m(oth) + oth) in one launch, in place (conv2d_dblock.h)
        if (!(dil == 16 || dil == 32 || dil == 64) || p[1].dil != dil || p[2].dil != dil || p[3].dil != dil)
          return fail(ORCA_EINVAL, "decoder block %d: dilation %d unsupported", i, dil);
        DBlockArgs da;
        da.cur = cur; da.bs = sz64; da.H = n; da.W = n; da.dil = dil; da.flag = ctx->d_flag;
        for (int k = 0; k < 4; ++k) {
          if (!bf16 && !p[k].f16_ok) return fail(ORCA_EINVAL, "layer weights exceed the fp16 range");
          da.w[k] = bf16 ? p[k].d_wb16p : p[k].d_wf16;
          da.bias[k] = p[k].d_bias;
        }
        hipLaunchKernelGGL((conv2d_dblock_kernel<NS, DT>), dim3(256, (unsigned)nb), dim3(512), 0, ctx->stream, da);
        LAUNCHCHECK("conv2d_dblock_kernel");
        continue;
      }
      C2(p[0], cur, sz64, 8, T, sz32, 4, nullptr, 0, 0);
      C2(p[1], T, sz32, 4, oth, sz64, 8, cur, sz64, 0);
      C2(p[2], oth, sz64, 8, T, sz32, 4, nullptr, 0, 1);
      C2(p[3], T, sz32, 4, cur, sz64, 8, oth, sz64, 1);
    }
#undef C2
    const ConvLayer& fa = net->convs[net->convs.size() - 2];
    const ConvLayer& fb = net->convs[net->convs.size() - 1];
    FinalArgs fa_;
    fa_.cur = reinterpret_cast<const float*>(cur); fa_.w1 = fa.d_w; fa_.b1 = fa.d_bias; fa_.w2 = fb.d_w; fa_.b2 = fb.d_bias; fa_.out = out + (size_t)b0 * nt2 * n * n;
    fa_.cur_bs = sz64; fa_.out_bs = (long)nt2 * n * n; fa_.n = n; fa_.accumulate = accumulate; fa_.T = nt2; fa_.F = fa.cout;
    hipLaunchKernelGGL((final_sym_m16_kernel<NS, DT>), dim3(136u, (unsigned)nb), dim3(256), 0, s, fa_);   // 16 x 16 tile pairs of the upper triangle
    LAUNCHCHECK("final_sym_m16_kernel");
    return ORCA_OK;
  };
  // A Decoder is a chain of ~90 dependent launches per map; a launch carries the WHOLE batch (conv2d_m16q.h: both strands of a level are
  // 252-256 workgroups = one round on 256 CUs).  Half-batches on two streams were measured twice (rounds 3 and 5: slower or noise) and are gone.
  return run(0, B);
}

static int decoder_common(orca_ctx* ctx, orca_net* net, const RowSrc& x, long sx_c, long sx_l, const RowSrc& de,
                          long sd_c, long sd_h, long sd_w, const RowSrc& y, long sy_c, long sy_h, long sy_w,
                          int B, int n, float* out, int accumulate) {
  const int nt2 = net->num_2d;
  if (n <= 0 || n > ORCA_LDW || (n & 1)) return fail(ORCA_EINVAL, "map size %d unsupported (even, <=256)", n);
  if (B <= 0) return ORCA_OK;
  HIPCHECK(hipSetDevice(ctx->device));
  if (net->precision == ORCA_PRECISION_F16X2)
    return decoder_m16<2, 1>(ctx, net, x, sx_c, sx_l, de, sd_c, sd_h, sd_w, y, sy_c, sy_h, sy_w, B, n, out, accumulate);
  if (net->precision == ORCA_PRECISION_BF16)
    return decoder_m16<1, 0>(ctx, net, x, sx_c, sx_l, de, sd_c, sd_h, sd_w, y, sy_c, sy_h, sy_w, B, n, out, accumulate);
  if (net->precision == ORCA_PRECISION_F16)
    return decoder_m16<1, 1>(ctx, net, x, sx_c, sx_l, de, sd_c, sd_h, sd_w, y, sy_c, sy_h, sy_w, B, n, out, accumulate);
  const bool is1m = net->kind == ORCA_NET_DECODER_1M;
  const size_t plane = (size_t)n * ORCA_LDW;
  const int cin0 = is1m ? 128 : 136;
  const size_t szIN = plane * cin0, szA = plane * 72, sz64 = plane * 64, sz32 = plane * 32;
  const size_t need = ru256(B * szIN * 4) + ru256(B * szA * 4) + 3 * ru256(B * sz64 * 4) + ru256(B * sz32 * 4);
  ORCA_TRY(ws_ensure(ctx, need));
  float* IN = ws_take(ctx, B * szIN);
  float* A = ws_take(ctx, B * szA);
  float* Bf = ws_take(ctx, B * sz64);
  float* Cf = ws_take(ctx, B * sz64);
  float* Df = ws_take(ctx, B * sz64);
  float* T = ws_take(ctx, B * sz32);
  hipStream_t s = ctx->stream;
  for (int b = 0; b < B; ++b) {
    hipLaunchKernelGGL(outer_sum_kernel, dim3((unsigned)n, (unsigned)cin0), dim3(64), 0, s, x.at(b), sx_c, sx_l,
                       de.at(b), sd_c, sd_h, sd_w, nt2, IN + b * szIN, n, cin0);
    LAUNCHCHECK("outer_sum_kernel");
  }
  const ConvLayer* L = net->convs.data();
  const ConvLayer* pairs;
  int npairs;
  if (!is1m) {
    ORCA_TRY(launch_conv2d(ctx, L[0], IN, szIN, Bf, sz64, nullptr, 0, B, n, 0));
    ORCA_TRY(launch_conv2d(ctx, L[1], Bf, sz64, Cf, sz64, nullptr, 0, B, n, 0));     // Cf = lcombinerD(mat)
    ORCA_TRY(launch_conv2d(ctx, L[2], Cf, sz64, Bf, sz64, nullptr, 0, B, n, 1));
    ORCA_TRY(launch_conv2d(ctx, L[3], Bf, sz64, A, szA, Cf, sz64, B, n, 1));         // A[0:64] = combinerD(.)+.
    pairs = L + 8; npairs = 28;
    if (y) {
      for (int b = 0; b < B; ++b) {
        hipLaunchKernelGGL(upsample2d_x2_kernel, dim3((unsigned)n, 8), dim3(ORCA_LDW), 0, s, y.at(b), sy_c, sy_h, sy_w, nt2,
                           A + b * szA + 64 * plane, n, net->upsample_mode == ORCA_UPSAMPLE_BILINEAR ? 1 : 0, 8);
        LAUNCHCHECK("upsample2d_x2_kernel");
      }
      ORCA_TRY(launch_conv2d(ctx, L[4], A, szA, Bf, sz64, nullptr, 0, B, n, 0));
      ORCA_TRY(launch_conv2d(ctx, L[5], Bf, sz64, Cf, sz64, nullptr, 0, B, n, 0));   // Cf = lcombiner(cat)
      ORCA_TRY(launch_conv2d(ctx, L[6], Cf, sz64, Bf, sz64, nullptr, 0, B, n, 1));
      ORCA_TRY(launch_conv2d(ctx, L[7], Bf, sz64, Df, sz64, Cf, sz64, B, n, 1));     // Df = combiner(.)+.
    } else {
      ORCA_TRY(launch_conv2d(ctx, pairs[0], A, szA, T, sz32, nullptr, 0, B, n, 0));
      ORCA_TRY(launch_conv2d(ctx, pairs[1], T, sz32, Cf, sz64, nullptr, 0, B, n, 0));  // Cf = lm0(mat) (no residual, :477)
      ORCA_TRY(launch_conv2d(ctx, pairs[2], Cf, sz64, T, sz32, nullptr, 0, B, n, 1));
      ORCA_TRY(launch_conv2d(ctx, pairs[3], T, sz32, Df, sz64, Cf, sz64, B, n, 1));
    }
  } else {
    pairs = L; npairs = 19;
    ORCA_TRY(launch_conv2d(ctx, pairs[0], IN, szIN, T, sz32, nullptr, 0, B, n, 0));
    ORCA_TRY(launch_conv2d(ctx, pairs[1], T, sz32, Cf, sz64, nullptr, 0, B, n, 0));
    ORCA_TRY(launch_conv2d(ctx, pairs[2], Cf, sz64, T, sz32, nullptr, 0, B, n, 1));
    ORCA_TRY(launch_conv2d(ctx, pairs[3], T, sz32, Df, sz64, Cf, sz64, B, n, 1));
  }
  float* cur = Df;
  float* oth = Cf;
  for (int i = 1; i < npairs; ++i) {
    const ConvLayer* p = pairs + 4 * i;
    ORCA_TRY(launch_conv2d(ctx, p[0], cur, sz64, T, sz32, nullptr, 0, B, n, 0));
    ORCA_TRY(launch_conv2d(ctx, p[1], T, sz32, oth, sz64, cur, sz64, B, n, 0));   // oth = lm(cur)+cur
    ORCA_TRY(launch_conv2d(ctx, p[2], oth, sz64, T, sz32, nullptr, 0, B, n, 1));
    ORCA_TRY(launch_conv2d(ctx, p[3], T, sz32, cur, sz64, oth, sz64, B, n, 1));   // cur = m(oth)+oth
  }
  return launch_final(ctx, net, cur, sz64, out, B, n, accumulate);
}

extern "C" int orca_decoder_forward(orca_ctx* ctx, orca_net* net, const float* x, int64_t sx_b, int64_t sx_c, int64_t sx_l,
                                    const float* distenc, int64_t sd_b, int64_t sd_h, int64_t sd_w, const float* y,
                                    int64_t sy_b, int64_t sy_h, int64_t sy_w, int B, int n, float* out, int accumulate) {
  if (!ctx || !net || !x || !distenc || !out) return fail(ORCA_EINVAL, "orca_decoder_forward: NULL argument");
  if (net->kind != ORCA_NET_DECODER) return fail(ORCA_EINVAL, "orca_decoder_forward: net is not a Decoder");
  if (net->num_2d != 1) return fail(ORCA_EINVAL, "orca_decoder_forward: net predicts %d maps, use orca_decoder_forward_mt", net->num_2d);
  RowSrc xs, ds, ys;
  xs.base = x; xs.bs = sx_b; ds.base = distenc; ds.bs = sd_b; ys.base = y; ys.bs = sy_b;
  return decoder_common(ctx, net, xs, sx_c, sx_l, ds, 0, sd_h, sd_w, ys, 0, sy_h, sy_w, B, n, out, accumulate);
}

extern "C" int orca_decoder_forward_mt(orca_ctx* ctx, orca_net* net, const float* x, int64_t sx_b, int64_t sx_c, int64_t sx_l,
                                       const float* distenc, int64_t sd_b, int64_t sd_c, int64_t sd_h, int64_t sd_w, const float* y,
                                       int64_t sy_b, int64_t sy_c, int64_t sy_h, int64_t sy_w, int B, int n, float* out,
                                       int accumulate) {
  if (!ctx || !net || !x || !distenc || !out) return fail(ORCA_EINVAL, "orca_decoder_forward_mt: NULL argument");
  if (net->kind != ORCA_NET_DECODER) return fail(ORCA_EINVAL, "orca_decoder_forward_mt: net is not a Decoder");
  RowSrc xs, ds, ys;
  xs.base = x; xs.bs = sx_b; ds.base = distenc; ds.bs = sd_b; ys.base = y; ys.bs = sy_b;
  return decoder_common(ctx, net, xs, sx_c, sx_l, ds, sd_c, sd_h, sd_w, ys, sy_c, sy_h, sy_w, B, n, out, accumulate);
}

extern "C" int orca_decoder_forward_rows(orca_ctx* ctx, orca_net* net, const float* const* x_rows, int64_t sx_c, int64_t sx_l,
                                         const float* const* distenc_rows, int64_t sd_c, int64_t sd_h, int64_t sd_w,
                                         const float* const* y_rows, int64_t sy_c, int64_t sy_h, int64_t sy_w, int B, int n, float* out,
                                         int accumulate) {
  if (!ctx || !net || !x_rows || !distenc_rows || !out) return fail(ORCA_EINVAL, "orca_decoder_forward_rows: NULL argument");
  if (net->kind != ORCA_NET_DECODER) return fail(ORCA_EINVAL, "orca_decoder_forward_rows: net is not a Decoder");
  for (int b = 0; b < B; ++b)
    if (!x_rows[b] || !distenc_rows[b] || (y_rows && !y_rows[b])) return fail(ORCA_EINVAL, "orca_decoder_forward_rows: NULL row pointer %d", b);
  RowSrc xs, ds, ys;
  xs.rows = x_rows; ds.rows = distenc_rows; ys.rows = y_rows;
  return decoder_common(ctx, net, xs, sx_c, sx_l, ds, sd_c, sd_h, sd_w, ys, sy_c, sy_h, sy_w, B, n, out, accumulate);
}

extern "C" int orca_decoder1m_forward_rows(orca_ctx* ctx, orca_net* net, const float* const* x_rows, int64_t sx_c, int64_t sx_l, int B, int n,
                                           float* out, int accumulate) {
  if (!ctx || !net || !x_rows || !out) return fail(ORCA_EINVAL, "orca_decoder1m_forward_rows: NULL argument");
  if (net->kind != ORCA_NET_DECODER_1M) return fail(ORCA_EINVAL, "orca_decoder1m_forward_rows: net is not a Decoder_1m");
  for (int b = 0; b < B; ++b)
    if (!x_rows[b]) return fail(ORCA_EINVAL, "orca_decoder1m_forward_rows: NULL row pointer %d", b);
  RowSrc xs, none;
  xs.rows = x_rows;
  return decoder_common(ctx, net, xs, sx_c, sx_l, none, 0, 0, 0, none, 0, 0, 0, B, n, out, accumulate);
}

extern "C" int orca_decoder1m_forward(orca_ctx* ctx, orca_net* net, const float* x, int64_t sx_b, int64_t sx_c, int64_t sx_l,
                                      int B, int n, float* out, int accumulate) {
  if (!ctx || !net || !x || !out) return fail(ORCA_EINVAL, "orca_decoder1m_forward: NULL argument");
  if (net->kind != ORCA_NET_DECODER_1M) return fail(ORCA_EINVAL, "orca_decoder1m_forward: net is not a Decoder_1m");
  RowSrc xs, none;
  xs.base = x; xs.bs = sx_b;
  return decoder_common(ctx, net, xs, sx_c, sx_l, none, 0, 0, 0, none, 0, 0, 0, B, n, out, accumulate);
}

extern "C" int orca_strand_merge(orca_ctx* ctx, const float* fwd, const float* rev, float* out, int n) {
  if (!ctx || !fwd || !rev || !out || n <= 0) return fail(ORCA_EINVAL, "orca_strand_merge: bad argument");
  HIPCHECK(hipSetDevice(ctx->device));
  hipLaunchKernelGGL(strand_merge_kernel, dim3((unsigned)((n * n + 255) / 256)), dim3(256), 0, ctx->stream, fwd, rev, out, n);
  LAUNCHCHECK("strand_merge_kernel");
  return ORCA_OK;
}

extern "C" int orca_block_mean_f64(orca_ctx* ctx, const double* mat, int64_t ld, int64_t row0, int64_t col0, int nb, int npix, double* mean_out,
                                   float* log_out, int flip) {
  if (!ctx || !mat || nb <= 0 || npix <= 0 || (!mean_out && !log_out)) return fail(ORCA_EINVAL, "orca_block_mean_f64: bad argument");
  HIPCHECK(hipSetDevice(ctx->device));
  hipLaunchKernelGGL(block_mean_f64_kernel, dim3((unsigned)((npix + 63) / 64), (unsigned)npix), dim3(64), 0, ctx->stream, mat, (long)ld, (long)row0,
                     (long)col0, nb, npix, mean_out, log_out, flip);
  LAUNCHCHECK("block_mean_f64_kernel");
  return ORCA_OK;
}

extern "C" int orca_adaptive_coarsegrain(orca_ctx* ctx, const float* ar, const float* countar, int64_t ld, int n, float cutoff, int max_levels,
                                         int min_shape, float* out, int64_t ld_out) {
  if (!ctx || !ar || !countar || !out) return fail(ORCA_EINVAL, "orca_adaptive_coarsegrain: NULL argument");
  if (n <= 0 || n > 32768 || ld < n || ld_out < n || max_levels < 0 || min_shape < 1) return fail(ORCA_EINVAL, "orca_adaptive_coarsegrain: bad shape");
  HIPCHECK(hipSetDevice(ctx->device));
  int N = 1;
  while (N < n) N <<= 1;
  std::vector<int> sides{N};
  for (int i = 0; i < max_levels; ++i)
    if (sides.back() > min_shape) sides.push_back(sides.back() / 2);
  size_t need = 0;
  for (int sd : sides) need += 3 * ru256((size_t)sd * sd * 4);
  ORCA_TRY(ws_ensure(ctx, need));
  std::vector<float*> v(sides.size()), c(sides.size());
  std::vector<int*> m(sides.size());
  for (size_t l = 0; l < sides.size(); ++l) {
    const size_t e = (size_t)sides[l] * sides[l];
    v[l] = ws_take(ctx, e); c[l] = ws_take(ctx, e); m[l] = reinterpret_cast<int*>(ws_take(ctx, e));
  }
  hipStream_t s = ctx->stream;
  auto blocks = [](long e) { return dim3((unsigned)((e + 255) / 256)); };
  hipLaunchKernelGGL(cg_init_kernel, blocks((long)N * N), dim3(256), 0, s, ar, countar, (long)ld, n, N, v[0], c[0], m[0]);
  for (size_t l = 1; l < sides.size(); ++l)
    hipLaunchKernelGGL(cg_coarsen_kernel, blocks((long)sides[l] * sides[l]), dim3(256), 0, s, v[l - 1], c[l - 1], m[l - 1], sides[l], v[l], c[l], m[l]);
  for (size_t l = sides.size() - 1; l >= 1; --l)
    hipLaunchKernelGGL(cg_refine_kernel, blocks((long)sides[l] * sides[l]), dim3(256), 0, s, v[l], m[l], sides[l], cutoff, v[l - 1], c[l - 1], m[l - 1]);
  hipLaunchKernelGGL(cg_finish_kernel, blocks((long)n * n), dim3(256), 0, s, v[0], m[0], N, n, out, (long)ld_out);
  LAUNCHCHECK("adaptive coarse-graining kernels");
  return ORCA_OK;
}

extern "C" int orca_genome_unpack_2bit(orca_ctx* ctx, const uint8_t* two_bit, const uint8_t* nmask, int64_t start, int64_t n, uint8_t* codes) {
  if (!ctx || !two_bit || !nmask || !codes) return fail(ORCA_EINVAL, "orca_genome_unpack_2bit: NULL argument");
  if (start < 0 || n < 0) return fail(ORCA_EINVAL, "orca_genome_unpack_2bit: negative window");
  if (n == 0) return ORCA_OK;
  HIPCHECK(hipSetDevice(ctx->device));
  const long nq = (n + 3) / 4;
  hipLaunchKernelGGL(genome_unpack_2bit_kernel, dim3((unsigned)((nq + 255) / 256)), dim3(256), 0, ctx->stream, two_bit, nmask, (long)start, (long)n, codes);
  LAUNCHCHECK("genome_unpack_2bit_kernel");
  return ORCA_OK;
}

