// orca_encoder.hip - Conv1d launchers, the Encoder (orca_modules.py:929-980) and the U-net encoders Encoder2 / Encoder3 / Encoder2b (:1151-1169, :1388-1406, :1262-1276)
// Part of liborca_hip.so (include/orca_hip.h is the ABI; orca_internal.h what the units share).
#include "orca_internal.h"

#include "conv_kernels.h"
#include "conv_bf16s.h"
#include "conv_stage1.h"
#include "conv_p16.h"
#include "conv_ws.h"
#include "conv_p16w1.h"
#include "conv_p16p5.h"
#include "conv_p16x.h"
#include "conv_small.h"
#include "misc_kernels.h"
#include "p16_planes.h"

// ---------------------------------------------------------------------------
// kernel launch helpers
// ---------------------------------------------------------------------------
// position tile used for long 128-channel convs: 128 (2 waves/SIMD, 4 accumulators/wave); 256 (1 wave/SIMD, 8 accumulators/wave) on request (`tile`)
static const int g_big_tile128 = 128;

template <int COUT, int MW, int NW, int WM, int WN, int KC>
static void launch_conv1d_t(hipStream_t s, const Conv1dArgs& a, int B) {
  constexpr int MT = WM * MW * 32;
  dim3 grid((unsigned)((a.n + MT - 1) / MT), (unsigned)B);
  hipLaunchKernelGGL((conv1d_k9_kernel<COUT, MW, NW, WM, WN, KC>), grid, dim3(WM * WN * 64), 0, s, a);
}

int launch_conv1d(orca_ctx* ctx, const ConvLayer& L, const float* x, long x_bs, long ldx, float* y, long y_bs,
                         long ldy, const float* r1, const float* r2, int B, long n, int relu, int tile, int y_nlc) {
  if (L.ksize != 9) return fail(ORCA_EINVAL, "launch_conv1d on a non-1d layer");
  if (n <= 0 || B <= 0) return ORCA_OK;
  Conv1dArgs a;
  a.x = x; a.w = L.d_w; a.bias = L.d_bias; a.y = y; a.r1 = r1; a.r2 = r2;
  a.x_bs = x_bs; a.y_bs = y_bs; a.ldx = ldx; a.ldy = ldy; a.n = n; a.nchunks = L.nchunks; a.relu = relu; a.y_nlc = y_nlc;
  a.x_vec_ok = al16(x) && (ldx % 4 == 0) && (x_bs % 4 == 0);
  a.y_vec_ok = al16(y) && (ldy % 4 == 0) && (y_bs % 4 == 0) && (!r1 || al16(r1)) && (!r2 || al16(r2));
  hipStream_t s = ctx->stream;
  const bool timed = ctx->timing && n >= 65536;
  TimedLaunch tl;
  if (timed) {
    HIPCHECK(hipEventCreate(&tl.e0));
    HIPCHECK(hipEventCreate(&tl.e1));
    HIPCHECK(hipEventRecord(tl.e0, s));
  }
  int used_tile = 256;
  if (L.cout == 64) {
    if (L.kc == 4) launch_conv1d_t<64, 2, 2, 4, 1, 4>(s, a, B);
    else launch_conv1d_t<64, 2, 2, 4, 1, 8>(s, a, B);
  } else if (L.cout == 96) {
    launch_conv1d_t<96, 2, 3, 4, 1, 8>(s, a, B);
  } else {
    if (tile == 0) tile = (n >= 65536) ? g_big_tile128 : (n >= 8192 ? 64 : 32);
    if (tile == 256) launch_conv1d_t<128, 4, 2, 2, 2, 8>(s, a, B);
    else if (tile == 128) launch_conv1d_t<128, 2, 2, 2, 2, 8>(s, a, B);
    else if (tile == 64) launch_conv1d_t<128, 1, 2, 2, 2, 8>(s, a, B);
    else if (tile == 32) launch_conv1d_t<128, 1, 1, 1, 4, 8>(s, a, B);
    else return fail(ORCA_EINVAL, "conv1d tile %d unsupported", tile);
    used_tile = tile;
  }
  LAUNCHCHECK("conv1d_k9_kernel");
  if (timed) {
    HIPCHECK(hipEventRecord(tl.e1, s));
    tl.rec.cout = L.cout; tl.rec.cin = L.cin; tl.rec.tile = used_tile; tl.rec.batch = B; tl.rec.n = n; tl.rec.ms = 0.f; tl.rec.ksize = 9;
    ctx->timed.push_back(tl);
  }
  return ORCA_OK;
}

int launch_pool(orca_ctx* ctx, const float* x, long ldx, float* y, long ldy, long rows, long n_out, int k) {
  if (n_out <= 0 || rows <= 0) return ORCA_OK;
  dim3 grid((unsigned)((n_out + 255) / 256), (unsigned)rows);
  switch (k) {
    case 2: hipLaunchKernelGGL((maxpool1d_kernel<2>), grid, dim3(256), 0, ctx->stream, x, ldx, y, ldy, n_out); break;
    case 4: hipLaunchKernelGGL((maxpool1d_kernel<4>), grid, dim3(256), 0, ctx->stream, x, ldx, y, ldy, n_out); break;
    case 5: hipLaunchKernelGGL((maxpool1d_kernel<5>), grid, dim3(256), 0, ctx->stream, x, ldx, y, ldy, n_out); break;
    default: return fail(ORCA_EINVAL, "maxpool k=%d unsupported", k);
  }
  LAUNCHCHECK("maxpool1d_kernel");
  return ORCA_OK;
}

// ---- bf16 split-operand conv1d (channel-last activations) --------------------------------
template <int COUT, int MW, int NW, int WM, int WN, int NS, int DT>
static void launch_b16_t(hipStream_t s, ConvB16Args a, int B) {
  constexpr int MT = WM * MW * 32;
  // persistent grid: CUs x resident workgroups per CU (queried once per instantiation)
  static int resident = [] {
    int dev = 0, ncu = 256, per_cu = 1;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev);
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, conv1d_k9_bf16s_kernel<COUT, MW, NW, WM, WN, NS, DT>, WM * WN * 64, 0) != hipSuccess || per_cu < 1) {
      (void)hipGetLastError();
      per_cu = 1;
    }
    return ncu * per_cu;
  }();
  a.tiles_per_row = (a.n + MT - 1) / MT;
  a.batch = B;
  const long ntiles = a.tiles_per_row * B;
  dim3 grid((unsigned)(ntiles < resident ? ntiles : resident));
  hipLaunchKernelGGL((conv1d_k9_bf16s_kernel<COUT, MW, NW, WM, WN, NS, DT>), grid, dim3(WM * WN * 64), 0, s, a);
}

template <int NS, int DT>
static int launch_conv1d_b16_ns(orca_ctx* ctx, const ConvLayer& L, const ConvB16Args& a, int B) {
  hipStream_t s = ctx->stream;
  if (L.cout == 64) launch_b16_t<64, 2, 2, 4, 1, NS, DT>(s, a, B);
  else if (L.cout == 96) launch_b16_t<96, 1, 3, 8, 1, NS, DT>(s, a, B);
  else if (L.cout == 128) {
    // stages 6-7 of the Encoder (16 000 / 8 000 positions) make 63 / 32 tiles of 256 positions for 256 CUs: 64-position tiles there
    if constexpr (NS <= 2) {
      if (((a.n + 255) / 256) * B < 200) { launch_b16_t<128, 1, 2, 2, 2, NS, DT>(s, a, B); return ORCA_OK; }
    }
    launch_b16_t<128, 2, 2, 4, 2, NS, DT>(s, a, B);
  }
  else return fail(ORCA_EINVAL, "bf16s conv1d cout %d unsupported", L.cout);
  return ORCA_OK;
}

// x [B][n][cin], y/r1 [B][n][cout] channel-last.  precision: ORCA_PRECISION_BF16 / _BF16X2 / _BF16X3
int launch_conv1d_b16(orca_ctx* ctx, const ConvLayer& L, int precision, const float* x, long x_bs, float* y, long y_bs,
                             const float* r1, int B, long n, int relu, int pool4, const float* r2) {
  if (!L.d_wb16) return fail(ORCA_EINVAL, "layer has no bf16 split pack (cin %d)", L.cin);
  if (n <= 0 || B <= 0) return ORCA_OK;
  ConvB16Args a;
  a.x = x; a.w = L.d_wb16; a.bias = L.d_bias; a.y = y; a.r1 = r1; a.r2 = r2; a.x_bs = x_bs; a.y_bs = y_bs; a.n = n;
  a.pool4 = pool4; a.r_bs = (long)n * L.cout;
  a.cin = L.cin; a.nchunks = L.cin / 16; a.relu = relu; a.stagger = 2;
  a.flag = ctx->d_flag;
  if (precision == ORCA_PRECISION_F16X2) {
    if (!L.f16_ok) return fail(ORCA_EINVAL, "layer weights exceed the fp16 range: use ORCA_PRECISION_BF16X3");
    a.w = L.d_wf16;
  }
  const bool timed = ctx->timing && n >= 65536;
  TimedLaunch tl;
  if (timed) {
    HIPCHECK(hipEventCreate(&tl.e0));
    HIPCHECK(hipEventCreate(&tl.e1));
    HIPCHECK(hipEventRecord(tl.e0, ctx->stream));
  }
  int rc;
  a.cout = L.cout;
  // short rows (the Encoder's stages 5-7 of a local re-encode, one- to three-bin inputs): the K-chunks of a tile side by side, one global
  // round trip and one barrier per launch instead of a chain of eight (conv_small.h: ~6 against 23-27 us).  A function of n alone, so that
  // a row's result never depends on the batch it is computed in
  if (n <= 2048 && !pool4 && L.cout % 32 == 0 && L.cin % 16 == 0) {
    const dim3 grid((unsigned)(((n + 31) / 32) * (L.cout / 32)), (unsigned)B);
    if (precision == ORCA_PRECISION_BF16X3) hipLaunchKernelGGL((conv1d_k9_small_kernel<3, 0>), grid, dim3(512), 0, ctx->stream, a);
    else if (precision == ORCA_PRECISION_BF16X2) hipLaunchKernelGGL((conv1d_k9_small_kernel<2, 0>), grid, dim3(512), 0, ctx->stream, a);
    else if (precision == ORCA_PRECISION_F16X2) hipLaunchKernelGGL((conv1d_k9_small_kernel<2, 1>), grid, dim3(512), 0, ctx->stream, a);
    else hipLaunchKernelGGL((conv1d_k9_small_kernel<1, 0>), grid, dim3(512), 0, ctx->stream, a);
    LAUNCHCHECK("conv1d_k9_small_kernel");
    ctx->counts[0]++;
    return ORCA_OK;
  }
  if (precision == ORCA_PRECISION_BF16X3) rc = launch_conv1d_b16_ns<3, 0>(ctx, L, a, B);
  else if (precision == ORCA_PRECISION_BF16X2) rc = launch_conv1d_b16_ns<2, 0>(ctx, L, a, B);
  else if (precision == ORCA_PRECISION_F16X2) rc = launch_conv1d_b16_ns<2, 1>(ctx, L, a, B);
  else rc = launch_conv1d_b16_ns<1, 0>(ctx, L, a, B);
  if (rc != ORCA_OK) return rc;
  LAUNCHCHECK("conv1d_k9_bf16s_kernel");
  ctx->counts[1]++;
  if (timed) {
    HIPCHECK(hipEventRecord(tl.e1, ctx->stream));
    tl.rec.cout = L.cout; tl.rec.cin = L.cin; tl.rec.tile = -precision; tl.rec.batch = B; tl.rec.n = n; tl.rec.ms = 0.f; tl.rec.ksize = 9;
    ctx->timed.push_back(tl);
  }
  return ORCA_OK;
}

// ---- P16 (planar split fp16) conv1d with LDS-DMA staging (conv_p16.h) -------------------------------------
// plane length in 16-byte units: P16_GUARD = 8 guard units on the left, >= 24 on the right (a 17-tap conv's second tap
// half reads 9 units past the last tile); the +1 keeps the zero stores of a pooled output's ragged last tile
// (128 * ceil(4n'/512) positions) inside the plane for every n'
int launch_p16_zero_pads(orca_ctx* ctx, float* base, int C, long n_valid, int fmt) {
  hipLaunchKernelGGL(p16_zero_pads_kernel, dim3((unsigned)(fmt == 1 ? C / 8 : C / 8 * 2)), dim3(256), 0, ctx->stream, reinterpret_cast<f32x4*>(base),
                     p16_plen(n_valid), n_valid);
  LAUNCHCHECK("p16_zero_pads_kernel");
  return ORCA_OK;
}

template <int CT, int MW, int NW, int WM, int OM, bool R1, int FMT = 0>
static void launch_p16_k(hipStream_t s, ConvP16Args a) {
  constexpr int MT = WM * MW * 32;
  static int resident = [] {
    int dev = 0, ncu = 256, per_cu = 1;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev);
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, conv1d_k9_p16_kernel<CT, MW, NW, WM, OM, R1, 0, false, FMT>, WM * 64, 0) != hipSuccess || per_cu < 1) {
      (void)hipGetLastError();
      per_cu = 1;
    }
    return ncu * per_cu;
  }();
  a.tiles_per_row = (a.n + MT - 1) / MT;
  const long ntiles = a.tiles_per_row * (a.cout / CT);
  dim3 grid((unsigned)(ntiles < resident ? ntiles : resident));
  hipLaunchKernelGGL((conv1d_k9_p16_kernel<CT, MW, NW, WM, OM, R1, 0, false, FMT>), grid, dim3(WM * 64), 0, s, a);
}

// the conv that follows the first layer, with the first layer fused into its input-tile producer (conv_p16.h, F1)
static void launch_p16_fused_first(hipStream_t s, ConvP16Args a) {
  constexpr int MT = 512;
  static int resident = [] {
    int dev = 0, ncu = 256, per_cu = 1;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev);
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, conv1d_k9_p16_kernel<64, 2, 2, 8, 0, false, 0, true>, 512, 0) != hipSuccess || per_cu < 1) {
      (void)hipGetLastError();
      per_cu = 1;
    }
    return ncu * per_cu;
  }();
  a.tiles_per_row = (a.n + MT - 1) / MT;
  const long ntiles = a.tiles_per_row;
  dim3 grid((unsigned)(ntiles < resident ? ntiles : resident));
  hipLaunchKernelGGL((conv1d_k9_p16_kernel<64, 2, 2, 8, 0, false, 0, true>), grid, dim3(512), 0, s, a);
}

// stage 1's pooled conv on P16 planes with the residual computed from the bases in its epilogue (conv_p16.h, RL)
static void launch_p16_res_bases(hipStream_t s, ConvP16Args a) {
  constexpr int FMT = 0;
  constexpr int MT = 512;
  static int resident = [] {
    int dev = 0, ncu = 256, per_cu = 1;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev);
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, conv1d_k9_p16_kernel<64, 2, 2, 8, 1, false, 0, false, FMT, true>, 512, 0) != hipSuccess || per_cu < 1) {
      (void)hipGetLastError();
      per_cu = 1;
    }
    return ncu * per_cu;
  }();
  a.tiles_per_row = (a.n + MT - 1) / MT;
  const long ntiles = a.tiles_per_row;
  dim3 grid((unsigned)(ntiles < resident ? ntiles : resident));
  hipLaunchKernelGGL((conv1d_k9_p16_kernel<64, 2, 2, 8, 1, false, 0, false, FMT, true>), grid, dim3(512), 0, s, a);
}

// the 96-cout layers of B16 planes on 512-position tiles with ONE half-by-half refilled weight buffer (conv_p16w1.h)
template <int OM, bool R1>
static void launch_p16w1_k(hipStream_t s, ConvP16Args a) {
  static int ncu = [] { int dev = 0, v = 256; (void)hipGetDevice(&dev); (void)hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev); return v; }();
  a.tiles_per_row = (a.n + 511) / 512;
  dim3 grid((unsigned)(a.tiles_per_row < ncu ? a.tiles_per_row : ncu));
  hipLaunchKernelGGL((conv1d_k9_p16w1_kernel<OM, R1, 1>), grid, dim3(512), 0, s, a);
}
// the same layers of P16 planes on the 16 x 16 x 32 matrix instruction (conv_p16x.h)
template <int OM, bool R1, int CT>
static void launch_p16x_k(hipStream_t s, ConvP16Args a) {
  static int ncu = [] { int dev = 0, v = 256; (void)hipGetDevice(&dev); (void)hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev); return v; }();
  a.tiles_per_row = (a.n + 511) / 512;
  const long ntiles = a.tiles_per_row * (a.cout / CT);
  dim3 grid((unsigned)(ntiles < ncu ? ntiles : ncu));
  hipLaunchKernelGGL((conv1d_k9_p16x_kernel<OM, R1, CT / 32, CT>), grid, dim3(1024), 0, s, a);
}
static bool launch_p16x(hipStream_t s, const ConvP16Args& a) {      // false: this (cout, out_mode, residual) combination stays on the 32 x 32 x 16 kernels
  const bool r1 = a.r1 != nullptr;
  if (a.cout == 96) {
    if (a.out_mode == 0 && !r1) launch_p16x_k<0, false, 96>(s, a);
    else if (a.out_mode == 1 && r1) launch_p16x_k<1, true, 96>(s, a);
    else return false;
  } else return false;
  return true;
}
static bool launch_p16w1(hipStream_t s, const ConvP16Args& a) {     // false: this (out_mode, residual) pair stays on the 256-position kernel
  const bool r1 = a.r1 != nullptr;
  if (a.out_mode == 0 && !r1) launch_p16w1_k<0, false>(s, a);
  else if (a.out_mode == 1 && r1) launch_p16w1_k<1, true>(s, a);
  else return false;
  return true;
}

// a 128-cout layer with ReLU, residual and MaxPool1d(5) fused (conv_p16p5.h): out_mode 3
template <int FMT>
static void launch_p16p5(hipStream_t s, ConvP16Args a) {
  static int ncu = [] { int dev = 0, v = 256; (void)hipGetDevice(&dev); (void)hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev); return v; }();
  a.tiles_per_row = (a.n + 319) / 320;
  dim3 grid((unsigned)(a.tiles_per_row < ncu ? a.tiles_per_row : ncu));
  if (a.r1) hipLaunchKernelGGL((conv1d_k9_p16p5_kernel<true, FMT>), grid, dim3(512), 0, s, a);
  else hipLaunchKernelGGL((conv1d_k9_p16p5_kernel<false, FMT>), grid, dim3(512), 0, s, a);
}


// W-stationary barrier-free form (conv_ws.h): persistent, one workgroup per CU; the grid is a multiple of the number
// of cout blocks (of 8 x that where possible: the blocks of one position range then share an XCD)
template <int FMT, int CIN, int CT, int MW, int NW, int OM, bool R1>
static void launch_ws_k(hipStream_t s, const ConvP16Args& a) {
  static int ncu = [] { int dev = 0, n = 256; (void)hipGetDevice(&dev); (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev); return n; }();
  const int ncb = a.cout / CT;
  int grid = (ncu / (8 * ncb)) * (8 * ncb);
  if (grid < ncu - 8) grid = (ncu / ncb) * ncb;
  const long ntw = (a.n + MW * 32 - 1) / (MW * 32);
  const long need = ((ntw + 7) / 8) * ncb;            // workgroups that get at least one wave tile
  if (need < grid) grid = (int)(((need + 8 * ncb - 1) / (8 * ncb)) * (8 * ncb));
  hipLaunchKernelGGL((conv1d_k9_ws_kernel<FMT, CIN, CT, MW, NW, OM, R1>), dim3((unsigned)grid), dim3(512), 0, s, a);
}
template <int FMT, int CIN, int CT, int MW, int NW>
static void launch_ws_t(hipStream_t s, const ConvP16Args& a) {
  const bool r1 = a.r1 != nullptr;
  switch (a.out_mode * 2 + (r1 ? 1 : 0)) {
    case 0: launch_ws_k<FMT, CIN, CT, MW, NW, 0, false>(s, a); break;
    case 1: launch_ws_k<FMT, CIN, CT, MW, NW, 0, true>(s, a); break;
    case 2: launch_ws_k<FMT, CIN, CT, MW, NW, 1, false>(s, a); break;
    case 3: launch_ws_k<FMT, CIN, CT, MW, NW, 1, true>(s, a); break;
    case 4: launch_ws_k<FMT, CIN, CT, MW, NW, 2, false>(s, a); break;
    default: launch_ws_k<FMT, CIN, CT, MW, NW, 2, true>(s, a); break;
  }
}

// out_mode and the residual are compile-time in the kernel (its epilogue is branch-free)
template <int CT, int MW, int NW, int WM, int FMT = 0>
static void launch_p16_t(hipStream_t s, const ConvP16Args& a) {
  const bool r1 = a.r1 != nullptr;
  switch (a.out_mode * 2 + (r1 ? 1 : 0)) {
    case 0: launch_p16_k<CT, MW, NW, WM, 0, false, FMT>(s, a); break;
    case 1: launch_p16_k<CT, MW, NW, WM, 0, true, FMT>(s, a); break;
    case 2: launch_p16_k<CT, MW, NW, WM, 1, false, FMT>(s, a); break;
    case 3: launch_p16_k<CT, MW, NW, WM, 1, true, FMT>(s, a); break;
    case 4: launch_p16_k<CT, MW, NW, WM, 2, false, FMT>(s, a); break;
    default: launch_p16_k<CT, MW, NW, WM, 2, true, FMT>(s, a); break;
  }
}

// x: P16 [cin] of n positions; y: P16 (out_mode 0: n positions, 1: n/4 pooled) or fp32 [n][cout] (2); r1: P16 [cout], n
struct FusedFirst {   // packed bases + first-layer table: the conv's input is produced instead of read (x may be NULL)
  const unsigned char* codes = nullptr;
  const unsigned char* nmask = nullptr;   // 2-bit genome window (conv_p16.h: p16_base_at)
  long origin = 0;
  long codes_L = 0, codes_off = 0;
  int reverse = 0;
  const float* table = nullptr;
  const float* bias = nullptr;
  bool residual = false;   // RL form instead: x IS read; the bases + the 17-tap pack (in `table`) give the residual in the epilogue
};

// fmt 0: P16 activations (fp32-class f16x2 arithmetic); fmt 1: B16 activations (plain bf16, BASELINE config 3)
int launch_conv1d_p16(orca_ctx* ctx, const ConvLayer& L, const float* x, void* y, const float* r1, long n, int relu,
                             int out_mode, const FusedFirst* f1, int fmt) {
  const bool k17 = L.ksize == 17;
  if ((L.ksize != 9 && !k17) || (fmt == 0 ? !L.d_wf16 : !L.d_wb16p)) return fail(ORCA_EINVAL, "layer has no %s pack", fmt == 0 ? "fp16 split" : "bf16");
  if (fmt == 0 && !L.f16_ok) return fail(ORCA_EINVAL, "layer weights exceed the fp16 range: use ORCA_PRECISION_BF16X3");
  if (n <= 0) return ORCA_OK;
  ctx->counts[2]++;
  ConvP16Args a;
  a.x = reinterpret_cast<const f32x4*>(x); a.w = reinterpret_cast<const f32x4*>(fmt == 0 ? L.d_wf16 : L.d_wb16p); a.bias = L.d_bias; a.y = y;
  a.r1 = reinterpret_cast<const f32x4*>(r1); a.x_plen = p16_plen(n); a.y_plen = p16_plen(out_mode == 1 ? n / 4 : out_mode == 3 ? n / 5 : n); a.n = n;
  a.nchunks = (fmt == 0 ? L.cin / 16 : L.cin / 32) * (k17 ? 2 : 1); a.cout = L.cout; a.relu = relu; a.out_mode = out_mode; a.flag = ctx->d_flag;
  a.k17 = k17 ? 1 : 0;
  const bool timed = ctx->timing && n >= 65536;
  TimedLaunch tl;
  if (timed) {
    HIPCHECK(hipEventCreate(&tl.e0));
    HIPCHECK(hipEventCreate(&tl.e1));
    HIPCHECK(hipEventRecord(tl.e0, ctx->stream));
  }
  a.f1_codes = nullptr; a.f1_nmask = nullptr; a.f1_origin = 0; a.f1_codes_L = a.f1_codes_off = 0; a.f1_reverse = 0; a.f1_table = a.f1_bias = nullptr; a.stamps = nullptr; a.rl_w = nullptr;
  const bool ws_ok = fmt == 1 && !k17;     // W-stationary barrier-free kernel (conv_ws.h; P16: measured 3 % slower than the tiled kernel - profiles/HISTORY.md)
  int tile_tag = fmt == 1 ? -6 : -5;
  if (f1 && f1->residual) {
    if (L.cout != 64 || L.cin != 64 || out_mode != 1 || r1 || k17) return fail(ORCA_EINVAL, "residual from the bases: only stage 1's pooled 64 -> 64 planar conv");
    a.f1_codes = f1->codes; a.f1_nmask = f1->nmask; a.f1_origin = f1->origin; a.f1_codes_L = f1->codes_L; a.f1_codes_off = f1->codes_off; a.f1_reverse = f1->reverse;
    a.rl_w = reinterpret_cast<const f32x4*>(f1->table); a.f1_bias = f1->bias;
    if (fmt == 1) return fail(ORCA_EINVAL, "residual from the bases on B16 planes: the stage runs as ONE kernel (conv_stage1.h)");
    launch_p16_res_bases(ctx->stream, a);
  } else if (f1) {
    if (L.cout != 64 || L.cin != 64 || out_mode != 0 || r1 || fmt != 0 || k17) return fail(ORCA_EINVAL, "fused first layer: only the 64 -> 64 P16 conv that follows it");
    a.f1_codes = f1->codes; a.f1_nmask = f1->nmask; a.f1_origin = f1->origin; a.f1_codes_L = f1->codes_L; a.f1_codes_off = f1->codes_off; a.f1_reverse = f1->reverse;
    a.f1_table = f1->table; a.f1_bias = f1->bias;
    launch_p16_fused_first(ctx->stream, a);
  } else if (ws_ok && fmt == 1 && L.cin == 64 && L.cout == 64) {
    launch_ws_t<1, 64, 64, 2, 2>(ctx->stream, a);
    tile_tag = -8;
  } else if (out_mode == 3) {
    if (k17 || L.cout != 128) return fail(ORCA_EINVAL, "fused MaxPool1d(5): only the 128-cout k9 conv (conv_p16p5.h)");
    if (fmt == 1) launch_p16p5<1>(ctx->stream, a); else launch_p16p5<0>(ctx->stream, a);
    tile_tag = fmt == 1 ? -13 : -12;
  } else if (L.cout == 96 && n >= 65536 && fmt == 0 && launch_p16x(ctx->stream, a)) {
    tile_tag = -14;                      // stage 2 of the Encoder on P16 planes: 16 x 16 x 32 matrix instruction (conv_p16x.h)
  } else if (L.cout == 96 && n >= 65536 && fmt == 1 && launch_p16w1(ctx->stream, a)) {
    tile_tag = -10;                      // ... on B16 planes: 512-position tiles, one half-by-half refilled weight buffer (conv_p16w1.h)
  } else if (fmt == 1) {
    if (L.cout == 96) launch_p16_t<96, 1, 3, 8, 1>(ctx->stream, a);
    else if (L.cout % 64 == 0) launch_p16_t<64, 2, 2, 8, 1>(ctx->stream, a);
    else return fail(ORCA_EINVAL, "b16 conv1d cout %d unsupported", L.cout);
  } else
  if (L.cout == 96) launch_p16_t<96, 1, 3, 8>(ctx->stream, a);
  else if (L.cout % 64 == 0) launch_p16_t<64, 2, 2, 8>(ctx->stream, a);
  else return fail(ORCA_EINVAL, "p16 conv1d cout %d unsupported", L.cout);
  LAUNCHCHECK("conv1d_k9_p16_kernel");
  if (timed) {
    HIPCHECK(hipEventRecord(tl.e1, ctx->stream));
    tl.rec.cout = L.cout; tl.rec.cin = L.cin; tl.rec.tile = tile_tag; tl.rec.batch = 1; tl.rec.n = n; tl.rec.ms = 0.f; tl.rec.ksize = k17 ? 17 : 9;
    ctx->timed.push_back(tl);
  }
  return ORCA_OK;
}

// the end positions of a composed linear group, recomputed conv by conv (conv_p16.h: lconv_edge_layer_kernel).  layers[l] with
// relu[l]; the last layer covers `half_last` positions per end, layer l four more per layer behind it; ys[l] (may be NULL) receives
// the outermost stores[l] positions per end of layer l.  Scratch: one 40 x 128 float slab per layer in the context.
static int launch_edge_chain(orca_ctx* ctx, const ConvLayer* const* layers, const int* relu, int nl, int half_last, EdgeFixArgs src, long n,
                             float* const* ys, const int* stores, int fmt, long ld_f32 = 0) {
  src.n = n;
  if (nl > 4 || 2 * (half_last + 4 * (nl - 1)) > 40) return fail(ORCA_EINVAL, "edge fix: chain too deep");
  for (int l = 0; l < nl; ++l) {
    const ConvLayer& L = *layers[l];
    if (!L.d_w) return fail(ORCA_EINVAL, "edge fix: layer without an fp32 pack");
    EdgeLayerArgs a{};
    a.in = src;
    if (l > 0) a.in.in_mode = -1;
    a.half = half_last + 4 * (nl - 1 - l); a.half_in = a.half + 4;
    a.relu = relu[l]; a.cin = L.cin; a.cout = L.cout; a.kc = L.kc; a.w = L.d_w; a.b = L.d_bias;
    a.sin = l > 0 ? ctx->d_edge + (l - 1) * ORCA_EDGE_SLAB : nullptr;
    a.sout = ctx->d_edge + l * ORCA_EDGE_SLAB;
    a.y = reinterpret_cast<f32x4*>(ys[l]); a.y_plen = fmt == 2 ? ld_f32 : p16_plen(n); a.out_fmt = fmt;      // fmt 2: fp32 channel-major, row stride ld_f32
    a.store_half = stores[l];
    hipLaunchKernelGGL(lconv_edge_layer_kernel, dim3((unsigned)(2 * a.half)), dim3(512), 0, ctx->stream, a);
  }
  LAUNCHCHECK("lconv_edge_layer_kernel");
  return ORCA_OK;
}

static int launch_pool_nlc(orca_ctx* ctx, const float* x, float* y, long n_out, int C, int k) {
  if (n_out <= 0) return ORCA_OK;
  if (C % 4 || C / 4 > 256) return fail(ORCA_EINVAL, "maxpool (channel-last): %d channels unsupported", C);
  const long per_block = 2 * (256 / (C / 4));
  dim3 grid((unsigned)((n_out + per_block - 1) / per_block));
  switch (k) {
    case 2: hipLaunchKernelGGL((maxpool1d_nlc_kernel<2>), grid, dim3(256), 0, ctx->stream, x, y, n_out, C); break;
    case 4: hipLaunchKernelGGL((maxpool1d_nlc_kernel<4>), grid, dim3(256), 0, ctx->stream, x, y, n_out, C); break;
    case 5: hipLaunchKernelGGL((maxpool1d_nlc_kernel<5>), grid, dim3(256), 0, ctx->stream, x, y, n_out, C); break;
    default: return fail(ORCA_EINVAL, "maxpool k=%d unsupported", k);
  }
  LAUNCHCHECK("maxpool1d_nlc_kernel");
  return ORCA_OK;
}

static int launch_copy2d(orca_ctx* ctx, const float* src, long lds_, long scol, float* dst, long ldd, long rows, long cols) {
  if (rows <= 0 || cols <= 0) return ORCA_OK;
  dim3 grid((unsigned)((cols + 255) / 256), (unsigned)rows);
  hipLaunchKernelGGL(copy2d_kernel, grid, dim3(256), 0, ctx->stream, src, lds_, scol, dst, ldd, cols);
  LAUNCHCHECK("copy2d_kernel");
  return ORCA_OK;
}

// ---------------------------------------------------------------------------
// Encoder (orca_modules.py:929-980)
// ---------------------------------------------------------------------------
static const int kEncPools[7] = {1, 4, 4, 5, 5, 5, 2};
static const long kHaloBp = 112000;  // x_padding, orca_modules.py:932
static const long kBinBp = 4000;

extern "C" int64_t orca_encoder_num_bins(int64_t L) {
  long n = L;
  for (int i = 1; i < 7; ++i) n /= kEncPools[i];
  return n;
}

// One chunk: x (strided [4][n1]) -> 128 x n7 in the returned buffer.
struct SeqSource {            // where a chunk's input comes from: a float [.,4] view or packed base codes
  const float* x = nullptr;   // already offset to the chunk start
  long sx_c = 0, sx_l = 0;
  const unsigned char* codes = nullptr;   // whole sequence of this batch row: 1 byte per base, or (nmask set) a 2-bit genome plane + N mask, the
  const unsigned char* nmask = nullptr;   // sequence starting at genome index `origin` (conv_p16.h: p16_base_at)
  long origin = 0;
  long codes_L = 0, codes_off = 0;
  int reverse = 0;
};

// `part` (P16 planes only, SV screens off the 4 kb grid - orca_encoder_stage3_planes and friends below): the FRONT of the Encoder is stages 1-3
// (99 % of its FLOPs; translation-covariant on a 16-base grid with a reach of 351 bases), the BACK stages 4-7 from the MaxPool1d(5)'d stage-3 output
// (stage 4 - 1.2 of the back's 2.0 ms per window - moves into the cache too: ENC_STAGE4 runs it alone on a stage-4 input and hands its fp32 rows over,
// ENC_FRONT4 = the front + stage 4 for the snippets, ENC_BACK5 = stages 5-7 from the MaxPool1d(5)'d rows: covariant on an 80-base grid, reach 1 631 bases)
enum { ENC_FULL = 0, ENC_FRONT_POOLED = 1, ENC_FRONT_UNPOOLED = 2, ENC_BACK = 3, ENC_STAGE4 = 4, ENC_FRONT4 = 5, ENC_BACK5 = 6 };

static int encoder_chunk(orca_ctx* ctx, orca_net* net, const SeqSource& src, long n1, float* const buf[3],
                         long ld1, float** out, long* out_ld, long* out_n, int part = ENC_FULL) {
  hipStream_t s = ctx->stream;
  int P = 0;
  long n = n1, ld = ld1;
  const float* x = src.x;
  long sx_c = src.sx_c, sx_l = src.sx_l;
  // planar 16-bit activation formats of conv_p16.h for stages 1-3 (96 % of the FLOPs):
  //   f16x2 -> P16 (2-way split fp16, fp32-class);  bf16 -> B16 (one bf16 plane, throughput mode of BASELINE config 3)
  const bool use_b16 = net->precision == ORCA_PRECISION_BF16;
  const bool use_p16 = net->precision == ORCA_PRECISION_F16X2 || use_b16;
  const int fmt = use_b16 ? 1 : 0;
  if (part != ENC_FULL && !(use_p16 && fmt == 0)) return fail(ORCA_EINVAL, "the Encoder's front / back parts exist in the f16x2 arithmetic (P16 planes) only");
  // the channel-last split-operand pipeline (bf16x3 / bf16x2): stage 1 composed - from PACKED bases only: the
  // first-layer GEMM splits its X operand into fp16 parts, exact for 0 / 0.25 / 1, while these modes promise fp32 range for arbitrary float rows
  const bool compose_nlc = !use_p16 && net->precision != ORCA_PRECISION_F32 && net->d_c1a_w16 && net->enc_form < ORCA_ENCODER_FORM_LCONV1_ONLY && src.codes;
  // exact-fp32 mode: stage 1's linear groups composed as in the 16-bit modes (conv_p16.h: first_taps_f32_kernel reads the source directly)
  const bool compose32 = net->precision == ORCA_PRECISION_F32 && net->d_l1_f32 && net->enc_form < ORCA_ENCODER_FORM_LCONV1_ONLY;
  if (src.codes && !use_p16 && !compose32 && !compose_nlc) {
    // the other arithmetic modes start from float rows: expand the packed bases into buf[2] as [n][4]
    hipLaunchKernelGGL(expand_codes_kernel, dim3((unsigned)((n1 + 255) / 256)), dim3(256), 0, s, src.codes, src.nmask, src.origin, src.codes_L, src.codes_off,
                       src.reverse, n1, buf[2]);
    LAUNCHCHECK("expand_codes_kernel");
    x = buf[2]; sx_c = 1; sx_l = 4;
  }
  if (!use_p16 && !compose32 && !compose_nlc) {
    hipLaunchKernelGGL(seq_to_channel_major_kernel, dim3((unsigned)((n1 + 255) / 256)), dim3(256), 0, s, x, sx_c, sx_l, n1, buf[P], ld1);
    LAUNCHCHECK("seq_to_channel_major_kernel");
  }
  if (net->precision != ORCA_PRECISION_F32) {
    // channel-last pipeline on the bf16 matrix cores (conv_bf16s.h); the 4-channel first layer stays on
    // the fp32 kernel (K = 36, 1 % of the FLOPs) and writes channel-last.
    const int prec = net->precision;
    int st0 = part == ENC_BACK5 ? 4 : 0;          // (ENC_BACK5: stage 5's input - n1 positions, fp32 channel-last - is in buf[0])
    if (use_p16 && part != ENC_BACK5) {
      // stages 1-3 on planar 16-bit activations with LDS-DMA staging (conv_p16.h)
      const ConvLayer* L = net->convs.data();
      FirstP16Args fa;
      fa.x = x; fa.sc = sx_c; fa.sl = sx_l; fa.n = n1; fa.w = nullptr; fa.bias = L[0].d_bias; fa.y = reinterpret_cast<f32x4*>(buf[1]);
      fa.y_plen = p16_plen(n1); fa.flag = ctx->d_flag;
      fa.w = net->d_first_w;
      // Composed linear pairs (default; orca_net_set_encoder_form(ORCA_ENCODER_FORM_TWO_CONV) = the reference's layer sequence - also what a group
      // whose composed weights leave the fp16 range falls back to): lconv1 is ONE
      // 17-tap first layer straight into buf[LO] (K = 68 MFMA GEMM from the bases / float rows), lconv2 / lconv3 are 17-tap planar
      // convs; conv1.a, linear up to its ReLU, is composed with lconv1 as well (25 taps from the bases).  The end positions of each
      // group are redone exactly by the edge-fix chain (lconv_edge_layer_kernel).
      const bool compose = net->enc_form < ORCA_ENCODER_FORM_TWO_CONV && net->d_l1_w16 != nullptr;
      // two-conv form from packed input: the first layer is fused into the input-tile producer of the conv that follows it (conv_p16.h, F1)
      const bool fuse1 = src.codes && fmt == 0 && !compose;
      FusedFirst f1;
      f1.codes = src.codes; f1.nmask = src.nmask; f1.origin = src.origin; f1.codes_L = src.codes_L; f1.codes_off = src.codes_off; f1.reverse = src.reverse;
      f1.table = net->d_first_tab; f1.bias = L[0].d_bias;
      int T = 1, LO = 2, S = 0;   // buffer roles: T holds the current input
      // (guards and tails of every planar tensor are zeroed by p16_zero_pads_kernel AFTER its producer: the conv kernels
      // write the units of a ragged last tile unmasked)
      const bool flat = src.codes || (sx_c == 1 && sx_l == 4 && al16(x));
      // conv1.a joins the composed group (25 taps from the bases + ReLU, K = 112); _FORM_LCONV1_ONLY keeps it a 64 -> 64 launch
      const bool compose25 = compose && net->enc_form < ORCA_ENCODER_FORM_LCONV1_ONLY && net->d_c1a_w16 != nullptr;
      // ... and with packed bases the residual lout1 is computed inside conv1.b's epilogue (conv_p16.h, RL) instead of being stored by a
      // 17-tap first-layer launch and re-read (float rows, _FORM_STORED_RESIDUAL: the stored form)
      const bool res_from_bases = compose25 && src.codes && net->enc_form < ORCA_ENCODER_FORM_STORED_RESIDUAL;
      // ... and in the throughput mode (B16 planes) the whole stage is then ONE kernel from the bases (conv_stage1.h): conv1.b's input tiles are
      // produced in LDS by the matrix cores, a1 is neither written nor re-read
      const bool stage1_fused = res_from_bases && fmt == 1;
      float* first_out = compose ? buf[LO] : buf[T];
      const float* rows = x;     // flat [n][4] float rows for the MFMA first-layer kernels (unused with packed input)
      // one first-layer GEMM launch: ntap 9 (lconv1.a alone), 17 (lconv1 composed), 25 (conv1.a o lconv1, + ReLU)
      auto launch_first = [&](int ntap, const void* w16, const float* bias, int relu, float* out) -> int {
        FirstMfmaArgs fm;
        fm.codes = src.codes; fm.nmask = src.nmask; fm.origin = src.origin; fm.codes_L = src.codes_L; fm.codes_off = src.codes_off; fm.reverse = src.reverse;
        fm.x = src.codes ? nullptr : rows; fm.n = n1;
        fm.w = reinterpret_cast<const f32x4*>(w16); fm.bias = bias; fm.relu = relu;
        fm.y = reinterpret_cast<f32x4*>(out); fm.y_plen = p16_plen(n1); fm.flag = ctx->d_flag;
        const long nt = (n1 + 255) / 256;
        const dim3 grid((unsigned)(nt < 2048 ? nt : 2048));
        const bool timed = ctx->timing && n1 >= 65536 && ntap > 9;
        TimedLaunch tl;
        if (timed) {
          HIPCHECK(hipEventCreate(&tl.e0));
          HIPCHECK(hipEventCreate(&tl.e1));
          HIPCHECK(hipEventRecord(tl.e0, s));
        }
        switch (ntap * 2 + fmt) {
          case 18: hipLaunchKernelGGL((conv1d_first_mfma_p16_kernel<0, 0, 9>), grid, dim3(256), 0, s, fm); break;
          case 19: hipLaunchKernelGGL((conv1d_first_mfma_p16_kernel<0, 1, 9>), grid, dim3(256), 0, s, fm); break;
          case 34: hipLaunchKernelGGL((conv1d_first_mfma_p16_kernel<0, 0, 17>), grid, dim3(256), 0, s, fm); break;
          case 35: hipLaunchKernelGGL((conv1d_first_mfma_p16_kernel<0, 1, 17>), grid, dim3(256), 0, s, fm); break;
          case 50: hipLaunchKernelGGL((conv1d_first_mfma_p16_kernel<0, 0, 25>), grid, dim3(256), 0, s, fm); break;
          default: hipLaunchKernelGGL((conv1d_first_mfma_p16_kernel<0, 1, 25>), grid, dim3(256), 0, s, fm); break;
        }
        LAUNCHCHECK("conv1d_first_mfma_p16_kernel");
        if (timed) {
          HIPCHECK(hipEventRecord(tl.e1, s));
          tl.rec.cout = 64; tl.rec.cin = 4; tl.rec.tile = fmt == 1 ? -6 : -5; tl.rec.batch = 1; tl.rec.n = n1; tl.rec.ms = 0.f; tl.rec.ksize = ntap;
          ctx->timed.push_back(tl);
        }
        return ORCA_OK;
      };
      if (part == ENC_BACK || part == ENC_STAGE4) {
        // stage 4's input (n1 positions, 128 channels) is in buf[S] already
      } else {
      if (fuse1) {
        // nothing to launch: buf[1] is never materialised
      } else if (compose || flat || fmt == 1) {
        if (!flat) {
          // strided float rows: gather them into a flat [n][4] copy (buf[S] is free until the stage's last conv), then the MFMA kernel
          hipLaunchKernelGGL(seq_to_rows_kernel, dim3((unsigned)((n1 + 255) / 256)), dim3(256), 0, s, x, sx_c, sx_l, n1, buf[S]);
          LAUNCHCHECK("seq_to_rows_kernel");
          rows = buf[S];
        }
        if (res_from_bases) { /* no lout1 tensor */ }
        else if (compose) ORCA_TRY(launch_first(17, net->d_l1_w16, net->d_l1_bias, 0, buf[LO]));
        else ORCA_TRY(launch_first(9, net->d_first_w16, L[0].d_bias, 0, buf[T]));
        if (compose25 && !stage1_fused) {
          ORCA_TRY(launch_first(25, net->d_c1a_w16, net->d_c1a_bias, 1, buf[T]));
          ORCA_TRY(launch_p16_zero_pads(ctx, buf[T], 64, n1, fmt));
        }
      } else {
        hipLaunchKernelGGL(conv1d_first_p16_kernel, dim3((unsigned)((n1 + 255) / 256), 8), dim3(256), 0, s, fa);
        LAUNCHCHECK("conv1d_first_p16_kernel");
      }
      if (!fuse1 && !res_from_bases) ORCA_TRY(launch_p16_zero_pads(ctx, first_out, 64, n1, fmt));
      if (compose) {
        EdgeFixArgs ef{};
        if (src.codes) { ef.in_mode = 1; ef.codes = src.codes; ef.nmask = src.nmask; ef.origin = src.origin; ef.codes_L = src.codes_L; ef.codes_off = src.codes_off; ef.reverse = src.reverse; }
        else { ef.in_mode = 0; ef.x = x; ef.sc = sx_c; ef.sl = sx_l; }
        const ConvLayer* chain[4] = {&L[0], &L[1], &L[2], &L[3]};
        const int relus[4] = {0, 0, 1, 1};
        if (res_from_bases) {        // lout1 is never stored; conv1.b's own end positions are needed for the pooled windows (after the conv, below)
          float* ys[4] = {nullptr, nullptr, buf[T], nullptr};
          const int st_[4] = {0, 0, 8, 0};
          ORCA_TRY(launch_edge_chain(ctx, chain, relus, 4, 8, ef, n1, ys, st_, fmt));
        } else if (compose25) {
          float* ys[3] = {nullptr, buf[LO], buf[T]};
          const int st_[3] = {0, 4, 8};
          ORCA_TRY(launch_edge_chain(ctx, chain, relus, 3, 8, ef, n1, ys, st_, fmt));
        } else {
          float* ys[2] = {nullptr, buf[LO]};
          const int st_[2] = {0, 4};
          ORCA_TRY(launch_edge_chain(ctx, chain, relus, 2, 4, ef, n1, ys, st_, fmt));
        }
      }
      }
      n = n1;
      const int nplanar = 4;                                                // stages on the planar kernels (pools 4, 4, 5 fused into the conv in front of them)
      const bool from4 = part == ENC_BACK || part == ENC_STAGE4;
      for (st0 = from4 ? 3 : 0; st0 < nplanar; ++st0) {
        const ConvLayer* Ls = L + 4 * st0;
        const int C = Ls[3].cout;
        if (kEncPools[st0] == 5 && !from4) n /= 5;   // the previous stage's last conv already pooled (conv_p16p5.h): buf[S] holds n / 5 positions
        // the stage's linear pair: (pooled) previous output buf[S] -> lout in buf[LO]
        const bool comp_st = compose && (st0 == 0 || (st0 <= 2 && net->comp[st0].d_wf16));
        if (comp_st && st0 > 0) {
          ORCA_TRY(launch_conv1d_p16(ctx, net->comp[st0], buf[S], buf[LO], nullptr, n, 0, 0, nullptr, fmt));
          ORCA_TRY(launch_p16_zero_pads(ctx, buf[LO], C, n, fmt));
          EdgeFixArgs ef{};
          ef.in_mode = fmt == 1 ? 3 : 2; ef.xp = reinterpret_cast<const f32x4*>(buf[S]); ef.x_plen = p16_plen(n);
          const ConvLayer* chain[2] = {&Ls[0], &Ls[1]};
          const int relus[2] = {0, 0};
          float* ys[2] = {nullptr, buf[LO]};
          const int st_[2] = {0, 4};
          ORCA_TRY(launch_edge_chain(ctx, chain, relus, 2, 4, ef, n, ys, st_, fmt));
        } else if (!comp_st) {
          if (st0 > 0) {  // first conv of the stage: previous (pooled) output in buf[S] -> buf[T]
            ORCA_TRY(launch_conv1d_p16(ctx, Ls[0], buf[S], buf[T], nullptr, n, 0, 0, nullptr, fmt));
            ORCA_TRY(launch_p16_zero_pads(ctx, buf[T], C, n, fmt));
          }
          ORCA_TRY(launch_conv1d_p16(ctx, Ls[1], buf[T], buf[LO], nullptr, n, 0, 0, (st0 == 0 && fuse1) ? &f1 : nullptr, fmt));   // lout
          ORCA_TRY(launch_p16_zero_pads(ctx, buf[LO], C, n, fmt));
        }
        if (!(st0 == 0 && compose25)) {     // (stage 1, composed: conv1.a's output is already in buf[T], straight from the bases)
          ORCA_TRY(launch_conv1d_p16(ctx, Ls[2], buf[LO], buf[T], nullptr, n, 1, 0, nullptr, fmt));
          ORCA_TRY(launch_p16_zero_pads(ctx, buf[T], C, n, fmt));
        }
        if (st0 == 0 && res_from_bases) {
          FusedFirst rl;     // relu(.) + lout1 computed from the bases in the epilogue, MaxPool1d(4)
          rl.codes = src.codes; rl.nmask = src.nmask; rl.origin = src.origin; rl.codes_L = src.codes_L; rl.codes_off = src.codes_off; rl.reverse = src.reverse;
          rl.table = reinterpret_cast<const float*>(net->d_l1_w16); rl.bias = net->d_l1_bias; rl.residual = true;
          if (stage1_fused) {
            Stage1Args sa;
            ConvP16Args& a1 = sa.c;
            a1.x = nullptr; a1.w = reinterpret_cast<const f32x4*>(Ls[3].d_wb16p); a1.bias = Ls[3].d_bias; a1.y = buf[S]; a1.r1 = nullptr;
            a1.x_plen = p16_plen(n); a1.y_plen = p16_plen(n / 4); a1.n = n; a1.tiles_per_row = (n + 63) / 64; a1.nchunks = 2; a1.cout = 64;
            a1.relu = 1; a1.out_mode = 1; a1.k17 = 0; a1.flag = ctx->d_flag; a1.stamps = nullptr;
            a1.f1_codes = rl.codes; a1.f1_nmask = rl.nmask; a1.f1_origin = rl.origin; a1.f1_codes_L = rl.codes_L; a1.f1_codes_off = rl.codes_off; a1.f1_reverse = rl.reverse;
            a1.rl_w = reinterpret_cast<const f32x4*>(net->d_l1_w16); a1.f1_table = nullptr; a1.f1_bias = net->d_l1_bias;
            sa.w25 = reinterpret_cast<const f32x4*>(net->d_c1a_w16); sa.b25 = net->d_c1a_bias;
            sa.a1_edge = reinterpret_cast<const f32x4*>(buf[T]); sa.a1_plen = p16_plen(n);
            static int ncu_ = [] { int dev = 0, v = 256; (void)hipGetDevice(&dev); (void)hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev); return v; }();
            const long ntw_ = (n + 63) / 64;
            const long need_ = (ntw_ + 7) / 8;
            const bool timed_ = ctx->timing && n >= 65536;
            TimedLaunch tl_;
            if (timed_) {
              HIPCHECK(hipEventCreate(&tl_.e0));
              HIPCHECK(hipEventCreate(&tl_.e1));
              HIPCHECK(hipEventRecord(tl_.e0, s));
            }
            hipLaunchKernelGGL((conv1d_stage1_b16_kernel<1>), dim3((unsigned)(need_ < ncu_ ? need_ : ncu_)), dim3(512), 0, s, sa);
            LAUNCHCHECK("conv1d_stage1_b16_kernel");
            ctx->counts[2]++;
            if (timed_) {
              HIPCHECK(hipEventRecord(tl_.e1, s));
              // (both 64 -> 64 convs of the stage in one launch: recorded as cin = 128 so that 2 * 9 * cin * cout is the pair's algorithmic work)
              tl_.rec.cout = 64; tl_.rec.cin = 128; tl_.rec.tile = -15; tl_.rec.batch = 1; tl_.rec.n = n; tl_.rec.ms = 0.f; tl_.rec.ksize = 9;
              ctx->timed.push_back(tl_);
            }
          } else {
            ORCA_TRY(launch_conv1d_p16(ctx, Ls[3], buf[T], buf[S], nullptr, n, 1, 1, &rl, fmt));
          }
          ORCA_TRY(launch_p16_zero_pads(ctx, buf[S], C, n / 4, fmt));
          EdgePoolArgs ep{};
          ep.sc = ctx->d_edge + 3 * ORCA_EDGE_SLAB; ep.half_c = 8; ep.sl = ctx->d_edge + 1 * ORCA_EDGE_SLAB; ep.half_l = 16; ep.n = n; ep.cout = C;
          ep.y = reinterpret_cast<f32x4*>(buf[S]); ep.y_plen = p16_plen(n / 4); ep.out_fmt = fmt;
          if (n / 4 > 0) hipLaunchKernelGGL(lconv_edge_pool_kernel, dim3(3), dim3(128), 0, s, ep);
          LAUNCHCHECK("lconv_edge_pool_kernel");
          n /= 4;
        } else if (st0 + 1 < nplanar && kEncPools[st0 + 1] == 4) {
          ORCA_TRY(launch_conv1d_p16(ctx, Ls[3], buf[T], buf[S], buf[LO], n, 1, 1, nullptr, fmt));          // relu(.)+lout, MaxPool1d(4)
          ORCA_TRY(launch_p16_zero_pads(ctx, buf[S], C, n / 4, fmt));
          n /= 4;
        } else if (st0 + 1 < nplanar) {      // (the pool in front of stage 4)
          if (kEncPools[st0 + 1] != 5 || C != 128) return fail(ORCA_EINVAL, "internal: planar stage %d followed by an unexpected pool", st0 + 1);
          if (part == ENC_FRONT_UNPOOLED) {   // stage 3's output as it is, every position: what a stage-3 cache keeps (the pool is the reader's)
            ORCA_TRY(launch_conv1d_p16(ctx, Ls[3], buf[T], buf[S], buf[LO], n, 1, 0, nullptr, fmt));
            ORCA_TRY(launch_p16_zero_pads(ctx, buf[S], C, n, fmt));
            *out = buf[S]; *out_ld = 0; *out_n = n;
            return ORCA_OK;
          }
          ORCA_TRY(launch_conv1d_p16(ctx, Ls[3], buf[T], buf[S], buf[LO], n, 1, 3, nullptr, fmt));          // relu(.)+lout, MaxPool1d(5) (conv_p16p5.h)
          ORCA_TRY(launch_p16_zero_pads(ctx, buf[S], C, n / 5, fmt));
          if (part == ENC_FRONT_POOLED) { *out = buf[S]; *out_ld = 0; *out_n = n / 5; return ORCA_OK; }
        } else {
          ORCA_TRY(launch_conv1d_p16(ctx, Ls[3], buf[T], buf[S], buf[LO], n, 1, 2, nullptr, fmt));          // fp32 channel-last hand-over
        }
      }
      P = S;
      if (part == ENC_STAGE4 || part == ENC_FRONT4) { *out = buf[P]; *out_ld = -128; *out_n = n; return ORCA_OK; }     // stage 4's output, fp32 [n][128], before MaxPool1d(5)
    }
    for (int st = st0; st < 7; ++st) {
      const ConvLayer* L = &net->convs[4 * st];
      if (kEncPools[st] == 4 && st0 == 0) {
        n = n / 4;  // MaxPool1d(4) was fused into the epilogue of the previous stage's last conv
      } else if (kEncPools[st] > 1 && !(part == ENC_BACK5 && st == 4)) {
        const long n2 = n / kEncPools[st];
        const int Q = (P + 1) % 3;
        ORCA_TRY(launch_pool_nlc(ctx, buf[P], buf[Q], n2, L[0].cin, kEncPools[st]));
        P = Q; n = n2;
      }
      const int T = (P + 1) % 3, LO = (P + 2) % 3;
      if (st == 0 && compose_nlc) {
        // stage 1's linear groups composed here too (the fallback of the fp16-range guard runs this branch in bf16x3): lconv1 and
        // conv1.a o lconv1 as 17- / 25-tap first-layer GEMMs writing fp32 channel-last, exact ends by the edge chain
        FirstMfmaArgs fm;
        fm.codes = src.codes; fm.nmask = src.nmask; fm.origin = src.origin; fm.codes_L = src.codes_L; fm.codes_off = src.codes_off; fm.reverse = src.reverse;
        fm.x = src.codes ? nullptr : src.x; fm.n = n; fm.y_plen = 0; fm.flag = nullptr;
        const long nt = (n + 255) / 256;
        const dim3 grid((unsigned)(nt < 2048 ? nt : 2048));
        fm.w = reinterpret_cast<const f32x4*>(net->d_l1_w16); fm.bias = net->d_l1_bias; fm.relu = 0; fm.y = reinterpret_cast<f32x4*>(buf[LO]);
        hipLaunchKernelGGL((conv1d_first_mfma_p16_kernel<0, 2, 17>), grid, dim3(256), 0, s, fm);
        fm.w = reinterpret_cast<const f32x4*>(net->d_c1a_w16); fm.bias = net->d_c1a_bias; fm.relu = 1; fm.y = reinterpret_cast<f32x4*>(buf[T]);
        hipLaunchKernelGGL((conv1d_first_mfma_p16_kernel<0, 2, 25>), grid, dim3(256), 0, s, fm);
        LAUNCHCHECK("conv1d_first_mfma_p16_kernel");
        EdgeFixArgs ef{};
        if (src.codes) { ef.in_mode = 1; ef.codes = src.codes; ef.nmask = src.nmask; ef.origin = src.origin; ef.codes_L = src.codes_L; ef.codes_off = src.codes_off; ef.reverse = src.reverse; }
        else { ef.in_mode = 0; ef.x = src.x; ef.sc = src.sx_c; ef.sl = src.sx_l; }
        const ConvLayer* chain[3] = {&L[0], &L[1], &L[2]};
        const int relus[3] = {0, 0, 1};
        float* ys[3] = {nullptr, buf[LO], buf[T]};
        const int st_[3] = {0, 4, 8};
        ORCA_TRY(launch_edge_chain(ctx, chain, relus, 3, 8, ef, n, ys, st_, 3));
      } else {
      if (st == 0) ORCA_TRY(launch_conv1d(ctx, L[0], buf[P], 0, ld1, buf[T], 0, 0, nullptr, nullptr, 1, n, 0, 0, 1));
      else ORCA_TRY(launch_conv1d_b16(ctx, L[0], prec, buf[P], 0, buf[T], 0, nullptr, 1, n, 0));
      ORCA_TRY(launch_conv1d_b16(ctx, L[1], prec, buf[T], 0, buf[LO], 0, nullptr, 1, n, 0));
      ORCA_TRY(launch_conv1d_b16(ctx, L[2], prec, buf[LO], 0, buf[T], 0, nullptr, 1, n, 1));
      }
      ORCA_TRY(launch_conv1d_b16(ctx, L[3], prec, buf[T], 0, buf[P], 0, st < 6 ? buf[LO] : nullptr, 1, n, 1,
                                 (st < 6 && kEncPools[st + 1] == 4) ? 1 : 0));
    }
    *out = buf[P]; *out_ld = -128; *out_n = n;   // negative ld: result is channel-last [n][128]
    return ORCA_OK;
  }
  int cprev = 4;
  for (int st = 0; st < 7; ++st) {
    const ConvLayer* L = &net->convs[4 * st];
    if (kEncPools[st] > 1) {
      const long n2 = n / kEncPools[st], ld2 = ru4(n2);
      const int Q = (P + 1) % 3;
      ORCA_TRY(launch_pool(ctx, buf[P], ld, buf[Q], ld2, cprev, n2, kEncPools[st]));
      P = Q; n = n2; ld = ld2;
    }
    const int T = (P + 1) % 3, LO = (P + 2) % 3;
    if (st == 0 && compose32) {
      // lconv1 (17 taps) and conv1.a o lconv1 (25 taps + ReLU) straight from the source, fp32 FMAs; exact ends by the edge chain
      FirstF32Args fa{};
      if (src.codes) { fa.in.in_mode = 1; fa.in.codes = src.codes; fa.in.nmask = src.nmask; fa.in.origin = src.origin; fa.in.codes_L = src.codes_L; fa.in.codes_off = src.codes_off; fa.in.reverse = src.reverse; }
      else { fa.in.in_mode = 0; fa.in.x = src.x; fa.in.sc = src.sx_c; fa.in.sl = src.sx_l; }
      fa.in.n = n; fa.ldy = ld;
      const long nt = (n + 127) / 128;
      const dim3 grid((unsigned)(nt < 4096 ? nt : 4096));
      fa.w = net->d_l1_f32; fa.bias = net->d_l1_bias32; fa.relu = 0; fa.y = buf[LO];
      hipLaunchKernelGGL((first_taps_f32_kernel<17>), grid, dim3(256), 0, s, fa);
      fa.w = net->d_c1a_f32; fa.bias = net->d_c1a_bias32; fa.relu = 1; fa.y = buf[T];
      hipLaunchKernelGGL((first_taps_f32_kernel<25>), grid, dim3(256), 0, s, fa);
      LAUNCHCHECK("first_taps_f32_kernel");
      const ConvLayer* chain[3] = {&L[0], &L[1], &L[2]};
      const int relus[3] = {0, 0, 1};
      float* ys[3] = {nullptr, buf[LO], buf[T]};
      const int st_[3] = {0, 4, 8};
      ORCA_TRY(launch_edge_chain(ctx, chain, relus, 3, 8, fa.in, n, ys, st_, 2, ld));
    } else {
    ORCA_TRY(launch_conv1d(ctx, L[0], buf[P], 0, ld, buf[T], 0, ld, nullptr, nullptr, 1, n, 0, 0));
    ORCA_TRY(launch_conv1d(ctx, L[1], buf[T], 0, ld, buf[LO], 0, ld, nullptr, nullptr, 1, n, 0, 0));
    ORCA_TRY(launch_conv1d(ctx, L[2], buf[LO], 0, ld, buf[T], 0, ld, nullptr, nullptr, 1, n, 1, 0));
    }
    ORCA_TRY(launch_conv1d(ctx, L[3], buf[T], 0, ld, buf[P], 0, ld, st < 6 ? buf[LO] : nullptr, nullptr, 1, n, 1, 0));
    cprev = L[3].cout;
  }
  *out = buf[P]; *out_ld = ld; *out_n = n;
  return ORCA_OK;
}

static int encoder_forward_impl(orca_ctx* ctx, orca_net* net, const float* x, int64_t sx_b, int64_t sx_c, int64_t sx_l,
                                const unsigned char* codes, int64_t sc_b, int reverse,
                                int B, int64_t L, int64_t bin_lo, int64_t bin_hi, float* out, int64_t so_b,
                                int64_t so_c, int64_t chunk_bp, int64_t win_origin = 0, int64_t win_len = -1,
                                const unsigned char* nmask = nullptr, int64_t two_origin = 0) {
  if (!ctx || !net || (!x && !codes) || !out) return fail(ORCA_EINVAL, "orca_encoder_forward: NULL argument");
  if (net->kind != ORCA_NET_ENCODER) return fail(ORCA_EINVAL, "orca_encoder_forward: net is not an Encoder");
  HIPCHECK(hipSetDevice(ctx->device));
  const long total = orca_encoder_num_bins(L);
  if (bin_hi <= 0) bin_hi = total;
  if (bin_lo < 0 || bin_lo > bin_hi || bin_hi > total) return fail(ORCA_EINVAL, "bin range [%ld,%ld) outside [0,%ld)", (long)bin_lo, (long)bin_hi, total);
  if (bin_lo == bin_hi || B <= 0) return ORCA_OK;
  bool chunk_auto = false;
  if (chunk_bp <= 0) {
    // a 32 Mb window is one chunk; longer inputs (the 256 Mb models) run in 128 Mb chunks: 98 GB of workspace (3 x 64 channels x 4 B per base) of
    // the 288 GB, a quarter of the chunk seams (each costs a 224 kb halo and one latency-bound pass through stages 5-7): 418 -> 406 ms per
    // genomepredict_256Mb call against 32 Mb chunks, 412 with 64 Mb (same box).  $ORCA_ENCODER_CHUNK_BP overrides.
    // The chunk size does not change a result beyond fp32 round-off of the last stages (tests/test_gpu_e2e.py::test_encoder_256mb_chunk_sizes).
    // The choice is DETERMINISTIC (free memory varies per rank and per run and does not count what torch's allocator holds): the
    // largest of 128 / 64 / 32 Mb whose workspace (768 B per base) stays under 40 % of the device's TOTAL memory divided by
    // $ORCA_RANKS_PER_DEVICE (default 1; bench.py and the tests set it when several ranks share one GPU).
    const char* e = getenv("ORCA_ENCODER_CHUNK_BP");
    chunk_bp = e ? atol(e) : 32000000L;
    chunk_auto = !e;
    if (!e && L > 32000000L) {
      size_t fr = 0, tot = 0;
      const char* r = getenv("ORCA_RANKS_PER_DEVICE");
      const long rpd = r && atol(r) > 0 ? atol(r) : 1;
      if (hipMemGetInfo(&fr, &tot) == hipSuccess)
        for (long c : {128000000L, 64000000L})
          if ((double)(c + 2 * kHaloBp + 4096) * 768.0 <= 0.40 * (double)tot / (double)rpd) { chunk_bp = c; break; }
    }
  }
  if (chunk_bp % kBinBp) return fail(ORCA_EINVAL, "chunk_bp must be a multiple of 4000");
  long chunk_bins = 0, ld1 = 0;
  for (;;) {
    chunk_bins = chunk_bp / kBinBp;
    long max_n1 = 0;
    for (long cb0 = bin_lo; cb0 < bin_hi; cb0 += chunk_bins) {
      const long cb1 = cb0 + chunk_bins < bin_hi ? cb0 + chunk_bins : bin_hi;
      const long lo = cb0 * kBinBp - kHaloBp > 0 ? cb0 * kBinBp - kHaloBp : 0;
      const long hi = (cb1 == total) ? L : (cb1 * kBinBp + kHaloBp < L ? cb1 * kBinBp + kHaloBp : L);
      if (hi - lo > max_n1) max_n1 = hi - lo;
    }
    ld1 = ru4(max_n1) + 1024;   // slack: P16 planes are padded to 512 positions + guards
    const int rc = ws_ensure(ctx, 3 * ru256((size_t)64 * ld1 * sizeof(float)));
    if (rc == ORCA_OK) break;
    // the deterministic choice above looks at TOTAL memory; if the device cannot give that much right now (another process, ranks
    // sharing it without $ORCA_RANKS_PER_DEVICE, torch's allocator holding most of it) fall back chunk size by chunk size - same result
    if (rc != ORCA_ENOMEM || !chunk_auto || chunk_bp <= 32000000L) return rc;
    (void)hipGetLastError();
    chunk_bp /= 2;
  }
  float* buf[3];
  for (int i = 0; i < 3; ++i) buf[i] = ws_take(ctx, (size_t)64 * ld1);
  for (int b = 0; b < B; ++b) {
    for (long cb0 = bin_lo; cb0 < bin_hi; cb0 += chunk_bins) {
      const long cb1 = cb0 + chunk_bins < bin_hi ? cb0 + chunk_bins : bin_hi;
      const long lo = cb0 * kBinBp - kHaloBp > 0 ? cb0 * kBinBp - kHaloBp : 0;
      const long hi = (cb1 == total) ? L : (cb1 * kBinBp + kHaloBp < L ? cb1 * kBinBp + kHaloBp : L);
      float* res; long rld, rn;
      SeqSource src;
      if (codes && win_len >= 0) {
        // the caller holds only bases [win_origin, win_origin + win_len) of the L-base sequence: this chunk reads strand positions
        // [lo, hi) = bases [lo, hi) (forward) or [L - hi, L - lo) (reverse complement)
        const long b0 = reverse ? L - hi : lo, b1 = reverse ? L - lo : hi;
        if (b0 < win_origin || b1 > win_origin + win_len)
          return fail(ORCA_EINVAL, "code window [%ld,%ld) does not cover bases [%ld,%ld) needed for bins [%ld,%ld) (112 kb halo included)",
                      (long)win_origin, (long)(win_origin + win_len), b0, b1, (long)cb0, (long)cb1);
      }
      if (codes) { src.codes = codes + (nmask ? 0 : (long)b * sc_b - win_origin); src.nmask = nmask; src.origin = two_origin; src.codes_L = L; src.codes_off = lo; src.reverse = reverse; }
      else { src.x = x + (long)b * sx_b + lo * sx_l; src.sx_c = sx_c; src.sx_l = sx_l; }
      ORCA_TRY(encoder_chunk(ctx, net, src, hi - lo, buf, ru4(hi - lo), &res, &rld, &rn));
      const long keep = cb0 - lo / kBinBp;
      if (keep + (cb1 - cb0) > rn) return fail(ORCA_EINVAL, "internal: chunk produced %ld bins, need %ld", rn, keep + (cb1 - cb0));
      if (rld < 0)  // channel-last result [bins][128] -> out[c][bin]
        ORCA_TRY(launch_copy2d(ctx, res + keep * 128, 1, 128, out + (long)b * so_b + (cb0 - bin_lo), so_c, 128, cb1 - cb0));
      else
        ORCA_TRY(launch_copy2d(ctx, res + keep, rld, 1, out + (long)b * so_b + (cb0 - bin_lo), so_c, 128, cb1 - cb0));
    }
  }
  return ORCA_OK;
}

extern "C" int orca_encoder_forward(orca_ctx* ctx, orca_net* net, const float* x, int64_t sx_b, int64_t sx_c, int64_t sx_l,
                                    int B, int64_t L, int64_t bin_lo, int64_t bin_hi, float* out, int64_t so_b,
                                    int64_t so_c, int64_t chunk_bp) {
  return encoder_forward_impl(ctx, net, x, sx_b, sx_c, sx_l, nullptr, 0, 0, B, L, bin_lo, bin_hi, out, so_b, so_c, chunk_bp);
}

extern "C" int orca_encoder_forward_codes(orca_ctx* ctx, orca_net* net, const uint8_t* codes, int64_t sc_b, int reverse, int B,
                                          int64_t L, int64_t bin_lo, int64_t bin_hi, float* out, int64_t so_b, int64_t so_c,
                                          int64_t chunk_bp) {
  return encoder_forward_impl(ctx, net, nullptr, 0, 0, 0, codes, sc_b, reverse, B, L, bin_lo, bin_hi, out, so_b, so_c, chunk_bp);
}

extern "C" int orca_encoder_forward_codes_window(orca_ctx* ctx, orca_net* net, const uint8_t* codes, int64_t sc_b, int64_t win_origin, int64_t win_len,
                                                 int reverse, int B, int64_t L, int64_t bin_lo, int64_t bin_hi, float* out, int64_t so_b,
                                                 int64_t so_c, int64_t chunk_bp) {
  if (win_origin < 0 || win_len < 0 || win_origin + win_len > L) return fail(ORCA_EINVAL, "code window [%ld,+%ld) outside the %ld-base sequence", (long)win_origin, (long)win_len, (long)L);
  return encoder_forward_impl(ctx, net, nullptr, 0, 0, 0, codes, sc_b, reverse, B, L, bin_lo, bin_hi, out, so_b, so_c, chunk_bp, win_origin, win_len);
}

// Encoder straight from a 2-bit genome resident in HBM (orca_amd/genome.py TwoBitGenome: 2 bits per base + 1 N bit = 3/8 byte per base):
// the sequence is bases [start, start + L) of the chromosome whose planes are `two` / `nmask`; the one-hot expansion of a base happens
// where the 1-byte codes are expanded - in LDS inside the first-layer kernels / the residual-from-the-bases epilogue (conv_p16.h:
// p16_base_at) - so neither the unpack pass (orca_genome_unpack_2bit) nor the 1 byte/base window ever exists.
extern "C" int orca_encoder_forward_2bit(orca_ctx* ctx, orca_net* net, const uint8_t* two, const uint8_t* nmask, int64_t start, int reverse,
                                         int64_t L, int64_t bin_lo, int64_t bin_hi, float* out, int64_t so_c, int64_t chunk_bp) {
  if (!two || !nmask || start < 0) return fail(ORCA_EINVAL, "orca_encoder_forward_2bit: NULL plane or negative start");
  return encoder_forward_impl(ctx, net, nullptr, 0, 0, 0, two, 0, reverse, 1, L, bin_lo, bin_hi, out, 0, so_c, chunk_bp, 0, -1, nmask, start);
}

// ---------------------------------------------------------------------------
// The Encoder in two parts (f16x2 / P16 planes): stage-3 cache of a chromosome -> stage-4 input of a window -> bins.
// Stages 1-3 hold 99 % of the Encoder's FLOPs and are translation-covariant on a 16-base grid (pools 4 x 4) with a reach of 351 bases
// (orca_modules.py:811-852): their output on a chromosome, kept once per strand and per phase mod 16 (512 bytes per base and strand), serves
// EVERY window of every allele built from pieces of that chromosome, whatever its phase on the 4 kb grid - the reference's structural-variant
// drivers place their windows at the variant's own phase (orca_predict.py:1613).  A window then costs a MaxPool1d(5) gather from the cache, the
// front on a few kb around its ends and junctions, and stages 4-7.
// ---------------------------------------------------------------------------
extern "C" int64_t orca_p16_plane_units(int64_t n) { return p16_plen(n); }

static int front_run(orca_ctx* ctx, orca_net* net, const uint8_t* codes, int64_t L, int reverse, int64_t base0, int64_t nbases, int part, float** res, long* rn) {
  if (!ctx || !net || !codes) return fail(ORCA_EINVAL, "encoder front: NULL argument");
  if (net->kind != ORCA_NET_ENCODER) return fail(ORCA_EINVAL, "encoder front: net is not an Encoder");
  if (net->precision != ORCA_PRECISION_F16X2) return fail(ORCA_EINVAL, "encoder front: f16x2 arithmetic only (P16 planes)");
  if (nbases <= 0 || nbases % 80 || base0 < 0 || base0 + nbases > L) return fail(ORCA_EINVAL, "encoder front: bases [%ld,+%ld) of %ld: a positive multiple of 80 inside the sequence", (long)base0, (long)nbases, (long)L);
  HIPCHECK(hipSetDevice(ctx->device));
  const long ld1 = ru4(nbases) + 1024;
  ORCA_TRY(ws_ensure(ctx, 3 * ru256((size_t)64 * ld1 * sizeof(float))));
  float* buf[3];
  for (int i = 0; i < 3; ++i) buf[i] = ws_take(ctx, (size_t)64 * ld1);
  SeqSource src;
  src.codes = codes; src.codes_L = L; src.codes_off = base0; src.reverse = reverse;
  long rld;
  return encoder_chunk(ctx, net, src, nbases, buf, ru4(nbases), res, &rld, rn, part);
}

extern "C" int orca_encoder_stage3_planes(orca_ctx* ctx, orca_net* net, const uint8_t* codes, int64_t L, int reverse, float* planes, int64_t plane_units) {
  if (!planes) return fail(ORCA_EINVAL, "orca_encoder_stage3_planes: NULL argument");
  float* res; long rn;
  ORCA_TRY(front_run(ctx, net, codes, L, reverse, 0, L, ENC_FRONT_UNPOOLED, &res, &rn));
  if (plane_units != p16_plen(rn)) return fail(ORCA_EINVAL, "orca_encoder_stage3_planes: planes of %ld units, %ld positions need %ld", (long)plane_units, rn, p16_plen(rn));
  HIPCHECK(hipMemcpyAsync(planes, res, (size_t)32 * plane_units * 16, hipMemcpyDeviceToDevice, ctx->stream));
  return ORCA_OK;
}

extern "C" int orca_p16_pool5_into(orca_ctx* ctx, const float* src, int64_t src_units, int64_t src_pos0, float* dst, int64_t dst_units, int64_t dst_pos0, int64_t count) {
  if (!ctx || !src || !dst) return fail(ORCA_EINVAL, "orca_p16_pool5_into: NULL argument");
  if (count <= 0) return ORCA_OK;
  if (src_pos0 < 0 || dst_pos0 < 0 || P16_GUARD + src_pos0 + 5 * count > src_units || P16_GUARD + dst_pos0 + count > dst_units)
    return fail(ORCA_EINVAL, "orca_p16_pool5_into: positions [%ld,+5 x %ld) / [%ld,+%ld) outside planes of %ld / %ld units", (long)src_pos0, (long)count, (long)dst_pos0, (long)count, (long)src_units, (long)dst_units);
  HIPCHECK(hipSetDevice(ctx->device));
  hipLaunchKernelGGL(p16_pool5_into_kernel, dim3((unsigned)((count + 255) / 256), 16), dim3(256), 0, ctx->stream, reinterpret_cast<const f32x4*>(src), (long)src_units, (long)src_pos0,
                     reinterpret_cast<f32x4*>(dst), (long)dst_units, (long)dst_pos0, (long)count);
  LAUNCHCHECK("p16_pool5_into_kernel");
  return ORCA_OK;
}

extern "C" int orca_encoder_front_snippet(orca_ctx* ctx, orca_net* net, const uint8_t* codes, int64_t L, int reverse, int64_t base0, int64_t nbases, int64_t skip,
                                          int64_t count, float* dst, int64_t dst_units, int64_t dst_pos0) {
  if (!dst) return fail(ORCA_EINVAL, "orca_encoder_front_snippet: NULL argument");
  float* res; long rn;
  ORCA_TRY(front_run(ctx, net, codes, L, reverse, base0, nbases, ENC_FRONT_POOLED, &res, &rn));
  if (skip < 0 || count <= 0 || skip + count > rn || dst_pos0 < 0 || P16_GUARD + dst_pos0 + count > dst_units)
    return fail(ORCA_EINVAL, "orca_encoder_front_snippet: pooled positions [%ld,+%ld) of %ld -> [%ld,..) of planes of %ld units", (long)skip, (long)count, rn, (long)dst_pos0, (long)dst_units);
  hipLaunchKernelGGL(p16_copy_units_kernel, dim3((unsigned)((count + 255) / 256), 32), dim3(256), 0, ctx->stream, reinterpret_cast<const f32x4*>(res), p16_plen(rn), (long)skip,
                     reinterpret_cast<f32x4*>(dst), (long)dst_units, (long)dst_pos0, (long)count);
  LAUNCHCHECK("p16_copy_units_kernel");
  return ORCA_OK;
}

extern "C" int orca_encoder_back(orca_ctx* ctx, orca_net* net, const float* s4, int64_t s4_units, int64_t n4, float* out, int64_t so_c) {
  if (!ctx || !net || !s4 || !out) return fail(ORCA_EINVAL, "orca_encoder_back: NULL argument");
  if (net->kind != ORCA_NET_ENCODER || net->precision != ORCA_PRECISION_F16X2) return fail(ORCA_EINVAL, "orca_encoder_back: an Encoder net in the f16x2 arithmetic");
  if (n4 <= 0 || n4 % 50 || s4_units != p16_plen(n4)) return fail(ORCA_EINVAL, "orca_encoder_back: %ld stage-4 positions (a multiple of 50) in planes of %ld units (need %ld)", (long)n4, (long)s4_units, p16_plen(n4));
  HIPCHECK(hipSetDevice(ctx->device));
  const long ld = ru4(n4) + 1024;                         // per buffer: 128 channels x ld floats (P16 planes of 128 channels; fp32 [n][128] behind stage 4)
  ORCA_TRY(ws_ensure(ctx, 3 * ru256((size_t)128 * ld * sizeof(float))));
  float* buf[3];
  for (int i = 0; i < 3; ++i) buf[i] = ws_take(ctx, (size_t)128 * ld);
  HIPCHECK(hipMemcpyAsync(buf[0], s4, (size_t)32 * s4_units * 16, hipMemcpyDeviceToDevice, ctx->stream));
  ORCA_TRY(launch_p16_zero_pads(ctx, buf[0], 128, n4, 0));
  SeqSource src;
  float* res; long rld, rn;
  ORCA_TRY(encoder_chunk(ctx, net, src, n4, buf, ru4(n4), &res, &rld, &rn, ENC_BACK));
  if (rld >= 0 || rn != n4 / 50) return fail(ORCA_EINVAL, "internal: the Encoder's back part produced %ld bins for %ld positions", rn, (long)n4);
  return launch_copy2d(ctx, res, 1, 128, out, so_c, 128, rn);
}

// ---- one level further: stage 4 in the cache (fp32 rows [n][128] on the 80-base grid, phases mod 80; same 512 bytes per base and strand) ----
extern "C" int orca_encoder_stage4_rows(orca_ctx* ctx, orca_net* net, const float* s4, int64_t s4_units, int64_t n4, float* rows) {
  if (!ctx || !net || !s4 || !rows) return fail(ORCA_EINVAL, "orca_encoder_stage4_rows: NULL argument");
  if (net->kind != ORCA_NET_ENCODER || net->precision != ORCA_PRECISION_F16X2) return fail(ORCA_EINVAL, "orca_encoder_stage4_rows: an Encoder net in the f16x2 arithmetic");
  if (n4 <= 0 || s4_units != p16_plen(n4)) return fail(ORCA_EINVAL, "orca_encoder_stage4_rows: %ld positions in planes of %ld units (need %ld)", (long)n4, (long)s4_units, p16_plen(n4));
  HIPCHECK(hipSetDevice(ctx->device));
  const long ld = ru4(n4) + 1024;
  ORCA_TRY(ws_ensure(ctx, 3 * ru256((size_t)128 * ld * sizeof(float))));
  float* buf[3];
  for (int i = 0; i < 3; ++i) buf[i] = ws_take(ctx, (size_t)128 * ld);
  HIPCHECK(hipMemcpyAsync(buf[0], s4, (size_t)32 * s4_units * 16, hipMemcpyDeviceToDevice, ctx->stream));
  ORCA_TRY(launch_p16_zero_pads(ctx, buf[0], 128, n4, 0));
  SeqSource src;
  float* res; long rld, rn;
  ORCA_TRY(encoder_chunk(ctx, net, src, n4, buf, ru4(n4), &res, &rld, &rn, ENC_STAGE4));
  if (rld >= 0 || rn != n4) return fail(ORCA_EINVAL, "internal: stage 4 produced %ld rows for %ld positions", rn, (long)n4);
  HIPCHECK(hipMemcpyAsync(rows, res, (size_t)n4 * 128 * sizeof(float), hipMemcpyDeviceToDevice, ctx->stream));
  return ORCA_OK;
}

static int launch_rows_pool5(orca_ctx* ctx, const float* src, long src_pos0, float* dst, long dst_pos0, long count) {
  if (count <= 0) return ORCA_OK;
  hipLaunchKernelGGL(rows_pool5_into_kernel, dim3((unsigned)((count * 32 + 255) / 256)), dim3(256), 0, ctx->stream, reinterpret_cast<const f32x4*>(src), src_pos0,
                     reinterpret_cast<f32x4*>(dst), dst_pos0, count);
  LAUNCHCHECK("rows_pool5_into_kernel");
  return ORCA_OK;
}

extern "C" int orca_rows_pool5_into(orca_ctx* ctx, const float* src, int64_t src_rows, int64_t src_pos0, float* dst, int64_t dst_rows, int64_t dst_pos0, int64_t count) {
  if (!ctx || !src || !dst) return fail(ORCA_EINVAL, "orca_rows_pool5_into: NULL argument");
  if (count < 0 || src_pos0 < 0 || dst_pos0 < 0 || src_pos0 + 5 * count > src_rows || dst_pos0 + count > dst_rows)
    return fail(ORCA_EINVAL, "orca_rows_pool5_into: rows [%ld,+5 x %ld) of %ld -> [%ld,+%ld) of %ld", (long)src_pos0, (long)count, (long)src_rows, (long)dst_pos0, (long)count, (long)dst_rows);
  HIPCHECK(hipSetDevice(ctx->device));
  return launch_rows_pool5(ctx, src, (long)src_pos0, dst, (long)dst_pos0, (long)count);
}

extern "C" int orca_encoder_front4_snippet(orca_ctx* ctx, orca_net* net, const uint8_t* codes, int64_t L, int reverse, int64_t base0, int64_t nbases, int64_t skip,
                                           int64_t count, float* dst, int64_t dst_rows, int64_t dst_pos0) {
  if (!dst) return fail(ORCA_EINVAL, "orca_encoder_front4_snippet: NULL argument");
  if (base0 % 400 || nbases % 400) return fail(ORCA_EINVAL, "orca_encoder_front4_snippet: bases [%ld,+%ld): multiples of 400 (the pooled rows must line up with the window's)", (long)base0, (long)nbases);
  float* res; long rn;
  ORCA_TRY(front_run(ctx, net, codes, L, reverse, base0, nbases, ENC_FRONT4, &res, &rn));
  if (skip < 0 || count <= 0 || 5 * (skip + count) > rn || dst_pos0 < 0 || dst_pos0 + count > dst_rows)
    return fail(ORCA_EINVAL, "orca_encoder_front4_snippet: pooled rows [%ld,+%ld) of %ld -> [%ld,..) of %ld", (long)skip, (long)count, rn / 5, (long)dst_pos0, (long)dst_rows);
  return launch_rows_pool5(ctx, res, 5 * (long)skip, dst, (long)dst_pos0, (long)count);
}

// the same for SEVERAL ranges of one front run: the snippets of a window strand concatenated (the window's ends first and last, so that the run's own ends
// are the window's; the seams between snippets lie inside the margins nobody reads) - one chain of ~44 small launches per strand instead of one per snippet
extern "C" int orca_encoder_front4_ranges(orca_ctx* ctx, orca_net* net, const uint8_t* codes, int64_t L, int reverse, int n_ranges, const int64_t* ranges_host,
                                          float* dst, int64_t dst_rows) {
  if (!dst || !ranges_host || n_ranges <= 0) return fail(ORCA_EINVAL, "orca_encoder_front4_ranges: NULL / empty argument");
  if (L % 400) return fail(ORCA_EINVAL, "orca_encoder_front4_ranges: %ld bases: a multiple of 400", (long)L);
  float* res; long rn;
  ORCA_TRY(front_run(ctx, net, codes, L, reverse, 0, L, ENC_FRONT4, &res, &rn));
  for (int k = 0; k < n_ranges; ++k) {
    const long skip = (long)ranges_host[3 * k], count = (long)ranges_host[3 * k + 1], pos0 = (long)ranges_host[3 * k + 2];
    if (skip < 0 || count <= 0 || 5 * (skip + count) > rn || pos0 < 0 || pos0 + count > dst_rows)
      return fail(ORCA_EINVAL, "orca_encoder_front4_ranges: range %d: pooled rows [%ld,+%ld) of %ld -> [%ld,..) of %ld", k, skip, count, rn / 5, pos0, (long)dst_rows);
    ORCA_TRY(launch_rows_pool5(ctx, res, 5 * skip, dst, pos0, count));
  }
  return ORCA_OK;
}

extern "C" int orca_encoder_back5(orca_ctx* ctx, orca_net* net, const float* rows, int64_t n5, float* out, int64_t so_c) {
  if (!ctx || !net || !rows || !out) return fail(ORCA_EINVAL, "orca_encoder_back5: NULL argument");
  if (net->kind != ORCA_NET_ENCODER || net->precision != ORCA_PRECISION_F16X2) return fail(ORCA_EINVAL, "orca_encoder_back5: an Encoder net in the f16x2 arithmetic");
  if (n5 <= 0 || n5 % 10) return fail(ORCA_EINVAL, "orca_encoder_back5: %ld stage-5 positions (a multiple of 10)", (long)n5);
  HIPCHECK(hipSetDevice(ctx->device));
  const long ld = ru4(n5) + 1024;
  ORCA_TRY(ws_ensure(ctx, 3 * ru256((size_t)128 * ld * sizeof(float))));
  float* buf[3];
  for (int i = 0; i < 3; ++i) buf[i] = ws_take(ctx, (size_t)128 * ld);
  HIPCHECK(hipMemcpyAsync(buf[0], rows, (size_t)n5 * 128 * sizeof(float), hipMemcpyDeviceToDevice, ctx->stream));
  SeqSource src;
  float* res; long rld, rn;
  ORCA_TRY(encoder_chunk(ctx, net, src, n5, buf, ru4(n5), &res, &rld, &rn, ENC_BACK5));
  if (rld >= 0 || rn != n5 / 10) return fail(ORCA_EINVAL, "internal: stages 5-7 produced %ld bins for %ld positions", rn, (long)n5);
  return launch_copy2d(ctx, res, 1, 128, out, so_c, 128, rn);
}

extern "C" int orca_pack_sequence(orca_ctx* ctx, const float* x, int64_t sx_c, int64_t sx_l, int64_t L, uint8_t* codes, int* packable) {
  if (!ctx || !x || !codes || !packable) return fail(ORCA_EINVAL, "orca_pack_sequence: NULL argument");
  HIPCHECK(hipSetDevice(ctx->device));
  HIPCHECK(hipMemsetAsync(ctx->d_flag + 1, 0, sizeof(unsigned), ctx->stream));
  hipLaunchKernelGGL(pack_sequence_kernel, dim3((unsigned)((L + 255) / 256)), dim3(256), 0, ctx->stream, x, sx_c, sx_l, L, codes, ctx->d_flag + 1);
  LAUNCHCHECK("pack_sequence_kernel");
  unsigned h = 0;
  HIPCHECK(hipMemcpyAsync(&h, ctx->d_flag + 1, sizeof h, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHECK(hipStreamSynchronize(ctx->stream));
  *packable = h ? 0 : 1;
  return ORCA_OK;
}

// ---------------------------------------------------------------------------
// Encoder2 / Encoder3 (orca_modules.py:1151-1169, :1388-1406)
// ---------------------------------------------------------------------------
static int launch_transpose(orca_ctx* ctx, const float* src, long lds_, long src_bs, float* dst, long ldd, long dst_bs, long rows, long cols, int B) {
  if (rows <= 0 || cols <= 0 || B <= 0) return ORCA_OK;
  dim3 grid((unsigned)((cols + 31) / 32), (unsigned)((rows + 31) / 32), (unsigned)B);
  hipLaunchKernelGGL(transpose2d_kernel, grid, dim3(32, 8), 0, ctx->stream, src, lds_, src_bs, dst, ldd, dst_bs, rows, cols);
  LAUNCHCHECK("transpose2d_kernel");
  return ORCA_OK;
}

// The U-net encoders on the 16-bit matrix cores (conv_bf16s.h: channel-last fp32 activations [B][n][128], split operands).  Same graph as
// the fp32 path below; the skip connection of the expanding path is the kernel's second residual and the result of a level overwrites the
// contracting-path encoding it consumed.  The outputs are handed over channel-major ([B][128][n], the C ABI's layout) by tiled transposes.
static int unet_forward_nlc(orca_ctx* ctx, orca_net* net, const float* x, int64_t sx_b, int64_t sx_c, int64_t sx_l, int B, int n,
                            float* const* outs, int nlev, bool up_only) {
  const int prec = net->precision;
  const size_t full = (size_t)B * 128 * n;
  size_t need = 3 * ru256(full * sizeof(float));
  for (int i = 0; i <= nlev; ++i) need += ru256((full >> i) * sizeof(float));
  ORCA_TRY(ws_ensure(ctx, need));
  std::vector<float*> encs(nlev + 1);
  for (int i = 0; i <= nlev; ++i) encs[i] = ws_take(ctx, full >> i);
  float* t0 = ws_take(ctx, full);
  float* t1 = ws_take(ctx, full);
  float* t2 = ws_take(ctx, full);
  hipStream_t s = ctx->stream;
  // the (possibly strided) channel-major input -> [B][n][128]
  if (sx_l == 1) ORCA_TRY(launch_transpose(ctx, x, sx_c, sx_b, encs[0], 128, (long)n * 128, 128, n, B));
  else
    for (int b = 0; b < B; ++b) ORCA_TRY(launch_copy2d(ctx, x + (long)b * sx_b, sx_l, sx_c, encs[0] + (size_t)b * n * 128, 128, n, 128));
  const ConvLayer* L = net->convs.data();
  for (int i = 0; i < nlev; ++i) {      // contracting path
    const long no = n >> (i + 1), bs = 128 * no;
    ORCA_TRY(launch_pool_nlc(ctx, encs[i], t0, (long)B * no, 128, 2));          // rows of all batch entries in one pass (n >> i is even)
    ORCA_TRY(launch_conv1d_b16(ctx, L[4 * i + 0], prec, t0, bs, t1, bs, nullptr, B, no, 0));
    ORCA_TRY(launch_conv1d_b16(ctx, L[4 * i + 1], prec, t1, bs, t2, bs, nullptr, B, no, 0));   // lout
    ORCA_TRY(launch_conv1d_b16(ctx, L[4 * i + 2], prec, t2, bs, t1, bs, nullptr, B, no, 1));
    ORCA_TRY(launch_conv1d_b16(ctx, L[4 * i + 3], prec, t1, bs, encs[i + 1], bs, t2, B, no, 1));
  }
  auto hand_over = [&](int lev) { const long nl = n >> lev; return launch_transpose(ctx, encs[lev], 128, nl * 128, outs[lev], nl, 128 * nl, nl, 128, B); };
  ORCA_TRY(hand_over(nlev));
  if (up_only) {
    for (int lev = 0; lev < nlev; ++lev) ORCA_TRY(hand_over(lev));
    return ORCA_OK;
  }
  const float* cur = encs[nlev];
  for (int i = 0; i < nlev; ++i) {      // expanding path
    const int lev = nlev - 1 - i;
    const long no = n >> lev, bs = 128 * no;
    const long total = (long)B * no * 32;
    hipLaunchKernelGGL(upsample1d_x2_nlc_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, cur, t0, (long)B * no, 128);
    LAUNCHCHECK("upsample1d_x2_nlc_kernel");
    const ConvLayer* D = L + 4 * nlev + 4 * i;
    ORCA_TRY(launch_conv1d_b16(ctx, D[0], prec, t0, bs, t1, bs, nullptr, B, no, 0));
    ORCA_TRY(launch_conv1d_b16(ctx, D[1], prec, t1, bs, t2, bs, nullptr, B, no, 0));           // lout
    ORCA_TRY(launch_conv1d_b16(ctx, D[2], prec, t2, bs, t1, bs, nullptr, B, no, 1));
    ORCA_TRY(launch_conv1d_b16(ctx, D[3], prec, t1, bs, encs[lev], bs, t2, B, no, 1, 0, encs[lev]));   // + lout + skip, in place of the skip
    ORCA_TRY(hand_over(lev));
    cur = encs[lev];
  }
  return ORCA_OK;
}

extern "C" int orca_unet_forward(orca_ctx* ctx, orca_net* net, const float* x, int64_t sx_b, int64_t sx_c, int64_t sx_l, int B,
                                 int n, float* const* outs, int n_outs) {
  if (!ctx || !net || !x || !outs) return fail(ORCA_EINVAL, "orca_unet_forward: NULL argument");
  if (net->kind != ORCA_NET_ENCODER2 && net->kind != ORCA_NET_ENCODER3 && net->kind != ORCA_NET_ENCODER2B)
    return fail(ORCA_EINVAL, "orca_unet_forward: wrong net kind");
  HIPCHECK(hipSetDevice(ctx->device));
  const bool up_only = net->kind == ORCA_NET_ENCODER2B;   // Encoder2b: the contracting path IS the output
  const int nlev = net->kind == ORCA_NET_ENCODER3 ? 3 : 5;
  if (n_outs != nlev + 1) return fail(ORCA_EINVAL, "expected %d output pointers, got %d", nlev + 1, n_outs);
  if (n <= 0 || (n % (1 << nlev))) return fail(ORCA_EINVAL, "length %d not divisible by %d", n, 1 << nlev);
  if (B <= 0) return ORCA_OK;
  // Every precision but "f32" runs the channel-last split-operand path: levels of <= 2048 positions on conv_small.h (the K-chunks of a tile
  // side by side: ~7 us per launch where the exact fp32 kernel's latency chain took 31 - 28 of an Encoder2's 40 convs at 8 000 bins), the
  // longer ones on conv_bf16s.h.  65.6 against 66.4 ms per bench step (same box, alternating); until round 4 the exact fp32 kernels
  // served every precision below 32 000 positions per launch - which also made a row's bits depend on the batch it was computed in.
  if (net->precision != ORCA_PRECISION_F32) return unet_forward_nlc(ctx, net, x, sx_b, sx_c, sx_l, B, n, outs, nlev, up_only);
  const size_t full = (size_t)B * 128 * n;
  size_t need = 0;
  for (int i = 0; i < nlev; ++i) need += ru256((full >> i) * sizeof(float));  // encs[0..nlev-1]
  need += 3 * ru256(full * sizeof(float));
  ORCA_TRY(ws_ensure(ctx, need));
  std::vector<float*> encs(nlev + 1);
  for (int i = 0; i < nlev; ++i) encs[i] = up_only ? outs[i] : ws_take(ctx, full >> i);
  encs[nlev] = outs[nlev];
  float* t0 = ws_take(ctx, full);
  float* t1 = ws_take(ctx, full);
  float* t2 = ws_take(ctx, full);
  hipStream_t s = ctx->stream;
  // stage the (possibly strided) input as contiguous [B][128][n]
  for (int b = 0; b < B; ++b)
    ORCA_TRY(launch_copy2d(ctx, x + (long)b * sx_b, sx_c, sx_l, encs[0] + (size_t)b * 128 * n, n, 128, n));
  const ConvLayer* L = net->convs.data();
  // contracting path
  for (int i = 0; i < nlev; ++i) {
    const long ni = n >> i, no = n >> (i + 1);
    ORCA_TRY(launch_pool(ctx, encs[i], ni, t0, no, (long)B * 128, no, 2));
    const long bs = 128 * no;
    ORCA_TRY(launch_conv1d(ctx, L[4 * i + 0], t0, bs, no, t1, bs, no, nullptr, nullptr, B, no, 0, 0));
    ORCA_TRY(launch_conv1d(ctx, L[4 * i + 1], t1, bs, no, t2, bs, no, nullptr, nullptr, B, no, 0, 0));  // lout
    ORCA_TRY(launch_conv1d(ctx, L[4 * i + 2], t2, bs, no, t1, bs, no, nullptr, nullptr, B, no, 1, 0));
    ORCA_TRY(launch_conv1d(ctx, L[4 * i + 3], t1, bs, no, encs[i + 1], bs, no, t2, nullptr, B, no, 1, 0));
  }
  if (up_only) return ORCA_OK;
  // expanding path
  const float* cur = encs[nlev];
  for (int i = 0; i < nlev; ++i) {
    const int lev = nlev - 1 - i;
    const long ni = n >> (lev + 1), no = n >> lev;
    dim3 grid((unsigned)((no + 255) / 256), (unsigned)(B * 128));
    hipLaunchKernelGGL(upsample1d_x2_kernel, grid, dim3(256), 0, s, cur, ni, t0, no, no);
    LAUNCHCHECK("upsample1d_x2_kernel");
    const long bs = 128 * no;
    const ConvLayer* D = L + 4 * nlev + 4 * i;
    ORCA_TRY(launch_conv1d(ctx, D[0], t0, bs, no, t1, bs, no, nullptr, nullptr, B, no, 0, 0));
    ORCA_TRY(launch_conv1d(ctx, D[1], t1, bs, no, t2, bs, no, nullptr, nullptr, B, no, 0, 0));  // lout
    ORCA_TRY(launch_conv1d(ctx, D[2], t2, bs, no, t1, bs, no, nullptr, nullptr, B, no, 1, 0));
    ORCA_TRY(launch_conv1d(ctx, D[3], t1, bs, no, outs[lev], bs, no, t2, encs[lev], B, no, 1, 0));
    cur = outs[lev];
  }
  return ORCA_OK;
}

