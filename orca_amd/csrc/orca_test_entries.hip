// orca_test_entries.hip - single-layer entry points (one conv / pool of the library on caller-owned tensors): what tests/test_gpu_kernels.py and the tools drive
// Part of liborca_hip.so (include/orca_hip.h is the ABI; orca_internal.h what the units share).
#include "orca_internal.h"

#include "conv_p16.h"        // nlc <-> P16 / B16 conversions
#include "conv2d_m16.h"      // nchw <-> M16 conversions
#include "misc_kernels.h"

// ---------------------------------------------------------------------------
// single-layer entry points
// ---------------------------------------------------------------------------
extern "C" int orca_conv1d_forward(orca_ctx* ctx, const orca_conv_desc* conv, const float* x, int64_t x_bs, int64_t ldx, float* y,
                                   int64_t y_bs, int64_t ldy, const float* r1, const float* r2, int B, int64_t n, int relu,
                                   int tile) {
  if (!ctx || !conv || !x || !y) return fail(ORCA_EINVAL, "orca_conv1d_forward: NULL argument");
  HIPCHECK(hipSetDevice(ctx->device));
  ConvLayer L;
  ORCA_TRY(make_layer(*conv, &L));
  int rc = launch_conv1d(ctx, L, x, x_bs, ldx, y, y_bs, ldy, r1, r2, B, n, relu, tile);
  (void)hipStreamSynchronize(ctx->stream);
  free_layer(L);
  return rc;
}

extern "C" int orca_conv1d_nlc_forward(orca_ctx* ctx, const orca_conv_desc* conv, int precision, const float* x, float* y,
                                       const float* r1, int B, int64_t n, int relu) {
  if (!ctx || !conv || !x || !y) return fail(ORCA_EINVAL, "orca_conv1d_nlc_forward: NULL argument");
  if (precision < ORCA_PRECISION_BF16 || precision > ORCA_PRECISION_F16X2) return fail(ORCA_EINVAL, "precision %d unsupported here", precision);
  HIPCHECK(hipSetDevice(ctx->device));
  ConvLayer L;
  ORCA_TRY(make_layer(*conv, &L));
  int rc = launch_conv1d_b16(ctx, L, precision, x, (long)n * conv->cin, y, (long)n * conv->cout, r1, B, n, relu);
  (void)hipStreamSynchronize(ctx->stream);
  free_layer(L);
  return rc;
}

static int conv1d_planar_test(orca_ctx* ctx, const orca_conv_desc* conv, const float* x, float* y, const float* r1, int64_t n, int relu,
                              int out_mode, int fmt) {
  if (!ctx || !conv || !x || !y) return fail(ORCA_EINVAL, "orca_conv1d_p16/b16_forward: NULL argument");
  HIPCHECK(hipSetDevice(ctx->device));
  ConvLayer L;
  if (conv->ksize == 17) {     // a 17-tap layer as the Encoder's composed pairs run it: weight_host [cout][cin][17]
    std::vector<double> w17((size_t)conv->cout * conv->cin * 17), b17(conv->cout);
    for (size_t i = 0; i < w17.size(); ++i) w17[i] = conv->weight_host[i];
    for (int i = 0; i < conv->cout; ++i) b17[i] = conv->bias_host[i];
    ORCA_TRY(make_layer17(conv->cin, conv->cout, w17, b17, &L));
  } else
  ORCA_TRY(make_layer(*conv, &L));
  const long nout = out_mode == 1 ? n / 4 : out_mode == 3 ? n / 5 : n;
  const size_t sx = (size_t)conv->cin * p16_plen(n), sy = (size_t)conv->cout * p16_plen(nout), sr = (size_t)conv->cout * p16_plen(n);
  int rc = ws_ensure(ctx, ru256(sx * 4) + ru256(sy * 4) + ru256(sr * 4));
  if (rc == ORCA_OK) {
    float* xp = ws_take(ctx, sx);
    float* yp = ws_take(ctx, sy);
    float* rp = ws_take(ctx, sr);
    hipStream_t s = ctx->stream;
    auto blocks = [](long n_, int C) { return dim3((unsigned)((n_ * (C / 4) + 255) / 256)); };
    auto to_planar = [&](const float* src, float* dst, int C) {
      if (fmt == 1) hipLaunchKernelGGL(nlc_to_b16_kernel, blocks(n, C), dim3(256), 0, s, src, reinterpret_cast<f32x4*>(dst), (long)n, C, p16_plen(n));
      else hipLaunchKernelGGL(nlc_to_p16_kernel, blocks(n, C), dim3(256), 0, s, src, reinterpret_cast<f32x4*>(dst), (long)n, C, p16_plen(n));
    };
    (void)launch_p16_zero_pads(ctx, xp, conv->cin, n, fmt);
    to_planar(x, xp, conv->cin);
    if (r1) {
      (void)launch_p16_zero_pads(ctx, rp, conv->cout, n, fmt);
      to_planar(r1, rp, conv->cout);
    }
    rc = launch_conv1d_p16(ctx, L, xp, out_mode == 2 ? (void*)y : (void*)yp, r1 ? rp : nullptr, n, relu, out_mode, nullptr, fmt);
    if (rc == ORCA_OK && out_mode != 2 && nout > 0) (void)launch_p16_zero_pads(ctx, yp, conv->cout, nout, fmt);   // as in the Encoder: pads after the producer
    if (rc == ORCA_OK && out_mode != 2 && nout > 0) {
      if (fmt == 1) hipLaunchKernelGGL(b16_to_nlc_kernel, blocks(nout, conv->cout), dim3(256), 0, s, reinterpret_cast<const f32x4*>(yp), y, nout, conv->cout, p16_plen(nout));
      else hipLaunchKernelGGL(p16_to_nlc_kernel, blocks(nout, conv->cout), dim3(256), 0, s, reinterpret_cast<const f32x4*>(yp), y, nout, conv->cout, p16_plen(nout));
    }
    hipError_t e = hipGetLastError();
    if (rc == ORCA_OK && e != hipSuccess) rc = fail(ORCA_EHIP, "planar conv test path: %s", hipGetErrorString(e));
  }
  (void)hipStreamSynchronize(ctx->stream);
  free_layer(L);
  return rc;
}

extern "C" int orca_conv1d_p16_forward(orca_ctx* ctx, const orca_conv_desc* conv, const float* x, float* y, const float* r1, int64_t n,
                                       int relu, int out_mode) {
  return conv1d_planar_test(ctx, conv, x, y, r1, n, relu, out_mode, 0);
}

extern "C" int orca_conv1d_b16_forward(orca_ctx* ctx, const orca_conv_desc* conv, const float* x, float* y, const float* r1, int64_t n,
                                       int relu, int out_mode) {
  if (conv && conv->cin % 32) return fail(ORCA_EINVAL, "orca_conv1d_b16_forward: cin %d is not a multiple of 32", conv->cin);
  return conv1d_planar_test(ctx, conv, x, y, r1, n, relu, out_mode, 1);
}

extern "C" int orca_conv2d_forward(orca_ctx* ctx, const orca_conv_desc* conv, const float* x, float* y, const float* r, int B,
                                   int n, int relu) {
  if (!ctx || !conv || !x || !y) return fail(ORCA_EINVAL, "orca_conv2d_forward: NULL argument");
  if (n <= 0 || n > ORCA_LDW) return fail(ORCA_EINVAL, "map size %d unsupported", n);
  HIPCHECK(hipSetDevice(ctx->device));
  ConvLayer L;
  ORCA_TRY(make_layer(*conv, &L));
  const int cpad = L.nchunks * 8;
  const size_t plane = (size_t)n * ORCA_LDW;
  int rc = ws_ensure(ctx, ru256(B * plane * cpad * 4) + 2 * ru256(B * plane * L.cout * 4));
  if (rc == ORCA_OK) {
    float* xp = ws_take(ctx, B * plane * cpad);
    float* yp = ws_take(ctx, B * plane * L.cout);
    float* rp = ws_take(ctx, B * plane * L.cout);
    hipStream_t s = ctx->stream;
    (void)hipMemsetAsync(xp, 0, B * plane * cpad * 4, s);
    for (int b = 0; b < B; ++b)
      hipLaunchKernelGGL(pad_rows_kernel, dim3((unsigned)(conv->cin * n)), dim3(ORCA_LDW), 0, s, x + (size_t)b * conv->cin * n * n,
                         xp + b * plane * cpad, n, 1);
    if (r) hipLaunchKernelGGL(pad_rows_kernel, dim3((unsigned)(B * L.cout * n)), dim3(ORCA_LDW), 0, s, r, rp, n, 1);
    rc = launch_conv2d(ctx, L, xp, plane * cpad, yp, plane * L.cout, r ? rp : nullptr, plane * L.cout, B, n, relu);
    if (rc == ORCA_OK) {
      hipLaunchKernelGGL(pad_rows_kernel, dim3((unsigned)(B * L.cout * n)), dim3(ORCA_LDW), 0, s, yp, y, n, 0);
      hipError_t e = hipGetLastError();
      if (e != hipSuccess) rc = fail(ORCA_EHIP, "pad_rows_kernel: %s", hipGetErrorString(e));
    }
  }
  (void)hipStreamSynchronize(ctx->stream);
  free_layer(L);
  return rc;
}

// single dilated 3x3 layer on M16 maps (conv2d_m16.h; dilations 1-8) - or, for dilation 16 / 32 / 64, a whole residual block
template <int NS, int DT>
static int conv2d_m16_test(orca_ctx* ctx, const ConvLayer& L, int mode, const float* x, float* y, const float* r, int B, int n, int relu) {
  const int xo = 2 * ((L.cin + 15) / 16), yo = L.cout / 8;
  const size_t upo = (size_t)NS * n * ORCA_LDW;
  ORCA_TRY(ws_ensure(ctx, ru256(B * upo * xo * 16) + 2 * ru256(B * upo * yo * 16)));
  f32x4* xp = reinterpret_cast<f32x4*>(ws_take(ctx, B * upo * xo * 4));
  f32x4* yp = reinterpret_cast<f32x4*>(ws_take(ctx, B * upo * yo * 4));
  f32x4* rp = reinterpret_cast<f32x4*>(ws_take(ctx, B * upo * yo * 4));
  hipStream_t s = ctx->stream;
  for (int b = 0; b < B; ++b) {
    hipLaunchKernelGGL((nchw_to_m16_kernel<NS, DT>), dim3((unsigned)n), dim3(ORCA_LDW), 0, s, x + (size_t)b * L.cin * n * n, L.cin, n, xp + b * upo * xo, xo);
    if (r) hipLaunchKernelGGL((nchw_to_m16_kernel<NS, DT>), dim3((unsigned)n), dim3(ORCA_LDW), 0, s, r + (size_t)b * L.cout * n * n, L.cout, n, rp + b * upo * yo, yo);
  }
  ORCA_TRY(launch_conv2d_m16(ctx, L, xp, upo * xo, xo, yp, upo * yo, yo, r ? rp : nullptr, upo * yo, B, n, relu, mode));
  for (int b = 0; b < B; ++b)
    hipLaunchKernelGGL((m16_to_nchw_kernel<NS, DT>), dim3((unsigned)n), dim3(ORCA_LDW), 0, s, yp + b * upo * yo, L.cout, n, y + (size_t)b * L.cout * n * n);
  LAUNCHCHECK("conv2d_m16 test path");
  return ORCA_OK;
}

extern "C" int orca_conv2d_m16_forward(orca_ctx* ctx, const orca_conv_desc* conv, int precision, const float* x, float* y, const float* r,
                                       int B, int n, int relu) {
  if (!ctx || !conv || !x || !y) return fail(ORCA_EINVAL, "orca_conv2d_m16_forward: NULL argument");
  if (n <= 0 || n > ORCA_LDW) return fail(ORCA_EINVAL, "map size %d unsupported", n);
  HIPCHECK(hipSetDevice(ctx->device));
  ConvLayer L;
  ORCA_TRY(make_layer(*conv, &L));
  int rc;
  if (precision == ORCA_PRECISION_F16X2) rc = conv2d_m16_test<2, 1>(ctx, L, 0, x, y, r, B, n, relu);
  else if (precision == ORCA_PRECISION_BF16) rc = conv2d_m16_test<1, 0>(ctx, L, 1, x, y, r, B, n, relu);
  else if (precision == ORCA_PRECISION_F16) rc = conv2d_m16_test<1, 1>(ctx, L, 2, x, y, r, B, n, relu);
  else rc = fail(ORCA_EINVAL, "orca_conv2d_m16_forward: precision %d has no M16 kernel", precision);
  (void)hipStreamSynchronize(ctx->stream);
  free_layer(L);
  return rc;
}

extern "C" int orca_pointwise1d_forward(orca_ctx* ctx, const float* w_dev, const float* bias_dev, int cout, int cin, const float* x,
                                        int64_t x_bs, int64_t ldx, float* y, int64_t y_bs, int64_t ldy, int B, int64_t n, int act) {
  if (!ctx || !w_dev || !bias_dev || !x || !y) return fail(ORCA_EINVAL, "orca_pointwise1d_forward: NULL argument");
  if (cout <= 0 || cin <= 0 || B <= 0 || n < 0 || act < 0 || act > 2) return fail(ORCA_EINVAL, "orca_pointwise1d_forward: bad shape / activation");
  if (n == 0) return ORCA_OK;
  HIPCHECK(hipSetDevice(ctx->device));
  dim3 grid((unsigned)((n + 255) / 256), (unsigned)cout, (unsigned)B);
  hipLaunchKernelGGL(pointwise1d_kernel, grid, dim3(256), 0, ctx->stream, w_dev, bias_dev, cin, x, (long)x_bs, (long)ldx, y, (long)y_bs,
                     (long)ldy, (long)n, act);
  LAUNCHCHECK("pointwise1d_kernel");
  return ORCA_OK;
}

extern "C" int orca_maxpool1d_forward(orca_ctx* ctx, const float* x, int64_t ldx, float* y, int64_t ldy, int64_t rows,
                                      int64_t n_out, int k) {
  if (!ctx || !x || !y) return fail(ORCA_EINVAL, "orca_maxpool1d_forward: NULL argument");
  HIPCHECK(hipSetDevice(ctx->device));
  return launch_pool(ctx, x, ldx, y, ldy, rows, n_out, k);
}

