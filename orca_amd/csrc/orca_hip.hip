// orca_hip.hip - C-ABI implementation (include/orca_hip.h) and network-level
// orchestration of the HIP kernels for gfx950.
//
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -I../../include orca_hip.hip -o liborca_hip.so
#include <hip/hip_runtime.h>

#include <dlfcn.h>
#include <link.h>

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "conv_kernels.h"
#include "conv_bf16s.h"
#include "conv2d_m16.h"
#include "conv2d_m16q.h"
#include "conv2d_dblock.h"
#include "conv_stage1.h"
#include "conv_p16.h"
#include "conv_ws.h"
#include "conv_p16w1.h"
#include "conv_p16p5.h"
#include "conv_p16x.h"
#include "conv_small.h"
#include "misc_kernels.h"
#include "coarsegrain.h"
#include "orca_hip.h"

// ---------------------------------------------------------------------------
// errors
// ---------------------------------------------------------------------------
static thread_local std::string g_err;

static int fail(int code, const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}

#define HIPCHECK(expr)                                                                          \
  do {                                                                                          \
    hipError_t _e = (expr);                                                                     \
    if (_e != hipSuccess) return fail(ORCA_EHIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
  } while (0)

#define LAUNCHCHECK(name)                                                                        \
  do {                                                                                          \
    hipError_t _e = hipGetLastError();                                                          \
    if (_e != hipSuccess) return fail(ORCA_EHIP, "launch of %s failed: %s", name, hipGetErrorString(_e)); \
  } while (0)

#define ORCA_TRY(expr)        \
  do {                        \
    int _rc = (expr);         \
    if (_rc != ORCA_OK) return _rc; \
  } while (0)

static inline long ru4(long v) { return (v + 3) & ~3L; }
static inline size_t ru256(size_t v) { return (v + 255) & ~size_t(255); }

// ---------------------------------------------------------------------------
// context + workspace arena
// ---------------------------------------------------------------------------
struct TimedLaunch {
  orca_kernel_time rec;
  hipEvent_t e0, e1;
};

struct orca_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  char* ws = nullptr;
  size_t ws_bytes = 0;
  size_t ws_off = 0;
  hipStream_t ws_stream = nullptr;   // stream of the last call that used the arena
  bool ws_used = false;
  bool timing = false;
  std::vector<TimedLaunch> timed;
  long long counts[4] = {0, 0, 0, 0};   // launches since creation: conv_small.h | conv_bf16s.h (channel-last convs) | planar P16 / B16 convs | fused Decoder pairs (orca_ctx_launch_counts)
  unsigned* d_flag = nullptr;   // fp16-range overflow flag written by the f16x2 kernels
  float* d_edge = nullptr;      // 4 x 40 x 128 floats: one scratch slab per layer of the edge-fix chain (lconv_edge_layer_kernel)
  float* d_zero = nullptr;      // 256 bytes of zeros (the source of out-of-map units in conv2d_3x3_m16q_kernel)
  // Decoders with an even batch run as two half-batches on two streams (see decoder_nhwc)
  hipStream_t aux = nullptr;
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
};

static int ws_ensure(orca_ctx* ctx, size_t bytes) {
  ctx->ws_off = 0;
  // the arena is reused by every call: kernels of an EARLIER call on a different stream may still be reading it
  if (ctx->ws_used && ctx->ws_stream != ctx->stream) HIPCHECK(hipStreamSynchronize(ctx->ws_stream));
  ctx->ws_stream = ctx->stream;
  ctx->ws_used = true;
  if (bytes <= ctx->ws_bytes) return ORCA_OK;
  if (ctx->ws) {
    HIPCHECK(hipStreamSynchronize(ctx->stream));
    HIPCHECK(hipFree(ctx->ws));
    ctx->ws = nullptr;
    ctx->ws_bytes = 0;
  }
  void* p = nullptr;
  hipError_t e = hipMalloc(&p, bytes);
  if (e != hipSuccess) return fail(ORCA_ENOMEM, "hipMalloc of %zu workspace bytes failed: %s", bytes, hipGetErrorString(e));
  ctx->ws = static_cast<char*>(p);
  ctx->ws_bytes = bytes;
  return ORCA_OK;
}

static float* ws_take(orca_ctx* ctx, size_t nfloats) {
  size_t bytes = ru256(nfloats * sizeof(float));
  if (ctx->ws_off + bytes > ctx->ws_bytes) return nullptr;
  float* p = reinterpret_cast<float*>(ctx->ws + ctx->ws_off);
  ctx->ws_off += bytes;
  return p;
}

// ---------------------------------------------------------------------------
// layers / nets
// ---------------------------------------------------------------------------
struct ConvLayer {
  int cin = 0, cout = 0, ksize = 0, dil = 1;
  int kc = 8, nchunks = 0;
  float* d_w = nullptr;     // packed (k=9 / k=3) or raw [cout][cin] (k=1)
  float* d_bias = nullptr;
  void* d_wb16 = nullptr;   // k=9, cin%16==0: bf16 3-way split pack [cin/16][3][9][2][cout][8]
  void* d_wf16 = nullptr;   // same, fp16 2-way split pack [cin/16][2][9][2][cout][8]
  void* d_wb16p = nullptr;  // k=9, cin%32==0: plain bf16 pack [cin/32][2 k-pairs][9][2][cout][8] (B16 format of conv_p16.h)
  bool f16_ok = true;       // all |w| < 65504
};

struct orca_net {
  orca_ctx* ctx = nullptr;
  int kind = 0;
  int precision = ORCA_PRECISION_F32;
  float* d_first_w = nullptr;   // Encoder: folded [64][4][9] weights of the first layer, unpacked (conv1d_first_p16_kernel)
  void* d_first_w16 = nullptr;  // same as a K=48 fp16 split pack [2][3][2][64][8] (conv1d_first_mfma_p16_kernel)
  float* d_first_tab = nullptr; // same as a per-base-code table [9 taps][6 codes][64] (fused first layer of conv1d_k9_p16_kernel)
  int upsample_mode = ORCA_UPSAMPLE_BILINEAR;
  int num_2d = 1;               // Decoder / Decoder_1m: target maps per prediction (orca_leukemia.py:512-990); 1 = the Orca models
  float* d_sep = nullptr;       // Decoder: tap-summed weights of lcombinerD.a for the separable part of the first conv (sep_tables_kernel)
  // Encoder: composed linear pairs (compose_pair).  lconv1 as ONE 17-tap first layer (K = 68 -> 80 fp16 split pack + bias),
  // lconv2 / lconv3 as 17-tap planar convs (comp[1], comp[2]; ksize 17)
  void* d_l1_w16 = nullptr;
  float* d_l1_bias = nullptr;
  void* d_c1a_w16 = nullptr;    // conv1.a o lconv1: 25 taps from the bases, K = 100 -> 112 fp16 split pack (ReLU follows)
  float* d_c1a_bias = nullptr;
  float* d_l1_f32 = nullptr;    // the same two groups as fp32 [ntap][4][64] tables + biases for the exact-fp32 mode (first_taps_f32_kernel)
  float* d_c1a_f32 = nullptr;
  float* d_l1_bias32 = nullptr;
  float* d_c1a_bias32 = nullptr;
  ConvLayer comp[3];
  std::vector<ConvLayer> convs;
};

static int upload(const std::vector<float>& h, float** d) {
  HIPCHECK(hipMalloc(reinterpret_cast<void**>(d), h.size() * sizeof(float)));
  HIPCHECK(hipMemcpy(*d, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice));
  return ORCA_OK;
}

static inline uint16_t bf16_rne(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  if ((u & 0x7f800000u) == 0x7f800000u) return (uint16_t)(u >> 16);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
static inline float bf16_f32(uint16_t h) {
  uint32_t u = (uint32_t)h << 16;
  float f;
  memcpy(&f, &u, 4);
  return f;
}

static void free_layer(ConvLayer& L) {
  if (L.d_wb16) (void)hipFree(L.d_wb16);
  if (L.d_wf16) (void)hipFree(L.d_wf16);
  if (L.d_wb16p) (void)hipFree(L.d_wb16p);
  L.d_wb16 = L.d_wf16 = L.d_wb16p = nullptr;
  if (L.d_w) (void)hipFree(L.d_w);
  if (L.d_bias) (void)hipFree(L.d_bias);
  L.d_w = L.d_bias = nullptr;
}

// Re-layout of reference-format weights for the MFMA kernels:
//   conv1d  [cout][cin][9]     -> [cin/KC][9][KC][cout]
//   conv2d  [cout][cin][3][3]  -> [cin_pad/8][9][8][cout]  (pad channels = 0)
//   1x1     kept as [cout][cin]
static int make_layer(const orca_conv_desc& d, ConvLayer* out) {
  ConvLayer L;
  L.cin = d.cin; L.cout = d.cout; L.ksize = d.ksize; L.dil = d.dilation > 0 ? d.dilation : 1;
  if (!d.weight_host || !d.bias_host) return fail(ORCA_EINVAL, "conv desc with NULL weight/bias");
  std::vector<float> w;
  if (d.ksize == 9) {
    if (!(d.cout == 64 || d.cout == 96 || d.cout == 128)) return fail(ORCA_EINVAL, "conv1d cout %d unsupported", d.cout);
    L.kc = (d.cin % 8 == 0) ? 8 : 4;
    if (d.cin % L.kc) return fail(ORCA_EINVAL, "conv1d cin %d not a multiple of %d", d.cin, L.kc);
    if (L.kc == 4 && d.cout != 64) return fail(ORCA_EINVAL, "conv1d cin %d only supported with cout 64", d.cin);
    L.nchunks = d.cin / L.kc;
    w.assign((size_t)L.nchunks * 9 * L.kc * d.cout, 0.f);
    for (int co = 0; co < d.cout; ++co)
      for (int ci = 0; ci < d.cin; ++ci)
        for (int t = 0; t < 9; ++t) {
          const int c = ci / L.kc, k = ci % L.kc;
          w[(((size_t)c * 9 + t) * L.kc + k) * d.cout + co] = d.weight_host[((size_t)co * d.cin + ci) * 9 + t];
        }
  } else if (d.ksize == 3) {
    if (!(d.cout == 32 || d.cout == 64)) return fail(ORCA_EINVAL, "conv2d cout %d unsupported", d.cout);
    L.kc = 8;
    const int cpad = (d.cin + 7) / 8 * 8;
    L.nchunks = cpad / 8;
    w.assign((size_t)L.nchunks * 9 * 8 * d.cout, 0.f);
    for (int co = 0; co < d.cout; ++co)
      for (int ci = 0; ci < d.cin; ++ci)
        for (int t = 0; t < 9; ++t) {
          const int c = ci / 8, k = ci % 8;
          w[(((size_t)c * 9 + t) * 8 + k) * d.cout + co] = d.weight_host[((size_t)co * d.cin + ci) * 9 + t];
        }
  } else if (d.ksize == 1) {
    w.assign(d.weight_host, d.weight_host + (size_t)d.cout * d.cin);
  } else {
    return fail(ORCA_EINVAL, "unsupported kernel size %d", d.ksize);
  }
  std::vector<float> bias(d.bias_host, d.bias_host + d.cout);
  ORCA_TRY(upload(w, &L.d_w));
  int rc = upload(bias, &L.d_bias);
  if (rc != ORCA_OK) { free_layer(L); return rc; }
  if (d.ksize == 9 && d.cin % 16 == 0) {
    // bf16 split pack for conv_bf16s.h: w = w1 + w2 + w3 (successive RNE residuals)
    const int nc = d.cin / 16;
    std::vector<uint16_t> pk((size_t)nc * 3 * 9 * 2 * d.cout * 8);
    for (int co = 0; co < d.cout; ++co)
      for (int ci = 0; ci < d.cin; ++ci)
        for (int t = 0; t < 9; ++t) {
          float v = d.weight_host[((size_t)co * d.cin + ci) * 9 + t];
          const int c = ci / 16, gg = (ci % 16) / 8, e = ci % 8;
          for (int sp = 0; sp < 3; ++sp) {
            const uint16_t h = bf16_rne(v);
            v -= bf16_f32(h);
            pk[(((((size_t)c * 3 + sp) * 9 + t) * 2 + gg) * d.cout + co) * 8 + e] = h;
          }
        }
    hipError_t e1 = hipMalloc(&L.d_wb16, pk.size() * 2);
    if (e1 == hipSuccess) e1 = hipMemcpy(L.d_wb16, pk.data(), pk.size() * 2, hipMemcpyHostToDevice);
    if (e1 != hipSuccess) { free_layer(L); return fail(ORCA_EHIP, "bf16 weight upload failed: %s", hipGetErrorString(e1)); }
    // fp16 2-way split pack: w = h1 + h2 (RNE residuals)
    std::vector<uint16_t> pf((size_t)nc * 2 * 9 * 2 * d.cout * 8);
    for (int co = 0; co < d.cout; ++co)
      for (int ci = 0; ci < d.cin; ++ci)
        for (int t = 0; t < 9; ++t) {
          float v = d.weight_host[((size_t)co * d.cin + ci) * 9 + t];
          if (!(v > -65504.f && v < 65504.f)) L.f16_ok = false;
          const int c = ci / 16, gg = (ci % 16) / 8, e = ci % 8;
          for (int sp = 0; sp < 2; ++sp) {
            const _Float16 h = (_Float16)v;
            v -= (float)h;
            uint16_t bits;
            memcpy(&bits, &h, 2);
            pf[(((((size_t)c * 2 + sp) * 9 + t) * 2 + gg) * d.cout + co) * 8 + e] = bits;
          }
        }
    e1 = hipMalloc(&L.d_wf16, pf.size() * 2);
    if (e1 == hipSuccess) e1 = hipMemcpy(L.d_wf16, pf.data(), pf.size() * 2, hipMemcpyHostToDevice);
    if (e1 != hipSuccess) { free_layer(L); return fail(ORCA_EHIP, "fp16 weight upload failed: %s", hipGetErrorString(e1)); }
  }
  if (d.ksize == 9 && d.cin % 32 == 0) {
    // plain bf16 pack for the B16 format: channel ci = 32 c + 16 kp + 8 g + e
    const int nc = d.cin / 32;
    std::vector<uint16_t> pk((size_t)nc * 2 * 9 * 2 * d.cout * 8);
    for (int co = 0; co < d.cout; ++co)
      for (int ci = 0; ci < d.cin; ++ci)
        for (int t = 0; t < 9; ++t) {
          const float v = d.weight_host[((size_t)co * d.cin + ci) * 9 + t];
          const int c = ci / 32, kp = (ci % 32) / 16, gg = (ci % 16) / 8, e = ci % 8;
          pk[(((((size_t)c * 2 + kp) * 9 + t) * 2 + gg) * d.cout + co) * 8 + e] = bf16_rne(v);
        }
    hipError_t e1 = hipMalloc(&L.d_wb16p, pk.size() * 2);
    if (e1 == hipSuccess) e1 = hipMemcpy(L.d_wb16p, pk.data(), pk.size() * 2, hipMemcpyHostToDevice);
    if (e1 != hipSuccess) { free_layer(L); return fail(ORCA_EHIP, "bf16 plain weight upload failed: %s", hipGetErrorString(e1)); }
  }
  if (d.ksize == 3) {
    // fp16 2-way split pack for conv2d_m16.h / conv2d_dblock.h: [cin_pad16/16][2][9][2][cout][8], pad channels = 0
    const int nc = (d.cin + 15) / 16;
    std::vector<uint16_t> pf((size_t)nc * 2 * 9 * 2 * d.cout * 8, 0);
    for (int co = 0; co < d.cout; ++co)
      for (int ci = 0; ci < d.cin; ++ci)
        for (int t = 0; t < 9; ++t) {
          float v = d.weight_host[((size_t)co * d.cin + ci) * 9 + t];
          if (!(v > -65504.f && v < 65504.f)) L.f16_ok = false;
          const int c = ci / 16, gg = (ci % 16) / 8, e = ci % 8;
          for (int sp = 0; sp < 2; ++sp) {
            const _Float16 h = (_Float16)v;
            v -= (float)h;
            uint16_t bits;
            memcpy(&bits, &h, 2);
            pf[(((((size_t)c * 2 + sp) * 9 + t) * 2 + gg) * d.cout + co) * 8 + e] = bits;
          }
        }
    hipError_t e1 = hipMalloc(&L.d_wf16, pf.size() * 2);
    if (e1 == hipSuccess) e1 = hipMemcpy(L.d_wf16, pf.data(), pf.size() * 2, hipMemcpyHostToDevice);
    if (e1 != hipSuccess) { free_layer(L); return fail(ORCA_EHIP, "fp16 conv2d weight upload failed: %s", hipGetErrorString(e1)); }
    // plain bf16 pack (one product): [cin_pad16/16][9][2][cout][8]
    std::vector<uint16_t> pb((size_t)nc * 9 * 2 * d.cout * 8, 0);
    for (int co = 0; co < d.cout; ++co)
      for (int ci = 0; ci < d.cin; ++ci)
        for (int t = 0; t < 9; ++t) {
          const int c = ci / 16, gg = (ci % 16) / 8, e = ci % 8;
          pb[((((size_t)c * 9 + t) * 2 + gg) * d.cout + co) * 8 + e] = bf16_rne(d.weight_host[((size_t)co * d.cin + ci) * 9 + t]);
        }
    e1 = hipMalloc(&L.d_wb16p, pb.size() * 2);
    if (e1 == hipSuccess) e1 = hipMemcpy(L.d_wb16p, pb.data(), pb.size() * 2, hipMemcpyHostToDevice);
    if (e1 != hipSuccess) { free_layer(L); return fail(ORCA_EHIP, "bf16 conv2d weight upload failed: %s", hipGetErrorString(e1)); }
  }
  *out = L;
  return ORCA_OK;
}

// A linear pair Conv(k9, W1, b1) -> Conv(k9, W2, b2) (BatchNorms folded; no nonlinearity in between: the Encoder's lconv_i,
// orca_modules.py:811-816, 829-835, 846-852) is ONE affine 17-tap conv:
//   out[p] = b2 + sum_t2 W2[t2] (b1 + sum_t1 W1[t1] x[p + t1 + t2 - 8])  =>  W17[co][ci][t] = sum_m sum_{t1+t2=t} W2[co][m][t2] W1[m][ci][t1],
//   b17[co] = b2[co] + sum_m sum_t2 W2[co][m][t2] b1[m]        (composed in fp64)
// exact wherever the intermediate is not zero-padded, i.e. everywhere but the 4 outputs next to each end (lconv_edge_fix_kernel).
static void compose_taps(const std::vector<double>& w1, const std::vector<double>& b1, int cin, int cm, int k1, const orca_conv_desc& c2,
                         std::vector<double>* wo_, std::vector<double>* bo_) {
  // w1 [cm][cin][k1], b1 [cm] (already composed or a plain conv) followed by c2 = Conv(cm -> cout, k9): [cout][cin][k1 + 8]
  const int cout = c2.cout, ko = k1 + 8;
  std::vector<double> wout((size_t)cout * cin * ko, 0.0), bout(cout, 0.0);
  for (int co = 0; co < cout; ++co) {
    double bb = c2.bias_host[co];
    double* wo = wout.data() + (size_t)co * cin * ko;
    for (int m = 0; m < cm; ++m) {
      const float* w2 = c2.weight_host + ((size_t)co * cm + m) * 9;
      double s2 = 0.0;
      for (int t2 = 0; t2 < 9; ++t2) s2 += w2[t2];
      bb += s2 * b1[m];
      const double* wm = w1.data() + (size_t)m * cin * k1;
      for (int ci = 0; ci < cin; ++ci)
        for (int t2 = 0; t2 < 9; ++t2) {
          const double v2 = w2[t2];
          for (int t1 = 0; t1 < k1; ++t1) wo[ci * ko + t1 + t2] += v2 * wm[ci * k1 + t1];
        }
    }
    bout[co] = bb;
  }
  wo_->swap(wout);
  bo_->swap(bout);
}
static void compose_pair(const orca_conv_desc& c1, const orca_conv_desc& c2, std::vector<double>* w17, std::vector<double>* b17) {
  std::vector<double> w1((size_t)c1.cout * c1.cin * 9), b1(c1.cout);
  for (size_t i = 0; i < w1.size(); ++i) w1[i] = c1.weight_host[i];
  for (int i = 0; i < c1.cout; ++i) b1[i] = c1.bias_host[i];
  compose_taps(w1, b1, c1.cin, c1.cout, 9, c2, w17, b17);
}

// 17-tap planar conv layer: packs in the layout of the k9 kernels with TWICE the K-chunks - chunk 2c + h holds taps 9h .. 9h+8
// of input channels 16c .. 16c+15 (32c .. in the bf16 pack); tap 17 does not exist: zero weights
static int make_layer17(int cin, int cout, const std::vector<double>& w17, const std::vector<double>& b17, ConvLayer* out) {
  ConvLayer L;
  L.cin = cin; L.cout = cout; L.ksize = 17; L.kc = 16; L.nchunks = 2 * (cin / 16);
  if (cin % 32 || !(cout == 64 || cout == 96 || cout == 128)) return fail(ORCA_EINVAL, "composed conv %d -> %d unsupported", cin, cout);
  std::vector<float> bias(cout);
  for (int i = 0; i < cout; ++i) bias[i] = (float)b17[i];
  ORCA_TRY(upload(bias, &L.d_bias));
  std::vector<uint16_t> pf((size_t)L.nchunks * 2 * 9 * 2 * cout * 8, 0), pb((size_t)2 * (cin / 32) * 2 * 9 * 2 * cout * 8, 0);
  for (int co = 0; co < cout; ++co)
    for (int ci = 0; ci < cin; ++ci)
      for (int t = 0; t < 17; ++t) {
        const double wd = w17[((size_t)co * cin + ci) * 17 + t];
        float v = (float)wd;
        if (!(v > -65504.f && v < 65504.f)) L.f16_ok = false;
        const int h = t / 9, tt = t % 9;
        {
          const int c = ci / 16, gg = (ci % 16) / 8, e = ci % 8;
          for (int sp = 0; sp < 2; ++sp) {
            const _Float16 hh = (_Float16)v;
            v -= (float)hh;
            uint16_t bits;
            memcpy(&bits, &hh, 2);
            pf[((((((size_t)c * 2 + h) * 2 + sp) * 9 + tt) * 2 + gg) * cout + co) * 8 + e] = bits;
          }
        }
        {
          const int c = ci / 32, kp = (ci % 32) / 16, gg = (ci % 16) / 8, e = ci % 8;
          pb[((((((size_t)c * 2 + h) * 2 + kp) * 9 + tt) * 2 + gg) * cout + co) * 8 + e] = bf16_rne((float)wd);
        }
      }
  hipError_t e1 = hipMalloc(&L.d_wf16, pf.size() * 2);
  if (e1 == hipSuccess) e1 = hipMemcpy(L.d_wf16, pf.data(), pf.size() * 2, hipMemcpyHostToDevice);
  if (e1 == hipSuccess) e1 = hipMalloc(&L.d_wb16p, pb.size() * 2);
  if (e1 == hipSuccess) e1 = hipMemcpy(L.d_wb16p, pb.data(), pb.size() * 2, hipMemcpyHostToDevice);
  if (e1 != hipSuccess) { free_layer(L); return fail(ORCA_EHIP, "composed weight upload failed: %s", hipGetErrorString(e1)); }
  *out = L;
  return ORCA_OK;
}

// ---------------------------------------------------------------------------
// kernel launch helpers
// ---------------------------------------------------------------------------
static bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// position tile used for long 128-channel convs: 128 (2 waves/SIMD, 4 accumulators/wave)
// or 256 (1 wave/SIMD, 8 accumulators/wave); ORCA_CONV1D_TILE128 overrides for A/B runs.
static int g_big_tile128 = [] { const char* e = getenv("ORCA_CONV1D_TILE128"); int v = e ? atoi(e) : 128; return (v == 256) ? 256 : 128; }();

template <int COUT, int MW, int NW, int WM, int WN, int KC>
static void launch_conv1d_t(hipStream_t s, const Conv1dArgs& a, int B) {
  constexpr int MT = WM * MW * 32;
  dim3 grid((unsigned)((a.n + MT - 1) / MT), (unsigned)B);
  hipLaunchKernelGGL((conv1d_k9_kernel<COUT, MW, NW, WM, WN, KC>), grid, dim3(WM * WN * 64), 0, s, a);
}

static int launch_conv1d(orca_ctx* ctx, const ConvLayer& L, const float* x, long x_bs, long ldx, float* y, long y_bs,
                         long ldy, const float* r1, const float* r2, int B, long n, int relu, int tile, int y_nlc = 0) {
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

static int launch_conv2d(orca_ctx* ctx, const ConvLayer& L, const float* x, long x_bs, float* y, long y_bs,
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
static int launch_conv2d_m16(orca_ctx* ctx, const ConvLayer& L, const f32x4* x, long x_bs, int x_oct, f32x4* y, long y_bs, int y_oct,
                             const f32x4* r, long r_bs, int B, int n, int relu, int mode, int chunk0 = 0, int nchunks_ = 0,
                             const float* tab = nullptr, long tab_bs = 0) {
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
  // half the chip: it stays on the one-row kernel (1.40 against 1.62 ms).  ORCA_NO_M16Q=1 (read per call) selects the one-row kernel
  // everywhere - the A/B and parity switch; ORCA_M16Q_ALWAYS=1 the four-row kernel for single maps too (tests).
  if ((B >= 2 || getenv("ORCA_M16Q_ALWAYS") != nullptr) && getenv("ORCA_NO_M16Q") == nullptr) {
    ConvM16QArgs aq;
    aq.c = a; aq.c.banded = 0; aq.zero = reinterpret_cast<const f32x4*>(ctx->d_zero);
    aq.ngroups = ((n + 4 * L.dil - 1) / (4 * L.dil)) * L.dil;
    aq.nb = B;
    // batches of more than one round (SV screen: 4 strands, config 3: 8): the grid is ONE round, a workgroup walks the maps b, b + grid.y, ...
    // of its tile and requests the next map's first piece under the last piece of the current one (needs an even chunk count: the heads'
    // 16- / 80- / 144-channel layers keep one workgroup per map and tile).  ORCA_NO_M16Q_WALK=1 (read per call): the A/B and parity switch
    static int ncu = [] { int dev = 0, v = 256; (void)hipGetDevice(&dev); (void)hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev); return v; }();
    const int gx = (aq.ngroups * 2 + 7) / 8 * 8;
    int gy = B;
    // (single-plane modes, 32 couts: 73.7 KB of LDS and 111 VGPRs - TWO workgroups fit a CU, one's transfers and epilogue under the other's
    // MFMAs: the resident round is twice as large.  ORCA_M16Q_ONE_PER_CU=1: the A/B switch)
    const int res = (mode != 0 && L.cout == 32 && getenv("ORCA_M16Q_ONE_PER_CU") == nullptr) ? 2 * ncu : ncu;
    if (a.nchunks % 2 == 0 && gx * B > res && getenv("ORCA_NO_M16Q_WALK") == nullptr) gy = res / gx > 1 ? res / gx : 1;
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
  static const bool no_banded = getenv("ORCA_NO_BANDED") != nullptr;   // A/B switch
  a.banded = (n >= 64 && !no_banded) ? 1 : 0;
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

static int launch_pool(orca_ctx* ctx, const float* x, long ldx, float* y, long ldy, long rows, long n_out, int k) {
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
    static const bool no_small = getenv("ORCA_NO_SMALL_TILES") != nullptr;   // A/B switch
    if constexpr (NS <= 2) {
      if (!no_small && ((a.n + 255) / 256) * B < 200) { launch_b16_t<128, 1, 2, 2, 2, NS, DT>(s, a, B); return ORCA_OK; }
    }
    launch_b16_t<128, 2, 2, 4, 2, NS, DT>(s, a, B);
  }
  else return fail(ORCA_EINVAL, "bf16s conv1d cout %d unsupported", L.cout);
  return ORCA_OK;
}

// x [B][n][cin], y/r1 [B][n][cout] channel-last.  precision: ORCA_PRECISION_BF16 / _BF16X2 / _BF16X3
static int launch_conv1d_b16(orca_ctx* ctx, const ConvLayer& L, int precision, const float* x, long x_bs, float* y, long y_bs,
                             const float* r1, int B, long n, int relu, int pool4 = 0, const float* r2 = nullptr) {
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
  // a row's result never depends on the batch it is computed in; ORCA_NO_SMALL_CONV=1 (read per call): the A/B and parity switch
  if (n <= 2048 && !pool4 && L.cout % 32 == 0 && L.cin % 16 == 0 && getenv("ORCA_NO_SMALL_CONV") == nullptr) {
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
static inline long p16_plen(long n) { return ((n + 512) / 512) * 512 + 32; }

static int launch_p16_zero_pads(orca_ctx* ctx, float* base, int C, long n_valid, int fmt = 0) {
  hipLaunchKernelGGL(p16_zero_pads_kernel, dim3((unsigned)(fmt == 1 ? C / 8 : C / 8 * 2)), dim3(256), 0, ctx->stream, reinterpret_cast<f32x4*>(base),
                     p16_plen(n_valid), n_valid);
  LAUNCHCHECK("p16_zero_pads_kernel");
  return ORCA_OK;
}

// MaxPool1d(5) between planar stages: x [C] planes of n_in positions -> y of n_in / 5 (conv2d_m16.h: p16_maxpool_kernel)
static int launch_p16_pool5(orca_ctx* ctx, const float* x, float* y, int C, long n_in, int fmt) {
  const long n_out = n_in / 5;
  if (n_out <= 0) return ORCA_OK;
  dim3 grid((unsigned)((n_out + 255) / 256), (unsigned)(C / 8));
  if (fmt == 1) hipLaunchKernelGGL((p16_maxpool_kernel<5, 1, 0>), grid, dim3(256), 0, ctx->stream, reinterpret_cast<const f32x4*>(x), p16_plen(n_in), reinterpret_cast<f32x4*>(y), p16_plen(n_out), n_out);
  else hipLaunchKernelGGL((p16_maxpool_kernel<5, 2, 1>), grid, dim3(256), 0, ctx->stream, reinterpret_cast<const f32x4*>(x), p16_plen(n_in), reinterpret_cast<f32x4*>(y), p16_plen(n_out), n_out);
  LAUNCHCHECK("p16_maxpool_kernel");
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

// stage 1's pooled conv with the residual computed from the bases in its epilogue (conv_p16.h, RL); FMT 0 = P16, 1 = B16
template <int FMT>
static void launch_p16_res_bases(hipStream_t s, ConvP16Args a) {
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

// the 96-cout layers on 512-position tiles with ONE half-by-half refilled weight buffer (conv_p16w1.h)
template <int OM, bool R1, int FMT>
static void launch_p16w1_k(hipStream_t s, ConvP16Args a) {
  static int ncu = [] { int dev = 0, v = 256; (void)hipGetDevice(&dev); (void)hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev); return v; }();
  a.tiles_per_row = (a.n + 511) / 512;
  dim3 grid((unsigned)(a.tiles_per_row < ncu ? a.tiles_per_row : ncu));
  hipLaunchKernelGGL((conv1d_k9_p16w1_kernel<OM, R1, FMT>), grid, dim3(512), 0, s, a);
}
// the same layers on the 16 x 16 x 32 matrix instruction (conv_p16x.h; P16 only)
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
template <int FMT>
static bool launch_p16w1(hipStream_t s, const ConvP16Args& a) {     // false: this (out_mode, residual) pair stays on the 256-position kernel
  const bool r1 = a.r1 != nullptr;
  if (a.out_mode == 0 && !r1) launch_p16w1_k<0, false, FMT>(s, a);
  else if (a.out_mode == 1 && r1) launch_p16w1_k<1, true, FMT>(s, a);
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
static int launch_conv1d_p16(orca_ctx* ctx, const ConvLayer& L, const float* x, void* y, const float* r1, long n, int relu,
                             int out_mode, const FusedFirst* f1 = nullptr, int fmt = 0) {
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
  static const bool no_ws = getenv("ORCA_NO_WS") != nullptr;   // A/B switch: W-stationary barrier-free kernel (conv_ws.h)
  const bool ws_ok = !no_ws && fmt == 1 && !k17;     // (P16: measured 3 % slower than the tiled kernel - profiles/HISTORY.md)
  int tile_tag = fmt == 1 ? -6 : -5;
  if (f1 && f1->residual) {
    if (L.cout != 64 || L.cin != 64 || out_mode != 1 || r1 || k17) return fail(ORCA_EINVAL, "residual from the bases: only stage 1's pooled 64 -> 64 planar conv");
    a.f1_codes = f1->codes; a.f1_nmask = f1->nmask; a.f1_origin = f1->origin; a.f1_codes_L = f1->codes_L; a.f1_codes_off = f1->codes_off; a.f1_reverse = f1->reverse;
    a.rl_w = reinterpret_cast<const f32x4*>(f1->table); a.f1_bias = f1->bias;
    if (fmt == 1) launch_p16_res_bases<1>(ctx->stream, a);
    else launch_p16_res_bases<0>(ctx->stream, a);
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
  } else if (L.cout == 96 && n >= 65536 && fmt == 0 && getenv("ORCA_NO_P16X") == nullptr &&
             launch_p16x(ctx->stream, a)) {
    tile_tag = -14;                      // 16 x 16 x 32 matrix instruction (conv_p16x.h)
  } else if (L.cout == 96 && n >= 65536 && getenv("ORCA_NO_P16W1") == nullptr && (fmt == 1 ? launch_p16w1<1>(ctx->stream, a) : launch_p16w1<0>(ctx->stream, a))) {
    tile_tag = fmt == 1 ? -10 : -9;      // stage 2 of the Encoder: 512-position tiles (conv_p16w1.h)
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
#define ORCA_EDGE_SLAB (40 * 128)
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
// C ABI: library / context
// ---------------------------------------------------------------------------
extern "C" int orca_abi_version(void) { return ORCA_ABI_VERSION; }
extern "C" const char* orca_last_error(void) { return g_err.c_str(); }

extern "C" int orca_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
  return n;
}

extern "C" int orca_ctx_create(int device, void* hip_stream, orca_ctx** out) {
  if (!out) return fail(ORCA_EINVAL, "orca_ctx_create: out is NULL");
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) { (void)hipGetLastError(); return fail(ORCA_ENODEV, "no HIP device visible"); }
  if (device < 0 || device >= n) return fail(ORCA_EINVAL, "device %d out of range (have %d)", device, n);
  HIPCHECK(hipSetDevice(device));
  hipDeviceProp_t prop;
  HIPCHECK(hipGetDeviceProperties(&prop, device));
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    return fail(ORCA_ENODEV, "device %d is %s; liborca_hip is built for gfx950 only", device, prop.gcnArchName);
  orca_ctx* c = new orca_ctx();
  c->device = device;
  c->stream = static_cast<hipStream_t>(hip_stream);
  if (hipMalloc(reinterpret_cast<void**>(&c->d_flag), 4 * sizeof(unsigned)) != hipSuccess || hipMemset(c->d_flag, 0, 4 * sizeof(unsigned)) != hipSuccess ||
      hipMalloc(reinterpret_cast<void**>(&c->d_edge), 4 * 40 * 128 * sizeof(float)) != hipSuccess ||
      hipMalloc(reinterpret_cast<void**>(&c->d_zero), 256) != hipSuccess || hipMemset(c->d_zero, 0, 256) != hipSuccess) {
    delete c;
    return fail(ORCA_ENOMEM, "could not allocate the context flag word");
  }
  *out = c;
  return ORCA_OK;
}

extern "C" int orca_ctx_destroy(orca_ctx* ctx) {
  if (!ctx) return ORCA_OK;
  (void)hipSetDevice(ctx->device);
  if (ctx->ws) {
    if (ctx->ws_used && ctx->ws_stream != ctx->stream) (void)hipStreamSynchronize(ctx->ws_stream);
    (void)hipStreamSynchronize(ctx->stream);
    (void)hipFree(ctx->ws);
  }
  if (ctx->d_flag) (void)hipFree(ctx->d_flag);
  if (ctx->d_edge) (void)hipFree(ctx->d_edge);
  if (ctx->d_zero) (void)hipFree(ctx->d_zero);
  if (ctx->aux) { (void)hipStreamSynchronize(ctx->aux); (void)hipStreamDestroy(ctx->aux); (void)hipEventDestroy(ctx->ev_fork); (void)hipEventDestroy(ctx->ev_join); }
  delete ctx;
  return ORCA_OK;
}

extern "C" int orca_ctx_set_stream(orca_ctx* ctx, void* hip_stream) {
  if (!ctx) return fail(ORCA_EINVAL, "ctx is NULL");
  ctx->stream = static_cast<hipStream_t>(hip_stream);
  return ORCA_OK;
}

extern "C" int orca_ctx_set_timing(orca_ctx* ctx, int enable) {
  if (!ctx) return fail(ORCA_EINVAL, "ctx is NULL");
  ctx->timing = enable != 0;
  return ORCA_OK;
}

extern "C" int orca_ctx_launch_counts(orca_ctx* ctx, int64_t* counts4) {
  if (!ctx || !counts4) return fail(ORCA_EINVAL, "NULL argument");
  for (int i = 0; i < 4; ++i) counts4[i] = ctx->counts[i];
  return ORCA_OK;
}

extern "C" int orca_ctx_get_timing(orca_ctx* ctx, orca_kernel_time* out, int max, int* n) {
  if (!ctx || !n) return fail(ORCA_EINVAL, "NULL argument");
  HIPCHECK(hipSetDevice(ctx->device));
  HIPCHECK(hipStreamSynchronize(ctx->stream));
  *n = (int)ctx->timed.size();
  for (size_t i = 0; i < ctx->timed.size(); ++i) {
    TimedLaunch& t = ctx->timed[i];
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, t.e0, t.e1);
    t.rec.ms = ms;
    if (out && (int)i < max) out[i] = t.rec;
    (void)hipEventDestroy(t.e0);
    (void)hipEventDestroy(t.e1);
  }
  ctx->timed.clear();
  return ORCA_OK;
}

extern "C" int orca_ctx_take_overflow(orca_ctx* ctx, int* flag) {
  if (!ctx || !flag) return fail(ORCA_EINVAL, "NULL argument");
  HIPCHECK(hipSetDevice(ctx->device));
  unsigned h = 0;
  HIPCHECK(hipMemcpyAsync(&h, ctx->d_flag, sizeof h, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHECK(hipMemsetAsync(ctx->d_flag, 0, sizeof h, ctx->stream));
  HIPCHECK(hipStreamSynchronize(ctx->stream));
  *flag = (int)h;
  return ORCA_OK;
}

extern "C" int orca_ctx_workspace_bytes(orca_ctx* ctx, size_t* out) {
  if (!ctx || !out) return fail(ORCA_EINVAL, "NULL argument");
  *out = ctx->ws_bytes;
  return ORCA_OK;
}

extern "C" int orca_ctx_release_workspace(orca_ctx* ctx) {
  if (!ctx) return fail(ORCA_EINVAL, "ctx is NULL");
  HIPCHECK(hipSetDevice(ctx->device));
  if (ctx->ws) {
    if (ctx->ws_used && ctx->ws_stream != ctx->stream) HIPCHECK(hipStreamSynchronize(ctx->ws_stream));
    HIPCHECK(hipStreamSynchronize(ctx->stream));
    HIPCHECK(hipFree(ctx->ws));
  }
  ctx->ws = nullptr; ctx->ws_bytes = 0; ctx->ws_off = 0; ctx->ws_used = false;
  return ORCA_OK;
}

// ---------------------------------------------------------------------------
// C ABI: weights
// ---------------------------------------------------------------------------
struct Shape { int cout, cin, k; };

static void expected_shapes(int kind, std::vector<Shape>* s, int T = 1) {
  const int F = T > 5 ? T : 5;   // hidden width of the `final` head
  s->clear();
  auto c1 = [&](int co, int ci) { s->push_back({co, ci, 9}); };
  auto c2 = [&](int co, int ci) { s->push_back({co, ci, 3}); };
  if (kind == ORCA_NET_ENCODER) {
    const int ch[7] = {64, 96, 128, 128, 128, 128, 128};
    int prev = 4;
    for (int i = 0; i < 7; ++i) { c1(ch[i], prev); c1(ch[i], ch[i]); c1(ch[i], ch[i]); c1(ch[i], ch[i]); prev = ch[i]; }
  } else if (kind == ORCA_NET_ENCODER2 || kind == ORCA_NET_ENCODER3) {
    const int nlev = kind == ORCA_NET_ENCODER2 ? 5 : 3;
    for (int i = 0; i < 8 * nlev; ++i) c1(128, 128);
  } else if (kind == ORCA_NET_ENCODER2B) {
    for (int i = 0; i < 4 * 5; ++i) c1(128, 128);
  } else if (kind == ORCA_NET_DECODER) {
    c2(64, 128 + T); c2(64, 64); c2(64, 64); c2(64, 64);  // lcombinerD, combinerD
    c2(64, 64 + T); c2(64, 64); c2(64, 64); c2(64, 64);   // lcombiner, combiner
    for (int i = 0; i < 28; ++i) { c2(32, 64); c2(64, 32); c2(32, 64); c2(64, 32); }
    s->push_back({F, 64, 1}); s->push_back({T, F, 1});
  } else if (kind == ORCA_NET_DECODER_1M) {
    for (int i = 0; i < 19; ++i) { c2(32, i == 0 ? 128 : 64); c2(64, 32); c2(32, 64); c2(64, 32); }
    s->push_back({F, 64, 1}); s->push_back({T, F, 1});
  }
}

extern "C" int orca_net_create(orca_ctx* ctx, int kind, const orca_conv_desc* convs, int n_convs, int upsample_mode, orca_net** out) {
  if (!ctx || !convs || !out) return fail(ORCA_EINVAL, "orca_net_create: NULL argument");
  HIPCHECK(hipSetDevice(ctx->device));
  std::vector<Shape> exp;
  int T = 1;   // multi-target decoders: the number of maps is the width of the last layer
  if ((kind == ORCA_NET_DECODER || kind == ORCA_NET_DECODER_1M) && n_convs > 0) {
    T = convs[n_convs - 1].cout;
    if (T < 1 || T > ORCA_MAX_TARGETS) return fail(ORCA_EINVAL, "decoder with %d target maps (supported: 1..%d)", T, ORCA_MAX_TARGETS);
  }
  expected_shapes(kind, &exp, T);
  if (exp.empty()) return fail(ORCA_EINVAL, "unknown net kind %d", kind);
  if ((int)exp.size() != n_convs) return fail(ORCA_EINVAL, "net kind %d expects %zu convs, got %d", kind, exp.size(), n_convs);
  for (int i = 0; i < n_convs; ++i)
    if (convs[i].cout != exp[i].cout || convs[i].cin != exp[i].cin || convs[i].ksize != exp[i].k)
      return fail(ORCA_EINVAL, "net kind %d conv %d: expected cout=%d cin=%d k=%d, got cout=%d cin=%d k=%d", kind, i,
                  exp[i].cout, exp[i].cin, exp[i].k, convs[i].cout, convs[i].cin, convs[i].ksize);
  orca_net* net = new orca_net();
  net->ctx = ctx; net->kind = kind; net->upsample_mode = upsample_mode; net->num_2d = T;
  net->convs.resize(n_convs);
  for (int i = 0; i < n_convs; ++i) {
    int rc = make_layer(convs[i], &net->convs[i]);
    if (rc != ORCA_OK) { orca_net_free(net); return rc; }
  }
  if (kind == ORCA_NET_DECODER) {
    // lcombinerD.a on mat[c][i][j] = x[c][i] + x[c][j] (c < 128) is separable:
    //   sum_{ky,kx valid} w[ky][kx] (x[i+ky-1] + x[j+kx-1]) = sum_ky (sum_{kx valid at j} w[ky][kx]) x[i+ky-1] + sum_kx (sum_{ky valid at i} w[ky][kx]) x[j+kx-1]
    // - two 3-tap 1-D convs of the encoding per border class (first / interior / last column resp. row).  wsep[which][class][k][c][co]:
    // which 0 = row term (k = ky, summed over the kx valid in column class), 1 = column term (k = kx, summed over the ky valid in row class).
    const orca_conv_desc& d0 = convs[0];
    std::vector<float> ws((size_t)2 * 3 * 3 * 128 * 64);
    for (int which = 0; which < 2; ++which)
      for (int cls = 0; cls < 3; ++cls)
        for (int k = 0; k < 3; ++k)
          for (int c = 0; c < 128; ++c)
            for (int co = 0; co < 64; ++co) {
              double acc = 0.0;
              for (int o = (cls == 0 ? 1 : 0); o <= (cls == 2 ? 1 : 2); ++o) {   // the other axis' taps that stay inside the map
                const int ky = which == 0 ? k : o, kx = which == 0 ? o : k;
                acc += d0.weight_host[(((size_t)co * d0.cin + c) * 3 + ky) * 3 + kx];
              }
              ws[((((size_t)which * 3 + cls) * 3 + k) * 128 + c) * 64 + co] = (float)acc;
            }
    int rc = upload(ws, &net->d_sep);
    if (rc != ORCA_OK) { orca_net_free(net); return rc; }
  }
  if (kind == ORCA_NET_ENCODER) {
    std::vector<float> w0(convs[0].weight_host, convs[0].weight_host + 64 * 4 * 9);
    int rc = upload(w0, &net->d_first_w);
    if (rc != ORCA_OK) { orca_net_free(net); return rc; }
    std::vector<uint16_t> pk((size_t)2 * 3 * 2 * 64 * 8, 0);   // k = tap*4 + ci; k16 step kk, half g, element e
    for (int co = 0; co < 64; ++co)
      for (int k = 0; k < 36; ++k) {
        float v = w0[((size_t)co * 4 + (k & 3)) * 9 + (k >> 2)];
        const int kk = k / 16, gg = (k % 16) / 8, e = k % 8;
        for (int sp = 0; sp < 2; ++sp) {
          const _Float16 hh = (_Float16)v;
          v -= (float)hh;
          uint16_t bits;
          memcpy(&bits, &hh, 2);
          pk[((((size_t)sp * 3 + kk) * 2 + gg) * 64 + co) * 8 + e] = bits;
        }
      }
    if (hipMalloc(&net->d_first_w16, pk.size() * 2) != hipSuccess ||
        hipMemcpy(net->d_first_w16, pk.data(), pk.size() * 2, hipMemcpyHostToDevice) != hipSuccess) {
      orca_net_free(net);
      return fail(ORCA_EHIP, "first-layer fp16 pack upload failed");
    }
    // table form for one-hot input: T[tap][K-chunk][code][quad][4] (cout = 16*chunk + 4*quad + e); code 0..3 = the base's
    // weight column, 4 = 'N' (0.25 of each), 5 = zero
    std::vector<float> tab((size_t)9 * 6 * 64, 0.f);
    auto at = [&](int t, int code, int co) -> float& { return tab[((((size_t)t * 4 + co / 16) * 6 + code) * 4 + (co % 16) / 4) * 4 + co % 4]; };
    for (int co = 0; co < 64; ++co)
      for (int t = 0; t < 9; ++t) {
        float sum = 0.f;
        for (int bse = 0; bse < 4; ++bse) {
          const float v = w0[((size_t)co * 4 + bse) * 9 + t];
          at(t, bse, co) = v;
          sum += 0.25f * v;
        }
        at(t, 4, co) = sum;
      }
    rc = upload(tab, &net->d_first_tab);
    if (rc != ORCA_OK) { orca_net_free(net); return rc; }
    // composed linear pairs (compose_pair): lconv1 -> K = 68 (tap*4 + ci) fp16 split pack for conv1d_first_mfma_p16_kernel<.,.,17>
    {
      std::vector<double> w17, b17, w25, b25;
      compose_pair(convs[0], convs[1], &w17, &b17);
      compose_taps(w17, b17, 4, 64, 17, convs[2], &w25, &b25);       // conv1.a (BN folded; its ReLU stays in the kernel)
      // K = tap*4 + ci fp16 split pack [2 splits][KS][2 g][64][8] for conv1d_first_mfma_p16_kernel<., ., ntap>
      auto pack_first = [&](const std::vector<double>& w, const std::vector<double>& b, int ntap, void** d_w, float** d_b) -> int {
        const int KS = (4 * ntap + 15) / 16;
        std::vector<uint16_t> pk((size_t)2 * KS * 2 * 64 * 8, 0);
        for (int co = 0; co < 64; ++co)
          for (int k = 0; k < 4 * ntap; ++k) {
            float v = (float)w[((size_t)co * 4 + (k & 3)) * ntap + (k >> 2)];
            const int kk = k / 16, gg = (k % 16) / 8, e = k % 8;
            for (int sp = 0; sp < 2; ++sp) {
              const _Float16 hh = (_Float16)v;
              v -= (float)hh;
              uint16_t bits;
              memcpy(&bits, &hh, 2);
              pk[((((size_t)sp * KS + kk) * 2 + gg) * 64 + co) * 8 + e] = bits;
            }
          }
        std::vector<float> bf(64);
        for (int i = 0; i < 64; ++i) bf[i] = (float)b[i];
        if (hipMalloc(d_w, pk.size() * 2) != hipSuccess || hipMemcpy(*d_w, pk.data(), pk.size() * 2, hipMemcpyHostToDevice) != hipSuccess) return ORCA_EHIP;
        return upload(bf, d_b);
      };
      // A composed weight is a sum of products of folded weights: with extreme checkpoints it may leave the fp16 range even though
      // every single layer fits.  Such a group simply keeps the reference's two-conv form (its packs stay NULL).
      auto in_f16 = [](const std::vector<double>& w) {
        for (double v : w) if (!(v > -65504.0 && v < 65504.0)) return false;
        return true;
      };
      {   // fp32 tables [t][ci][co] for the exact-fp32 mode (no range limit there)
        auto tab32 = [&](const std::vector<double>& w, const std::vector<double>& b, int ntap, float** d_w, float** d_b) -> int {
          std::vector<float> t((size_t)ntap * 4 * 64), bf(64);
          for (int co = 0; co < 64; ++co) {
            bf[co] = (float)b[co];
            for (int ci = 0; ci < 4; ++ci)
              for (int tt = 0; tt < ntap; ++tt) t[((size_t)tt * 4 + ci) * 64 + co] = (float)w[((size_t)co * 4 + ci) * ntap + tt];
          }
          return upload(t, d_w) != ORCA_OK ? ORCA_EHIP : upload(bf, d_b);
        };
        if (tab32(w17, b17, 17, &net->d_l1_f32, &net->d_l1_bias32) != ORCA_OK || tab32(w25, b25, 25, &net->d_c1a_f32, &net->d_c1a_bias32) != ORCA_OK) {
          orca_net_free(net);
          return fail(ORCA_EHIP, "composed first-layer upload failed");
        }
      }
      if (in_f16(w17) && in_f16(b17)) {
        if (pack_first(w17, b17, 17, &net->d_l1_w16, &net->d_l1_bias) != ORCA_OK) { orca_net_free(net); return fail(ORCA_EHIP, "composed first-layer upload failed"); }
        if (in_f16(w25) && in_f16(b25) && pack_first(w25, b25, 25, &net->d_c1a_w16, &net->d_c1a_bias) != ORCA_OK) {
          orca_net_free(net);
          return fail(ORCA_EHIP, "composed first-layer upload failed");
        }
      }
      for (int st = 1; st <= 2; ++st) {
        compose_pair(convs[4 * st], convs[4 * st + 1], &w17, &b17);
        if (!in_f16(w17)) continue;
        rc = make_layer17(convs[4 * st].cin, convs[4 * st + 1].cout, w17, b17, &net->comp[st]);
        if (rc != ORCA_OK) { orca_net_free(net); return rc; }
      }
    }
  }
  *out = net;
  return ORCA_OK;
}

extern "C" int orca_net_set_precision(orca_net* net, int precision) {
  if (!net) return fail(ORCA_EINVAL, "net is NULL");
  if (precision < ORCA_PRECISION_F32 || precision > ORCA_PRECISION_F16) return fail(ORCA_EINVAL, "unknown precision %d", precision);
  const bool dec = net->kind == ORCA_NET_DECODER || net->kind == ORCA_NET_DECODER_1M;
  const bool unet = net->kind == ORCA_NET_ENCODER2 || net->kind == ORCA_NET_ENCODER3 || net->kind == ORCA_NET_ENCODER2B;
  if (precision != ORCA_PRECISION_F32 && !(((net->kind == ORCA_NET_ENCODER || unet) && precision != ORCA_PRECISION_F16) ||
                                           (dec && (precision == ORCA_PRECISION_F16X2 || precision == ORCA_PRECISION_BF16 || precision == ORCA_PRECISION_F16))))
    return fail(ORCA_EINVAL, "precision %d is not implemented for net kind %d", precision, net->kind);
  net->precision = precision;
  return ORCA_OK;
}

extern "C" int orca_net_free(orca_net* net) {
  if (!net) return ORCA_OK;
  if (net->ctx) (void)hipSetDevice(net->ctx->device);
  for (auto& L : net->convs) free_layer(L);
  if (net->d_first_w) (void)hipFree(net->d_first_w);
  if (net->d_first_w16) (void)hipFree(net->d_first_w16);
  if (net->d_first_tab) (void)hipFree(net->d_first_tab);
  if (net->d_sep) (void)hipFree(net->d_sep);
  if (net->d_l1_w16) (void)hipFree(net->d_l1_w16);
  if (net->d_l1_bias) (void)hipFree(net->d_l1_bias);
  if (net->d_c1a_w16) (void)hipFree(net->d_c1a_w16);
  if (net->d_c1a_bias) (void)hipFree(net->d_c1a_bias);
  if (net->d_l1_f32) (void)hipFree(net->d_l1_f32);
  if (net->d_c1a_f32) (void)hipFree(net->d_c1a_f32);
  if (net->d_l1_bias32) (void)hipFree(net->d_l1_bias32);
  if (net->d_c1a_bias32) (void)hipFree(net->d_c1a_bias32);
  for (auto& L : net->comp) free_layer(L);
  delete net;
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

static int encoder_chunk(orca_ctx* ctx, orca_net* net, const SeqSource& src, long n1, float* const buf[3],
                         long ld1, float** out, long* out_ld, long* out_n) {
  hipStream_t s = ctx->stream;
  int P = 0;
  long n = n1, ld = ld1;
  const float* x = src.x;
  long sx_c = src.sx_c, sx_l = src.sx_l;
  // planar 16-bit activation formats of conv_p16.h for stages 1-3 (96 % of the FLOPs):
  //   f16x2 -> P16 (2-way split fp16, fp32-class);  bf16 -> B16 (one bf16 plane, throughput mode of BASELINE config 3)
  static const bool no_p16 = getenv("ORCA_NO_P16") != nullptr, no_b16 = getenv("ORCA_NO_B16") != nullptr;   // A/B switches
  const bool use_b16 = net->precision == ORCA_PRECISION_BF16 && !no_b16;
  const bool use_p16 = (net->precision == ORCA_PRECISION_F16X2 && !no_p16) || use_b16;
  const int fmt = use_b16 ? 1 : 0;
  // the channel-last split-operand pipeline (bf16x3 / bf16x2, or f16x2 with ORCA_NO_P16): stage 1 composed - from PACKED bases only: the
  // first-layer GEMM splits its X operand into fp16 parts, exact for 0 / 0.25 / 1, while these modes promise fp32 range for arbitrary float rows
  const bool compose_nlc = !use_p16 && net->precision != ORCA_PRECISION_F32 && net->d_c1a_w16 && getenv("ORCA_NO_COMPOSE") == nullptr &&
                           getenv("ORCA_NO_COMPOSE25") == nullptr && src.codes;
  // exact-fp32 mode: stage 1's linear groups composed as in the 16-bit modes (conv_p16.h: first_taps_f32_kernel reads the source directly)
  const bool compose32 = net->precision == ORCA_PRECISION_F32 && net->d_l1_f32 && getenv("ORCA_NO_COMPOSE") == nullptr && getenv("ORCA_NO_COMPOSE25") == nullptr;
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
    int st0 = 0;
    if (use_p16) {
      // stages 1-3 on planar 16-bit activations with LDS-DMA staging (conv_p16.h)
      const ConvLayer* L = net->convs.data();
      FirstP16Args fa;
      fa.x = x; fa.sc = sx_c; fa.sl = sx_l; fa.n = n1; fa.w = nullptr; fa.bias = L[0].d_bias; fa.y = reinterpret_cast<f32x4*>(buf[1]);
      fa.y_plen = p16_plen(n1); fa.flag = ctx->d_flag;
      fa.w = net->d_first_w;
      // Composed linear pairs (default; ORCA_NO_COMPOSE=1 = the two-conv form, kept as the in-library A/B reference): lconv1 is ONE
      // 17-tap first layer straight into buf[LO] (K = 68 MFMA GEMM from the bases / float rows), lconv2 / lconv3 are 17-tap planar
      // convs; conv1.a, linear up to its ReLU, is composed with lconv1 as well (25 taps from the bases).  The end positions of each
      // group are redone exactly by the edge-fix chain (lconv_edge_layer_kernel).
      const bool no_compose = getenv("ORCA_NO_COMPOSE") != nullptr;   // read per call: the tests flip it
      const bool compose = !no_compose && net->d_l1_w16 != nullptr;
      // packed input: the first layer is fused into the input-tile producer of the conv that follows it (conv_p16.h, F1)
      static const bool no_fuse1 = getenv("ORCA_NO_FUSE1") != nullptr;   // A/B switch
      const bool fuse1 = src.codes && !no_fuse1 && fmt == 0 && !compose;
      FusedFirst f1;
      f1.codes = src.codes; f1.nmask = src.nmask; f1.origin = src.origin; f1.codes_L = src.codes_L; f1.codes_off = src.codes_off; f1.reverse = src.reverse;
      f1.table = net->d_first_tab; f1.bias = L[0].d_bias;
      int T = 1, LO = 2, S = 0;   // buffer roles: T holds the current input
      // (guards and tails of every planar tensor are zeroed by p16_zero_pads_kernel AFTER its producer: the conv kernels
      // write the units of a ragged last tile unmasked)
      const bool flat = src.codes || (sx_c == 1 && sx_l == 4 && al16(x));
      // conv1.a joins the composed group (25 taps from the bases + ReLU, K = 112): ORCA_NO_COMPOSE25=1 keeps it a 64 -> 64 launch
      const bool compose25 = compose && getenv("ORCA_NO_COMPOSE25") == nullptr && net->d_c1a_w16 != nullptr;
      // ... and with packed bases the residual lout1 is computed inside conv1.b's epilogue (conv_p16.h, RL) instead of being stored by a
      // 17-tap first-layer launch and re-read: ORCA_NO_RL=1 keeps the stored form
      const bool res_from_bases = compose25 && src.codes && getenv("ORCA_NO_RL") == nullptr;
      // ... and in the throughput mode (B16 planes) the whole stage is ONE kernel from the bases (conv_stage1.h): conv1.b's input tiles are
      // produced in LDS by the matrix cores, a1 is neither written nor re-read.  ORCA_NO_STAGE1_FUSE=1 (read per call): the two-launch form
      const bool stage1_fused = res_from_bases && fmt == 1 && L[3].cin == 64 && L[3].cout == 64 && L[3].d_wb16p && getenv("ORCA_NO_STAGE1_FUSE") == nullptr;
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
      n = n1;
      static const bool no_st4 = getenv("ORCA_NO_P16_STAGE4") != nullptr;   // A/B switch: stage 4 on the register-staged kernel again
      const int nplanar = no_st4 ? 3 : 4;                                   // stages on the planar kernels (pools 4, 4 fused; 5 as a planar pass)
      bool pooled_ahead = false;
      for (st0 = 0; st0 < nplanar; ++st0) {
        const ConvLayer* Ls = L + 4 * st0;
        const int C = Ls[3].cout;
        if (kEncPools[st0] == 5 && pooled_ahead) {   // the previous stage's last conv already pooled (conv_p16p5.h): buf[S] holds n / 5 positions
          n /= 5;
          pooled_ahead = false;
        } else if (kEncPools[st0] == 5) {   // MaxPool1d(5) in front of this stage: previous output buf[S] (n positions) -> buf[LO] -> becomes S
          ORCA_TRY(launch_p16_pool5(ctx, buf[S], buf[LO], Ls[0].cin, n, fmt));
          n /= 5;
          ORCA_TRY(launch_p16_zero_pads(ctx, buf[LO], Ls[0].cin, n, fmt));
          const int t_ = S; S = LO; LO = t_;
        }
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
        } else if (st0 + 1 < nplanar && kEncPools[st0 + 1] == 5 && C == 128 && getenv("ORCA_NO_POOL5_FUSE") == nullptr) {
          ORCA_TRY(launch_conv1d_p16(ctx, Ls[3], buf[T], buf[S], buf[LO], n, 1, 3, nullptr, fmt));          // relu(.)+lout, MaxPool1d(5) (conv_p16p5.h)
          ORCA_TRY(launch_p16_zero_pads(ctx, buf[S], C, n / 5, fmt));
          pooled_ahead = true;
        } else if (st0 + 1 < nplanar) {
          ORCA_TRY(launch_conv1d_p16(ctx, Ls[3], buf[T], buf[S], buf[LO], n, 1, 0, nullptr, fmt));          // relu(.)+lout, planar; pooled by the next stage
          ORCA_TRY(launch_p16_zero_pads(ctx, buf[S], C, n, fmt));
        } else {
          ORCA_TRY(launch_conv1d_p16(ctx, Ls[3], buf[T], buf[S], buf[LO], n, 1, 2, nullptr, fmt));          // fp32 channel-last hand-over
        }
      }
      P = S;
    }
    for (int st = st0; st < 7; ++st) {
      const ConvLayer* L = &net->convs[4 * st];
      if (kEncPools[st] == 4 && st0 == 0) {
        n = n / 4;  // MaxPool1d(4) was fused into the epilogue of the previous stage's last conv
      } else if (kEncPools[st] > 1) {
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
  // $ORCA_UNET_NLC_MIN (positions per launch below which the exact kernels run; read per call) restores that for A/B checks.
  const char* nlc_env = getenv("ORCA_UNET_NLC_MIN");
  const long nlc_min = nlc_env ? atol(nlc_env) : 0;
  if (net->precision != ORCA_PRECISION_F32 && (long)B * n >= nlc_min) return unet_forward_nlc(ctx, net, x, sx_b, sx_c, sx_l, B, n, outs, nlev, up_only);
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
  // A Decoder is a chain of ~90 dependent launches per map.  With the four-row kernel (conv2d_m16q.h) a launch carries the WHOLE batch: both
  // strands of a level are 252-256 workgroups = one round on 256 CUs.  The one-row kernel of rounds 2-3 (ORCA_NO_M16Q=1) ran an even batch
  // as two half-batches on two streams, one half's ramps and tails filled by the other half's workgroups; ORCA_DECODER_TWO_STREAMS=1 /
  // ORCA_DECODER_ONE_STREAM=1 force either (read per call).
  const bool one_stream = getenv("ORCA_DECODER_ONE_STREAM") != nullptr || (getenv("ORCA_NO_M16Q") == nullptr && getenv("ORCA_DECODER_TWO_STREAMS") == nullptr);
  if (B < 2 || (B & 1) || one_stream) return run(0, B);
  if (!ctx->aux) {
    HIPCHECK(hipStreamCreateWithFlags(&ctx->aux, hipStreamNonBlocking));
    HIPCHECK(hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming));
    HIPCHECK(hipEventCreateWithFlags(&ctx->ev_join, hipEventDisableTiming));
  }
  // (the aux stream is internal: whatever happens below, the caller's stream is made to wait for it before this returns, so that
  // work queued on aux can never outlive the call unordered - the fp16-range flag read-back and the caller's allocator see it)
  hipStream_t main_s = ctx->stream;
  int rc = ORCA_OK;
  if (hipEventRecord(ctx->ev_fork, main_s) != hipSuccess || hipStreamWaitEvent(ctx->aux, ctx->ev_fork, 0) != hipSuccess) {
    (void)hipGetLastError();
    return run(0, B);                    // could not fork: everything on the caller's stream
  }
  rc = run(0, B / 2);
  ctx->stream = ctx->aux;
  const int rc2 = rc == ORCA_OK ? run(B / 2, B / 2) : rc;
  ctx->stream = main_s;
  const hipError_t e1 = hipEventRecord(ctx->ev_join, ctx->aux);
  const hipError_t e2 = e1 == hipSuccess ? hipStreamWaitEvent(main_s, ctx->ev_join, 0) : e1;
  if (e2 != hipSuccess) {                // the join itself failed: fall back to a host-side wait, then report
    (void)hipStreamSynchronize(ctx->aux);
    if (rc2 == ORCA_OK) return fail(ORCA_EHIP, "decoder: joining the internal stream failed: %s", hipGetErrorString(e2));
  }
  return rc2;
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

extern "C" int orca_net_num_targets(orca_net* net, int* num_2d) {
  if (!net || !num_2d) return fail(ORCA_EINVAL, "orca_net_num_targets: NULL argument");
  *num_2d = net->num_2d;
  return ORCA_OK;
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

// ---------------------------------------------------------------------------
// multi-GPU exchange: RCCL, resolved at run time
// ---------------------------------------------------------------------------
struct Id128 { char internal[ORCA_COMM_ID_BYTES]; };   // = ncclUniqueId (rccl.h: NCCL_UNIQUE_ID_BYTES 128)
namespace {
struct RcclApi {
  void* lib = nullptr;
  int (*GetUniqueId)(void*) = nullptr;
  int (*CommInitRank)(void**, int, /* ncclUniqueId by value */ Id128, int) = nullptr;
  int (*CommDestroy)(void*) = nullptr;
  int (*AllGather)(const void*, void*, size_t, int, void*, hipStream_t) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
};
}  // namespace

static int find_loaded_rccl(struct dl_phdr_info* info, size_t, void* out) {
  const char* n = info->dlpi_name;
  if (n && strstr(n, "librccl.so")) { *static_cast<std::string*>(out) = n; return 1; }
  return 0;
}

static std::string g_rccl_err;      // why RCCL could not be loaded (written once, under the call_once below)
static RcclApi* rccl_api() {
  static RcclApi api;
  static std::once_flag once;
  std::call_once(once, [] {
    std::string loaded;
    dl_iterate_phdr(find_loaded_rccl, &loaded);      // PyTorch-ROCm brings its own RCCL: share it
    void* h = nullptr;
    if (!loaded.empty()) h = dlopen(loaded.c_str(), RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) {
      const char* e = dlerror();       // once: the call clears the state
      g_rccl_err = e ? e : "librccl.so not found";
      return;
    }
    api.GetUniqueId = reinterpret_cast<decltype(api.GetUniqueId)>(dlsym(h, "ncclGetUniqueId"));
    api.CommInitRank = reinterpret_cast<decltype(api.CommInitRank)>(dlsym(h, "ncclCommInitRank"));
    api.CommDestroy = reinterpret_cast<decltype(api.CommDestroy)>(dlsym(h, "ncclCommDestroy"));
    api.AllGather = reinterpret_cast<decltype(api.AllGather)>(dlsym(h, "ncclAllGather"));
    api.GetErrorString = reinterpret_cast<decltype(api.GetErrorString)>(dlsym(h, "ncclGetErrorString"));
    if (!api.GetUniqueId || !api.CommInitRank || !api.CommDestroy || !api.AllGather) {
      g_rccl_err = "librccl.so lacks ncclGetUniqueId / ncclCommInitRank / ncclCommDestroy / ncclAllGather";
      return;
    }
    api.lib = h;
  });
  return api.lib ? &api : nullptr;
}

struct orca_comm {
  void* comm = nullptr;   // ncclComm_t
  int nranks = 1, rank = 0, device = 0;
};

static int rccl_fail(RcclApi* r, const char* what, int rc) {
  return fail(ORCA_EHIP, "%s failed: %s (RCCL result %d)", what, r && r->GetErrorString ? r->GetErrorString(rc) : "?", rc);
}

extern "C" int orca_comm_unique_id(void* id128_host) {
  if (!id128_host) return fail(ORCA_EINVAL, "orca_comm_unique_id: NULL argument");
  RcclApi* r = rccl_api();
  if (!r) return fail(ORCA_ENODEV, "RCCL (librccl.so) could not be loaded: %s", g_rccl_err.c_str());
  Id128 id;
  const int rc = r->GetUniqueId(&id);
  if (rc != 0) return rccl_fail(r, "ncclGetUniqueId", rc);
  memcpy(id128_host, &id, sizeof id);
  return ORCA_OK;
}

extern "C" int orca_comm_init_rank(orca_ctx* ctx, int nranks, int rank, const void* id128_host, orca_comm** out) {
  if (!ctx || !id128_host || !out) return fail(ORCA_EINVAL, "orca_comm_init_rank: NULL argument");
  if (nranks < 1 || rank < 0 || rank >= nranks) return fail(ORCA_EINVAL, "orca_comm_init_rank: rank %d of %d", rank, nranks);
  RcclApi* r = rccl_api();
  if (!r) return fail(ORCA_ENODEV, "RCCL (librccl.so) could not be loaded: %s", g_rccl_err.c_str());
  HIPCHECK(hipSetDevice(ctx->device));
  Id128 id;
  memcpy(&id, id128_host, sizeof id);
  orca_comm* c = new orca_comm();
  c->nranks = nranks; c->rank = rank; c->device = ctx->device;
  const int rc = r->CommInitRank(&c->comm, nranks, id, rank);
  if (rc != 0) { delete c; return rccl_fail(r, "ncclCommInitRank", rc); }
  *out = c;
  return ORCA_OK;
}

extern "C" int orca_comm_destroy(orca_comm* comm) {
  if (!comm) return ORCA_OK;
  RcclApi* r = rccl_api();
  (void)hipSetDevice(comm->device);
  if (r && comm->comm) (void)r->CommDestroy(comm->comm);
  delete comm;
  return ORCA_OK;
}

extern "C" int orca_allgather(orca_ctx* ctx, orca_comm* comm, const float* send, float* recv, size_t count) {
  if (!ctx || !comm || !send || !recv) return fail(ORCA_EINVAL, "orca_allgather: NULL argument");
  if (comm->device != ctx->device) return fail(ORCA_EINVAL, "orca_allgather: communicator lives on device %d, context on %d", comm->device, ctx->device);
  if (count == 0) return ORCA_OK;
  RcclApi* r = rccl_api();
  if (!r) return fail(ORCA_ENODEV, "RCCL is not loaded");
  HIPCHECK(hipSetDevice(ctx->device));
  const int rc = r->AllGather(send, recv, count, /* ncclFloat32 */ 7, comm->comm, ctx->stream);
  if (rc != 0) return rccl_fail(r, "ncclAllGather", rc);
  return ORCA_OK;
}

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
