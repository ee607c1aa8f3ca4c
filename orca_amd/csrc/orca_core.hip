// orca_core.hip - errors, contexts and the workspace arena, weight upload (BatchNorm-folded convs -> the kernels' operand packs, composed linear groups), nets
// Part of liborca_hip.so (include/orca_hip.h is the ABI; orca_internal.h what the units share).
#include "orca_internal.h"

// ---------------------------------------------------------------------------
// errors
// ---------------------------------------------------------------------------
static thread_local std::string g_err;

int fail(int code, const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}

// ---------------------------------------------------------------------------
// workspace arena
// ---------------------------------------------------------------------------
int ws_ensure(orca_ctx* ctx, size_t bytes) {
  ctx->ws_off = 0;
  // the arena is reused by every call: kernels of an EARLIER call on a different stream may still be reading it
  if (ctx->ws_used && ctx->ws_stream != ctx->stream) HIPCHECK(hipStreamSynchronize(ctx->ws_stream));
  ctx->ws_stream = ctx->stream;
  ctx->ws_used = true;
  if (bytes <= ctx->ws_bytes) return ORCA_OK;
  // grow in steps of 1 GiB (64 MiB below that): a call that needs a few kilobytes more than the last one must not free and re-allocate a 30 GB
  // arena - that pair costs 1.1-1.5 s on some boxes of this platform (found in the stage cache's build right behind a 40 Mb Encoder call)
  const size_t step = bytes >= (size_t(1) << 30) ? (size_t(1) << 30) : (size_t(64) << 20);
  bytes = (bytes + step - 1) / step * step;
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

float* ws_take(orca_ctx* ctx, size_t nfloats) {
  size_t bytes = ru256(nfloats * sizeof(float));
  if (ctx->ws_off + bytes > ctx->ws_bytes) return nullptr;
  float* p = reinterpret_cast<float*>(ctx->ws + ctx->ws_off);
  ctx->ws_off += bytes;
  return p;
}

// ---------------------------------------------------------------------------
// layers: one folded conv -> the packs its kernels read
// ---------------------------------------------------------------------------
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

void free_layer(ConvLayer& L) {
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
int make_layer(const orca_conv_desc& d, ConvLayer* out) {
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
void compose_pair(const orca_conv_desc& c1, const orca_conv_desc& c2, std::vector<double>* w17, std::vector<double>* b17) {
  std::vector<double> w1((size_t)c1.cout * c1.cin * 9), b1(c1.cout);
  for (size_t i = 0; i < w1.size(); ++i) w1[i] = c1.weight_host[i];
  for (int i = 0; i < c1.cout; ++i) b1[i] = c1.bias_host[i];
  compose_taps(w1, b1, c1.cin, c1.cout, 9, c2, w17, b17);
}

// 17-tap planar conv layer: packs in the layout of the k9 kernels with TWICE the K-chunks - chunk 2c + h holds taps 9h .. 9h+8
// of input channels 16c .. 16c+15 (32c .. in the bf16 pack); tap 17 does not exist: zero weights
int make_layer17(int cin, int cout, const std::vector<double>& w17, const std::vector<double>& b17, ConvLayer* out) {
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

extern "C" int orca_net_set_encoder_form(orca_net* net, int form) {
  if (!net || net->kind != ORCA_NET_ENCODER) return fail(ORCA_EINVAL, "orca_net_set_encoder_form: not an Encoder net");
  if (form < ORCA_ENCODER_FORM_DEFAULT || form > ORCA_ENCODER_FORM_TWO_CONV) return fail(ORCA_EINVAL, "orca_net_set_encoder_form: form %d unknown", form);
  net->enc_form = form;
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


extern "C" int orca_net_num_targets(orca_net* net, int* num_2d) {
  if (!net || !num_2d) return fail(ORCA_EINVAL, "orca_net_num_targets: NULL argument");
  *num_2d = net->num_2d;
  return ORCA_OK;
}

