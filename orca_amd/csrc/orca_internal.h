// orca_internal.h - host-side declarations shared by the translation units of liborca_hip.so (not installed; include/orca_hip.h is the ABI):
//   orca_core.hip          errors, contexts + workspace arena, weight upload (BatchNorm-folded convs -> the kernels' packs, composed groups), nets
//   orca_encoder.hip       Conv1d launchers, the Encoder (orca_modules.py:929-980) and the U-net encoders (:1151-1169, :1388-1406)
//   orca_decoder.hip       Conv2d launchers, Decoder / Decoder_1m (:461-488, :782-800), strand merge, background block means, the observed-data
//                          smoother, the 2-bit genome expander
//   orca_comm.hip          the RCCL communicator of the sharded Encoder (dlopen'ed)
//   orca_test_entries.hip  single-layer entry points for the kernel tests
// Kernels live in the *.h files next to these; every non-template kernel there is `static`, so a header may be included by several units.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "conv_kernels.h"   // f32x4
#include "orca_hip.h"

// ---- errors (orca_core.hip) -----------------------------------------------------------------------------------------------------------
int fail(int code, const char* fmt, ...) __attribute__((format(printf, 2, 3)));   // sets this thread's orca_last_error(), returns `code`

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

// ---- context + workspace arena -----------------------------------------------------------------------------------------------------
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
};

int ws_ensure(orca_ctx* ctx, size_t bytes);        // start of a call: the arena holds at least `bytes` (waits for an earlier call on another stream)
float* ws_take(orca_ctx* ctx, size_t nfloats);     // next 256-byte aligned piece of the arena (NULL when it is exhausted)

#ifndef ORCA_MAX_TARGETS
#define ORCA_MAX_TARGETS 8   // num_2d of the multi-target decoders (orca_leukemia.py:512-990); hidden width F = max(5, T) (also misc_kernels.h)
#endif
#define ORCA_EDGE_SLAB (40 * 128)   // floats per scratch slab of the edge-fix chain (orca_ctx::d_edge holds four; lconv_edge_layer_kernel)

// ---- layers / nets ----------------------------------------------------------------------------------------------------------------------
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
  int enc_form = ORCA_ENCODER_FORM_DEFAULT;   // Encoder: which of the algebraically equal forms of stage 1-3's linear groups runs (orca_net_set_encoder_form)
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

int make_layer(const orca_conv_desc& d, ConvLayer* out);                                   // one BN-folded conv -> every pack its kernels read
int make_layer17(int cin, int cout, const std::vector<double>& w17, const std::vector<double>& b17, ConvLayer* out);   // a composed 17-tap group
void compose_pair(const orca_conv_desc& c1, const orca_conv_desc& c2, std::vector<double>* w17, std::vector<double>* b17);
void free_layer(ConvLayer& L);

// ---- launchers used across units -------------------------------------------------------------------------------------------------------
static inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
// plane length of a planar 16-bit sequence tensor in 16-byte units: 8 guard units on the left, >= 24 on the right (a 17-tap conv's second tap
// half reads 9 units past the last tile); the +1 keeps the zero stores of a pooled output's ragged last tile inside the plane for every n
static inline long p16_plen(long n) { return ((n + 512) / 512) * 512 + 32; }
struct FusedFirst;
// orca_encoder.hip
int launch_conv1d(orca_ctx* ctx, const ConvLayer& L, const float* x, long x_bs, long ldx, float* y, long y_bs, long ldy, const float* r1, const float* r2,
                  int B, long n, int relu, int tile, int y_nlc = 0);
int launch_conv1d_b16(orca_ctx* ctx, const ConvLayer& L, int precision, const float* x, long x_bs, float* y, long y_bs, const float* r1, int B, long n, int relu,
                      int pool4 = 0, const float* r2 = nullptr);
int launch_conv1d_p16(orca_ctx* ctx, const ConvLayer& L, const float* x, void* y, const float* r1, long n, int relu, int out_mode, const FusedFirst* f1 = nullptr,
                      int fmt = 0);
int launch_p16_zero_pads(orca_ctx* ctx, float* base, int C, long n_valid, int fmt = 0);
int launch_pool(orca_ctx* ctx, const float* x, long ldx, float* y, long ldy, long rows, long n_out, int k);
// orca_decoder.hip
int launch_conv2d(orca_ctx* ctx, const ConvLayer& L, const float* x, long x_bs, float* y, long y_bs, const float* r, long r_bs, int B, int n, int relu);
// dilated 3x3 conv on M16 maps; mode 0 = f16x2, 1 = bf16, 2 = f16 (see the definition)
int launch_conv2d_m16(orca_ctx* ctx, const ConvLayer& L, const f32x4* x, long x_bs, int x_oct, f32x4* y, long y_bs, int y_oct, const f32x4* r, long r_bs, int B,
                      int n, int relu, int mode, int chunk0 = 0, int nchunks_ = 0, const float* tab = nullptr, long tab_bs = 0);
