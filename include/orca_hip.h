/*
 * orca_hip.h - C ABI of liborca_hip.so, the MI355X (gfx950) implementation of
 * the Orca hot path (Encoder -> Encoder2/Encoder3 -> Decoder cascade).
 *
 * The reference (jzhoulab/orca) has no FFI / plugin registry: the hot path is
 * reached through a duck-typed Python model protocol.
 * Each entry point below states which reference call it replaces; the Python
 * host in orca_amd/ binds them with ctypes (INTEGRATION.md shows the stub a
 * reference maintainer would add).
 *
 * Conventions
 *  - plain C types only; every function returns 0 (ORCA_OK) or a negative
 *    ORCA_E* code and records a message retrievable with orca_last_error()
 *    (thread-local).  No exceptions cross the boundary.
 *  - all `const float*` / `float*` data arguments are CALLER-OWNED DEVICE
 *    pointers (e.g. torch.Tensor.data_ptr() of a ROCm tensor) unless the
 *    parameter name ends in `_host`.
 *  - strides are in ELEMENTS (floats), so non-contiguous torch views (the
 *    `.transpose(1,2)` of orca_predict.py:334, the `[:, :, s:s+250]` slices of
 *    :358) are passed without a copy.
 *  - work is enqueued asynchronously on the HIP stream the context was
 *    created with (or was last given via orca_ctx_set_stream); the library
 *    never synchronises the device on the forward path.
 *  - the library owns only its weight copies (orca_net) and a per-context
 *    workspace that grows on demand.
 *  - one orca_ctx per (host thread, GPU).  No global mutable state.
 */
#ifndef ORCA_HIP_H
#define ORCA_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORCA_OK 0
#define ORCA_EINVAL (-1)  /* bad argument / shape mismatch            */
#define ORCA_EHIP (-2)    /* a HIP runtime call failed                */
#define ORCA_ENOMEM (-3)  /* workspace / weight allocation failed     */
#define ORCA_ENODEV (-4)  /* no usable gfx950 device                  */

#define ORCA_ABI_VERSION 1

typedef struct orca_ctx orca_ctx;
typedef struct orca_net orca_net;

/* Which reference nn.Module an orca_net stands for. */
enum orca_net_kind {
  ORCA_NET_ENCODER = 1,    /* orca_modules.Encoder     (orca_modules.py:803-980)   */
  ORCA_NET_ENCODER2 = 2,   /* orca_modules.Encoder2    (orca_modules.py:984-1169)  */
  ORCA_NET_ENCODER3 = 3,   /* orca_modules.Encoder3    (orca_modules.py:1279-1406) */
  ORCA_NET_DECODER = 4,    /* orca_modules.Decoder     (orca_modules.py:16-488)    */
  ORCA_NET_DECODER_1M = 5, /* orca_modules.Decoder_1m  (orca_modules.py:491-800)   */
  ORCA_NET_ENCODER2B = 6   /* orca_modules.Encoder2b   (orca_modules.py:1173-1276): Encoder2 without the expanding path */
};

/* Decoder(upsample_mode=...) of orca_modules.py:17,430; containers use bilinear
 * (orca_models.py:45-50). */
#define ORCA_UPSAMPLE_NEAREST 0
#define ORCA_UPSAMPLE_BILINEAR 1

/* One convolution with eval-mode BatchNorm already folded into it on the host
 * (w' = w*gamma/sqrt(var+1e-5), b' = (b-mean)*gamma/sqrt(var+1e-5)+beta).
 * weight_host: [cout][cin][ksize] (1-D, ksize=9) or [cout][cin][ksize][ksize]
 * (2-D, ksize=3 or 1), fp32, row-major exactly as in the reference checkpoints
 * (`*.statedict`, orca_models.py:53-123). */
typedef struct orca_conv_desc {
  const float* weight_host;
  const float* bias_host;
  int32_t cout;
  int32_t cin;
  int32_t ksize;
  int32_t dilation;
} orca_conv_desc;

/* ---- library / context ------------------------------------------------- */

int orca_abi_version(void);
const char* orca_last_error(void);

/* Number of visible HIP devices (0 if none; never fails). */
int orca_device_count(void);

/* Replaces: implicit torch device/stream state used by `.cuda()` at
 * orca_predict.py:334.  hip_stream may be NULL (default stream). */
int orca_ctx_create(int device, void* hip_stream, orca_ctx** out);
int orca_ctx_destroy(orca_ctx* ctx);
int orca_ctx_set_stream(orca_ctx* ctx, void* hip_stream);
/* Bytes of device workspace currently held by the context. */
int orca_ctx_workspace_bytes(orca_ctx* ctx, size_t* out);
/* Free the workspace (it is re-grown on demand). */
int orca_ctx_release_workspace(orca_ctx* ctx);

/* ---- per-kernel timing (bench.py roofline) ---------------------------------
 * When enabled, every conv1d launch over >= 65536 positions (the Encoder's
 * stage 1-4 convolutions, >96 % of the path's FLOPs) is bracketed by HIP events
 * on the context's stream.  orca_ctx_get_timing synchronises the stream, fills
 * up to `max` records (oldest first), stores the number available in *n and
 * clears the log. */
typedef struct orca_kernel_time {
  int32_t cout, cin, tile, batch; /* tile > 0: position tile of the exact-fp32 kernel; < 0: kernel family = -1..-4 conv_bf16s.h (bf16, bf16x2, bf16x3, f16x2),
                                   * -5 / -6 conv_p16.h (P16 / B16), -7 / -8 conv_ws.h, -9 / -10 conv_p16w1.h (P16 / B16), -12 / -13 conv_p16p5.h (P16 / B16), -14 conv_p16x.h */
  int64_t n;      /* positions per batch row            */
  float ms;       /* elapsed between the two HIP events */
  int32_t ksize;  /* taps of the launch: 9, or 17 = a composed linear pair (its algorithmic FLOPs are the pair's) */
} orca_kernel_time;
int orca_ctx_set_timing(orca_ctx* ctx, int enable);
int orca_ctx_get_timing(orca_ctx* ctx, orca_kernel_time* out, int max, int* n);

/* Launch counters of the context since its creation, counts4 = { channel-last Conv1d launches on the short-row kernel (conv_small.h:
 * rows of <= 2 048 positions), on the chunk-after-chunk kernel (conv_bf16s.h), planar P16 / B16 Conv1d launches, fused Decoder pair
 * launches }.  Diagnostics only (which kernel family a shard's short last stages ran on: bench.py's N-rank sections); no reference
 * counterpart. */
int orca_ctx_launch_counts(orca_ctx* ctx, int64_t* counts4);

/* ---- weights -------------------------------------------------------------
 * Replaces: nn.Module.load_state_dict + .cuda() of the reference containers
 * (orca_models.py:53-133).  `convs` lists the folded convolutions of the module
 * in forward order (documented per kind in DESIGN.md section "conv order");
 * shapes are validated against the architecture.  The library re-lays the
 * weights out for its MFMA kernels and keeps the device copy until
 * orca_net_free. */
int orca_net_create(orca_ctx* ctx, int kind, const orca_conv_desc* convs, int n_convs,
                    int upsample_mode, orca_net** out);
int orca_net_free(orca_net* net);

/* Arithmetic used for the Conv1d stacks of an Encoder net (Decoder / Decoder_1m nets take _F32, _F16X2 and _BF16).
 *  ORCA_PRECISION_F32   : v_mfma_f32_32x32x2_f32, exact fp32 products (default)
 *  ORCA_PRECISION_BF16X3: operands split into 3 bf16 parts, 6 bf16-MFMA products per fp32
 *                         product, fp32 accumulate - fp32-class error
 *                         at 2.67x the fp32-MFMA rate
 *  ORCA_PRECISION_BF16X2: 2-way split, 3 products (~2^-17 relative per product)
 *  ORCA_PRECISION_BF16  : plain bf16 operands, one product (the throughput mode of BASELINE config 3): Encoder stages 1-3
 *                         on single-plane bf16 activations (2 bytes per element in HBM, conv_p16.h "B16"), Decoders with
 *                         bf16 operands and fp32 feature maps */
#define ORCA_PRECISION_F32 0
#define ORCA_PRECISION_BF16 1
#define ORCA_PRECISION_BF16X2 2
#define ORCA_PRECISION_BF16X3 3
/*  ORCA_PRECISION_F16X2 : operands split into 2 fp16 parts (22 significant bits), 3 fp16-MFMA products,
 *                         fp32 accumulate: ~2^-22 relative error (fp32-class end to end) at 5.3x the
 *                         fp32-MFMA rate; requires |activation|, |weight| < 65504 (fp16 range). */
#define ORCA_PRECISION_F16X2 4
/*  ORCA_PRECISION_F16   : Decoder / Decoder_1m nets only - ONE fp16 plane per feature map and one MFMA product (the rate and
 *                         traffic of _BF16, 11 instead of 8 significant bits; fp16 range guard as _F16X2). */
#define ORCA_PRECISION_F16 5
int orca_net_set_precision(orca_net* net, int precision);
/* ORCA_NET_ENCODER only: which of the ALGEBRAICALLY EQUAL forms of stage 1-3's linear groups runs (orca_modules.py:811-852: `lconv_i` is
 * Conv-BN-Conv-BN with no nonlinearity, `conv_i`'s first conv is linear up to its ReLU; eval-mode BatchNorm is affine, so each group is ONE
 * convolution whose weights the library composes on the host in float64).  The default runs every composed form the weights allow; a group
 * whose composed weights leave the fp16 range keeps the reference's layer sequence by itself.  The other values FORCE a less composed form:
 * they exist so that the parity suite can run the fallbacks on ordinary weights and compare them with the default (tests/test_gpu_nets.py).
 *   _DEFAULT         lconv1 = 17 taps, conv1.a o lconv1 = 25 taps, both straight from the bases; with packed bases lout1 is computed in
 *                    conv1.b's epilogue (never stored); lconv2 / lconv3 = single 17-tap convs
 *   _STORED_RESIDUAL as _DEFAULT, lout1 written by a 17-tap first-layer launch and re-read (what float input rows get anyway)
 *   _LCONV1_ONLY     conv1.a stays a 64 -> 64 launch of its own
 *   _TWO_CONV        the reference's layer sequence, conv by conv */
#define ORCA_ENCODER_FORM_DEFAULT 0
#define ORCA_ENCODER_FORM_STORED_RESIDUAL 1
#define ORCA_ENCODER_FORM_LCONV1_ONLY 2
#define ORCA_ENCODER_FORM_TWO_CONV 3
int orca_net_set_encoder_form(orca_net* net, int form);
/* ORCA_PRECISION_F16X2 only: the kernels raise a device flag when an activation leaves the fp16
 * range (the result of that forward is then invalid).  This call waits for the context's stream,
 * returns the flag in *flag and clears it - the host falls back to ORCA_PRECISION_BF16X3. */
int orca_ctx_take_overflow(orca_ctx* ctx, int* flag);

/* ---- forward passes -------------------------------------------------------- */

/* Replaces: model.net0(x) = Encoder.forward (orca_modules.py:929-980).
 * x: [B,4,L] fp32 with element strides (sx_b,sx_c,sx_l); any float content
 * (one-hot rows, 0.25 'N' rows, ...).  Computes output bins [bin_lo,bin_hi)
 * (4000 bp each; bin_hi<=0 means "to the end") and writes
 * out[b*so_b + c*so_c + (bin-bin_lo)], c<128.  A sub-range is how the
 * independent sequence blocks shard across GPUs.
 * chunk_bp: internal processing chunk (multiple of 4000; <=0 = default: the whole input up to 32 Mb, 128 Mb chunks beyond - workspace = 768 B per chunk base);
 * chunks carry a 112 kb input halo each side exactly like the reference's
 * 800 kb blocks, so results do not depend on it. */
int orca_encoder_forward(orca_ctx* ctx, orca_net* net, const float* x, int64_t sx_b, int64_t sx_c,
                         int64_t sx_l, int B, int64_t L, int64_t bin_lo, int64_t bin_hi, float* out,
                         int64_t so_b, int64_t so_c, int64_t chunk_bp);

/* Packed-sequence input.  The reference materialises every window as a 512 MB float [L,4] array per strand
 * (selene_utils2.py:216-272) and a second, flipped copy for the reverse complement (orca_predict.py:324-329).
 * orca_pack_sequence turns a float view (device) into 1 byte per base - 0..3 = A,C,G,T one-hot rows, 4 = the
 * 0.25 x 4 'N' row - and reports *packable = 0 if some row is neither (the float path must then be used; it
 * synchronises the stream to return that flag).  orca_encoder_forward_codes runs the Encoder straight from the
 * codes ([B][L] bytes, batch stride sc_b); reverse != 0 encodes the REVERSE COMPLEMENT of the same buffer
 * (index L-1-i, code 3-c), so no second sequence copy ever exists.  The one-hot expansion happens in LDS inside
 * the first-layer kernel. */
int orca_pack_sequence(orca_ctx* ctx, const float* x, int64_t sx_c, int64_t sx_l, int64_t L, uint8_t* codes_out, int* packable);
int orca_encoder_forward_codes(orca_ctx* ctx, orca_net* net, const uint8_t* codes, int64_t sc_b, int reverse, int B, int64_t L,
                               int64_t bin_lo, int64_t bin_hi, float* out, int64_t so_b, int64_t so_c, int64_t chunk_bp);
/* The same from a WINDOW of the sequence: `codes` holds only bases [win_origin, win_origin + win_len) of the L-base
 * sequence (batch stride sc_b) - what a rank of a sharded run keeps of it: its bin range +- the 112 kb halo, i.e.
 * bases [bin_lo*4000 - 112000, bin_hi*4000 + 112000) for the forward strand and the mirror image
 * [L - bin_hi*4000 - 112000, L - bin_lo*4000 + 112000) for the reverse complement (clipped to [0, L)).  A window that
 * does not cover what the requested bins read is an error (ORCA_EINVAL), never a silent zero fill.
 * Replaces: the replicated `.cuda()` of the whole sequence per DataParallel replica (orca_predict.py:334, :675-683). */
int orca_encoder_forward_codes_window(orca_ctx* ctx, orca_net* net, const uint8_t* codes, int64_t sc_b, int64_t win_origin,
                                      int64_t win_len, int reverse, int B, int64_t L, int64_t bin_lo, int64_t bin_hi,
                                      float* out, int64_t so_b, int64_t so_c, int64_t chunk_bp);
/* The same from a 2-bit genome resident in HBM (selene_utils2.py:38-272 keeps the genome as text / expands every window to float32 [L,4]
 * on the host; orca_amd/genome.py TwoBitGenome: base j of a chromosome in bits 2 (j % 4).. of byte j / 4 of `two`, N flag in bit j % 8 of byte
 * j / 8 of `nmask`): the sequence is bases [start, start + L) of that chromosome (B = 1).  The first-layer kernels read the two planes
 * directly - the one-hot expansion happens in LDS as for the 1-byte codes - so no unpacked copy of the window exists. */
int orca_encoder_forward_2bit(orca_ctx* ctx, orca_net* net, const uint8_t* two, const uint8_t* nmask, int64_t start, int reverse, int64_t L,
                              int64_t bin_lo, int64_t bin_hi, float* out, int64_t so_c, int64_t chunk_bp);

/* ---- the Encoder in two parts: structural-variant screens at arbitrary base positions ----------------------------
 * The reference's drivers place every window at the variant's own phase (orca_predict.py:1613: wpos = coord_clip(mstart) - no rounding to
 * the 4 kb grid) and push each 32 Mb window through the whole Encoder (orca_predict.py:231).  Stages 1-3 of the Encoder - 99 % of its work -
 * are translation-covariant on a 16-BASE grid (MaxPool1d(4) twice, orca_modules.py:829, :846) with a reach of 351 bases, so their output on a
 * chromosome, computed once per strand and per phase mod 16, serves every window of every allele made of pieces of that chromosome:
 * 512 bytes per base and strand of HBM (41 GB for both strands of a 40 Mb chromosome), against 24.6 ms of Encoder per strand and window.
 *   orca_encoder_stage3_planes  stage 3's output (ReLU + residual, BEFORE MaxPool1d(5)) of the L bases `codes` (reverse != 0: of their reverse
 *                               complement) as P16 planes: 32 planes (16 channel octets x {hi, lo} fp16 parts) of plane_units =
 *                               orca_p16_plane_units(L / 16) 16-byte units each, position j of the output at unit 8 + j of a plane.  L % 80 == 0.
 *   orca_p16_pool5_into         MaxPool1d(5) (orca_modules.py:853) of positions [src_pos0, src_pos0 + 5 count) of such planes into positions
 *                               [dst_pos0, dst_pos0 + count) of the planes of a stage-4 input (planes of dst_units units)
 *   orca_encoder_front_snippet  stages 1-3 + MaxPool1d(5) on bases [base0, base0 + nbases) of the L-base sequence `codes` (an allele window; both
 *                               ends of the snippet are treated as sequence ends, as the reference's zero padding treats a window's), pooled
 *                               positions [skip, skip + count) written to positions [dst_pos0, ..) of the stage-4 input: window ends and
 *                               junctions between pieces, where the cached values do not apply
 *   orca_encoder_back           stages 4-7 from a stage-4 input of n4 positions (n4 % 50 == 0; planes of s4_units = orca_p16_plane_units(n4)
 *                               units) -> out[c * so_c + bin], 128 x n4 / 50 bins
 * ORCA_PRECISION_F16X2 nets only (the planes are the operand image of that arithmetic); the fp16-range guard applies as everywhere. */
int64_t orca_p16_plane_units(int64_t n);
int orca_encoder_stage3_planes(orca_ctx* ctx, orca_net* net, const uint8_t* codes, int64_t L, int reverse, float* planes, int64_t plane_units);
int orca_p16_pool5_into(orca_ctx* ctx, const float* src, int64_t src_units, int64_t src_pos0, float* dst, int64_t dst_units, int64_t dst_pos0, int64_t count);
int orca_encoder_front_snippet(orca_ctx* ctx, orca_net* net, const uint8_t* codes, int64_t L, int reverse, int64_t base0, int64_t nbases, int64_t skip,
                               int64_t count, float* dst, int64_t dst_units, int64_t dst_pos0);
int orca_encoder_back(orca_ctx* ctx, orca_net* net, const float* s4, int64_t s4_units, int64_t n4, float* out, int64_t so_c);
/* One level further up (what sv.Stage4Cache keeps): stage 4 - 1.2 of the back's 2.0 ms per window strand - is covariant on the 80-BASE grid of its
 * input (MaxPool1d(5) behind stage 3) with a reach of 1 631 bases; its output as fp32 rows [n][128] per strand and phase mod 80 costs the same 512 bytes
 * per base and strand, and a window strand is then a MaxPool1d(5) gather of rows, the front + stage 4 on its ends and junctions, and stages 5-7.
 *   orca_encoder_stage4_rows     stage 4 alone: a stage-4 input of n4 positions (P16 planes, as for orca_encoder_back) -> rows[n4][128] (ReLU + residual,
 *                                before the MaxPool1d(5) in front of stage 5, orca_modules.py:866-872)
 *   orca_rows_pool5_into         MaxPool1d(5) of rows [src_pos0, src_pos0 + 5 count) of src[src_rows][128] into rows [dst_pos0, ..) of dst[dst_rows][128]
 *   orca_encoder_front4_snippet  stages 1-4 on bases [base0, base0 + nbases) (multiples of 400) of the L-base sequence `codes`, MaxPool1d(5); pooled rows
 *                                [skip, skip + count) -> rows [dst_pos0, ..) of the stage-5 input dst[dst_rows][128]
 *   orca_encoder_back5           stages 5-7 from a stage-5 input of n5 rows (n5 % 10 == 0) -> out[c * so_c + bin], 128 x n5 / 10 bins */
int orca_encoder_stage4_rows(orca_ctx* ctx, orca_net* net, const float* s4, int64_t s4_units, int64_t n4, float* rows);
int orca_rows_pool5_into(orca_ctx* ctx, const float* src, int64_t src_rows, int64_t src_pos0, float* dst, int64_t dst_rows, int64_t dst_pos0, int64_t count);
int orca_encoder_front4_snippet(orca_ctx* ctx, orca_net* net, const uint8_t* codes, int64_t L, int reverse, int64_t base0, int64_t nbases, int64_t skip,
                                int64_t count, float* dst, int64_t dst_rows, int64_t dst_pos0);
/* ... and for the snippets of one window strand CONCATENATED into one L-base sequence (the window's ends first and last): ranges_host = n_ranges triples
 * (skip, count, dst_pos0) of pooled rows of that one run. */
int orca_encoder_front4_ranges(orca_ctx* ctx, orca_net* net, const uint8_t* codes, int64_t L, int reverse, int n_ranges, const int64_t* ranges_host,
                               float* dst, int64_t dst_rows);
int orca_encoder_back5(orca_ctx* ctx, orca_net* net, const float* rows, int64_t n5, float* out, int64_t so_c);

/* Number of 4 kb bins Encoder emits for an L-bp input (floor through the
 * 4,4,5,5,5,2 pooling chain). */
int64_t orca_encoder_num_bins(int64_t L);

/* Replaces: model.net(enc0) / model.net1(enc0) = Encoder2.forward
 * (orca_modules.py:1151-1169) and Encoder3.forward (:1388-1406).
 * x: [B,128,n] (strides in elements), n divisible by 2^nlev (nlev = 5 / 3).
 * An ORCA_NET_ENCODER2B net (Encoder2b.forward, :1262-1276; 5 levels) returns the contracting path's encodings.
 * outs: HOST array of nlev+1 device pointers; outs[i] receives the
 * contiguous [B,128,n>>i] encoding (fine -> coarse, as the reference returns).
 * Arithmetic: orca_net_set_precision (F32, F16X2, BF16X3, BF16X2, BF16).  The split-operand modes run on channel-last
 * activations from B*n >= 32000 positions on (the 256 Mb model's 64 000 bins); smaller problems use the exact fp32 kernels,
 * which are faster there, whatever the precision asked for. */
int orca_unet_forward(orca_ctx* ctx, orca_net* net, const float* x, int64_t sx_b, int64_t sx_c,
                      int64_t sx_l, int B, int n, float* const* outs_host, int n_outs);

/* Replaces: model.denets[level].forward(x, distenc, y) = Decoder.forward
 * (orca_modules.py:461-488).  x: [B,128,n] slice of an encoding (n<=256,
 * 250 in the reference); distenc: [B,1,n,n] log-background (sd_b may be 0
 * for the `.expand` of orca_predict.py:353); y: NULL or the [B,1,n/2,n/2]
 * crop of the coarser prediction (orca_predict.py:374-379).
 * out: contiguous [B,1,n,n].  accumulate!=0 adds into out instead of
 * overwriting (used for `+ denet_1_pt(...)`, orca_predict.py:362-366).
 * Streams: every launch of the call goes to the context's stream (a launch carries the whole batch; the call can be captured into a graph). */
int orca_decoder_forward(orca_ctx* ctx, orca_net* net, const float* x, int64_t sx_b, int64_t sx_c,
                         int64_t sx_l, const float* distenc, int64_t sd_b, int64_t sd_h, int64_t sd_w,
                         const float* y, int64_t sy_b, int64_t sy_h, int64_t sy_w, int B, int n,
                         float* out, int accumulate);

/* Multi-target decoders (orca_leukemia.py:512-990 `Decoder(num_2d)`, :996-1316 `Decoder_1m(num_2d)`; containers
 * OrcaLeukemiaA/B :1604-1873): the same network with T = num_2d maps per prediction - distenc [B,T,n,n], coarse y
 * [B,T,n/2,n/2], `final` 64 -> max(5,T) -> T, out contiguous [B,T,n,n].  T is taken from the width of the last
 * layer handed to orca_net_create (1 <= T <= 8).  Replaces: the same call site as orca_decoder_forward,
 * `model.denets[level].forward(x, distenc, y)` (orca_predict.py:356-379), for models whose normmats are 3-D. */
int orca_decoder_forward_mt(orca_ctx* ctx, orca_net* net, const float* x, int64_t sx_b, int64_t sx_c,
                            int64_t sx_l, const float* distenc, int64_t sd_b, int64_t sd_c, int64_t sd_h,
                            int64_t sd_w, const float* y, int64_t sy_b, int64_t sy_c, int64_t sy_h, int64_t sy_w,
                            int B, int n, float* out, int accumulate);

/* T (num_2d) of a Decoder / Decoder_1m net; 1 for every other kind. */
int orca_net_num_targets(orca_net* net, int* num_2d);

/* orca_decoder_forward_mt / orca_decoder1m_forward with ONE DEVICE POINTER PER BATCH ROW (host arrays of B pointers;
 * y_rows may be NULL) instead of a batch stride: the rows of a batch need not be slices of one tensor.  genomepredict
 * crops every strand's encoding (`[:, :, s:s+250]`) and coarse prediction (`[:, :, i:i+125, i:i+125]`) at the strand's
 * own offset (orca_predict.py:356-379); with row pointers the two strands go through every decoder level as one batch
 * without being copied next to each other first.  The pointer arrays are read during the call only. */
int orca_decoder_forward_rows(orca_ctx* ctx, orca_net* net, const float* const* x_rows, int64_t sx_c, int64_t sx_l,
                              const float* const* distenc_rows, int64_t sd_c, int64_t sd_h, int64_t sd_w,
                              const float* const* y_rows, int64_t sy_c, int64_t sy_h, int64_t sy_w, int B, int n,
                              float* out, int accumulate);
int orca_decoder1m_forward_rows(orca_ctx* ctx, orca_net* net, const float* const* x_rows, int64_t sx_c, int64_t sx_l,
                                int B, int n, float* out, int accumulate);

/* Replaces: model.denet_1_pt.forward(x) = Decoder_1m.forward (orca_modules.py:782-800); out [B,T,n,n]. */
int orca_decoder1m_forward(orca_ctx* ctx, orca_net* net, const float* x, int64_t sx_b, int64_t sx_c,
                           int64_t sx_l, int B, int n, float* out, int accumulate);

/* Replaces: the strand merge `0.5*fwd + 0.5*rev[::-1, ::-1]`
 * (orca_predict.py:514-523) for contiguous [n,n] maps. */
int orca_strand_merge(orca_ctx* ctx, const float* fwd, const float* rev, float* out, int n);

/* Replaces: the per-level background of genomepredict_256Mb (orca_predict.py:724-737, :692-697, :703):
 *   normmat_r = np.nanmean(np.nanmean(np.reshape(normmat[s : s + 250 nb, s : s + 250 nb], (1, 250, nb, 250, nb)), axis=4), axis=2)
 *   distenc   = torch.log(torch.FloatTensor(normmat_r))        [torch.flip(distenc, [2, 3]) on the reverse strand]
 * on an 8000 x 8000 float64 background held in HBM (row stride ld, window origin (row0, col0), nb = level // 8 entries per
 * pixel side, npix = 250).  mean_out [npix][npix] float64 is BIT-IDENTICAL to numpy's result (same pairwise / sequential
 * summation order); log_out [npix][npix] float32 = logf of its float32 rounding, written flipped in both axes when
 * `flip`.  Either output may be NULL.  Device pointers. */
int orca_block_mean_f64(orca_ctx* ctx, const double* mat, int64_t ld, int64_t row0, int64_t col0, int nb, int npix,
                        double* mean_out, float* log_out, int flip);

/* Replaces: `adaptive_coarsegrain_gpu(ar, countar, cutoff, max_levels, min_shape)` (selene_utils2.py:274-463), the
 * 2x2-pooling smoother that `Genomic2DFeatures(cg=True)` applies to OBSERVED Hi-C matrices (:551-556) before they become
 * output["experiments"].  ar = balanced matrix (NaN = masked), countar = raw counts, both [n][n] fp32 device arrays with
 * row stride ld; out [n][n] (row stride ld_out), NaN at invalid pixels.  Bit-identical to the reference (same float32
 * operation order).  The reference's defaults: cutoff 5, max_levels 8 (12 through its wrapper), min_shape 8. */
int orca_adaptive_coarsegrain(orca_ctx* ctx, const float* ar, const float* countar, int64_t ld, int n, float cutoff,
                              int max_levels, int min_shape, float* out, int64_t ld_out);

/* Genome store, 2 bits per base + N bit-mask (replaces the float32 one-hot memmap of
 * selene_utils2.MemmapGenome, selene_utils2.py:38-272): expands the window [start, start+n) of a chromosome to the
 * 1-byte base codes orca_encoder_forward_codes reads (0..3 = A,C,G,T, 4 = N).  two_bit: base i in bits 2*(i%4).. of
 * byte i/4; nmask: base i in bit i%8 of byte i/8.  Device pointers. */
int orca_genome_unpack_2bit(orca_ctx* ctx, const uint8_t* two_bit, const uint8_t* nmask, int64_t start, int64_t n, uint8_t* codes);

/* ---- multi-GPU exchange ----------------------------------
 * The path shards at ONE place: the Encoder's 4 kb bins (bin_lo / bin_hi of orca_encoder_forward; blocks are
 * independent given the 112 kb input halo, orca_modules.py:955-977), followed by ONE all-gather of the per-rank
 * [B,128,bins] slabs before the replicated Encoder2 / Encoder3 / decoder tail.  Replaces: the `nn.DataParallel`
 * wrappers that are the reference's only multi-GPU mechanism (orca_models.py:44-50) around `model.net0(x)`
 * (orca_predict.py:334, :675-683).  One process per GPU; the communicator is RCCL over xGMI.
 *
 * RCCL is resolved at run time (dlopen; an RCCL already in the process - e.g. the one PyTorch-ROCm carries - is
 * preferred), so single-GPU users need no RCCL at all.  orca_comm_unique_id is called on ONE rank; the 128 bytes are
 * handed to every rank by the host's own channel (the Python host broadcasts them through torch.distributed's
 * store) and passed to orca_comm_init_rank, which is collective over the ranks. */
typedef struct orca_comm orca_comm;
#define ORCA_COMM_ID_BYTES 128
int orca_comm_unique_id(void* id128_host);
int orca_comm_init_rank(orca_ctx* ctx, int nranks, int rank, const void* id128_host, orca_comm** out);
int orca_comm_destroy(orca_comm* comm);
/* recv[r*count .. (r+1)*count) = rank r's send[0 .. count): fp32 device buffers, enqueued on the context's stream. */
int orca_allgather(orca_ctx* ctx, orca_comm* comm, const float* send, float* recv, size_t count);

/* Replaces: the kernel-size-1 Conv1d (+ folded BatchNorm) + activation layers of `Net.final_1d`, the auxiliary
 * 1-D head of the 1 Mb model (orca_modules.py:1824-1830, :1854).
 * y[b][co][m] = act(bias[co] + sum_ci w[co][ci] * x[b][ci][m]);  w [cout][cin] and bias [cout] are DEVICE memory
 * (folded on the host, uploaded once by the caller); x/y rows have stride ldx/ldy, batches x_bs/y_bs (elements).
 * act: 0 = identity, 1 = ReLU, 2 = sigmoid. */
int orca_pointwise1d_forward(orca_ctx* ctx, const float* w_dev, const float* bias_dev, int cout, int cin, const float* x,
                             int64_t x_bs, int64_t ldx, float* y, int64_t y_bs, int64_t ldy, int B, int64_t n, int act);

/* ---- single-layer entry points (kernel unit tests; same kernels the nets use) ---
 * y = [relu](conv1d_k9(x) + b) [+ r1] [+ r2], all [B,C,n] with row stride ld
 * (elements) and batch stride bs.  w/b as in orca_conv_desc (host, folded).
 * tile: 0 = auto, else force the position-tile size (32/64/256). */
int orca_conv1d_forward(orca_ctx* ctx, const orca_conv_desc* conv, const float* x, int64_t x_bs, int64_t ldx,
                        float* y, int64_t y_bs, int64_t ldy, const float* r1, const float* r2, int B, int64_t n,
                        int relu, int tile);
/* Channel-last variant on the bf16 matrix cores with split operands (precision = ORCA_PRECISION_BF16 /
 * _BF16X2 / _BF16X3): x [B][n][cin], y and r1 [B][n][cout], all contiguous; cin % 16 == 0. */
int orca_conv1d_nlc_forward(orca_ctx* ctx, const orca_conv_desc* conv, int precision, const float* x, float* y,
                            const float* r1, int B, int64_t n, int relu);
/* The P16 / LDS-DMA conv1d of the Encoder's stages 1-3 (conv_p16.h), wrapped for tests: channel-last fp32
 * in/out (x [n][cin], r1 [n][cout], y [n][cout] or, with out_mode 1 = fused MaxPool1d(4), [n/4][cout], with out_mode 3 = fused
 * MaxPool1d(5) (128 couts: conv_p16p5.h), [n/5][cout]);
 * out_mode 0/1 round-trip through the planar split-fp16 storage, out_mode 2 writes fp32 directly.
 * conv->ksize may be 9, or 17 with weight_host [cout][cin][17] - the form the Encoder's composed linear pairs
 * (Conv-BN-Conv-BN without a nonlinearity, orca_modules.py:829-835, 846-852) run as; cin % 32 == 0 then. */
int orca_conv1d_p16_forward(orca_ctx* ctx, const orca_conv_desc* conv, const float* x, float* y, const float* r1,
                            int64_t n, int relu, int out_mode);
/* The same kernel on "B16" activations - ONE bf16 plane per 8 channels, plain bf16 operands, one MFMA product,
 * 32 input channels per step (the ORCA_PRECISION_BF16 Encoder of BASELINE config 3); cin % 32 == 0.  Inputs
 * and outputs are rounded to bf16 on the way through the planar storage (out_mode 2 writes fp32). */
int orca_conv1d_b16_forward(orca_ctx* ctx, const orca_conv_desc* conv, const float* x, float* y, const float* r1,
                            int64_t n, int relu, int out_mode);
/* y = [relu](conv2d_3x3_dilated(x) + b) [+ r]; x: contiguous [B,cin,n,n], y/r: [B,cout,n,n]. */
int orca_conv2d_forward(orca_ctx* ctx, const orca_conv_desc* conv, const float* x, float* y, const float* r,
                        int B, int n, int relu);
/* The Decoders' 16-bit conv (conv2d_m16.h; precision ORCA_PRECISION_F16X2 / _BF16 / _F16; dilation 1..8), wrapped for
 * tests: contiguous [B,cin,n,n] in, [B,cout,n,n] out / residual; the maps make a round trip through the M16 storage
 * (two fp16 planes, or one bf16 / fp16 plane). */
int orca_conv2d_m16_forward(orca_ctx* ctx, const orca_conv_desc* conv, int precision, const float* x, float* y,
                            const float* r, int B, int n, int relu);
/* y[c][m] = max_{j<k} x[c][k*m+j]  (nn.MaxPool1d(k,k)); x: [rows][ldx], y: [rows][ldy]. */
int orca_maxpool1d_forward(orca_ctx* ctx, const float* x, int64_t ldx, float* y, int64_t ldy, int64_t rows,
                           int64_t n_out, int k);

#ifdef __cplusplus
}
#endif
#endif /* ORCA_HIP_H */
