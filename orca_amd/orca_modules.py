"""Drop-in counterparts of the reference ``orca_modules`` classes
(/root/reference/orca_modules.py: Encoder :803-980, Encoder2 :984-1169,
Encoder3 :1279-1406, Decoder :16-488, Decoder_1m :491-800, Net :1409-1900).

Each class is an ``nn.Module`` whose parameter tree has exactly the reference's
``state_dict`` keys and shapes (checked against tests/golden/G0_manifest.npz),
so the published ``*.statedict`` checkpoints load unchanged.  The parameters
are only CONTAINERS: ``forward`` folds the eval-mode BatchNorms into the
convolutions on the host, uploads the weights once through the C ABI
(orca_net_create) and runs the hand-written HIP kernels.  There is no torch
arithmetic on the data path and no CPU fallback - inputs must be ROCm tensors.

Inference engine only: BatchNorm always uses running statistics and Dropout is
the identity, i.e. the reference's ``.eval()`` behaviour (orca_models.py:125-133).
"""
import os

import numpy as np
import torch
from torch import nn

from . import _lib, engine

ENCODER_CHANNELS = (64, 96, 128, 128, 128, 128, 128)
ENCODER_POOLS = (1, 4, 4, 5, 5, 5, 2)
DECODER_DILATIONS = tuple([1, 2, 4, 8, 16, 32, 64] * 4)
DECODER1M_DILATIONS = tuple([1, 2, 4, 8, 16, 32, 64] + [2, 4, 8, 16, 32, 64] * 2)


def _linear_pair(conv, bn, cin, cmid, cout, lead=None, **kw):
    """[lead,] Conv, BN, Conv, BN  (the reference's 'l' blocks: no ReLU)."""
    mods = [] if lead is None else [lead]
    mods += [conv(cin, cmid, **kw), bn(cmid), conv(cmid, cout, **kw), bn(cout)]
    return nn.Sequential(*mods)


def _relu_pair(conv, bn, cin, cmid, cout, second_bn=True, **kw):
    """Conv, BN, ReLU, Conv, [BN,] ReLU."""
    mods = [conv(cin, cmid, **kw), bn(cmid), nn.ReLU(inplace=True), conv(cmid, cout, **kw)]
    if second_bn:
        mods.append(bn(cout))
    mods.append(nn.ReLU(inplace=True))
    return nn.Sequential(*mods)


def _conv_indices(seq):
    """[(conv_idx, bn_idx or None)] of a Sequential, in order."""
    out = []
    mods = list(seq)
    for i, m in enumerate(mods):
        if isinstance(m, (nn.Conv1d, nn.Conv2d)):
            nxt = mods[i + 1] if i + 1 < len(mods) else None
            out.append((i, i + 1 if isinstance(nxt, (nn.BatchNorm1d, nn.BatchNorm2d)) else None))
    return out


class _HipModule(nn.Module):
    """Shared plumbing: lazily built, per-device engine handle."""

    _kind = None
    _upsample = _lib.ORCA_UPSAMPLE_BILINEAR

    def __init__(self):
        super().__init__()
        self._nets = {}
        self._wver = 0          # bumped with every change of the parameter containers: whoever caches this module's OUTPUTS keys on it (sv_drivers._store)

    # any change of the parameter containers invalidates the uploaded copy
    def invalidate(self):
        self._nets = {}
        self._wver = getattr(self, "_wver", 0) + 1

    def _apply(self, fn, *a, **k):
        self._nets = {}
        self._wver = getattr(self, "_wver", 0) + 1
        return super()._apply(fn, *a, **k)

    def load_state_dict(self, state_dict, *a, **k):
        self._nets = {}
        self._wver = getattr(self, "_wver", 0) + 1
        # tolerate the reference's DataParallel prefixes (orca_models.py:104-123)
        own = set(self.state_dict().keys())
        fixed = {}
        for key, v in state_dict.items():
            kk = key
            while kk not in own and kk.startswith("module."):
                kk = kk[len("module."):]
            fixed[kk] = v
        return super().load_state_dict(fixed, *a, **k)

    def _fold_sequentials(self, items):
        """items: [(prefix, nn.Sequential, dilation)] -> folded conv dicts in order."""
        sd = self.state_dict()
        convs = []
        for prefix, seq, dil in items:
            for ci, bi in _conv_indices(seq):
                convs.append(engine.fold_conv(sd, f"{prefix}.{ci}", None if bi is None else f"{prefix}.{bi}", dil))
        return convs

    def _conv_items(self):
        raise NotImplementedError

    def _net(self, device):
        ctx = engine.get_context(device)
        key = (ctx.device_index, id(ctx))
        net = self._nets.get(key)
        if net is None:
            net = engine.Net(ctx, self._kind, self._fold_sequentials(self._conv_items()), self._upsample)
            self._nets[key] = net
        return net

    @staticmethod
    def _apply_precision(net, name):
        if getattr(net, "_precision", None) != name:
            net.set_precision(name)
            net._precision = name

    def _run_guarded(self, net, fn, fallback):
        """Run fn() in self.precision; in the fp16-split mode check the device range flag and redo the
        forward in the range-safe ``fallback`` arithmetic if an activation left the fp16 range."""
        ranged = ("f16x2", "f16")          # the modes with fp16 operands: device range guard
        prec = fallback if (engine._guard["force_safe"] and self.precision in ranged) else self.precision
        self._apply_precision(net, prec)
        out = fn()
        if prec in ranged and not engine._guard["defer"] and net.ctx.take_overflow():
            import warnings
            warnings.warn(f"orca_amd.{type(self).__name__}: an activation left the fp16 range; recomputing this forward "
                          f"with precision='{fallback}' (set .precision='{fallback}' to avoid the retry)")
            self._apply_precision(net, fallback)
            out = fn()
        return out


class Encoder(_HipModule):
    """bp-resolution sequence -> 4 kb bins (orca_modules.py:803-980)."""

    _kind = _lib.ORCA_NET_ENCODER

    def __init__(self, precision=None):
        """precision: arithmetic of the Conv1d stacks (all accumulate in fp32)
          "f32"    fp32 MFMA, exact fp32 products (157 TFLOP/s class)
          "f16x2"  operands split into 2 fp16 parts, 3 MFMA products: ~2^-22 relative error, 5.3x the
                   fp32-MFMA rate; needs |activations| < 65504 - checked on the device, and the forward
                   is transparently redone in "bf16x3" if the check fires (default)
          "bf16x3" 3 bf16 parts, 6 products: fp32-class error for ANY finite fp32 input, 2.67x
          "bf16x2" / "bf16"  reduced-precision throughput modes
        Default: $ORCA_ENCODER_PRECISION or "f16x2"."""
        super().__init__()
        self.precision = precision or os.environ.get("ORCA_ENCODER_PRECISION", "f16x2")
        if self.precision not in _lib.PRECISIONS:
            raise ValueError(f"precision must be one of {sorted(_lib.PRECISIONS)}")
        prev = 4
        for i, (ch, pool) in enumerate(zip(ENCODER_CHANNELS, ENCODER_POOLS), start=1):
            lead = nn.MaxPool1d(kernel_size=pool, stride=pool) if pool > 1 else None
            setattr(self, f"lconv{i}", _linear_pair(nn.Conv1d, nn.BatchNorm1d, prev, ch, ch, lead, kernel_size=9, padding=4))
            setattr(self, f"conv{i}", _relu_pair(nn.Conv1d, nn.BatchNorm1d, ch, ch, ch, kernel_size=9, padding=4))
            prev = ch

    def _conv_items(self):
        items = []
        for i in range(1, 8):
            items += [(f"lconv{i}", getattr(self, f"lconv{i}"), 1), (f"conv{i}", getattr(self, f"conv{i}"), 1)]
        return items

    # Which of the algebraically equal forms of stage 1-3's linear groups the library runs: "default" (every composed form the weights allow),
    # "stored_residual", "lconv1_only", "two_conv" (the reference's layer sequence, orca_modules.py:811-852).  The forced forms exist for the
    # parity suite: they are what extreme weights / float input rows fall back to (include/orca_hip.h: ORCA_ENCODER_FORM_*).
    form = "default"

    def _net(self, device):
        net = super()._net(device)
        if getattr(net, "_form", "default") != self.form:
            net.set_encoder_form(self.form)
            net._form = self.form
        return net

    def forward(self, x, bin_lo=0, bin_hi=0, chunk_bp=0):
        """x: [B,4,L] float32 ROCm tensor (any strides).  Returns [B,128,L//4000].
        ``bin_lo/bin_hi`` restrict the output to a bin range (multi-GPU sharding of
        the independent sequence blocks, orca_modules.py:955-977)."""
        net = self._net(x.device)
        return self._run_guarded(net, lambda: engine.encoder_forward(net, x, bin_lo, bin_hi, chunk_bp), "bf16x3")


    def forward_codes(self, codes, reverse=False, bin_lo=0, bin_hi=0, chunk_bp=0, out=None):
        """Encoder straight from packed bases: ``codes`` [B,L] uint8 on the MI355X (0..3 = A,C,G,T, 4 = N, see
        engine.pack_sequence).  ``reverse=True`` encodes the reverse complement of the same buffer."""
        if (self.precision == "bf16" and isinstance(codes, torch.Tensor) and codes.is_cuda and codes.dim() == 2 and codes.shape[0] >= 2
                and codes.shape[1] <= 64_000_000 and engine.batch_streams() and not engine.strand_streams() and not engine.in_pool_run()):
            # throughput mode, a batch: two halves on two contexts (engine.batch_streams) - each row is computed exactly as alone
            B, L = codes.shape
            hi = engine.encoder_num_bins(L) if bin_hi <= 0 else bin_hi
            if out is None:
                out = torch.empty((B, 128, hi - bin_lo), dtype=torch.float32, device=codes.device)
            h = (B + 1) // 2
            pool = engine.context_pool(codes.device, 1)
            pool.fork()
            try:
                pool.run(0, lambda: self.forward_codes(codes[h:], reverse, bin_lo, bin_hi, chunk_bp, out[h:]))
                net = self._net(codes.device)
                self._run_guarded(net, lambda: engine.encoder_forward_codes(net, codes[:h], reverse, bin_lo, bin_hi, chunk_bp, out[:h]), "bf16x3")
            finally:
                pool.join()          # whatever happened: the caller's stream continues behind what was issued on the pool's
            return out
        net = self._net(codes.device)
        return self._run_guarded(net, lambda: engine.encoder_forward_codes(net, codes, reverse, bin_lo, bin_hi, chunk_bp, out), "bf16x3")

    forward_codes.__doc__ += "  (Replaces the float [B,4,L] input of orca_predict.py:334.)"

    # ---- the Encoder in two parts (sv.Stage3Cache; include/orca_hip.h: orca_encoder_stage3_planes ...).  f16x2 arithmetic only: the planes are
    # its operand image.  No retry inside: the caller's chain carries the fp16-range check (engine.run_with_overflow_retry, sv.sv_screen) and
    # takes the whole-window route under engine.force_safe_precision() ----
    def two_part_ok(self):
        return self.precision == "f16x2" and self.form == "default" and not engine._guard["force_safe"]

    def _parts_net(self, device):
        if not self.two_part_ok():
            raise RuntimeError("the Encoder's two-part form (stage-3 cache) exists in the default f16x2 arithmetic only")
        net = self._net(device)
        self._apply_precision(net, "f16x2")
        return net

    def stage3_planes(self, codes, reverse=False):
        """Stage 3's output of ``codes`` [L] uint8 (before MaxPool1d(5); orca_modules.py:846-852) as P16 planes [32, units, 4]; None (and a
        warning) when an activation left the fp16 range and the check is immediate."""
        net = self._parts_net(codes.device)
        planes = engine.encoder_stage3_planes(net, codes, reverse)
        if not engine._guard["defer"] and net.ctx.take_overflow():
            import warnings
            warnings.warn("orca_amd.Encoder: an activation left the fp16 range while building a stage-3 cache entry; windows will be encoded whole")
            return None
        return planes

    def front_snippet(self, codes, reverse, base0, nbases, skip, count, dst, dst_pos0):
        engine.encoder_front_snippet(self._parts_net(codes.device), codes, reverse, base0, nbases, skip, count, dst, dst_pos0)

    def back(self, s4, n4, out):
        return engine.encoder_back(self._parts_net(s4.device), s4, n4, out)

    def stage4_rows(self, s4, n4):
        return engine.encoder_stage4_rows(self._parts_net(s4.device), s4, n4)

    def front4_snippet(self, codes, reverse, base0, nbases, skip, count, dst, dst_pos0):
        engine.encoder_front4_snippet(self._parts_net(codes.device), codes, reverse, base0, nbases, skip, count, dst, dst_pos0)

    def front4_ranges(self, codes, reverse, ranges, dst):
        engine.encoder_front4_ranges(self._parts_net(codes.device), codes, reverse, ranges, dst)

    def back5(self, rows, out):
        return engine.encoder_back5(self._parts_net(rows.device), rows, out)

    def forward_2bit(self, genome, chrom, start, end, reverse=False, bin_lo=0, bin_hi=0, out=None):
        """Encoder on `chrom`[start:end) of a genome.TwoBitGenome resident on the MI355X, read in place: 2 bits per base + N mask straight
        into the first-layer kernels (one-hot expansion in LDS) - no unpacked 1 byte/base window, no float window (selene_utils2.py:216-222)."""
        two, nmask = genome.planes(chrom)
        net = self._net(two.device)
        return self._run_guarded(net, lambda: engine.encoder_forward_2bit(net, two, nmask, start, end - start, reverse, bin_lo, bin_hi, 0, out), "bf16x3")


def _unet_precision(precision):
    p = precision or os.environ.get("ORCA_ENCODER2_PRECISION", "f16x2")
    if p not in ("f32", "f16x2", "bf16x3", "bf16x2", "bf16"):
        raise ValueError("precision must be one of ['bf16', 'bf16x2', 'bf16x3', 'f16x2', 'f32']")
    return p


class _UNetEncoder(_HipModule):
    _nlev = 0

    def __init__(self, precision=None):
        """precision: as Encoder's ("f16x2" default - split fp16 operands under the device range guard, fp32-class results; "f32" = the
        exact fp32-MFMA kernels).  Default: $ORCA_ENCODER2_PRECISION or "f16x2"."""
        super().__init__()
        self.precision = _unet_precision(precision)
        n = self._nlev
        c1 = dict(kernel_size=9, padding=4)
        self.lblocks = nn.ModuleList(
            [_linear_pair(nn.Conv1d, nn.BatchNorm1d, 128, 128, 128, nn.MaxPool1d(kernel_size=2, stride=2), **c1) for _ in range(n)])
        self.blocks = nn.ModuleList([_relu_pair(nn.Conv1d, nn.BatchNorm1d, 128, 128, 128, **c1) for _ in range(n)])
        self.downlblocks = nn.ModuleList(
            [_linear_pair(nn.Conv1d, nn.BatchNorm1d, 128, 128, 128, nn.Upsample(scale_factor=2), **c1) for _ in range(n)])
        self.downblocks = nn.ModuleList(
            [_relu_pair(nn.Conv1d, nn.BatchNorm1d, 128, 128, 128, second_bn=False, **c1) for _ in range(n)])

    def _conv_items(self):
        items = []
        for i in range(self._nlev):
            items += [(f"lblocks.{i}", self.lblocks[i], 1), (f"blocks.{i}", self.blocks[i], 1)]
        for i in range(self._nlev):
            items += [(f"downlblocks.{i}", self.downlblocks[i], 1), (f"downblocks.{i}", self.downblocks[i], 1)]
        return items

    def forward(self, x):
        """x: [B,128,n] -> list of nlev+1 encodings [B,128,n>>i], fine to coarse."""
        net = self._net(x.device)
        return self._run_guarded(net, lambda: engine.unet_forward(net, x, self._nlev), "bf16x3")


class Encoder2(_UNetEncoder):
    """4 kb -> 128 kb U-shaped encoder (orca_modules.py:984-1169)."""
    _kind = _lib.ORCA_NET_ENCODER2
    _nlev = 5


class Encoder2b(_HipModule):
    """Encoder2 without the expanding path (orca_modules.py:1173-1276; the HCTnoc model): returns the six encodings
    of the contracting path, fine to coarse."""
    _kind = _lib.ORCA_NET_ENCODER2B
    _nlev = 5

    def __init__(self, precision=None):
        super().__init__()
        self.precision = _unet_precision(precision)
        c1 = dict(kernel_size=9, padding=4)
        self.lblocks = nn.ModuleList(
            [_linear_pair(nn.Conv1d, nn.BatchNorm1d, 128, 128, 128, nn.MaxPool1d(kernel_size=2, stride=2), **c1) for _ in range(5)])
        self.blocks = nn.ModuleList([_relu_pair(nn.Conv1d, nn.BatchNorm1d, 128, 128, 128, **c1) for _ in range(5)])

    def _conv_items(self):
        items = []
        for i in range(5):
            items += [(f"lblocks.{i}", self.lblocks[i], 1), (f"blocks.{i}", self.blocks[i], 1)]
        return items

    def forward(self, x):
        net = self._net(x.device)
        return self._run_guarded(net, lambda: engine.unet_forward(net, x, 5), "bf16x3")


class Encoder3(_UNetEncoder):
    """128 kb -> 1024 kb U-shaped encoder (orca_modules.py:1279-1406)."""
    _kind = _lib.ORCA_NET_ENCODER3
    _nlev = 3


def _c2(d):
    return dict(kernel_size=(3, 3), padding=d, dilation=d)


MAX_TARGETS = 8   # ORCA_MAX_TARGETS of the HIP library


def _final_head(num_2d=1):
    """64 -> max(5, num_2d) -> num_2d (orca_modules.py:423-428; orca_leukemia.py:922-927 for num_2d > 1)."""
    hidden = num_2d if num_2d > 5 else 5
    return nn.Sequential(nn.Conv2d(64, hidden, kernel_size=(1, 1), padding=0), nn.BatchNorm2d(hidden), nn.ReLU(inplace=True),
                         nn.Conv2d(hidden, num_2d, kernel_size=(1, 1), padding=0))


def _check_num_2d(num_2d):
    if not (isinstance(num_2d, int) and 1 <= num_2d <= MAX_TARGETS):
        raise ValueError(f"num_2d must be an int in 1..{MAX_TARGETS}, got {num_2d!r}")
    return num_2d


class Decoder(_HipModule):
    """1-D encoding -> 2-D log-fold contact map (orca_modules.py:16-488)."""

    _kind = _lib.ORCA_NET_DECODER

    def __init__(self, upsample_mode="nearest", precision=None, num_2d=1):
        """precision: "f16x2" (dilated 3x3 convs on the fp16 matrix cores with 2-way split fp32 operands,
        ~2^-22 relative error, device range guard with automatic "f32" retry; default), "f32" (fp32 MFMA) or
        "bf16" / "f16" (ONE bf16 / fp16 plane per feature map, one product, fp32 accumulate: the throughput modes - "bf16" is
        BASELINE config 3 as named, "f16" the same rate with 11 instead of 8 significant bits and the fp16 range guard).
        Default: $ORCA_DECODER_PRECISION or "f16x2".
        num_2d: maps per prediction (the multi-target decoders of orca_leukemia.py:512-990): distenc and the
        coarse prediction y then carry num_2d channels, and so does the output."""
        super().__init__()
        self.num_2d = _check_num_2d(num_2d)
        self.precision = precision or os.environ.get("ORCA_DECODER_PRECISION", "f16x2")
        if self.precision not in ("f16x2", "f32", "bf16", "f16"):
            raise ValueError("Decoder precision must be 'f16x2', 'f32', 'bf16' or 'f16'")
        if upsample_mode not in ("nearest", "bilinear"):
            raise ValueError("upsample_mode must be 'nearest' or 'bilinear'")
        self._upsample = _lib.ORCA_UPSAMPLE_BILINEAR if upsample_mode == "bilinear" else _lib.ORCA_UPSAMPLE_NEAREST
        self.lconvtwos = nn.ModuleList([
            _linear_pair(nn.Conv2d, nn.BatchNorm2d, 64, 32, 64, nn.Dropout(p=0.1) if i == 0 else None, **_c2(d))
            for i, d in enumerate(DECODER_DILATIONS)])
        self.convtwos = nn.ModuleList([_relu_pair(nn.Conv2d, nn.BatchNorm2d, 64, 32, 64, **_c2(d)) for d in DECODER_DILATIONS])
        self.final = _final_head(num_2d)
        self.upsample = nn.Upsample(scale_factor=(2, 2), mode=upsample_mode)
        self.lcombiner = _linear_pair(nn.Conv2d, nn.BatchNorm2d, 64 + num_2d, 64, 64, nn.Dropout(p=0.1), **_c2(1))
        self.combiner = _relu_pair(nn.Conv2d, nn.BatchNorm2d, 64, 64, 64, **_c2(1))
        self.lcombinerD = _linear_pair(nn.Conv2d, nn.BatchNorm2d, 128 + num_2d, 64, 64, **_c2(1))
        self.combinerD = _relu_pair(nn.Conv2d, nn.BatchNorm2d, 64, 64, 64, **_c2(1))

    def _conv_items(self):
        items = [("lcombinerD", self.lcombinerD, 1), ("combinerD", self.combinerD, 1),
                 ("lcombiner", self.lcombiner, 1), ("combiner", self.combiner, 1)]
        for i, d in enumerate(DECODER_DILATIONS):
            items += [(f"lconvtwos.{i}", self.lconvtwos[i], d), (f"convtwos.{i}", self.convtwos[i], d)]
        items.append(("final", self.final, 1))
        return items

    def forward(self, x, distenc, y=None):
        """x [B,128,n], distenc [B,num_2d,n,n] (log background), y None or [B,num_2d,n/2,n/2] -> [B,num_2d,n,n]."""
        net = self._net(x.device)
        return self._run_guarded(net, lambda: engine.decoder_forward(net, x, distenc, y), "f32")

    def forward_into(self, out, x, distenc, y=None, accumulate=False):
        net = self._net(x.device)
        if accumulate and self.precision in ("f16x2", "f16"):
            base = out.clone()   # a retry must not accumulate twice
            return self._run_guarded(net, lambda: engine.decoder_forward(net, x, distenc, y, out=out.copy_(base), accumulate=True), "f32")
        return self._run_guarded(net, lambda: engine.decoder_forward(net, x, distenc, y, out=out, accumulate=accumulate), "f32")

    def forward_rows(self, xs, distencs, ys=None):
        """forward() on a batch given row by row (lists of views [128,n] / [T,n,n] / [T,n/2,n/2]): no stacking copies."""
        net = self._net(xs[0].device)
        return self._run_guarded(net, lambda: engine.decoder_forward_rows(net, xs, distencs, ys), "f32")


class Decoder_1m(_HipModule):
    """Decoder of the 1 Mb module (orca_modules.py:491-800)."""

    _kind = _lib.ORCA_NET_DECODER_1M

    def __init__(self, precision=None, num_2d=1):
        super().__init__()
        self.num_2d = _check_num_2d(num_2d)
        self.precision = precision or os.environ.get("ORCA_DECODER_PRECISION", "f16x2")
        if self.precision not in ("f16x2", "f32", "bf16", "f16"):
            raise ValueError("Decoder_1m precision must be 'f16x2', 'f32', 'bf16' or 'f16'")
        self.lconvtwos = nn.ModuleList([
            _linear_pair(nn.Conv2d, nn.BatchNorm2d, 128 if i == 0 else 64, 32, 64, nn.Dropout(p=0.1) if i == 0 else None, **_c2(d))
            for i, d in enumerate(DECODER1M_DILATIONS)])
        self.convtwos = nn.ModuleList([_relu_pair(nn.Conv2d, nn.BatchNorm2d, 64, 32, 64, **_c2(d)) for d in DECODER1M_DILATIONS])
        self.final = _final_head(num_2d)

    def _conv_items(self):
        items = []
        for i, d in enumerate(DECODER1M_DILATIONS):
            items += [(f"lconvtwos.{i}", self.lconvtwos[i], d), (f"convtwos.{i}", self.convtwos[i], d)]
        items.append(("final", self.final, 1))
        return items

    def forward(self, x):
        net = self._net(x.device)
        return self._run_guarded(net, lambda: engine.decoder1m_forward(net, x), "f32")

    def forward_into(self, out, x, accumulate=False):
        net = self._net(x.device)
        if accumulate and self.precision in ("f16x2", "f16"):
            base = out.clone()   # a retry must not accumulate twice
            return self._run_guarded(net, lambda: engine.decoder1m_forward(net, x, out=out.copy_(base), accumulate=True), "f32")
        return self._run_guarded(net, lambda: engine.decoder1m_forward(net, x, out=out, accumulate=accumulate), "f32")

    def forward_rows_into(self, out, xs, accumulate=False):
        net = self._net(xs[0].device)
        if accumulate and self.precision in ("f16x2", "f16"):
            base = out.clone()
            return self._run_guarded(net, lambda: engine.decoder1m_forward_rows(net, xs, out=out.copy_(base), accumulate=True), "f32")
        return self._run_guarded(net, lambda: engine.decoder1m_forward_rows(net, xs, out=out, accumulate=accumulate), "f32")


class Net(nn.Module):
    """The 1 Mb Orca model (orca_modules.py:1409-1900): the Encoder's seven stages on a [B,4,1000000] sequence,
    the 19-pair 2-D head of Decoder_1m on the pairwise sum of the resulting [B,128,250] encoding and - with
    ``num_1d`` - the auxiliary 1-D head ``final_1d`` (two kernel-size-1 convolutions, sigmoid).

    The state dict is the reference's (Encoder keys + Decoder_1m keys + ``final_1d.*``, checked in
    tests/golden/G0_manifest.npz), so ``orca_<cell>.net0.statedict`` loads unchanged.  The parameter containers are
    SHARED with an internal Encoder and Decoder_1m, which own the two device-side engine nets; ``forward`` is
    ``Decoder_1m(Encoder(x))`` on the HIP kernels (one Encoder chunk: the reference runs the stack over the whole
    1 Mb at once, `run0` :1836-1857) plus ``orca_pointwise1d_forward`` for the 1-D head."""

    def __init__(self, num_1d=None, precision=None, num_2d=1):
        super().__init__()
        enc, dec = Encoder(precision), Decoder_1m(num_2d=num_2d)
        self.num_2d = num_2d
        for name, child in list(enc.named_children()) + list(dec.named_children()):
            self.add_module(name, child)
        if num_1d is not None:
            self.final_1d = nn.Sequential(nn.Conv1d(128, 128, kernel_size=1, padding=0), nn.BatchNorm1d(128), nn.ReLU(inplace=True),
                                          nn.Conv1d(128, num_1d, kernel_size=1, padding=0), nn.Sigmoid())
        self.num_1d = num_1d
        # not registered as sub-modules: they only borrow the containers above
        object.__setattr__(self, "_enc", enc)
        object.__setattr__(self, "_dec", dec)
        self._head1d = {}

    @property
    def precision(self):
        return self._enc.precision

    @precision.setter
    def precision(self, name):
        self._enc.precision = name

    def _invalidate(self):
        self._enc.invalidate()
        self._dec.invalidate()
        self._head1d = {}

    def _apply(self, fn, *a, **k):
        self._invalidate()
        return super()._apply(fn, *a, **k)

    def load_state_dict(self, state_dict, *a, **k):
        self._invalidate()
        own = set(self.state_dict().keys())
        fixed = {}
        for key, v in state_dict.items():
            kk = key
            while kk not in own and kk.startswith("module."):
                kk = kk[len("module."):]
            fixed[kk] = v
        return super().load_state_dict(fixed, *a, **k)

    def _head1d_weights(self, device):
        w = self._head1d.get(device)
        if w is None:
            sd = self.state_dict()
            a = engine.fold_conv(sd, "final_1d.0", "final_1d.1")
            b = engine.fold_conv(sd, "final_1d.3")
            w = tuple(torch.from_numpy(np.ascontiguousarray(t)).to(device) for t in
                      (a["w"][:, :, 0], a["b"], b["w"][:, :, 0], b["b"]))
            self._head1d[device] = w
        return w

    def forward(self, x):
        """x: [B,4,L] float32 ROCm tensor, L = 1000000 in the reference (any L whose 4 kb-bin count is <= 256).
        Returns the [B,1,n,n] map, or (map, [B,num_1d,n]) with the auxiliary head."""
        enc = self._enc(x)
        cur = self._dec(enc)
        if self.num_1d:
            w1, b1, w2, b2 = self._head1d_weights(x.device)
            return cur, engine.pointwise1d(engine.pointwise1d(enc, w1, b1, "relu"), w2, b2, "sigmoid")
        return cur
