"""Thin host layer over the C ABI: contexts, BatchNorm folding, net handles and
forward wrappers that take torch ROCm tensors purely as device containers
(``data_ptr()`` + strides).  No arithmetic on the data path happens in torch.
"""
import ctypes
import threading
import weakref

import numpy as np
import torch

from . import _lib
from ._lib import ConvDesc, OrcaHipError, check

BN_EPS = 1e-5  # nn.BatchNorm1d/2d default used throughout orca_modules.py


# ---------------------------------------------------------------------------
# contexts: one per (host thread, device)
# ---------------------------------------------------------------------------
_tls = threading.local()


class Context:
    def __init__(self, device_index):
        lib = _lib.load()
        self.device_index = device_index
        self.handle = ctypes.c_void_p()
        stream = torch.cuda.current_stream(device_index).cuda_stream
        check(lib.orca_ctx_create(device_index, ctypes.c_void_p(stream), ctypes.byref(self.handle)), "orca_ctx_create")
        self._finalizer = weakref.finalize(self, lib.orca_ctx_destroy, self.handle)

    def sync_stream(self):
        """Point the context at torch's current stream for this device."""
        stream = torch.cuda.current_stream(self.device_index).cuda_stream
        check(_lib.load().orca_ctx_set_stream(self.handle, ctypes.c_void_p(stream)), "orca_ctx_set_stream")

    def workspace_bytes(self):
        n = ctypes.c_size_t()
        check(_lib.load().orca_ctx_workspace_bytes(self.handle, ctypes.byref(n)))
        return n.value

    def take_overflow(self):
        """f16x2 mode: True if an activation left the fp16 range since the last call (syncs the stream)."""
        f = ctypes.c_int()
        check(_lib.load().orca_ctx_take_overflow(self.handle, ctypes.byref(f)), "orca_ctx_take_overflow")
        return bool(f.value)

    def set_timing(self, enable):
        check(_lib.load().orca_ctx_set_timing(self.handle, 1 if enable else 0))

    def get_timing(self, max_records=4096):
        """[(cout, cin, tile, batch, n, ms, ksize)] of the conv1d launches timed since the last call."""
        buf = (_lib.KernelTime * max_records)()
        n = ctypes.c_int()
        check(_lib.load().orca_ctx_get_timing(self.handle, buf, max_records, ctypes.byref(n)))
        return [(r.cout, r.cin, r.tile, r.batch, r.n, r.ms, r.ksize) for r in buf[: min(n.value, max_records)]]

    def launch_counts(self):
        """{"conv_small": n, "conv_bf16s": n, "planar": n, "decoder_pairs": n} launches on this context since it was created."""
        c = (ctypes.c_int64 * 4)()
        check(_lib.load().orca_ctx_launch_counts(self.handle, c), "orca_ctx_launch_counts")
        return dict(zip(("conv_small", "conv_bf16s", "planar", "decoder_pairs"), [int(x) for x in c]))

    def release_workspace(self):
        check(_lib.load().orca_ctx_release_workspace(self.handle))


def get_context(device):
    if isinstance(device, torch.device):
        if device.type != "cuda":
            raise OrcaHipError(f"orca_amd runs on MI355X only; got a tensor on '{device}'. There is no CPU path.")
        idx = device.index if device.index is not None else torch.cuda.current_device()
    else:
        idx = int(device)
    ctx = getattr(_tls, "override", None)          # ContextPool.run: an auxiliary context is current on this thread
    if ctx is None or ctx.device_index != idx:
        ctxs = getattr(_tls, "ctxs", None)
        if ctxs is None:
            ctxs = _tls.ctxs = {}
        if idx not in ctxs:
            ctxs[idx] = Context(idx)
        ctx = ctxs[idx]
    ctx.sync_stream()
    return ctx


# ---------------------------------------------------------------------------
# fp16-range guard policy (see orca_modules._HipModule._run_guarded)
#   immediate (default): every module forward in "f16x2" checks the device flag right away (one stream sync)
#   deferred: a caller that runs a whole cascade checks ONCE at the end and reruns it under force_safe()
# ---------------------------------------------------------------------------
import contextlib


class _GuardState(threading.local):
    """Per-THREAD policy (contexts, streams and overflow flags are per thread too, see _tls): one thread's deferred scope
    must not switch off another thread's immediate checks."""
    defer = False
    force_safe = False

    def __getitem__(self, key):      # dict-style access kept for the modules
        return getattr(self, key)

    def __setitem__(self, key, value):
        setattr(self, key, value)


_guard = _GuardState()


@contextlib.contextmanager
def _guard_scope(key, value):
    old = _guard[key]
    _guard[key] = value
    try:
        yield
    finally:
        _guard[key] = old


def defer_overflow_guard():
    return _guard_scope("defer", True)


def immediate_overflow_guard():
    """Inside a deferred scope: check (and retry) right away again.  Used around rank-LOCAL work that is followed by a
    collective (dist.ShardedEncoder): a retry decided after the collective would re-enter it on one rank only."""
    return _guard_scope("defer", False)


def force_safe_precision():
    return _guard_scope("force_safe", True)


class ContextPool:
    """K auxiliary contexts on one device, each with its own HIP stream, workspace arena, edge-fix scratch, fp16-range flag and weight
    upload (a context's arena is reused by every call, so calls that are to overlap need a context each) - for INDEPENDENT short calls
    whose launches fill a fraction of the chip each: the local re-encodes of an SV allele window (sv.encode_windows: 14 calls of ~50
    dependent launches, 0.65 ms of GPU time each in stream order against 0.15 ms of host time to issue one).  All issued from the
    CALLER's thread: run(i, fn) makes context i the thread's current context (get_context) and its stream torch's current stream while
    fn() runs.  fork() orders the pool's streams behind a stream of the caller's, host_join() marks the end of what has been issued,
    wait_join() orders a stream behind those marks (join() = both, now); `side` is one more stream of the caller's for preparation
    work.  (Worker THREADS per context were measured too: 42.0 against 40.6 ms per variant - the host is not the bound.)"""

    def __init__(self, device, k):
        self.index = device.index if isinstance(device, torch.device) and device.index is not None else torch.cuda.current_device()
        self.streams = [torch.cuda.Stream(device=self.index) for _ in range(k)]
        self.ctxs = []
        for s in self.streams:
            with torch.cuda.stream(s):
                self.ctxs.append(Context(self.index))
        self._fork = torch.cuda.Event()
        self._join = [torch.cuda.Event() for _ in range(k)]
        self.side = torch.cuda.Stream(device=self.index)     # a stream beside the caller's current one (preparation work)
        self._dirty = set()                                   # contexts that ran something since their range flag was last read

    def __len__(self):
        return len(self.ctxs)

    def fork(self, stream=None):
        """The pool's streams continue behind ``stream`` (default: the caller's current stream)."""
        self._fork.record(stream or torch.cuda.current_stream(self.index))
        for s in self.streams:
            s.wait_event(self._fork)

    def run(self, i, fn):
        i %= len(self.ctxs)
        self._dirty.add(i)
        prev = getattr(_tls, "override", None)
        _tls.override = self.ctxs[i]
        try:
            with torch.cuda.stream(self.streams[i]):
                return fn()
        finally:
            _tls.override = prev

    def host_join(self):
        """Mark the end of the work issued so far on the pool's streams (wait_join orders a stream behind those marks - now or later,
        e.g. after more work has been issued on it: sv.sv_screen issues a variant's local encodes under the previous variant's decoders)."""
        for s, ev in zip(self.streams, self._join):
            ev.record(s)

    def wait_join(self, stream=None):
        cur = stream or torch.cuda.current_stream(self.index)
        for ev in self._join:
            cur.wait_event(ev)

    def join(self):
        self.host_join()
        self.wait_join()

    def release_workspaces(self):
        """Free the contexts' workspace arenas (they grow again on demand): a whole-strand Encoder call leaves 25 GB in its context."""
        for c in self.ctxs:
            c.release_workspace()

    def take_overflow(self):
        """True if an activation left the fp16 range in a call run on one of the pool's contexts since the last check (reads - stream
        sync - and clears the flags of the contexts used since then)."""
        dirty, self._dirty = sorted(self._dirty), set()
        return any([self.ctxs[i].take_overflow() for i in dirty])      # a list: every flag is read (and cleared)


def _thread_pools():
    pools = getattr(_tls, "pools", None)
    if pools is None:
        pools = _tls.pools = {}
    return pools


def context_pool(device, k):
    """THE pool of k auxiliary contexts of a device FOR THE CALLING THREAD (made once per thread: modules cache a weight upload per
    context they ran on).  Per thread like every other context (`_tls.ctxs`): a pool's workspace arenas, edge scratch, streams, `_dirty`
    set and fork / join events belong to one chain of calls; two threads sharing them would overwrite each other's workspace and drain
    each other's fp16-range flags."""
    idx = device.index if isinstance(device, torch.device) and device.index is not None else torch.cuda.current_device()
    pools = _thread_pools()
    if (idx, k) not in pools:
        pools[(idx, k)] = ContextPool(torch.device("cuda", idx), k)
    return pools[(idx, k)]


def strand_streams():
    """$ORCA_STRAND_STREAMS=1: `genomepredict` / `cascade_32m` encode the reverse strand on an auxiliary context while the forward strand
    runs on the caller's stream.  The two Encoders' launches then share the chip (two persistent kernels side by side, each about twice as
    long): a first layer bound by HBM writes runs beside a conv bound by the matrix pipe, the small launches of stages 5-7 beside full ones -
    63.8 against 65.4 ms per step (same box, alternating).  Opt-in: a kernel's duration no longer says what the kernel can do (the bench's
    per-kernel roofline is defined on launches that have the chip to themselves), and the second context holds its own 25 GB workspace."""
    import os
    return os.environ.get("ORCA_STRAND_STREAMS", "0") == "1"


def batch_streams():
    """The bf16-plane Encoder (config 3's throughput mode) runs a batch of >= 2 sequences as two halves on two contexts / HIP streams
    (`orca_modules.Encoder.forward_codes`): stage 1's HBM-bound launch of one half beside the matrix-bound convs of the other - 74.7 -> 71.8 ms
    per 8 strands (tools/scratch measurement, late round 5), same bits.  $ORCA_BATCH_STREAMS=0: one context (a kernel's HIP-event time then
    measures the kernel, not its share of the chip - bench.py's config-3 roofline pass).  The fp32-class default arithmetic does not gain
    (197.3 -> 197.6 ms) and stays on one context."""
    import os
    return os.environ.get("ORCA_BATCH_STREAMS", "1") != "0"


def in_pool_run():
    """True while the calling thread is inside ContextPool.run (its current context is one of a pool's)."""
    return getattr(_tls, "override", None) is not None


def tentative(undo):
    """A result computed in the fp16-split arithmetic was just put where LATER calls will find it (a cached Encoder output, a
    chromosome encoding of the SV drivers' store) while the range check of the pass that produced it is still pending
    (`run_with_overflow_retry`: one check at the end of the chain).  ``undo()`` takes it out again; it runs if and only if that
    check fires, before the range-safe retry.  Outside such a pass (immediate checks: the module's own retry has already
    happened; the retry itself: range-safe arithmetic) nothing is registered."""
    lst = getattr(_tls, "tentative", None)
    if lst is not None and _guard.defer and not _guard.force_safe:
        lst.append(undo)


def run_with_overflow_retry(fn, device, pool=None):
    """Run fn() (a chain of module forwards on ``device``) with ONE fp16-range check at the end instead of one
    per module; if an activation left the fp16 range anywhere, redo the whole chain in the range-safe arithmetic.
    ``pool``: a ContextPool fn() also launched on (its contexts carry their own range flags).
    Whatever fn() cached for later calls on the way (`tentative`) is dropped first when the check fires."""
    if not (isinstance(device, torch.device) and device.type == "cuda"):
        return fn()
    ctx = get_context(device)
    outer, _tls.tentative = getattr(_tls, "tentative", None), []
    try:
        with defer_overflow_guard():
            out = fn()
        over = ctx.take_overflow()
        for p in ([pool] if pool is not None else []) + [q for q in _thread_pools().values() if q is not pool and q.index == ctx.device_index]:
            over = p.take_overflow() or over          # THIS thread's auxiliary contexts fn() ran on (e.g. the reverse strand's Encoder, strand_streams())
        if over:
            import warnings
            warnings.warn("orca_amd: an activation left the fp16 range; recomputing with range-safe arithmetic (bf16x3 / f32)")
            for undo in _tls.tentative:
                undo()
            _tls.tentative = []
            with force_safe_precision():
                out = fn()
    finally:
        _tls.tentative = outer
    return out


# ---------------------------------------------------------------------------
# BatchNorm folding (host, float64) - checkpoint format of orca_models.py:53-123
# ---------------------------------------------------------------------------
def _np64(v):
    if isinstance(v, torch.Tensor):
        return v.detach().cpu().double().numpy()
    return np.asarray(v, dtype=np.float64)


def fold_conv(sd, conv, bn=None, dilation=1):
    """(weight, bias[, BN running stats]) -> folded fp32 (w, b) + shape info.
    Eval-mode BN is the per-channel affine y = (x-mean)/sqrt(var+eps)*gamma+beta."""
    w = _np64(sd[conv + ".weight"])
    b = _np64(sd[conv + ".bias"])
    if bn is not None:
        s = _np64(sd[bn + ".weight"]) / np.sqrt(_np64(sd[bn + ".running_var"]) + BN_EPS)
        w = w * s.reshape((-1,) + (1,) * (w.ndim - 1))
        b = (b - _np64(sd[bn + ".running_mean"])) * s + _np64(sd[bn + ".bias"])
    return {"w": np.ascontiguousarray(w, dtype=np.float32), "b": np.ascontiguousarray(b, dtype=np.float32),
            "cout": w.shape[0], "cin": w.shape[1], "k": w.shape[2], "dil": int(dilation)}


def make_descs(convs):
    arr = (ConvDesc * len(convs))()
    for i, c in enumerate(convs):
        arr[i].weight_host = c["w"].ctypes.data
        arr[i].bias_host = c["b"].ctypes.data
        arr[i].cout, arr[i].cin, arr[i].ksize, arr[i].dilation = c["cout"], c["cin"], c["k"], c["dil"]
    return arr


class Net:
    """Device-resident weights of one reference module (orca_net)."""

    def __init__(self, ctx, kind, convs, upsample_mode=_lib.ORCA_UPSAMPLE_BILINEAR):
        lib = _lib.load()
        self.ctx, self.kind = ctx, kind
        self.handle = ctypes.c_void_p()
        descs = make_descs(convs)  # `convs` keeps the numpy arrays alive during the call
        check(lib.orca_net_create(ctx.handle, kind, descs, len(convs), upsample_mode, ctypes.byref(self.handle)), "orca_net_create")
        self._finalizer = weakref.finalize(self, lib.orca_net_free, self.handle)

    def num_targets(self):
        """Maps per prediction of a Decoder / Decoder_1m net (num_2d); 1 for the other kinds."""
        if getattr(self, "_num_targets", None) is None:
            t = ctypes.c_int(0)
            check(_lib.load().orca_net_num_targets(self.handle, ctypes.byref(t)), "orca_net_num_targets")
            self._num_targets = int(t.value)
        return self._num_targets

    def set_precision(self, name):
        if name not in _lib.PRECISIONS:
            raise ValueError(f"precision must be one of {sorted(_lib.PRECISIONS)}, got {name!r}")
        check(_lib.load().orca_net_set_precision(self.handle, _lib.PRECISIONS[name]), "orca_net_set_precision")

    def set_encoder_form(self, name):
        """Encoder nets: force a less composed form of stage 1-3's linear groups (include/orca_hip.h: ORCA_ENCODER_FORM_*; the parity suite's handle)."""
        if name not in _lib.ENCODER_FORMS:
            raise ValueError(f"form must be one of {sorted(_lib.ENCODER_FORMS)}, got {name!r}")
        check(_lib.load().orca_net_set_encoder_form(self.handle, _lib.ENCODER_FORMS[name]), "orca_net_set_encoder_form")


# ---------------------------------------------------------------------------
# forward wrappers
# ---------------------------------------------------------------------------
def _f32_cuda(t, name):
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name} must be a torch.Tensor")
    if not t.is_cuda:
        raise OrcaHipError(f"{name} is on '{t.device}': orca_amd has no CPU path (move it to the MI355X with .cuda())")
    if t.dtype != torch.float32:
        t = t.float()
    return t


def _p(t):
    return ctypes.c_void_p(t.data_ptr())


def encoder_num_bins(L):
    return int(_lib.load().orca_encoder_num_bins(int(L)))


def encoder_forward(net, x, bin_lo=0, bin_hi=0, chunk_bp=0, out=None):
    x = _f32_cuda(x, "x")
    if x.dim() != 3 or x.shape[1] != 4:
        raise ValueError(f"Encoder input must be [B,4,L], got {tuple(x.shape)}")
    B, _, L = x.shape
    total = encoder_num_bins(L)
    hi = total if bin_hi <= 0 else bin_hi
    if out is None:
        out = torch.empty((B, 128, hi - bin_lo), dtype=torch.float32, device=x.device)
    net.ctx.sync_stream()
    check(_lib.load().orca_encoder_forward(net.ctx.handle, net.handle, _p(x), x.stride(0), x.stride(1), x.stride(2), B, L,
                                           bin_lo, hi, _p(out), out.stride(0), out.stride(1), chunk_bp), "orca_encoder_forward")
    return out


def pack_sequence(x):
    """x: [B,4,L] float ROCm tensor (any strides) -> (codes uint8 [B,L], packable).  ``packable`` is False if some
    row is neither one-hot nor the 0.25 'N' row; the codes are then meaningless and the float path must be used."""
    x = _f32_cuda(x, "x")
    B, C, L = x.shape
    if C != 4:
        raise ValueError(f"sequence must be [B,4,L], got {tuple(x.shape)}")
    codes = torch.empty((B, L), dtype=torch.uint8, device=x.device)
    ctx = get_context(x.device)
    ok = True
    for b in range(B):
        flag = ctypes.c_int()
        check(_lib.load().orca_pack_sequence(ctx.handle, ctypes.c_void_p(x[b].data_ptr()), x.stride(1), x.stride(2), L,
                                             ctypes.c_void_p(codes[b].data_ptr()), ctypes.byref(flag)), "orca_pack_sequence")
        ok = ok and bool(flag.value)
    return codes, ok


class CodeWindow:
    """A rank's share of a packed sequence: ``codes`` [B, n] = bases [origin, origin + n) of an L-base sequence (see
    `code_window_range` for what a bin range needs).  Accepted wherever the Encoder takes packed bases."""

    def __init__(self, codes, origin, L):
        self.codes, self.origin, self.L = codes, int(origin), int(L)

    @property
    def shape(self):
        return (self.codes.shape[0], self.L)

    device = property(lambda self: self.codes.device)
    dtype = property(lambda self: self.codes.dtype)


def code_window_range(L, bin_lo, bin_hi, reverse=False, halo=112000, binsize=4000):
    """[b0, b1): bases of the FORWARD sequence the Encoder reads for bins [bin_lo, bin_hi) of the forward strand / of the reverse
    complement (input halo of orca_modules.py:929-980 included; the last bin range runs to the end of the sequence)."""
    total = encoder_num_bins(L)
    lo = max(0, bin_lo * binsize - halo)
    hi = L if bin_hi >= total else min(L, bin_hi * binsize + halo)
    return (L - hi, L - lo) if reverse else (lo, hi)


def encoder_forward_codes(net, codes, reverse=False, bin_lo=0, bin_hi=0, chunk_bp=0, out=None):
    win = codes if isinstance(codes, CodeWindow) else None
    if win is not None:
        codes = win.codes
    if not (isinstance(codes, torch.Tensor) and codes.is_cuda and codes.dtype == torch.uint8 and codes.dim() == 2):
        raise ValueError("codes must be a [B,L] uint8 ROCm tensor")
    if codes.stride(1) != 1:
        codes = codes.contiguous()
    B, L = codes.shape
    if win is not None:
        L = win.L
    total = encoder_num_bins(L)
    hi = total if bin_hi <= 0 else bin_hi
    if out is None:
        out = torch.empty((B, 128, hi - bin_lo), dtype=torch.float32, device=codes.device)
    elif tuple(out.shape) != (B, 128, hi - bin_lo) or out.stride(2) != 1:
        raise ValueError(f"out must be a [{B},128,{hi - bin_lo}] view with unit stride along the bins")
    net.ctx.sync_stream()
    if win is not None:
        check(_lib.load().orca_encoder_forward_codes_window(net.ctx.handle, net.handle, _p(codes), codes.stride(0), win.origin, codes.shape[1],
                                                            1 if reverse else 0, B, L, bin_lo, hi, _p(out), out.stride(0), out.stride(1), chunk_bp),
              "orca_encoder_forward_codes_window")
        return out
    check(_lib.load().orca_encoder_forward_codes(net.ctx.handle, net.handle, _p(codes), codes.stride(0), 1 if reverse else 0, B, L,
                                                 bin_lo, hi, _p(out), out.stride(0), out.stride(1), chunk_bp), "orca_encoder_forward_codes")
    return out


# ---- the Encoder in two parts (include/orca_hip.h: orca_encoder_stage3_planes ...; sv.Stage3Cache is the user) -------------------------------------
def p16_plane_units(n):
    """16-byte units per plane of a P16 sequence tensor of n positions (guards and tile padding included)."""
    return int(_lib.load().orca_p16_plane_units(int(n)))


def _codes1d(codes):
    if not (isinstance(codes, torch.Tensor) and codes.is_cuda and codes.dtype == torch.uint8 and codes.dim() == 1 and codes.is_contiguous()):
        raise ValueError("codes must be a contiguous [L] uint8 ROCm tensor")
    return codes


def _planes(t, units, name):
    if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() and tuple(t.shape) == (32, units, 4)):
        raise ValueError(f"{name}: a contiguous [32,{units},4] float32 ROCm tensor (32 P16 planes of {units} 16-byte units)")
    return t


def encoder_stage3_planes(net, codes, reverse=False):
    """Stage 3's output (before MaxPool1d(5)) of the bases ``codes`` [L] (L % 80 == 0; ``reverse``: of their reverse complement) as 32 P16
    planes: a [32, units, 4] float32 tensor, position j at unit 8 + j."""
    codes = _codes1d(codes)
    L = codes.numel()
    if L <= 0 or L % 80:
        raise ValueError("stage-3 planes: the number of bases must be a positive multiple of 80")
    units = p16_plane_units(L // 16)
    planes = torch.empty((32, units, 4), dtype=torch.float32, device=codes.device)
    net.ctx.sync_stream()
    check(_lib.load().orca_encoder_stage3_planes(net.ctx.handle, net.handle, _p(codes), L, 1 if reverse else 0, _p(planes), units), "orca_encoder_stage3_planes")
    return planes


def p16_pool5_into(ctx, src, src_pos0, dst, dst_pos0, count):
    """MaxPool1d(5) of positions [src_pos0, src_pos0 + 5 count) of the planes ``src`` into positions [dst_pos0, dst_pos0 + count) of ``dst``."""
    _planes(src, src.shape[1], "src")
    _planes(dst, dst.shape[1], "dst")
    ctx.sync_stream()
    check(_lib.load().orca_p16_pool5_into(ctx.handle, _p(src), src.shape[1], int(src_pos0), _p(dst), dst.shape[1], int(dst_pos0), int(count)), "orca_p16_pool5_into")


def encoder_front_snippet(net, codes, reverse, base0, nbases, skip, count, dst, dst_pos0):
    """Stages 1-3 + MaxPool1d(5) on strand positions [base0, base0 + nbases) of ``codes`` [L]; pooled positions [skip, skip + count) go to
    positions [dst_pos0, ..) of the stage-4 input planes ``dst``."""
    codes = _codes1d(codes)
    _planes(dst, dst.shape[1], "dst")
    net.ctx.sync_stream()
    check(_lib.load().orca_encoder_front_snippet(net.ctx.handle, net.handle, _p(codes), codes.numel(), 1 if reverse else 0, int(base0), int(nbases), int(skip), int(count),
                                                 _p(dst), dst.shape[1], int(dst_pos0)), "orca_encoder_front_snippet")


def encoder_back(net, s4, n4, out):
    """Stages 4-7 from the stage-4 input planes ``s4`` ([32, p16_plane_units(n4), 4]) into ``out`` [128, n4 / 50] (unit stride along the bins)."""
    _planes(s4, p16_plane_units(n4), "s4")
    if not (out.is_cuda and out.dtype == torch.float32 and tuple(out.shape) == (128, n4 // 50) and out.stride(1) == 1):
        raise ValueError(f"out must be a [128,{n4 // 50}] float32 view with unit stride along the bins")
    net.ctx.sync_stream()
    check(_lib.load().orca_encoder_back(net.ctx.handle, net.handle, _p(s4), s4.shape[1], int(n4), _p(out), out.stride(0)), "orca_encoder_back")
    return out


def _rows(t, name):
    if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() and t.dim() == 2 and t.shape[1] == 128):
        raise ValueError(f"{name}: a contiguous [n,128] float32 ROCm tensor")
    return t


def encoder_stage4_rows(net, s4, n4):
    """Stage 4 alone: stage-4 input planes ``s4`` ([32, p16_plane_units(n4), 4]) -> its output rows [n4, 128] (before the MaxPool1d(5) in front of stage 5)."""
    _planes(s4, p16_plane_units(n4), "s4")
    rows = torch.empty((n4, 128), dtype=torch.float32, device=s4.device)
    net.ctx.sync_stream()
    check(_lib.load().orca_encoder_stage4_rows(net.ctx.handle, net.handle, _p(s4), s4.shape[1], int(n4), _p(rows)), "orca_encoder_stage4_rows")
    return rows


def rows_pool5_into(ctx, src, src_pos0, dst, dst_pos0, count):
    """MaxPool1d(5) of rows [src_pos0, src_pos0 + 5 count) of ``src`` [n,128] into rows [dst_pos0, dst_pos0 + count) of ``dst`` [m,128]."""
    _rows(src, "src")
    _rows(dst, "dst")
    ctx.sync_stream()
    check(_lib.load().orca_rows_pool5_into(ctx.handle, _p(src), src.shape[0], int(src_pos0), _p(dst), dst.shape[0], int(dst_pos0), int(count)), "orca_rows_pool5_into")


def encoder_front4_snippet(net, codes, reverse, base0, nbases, skip, count, dst, dst_pos0):
    """Stages 1-4 + MaxPool1d(5) on strand positions [base0, base0 + nbases) of ``codes`` [L] (multiples of 400); pooled rows [skip, skip + count)
    go to rows [dst_pos0, ..) of the stage-5 input ``dst`` [n5,128]."""
    codes = _codes1d(codes)
    _rows(dst, "dst")
    net.ctx.sync_stream()
    check(_lib.load().orca_encoder_front4_snippet(net.ctx.handle, net.handle, _p(codes), codes.numel(), 1 if reverse else 0, int(base0), int(nbases), int(skip), int(count),
                                                  _p(dst), dst.shape[0], int(dst_pos0)), "orca_encoder_front4_snippet")


def encoder_front4_ranges(net, codes, reverse, ranges, dst):
    """Stages 1-4 + MaxPool1d(5) on ALL of ``codes`` [L] (the snippets of one window strand, concatenated); ranges = [(skip, count, dst_pos0)]: pooled
    rows [skip, skip + count) of that run go to rows [dst_pos0, ..) of ``dst`` [n5,128]."""
    codes = _codes1d(codes)
    _rows(dst, "dst")
    flat = (ctypes.c_int64 * (3 * len(ranges)))(*[int(v) for r in ranges for v in r])
    net.ctx.sync_stream()
    check(_lib.load().orca_encoder_front4_ranges(net.ctx.handle, net.handle, _p(codes), codes.numel(), 1 if reverse else 0, len(ranges), flat, _p(dst), dst.shape[0]),
          "orca_encoder_front4_ranges")


def encoder_back5(net, rows, out):
    """Stages 5-7 from the stage-5 input ``rows`` [n5,128] into ``out`` [128, n5 / 10] (unit stride along the bins)."""
    _rows(rows, "rows")
    n5 = rows.shape[0]
    if not (out.is_cuda and out.dtype == torch.float32 and tuple(out.shape) == (128, n5 // 10) and out.stride(1) == 1):
        raise ValueError(f"out must be a [128,{n5 // 10}] float32 view with unit stride along the bins")
    net.ctx.sync_stream()
    check(_lib.load().orca_encoder_back5(net.ctx.handle, net.handle, _p(rows), n5, _p(out), out.stride(0)), "orca_encoder_back5")
    return out


def encoder_forward_2bit(net, two, nmask, start, L, reverse=False, bin_lo=0, bin_hi=0, chunk_bp=0, out=None):
    """Encoder on bases [start, start + L) of a chromosome stored as 2 bits per base + N mask in HBM (genome.TwoBitGenome planes): no
    unpacked window is made (orca_encoder_forward_2bit).  Returns [1,128,bins]."""
    if not (two.is_cuda and nmask.is_cuda and two.dtype == torch.uint8 and nmask.dtype == torch.uint8 and two.is_contiguous() and nmask.is_contiguous()):
        raise ValueError("two / nmask: contiguous uint8 ROCm tensors")
    if start < 0 or (start + L + 3) // 4 > two.numel() or (start + L + 7) // 8 > nmask.numel():
        raise ValueError("2-bit window outside the chromosome")
    total = encoder_num_bins(L)
    hi = total if bin_hi <= 0 else bin_hi
    if out is None:
        out = torch.empty((1, 128, hi - bin_lo), dtype=torch.float32, device=two.device)
    elif tuple(out.shape) != (1, 128, hi - bin_lo) or out.stride(2) != 1:
        raise ValueError(f"out must be a [1,128,{hi - bin_lo}] view with unit stride along the bins")
    net.ctx.sync_stream()
    check(_lib.load().orca_encoder_forward_2bit(net.ctx.handle, net.handle, _p(two), _p(nmask), int(start), 1 if reverse else 0, int(L), bin_lo, hi,
                                                _p(out), out.stride(1), chunk_bp), "orca_encoder_forward_2bit")
    return out


def unet_forward(net, x, nlev):
    x = _f32_cuda(x, "x")
    if x.dim() != 3 or x.shape[1] != 128:
        raise ValueError(f"Encoder2/3 input must be [B,128,n], got {tuple(x.shape)}")
    B, _, n = x.shape
    outs = [torch.empty((B, 128, n >> i), dtype=torch.float32, device=x.device) for i in range(nlev + 1)]
    ptrs = (ctypes.c_void_p * (nlev + 1))(*[o.data_ptr() for o in outs])
    net.ctx.sync_stream()
    check(_lib.load().orca_unet_forward(net.ctx.handle, net.handle, _p(x), x.stride(0), x.stride(1), x.stride(2), B, n, ptrs,
                                        nlev + 1), "orca_unet_forward")
    return outs


def decoder_forward(net, x, distenc, y=None, out=None, accumulate=False):
    """x [B,128,n]; distenc [B or 1, T, n, n]; y None or [B,T,n/2,n/2]; returns / fills [B,T,n,n]
    (T = the net's number of target maps: 1 for the Orca models, num_2d for the orca_leukemia decoders)."""
    x = _f32_cuda(x, "x")
    distenc = _f32_cuda(distenc, "distenc")
    B, C, n = x.shape
    T = net.num_targets()
    if C != 128:
        raise ValueError(f"Decoder input must be [B,128,n], got {tuple(x.shape)}")
    if distenc.dim() != 4 or distenc.shape[1] != T or distenc.shape[2] != n or distenc.shape[3] != n:
        raise ValueError(f"distenc must be [B,{T},{n},{n}], got {tuple(distenc.shape)}")
    if distenc.shape[0] not in (1, B):
        raise ValueError("distenc batch mismatch")
    sd_b = distenc.stride(0) if distenc.shape[0] == B else 0
    yp, sy = ctypes.c_void_p(0), (0, 0, 0, 0)
    if y is not None:
        y = _f32_cuda(y, "y")
        if tuple(y.shape) != (B, T, n // 2, n // 2):
            raise ValueError(f"coarse prediction must be [{B},{T},{n // 2},{n // 2}], got {tuple(y.shape)}")
        yp, sy = _p(y), (y.stride(0), y.stride(1), y.stride(2), y.stride(3))
    if out is None:
        out = torch.empty((B, T, n, n), dtype=torch.float32, device=x.device)
        accumulate = False
    elif tuple(out.shape) != (B, T, n, n) or not out.is_contiguous():
        raise ValueError(f"out must be a contiguous [{B},{T},{n},{n}] tensor")
    net.ctx.sync_stream()
    if T == 1:
        check(_lib.load().orca_decoder_forward(net.ctx.handle, net.handle, _p(x), x.stride(0), x.stride(1), x.stride(2), _p(distenc),
                                               sd_b, distenc.stride(2), distenc.stride(3), yp, sy[0], sy[2], sy[3], B, n, _p(out),
                                               1 if accumulate else 0), "orca_decoder_forward")
    else:
        check(_lib.load().orca_decoder_forward_mt(net.ctx.handle, net.handle, _p(x), x.stride(0), x.stride(1), x.stride(2),
                                                  _p(distenc), sd_b, distenc.stride(1), distenc.stride(2), distenc.stride(3), yp,
                                                  sy[0], sy[1], sy[2], sy[3], B, n, _p(out), 1 if accumulate else 0),
              "orca_decoder_forward_mt")
    return out


def _row_ptrs(rows):
    return (ctypes.c_void_p * len(rows))(*[r.data_ptr() for r in rows])


def _same_strides(rows, name):
    st = rows[0].stride()
    if any(r.stride() != st or r.shape != rows[0].shape for r in rows):
        raise ValueError(f"{name}: the rows of a batch must share one shape and one set of strides")
    return st


def decoder_forward_rows(net, xs, des, ys=None, out=None, accumulate=False):
    """Decoder on a batch given ROW BY ROW: xs = B views [128,n], des = B views [T,n,n] (the same view may repeat),
    ys = None or B views [T,n/2,n/2] - any strides, no copies (orca_decoder_forward_rows)."""
    xs = [_f32_cuda(t, "x") for t in xs]
    des = [_f32_cuda(t, "distenc") for t in des]
    B, T = len(xs), net.num_targets()
    C, n = xs[0].shape
    if C != 128 or tuple(des[0].shape) != (T, n, n) or len(des) != B:
        raise ValueError(f"decoder rows: x [128,n] and distenc [{T},n,n] per row, got {tuple(xs[0].shape)} / {tuple(des[0].shape)}")
    sx, sd = _same_strides(xs, "x"), _same_strides(des, "distenc")
    yp, sy = None, (0, 0, 0)
    if ys is not None:
        ys = [_f32_cuda(t, "y") for t in ys]
        if len(ys) != B or tuple(ys[0].shape) != (T, n // 2, n // 2):
            raise ValueError(f"coarse prediction rows must be [{T},{n // 2},{n // 2}]")
        yp, sy = _row_ptrs(ys), _same_strides(ys, "y")
    if out is None:
        out = torch.empty((B, T, n, n), dtype=torch.float32, device=xs[0].device)
        accumulate = False
    net.ctx.sync_stream()
    check(_lib.load().orca_decoder_forward_rows(net.ctx.handle, net.handle, _row_ptrs(xs), sx[0], sx[1], _row_ptrs(des), sd[0], sd[1], sd[2],
                                                yp, sy[0], sy[1], sy[2], B, n, _p(out), 1 if accumulate else 0), "orca_decoder_forward_rows")
    return out


def decoder1m_forward_rows(net, xs, out=None, accumulate=False):
    xs = [_f32_cuda(t, "x") for t in xs]
    B, T = len(xs), net.num_targets()
    C, n = xs[0].shape
    sx = _same_strides(xs, "x")
    if out is None:
        out = torch.empty((B, T, n, n), dtype=torch.float32, device=xs[0].device)
        accumulate = False
    net.ctx.sync_stream()
    check(_lib.load().orca_decoder1m_forward_rows(net.ctx.handle, net.handle, _row_ptrs(xs), sx[0], sx[1], B, n, _p(out), 1 if accumulate else 0),
          "orca_decoder1m_forward_rows")
    return out


def decoder1m_forward(net, x, out=None, accumulate=False):
    x = _f32_cuda(x, "x")
    B, C, n = x.shape
    T = net.num_targets()
    if C != 128:
        raise ValueError(f"Decoder_1m input must be [B,128,n], got {tuple(x.shape)}")
    if out is None:
        out = torch.empty((B, T, n, n), dtype=torch.float32, device=x.device)
        accumulate = False
    elif tuple(out.shape) != (B, T, n, n) or not out.is_contiguous():
        raise ValueError(f"out must be a contiguous [{B},{T},{n},{n}] tensor")
    net.ctx.sync_stream()
    check(_lib.load().orca_decoder1m_forward(net.ctx.handle, net.handle, _p(x), x.stride(0), x.stride(1), x.stride(2), B, n,
                                             _p(out), 1 if accumulate else 0), "orca_decoder1m_forward")
    return out


def strand_merge(fwd, rev):
    """0.5*fwd + 0.5*rev[::-1, ::-1] for contiguous [n,n] maps (orca_predict.py:514-523)."""
    fwd, rev = _f32_cuda(fwd, "fwd").contiguous(), _f32_cuda(rev, "rev").contiguous()
    n = fwd.shape[-1]
    out = torch.empty_like(fwd)
    ctx = get_context(fwd.device)
    check(_lib.load().orca_strand_merge(ctx.handle, _p(fwd), _p(rev), _p(out), n), "orca_strand_merge")
    return out


def block_mean(normmat, start, nb, npix=250, flip=False, want_mean=True):
    """Background of one level of the 256 Mb cascade on the device (orca_predict.py:724-737, :692-703): block means of the
    window [start, start + npix*nb)^2 of a float64 CUDA matrix -> (float64 [npix,npix] means or None, float32 [1,1,npix,npix]
    log-background, flipped in both axes if ``flip``).  The means are bit-identical to numpy's nanmean-of-nanmean."""
    if not (normmat.is_cuda and normmat.dtype == torch.float64 and normmat.dim() == 2 and normmat.stride(1) == 1):
        raise OrcaHipError("block_mean: a float64 [n,n] ROCm tensor with unit column stride is required")
    if start < 0 or start + npix * nb > min(normmat.shape):
        raise ValueError(f"block_mean: window [{start}, {start + npix * nb}) outside the {tuple(normmat.shape)} background")
    ctx = get_context(normmat.device)
    ctx.sync_stream()
    mean = torch.empty((npix, npix), dtype=torch.float64, device=normmat.device) if want_mean else None
    logt = torch.empty((1, 1, npix, npix), dtype=torch.float32, device=normmat.device)
    check(_lib.load().orca_block_mean_f64(ctx.handle, _p(normmat), normmat.stride(0), int(start), int(start), int(nb), int(npix),
                                          _p(mean) if mean is not None else None, _p(logt), 1 if flip else 0), "orca_block_mean_f64")
    return mean, logt


# ---- single layers (kernel unit tests) -------------------------------------
def conv1d(x, w, b, relu=False, r1=None, r2=None, tile=0):
    """y = [relu](conv1d_k9(x, w) + b) [+ r1] [+ r2]; x [B,cin,n] contiguous cuda."""
    x = _f32_cuda(x, "x").contiguous()
    B, cin, n = x.shape
    w = np.ascontiguousarray(w, dtype=np.float32)
    b = np.ascontiguousarray(b, dtype=np.float32)
    cout = w.shape[0]
    d = make_descs([{"w": w, "b": b, "cout": cout, "cin": cin, "k": 9, "dil": 1}])
    y = torch.empty((B, cout, n), dtype=torch.float32, device=x.device)
    ctx = get_context(x.device)
    r1 = r1.contiguous() if r1 is not None else None
    r2 = r2.contiguous() if r2 is not None else None
    check(_lib.load().orca_conv1d_forward(ctx.handle, d, _p(x), cin * n, n, _p(y), cout * n, n,
                                          _p(r1) if r1 is not None else None, _p(r2) if r2 is not None else None, B, n,
                                          1 if relu else 0, tile), "orca_conv1d_forward")
    return y


def conv1d_nlc(x, w, b, precision="bf16x3", relu=False, r1=None):
    """channel-last split-bf16 conv: x [B,n,cin] -> [B,n,cout]."""
    x = _f32_cuda(x, "x").contiguous()
    B, n, cin = x.shape
    w = np.ascontiguousarray(w, dtype=np.float32)
    b = np.ascontiguousarray(b, dtype=np.float32)
    cout = w.shape[0]
    d = make_descs([{"w": w, "b": b, "cout": cout, "cin": cin, "k": 9, "dil": 1}])
    y = torch.empty((B, n, cout), dtype=torch.float32, device=x.device)
    ctx = get_context(x.device)
    r1 = r1.contiguous() if r1 is not None else None
    check(_lib.load().orca_conv1d_nlc_forward(ctx.handle, d, _lib.PRECISIONS[precision], _p(x), _p(y),
                                              _p(r1) if r1 is not None else None, B, n, 1 if relu else 0), "orca_conv1d_nlc_forward")
    return y


def conv1d_p16(x, w, b, relu=False, r1=None, out_mode=0, fmt="p16"):
    """P16 / LDS-DMA conv (test wrapper): x [n,cin] channel-last fp32 -> [n, n/4 (out_mode 1) or n/5 (out_mode 3: 128 couts), cout].
    fmt="b16": the same kernel on single-plane bf16 activations (cin % 32 == 0)."""
    x = _f32_cuda(x, "x").contiguous()
    n, cin = x.shape
    w = np.ascontiguousarray(w, dtype=np.float32)
    b = np.ascontiguousarray(b, dtype=np.float32)
    cout = w.shape[0]
    d = make_descs([{"w": w, "b": b, "cout": cout, "cin": cin, "k": int(w.shape[2]), "dil": 1}])     # 9 taps, or 17 (the composed-pair form)
    y = torch.empty((n // 4 if out_mode == 1 else n // 5 if out_mode == 3 else n, cout), dtype=torch.float32, device=x.device)
    if y.numel() == 0:
        return y
    ctx = get_context(x.device)
    r1 = r1.contiguous() if r1 is not None else None
    fn = _lib.load().orca_conv1d_b16_forward if fmt == "b16" else _lib.load().orca_conv1d_p16_forward
    check(fn(ctx.handle, d, _p(x), _p(y), _p(r1) if r1 is not None else None, n, 1 if relu else 0, out_mode), f"orca_conv1d_{fmt}_forward")
    return y


def conv2d(x, w, b, dilation=1, relu=False, r=None):
    x = _f32_cuda(x, "x").contiguous()
    B, cin, n, _ = x.shape
    w = np.ascontiguousarray(w, dtype=np.float32)
    b = np.ascontiguousarray(b, dtype=np.float32)
    cout = w.shape[0]
    d = make_descs([{"w": w, "b": b, "cout": cout, "cin": cin, "k": 3, "dil": dilation}])
    y = torch.empty((B, cout, n, n), dtype=torch.float32, device=x.device)
    ctx = get_context(x.device)
    r = r.contiguous() if r is not None else None
    check(_lib.load().orca_conv2d_forward(ctx.handle, d, _p(x), _p(y), _p(r) if r is not None else None, B, n,
                                          1 if relu else 0), "orca_conv2d_forward")
    return y


def conv2d_m16(x, w, b, dilation=1, relu=False, r=None, precision="f16x2"):
    """The Decoders' 16-bit 3x3 conv on M16 maps (test wrapper): x [B,cin,n,n] -> [B,cout,n,n]; dilation 1..8."""
    x = _f32_cuda(x, "x").contiguous()
    B, cin, n, _ = x.shape
    w = np.ascontiguousarray(w, dtype=np.float32)
    b = np.ascontiguousarray(b, dtype=np.float32)
    cout = w.shape[0]
    d = make_descs([{"w": w, "b": b, "cout": cout, "cin": cin, "k": 3, "dil": dilation}])
    y = torch.empty((B, cout, n, n), dtype=torch.float32, device=x.device)
    ctx = get_context(x.device)
    r = r.contiguous() if r is not None else None
    check(_lib.load().orca_conv2d_m16_forward(ctx.handle, d, _lib.PRECISIONS[precision], _p(x), _p(y), _p(r) if r is not None else None, B, n,
                                              1 if relu else 0), "orca_conv2d_m16_forward")
    return y


def maxpool1d(x, k):
    x = _f32_cuda(x, "x").contiguous()
    B, C, n = x.shape
    y = torch.empty((B, C, n // k), dtype=torch.float32, device=x.device)
    ctx = get_context(x.device)
    check(_lib.load().orca_maxpool1d_forward(ctx.handle, _p(x), n, _p(y), n // k, B * C, n // k, k), "orca_maxpool1d_forward")
    return y


def pointwise1d(x, w_dev, b_dev, act="none"):
    """Kernel-size-1 Conv1d + activation (Net.final_1d layers): x [B,cin,n] ROCm fp32, w_dev [cout,cin] / b_dev [cout]
    ROCm tensors (folded weights).  act: "none" | "relu" | "sigmoid"."""
    x = _f32_cuda(x, "x").contiguous()
    B, cin, n = x.shape
    cout = w_dev.shape[0]
    if tuple(w_dev.shape) != (cout, cin) or not w_dev.is_cuda or not w_dev.is_contiguous():
        raise ValueError("w_dev must be a contiguous [cout,cin] ROCm tensor")
    y = torch.empty((B, cout, n), dtype=torch.float32, device=x.device)
    ctx = get_context(x.device)
    check(_lib.load().orca_pointwise1d_forward(ctx.handle, _p(w_dev), _p(b_dev), cout, cin, _p(x), cin * n, n, _p(y), cout * n, n, B, n,
                                               {"none": 0, "relu": 1, "sigmoid": 2}[act]), "orca_pointwise1d_forward")
    return y
