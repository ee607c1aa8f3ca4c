"""Multiscale prediction drivers with the reference's signatures and output
dictionaries (/root/reference/orca_predict.py: genomepredict :231-540,
genomepredict_256Mb :543-878).

This is the host-side caller of the hot path: it walks the coarse-to-fine
cascade (both strands, every model), hands every numeric step to the model
protocol (``net0 / net / net1 / denets[level] / denet_1_pt``) and assembles the
result dict.  With the orca_amd models those protocol calls are the HIP
kernels; the only tensor work done here is slicing/expanding views and the
H2D/D2H copies at the ends.  ``use_cuda=False`` simply leaves tensors on the
host, which the orca_amd modules refuse (no CPU path) - it exists so that the
cascade bookkeeping can be exercised with foreign ``torch.nn.Module`` models.
"""
import os

import numpy as np
import torch

from . import engine

model_dict_global = {}


def load_resources(models=("32M",), use_cuda=True, model_dir=None):
    """Build the named model containers into ``model_dict_global``
    (orca_predict.py:42-116; genome / cooler resources are out of scope)."""
    from . import orca_models
    if "32M" in models:
        for key, cls in (("h1esc", orca_models.H1esc), ("hff", orca_models.Hff)):
            if key not in model_dict_global:
                model_dict_global[key] = cls(model_dir=model_dir)
    if "256M" in models:
        for key, cls in (("h1esc_256m", orca_models.H1esc_256M), ("hff_256m", orca_models.Hff_256M)):
            if key not in model_dict_global:
                model_dict_global[key] = cls(model_dir=model_dir)
    if "1M" in models or "1m" in models:      # orca_predict.py:120-133
        for key, cls in (("h1esc_1m", orca_models.H1esc_1M), ("hff_1m", orca_models.Hff_1M)):
            if key not in model_dict_global:
                model_dict_global[key] = cls(model_dir=model_dir)
    if use_cuda:
        for m in model_dict_global.values():
            m.cuda()


def _resolve_models(models, group, use_cuda):
    objs = []
    for m in models:
        if isinstance(m, torch.nn.Module):
            objs.append(m)
        else:
            if m not in model_dict_global:
                load_resources(models=[group], use_cuda=use_cuda)
            objs.append(model_dict_global[m])
    return objs


import contextlib as _contextlib
import threading as _threading

_enc_scope = _threading.local()
SHARE_ENCODINGS = True      # False: every view of a 256 Mb driver call is encoded on its own, as the reference does (the tests' A/B handle)


@_contextlib.contextmanager
def shared_encodings():
    """Within this scope (one thread) `genomepredict_256Mb` keeps the Encoder output of a packed sequence per (sequence storage, Encoder) and
    reuses it when the SAME sequence comes again: the structural-variant drivers predict one 256 Mb sequence at two anchors (the reference
    chromosome at the variant's left and right end, `orca_predict.py:1335 / :1389`; both views of an inversion) - the reference encodes it
    twice, 2 x 128 Mb strand pairs of Encoder work each.  Nothing outlives the scope."""
    prev = getattr(_enc_scope, "cache", None)
    if not SHARE_ENCODINGS:
        yield
        return
    _enc_scope.cache = {} if prev is None else prev
    try:
        yield
    finally:
        _enc_scope.cache = prev


class _StrandInputs:
    """The two strands of a [B,L,4] float sequence as the models want them.

    Reference (orca_predict.py:324-337): the reverse complement is a second, flipped 512 MB host copy
    (`sequence[:, ::-1, ::-1].copy()`), and each strand is uploaded per model.  Here the forward strand is
    uploaded ONCE; if every row is one-hot or the 0.25 'N' row it is packed to 1 byte per base on the device and an
    orca_amd Encoder encodes both strands from that one buffer (reverse complement = index/ code flip inside the
    first-layer kernel).  Anything else (arbitrary floats, foreign models, CPU) uses float views as the reference does."""

    def __init__(self, sequence, use_cuda):
        self.use_cuda = use_cuda
        self._fwd = self._rev = self._codes = None
        self._packable = None
        self._device = None        # where the strands live, recorded at the first upload (never re-materialises .fwd)
        self.uploads = 0           # H2D copies of the float window (tests: at most one per call, whatever the model count)
        if isinstance(sequence, torch.Tensor) and sequence.dtype == torch.uint8:
            # already packed: [B,L] base codes on the MI355X (orca_amd extension, see orca_amd/sv.py)
            if not (use_cuda and sequence.is_cuda and sequence.dim() == 2):
                raise ValueError("packed input must be a [B,L] uint8 ROCm tensor with use_cuda=True")
            self.seq, self._codes, self._packable = None, sequence, True
            self._device = sequence.device
            self.batch = sequence.shape[0]
            return
        self.seq = np.asarray(sequence, dtype=np.float32)
        self.batch = self.seq.shape[0]

    @property
    def device(self):
        if self._device is None:
            self._device = self.fwd.device
        return self._device

    @property
    def fwd(self):
        if self._fwd is None:
            t = torch.from_numpy(np.ascontiguousarray(self.seq))
            if self.use_cuda:
                t = t.cuda()
                self.uploads += 1
            self._fwd = t.transpose(1, 2)
            self._device = self._fwd.device
        return self._fwd

    @property
    def rev(self):
        if self._rev is None:
            if self.use_cuda:   # flip length and channel axes on the device instead of a second host copy + upload
                self._rev = torch.flip(self.fwd.transpose(1, 2), [1, 2]).transpose(1, 2)
            else:
                self._rev = torch.from_numpy(np.ascontiguousarray(self.seq[:, ::-1, ::-1])).transpose(1, 2)
        return self._rev

    def _pack(self):
        if self._packable is None:
            self._codes, self._packable = engine.pack_sequence(self.fwd)
            if self._packable:
                self._fwd = None   # the float copy is no longer needed on the device
        return self._packable

    def encode(self, net0):
        """[2B,128,n_bins]: forward strand rows first, reverse strand rows second."""
        from .dist import ShardedEncoder
        from .orca_modules import Encoder
        if self.seq is None and not isinstance(net0, (Encoder, ShardedEncoder)):
            raise TypeError("packed (uint8) input needs an orca_amd Encoder as model.net0")
        if self.use_cuda and isinstance(net0, (Encoder, ShardedEncoder)) and self._pack():
            if isinstance(net0, Encoder):     # both strands straight into the halves of one [2B,128,bins] tensor
                B, L = self._codes.shape
                cache = getattr(_enc_scope, "cache", None)          # `shared_encodings()`: the same packed sequence at another anchor
                key = (self._codes.data_ptr(), tuple(self._codes.shape), self._codes._version, id(net0), net0.precision, engine._guard["force_safe"])
                if cache is not None and key in cache:
                    return cache[key][1]
                enc0 = torch.empty((2 * B, 128, engine.encoder_num_bins(L)), dtype=torch.float32, device=self._codes.device)
                _encode_two_strands(lambda rev, out: net0.forward_codes(self._codes, reverse=rev, out=out), enc0, B)
                if cache is not None:
                    cache[key] = (self._codes, enc0)      # (the codes tensor is HELD: its storage - the key - cannot be handed to another sequence meanwhile)
                    # inside a deferred range check (genomepredict_256Mb's forward) this pass is not known to be clean yet: if the check
                    # at the end of the cascade fires, the entry goes before the range-safe retry - and before another anchor can hit it
                    engine.tentative(lambda cache=cache, key=key: cache.pop(key, None))
                return enc0
            return torch.cat([net0.forward_codes(self._codes, reverse=False), net0.forward_codes(self._codes, reverse=True)], dim=0)
        return torch.cat([net0(self.fwd), net0(self.rev)], dim=0)


def _encode_two_strands(encode, enc0, B):
    """encode(reverse, out) for the forward strand into enc0[:B] and the reverse complement into enc0[B:]: one after the other on the
    caller's stream, or (engine.strand_streams(), opt-in) the reverse strand on an auxiliary context beside the forward one."""
    if engine.strand_streams() and enc0.is_cuda:
        pool = engine.context_pool(enc0.device, 1)
        pool.fork()
        pool.run(0, lambda: encode(True, enc0[B:]))
        encode(False, enc0[:B])
        pool.join()
    else:
        encode(False, enc0[:B])
        encode(True, enc0[B:])


def _log_background(bg, batch, use_cuda, flip=False):
    """log(background) as a [B,1,n,n] (expanded) tensor, orca_predict.py:349-353 / :692-697,:703."""
    a = np.asarray(bg)
    a = a[(None,) * (4 - a.ndim)]
    t = torch.log(torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)))
    if flip:
        t = torch.flip(t, [2, 3])
    if use_cuda:
        # pinned staging + asynchronous copy: a pageable H2D copy blocks the host until the stream reaches it, i.e. behind the whole Encoder
        # (0.58 s of genomepredict_256Mb's 0.84 s were spent in this call; the block means of the later levels then ran with the GPU idle)
        t = t.pin_memory().cuda(non_blocking=True)
    return t.expand(batch, -1, -1, -1)


def _cached_log_background(model, level, use_cuda):
    """log(normmats[level]) on the device, computed/uploaded once per model (the pageable H2D copy of the reference's
    per-call `torch.FloatTensor(normmat).cuda()` stalls the host until the GPU queue drains)."""
    arr = model.normmats[level]
    cache = model.__dict__.setdefault("_orca_amd_bg_cache", {})
    key = (level, bool(use_cuda), id(arr))
    if key not in cache:
        cache[key] = _log_background(arr, 1, use_cuda)
    return cache[key]


def _decode(model, level, enc_slice, distenc, coarse, add_1m):
    dec = model.denets[level]
    if coarse is None:
        pred = dec.forward(enc_slice, distenc)
    else:
        pred = dec.forward(enc_slice, distenc, coarse)
    if add_1m:
        pt = model.denet_1_pt
        if hasattr(pt, "forward_into") and pred.is_cuda:
            pt.forward_into(pred, enc_slice, accumulate=True)  # fused `+ denet_1_pt(x)`
        else:
            pred = pred + pt.forward(enc_slice)
    return pred


def _coarse_grain(mat, nblock, nan_thresh):
    """nan-aware block mean of the leading [T, 250*nblock, 250*nblock] window
    (orca_predict.py:404-436): mean over the block rows of the per-row means, blocks with more than
    ``nan_thresh`` missing entries set to NaN.  NaN-free windows (every background after the reference's in-place
    fill, orca_predict.py:664-667) take plain means - bit-identical to ``nanmean`` there (same pairwise sums,
    same counts) and 8x quicker on the 8000 x 8000 level-256 window."""
    T = mat.shape[0]
    r = np.reshape(mat, (T, 250, nblock, 250, nblock))
    if not np.isnan(mat).any():
        return np.mean(np.mean(r, axis=4), axis=2)
    with np.errstate(invalid="ignore"):
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("ignore", category=RuntimeWarning)
            mean = np.nanmean(np.nanmean(r, axis=4), axis=2)
    frac = np.mean(np.mean(np.isnan(r), axis=4), axis=2)
    mean[frac > nan_thresh] = np.nan
    return mean


class Background256:
    """Per-level, per-strand log-background of the 256 Mb cascade (orca_predict.py:692-703, :724-737): block means of the
    window of the 8000 x 8000 (32 kb bins) background at the strand's current start, log in float32, flipped in both axes on
    the reverse strand.  ``normmat`` is either the reference's host array (numpy float64; block means with numpy, overlapped
    with the Encoder's queue) or a float64 ROCm tensor RESIDENT IN HBM (`Background256.to_device`, `sv_drivers` build it there):
    then the block means run in one kernel per level (engine.block_mean, bit-identical float64 means) and nothing of the
    8000 x 8000 matrix crosses PCIe or the host's caches per call.  NaNs are filled with the smallest finite entry, in place,
    exactly once (:664-667).  Callable as run_cascade's ``background(level, k, start)`` for the strands ``reverse_flags``."""

    def __init__(self, normmat, reverse_flags=(False, True), use_cuda=True):
        self.on_device = isinstance(normmat, torch.Tensor) and normmat.is_cuda
        if self.on_device and normmat.dtype != torch.float64:
            raise ValueError("device-resident background: float64 expected (the reference's dtype)")
        if not self.on_device and isinstance(normmat, torch.Tensor):
            normmat = normmat.numpy()
        self.normmat, self.reverse_flags, self.use_cuda = normmat, [bool(r) for r in reverse_flags], use_cuda
        self.ns = [{} for _ in self.reverse_flags]            # per strand: {level: (start, block means[, {flip: log tensor}])}
        self._filled = False

    def _fill_nan(self):
        """`normmat[isnan] = nanmin(normmat)` in place (:664-667) - on first use, i.e. with the Encoder already enqueued."""
        if self._filled:
            return
        self._filled = True
        m = self.normmat
        if self.on_device:
            # in place, as the reference does (:664-667) - and ONCE per tensor: callers hand the same resident 8000 x 8000 matrix to every call
            # (bench.py, dist.strand_tail_256m), and the isnan pass over its 512 MB plus the host sync of `.any()` sat inside every timed tail
            # (the "already filled" state belongs to the STORAGE and its version counter, not to the caller's tensor object: an
            # in-place refill with a matrix that holds NaN bumps `_version` and is filled again; views cannot lose it)
            # the state lives ON the storage object (PyTorch keeps one Python object per live storage), so it dies with the
            # allocation - a key of (pointer, shape, version) outlived the tensor and matched the caching allocator's next block at that address
            st = m.untyped_storage()
            if getattr(st, "_orca_nan_filled", None) == m._version:
                return
            nan = torch.isnan(m)
            if bool(nan.any()):
                m[nan] = m[~nan].min()
            st._orca_nan_filled = m._version
        else:
            isnan = np.isnan(m)
            if np.any(isnan):
                m[isnan] = np.nanmin(m[~isnan])

    @staticmethod
    def to_device(normmat, device):
        """Upload a host background once (512 MB of float64) - for callers that reuse it over many calls."""
        return torch.from_numpy(np.ascontiguousarray(normmat, dtype=np.float64)).to(device)

    def reset(self):
        for d in self.ns:
            d.clear()

    def means(self, k, level):
        """float64 [250,250] block means strand k used at ``level`` (host array; output["normmats"])."""
        m = self.ns[k][level][1]
        return m.cpu().numpy() if isinstance(m, torch.Tensor) else m

    def __call__(self, level, k, start):
        self._fill_nan()
        nb, flip = level // 8, self.reverse_flags[k]
        hit = self.ns[k].get(level)
        if hit is None or hit[0] != start:
            hit = None
            for other in self.ns:                             # strands that start a level at the same bin share the means
                o = other.get(level)
                if o is not None and o[0] == start:
                    self.ns[k][level] = hit = o
                    break
        if self.on_device:
            if hit is not None and flip in hit[2]:
                return hit[2][flip]
            mean, logt = engine.block_mean(self.normmat, start, nb, 250, flip, want_mean=hit is None)
            if hit is None:
                self.ns[k][level] = hit = (start, mean, {})
            hit[2][flip] = logt
            return logt
        if hit is None:
            w = 250 * nb
            self.ns[k][level] = hit = (start, _coarse_grain(self.normmat[None, start: start + w, start: start + w], nb, 1))
        return _log_background(hit[1], 1, self.use_cuda, flip=flip)


def _scale_annotation(annotation, newstart, newend):
    """Clip / rescale plot annotations into the current window (orca_predict.py:451-468)."""
    span = newend - newstart
    out = []
    for r in annotation:
        if len(r) == 3:
            if not (r[0] >= newend or r[1] <= newstart):
                out.append((np.fmax((r[0] - newstart) / span, 0), np.fmin((r[1] - newstart) / span, 1), r[2]))
        elif newstart <= r[0] < newend:
            out.append(((r[0] - newstart) / span, r[1]))
    return out


def zoom_index_32m(level, start, mpos, wpos, reverse):
    """Offset (0..125 pixels of the current map) of the next, half-size window so that it
    is centred on ``mpos``: floor on the forward strand, ceil on the mirrored reverse
    strand (orca_predict.py:470-499).  ``start`` is in 4 kb bins from the window start."""
    half = level * 1000000 / 4
    if not reverse:
        v = np.floor(((mpos - half) - (wpos - 16000000 + start * 4000)) / (4000 * level))
    else:
        v = np.ceil(((wpos + 16000000 - start * 4000) - (mpos + half)) / (4000 * level))
    return int(np.clip(v, 0, 125))


def zoom_index_256m(level, start, mpos, wpos, chrlen, reverse):
    """256 Mb cascade: window offset with chromosome-end bounds (orca_predict.py:813-835); ``start`` in 32 kb bins."""
    half = level * 1000000 / 4
    if not reverse:
        proposed = (mpos - half) - (wpos - 128000000 + start * 32000)
    else:
        proposed = (mpos - half) - (wpos + 128000000 - start * 32000 - level * 1000000)
    if chrlen is not None:
        lo = 0 - (wpos - 128000000)
        hi = chrlen - level * 1000000 / 2 - (wpos - 128000000)
        proposed = np.clip(proposed, lo, hi) if lo < hi else lo
    i = int(np.clip(np.floor(proposed / (4000 * level)), 0, 125))
    return 250 - (i + 125) if reverse else i


def _rows(t, B, idx, width, dims):
    """Per-strand crops of a [nstrands*B, ...] tensor (strand k uses offset idx[k]) as a list of nstrands*B row VIEWS."""
    out = []
    for k, i in enumerate(idx):
        for b in range(B):
            sl = t[k * B + b]
            for d in dims:
                sl = sl.narrow(d - 1, i, width)
            out.append(sl)
    return out


def _gather(t, B, idx, width, dims):
    """The same crops restacked along the batch axis (a copy; used for models that are not orca_amd modules)."""
    parts = []
    for k, i in enumerate(idx):
        sl = t[k * B:(k + 1) * B]
        for d in dims:
            sl = sl.narrow(d, i, width)
        parts.append(sl)
    return parts[0] if len(parts) == 1 else torch.cat(parts, dim=0)


def run_cascade(model, encodings, levels, unit, B, reverse_flags, background, zoom, add_1m_level=None, on_level=None):
    """Coarse-to-fine decoder cascade for several strands AT ONCE: the strands are stacked along the batch axis
    (strand k = rows k*B..(k+1)*B-1 of every tensor) so each decoder level is ONE batched call - the maps are
    only 250x250, so batching the two strands fills the GPU twice as well.  Per strand the bookkeeping is exactly
    the reference's (orca_predict.py:389-500 / :723-838): slice the level's encoding at the strand's current
    start, crop the previous map at the strand's zoom offset, advance ``starts``.

    encodings: {level: [S*B,128,n]};  unit(level): encoding bins per map pixel at that level;
    background(level, k, start): [S... ] -> log-background tensor [B or 1,1,250,250] for strand k;
    zoom(level, start, reverse) -> 0..125 (or one such callable per strand: strands of different windows).  Returns (preds[level_idx] [S*B,1,250,250], starts[k][level_idx])."""
    S = len(reverse_flags)
    starts = [[0] for _ in range(S)]
    zoom_idx = [0] * S
    preds = []
    for j, level in enumerate(levels):
        u = unit(level)
        sl = [int(starts[k][j] / u) for k in range(S)]
        bgs = [background(level, k, starts[k][j]) for k in range(S)]
        dec = model.denets[level]
        if hasattr(dec, "forward_rows") and encodings[level].is_cuda:
            # orca_amd Decoders take the batch row by row: every strand's crops stay views (no torch.cat on the path)
            xs = _rows(encodings[level], B, sl, 250, (2,))
            des = [bgs[k][b if bgs[k].shape[0] > 1 else 0] for k in range(S) for b in range(B)]
            ys = _rows(preds[j - 1], B, zoom_idx, 125, (2, 3)) if j > 0 else None
            pred = dec.forward_rows(xs, des, ys)
            if level == add_1m_level:
                model.denet_1_pt.forward_rows_into(pred, xs, accumulate=True)     # fused `+ denet_1_pt(x)`
            preds.append(pred)
        else:
            enc = _gather(encodings[level], B, sl, 250, (2,))
            if all(b is bgs[0] for b in bgs):
                distenc = bgs[0].expand(S * B, -1, -1, -1)
            else:
                distenc = torch.cat([b.expand(B, -1, -1, -1) for b in bgs], dim=0)
            coarse = _gather(preds[j - 1], B, zoom_idx, 125, (2, 3)) if j > 0 else None
            preds.append(_decode(model, level, enc, distenc, coarse, level == add_1m_level))
        if on_level is not None:
            on_level(j, level, [starts[k][j] for k in range(S)])
        for k in range(S):
            zoom_idx[k] = (zoom[k] if isinstance(zoom, (list, tuple)) else zoom)(level, starts[k][j], reverse_flags[k])
            starts[k].append(starts[k][j] + zoom_idx[k] * u)
    return preds, [st[:-1] for st in starts]


def cascade_32m(model, xs, mpos, wpos, reverse_flags, distencs=None, merge=False):
    """32 Mb model on device-resident strands: ``xs`` = list of [B,4,L] tensors (e.g. forward strand and reverse
    complement), ``reverse_flags`` the matching booleans.  net0 per strand, then Encoder2 and the six decoder
    levels batched over the strands.  Returns (preds[6] each [S*B,1,250,250], starts[k][6] in 4 kb bins).
    A strand may also be given as packed bases - a uint8 [B,L] tensor of the FORWARD strand (engine.pack_sequence);
    its flag then also tells the Encoder to read it as the reverse complement (no second copy exists).
    ``merge=True`` (two strands): also returns the strand-merged maps (`_merge_device`), enqueued before the guard's sync."""
    B = xs[0].shape[0]
    cache = {}

    def background(level, k, start):
        if distencs is not None:
            return distencs[level]
        if level not in cache:
            cache[level] = _cached_log_background(model, level, xs[0].is_cuda)
        return cache[level]

    def forward():
        def encode(x, rev, out=None):
            return model.net0.forward_codes(x, reverse=rev, out=out) if x.dtype == torch.uint8 else model.net0(x)
        if len(xs) > 1 and all(x.dtype == torch.uint8 for x in xs) and hasattr(model.net0, "_net"):
            # packed strands: each is encoded straight into its rows of one [S*B,128,bins] tensor
            enc0 = torch.empty((len(xs) * B, 128, engine.encoder_num_bins(xs[0].shape[1])), dtype=torch.float32, device=xs[0].device)
            if len(xs) == 2 and xs[0] is xs[1] and list(reverse_flags) == [False, True]:
                _encode_two_strands(lambda rev, out: encode(xs[0], rev, out), enc0, B)
            else:
                for k, (x, r) in enumerate(zip(xs, reverse_flags)):
                    encode(x, r, enc0[k * B:(k + 1) * B])
        else:
            enc0 = torch.cat([encode(x, r) for x, r in zip(xs, reverse_flags)], dim=0) if len(xs) > 1 else encode(xs[0], reverse_flags[0])
        encodings = dict(zip([1, 2, 4, 8, 16, 32], model.net(enc0)))
        preds, starts = run_cascade(model, encodings, [32, 16, 8, 4, 2, 1], lambda lv: lv, B, list(reverse_flags), background,
                                    lambda lv, st, rev: zoom_index_32m(lv, st, mpos, wpos, rev), add_1m_level=1)
        return (preds, starts, _merge_device(preds, B)) if merge else (preds, starts)

    return engine.run_with_overflow_retry(forward, xs[0].device)


def starts_32m(mpos, wpos, reverse):
    """Window starts (4 kb bins) of the six levels for one strand: the zoom path depends on the coordinates only
    (orca_predict.py:471-500), not on the predictions."""
    st = [0]
    for level in [32, 16, 8, 4, 2, 1]:
        st.append(st[-1] + zoom_index_32m(level, st[-1], mpos, wpos, reverse) * level)
    return st[:-1]


def denet1m_32m_from_enc(model, enc0, mpos, wpos, reverse_flags):
    """The `+ model.denet_1_pt.forward(...)` term of the 4 kb level ALONE (orca_predict.py:362), for the strands of ``enc0``
    [S*B,128,8000]: it reads the level-1 encoding at the strand's last window and nothing of the decoder cascade, so another rank can
    compute it (dist.strand_bin_sharded_32m from 4 ranks).  Returns [S*B,C,250,250]."""
    S = len(reverse_flags)
    B = enc0.shape[0] // S

    def forward():
        enc1 = model.net(enc0)[0]
        xs = []
        for k in range(S):
            s = int(starts_32m(mpos, wpos, bool(reverse_flags[k]))[5])
            xs.append(enc1[k * B: (k + 1) * B, :, s: s + 250])
        return model.denet_1_pt(torch.cat(xs, dim=0).contiguous())

    return engine.run_with_overflow_retry(forward, enc0.device)


def cascade_32m_from_enc(model, enc0, mpos, wpos, reverse_flags, distencs=None, with_1m=True):
    """The part of `cascade_32m` AFTER the Encoder: ``enc0`` [S*B,128,8000] (strand k = rows k*B..) -> Encoder2 -> six decoder
    levels (+ denet_1_pt at 4 kb unless ``with_1m`` is False: `denet1m_32m_from_enc` then supplies that term).  Returns
    (preds[6], starts[k][6]).  (Multi-GPU: every rank holds the gathered encodings and runs the strands it owns,
    dist.strand_bin_sharded_32m.)"""
    S = len(reverse_flags)
    B = enc0.shape[0] // S
    cache = {}

    def background(level, k, start):
        if distencs is not None:
            return distencs[level]
        if level not in cache:
            cache[level] = _cached_log_background(model, level, enc0.is_cuda)
        return cache[level]

    def forward():
        encodings = dict(zip([1, 2, 4, 8, 16, 32], model.net(enc0)))
        return run_cascade(model, encodings, [32, 16, 8, 4, 2, 1], lambda lv: lv, B, [bool(r) for r in reverse_flags], background,
                           lambda lv, st, rev: zoom_index_32m(lv, st, mpos, wpos, rev), add_1m_level=1 if with_1m else None)

    return engine.run_with_overflow_retry(forward, enc0.device)


def cascade_256m(model, enc0, mpos, wpos, chrlen, normmat, reverse_flags=(False, True)):
    """Device part of genomepredict_256Mb AFTER the Encoder (orca_predict.py:675-838): ``enc0`` [S*B,128,64000] (strand k =
    rows k*B..) -> net1 -> [-1] -> net -> the four decoder levels, numerically the same as the full call: every strand's
    background is the block mean of ``normmat`` at THAT strand's window start, flipped on the reverse strand (:703, :724-737).
    ``normmat``: the 8000 x 8000 background (host numpy array, or a float64 ROCm tensor for device-side block means), or a
    ready `Background256` (its ``reverse_flags`` must be the ones given here).  Returns (preds[4], starts[k][4])."""
    S = len(reverse_flags)
    B = enc0.shape[0] // S
    bg = normmat if isinstance(normmat, Background256) else Background256(normmat, reverse_flags, enc0.is_cuda)
    if list(bg.reverse_flags) != [bool(r) for r in reverse_flags]:
        raise ValueError("cascade_256m: the Background256 was built for other strands")

    def forward():
        bg.reset()
        encodings = dict(zip([32, 64, 128, 256], model.net(model.net1(enc0)[-1])))
        return run_cascade(model, encodings, [256, 128, 64, 32], lambda lv: lv // 8, B, [bool(r) for r in reverse_flags],
                           bg, lambda lv, st, rev: zoom_index_256m(lv, st, mpos, wpos, chrlen, rev))

    return engine.run_with_overflow_retry(forward, enc0.device)


def _merge(preds, B):
    """0.5*fwd + 0.5*rev[::-1,::-1], batch row 0 only (orca_predict.py:510-523).  ``preds``: per level a
    [2B,C,250,250] tensor, forward strand in rows 0..B-1 and reverse strand in rows B..2B-1."""
    merged = []
    for p in preds:
        fwd, rev = p[0], p[B]
        if p.is_cuda and p.shape[1] == 1:
            merged.append(engine.strand_merge(fwd[0], rev[0]).cpu().numpy())
            continue
        f, r = fwd.cpu().detach().numpy(), rev.cpu().detach().numpy()
        if f.shape[0] == 1:
            merged.append(f[0, :, :] * 0.5 + r[0, ::-1, ::-1] * 0.5)
        else:
            merged.append(f[:, :, :] * 0.5 + r[:, ::-1, ::-1] * 0.5)
    return merged


def _merge_device(preds, B):
    """`_merge` on the MI355X, without leaving it: per level a [C,250,250] tensor.  Called INSIDE the guarded forward, i.e.
    enqueued before the host waits for the fp16-range flag (the merges used to start 0.6 ms after the last Decoder: flag
    read-back, host wake-up, then six launches each followed by its own blocking copy)."""
    merged = []
    for p in preds:
        fwd, rev = p[0], p[B]
        if p.shape[1] == 1:
            merged.append(engine.strand_merge(fwd[0], rev[0])[None])
        else:
            merged.append(fwd * 0.5 + torch.flip(rev, [1, 2]) * 0.5)
    return merged


def _merged_to_host(merged):
    """One device -> host copy for all levels; [250,250] per level for single-target models, [C,250,250] otherwise (:510-523)."""
    a = torch.stack(merged).cpu().numpy()
    return [m[0] if m.shape[0] == 1 else m for m in a]


def genomepredict(sequence, mchr, mpos=-1, wpos=-1, models=["h1esc", "hff"], targets=None, annotation=None,
                  use_cuda=True, nan_thresh=1):
    """Multiscale prediction for a 32 Mb sequence (orca_predict.py:231-540): six maps
    (32, 16, 8, 4, 2, 1 Mb windows, 250x250 each) zooming into ``mpos``.
    ``sequence``: float array [1, 32000000, 4]; ``wpos``: window centre coordinate."""
    models = _resolve_models(models, "32M", use_cuda)
    levels = [32, 16, 8, 4, 2, 1]
    batch = sequence.shape[0]   # float [B,L,4] (reference) or packed uint8 [B,L] codes on the device (extension)
    predictions, allstarts, alltargets, allannos = [], [], [], []
    with torch.no_grad():
        strands = _StrandInputs(sequence, use_cuda)     # forward strand + reverse complement
        for ii, model in enumerate(models):
            ts, annos = [], []

            def on_level(j, level, starts_now, model=model, ii=ii, ts=ts, annos=annos):
                s0 = starts_now[0]                      # bookkeeping follows the forward strand only
                if targets:
                    tgt = targets[ii]
                    tgt = tgt.numpy() if isinstance(tgt, torch.Tensor) else np.asarray(tgt)
                    w = 250 * level
                    tr = _coarse_grain(tgt[:, s0: s0 + w, s0: s0 + w], level, nan_thresh)
                    lf = np.log((tr + model.epss[level]) / (model.normmats[level] + model.epss[level]))
                    ts.append(lf[0, :, :] if tr.shape[0] == 1 else lf)
                if annotation is not None:
                    annos.append(_scale_annotation(annotation, s0 / 8000.0, (s0 + 250 * level) / 8000.0))

            bg_cache = {}

            def background(level, k, start, model=model, bg_cache=bg_cache):
                if level not in bg_cache:
                    bg_cache[level] = _cached_log_background(model, level, use_cuda)
                return bg_cache[level]

            def forward(model=model, background=background, on_level=on_level, ts=ts, annos=annos):
                del ts[:], annos[:]
                enc0 = strands.encode(model.net0)
                encodings = dict(zip([1, 2, 4, 8, 16, 32], model.net(enc0)))
                preds, starts = run_cascade(model, encodings, levels, lambda lv: lv, batch, [False, True], background,
                                            lambda lv, st, rev: zoom_index_32m(lv, st, mpos, wpos, rev), add_1m_level=1,
                                            on_level=on_level)
                return preds, starts, (_merge_device(preds, batch) if preds[0].is_cuda else None)

            preds, starts, merged = engine.run_with_overflow_retry(forward, strands.device)
            predictions.append(_merged_to_host(merged) if merged is not None else _merge(preds, batch))
            allstarts.append(starts[0])
            if targets:
                alltargets.append(ts)
            if annotation is not None:
                allannos.append(annos)
    output = {"predictions": predictions}
    output["experiments"] = alltargets if targets else None
    output["start_coords"] = [wpos - 16000000 + s * 4000 for s in allstarts[0]]
    output["end_coords"] = [int(output["start_coords"][ii] + 32000000 / 2 ** (ii)) for ii in range(6)]
    output["chr"] = mchr
    output["annos"] = allannos[0] if annotation is not None else None
    output["normmats"] = [[model.normmats[ii] for ii in levels] for model in models]
    return output


def genomepredict_256Mb(sequence, mchr, normmats, chrlen, mpos=-1, wpos=-1, models=["h1esc_256m", "hff_256m"],
                        targets=None, annotation=None, padding_chr=None, use_cuda=True, nan_thresh=1):
    """Multiscale prediction for a 256 Mb sequence (orca_predict.py:543-878): four maps
    (256, 128, 64, 32 Mb).  ``normmats``: one 8000x8000 (32 kb bins) background per model;
    ``chrlen``: length of the (first) chromosome, bounding the zoom."""
    models = _resolve_models(models, "256M", use_cuda)
    levels = [256, 128, 64, 32]
    batch = sequence.shape[0]
    predictions, allstarts, allnormmats, allnormmats_rev, alltargets, allannos = [], [], [], [], [], []

    def zoom(level, start, reverse):
        return zoom_index_256m(level, start, mpos, wpos, chrlen, reverse)

    with torch.no_grad():
        strands = _StrandInputs(sequence, use_cuda)
        for ii, model in enumerate(models):
            bg = Background256(normmats[ii], [False, True], use_cuda)
            ts, annos = [], []

            def on_level(j, level, starts_now, ii=ii, bg=bg, ts=ts, annos=annos):
                s0, w = starts_now[0], 250 * (level // 8)
                if targets:
                    tgt = targets[ii]
                    tgt = tgt.numpy() if isinstance(tgt, torch.Tensor) else np.asarray(tgt)
                    tr = _coarse_grain(tgt[:, s0: s0 + w, s0: s0 + w], level // 8, nan_thresh)
                    nm0 = bg.means(0, level)
                    eps = np.nanmin(nm0)
                    lf = np.log((tr + eps) / (nm0 + eps))
                    ts.append(lf[0, :, :] if tr.shape[0] == 1 else lf)
                if annotation is not None:
                    annos.append(_scale_annotation(annotation, s0 / 8000.0, (s0 + w) / 8000.0))

            def forward(model=model, background=bg, on_level=on_level, ts=ts, annos=annos):
                del ts[:], annos[:]
                background.reset()
                enc0 = strands.encode(model.net0)
                encodings = dict(zip([32, 64, 128, 256], model.net(model.net1(enc0)[-1])))
                preds, starts = run_cascade(model, encodings, levels, lambda lv: lv // 8, batch, [False, True], background, zoom,
                                            add_1m_level=None, on_level=on_level)
                return preds, starts, (_merge_device(preds, batch) if preds[0].is_cuda else None)

            preds, starts, merged = engine.run_with_overflow_retry(forward, strands.device)
            predictions.append(_merged_to_host(merged) if merged is not None else _merge(preds, batch))
            allstarts.append(starts[0])
            allnormmats.append({lv: bg.means(0, lv) for lv in bg.ns[0]})
            allnormmats_rev.append({lv: bg.means(1, lv) for lv in bg.ns[1]})
            if targets:
                alltargets.append(ts)
            if annotation is not None:
                allannos.append(annos)
    output = {"predictions": predictions}
    output["experiments"] = alltargets if targets else None
    output["start_coords"] = [wpos - 128000000 + s * 32000 for s in allstarts[0]]
    output["end_coords"] = [np.fmin(int(output["start_coords"][ii] + 256000000 / 2 ** (ii)), chrlen) for ii in range(4)]
    output["annos"] = allannos[0] if annotation is not None else None
    output["chr"] = mchr
    output["padding_chr"] = padding_chr
    output["normmats"] = allnormmats + allnormmats_rev   # reference order: forward-strand dicts, then reverse-strand
    return output


# structural-variant drivers (orca_predict.py:983-3057) live in sv_drivers.py; re-exported under the reference's names
from .sv_drivers import (predict_sequence_string, process_custom, process_del, process_dup, process_ins, process_inv,  # noqa: E402,F401
                         process_region, process_seqstr, process_single_breakpoint)
