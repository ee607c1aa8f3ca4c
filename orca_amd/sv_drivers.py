"""Structural-variant drivers with the reference's signatures: `process_region / process_dup /
process_del / process_inv / process_ins / process_custom / process_single_breakpoint / process_seqstr`
(/root/reference/orca_predict.py:983-3161), for the 32 Mb (`window_radius=16000000`) and the 256 Mb models.

Each driver is a list of VIEWS of a reference or mutated chromosome: (allele, anchor position, chromosome length
used for window clipping, annotation).  A view is materialised as the 32 Mb window sequence and handed to
`genomepredict`; the result dictionaries are the ones `genomepredict` returns, in the reference's order.

What differs from the reference is where the window sequence lives.  The reference concatenates 512 MB float one-hot
arrays on the host (`genome.get_encoding_from_coords` per piece, `[::-1, ::-1]` for '-' pieces).  With a
`orca_amd.genome.PackedGenome` resident on the MI355X (`genome.to("cuda")`) the pieces are gathered as 1-byte base
codes on the device and go straight into the Encoder's packed-input path - no float window ever exists.  Any other
genome object with the selene API (`get_chr_lens`, `get_encoding_from_coords`) takes the reference's host route.

`window_radius=128000000` selects the 256 Mb models: the whole (mutated) chromosome, rounded down to 32 kb and padded
with `padding_chr` to 256 Mb, goes through `genomepredict_256Mb` together with a distance background assembled per
region pair by `_retrieve_multi` (`orca_predict.py:881-980`).  Not offered at 256 Mb, because the reference itself
cannot run them: `process_ins` (its alt.r view is assigned to the wrong name and inserted pieces are looked up in the
genome, `:2412-2490`) and `process_custom` (always calls the 32 Mb `genomepredict`, `:2660`).

Out of scope here: plotting (`file=...` must stay None; the reference's `process_dup` even REQUIRES a file name at
256 Mb, `:1365-1367`) and the bundled Micro-C targets (`target=True` behaves as it does in the reference when the
resources are not loaded: no targets).
"""
import numpy as np
import torch

from . import genome as _genome
from .orca_utils import StructuralChange2, coord_clip, coord_round, process_anno

_R32 = 16000000
_R256 = 128000000
_BIN256 = 32000


def _setup(custom_models, window_radius, model_labels, file, allow_256=True):
    if file is not None:
        raise NotImplementedError("plotting (file=...) is not part of orca_amd; plot the returned dicts with the reference's genomeplot")
    if window_radius not in (_R32, _R256):
        raise ValueError("Only window_radius 16000000 (32Mb models) or 128000000 (256Mb models) are supported")
    if window_radius == _R256 and not allow_256:
        raise NotImplementedError("this driver has no working 256 Mb form in the reference (see orca_amd/sv_drivers.py)")
    if custom_models is None:
        return (["h1esc", "hff"] if window_radius == _R32 else ["h1esc_256m", "hff_256m"]), ["H1-ESC", "HFF"]
    if model_labels is None:
        model_labels = [f"Model {i}" for i in range(len(custom_models))]
    return custom_models, model_labels


def _targets(target):
    """`target=True` asks for the bundled Micro-C datasets, which do not exist offline: as in the reference when
    `target_dict_global` lacks them (`orca_predict.py:1232-1241`), fall back to no targets."""
    if not target or target is True:
        return False
    if any(not hasattr(t, "get_feature_data") for t in target):
        return False
    return list(target)


def _target_windows(target, chrom, w0, w1):
    if not target:
        return None
    return [torch.FloatTensor(t.get_feature_data(chrom, coord_round(w0), coord_round(w1))[None, :]) for t in target]


def _chrlen(genome, chrom):
    return [l for c, l in genome.get_chr_lens() if c == chrom].pop()


def _on_device(genome, use_cuda):
    return use_cuda and getattr(genome, "device", None) is not None and hasattr(genome, "get_codes_from_coords")


def _assemble(genome, pieces, use_cuda, ins_seq=None, pad_to=None):
    """Window sequence from reference pieces `(chrom, start, end, strand)`: packed codes `[1, L]` on the device when
    the genome lives there, else float32 `[1, L, 4]` on the host (`orca_predict.py:1428-1436`).  Pieces of an
    inserted sequence (chromosome name 'ins...') are cut from `ins_seq`; `pad_to` appends 'N' rows."""
    if _on_device(genome, use_cuda):
        parts = []
        for chrom, start, end, strand in pieces:
            if ins_seq is not None and chrom.startswith("ins"):
                codes = torch.from_numpy(_genome.sequence_to_codes(ins_seq[start:end])).to(genome.device)
            else:
                codes = genome.get_codes_from_coords(chrom, start, end)
            parts.append(_genome.revcomp_codes(codes) if strand == "-" else codes)
        n = sum(int(p.shape[0]) for p in parts)
        if pad_to and n < pad_to:
            parts.append(torch.full((pad_to - n,), _genome.N_CODE, dtype=torch.uint8, device=genome.device))
        return torch.cat(parts)[None, :]
    parts = []
    for chrom, start, end, strand in pieces:
        if ins_seq is not None and chrom.startswith("ins"):
            seq = _genome.sequence_to_encoding(ins_seq[start:end])
        else:
            seq = genome.get_encoding_from_coords(chrom, start, end)
        parts.append(seq[::-1, ::-1] if strand == "-" else seq)
    seq = np.concatenate(parts, axis=0)
    if pad_to and seq.shape[0] < pad_to:
        seq = np.concatenate((seq, np.full((pad_to - seq.shape[0], 4), 0.25, dtype=seq.dtype)), axis=0)
    return seq[None, :, :]


def _predict(sequence, mchr, mpos, wpos, models, annotation, targets, use_cuda):
    from .orca_predict import genomepredict
    return genomepredict(sequence, mchr, mpos, wpos, models=models, annotation=annotation, targets=targets, use_cuda=use_cuda)


class _View:
    """One 32 Mb window a driver wants predicted: reference pieces `(chrom, start, end, strand)`, the `genomepredict` arguments that go
    with them, and what assembling needs (`ins_seq`, `pad_to`).  A driver first LISTS its views, then `_run_views` predicts them."""

    def __init__(self, pieces, label, mpos, wpos, annotation, targets=None, ins_seq=None, pad_to=None):
        self.pieces, self.label, self.mpos, self.wpos = [tuple(p) for p in pieces], label, mpos, wpos
        self.annotation, self.targets, self.ins_seq, self.pad_to = annotation, targets, ins_seq, pad_to


def _ref_view(genome, chrom, anchor, anno_regions, models, target, use_cuda, chrlen=None):
    """Reference-allele window clipped around `anchor`; `anno_regions(w0, w1)` builds the unscaled annotation."""
    chrlen = _chrlen(genome, chrom) if chrlen is None else chrlen
    wpos = coord_clip(anchor, chrlen)
    w0, w1 = wpos - _R32, wpos + _R32
    anno = process_anno(anno_regions(w0, w1), base=w0, window_radius=_R32)
    return _View([(chrom, w0, w1, "+")], chrom, anchor, wpos, anno, _target_windows(target, chrom, w0, w1))


def _alt_view(genome, sc, label, anchor, chrlen_alt, anno_regions, models, use_cuda, ins_seq=None):
    wpos = coord_clip(anchor, chrlen_alt)
    w0, w1 = wpos - _R32, wpos + _R32
    anno = process_anno(anno_regions(w0, w1), base=w0, window_radius=_R32)
    return _View(sc[w0:w1], label, anchor, wpos, anno, None, ins_seq=ins_seq)


# ---- running a driver's views ---------------------------------------------------------------------------------------------------
# The reference pushes every view through the whole Encoder (orca_predict.py:1335, :1389, :1484 ...: three or four `genomepredict` calls
# per driver call, each encoding 2 x 32 Mb).  With the genome resident on the MI355X and orca_amd containers the views of a call are run
# TOGETHER (round 5): the Encoder is translation-covariant on the 4 kb grid with a reach of 104 016 bases
# (orca_modules.py:811-927), so
#   * the reference views are encoded first and their Encoder outputs kept as SEGMENTS of the chromosome (`sv.ChromEncodings`, both strands);
#   * an alternative allele is pieces of the reference: every bin whose receptive field lies inside one piece is copied from a segment of
#     the same 4 kb phase - a '-' piece from the OTHER strand's segment (an inversion anchored at its left end takes the inverted bins from
#     the reverse strand of the view anchored at its right end: that is where the phases agree) - and only the bins at window ends and at
#     junctions go through the Encoder's bin-range form (on a pool of auxiliary contexts);
#   * segments stay (per thread, genome and Encoder; LRU-bounded) for later calls, and a (chromosome, strand, phase) that keeps being
#     asked for is encoded ONCE as a whole (`ChromEncodings.auto_threshold`): from then on a call at that phase - coordinates on one 4 kb
#     grid, as a screen's usually are - encodes 1-2 % of its bins;
#   * Encoder2 and every decoder level run once per call and model for ALL views' strands as one batch (6-8 maps per launch).
# The maps are the whole-window route's to a few 1e-6 (other tile instantiations in the short calls; tests/test_gpu_e2e.py against the
# REFERENCE's process_dup / process_inv / process_del with its real networks: G22, G23).  ORCA_SV_INCREMENTAL=0 (read per call): every view
# through `genomepredict`, as in rounds 1-4.
import threading as _threading

_tls = _threading.local()
_MAX_STORES = 6


def clear_encoding_cache():
    """Drop the Encoder outputs the drivers keep for reuse (this thread's)."""
    _tls.stores = {}


def _store(genome, net0):
    from . import sv
    stores = getattr(_tls, "stores", None)
    if stores is None:
        stores = _tls.stores = {}
    # (the Encoder's arithmetic mode and the version of its weights are part of the key: kept encodings must not outlive a `precision = ...`
    # or a `load_state_dict` on the same module object)
    key = (id(genome), id(net0), getattr(net0, "precision", None), getattr(net0, "_wver", 0))
    for k in [k for k, h in stores.items() if h[0]() is None or h[1]() is None]:
        del stores[k]                                        # the genome or the Encoder is gone: its encodings (and the HBM they hold) go too
    hit = stores.pop(key, None)
    if hit is not None and (hit[0]() is not genome or hit[1]() is not net0):
        hit = None                                           # an id reused by another object
    if hit is None:
        import weakref
        chrlens = dict(genome.get_chr_lens())
        wg = weakref.ref(genome)
        # the store must not keep what it is keyed on alive - the GenomeEncodings sees the Encoder through a weak proxy and the
        # genome through a weak reference (a deleted genome's GBs of HBM, its unpacked chromosomes and segments are released with it)
        hit = (wg, weakref.ref(net0), sv.GenomeEncodings(weakref.proxy(net0), lambda c: wg().get_codes_from_coords(c, 0, chrlens[c]), chrlens))
    stores[key] = hit                                        # most recently used last
    while len(stores) > _MAX_STORES:
        stores.pop(next(iter(stores)))
    return hit[2]


def _incremental_ok(genome, models, use_cuda):
    import os
    if os.environ.get("ORCA_SV_INCREMENTAL", "1") == "0" or not _on_device(genome, use_cuda):
        return False
    return all(hasattr(getattr(m, "net0", None), "forward_codes") and hasattr(m, "denets") and hasattr(m.denets.get(1, None), "forward_rows")
               for m in models)


def _bookkeeping(model, ii, starts, view, nan_thresh=1):
    """`genomepredict`'s per-level targets and annotations (orca_predict.py:404-468) from the forward strand's window starts."""
    from .orca_predict import _coarse_grain, _scale_annotation
    ts, annos = [], []
    for level, s0 in zip([32, 16, 8, 4, 2, 1], starts):
        if view.targets:
            tgt = view.targets[ii]
            tgt = tgt.numpy() if isinstance(tgt, torch.Tensor) else np.asarray(tgt)
            w = 250 * level
            tr = _coarse_grain(tgt[:, s0: s0 + w, s0: s0 + w], level, nan_thresh)
            lf = np.log((tr + model.epss[level]) / (model.normmats[level] + model.epss[level]))
            ts.append(lf[0, :, :] if tr.shape[0] == 1 else lf)
        if view.annotation is not None:
            annos.append(_scale_annotation(view.annotation, s0 / 8000.0, (s0 + 250 * level) / 8000.0))
    return ts, annos


def _run_views(genome, views, models, use_cuda, stats=None):
    """`genomepredict` output dicts of a driver's views, in order."""
    from . import engine, orca_predict, sv
    if not (use_cuda and views) or not _on_device(genome, use_cuda):
        return [_predict(_assemble(genome, v.pieces, use_cuda, ins_seq=v.ins_seq, pad_to=v.pad_to), v.label, v.mpos, v.wpos, models,
                         v.annotation, v.targets, use_cuda) for v in views]
    resolved = orca_predict._resolve_models(models, "32M", use_cuda)
    if not _incremental_ok(genome, resolved, use_cuda):
        return [_predict(_assemble(genome, v.pieces, use_cuda, ins_seq=v.ins_seq, pad_to=v.pad_to), v.label, v.mpos, v.wpos, models,
                         v.annotation, v.targets, use_cuda) for v in views]
    dev = genome.device
    W, nb = len(views), 2 * _R32 // sv.BIN
    levels = [32, 16, 8, 4, 2, 1]
    with torch.no_grad():
        codes = torch.cat([_assemble(genome, v.pieces, True, ins_seq=v.ins_seq, pad_to=v.pad_to) for v in views])      # [W, L]
        pieces = []
        for v in views:
            p4 = [(c, s, e - s, st) for c, s, e, st in v.pieces]
            n = sum(p[2] for p in p4)
            if n < 2 * _R32:
                p4.append(("__pad__", 0, 2 * _R32 - n, "+"))
            pieces.append(p4)
        chrlens = dict(genome.get_chr_lens())
        single = [len(p) == 1 and p[0][3] == "+" and p[0][0] in chrlens for p in pieces]
        order = [w for w in range(W) if single[w]] + [w for w in range(W) if not single[w]]      # reference views first: the others reuse them
        pool = engine.context_pool(dev, 4)
        outs = [{"predictions": [], "experiments": [] if v.targets else None, "chr": v.label, "annos": None, "normmats": []} for v in views]
        for ii, model in enumerate(resolved):
            store = _store(genome, model.net0)
            box = {}

            def forward(model=model, store=store, box=box):
                enc0 = torch.empty((2 * W, 128, nb), dtype=torch.float32, device=dev)
                own, encoded = {}, 0              # this call's own segments: visible to its later views, committed to the store on success
                for w in order:
                    encoded += sv.encode_windows(store, [pieces[w]], codes[w: w + 1], enc0[2 * w: 2 * w + 2], build="auto", pool=pool,
                                                 extra=own, big_on_caller=True)
                    if single[w]:
                        chrom, start, ln, _ = pieces[w][0]
                        own.setdefault(chrom, []).extend([["+", start, enc0[2 * w].clone()], ["-", chrlens[chrom] - start - ln, enc0[2 * w + 1].clone()]])
                box["own"], box["encoded"] = own, encoded
                return sv._cascade_windows(model, enc0, [(v.mpos, v.wpos) for v in views])

            merged, starts = engine.run_with_overflow_retry(forward, dev, pool)
            for chrom, segs in box["own"].items():          # (after a range-safe retry these are the retry's)
                for seg in segs:
                    store.of(chrom).add_segment(*seg)
            if stats is not None:
                stats["bins_encoded"] = stats.get("bins_encoded", 0) + box["encoded"]
                stats["bins_total"] = stats.get("bins_total", 0) + 2 * W * nb
                stats["chromosome_encodings"] = store.builds
            host = merged.cpu().numpy()
            for w, v in enumerate(views):
                o = outs[w]
                o["predictions"].append([host[w, j, 0] if host.shape[2] == 1 else host[w, j] for j in range(6)])
                o["normmats"].append([model.normmats[lv] for lv in levels])
                ts, annos = _bookkeeping(model, ii, starts[2 * w], v)
                if v.targets:
                    o["experiments"].append(ts)
                if ii == 0:
                    o["start_coords"] = [v.wpos - _R32 + s * 4000 for s in starts[2 * w]]
                    o["end_coords"] = [int(o["start_coords"][k] + 32000000 / 2 ** k) for k in range(6)]
                    o["annos"] = annos if v.annotation is not None else None
    keys = ["predictions", "experiments", "start_coords", "end_coords", "chr", "annos", "normmats"]       # genomepredict's key order
    return [{k: o[k] for k in keys} for o in outs]


def _left_anchored(mstart, mend, colour):
    """Region annotation of a window anchored at the variant's LEFT end: cut at the window's right edge."""
    return lambda w0, w1: [[mstart, mend if w1 > mend else w1, colour]]


def _right_anchored(mstart, mend, colour):
    return lambda w0, w1: [[mstart if w0 < mstart else w0, mend, colour]]


# ---- 256 Mb models -----------------------------------------------------------------------------------------
def _background_objects(models, normmat):
    """Objects carrying `background_cis` / `background_trans`.  The reference always takes the module-level default
    pair `h1esc_256m, hff_256m` (`orca_predict.py:938-941`), even next to custom models; here custom models that
    bring their own backgrounds are used as they are, anything else falls back to the registered default pair."""
    if isinstance(normmat, (list, tuple)):
        return list(normmat)
    from .orca_predict import _resolve_models
    if models is not None and all(hasattr(m, "background_cis") for m in models if not isinstance(m, str)):
        return _resolve_models(models, "256M", False)
    return _resolve_models(["h1esc_256m", "hff_256m"], "256M", False)


def _retrieve_multi(regionlist, genome, target=True, normmat=True, normmat_regionlist=None, use_cuda=False, models=None):
    """Sequence (+ per-model distance background, + targets) of a concatenation of regions `(chrom, start, end[,
    strand])` (`orca_predict.py:881-980`).  The background of a region pair on the same chromosome is the cis
    expectation at the pair's 32 kb-bin distances, flipped for '-' regions; pairs on different chromosomes get the
    scalar trans background."""
    regions = [tuple(r) if len(r) == 4 else (r[0], r[1], r[2], "+") for r in regionlist]
    # get_encoding_from_coords(strand='-') is already the reverse complement: regions are passed with their strand
    sequence = _assemble(genome, regions, use_cuda)
    out = (sequence,)
    def block_matrix(rl, block):
        """Square matrix over the 32 kb bins of the concatenated regions, filled block by block in place."""
        nbins = [int((end - start) / _BIN256) for _, start, end, _ in rl]
        offs = np.concatenate([[0], np.cumsum(nbins)])
        out = np.empty((offs[-1], offs[-1]), dtype=np.float64)
        for i, ra in enumerate(rl):
            for j, rb in enumerate(rl):
                blk = block(ra, rb, nbins[i], nbins[j])
                if ra[3] == "-":
                    blk = blk[::-1, :]
                if rb[3] == "-":
                    blk = blk[:, ::-1]
                out[offs[i]:offs[i + 1], offs[j]:offs[j + 1]] = blk
        return out

    if normmat:
        nrl = [tuple(r) for r in (regions if normmat_regionlist is None else normmat_regionlist)]
        normmats = []
        for obj in _background_objects(models, normmat):
            def background(ra, rb, na, nb, obj=obj):
                if ra[0] != rb[0]:
                    return np.full((na, nb), obj.background_trans)
                if (ra[2] - ra[1]) == na * _BIN256 and (rb[2] - rb[1]) == nb * _BIN256:
                    # bins are exactly 32 kb apart, so the block is Toeplitz in i - j: a strided VIEW of one diagonal
                    # profile instead of an [na, nb] gather (bit-identical to the general formula below)
                    k = np.arange(-(nb - 1), na, dtype=np.int64)
                    prof = np.ascontiguousarray(obj.background_cis[np.abs((ra[1] - rb[1]) + _BIN256 * k) // _BIN256])
                    st = prof.strides[0]
                    return np.lib.stride_tricks.as_strided(prof[nb - 1:], shape=(na, nb), strides=(st, -st), writeable=False)
                acoor = np.linspace(ra[1], ra[2], na + 1)[:-1]
                bcoor = np.linspace(rb[1], rb[2], nb + 1)[:-1]
                return obj.background_cis[(np.abs(acoor[:, None] - bcoor[None, :]) / _BIN256).astype(int)]
            normmats.append(block_matrix(nrl, background))
        out = out + (normmats,)
    if isinstance(target, (list, tuple)) and target:
        targets = []
        for t in target:
            def observed(ra, rb, na, nb, t=t):
                return t.get_feature_data(ra[0], ra[1], ra[2], chrom2=rb[0], start2=rb[1], end2=rb[2])
            targets.append(torch.from_numpy(block_matrix(regions, observed).astype(np.float32)[None, :, :]))
        out = out + (targets,)
    return out


def _predict256(sequence, mchr, normmats, chrlen, mpos, wpos, models, annotation, padding_chr, targets, use_cuda):
    from .orca_predict import genomepredict_256Mb
    return genomepredict_256Mb(sequence, mchr, normmats, chrlen, mpos, wpos, models=models, targets=targets,
                               annotation=annotation, padding_chr=padding_chr, use_cuda=use_cuda)


class _Chrom256:
    """The whole reference chromosome, 32 kb-rounded and padded to 256 Mb, fetched once and viewed at several anchors
    (the window is always the full 256 Mb: wpos = 128 000 000)."""

    def __init__(self, genome, chrom, padding_chr, target, models, use_cuda):
        self.chrom, self.padding_chr, self.models, self.use_cuda = chrom, padding_chr, models, use_cuda
        chrlen = _chrlen(genome, chrom)
        self.chrlen_round = chrlen - chrlen % _BIN256
        self.regions = [[chrom, 0, self.chrlen_round, "+"], [padding_chr, 0, 2 * _R256 - self.chrlen_round, "+"]]
        got = _retrieve_multi(self.regions, genome, target=target, use_cuda=use_cuda, models=models)
        self.sequence, self.normmats = got[0], got[1]
        self.targets = got[2] if len(got) > 2 else None

    def view(self, anchor, anno_regions, sequence=None, targets="ref"):
        anno = process_anno(anno_regions(0, 2 * _R256), base=0, window_radius=_R256)
        return _predict256(self.sequence if sequence is None else sequence, self.chrom, self.normmats, self.chrlen_round,
                           anchor, _R256, self.models, anno, self.padding_chr, self.targets if targets == "ref" else None,
                           self.use_cuda)


def _alt_view_256(genome, sc, mchr, anchor, chrlen_alt, anno_regions, models, padding_chr, use_cuda):
    """Mutated chromosome at 256 Mb: whole chromosome + padding while it fits, else a clipped 256 Mb window of it
    (`orca_predict.py:1438-1460`)."""
    alt_round = chrlen_alt - chrlen_alt % _BIN256
    if alt_round < 2 * _R256:
        wpos = _R256
        seq, normmats = _retrieve_multi(list(sc[0:alt_round]) + [[padding_chr, 0, 2 * _R256 - alt_round, "+"]], genome,
                                        target=False, normmat=True, use_cuda=use_cuda, models=models,
                                        normmat_regionlist=[[mchr, 0, alt_round, "+"], [padding_chr, 0, 2 * _R256 - alt_round, "+"]])
    else:
        wpos = coord_clip(anchor, alt_round, window_radius=_R256)
        seq, normmats = _retrieve_multi(list(sc[wpos - _R256: wpos + _R256]), genome, target=False, normmat=True,
                                        use_cuda=use_cuda, models=models,
                                        normmat_regionlist=[[mchr, wpos - _R256, wpos + _R256, "+"]])
    anno = process_anno(anno_regions(wpos - _R256, wpos + _R256), base=wpos - _R256, window_radius=_R256)
    return _predict256(seq, mchr, normmats, alt_round, anchor, wpos, models, anno, padding_chr, None, use_cuda)


def _shares_encodings(fn):
    """A driver call runs inside `orca_predict.shared_encodings()`: at 256 Mb the same packed sequence predicted at two anchors is encoded once."""
    import functools

    @functools.wraps(fn)
    def wrapped(*a, **k):
        from .orca_predict import shared_encodings
        with shared_encodings():
            return fn(*a, **k)
    return wrapped


@_shares_encodings
def process_region(mchr, mstart, mend, genome, file=None, custom_models=None, target=True, show_genes=True, show_tracks=False,
                   window_radius=16000000, padding_chr="chr1", model_labels=None, use_cuda=True):
    """Multiscale prediction centred on a region (`orca_predict.py:983-1169`)."""
    models, _ = _setup(custom_models, window_radius, model_labels, file)
    target = _targets(target)
    mpos = int((int(mstart) + int(mend)) / 2)
    anno = lambda w0, w1: [[np.clip(mstart, w0, w1), np.clip(mend, w0, w1), "black"]]
    if window_radius == _R256:
        return _Chrom256(genome, mchr, padding_chr, target, models, use_cuda).view(mpos, anno)
    return _run_views(genome, [_ref_view(genome, mchr, mpos, anno, models, target, use_cuda)], models, use_cuda)[0]


def _ref_pair(genome, mchr, mstart, mend, models, target, use_cuda, window_radius=_R32, padding_chr=None):
    """The two reference views every interval variant starts with: anchored at its left and at its right end.
    Returns (ref_l, ref_r, whole-chromosome handle) - at 256 Mb the two output dicts, at 32 Mb two `_View`s (and None) for `_run_views`."""
    if window_radius == _R256:
        ref = _Chrom256(genome, mchr, padding_chr, target, models, use_cuda)
        return ref.view(mstart, _left_anchored(mstart, mend, "black")), ref.view(mend, _right_anchored(mstart, mend, "black")), ref
    ref_l = _ref_view(genome, mchr, mstart, _left_anchored(mstart, mend, "black"), models, target, use_cuda)
    ref_r = _ref_view(genome, mchr, mend, _right_anchored(mstart, mend, "black"), models, target, use_cuda)
    return ref_l, ref_r, None


@_shares_encodings
def process_dup(mchr, mstart, mend, genome, file=None, custom_models=None, target=True, show_genes=True, show_tracks=False,
                window_radius=16000000, padding_chr="chr1", model_labels=None, use_cuda=True):
    """Tandem duplication of [mstart, mend) (`orca_predict.py:1172-1507`): ref.l, ref.r, alt (anchored at the
    junction `mend` between the two copies; original copy black, new copy gray)."""
    models, _ = _setup(custom_models, window_radius, model_labels, file)
    target = _targets(target)
    chrlen = _chrlen(genome, mchr)
    ref_l, ref_r, _ = _ref_pair(genome, mchr, mstart, mend, models, target, use_cuda, window_radius, padding_chr)
    sc = StructuralChange2(mchr, chrlen)
    sc.duplicate(mstart, mend)
    copy_end = mend + mend - mstart

    def anno(w0, w1):
        return [[mstart if w0 < mstart else w0, mend, "black"], [mend, copy_end if copy_end < w1 else w1, "gray"]]

    if window_radius == _R256:
        return ref_l, ref_r, _alt_view_256(genome, sc, mchr, mend, chrlen + mend - mstart, anno, models, padding_chr, use_cuda)
    alt = _alt_view(genome, sc, mchr, mend, chrlen + mend - mstart, anno, models, use_cuda)
    return tuple(_run_views(genome, [ref_l, ref_r, alt], models, use_cuda))


@_shares_encodings
def process_del(mchr, mstart, mend, genome, cmap=None, file=None, custom_models=None, target=True, show_genes=True,
                show_tracks=False, window_radius=16000000, padding_chr="chr1", model_labels=None, use_cuda=True):
    """Deletion of [mstart, mend) (`orca_predict.py:1510-1817`): ref.l, ref.r, alt at the breakpoint."""
    models, _ = _setup(custom_models, window_radius, model_labels, file)
    target = _targets(target)
    chrlen = _chrlen(genome, mchr)
    ref_l, ref_r, _ = _ref_pair(genome, mchr, mstart, mend, models, target, use_cuda, window_radius, padding_chr)
    sc = StructuralChange2(mchr, chrlen)
    sc.delete(mstart, mend)
    anno = lambda w0, w1: [[mstart, "double"]]
    if window_radius == _R256:
        return ref_l, ref_r, _alt_view_256(genome, sc, mchr, mstart, chrlen - (mend - mstart), anno, models, padding_chr, use_cuda)
    alt = _alt_view(genome, sc, mchr, mstart, chrlen - (mend - mstart), anno, models, use_cuda)
    return tuple(_run_views(genome, [ref_l, ref_r, alt], models, use_cuda))


@_shares_encodings
def process_inv(mchr, mstart, mend, genome, file=None, custom_models=None, target=True, show_genes=True, show_tracks=False,
                window_radius=16000000, padding_chr="chr1", model_labels=None, use_cuda=True):
    """Inversion of [mstart, mend) (`orca_predict.py:1820-2175`): ref.l, ref.r, alt.l, alt.r."""
    models, _ = _setup(custom_models, window_radius, model_labels, file)
    target = _targets(target)
    chrlen = _chrlen(genome, mchr)
    ref_l, ref_r, ref = _ref_pair(genome, mchr, mstart, mend, models, target, use_cuda, window_radius, padding_chr)
    sc = StructuralChange2(mchr, chrlen)
    sc.invert(mstart, mend)
    anno_l = lambda w0, w1: [[mstart, mend if mend < w1 else w1, "gray"]]
    anno_r = lambda w0, w1: [[mstart if mstart > w0 else w0, mend, "gray"]]
    if window_radius == _R256:   # an inversion changes neither the length nor the distance background (`:2017-2024`)
        (seq,) = _retrieve_multi(list(sc[0:ref.chrlen_round]) + [ref.regions[1]], genome, target=False, normmat=False, use_cuda=use_cuda)
        return ref_l, ref_r, ref.view(mstart, anno_l, sequence=seq, targets=None), ref.view(mend, anno_r, sequence=seq, targets=None)
    alt_l = _alt_view(genome, sc, mchr, mstart, chrlen, anno_l, models, use_cuda)
    alt_r = _alt_view(genome, sc, mchr, mend, chrlen, anno_r, models, use_cuda)
    return tuple(_run_views(genome, [ref_l, ref_r, alt_l, alt_r], models, use_cuda))


def process_ins(mchr, mpos, ins_seq, genome, strand="+", file=None, custom_models=None, target=True, show_genes=True,
                show_tracks=False, window_radius=16000000, padding_chr="chr1", model_labels=None, use_cuda=True):
    """Insertion of the string `ins_seq` at `mpos` (`orca_predict.py:2178-2497`): ref, alt.l, alt.r (anchored at the
    two ends of the insert).  The reference forgets `models=` in the alt.r call (`:2453`) and so always uses the
    default H1-ESC + HFF pair there; here alt.r uses the same models as the other views."""
    models, _ = _setup(custom_models, window_radius, model_labels, file, allow_256=False)
    target = _targets(target)
    chrlen = _chrlen(genome, mchr)
    n_ins = len(ins_seq)
    ref = _ref_view(genome, mchr, mpos, lambda w0, w1: [[mpos, "single"]], models, target, use_cuda, chrlen)
    sc = StructuralChange2(mchr, chrlen)
    sc.insert(mpos, n_ins, strand=strand)
    alt_l = _alt_view(genome, sc, mchr, mpos, chrlen + n_ins,
                      lambda w0, w1: [[mpos, mpos + n_ins if mpos + n_ins < w1 else w1, "gray"]], models, use_cuda, ins_seq=ins_seq)
    alt_r = _alt_view(genome, sc, mchr, mpos + n_ins, chrlen + n_ins,
                      lambda w0, w1: [[mpos if mpos > w0 else w0, mpos + n_ins, "gray"]], models, use_cuda, ins_seq=ins_seq)
    return tuple(_run_views(genome, [ref, alt_l, alt_r], models, use_cuda))


def process_custom(region_list, ref_region_list, mpos, genome, ref_mpos_list=None, anno_list=None, ref_anno_list=None,
                   custom_models=None, target=True, file=None, show_genes=True, show_tracks=False, window_radius=16000000,
                   model_labels=None, use_cuda=True):
    """Arbitrary rearrangement given as the list of reference pieces that make up the 32 Mb alternative window
    (`orca_predict.py:2500-2681`).  Like the reference, returns the LAST reference view and the alternative view."""
    models, _ = _setup(custom_models, window_radius, model_labels, file, allow_256=False)
    target = _targets(target)

    def validate(regions, enforce_strand=None):
        total = 0
        for chrm, start, end, strand in regions:
            assert start >= 0 and end <= _chrlen(genome, chrm)
            total += end - start
            if enforce_strand and strand != enforce_strand:
                raise ValueError("The specified strand must be " + enforce_strand)
        assert total == 2 * window_radius

    validate(region_list)
    views = []
    for i, ref_region in enumerate(ref_region_list):
        validate([ref_region], enforce_strand="+")
        chrm, start, end = ref_region[0], ref_region[1], ref_region[2]
        anno = process_anno(ref_anno_list, base=0, window_radius=window_radius)
        views.append(_View([(chrm, start, end, "+")], chrm, start + window_radius if ref_mpos_list is None else ref_mpos_list[i],
                           start + window_radius, anno, _target_windows(target, chrm, start, end)))
    anno = process_anno(anno_list, base=0, window_radius=window_radius)
    views.append(_View(region_list, "chimeric", mpos, window_radius, anno, None))
    outs = _run_views(genome, views, models, use_cuda)
    return (outs[-2] if len(outs) > 1 else None), outs[-1]


@_shares_encodings
def process_single_breakpoint(chr1, pos1, chr2, pos2, orientation1, orientation2, genome, custom_models=None, target=True,
                              file=None, show_genes=True, show_tracks=False, window_radius=16000000, padding_chr="chr1",
                              model_labels=None, use_cuda=True):
    """Simple translocation joining `chr1:pos1` and `chr2:pos2` (`orca_predict.py:2684-3057`): the two reference
    views and the fused chromosome around the breakpoint.  orientation1 '+' keeps chr1[0:pos1), '-' keeps
    chr1[pos1-1:] reverse-complemented; orientation2 '-' keeps chr2[pos2-1:], '+' keeps chr2[0:pos2) reversed."""
    models, _ = _setup(custom_models, window_radius, model_labels, file)
    target = _targets(target)
    len1, len2 = _chrlen(genome, chr1), _chrlen(genome, chr2)
    if window_radius == _R256:
        ref_1 = _Chrom256(genome, chr1, padding_chr, target, models, use_cuda).view(pos1, lambda w0, w1: [[pos1, "single"]])
        ref_2 = _Chrom256(genome, chr2, padding_chr, target, models, use_cuda).view(pos2, lambda w0, w1: [[pos2, "single"]])
    else:
        ref_1 = _ref_view(genome, chr1, pos1, lambda w0, w1: [[pos1, "single"]], models, target, use_cuda, len1)
        ref_2 = _ref_view(genome, chr2, pos2, lambda w0, w1: [[pos2, "single"]], models, target, use_cuda, len2)

    s = StructuralChange2(chr1, len1)
    if orientation1 == "+":
        s.delete(pos1, len1)
    else:
        s.delete(0, pos1 - 1)
        s.invert(0, len1 - pos1 + 1)
    s2 = StructuralChange2(chr2, len2)
    if orientation2 == "-":
        s2.delete(0, pos2 - 1)
    else:
        s2.delete(pos2, len2)
        s2.invert(0, pos2)
    breakpos = s.coord_points[-1]
    s = s + s2
    total = s.coord_points[-1]
    fused = chr1 + "|" + chr2
    if window_radius == _R256:
        alt_round = total - total % _BIN256
        if alt_round < 2 * _R256:
            wpos, pieces = _R256, s[0:alt_round]
            seq, normmats = _retrieve_multi(list(pieces) + [[padding_chr, 0, 2 * _R256 - alt_round, "+"]], genome, target=False,
                                            normmat=True, use_cuda=use_cuda, models=models,
                                            normmat_regionlist=[[fused, 0, alt_round, "+"], [padding_chr, 0, 2 * _R256 - alt_round, "+"]])
        else:
            wpos = coord_clip(breakpos, alt_round, window_radius=_R256)
            pieces = s[wpos - _R256: wpos + _R256]
            seq, normmats = _retrieve_multi(list(pieces), genome, target=False, normmat=True, use_cuda=use_cuda, models=models,
                                            normmat_regionlist=[[fused, wpos - _R256, wpos + _R256, "+"]])
        anno = process_anno([[pieces[0].end - pieces[0].start, "double"]], base=0, window_radius=_R256)
        alt = _predict256(seq, fused, normmats, alt_round, breakpos, wpos, models, anno, padding_chr, None, use_cuda)
        return ref_1, ref_2, alt
    if total < 2 * window_radius + 128000:   # fused chromosome shorter than a window (+ one coord_clip bin)
        radius = total // 2
        wpos = radius
    else:
        radius = window_radius
        wpos = coord_clip(breakpos, total, window_radius=radius)
    pieces = s[wpos - radius: wpos + radius]
    first_len = pieces[0].end - pieces[0].start
    n = sum(p.end - p.start for p in pieces)
    if n != 32000000:
        wpos = wpos + (32000000 - n) // 2
    anno = process_anno([[first_len, "double"]], base=0, window_radius=window_radius)
    alt = _View(pieces, fused, breakpos, wpos, anno, None, pad_to=32000000)
    return tuple(_run_views(genome, [ref_1, ref_2, alt], models, use_cuda))


def predict_sequence_string(sequence_str, mpos=None, models=("h1esc", "hff"), use_cuda=True):
    """The body of `process_seqstr` after the Seqstr lookup (orca_predict.py:3113-3148): the middle 32 Mb of a DNA string
    (ValueError below 32 Mb), window centre = its midpoint, zoom position ``mpos`` (default: the midpoint), chromosome
    label "customized seq", no targets.  With ``use_cuda`` the string becomes 1 byte/base codes that go straight into the
    Encoder's packed-input path (the reference builds the 512 MB float one-hot array with `Genome.sequence_to_encoding`)."""
    from . import orca_predict
    midpoint = int(len(sequence_str) / 2)
    if midpoint < _R32:
        raise ValueError("Sequence length needs to be at least 32Mb long.\n" + " Current length is " + str(len(sequence_str)))
    if midpoint > _R32:
        print("Sequence length is longer than 32Mb. Only the middle 32Mb will be used.")
        sequence_str = sequence_str[midpoint - _R32: midpoint + _R32]
    midpoint = int(len(sequence_str) / 2)
    wpos = midpoint
    if mpos is None:
        mpos = midpoint
    codes = _genome.sequence_to_codes(sequence_str)
    if use_cuda:
        sequence = torch.from_numpy(codes)[None].cuda()
    else:
        sequence = _genome.codes_to_encoding(codes)[None, :, :]
    return orca_predict.genomepredict(sequence, "customized seq", mpos, wpos, models=list(models), targets=False, use_cuda=use_cuda)


def process_seqstr(seqstr_input, file=None, mpos=None, custom_models=None, model_labels=None, use_cuda=True):
    """`process_region` for a Seqstr specification, e.g. '[hg38]chr9:94904000-126904000 +' (orca_predict.py:3060-3161):
    the first sequence the `seqstr` package returns for the one-line input, then `predict_sequence_string`.
    ImportError when `seqstr` is not installed, as in the reference; `file=` (plotting) is not part of orca_amd."""
    try:
        from seqstr import seqstr
    except ImportError:
        raise ImportError("Seqstr is not installed. Please install it first.\n" + "pip install seqstr")
    models, _ = _setup(custom_models, _R32, model_labels, file)
    return predict_sequence_string(seqstr(seqstr_input)[0].Seq, mpos=mpos, models=models, use_cuda=use_cuda)
