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


def _strands(sequence, use_cuda):
    """Forward strand and reverse complement (flip of the length AND channel axes,
    orca_predict.py:324-329) as [B,4,L] views of [B,L,4] storage."""
    seq = np.asarray(sequence, dtype=np.float32)
    for s in (seq, seq[:, ::-1, ::-1]):
        t = torch.from_numpy(np.ascontiguousarray(s))
        if use_cuda:
            t = t.cuda()
        yield t.transpose(1, 2)


def _log_background(bg, batch, use_cuda, flip=False):
    """log(background) as a [B,1,n,n] (expanded) tensor, orca_predict.py:349-353 / :692-697,:703."""
    a = np.asarray(bg)
    a = a[(None,) * (4 - a.ndim)]
    t = torch.log(torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)))
    if flip:
        t = torch.flip(t, [2, 3])
    if use_cuda:
        t = t.cuda()
    return t.expand(batch, -1, -1, -1)


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
    (orca_predict.py:404-436)."""
    T = mat.shape[0]
    r = np.reshape(mat, (T, 250, nblock, 250, nblock))
    with np.errstate(invalid="ignore"):
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("ignore", category=RuntimeWarning)
            mean = np.nanmean(np.nanmean(r, axis=4), axis=2)
    frac = np.mean(np.mean(np.isnan(r), axis=4), axis=2)
    mean[frac > nan_thresh] = np.nan
    return mean


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


def cascade_32m(model, x, mpos, wpos, reverse, distencs=None):
    """One strand, one model, tensors already on the device: net0 -> net -> six decoders
    (+ denet_1_pt at 4 kb).  Returns (preds[6] each [B,1,250,250], starts[6] in 4 kb bins).
    ``distencs``: optional {level: [B,1,250,250] log-background tensor} to skip the upload."""
    batch = x.shape[0]
    encs = model.net(model.net0(x))
    encodings = dict(zip([1, 2, 4, 8, 16, 32], encs))
    preds, starts, start_index = [], [0], 0
    for j, level in enumerate([32, 16, 8, 4, 2, 1]):
        s = int(starts[j] / level)
        coarse = None
        if j > 0:
            coarse = preds[j - 1][:, :, start_index: start_index + 125, start_index: start_index + 125]
        de = distencs[level] if distencs is not None else _log_background(model.normmats[level], batch, x.is_cuda)
        preds.append(_decode(model, level, encodings[level][:, :, s: s + 250], de, coarse, level == 1))
        start_index = zoom_index_32m(level, starts[j], mpos, wpos, reverse)
        starts.append(starts[j] + start_index * level)
    return preds, starts[:-1]


def _merge(allpreds, n_models):
    """0.5*fwd + 0.5*rev[::-1,::-1], batch row 0 only (orca_predict.py:510-523)."""
    merged = [[] for _ in range(n_models)]
    for i in range(n_models):
        for fwd, rev in zip(allpreds[i], allpreds[i + n_models]):
            if fwd.is_cuda and fwd.shape[1] == 1:
                merged[i].append(engine.strand_merge(fwd[0, 0], rev[0, 0]).cpu().numpy())
                continue
            f, r = fwd.cpu().detach().numpy(), rev.cpu().detach().numpy()
            if f.shape[1] == 1:
                merged[i].append(f[0, 0, :, :] * 0.5 + r[0, 0, ::-1, ::-1] * 0.5)
            else:
                merged[i].append(f[0, :, :, :] * 0.5 + r[0, :, ::-1, ::-1] * 0.5)
    return merged


def genomepredict(sequence, mchr, mpos=-1, wpos=-1, models=["h1esc", "hff"], targets=None, annotation=None,
                  use_cuda=True, nan_thresh=1):
    """Multiscale prediction for a 32 Mb sequence (orca_predict.py:231-540): six maps
    (32, 16, 8, 4, 2, 1 Mb windows, 250x250 each) zooming into ``mpos``.
    ``sequence``: float array [1, 32000000, 4]; ``wpos``: window centre coordinate."""
    models = _resolve_models(models, "32M", use_cuda)
    n_models = len(models)
    levels = [32, 16, 8, 4, 2, 1]
    batch = sequence.shape[0]
    allpreds, allstarts, alltargets, allannos = [], [], [], []
    with torch.no_grad():
        for iii, x in enumerate(_strands(sequence, use_cuda)):
            for ii, model in enumerate(models):
                encs = model.net(model.net0(x))
                encodings = dict(zip([1, 2, 4, 8, 16, 32], encs))
                preds, starts, ts, annos = [], [0], [], []
                start_index = 0
                for j, level in enumerate(levels):
                    s = int(starts[j] / level)
                    coarse = None
                    if j > 0:
                        coarse = preds[j - 1][:, :, start_index: start_index + 125, start_index: start_index + 125]
                    pred = _decode(model, level, encodings[level][:, :, s: s + 250],
                                   _log_background(model.normmats[level], batch, use_cuda), coarse, level == 1)
                    if targets and iii == 0:
                        tgt = targets[ii]
                        tgt = tgt.numpy() if isinstance(tgt, torch.Tensor) else np.asarray(tgt)
                        w = 250 * level
                        tr = _coarse_grain(tgt[:, starts[j]: starts[j] + w, starts[j]: starts[j] + w], level, nan_thresh)
                        lf = np.log((tr + model.epss[level]) / (model.normmats[level] + model.epss[level]))
                        ts.append(lf[0, :, :] if tr.shape[0] == 1 else lf)
                    if annotation is not None and iii == 0:
                        annos.append(_scale_annotation(annotation, starts[j] / 8000.0, (starts[j] + 250 * level) / 8000.0))
                    start_index = zoom_index_32m(level, starts[j], mpos, wpos, reverse=(iii != 0))
                    starts.append(starts[j] + start_index * level)
                    preds.append(pred)
                allpreds.append(preds)
                if iii == 0:
                    allstarts.append(starts[:-1])
                    if targets:
                        alltargets.append(ts)
                    if annotation is not None:
                        allannos.append(annos)
    output = {"predictions": _merge(allpreds, n_models)}
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
    n_models = len(models)
    levels = [256, 128, 64, 32]
    batch = sequence.shape[0]
    allpreds, allstarts, allnormmats, alltargets, allannos = [], [], [], [], []
    with torch.no_grad():
        for iii, x in enumerate(_strands(sequence, use_cuda)):
            for ii, model in enumerate(models):
                normmat = normmats[ii]
                isnan = np.isnan(normmat)
                if np.any(isnan):
                    normmat[isnan] = np.nanmin(normmat[~isnan])   # in place, as the reference (:664-667)
                encs = model.net(model.net1(model.net0(x))[-1])
                encodings = dict(zip([32, 64, 128, 256], encs))
                preds, starts, ns, ts, annos = [], [0], {}, [], []
                start_index = 0
                for j, level in enumerate(levels):
                    unit = level // 8                      # 32 kb bins per map pixel
                    w = 250 * unit
                    ns[level] = _coarse_grain(normmat[None, starts[j]: starts[j] + w, starts[j]: starts[j] + w], unit, 1)
                    s = int(starts[j] / unit)
                    coarse = None
                    if j > 0:
                        coarse = preds[j - 1][:, :, start_index: start_index + 125, start_index: start_index + 125]
                    pred = _decode(model, level, encodings[level][:, :, s: s + 250],
                                   _log_background(ns[level], batch, use_cuda, flip=(iii != 0)), coarse, False)
                    if targets and iii == 0:
                        tgt = targets[ii]
                        tgt = tgt.numpy() if isinstance(tgt, torch.Tensor) else np.asarray(tgt)
                        tr = _coarse_grain(tgt[:, starts[j]: starts[j] + w, starts[j]: starts[j] + w], unit, nan_thresh)
                        eps = np.nanmin(ns[level])
                        lf = np.log((tr + eps) / (ns[level] + eps))
                        ts.append(lf[0, :, :] if tr.shape[0] == 1 else lf)
                    if annotation is not None and iii == 0:
                        annos.append(_scale_annotation(annotation, starts[j] / 8000.0, (starts[j] + w) / 8000.0))
                    # zoom with chromosome-end bounds (orca_predict.py:813-835)
                    half = level * 1000000 / 4
                    if iii == 0:
                        proposed = (mpos - half) - (wpos - 128000000 + starts[j] * 32000)
                    else:
                        proposed = (mpos - half) - (wpos + 128000000 - starts[j] * 32000 - level * 1000000)
                    if chrlen is not None:
                        lo = 0 - (wpos - 128000000)
                        hi = chrlen - level * 1000000 / 2 - (wpos - 128000000)
                        proposed = np.clip(proposed, lo, hi) if lo < hi else lo
                    start_index = int(np.clip(np.floor(proposed / (4000 * level)), 0, 125))
                    if iii != 0:
                        start_index = 250 - (start_index + 125)
                    starts.append(starts[j] + start_index * unit)
                    preds.append(pred)
                allpreds.append(preds)
                allnormmats.append(ns)
                if iii == 0:
                    allstarts.append(starts[:-1])
                    if targets:
                        alltargets.append(ts)
                    if annotation is not None:
                        allannos.append(annos)
    output = {"predictions": _merge(allpreds, n_models)}
    output["experiments"] = alltargets if targets else None
    output["start_coords"] = [wpos - 128000000 + s * 32000 for s in allstarts[0]]
    output["end_coords"] = [np.fmin(int(output["start_coords"][ii] + 256000000 / 2 ** (ii)), chrlen) for ii in range(4)]
    output["annos"] = allannos[0] if annotation is not None else None
    output["chr"] = mchr
    output["padding_chr"] = padding_chr
    output["normmats"] = allnormmats
    return output
