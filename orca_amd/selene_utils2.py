"""Observed-data side of the hot path: the counterparts of the reference's
`selene_utils2.adaptive_coarsegrain_gpu` (:274-463), `_adaptive_coarsegrain` (:466-504) and `Genomic2DFeatures`
(:507-584) - what turns a cooler file into the `targets` that `genomepredict` coarse-grains into
`output["experiments"]`.

The smoother runs on the MI355X through the C ABI (`orca_adaptive_coarsegrain`, bit-identical to the reference's
float32 arithmetic); there is no CPU path.  `cooler` / `h5py` are not part of this image: `Genomic2DFeatures` opens
`cooler.Cooler(path)` when the module is importable and otherwise accepts, in place of a path, any object with the
two calls it makes on a Cooler - `obj.matrix(balance=True|False).fetch(region[, region2])` - e.g. `MatrixCooler`
below (dense matrices per chromosome, from memory or an `.npz`)."""
import ctypes

import numpy as np
import torch

from . import _lib, engine


def adaptive_coarsegrain_gpu(ar, countar, cutoff=5, max_levels=8, min_shape=8, device=None):
    """selene_utils2.py:274-463 on the MI355X.  ar / countar: square numpy arrays or ROCm tensors of one shape;
    returns a float32 numpy array (as the reference does: `.cpu().numpy()`, :463)."""
    dev = device if device is not None else (ar.device if isinstance(ar, torch.Tensor) and ar.is_cuda else torch.device("cuda", torch.cuda.current_device()))

    def up(a):
        t = a if isinstance(a, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))
        return t.to(device=dev, dtype=torch.float32).contiguous()

    a, c = up(ar), up(countar)
    if a.dim() != 2 or a.shape[0] != a.shape[1] or a.shape != c.shape:
        raise ValueError(f"adaptive_coarsegrain_gpu needs two square matrices of one shape, got {tuple(a.shape)} and {tuple(c.shape)}")
    n = a.shape[0]
    out = torch.empty((n, n), dtype=torch.float32, device=dev)
    ctx = engine.get_context(dev)
    _lib.check(_lib.load().orca_adaptive_coarsegrain(ctx.handle, ctypes.c_void_p(a.data_ptr()), ctypes.c_void_p(c.data_ptr()), n, n, float(cutoff),
                                                     int(max_levels), int(min_shape), ctypes.c_void_p(out.data_ptr()), n), "orca_adaptive_coarsegrain")
    return out.cpu().numpy()


def _adaptive_coarsegrain(ar, countar, max_levels=12, cuda=True):
    """selene_utils2.py:466-504: tiny (< 9 x 9) and non-square inputs are padded with NaN to a square first.  ``cuda`` is
    kept for the signature; the reference's CPU variant is cooltools' (not vendored) - here the kernel always runs."""
    ar, countar = np.asarray(ar), np.asarray(countar)
    assert ar.shape == countar.shape
    h, w = ar.shape
    if h == w and h >= 9:
        return adaptive_coarsegrain_gpu(ar, countar, max_levels=max_levels)
    n = 9 if (h < 9 and w < 9) else max(h, w)
    a = np.full((n, n), np.nan, dtype=np.float32)
    c = np.full((n, n), np.nan, dtype=np.float32)
    a[:h, :w], c[:h, :w] = ar, countar
    return adaptive_coarsegrain_gpu(a, c, max_levels=max_levels)[:h, :w]


class MatrixCooler:
    """The two `cooler.Cooler` calls `Genomic2DFeatures` makes, on dense per-chromosome matrices:
    ``balanced[chrom]`` / ``raw[chrom]`` are [nbins, nbins] arrays at ``binsize`` bp (NaN = masked bin in `balanced`);
    trans blocks come from ``balanced[(chrom, chrom2)]`` when given."""

    def __init__(self, balanced, raw, binsize):
        self.balanced, self.raw, self.binsize = balanced, raw, int(binsize)

    @classmethod
    def from_npz(cls, path):
        z = np.load(path, allow_pickle=False)
        chroms = sorted({k.split("|", 1)[1] for k in z.files if "|" in k})
        return cls({c: z["balanced|" + c] for c in chroms}, {c: z["raw|" + c] for c in chroms}, int(z["binsize"]))

    def matrix(self, balance=True):
        src, bs = (self.balanced if balance else self.raw), self.binsize

        class _Sel:
            @staticmethod
            def fetch(region, region2=None):
                (c1, s1, e1), (c2, s2, e2) = region, (region2 or region)
                m = src[c1] if c1 == c2 else src[(c1, c2)]
                return np.asarray(m[s1 // bs: -(-e1 // bs), s2 // bs: -(-e2 // bs)])
        return _Sel()


class Genomic2DFeatures:
    """selene_utils2.py:507-584: one or several Hi-C style datasets queried by genomic window; ``cg=True`` applies the
    adaptive coarse-graining.  Same constructor, attributes and `get_feature_data` contract as the reference."""

    def __init__(self, input_paths, features, shape, cg=False, cuda=False):
        if isinstance(features, str):
            input_paths, features = [input_paths], [features]
        self.input_paths = list(input_paths)
        self._initialized = False
        self.n_features = len(features)
        self.feature_index_dict = dict((feat, index) for index, feat in enumerate(features))
        self.shape, self.cg, self.cuda = shape, cg, cuda

    def _open(self, src):
        if hasattr(src, "matrix"):
            return src
        if isinstance(src, str) and src.endswith(".npz"):
            return MatrixCooler.from_npz(src)
        try:
            import cooler
        except ImportError as e:
            raise ImportError(f"cannot open {src!r}: the `cooler` package is not installed; pass a MatrixCooler (or an .npz written "
                              "for MatrixCooler.from_npz) instead") from e
        return cooler.Cooler(src)

    def get_feature_data(self, chrom, start, end, chrom2=None, start2=None, end2=None):
        """[n_features, rows, cols] float32 (2-D for a single dataset) for the window, or for the rectangle between two
        windows when the second one is given; `cg=True` smooths every dataset with its own raw counts."""
        if not self._initialized:
            self.data, self._initialized = [self._open(src) for src in self.input_paths], True
        self.chrom, self.start, self.end = chrom, start, end
        regions = [(chrom, start, end)]
        if None not in (chrom2, start2, end2):
            regions.append((chrom2, start2, end2))
        mats = []
        for source in self.data:
            balanced = np.asarray(source.matrix(balance=True).fetch(*regions))
            if self.cg:
                balanced = _adaptive_coarsegrain(balanced, np.asarray(source.matrix(balance=False).fetch(*regions)), cuda=self.cuda)
            mats.append(balanced.astype(np.float32))
        return mats[0] if len(mats) == 1 else np.stack(mats, axis=0)
