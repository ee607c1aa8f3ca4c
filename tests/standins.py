"""Stand-ins used by the tests and the fixture generators only (NOT part of the product package): cheap replacements for the
networks (`FakeNet0`, `FakeNet`, `FakeDecoder`, `FakeModel32`), for the absent `seqstr` package, for the 256 Mb background carriers
and targets, and a recorder that takes the place of `genomepredict_256Mb` so that the 256 Mb structural-variant drivers are pinned
without a 256 Mb forward.  `tools/make_golden.py` drives the REFERENCE's drivers with the same objects (fixtures G7-G13, G19)."""
import numpy as np
import torch

from orca_amd.synth import seqstr_string, synth_normmats_32m


class FakeNet0(torch.nn.Module):
    """Cheap stand-in for Encoder used by cascade (host-logic) tests and
    fixtures: bins the [B,4,L] input into ``nbins`` windows, projects 4->128
    with a seeded matrix and adds a positional term so bins differ.  Pure
    torch ops; NOT part of the product path."""

    def __init__(self, nbins, seed=0):
        super().__init__()
        rs = np.random.RandomState(1000 + seed)
        self.nbins = nbins
        self.register_buffer("proj", torch.from_numpy(rs.normal(0, 1.0, (128, 4)).astype(np.float32)))
        t = np.arange(nbins, dtype=np.float64)[None, :]
        c = (np.arange(128, dtype=np.float64)[:, None] % 7 + 1)
        self.register_buffer("posterm", torch.from_numpy((0.3 * np.sin(0.01 * t * c)).astype(np.float32)))

    def forward(self, x):
        B, C, L = x.shape
        k = L // self.nbins
        pooled = x[:, :, : k * self.nbins].reshape(B, C, self.nbins, k).mean(dim=3)
        return torch.einsum("oc,bct->bot", self.proj, pooled) + self.posterm[None]

class FakeNet(torch.nn.Module):
    """Stand-in for Encoder2 in driver tests: the 6-level pyramid by average pooling (fine -> coarse)."""

    def forward(self, enc0):
        return [enc0 if k == 1 else torch.nn.functional.avg_pool1d(enc0, k, k) for k in (1, 2, 4, 8, 16, 32)]

class FakeDecoder(torch.nn.Module):
    """Stand-in for Decoder / Decoder_1m in driver tests: symmetric, depends on every input it is given."""

    def __init__(self, seed=0):
        super().__init__()
        self.register_buffer("v", torch.from_numpy(np.random.RandomState(2000 + seed).normal(0, 0.3, 128).astype(np.float32)))

    def forward(self, x, distenc=None, y=None):
        a = torch.einsum("c,bct->bt", self.v, x)
        out = (a[:, :, None] + a[:, None, :])[:, None]
        if distenc is not None:
            out = out + 0.05 * distenc
        if y is not None:
            out = out + 0.3 * torch.nn.functional.interpolate(y, scale_factor=2, mode="nearest")
        return out

class FakeModel32(torch.nn.Module):
    """A whole 32 Mb model container made of the cheap stand-ins above (seconds on a CPU): used to pin the
    structural-variant DRIVERS (sequence assembly, window clipping, annotations) against the reference's, whose
    numerics are pinned separately (G1-G9)."""

    def __init__(self, seed=0):
        super().__init__()
        self.net0 = FakeNet0(nbins=8000, seed=seed)
        self.net = FakeNet()
        self.denets = {lv: FakeDecoder(seed + lv) for lv in (1, 2, 4, 8, 16, 32)}
        self.denet_1_pt = FakeDecoder(seed + 100)
        self.normmats, self.epss = synth_normmats_32m(seed)
        self.levels = [1, 2, 4, 8, 16, 32]


class FakeSeqstr:
    """`seqstr(spec)` -> list of records with `.Seq`; the spec names a (length, seed) pair."""

    class _Rec:
        def __init__(self, seq):
            self.Seq = seq

    def __call__(self, spec):
        n, seed = (int(v) for v in spec.strip("[]").split(","))
        return [self._Rec(seqstr_string(n, seed)), self._Rec("ACGT")]      # only the first record is used (:3113)


class Background256:
    """Carrier of `background_cis` / `background_trans` (what `_retrieve_multi` reads from the 256 Mb models)."""

    def __init__(self, seed):
        d = np.arange(8000 + 2000, dtype=np.float64)
        self.background_cis = np.exp(-1.1 * np.log1p(d) - 2.0 + 0.02 * np.cos(d / 53.0 + seed))
        self.background_trans = float(np.exp(-12.5 - 0.1 * seed))


class FakeTarget256:
    """Stand-in for `selene_utils2.Genomic2DFeatures` at 32 kb resolution: a smooth function of the two absolute
    coordinates (cis) or a constant (trans).  The reference's `process_del` cannot run at 256 Mb WITHOUT targets
    (`targets` is unbound otherwise, `orca_predict.py:1627-1669`), so the 256 Mb fixtures carry one."""

    def get_feature_data(self, chrom, start, end, chrom2=None, start2=None, end2=None):
        if chrom2 is None:
            chrom2, start2, end2 = chrom, start, end
        a = np.arange(start, end, 32000, dtype=np.float64)[: int((end - start) / 32000)]
        b = np.arange(start2, end2, 32000, dtype=np.float64)[: int((end2 - start2) / 32000)]
        if chrom != chrom2:
            return np.full((a.shape[0], b.shape[0]), 1e-3 * (1 + len(chrom) + len(chrom2)))
        return 1.0 / (1.0 + np.abs(a[:, None] - b[None, :]) / 32000.0) + 1e-9 * (a[:, None] + 2 * b[None, :])


def seq_digest(sequence, binsize=1_024_000):
    """Exact digest of a [1,L,4] float one-hot sequence or of [1,L] base codes: per `binsize` bin,
    sum over positions of (pos % 1000 + 1) * sum_c (c + 1) * x[pos, c]  (an 'N' row counts 2.5)."""
    if hasattr(sequence, "detach"):
        sequence = sequence.detach().cpu().numpy()
    x = np.asarray(sequence)[0]
    L = x.shape[0]
    assert L % binsize == 0 and binsize % 1000 == 0
    w = np.tile(np.arange(1, 1001, dtype=np.float64), binsize // 1000)
    out = np.zeros(L // binsize)
    lut = np.array([1.0, 2.0, 3.0, 4.0, 2.5], dtype=np.float64)
    for b in range(L // binsize):
        blk = x[b * binsize:(b + 1) * binsize]
        v = lut[np.minimum(blk, 4)] if blk.ndim == 1 else blk.astype(np.float64) @ lut[:4]
        out[b] = float(v @ w)
    return out


class Recorder256:
    """Drop-in for genomepredict_256Mb (reference keyword/positional order, `orca_predict.py:543-556`)."""

    def __init__(self):
        self.calls = []

    def __call__(self, sequence, mchr, normmats, chrlen, mpos=-1, wpos=-1, models=None, targets=None, annotation=None,
                 padding_chr=None, use_cuda=True, nan_thresh=1):
        rec = {"seq": seq_digest(sequence), "chr": str(mchr), "chrlen": int(chrlen), "mpos": int(mpos), "wpos": int(wpos),
               "padding_chr": str(padding_chr), "has_targets": targets is not None,
               "anno": repr([[float(v) if not isinstance(v, str) else v for v in r] for r in annotation]) if annotation is not None else "None"}
        mats = [(f"nm{k}", nm) for k, nm in enumerate(normmats)]
        if targets is not None:
            mats += [(f"tgt{k}", t[0]) for k, t in enumerate(targets)]
        for key, nm in mats:
            nm = np.asarray(nm.numpy() if hasattr(nm, "numpy") else nm, dtype=np.float64)
            rec[f"{key}_shape"] = np.array(nm.shape)
            flat = nm.ravel()
            rec[f"{key}_stats"] = np.array([flat.sum(), float(np.dot(flat, flat)), flat.max(), flat.min()])
            rec[f"{key}_sub"] = nm[::200, ::200].copy()
        self.calls.append(rec)
        return {"call": len(self.calls) - 1}

    def summary(self, first=0):
        d = {}
        for i, rec in enumerate(self.calls[first:]):
            for k, v in rec.items():
                d[f"v{i}_{k}"] = v if isinstance(v, np.ndarray) else np.array([v])
        return d
