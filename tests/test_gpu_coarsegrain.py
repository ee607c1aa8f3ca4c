"""-m gpu: the observed-data path - `orca_adaptive_coarsegrain` through the C ABI against the oracle
and against the reference's own outputs (G18), bit-exact including the NaN pattern; `Genomic2DFeatures(cg=True)` on a
dense stand-in for a cooler file; and `genomepredict(targets=...)` filling output["experiments"] from it."""
import numpy as np
import pytest
import torch

from oracle import coarsegrain_oracle as CO
from orca_amd import selene_utils2 as S, synth
from tests.util import golden

pytestmark = pytest.mark.gpu


def _same(a, b):
    return a.shape == b.shape and np.array_equal(np.isnan(a), np.isnan(b)) and np.array_equal(np.nan_to_num(a, nan=0.0), np.nan_to_num(b, nan=0.0))


def test_coarsegrain_vs_reference_outputs_and_oracle(cuda):
    g = golden("G18_coarsegrain.npz")
    for name, n, m, seed in synth.COARSEGRAIN_CASES:
        a, c = synth.synth_hic(n, seed, m=m)
        out = S._adaptive_coarsegrain(a, c, cuda=True)
        assert out.dtype == np.float32 and _same(out, CO.adaptive_coarsegrain_any_shape(a, c).astype(np.float32)), name
        if name in g.files:
            assert _same(out, g[name]), name
        else:
            assert _same(out[:64, :64], g[name + "_corner"]), name


@pytest.mark.parametrize("n,cutoff,levels", [(256, 5, 8), (513, 3, 12), (2000, 5, 12), (64, 1, 2), (40, 50, 8)])
def test_coarsegrain_sizes_and_parameters_vs_oracle(cuda, n, cutoff, levels):
    a, c = synth.synth_hic(n, 100 + n, nan_frac=0.05 if n != 64 else 0.0, depth=1.0 if n > 1000 else 4.0)
    out = S.adaptive_coarsegrain_gpu(a, c, cutoff=cutoff, max_levels=levels)
    assert _same(out, CO.adaptive_coarsegrain(a, c, cutoff=cutoff, max_levels=levels).astype(np.float32))


def test_genomic2dfeatures_on_a_dense_source_fills_experiments(cuda):
    bs, nb = 4000, 8000
    bal, raw = synth.synth_hic(nb, 3, depth=2.0)
    src = S.MatrixCooler({"chrS": bal.astype(np.float32)}, {"chrS": raw.astype(np.float32)}, bs)
    t = S.Genomic2DFeatures([src], ["synthetic"], (nb, nb), cg=True)
    w = t.get_feature_data("chrS", 4_000_000, 12_000_000)          # 2000 x 2000 bins
    assert w.shape == (2000, 2000) and w.dtype == np.float32
    assert _same(w, CO.adaptive_coarsegrain(bal[1000:3000, 1000:3000], raw[1000:3000, 1000:3000], max_levels=12).astype(np.float32))
    plain = S.Genomic2DFeatures(src, "synthetic", (nb, nb)).get_feature_data("chrS", 0, 400_000)
    assert _same(plain, bal[:100, :100].astype(np.float32))
    # the smoothed window as `targets` of genomepredict: experiments = log((coarse-grained target + eps) / (background + eps))
    from orca_amd import orca_models, orca_predict as P
    model = orca_models.H1esc(synthetic_seed=0)
    full = t.get_feature_data("chrS", 0, 32_000_000)
    seq = torch.from_numpy(synth.synth_base_codes(32_000_000, seed=1)[None]).to(cuda)
    out = P.genomepredict(seq, "chrS", 16_000_000 + 1_234_567, 16_000_000, models=[model], targets=[torch.from_numpy(full[None])])
    exp = out["experiments"][0]
    assert len(exp) == 6 and all(e.shape == (250, 250) for e in exp)
    tr = P._coarse_grain(full[None, :8000, :8000], 32, 1)
    want = np.log((tr + model.epss[32]) / (model.normmats[32] + model.epss[32]))[0]
    assert np.allclose(np.nan_to_num(exp[0]), np.nan_to_num(want), rtol=1e-6, atol=1e-6)
