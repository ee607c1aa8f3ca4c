"""oracle/coarsegrain_oracle.py (numpy restatement of selene_utils2.py:274-504) against G18 - outputs of the REFERENCE's
own adaptive_coarsegrain_gpu / _adaptive_coarsegrain on synthetic observed Hi-C blocks (tools/make_golden.py
--coarsegrain).  Float32 pooling sums in the reference's order: bit-exact, NaN pattern included."""
import numpy as np

from oracle import coarsegrain_oracle as CO
from orca_amd import synth
from tests.util import golden


def _same(a, b):
    return a.shape == b.shape and np.array_equal(np.isnan(a), np.isnan(b)) and np.array_equal(np.nan_to_num(a, nan=0.0), np.nan_to_num(b, nan=0.0))


def test_oracle_matches_reference_outputs():
    g = golden("G18_coarsegrain.npz")
    for name, n, m, seed in synth.COARSEGRAIN_CASES:
        a, c = synth.synth_hic(n, seed, m=m)
        out = CO.adaptive_coarsegrain_any_shape(a, c).astype(np.float32)
        if name in g.files:
            assert _same(out, g[name]), name
        else:
            fin = np.isfinite(out)
            st = np.array([out[fin].astype(np.float64).sum(), (out[fin].astype(np.float64) ** 2).sum(), float((~fin).sum())])
            assert np.array_equal(st, g[name + "_stats"]) and _same(out[:64, :64], g[name + "_corner"]), name


def test_oracle_properties():
    a, c = synth.synth_hic(96, 11, nan_frac=0.0, depth=50.0)
    out = CO.adaptive_coarsegrain(a, c, cutoff=1)
    # deep data, no masked bins: only zero-count pixels are pooled, everything else is untouched
    keep = c >= 1
    blocks_ok = np.ones_like(keep)
    k = keep.reshape(48, 2, 48, 2).all(axis=(1, 3))
    blocks_ok = np.repeat(np.repeat(k, 2, 0), 2, 1)
    assert np.array_equal(out[blocks_ok], a.astype(np.float32)[blocks_ok]) and np.isfinite(out).all()
    # pooling conserves the block sums (the replacement spreads V_cg over the valid pixels, selene_utils2.py:420-430)
    assert np.allclose(out.astype(np.float64).sum(), a.astype(np.float32).astype(np.float64).sum(), rtol=1e-5)
