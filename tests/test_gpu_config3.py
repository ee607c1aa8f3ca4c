"""-m gpu: BASELINE.json configs[2] - HFF-shaped 32 Mb model, batch of 8 random 32 Mb sequences, module-level forward
(net0 -> net -> six Decoder levels + denet_1_pt; `genomepredict` keeps only batch row 0, orca_predict.py:514-523, so
batch > 1 is a module-level mode, train_h1esc_b.py:170) - against rows 0 and 5 computed by the REFERENCE's own
nn.Modules on CPU (tests/golden/G17_config3.npz, tools/make_golden.py --config3).

  * default arithmetic (fp32-class f16x2): north-star tolerance, 1e-4 max-abs per level;
  * "bf16" throughput mode (2-byte activations end to end: Encoder stages 1-3 on single-plane bf16 activations, Decoder
    feature maps - residual stream included - as single bf16 planes, bf16 operands, one MFMA product, fp32 accumulate): bf16
    keeps 8 significant bits and the Decoders round their residual stream 56 times, so the maps agree to 1-2 decimal digits -
    stated tolerance: max-abs 0.6 on maps of range ~+-3 (measured 0.40), Pearson r >= 0.999 per level (measured 0.9998);
  * the same with the Decoders on single fp16 planes ("f16": same traffic and rate, the residual stream keeps 11 significant bits, fp16
    range guard) - THE config-3 mode of bench.py / tools/run_configs.py: stated tolerance max-abs 0.1 (2 x the measured 0.045),
    Pearson r >= 0.99999 (measured 0.999994; the Encoder's bf16 rounding dominates)."""
import numpy as np
import pytest
import torch

from orca_amd import orca_models, orca_predict as P, synth
from tests.util import golden, maxabs, pearson, stats

pytestmark = pytest.mark.gpu
CFG = {"seed": 7, "rows": (0, 5), "row_seed0": 10, "L": 32_000_000, "mpos": 17_234_567, "wpos": 16_000_000}   # = tools/make_golden.py CONFIG3


@pytest.fixture(scope="module")
def setup(cuda):
    model = orca_models.Hff(synthetic_seed=CFG["seed"])
    codes = torch.from_numpy(np.stack([synth.synth_base_codes(CFG["L"], seed=CFG["row_seed0"] + b) for b in range(8)])).to(cuda)
    de = {lv: torch.log(torch.from_numpy(model.normmats[lv][None, None].astype(np.float32))).to(cuda) for lv in model.levels}
    return model, codes, de


def _forward(model, codes, de, precision, dec_precision=None):
    dec_precision = dec_precision or precision
    model.net0.precision = precision
    for lv in model.levels:
        model.denets[lv].precision = dec_precision
    model.denet_1_pt.precision = dec_precision
    enc0 = model.net0.forward_codes(codes)
    encs = dict(zip([1, 2, 4, 8, 16, 32], model.net(enc0)))
    preds, starts = P.run_cascade(model, encs, [32, 16, 8, 4, 2, 1], lambda lv: lv, codes.shape[0], [False], lambda lv, k, st: de[lv],
                                  lambda lv, st, rev: P.zoom_index_32m(lv, st, CFG["mpos"], CFG["wpos"], rev), add_1m_level=1)
    return enc0.cpu().numpy(), [p[:, 0].cpu().numpy() for p in preds], starts[0]


def test_config3_default_arithmetic_vs_reference(setup):
    model, codes, de = setup
    g = golden("G17_config3.npz")
    enc0, maps, starts = _forward(model, codes, de, "f16x2")
    assert enc0.shape == (8, 128, 8000) and all(m.shape == (8, 250, 250) for m in maps)
    for b in CFG["rows"]:
        assert list(starts) == list(g[f"starts_row{b}"])
        assert maxabs(enc0[b][:, :64], g[f"enc0_first64_row{b}"]) < 1e-4 and maxabs(enc0[b][:, -64:], g[f"enc0_last64_row{b}"]) < 1e-4
        st, gs = stats(enc0[b]), g[f"enc0_stats_row{b}"]
        assert abs(st[0] - gs[0]) < 1e-4 * enc0[b].size and abs(st[1] / gs[1] - 1) < 1e-4
        for j in range(6):
            err, r = maxabs(maps[j][b], g[f"maps_row{b}"][j]), pearson(maps[j][b], g[f"maps_row{b}"][j])
            assert err < 1e-4 and r > 0.999999, (b, j, err, r)
    # the rows of a batch are different sequences: their maps differ by far more than the tolerance (random sequences
    # through random weights average out to similar maps - a few 1e-3 apart at 32 Mb, more at 1 Mb)
    assert max(maxabs(m[0], m[5]) for m in maps) > 1e-3


@pytest.mark.parametrize("dec_precision,tol,rmin", [("bf16", 0.6, 0.999), ("f16", 0.1, 0.99999)])
def test_config3_bf16_throughput_mode_vs_reference(setup, dec_precision, tol, rmin):
    model, codes, de = setup
    g = golden("G17_config3.npz")
    enc0, maps, starts = _forward(model, codes, de, "bf16", dec_precision)
    worst = (0.0, 1.0)
    for b in CFG["rows"]:
        assert list(starts) == list(g[f"starts_row{b}"])
        for j in range(6):
            err, r = maxabs(maps[j][b], g[f"maps_row{b}"][j]), pearson(maps[j][b], g[f"maps_row{b}"][j])
            worst = (max(worst[0], err), min(worst[1], r))
            assert err < tol and r > rmin, (b, j, err, r)
    print(f"config 3, Encoder bf16 / Decoders {dec_precision} vs reference: worst max-abs {worst[0]:.4g}, worst Pearson {worst[1]:.6f}")
    _forward(model, codes, de, "f16x2")   # leave the module-scoped model in its default arithmetic
