"""-m gpu: the HBM-resident genome stores: windows expanded on the device by the library's
`orca_genome_unpack_2bit` kernel equal the 1-byte store's (bit-exact; integer / byte work)."""
import numpy as np
import pytest
import torch

from orca_amd.genome import PackedGenome, TwoBitGenome

pytestmark = pytest.mark.gpu


def test_two_bit_genome_on_device_equals_one_byte_store(cuda):
    g1 = PackedGenome.random({"chrA": 1_000_003, "chrB": 70_001, "chrC": 5}, seed=4, n_runs=3)
    g2 = TwoBitGenome.from_packed(g1).to(cuda)
    g1d = PackedGenome({c: g1._host[c] for c in g1.get_chrs()}).to(cuda)
    rs = np.random.RandomState(0)
    for chrom, n in g1.get_chr_lens():
        for _ in range(25):
            a, b = sorted(rs.randint(-50, n + 50, 2))
            a, b = min(a, n), max(b, 0)          # a padded window has to touch the chromosome (G26)
            for strand in "+-":
                ref = g1.get_codes_from_coords(chrom, a, b, strand, pad=True)
                got = g2.get_codes_from_coords(chrom, a, b, strand, pad=True)
                assert isinstance(got, torch.Tensor) and got.is_cuda
                np.testing.assert_array_equal(got.cpu().numpy(), ref)
                np.testing.assert_array_equal(g1d.get_codes_from_coords(chrom, a, b, strand, pad=True).cpu().numpy(), ref)
    # a whole 1 Mb window feeds the Encoder's packed input unchanged
    w = g2.get_codes_from_coords("chrA", 0, 1_000_000)
    np.testing.assert_array_equal(w.cpu().numpy(), g1._host["chrA"][:1_000_000])


@pytest.mark.parametrize("precision", ["f16x2", "bf16", "f32", "bf16x3"])
def test_encoder_reads_the_two_bit_genome_in_place(cuda, precision):
    """`Encoder.forward_2bit` (orca_encoder_forward_2bit): the first-layer kernels read the 2-bit plane + N mask of a chromosome in HBM
    directly - no unpack pass, no 1 byte/base window - and give bit for bit what the unpacked codes give (`forward_codes`), on both
    strands, at odd window starts (every 2-bit / mask phase), with N runs, through bin ranges, in every arithmetic mode (composed first
    layers, residual from the bases, the fp32 tap sums, the expanded float rows of the 3-way split mode)."""
    from tests.util import product_module
    g1 = PackedGenome.random({"chrA": 2_000_003}, seed=4, n_runs=5)
    g2 = TwoBitGenome.from_packed(g1).to(cuda)
    enc = product_module("Encoder", 0, precision=precision)
    for start, nb in ((0, 60), (4001, 60), (123_457, 97), (1_000_002, 33)):
        end = start + 4000 * nb
        codes = g2.get_codes_from_coords("chrA", start, end)[None]
        for rev in (False, True):
            ref = enc.forward_codes(codes, reverse=rev)
            got = enc.forward_2bit(g2, "chrA", start, end, reverse=rev)
            assert got.shape == (1, 128, nb) and torch.equal(got, ref), (precision, start, nb, rev)
        part = enc.forward_2bit(g2, "chrA", start, end, bin_lo=11, bin_hi=29)
        assert torch.equal(part, enc.forward_codes(codes, bin_lo=11, bin_hi=29))
    with pytest.raises(Exception):
        enc.forward_2bit(g2, "chrA", 1_999_000, 2_003_000)              # beyond the chromosome


def _g26():
    import os
    return np.load(os.path.join(os.path.dirname(__file__), "golden", "G26_genome.npz"))


def test_hbm_resident_stores_against_the_reference_memmap_genome(cuda):
    """G26 (the reference's `MemmapGenome.get_encoding_from_coords`, selene_utils2.py:186-272, query by query) against the genome
    stores RESIDENT IN HBM: the 1-byte store's device slices and the 2-bit store's windows expanded by `orca_genome_unpack_2bit`,
    on both strands, padded over either end, touching an end; the reference's asserts fire on the device path too."""
    from orca_amd.genome import codes_to_encoding, sequence_to_codes
    d = _g26()
    recs = {str(c): str(d["seq_" + str(c)]) for c in d["chrs"]}
    g1 = PackedGenome({c: sequence_to_codes(s) for c, s in recs.items()}).to(cuda)
    g2 = TwoBitGenome.from_packed(g1).to(cuda)
    rows, offs = d["rows"], d["row_offsets"]
    n_ok = 0
    for i in range(d["q_chrom"].shape[0]):
        c, a, b, strand, pad = str(d["q_chrom"][i]), int(d["q_start"][i]), int(d["q_end"][i]), str(d["q_strand"][i]), bool(d["q_pad"][i])
        if i % 2 and b - a < 30000:
            continue
        for g in (g1, g2):
            if str(d["status"][i]) == "ok":
                got = g.get_codes_from_coords(c, a, b, strand, pad)
                assert isinstance(got, torch.Tensor) and got.is_cuda and got.dtype == torch.uint8
                assert np.array_equal(codes_to_encoding(got.cpu().numpy()), rows[offs[i]:offs[i + 1]]), (i, c, a, b, strand, pad)
                n_ok += 1
            else:
                with pytest.raises(AssertionError):
                    g.get_codes_from_coords(c, a, b, strand, pad)
    assert n_ok > 500


@pytest.mark.parametrize("precision", ["f16x2", "f32"])
def test_encoder_on_genome_windows_against_the_reference_rows(cuda, precision):
    """The Encoder fed from the HBM-resident stores - codes of a padded window (`forward_codes`), the 2-bit planes read in place
    (`forward_2bit`, with the '-' strand derived on the device) - against the Encoder on the float rows the REFERENCE's genome store
    returns for the same query (G26; `orca_predict.py:324-337` is what would upload them): the window's bases, N runs, 0.25 padding
    and reverse complement are the reference's, the arithmetic is the same network - 1e-5."""
    from orca_amd.genome import sequence_to_codes
    from tests.util import product_module
    d = _g26()
    recs = {str(c): str(d["seq_" + str(c)]) for c in d["chrs"]}
    g2 = TwoBitGenome.from_packed(PackedGenome({c: sequence_to_codes(s) for c, s in recs.items()})).to(cuda)
    enc = product_module("Encoder", 0, precision=precision)
    rows, offs = d["rows"], d["row_offsets"]
    n = 0
    for i in range(d["q_chrom"].shape[0]):
        c, a, b, strand, pad = str(d["q_chrom"][i]), int(d["q_start"][i]), int(d["q_end"][i]), str(d["q_strand"][i]), bool(d["q_pad"][i])
        if b - a < 20000 or (b - a) % 4000 or str(d["status"][i]) != "ok":
            continue
        x = torch.from_numpy(rows[offs[i]:offs[i + 1]].copy())[None].transpose(1, 2).to(cuda)      # the reference's call form: [1, L, 4] rows, transposed view
        ref = enc(x)
        got = enc.forward_codes(g2.get_codes_from_coords(c, a, b, strand, pad)[None])
        assert float((got - ref).abs().max()) < 1e-5, (i, c, a, b, strand)
        if 0 <= a and b <= len(recs[c]):          # inside the chromosome: the planes are read in place, '-' = the window's reverse complement
            got2 = enc.forward_2bit(g2, c, a, b, reverse=strand == "-")
            assert float((got2 - ref).abs().max()) < 1e-5, (i, c, a, b, strand)
            n += 1
    assert n >= 4
