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
