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
