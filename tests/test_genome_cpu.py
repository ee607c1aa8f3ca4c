"""PackedGenome: the selene query API of the reference's MemmapGenome (selene_utils2.py:38-272) on 1 byte/base storage.
The reference pins no encoding test (selene is not vendored); the semantics asserted here are the ones its code
states: channel order A,C,G,T, unknown symbols 0.25 x 4, '-' strand = both axes flipped, pad = 0.25 beyond the ends."""
import numpy as np
import pytest

from orca_amd.genome import PackedGenome, codes_to_encoding, revcomp_codes, sequence_to_codes, sequence_to_encoding


def _write_fasta(path, records, width=60):
    with open(path, "w") as f:
        for name, seq in records.items():
            f.write(f">{name} some description\n")
            for i in range(0, len(seq), width):
                f.write(seq[i:i + width] + "\n")


@pytest.fixture
def fasta(tmp_path):
    rs = np.random.RandomState(5)
    recs = {f"chr{k}": "".join(rs.choice(list("ACGTacgtNRY"), size=n)) for k, n in ((1, 1234), (2, 60), (3, 7), (4, 0), (5, 601))}
    p = str(tmp_path / "g.fa")
    _write_fasta(p, recs)
    return p, recs


def test_fasta_scan_and_indexed_reads_agree(fasta):
    path, recs = fasta
    g = PackedGenome.from_fasta(path)
    assert g.get_chr_lens() == sorted((k, len(v)) for k, v in recs.items())
    for k, v in recs.items():
        assert np.array_equal(g.get_codes_from_coords(k, 0, len(v)), sequence_to_codes(v))
    sub = PackedGenome.from_fasta(path, chroms=["chr5", "chr2"])          # no index yet: scan, selected records only
    assert sub.get_chrs() == ["chr2", "chr5"]
    PackedGenome.write_fai(path)
    idx = PackedGenome.from_fasta(path, chroms=["chr5", "chr1", "chr4", "chr3"])   # one seek per record
    for k in ("chr1", "chr3", "chr4", "chr5"):
        assert np.array_equal(idx.get_codes_from_coords(k, 0, len(recs[k])), sequence_to_codes(recs[k])), k
    with pytest.raises(KeyError):
        PackedGenome.from_fasta(path, chroms=["chrX"])


def test_encoding_semantics():
    enc = sequence_to_encoding("ACGTNacgtR")
    assert enc.dtype == np.float32 and enc.shape == (10, 4)
    assert np.array_equal(enc[:4], np.eye(4, dtype=np.float32)) and np.array_equal(enc[5:9], np.eye(4, dtype=np.float32))
    assert np.all(enc[4] == 0.25) and np.all(enc[9] == 0.25)
    g = PackedGenome({"c": sequence_to_codes("AACCGGTTNN")})
    plus = g.get_encoding_from_coords("c", 2, 8)
    minus = g.get_encoding_from_coords("c", 2, 8, strand="-")
    assert np.array_equal(minus, plus[::-1, ::-1])                       # selene_utils2.py:227-228
    assert np.array_equal(codes_to_encoding(revcomp_codes(g.get_codes_from_coords("c", 2, 8))), minus)
    padded = g.get_encoding_from_coords("c", -3, 12, pad=True)
    assert padded.shape == (15, 4) and np.all(padded[:3] == 0.25) and np.all(padded[13:] == 0.25)
    assert np.array_equal(padded[3:13], g.get_encoding_from_coords("c", 0, 10))
    with pytest.raises(AssertionError):
        g.get_encoding_from_coords("c", 5, 11)                           # beyond the end without pad (selene_utils2.py:257)
    _, unk = g.get_encoding_from_coords_check_unk("c", 8, 10)
    assert unk                                                           # looks at the first row only (:264-272)
    _, unk = g.get_encoding_from_coords_check_unk("c", 7, 10)
    assert not unk


def test_padded_window_semantics_and_the_reference_errors():
    """pad=True fills 0.25 beyond either end as long as the window at least TOUCHES the chromosome; a window that lies
    entirely beyond an end, or has negative length, fails the reference's length assert (selene_utils2.py:261; pinned by G26
    below - round 1's all-padding answer for such windows was an extension the reference does not have)."""
    from orca_amd.genome import N_CODE
    g = PackedGenome({"c": np.array([0, 1, 2, 3, 0, 1], dtype=np.uint8)})
    assert list(g.get_codes_from_coords("c", -2, 8, pad=True)) == [N_CODE] * 2 + [0, 1, 2, 3, 0, 1] + [N_CODE] * 2
    assert list(g.get_codes_from_coords("c", 4, 9, pad=True)) == [0, 1] + [N_CODE] * 3
    assert list(g.get_codes_from_coords("c", 6, 10, pad=True)) == [N_CODE] * 4          # touches the end
    enc = g.get_encoding_from_coords("c", -3, 0, pad=True)                              # touches the start
    assert enc.shape == (3, 4) and np.all(enc == 0.25)
    for bad in ((-100, -10), (10, 14), (7, 7), (-3, -3), (5, 2)):
        with pytest.raises(AssertionError):
            g.get_codes_from_coords("c", *bad, pad=True)
    for bad in ((-1, 3), (2, 7), (5, 2)):
        with pytest.raises(AssertionError):      # the reference's error type (selene_utils2.py:257), raised explicitly
            g.get_codes_from_coords("c", *bad)


def _g26():
    import os
    return np.load(os.path.join(os.path.dirname(__file__), "golden", "G26_genome.npz"))


def _g26_queries(d):
    for i in range(d["q_chrom"].shape[0]):
        yield i, (str(d["q_chrom"][i]), int(d["q_start"][i]), int(d["q_end"][i]), str(d["q_strand"][i]), bool(d["q_pad"][i]))


@pytest.mark.parametrize("store", ["packed", "twobit", "fasta", "fasta_indexed_twobit"])
def test_genome_store_against_the_reference_memmap_genome(store, tmp_path):
    """SURVEY 8(f2) pinned: tests/golden/G26_genome.npz holds what the REFERENCE's `MemmapGenome.get_encoding_from_coords` and
    `get_encoding_from_coords_check_unk` (selene_utils2.py:186-272; tools/make_golden.py --genome) return for 1 269 queries on a
    five-chromosome genome with lower case, IUPAC symbols and N runs: both strands and '.', pad on / off, windows over either
    end, touching an end, entirely outside, of zero and negative length.  Each store must give the same rows (exactly: the
    values are 0, 0.25 and 1), raise where the reference's asserts fire and nowhere else, and reproduce `_check_unk`'s two
    quirks (the window is always padded: `pad=strand`, :271; the flag reads the first row only; no first row: IndexError)."""
    from orca_amd.genome import TwoBitGenome
    d = _g26()
    recs = {str(c): str(d["seq_" + str(c)]) for c in d["chrs"]}
    if store.startswith("fasta"):
        path = str(tmp_path / "g26.fa")
        _write_fasta(path, recs, width=61)
        if store == "fasta":
            g = PackedGenome.from_fasta(path)
        else:
            PackedGenome.write_fai(path)
            g = TwoBitGenome.from_fasta(path, chroms=list(recs))
    else:
        g = PackedGenome({c: sequence_to_codes(s) for c, s in recs.items()})
        if store == "twobit":
            g = TwoBitGenome.from_packed(g)
    assert g.get_chrs() == [str(c) for c in d["chrs"]]                      # the reference sorts the names (:100)
    rows, offs = d["rows"], d["row_offsets"]
    n_ok = n_err = 0
    for i, (c, a, b, strand, pad) in _g26_queries(d):
        if b - a > 30000 and store != "packed" and i % 3:                   # the long windows once per store is enough
            continue
        want = str(d["status"][i])
        if want == "ok":
            got = g.get_encoding_from_coords(c, a, b, strand=strand, pad=pad)
            assert got.dtype == np.float32 and got.shape == (b - a, 4), (i, c, a, b, strand, pad)
            assert np.array_equal(got, rows[offs[i]:offs[i + 1]]), (i, c, a, b, strand, pad)
            n_ok += 1
        else:
            assert want == "AssertionError"
            with pytest.raises(AssertionError):
                g.get_encoding_from_coords(c, a, b, strand=strand, pad=pad)
            n_err += 1
        uwant = str(d["unk_status"][i])
        if uwant == "ok":
            enc, unk = g.get_encoding_from_coords_check_unk(c, a, b, strand=strand, pad=pad)
            assert unk == bool(d["unk_flag"][i]) and enc.shape == (b - a, 4), (i, c, a, b, strand, pad)
        else:
            with pytest.raises(AssertionError if uwant == "AssertionError" else IndexError):
                g.get_encoding_from_coords_check_unk(c, a, b, strand=strand, pad=pad)
    assert n_ok > 250 and n_err > 130


def test_two_bit_store_is_bit_exact_with_the_one_byte_store(fasta):
    """2 bits per base + N bit-mask, same query API; every window equals the 1-byte store's, on FASTA with
    lower case, IUPAC codes and N runs, on both strands, with and without padding, at every alignment of the window."""
    from orca_amd.genome import TwoBitGenome, pack_2bit, unpack_2bit
    path, recs = fasta
    g1 = PackedGenome.from_fasta(path)
    g2 = TwoBitGenome.from_fasta(path)
    assert g2.get_chr_lens() == g1.get_chr_lens()
    assert g2.nbytes() <= sum(len(v) for v in recs.values()) * 3 / 8 + 2 * len(recs) + 2
    rs = np.random.RandomState(3)
    for chrom, n in g1.get_chr_lens():
        for _ in range(40 if n else 1):
            a, b = sorted(rs.randint(-9, n + 10, 2))
            a, b = min(a, n), max(b, 0)          # a padded window has to touch the chromosome (G26)
            for strand in "+-":
                np.testing.assert_array_equal(g2.get_codes_from_coords(chrom, a, b, strand, pad=True), g1.get_codes_from_coords(chrom, a, b, strand, pad=True))
            a, b = max(a, 0), min(b, n)
            if a <= b:
                np.testing.assert_array_equal(g2.get_encoding_from_coords(chrom, a, b), g1.get_encoding_from_coords(chrom, a, b))
    with pytest.raises(AssertionError):
        g2.get_codes_from_coords("chr1", 5, 5000)
    codes = rs.randint(0, 6, 1001).astype(np.uint8)          # 4 and 5: both read back as N
    two, mask = pack_2bit(codes)
    assert two.shape[0] == 251 and mask.shape[0] == 126
    np.testing.assert_array_equal(unpack_2bit(two, mask, 0, 1001), np.minimum(codes, 4))
