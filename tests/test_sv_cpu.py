"""SV-screen host logic (orca_amd/sv.py) on CPU: coordinate helpers vs values produced by the reference
(tests/golden/G10_coords.npz), and the piece algebra vs explicit editing of a small chromosome string."""
import numpy as np

from orca_amd import sv
from tests.util import golden


def test_coord_clip_and_round_match_reference():
    rows = golden("G10_coords.npz")["rows"]
    for pos, chrlen, clip32, rnd, clip256 in rows.tolist():
        assert sv.coord_clip(pos, chrlen) == clip32
        assert sv.coord_round(pos) == rnd
        assert sv.coord_clip(pos, max(chrlen, 260_000_000), binsize=1024000, window_radius=128000000) == clip256


def _edit(chrom, v):
    s, e = v.start, v.end
    comp = lambda a: np.where(a < 4, 3 - a, a).astype(np.uint8)
    if v.kind == "del":
        return np.concatenate([chrom[:s], chrom[e:]])
    if v.kind == "dup":
        return np.concatenate([chrom[:e], chrom[s:e], chrom[e:]])
    return np.concatenate([chrom[:s], comp(chrom[s:e][::-1]), chrom[e:]])


def test_allele_windows_equal_explicit_editing():
    rs = np.random.RandomState(3)
    chrlen, L = 5000, 3200
    chrom = rs.randint(0, 5, chrlen).astype(np.uint8)
    for kind in ("del", "dup", "inv"):
        for _ in range(50):
            a = int(rs.randint(10, chrlen - 400))
            v = sv.SV(kind, a, a + int(rs.randint(1, 380)))
            edited = _edit(chrom, v)
            pieces = sv.allele_pieces(v, chrlen)
            assert sum(p[1] for p in pieces) == len(edited)
            for w0 in (0, 7, len(edited) - L, int(rs.randint(0, len(edited) - L))):
                got = sv.assemble_codes(chrom, sv.window_pieces(pieces, w0, L))
                assert np.array_equal(got, edited[w0: w0 + L]), (kind, v, w0)


def test_sv_windows_are_centred_and_full_length():
    chrlen = 40_000_000
    off = [v for v in sv.synth_svs(64, chrlen) if v.start % 4000 or v.end % 4000]
    assert len(off) >= 60                                     # the default set is unaligned (log-uniform sizes, no alignment)
    old = sv.synth_svs(64, chrlen, align=4000)                # rounds 3-5's set: the same draws rounded down to the 4 kb grid
    assert [v.kind for v in old] == [v.kind for v in sv.synth_svs(64, chrlen)] and old[0] == sv.SV("del", 30_172_000, 33_840_000)
    for v in sv.synth_svs(64, chrlen) + old:
        assert v.kind in ("del", "dup", "inv") and 0 < v.start < v.end < chrlen and 10_000 - 4000 < v.end - v.start <= 5_000_000
        rp, rw, rm, ap, aw, am = sv.sv_windows(v, chrlen)
        assert sum(p[1] for p in rp) == sum(p[1] for p in ap) == sv.WINDOW
        assert rw - 16_000_000 <= rm <= rw + 16_000_000 and aw - 16_000_000 <= am <= aw + 16_000_000
        assert rp == [(rw - 16_000_000, sv.WINDOW, "+")]
