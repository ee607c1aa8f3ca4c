"""Incremental allele-window encoding (orca_amd/sv.py: ChromEncodings / reuse_plan / encode_window) on the CPU with a toy Encoder that
has the real one's structure - 4 kb bins, a receptive reach of 104 016 bases beyond the bin, zero padding at the ends of whatever
sequence it is given, reverse complement by index / code flip - in exact integer arithmetic: the windows assembled from chromosome-level
encodings must EQUAL the windows encoded whole, for deletions, duplications and inversions, aligned to the 4 kb grid or not."""
import numpy as np
import pytest
import torch

from orca_amd import sv

R = sv.RF_BP


class ToyNet0:
    """out[0, j] = sum of t[code] over the bases within reach of bin j, out[1, j] = the same weighted by the base's offset from the
    bin start (orientation- and phase-sensitive), out[2:] = 0; float64 holds the integers exactly."""
    TAB = np.array([3, 5, 11, 17, 1], dtype=np.int64)

    def __init__(self):
        self.calls = []

    def forward_codes(self, codes, reverse=False, bin_lo=0, bin_hi=0, chunk_bp=0, out=None):
        c = codes[0].numpy()
        if reverse:
            c = c[::-1]
            c = np.where(c < 4, 3 - c, c)
        L = c.shape[0]
        nb = L // 4000
        hi = nb if bin_hi <= 0 else bin_hi
        v = self.TAB[c]
        pos = np.arange(L, dtype=np.int64)
        s0 = np.concatenate([[0], np.cumsum(v)])
        s1 = np.concatenate([[0], np.cumsum(v * pos)])
        res = np.zeros((1, 128, hi - bin_lo))
        for j in range(bin_lo, hi):
            a, b = max(0, 4000 * j - R), min(L, 4000 * (j + 1) + R)
            res[0, 0, j - bin_lo] = s0[b] - s0[a]
            res[0, 1, j - bin_lo] = (s1[b] - s1[a]) - 4000 * j * (s0[b] - s0[a])
        self.calls.append((bool(reverse), bin_lo, hi))
        t = torch.from_numpy(res)
        if out is not None:
            out.copy_(t)
            return out
        return t


C, LEN = 2_400_000, 1_600_000


def _window_whole(net, codes, pieces):
    w = sv.assemble_codes(codes, pieces)
    return torch.cat([net.forward_codes(w[None], reverse=False), net.forward_codes(w[None], reverse=True)], dim=0), w


@pytest.mark.parametrize("variant", [sv.SV("del", 1_000_000, 1_200_000), sv.SV("dup", 900_000, 1_140_000), sv.SV("inv", 700_000, 1_300_000),
                                     sv.SV("inv", 1_000_000, 1_012_000), sv.SV("del", 1_001_234, 1_203_210), sv.SV("inv", 801_111, 1_399_007),
                                     sv.SV("dup", 1_100_000, 1_108_000)])
def test_incremental_windows_equal_whole_windows(variant):
    rs = np.random.RandomState(3)
    codes = torch.from_numpy(rs.randint(0, 5, C).astype(np.uint8))
    net = ToyNet0()
    cache = sv.ChromEncodings(net, codes, max_entries=16)
    rp, rw, rm, ap, aw, am = sv.sv_windows(variant, C, LEN)
    nb = LEN // 4000
    for pieces in (rp, ap):
        ref, w = _window_whole(ToyNet0(), codes, pieces)
        out = torch.full((2, 128, nb), -1.0, dtype=torch.float64)
        n = sv.encode_window(cache, pieces, w, out)
        assert torch.equal(out, ref), (variant, pieces)
        assert 2 * 2 * sv.RF_BINS <= n < 2 * nb                      # the window ends are always encoded, never everything
    # without chromosome encodings (phases not held, none may be built): everything is encoded, same result
    cache2 = sv.ChromEncodings(ToyNet0(), codes)
    out = torch.full((2, 128, nb), -1.0, dtype=torch.float64)
    assert sv.encode_window(cache2, ap, sv.assemble_codes(codes, ap), out, build=False) == 2 * nb
    assert torch.equal(out, _window_whole(ToyNet0(), codes, ap)[0])


def test_reuse_plan_and_phases():
    v = sv.SV("inv", 700_000, 1_300_000)
    rp, rw, rm, ap, aw, am = sv.sv_windows(v, C, LEN)
    plan = sv.reuse_plan(ap, C, LEN // 4000)
    assert [p[2] for p in plan] == ["+", "-", "+"]                      # flank, inverted interior (from the other strand's encoding), flank
    assert all(hi - lo > 0 and lo >= sv.RF_BINS for lo, hi, _, _ in plan)
    rplan = sv.reuse_plan(sv.revcomp_pieces(ap), C, LEN // 4000)
    assert [p[2] for p in rplan] == ["-", "+", "-"]
    want = sv.needed_phases([v, sv.SV("del", 1_001_234, 1_203_210)], C, LEN)
    assert all(0 <= ph < 4000 for _, ph in want) and sum(want.values()) >= 12


def _driver_views(variant, chrlen, length):
    """The views a structural-variant driver lists (orca_amd/sv_drivers.py): reference windows anchored at the variant's two ends, then
    the alternative allele(s) - as 4-tuple pieces (chrom, start, length, strand) of chromosome "c"."""
    r = length // 2
    s, e = variant.start, variant.end
    ref = [[("c", sv.coord_clip(a, chrlen, window_radius=r) - r, length, "+")] for a in (s, e)]
    alt_pieces = sv.allele_pieces(variant, chrlen)
    alt_len = sum(p[1] for p in alt_pieces)
    anchors = {"del": [s], "dup": [e], "inv": [s, e]}[variant.kind]
    alts = [[("c",) + p for p in sv.window_pieces(alt_pieces, sv.coord_clip(a, alt_len, window_radius=r) - r, length)] for a in anchors]
    return ref, alts


@pytest.mark.parametrize("variant", [sv.SV("del", 1_001_234, 1_203_210), sv.SV("inv", 801_111, 1_399_007), sv.SV("dup", 900_000, 1_140_000),
                                     sv.SV("inv", 1_000_000, 1_012_000), sv.SV("dup", 1_100_777, 1_108_001)])
def test_alternative_alleles_from_the_reference_views_of_the_same_call(variant):
    """Round 5 (sv_drivers._run_views): no chromosome encoding at all - the reference views of a call are encoded whole and kept as
    segments (both strands), the alternative allele takes its bins from them (the inverted segment from the OTHER strand's segment of the
    view whose phase agrees) and only window ends and junctions are encoded.  Exact on the integer toy Encoder, coordinates off the grid."""
    rs = np.random.RandomState(4)
    codes = torch.from_numpy(rs.randint(0, 5, C).astype(np.uint8))
    net = ToyNet0()
    store = sv.GenomeEncodings(net, lambda c: codes, {"c": C})
    nb = LEN // 4000
    refs, alts = _driver_views(variant, C, LEN)
    own = {}
    for pieces in refs:
        w = sv.assemble_codes(codes, [p[1:] for p in pieces])
        out = torch.full((2, 128, nb), -1.0, dtype=torch.float64)
        sv.encode_windows(store, [pieces], w[None], out, build=False, extra=own)
        assert torch.equal(out, _window_whole(ToyNet0(), codes, [p[1:] for p in pieces])[0])
        _, start, ln, _ = pieces[0]
        own.setdefault("c", []).extend([["+", start, out[0].clone()], ["-", C - start - ln, out[1].clone()]])
    for pieces in alts:
        w = sv.assemble_codes(codes, [p[1:] for p in pieces])
        out = torch.full((2, 128, nb), -1.0, dtype=torch.float64)
        n = sv.encode_windows(store, [pieces], w[None], out, build=False, extra=own)
        assert torch.equal(out, _window_whole(ToyNet0(), codes, [p[1:] for p in pieces])[0]), (variant, pieces)
        assert n < 0.6 * 2 * nb, (variant, n, 2 * nb)                # most of the alternative allele came from the reference views
    assert store.builds == 0


def test_segments_persist_and_a_repeated_phase_gets_a_chromosome_encoding():
    rs = np.random.RandomState(5)
    codes = torch.from_numpy(rs.randint(0, 5, C).astype(np.uint8))
    net = ToyNet0()
    store = sv.GenomeEncodings(net, lambda c: codes, {"c": C})
    nb = LEN // 4000
    ce = store.of("c")
    assert store.of("ins0") is None and ce.auto_threshold() == 2
    ce.miss_bins = 50             # (the toy window is 400 bins; 1 000 of a real window's 8 000)
    pieces = [("c", 400_000, LEN, "+")]
    w = sv.assemble_codes(codes, [p[1:] for p in pieces])
    out = torch.zeros((2, 128, nb), dtype=torch.float64)
    assert sv.encode_windows(store, [pieces], w[None], out, build="auto") == 2 * nb          # nothing held: one miss per strand counted
    ce.add_segment("+", 400_000, out[0].clone())
    ce.add_segment("-", C - 400_000 - LEN, out[1].clone())
    # the same window again: only its ends are encoded; a window shifted by 40 bins: the overlap comes from the segment
    out2 = torch.zeros_like(out)
    assert sv.encode_windows(store, [pieces], w[None], out2, build="auto") == 2 * 2 * sv.RF_BINS and torch.equal(out2, out)
    shifted = [("c", 560_000, LEN, "+")]
    w3 = sv.assemble_codes(codes, [p[1:] for p in shifted])
    out3 = torch.zeros_like(out)
    n3 = sv.encode_windows(store, [shifted], w3[None], out3, build="auto")
    assert n3 < 2 * nb and torch.equal(out3, _window_whole(ToyNet0(), codes, [p[1:] for p in shifted])[0])
    assert store.builds == 0
    # a second window-sized miss at the same phase elsewhere on the chromosome: the whole chromosome is encoded at that phase
    far = [("c", 4_000, LEN, "+")]
    out4 = torch.zeros_like(out)
    sv.encode_windows(store, [far], sv.assemble_codes(codes, [p[1:] for p in far])[None], out4, build="auto")
    assert store.builds == 2 and torch.equal(out4, _window_whole(ToyNet0(), codes, [p[1:] for p in far])[0])


@pytest.mark.parametrize("grid,reach,kw", [(16, 351, {}), (80, 1631, dict(margin=sv.S4_MARGIN_BP, grid=sv.S4_GRID, pad=sv.S4_PAD_BP, min_snippet=sv.S4_MIN_SNIPPET_BP))])
def test_stage3_plan_is_exact_on_a_toy_front(grid, reach, kw):
    """`sv.s3_plan` (the stage-3 cache's route, round 6) on CPU: a toy "front" - a LINEAR integer filter with the real reach (351 bases
    either side of a 16-base cell, zero padded at the ends of whatever sequence it runs on) and MaxPool1d(5) - so that equality is exact.
    For windows of one piece, deletions, inversions ('-' pieces from the other strand's cache), short duplicated pieces, both strands, and
    windows at the chromosome's ends: every pooled position comes from exactly one source - a cache entry of the right (strand, phase mod 16)
    at the right offset, or a snippet of the assembled window - and equals the front on the whole assembled window.  Second parameter set:
    the same plan one level up (sv.Stage4Cache: stage 4's output on the 80-base grid, reach 1 631 bases, phases mod 80, snippets on the 400-base grid)."""
    rs = np.random.RandomState(0)
    C, L = 400_000, 160_000
    chrom = rs.randint(0, 4, C).astype(np.int64)
    wts = rs.randint(-3, 4, size=(reach * 2 + grid,)).astype(np.int64)

    def revcomp(c):
        return (3 - c)[::-1]

    def stage3(codes):
        x = np.concatenate([np.zeros(reach, np.int64), codes + 1, np.zeros(reach, np.int64)])
        idx = np.arange(len(codes) // grid)[:, None] * grid + np.arange(len(wts))[None, :]
        return (x[idx] * wts[None, :]).sum(1)

    def pool5(v):
        return v[: len(v) // 5 * 5].reshape(-1, 5).max(1)

    cache = {}

    def entry(strand, phase, region):      # (planes, strand coordinate of position 0): what sv.Stage3Cache.get / _origin keep
        if (strand, phase, region) not in cache:
            lo, hi = region if strand == "+" else (C - region[1], C - region[0])
            e0 = lo + (phase - lo) % grid
            n = (hi - e0) // (5 * grid) * (5 * grid)
            cache[(strand, phase, region)] = (stage3((chrom if strand == "+" else revcomp(chrom))[e0: e0 + n]), e0)
        return cache[(strand, phase, region)]

    def check(pieces, region=None):
        for rev in (False, True):
            pcs = sv.revcomp_pieces(pieces) if rev else pieces
            win = np.concatenate([chrom[s: s + n] if st == "+" else revcomp(chrom[s: s + n]) for s, n, st in pcs])
            ref = pool5(stage3(win))
            takes, snips = sv.s3_plan(pcs, C, L, regions=region, **kw)
            got, cov = np.zeros(L // (5 * grid), np.int64), np.zeros(L // (5 * grid), int)
            for m_lo, m_hi, _, strand, phase, c in takes:
                e, e0 = entry(strand, phase, region or (0, C))
                j0 = (c - e0) // grid
                assert phase == c % grid and (c - e0) % grid == 0 and j0 >= 0 and j0 + 5 * (m_hi - m_lo) <= len(e)
                got[m_lo:m_hi] = pool5(e[j0: j0 + 5 * (m_hi - m_lo)])
                cov[m_lo:m_hi] += 1
            for ga, gb, b0, nb, skip in snips:
                assert nb % (5 * grid) == 0 and b0 % (5 * grid) == 0 and 0 <= b0 and b0 + nb <= L and nb >= min(L, kw.get("min_snippet", sv.S3_MIN_SNIPPET_BP))
                got[ga:gb] = pool5(stage3(win[b0: b0 + nb]))[skip: skip + gb - ga]
                cov[ga:gb] += 1
            assert (cov == 1).all() and (got == ref).all(), (pieces, rev)
            if len(snips) > 1 and snips[0][0] == 0 and snips[-1][1] == L // (5 * grid):
                # the strand's snippets as ONE run (sv._s4_encode): concatenated in strand order, the window's ends first and last - the seams
                # between snippets lie inside the pads nobody reads
                cat = pool5(stage3(np.concatenate([win[b0: b0 + nb] for _, _, b0, nb, _ in snips])))
                off = 0
                for ga, gb, b0, nb, skip in snips:
                    assert (cat[off // (5 * grid) + skip: off // (5 * grid) + skip + gb - ga] == ref[ga:gb]).all(), (pieces, rev, ga)
                    off += nb
            assert region is not None or sum(sn[3] for sn in snips) < 0.3 * L

    for trial in range(24):
        s = int(rs.randint(1000, C - L - 60000))
        if trial % 4 == 0:
            pieces = [(s, L, "+")]
        elif trial % 4 == 1:
            a, d = int(rs.randint(20000, L - 20000)), int(rs.randint(1, 50000))
            pieces = [(s, a, "+"), (s + a + d, L - a, "+")]
        elif trial % 4 == 2:
            a, b = int(rs.randint(20000, 60000)), int(rs.randint(1, 40000))
            pieces = [(s, a, "+"), (s + a, b, "-"), (s + a + b, L - a - b, "+")]
        else:
            a, b = int(rs.randint(20000, 60000)), 5 * int(rs.randint(1, 300))
            pieces = [(s, a, "+"), (s + a - b, b, "+"), (s + a, L - a - b, "+")]
        check(pieces)
    check([(0, L, "+")])
    check([(C - L, L, "+")])
    # a cache that holds a REGION of the chromosome only: what lies outside goes through the front
    check([(100_000, L, "+")], region=(120_000, 250_000))
    check([(100_000, 70_000, "+"), (200_003, 40_000, "-"), (290_000, 50_000, "+")], region=(150_000, 330_000))


def test_drivers_store_caches_a_chromosome_a_locus_at_a_time():
    """`GenomeEncodings.stage3_caches` (round 6): 1 KB of HBM per base means a real chromosome (chr1: 248 Mb) cannot be cached whole under the
    store's budget - after `s3_after` window strands nobody served, the REGION those strands spanned (+- 4 Mb) is cached; strands inside it
    are served, a run of strands outside it makes the region grow (or move, when both do not fit), least recently used chromosomes give way.
    The builder is substituted: no GPU here."""
    class FakeNet:
        def two_part_ok(self):
            return True

    class FakeCache:
        poisoned = False

        def __init__(self, region):
            self.region = region

    built = []

    def make(ce, region, need):
        built.append((ce.C, region, need))
        return FakeCache(region)

    genc = sv.GenomeEncodings(FakeNet(), lambda c: None, {"chr1": 248_000_000, "chr2": 242_000_000, "chrS": 40_000_000})
    genc.s3_after, genc.s3_budget = 4, 120e9
    window = lambda chrom, start: [(chrom, start, 32_000_000, "+")]
    # three unserved strands: nothing yet; the fourth builds the hull of the four windows +- 4 Mb
    for k in range(3):
        assert genc.stage3_caches(window("chr1", 100_000_000 + 1_000_003 * k), True, make) == {} and not built
    got = genc.stage3_caches(window("chr1", 103_000_009), True, make)
    assert built == [(248_000_000, (96_000_000, 139_000_080), sv.Stage3Cache.bytes_needed(43_000_080))] and got["chr1"].region == (96_000_000, 139_000_080)
    # inside the region: served, nothing counted; build=False never builds
    assert genc.stage3_caches(window("chr1", 101_234_567), True, make)["chr1"] is got["chr1"] and genc.of("chr1").s3_misses == 0
    assert genc.stage3_caches(window("chr2", 5_000_000), False, make) == {} and len(built) == 1
    # a run of windows beyond the region: the old cache still serves those it covers by half, then the region grows to hold both
    for k in range(3):
        out = genc.stage3_caches(window("chr1", 120_000_000 + k), True, make)
        assert out["chr1"] is got["chr1"] and len(built) == 1
    grown = genc.stage3_caches(window("chr1", 120_000_003), True, make)["chr1"]
    assert grown.region == (96_000_000, 156_000_080) and len(built) == 2
    # far away: hull with the held region = 200 Mb > budget (117 Mb): the cache MOVES to the new locus
    for k in range(4):
        out = genc.stage3_caches(window("chr1", 200_000_000 + 7 * k), True, make)
    assert out["chr1"].region == (196_000_000, 236_000_080) and len(built) == 3
    # a whole small chromosome, and the budget across chromosomes: chr1's 40 Mb + chrS's 40 Mb fit; chr2's 100 Mb (102 of the 120 GB) evicts both
    for k in range(4):
        out = genc.stage3_caches(window("chrS", 1_000 * k + (8_000_000 if k & 1 else 0)), True, make)
    assert out["chrS"].region == (0, 40_000_000) and genc.of("chr1").stage3 is not None
    for k in range(4):
        out = genc.stage3_caches([("chr2", 10_000_000 + 20_000_000 * k, 32_000_000, "-")], True, make)
    assert out["chr2"].region == (6_000_000, 106_000_000)
    held = {c for c in ("chr1", "chr2", "chrS") if genc.of(c).stage3 is not None}
    assert held == {"chr2"}, held
    # a cache whose entries were dropped because the range check of the pass that built it fired is abandoned, and not rebuilt at once
    genc.of("chr2").stage3.poisoned = True
    n_built = len(built)
    for k in range(8):
        assert genc.stage3_caches([("chr2", 10_000_000, 32_000_000, "+")], True, make) == {}
    assert genc.of("chr2").stage3 is None and len(built) == n_built
    # pieces of anything that is not a chromosome of the genome (an inserted string, padding) do not count
    assert genc.stage3_caches([("__pad__", 0, 32_000_000, "+")], True, make) == {}
