"""The incremental SV screen on the MI355X (orca_amd/sv.py, BASELINE configs[4]): allele windows assembled from chromosome-level Encoder
outputs + locally encoded ends / junctions, ref and alt decoded as one batch - against the reference's cost structure (two whole
`genomepredict` calls per variant, `incremental=False`) on the same variants."""
import numpy as np
import pytest
import torch

from orca_amd import orca_models, sv

pytestmark = pytest.mark.gpu
CHR = 40_000_000


@pytest.fixture(scope="module")
def setup(cuda):
    g = torch.Generator(device=cuda).manual_seed(11)
    genome = torch.randint(0, 4, (CHR,), device=cuda, generator=g, dtype=torch.uint8)
    genome[7_000_000:7_003_000] = 4                                     # an N run inside the windows
    return orca_models.H1esc(synthetic_seed=0), genome


VARIANTS = [sv.SV("del", 18_000_000, 19_200_000), sv.SV("dup", 21_004_000, 21_500_000), sv.SV("inv", 15_000_000, 18_000_000),
            sv.SV("inv", 20_000_000, 20_012_000), sv.SV("del", 17_001_234, 17_803_210)]


def test_window_encodings_equal_whole_encodings(setup):
    """encode_window == the Encoder on the assembled window.  Not bit for bit: the short bin-range calls run other kernel instantiations
    than a 32 Mb / 40 Mb sequence (tile shapes chosen by length: another summation order) - measured max-abs 2.3e-6 on encodings of
    range 0..8; the bound here is 1e-5, an order below what the 1e-4 parity bar of the maps needs."""
    model, genome = setup
    cache = sv.ChromEncodings(model.net0, genome, max_entries=16)
    for v in VARIANTS:
        rp, rw, rm, ap, aw, am = sv.sv_windows(v, CHR)
        for pieces in (rp, ap):
            w = sv.assemble_codes(genome, pieces)
            whole = torch.cat([model.net0.forward_codes(w[None], reverse=False), model.net0.forward_codes(w[None], reverse=True)], dim=0)
            out = torch.full((2, 128, 8000), float("nan"), device=genome.device)
            n = sv.encode_window(cache, pieces, w, out)
            assert n < 2 * 8000 // 10, (v, n)
            assert float((out - whole).abs().max()) <= 1e-5, v


def test_screen_incremental_equals_two_genomepredict_calls(setup):
    model, genome = setup
    stats = {}
    inc = sv.sv_screen([model], genome, VARIANTS, CHR, min_uses=1, stats=stats)
    full = sv.sv_screen([model], genome, VARIANTS, CHR, incremental=False)
    assert stats["bins_encoded"] < 0.1 * stats["bins_total"] and stats["chromosome_encodings"] >= 4
    for i in range(len(VARIANTS)):
        for allele in ("ref", "alt"):
            a, b = inc[i][allele], full[i][allele]
            assert a["start_coords"] == b["start_coords"] and a["end_coords"] == b["end_coords"] and a["chr"] == b["chr"]
            for j in range(6):
                assert float(np.abs(a["predictions"][0][j] - b["predictions"][0][j]).max()) <= 2e-5, (VARIANTS[i], allele, j)


def test_screen_falls_back_when_phases_are_not_held(setup):
    """A lone variant off the 4 kb grid does not justify chromosome encodings of its phases (min_uses): its windows are encoded whole -
    same maps."""
    model, genome = setup
    v = [VARIANTS[4]]
    stats = {}
    inc = sv.sv_screen([model], genome, v, CHR, min_uses=100, stats=stats)
    full = sv.sv_screen([model], genome, v, CHR, incremental=False)
    assert stats["chromosome_encodings"] == 0 and stats["bins_encoded"] == stats["bins_total"]
    for allele in ("ref", "alt"):
        for j in range(6):
            # (not 0: the four strands of a variant are 32 000 Encoder2 positions - its split-operand kernels - where a window's two run the exact-fp32 ones)
            assert float(np.abs(inc[0][allele]["predictions"][0][j] - full[0][allele]["predictions"][0][j]).max()) <= 2e-5


def test_local_encodes_on_a_context_pool_equal_the_one_stream_path(setup):
    """engine.ContextPool: the (window, strand, range) jobs of encode_windows dealt to four auxiliary contexts (own stream, workspace,
    edge scratch, range flag) - the same kernels on the same inputs, so the encodings are equal bit for bit; repeated, so that a
    race between overlapping calls (a shared buffer, a missing fork / join edge) would show; and the screen gives the same maps."""
    from orca_amd import engine
    model, genome = setup
    cache = sv.ChromEncodings(model.net0, genome, max_entries=16)
    pool = engine.context_pool(genome.device, 4)
    for v in VARIANTS[:3]:
        rp, rw, rm, ap, aw, am = sv.sv_windows(v, CHR)
        codes = torch.stack([sv.assemble_codes(genome, rp), sv.assemble_codes(genome, ap)])
        one = torch.full((4, 128, 8000), float("nan"), device=genome.device)
        n1 = sv.encode_windows(cache, [rp, ap], codes, one)
        for _ in range(3):
            out = torch.full((4, 128, 8000), float("nan"), device=genome.device)
            n2 = sv.encode_windows(cache, [rp, ap], codes, out, pool=pool)
            assert n1 == n2 and torch.equal(out, one), v
    assert not pool.take_overflow()
    a = sv.sv_screen([model], genome, VARIANTS[:2], CHR, min_uses=1, streams=4)
    b = sv.sv_screen([model], genome, VARIANTS[:2], CHR, min_uses=1, streams=0)
    for i in range(2):
        for allele in ("ref", "alt"):
            for j in range(6):
                assert np.array_equal(a[i][allele]["predictions"][0][j], b[i][allele]["predictions"][0][j])


def test_pipelined_screen_redoes_a_unit_whose_encodes_left_the_fp16_range(setup, monkeypatch):
    """The screen issues unit k + 1's local encodes under unit k's decoders; their range flags are read per unit and attributed to the unit
    they belong to.  Forced here (the pool reports a raised flag once, for the SECOND variant's encodes): that unit alone is redone in the
    range-safe arithmetic (bf16x3 Encoder calls, f32 Decoders: maps equal within the 1e-4 parity bar), the others are bit-identical."""
    from orca_amd import engine
    model, genome = setup
    base = sv.sv_screen([model], genome, VARIANTS[:3], CHR, min_uses=1, streams=4, group=1)     # unit = one variant
    calls = {"n": 0}
    real = engine.ContextPool.take_overflow

    def fake(self):
        calls["n"] += 1
        return bool(real(self)) or calls["n"] == 2          # call 1: after prep(0); call 2: unit 1's encodes, issued under unit 0's decoders
    monkeypatch.setattr(engine.ContextPool, "take_overflow", fake)
    with pytest.warns(UserWarning, match="fp16 range"):
        redo = sv.sv_screen([model], genome, VARIANTS[:3], CHR, min_uses=1, streams=4, group=1)
    for i in range(3):
        for allele in ("ref", "alt"):
            for j in range(6):
                a, b = redo[i][allele]["predictions"][0][j], base[i][allele]["predictions"][0][j]
                if i == 1:
                    assert float(np.abs(a - b).max()) <= 1e-4 and not np.array_equal(a, b)
                else:
                    assert np.array_equal(a, b)


def test_pipelined_screen_with_two_models(setup):
    """Units of the pipeline are (variant, model) pairs: two models on two variants - the entries carry both models' maps in model order and
    equal the one-stream screen's bit for bit."""
    model, genome = setup
    hff = orca_models.Hff(synthetic_seed=1)
    a = sv.sv_screen([model, hff], genome, VARIANTS[:2], CHR, min_uses=1, streams=4)
    b = sv.sv_screen([model, hff], genome, VARIANTS[:2], CHR, min_uses=1, streams=0)
    one = sv.sv_screen([hff], genome, VARIANTS[:1], CHR, min_uses=1, streams=0)
    for i in range(2):
        for allele in ("ref", "alt"):
            assert len(a[i][allele]["predictions"]) == 2 and len(a[i][allele]["normmats"]) == 2
            for m in range(2):
                for j in range(6):
                    assert np.array_equal(a[i][allele]["predictions"][m][j], b[i][allele]["predictions"][m][j])
    for j in range(6):
        assert np.array_equal(a[0]["alt"]["predictions"][1][j], one[0]["alt"]["predictions"][0][j])


def test_screen_groups_of_variants_per_decoder_batch_are_bit_identical(setup):
    """`group` variants go through Encoder2 and every decoder level as ONE batch of 4 x group maps (default 2: a Decoder forward is cheaper per
    map at B = 8 than at B = 4).  A map does not depend on the batch it is computed in: groups of 1, 2 (with an odd variant left over) and 3,
    one and two models, with and without the auxiliary contexts - the same bits, the same entries in the same order."""
    model, genome = setup
    hff = orca_models.Hff(synthetic_seed=1)
    base = sv.sv_screen([model], genome, VARIANTS[:3], CHR, min_uses=1, group=1, on_result=None)
    for group, streams in ((2, 4), (3, 4), (2, 0)):
        order = []
        got = {}
        sv.sv_screen([model], genome, VARIANTS[:3], CHR, min_uses=1, group=group, streams=streams, on_result=lambda i, e: (order.append(i), got.__setitem__(i, e)))
        assert order == [0, 1, 2]
        for i in range(3):
            assert got[i]["sv"] == VARIANTS[i]
            for allele in ("ref", "alt"):
                assert got[i][allele]["start_coords"] == base[i][allele]["start_coords"]
                for j in range(6):
                    assert np.array_equal(got[i][allele]["predictions"][0][j], base[i][allele]["predictions"][0][j]), (group, streams, i, allele, j)
    two1 = sv.sv_screen([model, hff], genome, VARIANTS[:3], CHR, min_uses=1, group=1)
    two2 = sv.sv_screen([model, hff], genome, VARIANTS[:3], CHR, min_uses=1, group=2)
    for i in range(3):
        for allele in ("ref", "alt"):
            assert len(two2[i][allele]["predictions"]) == 2
            for m in range(2):
                for j in range(6):
                    assert np.array_equal(two1[i][allele]["predictions"][m][j], two2[i][allele]["predictions"][m][j])


UNALIGNED = [sv.SV("del", 18_000_123, 19_200_777), sv.SV("dup", 21_004_001, 21_500_017), sv.SV("inv", 15_000_013, 18_000_002),
             sv.SV("inv", 20_000_003, 20_012_345), sv.SV("del", 17_001_234, 17_803_210), sv.SV("dup", 9_999_999, 10_020_020)]


@pytest.mark.parametrize("level", [3, 4])
def test_stage3_route_equals_the_whole_encoder(setup, level):
    """`sv.Stage3Cache` / `sv.Stage4Cache` (round 6; level 4 = stage 4 in the cache too: rows on the 80-base grid, the front + stage 4 on the snippets,
    stages 5-7 per window - what the screen and the drivers use): a window at an ARBITRARY base position - MaxPool1d(5) gather of stages 1-3 from the chromosome's cached planes
    (16 phases x 2 strands), the Encoder's front on its ends and junctions, stages 4-7 - against the Encoder on the assembled window, both
    strands, for reference and alternative alleles of deletions, duplications and inversions off the 4 kb grid (an N run inside).  Not bit for
    bit (the snippets run other tile instantiations than a 32 Mb sequence; the pool re-splits a stored value): the bound is 1e-5 on encodings of
    range 0..8, an order below what the maps' 1e-4 needs."""
    model, genome = setup
    Cache = sv.Stage3Cache if level == 3 else sv.Stage4Cache
    s3 = Cache(model.net0, genome)
    worst, front_bases = 0.0, 0
    for v in UNALIGNED:
        rp, rw, rm, ap, aw, am = sv.sv_windows(v, CHR)
        for pieces in (rp, ap):
            w = sv.assemble_codes(genome, pieces)
            for rev in (False, True):
                whole = model.net0.forward_codes(w[None], reverse=rev)[0]
                out = torch.full((128, 8000), float("nan"), device=genome.device)
                front_bases += s3.encode(sv.revcomp_pieces(pieces) if rev else pieces, w, rev, out)
                worst = max(worst, float((out - whole).abs().max()))
                assert worst <= 1e-5, (v, rev, worst)
    assert len(s3.entries) <= (32 if level == 3 else 160) and front_bases < 0.004 * len(UNALIGNED) * 4 * sv.WINDOW       # a few kb per end and junction, not windows
    # a cache of a REGION of the chromosome (a locus of a real one): what lies outside goes through the Encoder's front
    del s3
    part = Cache(model.net0, genome, region=(9_000_000, 33_000_000))
    rp, rw, rm, ap, aw, am = sv.sv_windows(UNALIGNED[0], CHR)
    w = sv.assemble_codes(genome, ap)
    for rev in (False, True):
        out = torch.full((128, 8000), float("nan"), device=genome.device)
        n = part.encode(sv.revcomp_pieces(ap) if rev else ap, w, rev, out)
        assert float((out - model.net0.forward_codes(w[None], reverse=rev)[0]).abs().max()) <= 1e-5 and 4_000_000 < n < 12_000_000, (rev, n)


def test_screen_off_the_grid_through_the_stage3_cache(setup):
    """The screen on variants at arbitrary base positions: no two windows share a 4 kb phase (every window of the plain incremental route is
    encoded whole), the stage-3 cache serves all of them - same dictionaries as two whole `genomepredict` calls per variant, maps within 2e-5;
    and the range-safe retry of a unit (forced) takes the whole-window route and gives the same maps."""
    from orca_amd import engine
    model, genome = setup
    stats = {}
    inc = sv.sv_screen([model], genome, UNALIGNED, CHR, stats=stats, stage3=True)
    full = sv.sv_screen([model], genome, UNALIGNED, CHR, incremental=False)
    assert stats["stage3_cache"]["entries"] == 160
    assert stats["bins_encoded"] < 0.01 * stats["bins_total"], stats
    for i in range(len(UNALIGNED)):
        for allele in ("ref", "alt"):
            a, b = inc[i][allele], full[i][allele]
            assert a["start_coords"] == b["start_coords"] and a["end_coords"] == b["end_coords"]
            for j in range(6):
                assert float(np.abs(a["predictions"][0][j] - b["predictions"][0][j]).max()) <= 2e-5, (UNALIGNED[i], allele, j)
    plain = {}
    sv.sv_screen([model], genome, UNALIGNED[:2], CHR, stats=plain, stage3=False)
    assert plain["stage3_cache"] is None and plain["bins_encoded"] == plain["bins_total"]
    with engine.force_safe_precision():
        assert not model.net0.two_part_ok()


def test_stage4_route_over_many_random_variants(setup):
    """A wider net for the route the screen uses: 36 variants of `synth_svs` (log-uniform sizes 10 kb - 5 Mb, arbitrary bases, all three kinds) plus
    the extremes - a variant hard against either end of the chromosome (windows clipped by `coord_clip`: a window end IS the chromosome end), a
    3-base deletion, a duplication shorter than the cache's margin (its middle piece has no interior: three pieces' worth of snippets merge) - both
    alleles, both strands, against the Encoder on the assembled window at 1e-5."""
    model, genome = setup
    cache = sv.Stage4Cache(model.net0, genome)
    assert cache.build_all() and len(cache.entries) == 160
    variants = sv.synth_svs(36, CHR, seed=4242) + [sv.SV("del", 1_000_003, 1_400_001), sv.SV("inv", 38_700_011, 39_900_007), sv.SV("del", 20_000_001, 20_000_004),
                                               sv.SV("dup", 12_345_678, 12_346_000), sv.SV("inv", 25_000_000, 25_000_900), sv.SV("dup", 30_000_001, 34_999_999)]
    worst = 0.0
    for v in variants:
        rp, rw, rm, ap, aw, am = sv.sv_windows(v, CHR)
        for pieces in (rp, ap):
            w = sv.assemble_codes(genome, pieces)
            for rev in (False, True):
                out = torch.full((128, 8000), float("nan"), device=genome.device)
                cache.encode(sv.revcomp_pieces(pieces) if rev else pieces, w, rev, out)
                d = float((out - model.net0.forward_codes(w[None], reverse=rev)[0]).abs().max())
                worst = max(worst, d)
                assert d <= 1e-5, (v, rev, d)
    print(f"stage-4 route vs the whole Encoder over {len(variants)} variants x 2 alleles x 2 strands: worst max-abs {worst:.3g}")
