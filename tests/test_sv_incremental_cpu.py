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
