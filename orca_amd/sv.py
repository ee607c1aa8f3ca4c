"""Structural-variant screen on packed genomes (BASELINE.json configs[4]).

The reference's SV drivers (`orca_predict.py` process_del / process_dup / process_inv, `orca_utils.StructuralChange2`)
edit a chromosome's coordinate map and materialise, per allele, a 512 MB float one-hot window on the host.  Here
an allele is a LIST OF PIECES `(src_start, length, strand)` of the packed (1 byte/base) chromosome that already
lives in HBM; a 32 Mb window of the edited chromosome is assembled on the device by gathering those pieces
(reverse-complemented where the strand is '-') and goes straight into the Encoder's packed-input path.

Scope: one alternative-allele window per variant, centred on the variant like the reference's `coord_clip` centring
of the reference window; the reference's multiple anchorings / plotting are out of scope.
"""
from collections import namedtuple

import numpy as np
import torch

SV = namedtuple("SV", "kind start end")   # kind in {"del", "dup", "inv"}; half-open [start, end) on the chromosome
WINDOW = 32_000_000


def coord_round(x, gridsize=4000):
    """Round down to the prediction grid (orca_utils.py:1044-1060)."""
    return x - x % gridsize


def coord_clip(pos, chrlen, binsize=128000, window_radius=16000000):
    """Centre for a full window around ``pos`` that stays inside the chromosome while keeping pos's phase on the
    ``binsize`` grid where possible (orca_utils.py:1009-1041)."""
    if pos < binsize or pos > chrlen - binsize:
        return int(np.clip(pos, window_radius, chrlen - window_radius))
    edge = (chrlen - window_radius) % binsize - pos % binsize
    endclip = chrlen - window_radius - edge if edge > 0 else chrlen - window_radius - binsize - edge
    return int(np.clip(pos, window_radius + pos % binsize, endclip))


def synth_svs(n, chrlen, seed=1000):
    """n synthetic variants: type uniform over del/dup/inv, size log-uniform 10 kb - 5 Mb, 4 kb-aligned."""
    out = []
    for k in range(n):
        rs = np.random.RandomState(seed + k)
        kind = ("del", "dup", "inv")[rs.randint(3)]
        size = int(coord_round(int(np.exp(rs.uniform(np.log(10_000), np.log(5_000_000)))))) or 4000
        start = int(coord_round(int(rs.randint(6_000_000, chrlen - 6_000_000 - size))))
        out.append(SV(kind, start, start + size))
    return out


def allele_pieces(sv, chrlen):
    """The edited chromosome as pieces of the original."""
    s, e = sv.start, sv.end
    if sv.kind == "del":
        return [(0, s, "+"), (e, chrlen - e, "+")]
    if sv.kind == "dup":      # tandem duplication
        return [(0, e, "+"), (s, e - s, "+"), (e, chrlen - e, "+")]
    if sv.kind == "inv":
        return [(0, s, "+"), (s, e - s, "-"), (e, chrlen - e, "+")]
    raise ValueError(sv.kind)


def window_pieces(pieces, w0, length):
    """Pieces covering [w0, w0+length) of the concatenation of ``pieces``."""
    out, pos = [], 0
    for src, ln, strand in pieces:
        lo, hi = max(w0, pos), min(w0 + length, pos + ln)
        if lo < hi:
            off, cnt = lo - pos, hi - lo
            # a '-' piece is read backwards: its local offset counts from the END of the source interval
            out.append((src + off, cnt, "+") if strand == "+" else (src + ln - off - cnt, cnt, "-"))
        pos += ln
    if sum(p[1] for p in out) != length:
        raise ValueError("window exceeds the allele")
    return out


def sv_windows(sv, chrlen, length=WINDOW):
    """(ref_pieces, ref_wpos, ref_mpos, alt_pieces, alt_wpos, alt_mpos): both windows centred (coord_clip) on the
    variant's midpoint, in reference resp. alternative-allele coordinates."""
    r = length // 2
    mid = (sv.start + sv.end) // 2
    ref_c = coord_clip(mid, chrlen, window_radius=r)
    pieces = allele_pieces(sv, chrlen)
    alt_len = sum(p[1] for p in pieces)
    alt_mid = {"del": sv.start, "dup": sv.end, "inv": mid}[sv.kind]   # breakpoint / junction / centre in alt coordinates
    alt_c = coord_clip(alt_mid, alt_len, window_radius=r)
    return (window_pieces([(0, chrlen, "+")], ref_c - r, length), ref_c, mid,
            window_pieces(pieces, alt_c - r, length), alt_c, alt_mid)


def assemble_codes(genome_codes, pieces):
    """Gather pieces of a packed chromosome ([L] uint8: 0..3 = ACGT, 4 = N; torch tensor on any device or numpy)
    into one window; '-' pieces are reverse-complemented (index flip, code 3-c)."""
    if isinstance(genome_codes, np.ndarray):
        parts = []
        for src, ln, strand in pieces:
            seg = genome_codes[src: src + ln]
            if strand == "-":
                seg = seg[::-1]
                seg = np.where(seg < 4, 3 - seg, seg).astype(np.uint8)
            parts.append(seg)
        return np.concatenate(parts)
    parts = []
    for src, ln, strand in pieces:
        seg = genome_codes[src: src + ln]
        if strand == "-":
            seg = torch.flip(seg, [0])
            seg = torch.where(seg < 4, 3 - seg, seg)
        parts.append(seg)
    return torch.cat(parts)


def sv_screen(models, genome_codes, svs, chrlen, mchr="chrS", rank=0, world=1):
    """Predict reference and alternative allele (6 maps each, per model) for this rank's share of ``svs``
    (independent windows: replicas, no collective).  genome_codes: [chrlen] uint8 tensor on the MI355X.
    Returns {sv_index: {"sv": SV, "ref": output_dict, "alt": output_dict}} with genomepredict's output dicts."""
    from . import dist, orca_predict
    res = {}
    for i in dist.shard_indices(len(svs), rank, world):
        sv = svs[i]
        rp, rw, rm, ap, aw, am = sv_windows(sv, chrlen)
        ref = orca_predict.genomepredict(assemble_codes(genome_codes, rp)[None], mchr, rm, rw, models=models)
        alt = orca_predict.genomepredict(assemble_codes(genome_codes, ap)[None], mchr, am, aw, models=models)
        res[i] = {"sv": sv, "ref": ref, "alt": alt}
    return res
