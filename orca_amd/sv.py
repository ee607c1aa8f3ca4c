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

from . import engine

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


def synth_svs(n, chrlen, seed=1000, align=1):
    """n synthetic variants as SURVEY.md 8(d) config 5 defines them: type uniform over del/dup/inv, size log-uniform 10 kb - 5 Mb, start
    uniform, NO alignment (``align=1``, the default since round 6: real breakpoints sit at arbitrary bases and the reference's windows keep
    the variant's phase, orca_predict.py:1613, orca_utils.py:1009-1041).  ``align=4000``: sizes and starts rounded down to the 4 kb
    prediction grid - the set rounds 3-5 quoted, on which every window of the screen shares ONE phase with the chromosome encodings (the
    incremental screen's best case; bench.py reports it as a labelled second figure)."""
    out = []
    for k in range(n):
        rs = np.random.RandomState(seed + k)
        kind = ("del", "dup", "inv")[rs.randint(3)]
        size = int(coord_round(int(np.exp(rs.uniform(np.log(10_000), np.log(5_000_000)))), align)) or align
        start = int(coord_round(int(rs.randint(6_000_000, chrlen - 6_000_000 - size)), align))
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


# ---------------------------------------------------------------------------------------------------------------------------------
# Incremental encoding (round 4).  The reference pushes every allele window through the whole Encoder (orca_predict.py:1510-1817 build
# ref / alt windows, :231 encodes each): a screen of 1 024 variants on one chromosome encodes the same bases ~2 000 times - 74 % of
# the screen's time.  The Encoder is a translation-covariant stack on a 4 kb grid: Encoder bin j of a sequence depends on the bases
# within RF_BP of the bin only (orca_modules.py:811-927: receptive reach 104 016 bp; the 112 kb halo of :929-980 exists because of it).
# So the chromosome's two strands are encoded ONCE per 4 kb phase (`ChromEncodings`), and a window of an allele - a list of pieces
# of the chromosome - takes every bin whose receptive field lies inside ONE piece from there ('+' pieces from the forward strand's
# encoding, '-' pieces from the reverse complement's); only the bins next to a window end (the reference zero-pads every layer there,
# orca_modules.py:955-977) or next to a junction between pieces are run through the Encoder (its bin-range form, on the assembled window).
# ---------------------------------------------------------------------------------------------------------------------------------
RF_BP = 104_016          # receptive reach of the Encoder beyond a bin, in bases
RF_BINS = 27             # ... in 4 kb bins (27 * 4000 = 108 000 >= RF_BP)
BIN = 4000


class ChromEncodings:
    """Encoder outputs of a packed chromosome that are held in HBM for reuse, per strand and 4 kb phase:
    * ENTRIES - the whole chromosome (128 x chrlen/4000 floats: 5 MB per entry for 40 Mb), built on demand.  Strand '+', phase p: bins of
      chrom[p:], bin i = bases [p + 4000 i, ..); strand '-', phase p: bins of revcomp(chrom)[p:] in reverse-complement coordinates
      (q = chrlen - 1 - forward position);
    * SEGMENTS (round 5) - the Encoder output of a WINDOW that some call has encoded anyway (`add_segment`: a reference-allele view of a
      structural-variant driver, both strands): the bins of strand coordinates [c0, c0 + 4000 n).  A later window - the alternative allele
      of the same call, or a later call at the same phase - takes its bins from there.
    A bin is served from a source only if the source's phase matches and the bin lies RF_BINS bins inside it (the bins next to a source's
    ends saw its zero padding).  ``codes``: the chromosome's [chrlen] uint8 tensor, or a callable returning it (a 2-bit genome unpacks
    only when an entry is really built); ``chrlen`` is then required."""

    def __init__(self, net0, codes, max_entries=8, chrlen=None, max_segments=24):
        if callable(codes):
            if chrlen is None:
                raise ValueError("ChromEncodings: chrlen is required with a codes callable")
            self._codes, self._codes_fn, self.C = None, codes, int(chrlen)
        else:
            if not (isinstance(codes, torch.Tensor) and codes.dtype == torch.uint8 and codes.dim() == 1):
                raise ValueError("codes: a [chrlen] uint8 tensor (on the MI355X for an orca_amd Encoder)")
            self._codes, self._codes_fn, self.C = codes, None, int(codes.shape[0])
        self.net0, self.max_entries, self.max_segments = net0, max_entries, max_segments
        self.entries = {}
        self.segments = []        # [strand, c0, tensor [128, n]], least recently used first
        self.miss_bins = 1000     # bins a run must miss to count as a request for a chromosome encoding (an eighth of a 32 Mb window)
        self.requests = {}        # (strand, phase) -> window-sized runs that no source could serve so far (`build="auto"`)
        self.builds = 0
        self.s3_misses = 0        # window strands of this chromosome that nobody could serve since the last attempt to build `stage3` (GenomeEncodings)
        self.s3_spans = []        # ... and what each of them took from the chromosome, (lo, hi) in forward coordinates: the region to cache
        self.stage3 = None        # a Stage3Cache of this chromosome (sv_screen builds one for variants off the 4 kb grid): `encode_windows` sends
                                  # strands whose bins nobody holds through it instead of through the whole Encoder

    @property
    def codes(self):
        if self._codes is None:
            self._codes = self._codes_fn()
        return self._codes

    def auto_threshold(self):
        """Unserved window-sized runs of one (strand, phase) after which the whole chromosome is encoded at that phase: an entry costs
        chrlen / 32 Mb windows' worth of Encoder time and saves about one window's worth per run from then on."""
        return max(2, -(-self.C // WINDOW))

    def get(self, strand, phase, build=True):
        key = (strand, int(phase))
        e = self.entries.get(key)
        if e is None and build and len(self.entries) < self.max_entries:
            nb = (self.C - key[1]) // BIN
            if nb <= 2 * RF_BINS:
                return None
            if strand == "+":
                e = self.net0.forward_codes(self.codes[None, key[1]: key[1] + nb * BIN], reverse=False)[0]
            else:
                e = self.net0.forward_codes(self.codes[None, self.C - key[1] - nb * BIN: self.C - key[1]], reverse=True)[0]
            self.entries[key] = e
            self.builds += 1
            # built inside a driver call's deferred range check (`build="auto"` under sv_drivers._run_views): the entry outlives the call, so it
            # must not survive a pass whose check fires - the range-safe retry then builds its own, under force_safe
            engine.tentative(lambda key=key, e=e: self._drop_entry(key, e))
        return e

    def _drop_entry(self, key, e):
        if self.entries.get(key) is e:
            del self.entries[key]
            self.builds -= 1

    def add_segment(self, strand, c0, enc):
        """Keep ``enc`` [128, n] = the bins of strand coordinates [c0, c0 + 4000 n) (a copy is NOT made: hand over a tensor of your own)."""
        if enc.shape[1] > 2 * RF_BINS:
            self.segments.append([strand, int(c0), enc])
            del self.segments[: max(0, len(self.segments) - self.max_segments)]

    def cover(self, strand, coord, nbins, build=True, extra=()):
        """What the held sources can serve of the ``nbins`` bins starting at strand coordinate ``coord``: [(a, b, view [128, b - a])] with
        0 <= a < b <= nbins, ascending and disjoint.  ``build``: True = encode the chromosome at this phase if it is not held, False = never,
        "auto" = once `auto_threshold()` window-sized runs of this (strand, phase) went unserved.  ``extra``: more segments (a call's own)."""
        phase = coord % BIN
        key = (strand, phase)
        got = []                                             # (a, b, tensor, first bin of the tensor to take)

        def offer(c0, t):
            i0 = (coord - c0) // BIN
            a, b = max(0, RF_BINS - i0), min(nbins, t.shape[1] - RF_BINS - i0)
            if b > a:
                got.append((a, b, t, i0 + a))

        def entry(bld):
            e = self.get(strand, phase, bld)
            if e is not None:
                offer(phase, e)
        entry(build is True)
        used = []
        for k, seg in enumerate(list(self.segments) + list(extra)):
            if seg[0] == strand and (coord - seg[1]) % BIN == 0:
                n0 = len(got)
                offer(seg[1], seg[2])
                if len(got) > n0 and k < len(self.segments):
                    used.append(k)
        if used:                                             # used: most recent (by identity: the lists hold tensors)
            self.segments = [g for k, g in enumerate(self.segments) if k not in used] + [self.segments[k] for k in used]
        got.sort(key=lambda g: (g[0], -g[1]))
        out, pos = [], 0
        for a, b, t, j in got:
            if b <= pos:
                continue
            if a < pos:
                j, a = j + (pos - a), pos
            out.append((a, b, t[:, j: j + (b - a)]))
            pos = b
        if build == "auto" and key not in self.entries:
            missing = nbins - sum(b - a for a, b, _ in out)
            if missing >= self.miss_bins:                       # a window-sized miss
                self.requests[key] = self.requests.pop(key, 0) + 1
                while len(self.requests) > 64:                  # one key per off-grid phase ever seen: keep the most recent ones
                    del self.requests[next(iter(self.requests))]
                if self.requests[key] >= self.auto_threshold() and len(self.entries) < self.max_entries:
                    return self.cover(strand, coord, nbins, True, extra)
        return out

    def lookup(self, strand, coord, nbins, build=True):
        """[128, nbins] view of the bins starting at strand coordinate ``coord`` (forward position, or reverse-complement coordinate for
        '-') when ONE source holds all of them, else None (no source of that phase is held / may be built, or the range touches the
        RF_BINS bins at an end of a source: those saw its zero padding)."""
        for a, b, v in self.cover(strand, coord, nbins, build):
            if a == 0 and b == nbins:
                return v
        return None


class GenomeEncodings:
    """`ChromEncodings` per chromosome of one genome for one Encoder: what `encode_windows` looks pieces `(chrom, start, length, strand)`
    up in.  ``chrom_codes(chrom)`` -> the chromosome's [chrlen] uint8 codes on the device; ``chrlens``: {chrom: length}."""

    def __init__(self, net0, chrom_codes, chrlens, max_entries=4, max_chroms=8):
        self.net0, self.chrom_codes, self.chrlens, self.max_entries, self.max_chroms = net0, chrom_codes, dict(chrlens), max_entries, max_chroms
        self.chroms = {}           # least recently used first; at most max_chroms chromosomes keep encodings (<= 24 segments of 4 MB + max_entries
        self._builds_gone = 0      # whole-chromosome encodings each: a bound on what a long session over a whole genome holds in HBM)
        self.s3_budget = 120e9     # bytes of stage-3 caches (1 KB per base of a chromosome) this store may hold: a chromosome that does not fit
        self.s3_after = 16         # beside the others' is served without one; built after this many window strands nobody could serve

    def of(self, chrom):
        if chrom not in self.chrlens:
            return None                                        # an inserted sequence, padding: nothing to reuse
        c = self.chroms.pop(chrom, None)
        if c is None:
            c = ChromEncodings(self.net0, lambda chrom=chrom: self.chrom_codes(chrom), self.max_entries, chrlen=self.chrlens[chrom])
        self.chroms[chrom] = c
        while len(self.chroms) > self.max_chroms:
            self._builds_gone += self.chroms.pop(next(iter(self.chroms))).builds
        return c

    @property
    def builds(self):
        return self._builds_gone + sum(c.builds for c in self.chroms.values())

    def stage3_caches(self, pcs, build, make=None):
        """{chrom: stage cache} for a window strand given as strand-oriented 4-tuple pieces: the caches of its chromosomes that cover at least
        half of what the window takes from them.  ``build``: the strand just went unserved - where no cache covers it, count it (and remember
        the span of the chromosome it touched), and after `s3_after` of them build a cache of the REGION those strands spanned (+- 4 Mb; the
        whole chromosome when that is what they spanned; grown from the region already held if both fit) - 1 KB of HBM per base, so a real
        chromosome is cached a locus at a time: whatever exceeds `s3_budget` (least recently used chromosomes go first) or the device (beside
        40 GB of workspace) is served without one.  ``make(chrom_encodings, region, nbytes)``: builds the cache (tests substitute it)."""
        spans = {}
        for p in pcs:
            chrom, src, ln, _ = _p4(p)
            if chrom in self.chrlens:
                a, b = spans.get(chrom, (src, src + ln))
                spans[chrom] = (min(a, src), max(b, src + ln))
        out = {}
        for chrom, (lo, hi) in spans.items():
            ce = self.of(chrom)
            s3 = ce.stage3
            if s3 is not None and getattr(s3, "poisoned", False):     # built inside a pass whose fp16-range check fired: gone, and not tried again soon
                ce.stage3 = s3 = None
                ce.s3_misses, ce.s3_spans = -16 * self.s3_after, []
            covered = s3 is not None and s3.region[0] <= lo and hi <= s3.region[1]
            if not covered and build and self.net0.two_part_ok():
                ce.s3_misses += 1
                ce.s3_spans = (ce.s3_spans + [(lo, hi)])[-64:]
                if ce.s3_misses >= self.s3_after:
                    region = self._s3_region(ce, ce.s3_spans, s3.region if s3 is not None else None)
                    if region is None and s3 is not None:
                        region = self._s3_region(ce, ce.s3_spans)                     # the old region and the new one do not fit together
                    if region is not None:
                        need = Stage3Cache.bytes_needed(region[1] - region[0])
                        ce.stage3 = s3 = None                                         # (its planes are free for the new one)
                        held = [(c, k) for c, k in self.chroms.items() if k.stage3 is not None]
                        while held and sum(Stage3Cache.bytes_needed(k.stage3.region[1] - k.stage3.region[0]) for _, k in held) + need > self.s3_budget:
                            held.pop(0)[1].stage3 = None
                        ce.stage3 = s3 = (make or self._s3_make)(ce, region, need)
                    ce.s3_misses, ce.s3_spans = 0, []
            if s3 is not None and min(hi, s3.region[1]) - max(lo, s3.region[0]) >= (hi - lo) // 2:
                out[chrom] = s3
        return out

    def _s3_region(self, ce, spans, keep=None):
        """(r0, r1), multiples of 80: the spans' hull + 4 Mb either side inside the chromosome (and the region ``keep``); None if that is
        more than the budget holds."""
        r0 = max(0, min(a for a, _ in spans) - 4_000_000) // 80 * 80
        r1 = min(ce.C, -(-(max(b for _, b in spans) + 4_000_000) // 80) * 80)
        if keep is not None:
            r0, r1 = min(r0, keep[0]), max(r1, keep[1])
        return (r0, r1) if Stage3Cache.bytes_needed(r1 - r0) <= self.s3_budget else None

    def _s3_make(self, ce, region, need):
        dev = ce.codes.device
        if dev.type != "cuda" or need + 40e9 >= _hbm_available(dev):
            return None
        s3 = Stage4Cache(self.net0, ce.codes, region)
        return s3 if s3.build_all() else None


def _p4(piece):
    """Pieces are `(src_start, length, strand)` of ONE chromosome (the screen) or `(chrom, src_start, length, strand)` (the drivers)."""
    return piece if len(piece) == 4 else (None,) + tuple(piece)


def strand_coord(piece, chrlen):
    """Strand coordinate of the first base a piece contributes: forward position for '+', reverse-complement coordinate for '-'."""
    _, src, ln, strand = _p4(piece)
    return src if strand == "+" else chrlen - src - ln


def revcomp_pieces(pieces):
    """The pieces of the reverse complement of a sequence given by ``pieces``."""
    return [p[:-1] + ("-" if p[-1] == "+" else "+",) for p in reversed([tuple(q) for q in pieces])]


def reuse_plan(pieces, chrlen, nbins):
    """For a sequence of ``nbins`` * 4000 bases given as pieces: [(bin_lo, bin_hi, strand, coord)] - runs of bins whose receptive field
    lies inside one piece (and off the sequence's ends), with the strand coordinate of bin_lo's first base.  With 4-tuple pieces
    `(chrom, start, length, strand)` and ``chrlen`` a {chrom: length} mapping: [(bin_lo, bin_hi, strand, coord, chrom)], pieces of
    anything that is not a chromosome of the mapping (an inserted string, padding) left out."""
    out, o = [], 0
    L = nbins * BIN
    for piece in pieces:
        chrom, src, ln, strand = _p4(piece)
        lo = max(-(-(o + RF_BINS * BIN) // BIN), RF_BINS)                          # ceil
        hi = min((o + ln - RF_BINS * BIN) // BIN, nbins - RF_BINS)                 # exclusive
        if hi > lo:
            if len(piece) == 3:
                out.append((lo, hi, strand, strand_coord(piece, chrlen) + lo * BIN - o))
            elif chrom in chrlen:
                out.append((lo, hi, strand, strand_coord(piece, chrlen[chrom]) + lo * BIN - o, chrom))
        o += ln
    if o != L:
        raise ValueError("pieces do not add up to the window")
    return out


# ---------------------------------------------------------------------------------------------------------------------------------
# Stage-3 cache (round 6): windows at ARBITRARY base positions.  The bins above are reusable only between windows that share a 4 kb phase -
# and the reference's drivers place every window at the variant's own phase (orca_predict.py:1613), so a screen of real breakpoints shares
# nothing at that level.  Two stages of pooling further up the Encoder the grid is 16 bases: stages 1-3 (MaxPool1d(4) twice,
# orca_modules.py:811-852; 99 % of the Encoder's FLOPs) are translation-covariant on it with a reach of 351 bases.  Their output on the
# chromosome, kept once per strand and phase mod 16 as the P16 planes the next conv reads (512 bytes per base and strand: 41 GB for both strands
# of a 40 Mb chromosome - what 288 GB of HBM are for), serves every window of every allele: a strand of a window is a MaxPool1d(5) gather
# from the cache, the front of the Encoder on a few kb at its ends and junctions, and stages 4-7 (`orca_encoder_back`): ~3 instead of 24.6 ms.
# ---------------------------------------------------------------------------------------------------------------------------------
S3_GRID = 16             # bases per stage-3 position
S3_POOL = 5              # MaxPool1d(5) in front of stage 4: a stage-4 position is 80 bases
S3_MARGIN_BP = 512       # >= reach of a stage-3 position beyond its 16 bases (16 + 3 + 64 + 12 + 256 = 351), in bases, a multiple of 16
S3_PAD_BP = 1600         # bases a snippet extends beyond the pooled positions it is run for (>= S3_MARGIN_BP, a multiple of 80)
S3_MIN_SNIPPET_BP = 4000
# ... and one level further (Stage4Cache): stage 4's output on the 80-base grid of ITS input, reach 351 + 16 * 80 = 1 631 bases
S4_GRID = 80
S4_MARGIN_BP = 1760      # >= 1 631, a multiple of 80
S4_PAD_BP = 2400         # >= S4_MARGIN_BP, a multiple of 400 (a snippet's pooled rows must line up with the window's)
S4_MIN_SNIPPET_BP = 8000
S4_CONCAT_MAX_BP = 400_000   # snippets of a strand longer than this in all (pieces the cache does not hold) run one by one


def _hbm_available(dev):
    """Bytes a new tensor can get on ``dev``: what the driver reports free plus what torch's caching allocator holds without using it."""
    return torch.cuda.mem_get_info(dev)[0] + torch.cuda.memory_reserved(dev) - torch.cuda.memory_allocated(dev)


def s3_plan(pcs, chrlens, L, margin=S3_MARGIN_BP, regions=None, grid=S3_GRID, pad=S3_PAD_BP, min_snippet=S3_MIN_SNIPPET_BP):
    """Stage-4 input of a window given as strand-oriented pieces (3-tuples of ONE chromosome of length ``chrlens``, or 4-tuples with
    ``chrlens`` a {chrom: length} mapping): (takes, snippets).  takes = [(m_lo, m_hi, chrom, strand, phase, c)]: pooled positions
    [m_lo, m_hi) are the MaxPool1d(5) of cache entry (chrom, strand, phase) from the stage-3 position that starts at strand coordinate c;
    snippets = [(ga, gb, base0, nbases, skip)]: pooled positions [ga, gb) come from the Encoder's front run on window bases [base0, base0 +
    nbases), whose pooled position ``skip`` is position ga.  Every pooled position is in exactly one of them.  ``regions``: the part of a
    chromosome the cache holds, (r0, r1) in forward coordinates (or {chrom: (r0, r1)}; default the whole chromosome) - bases outside it
    go through the front like anything else the cache does not know.  ``grid`` = bases per cached position (16: stage 3, 80: stage 4 -
    "stage-3 position" then reads stage-4 position and the pooled positions are stage 5's input), with its ``margin`` / ``pad``."""
    cell = grid * S3_POOL
    if L % cell:
        raise ValueError(f"window length must be a multiple of {cell}")
    n4 = L // cell
    takes, o = [], 0
    for piece in pcs:
        chrom, src, ln, strand = _p4(piece)
        C = chrlens.get(chrom) if isinstance(chrlens, dict) else chrlens
        if C is not None:
            c0 = strand_coord(piece, C)
            r0, r1 = (regions.get(chrom, (0, C)) if isinstance(regions, dict) else regions) if regions else (0, C)
            lo_s, hi_s = (r0, r1) if strand == "+" else (C - r1, C - r0)      # what the cache holds, in this strand's coordinates
            i_lo = -(-(o + margin) // grid)                         # first / one-past-last stage-3 position whose reach lies inside the piece
            i_hi = (o + ln - margin) // grid
            i_lo = max(i_lo, -(-(lo_s + margin - c0 + o) // grid)) # ... and inside the cached range's own interior (the cache saw ITS ends' padding)
            i_hi = min(i_hi, (hi_s - margin - c0 + o) // grid)
            m_lo, m_hi = max(-(-i_lo // S3_POOL), 0), min(i_hi // S3_POOL, n4)
            if m_hi > m_lo:
                c = c0 + cell * m_lo - o                               # strand coordinate of pooled position m_lo's first base
                takes.append((m_lo, m_hi, chrom, strand, c % grid, c))
        o += ln
    if o != L:
        raise ValueError("pieces do not add up to the window")
    snippets, pos = [], 0
    for ga, gb in [(t[1], u[0]) for t, u in zip([(0, 0)] + takes, takes + [(n4, n4)])]:
        if gb <= ga:
            continue
        b0 = 0 if ga == 0 else max(0, ga * cell - pad)
        b1 = L if gb == n4 else min(L, gb * cell + pad)
        while b1 - b0 < min_snippet and (b0 > 0 or b1 < L):      # very short runs: a snippet of at least one bin
            if b1 < L:
                b1 = min(L, b1 + cell)
            else:
                b0 = max(0, b0 - cell)
        snippets.append((ga, gb, b0, b1 - b0, (ga * cell - b0) // cell))
    return takes, snippets


class Stage3Cache:
    """Stage-3 output of ONE chromosome's two strands (or of a ``region`` (r0, r1) of it, forward coordinates) at the 16 phases of the
    16-base grid, as P16 planes in HBM (`Encoder.stage3_planes`), and the route from it to the bins of a window (`encode`).  Entry (strand,
    phase) starts at the first strand coordinate e0 >= the region's start with e0 % 16 == phase: position j = the 16 bases from e0 + 16 j.
    ``codes``: the chromosome's [chrlen] uint8 codes on the device."""
    level, grid = 3, S3_GRID

    def __init__(self, net0, codes, region=None):
        if not (isinstance(codes, torch.Tensor) and codes.dtype == torch.uint8 and codes.dim() == 1 and codes.is_cuda):
            raise ValueError("codes: a [chrlen] uint8 tensor on the MI355X")
        self.net0, self.codes, self.C = net0, codes, int(codes.shape[0])
        r0, r1 = (0, self.C) if region is None else (max(0, int(region[0])), min(self.C, int(region[1])))
        if r1 - r0 < 2 * S3_MARGIN_BP + 80:
            raise ValueError("stage-3 cache: empty region")
        self.region = (r0, r1)
        self.entries = {}
        self.builds = 0
        self.poisoned = False

    @staticmethod
    def bytes_needed(nbases):
        return 2 * 16 * 32 * 16 * (nbases // S3_GRID + 1100)          # strands x phases x planes x bytes per unit x units

    def _origin(self, strand, phase):
        lo = self.region[0] if strand == "+" else self.C - self.region[1]
        return lo + (phase - lo) % self.grid

    def _planes3(self, strand, p16):
        """Stage-3 planes of the region at phase ``p16`` of the 16-base grid: (planes or None, origin, positions)."""
        lo, hi = (self.region[0], self.region[1]) if strand == "+" else (self.C - self.region[1], self.C - self.region[0])
        e0 = lo + (p16 - lo) % S3_GRID
        n = (hi - e0) // 80 * 80
        seg = self.codes[e0: e0 + n] if strand == "+" else self.codes[self.C - e0 - n: self.C - e0]
        return self.net0.stage3_planes(seg.contiguous(), reverse=(strand == "-")), e0, n // S3_GRID

    def get(self, strand, phase):
        key = (strand, int(phase))
        e = self.entries.get(key)
        if e is None:
            e = self._planes3(*key)[0]
            if e is None:
                return None
            self.entries[key] = e
            self.builds += 1
            engine.tentative(lambda key=key, e=e: self._drop(key, e))     # built inside a deferred range check: must not survive it firing
        return e

    def _drop(self, key, e):
        """(engine.tentative) the deferred range check of the pass that built this entry fired: the entry goes, and the cache is marked - its
        owner stops using it (the weights overflow the fp16 range on this genome: every rebuild would end the same way)."""
        if self.entries.get(key) is e:
            del self.entries[key]
            self.builds -= 1
        self.poisoned = True

    def build_all(self):
        """All entries (32; 160 one level up); False (nothing kept) if the fp16-range guard fired on one of them."""
        for strand in "+-":
            for phase in range(self.grid):
                if self.get(strand, phase) is None:
                    self.entries.clear()
                    return False
        return True

    def encode(self, pcs, codes_w, reverse, out_row):
        """Bins of one strand of a window: ``pcs`` = the strand-oriented pieces (of this chromosome) of the window whose FORWARD codes are
        ``codes_w`` [L] (``reverse``: the strand is the reverse complement of those codes), ``out_row`` [128, L / 4000].  Returns the number
        of bases that went through the Encoder's front again (window ends and junctions)."""
        return s3_encode(self.net0, {None: self}, pcs, codes_w, reverse, out_row)


class Stage4Cache(Stage3Cache):
    """The same one level further up the Encoder: stage 4's output (ReLU + residual, before the MaxPool1d(5) in front of stage 5) as fp32
    rows [n, 128] per strand and phase mod 80 - the same 512 bytes per base and strand, 160 entries - so that a window strand runs only
    stages 5-7 (0.8 instead of 2.0 ms) behind a gather of rows and the front + stage 4 on its ends and junctions (reach 1 631 bases).
    Entries are DERIVED: the stage-3 planes of a phase mod 16 exist while its five phases mod 80 are pooled and run through stage 4."""
    level, grid = 4, S4_GRID

    def get(self, strand, phase):
        key = (strand, int(phase))
        if key not in self.entries:
            self._build_group(strand, key[1] % S3_GRID)
        return self.entries.get(key)

    def _build_group(self, strand, p16):
        planes, e0, n3 = self._planes3(strand, p16)
        if planes is None:
            return
        hi = self.region[1] if strand == "+" else self.C - self.region[0]
        ctx = engine.get_context(self.codes.device)
        for k in range(S4_GRID // S3_GRID):
            key = (strand, (p16 + S3_GRID * k) % S4_GRID)
            e4 = self._origin(*key)
            j0 = (e4 - e0) // S3_GRID
            n4 = min((hi - e4) // S4_GRID, (n3 - j0) // S3_POOL)
            if n4 <= 0:
                continue
            s4 = torch.empty((32, engine.p16_plane_units(n4), 4), dtype=torch.float32, device=self.codes.device)
            engine.p16_pool5_into(ctx, planes, j0, s4, 0, n4)
            e = self.net0.stage4_rows(s4, n4)
            self.entries[key] = e
            self.builds += 1
            engine.tentative(lambda key=key, e=e: self._drop(key, e))
        if not engine._guard["defer"] and ctx.take_overflow():
            import warnings
            warnings.warn("orca_amd: an activation left the fp16 range while building a stage-4 cache entry; windows will be encoded whole")
            for k in range(S4_GRID // S3_GRID):
                self.entries.pop((strand, (p16 + S3_GRID * k) % S4_GRID), None)

    def build_all(self, times=None):
        """All 160 entries; False (nothing kept) if the fp16-range guard fired.  ``times``: a list that receives the seconds each of the 32
        groups took (a stream sync per group: diagnostics only)."""
        import time
        for strand in "+-":
            for p16 in range(S3_GRID):
                if times is not None:
                    torch.cuda.synchronize(self.codes.device)
                    t0 = time.perf_counter()
                self._build_group(strand, p16)
                if times is not None:
                    torch.cuda.synchronize(self.codes.device)
                    times.append(time.perf_counter() - t0)
                if (strand, p16) not in self.entries:
                    self.entries.clear()
                    return False
        return True


def s3_encode(net0, caches, pcs, codes_w, reverse, out_row):
    """`Stage3Cache.encode` for pieces of several chromosomes: ``caches`` = {chrom: Stage3Cache} (key None: 3-tuple pieces of the one
    chromosome); pieces of anything else (an inserted string, padding, a chromosome without a cache) go through the Encoder's front.
    `Stage4Cache`s (all of ``caches`` one kind): the same with rows, the front + stage 4, and stages 5-7."""
    if next(iter(caches.values())).level == 4:
        return _s4_encode(net0, caches, pcs, codes_w, reverse, out_row)
    L = int(codes_w.numel())
    n4 = L // (S3_GRID * S3_POOL)
    if None in caches:
        takes, snippets = s3_plan(pcs, caches[None].C, L, regions=caches[None].region)
    else:
        takes, snippets = s3_plan(pcs, {c: k.C for c, k in caches.items()}, L, regions={c: k.region for c, k in caches.items()})
    s4 = torch.empty((32, engine.p16_plane_units(n4), 4), dtype=torch.float32, device=codes_w.device)
    ctx = engine.get_context(codes_w.device)
    for m_lo, m_hi, chrom, strand, phase, c in takes:
        src = caches[chrom].get(strand, phase)
        if src is None:
            raise RuntimeError("stage-3 cache entry unavailable (fp16 range)")
        engine.p16_pool5_into(ctx, src, (c - caches[chrom]._origin(strand, phase)) // S3_GRID, s4, m_lo, m_hi - m_lo)
    for ga, gb, b0, nb, skip in snippets:
        net0.front_snippet(codes_w, reverse, b0, nb, skip, gb - ga, s4, ga)
    net0.back(s4, n4, out_row)
    return sum(sn[3] for sn in snippets)


def _s4_encode(net0, caches, pcs, codes_w, reverse, out_row):
    L = int(codes_w.numel())
    n5 = L // (S4_GRID * S3_POOL)
    kw = dict(margin=S4_MARGIN_BP, grid=S4_GRID, pad=S4_PAD_BP, min_snippet=S4_MIN_SNIPPET_BP)
    if None in caches:
        takes, snippets = s3_plan(pcs, caches[None].C, L, regions=caches[None].region, **kw)
    else:
        takes, snippets = s3_plan(pcs, {c: k.C for c, k in caches.items()}, L, regions={c: k.region for c, k in caches.items()}, **kw)
    s5 = torch.empty((n5, 128), dtype=torch.float32, device=codes_w.device)
    ctx = engine.get_context(codes_w.device)
    for m_lo, m_hi, chrom, strand, phase, c in takes:
        src = caches[chrom].get(strand, phase)
        if src is None:
            raise RuntimeError("stage-4 cache entry unavailable (fp16 range)")
        engine.rows_pool5_into(ctx, src, (c - caches[chrom]._origin(strand, phase)) // S4_GRID, s5, m_lo, m_hi - m_lo)
    if len(snippets) > 1 and snippets[0][0] == 0 and snippets[-1][1] == n5 and sum(sn[3] for sn in snippets) <= S4_CONCAT_MAX_BP:
        # ONE front run for the strand's snippets, concatenated in strand order: the window's two ends are first and last (the run's ends ARE the
        # window's, zero padding included), the seams between snippets lie inside the pads nobody reads.  A reverse-complement strand's sequence is
        # read backwards from the forward codes: its concatenation is the forward slices in reverse order.
        cat = torch.cat([codes_w[b0: b0 + nb] for _, _, b0, nb, _ in snippets] if not reverse else
                        [codes_w[L - b0 - nb: L - b0] for _, _, b0, nb, _ in reversed(snippets)])
        ranges, off = [], 0
        for ga, gb, b0, nb, skip in snippets:
            ranges.append((off // (S4_GRID * S3_POOL) + skip, gb - ga, ga))
            off += nb
        net0.front4_ranges(cat, reverse, ranges, s5)
    else:
        for ga, gb, b0, nb, skip in snippets:
            net0.front4_snippet(codes_w, reverse, b0, nb, skip, gb - ga, s5, ga)
    net0.back5(s5, out_row)
    return sum(sn[3] for sn in snippets)


POOL_MAX_BINS = 500     # longer runs (whole windows: phases that are not held) are not spread over the pool's contexts - they fill the chip on their own


def encode_windows(cache, pieces_list, win_codes, out, merge_gap=2 * RF_BINS, build=True, pool=None, defer_join=False, extra=None, big_on_caller=False):
    """Both strands of W allele windows into ``out`` [2W,128,nbins] (window w: row 2w forward, row 2w + 1 reverse complement): bins whose
    receptive field lies inside one piece are copied from `cache` (ChromEncodings; GenomeEncodings for pieces that name their chromosome),
    the rest - window ends, junctions, pieces of a phase that is not held - go through the Encoder's bin-range form on ``win_codes`` [W,L]
    (the assembled windows).  Bin ranges that several windows have in common (the window ends, as a rule) are ONE batched call.  Returns
    the number of bins encoded (of 2W * nbins).  ``extra``: {chrom: [segment]} - segments of the caller's own beside the cache's.

    ``pool`` (engine.ContextPool): the bin-range calls are independent of each other and small (220-440 kb of bases: ~50 dependent launches,
    0.65-0.9 ms of GPU time in stream order, most of them on a fraction of the chip) - each (window, strand, range) becomes a job of its own,
    the jobs are dealt to the pool's contexts (own stream and workspace), longest first, and their kernels run side by side (9.1 -> 5.4 ms
    per variant); the caller's stream continues behind all of them (``defer_join``: it does not - the caller orders a stream behind them
    later with pool.wait_join: `sv_screen` issues a variant's local encodes under the previous variant's decoders).  Runs longer than
    POOL_MAX_BINS (whole windows) go to the pool's context 0 one after the other, or (``big_on_caller``) stay on the caller's context."""
    nbins = out.shape[2]
    multi = isinstance(cache, GenomeEncodings)
    W = len(pieces_list)
    runs = {}                                                           # (reverse, lo, hi) -> [window]
    # strands whose bins are NOT held (a phase of their own: variants off the 4 kb grid) go through the stage-3 cache when the caller has
    # built one (`cache.stage3`: sv_screen) and the Encoder is in its default arithmetic - never inside a range-safe retry
    s3 = None if multi else getattr(cache, "stage3", None)
    s3_able = win_codes.is_cuda and getattr(cache.net0, "two_part_ok", lambda: False)()
    if not s3_able:
        s3 = None
    s3_jobs = []
    for w, pieces in enumerate(pieces_list):
        for rev, pcs in ((False, pieces), (True, revcomp_pieces(pieces))):
            row = 2 * w + int(rev)
            have = []
            for run in reuse_plan(pcs, cache.chrlens if multi else cache.C, nbins):
                lo, hi, strand, coord = run[:4]
                ce = cache.of(run[4]) if multi else cache
                if ce is None:
                    continue
                for a, b, src in ce.cover(strand, coord, hi - lo, build, (extra or {}).get(run[4] if multi else None, ())):
                    out[row, :, lo + a: lo + b].copy_(src)
                    have.append((lo + a, lo + b))
            # the complement: runs of bins still to encode; runs closer than merge_gap are encoded as one (a call costs ~40 launches)
            todo, pos = [], 0
            for lo, hi in have:
                if lo > pos:
                    todo.append([pos, lo])
                pos = hi
            if pos < nbins:
                todo.append([pos, nbins])
            merged = []
            for r in todo:
                if merged and r[0] - merged[-1][1] < merge_gap:
                    merged[-1][1] = r[1]
                else:
                    merged.append(r)
            if sum(hi - lo for lo, hi in merged) > POOL_MAX_BINS and s3_able:
                if multi:       # the drivers' store: per-chromosome caches, built once enough strands went unserved (`build="auto"`)
                    s3w = cache.stage3_caches(pcs, build == "auto")
                else:
                    s3w = {None: s3} if s3 is not None else {}
                if s3w:
                    s3_jobs.append((w, rev, pcs, s3w))                  # the whole strand (what was copied above is overwritten)
                    continue
            for lo, hi in merged:
                runs.setdefault((rev, lo, hi), []).append(w)
    encoded = sum((hi - lo) * len(ws) for (rev, lo, hi), ws in runs.items())
    if pool is not None and len(pool) > 0:
        jobs = sorted(((hi - lo, rev, lo, hi, w) for (rev, lo, hi), ws in runs.items() for w in ws), reverse=True)
        pool.fork()                                                     # the windows' codes are complete on the caller's stream
        k = 0
        for w, rev, pcs, s3w in s3_jobs:                                # ~3 ms each, the last third latency-bound: one context each, side by side
            encoded += -(-pool.run(k, lambda w=w, rev=rev, pcs=pcs, s3w=s3w: s3_encode(cache.net0, s3w, pcs, win_codes[w], rev, out[2 * w + int(rev)])) // BIN)
            k += 1
        for n_, rev, lo, hi, w in jobs:
            row = 2 * w + int(rev)
            if n_ > POOL_MAX_BINS and big_on_caller:      # a whole window fills the chip on its own: the caller's context and stream
                cache.net0.forward_codes(win_codes[w:w + 1], reverse=rev, bin_lo=lo, bin_hi=hi, out=out[row:row + 1, :, lo:hi])
                continue
            if n_ > POOL_MAX_BINS:      # all on context 0, one after the other (one 25 GB workspace; the range flag stays with the pool's)
                pool.run(0, lambda w=w, rev=rev, lo=lo, hi=hi, row=row: cache.net0.forward_codes(win_codes[w:w + 1], reverse=rev, bin_lo=lo, bin_hi=hi,
                                                                                             out=out[row:row + 1, :, lo:hi]))
                continue
            pool.run(k, lambda w=w, rev=rev, lo=lo, hi=hi, row=row: cache.net0.forward_codes(win_codes[w:w + 1], reverse=rev, bin_lo=lo, bin_hi=hi,
                                                                                         out=out[row:row + 1, :, lo:hi]))
            k += 1
        pool.host_join()
        if not defer_join:
            pool.wait_join()
        return encoded
    for w, rev, pcs, s3w in s3_jobs:
        encoded += -(-s3_encode(cache.net0, s3w, pcs, win_codes[w], rev, out[2 * w + int(rev)]) // BIN)
    for (rev, lo, hi), ws in runs.items():
        if len(ws) == W and W > 1:                                      # every window: the rows of one strand are a strided view of `out`
            cache.net0.forward_codes(win_codes, reverse=rev, bin_lo=lo, bin_hi=hi, out=out[int(rev)::2, :, lo:hi])
        else:
            for w in ws:
                cache.net0.forward_codes(win_codes[w:w + 1], reverse=rev, bin_lo=lo, bin_hi=hi, out=out[2 * w + int(rev):2 * w + int(rev) + 1, :, lo:hi])
    return encoded


def encode_window(cache, pieces, win_codes, out, merge_gap=2 * RF_BINS, build=True):
    """`encode_windows` for one window: ``win_codes`` [L], ``out`` [2,128,nbins]."""
    return encode_windows(cache, [pieces], win_codes[None], out, merge_gap, build)


def needed_phases(svs, chrlen, length=WINDOW):
    """{(strand, phase): number of runs that want it} over the windows of ``svs`` - what `sv_screen` builds before the loop."""
    want = {}
    nbins = length // BIN
    for sv in svs:
        rp, _, _, ap, _, _ = sv_windows(sv, chrlen, length)
        for pcs in (rp, ap):
            for p in (pcs, revcomp_pieces(pcs)):
                for lo, hi, strand, coord in reuse_plan(p, chrlen, nbins):
                    want[(strand, coord % BIN)] = want.get((strand, coord % BIN), 0) + 1
    return want


def _cascade_windows(model, enc0, params):
    """Encoder2 + the six decoder levels for W windows whose Encoder outputs are given: ``enc0`` [2W,128,8000] (window w: forward strand
    in row 2w, reverse complement in row 2w + 1), params[w] = (mpos, wpos).  ONE cascade: every decoder level is a batch of 2W maps.
    Returns (strand-merged maps [W,6,C,250,250] on the device - every target channel of a multi-target model, as `genomepredict`
    keeps them -, starts[k][6])."""
    from . import orca_predict as OP
    W = len(params)
    flags = [bool(k & 1) for k in range(2 * W)]
    zooms = [(lambda lv, st, rev, m=params[k // 2][0], w=params[k // 2][1]: OP.zoom_index_32m(lv, st, m, w, rev)) for k in range(2 * W)]
    bg_cache = {}

    def background(level, k, start):
        if level not in bg_cache:
            bg_cache[level] = OP._cached_log_background(model, level, enc0.is_cuda)
        return bg_cache[level]

    encodings = dict(zip([1, 2, 4, 8, 16, 32], model.net(enc0)))
    preds, starts = OP.run_cascade(model, encodings, [32, 16, 8, 4, 2, 1], lambda lv: lv, 1, flags, background, zooms, add_1m_level=1)
    merged = [torch.stack([torch.stack([engine.strand_merge(p[2 * w, c], p[2 * w + 1, c]) for c in range(p.shape[1])]) for p in preds])
              for w in range(W)]
    return torch.stack(merged), starts


def _window_outputs(model, merged, starts, params, mchr):
    """`genomepredict`'s output dict (one model) per window."""
    levels = [32, 16, 8, 4, 2, 1]
    host = merged.cpu().numpy()
    outs = []
    for w, (mpos, wpos) in enumerate(params):
        sc = [wpos - 16000000 + s * 4000 for s in starts[2 * w]]
        # [250,250] per level for single-target models, [C,250,250] otherwise (orca_predict.py:510-523)
        outs.append({"predictions": [[host[w, j, 0] if host.shape[2] == 1 else host[w, j] for j in range(6)]], "experiments": None, "start_coords": sc,
                     "end_coords": [int(sc[ii] + 32000000 / 2 ** ii) for ii in range(6)], "chr": mchr, "annos": None,
                     "normmats": [[model.normmats[ii] for ii in levels]]})
    return outs


def sv_screen(models, genome_codes, svs, chrlen, mchr="chrS", rank=0, world=1, incremental=True, min_uses=3, stats=None, on_result=None,
              streams=None, group=2, stage3="auto"):
    """Predict reference and alternative allele (6 maps each, per model) for this rank's share of ``svs``
    (independent windows: replicas, no collective).  genome_codes: [chrlen] uint8 tensor on the MI355X.
    Returns {sv_index: {"sv": SV, "ref": output_dict, "alt": output_dict}} with genomepredict's output dicts.

    ``incremental`` (default): the chromosome's strands are encoded once per 4 kb phase that at least ``min_uses`` window runs share
    (a chromosome encoding costs chrlen / 32 Mb windows' worth of Encoder time), windows reuse those bins (`encode_window`), and the
    four strands of a variant (ref / alt x forward / reverse) go through Encoder2 and every decoder level as ONE batch.  Variants whose
    phases are not held fall back to encoding their windows whole - same maps either way (tests/test_gpu_sv_incremental.py).
    LIMIT: reuse needs window runs that SHARE a 4 kb phase - coordinates on one 4 kb grid, as `synth_svs` draws them and as
    screens of binned calls have them.  Variants at arbitrary base positions have a phase of their own each: they take the whole-window
    route (`stats["bins_encoded"] / stats["bins_total"]` says how much of the screen did; `stats["whole_window_runs"]` counts the runs),
    and the quoted ~30 SV/s does not apply to them - the structural-variant drivers (`sv_drivers._run_views`) then still share work between
    the views of one call (1.3-1.6x).
    ``incremental=False`` is the reference's cost structure: two independent `genomepredict` calls per variant.
    ``on_result(i, entry)``: called per variant INSTEAD of collecting the entries (a 1 024-variant screen is 12 288 maps = 3 GB).
    ``streams``: auxiliary contexts the local re-encodes of a variant are dealt to (`encode_windows`; default 4, 0 = all
    on the caller's stream).
    ``group``: variants per pass of Encoder2 + decoders (default 2 = batches of 8 maps per level: a Decoder forward costs 0.97 ms per map at
    B = 8 against 1.00 at B = 4 and 1.12 at B = 2, tools/prof_decoder.py; a map does not depend on the batch it is computed in, so the
    results are the same bit for bit - tests/test_gpu_sv_incremental.py).
    ``stage3``: the stage-3 cache (`Stage3Cache`: windows at arbitrary base positions take stages 1-3 of the Encoder from the chromosome's
    cached planes - 1 KB of HBM per base of the chromosome and model).  "auto" (default): built when the screen has at least 32 window
    strands per 32 Mb of chromosome whose 4 kb phase is not shared, the Encoders run the default arithmetic and the planes fit beside 40 GB of
    workspace; True / False force / forbid it."""
    import time
    from . import dist, orca_predict
    res = {}
    mine = list(dist.shard_indices(len(svs), rank, world))
    if not incremental:
        for i in mine:
            sv = svs[i]
            rp, rw, rm, ap, aw, am = sv_windows(sv, chrlen)
            ref = orca_predict.genomepredict(assemble_codes(genome_codes, rp)[None], mchr, rm, rw, models=models)
            alt = orca_predict.genomepredict(assemble_codes(genome_codes, ap)[None], mchr, am, aw, models=models)
            entry = {"sv": sv, "ref": ref, "alt": alt}
            if on_result is not None:
                on_result(i, entry)
            else:
                res[i] = entry
        return res
    models = orca_predict._resolve_models(models, "32M", True)
    nbins = WINDOW // BIN
    want = needed_phases([svs[i] for i in mine], chrlen)
    caches = []
    with torch.no_grad():
        caches = [ChromEncodings(model.net0, genome_codes) for model in models]
        # windows whose phase is held by nobody: the stage cache serves them at ANY phase
        whole_runs = int(sum(n for key, n in want.items() if n < min_uses))
        s3_info = None
        if stage3 and genome_codes.is_cuda and all(getattr(m.net0, "two_part_ok", lambda: False)() for m in models):
            # the part of the chromosome this rank's windows touch (a whole 40 Mb chromosome for a screen across it; a locus of a real one)
            spans = [(p[0], p[0] + p[1]) for i in mine for pcs in sv_windows(svs[i], chrlen)[0:4:3] for p in pcs]
            region = (max(0, min(a for a, _ in spans) // 80 * 80), min(chrlen, -(-max(b for _, b in spans) // 80) * 80)) if spans else (0, chrlen)
            need = Stage3Cache.bytes_needed(region[1] - region[0]) * len(models)
            if (stage3 is True or whole_runs >= 32 * -(-(region[1] - region[0]) // WINDOW)) and need + 40e9 < _hbm_available(genome_codes.device):
                t0 = time.perf_counter()
                group_s = []
                for cache in caches:
                    s3c = Stage4Cache(cache.net0, genome_codes, region)
                    cache.stage3 = s3c if s3c.build_all(group_s) else None
                torch.cuda.synchronize(genome_codes.device)
                s3_info = {"entries": sum(len(c.stage3.entries) for c in caches if c.stage3 is not None), "GB": round(need / 1e9, 1), "region": list(region),
                           "group_ms_min_median_max": [round(1e3 * x, 1) for x in (min(group_s), sorted(group_s)[len(group_s) // 2], max(group_s))] if group_s else None,
                           "build_s": round(time.perf_counter() - t0, 3)}
        # whole-chromosome bins per 4 kb phase that enough runs share: an entry costs chrlen / 32 Mb windows' worth of Encoder time and serves a
        # strand by a copy; with a stage cache behind it (2 ms per strand at any phase) it pays only from ~16 runs on
        for cache in caches:
            uses = max(min_uses, 16) if cache.stage3 is not None else min_uses
            for key, n in sorted(want.items(), key=lambda kv: -kv[1]):
                if n >= uses:
                    cache.get(*key)
        encoded = 0
        if streams is None:
            streams = 4
        pool = engine.context_pool(genome_codes.device, streams) if (streams > 0 and genome_codes.is_cuda) else None
        dev = genome_codes.device
        group = max(1, int(group))
        chunks = [tuple(mine[c:c + group]) for c in range(0, len(mine), group)]
        units = [(ch, mi) for ch in chunks for mi in range(len(models))]     # one (group of variants, model) per pass of the pipeline below
        slots = [{"enc0": torch.empty((4 * group, 128, nbins), dtype=torch.float32, device=dev), "ev": torch.cuda.Event() if pool else None} for _ in range(2)]
        plans = {}

        def plan(ch):
            """windows of a group of variants: [ref, alt] of each, in order"""
            if ch not in plans:
                plans.clear()
                pieces, params, codes = [], [], []
                for i in ch:
                    rp, rw, rm, ap, aw, am = sv_windows(svs[i], chrlen)
                    pieces += [rp, ap]
                    params += [(rm, rw), (am, aw)]
                    codes += [assemble_codes(genome_codes, rp), assemble_codes(genome_codes, ap)]
                plans[ch] = {"pieces": pieces, "params": params, "codes": torch.stack(codes)}
            return plans[ch]

        def prep(k):
            """Issue unit k's Encoder outputs into slot k % 2: on the pool's side stream and contexts when there is a pool (the caller's
            stream - the previous unit's decoders - is not involved), else right here."""
            ch, mi = units[k]
            slot = slots[k % 2]
            if pool is None:
                pl = plan(ch)
                slot["n"] = encode_windows(caches[mi], pl["pieces"], pl["codes"], slot["enc0"][:4 * len(ch)], build=False)
                return
            with torch.cuda.stream(pool.side):
                pl = plan(ch)
                slot["n"] = encode_windows(caches[mi], pl["pieces"], pl["codes"], slot["enc0"][:4 * len(ch)], build=False, pool=pool, defer_join=True)
                slot["ev"].record(pool.side)

        ctx = engine.get_context(dev) if genome_codes.is_cuda else None
        main = torch.cuda.current_stream(dev) if pool else None
        with engine.defer_overflow_guard():                   # ONE fp16-range check per unit (below), not one per module forward
            if pool is not None:
                pool.side.wait_stream(main)                   # the chromosome encodings above
            if units:
                prep(0)
            enc_over = pool.take_overflow() if pool else False
            entries = {}
            for k, (ch, mi) in enumerate(units):
                model, slot, pl = models[mi], slots[k % 2], plan(ch)
                enc0 = slot["enc0"][:4 * len(ch)]
                if pool is not None:
                    main.wait_event(slot["ev"])
                    pool.wait_join(main)
                merged, starts = _cascade_windows(model, enc0, pl["params"])
                # the NEXT unit's local encodes are issued now: they run on the pool's streams while this unit's decoders (a chain of
                # launches with ramps and tails, matrix pipe busy a quarter of the time) hold the caller's stream
                nxt_over = False
                if pool is not None and k + 1 < len(units):
                    prep(k + 1)
                    nxt_over = pool.take_overflow()            # (waits for those encodes: the host has nothing else to issue)
                if ctx is not None:
                    ctx.sync_stream()                          # (a whole-window fallback in prep() ran this context on the side stream)
                over = (ctx.take_overflow() if ctx is not None else False) or enc_over
                enc_over = nxt_over
                if over:                                      # an activation left the fp16 range: this unit again, range-safe arithmetic
                    import warnings
                    warnings.warn("orca_amd: an activation left the fp16 range; recomputing with range-safe arithmetic (bf16x3 / f32)")
                    with engine.force_safe_precision():
                        slot["n"] = encode_windows(caches[mi], pl["pieces"], pl["codes"], enc0, build=False)
                        merged, starts = _cascade_windows(model, enc0, pl["params"])
                    ctx.take_overflow()
                encoded += slot["n"]
                outs = _window_outputs(model, merged, starts, pl["params"], mchr)
                for v, i in enumerate(ch):
                    r, a = outs[2 * v], outs[2 * v + 1]
                    if mi == 0:
                        entries[i] = {"sv": svs[i], "ref": r, "alt": a}
                    else:
                        for o, n in ((entries[i]["ref"], r), (entries[i]["alt"], a)):
                            o["predictions"] += n["predictions"]
                            o["normmats"] += n["normmats"]
                if pool is None and k + 1 < len(units):
                    prep(k + 1)
                if mi == len(models) - 1:
                    for i in ch:
                        entry = entries.pop(i)
                        if on_result is not None:
                            on_result(i, entry)
                        else:
                            res[i] = entry
    if stats is not None:
        stats.update({"bins_encoded": encoded, "bins_total": len(mine) * len(models) * 4 * nbins,
                      "chromosome_encodings": sum(c.builds for c in caches),
                      "phases_wanted": len(want), "phases_held": sum(len(c.entries) for c in caches) // max(1, len(caches)),
                      "whole_window_runs": whole_runs, "stage3_cache": s3_info})
    return res
