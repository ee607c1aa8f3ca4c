"""Coordinate algebra of the structural-variant drivers (reference `orca_utils.py:733-1060`), without the plotting.

`StructuralChange2` keeps a mutated chromosome as an ordered list of pieces of reference chromosomes and answers
"which reference intervals make up [a, b) of the mutated chromosome".  Behaviour follows the reference class method by
method (same attribute names: `segments` = list of `LGRange(len, GRange(chr, start, end, strand))`, `coord_points`),
including two quirks its callers rely on:
  * `invert` marks every inverted piece '-' - a piece that was already '-' stays '-' (`orca_utils.py:865-870`);
  * `query` does not trim a last piece that ends exactly at the chromosome end (`orca_utils.py:899-903`).
"""
from bisect import bisect_right
from collections import namedtuple
from copy import deepcopy

import numpy as np

from .sv import coord_clip, coord_round  # noqa: F401  (orca_utils.py:1009-1060)

GRange = namedtuple("GRange", ["chr", "start", "end", "strand"])
LGRange = namedtuple("LGRange", ["len", "ref"])


def _piece(chrom, start, end, strand):
    return LGRange(end - start, GRange(chrom, start, end, strand))


class StructuralChange2(object):
    def __init__(self, chr_name, length):
        self.chr_name = chr_name
        self.segments = [_piece(chr_name, 0, length, "+")]
        self.coord_points = [0, length]

    # -- bookkeeping ---------------------------------------------------------------------------------------
    def _coord_sync(self):
        pts = [0]
        for seg in self.segments:
            pts.append(pts[-1] + seg.len)
        self.coord_points = pts

    def _index(self, pos):
        """Piece that contains mutated-chromosome position `pos` (a junction belongs to the piece on its right)."""
        return bisect_right(self.coord_points, pos) - 1

    def _split(self, pos):
        """Make `pos` a junction."""
        k = self._index(pos)
        off = pos - self.coord_points[k]
        if off:
            chrom, s, e, strand = self.segments[k].ref
            if strand == "+":
                left, right = _piece(chrom, s, s + off, "+"), _piece(chrom, s + off, e, "+")
            else:   # a '-' piece is read from its reference END backwards
                left, right = _piece(chrom, e - off, e, "-"), _piece(chrom, s, e - off, "-")
            self.segments[k:k + 1] = [left, right]
        self._coord_sync()

    def _span(self, start, end):
        self._split(start)
        self._split(end)
        return self._index(start), self._index(end)

    # -- edits (coordinates always refer to the CURRENT state of the chromosome) -----------------------------
    def __add__(self, other):
        out = deepcopy(self)
        base = out.coord_points[-1]
        out.segments = out.segments + other.segments
        out.coord_points = out.coord_points + [base + p for p in other.coord_points[1:]]
        return out

    def duplicate(self, start, end):
        i, j = self._span(start, end)
        self.segments[j:j] = [deepcopy(seg) for seg in self.segments[i:j]]   # tandem copy right after the original
        self._coord_sync()

    def insert(self, start, length, strand="+", name=None):
        self._split(start)
        k = self._index(start)
        if not name:
            name = "ins" + str(start) + "_" + str(length)
        self.segments.insert(k, _piece(name, 0, length, strand))
        self._coord_sync()

    def delete(self, start, end):
        i, j = self._span(start, end)
        del self.segments[i:j]
        self._coord_sync()

    def invert(self, start, end):
        i, j = self._span(start, end)
        self.segments[i:j] = [LGRange(seg.len, GRange(seg.ref.chr, seg.ref.start, seg.ref.end, "-"))
                              for seg in reversed(self.segments[i:j])]
        self._coord_sync()

    # -- queries -------------------------------------------------------------------------------------------
    def query(self, start, end):
        """Reference pieces (GRange list) that make up [start, end) of the mutated chromosome."""
        i = self._index(start)
        j = bisect_right(self.coord_points, end - 1)
        if i < 0:
            raise ValueError(f"Warning: query start {start} exceed limit {self.coord_points[0]}!")
        if j == len(self.coord_points) and end > self.coord_points[-1]:
            raise ValueError(f"Warning: query end {end} exceed limit {self.coord_points[-1]}!")
        out = [seg.ref for seg in self.segments[i:j]]
        if out:
            head = start - self.coord_points[i]          # trim the first piece on its leading side
            c, s, e, strand = out[0]
            out[0] = GRange(c, s + head, e, strand) if strand == "+" else GRange(c, s, e - head, strand)
            if j < len(self.coord_points):               # trim the last piece on its trailing side
                tail = self.coord_points[j] - end
                c, s, e, strand = out[-1]
                out[-1] = GRange(c, s, e - tail, strand) if strand == "+" else GRange(c, s + tail, e, strand)
        return out

    def query_ref(self, chr_name, start, end):
        """Where a reference interval ended up: (clipped reference intervals, [start, end, strand] in the mutated
        chromosome), one entry per piece of `chr_name` (the reference's overlap test `start < piece.end or
        end >= piece.start` holds for every piece, `orca_utils.py:937`)."""
        ref_coords, cur_coords = [], []
        for k, (seglen, ref) in enumerate(self.segments):
            if ref.chr != chr_name or not (start < ref.end or end >= ref.start):
                continue
            ref_coords.append([np.clip(start, ref.start, ref.end), np.clip(end, ref.start, ref.end)])
            a, b = np.clip(start - ref.start, 0, seglen), np.clip(end - ref.start, 0, seglen)
            if ref.strand == "+":
                cur_coords.append([self.coord_points[k] + a, self.coord_points[k] + b, "+"])
            else:
                cur_coords.append([self.coord_points[k + 1] - a, self.coord_points[k + 1] - b, "-"])
        return ref_coords, cur_coords

    def __getitem__(self, key):
        if isinstance(key, slice):
            return self.query(key.start, key.stop)


def process_anno(anno_scaled, base=0, window_radius=16000000):
    """Annotations -> fractions of the 2 x window_radius window (`orca_utils.py:968-1006`): regions
    `[start, end, colour]`, sites `[pos, 'single'|'double']`."""
    span = window_radius * 2
    out = []
    for r in anno_scaled:
        if len(r) == 3:
            out.append([(r[0] - base) / span, (r[1] - base) / span, r[2]])
        elif len(r) == 2:
            out.append([(r[0] - base) / span, r[1]])
        else:
            raise ValueError
    return out
