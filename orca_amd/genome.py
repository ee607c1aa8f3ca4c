"""Packed genome store: 1 byte per base (PackedGenome) or 2 bits per base + an N bit-mask (TwoBitGenome, 3 bits per
base: ~1.2 GB for hg38), resident on the host or in HBM.

The reference keeps the genome as a float32 one-hot memmap (`selene_utils2.MemmapGenome`, ~40 GB for hg38,
`selene_utils2.py:38-272`) and materialises a 512 MB float window per `get_encoding_from_coords` call.  Here a
chromosome is ONE uint8 array of base codes - 0..3 = A,C,G,T (the reference's channel order), 4 = N / anything else
(the 0.25 x 4 row) - ~3.1 GB for hg38, which fits HBM many times over.  The selene query API the SV drivers use is
kept (`get_chr_lens`, `get_encoding_from_coords(chrom, start, end, strand="+", pad=False)` returning float32
`[end-start, 4]`, reverse complement = both axes flipped, `pad=True` filling 0.25 beyond the chromosome ends); the
device path adds `get_codes_from_coords`, which returns the same window as codes and feeds the Encoder's packed
input (`orca_encoder_forward_codes`) without ever expanding to floats.

One-hot encoding parity is unpinned by the reference (selene is not vendored): A,C,G,T in that channel
order, case-insensitive, every other symbol -> 0.25 x 4, consistent with `selene_utils2.py:216-222,272`.
"""
import os

import numpy as np
import torch

N_CODE = 4
_LUT = np.full(256, N_CODE, dtype=np.uint8)
for _i, _b in enumerate("ACGT"):
    _LUT[ord(_b)] = _i
    _LUT[ord(_b.lower())] = _i
_ONEHOT = np.concatenate([np.eye(4, dtype=np.float32), np.full((1, 4), 0.25, dtype=np.float32)], axis=0)   # [5,4]


def sequence_to_codes(seq):
    """str / bytes -> uint8 base codes."""
    if isinstance(seq, str):
        seq = seq.encode("ascii")
    return _LUT[np.frombuffer(seq, dtype=np.uint8)]


def codes_to_encoding(codes):
    """uint8 codes [L] -> float32 one-hot [L,4] (N -> 0.25 x 4)."""
    return _ONEHOT[np.minimum(np.asarray(codes), N_CODE)]


def sequence_to_encoding(seq):
    """Counterpart of `selene_sdk.sequences.Genome.sequence_to_encoding` (call site `orca_predict.py:2299`)."""
    return codes_to_encoding(sequence_to_codes(seq))


def revcomp_codes(codes):
    """Reverse complement of a code array (numpy or torch): reversed order, A<->T, C<->G, N stays N."""
    if isinstance(codes, torch.Tensor):
        r = torch.flip(codes, [-1])
        return torch.where(r < 4, 3 - r, r)
    r = np.asarray(codes)[..., ::-1]
    return np.where(r < 4, 3 - r, r).astype(np.uint8)


class PackedGenome:
    """Chromosome name -> uint8 code array.  `to(device)` additionally keeps a ROCm copy for `get_codes_from_coords`."""

    def __init__(self, chroms):
        self._host = {str(k): np.ascontiguousarray(np.asarray(v, dtype=np.uint8)) for k, v in chroms.items()}
        self._dev = {}
        self.device = None

    # ---- construction ------------------------------------------------------------------------------------
    @classmethod
    def from_fasta(cls, path, chroms=None):
        """Multi-FASTA reader.  With ``chroms`` (an iterable of names) only those records are packed; if a samtools
        ``.fai`` index sits next to the file (``path + ".fai"``, the reference's pyfaidx layout: name, length, byte
        offset, bases per line, bytes per line) each wanted record is read with one seek instead of a scan of the whole
        file.  The reference indexes with pyfaidx (selene_utils2.py:97-140)."""
        want = None if chroms is None else {str(c) for c in chroms}
        fai = path + ".fai"
        if want is not None and os.path.exists(fai):
            return cls(cls._read_indexed(path, fai, want))
        out, name, parts = {}, None, []

        def flush():
            if name is not None and (want is None or name in want):
                out[name] = np.concatenate(parts) if parts else np.zeros(0, np.uint8)

        with open(path, "rb") as f:
            for line in f:
                if line.startswith(b">"):
                    flush()
                    name, parts = line[1:].split()[0].decode("ascii"), []
                elif want is None or name in want:
                    parts.append(sequence_to_codes(line.strip()))
        flush()
        if want is not None and want - set(out):
            raise KeyError(f"{path}: no record named {sorted(want - set(out))}")
        return cls(out)

    @staticmethod
    def _read_indexed(path, fai, want):
        index = {}
        with open(fai) as f:
            for line in f:
                p = line.rstrip("\n").split("\t")
                if len(p) >= 5:
                    index[p[0]] = tuple(int(v) for v in p[1:5])
        missing = want - set(index)
        if missing:
            raise KeyError(f"{fai}: no record named {sorted(missing)}")
        out = {}
        with open(path, "rb") as f:
            for name in sorted(want, key=lambda c: index[c][1]):
                length, offset, linebases, linewidth = index[name]
                nlines = (length + linebases - 1) // linebases if linebases else 0
                f.seek(offset)
                raw = np.frombuffer(f.read(length + nlines * (linewidth - linebases)), dtype=np.uint8)
                codes = _LUT[raw[(raw != 10) & (raw != 13)]]     # drop the line ends
                if codes.shape[0] < length:
                    raise ValueError(f"{path}: record {name} is shorter than its index entry ({codes.shape[0]} < {length})")
                out[name] = codes[:length]
        return out

    @staticmethod
    def write_fai(path):
        """Write ``path + ".fai"`` (samtools faidx format) for a FASTA with uniform line lengths per record."""
        rows, name, length, offset, lb, lw, pos = [], None, 0, 0, 0, 0, 0
        with open(path, "rb") as f:
            for line in f:
                if line.startswith(b">"):
                    if name is not None:
                        rows.append((name, length, offset, lb, lw))
                    name, length, offset, lb, lw = line[1:].split()[0].decode("ascii"), 0, pos + len(line), 0, 0
                else:
                    n = len(line.rstrip(b"\r\n"))
                    if lb == 0:
                        lb, lw = n, len(line)
                    length += n
                pos += len(line)
        if name is not None:
            rows.append((name, length, offset, lb, lw))
        with open(path + ".fai", "w") as f:
            for r in rows:
                f.write("\t".join(str(v) for v in r) + "\n")
        return path + ".fai"

    @classmethod
    def random(cls, lengths, seed=0, n_runs=0, fast=False):
        """Synthetic genome (tests / benchmarks): uniform bases with `n_runs` runs of N per chromosome.
        `fast`: bases from the generator's raw byte stream (5x quicker for 100 Mb+ chromosomes; a different genome)."""
        chroms = {}
        for k, (name, L) in enumerate(lengths.items() if isinstance(lengths, dict) else lengths):
            rs = np.random.RandomState(seed + 7919 * k)
            c = (np.frombuffer(rs.bytes(int(L)), dtype=np.uint8) & 3) if fast else rs.randint(0, 4, int(L)).astype(np.uint8)
            for _ in range(n_runs):
                s = int(rs.randint(0, max(1, L - 50000)))
                c[s:s + int(rs.randint(1000, 50000))] = N_CODE
            chroms[name] = c
        return cls(chroms)

    def to(self, device):
        self.device = torch.device(device)
        self._dev = {k: torch.from_numpy(v).to(self.device) for k, v in self._host.items()}
        return self

    # ---- selene Genome API used by the reference's drivers -------------------------------------------------
    def get_chrs(self):
        return sorted(self._host)

    def get_chr_lens(self):
        return [(c, self._length(c)) for c in self.get_chrs()]

    def _bounds(self, chrom, start, end, pad):
        """Clamped slice + padding counts of a query, with the reference's error behaviour (pinned by tests/golden/G26, generated
        from `MemmapGenome.get_encoding_from_coords`, selene_utils2.py:231-262): every query that cannot yield `end - start` rows
        fails its `assert` there - without `pad` anything beyond the chromosome (:257), with `pad` a window that does not at least
        touch the chromosome (the padded pieces then do not add up to `end - start`, :261) and any window of negative length."""
        n = self._length(chrom)
        if pad:
            if not (start <= end and start <= n and end >= 0):
                raise AssertionError(f"window [{start}, {end}) does not touch chromosome {chrom} of length {n}: the reference's padded "
                                     "pieces do not add up to end - start (selene_utils2.py:261)")
            qs, qe = max(start, 0), min(end, n)
            return qs, qe, qs - start, end - qe
        if not (0 <= start <= end <= n):   # the reference's error type (its `assert`s, selene_utils2.py:257,261), raised explicitly: survives python -O
            raise AssertionError(f"coordinates [{start}, {end}) exceed chromosome {chrom} of length {n} (selene_utils2.py:257)")
        return start, end, 0, 0

    def _length(self, chrom):
        return int(self._host[chrom].shape[0])

    def _slice(self, chrom, qs, qe, on_dev):
        return self._dev[chrom][qs:qe] if on_dev else self._host[chrom][qs:qe]

    def get_codes_from_coords(self, chrom, start, end, strand="+", pad=False, device=None):
        """The window as base codes: numpy uint8 [end-start], or a ROCm tensor when the genome lives on a device
        (`device=False` forces the host copy)."""
        qs, qe, pl, pr = self._bounds(chrom, start, end, pad)
        on_dev = bool(self._dev) and device is not False
        if on_dev:
            c = self._slice(chrom, qs, qe, True)
            if pl or pr:
                c = torch.cat([torch.full((pl,), N_CODE, dtype=torch.uint8, device=c.device), c,
                               torch.full((pr,), N_CODE, dtype=torch.uint8, device=c.device)])
        else:
            c = self._slice(chrom, qs, qe, False)
            if pl or pr:
                c = np.concatenate([np.full(pl, N_CODE, np.uint8), c, np.full(pr, N_CODE, np.uint8)])
        if strand == "-":
            c = revcomp_codes(c)
        assert c.shape[0] == end - start
        return c

    def get_encoding_from_coords(self, chrom, start, end, strand="+", pad=False):
        """[end-start, 4] rows exactly as `MemmapGenome.get_encoding_from_coords` (`selene_utils2.py:186-262`; pinned query by query
        by tests/golden/G26): 0.25 rows beyond either end with `pad`, '-' = both axes flipped, any other strand symbol = '+'.
        Always float32 - the reference's padded form is float64 only because `np.ones(...) * 0.25` promotes its `hstack`
        (:238-244); every caller converts to float32 (`torch.FloatTensor`, orca_predict.py:324)."""
        return codes_to_encoding(self.get_codes_from_coords(chrom, start, end, strand, pad, device=False))

    def get_encoding_from_coords_check_unk(self, chrom, start, end, strand="+", pad=False):
        """As the reference (`selene_utils2.py:264-272`): it passes `pad=strand`, so the window is ALWAYS padded whatever the
        caller's `pad` says, and the flag looks at the FIRST position's row only (a zero-length window has none: IndexError)."""
        enc = self.get_encoding_from_coords(chrom, start, end, strand=strand, pad=True)
        return enc, bool(np.any(enc[0, :] == 0.25))

    sequence_to_encoding = staticmethod(sequence_to_encoding)



def pack_2bit(codes):
    """uint8 codes [n] (0..3, anything else = N) -> (2-bit array [ceil(n/4)] with base i in bits 2*(i%4).. of byte i//4,
    N bit-mask [ceil(n/8)] with base i in bit i%8 of byte i//8; N positions carry 0 in the 2-bit array)."""
    c = np.asarray(codes, dtype=np.uint8)
    n = c.shape[0]
    isn = c > 3
    b = np.where(isn, 0, c).astype(np.uint8)
    b = np.concatenate([b, np.zeros((-n) % 4, np.uint8)]).reshape(-1, 4)
    two = (b[:, 0] | (b[:, 1] << 2) | (b[:, 2] << 4) | (b[:, 3] << 6)).astype(np.uint8)
    mask = np.packbits(isn, bitorder="little")
    return two, mask


def unpack_2bit(two, mask, start, end):
    """Inverse of pack_2bit on the window [start, end) (host)."""
    i = np.arange(start, end, dtype=np.int64)
    c = (two[i >> 2] >> ((i & 3) * 2).astype(np.uint8)) & 3
    isn = (mask[i >> 3] >> (i & 7).astype(np.uint8)) & 1
    return np.where(isn == 1, N_CODE, c).astype(np.uint8)


class TwoBitGenome(PackedGenome):
    """The same query API on 2 bits per base + a 1-bit N mask ('a 2-bit packing alone cannot represent
    N'): 3/8 of the 1-byte store.  Windows are expanded to base codes on demand - on the host with numpy, in HBM by
    the library's `orca_genome_unpack_2bit` kernel (the expanded window is what `orca_encoder_forward_codes` reads)."""

    def __init__(self, chroms):
        self._host, self._dev, self.device = {}, {}, None
        for k, v in chroms.items():
            v = np.asarray(v, dtype=np.uint8)
            self._host[str(k)] = pack_2bit(v) + (int(v.shape[0]),)

    @classmethod
    def from_packed(cls, genome):
        """From a 1-byte PackedGenome (e.g. `PackedGenome.from_fasta`)."""
        return cls({c: genome._host[c] for c in genome.get_chrs()})

    @classmethod
    def from_fasta(cls, path, chroms=None):
        return cls.from_packed(PackedGenome.from_fasta(path, chroms))

    def nbytes(self):
        return sum(t.nbytes + m.nbytes for t, m, _ in self._host.values())

    def to(self, device):
        self.device = torch.device(device)
        self._dev = {k: (torch.from_numpy(t).to(self.device), torch.from_numpy(m).to(self.device)) for k, (t, m, _) in self._host.items()}
        return self

    def _length(self, chrom):
        return self._host[chrom][2]

    def planes(self, chrom):
        """(2-bit plane, N mask) of a chromosome in HBM - what `Encoder.forward_2bit` reads in place."""
        if chrom not in self._dev:
            raise ValueError("TwoBitGenome.planes: the genome is not resident on a device (.to(device) first)")
        return self._dev[chrom]

    def _slice(self, chrom, qs, qe, on_dev):
        if not on_dev:
            t, m, _ = self._host[chrom]
            return unpack_2bit(t, m, qs, qe)
        import ctypes

        from . import _lib, engine
        t, m = self._dev[chrom]
        out = torch.empty(qe - qs, dtype=torch.uint8, device=t.device)
        if qe > qs:
            ctx = engine.get_context(t.device)
            _lib.check(_lib.load().orca_genome_unpack_2bit(ctx.handle, ctypes.c_void_p(t.data_ptr()), ctypes.c_void_p(m.data_ptr()), qs, qe - qs,
                                                           ctypes.c_void_p(out.data_ptr())), "orca_genome_unpack_2bit")
        return out
