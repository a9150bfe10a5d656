"""Pre-extracted candidate-site shards: what the reference's per-shard caller holds for every site right before
it featurises and scores it (reference python/caller_calling.py:795-843: reads sampled from the BAMs, candidate
alleles and read -> allele support from AlleleSearcherLite, the site's reference window), stored as flat arrays.

BAM / FASTA ingestion, hotspot detection and allele assembly are upstream of the scoring path (SURVEY.md section 2,
rows 9-13: they need pysam and the C++ searcher) and stay with the reference; a shard file is the hand-over point.
One file per shard (the reference's unit of work, ``shard<N>.txt``, python/call.py:162-221):

  site arrays   chromosome names (byte table) + chromosome_of_site, start, stop (allele span, genome coordinates),
                window_start, ref_off[S+1] into ref (ASCII bytes of every site's reference window, wide enough for the
                feature window and one anchor base left of the site), alleles_per_site[S], allele strings (byte table:
                allele_text + allele_text_off[A+1]), has_second
  read arrays   per technology t in (0, 1): reads_per_allele<t>[A] (0 = no supporting read: the engine gets the
                all-zero dummy read, c++/src/AlleleSearcherLiteFiltered.cpp:1037-1043), bases / quals (concatenated),
                read_off, cigars (BAM packing length << 4 | op), cigar_off, ref_start, mapq, orientation, hp

File format ``.hshard`` (``write_shard`` / ``PackedShard.from_file``): ``HSHARD01``, a little-endian uint64 header length,
a JSON header {array name: [dtype, shape, byte offset]}, then the raw arrays at 64-byte aligned offsets -- ONE read brings
a shard into memory and every array is a view of that buffer (a reference-sized shard of ~400 sites loads in ~0.1 ms;
the same arrays as a NumPy ``.npz`` take ~4 ms of zip / header parsing per shard, which at the engine's rate is more host
time than the GPU spends scoring them).  ``.npz`` shards (strings as NumPy unicode arrays, optionally compressed) are
still read and written.
"""
from __future__ import annotations

import json
import os
from dataclasses import dataclass, field
from typing import List, Optional, Sequence, Tuple

import numpy as np

from .featurizer import AlignedRead, SiteReads


@dataclass
class CandidateSite:
    chromosome: str
    start: int                                  # allele span [start, stop) in genome coordinates
    stop: int
    reference: str                              # reference window
    window_start: int                           # genome position of reference[0]
    alleles: List[Tuple[str, List[AlignedRead], Optional[List[AlignedRead]]]] = field(default_factory=list)

    def site_reads(self, tech: int) -> SiteReads:
        return SiteReads(self.reference, self.window_start, self.start, self.stop,
                         [(a, (r0 if tech == 0 else (r1 or []))) for a, r0, r1 in self.alleles])

    def base(self, position: int) -> str:
        return self.reference[position - self.window_start]

    def ref_allele(self) -> str:
        return self.reference[self.start - self.window_start:self.stop - self.window_start]


def _pack_reads(groups: Sequence[Sequence[AlignedRead]]):
    bases, quals, cigars, read_off, cigar_off = [], [], [], [0], [0]
    ref_start, mapq, orient, hp, counts = [], [], [], [], []
    for reads in groups:
        counts.append(len(reads))
        for rd in reads:
            bases.append(rd.bases.encode("ascii"))
            quals.append(bytes(bytearray(int(q) & 0xFF for q in rd.quals)))
            read_off.append(read_off[-1] + len(rd.bases))
            cigars += [(int(n) << 4) | int(op) for op, n in rd.cigar]
            cigar_off.append(cigar_off[-1] + len(rd.cigar))
            ref_start.append(rd.ref_start)
            mapq.append(min(int(rd.mapq), 255))
            orient.append(1 if rd.orientation > 0 else -1)
            hp.append(int(rd.hp))
    return dict(reads_per_allele=np.asarray(counts, np.int32),
                bases=np.frombuffer(b"".join(bases), np.uint8).copy(), quals=np.frombuffer(b"".join(quals), np.uint8).copy(),
                read_off=np.asarray(read_off, np.int64), cigars=np.asarray(cigars, np.uint32),
                cigar_off=np.asarray(cigar_off, np.int64), ref_start=np.asarray(ref_start, np.int64),
                mapq=np.asarray(mapq, np.uint8), orientation=np.asarray(orient, np.int8), hp=np.asarray(hp, np.uint8))


def text_table(strings) -> Tuple[np.ndarray, np.ndarray]:
    """A sequence / NumPy array of ASCII strings -> (uint8 blob, int64 offsets [n + 1]), vectorised."""
    arr = np.asarray(strings)
    if arr.size == 0:
        return np.zeros(0, np.uint8), np.zeros(1, np.int64)
    fixed = np.char.encode(arr, "ascii") if arr.dtype.kind == "U" else arr
    width = fixed.dtype.itemsize
    lengths = np.char.str_len(fixed).astype(np.int64)
    off = np.concatenate([[0], np.cumsum(lengths)]).astype(np.int64)
    if width == 0:
        return np.zeros(0, np.uint8), off
    grid = np.frombuffer(fixed.tobytes(), np.uint8).reshape(-1, width)
    return grid[np.arange(width)[None, :] < lengths[:, None]], off


def table_strings(text: np.ndarray, off: np.ndarray) -> List[str]:
    whole = np.asarray(text, np.uint8).tobytes().decode("ascii")
    return [whole[int(a):int(b)] for a, b in zip(off[:-1], off[1:])]


def _payload(sites: Sequence[CandidateSite]) -> dict:
    hybrid = any(r1 is not None for s in sites for _, _, r1 in s.alleles)
    names = sorted({s.chromosome for s in sites})
    index = {c: i for i, c in enumerate(names)}
    chrom_text, chrom_off = text_table(np.array(names, dtype="U"))
    allele_text, allele_off = text_table(np.array([a for s in sites for a, _, _ in s.alleles], dtype="U"))
    payload = dict(
        chromosome_text=chrom_text, chromosome_text_off=chrom_off,
        chromosome_of_site=np.array([index[s.chromosome] for s in sites], np.int32),
        start=np.array([s.start for s in sites], np.int64),
        stop=np.array([s.stop for s in sites], np.int64), window_start=np.array([s.window_start for s in sites], np.int64),
        ref=np.frombuffer("".join(s.reference for s in sites).encode("ascii"), np.uint8).copy(),
        ref_off=np.concatenate([[0], np.cumsum([len(s.reference) for s in sites])]).astype(np.int64),
        alleles_per_site=np.array([len(s.alleles) for s in sites], np.int32),
        allele_text=allele_text, allele_text_off=allele_off, has_second=np.array(int(hybrid)))
    for tech in (0, 1) if hybrid else (0,):
        groups = [(r0 if tech == 0 else (r1 or [])) for s in sites for _, r0, r1 in s.alleles]
        payload.update({f"{k}{tech}": v for k, v in _pack_reads(groups).items()})
    return payload


MAGIC = b"HSHARD01"


def write_flat(path: str, arrays: dict) -> str:
    header, at, chunks = {}, 0, []
    for name, a in arrays.items():
        shape = list(np.shape(a))
        a = np.ascontiguousarray(a)                       # (at least one-dimensional: the header keeps the real shape)
        header[name] = [a.dtype.str, shape, at]
        chunks.append(a)
        at += (a.nbytes + 63) & ~63
    head = json.dumps(header).encode("ascii")
    pad = (-(16 + len(head))) % 64
    with open(path, "wb") as fh:
        fh.write(MAGIC + np.uint64(len(head) + pad).tobytes() + head + b" " * pad)
        for a in chunks:
            fh.write(a.tobytes())
            fh.write(b"\0" * ((-a.nbytes) % 64))
    return path


def _header_entry(path: str, name: str, entry, room: int, base: int = 0):
    """One checked entry of a ``.hshard`` header -> (dtype, shape, offset, bytes).  The header is the file's own claim
    about itself: a negative offset or dimension would make NumPy slice from the END of the buffer and hand back a wrong
    view without an error, so everything is refused here by name."""
    try:
        dtype, shape, at = entry
        dt = np.dtype(dtype)
        shape = list(shape)
    except (TypeError, ValueError) as e:
        raise ValueError(f"malformed shard {path}: header entry of array {name}: {e}") from None
    # JSON numbers that are not integers (1.9, true) would be truncated to a plausible shape by int(): refused instead
    if not all(isinstance(d, int) and not isinstance(d, bool) for d in shape + [at]):
        raise ValueError(f"malformed shard {path}: array {name} has a non-integer offset or dimension ({at!r}, {shape!r})")
    if dt.kind not in "iuf" or dt.itemsize not in (1, 2, 4, 8) or dt.hasobject:
        raise ValueError(f"malformed shard {path}: array {name} has dtype {dtype!r} (integer or float arrays only)")
    if at < 0 or any(d < 0 for d in shape):
        raise ValueError(f"malformed shard {path}: array {name} has a negative offset or dimension ({at}, {shape})")
    count = 1
    for d in shape:
        count *= d
    nbytes = count * dt.itemsize
    if at + nbytes > room:
        raise ValueError(f"malformed shard {path}: array {name} runs past the end of the file")
    if (base + at) % dt.itemsize:
        # the view is handed by pointer to C and to the GPU: its ABSOLUTE position in the file (header length included: a foreign
        # writer need not pad its header to 64 bytes like write_flat) must be aligned to its element
        raise ValueError(f"malformed shard {path}: array {name} starts at byte {base + at} of the file ({at} behind a header that ends at "
                         f"{base}), not a multiple of its {dt.itemsize}-byte element")
    return dt, tuple(shape), at, nbytes


def _refuse_overlaps(path: str, spans) -> None:
    """``spans`` = (name, offset, bytes) of every array of a header: two arrays sharing bytes is a file lying about itself."""
    end, last = 0, None
    for name, at, nbytes in sorted((s for s in spans if s[2] > 0), key=lambda s: s[1]):
        if at < end:
            raise ValueError(f"malformed shard {path}: arrays {last} and {name} overlap")
        end, last = at + nbytes, name


def _header(path: str, raw: bytes, n: int) -> dict:
    if len(raw) != n:
        raise ValueError(f"malformed shard {path}: the header is cut short ({len(raw)} of {n} bytes)")
    try:
        header = json.loads(raw)
    except ValueError as e:
        raise ValueError(f"malformed shard {path}: the header is not JSON ({e})") from None
    if not isinstance(header, dict):
        raise ValueError(f"malformed shard {path}: the header is not a table of arrays")
    return header


def read_flat(path: str) -> dict:
    """One read; every array is a view of the file's buffer."""
    buf = np.fromfile(path, dtype=np.uint8)
    if buf.shape[0] < 16 or buf[:8].tobytes() != MAGIC:
        raise ValueError(f"{path} is not a shard file (no {MAGIC.decode()} magic)")
    n = int(buf[8:16].view(np.uint64)[0])
    if n > buf.shape[0] - 16:
        raise ValueError(f"malformed shard {path}: the header is cut short ({buf.shape[0] - 16} of {n} bytes)")
    header = _header(path, buf[16:16 + n].tobytes(), n)
    base = 16 + n
    out, spans = {}, []
    for name, entry in header.items():
        dt, shape, at, nbytes = _header_entry(path, name, entry, buf.shape[0] - base, base)
        spans.append((name, at, nbytes))
        out[name] = buf[base + at:base + at + nbytes].view(dt).reshape(shape)
    _refuse_overlaps(path, spans)
    return out


def read_flat_arrays(path: str, names: Sequence[str]) -> dict:
    """Only the named arrays of a ``.hshard`` file (those it holds), without reading the rest."""
    with open(path, "rb") as fh:
        head = fh.read(16)
        if len(head) < 16 or head[:8] != MAGIC:
            raise ValueError(f"{path} is not a shard file (no {MAGIC.decode()} magic)")
        n = int(np.frombuffer(head[8:], np.uint64)[0])
        size = os.fstat(fh.fileno()).st_size
        if n > size - 16:
            raise ValueError(f"malformed shard {path}: the header is cut short ({size - 16} of {n} bytes)")
        header = _header(path, fh.read(n), n)
        _refuse_overlaps(path, [(k,) + _header_entry(path, k, e, size - 16 - n, 16 + n)[2:] for k, e in header.items()])
        out = {}
        for name in names:
            if name in header:
                dt, shape, at, nbytes = _header_entry(path, name, header[name], size - 16 - n, 16 + n)
                fh.seek(16 + n + at)
                raw = fh.read(nbytes)
                if len(raw) != nbytes:
                    raise ValueError(f"malformed shard {path}: array {name} runs past the end of the file")
                out[name] = np.frombuffer(raw, dt).reshape(shape)
    return out


def to_npz_arrays(payload: dict) -> dict:
    """The strings as NumPy stores them (``.npz`` shards)."""
    out = {k: v for k, v in payload.items() if k not in ("chromosome_text", "chromosome_text_off", "chromosome_of_site",
                                                          "allele_text", "allele_text_off")}
    names = np.array(table_strings(payload["chromosome_text"], payload["chromosome_text_off"]), dtype="U")
    out["chromosome"] = names[payload["chromosome_of_site"]] if names.size else np.array([], dtype="U1")
    out["alleles"] = np.array(table_strings(payload["allele_text"], payload["allele_text_off"]) or [""])
    return out


def from_npz_arrays(z: dict) -> dict:
    out = {k: v for k, v in z.items() if k not in ("chromosome", "alleles")}
    n_alleles = int(np.asarray(z["alleles_per_site"], np.int64).sum())
    if np.asarray(z["alleles"]).shape[0] < n_alleles:
        raise ValueError(f"malformed shard: {np.asarray(z['alleles']).shape[0]} allele strings for sum(alleles_per_site) = {n_alleles}")
    out["allele_text"], out["allele_text_off"] = text_table(np.asarray(z["alleles"])[:n_alleles])
    names, index = np.unique(np.asarray(z["chromosome"]), return_inverse=True) if np.asarray(z["chromosome"]).size else (np.array([], "U1"), np.zeros(0, np.int64))
    out["chromosome_text"], out["chromosome_text_off"] = text_table(names)
    out["chromosome_of_site"] = index.astype(np.int32)
    return out


def write_shard(path: str, sites: Sequence[CandidateSite], compressed: bool = False) -> str:
    """One shard file: the flat ``.hshard`` format, or -- for a path ending in ``.npz`` -- a NumPy archive (optionally
    compressed; a shard is mostly base / quality bytes, and inflating them costs the host more than scoring them costs
    the GPU)."""
    payload = _payload(sites)
    if not path.endswith(".npz"):
        return write_flat(path, payload)
    with open(path, "wb") as fh:
        (np.savez_compressed if compressed else np.savez)(fh, **to_npz_arrays(payload))
    return path


READ_ARRAYS = ("reads_per_allele", "bases", "quals", "read_off", "cigars", "cigar_off", "ref_start", "mapq", "orientation", "hp")
QUERY_OPS = np.zeros(16, np.int64)
QUERY_OPS[[0, 1, 4, 7, 8]] = 1                    # BAM operations that consume read bases: M, I, S, =, X


class PackedShard:
    """A shard kept as the flat arrays of its file: what the featurizer launch and the record stage need, without a
    Python object per read or per site (unpacking a shard into ``AlignedRead`` objects and flattening them again costs
    ~7 us per read on the host -- two orders of magnitude more than scoring the read on the GPU).

    The file is the hand-over format from the upstream stages, so everything the kernels will index with is checked
    here, vectorised (``validate``): a malformed shard raises ``ValueError`` naming the site or read instead of
    producing wrong pileups or an out-of-bounds access."""

    def __init__(self, arrays: dict, feature_length: int = 150, validate: bool = True):
        if "alleles" in arrays:                              # the arrays of an ``.npz`` shard: strings as NumPy unicode
            arrays = from_npz_arrays(arrays)
        self.z = arrays
        self.hybrid = bool(int(np.asarray(arrays["has_second"]).reshape(-1)[0]))
        self.alleles_per_site = np.asarray(arrays["alleles_per_site"], np.int32)
        self.n_sites = int(self.alleles_per_site.shape[0])
        self.allele_off = np.concatenate([[0], np.cumsum(self.alleles_per_site, dtype=np.int64)])
        self.n_alleles = int(self.allele_off[-1])
        self.allele_text = np.asarray(arrays["allele_text"], np.uint8)
        self.allele_text_off = np.asarray(arrays["allele_text_off"], np.int64)
        self.chromosome_text = np.asarray(arrays["chromosome_text"], np.uint8)
        self.chromosome_text_off = np.asarray(arrays["chromosome_text_off"], np.int64)
        self.chromosome_of_site = np.asarray(arrays["chromosome_of_site"], np.int32)
        self.start = np.asarray(arrays["start"], np.int64)
        self.stop = np.asarray(arrays["stop"], np.int64)
        self.window_start = np.asarray(arrays["window_start"], np.int64)
        self.ref_off = np.asarray(arrays["ref_off"], np.int64)
        self.ref = np.asarray(arrays["ref"], np.uint8)
        self.feature_length = feature_length
        self._names = self._chromosomes = self._chromosome_names = self._ref_text = None
        self._fa = {}
        if validate:
            self.validate(feature_length)

    @classmethod
    def from_file(cls, path: str, feature_length: int = 150) -> "PackedShard":
        if path.endswith(".npz"):
            with np.load(path, allow_pickle=False) as z:
                arrays = {k: z[k] for k in z.files}
        else:
            arrays = read_flat(path)
        try:
            shard = cls(arrays, feature_length)
        except ValueError as e:
            raise ValueError(f"{path}: {e}") from None
        shard.path = path
        return shard

    @classmethod
    def from_sites(cls, sites: Sequence[CandidateSite], feature_length: int = 150) -> "PackedShard":
        return cls(_payload(sites), feature_length)

    def __len__(self):
        return self.n_sites

    # -- strings (the per-site plug-in path, logs and tests; the throughput path works on the byte tables) ---------
    @property
    def chromosome_names(self) -> List[str]:
        if self._chromosome_names is None:
            self._chromosome_names = table_strings(self.chromosome_text, self.chromosome_text_off)
        return self._chromosome_names

    @property
    def allele_names(self) -> List[str]:
        if self._names is None:
            self._names = table_strings(self.allele_text, self.allele_text_off)
        return self._names

    @property
    def chromosomes(self) -> List[str]:
        if self._chromosomes is None:
            names = self.chromosome_names
            self._chromosomes = [names[i] for i in self.chromosome_of_site.tolist()]
        return self._chromosomes

    def names(self, s: int) -> List[str]:
        return self.allele_names[int(self.allele_off[s]):int(self.allele_off[s + 1])]

    def reference(self, s: int) -> str:
        if self._ref_text is None:
            self._ref_text = self.ref.tobytes().decode("ascii")
        return self._ref_text[int(self.ref_off[s]):int(self.ref_off[s + 1])]

    def has_reads(self, tech: int) -> bool:
        """Technology ``tech`` is part of this shard: technology 0 always, technology 1 exactly when ``has_second`` says so
        (``validate`` refuses a file whose arrays disagree with the flag, so what is consumed is what was checked)."""
        return f"reads_per_allele{tech}" in self.z and (tech == 0 or self.hybrid)

    def n_reads(self, tech: int = 0) -> int:
        """Reads the featurizer will write for technology ``tech`` (dummy reads of unsupported alleles included)."""
        if not self.has_reads(tech):
            return 0
        return int(np.maximum(np.asarray(self.z[f"reads_per_allele{tech}"], np.int64), 1).sum())

    # -- validation --------------------------------------------------------------------------------------------
    def validate(self, feature_length: int = 150):
        S, A = self.n_sites, self.n_alleles

        def bad(what, index=None, kind="site"):
            where = ""
            if index is not None:
                i = int(index)
                named = kind == "site" and i < min(S, self.start.shape[0], self.chromosome_of_site.shape[0])
                where = f" ({kind} {i}" + (f", {self.chromosomes[i]}:{int(self.start[i])}" if named else "") + ")"
            raise ValueError(f"malformed shard: {what}{where}")

        # every array the launch's staging block takes (shard_pipeline._fill concatenates them into typed views of one pinned block)
        # must be an integer array whose values fit the staging type: a float or out-of-range array is refused here, by name,
        # not by a casting error in the middle of a launch (or silently wrapped)
        staged = dict(ref=np.uint8, ref_off=np.int64, window_start=np.int64, start=np.int64, stop=np.int64, alleles_per_site=np.int32)
        for tech in (0, 1):
            staged.update({f"{k}{tech}": t for k, t in (("reads_per_allele", np.int32), ("bases", np.uint8), ("quals", np.uint8),
                                                         ("read_off", np.int64), ("cigars", np.uint32), ("cigar_off", np.int64),
                                                         ("ref_start", np.int64), ("mapq", np.uint8), ("orientation", np.int8), ("hp", np.uint8))})
        for name, want in staged.items():
            if name not in self.z:
                continue
            arr = np.asarray(self.z[name])
            if arr.dtype.kind not in "iu":
                bad(f"array {name} has dtype {arr.dtype} (the launch stages it as {np.dtype(want).name}: integer arrays only)")
            if arr.size and arr.dtype != np.dtype(want):
                lim = np.iinfo(want)
                if int(arr.min()) < lim.min or int(arr.max()) > lim.max:
                    bad(f"array {name} ({arr.dtype}) holds values outside {np.dtype(want).name}, the type the launch stages it as")
        if S and int(self.alleles_per_site.min()) < 1:
            bad("a site without alleles", int(np.argmin(self.alleles_per_site)))
        for name in ("chromosome_of_site", "start", "stop", "window_start"):
            if np.asarray(self.z[name]).shape[0] != S:
                bad(f"{name} holds {np.asarray(self.z[name]).shape[0]} entries for {S} sites")
        for text, off, n, what in ((self.allele_text, self.allele_text_off, A, "allele strings"),
                                   (self.chromosome_text, self.chromosome_text_off, None, "chromosome names")):
            if off.shape[0] < 1 or off[0] != 0 or np.any(np.diff(off) < 0) or int(off[-1]) != text.shape[0]:
                bad(f"the offsets of the {what} must run from 0 to the length of their text without decreasing")
            if n is not None and off.shape[0] != n + 1:
                bad(f"{off.shape[0] - 1} allele strings for sum(alleles_per_site) = {n}")
        if S and (int(self.chromosome_of_site.min()) < 0 or int(self.chromosome_of_site.max()) >= self.chromosome_text_off.shape[0] - 1):
            bad("chromosome_of_site points outside the chromosome names")
        if self.ref_off.shape[0] != S + 1 or self.ref_off[0] != 0 or (S and np.any(np.diff(self.ref_off) < 0)):
            bad("ref_off must hold S + 1 non-decreasing offsets from 0")
        if int(self.ref_off[-1]) != self.ref.shape[0]:
            bad(f"ref_off ends at {int(self.ref_off[-1])}, ref holds {self.ref.shape[0]} bytes")
        if S:
            if np.any(self.stop < self.start):
                bad("stop < start", np.argmax(self.stop < self.start))
            # the feature window [mid - L/2, mid - L/2 + L) (AlleleSearcherLiteFiltered.cpp:1031-1036) and the allele span
            # must lie inside the site's reference window
            lo = (self.start + self.stop) // 2 - feature_length // 2
            window_end = self.window_start + np.diff(self.ref_off)
            short = (self.window_start > np.minimum(lo, self.start)) | (window_end < np.maximum(lo + feature_length, self.stop))
            if np.any(short):
                s = int(np.argmax(short))
                bad(f"the reference window [{int(self.window_start[s])}, {int(window_end[s])}) does not cover the feature window "
                    f"[{int(lo[s])}, {int(lo[s]) + feature_length}) and the allele span", s)
        if not self.hybrid:
            stray = sorted(k for k in self.z if k.endswith("1") and k[:-1] in READ_ARRAYS)
            if stray:
                bad(f"has_second is 0 but the file carries arrays of a second technology ({', '.join(stray)})")
        for tech in (0, 1) if self.hybrid else (0,):
            missing = sorted(k for k in READ_ARRAYS if f"{k}{tech}" not in self.z)
            if missing:
                bad(("has_second is set but " if tech else "") + f"the arrays of technology {tech} are missing ({', '.join(missing)})")
            g = lambda k: np.asarray(self.z[f"{k}{tech}"])                            # noqa: E731
            counts, read_off, cigar_off = g("reads_per_allele").astype(np.int64), g("read_off").astype(np.int64), g("cigar_off").astype(np.int64)
            if counts.shape[0] != A:
                bad(f"reads_per_allele{tech} holds {counts.shape[0]} entries for {A} alleles")
            if A and int(counts.min()) < 0:
                bad(f"negative reads_per_allele{tech}", np.argmin(counts), "allele")
            R = int(counts.sum())
            for name in ("ref_start", "mapq", "orientation", "hp"):
                if g(name).shape[0] != R:
                    bad(f"{name}{tech} holds {g(name).shape[0]} entries for {R} reads")
            for name, off, data in (("read_off", read_off, "bases"), ("cigar_off", cigar_off, "cigars")):
                if off.shape[0] != R + 1 or off[0] != 0 or (R and np.any(np.diff(off) < 0)):
                    bad(f"{name}{tech} must hold R + 1 non-decreasing offsets from 0")
                if int(off[-1]) != g(data).shape[0]:
                    bad(f"{name}{tech} ends at {int(off[-1])}, {data}{tech} holds {g(data).shape[0]} entries")
            if g("quals").shape[0] != g("bases").shape[0]:
                bad(f"quals{tech} and bases{tech} differ in length")
            if R:
                cigars = g("cigars").astype(np.int64)
                consumed = np.concatenate([[0], np.cumsum((cigars >> 4) * QUERY_OPS[cigars & 15])])
                query = consumed[cigar_off[1:]] - consumed[cigar_off[:-1]]
                wrong = query != np.diff(read_off)
                if np.any(wrong):
                    r = int(np.argmax(wrong))
                    bad(f"the CIGAR of read {r} of technology {tech} consumes {int(query[r])} bases, the read holds "
                        f"{int(np.diff(read_off)[r])}", np.searchsorted(np.cumsum(counts), r, side="right"), "allele")

    # -- featurizer input --------------------------------------------------------------------------------------
    def featurizer_core(self, tech: int) -> dict:
        """The arrays of ``hello_engine_featurize`` for technology ``tech`` by index arithmetic on the file's arrays: an
        allele without supporting reads gets the dummy read with an empty CIGAR.  The large arrays (bases, quals,
        cigars) are the file's own, not copies, and may be empty."""
        if tech in self._fa:
            return self._fa[tech]
        z = self.z
        counts = np.asarray(z[f"reads_per_allele{tech}"], np.int64)
        n_alleles = self.n_alleles
        assert counts.shape[0] == n_alleles
        rpa = np.maximum(counts, 1)                                   # the dummy read of an unsupported allele
        new_off = np.concatenate([[0], np.cumsum(rpa)])
        old_off = np.concatenate([[0], np.cumsum(counts)])
        n_old, n_new = int(old_off[-1]), int(new_off[-1])
        read_off, cigar_off = np.asarray(z[f"read_off{tech}"], np.int64), np.asarray(z[f"cigar_off{tech}"], np.int64)
        if n_old == n_new:                                            # every allele is supported: the file's arrays as they are
            take = lambda name, default, dtype: np.asarray(z[f"{name}{tech}"]).astype(dtype, copy=False)   # noqa: E731
        else:
            allele_of_old = np.repeat(np.arange(n_alleles), counts)
            src = np.full(n_new, -1, np.int64)                        # new read -> read of the file, -1 = dummy
            src[new_off[allele_of_old] + (np.arange(n_old) - old_off[allele_of_old])] = np.arange(n_old)
            real = src >= 0
            pick = np.where(real, src, 0)

            def take(name, default, dtype):
                a = np.asarray(z[f"{name}{tech}"])
                if a.shape[0] == 0:
                    return np.full(n_new, default, dtype)
                return np.where(real, a[pick], default).astype(dtype)
            read_len = np.where(real, np.diff(read_off)[pick] if n_old else 0, 0)
            cigar_len = np.where(real, np.diff(cigar_off)[pick] if n_old else 0, 0)
            read_off = np.concatenate([[0], np.cumsum(read_len)]).astype(np.int64)
            cigar_off = np.concatenate([[0], np.cumsum(cigar_len)]).astype(np.int64)
        site_of_allele = np.repeat(np.arange(self.n_sites, dtype=np.int32), self.alleles_per_site)
        self._fa[tech] = dict(
            bases=np.asarray(z[f"bases{tech}"], np.uint8), quals=np.asarray(z[f"quals{tech}"], np.uint8),
            read_off=read_off, cigars=np.asarray(z[f"cigars{tech}"], np.uint32), cigar_off=cigar_off,
            ref_start=take("ref_start", 0, np.int64), mapq=take("mapq", 40, np.uint8),
            orientation=take("orientation", 1, np.int8), hp=take("hp", 0, np.uint8),
            site_of_read=np.repeat(site_of_allele, rpa),
            ref=self.ref, ref_off=self.ref_off, window_start=self.window_start, asm_start=self.start, asm_stop=self.stop,
            reads_per_allele=rpa.astype(np.int32), alleles_per_site=self.alleles_per_site)
        return self._fa[tech]

    def featurizer_arrays(self, tech: int) -> dict:
        """``featurizer_core`` with one element of padding behind the large arrays (a host pointer is taken of each, so
        none may be empty) -- element for element what ``featurizer.pack_sites`` builds from the unpacked sites."""
        core = dict(self.featurizer_core(tech))
        for name, dtype in (("bases", np.uint8), ("quals", np.uint8), ("cigars", np.uint32), ("ref", np.uint8)):
            core[name] = np.concatenate([core[name], np.zeros(1, dtype)])
        return core

    # -- record-stage input ------------------------------------------------------------------------------------
    def segment_onehot(self, genomes=None) -> np.ndarray:
        """caller_calling.py:53-97 (get_reference_segment + one_hot_encode) for every site: uint8 [S, L, 5], classes
        A, C, G, T, other -- from the site's reference window (validated to cover the segment)."""
        L = self.feature_length
        lo = (self.start + self.stop) // 2 - L // 2
        index = (self.ref_off[:-1] + (lo - self.window_start))[:, None] + np.arange(L)[None, :]
        return ONE_HOT[BASE_CLASS[self.ref[index]]]


BASE_CLASS = np.full(256, 4, np.uint8)
for _i, _b in enumerate(b"ACGT"):
    BASE_CLASS[_b] = _i
ONE_HOT = np.eye(5, dtype=np.uint8)


def read_shard(path: str) -> List[CandidateSite]:
    """A shard file back as ``CandidateSite`` objects (tests, tools: one Python object per read)."""
    shard = PackedShard.from_file(path)
    z = shard.z
    hybrid = shard.hybrid
    ref = shard.ref.tobytes().decode("ascii")
    ref_off, aps = shard.ref_off, shard.alleles_per_site
    allele_names, chromosomes = shard.allele_names, shard.chromosomes

    def unpack(tech):
        counts, bases, quals = z[f"reads_per_allele{tech}"], np.asarray(z[f"bases{tech}"]).tobytes().decode("ascii"), z[f"quals{tech}"]
        read_off, cigars, cigar_off = z[f"read_off{tech}"], z[f"cigars{tech}"], z[f"cigar_off{tech}"]
        ref_start, mapq, orient, hp = z[f"ref_start{tech}"], z[f"mapq{tech}"], z[f"orientation{tech}"], z[f"hp{tech}"]
        groups, r = [], 0
        for n in counts:
            reads = []
            for _ in range(int(n)):
                lo, hi = int(read_off[r]), int(read_off[r + 1])
                cg = cigars[int(cigar_off[r]):int(cigar_off[r + 1])]
                reads.append(AlignedRead(bases[lo:hi], quals[lo:hi].tolist(), [(int(c & 15), int(c >> 4)) for c in cg],
                                         int(ref_start[r]), int(mapq[r]), int(orient[r]), int(hp[r])))
                r += 1
            groups.append(reads)
        return groups

    g0 = unpack(0)
    g1 = unpack(1) if hybrid else None
    sites, a = [], 0
    for s in range(aps.shape[0]):
        alleles = []
        for _ in range(int(aps[s])):
            alleles.append((allele_names[a], g0[a], g1[a] if hybrid else None))
            a += 1
        sites.append(CandidateSite(chromosomes[s], int(shard.start[s]), int(shard.stop[s]),
                                   ref[int(ref_off[s]):int(ref_off[s + 1])], int(shard.window_start[s]), alleles))
    return sites
