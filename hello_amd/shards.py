"""Pre-extracted candidate-site shards: what the reference's per-shard caller holds for every site right before
it featurises and scores it (reference python/caller_calling.py:795-843: reads sampled from the BAMs, candidate
alleles and read -> allele support from AlleleSearcherLite, the site's reference window), stored as flat arrays.

BAM / FASTA ingestion, hotspot detection and allele assembly are upstream of the scoring path (SURVEY.md section 2,
rows 9-13: they need pysam and the C++ searcher) and stay with the reference; a shard file is the hand-over point.
One ``.npz`` per shard (the reference's unit of work, ``shard<N>.txt``, python/call.py:162-221):

  site arrays   chromosome (str), start, stop (allele span, genome coordinates), window_start, ref_off[S+1] into
                ref (ASCII bytes of every site's reference window, wide enough for the feature window and one anchor
                base left of the site), alleles_per_site[S], allele strings (one per allele), has_second
  read arrays   per technology t in (0, 1): reads_per_allele<t>[A] (0 = no supporting read: the engine gets the
                all-zero dummy read, c++/src/AlleleSearcherLiteFiltered.cpp:1037-1043), bases / quals (concatenated),
                read_off, cigars (BAM packing length << 4 | op), cigar_off, ref_start, mapq, orientation, hp
"""
from __future__ import annotations

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


def _payload(sites: Sequence[CandidateSite]) -> dict:
    hybrid = any(r1 is not None for s in sites for _, _, r1 in s.alleles)
    payload = dict(
        chromosome=np.array([s.chromosome for s in sites]), start=np.array([s.start for s in sites], np.int64),
        stop=np.array([s.stop for s in sites], np.int64), window_start=np.array([s.window_start for s in sites], np.int64),
        ref=np.frombuffer("".join(s.reference for s in sites).encode("ascii"), np.uint8).copy(),
        ref_off=np.concatenate([[0], np.cumsum([len(s.reference) for s in sites])]).astype(np.int64),
        alleles_per_site=np.array([len(s.alleles) for s in sites], np.int32),
        alleles=np.array([a for s in sites for a, _, _ in s.alleles] or [""]), has_second=np.array(int(hybrid)))
    for tech in (0, 1) if hybrid else (0,):
        groups = [(r0 if tech == 0 else (r1 or [])) for s in sites for _, r0, r1 in s.alleles]
        payload.update({f"{k}{tech}": v for k, v in _pack_reads(groups).items()})
    return payload


def write_shard(path: str, sites: Sequence[CandidateSite]) -> str:
    with open(path, "wb") as fh:
        np.savez_compressed(fh, **_payload(sites))
    return path


class PackedShard:
    """A shard kept as the flat arrays of its file: what the featurizer launch and the record loop need, without a
    Python object per read (unpacking a shard into ``AlignedRead`` objects and flattening them again costs ~7 us per
    read on the host -- two orders of magnitude more than scoring the read on the GPU)."""

    def __init__(self, arrays: dict):
        self.z = arrays
        self.hybrid = bool(int(arrays["has_second"]))
        self.alleles_per_site = np.asarray(arrays["alleles_per_site"], np.int32)
        self.n_sites = int(self.alleles_per_site.shape[0])
        self.allele_off = np.concatenate([[0], np.cumsum(self.alleles_per_site, dtype=np.int64)])
        self.allele_names = [str(a) for a in arrays["alleles"]][:int(self.allele_off[-1])]
        self.chromosomes = [str(c) for c in arrays["chromosome"]]
        self.start = np.asarray(arrays["start"], np.int64)
        self.stop = np.asarray(arrays["stop"], np.int64)
        self.window_start = np.asarray(arrays["window_start"], np.int64)
        self.ref_off = np.asarray(arrays["ref_off"], np.int64)
        self._ref_text = np.asarray(arrays["ref"], np.uint8).tobytes().decode("ascii")

    @classmethod
    def from_file(cls, path: str) -> "PackedShard":
        with np.load(path, allow_pickle=False) as z:
            return cls({k: z[k] for k in z.files})

    @classmethod
    def from_sites(cls, sites: Sequence[CandidateSite]) -> "PackedShard":
        return cls(_payload(sites))

    def __len__(self):
        return self.n_sites

    def names(self, s: int) -> List[str]:
        return self.allele_names[int(self.allele_off[s]):int(self.allele_off[s + 1])]

    def reference(self, s: int) -> str:
        return self._ref_text[int(self.ref_off[s]):int(self.ref_off[s + 1])]

    def has_reads(self, tech: int) -> bool:
        return f"reads_per_allele{tech}" in self.z

    def featurizer_arrays(self, tech: int) -> dict:
        """The arrays of ``hello_engine_featurize`` for technology ``tech`` -- element for element what
        ``featurizer.pack_sites`` builds from the unpacked sites (an allele without supporting reads gets the dummy
        read with an empty CIGAR), by index arithmetic on the file's arrays."""
        z = self.z
        counts = np.asarray(z[f"reads_per_allele{tech}"], np.int64)
        n_alleles = int(self.allele_off[-1])
        assert counts.shape[0] == n_alleles
        rpa = np.maximum(counts, 1)                                   # the dummy read of an unsupported allele
        new_off = np.concatenate([[0], np.cumsum(rpa)])
        old_off = np.concatenate([[0], np.cumsum(counts)])
        n_old, n_new = int(old_off[-1]), int(new_off[-1])
        allele_of_old = np.repeat(np.arange(n_alleles), counts)
        src = np.full(n_new, -1, np.int64)                            # new read -> read of the file, -1 = dummy
        src[new_off[allele_of_old] + (np.arange(n_old) - old_off[allele_of_old])] = np.arange(n_old)
        real = src >= 0
        pick = np.where(real, src, 0)

        def take(name, default, dtype):
            a = np.asarray(z[f"{name}{tech}"])
            if a.shape[0] == 0:
                return np.full(n_new, default, dtype)
            return np.where(real, a[pick], default).astype(dtype)
        read_off, cigar_off = np.asarray(z[f"read_off{tech}"], np.int64), np.asarray(z[f"cigar_off{tech}"], np.int64)
        read_len = np.where(real, np.diff(read_off)[pick] if n_old else 0, 0)
        cigar_len = np.where(real, np.diff(cigar_off)[pick] if n_old else 0, 0)
        allele_of_new = np.repeat(np.arange(n_alleles), rpa)
        site_of_allele = np.repeat(np.arange(self.n_sites), self.alleles_per_site)
        pad = lambda a, dtype: np.concatenate([np.asarray(a, dtype), np.zeros(1, dtype)])     # noqa: E731 (never empty)
        return dict(
            bases=pad(z[f"bases{tech}"], np.uint8), quals=pad(z[f"quals{tech}"], np.uint8),
            read_off=np.concatenate([[0], np.cumsum(read_len)]).astype(np.int64),
            cigars=pad(z[f"cigars{tech}"], np.uint32),
            cigar_off=np.concatenate([[0], np.cumsum(cigar_len)]).astype(np.int64),
            ref_start=take("ref_start", 0, np.int64), mapq=take("mapq", 40, np.uint8),
            orientation=take("orientation", 1, np.int8), hp=take("hp", 0, np.uint8),
            site_of_read=site_of_allele[allele_of_new].astype(np.int32),
            ref=pad(z["ref"], np.uint8), ref_off=self.ref_off.astype(np.int64),
            window_start=self.window_start, asm_start=self.start, asm_stop=self.stop,
            reads_per_allele=rpa.astype(np.int32), alleles_per_site=self.alleles_per_site)


def read_shard(path: str) -> List[CandidateSite]:
    with np.load(path, allow_pickle=False) as z:
        hybrid = bool(int(z["has_second"]))
        ref = z["ref"].tobytes().decode("ascii")
        ref_off, aps = z["ref_off"], z["alleles_per_site"]
        allele_names = [str(a) for a in z["alleles"]]

        def unpack(tech):
            counts, bases, quals = z[f"reads_per_allele{tech}"], z[f"bases{tech}"].tobytes().decode("ascii"), z[f"quals{tech}"]
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
            sites.append(CandidateSite(str(z["chromosome"][s]), int(z["start"][s]), int(z["stop"][s]),
                                       ref[int(ref_off[s]):int(ref_off[s + 1])], int(z["window_start"][s]), alleles))
    return sites
