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


def write_shard(path: str, sites: Sequence[CandidateSite]) -> str:
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
    with open(path, "wb") as fh:
        np.savez_compressed(fh, **payload)
    return path


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
