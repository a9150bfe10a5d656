"""CPU ORACLE -- TEST INFRASTRUCTURE ONLY (pileup-tensor producer).

Literal restatement of ``AlleleSearcherLiteFiltered::computeFeaturesColoredSimple`` and its colour
helpers (reference c++/src/AlleleSearcherLiteFiltered.cpp:971-1180, constants :369-384).  The C++ cannot
be built here (Boost.Python/numpy/log are absent).  **Pinned** by outputs of the reference itself: its own
Python statement of the encoding (python/test_aligner.py:15-180, the code its unit test holds the C++ to) is
executed in the build container by tests/golden/make_fixtures.py on the test's two cases (:279-384) and 336 random
reads (tests/golden/featurizer_reference.npz); tests/test_featurizer.py replays them bit for bit, beside the
hand-worked answers.  That encoder and the C++ are the same function only on CIGARs without clips, with insertions
no worse in quality than the base before them and deletions wholly inside or outside the window (the Python one
ignores clips and indexes out of the window otherwise): the fixtures stay inside that domain; outside it this
oracle follows the C++ text, and the GPU tests compare against it on random CIGARs of every kind.

Track order (:376-382): read base, reference base, base quality, mapping quality, strand, allele-position
marker, haplotag.  Output uint8 [reads][feature_length][channels].
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

import numpy as np

BAM_CMATCH, BAM_CINS, BAM_CDEL, BAM_CREF_SKIP, BAM_CSOFT_CLIP, BAM_CHARD_CLIP, BAM_CPAD, BAM_CEQUAL, BAM_CDIFF = range(9)

BASE_QUALITY_CAP, MAPPING_QUALITY_CAP = 40, 60            # :373-374
POSITIVE_STRAND, NEGATIVE_STRAND = 70, 240                # :375-376
ALLELE_POSITION, BACKGROUND_POSITION = 240, 70            # :377-378


def base_color(base: str) -> int:
    """:971-985: offsets 40 (A, G) / 30 (T, C), stride 70; anything else (gap '*', N) is 0."""
    return {"A": 40 + 3 * 70, "G": 40 + 2 * 70, "T": 30 + 1 * 70, "C": 30}.get(base, 0)


def quality_color(qual: int, cap: int) -> int:
    """:988-999: int(254 * (1.0 * min(qual, cap) / cap)) in double arithmetic."""
    return int(254 * (1.0 * min(qual, cap) / cap))


def strand_color(orientation: int) -> int:
    return POSITIVE_STRAND if orientation > 0 else NEGATIVE_STRAND       # :1002-1005


def hp_color(hp: int) -> int:
    return 120 if hp == 1 else (240 if hp == 2 else 0)                   # :1019-1028


@dataclass
class Read:
    bases: str
    quals: Sequence[int]
    cigar: Sequence[Tuple[int, int]]         # (operation, length)
    ref_start: int
    mapq: int = 40
    orientation: int = 1
    hp: int = 0


def features_for_reads(reads: Sequence[Read], reference: str, window_start: int, assembly_start: int,
                       assembly_stop: int, feature_length: int, include_hp: bool) -> np.ndarray:
    """:1032-1180 for the reads that support one allele (an empty list gives the single all-zero dummy
    read of :1037-1043).  ``reference`` is the window string starting at genome position ``window_start``."""
    channels = 7 if include_hp else 6
    if len(reads) == 0:
        return np.zeros((1, feature_length, channels), dtype=np.uint8)
    out = np.zeros((len(reads), feature_length, channels), dtype=np.uint8)
    mid = (assembly_start + assembly_stop) // 2
    start = mid - feature_length // 2
    end = start + feature_length

    def between(x, y, z):
        return x <= y < z

    def position_color(pos_in_window):                                     # :1008-1016
        inside = (assembly_start - window_start <= pos_in_window) and (pos_in_window < assembly_stop - window_start)
        return ALLELE_POSITION if inside else BACKGROUND_POSITION

    for n, rd in enumerate(reads):
        rf, rp = rd.ref_start, 0
        mapq_c = quality_color(rd.mapq, MAPPING_QUALITY_CAP)
        strand_c = strand_color(rd.orientation)
        hp_c = hp_color(rd.hp)
        for op, length in rd.cigar:
            if op in (BAM_CEQUAL, BAM_CDIFF, BAM_CMATCH):                  # :1074-1096
                for j in range(length):
                    if between(start, rf + j, end):
                        f = rf + j - start
                        out[n, f, 0] = base_color(rd.bases[rp + j])
                        out[n, f, 1] = base_color(reference[rf + j - window_start])
                        out[n, f, 2] = quality_color(rd.quals[rp + j], BASE_QUALITY_CAP)
                        out[n, f, 3] = mapq_c
                        out[n, f, 4] = strand_c
                        out[n, f, 5] = position_color(rf + j - window_start)
                        if include_hp:
                            out[n, f, 6] = hp_c
                rf += length
                rp += length
            elif op == BAM_CDEL:                                           # :1098-1125, falls through to :1126
                if between(start, rf - 1, end):
                    for i in range(rf - 1, rf + length):
                        if not between(start, i, end):
                            continue
                        f = i - start
                        out[n, f, 1] = base_color(reference[i - window_start])
                        out[n, f, 3] = mapq_c
                        out[n, f, 4] = strand_c
                        out[n, f, 5] = position_color(i - window_start)
                        if include_hp:
                            out[n, f, 6] = hp_c
                    f = rf - 1 - start
                    out[n, f, 0] = base_color("*")
                    out[n, f, 2] = quality_color(rd.quals[rp - 1], BASE_QUALITY_CAP) if rp > 0 else 0
                rf += length
            elif op == BAM_CREF_SKIP:                                      # :1126-1129
                rf += length
            elif op == BAM_CINS:                                           # :1131-1160, falls through to :1161
                if between(start, rf - 1, end):
                    lo = rp - 1 if rp > 0 else rp
                    qual_c = quality_color(min(rd.quals[lo:rp + length]), BASE_QUALITY_CAP)
                    f = rf - 1 - start
                    out[n, f, 0] = base_color("*")
                    out[n, f, 1] = base_color(reference[rf - 1 - window_start])
                    out[n, f, 2] = qual_c
                    out[n, f, 3] = mapq_c
                    out[n, f, 4] = strand_c
                    out[n, f, 5] = position_color(rf - 1 - window_start)
                    if include_hp:
                        out[n, f, 6] = hp_c
                rp += length
            elif op == BAM_CSOFT_CLIP:                                     # :1161-1164
                rp += length
            # hard clip / pad: the C++ switch has no case for them (no effect)
    return out
