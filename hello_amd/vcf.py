"""Genotype calls and VCF records from pair posteriors.

What the reference does with the network's output, per site:
  * per-shard caller: best unordered allele pair -> QUAL = -10 log10(1 - p) capped at p = 1 - 1e-8,
    genotype indices against the ALT list, one normalised VCF line
    (reference python/caller_calling.py:698-743, python/vcfFromContigs.py:139-227);
  * final VCF: the same on the meta-weighted mean of the three experts' posteriors
    (python/prepareVcf.py:36-105,138-168).

ALT alleles are emitted in sorted order (the reference's order is ``list(set(...))``, i.e. Python's
per-process hash order; SURVEY.md section 7 defines call identity on the ALT *set*).  Everything here is
string work on the host; the posteriors themselves come from the GPU (hello_amd.wrapper).
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence, Tuple

QUAL_CAP = 1 - 1e-8            # "Quality score restricted to value 80" (prepareVcf.py:61)


@dataclass
class Call:
    chromosome: str
    position: int               # 0-based, after normalisation
    ref: str
    alts: Tuple[str, ...]
    genotype: Tuple[int, int]
    qual: float
    info: str = "HELLO"
    filter: str = "PASS"

    def line(self) -> str:
        return "%s\t%d\t.\t%s\t%s\t%f\t%s\t%s\tGT\t%s" % (
            self.chromosome, self.position + 1, self.ref, ",".join(self.alts), self.qual, self.filter,
            self.info, "/".join(str(g) for g in self.genotype))

    def identity(self):
        """What two pipelines must agree on for a call to count as identical."""
        alleles = (self.ref,) + self.alts
        return (self.chromosome, self.position, self.ref, frozenset(self.alts),
                tuple(sorted(alleles[g] for g in self.genotype)))


def _pad_left_if_empty(pos: int, ref: str, alts: List[str], genome: str):
    alts = [a.replace("-", "") for a in alts]
    if min(len(x) for x in [ref] + alts) > 0:
        return False, pos, ref, alts
    anchor = genome[pos - 1]
    return True, pos - 1, anchor + ref, [anchor + a for a in alts]


def normalise(pos: int, ref: str, alts: Sequence[str], genome: str):
    """Trim shared suffix bases (re-anchoring on the previous reference base whenever an allele would become
    empty), then shared prefix bases while every allele keeps at least one (vcfFromContigs.py:176-209)."""
    _, pos, ref, alts = _pad_left_if_empty(pos, ref, list(alts), genome)
    if not alts or all(a == ref for a in alts):
        return None
    while True:
        trimmed = len({x[-1] for x in [ref] + alts}) == 1
        if trimmed:
            ref, alts = ref[:-1], [a[:-1] for a in alts]
        padded, pos, ref, alts = _pad_left_if_empty(pos, ref, alts, genome)
        if not (trimmed or padded):
            break
    while len(ref) > 1 and all(len(a) > 1 for a in alts) and len({x[0] for x in [ref] + alts}) == 1:
        pos, ref, alts = pos + 1, ref[1:], [a[1:] for a in alts]
    return pos, ref, alts


def call_site(posteriors: Dict[Tuple[str, str], float], chromosome: str, start: int, length: int,
              genome: str, info: str = "HELLO") -> Optional[Call]:
    """Pair posteriors of one site ({(a, b): p}, values float or 0-dim tensors) -> Call, or None when the
    site holds no non-reference allele."""
    ref_allele = genome[start:start + length]
    best_p, best_pair = max((float(p), pair) for pair, p in posteriors.items())
    qual = -10.0 * math.log10(1.0 - min(best_p, QUAL_CAP))
    alts = sorted(set(best_pair) - {ref_allele})
    if alts:
        genotype = tuple(0 if a == ref_allele else alts.index(a) + 1 for a in best_pair)
    else:
        genotype = (0, 0)
        alts = sorted({a for pair in posteriors for a in pair} - {ref_allele})
        if not alts:
            return None
    norm = normalise(start, ref_allele, alts, genome)
    if norm is None:
        return None
    pos, ref, alts = norm
    return Call(chromosome, pos, ref, tuple(alts), genotype, qual, info)


def mean_posteriors(experts: Sequence[Dict], meta: Sequence[float]) -> Dict:
    """Meta-weighted mean of the three experts' pair posteriors (prepareVcf.py:154-163), in float64."""
    w = [float(m) for m in meta]
    return {pair: sum(float(e[pair]) * wi for e, wi in zip(experts, w)) for pair in experts[0]}


def call_from_prediction(prediction, chromosome: str, start: int, length: int, genome: str) -> Optional[Call]:
    """``prediction`` = what the network returns for a site: a {pair: p} dict, or the 5-tuple
    (mix, e0, e1, e2, meta) -- then the final-VCF rule (mean of experts) is applied."""
    if isinstance(prediction, dict):
        return call_site(prediction, chromosome, start, length, genome)
    _, e0, e1, e2, meta = prediction
    return call_site(mean_posteriors((e0, e1, e2), meta), chromosome, start, length, genome)


# --------------------------------------------------------------------------------------------
# the per-shard ``.features`` records and what prepareVcf makes of them
# --------------------------------------------------------------------------------------------
def feature_record(prediction, chromosome: str, start: int, length: int) -> dict:
    """The dict the per-shard caller appends to its ``.features`` list for one site (reference
    python/caller_calling.py:743-754): chromosome, position (0-based site start), length of the reference
    allele, the meta-expert weights and the three experts' pair posteriors.  ``prediction`` is the network's
    5-tuple (mix, e0, e1, e2, meta).  Values are stored as plain floats / a float32 NumPy array, which is
    what prepareVcf.py:138-163 converts them to anyway, so the reference's own tool reads the file."""
    import numpy as np
    _, e0, e1, e2, meta = prediction
    plain = lambda d: {pair: float(v) for pair, v in d.items()}          # noqa: E731
    meta = np.asarray(meta.detach().cpu().numpy() if hasattr(meta, "detach") else meta, dtype=np.float32)
    return {"chromosome": chromosome, "position": int(start), "length": int(length), "meta": meta,
            "expertPredictions": (plain(e0), plain(e1), plain(e2))}


def write_features(path: str, records: Sequence[dict]) -> str:
    """``<prefix>.features``: one pickled list per shard (caller_calling.py:895-898)."""
    import pickle
    with open(path, "wb") as fh:
        pickle.dump(list(records), fh)
    return path


@dataclass
class ShardCalls:
    """prepareVcf.py:126-176 for one shard: per-expert calls, the call of the expert the meta-expert trusts
    most, the call on the meta-weighted mean (the one the final VCF is built from), and the choices BED rows."""
    expert: Tuple[List[Optional[Call]], List[Optional[Call]], List[Optional[Call]]]
    best: List[Optional[Call]]
    mean: List[Optional[Call]]
    choices: List[Tuple[str, int, int, int]]


def calls_from_features(records: Sequence[dict], genomes: Dict[str, str]) -> ShardCalls:
    """``genomes``: chromosome -> reference sequence (the reference reads it through its ReferenceCache)."""
    expert: Tuple[List, List, List] = ([], [], [])
    best, mean, choices = [], [], []
    for rec in records:
        chrom, pos, length = rec["chromosome"], rec["position"], rec["length"]
        genome = genomes[chrom]
        per_expert = [call_site(p, chrom, pos, length, genome) for p in rec["expertPredictions"]]
        for lst, c in zip(expert, per_expert):
            lst.append(c)
        meta = [float(m) for m in rec["meta"]]
        choice = max(range(3), key=lambda i: (meta[i], -i))              # np.argmax: first maximum
        best.append(per_expert[choice])
        mean.append(call_site(mean_posteriors(rec["expertPredictions"], meta), chrom, pos, length, genome))
        choices.append((chrom, pos, pos + length, choice))
    return ShardCalls(expert, best, mean, choices)
