"""CPU ORACLE -- TEST INFRASTRUCTURE ONLY (posterior -> genotype -> VCF record).

Literal restatement of the reference's record emission (vcfFromContigs.py:139-227, prepareVcf.py:36-105,126-176,
caller_calling.py:698-754).  The reference modules cannot be imported here (``vcfFromContigs`` imports Biopython,
``prepareVcf`` / ``caller_calling`` import pysam: ordinary ModuleNotFoundError in this container), but the functions
on this path use neither: tests/golden/make_fixtures.py executes exactly those source ranges in the build container
and records what they return (tests/golden/vcf_reference.json: 160 createVcfRecord cases, 240 callAlleles cases,
three models through caller_calling.vcfRecords with the reference's own network, one .features shard through
prepareVcf.vcfRecords).  **Pinned**: tests/test_vcf.py replays them (record lines equal character for character
in canonical ALT order); the hand-worked cases of each normalisation rule stay beside them.

One deliberate deviation ("VCF-identical"): the reference orders ALT alleles by
``list(set(...))`` (prepareVcf.py:63,75; caller_calling.py:712,718), i.e. by Python's per-process string
hash.  Both this oracle and the product sort them, which is the canonical form two runs can be compared in.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple


def fix_empty_alleles(location: int, ref: str, alts: List[str], genome: str):
    """vcfFromContigs.py:139-160: strip '-' from ALTs; if any allele is empty, prepend the previous
    reference base to all of them and move the record one base left."""
    alts = [a.replace("-", "") for a in alts]
    found = any(len(x) == 0 for x in [ref] + alts)
    if found:
        location -= 1
        pre = genome[location]
        ref = pre + ref
        alts = [pre + a for a in alts]
    return found, location, ref, alts


def create_vcf_record(chromosome: str, position: int, genome: str, ref: str, alts: List[str], gt: Sequence[int],
                      string: str = "HELLO", qual: float = 30.0, qualifier: str = "PASS") -> Optional[str]:
    """vcfFromContigs.py:162-227 for ONE (offset 0) variant: right parsimony with re-extension on empty
    alleles, then left parsimony, then the tab-separated line (QUAL printed with %f)."""
    _, location, ref, alts = fix_empty_alleles(position, ref, alts, genome)
    if len(alts) == 0 or all(a == ref for a in alts):
        return None                                         # reference: nothing is appended (:173-174,224)
    change = True
    while change:
        change = False
        right = {ref[-1]} | {a[-1] for a in alts}
        if len(right) == 1:
            ref = ref[:-1]
            alts = [a[:-1] for a in alts]
            change = True
        found, location, ref, alts = fix_empty_alleles(location, ref, alts, genome)
        change = change or found
    while (len(ref) > 1) and (min(len(a) for a in alts) > 1):
        left = {ref[0]} | {a[0] for a in alts}
        if len(left) != 1:
            break
        location += 1
        ref = ref[1:]
        alts = [a[1:] for a in alts]
    return "%s\t%d\t.\t%s\t%s\t%f\t%s\t%s\tGT\t%s" % (
        str(chromosome), location + 1, ref, ",".join(alts), qual, qualifier, string,
        "/".join(str(x) for x in gt))


def call_alleles(likelihoods: Dict[Tuple[str, str], float], chromosome: str, start: int, length: int,
                 genome: str, string: str = "HELLO") -> Optional[str]:
    """prepareVcf.py:36-105 (same logic as caller_calling.py:698-743): best pair, QUAL capped at 80,
    ALT list, genotype indices, record."""
    ref_allele = genome[start:start + length]
    likelihood, top = sorted([(float(v), k) for k, v in likelihoods.items()], reverse=True)[0]
    likelihood = min(float(likelihood), 1 - 1e-8)
    quality = -10 * math.log10(1 - likelihood)
    alt_alleles = sorted(set(top) - {ref_allele})
    at_site = sorted({a for key in likelihoods for a in key})
    if len(alt_alleles) == 0:
        genotypes = [0, 0]
        alt_alleles = sorted(set(at_site) - {ref_allele})
        if len(alt_alleles) == 0:
            return None
    else:
        genotypes = [0 if a == ref_allele else alt_alleles.index(a) + 1 for a in top]
    return create_vcf_record(chromosome, start, genome, ref_allele, alt_alleles, genotypes, string=string,
                             qual=quality)


def mean_of_experts(expert_predictions: Sequence[Dict], meta: Sequence[float]) -> Dict:
    """prepareVcf.py:154-163: per pair, sum_i expert_i * meta_i in float64."""
    return {pair: sum(float(expert_predictions[i][pair]) * float(meta[i]) for i in range(3))
            for pair in expert_predictions[0]}


def prepare_shard(items: Sequence[dict], genomes: Dict[str, str]):
    """prepareVcf.py:126-176 for one ``.features`` shard: -> (expert0 lines, expert1 lines, expert2 lines,
    best lines, mean lines, choices rows).  A site without an alternative allele contributes None."""
    import numpy as np
    e0, e1, e2, best, mean, choices = [], [], [], [], [], []
    for site in items:
        genome = genomes[site["chromosome"]]
        records = [call_alleles(d, site["chromosome"], site["position"], site["length"], genome)
                   for d in site["expertPredictions"]]
        e0.append(records[0]); e1.append(records[1]); e2.append(records[2])          # noqa: E702
        best.append(records[int(np.argmax(site["meta"]))])
        mean_dict = mean_of_experts(site["expertPredictions"], site["meta"])
        mean.append(call_alleles(mean_dict, site["chromosome"], site["position"], site["length"], genome))
        choices.append("\t".join([site["chromosome"], str(site["position"]), str(site["position"] + site["length"]),
                                  str(int(np.argmax(site["meta"])))]))
    return e0, e1, e2, best, mean, choices
