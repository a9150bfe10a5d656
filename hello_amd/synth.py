"""Seeded synthetic candidate sites (SURVEY.md section 8d generator).

Produces the *batched* argument layout the reference's training/eval loaders hand to the network
(reference MixtureOfExpertsDNNFast.py:165-219 ``collate_function``): all reads of all alleles of all
sites concatenated, plus reads-per-allele and alleles-per-site counts.  Reads are emitted
channels-last, uint8 ``[R, L, C]``, exactly as the C++ featurizer writes them
(c++/src/AlleleSearcherLiteFiltered.cpp:1045,1172); the value alphabet follows
AlleleSearcherLiteFiltered.cpp:369-384,971-1027 (cross-checked by python/test_aligner.py:15-60):

  ch0 read base   {0 gap, 30 C, 100 T, 180 G, 250 A}
  ch1 ref base    same alphabet
  ch2 base qual   int(254*min(q,40)/40)
  ch3 map qual    int(254*min(m,60)/60), constant per read
  ch4 strand      {70, 240}, constant per read
  ch5 position    240 inside the allele span around the window centre, else 70
  ch6 haplotag    {0, 120, 240}, constant per read (only with 7 channels)

Reads shorter than the window leave all channels zero on the uncovered flanks.  An allele without
supporting reads gets ONE all-zero dummy read (AlleleSearcherLiteFiltered.cpp:1037-1043).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional

import numpy as np

BASE_CODE = np.array([250, 30, 180, 100], dtype=np.uint8)   # A, C, G, T  (one-hot order 'ACGT')
ALLELE_PROBS = np.array([0.10, 0.70, 0.15, 0.05])


@dataclass
class SiteBatch:
    reads0: np.ndarray                      # uint8 [R0, L, C0]
    reads_per_allele0: np.ndarray           # int32 [A]
    alleles_per_site: np.ndarray            # int32 [S]
    ref_onehot: np.ndarray                  # uint8 [S, L, 5]
    reads1: Optional[np.ndarray] = None     # uint8 [R1, L, C1]
    reads_per_allele1: Optional[np.ndarray] = None

    @property
    def n_sites(self) -> int:
        return int(self.alleles_per_site.shape[0])

    @property
    def n_alleles(self) -> int:
        return int(self.reads_per_allele0.shape[0])

    def site_slice(self, lo: int, hi: int) -> "SiteBatch":
        """Sites [lo, hi) as a new batch (views where possible)."""
        aoff = np.concatenate([[0], np.cumsum(self.alleles_per_site)])
        a0, a1 = int(aoff[lo]), int(aoff[hi])

        def cut(reads, rpa):
            if reads is None:
                return None, None
            roff = np.concatenate([[0], np.cumsum(rpa)])
            return reads[int(roff[a0]):int(roff[a1])], rpa[a0:a1]

        r0, c0 = cut(self.reads0, self.reads_per_allele0)
        r1, c1 = cut(self.reads1, self.reads_per_allele1)
        return SiteBatch(r0, c0, self.alleles_per_site[lo:hi], self.ref_onehot[lo:hi], r1, c1)


def _split_reads(rng, total, n_alleles):
    """Dirichlet-multinomial(alpha=1) split of each site's reads over its alleles, vectorised:
    returns int32 [sum(n_alleles)] with zero-read alleles bumped to one dummy read (flag array)."""
    site_of = np.repeat(np.arange(len(n_alleles)), n_alleles)
    gam = rng.exponential(1.0, size=site_of.shape[0])           # Dirichlet(1) = normalised Exp(1)
    gsum = np.zeros(len(n_alleles))
    np.add.at(gsum, site_of, gam)
    p = gam / gsum[site_of]
    # multinomial by inverse-CDF on per-read uniforms would be O(R); use the cheap, still seeded,
    # largest-remainder rounding of total*p (exact totals, deterministic)
    want = p * total[site_of]
    base = np.floor(want).astype(np.int64)
    rem = want - base
    short = total - np.bincount(site_of, weights=base, minlength=len(n_alleles)).astype(np.int64)
    # give the `short[s]` largest remainders of site s one extra read
    order = np.lexsort((-rem, site_of))
    start = np.concatenate([[0], np.cumsum(n_alleles)])[:-1]
    rank = np.empty_like(order)
    rank[order] = np.arange(order.shape[0]) - start[site_of[order]]
    base += (rank < short[site_of]).astype(np.int64)
    dummy = base == 0
    base[dummy] = 1
    return base.astype(np.int32), dummy


def _make_reads(rng, site_of_read, dummy_read, ref_codes, channels, tech, marker_w):
    n = site_of_read.shape[0]
    L = ref_codes.shape[1]
    match_p = 0.98 if tech == "illumina" else 0.95
    out = np.zeros((n, L, channels), dtype=np.uint8)
    ref = ref_codes[site_of_read]                                   # [n, L]
    alphabet = np.array([0, 30, 100, 180, 250], dtype=np.uint8)
    noise = alphabet[rng.integers(0, 5, size=(n, L))]
    keep = rng.random((n, L)) < match_p
    out[:, :, 0] = np.where(keep, ref, noise)
    out[:, :, 1] = ref
    q = rng.integers(10, 41, size=(n, L))
    out[:, :, 2] = (254 * np.minimum(q, 40) // 40).astype(np.uint8)
    m = rng.integers(5, 61, size=n)
    out[:, :, 3] = (254 * np.minimum(m, 60) // 60).astype(np.uint8)[:, None]
    out[:, :, 4] = np.where(rng.random(n) < 0.5, 70, 240).astype(np.uint8)[:, None]
    pos = np.arange(L)[None, :]
    w = marker_w[site_of_read][:, None]
    centre = L // 2
    inside = (pos >= centre - w // 2) & (pos < centre - w // 2 + w)
    out[:, :, 5] = np.where(inside, 240, 70).astype(np.uint8)
    if channels == 7:
        out[:, :, 6] = (120 * rng.integers(0, 3, size=n)).astype(np.uint8)[:, None]
    if tech == "illumina":
        span = rng.integers(100, L + 1, size=n)
        left = (rng.random(n) * (L - span + 1)).astype(np.int64)
        covered = (pos >= left[:, None]) & (pos < (left + span)[:, None])
        out *= covered[:, :, None].astype(np.uint8)
    out[dummy_read] = 0
    return out


def make_sites(n_sites: int, seed: int = 0, coverage=30, channels: int = 6, tech: str = "illumina",
               window: int = 150, hybrid_coverage=None, max_reads: Optional[int] = None,
               channels1: int = 6) -> SiteBatch:
    """``coverage`` is a number (Poisson mean) or a (lo, hi) tuple: per-site mean ~ U{lo..hi}.
    ``hybrid_coverage`` adds a second ("pacbio") read set with that coverage."""
    rng = np.random.Generator(np.random.PCG64(seed))
    n_alleles = (rng.choice(4, size=n_sites, p=ALLELE_PROBS) + 1).astype(np.int32)
    ref_idx = rng.integers(0, 4, size=(n_sites, window))
    ref_codes = BASE_CODE[ref_idx]
    ref_onehot = np.zeros((n_sites, window, 5), dtype=np.uint8)
    np.put_along_axis(ref_onehot, ref_idx[:, :, None], 1, axis=2)
    marker_w = rng.integers(1, 11, size=n_sites)

    def one_tech(cov, ch, tech_name, cap):
        if isinstance(cov, (tuple, list)):
            lam = rng.integers(cov[0], cov[1] + 1, size=n_sites).astype(np.float64)
        else:
            lam = np.full(n_sites, float(cov))
        total = np.clip(rng.poisson(lam), 1, cap).astype(np.int64)
        rpa, dummy_allele = _split_reads(rng, total, n_alleles)
        allele_of_read = np.repeat(np.arange(rpa.shape[0]), rpa)
        site_of_allele = np.repeat(np.arange(n_sites), n_alleles)
        reads = _make_reads(rng, site_of_allele[allele_of_read], dummy_allele[allele_of_read],
                            ref_codes, ch, tech_name, marker_w)
        return reads, rpa

    cap0 = max_reads if max_reads is not None else (1000 if tech == "illumina" else 128)
    reads0, rpa0 = one_tech(coverage, channels, tech, cap0)
    reads1 = rpa1 = None
    if hybrid_coverage is not None:
        reads1, rpa1 = one_tech(hybrid_coverage, channels1, "pacbio", 128)
    return SiteBatch(reads0, rpa0, n_alleles, ref_onehot, reads1, rpa1)


def allele_names(batch: SiteBatch):
    """Synthetic allele strings per site ('A', 'AT', 'ATT', ...): first allele plays the reference."""
    names = []
    for a in batch.alleles_per_site:
        names.append(["A" + "T" * i for i in range(int(a))])
    return names
