"""Per-site plug-in surface: the object the reference caller holds as ``network``.

Reference behaviour being mirrored (python/caller_calling.py:612-654, 863-868;
python/MixtureOfExpertsAdvanced.py:487-589):

    network = torch.load(path); network.eval(); network.providePredictions = True
    out = network(featureDict, ref_segment)        # under torch.no_grad()

with ``featureDict = {allele: (FloatTensor[R, L, C], FloatTensor[R', L, C] | None)}`` in allele order
and ``ref_segment = FloatTensor[1, L, 5]``; ``out`` is ``{(a, b): 0-dim tensor}`` over unordered allele
pairs in first-seen ``itertools.product`` order, or the 5-tuple ``(mix, e0, e1, e2, meta)`` when
``providePredictions`` is set.

All arithmetic runs on the GPU through the C ABI (logits AND pair posteriors); this module only
packs inputs and shapes outputs.  ``score_sites`` is the throughput form of the same call: many
sites per launch, results returned in order (the site-batching shim of SURVEY.md 8f N3).
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

from . import netspec as ns
from .engine import Engine


def _to_u8(x, what: str) -> np.ndarray:
    """Pileup tensors reach the reference as float copies of the featurizer's uint8 arrays
    (caller_calling.py:633-639); the engine consumes the bytes themselves."""
    if hasattr(x, "detach"):
        x = x.detach().cpu().numpy()
    x = np.asarray(x)
    if x.dtype == np.uint8:
        return np.ascontiguousarray(x)
    u = x.astype(np.uint8)
    if not np.array_equal(u.astype(x.dtype), x):
        raise ValueError(f"{what}: pileup values must be integers in 0..255 (the featurizer's uint8 alphabet)")
    return np.ascontiguousarray(u)


class _Lazy:
    """torch is imported on first use (the module itself needs NumPy only)."""
    def __init__(self, make):
        self._make, self._value = make, None

    def clone(self):
        if self._value is None:
            self._value = self._make()
        return self._value.clone()


def _one_hot_meta():
    import torch
    return torch.tensor([1.0, 0.0, 0.0])


_SINGLE_EXPERT_META = _Lazy(_one_hot_meta)      # a single-expert model's mixing weights (MixtureOfExpertsAdvanced.py:575-579)


def pair_keys(alleles: Sequence) -> List[Tuple]:
    """Unordered allele pairs in first-seen itertools.product order (MixtureOfExpertsAdvanced.py:562-564)."""
    n = len(alleles)
    return [(alleles[i], alleles[j]) for i in range(n) for j in range(i, n)]


class _BatchedOperator:
    """``network.moeMerged``: the batched call form ``dnn(tensors, numAllelesPerSite, numReadsPerAllele,
    reference_segments, *extra)`` (MixtureOfExpertsDNNFast.py:128-134, MixtureOfExpertsAdvanced.py:161).
    ``tensors`` are [sumR, C, L] like the reference's (uint8 or integer-valued float)."""

    def __init__(self, engine: Engine, spec: ns.ModelSpec):
        self.engine = engine
        self.meta = object() if spec.ensemble else None     # the reference tests `moeMerged.meta is not None`

    def __call__(self, tensors, numAllelesPerSite, numReadsPerAllele, reference_segments=None, *extra, **kw):
        import torch
        t0 = _to_u8(tensors[0], "tensors[0]")
        t1 = _to_u8(tensors[1], "tensors[1]") if tensors[1] is not None else None
        ref = _to_u8(reference_segments, "reference_segments") if (
            reference_segments is not None and self.engine.program.uses_ref) else None
        logits, meta = self.engine.forward(
            t0, numReadsPerAllele[0], numAllelesPerSite, t1,
            numReadsPerAllele[1] if t1 is not None else None, ref, layout_rcl=True)
        if self.engine.n_experts == 1:
            return torch.from_numpy(logits[0][:, None].copy())
        return [torch.from_numpy(logits[e][:, None].copy()) for e in range(3)], torch.from_numpy(meta)

    forward = __call__


class ScoringNetwork:
    """Counterpart of ``MoEMergedWrapperAdvanced`` backed by the HIP engine."""

    def __init__(self, spec: ns.ModelSpec, state, device: int = 0, providePredictions: bool = False, fused: bool = True,
                 winograd: bool = True, arithmetic=None):
        self.spec = spec
        self.engine = Engine(spec, state, device=device, fused=fused, winograd=winograd, arithmetic=arithmetic)
        self.moeMerged = _BatchedOperator(self.engine, spec)
        self.providePredictions = providePredictions
        self.training = False

    # torch.nn.Module look-alikes the caller touches
    def eval(self):
        return self

    def train(self, mode: bool = False):
        if mode:
            raise NotImplementedError("inference-only engine")
        return self

    def close(self):
        self.engine.close()

    # -- packing ---------------------------------------------------------------------------------
    @staticmethod
    def _as_array(x) -> np.ndarray:
        if hasattr(x, "detach"):
            x = x.detach().cpu().numpy()                     # a view for a CPU tensor
        return np.asarray(x)

    @staticmethod
    def _bytes_of(arrays: List[np.ndarray], labels: List[str]) -> np.ndarray:
        """The pileups of one technology as ONE contiguous uint8 [sum R, L, C] array.  The reference hands over float copies of
        the featurizer's bytes (caller_calling.py:633-639): they are concatenated first and converted + validated in one pass
        (one NumPy call per site instead of three per allele); only a failed check walks the alleles again to name the culprit."""
        if all(a.dtype == np.uint8 for a in arrays):
            return np.concatenate(arrays, axis=0)
        whole = np.concatenate(arrays, axis=0)
        if whole.dtype == np.uint8:                          # (mixed uint8 / float inputs were promoted by the concatenation)
            return whole
        u = whole.astype(np.uint8)
        if not np.array_equal(u, whole):
            for a, label in zip(arrays, labels):
                _to_u8(a, label)                             # raises, naming the allele
            raise ValueError("pileup values must be integers in 0..255 (the featurizer's uint8 alphabet)")
        return u

    @classmethod
    def _pack(cls, sites: Sequence[Tuple[Dict, object]], need_ref: bool = True):
        """[(featureDict, segment)] -> contiguous channels-last uint8 batch + counts.  ``need_ref`` False: the model does not
        read the reference segment -- it is not converted."""
        r0, r1, l0, l1, rpa0, rpa1, aps, refs, names = [], [], [], [], [], [], [], [], []
        for feature_dict, segment in sites:
            alleles = list(feature_dict.keys())
            names.append(alleles)
            aps.append(len(alleles))
            for a in alleles:
                first, second = feature_dict[a]
                f = cls._as_array(first)
                r0.append(f)
                l0.append(a)
                rpa0.append(f.shape[0])
                if second is not None:
                    sec = cls._as_array(second)
                    r1.append(sec)
                    l1.append(a)
                    rpa1.append(sec.shape[0])
            if need_ref and segment is not None:
                seg = _to_u8(segment, "ref_segment")
                refs.append(seg.reshape(-1, seg.shape[-2], seg.shape[-1])[0])
        reads0 = cls._bytes_of(r0, [f"allele {a!r}" for a in l0])
        # like the reference (MixtureOfExpertsAdvanced.py:511-516): a missing second tensor anywhere
        # means "no second technology"
        complete = len(r1) == len(r0)
        reads1 = cls._bytes_of(r1, [f"allele {a!r} (second technology)" for a in l1]) if (r1 and complete) else None
        ref = np.stack(refs, axis=0) if (refs and len(refs) == len(sites)) else None
        return (reads0, np.asarray(rpa0, np.int32), reads1,
                np.asarray(rpa1, np.int32) if reads1 is not None else None,
                np.asarray(aps, np.int32), ref, names)

    # -- scoring ---------------------------------------------------------------------------------
    def score_sites(self, sites: Sequence[Tuple[Dict, object]]):
        """Score many sites in ONE engine launch; returns one result per site, in order, each shaped
        exactly like the reference's per-site return value."""
        import torch
        eng = self.engine
        reads0, rpa0, reads1, rpa1, aps, ref, names = self._pack(sites, need_ref=bool(eng.program.uses_ref))
        if eng.program.channels1 and reads1 is None:
            raise ValueError("this model scores two read technologies: every allele needs both tensors")
        logits, meta, post = eng.forward(reads0, rpa0, aps, reads1 if eng.program.channels1 else None,
                                         rpa1 if eng.program.channels1 else None,
                                         ref if eng.program.uses_ref else None, posteriors=True)
        results, col = [], 0
        n_pairs_total = post.shape[1]
        scalars = torch.from_numpy(np.ascontiguousarray(post).reshape(-1)).unbind(0)      # 0-dim tensors of all four rows, with one call
        for s, alleles in enumerate(names):
            keys = pair_keys(alleles)
            n = len(keys)
            rows = [dict(zip(keys, scalars[r * n_pairs_total + col:r * n_pairs_total + col + n])) for r in range(4)]
            col += n
            if self.providePredictions:
                m = torch.from_numpy(meta[s].copy()) if eng.has_meta else _SINGLE_EXPERT_META.clone()
                results.append((rows[0], rows[1], rows[2], rows[3], m))
            else:
                results.append(rows[0])
        return results

    def __call__(self, featureDict, segment):
        return self.score_sites([(featureDict, segment)])[0]

    forward = __call__


class SiteBatcher:
    """Buffers (featureDict, segment, tag) triples and flushes them through ``score_sites`` in launches
    of ``max_sites``; yields (tag, result) in submission order.  Drop this around the hot loop of the
    reference caller (caller_calling.py:872-891) to turn the per-site plug-in into a throughput path."""

    def __init__(self, network: ScoringNetwork, max_sites: int = 4096):
        self.network = network
        self.max_sites = max_sites
        self._pending: List[Tuple[Dict, object, object]] = []

    def submit(self, featureDict, segment, tag=None):
        self._pending.append((featureDict, segment, tag))
        if len(self._pending) >= self.max_sites:
            return self.flush()
        return []

    def flush(self):
        if not self._pending:
            return []
        batch, self._pending = self._pending, []
        results = self.network.score_sites([(fd, seg) for fd, seg, _ in batch])
        return [(tag, res) for (_, _, tag), res in zip(batch, results)]
