"""CPU ORACLE -- TEST INFRASTRUCTURE ONLY.

A plain NumPy restatement of the reference's variant-scoring forward.  It exists to *check* the HIP
engine; nothing in the product path (hello_amd/) imports it.  Allowed importers: tests/,
__graft_entry__.smoke() and bench.py's ``cpu_baseline`` leg.

Parity pinning: the reference's own tests hold no fixture for this path, so
the oracle is pinned against outputs of the reference itself, captured in the build container by
tests/golden/make_fixtures.py (which imports /root/reference/python and pushes this package's seeded
state dicts into the reference modules) and committed under tests/golden/*.npz;
tests/test_oracle_golden.py replays them.

Every function cites the reference file:line it restates (paths relative to /root/reference).
Layout convention follows the reference: activations are [N, C, L] float32.

The convolution arithmetic of the reference lives in a third-party dependency (PyTorch CPU
Conv1d/MaxPool1d/cumsum/Linear kernels; version not pinned by the reference).  Two
interchangeable conv back ends are provided: "numpy" (im2col + float32 matmul, the default and the
independent restatement) and "torch" (torch.nn.functional.conv1d on CPU: the same third-party
kernels the reference calls, used for the timed CPU baseline because it is the faster of the two).
"""
from __future__ import annotations

import itertools
import os
import sys
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hello_amd import netspec as ns          # noqa: E402  (architecture description only)
from hello_amd import weights as wts         # noqa: E402  (weight-norm / batch-norm folding)

F32 = np.float32


# --------------------------------------------------------------------------------------------
# primitives
# --------------------------------------------------------------------------------------------
def conv1d(x, w, b, stride=1, pad=0, groups=1, backend="numpy"):
    """torch.nn.Conv1d semantics (cross-correlation, zero padding), the op behind every
    ``Conv1d`` / ``WeightNormedConv1d`` entry of a reference config (NNTools.py:89-100,791-799).
    x [N, Cin, L], w [Cout, Cin/groups, k], b [Cout] -> [N, Cout, Lout]."""
    if backend == "torch":
        import torch
        import torch.nn.functional as TF
        y = TF.conv1d(torch.from_numpy(np.ascontiguousarray(x)), torch.from_numpy(w), torch.from_numpy(b),
                      stride=stride, padding=pad, groups=groups)
        return y.numpy()
    n, cin, length = x.shape
    cout, cg, k = w.shape
    if pad:
        x = np.pad(x, ((0, 0), (0, 0), (pad, pad)))
    lout = (length + 2 * pad - k) // stride + 1
    out = np.empty((n, cout, lout), dtype=F32)
    og = cout // groups
    for g in range(groups):
        xg = x[:, g * cg:(g + 1) * cg]
        # cols [N, Lout, cg*k], K ordered (channel, tap) like w.reshape(cout, cg*k)
        idx = (np.arange(lout) * stride)[:, None] + np.arange(k)[None, :]
        cols = xg[:, :, idx]                                   # [N, cg, Lout, k]
        cols = cols.transpose(0, 2, 1, 3).reshape(n, lout, cg * k)
        wg = w[g * og:(g + 1) * og].reshape(og, cg * k)
        out[:, g * og:(g + 1) * og] = np.matmul(cols, wg.T).transpose(0, 2, 1)
    out += b[None, :, None]
    return out


def max_pool1d(x, k, stride, pad=0):
    """torch.nn.MaxPool1d (floor mode), architectures/read_convolver.py:49-56."""
    if pad:
        x = np.pad(x, ((0, 0), (0, 0), (pad, pad)), constant_values=-np.inf)
    lout = (x.shape[2] - k) // stride + 1
    idx = (np.arange(lout) * stride)[:, None] + np.arange(k)[None, :]
    return x[:, :, idx].max(axis=3)


def activation(x, kind):
    if kind == "relu":
        return np.maximum(x, F32(0))
    if kind == "softplus":
        # torch.nn.Softplus(beta=1, threshold=20)
        return np.where(x > 20, x, np.log1p(np.exp(np.minimum(x, F32(20))))).astype(F32)
    if kind == "none":
        return x
    raise ValueError(kind)


def segment_sum(d, slots):
    """``reduceSlots`` (MixtureOfExpertsAdvanced.py:23-34): running sum over dim 0, picked at the
    last row of every segment, minus the previous pick.  Restated literally (cumsum in float32, then
    difference) so that the oracle carries the same rounding pattern as the reference."""
    slots = np.asarray(slots, dtype=np.int64)
    running = np.cumsum(d, axis=0, dtype=F32)
    picks = running[np.cumsum(slots) - 1]
    shifted = np.concatenate([np.zeros_like(picks[:1]), picks[:-1]], axis=0)
    return picks - shifted


# --------------------------------------------------------------------------------------------
# sub-network interpreter  (NNTools.Network.forward, NNTools.py:633-657)
# --------------------------------------------------------------------------------------------
def layer_norm_channels(x, gamma, beta, eps):
    """``LayerNormModule`` on a [N, C, L] tensor (NNTools.py:802-828): torch.nn.LayerNorm over the channels of every
    position -- mean and biased variance over C, (x - mean) / sqrt(var + eps) * gamma + beta, in float32."""
    mean = x.mean(axis=1, keepdims=True, dtype=F32)
    var = np.mean((x - mean) ** 2, axis=1, keepdims=True, dtype=F32)
    return ((x - mean) / np.sqrt(var + F32(eps)) * gamma[None, :, None] + beta[None, :, None]).astype(F32)


def run_net(nodes: Sequence[ns.Node], x, folded, backend="numpy", state=None):
    for node in nodes:
        if isinstance(node, ns.Conv):
            w, b = folded[node.key]
            x = conv1d(x, w, b, node.stride, node.pad, node.groups, backend)
            if node.norm == "ln":
                x = layer_norm_channels(x, *wts.layer_norm_params(node, state))
            x = activation(x, node.act)
        elif isinstance(node, ns.MaxPool):
            x = max_pool1d(x, node.k, node.stride, node.pad)
        elif isinstance(node, ns.Residual):
            # ResidualBlock.forward: ffNetwork(x) + shNetwork(x)  (NNTools.py:582-583)
            short = run_net(node.shortcut, x, folded, backend, state) if node.shortcut else x
            x = run_net(node.body, x, folded, backend, state) + short
        elif isinstance(node, ns.Head):
            # AdaptiveAvgPool1d(1) -> Flatten -> [BatchNorm folded] -> Linear  (NNTools.py:517-566)
            w, b = folded[node.key]
            pooled = x.mean(axis=2, dtype=F32)
            x = pooled @ w.T + b
        elif isinstance(node, ns.Mix):
            # Fork(Noop, SelectArgument(pick)) then LinearCombination (xattn_subtract.py:14-42,
            # NNTools.py:754-777): result = 0; result += c_i * arg_i
            allele, sites = x
            x = F32(node.coeffs[0]) * allele + F32(node.coeffs[1]) * sites[node.pick]
        elif isinstance(node, ns.Select):
            x = x[node.index]
        elif isinstance(node, ns.Transpose):
            x = np.ascontiguousarray(np.swapaxes(x, node.dim0, node.dim1))
        elif isinstance(node, ns.Concat):
            x = np.concatenate(list(x), axis=1)
        else:
            raise TypeError(node)
    return x


# --------------------------------------------------------------------------------------------
# MoEAttention  (MixtureOfExpertsAdvanced.py:71-252)
# --------------------------------------------------------------------------------------------
class Oracle:
    def __init__(self, spec: ns.ModelSpec, state: Dict[str, np.ndarray], backend: str = "numpy"):
        self.spec = spec
        self.folded = wts.fold(spec, state)
        self.state = state
        self.backend = backend

    def _net(self, name, x):
        return run_net(self.spec.nets[name], x, self.folded, self.backend, self.state)

    def compress_and_predict(self, frames_allele, alleles_per_site, idx):
        """MixtureOfExpertsAdvanced.py:117-159.  The site-level compressor call of line 136 is
        evaluated too (its result is only consumed by configs whose Mix picks element 0)."""
        compressor = f"compressor{idx}"
        ca = self._net(compressor, frames_allele)
        frames_site = segment_sum(frames_allele, alleles_per_site)
        cs0 = self._net(compressor, frames_site)
        cs1 = segment_sum(ca, alleles_per_site)
        xattn = f"xattn{idx}"
        pred = None
        if self.spec.has(xattn):
            e0 = np.repeat(cs0, alleles_per_site, axis=0)
            e1 = np.repeat(cs1, alleles_per_site, axis=0)
            pred = self._net(xattn, (ca, (e0, e1)))
        return pred, (cs0, cs1), ca

    def forward(self, tensors, alleles_per_site, reads_per_allele, reference_segments=None):
        """``MoEAttention.forward`` (MixtureOfExpertsAdvanced.py:161-252).

        tensors = (T0, T1|None) with T [sumR, C, L] (any dtype; cast to float32 like ``.float()``).
        Returns logits [sumA, 1] (single expert) or ([e0, e1, e2], meta [S, 3])."""
        spec = self.spec
        if spec.family == "merged":
            return self.forward_merged(tensors, alleles_per_site, reads_per_allele)
        aps = np.asarray(alleles_per_site, dtype=np.int64)
        rc0 = self._net("read_convolver0", np.asarray(tensors[0], dtype=F32))
        frames0 = segment_sum(rc0, reads_per_allele[0])
        p0, f0, ca0 = self.compress_and_predict(frames0, aps, 0)
        self.last = {"frames0": frames0, "ca0": ca0}
        if not spec.has("read_convolver1"):
            return p0

        rc1 = self._net("read_convolver1", np.asarray(tensors[1], dtype=F32))
        frames1 = segment_sum(rc1, reads_per_allele[1])
        p1, f1, ca1 = self.compress_and_predict(frames1, aps, 1)
        self.last.update({"frames1": frames1, "ca1": ca1})

        if spec.has("compressor2"):
            frames2 = frames0 + frames1
            p2, f2, _ = self.compress_and_predict(frames2, aps, 2)
            site_frames_for_meta = f2[0]
        elif spec.has("xattn2"):
            ca2 = self._net("combiner0", (ca0, ca1))
            cs2 = self._net("combiner1", (f0[1], f1[1]))
            p2 = self._net("xattn2", (ca2, (None, np.repeat(cs2, aps, axis=0))))
            site_frames_for_meta = cs2
        else:
            p2 = None
            site_frames_for_meta = segment_sum(frames0 + frames1, aps)

        meta = None
        if spec.has("meta"):
            ref = None if reference_segments is None else np.asarray(reference_segments, dtype=F32)
            logits = self._net("meta", (site_frames_for_meta, ref))
            logits = logits - logits.max(axis=-1, keepdims=True)
            e = np.exp(logits)
            meta = (e / e.sum(axis=-1, keepdims=True)).astype(F32)

        if p0 is None and p1 is None:
            return p2
        if p2 is None:
            p2 = np.zeros_like(p0)
        return [p0, p1, p2], meta


def _forward_merged(self, tensors, alleles_per_site, reads_per_allele):
    """``MoEMergedAdvanced.forward`` (MixtureOfExpertsAdvanced.py:398-484): read conv -> allele sums -> allele conv
    (:332-342); per-site frames (:369-370, :422-436); expert input a - (repeat(s) - a) with useAdditive, else
    cat(a, repeat(s) - a) along channels (:372-383; a hybrid model without useAdditive raises in the reference,
    :436, and here); separate meta read convolvers sum each site's reads directly (:344-367, :438-458); meta softmax
    over dim 1 (:480)."""
    spec = self.spec
    aps = np.asarray(alleles_per_site, dtype=np.int64)
    frames0 = segment_sum(self._net("readConv0", np.asarray(tensors[0], dtype=F32)), reads_per_allele[0])
    a0 = self._net("alleleConv0", frames0)
    frames = {"frames0": frames0}
    hybrid = spec.has("readConv1") and tensors[1] is not None
    additive = getattr(spec, "use_additive", True)
    if hybrid and not additive:
        raise RuntimeError("Boolean value of Tensor with more than one value is ambiguous")      # :436
    if hybrid:
        frames["frames1"] = segment_sum(self._net("readConv1", np.asarray(tensors[1], dtype=F32)), reads_per_allele[1])
        a1 = self._net("alleleConv1", frames["frames1"])
        a2 = self._net("alleleConvCombiner", (a0, a1)) if spec.has("alleleConvCombiner") else a0 + a1
    s0 = segment_sum(a0, aps)

    def expert(idx, allele, site):
        remaining = np.repeat(site, aps, axis=0) - allele
        x = allele - remaining if additive else np.concatenate([allele, remaining], axis=1)
        return self._net(f"expert{idx}", x)

    p0 = expert(0, a0, s0)
    self.last = dict(frames, ca0=a0)
    if not hybrid:
        return p0
    s1 = segment_sum(a1, aps)
    s2 = self._net("siteConvCombiner", (s0, s1)) if spec.has("siteConvCombiner") else segment_sum(a2, aps)
    p1, p2 = expert(1, a1, s1), expert(2, a2, s2)
    site_meta = s2
    if spec.has("readConv0Meta"):
        # preparePerSiteFramesFromReads (:344-367): every read of a site summed directly (reduceFrames)
        site_reads = [np.add.reduceat(np.asarray(r, dtype=np.int64), np.concatenate([[0], np.cumsum(aps)])[:-1])
                      for r in reads_per_allele]

        def per_site(name, x, counts):
            conv = self._net(name, np.asarray(x, dtype=F32))
            off = np.concatenate([[0], np.cumsum(counts)])
            return np.stack([conv[off[i]:off[i + 1]].sum(axis=0, dtype=F32) for i in range(len(counts))]).astype(F32)
        m0 = per_site("readConv0Meta", tensors[0], site_reads[0])
        m1 = per_site("readConv1Meta", tensors[1], site_reads[1])
        site_meta = self._net("siteConvCombiner", (m0, m1)) if spec.has("siteConvCombiner") else m0 + m1
    logits = self._net("meta", site_meta)
    logits = logits - logits.max(axis=1, keepdims=True)
    e = np.exp(logits)
    return [p0, p1, p2], (e / e.sum(axis=1, keepdims=True)).astype(F32)


Oracle.forward_merged = _forward_merged


# --------------------------------------------------------------------------------------------
# per-site wrapper  (MoEMergedWrapperAdvanced, MixtureOfExpertsAdvanced.py:487-589)
# --------------------------------------------------------------------------------------------
def sigmoid(x):
    return (F32(1) / (F32(1) + np.exp(-x.astype(F32)))).astype(F32)


def pair_order(n_alleles: int) -> List[Tuple[int, int]]:
    """First-seen order of unordered pairs in ``itertools.product(alleles, alleles)`` (:562-564)."""
    seen, order = set(), []
    for i, j in itertools.product(range(n_alleles), range(n_alleles)):
        if (i, j) in seen or (j, i) in seen:
            continue
        seen.add((i, j))
        order.append((i, j))
    return order


def expert_pair_probability(p, i, j):
    """``expertProbability`` (:543-548) for the target vector with ones at alleles i and j."""
    t = np.zeros_like(p)
    t[i] = 1
    t[j] = 1
    return np.exp(np.sum(np.log(p * t + (F32(1) - p) * (F32(1) - t) + F32(1e-10)), dtype=F32)).astype(F32)


def posteriors(expert_probs: Sequence[np.ndarray], meta: np.ndarray):
    """Mixture over experts of the pair probabilities (:573-584).  expert_probs: 3 arrays [A] of
    per-allele sigmoid outputs; meta [3].  Returns (mix, e0, e1, e2) arrays over pair_order(A)."""
    n = expert_probs[0].shape[0]
    pairs = pair_order(n)
    per_expert = [np.array([expert_pair_probability(p.astype(F32), i, j) for i, j in pairs], dtype=F32)
                  for p in expert_probs]
    mix = meta[0] * per_expert[0] + meta[1] * per_expert[1] + meta[2] * per_expert[2]
    return mix.astype(F32), per_expert[0], per_expert[1], per_expert[2]


class WrapperOracle:
    """Counterpart of ``MoEMergedWrapperAdvanced`` for ONE site (:520-589)."""

    def __init__(self, spec, state, backend="numpy", provide_predictions=True):
        self.net = Oracle(spec, state, backend)
        self.providePredictions = provide_predictions

    def __call__(self, feature_dict, segment):
        """feature_dict: {allele: (array [R, L, C], array [R', L, C] | None)} in allele order;
        segment: [1, L, 5].  Mirrors ``_singleFeatureDictData`` (:493-518) then ``forward``."""
        alleles = list(feature_dict.keys())
        rpa0 = [feature_dict[a][0].shape[0] for a in alleles]
        t0 = np.concatenate([np.transpose(feature_dict[a][0], (0, 2, 1)) for a in alleles], axis=0)
        second = [feature_dict[a][1] for a in alleles]
        if any(s is None for s in second):
            t1, rpa1 = None, None
        else:
            rpa1 = [s.shape[0] for s in second]
            t1 = np.concatenate([np.transpose(s, (0, 2, 1)) for s in second], axis=0)
        result = self.net.forward((t0, t1), [len(alleles)], (rpa0, rpa1), segment)
        if self.net.spec.ensemble:
            experts, meta = result
            meta = meta[0]
            experts = [sigmoid(e[:, 0]) for e in experts]
        else:
            e0 = sigmoid(result[:, 0])
            experts = [e0, np.zeros_like(e0), np.zeros_like(e0)]
            meta = np.array([1, 0, 0], dtype=F32)
        mix, p0, p1, p2 = posteriors(experts, meta)
        pairs = [(alleles[i], alleles[j]) for i, j in pair_order(len(alleles))]
        as_dict = lambda v: dict(zip(pairs, v))          # noqa: E731
        if self.providePredictions:
            return as_dict(mix), as_dict(p0), as_dict(p1), as_dict(p2), meta
        return as_dict(mix)


# --------------------------------------------------------------------------------------------
# batched convenience over a hello_amd.synth.SiteBatch (channels-last uint8 input)
# --------------------------------------------------------------------------------------------
def forward_batch(oracle: Oracle, batch, chunk_sites: int = 64):
    """Run ``oracle.forward`` over a SiteBatch in site chunks; returns (logits [n_experts, A], meta
    [S,3] | None).  Chunking changes nothing mathematically except the cumsum prefix each segment sum
    sees (the reference itself differs by <=4e-6 between per-site and batched calls)."""
    logits, metas = [], []
    for lo in range(0, batch.n_sites, chunk_sites):
        sub = batch.site_slice(lo, min(lo + chunk_sites, batch.n_sites))
        t0 = np.transpose(sub.reads0, (0, 2, 1))
        t1 = None if sub.reads1 is None else np.transpose(sub.reads1, (0, 2, 1))
        out = oracle.forward((t0, t1), sub.alleles_per_site, (sub.reads_per_allele0, sub.reads_per_allele1),
                             sub.ref_onehot)
        if isinstance(out, tuple):
            experts, meta = out
            logits.append(np.stack([e[:, 0] for e in experts], axis=0))
            metas.append(meta)
        else:
            logits.append(out[:, 0][None, :])
    logits = np.concatenate(logits, axis=1)
    meta = np.concatenate(metas, axis=0) if metas else None
    return logits, meta
