"""Model state handling: seeded synthetic weights and inference-time folding.

Trained HELLO weights are not distributed with the reference, so parity and
benchmarks use *synthetic* state dicts generated here.  A state dict uses exactly the reference's key
names and tensor shapes (``moeMerged.<net>.network.<i>[.ffNetwork|.shNetwork].network.<j>.conv1d.
{bias,weight_g,weight_v}`` ...), so the same dict can be pushed into the reference modules with
``load_state_dict`` (done only by tests/golden/make_fixtures.py in the build container) and into this
package's compiler.

Folding (done once at load instead of at every forward):
  * weight norm  W = g * v / ||v||_2, norm over every dim but 0 (reference NNTools.py:780-799 wraps
    ``torch.nn.utils.weight_norm``, which recomputes W in a forward pre-hook);
  * BatchNorm1d in eval mode  y = (x - mean) / sqrt(var + eps) * gamma + beta  folded into the
    preceding conv / the following linear (reference NNTools.py:102-108, 537-555).
"""
from __future__ import annotations

import zlib
from typing import Dict, Tuple

import numpy as np

from . import netspec as ns

BN_EPS = 1e-5  # torch.nn.BatchNorm1d default


def _rng(seed: int, key: str) -> np.random.Generator:
    return np.random.Generator(np.random.PCG64([seed, zlib.crc32(key.encode())]))


def _uniform(rng, shape, bound):
    return rng.uniform(-bound, bound, size=shape).astype(np.float32)


def synth_state(spec: ns.ModelSpec, seed: int = 0, gain: float = 1.0) -> Dict[str, np.ndarray]:
    """Deterministic state dict for ``spec``.

    v ~ U(+-sqrt(3/fan_in)) (unit-variance preserving up to the ReLU), g = gain_l * ||v|| with a
    per-layer jitter so that the weight-norm fold is exercised with g != ||v||, bias ~ U(+-0.1).
    The very first conv of a read convolver sees raw 0..254 bytes (the reference never rescales its
    input, MixtureOfExpertsAdvanced.py:162), so its gain is divided by 128 to keep activations O(1).
    """
    state: Dict[str, np.ndarray] = {}
    for net_name, nodes in spec.nets.items():
        first = True
        for node in ns.walk(nodes):
            rng = _rng(seed, node.key)
            if isinstance(node, ns.Conv):
                shape = (node.cout, node.cin // node.groups, node.k)
                fan_in = shape[1] * shape[2]
            else:
                shape = (node.cout, node.cin)
                fan_in = node.cin
            layer_gain = gain * float(rng.uniform(0.9, 1.3))
            if first and net_name.startswith(("read_convolver", "readConv")):
                layer_gain /= 128.0
            if first and net_name.startswith(("compressor", "alleleConv")) and "Combiner" not in net_name:
                layer_gain /= 24.0          # its input is a sum over ~tens of reads
            if first and net_name.startswith("expert"):
                layer_gain /= 4.0           # 2a - s over a handful of alleles
            if ".ffNetwork.network.3" in node.key:
                layer_gain *= 0.4           # keep x + f(x) from doubling the scale per block
            if isinstance(node, ns.Head):
                layer_gain *= 2.0
            first = False
            v = _uniform(rng, shape, np.sqrt(3.0 / fan_in))
            bias = _uniform(rng, (node.cout,), 0.1)
            if node.norm == "wn":
                axes = tuple(range(1, v.ndim))
                norm = np.sqrt(np.sum(v.astype(np.float64) ** 2, axis=axes, keepdims=True))
                chan = rng.uniform(0.8, 1.2, size=norm.shape)
                state[node.key + ".weight_v"] = v
                state[node.key + ".weight_g"] = (layer_gain * chan * norm).astype(np.float32)
                state[node.key + ".bias"] = bias
            else:
                state[node.key + ".weight"] = (v * np.float32(layer_gain)).astype(np.float32)
                state[node.key + ".bias"] = bias
            if node.norm == "ln":
                k = node.bn_key + ".normer"                       # LayerNormModule.normer = torch.nn.LayerNorm(cout)
                state[k + ".weight"] = rng.uniform(0.7, 1.3, size=node.cout).astype(np.float32)
                state[k + ".bias"] = _uniform(rng, (node.cout,), 0.1)
            if node.norm == "bn":
                c = node.cout if isinstance(node, ns.Conv) else node.cin
                k = node.bn_key
                state[k + ".weight"] = rng.uniform(0.7, 1.3, size=c).astype(np.float32)
                state[k + ".bias"] = _uniform(rng, (c,), 0.1)
                state[k + ".running_mean"] = _uniform(rng, (c,), 0.2)
                state[k + ".running_var"] = rng.uniform(0.5, 1.5, size=c).astype(np.float32)
                state[k + ".num_batches_tracked"] = np.array(1, dtype=np.int64)
    return state


def _wn_weight(state, key) -> np.ndarray:
    v = np.asarray(state[key + ".weight_v"], dtype=np.float32)
    g = np.asarray(state[key + ".weight_g"], dtype=np.float32)
    axes = tuple(range(1, v.ndim))
    norm = np.sqrt(np.sum(v * v, axis=axes, keepdims=True, dtype=np.float32))
    return (v * (g / norm)).astype(np.float32)


def fold_node(node, state) -> Tuple[np.ndarray, np.ndarray]:
    """Effective (weight, bias) of a Conv ([cout, cin/groups, k]) or Head ([cout, cin]) node.  A layer pickled
    with ``bias=None`` has no bias entry (zeros); a BatchNorm1d with ``affine=False`` no weight / bias entries
    (gamma 1, beta 0); its eps is the module's own (``node.bn_eps``)."""
    if node.norm == "wn":
        w = _wn_weight(state, node.key)
    else:
        w = np.asarray(state[node.key + ".weight"], dtype=np.float32)
    b = np.asarray(state[node.key + ".bias"], dtype=np.float32) if node.key + ".bias" in state else \
        np.zeros(w.shape[0], np.float32)
    if node.norm == "bn":
        k = node.bn_key
        mean = np.asarray(state[k + ".running_mean"], np.float32)
        var = np.asarray(state[k + ".running_var"], np.float32)
        gamma = np.asarray(state[k + ".weight"], np.float32) if k + ".weight" in state else np.ones_like(mean)
        beta = np.asarray(state[k + ".bias"], np.float32) if k + ".bias" in state else np.zeros_like(mean)
        scale = gamma / np.sqrt(var + np.float32(getattr(node, "bn_eps", BN_EPS)))
        if isinstance(node, ns.Conv):
            # conv -> BN: scale output channels
            w = w * scale[:, None, None]
            b = (b - mean) * scale + beta
        else:
            # BN -> linear: y = W (scale*(x-mean)+beta) + b
            shift = beta - mean * scale
            b = b + w @ shift
            w = w * scale[None, :]
    return np.ascontiguousarray(w, np.float32), np.ascontiguousarray(b, np.float32)


def layer_norm_params(node, state) -> Tuple[np.ndarray, np.ndarray, float]:
    """(gamma, beta, eps) of the LayerNorm that follows a Conv node with norm "ln" (NNTools.py:802-828)."""
    k = node.bn_key + ".normer"
    gamma = np.asarray(state[k + ".weight"], np.float32) if k + ".weight" in state else np.ones(node.cout, np.float32)
    beta = np.asarray(state[k + ".bias"], np.float32) if k + ".bias" in state else np.zeros(node.cout, np.float32)
    return gamma, beta, float(getattr(node, "bn_eps", 1e-5))


def fold(spec: ns.ModelSpec, state) -> Dict[str, Tuple[np.ndarray, np.ndarray]]:
    """key -> (weight, bias) for every parametrised node of ``spec``."""
    out = {}
    for nodes in spec.nets.values():
        for node in ns.walk(nodes):
            out[node.key] = fold_node(node, state)
    return out


def count_params(state) -> int:
    return int(sum(np.asarray(v).size for k, v in state.items() if not k.endswith("num_batches_tracked")))
