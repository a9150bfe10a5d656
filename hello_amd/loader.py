"""Model-loader surface: ``load(path)`` stands in for ``torch.load(args.network, map_location='cpu')``
of the reference caller (reference python/caller_calling.py:863-868).

Two on-disk formats are accepted:

  * the reference's ``*.wrapper.dnn`` -- ``torch.save`` of a whole ``MoEMergedWrapperAdvanced`` module
    (reference python/create_model_wrapper.py:7-10).  The pickle names classes from the reference's
    ``MixtureOfExpertsAdvanced`` and ``NNTools`` modules; this loader supplies inert stand-in classes
    under those module names while unpickling, so NO reference source has to be importable, then reads
    the architecture off the module tree and the weights off ``state_dict()``;
  * the build's native file -- an ``.npz`` holding the state dict and the name of one of
    ``hello_amd.netspec.CONFIGS`` (``save_native``).

Unpickling executes pickle opcodes: only load model files you trust (same caveat as the reference's
own ``torch.load``; torch >= 2.6 needs ``weights_only=False`` for whole-module pickles).
"""
from __future__ import annotations

import contextlib
import json
import os
import sys
import types
import zipfile
from typing import Dict, List, Tuple

import numpy as np

from . import netspec as ns

_NATIVE_KEY = "__hello_config__"


# --------------------------------------------------------------------------------------------
# native format
# --------------------------------------------------------------------------------------------
def save_native(path: str, config: str, state: Dict[str, np.ndarray], **config_kwargs) -> None:
    payload = {k: np.asarray(v) for k, v in state.items()}
    payload[_NATIVE_KEY] = np.array(json.dumps({"config": config, "kwargs": config_kwargs}))
    with open(path, "wb") as fh:
        np.savez(fh, **payload)


def _load_native(path: str) -> Tuple[ns.ModelSpec, Dict[str, np.ndarray]]:
    with np.load(path, allow_pickle=False) as z:
        meta = json.loads(str(z[_NATIVE_KEY]))
        state = {k: z[k] for k in z.files if k != _NATIVE_KEY}
    return ns.build(meta["config"], **meta.get("kwargs", {})), state


def _is_native(path: str) -> bool:
    try:
        with zipfile.ZipFile(path) as zf:
            return _NATIVE_KEY + ".npy" in zf.namelist()
    except zipfile.BadZipFile:
        return False


_LFS_MAGIC = b"version https://git-lfs"


def sniff(path: str) -> str:
    """What kind of file ``path`` is, from its first bytes -- before any unpickler sees it:
    "native" (the build's .npz), "torch-zip" (torch.save since 1.6), "torch-legacy" (torch.save before 1.6 or with
    _use_new_zipfile_serialization=False: a bare pickle stream, which torch.load still reads), "gzip", or -- raising a
    ValueError that names the cause and the fix -- a git-LFS pointer (what a fresh clone of the reference holds in models/:
    its .gitattributes tracks *.dnn with LFS, models/README.md:3-9), an empty file, or something that is not a model."""
    with open(path, "rb") as fh:
        head = fh.read(64)
    if head.startswith(_LFS_MAGIC):
        with open(path, "r", errors="replace") as fh:
            fields = dict(line.split(" ", 1) for line in fh.read().splitlines() if " " in line)
        raise ValueError(
            f"{path} is a git-LFS pointer ({os.path.getsize(path)} bytes of text, oid {fields.get('oid', '?')}, the real model is "
            f"{fields.get('size', '?')} bytes), not the model itself: the clone was made without git-lfs.  Fetch the model with "
            f"`git lfs install && git lfs pull --include '{path.rsplit('/', 1)[-1]}'` in the reference checkout (or download it "
            f"from the release the pointer belongs to) and pass that file to --network")
    if not head:
        raise ValueError(f"{path} is empty (0 bytes): an interrupted download or copy; fetch the model file again")
    if head.startswith(b"PK\x03\x04") or head.startswith(b"PK\x05\x06"):
        return "native" if _is_native(path) else "torch-zip"
    if head.startswith(b"\x1f\x8b"):
        return "gzip"
    if len(head) >= 2 and head[:1] == b"\x80" and 2 <= head[1] <= 5:
        return "torch-legacy"
    raise ValueError(f"{path} is neither a torch.save file (zip archive or legacy pickle stream) nor a native hello_amd .npz: it starts "
                     f"with {head[:16]!r}.  Expected the reference's *.wrapper.dnn (python/create_model_wrapper.py) or a file written "
                     f"by hello_amd.loader.save_native")


# --------------------------------------------------------------------------------------------
# reference pickles: inert stand-ins for the classes the pickle names
# --------------------------------------------------------------------------------------------
_NNTOOLS_CLASSES = ["Network", "ResidualBlock", "Noop", "Flatten", "GlobalPool", "Inception", "Pad1d",
                    "Compressor", "DotProduct", "ConcatenateChannels", "AdditiveLayer", "SelectArgument",
                    "Fork", "LinearCombination", "WeightNormedLinear", "WeightNormedConv1d",
                    "LayerNormModule", "Transposer"]
_MOE_CLASSES = ["MoEAttention", "MoEMergedAdvanced", "MoEMergedWrapperAdvanced", "ConvCombiner",
                "DummyGraphNetwork"]


@contextlib.contextmanager
def _stand_in_modules():
    import torch

    def make(modname, names):
        mod = types.ModuleType(modname)
        for n in names:
            setattr(mod, n, type(n, (torch.nn.Module,), {"__module__": modname}))
        return mod

    moe_modules = ("MixtureOfExpertsAdvanced", "MixtureOfExpertsAdvancedXferLearning")
    saved = {k: sys.modules.get(k) for k in ("NNTools",) + moe_modules}
    sys.modules["NNTools"] = make("NNTools", _NNTOOLS_CLASSES)
    for name in moe_modules:
        sys.modules[name] = make(name, _MOE_CLASSES)
    try:
        yield
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v


def _cls(m) -> str:
    return type(m).__name__


def _children(seq) -> List:
    return list(seq._modules.values())


def _conv_node(prefix, idx, layers, pos):
    """layers[pos] is a conv (weight-normed wrapper or plain Conv1d); consume an optional norm and
    activation that follow.  Returns (node, slots consumed)."""
    layer = layers[pos]
    wn = _cls(layer) == "WeightNormedConv1d"
    conv = layer._modules["conv1d"] if wn else layer
    key = f"{prefix}.{idx}.conv1d" if wn else f"{prefix}.{idx}"
    used, norm, bn_key, act, bn_eps = 1, ("wn" if wn else "none"), None, "none", 1e-5
    while pos + used < len(layers):
        nxt = _cls(layers[pos + used])
        if nxt == "BatchNorm1d" and norm != "bn" and act == "none":
            if wn:
                raise NotImplementedError(f"BatchNorm1d at {prefix}.{idx + used} (state-dict keys {prefix}.{idx + used}.weight / "
                                          f".running_mean ...) follows the weight-normed convolution {key}: folding both into one "
                                          f"weight is not implemented (no shipped configuration does this)")
            norm, bn_key = "bn", f"{prefix}.{idx + used}"
            bn_eps = float(layers[pos + used].eps)
        elif nxt == "LayerNormModule" and norm not in ("bn", "ln") and act == "none":
            if wn:
                raise NotImplementedError(f"LayerNormModule at {prefix}.{idx + used} follows the weight-normed convolution {key} "
                                          f"(state-dict keys {key}.weight_g / .weight_v): not implemented (the reference's "
                                          f"layer-norm configurations use plain convolutions, *_layer_norm.py)")
            norm, bn_key = "ln", f"{prefix}.{idx + used}"
            bn_eps = float(layers[pos + used]._modules["normer"].eps)
        elif nxt == "Noop" and act == "none":
            pass
        elif nxt == "ReLU" and act == "none":
            act = "relu"
        elif nxt == "Softplus" and act == "none":
            act = "softplus"
        else:
            break
        used += 1
        if act != "none":
            break
    (k,), (s,), (p,), (d,) = conv.kernel_size, conv.stride, conv.padding, conv.dilation
    if getattr(conv, "padding_mode", "zeros") != "zeros":
        raise NotImplementedError(f"convolution {key} (state-dict key {key}.{'weight_v' if wn else 'weight'}) pads with "
                                  f"{conv.padding_mode!r}: only zero padding is implemented")
    return ns.Conv(key, conv.in_channels, conv.out_channels, k, s, p, d, conv.groups, norm, bn_key, act, bn_eps), used


def _convert(network, prefix: str) -> List[ns.Node]:
    """A pickled ``NNTools.Network`` (its ``.network`` Sequential) -> node list.  A transfer-learning
    model holds ``Sequential(Network, Network)`` instead (MixtureOfExpertsAdvancedXferLearning.py:132-137)."""
    if _cls(network) == "Sequential":
        nodes: List[ns.Node] = []
        for name, child in network._modules.items():
            nodes += _convert(child, f"{prefix}.{name}")
        return nodes
    layers = _children(network._modules["network"])
    prefix = prefix + ".network"
    nodes: List[ns.Node] = []
    pos = 0
    while pos < len(layers):
        layer, name = layers[pos], _cls(layers[pos])
        if name in ("WeightNormedConv1d", "Conv1d"):
            node, used = _conv_node(prefix, pos, layers, pos)
            nodes.append(node)
            pos += used
        elif name == "MaxPool1d":
            k = layer.kernel_size if isinstance(layer.kernel_size, int) else layer.kernel_size[0]
            s = layer.stride if isinstance(layer.stride, int) else layer.stride[0]
            p = layer.padding if isinstance(layer.padding, int) else layer.padding[0]
            nodes.append(ns.MaxPool(k, s, p))
            pos += 1
        elif name == "ResidualBlock":
            body = _convert(layer._modules["ffNetwork"], f"{prefix}.{pos}.ffNetwork")
            shortcut = _convert(layer._modules["shNetwork"], f"{prefix}.{pos}.shNetwork")
            nodes.append(ns.Residual(body, shortcut))
            pos += 1
        elif name == "AdaptiveAvgPool1d":
            # terminus: AdaptiveAvgPool1d(1), Flatten, norm|Noop|Dropout, Linear
            if [_cls(x) for x in layers[pos:pos + 2]] != ["AdaptiveAvgPool1d", "Flatten"] or pos + 3 >= len(layers):
                raise NotImplementedError(f"pooling head at {prefix}.{pos}: expected AdaptiveAvgPool1d, Flatten, (norm | Noop | Dropout), "
                                          f"Linear (NNTools.py:517-566), found {[_cls(x) for x in layers[pos:pos + 4]]}")
            mid, lin = layers[pos + 2], layers[pos + 3]
            wn = _cls(lin) == "WeightNormedLinear"
            linear = lin._modules["linear"] if wn else lin
            key = f"{prefix}.{pos + 3}.linear" if wn else f"{prefix}.{pos + 3}"
            bn_eps = 1e-5
            if _cls(mid) == "BatchNorm1d":
                norm, bn_key, bn_eps = "bn", f"{prefix}.{pos + 2}", float(mid.eps)
                if wn:
                    raise NotImplementedError(f"BatchNorm1d at {prefix}.{pos + 2} before the weight-normed Linear {key} (state-dict keys "
                                              f"{key}.weight_g / .weight_v): not implemented")
            else:
                norm, bn_key = ("wn" if wn else "none"), None
            nodes.append(ns.Head(key, linear.in_features, linear.out_features, norm, bn_key, bn_eps))
            pos += 4
        elif name == "Fork":
            nets = [m for n, m in layer._modules.items() if n.startswith("net")]
            nxt = layers[pos + 1] if pos + 1 < len(layers) else None
            sel = _children(nets[1]._modules["network"])[0] if len(nets) == 2 else None
            if nxt is None or _cls(nxt) != "LinearCombination" or sel is None or _cls(sel) != "SelectArgument":
                raise NotImplementedError(f"Fork at {prefix}.{pos} with {len(nets)} branch(es) followed by "
                                          f"{_cls(nxt) if nxt is not None else 'nothing'}: only the xattn_subtract front end -- Fork(net0, "
                                          f"net1 = SelectArgument) + LinearCombination (architectures/xattn_subtract.py:14-42) -- is implemented")
            nodes.append(ns.Mix(tuple(float(c) for c in nxt.coefficients), int(sel.select)))
            pos += 2
        elif name == "SelectArgument":
            nodes.append(ns.Select(int(layer.select)))
            pos += 1
        elif name == "Transposer":
            nodes.append(ns.Transpose(int(layer.dim0), int(layer.dim1)))
            pos += 1
        elif name == "ConcatenateChannels":
            nodes.append(ns.Concat())
            pos += 1
        elif name in ("Noop", "Dropout"):
            pos += 1
        else:
            keys = [k for k, _ in getattr(layer, "named_parameters", lambda: [])()][:3]
            raise NotImplementedError(f"layer type {name!r} at {prefix}.{pos} is not supported"
                                      + (f" (state-dict keys {', '.join(f'{prefix}.{pos}.{k}' for k in keys)} ...)" if keys else "")
                                      + "; supported: Conv1d / WeightNormedConv1d (+ BatchNorm1d | LayerNormModule, ReLU | Softplus), "
                                        "MaxPool1d, ResidualBlock, the pooling head, Fork + LinearCombination, SelectArgument, Transposer, "
                                        "ConcatenateChannels, Noop, Dropout")
    return nodes


_MOE_ATTENTION_NETS = ["read_convolver0", "read_convolver1", "compressor0", "compressor1", "compressor2",
                       "xattn0", "xattn1", "xattn2", "combiner0", "combiner1", "meta"]


_MOE_MERGED_NETS = ["readConv0", "readConv1", "alleleConv0", "alleleConv1", "expert0", "expert1", "expert2", "meta",
                    "readConv0Meta", "readConv1Meta"]


def _spec_from_merged(moe) -> ns.ModelSpec:
    """Pickled ``MoEMergedAdvanced`` (MixtureOfExpertsAdvanced.py:255-331): the additive form (useAdditive=True),
    the class default (useAdditive=False: concatenated expert input, single technology -- the reference's own
    forward raises on a hybrid one, :436) and separate meta read convolvers (useSeparateMeta, :328-331)."""
    use_additive = bool(moe.__dict__.get("useAdditive", False))
    if not use_additive and moe._modules.get("readConv1") is not None:
        raise NotImplementedError(
            "hybrid MoEMergedAdvanced with useAdditive=False: the reference's own forward raises on it ('Boolean value "
            "of Tensor with more than one value is ambiguous', MixtureOfExpertsAdvanced.py:436), there is nothing to match")
    nets = {}
    for name in _MOE_MERGED_NETS:
        sub = moe._modules.get(name)
        if sub is not None:
            if _cls(sub) != "Network":
                raise NotImplementedError(f"moeMerged.{name} is a {_cls(sub)}, expected NNTools.Network (state-dict keys moeMerged.{name}.*)")
            nets[name] = _convert(sub, f"moeMerged.{name}")
    for name in ("alleleConvCombiner", "siteConvCombiner"):
        sub = moe._modules.get(name)
        if sub is not None:          # ConvCombiner: cat along channels, then its Network (:37-44)
            nets[name] = [ns.Concat()] + _convert(sub._modules["network"], f"moeMerged.{name}.network")
    first0 = next(ns.walk(nets["readConv0"]))
    c1 = next(ns.walk(nets["readConv1"])).cin if "readConv1" in nets else first0.cin
    return ns.ModelSpec(nets, name="reference_pickle", channels=(first0.cin, c1), family="merged",
                        use_additive=use_additive)


def spec_from_module(wrapper) -> Tuple[ns.ModelSpec, Dict[str, np.ndarray]]:
    """Pickled ``MoEMergedWrapperAdvanced`` (stand-in instance) -> (ModelSpec, state dict)."""
    moe = wrapper._modules.get("moeMerged")
    if moe is not None and _cls(moe) == "MoEMergedAdvanced":
        state = {k: v.detach().cpu().numpy() for k, v in wrapper.state_dict().items()}
        return _spec_from_merged(moe), state
    if moe is None or _cls(moe) != "MoEAttention":
        raise NotImplementedError(f"the pickled wrapper's .moeMerged is {_cls(moe) if moe is not None else 'missing'}: expected "
                                  f"MoEAttention or MoEMergedAdvanced (MixtureOfExpertsAdvanced.py:71,255); the pickle's top level is a "
                                  f"{_cls(wrapper)}")
    nets = {}
    for name in _MOE_ATTENTION_NETS:
        sub = moe._modules.get(name)
        if sub is not None:
            nets[name] = _convert(sub, f"moeMerged.{name}")
    first0 = next(ns.walk(nets["read_convolver0"]))
    c1 = next(ns.walk(nets["read_convolver1"])).cin if "read_convolver1" in nets else first0.cin
    spec = ns.ModelSpec(nets, name="reference_pickle", channels=(first0.cin, c1))
    state = {k: v.detach().cpu().numpy() for k, v in wrapper.state_dict().items()}
    return spec, state


def _load_reference_pickle(path: str, kind: str = "torch-zip"):
    import pickle
    import torch
    import warnings
    with _stand_in_modules(), warnings.catch_warnings():
        warnings.simplefilter("ignore")
        try:
            obj = torch.load(path, map_location="cpu", weights_only=False)
        except (pickle.UnpicklingError, AttributeError, ModuleNotFoundError, EOFError, RuntimeError) as exc:
            # a RuntimeError from torch.load is reported under its own type and text (a corrupt zip member, an unsupported storage ...):
            # the message below keeps both, so an unrelated torch failure is not mistaken for a class the stand-ins lack
            raise ValueError(f"{path} ({kind} stream) could not be unpickled as a reference model: {type(exc).__name__}: {exc}.  The loader supplies stand-ins "
                             f"for the classes of NNTools and MixtureOfExpertsAdvanced[XferLearning] only; a pickle naming other modules, "
                             f"or a truncated file, cannot be read") from exc
    if not hasattr(obj, "_modules"):
        raise ValueError(f"{path} holds a {type(obj).__name__}, not a module: the reference caller loads a whole "
                         f"MoEMergedWrapperAdvanced (python/create_model_wrapper.py:7-10), not a bare state dict -- wrap the weights "
                         f"with that script, or save them with hello_amd.loader.save_native(path, config, state)")
    return spec_from_module(obj)


def load_spec(path: str) -> Tuple[ns.ModelSpec, Dict[str, np.ndarray]]:
    """(ModelSpec, state dict) of a native file or a reference ``.wrapper.dnn`` pickle (zip form or the legacy stream of
    torch < 1.6).  A file that is neither -- most often the git-LFS pointer a plain clone of the reference leaves in models/ --
    is refused with a ValueError that says what it is and how to get the model (``sniff``)."""
    kind = sniff(path)
    if kind == "native":
        return _load_native(path)
    if kind == "gzip":
        raise ValueError(f"{path} is gzip-compressed: decompress it first (gunzip -k) and pass the .wrapper.dnn / .npz inside")
    return _load_reference_pickle(path, kind)


def load(path: str, device=0, shared: bool = False, **kw):
    """Drop-in for ``torch.load(path)`` in the reference caller: returns a network object with
    ``.eval()``, ``.providePredictions`` and ``__call__(featureDict, ref_segment)``.

    ``shared=True`` is for the reference's deployment form -- a pool of worker processes that each load the model and score one
    site per call (call.py:111,215-221): the object then holds no engine of its own; its calls go to ONE scoring server per
    (model file, GPU), started by the first worker, which coalesces the workers' concurrent sites into one launch
    (``hello_amd.shared``).  The calling process never touches the GPU.  ``device="auto"`` spreads a pool's workers over the node's GPUs
    (process id modulo the device count): one server per GPU."""
    if shared:
        from .shared import SharedScoringNetwork
        sniff(path)                                # a git-LFS pointer / an empty file is refused here, by name, not in the server's log
        return SharedScoringNetwork(path, device=device, **kw)
    from .wrapper import ScoringNetwork
    spec, state = load_spec(path)
    return ScoringNetwork(spec, state, device=device, **kw)
