"""Layer-level description of HELLO's mixture-of-experts scoring networks.

The reference describes every sub-network as a list of ``{"type", "kwargs"}`` dicts that
``NNTools.Network`` turns into a ``torch.nn.Sequential`` (reference python/NNTools.py:633-657).
This module is the build's own, framework-free statement of the same architectures: a sub-network is
a list of small dataclass nodes that carry (a) the arithmetic (channels, kernel, stride, padding,
normalisation kind, activation) and (b) the *state-dict key* under which the reference stores the
parameters, so that weights harvested from a reference ``.wrapper.dnn`` pickle, or generated
synthetically, address the same tensors.

Key bookkeeping mirrors how the reference lays layers out in its Sequential containers:
  * a conv "layer" occupies conv [+ norm] + activation slots (NNTools.py:72-115);
  * a residual block is ONE slot whose ``ffNetwork`` always has 6 slots -- conv, norm|Noop, act,
    conv, norm|Noop, act -- and whose ``shNetwork`` is a single Noop or a bias-carrying 1x1 conv
    (NNTools.py:118-294, 569-583);
  * the terminus is AdaptiveAvgPool1d, Flatten, norm|Noop, Linear (NNTools.py:517-566).

Only inference semantics are described (Dropout never appears with p > 0 in the shipped configs).
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple, Union


# --------------------------------------------------------------------------------------------
# nodes
# --------------------------------------------------------------------------------------------
@dataclass
class Conv:
    """Conv1d (+ folded normalisation) (+ activation).  ``key`` addresses the conv parameters:
    weight-normed convs store ``<key>.bias/.weight_g/.weight_v`` (NNTools.py:791-799), plain
    convs ``<key>.weight/.bias``; ``bn_key`` addresses a following BatchNorm1d, if any."""
    key: str
    cin: int
    cout: int
    k: int
    stride: int = 1
    pad: int = 0
    dilation: int = 1
    groups: int = 1
    norm: str = "wn"          # "wn" | "bn" | "ln" | "none"
    bn_key: Optional[str] = None   # key of the BatchNorm1d -- or, for "ln", of the LayerNormModule (its LayerNorm: <key>.normer)
    act: str = "relu"         # "relu" | "none" | "softplus"
    bn_eps: float = 1e-5      # eps of the following BatchNorm1d (torch default; a pickle may carry another)


@dataclass
class MaxPool:
    k: int
    stride: int
    pad: int = 0


@dataclass
class Residual:
    """out = body(x) + shortcut(x); the body ends in an activation, nothing follows the add
    (NNTools.py:582-583)."""
    body: List["Node"]
    shortcut: List["Node"]     # empty list = identity


@dataclass
class Head:
    """Mean over length, optional BatchNorm1d, Linear cin->cout (the reference's 'terminus')."""
    key: str
    cin: int
    cout: int
    norm: str = "wn"
    bn_key: Optional[str] = None
    bn_eps: float = 1e-5


@dataclass
class Mix:
    """Front-end of the 'xattn_subtract' expert: the network input is the tuple
    ``(allele_features, (site_frames0, site_frames1))``; the output is
    ``coeffs[0]*allele_features + coeffs[1]*site_frames[pick]`` (reference
    architectures/xattn_subtract.py:14-42: Fork(Noop, SelectArgument(1)) + LinearCombination([2,-1]))."""
    coeffs: Tuple[float, float] = (2.0, -1.0)
    pick: int = 1


@dataclass
class Select:
    """Pick one element of a tuple input (NNTools.py:745-751)."""
    index: int


@dataclass
class Transpose:
    """Swap (length, channels) of a [N, L, C] input so that convs see [N, C, L] (NNTools.py:831-838)."""
    dim0: int = 1
    dim1: int = 2


@dataclass
class Concat:
    """Concatenate a tuple of [N, C_i, L] tensors along channels (NNTools.py:727-733)."""


Node = Union[Conv, MaxPool, Residual, Head, Mix, Select, Transpose, Concat]


@dataclass
class ModelSpec:
    """A MoEAttention model (reference MixtureOfExpertsAdvanced.py:71-115,657-703): named
    sub-networks; a missing name means the reference attribute is None / deleted."""
    nets: Dict[str, List[Node]]
    name: str = ""
    window: int = 150
    channels: Tuple[int, int] = (6, 6)
    prefix: str = "moeMerged"
    family: str = "attention"        # "attention": MoEAttention (:71-252);  "merged": MoEMergedAdvanced (:255-484)
    # merged family only: expert input a - (s - a) (useAdditive=True) or cat(a, s - a) along channels (the class
    # default, MixtureOfExpertsAdvanced.py:270,372-383)
    use_additive: bool = True

    def has(self, net: str) -> bool:
        return net in self.nets and self.nets[net] is not None

    @property
    def hybrid_inputs(self) -> bool:
        """Two read technologies are convolved (read_convolver1 / readConv1 configured, :170-172, :401)."""
        return self.has("read_convolver1") or self.has("readConv1")

    @property
    def ensemble(self) -> bool:
        """The wrapper mixes three experts with meta weights (:509 'hybrid')."""
        return self.has("meta")


# --------------------------------------------------------------------------------------------
# sequential builder with reference-compatible slot numbering
# --------------------------------------------------------------------------------------------
class _Seq:
    def __init__(self, prefix: str, norm: str, act: str = "relu"):
        assert norm in ("wn", "bn", "ln", "none")
        self.prefix = prefix
        self.norm = norm
        self.act = act
        self.slot = 0
        self.nodes: List[Node] = []

    # -- helpers -------------------------------------------------------------------------------
    def _conv_node(self, base: str, idx: int, cin, cout, k, stride, pad, groups, act, norm=None):
        norm = self.norm if norm is None else norm
        if norm == "wn":
            return Conv(f"{base}.{idx}.conv1d", cin, cout, k, stride, pad, 1, groups, "wn", None, act)
        if norm in ("bn", "ln"):
            # plain conv followed by BatchNorm1d (folded at load) or LayerNormModule (NNTools.py:802-828: LayerNorm
            # over the channels of every position -- data dependent, so it stays a layer of its own)
            return Conv(f"{base}.{idx}", cin, cout, k, stride, pad, 1, groups, norm, f"{base}.{idx + 1}", act)
        return Conv(f"{base}.{idx}", cin, cout, k, stride, pad, 1, groups, "none", None, act)

    def skip(self, n: int = 1):
        self.slot += n
        return self

    def add(self, node: Node, slots: int = 1):
        self.nodes.append(node)
        self.slot += slots
        return self

    # -- layers --------------------------------------------------------------------------------
    def conv(self, cin, cout, k, pad=0, stride=1, groups=1):
        node = self._conv_node(self.prefix, self.slot, cin, cout, k, stride, pad, groups, self.act)
        # weight-normed conv + activation, or plain conv + (BatchNorm | Noop) + activation (NNTools.py:72-115)
        return self.add(node, 2 if self.norm == "wn" else 3)

    def maxpool(self, k, stride, pad=0):
        return self.add(MaxPool(k, stride, pad))

    def residual(self, cin, cout, stride=1, k=3, pad=1, groups=(1, 1, 1)):
        """stride == 1 and cin == cout -> identity shortcut; otherwise a strided 1x1 conv shortcut
        with bias and *no* padding, normalisation or activation."""
        base = f"{self.prefix}.{self.slot}"
        ff = f"{base}.ffNetwork.network"
        body = [
            self._conv_node(ff, 0, cin, cout, k, stride, pad, groups[0], self.act),
            self._conv_node(ff, 3, cout, cout, k, 1, pad, groups[1], self.act),
        ]
        if stride == 1 and cin == cout:
            shortcut: List[Node] = []
        else:
            sc_norm = "wn" if self.norm == "wn" else "none"
            shortcut = [self._conv_node(f"{base}.shNetwork.network", 0, cin, cout, 1, stride, 0,
                                        groups[2], "none", norm=sc_norm)]
        return self.add(Residual(body, shortcut))

    def head(self, cin, cout):
        """terminus (NNTools.py:517-566): weight-normed Linear, or BatchNorm1d + Linear -- the terminus keeps
        its default BatchNorm1d even in a network whose conv layers were generated with norm_type "Noop"."""
        lin = self.slot + 3
        if self.norm == "wn":
            node = Head(f"{self.prefix}.{lin}.linear", cin, cout, "wn", None)
        else:
            node = Head(f"{self.prefix}.{lin}", cin, cout, "bn", f"{self.prefix}.{self.slot + 2}")
        return self.add(node, 4)


# --------------------------------------------------------------------------------------------
# architectures (reference python/architectures/*.py); ``w`` scales channel widths (the *_wide files)
# --------------------------------------------------------------------------------------------
def read_convolver(prefix: str, norm="wn", in_channels=6, w=1, act="relu") -> List[Node]:
    """architectures/read_convolver.py:9-144 (6 in-channels), read_convolver_with_hp_channel.py
    (7 in-channels), read_convolver_wide.py (w=2).  150 -> 148 -> 146 -> 144 -> pool 71 -> 36."""
    s = _Seq(f"{prefix}.network", norm, act)
    s.conv(in_channels, 16 * w, 3).conv(16 * w, 16 * w, 3).conv(16 * w, 32 * w, 3)
    s.maxpool(3, 2, 0)
    for _ in range(3):
        s.residual(32 * w, 32 * w)
    s.residual(32 * w, 64 * w, stride=2)
    for _ in range(3):
        s.residual(64 * w, 64 * w)
    return s.nodes


def compressor(prefix: str, norm="wn", w=1, act="relu", blocks=2) -> List[Node]:
    """architectures/compressor_conv_small.py:8-55.  [64,36] -> [128,18].  ``blocks`` = identity-shortcut
    residual blocks after the strided one (3 in ExpertAlleleConvolver250FeatureMap.py)."""
    s = _Seq(f"{prefix}.network", norm, act)
    s.conv(64 * w, 64 * w, 1)
    s.residual(64 * w, 128 * w, stride=2)
    for _ in range(blocks):
        s.residual(128 * w, 128 * w)
    return s.nodes


def xattn_subtract(prefix: str, norm="wn", w=1, act="relu") -> List[Node]:
    """architectures/xattn_subtract.py:9-95.  (allele, (site0, site1)) -> 2*allele - site1 ->
    [256,9] -> logit."""
    s = _Seq(f"{prefix}.network", norm, act)
    s.add(Mix((2.0, -1.0), 1), slots=2)          # Fork + LinearCombination occupy two slots
    s.conv(128 * w, 128 * w, 1)
    s.residual(128 * w, 256 * w, stride=2)
    s.residual(256 * w, 256 * w)
    s.residual(256 * w, 256 * w)
    s.head(256 * w, 1)
    return s.nodes


def conv_combiner(prefix: str, norm="wn", w=1, act="relu") -> List[Node]:
    """architectures/conv_combiner.py:10-42.  cat([128,18],[128,18]) -> 512 (k3) -> 128 (k1)."""
    s = _Seq(f"{prefix}.network", norm, act)
    s.add(Concat())
    s.conv(256 * w, 512 * w, 3, pad=1)
    s.conv(512 * w, 128 * w, 1)
    return s.nodes


def meta_convolver(prefix: str, norm="wn", act="relu") -> List[Node]:
    """architectures/meta_convolver.py:10-77: site frames [128,18] -> 3 mixing logits."""
    s = _Seq(f"{prefix}.network", norm, act)
    s.add(Select(0))
    s.conv(128, 128, 1)
    s.residual(128, 256, stride=2)
    s.residual(256, 256)
    s.residual(256, 256)
    s.head(256, 3)
    return s.nodes


def meta_convolver_ref(prefix: str, norm="wn", act="relu") -> List[Node]:
    """architectures/meta_convolver_ref.py:14-106: one-hot reference segment [150,5] -> 3 logits."""
    s = _Seq(f"{prefix}.network", norm, act)
    s.add(Select(1))
    s.add(Transpose(1, 2))
    s.conv(5, 16, 1)
    s.residual(16, 32, stride=2)
    s.residual(32, 64, stride=2)
    s.residual(64, 128, stride=2)
    s.residual(128, 256, stride=2)
    s.head(256, 3)
    return s.nodes


# --------------------------------------------------------------------------------------------
# model configurations (reference python/moe_attention_config_*.py)
# --------------------------------------------------------------------------------------------
def _nets(prefix, table) -> Dict[str, List[Node]]:
    return {name: fn(f"{prefix}.{name}", **kw) for name, (fn, kw) in table.items()}


def single_tech(norm="wn", in_channels=6, prefix="moeMerged", act="relu") -> ModelSpec:
    """moe_attention_config_single_tech_old_equivalent_weight_norm.py:6-14 (and
    ..._with_hp_channel.py for in_channels=7).  ``act`` reaches the read convolver and the expert only:
    architectures/compressor_conv_small.py has no activation switch, so the compressor keeps ReLU even in
    the ..._layer_norm.py configuration that asks for Softplus."""
    nets = _nets(prefix, {
        "read_convolver0": (read_convolver, dict(norm=norm, in_channels=in_channels, act=act)),
        "compressor0": (compressor, dict(norm=norm)),
        "xattn0": (xattn_subtract, dict(norm=norm, act=act)),
    })
    name = "single_tech" + ("_hp" if in_channels == 7 else "")
    return ModelSpec(nets, name=name, channels=(in_channels, in_channels), prefix=prefix)


def hybrid_no_ensemble(norm="wn", prefix="moeMerged", w=1) -> ModelSpec:
    """moe_attention_config_full_hybrid_old_equivalent_weight_norm_no_ensemble.py:14-22 (w=1) and
    ..._no_ensemble_wide.py (w=2): two read convolvers + compressors, combiners, a single expert
    (xattn2); no xattn0/1, no meta."""
    nets = _nets(prefix, {
        "read_convolver0": (read_convolver, dict(norm=norm, w=w)),
        "read_convolver1": (read_convolver, dict(norm=norm, w=w)),
        "compressor0": (compressor, dict(norm=norm, w=w)),
        "compressor1": (compressor, dict(norm=norm, w=w)),
        "combiner0": (conv_combiner, dict(norm=norm, w=w)),
        "combiner1": (conv_combiner, dict(norm=norm, w=w)),
        "xattn2": (xattn_subtract, dict(norm=norm, w=w)),
    })
    return ModelSpec(nets, name="hybrid_no_ensemble" + ("_wide" if w == 2 else ""), prefix=prefix)


def hybrid_full(norm="wn", prefix="moeMerged") -> ModelSpec:
    """moe_attention_config_full_hybrid_old_equivalent_weight_norm.py: three experts + combiners +
    meta_convolver on the combined site frames."""
    nets = _nets(prefix, {
        "read_convolver0": (read_convolver, dict(norm=norm)),
        "read_convolver1": (read_convolver, dict(norm=norm)),
        "compressor0": (compressor, dict(norm=norm)),
        "compressor1": (compressor, dict(norm=norm)),
        "xattn0": (xattn_subtract, dict(norm=norm)),
        "xattn1": (xattn_subtract, dict(norm=norm)),
        "xattn2": (xattn_subtract, dict(norm=norm)),
        "combiner0": (conv_combiner, dict(norm=norm)),
        "combiner1": (conv_combiner, dict(norm=norm)),
        "meta": (meta_convolver, dict(norm=norm)),
    })
    return ModelSpec(nets, name="hybrid_full", prefix=prefix)


def hybrid_compressor2(norm="wn", prefix="moeMerged") -> ModelSpec:
    """The third way MoEAttention.forward makes hybrid features (MixtureOfExpertsAdvanced.py:181-192): a hybrid
    COMPRESSOR on the summed read frames of both technologies instead of the combiners -- ``compressor2`` + ``xattn2``;
    the meta-expert then reads that compressor's SITE-level output f2[0] (:192), the one place where the site-level
    compressor call of :136 is live.  No shipped configuration file selects it; the architecture modules are the
    shipped ones (the dict of ..._weight_norm.py with combiner0/1 replaced by compressor2)."""
    nets = _nets(prefix, {
        "read_convolver0": (read_convolver, dict(norm=norm)),
        "read_convolver1": (read_convolver, dict(norm=norm)),
        "compressor0": (compressor, dict(norm=norm)),
        "compressor1": (compressor, dict(norm=norm)),
        "compressor2": (compressor, dict(norm=norm)),
        "xattn0": (xattn_subtract, dict(norm=norm)),
        "xattn1": (xattn_subtract, dict(norm=norm)),
        "xattn2": (xattn_subtract, dict(norm=norm)),
        "meta": (meta_convolver, dict(norm=norm)),
    })
    return ModelSpec(nets, name="hybrid_compressor2", prefix=prefix)


def hybrid_ensemble2(norm="wn", prefix="moeMerged") -> ModelSpec:
    """moe_attention_config_full_hybrid_old_equivalent_weight_norm_ensemble2.py: two experts mixed
    by a meta-expert that reads the one-hot reference segment; third expert is all-zero logits."""
    nets = _nets(prefix, {
        "read_convolver0": (read_convolver, dict(norm=norm)),
        "read_convolver1": (read_convolver, dict(norm=norm)),
        "compressor0": (compressor, dict(norm=norm)),
        "compressor1": (compressor, dict(norm=norm)),
        "xattn0": (xattn_subtract, dict(norm=norm)),
        "xattn1": (xattn_subtract, dict(norm=norm)),
        "meta": (meta_convolver_ref, dict(norm=norm)),
    })
    return ModelSpec(nets, name="hybrid_ensemble2", prefix=prefix)


# --------------------------------------------------------------------------------------------
# transfer-learning "addendum" models (MixtureOfExpertsAdvancedXferLearning.py:71-181): every sub-network
# becomes Sequential(original Network, addendum Network) -- parameters under <net>.0.* and <net>.1.* -- and an
# expert first loses its terminus (everything after its last residual block), which the addendum re-adds.
# --------------------------------------------------------------------------------------------
def _addendum_blocks(prefix: str, norm: str, channels: int, head: Optional[int] = None) -> List[Node]:
    """architectures/read_convolver_addendum.py (64), compressor_conv_small_addendum.py (128),
    xattn_subtract_addendum.py (256 + terminus): two identity-shortcut residual blocks."""
    s = _Seq(f"{prefix}.network", norm)
    s.residual(channels, channels)
    s.residual(channels, channels)
    if head is not None:
        s.head(channels, head)
    return s.nodes


def _with_addendum(fn, channels, head=None):
    def build(prefix, norm="wn", **kw):
        base = fn(f"{prefix}.0", norm=norm, **kw)
        if head is not None:
            base = [n for n in base if not isinstance(n, Head)]
        return base + _addendum_blocks(f"{prefix}.1", norm, channels, head)
    return build


def single_tech_addendum(norm="wn", prefix="moeMerged") -> ModelSpec:
    """moe_attention_config_single_tech_old_equivalent_weight_norm_addendum.py on top of the single-tech model."""
    nets = _nets(prefix, {
        "read_convolver0": (_with_addendum(read_convolver, 64), dict(norm=norm)),
        "compressor0": (_with_addendum(compressor, 128), dict(norm=norm)),
        "xattn0": (_with_addendum(xattn_subtract, 256, head=1), dict(norm=norm)),
    })
    return ModelSpec(nets, name="single_tech_addendum", prefix=prefix)


def hybrid_no_ensemble_addendum(norm="wn", prefix="moeMerged") -> ModelSpec:
    """moe_attention_config_full_hybrid_old_equivalent_weight_norm_no_ensemble_addendum.py on top of the hybrid
    no-ensemble model (the combiners get no addendum)."""
    nets = _nets(prefix, {
        "read_convolver0": (_with_addendum(read_convolver, 64), dict(norm=norm)),
        "read_convolver1": (_with_addendum(read_convolver, 64), dict(norm=norm)),
        "compressor0": (_with_addendum(compressor, 128), dict(norm=norm)),
        "compressor1": (_with_addendum(compressor, 128), dict(norm=norm)),
        "combiner0": (conv_combiner, dict(norm=norm)),
        "combiner1": (conv_combiner, dict(norm=norm)),
        "xattn2": (_with_addendum(xattn_subtract, 256, head=1), dict(norm=norm)),
    })
    return ModelSpec(nets, name="hybrid_no_ensemble_addendum", prefix=prefix)


# --------------------------------------------------------------------------------------------
# the older family: MoEMergedAdvanced (reference MixtureOfExpertsAdvanced.py:255-484) built by
# createMoEFullMergedAdvancedModel (:614-654) from the "*Deeper" layer lists.  Same blocks, other names:
# readConv = MoEReadConvolverDeeper.py, alleleConv = ExpertAlleleConvolverDeeper.py,
# expert = ExpertGraphConvolverDeeper.py (no Mix node: the model forms a - (s - a) itself, :372-383),
# combiners = ConvCombinerResNetDeeper.py and meta = MetaCombinerDeeper.py (both always BatchNorm).
# --------------------------------------------------------------------------------------------
def graph_convolver(prefix: str, norm="wn") -> List[Node]:
    """ExpertGraphConvolverDeeper.py: [128,18] -> [256,9] -> logit."""
    s = _Seq(f"{prefix}.network", norm)
    s.conv(128, 128, 1)
    s.residual(128, 256, stride=2)
    s.residual(256, 256)
    s.residual(256, 256)
    s.head(256, 1)
    return s.nodes


def conv_combiner_deeper(prefix: str) -> List[Node]:
    """ConvCombinerResNetDeeper.py inside a ConvCombiner module (MixtureOfExpertsAdvanced.py:37-44):
    cat -> 256->512 k3 -> 512->128 k1, BatchNorm; parameters live under <prefix>.network.network."""
    s = _Seq(f"{prefix}.network.network", "bn")
    s.nodes.append(Concat())
    s.conv(256, 512, 3, pad=1)
    s.conv(512, 128, 1)
    return s.nodes


def conv_combiner_250(prefix: str) -> List[Node]:
    """ConvCombiner250FeatureMap.py inside a ConvCombiner module: cat -> grouped (2) k3 256->256 ->
    grouped strided residual block 256->512 -> 2 residual blocks 512 -> 1x1 512->128, BatchNorm."""
    s = _Seq(f"{prefix}.network.network", "bn")
    s.nodes.append(Concat())
    s.conv(256, 256, 3, pad=1, groups=2)
    s.residual(256, 512, stride=2, groups=(2, 2, 2))
    s.residual(512, 512)
    s.residual(512, 512)
    s.conv(512, 128, 1)
    return s.nodes


def meta_combiner_deeper(prefix: str) -> List[Node]:
    """MetaCombinerDeeper.py: site frames [128,18] -> 3 mixing logits, BatchNorm."""
    s = _Seq(f"{prefix}.network", "bn")
    s.conv(128, 128, 1)
    s.residual(128, 256, stride=2)
    s.residual(256, 256)
    s.residual(256, 256)
    s.head(256, 3)
    return s.nodes


def merged_single(norm="wn", prefix="moeMerged") -> ModelSpec:
    nets = _nets(prefix, {
        "readConv0": (read_convolver, dict(norm=norm)),
        "alleleConv0": (compressor, dict(norm=norm)),
        "expert0": (graph_convolver, dict(norm=norm)),
    })
    return ModelSpec(nets, name="merged_single", prefix=prefix, family="merged")


def merged_hybrid(norm="wn", prefix="moeMerged") -> ModelSpec:
    """Hybrid MoEMergedAdvanced with useAdditive=True and both combiners (the form
    MoEMergedConfig250FeatureMap.py:3-14 describes, at the shipped 150 bp window)."""
    nets = _nets(prefix, {
        "readConv0": (read_convolver, dict(norm=norm)),
        "readConv1": (read_convolver, dict(norm=norm)),
        "alleleConv0": (compressor, dict(norm=norm)),
        "alleleConv1": (compressor, dict(norm=norm)),
        "expert0": (graph_convolver, dict(norm=norm)),
        "expert1": (graph_convolver, dict(norm=norm)),
        "expert2": (graph_convolver, dict(norm=norm)),
        "alleleConvCombiner": (conv_combiner_deeper, dict()),
        "siteConvCombiner": (conv_combiner_deeper, dict()),
        "meta": (meta_combiner_deeper, dict()),
    })
    return ModelSpec(nets, name="merged_hybrid", prefix=prefix, family="merged")


def merged_hybrid_250(prefix="moeMerged") -> ModelSpec:
    """The 250 bp feature-map variant of MoEMergedAdvanced (MoEMergedConfig250FeatureMap.py:3-14 expressed
    in the current factory's keys): BatchNorm throughout, a three-block allele convolver, the grouped
    ConvCombiner250FeatureMap on allele features, no site-level combiner (site frames = sum of the combined
    allele features, MixtureOfExpertsAdvanced.py:434)."""
    nets = _nets(prefix, {
        "readConv0": (read_convolver, dict(norm="bn")),
        "readConv1": (read_convolver, dict(norm="bn")),
        "alleleConv0": (compressor, dict(norm="bn", blocks=3)),
        "alleleConv1": (compressor, dict(norm="bn", blocks=3)),
        "expert0": (graph_convolver, dict(norm="bn")),
        "expert1": (graph_convolver, dict(norm="bn")),
        "expert2": (graph_convolver, dict(norm="bn")),
        "alleleConvCombiner": (conv_combiner_250, dict()),
        "meta": (meta_combiner_deeper, dict()),
    })
    return ModelSpec(nets, name="merged_hybrid_250", window=250, prefix=prefix, family="merged")


CONFIGS = {
    "single_tech": lambda **kw: single_tech(**kw),
    "single_tech_hp": lambda **kw: single_tech(in_channels=7, **kw),
    "hybrid_no_ensemble": lambda **kw: hybrid_no_ensemble(**kw),
    "hybrid_no_ensemble_wide": lambda **kw: hybrid_no_ensemble(w=2, **kw),
    "hybrid_full": lambda **kw: hybrid_full(**kw),
    "hybrid_ensemble2": lambda **kw: hybrid_ensemble2(**kw),
    "hybrid_compressor2": lambda **kw: hybrid_compressor2(**kw),
    # moe_attention_config_single_tech_old_equivalent_layer_norm.py: plain convs, no normalisation, Softplus
    "single_tech_softplus": lambda **kw: single_tech(norm="none", act="softplus", **kw),
    # the same file with its commented-out line 14 active: norm_type = "LayerNormModule" (terminus stays BatchNorm)
    "single_tech_layernorm": lambda **kw: single_tech(norm="ln", act="softplus", **kw),
    "single_tech_addendum": lambda **kw: single_tech_addendum(**kw),
    "hybrid_no_ensemble_addendum": lambda **kw: hybrid_no_ensemble_addendum(**kw),
    "merged_single": lambda **kw: merged_single(**kw),
    "merged_hybrid": lambda **kw: merged_hybrid(**kw),
    "merged_hybrid_250": lambda **kw: merged_hybrid_250(**kw),
}

# name of the reference config module each spec corresponds to (used only by the fixture generator,
# which imports the reference in the build container)
REFERENCE_CONFIG_MODULE = {
    "single_tech": "moe_attention_config_single_tech_old_equivalent_weight_norm",
    "single_tech_hp": "moe_attention_config_single_tech_old_equivalent_weight_norm_with_hp_channel",
    "hybrid_no_ensemble": "moe_attention_config_full_hybrid_old_equivalent_weight_norm_no_ensemble",
    "hybrid_no_ensemble_wide": "moe_attention_config_full_hybrid_old_equivalent_weight_norm_no_ensemble_wide",
    "hybrid_full": "moe_attention_config_full_hybrid_old_equivalent_weight_norm",
    "hybrid_ensemble2": "moe_attention_config_full_hybrid_old_equivalent_weight_norm_ensemble2",
    "single_tech_softplus": "moe_attention_config_single_tech_old_equivalent_layer_norm",
    "single_tech_layernorm": "moe_attention_config_single_tech_old_equivalent_layer_norm",
    "single_tech_addendum": "moe_attention_config_single_tech_old_equivalent_weight_norm_addendum",
    "hybrid_no_ensemble_addendum": "moe_attention_config_full_hybrid_old_equivalent_weight_norm_no_ensemble_addendum",
}


def build(name: str, **kw) -> ModelSpec:
    return CONFIGS[name](**kw)


# --------------------------------------------------------------------------------------------
# traversal helpers
# --------------------------------------------------------------------------------------------
def walk(nodes: Sequence[Node]):
    """Yield every parametrised node (Conv / Head), depth first, in state-dict order."""
    for n in nodes:
        if isinstance(n, Residual):
            yield from walk(n.body)
            yield from walk(n.shortcut)
        elif isinstance(n, (Conv, Head)):
            yield n


def out_length(nodes: Sequence[Node], length: int) -> int:
    """Length after running ``nodes`` on an input of ``length`` positions."""
    for n in nodes:
        if isinstance(n, Conv):
            length = (length + 2 * n.pad - n.dilation * (n.k - 1) - 1) // n.stride + 1
        elif isinstance(n, MaxPool):
            length = (length + 2 * n.pad - n.k) // n.stride + 1
        elif isinstance(n, Residual):
            length = out_length(n.body, length)
        elif isinstance(n, Head):
            length = 1
    return length


def macs(nodes: Sequence[Node], length: int) -> int:
    """Multiply-accumulates for one item of ``length`` positions (SURVEY.md section 8d figures)."""
    total = 0
    for n in nodes:
        if isinstance(n, Conv):
            lo = out_length([n], length)
            total += lo * n.cout * (n.cin // n.groups) * n.k
            length = lo
        elif isinstance(n, MaxPool):
            length = out_length([n], length)
        elif isinstance(n, Residual):
            total += macs(n.body, length) + macs(n.shortcut, length)
            length = out_length(n.body, length)
        elif isinstance(n, Head):
            total += n.cin * n.cout
            length = 1
    return total
