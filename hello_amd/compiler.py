"""Lower a MoEAttention model (netspec.ModelSpec + state dict) to the engine's flat program.

The reference evaluates the model by walking ``torch.nn.Sequential`` containers inside
``MoEAttention.forward`` (reference python/MixtureOfExpertsAdvanced.py:117-252).  The engine instead
receives ONE program: a list of ops over row-domain-typed activation buffers plus a single blob of
folded weights (include/hello_mi355x.h).  This module performs that lowering once at load time:

  * weight-norm / batch-norm folding (weights.fold);
  * residual blocks become convs whose epilogue adds the shortcut (ReLU happens before the add,
    NNTools.py:582-583);
  * ``reduceSlots`` + ``repeat_interleave`` + ``LinearCombination`` (MixtureOfExpertsAdvanced.py:
    142-155, xattn_subtract.py:14-42) become SEGSUM + MIX with an owner index, so the per-site sum is
    never re-expanded in memory;
  * the site-level compressor call of :136 is dropped when nothing consumes it (it is dead for every
    xattn_subtract configuration);
  * a read convolver of the canonical shape is replaced by the fused read-convolver op;
  * virtual activations are packed into as few physical scratch buffers as liveness allows.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import numpy as np

from . import netspec as ns
from . import weights as wts

# mirror of include/hello_mi355x.h
ROWS_READS0, ROWS_READS1, ROWS_ALLELES, ROWS_SITES = 0, 1, 2, 3
SEG_R0A, SEG_R1A, SEG_AS = 0, 1, 2
BUF_NONE, BUF_READS0, BUF_READS1, BUF_REF, BUF_FIRST_SCRATCH = -1, 0, 1, 2, 3
OP_XATTN_FRONT = 11
(OP_CONV1D, OP_MAXPOOL, OP_SEGSUM, OP_MIX, OP_HEAD, OP_CONCAT, OP_ADD, OP_READCONV_FUSED, OP_LAYERNORM,
 OP_COMPRESSOR_FUSED) = range(1, 11)
FLAG_RELU, FLAG_SRC_U8, FLAG_SOFTMAX, FLAG_MIX_REST, FLAG_SOFTPLUS, FLAG_WINOGRAD, FLAG_BF16X3, FLAG_BF16X3_32 = 1, 2, 4, 8, 16, 32, 64, 128
FLAG_LANE_SHIFT, MAX_LANES = 8, 8            # bits 8..10 of an op's flags: the stream it runs on in a laned program (HELLO_FLAG_LANE_*)
OP_NAMES = {1: "conv1d", 2: "maxpool", 3: "segsum", 4: "mix", 5: "head", 6: "concat", 7: "add",
            8: "readconv_fused", 9: "layernorm", 10: "compressor_fused", 11: "xattn_front"}


@dataclass
class Value:
    """A virtual activation tensor [rows(domain)][length][channels]."""
    vid: int
    domain: int
    length: int
    channels: int
    u8: bool = False

    @property
    def floats_per_row(self):
        return self.length * self.channels


@dataclass
class Op:
    kind: int
    domain: int
    src0: int = BUF_NONE      # virtual ids until allocate() rewrites them
    src1: int = BUF_NONE
    dst: int = BUF_NONE
    res: int = BUF_NONE
    cin: int = 0
    cout: int = 0
    k: int = 0
    stride: int = 1
    pad: int = 0
    lin: int = 0
    lout: int = 0
    flags: int = 0
    seg: int = 0
    c1: int = 0
    a0: float = 0.0
    a1: float = 0.0
    w_off: int = 0
    b_off: int = 0
    name: str = ""
    macs_per_row: int = 0            # algorithmic (direct-form) MACs per row of the op's domain
    exec_macs_per_row: float = 0.0   # MACs the matrix instructions execute per row (0: same as macs_per_row)


@dataclass
class Program:
    spec_name: str
    window: int
    channels0: int
    channels1: int
    n_experts: int
    has_meta: bool
    uses_ref: bool
    ops: List[Op]
    buffers: List[Tuple[int, int]]          # (domain, floats_per_row) per physical buffer id
    weights: np.ndarray                      # float32 blob
    fused_read_convolver: bool = False
    fused_compressor: bool = False
    winograd: bool = False           # k3/s1/p1 convolutions run in Winograd form (F(2,3) / F(3,3)) where a kernel offers it
    arithmetic: str = "fp32"         # "bf16x3": the read convolver's 64-channel trunk as 3-term bf16 splits; "bf16x3+32": the 32-channel
                                     # blocks too (selectable modes, never the default)
    n_lanes: int = 1                 # > 1: the program's independent chains carry lane numbers (small launches: assign_lanes)

    def describe(self) -> str:
        lines = [f"program {self.spec_name}: {len(self.ops)} ops, {len(self.buffers)} buffers, "
                 f"{self.weights.nbytes / 1e6:.2f} MB weights"]
        for i, o in enumerate(self.ops):
            lines.append(f"  {i:3d} {OP_NAMES[o.kind]:14s} dom={o.domain} {o.src0}->{o.dst} "
                         f"cin={o.cin} cout={o.cout} k={o.k} s={o.stride} L {o.lin}->{o.lout} {o.name}")
        return "\n".join(lines)


class _WeightBlob:
    def __init__(self):
        self.parts: List[np.ndarray] = []
        self.size = 0

    def add(self, arr: np.ndarray) -> int:
        arr = np.ascontiguousarray(arr, dtype=np.float32).ravel()
        pad = (-self.size) % 4
        if pad:
            self.parts.append(np.zeros(pad, np.float32))
            self.size += pad
        off = self.size
        self.parts.append(arr)
        self.size += arr.size
        return off

    def finish(self) -> np.ndarray:
        return np.concatenate(self.parts) if self.parts else np.zeros(0, np.float32)


def grouped_native(node) -> bool:
    """A grouped convolution the kernels run group by group (a workgroup's channel block inside one group, hello_op.c1):
    no multiplications by the zero blocks of the block-diagonal dense form."""
    return (node.groups > 1 and node.cin % node.groups == 0 and node.cout % node.groups == 0
            and (node.cout // node.groups) % 128 == 0 and (node.cin // node.groups) % 16 == 0)


def pack_conv(w: np.ndarray, b: np.ndarray, groups: int = 1, expand: bool = True):
    """[cout, cin/groups, k] -> dense [cout_pad32][kpad32] with K index = tap*cin + c (channels-last
    im2col order); grouped convs are expanded to block-diagonal dense weights unless ``expand`` is False (then a
    row holds its own group's k * cin/groups inputs only: the kernels' grouped form)."""
    cout, cg, k = w.shape
    cin = cg * groups if expand else cg
    dense = np.zeros((cout, cin, k), np.float32)
    og = cout // groups
    if expand:
        for g in range(groups):
            dense[g * og:(g + 1) * og, g * cg:(g + 1) * cg] = w[g * og:(g + 1) * og]
    else:
        dense[:] = w
    kreal = k * cin
    kpad = -(-kreal // 32) * 32
    cpad = -(-cout // 32) * 32
    packed = np.zeros((cpad, kpad), np.float32)
    packed[:cout, :kreal] = dense.transpose(0, 2, 1).reshape(cout, kreal)
    bias = np.zeros(cpad, np.float32)
    bias[:cout] = b
    return packed, bias


def winograd_outputs_per_tile(length: int) -> int:
    """F(3,3) (5 contractions per 3 positions) when the row length is a multiple of 3, else F(2,3) (4 per 2): the
    rule of ``conv1d_wino_outputs_per_tile`` in hello_amd/csrc/conv_wino.hip."""
    return 3 if length % 3 == 0 else 2


from .readconv_pack import winograd_taps_f33  # noqa: E402  (the F(3,3) filter transform; also used by the fused kernel's packing)


def pack_conv_winograd(w: np.ndarray, b: np.ndarray, length: int):
    """[cout, cin, 3] -> [cout][cin/8][T Winograd taps][8 channels] (hello_amd/csrc/conv_wino.hip), bias [cout];
    T = 5 for rows whose length is a multiple of 3, else 4."""
    from .readconv_pack import winograd_taps
    cout, cin, k = w.shape
    assert k == 3 and cin % 8 == 0
    u = winograd_taps_f33(w) if winograd_outputs_per_tile(length) == 3 else winograd_taps(w)   # [cout, cin, T]
    t = u.shape[-1]
    packed = u.reshape(cout, cin // 8, 8, t).transpose(0, 1, 3, 2).reshape(cout, t * cin)
    return np.ascontiguousarray(packed, dtype=np.float32), b.astype(np.float32)


# canonical read-convolver shape the fused kernel implements (architectures/read_convolver.py)
def _canonical_read_convolver_extras(nodes, cin, act="relu") -> int:
    """-1 if ``nodes`` is not the canonical read convolver (with activation ``act`` throughout); otherwise the
    number of extra identity-shortcut 64-channel residual blocks appended to it (transfer-learning models add
    2, read_convolver_addendum.py)."""
    try:
        ref = ns.read_convolver("x", in_channels=cin, act=act)
    except Exception:
        return -1

    def sig(n):
        if isinstance(n, ns.Conv):
            # a LayerNorm between convolutions is a layer of its own: such a read convolver is not the fused kernel's
            return ("c", n.cin, n.cout, n.k, n.stride, n.pad, n.groups, n.act, n.norm == "ln")
        if isinstance(n, ns.MaxPool):
            return ("p", n.k, n.stride, n.pad)
        if isinstance(n, ns.Residual):
            return ("r", tuple(sig(m) for m in n.body), tuple(sig(m) for m in n.shortcut))
        return ("?", type(n).__name__)

    if len(nodes) < len(ref) or cin not in (6, 7):
        return -1
    if not all(sig(a) == sig(b) for a, b in zip(nodes, ref)):
        return -1
    extras = nodes[len(ref):]
    if not all(sig(x) == sig(ref[-1]) for x in extras):
        return -1
    return len(extras)


def _is_canonical_read_convolver(nodes, cin) -> bool:
    return _canonical_read_convolver_extras(nodes, cin) == 0


ARITHMETICS = ("fp32", "bf16x3", "bf16x3+32")


class _Lowering:
    def __init__(self, spec: ns.ModelSpec, state, fused: bool, winograd: bool = True, arithmetic: str = "fp32"):
        if arithmetic not in ARITHMETICS:
            raise ValueError(f"arithmetic must be one of {ARITHMETICS}, not {arithmetic!r}")
        self.arithmetic = arithmetic
        self.used_bf16x3 = False
        self.used_xattn_front = False
        self.spec = spec
        self.state = state
        self.folded = wts.fold(spec, state)
        self.fused = fused
        self.winograd = bool(winograd)
        self.ops: List[Op] = []
        self.values: Dict[int, Value] = {}
        self.blob = _WeightBlob()
        self.next_vid = 1000        # virtual ids live above any physical id
        self.used_fused = False
        self.used_fused_compressor = False
        self.uses_ref = False

    # -- values ----------------------------------------------------------------------------
    def new(self, domain, length, channels) -> Value:
        v = Value(self.next_vid, domain, length, channels)
        self.values[v.vid] = v
        self.next_vid += 1
        return v

    def input(self, buf, domain, length, channels) -> Value:
        v = Value(buf, domain, length, channels, u8=True)
        self.values[buf] = v
        return v

    # -- single nodes ------------------------------------------------------------------------
    @staticmethod
    def _winograd_ok(node: ns.Conv, x: Value) -> bool:
        return (node.k == 3 and node.stride == 1 and node.pad == 1 and (node.groups == 1 or grouped_native(node))
                and node.dilation == 1 and not x.u8 and node.cin % 8 == 0 and node.cout % 64 == 0)

    def conv(self, node: ns.Conv, x: Value, res: Optional[Value] = None) -> Value:
        if node.act not in ("relu", "none", "softplus"):
            raise NotImplementedError(f"activation {node.act!r} is not implemented by the HIP engine")
        if node.dilation != 1:
            raise NotImplementedError("dilated convs are not implemented by the HIP engine")
        assert x.channels == node.cin, (node.key, x.channels, node.cin)
        w, b = self.folded[node.key]
        wino = self.winograd and self._winograd_ok(node, x)
        native = grouped_native(node) and not x.u8        # w is [cout, cin / groups, k]: each row its own group's inputs
        packed, bias = (pack_conv_winograd(w, b, x.length) if wino else pack_conv(w, b, node.groups, expand=not native))
        lout = ns.out_length([node], x.length)
        m = winograd_outputs_per_tile(lout)
        y = self.new(x.domain, lout, node.cout)
        act_flag = {"relu": FLAG_RELU, "softplus": FLAG_SOFTPLUS, "none": 0}[node.act]
        layer_norm = node.norm == "ln"          # conv + bias, then LayerNorm over channels, activation, residual
        self.ops.append(Op(
            OP_CONV1D, x.domain, src0=x.vid, dst=y.vid, res=res.vid if (res is not None and not layer_norm) else BUF_NONE,
            cin=node.cin, cout=node.cout, k=node.k, stride=node.stride, pad=node.pad,
            lin=x.length, lout=lout,
            flags=(0 if layer_norm else act_flag) | (FLAG_SRC_U8 if x.u8 else 0) | (FLAG_WINOGRAD if wino else 0),
            c1=node.groups if native else 0,
            w_off=self.blob.add(packed), b_off=self.blob.add(bias), name=node.key,
            macs_per_row=lout * node.cout * (node.cin // node.groups) * node.k,
            exec_macs_per_row=float(-(-lout // m) * (m + 2) * node.cout * (node.cin // node.groups)) if wino else 0.0))
        if layer_norm:
            gamma, beta, eps = wts.layer_norm_params(node, self.state)
            z = self.new(x.domain, lout, node.cout)
            self.ops.append(Op(OP_LAYERNORM, x.domain, src0=y.vid, dst=z.vid, res=res.vid if res is not None else BUF_NONE,
                               cin=node.cout, cout=node.cout, lin=lout, lout=lout, flags=act_flag, a0=float(eps),
                               w_off=self.blob.add(gamma), b_off=self.blob.add(beta), name=node.bn_key + ".normer"))
            return z
        return y

    def net(self, nodes, x, head_slot: Optional[int] = None, softmax=False):
        from . import readconv_pack
        if (self.fused is True and self.winograd and isinstance(x, Value) and not x.u8 and (x.length, x.channels) == (36, 64)
                and readconv_pack.compressor_blocks(nodes) in readconv_pack.COMPRESSOR_BLOCKS):
            # the canonical allele-level compressor: one LDS-resident kernel instead of 4 + 2 blocks launches
            blocks = readconv_pack.compressor_blocks(nodes)
            y = self.new(x.domain, 18, 128)
            w_off = self.blob.add(readconv_pack.pack_compressor(nodes, self.folded))
            self.ops.append(Op(OP_COMPRESSOR_FUSED, x.domain, src0=x.vid, dst=y.vid, cin=64, cout=128, k=blocks, lin=36, lout=18,
                               flags=FLAG_WINOGRAD | FLAG_RELU, w_off=w_off, b_off=w_off, name=nodes[0].key.rsplit(".network", 1)[0],
                               macs_per_row=ns.macs(nodes, 36), exec_macs_per_row=readconv_pack.compressor_executed_macs(blocks)))
            self.used_fused_compressor = True
            return y
        front = readconv_pack.xattn_front_match(nodes) if (self.fused is True and self.winograd and isinstance(x, tuple)) else None
        if front is not None:
            allele, sites = x
            site = sites[front[0].pick]
            if (site is not None and (allele.length, allele.channels) == (18, 128) and allele.domain == ROWS_ALLELES
                    and site.domain == ROWS_SITES):
                # the expert's front in ONE LDS-resident launch: MIX + 1x1 + the strided block's first convolution and its
                # shortcut (xattn_front_kernel); the block's second convolution follows with the shortcut as its residual
                mix, conv11, blk = front
                y2, sc = self.new(ROWS_ALLELES, 9, 256), self.new(ROWS_ALLELES, 9, 256)
                w_off = self.blob.add(readconv_pack.pack_xattn_front(conv11, blk, self.folded))
                self.ops.append(Op(OP_XATTN_FRONT, ROWS_ALLELES, src0=allele.vid, src1=site.vid, dst=y2.vid, res=sc.vid,
                                   cin=128, cout=256, k=3, stride=2, pad=1, lin=18, lout=9, flags=FLAG_RELU, seg=SEG_AS,
                                   a0=float(mix.coeffs[0]), a1=float(mix.coeffs[1]), w_off=w_off, b_off=w_off,
                                   name=conv11.key.rsplit(".network", 1)[0] + ".front",
                                   macs_per_row=ns.macs([conv11], 18) + ns.macs([blk.body[0]], 18) + ns.macs(blk.shortcut, 18),
                                   exec_macs_per_row=readconv_pack.xattn_front_executed_macs()))
                self.used_xattn_front = True
                x = self.conv(blk.body[1], y2, res=sc)
                nodes = nodes[3:]
        for node in nodes:
            if isinstance(node, ns.Conv):
                x = self.conv(node, x)
            elif isinstance(node, ns.MaxPool):
                lout = ns.out_length([node], x.length)
                y = self.new(x.domain, lout, x.channels)
                self.ops.append(Op(OP_MAXPOOL, x.domain, src0=x.vid, dst=y.vid, cin=x.channels,
                                   cout=x.channels, k=node.k, stride=node.stride, pad=node.pad,
                                   lin=x.length, lout=lout))
                x = y
            elif isinstance(node, ns.Residual):
                short = self.net(node.shortcut, x) if node.shortcut else x
                h = x
                for i, sub in enumerate(node.body):
                    last = i == len(node.body) - 1
                    if not isinstance(sub, ns.Conv):
                        raise NotImplementedError("residual bodies must be conv stacks")
                    h = self.conv(sub, h, res=short if last else None)
                x = h
            elif isinstance(node, ns.Head):
                w, b = self.folded[node.key]
                assert head_slot is not None
                self.ops.append(Op(OP_HEAD, x.domain, src0=x.vid, dst=head_slot, cin=x.channels,
                                   cout=node.cout, lin=x.length, lout=1,
                                   flags=FLAG_SOFTMAX if softmax else 0,
                                   w_off=self.blob.add(w), b_off=self.blob.add(b), name=node.key,
                                   macs_per_row=node.cin * node.cout))
                x = None
            elif isinstance(node, ns.Mix):
                allele, sites = x
                site = sites[node.pick]
                assert site is not None and site.domain == ROWS_SITES and allele.domain == ROWS_ALLELES
                y = self.new(ROWS_ALLELES, allele.length, allele.channels)
                self.ops.append(Op(OP_MIX, ROWS_ALLELES, src0=allele.vid, src1=site.vid, dst=y.vid,
                                   cin=allele.channels, lin=allele.length, lout=allele.length,
                                   a0=float(node.coeffs[0]), a1=float(node.coeffs[1]), seg=SEG_AS))
                x = y
            elif isinstance(node, ns.Select):
                x = x[node.index]
            elif isinstance(node, ns.Transpose):
                # the only transposed input is the one-hot reference [S, L, 5] -> [S, 5, L]
                # (meta_convolver_ref.py:27-36); channels-last storage already is that view
                if not (isinstance(x, Value) and x.vid == BUF_REF):
                    raise NotImplementedError("Transpose is only supported on the reference segment input")
            elif isinstance(node, ns.Concat):
                a, b = x
                assert a.domain == b.domain and a.length == b.length
                y = self.new(a.domain, a.length, a.channels + b.channels)
                self.ops.append(Op(OP_CONCAT, a.domain, src0=a.vid, src1=b.vid, dst=y.vid, cin=a.channels,
                                   c1=b.channels, lin=a.length, lout=a.length))
                x = y
            else:
                raise TypeError(node)
        return x

    def segsum(self, x: Value, seg: int) -> Value:
        domain = ROWS_SITES if seg == SEG_AS else ROWS_ALLELES
        y = self.new(domain, x.length, x.channels)
        self.ops.append(Op(OP_SEGSUM, domain, src0=x.vid, dst=y.vid, cin=x.channels, lin=x.length,
                           lout=x.length, seg=seg))
        return y

    def add(self, a: Value, b: Value) -> Value:
        y = self.new(a.domain, a.length, a.channels)
        self.ops.append(Op(OP_ADD, a.domain, src0=a.vid, src1=b.vid, dst=y.vid, cin=a.channels,
                           lin=a.length, lout=a.length))
        return y

    # -- model -------------------------------------------------------------------------------
    def read_frames(self, tech: int, stem: str = "read_convolver", suffix: str = "") -> Value:
        spec = self.spec
        name = f"{stem}{tech}{suffix}"
        nodes = spec.nets[name]
        cin = spec.channels[tech]
        buf = BUF_READS0 if tech == 0 else BUF_READS1
        dom = ROWS_READS0 if tech == 0 else ROWS_READS1
        seg = SEG_R0A if tech == 0 else SEG_R1A
        x = self.input(buf, dom, spec.window, cin)
        from . import readconv_pack
        extras = _canonical_read_convolver_extras(nodes, cin)
        softplus = False
        if extras < 0 and _canonical_read_convolver_extras(nodes, cin, "softplus") == 0:
            extras, softplus = 0, True            # moe_attention_config_single_tech_old_equivalent_layer_norm.py
        fusable = (self.fused and readconv_pack.AVAILABLE and extras in readconv_pack.EXTRA_BLOCKS
                   and spec.window in readconv_pack.WINDOWS)
        if fusable and (spec.window != 150 or softplus):
            # the 250 bp geometry and the Softplus activation exist as the whole kernel (stem included) in
            # Winograd form only
            fusable = self.fused is True and self.winograd and extras == 0 and not (softplus and spec.window != 150)
        wide_blocks = readconv_pack.wide_trunk_nodes(nodes, cin)
        if (self.fused in (True, "trunk") and self.winograd and spec.window == 150 and wide_blocks is not None
                and readconv_pack.AVAILABLE):
            # the 2x-channel read convolver: ONE kernel from the bytes (stem, residual trunk, segment sum), or with
            # fused="trunk" the stem layer by layer and the kernel entered at the pooled rows
            y = self.new(ROWS_ALLELES, 36, 128)
            w_off = self.blob.add(readconv_pack.pack_wide(nodes, wide_blocks, self.folded))
            if self.fused == "trunk":
                pooled = self.net(nodes[:readconv_pack.TRUNK_FIRST_NODE], x)
                assert (pooled.length, pooled.channels) == (71, 64)
                self.ops.append(Op(OP_READCONV_FUSED, ROWS_ALLELES, src0=pooled.vid, dst=y.vid, cin=64, cout=128, k=0, lin=71,
                                   lout=36, seg=seg, w_off=w_off, b_off=w_off, name=name + ".trunk", flags=FLAG_WINOGRAD,
                                   macs_per_row=ns.macs(wide_blocks, 71),
                                   exec_macs_per_row=readconv_pack.wide_trunk_executed_macs()))
            else:
                self.ops.append(Op(OP_READCONV_FUSED, ROWS_ALLELES, src0=buf, dst=y.vid, cin=cin, cout=128, k=0, lin=spec.window,
                                   lout=36, seg=seg, w_off=w_off, b_off=w_off, name=name, flags=FLAG_SRC_U8 | FLAG_WINOGRAD,
                                   macs_per_row=ns.macs(nodes, spec.window),
                                   exec_macs_per_row=readconv_pack.wide_executed_macs(cin)))
            self.used_fused = True
            return y
        if fusable:
            _, l1, _, l2, _, _ = readconv_pack.geometry(spec.window)
            y = self.new(ROWS_ALLELES, l2, 64)
            packed = readconv_pack.pack(nodes, self.folded, cin, winograd=self.winograd, window=spec.window)
            wflag = FLAG_WINOGRAD if self.winograd else 0
            # arithmetic mode bf16x3 (never the default): the 64 -> 64 trunk convolutions of the whole-kernel form on the
            # bf16 matrix cores as 3-term splits; their split weights ride behind the fp32 blob
            if (self.arithmetic != "fp32" and self.fused is True and self.winograd and spec.window == 150 and extras == 0
                    and not softplus):
                packed = np.concatenate([packed, readconv_pack.pack_bf16x3(nodes, self.folded)])
                wflag |= FLAG_BF16X3 | (FLAG_BF16X3_32 if "+32" in self.arithmetic else 0)
                self.used_bf16x3 = True
            w_off = self.blob.add(packed)
            if self.fused == "trunk":
                # stem layer by layer (3 valid convs + max pool), fused residual trunk + segment sum
                pooled = self.net(nodes[:readconv_pack.TRUNK_FIRST_NODE], x)
                assert (pooled.length, pooled.channels) == (l1, 32)
                self.ops.append(Op(OP_READCONV_FUSED, ROWS_ALLELES, src0=pooled.vid, dst=y.vid, cin=32, cout=64,
                                   k=extras, lin=l1, lout=l2, seg=seg, w_off=w_off, b_off=w_off, name=name + ".trunk",
                                   flags=wflag,
                                   macs_per_row=ns.macs(nodes[readconv_pack.TRUNK_FIRST_NODE:], l1)))
            else:
                # the whole read convolver (stem included) + segment sum in one kernel, straight from the bytes
                self.ops.append(Op(OP_READCONV_FUSED, ROWS_ALLELES, src0=buf, dst=y.vid, cin=cin, cout=64,
                                   k=extras, lin=spec.window, lout=l2, seg=seg, w_off=w_off, b_off=w_off, name=name,
                                   flags=FLAG_SRC_U8 | wflag | (FLAG_SOFTPLUS if softplus else 0),
                                   macs_per_row=ns.macs(nodes, spec.window),
                                   exec_macs_per_row=readconv_pack.executed_macs_per_read(self.winograd, extras,
                                                                                          spec.window)))
            self.used_fused = True
            return y
        return self.segsum(self.net(nodes, x), seg)

    def compress_and_predict(self, frames: Value, idx: int, slot: Optional[int]):
        """MixtureOfExpertsAdvanced.py:117-159."""
        spec = self.spec
        comp = spec.nets[f"compressor{idx}"]
        ca = self.net(comp, frames)
        xattn = f"xattn{idx}"
        needs_cs0 = spec.has(xattn) and any(isinstance(n, ns.Mix) and n.pick == 0 for n in spec.nets[xattn])
        cs0 = self.net(comp, self.segsum(frames, SEG_AS)) if needs_cs0 else None
        cs1 = self.segsum(ca, SEG_AS)
        if spec.has(xattn):
            self.net(spec.nets[xattn], (ca, (cs0, cs1)), head_slot=slot)
        return (cs0, cs1), ca

    def lower_merged(self):
        """MoEMergedAdvanced.forward (MixtureOfExpertsAdvanced.py:398-484): useAdditive=True, the class default
        (concatenated expert input; single technology only -- the reference raises on a hybrid one, :436) and
        separate meta read convolvers (:438-458)."""
        spec = self.spec
        hybrid = spec.has("readConv1")
        additive = spec.use_additive
        if hybrid and not additive:
            raise ValueError("hybrid MoEMergedAdvanced with useAdditive=False: the reference's own forward raises on it "
                             "(MixtureOfExpertsAdvanced.py:436)")

        def mix(allele: Value, site: Value, a0, a1, flags=0) -> Value:
            x = self.new(ROWS_ALLELES, allele.length, allele.channels)
            self.ops.append(Op(OP_MIX, ROWS_ALLELES, src0=allele.vid, src1=site.vid, dst=x.vid,
                               cin=allele.channels, lin=allele.length, lout=allele.length,
                               a0=a0, a1=a1, seg=SEG_AS, flags=flags))
            return x

        def expert(idx, allele: Value, site: Value):
            from . import readconv_pack
            nodes = spec.nets[f"expert{idx}"]
            front = (readconv_pack.xattn_front_match(nodes, mixed_ahead=True)
                     if (additive and self.fused is True and self.winograd
                         and (allele.length, allele.channels) == (18, 128)) else None)
            if front is not None:
                # the same fused front as MoEAttention's experts (xattn_front_kernel), with x = a - (s - a) formed in that order
                _, conv11, blk = front
                y2, sc = self.new(ROWS_ALLELES, 9, 256), self.new(ROWS_ALLELES, 9, 256)
                w_off = self.blob.add(readconv_pack.pack_xattn_front(conv11, blk, self.folded))
                self.ops.append(Op(OP_XATTN_FRONT, ROWS_ALLELES, src0=allele.vid, src1=site.vid, dst=y2.vid, res=sc.vid,
                                   cin=128, cout=256, k=3, stride=2, pad=1, lin=18, lout=9, flags=FLAG_RELU | FLAG_MIX_REST, seg=SEG_AS,
                                   a0=2.0, a1=-1.0, w_off=w_off, b_off=w_off, name=conv11.key.rsplit(".network", 1)[0] + ".front",
                                   macs_per_row=ns.macs([conv11], 18) + ns.macs([blk.body[0]], 18) + ns.macs(blk.shortcut, 18),
                                   exec_macs_per_row=readconv_pack.xattn_front_executed_macs()))
                self.used_xattn_front = True
                self.net(nodes[2:], self.conv(blk.body[1], y2, res=sc), head_slot=idx)
                return
            if additive:
                x = mix(allele, site, 2.0, -1.0, FLAG_MIX_REST)              # a - (s - a), in that rounding order
            else:
                rest = mix(allele, site, -1.0, 1.0)                          # s - a  (= -a + s exactly)
                x = self.new(ROWS_ALLELES, allele.length, 2 * allele.channels)
                self.ops.append(Op(OP_CONCAT, ROWS_ALLELES, src0=allele.vid, src1=rest.vid, dst=x.vid, cin=allele.channels,
                                   c1=allele.channels, lin=allele.length, lout=allele.length))      # cat((a, s - a), dim=1)
            self.net(spec.nets[f"expert{idx}"], x, head_slot=idx)

        a0 = self.net(spec.nets["alleleConv0"], self.read_frames(0, "readConv"))
        s0 = self.segsum(a0, SEG_AS)
        expert(0, a0, s0)
        if not hybrid:
            return 1, False
        for need in ("alleleConv1", "expert1", "expert2", "meta"):
            if not spec.has(need):
                raise ValueError(f"hybrid MoEMergedAdvanced needs {need}")
        a1 = self.net(spec.nets["alleleConv1"], self.read_frames(1, "readConv"))
        s1 = self.segsum(a1, SEG_AS)
        expert(1, a1, s1)
        if spec.has("alleleConvCombiner"):
            a2 = self.net(spec.nets["alleleConvCombiner"], (a0, a1))
        else:
            a2 = self.add(a0, a1)                                                     # :419
        if spec.has("siteConvCombiner"):
            s2 = self.net(spec.nets["siteConvCombiner"], (s0, s1))
        else:
            s2 = self.segsum(a2, SEG_AS)                                              # :434
        expert(2, a2, s2)
        site_meta = s2
        if spec.has("readConv0Meta"):
            # separate read convolvers for the meta-expert: each site's reads summed (reads -> alleles -> sites)
            if not spec.has("readConv1Meta"):
                raise ValueError("readConv0Meta without readConv1Meta")
            m0 = self.segsum(self.read_frames(0, "readConv", "Meta"), SEG_AS)
            m1 = self.segsum(self.read_frames(1, "readConv", "Meta"), SEG_AS)
            site_meta = self.net(spec.nets["siteConvCombiner"], (m0, m1)) if spec.has("siteConvCombiner") else self.add(m0, m1)
        self.net(spec.nets["meta"], site_meta, head_slot=3, softmax=True)
        return 3, True

    def lower(self):
        spec = self.spec
        if spec.family == "merged":
            return self.lower_merged()
        hybrid = spec.has("read_convolver1")
        has = [spec.has(f"xattn{i}") for i in range(3)]
        if not hybrid:
            n_experts, slots = 1, [0, None, None]
        elif not has[0] and not has[1]:
            if not has[2]:
                raise ValueError("no expert prediction is valid")    # reference assert, :239
            n_experts, slots = 1, [None, None, 0]
        else:
            if not (has[0] and has[1]):
                raise ValueError("hybrid data provided, but only single tech prediction is available")  # :243
            n_experts, slots = 3, [0, 1, 2]
        frames0 = self.read_frames(0)
        f0, ca0 = self.compress_and_predict(frames0, 0, slots[0])
        has_meta = False
        if hybrid:
            frames1 = self.read_frames(1)
            f1, ca1 = self.compress_and_predict(frames1, 1, slots[1])
            if spec.has("compressor2"):
                if not has[2]:
                    raise ValueError("xattn2 is needed with compressor2")                     # :184
                frames2 = self.add(frames0, frames1)
                # meta reads the site-level compressor output f2[0] (:192): force it alive
                comp2 = spec.nets["compressor2"]
                ca2 = self.net(comp2, frames2)
                cs0_2 = self.net(comp2, self.segsum(frames2, SEG_AS))
                cs1_2 = self.segsum(ca2, SEG_AS)
                self.net(spec.nets["xattn2"], (ca2, (cs0_2, cs1_2)), head_slot=slots[2])
                site_frames_for_meta = cs0_2
            elif has[2]:
                ca2 = self.net(spec.nets["combiner0"], (ca0, ca1))
                cs2 = self.net(spec.nets["combiner1"], (f0[1], f1[1]))
                self.net(spec.nets["xattn2"], (ca2, (None, cs2)), head_slot=slots[2])
                site_frames_for_meta = cs2
            else:
                site_frames_for_meta = None     # built lazily below: only the meta expert reads it
            if spec.has("meta"):
                has_meta = True
                meta_nodes = spec.nets["meta"]
                pick = next(n.index for n in meta_nodes if isinstance(n, ns.Select))
                ref = None
                if pick == 1:
                    self.uses_ref = True
                    ref = self.input(BUF_REF, ROWS_SITES, spec.window, 5)
                elif site_frames_for_meta is None:
                    site_frames_for_meta = self.segsum(self.add(frames0, frames1), SEG_AS)   # :224-227
                self.net(meta_nodes, (site_frames_for_meta, ref), head_slot=3, softmax=True)
        return n_experts, has_meta


def _fold_site_sums(ops: List[Op]) -> List[Op]:
    """The fused expert front forms a site's row itself when that row is nothing but the sum of the site's rows of the
    front's own allele input (reduceSlots, MixtureOfExpertsAdvanced.py:142-147) and nobody else reads it: the SEGSUM op goes,
    the front's src1 becomes BUF_NONE.  Same bits: the kernel adds the rows in allele order from zero, as segsum_kernel does."""
    drop = set()
    for o in ops:
        if o.kind != OP_XATTN_FRONT or o.src1 == BUF_NONE:
            continue
        makers = [j for j, m in enumerate(ops) if m.kind == OP_SEGSUM and m.dst == o.src1]
        readers = [r for r in ops if r is not o and o.src1 in (r.src0, r.src1, r.res)]
        if len(makers) == 1 and not readers and ops[makers[0]].seg == SEG_AS and ops[makers[0]].src0 == o.src0:
            drop.add(makers[0])
            o.src1 = BUF_NONE
    return [o for j, o in enumerate(ops) if j not in drop]


def _fold_concats(ops: List[Op]) -> List[Op]:
    """A channel-wise CONCAT whose only reader is a dense Winograd convolution with ReLU and no residual (the combiners' first
    layer, ConvCombiner: MixtureOfExpertsAdvanced.py:205-214) is never materialised: the convolution reads its two sources
    chunk by chunk (conv1d_wino_kernel's two-source form: src1 = the second tensor, seg = channels of the first).  Same bits:
    the K loop visits the same channels in the same order."""
    drop = set()
    for j, c in enumerate(ops):
        if c.kind != OP_CONCAT or c.cin % 16 or c.c1 % 16:
            continue
        readers = [r for r in ops if r is not c and c.dst in (r.src0, r.src1, r.res)]
        if len(readers) != 1:
            continue
        r = readers[0]
        if (r.kind == OP_CONV1D and r.src0 == c.dst and r.src1 == BUF_NONE and r.res == BUF_NONE and r.c1 <= 1
                and (r.flags & FLAG_WINOGRAD) and (r.flags & FLAG_RELU) and not (r.flags & (FLAG_BF16X3 | FLAG_SOFTPLUS))
                and r.cin == c.cin + c.c1):
            r.src0, r.src1, r.seg = c.src0, c.src1, c.cin
            drop.add(j)
    return [o for j, o in enumerate(ops) if j not in drop]


def assign_lanes(ops: List[Op], max_lanes: int = 4) -> int:
    """Lane numbers (bits 8..10 of ``flags``) for the ops of a program still in virtual buffer ids: an op continues the lane of
    a producer of one of its inputs while that producer is the LAST op of its lane so far, else it opens a new lane (or, with
    none left, queues behind its first input's producer).  The chains this finds in a MoEAttention forward
    (MixtureOfExpertsAdvanced.py:161-252): technology 0 (read convolver, compressor, its expert), technology 1, the combined
    expert behind combiner0, and combiner1 + the meta network.  Streams are in order, so any assignment is correct as long as the
    engine orders lanes with events where an op reads another lane's output -- which it derives from the buffer ids
    (hello_engine_create).  -> number of lanes used."""
    producer: Dict[int, int] = {}
    lane_of: List[int] = []
    tail: Dict[int, int] = {}
    for i, o in enumerate(ops):
        front = o.kind == OP_XATTN_FRONT                      # writes dst AND res; every other op reads res
        reads = [v for v in (o.src0, o.src1, BUF_NONE if front else o.res) if v >= 1000 and v in producer]
        pick = next((lane_of[producer[v]] for v in reads if tail.get(lane_of[producer[v]]) == producer[v]), None)
        if pick is None:
            unused = [lane for lane in range(max_lanes) if lane not in tail]
            pick = unused[0] if unused else (lane_of[producer[reads[0]]] if reads else 0)
        lane_of.append(pick)
        tail[pick] = i
        o.flags = (o.flags & ~(7 << FLAG_LANE_SHIFT)) | (pick << FLAG_LANE_SHIFT)
        if o.kind != OP_HEAD:
            producer[o.dst] = i
        if front:
            producer[o.res] = i
    return len(tail)


# rough device time of one op in a launch of a few sites (us; tools/one_site_profile.py): only their ORDER of magnitude matters below
_SMALL_LAUNCH_US = {OP_READCONV_FUSED: 75.0, OP_COMPRESSOR_FUSED: 55.0, OP_XATTN_FRONT: 27.0, OP_CONV1D: 10.0, OP_HEAD: 8.0, OP_SEGSUM: 7.0}


def schedule_lanes(ops: List[Op]) -> List[Op]:
    """Submission order of a laned program (virtual buffer ids).  The host submits the ~40 launches of a three-expert model one after
    the other (~5 us each: 220 us, about as long as the longest chain runs): what it submits first should be what the result waits
    for longest.  List scheduling by bottom level: among the ops whose producers and whose lane's previous op have been submitted,
    take the one with the longest remaining path to the end (its own estimated time + the longest chain of consumers / lane
    successors behind it).  Every such order is correct: producers precede consumers (a stream may only wait for an event that has
    been recorded) and a lane's ops keep their order (streams are first in, first out)."""
    shift = FLAG_LANE_SHIFT
    n = len(ops)
    producer: Dict[int, int] = {}
    preds: List[set] = [set() for _ in range(n)]
    last_on_lane: Dict[int, int] = {}
    for i, o in enumerate(ops):
        front = o.kind == OP_XATTN_FRONT
        for v in (o.src0, o.src1, BUF_NONE if front else o.res):
            if v >= 1000 and v in producer:
                preds[i].add(producer[v])
        lane = (o.flags >> shift) & 7
        if lane in last_on_lane:
            preds[i].add(last_on_lane[lane])
        last_on_lane[lane] = i
        if o.kind != OP_HEAD:
            producer[o.dst] = i
        if front:
            producer[o.res] = i
    succs: List[List[int]] = [[] for _ in range(n)]
    for i in range(n):
        for j in preds[i]:
            succs[j].append(i)
    bottom = [0.0] * n
    for i in reversed(range(n)):                       # program order is topological
        bottom[i] = _SMALL_LAUNCH_US.get(ops[i].kind, 6.0) + max([bottom[j] for j in succs[i]] + [0.0])
    done, order = set(), []
    while len(order) < n:
        ready = [i for i in range(n) if i not in done and preds[i] <= done]
        pick = max(ready, key=lambda i: (bottom[i], -i))
        done.add(pick)
        order.append(pick)
    return [ops[i] for i in order]


def _allocate(ops: List[Op], values: Dict[int, Value], reuse: bool = True):
    """Greedy liveness packing of virtual activations into physical scratch buffers, per domain.  ``reuse`` False: every value
    its own buffer (a laned program's ops run concurrently: liveness in program order says nothing there)."""
    last_use: Dict[int, int] = {}
    for i, o in enumerate(ops):
        for v in (o.src0, o.src1, o.res):
            if v >= 1000:
                last_use[v] = i        # (the fused expert front WRITES its `res` buffer: its readers come later and overwrite this)
    phys: List[Tuple[int, int]] = [(0, 0)] * BUF_FIRST_SCRATCH      # reserved ids
    free: Dict[int, List[int]] = {d: [] for d in range(4)}
    assigned: Dict[int, int] = {}
    def place(vid):
        v = values[vid]
        need = v.floats_per_row
        pool = free[v.domain]
        if pool:
            # best fit: smallest buffer that is large enough, else the largest one (it grows)
            fits = [p for p in pool if phys[p][1] >= need]
            pick = min(fits, key=lambda p: phys[p][1]) if fits else max(pool, key=lambda p: phys[p][1])
            pool.remove(pick)
            phys[pick] = (v.domain, max(phys[pick][1], need))
        else:
            pick = len(phys)
            phys.append((v.domain, need))
        assigned[vid] = pick

    for i, o in enumerate(ops):
        if o.kind == OP_XATTN_FRONT:          # two outputs: dst (the strided convolution) and res (its shortcut)
            place(o.dst)
            place(o.res)
            for vsrc in {o.src0, o.src1}:
                if reuse and vsrc >= 1000 and last_use.get(vsrc) == i and vsrc in assigned:
                    free[values[vsrc].domain].append(assigned[vsrc])
            continue
        if o.kind != OP_HEAD:
            place(o.dst)
        # release inputs whose last use is this op (after the output was placed: no aliasing)
        for vsrc in {o.src0, o.src1, o.res}:
            if reuse and vsrc >= 1000 and last_use.get(vsrc) == i and vsrc in assigned:
                free[values[vsrc].domain].append(assigned[vsrc])
        # an output nobody reads (cannot happen in a well-formed program) would leak; ignore
    for o in ops:
        o.src0 = assigned.get(o.src0, o.src0)
        o.src1 = assigned.get(o.src1, o.src1)
        o.res = assigned.get(o.res, o.res)
        if o.kind != OP_HEAD:
            o.dst = assigned[o.dst]
    return phys


def compile_model(spec: ns.ModelSpec, state, fused: bool = True, winograd: bool = True, arithmetic: str = "fp32",
                  fold_site_sums: bool = True, lanes: bool = False) -> Program:
    """``winograd``: k3/s1/p1 convolutions are evaluated in Winograd form -- F(3,3) (5 instead of 9 contractions per
    3 positions) where the row length / the fused kernel's geometry is whole triples, else F(2,3) (4 instead of 6
    per pair) -- same fp32 arithmetic, results differ from the direct form by float re-association only.
    ``fold_site_sums``: glue ops folded into their consumers -- the expert front sums a site's alleles itself, a combiner's first
    convolution reads the two tensors of its CONCAT directly (one launch and one buffer less each; the same bits); False keeps
    the SEGSUM / CONCAT ops (tests compare the two).
    ``lanes``: the program for SMALL launches -- the same ops, but the independent chains of a two-technology / three-expert model
    carry lane numbers (``assign_lanes``) and no two values share a buffer, so that the engine may run the chains concurrently
    (a launch of a few sites is latency-bound: every chain is a handful of workgroups).  ``n_lanes`` == 1 for single-chain models."""
    low = _Lowering(spec, state, fused, winograd, arithmetic)
    n_experts, has_meta = low.lower()
    if arithmetic != "fp32" and not low.used_bf16x3:
        raise ValueError("arithmetic='bf16x3' / 'bf16x3+32' exists for the canonical 150 bp ReLU read convolver in the whole-kernel "
                         "Winograd form (fused=True, winograd=True): this model / these options do not run it")
    if fold_site_sums:
        low.ops = _fold_site_sums(low.ops)
        low.ops = _fold_concats(low.ops)
    n_lanes = assign_lanes(low.ops) if lanes else 1
    if n_lanes > 1:
        low.ops = schedule_lanes(low.ops)
    buffers = _allocate(low.ops, low.values, reuse=n_lanes == 1)
    return Program(
        spec_name=spec.name, window=spec.window, channels0=spec.channels[0],
        channels1=spec.channels[1] if spec.hybrid_inputs else 0,
        n_experts=n_experts, has_meta=has_meta, uses_ref=low.uses_ref, ops=low.ops,
        buffers=buffers, weights=low.blob.finish(), fused_read_convolver=low.used_fused,
        fused_compressor=low.used_fused_compressor, winograd=low.winograd, arithmetic=arithmetic, n_lanes=n_lanes)
