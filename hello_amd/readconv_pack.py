"""Weight packing for the fused read-convolver trunk kernel (hello_amd/csrc/readconv_fused.hip).

The kernel covers the residual part of the canonical read convolver (reference
architectures/read_convolver.py:58-143): 3 x ResidualBlock(32), the strided 32->64 block with its 1x1
shortcut, 3 x ResidualBlock(64).  Each wave owns one 16-channel output block and keeps that slice of a
layer in registers as the A operand of v_mfma_f32_16x16x4_f32, so the blob is laid out in exactly the
order the lanes load it:

    per conv:  [cout/16 blocks][k taps][cin/16 groups][64 lanes][4 floats], then bias[cout]
    lane l, float t  =  W[out = 16*block + (l & 15)][in = 16*group + 4*(l >> 4) + t][tap]
"""
import numpy as np

from . import netspec as ns

AVAILABLE = True
TRUNK_FIRST_NODE = 4          # nodes[0:4] = the stem (3 valid convs + max pool), run layer by layer
EXTRA_BLOCKS = (0, 2)         # identity 64-channel blocks the kernel accepts after the canonical three


def _pack_conv(w: np.ndarray, b: np.ndarray) -> np.ndarray:
    cout, cin, k = w.shape
    lanes = np.arange(64)
    out_idx = (np.arange(cout // 16)[:, None] * 16 + (lanes & 15)[None, :])                    # [cb, lane]
    in_idx = (np.arange(cin // 16)[:, None, None] * 16 + 4 * (lanes >> 4)[None, :, None]
              + np.arange(4)[None, None, :])                                                  # [m, lane, t]
    # result[cb, tap, m, lane, t]
    packed = w[out_idx[:, None, None, :, None], in_idx[None, None, :, :, :], np.arange(k)[None, :, None, None, None]]
    return np.concatenate([packed.astype(np.float32).ravel(), b.astype(np.float32).ravel()])


def trunk_convs(nodes):
    """The 15 convolutions of the canonical trunk in kernel order."""
    blocks = nodes[TRUNK_FIRST_NODE:TRUNK_FIRST_NODE + 7]
    assert len(blocks) == 7 and all(isinstance(b, ns.Residual) for b in blocks)
    order = []
    for blk in blocks[:3]:
        order += [blk.body[0], blk.body[1]]
    strided = blocks[3]
    order += [strided.body[0], strided.shortcut[0], strided.body[1]]
    for blk in blocks[4:]:
        order += [blk.body[0], blk.body[1]]
    return order


def _pack_first_conv(w: np.ndarray, b: np.ndarray, steps: int = 6) -> np.ndarray:
    """conv1 of the stem reads the pileup bytes directly: k = tap*C + c is the byte offset from the row
    start, walked 4 per MFMA step; lane l holds W[out = l & 15][k = 4*step + (l >> 4)] (zero past 3*C)."""
    cout, cin, k = w.shape
    assert cout == 16 and k == 3 and 3 * cin <= 4 * steps
    flat = np.zeros((cout, 4 * steps), np.float32)
    flat[:, :3 * cin] = w.transpose(0, 2, 1).reshape(cout, 3 * cin)          # [out][tap*C + c]
    lanes = np.arange(64)
    packed = flat[(lanes & 15)[None, :], 4 * np.arange(steps)[:, None] + (lanes >> 4)[None, :]]   # [step][lane]
    return np.concatenate([packed.ravel(), b.astype(np.float32).ravel()])


WINDOWS = (150, 250)          # pileup windows the fused kernel is instantiated for (250: Winograd form, whole kernel)


def geometry(window: int):
    """(reads per group, L1, RS1, L2, RS2, pool tiles per read) of readconv_fused.hip's Cfg for ``window``."""
    l1 = (window - 9) // 2 + 1
    l2 = (l1 - 1) // 2 + 1
    return (4 if window == 150 else 2), l1, l1 + 1, l2, (l2 if l2 % 2 == 0 else l2 + 1), (l1 + 6) // 7


def executed_macs_per_read(winograd: bool, extra_blocks: int = 0, window: int = 150) -> float:
    """MACs the fused kernel's MFMA instructions really execute per read (tile padding included), from its
    tile schedule in readconv_fused.hip: v_mfma_f32_16x16x4_f32 = 16*16*4 MACs.  Direct form: 5 052 MFMAs per
    wave and group of 4 reads; Winograd form: 3 948 on average (the residual blocks need 4 instead of 6
    contractions per pair of positions, minus what their half-empty last tile gives back)."""
    g, l1, rs1, _, rs2, ntt = geometry(window)
    t_stem12 = -(-(-(-window * g // 16)) // 4) * 4               # 16-row tiles of conv1 / conv2, 4 position groups
    # conv3 + pool: `ntt` stride-14 tiles of 24 MFMAs per read (2 blocks), or ceil(L1/15) Winograd tiles of 32
    conv3 = g * (-(-l1 // 15)) * 2 * 4 * 4 if winograd else g * ntt * 2 * 3 * 4
    # conv2: tiles of 16 pairs (16 MFMAs each) dealt to the 4 waves in whole rounds, or 3 taps x 4 per 16-row tile
    conv2 = -(-(-(-(window * g // 2) // 16)) // 4) * 4 * 16 if winograd else t_stem12 * 3 * 4
    stem = t_stem12 * 6 + conv2 + conv3
    t1, t2 = -(-rs1 * g // 16), -(-rs2 * g // 16)                # direct tiles at 32 / 64 channels
    n64 = 6 + 2 * extra_blocks
    if winograd:
        w1, w2 = -(-(rs1 // 2) * g // 16), -(-(rs2 // 2) * g // 16)   # tiles of 16 pairs
        strided = t2 * 4 * 6 * 4 + 2 * w2 * 4 * 2 * 4 + w2 * 4 * 4 * 16
        if f33(winograd, window):                    # shortcut on 3 x 3 tiles of 16 rows, second conv in F(3,3) form
            strided = t2 * 4 * 6 * 4 + 3 * (rs2 * g // 48) * 4 * 2 * 4 + (rs2 * g // 48) * 4 * 4 * 20
        # 64-channel blocks: F(3,3) where the image is whole tiles of 16 triples (20 MFMAs per tile, input group and wave)
        per64 = (rs2 * g // 48) * 4 * 4 * 20 if f33(winograd, window) else w2 * 4 * 4 * 16
        per32 = (rs1 * g // 48) * 2 * 2 * 20 if f33(winograd, window) else w1 * 2 * 2 * 16
        blocks = 6 * per32 + n64 * per64
    else:
        strided = t2 * 4 * 6 * 4 + t2 * 4 * 2 * 4 + t2 * 4 * 12 * 4
        blocks = 6 * (t1 * 2 * 6 * 4) + n64 * (t2 * 4 * 12 * 4)
    return (stem + strided + blocks) * 1024.0 / g


def winograd_taps(w: np.ndarray) -> np.ndarray:
    """[cout, cin, 3] -> [cout, cin, 4]: the F(2,3) filter transform U = G g, evaluated in float64 and rounded
    once: g0, (g0 + g1 + g2)/2, (g0 - g1 + g2)/2, g2."""
    g = w.astype(np.float64)
    u = np.stack([g[..., 0], (g[..., 0] + g[..., 1] + g[..., 2]) / 2, (g[..., 0] - g[..., 1] + g[..., 2]) / 2,
                  g[..., 2]], axis=-1)
    return u.astype(np.float32)


def winograd_taps_f33(w: np.ndarray) -> np.ndarray:
    """[cout, cin, 3] -> [cout, cin, 5]: the F(3,3) filter transform for the points 0, 1, -1, 2, inf, evaluated in
    float64 and rounded once: g0/2, -(g0+g1+g2)/2, (-g0+g1-g2)/6, (g0+2g1+4g2)/6, g2."""
    g = w.astype(np.float64)
    g0, g1, g2 = g[..., 0], g[..., 1], g[..., 2]
    u = np.stack([g0 / 2, -(g0 + g1 + g2) / 2, (-g0 + g1 - g2) / 6, (g0 + 2 * g1 + 4 * g2) / 6, g2], axis=-1)
    return u.astype(np.float32)


def f33(winograd: bool, window: int) -> bool:
    """The identity-shortcut residual blocks run in Winograd F(3,3) form: the 150 bp geometry, whose images (4 reads
    x 72 rows at 32 channels, 4 x 36 at 64) are whole tiles of 16 triples (``Cfg::F33`` / ``F33_32`` in
    readconv_fused.hip)."""
    return bool(winograd) and window == 150


def _pack_conv_f33(w: np.ndarray, b: np.ndarray) -> np.ndarray:
    """[cout/16 blocks][cin/16 groups][5 components][64 lanes][4]: wino3_layer walks the input groups outermost."""
    u = winograd_taps_f33(w)
    cout, cin, _ = u.shape
    n = (cout // 16) * 5 * (cin // 16) * 256
    blocks = _pack_conv(u, b)[:n].reshape(cout // 16, 5, cin // 16, 64, 4)            # [cb, c, m, lane, t]
    return np.concatenate([blocks.transpose(0, 2, 1, 3, 4).ravel(), b.astype(np.float32).ravel()])


def pack(nodes, folded, cin=None, winograd: bool = False, window: int = 150) -> np.ndarray:
    """trunk block followed by the stem block (conv1 bytes form, conv2, conv3), then any extra 64-channel
    blocks.  With ``winograd`` the two convolutions of every identity-shortcut residual block are stored as
    their four Winograd F(2,3) taps -- the 64-channel ones as five F(3,3) taps where ``f33(winograd, window)`` --
    (the strided block's first conv and shortcut stay in direct form)."""
    convs = trunk_convs(nodes)
    strided = {6, 7}                                 # kernel order: 6 x (32->32), strided a / shortcut / b, 6 x (64->64);
                                                     # the strided block's second conv (b) is k3/s1/p1 too
    use33 = f33(winograd, window)

    def one(i, c):
        w, b = folded[c.key]
        if use33 and i not in strided:               # every k3/s1 convolution of the trunk, 32 and 64 channels
            return _pack_conv_f33(w, b)
        return _pack_conv(winograd_taps(w), b) if (winograd and i not in strided) else _pack_conv(w, b)

    parts = [one(i, c) for i, c in enumerate(convs)]
    stem = nodes[:3]
    parts.append(_pack_first_conv(*folded[stem[0].key]))
    w2, b2 = folded[stem[1].key]                        # conv2 too
    parts.append(_pack_conv(winograd_taps(w2), b2) if winograd else _pack_conv(w2, b2))
    w3, b3 = folded[stem[2].key]                        # conv3 + pool runs in Winograd form with the rest
    parts.append(_pack_conv(winograd_taps(w3), b3) if winograd else _pack_conv(w3, b3))
    extras = nodes[TRUNK_FIRST_NODE + 7:]               # transfer-learning blocks follow the canonical blob
    assert len(extras) in EXTRA_BLOCKS
    for blk in extras:
        parts += [one(9, blk.body[0]), one(9, blk.body[1])]
    blob = np.concatenate(parts)
    kt = 4 if winograd else 3
    w32, w64 = 2 * kt * 2 * 256, 4 * kt * 4 * 256
    w64d = 4 * 5 * 4 * 256 if use33 else w64
    w32d = 2 * 5 * 2 * 256 if use33 else w32
    trunk = 6 * (w32d + 32) + (6144 + 64) + (2048 + 64) + (w64d + 64) + 6 * (w64d + 64)
    assert blob.size == trunk + (384 + 16) + (kt * 256 + 16) + (2 * kt * 256 + 32) + 2 * len(extras) * (w64d + 64), blob.size
    return blob


def xattn_front_match(nodes, mixed_ahead: bool = False):
    """(mix, conv 1x1, strided residual block) when ``nodes`` opens like the canonical allele-level expert
    (architectures/xattn_subtract.py:9-60): LinearCombination of the allele's and its site's frames, Conv 1x1 128->128 + ReLU,
    then a residual block whose body starts k3 s2 p1 128->256 + ReLU and whose shortcut is a 1x1 s2 128->256 convolution
    without activation; else None.  What xattn_front_kernel (readconv_fused.hip) fuses."""
    def conv_is(n, cin, cout, k, stride, pad, act):
        return (isinstance(n, ns.Conv) and (n.cin, n.cout, n.k, n.stride, n.pad, n.dilation, n.groups, n.act) ==
                (cin, cout, k, stride, pad, 1, 1, act) and n.norm != "ln")
    if mixed_ahead:          # MoEMergedAdvanced: the caller forms a - (s - a) itself, the expert opens with the 1x1
        nodes = [None] + list(nodes)
    if len(nodes) < 3 or not (mixed_ahead or isinstance(nodes[0], ns.Mix)) or not conv_is(nodes[1], 128, 128, 1, 1, 0, "relu"):
        return None
    blk = nodes[2]
    if not (isinstance(blk, ns.Residual) and len(blk.body) == 2 and len(blk.shortcut) == 1
            and conv_is(blk.body[0], 128, 256, 3, 2, 1, "relu") and conv_is(blk.shortcut[0], 128, 256, 1, 2, 0, "none")
            and isinstance(blk.body[1], ns.Conv)):
        return None
    return nodes[0], nodes[1], blk


def pack_xattn_front(conv11, blk, folded) -> np.ndarray:
    """Weights of xattn_front_kernel: the 1x1, the strided convolution and its shortcut, each in conv_layer's order
    [cout/16][taps][cin/16][64 lanes][4] followed by its bias."""
    blob = np.concatenate([_pack_conv(*folded[conv11.key]), _pack_conv(*folded[blk.body[0].key]),
                           _pack_conv(*folded[blk.shortcut[0].key])])
    assert blob.size == (8 * 8 * 256 + 128) + (16 * 3 * 8 * 256 + 256) + (16 * 8 * 256 + 256), blob.size
    return blob


def xattn_front_executed_macs() -> float:
    """MACs the kernel's MFMAs execute per item: 9 + 2 x 5 x (24 + 8) tiles-steps of 4 MFMAs (16 x 16 x 4) per wave, 8 waves,
    8 items (the 72 output rows of a workgroup fill 4.5 tiles: the fifth is half empty)."""
    return (9 * 8 + 2 * 5 * (24 + 8)) * 4 * 8 * 1024.0 / 8


def to_bf16_bits(x: np.ndarray) -> np.ndarray:
    """float32 -> bf16 bit patterns (uint16), round to nearest even (what v_cvt_pk_bf16_f32 does for finite values)."""
    u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32)
    return ((u + np.uint32(0x7FFF) + ((u >> np.uint32(16)) & np.uint32(1))) >> np.uint32(16)).astype(np.uint16)


def pack_bf16x3(nodes, folded) -> np.ndarray:
    """Arithmetic mode "bf16x3": the split weights of the 64 -> 64 trunk convolutions (the strided block's second conv,
    then the residual blocks' convs, in kernel order) and of the six 32 -> 32 ones for bf16x3_layer / bf16x3_layer32 in
    readconv_fused.hip, appended behind the fp32 blob
    (whose biases these layers keep using): w = hi + lo with hi = bf16(w), lo = bf16(w - hi);
        [layer][4 channel blocks][6 steps s = 2 tap + h][hi | lo][64 lanes][8]
        lane l, element i  =  W[out = 16 block + (l & 15)][in = 32 h + 8 (l >> 4) + i][tap]
    returned as float32 words holding the uint16 pairs."""
    convs = trunk_convs(nodes)
    layers = [convs[8]] + convs[9:] + [c for blk in nodes[TRUNK_FIRST_NODE + 7:] for c in (blk.body[0], blk.body[1])]
    lanes = np.arange(64)
    out_idx = np.arange(4)[:, None] * 16 + (lanes & 15)[None, :]                                    # [cb, lane]
    in_idx = 32 * np.arange(2)[:, None, None] + 8 * (lanes >> 4)[None, :, None] + np.arange(8)[None, None, :]   # [h, lane, i]
    parts = []
    for c in layers:
        w, _ = folded[c.key]
        assert w.shape == (64, 64, 3), w.shape
        w = w.astype(np.float32)
        hi = to_bf16_bits(w)
        lo = to_bf16_bits(w - (hi.astype(np.uint32) << np.uint32(16)).view(np.float32))
        for cb in range(4):
            for tap in range(3):
                for h in range(2):
                    for part in (hi, lo):
                        parts.append(part[out_idx[cb][:, None], in_idx[h], tap].ravel())            # [lane][i]
    # the six 32 -> 32 convolutions of the ResidualBlock(32)s: one 32-deep chunk per tap
    #     [layer][2 channel blocks][3 taps][hi | lo][64 lanes][8],  lane l, element i = W[16 block + (l & 15)][8 (l >> 4) + i][tap]
    out32 = np.arange(2)[:, None] * 16 + (lanes & 15)[None, :]
    in32 = 8 * (lanes >> 4)[:, None] + np.arange(8)[None, :]                                         # [lane, i]
    for c in convs[:6]:
        w, _ = folded[c.key]
        assert w.shape == (32, 32, 3), w.shape
        w = w.astype(np.float32)
        hi = to_bf16_bits(w)
        lo = to_bf16_bits(w - (hi.astype(np.uint32) << np.uint32(16)).view(np.float32))
        for cb in range(2):
            for tap in range(3):
                for part in (hi, lo):
                    parts.append(part[out32[cb][:, None], in32, tap].ravel())
    bits = np.concatenate(parts).astype(np.uint16)
    assert bits.size == len(layers) * 24576 + 6 * 6144
    return bits.view(np.float32).copy()


# ------------------------------------------------------------------------------------------------
# fused allele-level compressor (compressor_kernel in readconv_fused.hip)
# ------------------------------------------------------------------------------------------------
COMPRESSOR_BLOCKS = (2, 3, 4) # identity residual blocks the kernel is instantiated for (architectures/compressor_conv_small.py
                              # has 2, ExpertAlleleConvolver250FeatureMap.py 3, the transfer-learning addendum appends 2 to 2)


def compressor_blocks(nodes) -> int:
    """-1 unless ``nodes`` is the canonical compressor -- Conv 1x1 64->64 + ReLU, the strided residual block 64->128
    (k3 s2 p1 + ReLU, k3 s1 p1 + ReLU, 1x1 s2 shortcut), then identity residual blocks at 128 channels, no LayerNorm --
    otherwise the number of identity blocks."""
    def conv_is(n, cin, cout, k, stride, pad, act):
        return (isinstance(n, ns.Conv) and (n.cin, n.cout, n.k, n.stride, n.pad, n.dilation, n.groups, n.act) ==
                (cin, cout, k, stride, pad, 1, 1, act) and n.norm != "ln")
    if len(nodes) < 3 or not conv_is(nodes[0], 64, 64, 1, 1, 0, "relu"):
        return -1
    st = nodes[1]
    if not (isinstance(st, ns.Residual) and len(st.body) == 2 and len(st.shortcut) == 1 and
            conv_is(st.body[0], 64, 128, 3, 2, 1, "relu") and conv_is(st.body[1], 128, 128, 3, 1, 1, "relu") and
            conv_is(st.shortcut[0], 64, 128, 1, 2, 0, "none")):
        return -1
    for blk in nodes[2:]:
        if not (isinstance(blk, ns.Residual) and len(blk.body) == 2 and not blk.shortcut and
                all(conv_is(c, 128, 128, 3, 1, 1, "relu") for c in blk.body)):
            return -1
    return len(nodes) - 2


def pack_compressor(nodes, folded) -> np.ndarray:
    """1x1 conv, strided conv, shortcut in the direct layout of ``_pack_conv``; the strided block's second conv and the
    identity blocks' convs in the F(3,3) layout of ``_pack_conv_f33``; each followed by its bias (cc::Cfg offsets)."""
    assert compressor_blocks(nodes) in COMPRESSOR_BLOCKS
    st = nodes[1]
    parts = [_pack_conv(*folded[nodes[0].key]), _pack_conv(*folded[st.body[0].key]), _pack_conv(*folded[st.shortcut[0].key]),
             _pack_conv_f33(*folded[st.body[1].key])]
    for blk in nodes[2:]:
        parts += [_pack_conv_f33(*folded[blk.body[0].key]), _pack_conv_f33(*folded[blk.body[1].key])]
    blob = np.concatenate(parts)
    n = len(nodes) - 2
    assert blob.size == (4096 + 64) + (24576 + 128) + (8192 + 128) + (1 + 2 * n) * (81920 + 128), blob.size
    return blob


def compressor_executed_macs(blocks: int) -> float:
    """MACs the kernel's MFMAs execute per item: direct 1x1 / strided / shortcut convs, F(3,3) (5 contractions per 3
    positions) for the 1 + 2 blocks k3/s1 convolutions."""
    return 36 * 64 * 64 + 18 * 128 * 192 + 18 * 128 * 64 + (1 + 2 * blocks) * 6 * 5 * 128 * 128


# ---- the 2x-channel ("_wide") read convolver: residual trunk kernel (readconv_wide_trunk_kernel) ---------------------
def wide_trunk_nodes(nodes, cin):
    """The 7 residual blocks after the stem if ``nodes`` is the 2x-channel read convolver
    (architectures/read_convolver_wide.py: 6|7 -> 32 -> 32 -> 64, max pool, 3 x ResidualBlock(64), strided 64 -> 128,
    3 x ResidualBlock(128), ReLU, no LayerNorm), else None."""
    def conv_is(n, ci, co, k, stride, pad, act):
        return (isinstance(n, ns.Conv) and (n.cin, n.cout, n.k, n.stride, n.pad, n.dilation, n.groups, n.act) ==
                (ci, co, k, stride, pad, 1, 1, act) and n.norm != "ln")
    if len(nodes) != TRUNK_FIRST_NODE + 7 or cin not in (6, 7):
        return None
    stem = nodes[:TRUNK_FIRST_NODE]
    if not (conv_is(stem[0], cin, 32, 3, 1, 0, "relu") and conv_is(stem[1], 32, 32, 3, 1, 0, "relu") and
            conv_is(stem[2], 32, 64, 3, 1, 0, "relu") and isinstance(stem[3], ns.MaxPool) and
            (stem[3].k, stem[3].stride, stem[3].pad) == (3, 2, 0)):
        return None
    blocks = nodes[TRUNK_FIRST_NODE:]

    def identity(blk, c):
        return (isinstance(blk, ns.Residual) and len(blk.body) == 2 and not blk.shortcut and
                all(conv_is(x, c, c, 3, 1, 1, "relu") for x in blk.body))
    st = blocks[3]
    if not (all(identity(b, 64) for b in blocks[:3]) and all(identity(b, 128) for b in blocks[4:]) and
            isinstance(st, ns.Residual) and len(st.body) == 2 and len(st.shortcut) == 1 and
            conv_is(st.body[0], 64, 128, 3, 2, 1, "relu") and conv_is(st.body[1], 128, 128, 3, 1, 1, "relu") and
            conv_is(st.shortcut[0], 64, 128, 1, 2, 0, "none")):
        return None
    return blocks


def _pack_first_conv_wide(w: np.ndarray, b: np.ndarray, steps: int = 6) -> np.ndarray:
    """``_pack_first_conv`` for 32 output channels: [2 blocks][steps][64 lanes], then bias[32]."""
    cout, cin, k = w.shape
    assert cout == 32 and k == 3 and 3 * cin <= 4 * steps
    flat = np.zeros((cout, 4 * steps), np.float32)
    flat[:, :3 * cin] = w.transpose(0, 2, 1).reshape(cout, 3 * cin)
    lanes = np.arange(64)
    packed = flat[(16 * np.arange(2)[:, None, None] + (lanes & 15)[None, None, :]),
                  (4 * np.arange(steps)[None, :, None] + (lanes >> 4)[None, None, :])]           # [blk][step][lane]
    return np.concatenate([packed.ravel(), b.astype(np.float32).ravel()])


def pack_wide(nodes, blocks, folded) -> np.ndarray:
    """Trunk block (``pack_wide_trunk``) followed by the stem block: conv1 in the bytes form, conv2 and conv3 as their
    four Winograd F(2,3) taps (wt::Cfg::OFF_S1 / OFF_S2 / OFF_S3)."""
    stem = nodes[:3]
    parts = [pack_wide_trunk(blocks, folded), _pack_first_conv_wide(*folded[stem[0].key])]
    for conv in stem[1:]:
        w, b = folded[conv.key]
        parts.append(_pack_conv(winograd_taps(w), b))
    blob = np.concatenate(parts)
    assert blob.size == parts[0].size + (768 + 32) + (4096 + 32) + (8192 + 64), blob.size
    return blob


def wide_executed_macs(cin: int) -> float:
    """MACs the whole wide kernel's MFMAs execute per read: ``wide_trunk_executed_macs`` + the stem (conv1 40 tiles of
    16 rows x 24 k x 32 channels per 4 reads; conv2 19 tiles of 16 pairs, conv3 5 tiles of 16 pairs per read, 4
    contractions per pair)."""
    return wide_trunk_executed_macs() + 40 * 16 * 24 * 32 / 4 + 19 * 16 * 4 * 32 * 32 / 4 + 5 * 16 * 4 * 32 * 64


def pack_wide_trunk(blocks, folded) -> np.ndarray:
    """wt::Cfg offsets: 6 convs 64->64 in the F(3,3) layout, the strided conv and its shortcut in the direct layout, the
    strided block's second conv and 6 convs 128->128 in the F(3,3) layout; each followed by its bias."""
    parts = []
    for blk in blocks[:3]:
        parts += [_pack_conv_f33(*folded[blk.body[0].key]), _pack_conv_f33(*folded[blk.body[1].key])]
    st = blocks[3]
    parts += [_pack_conv(*folded[st.body[0].key]), _pack_conv(*folded[st.shortcut[0].key]), _pack_conv_f33(*folded[st.body[1].key])]
    for blk in blocks[4:]:
        parts += [_pack_conv_f33(*folded[blk.body[0].key]), _pack_conv_f33(*folded[blk.body[1].key])]
    blob = np.concatenate(parts)
    assert blob.size == 6 * (20480 + 64) + (24576 + 128) + (8192 + 128) + 7 * (81920 + 128), blob.size
    return blob


def wide_trunk_executed_macs() -> float:
    """MACs the trunk kernel's MFMAs execute per read: F(3,3) (5 contractions per 3 positions; 24 triples per read at 64
    channels incl. the shared zero row, 12 at 128), direct strided conv and shortcut."""
    return 6 * 24 * 5 * 64 * 64 + 36 * 128 * 192 + 36 * 128 * 64 + 7 * 12 * 5 * 128 * 128
