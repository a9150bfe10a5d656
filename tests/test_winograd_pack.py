"""Host side of the Winograd F(2,3) form: the filter transform, the two weight layouts the kernels read, the
executed-MAC accounting bench.py reports."""
import numpy as np

from hello_amd import compiler, netspec as ns, readconv_pack as rp, weights


def _direct(x, w):                     # x [cin, L] zero padded by 1, w [cout, cin, 3] -> [cout, L]
    L = x.shape[1]
    xp = np.pad(x, ((0, 0), (1, 1)))
    return sum(w[:, :, t] @ xp[:, t:t + L] for t in range(3))


def test_filter_transform_reproduces_the_direct_convolution():
    rng = np.random.default_rng(0)
    w = rng.standard_normal((8, 16, 3)).astype(np.float32)
    x = rng.standard_normal((16, 18)).astype(np.float32)
    u = rp.winograd_taps(w).astype(np.float64)                      # [cout, cin, 4]
    xp = np.pad(x.astype(np.float64), ((0, 0), (1, 2)))
    out = np.empty((8, 18))
    for p in range(9):                                              # pairs of positions (2p, 2p+1)
        d = [xp[:, 2 * p + i] for i in range(4)]
        v = [d[0] - d[2], d[1] + d[2], d[2] - d[1], d[1] - d[3]]
        m = [u[:, :, c] @ v[c] for c in range(4)]
        out[:, 2 * p] = m[0] + m[1] + m[2]
        out[:, 2 * p + 1] = m[1] - m[2] - m[3]
    np.testing.assert_allclose(out, _direct(x.astype(np.float64), w.astype(np.float64)), rtol=1e-6, atol=1e-6)


def test_f33_filter_transform_reproduces_the_direct_convolution():
    """F(3,3) on the points 0, 1, -1, 2, inf as conv_wino.hip evaluates it (rows whose length is a multiple of 3)."""
    rng = np.random.default_rng(2)
    w = rng.standard_normal((8, 16, 3)).astype(np.float32)
    x = rng.standard_normal((16, 9)).astype(np.float32)
    u = compiler.winograd_taps_f33(w).astype(np.float64)            # [cout, cin, 5]
    xp = np.pad(x.astype(np.float64), ((0, 0), (1, 1)))
    out = np.empty((8, 9))
    for p in range(3):                                              # triples of positions (3p, 3p+1, 3p+2)
        d = [xp[:, 3 * p + i] for i in range(5)]
        s31 = d[3] - d[1]
        v = [2 * (d[0] - d[2]) + s31, s31 - (d[1] + d[2]), 3 * (d[1] - d[2]) + s31, s31, (d[4] - d[2]) - 2 * s31]
        m = [u[:, :, c] @ v[c] for c in range(5)]
        out[:, 3 * p] = m[0] + m[1] + m[2] + m[3]
        out[:, 3 * p + 1] = m[1] - m[2] + 2 * m[3]
        out[:, 3 * p + 2] = m[1] + m[2] + 4 * m[3] + m[4]
    np.testing.assert_allclose(out, _direct(x.astype(np.float64), w.astype(np.float64)), rtol=1e-6, atol=1e-6)


def test_fused_f33_weight_layout():
    """64-channel residual-block convs of the 150 bp kernel: [cout/16][cin/16][5 components][64 lanes][4]."""
    rng = np.random.default_rng(3)
    w = rng.standard_normal((64, 64, 3)).astype(np.float32)
    b = rng.standard_normal(64).astype(np.float32)
    blob = rp._pack_conv_f33(w, b)
    assert blob.size == 4 * 4 * 5 * 256 + 64 and np.array_equal(blob[-64:], b)
    u = compiler.winograd_taps_f33(w)
    body = blob[:-64].reshape(4, 4, 5, 64, 4)
    for cb, m, c, lane, t in [(0, 0, 0, 0, 0), (3, 2, 4, 63, 3), (1, 3, 1, 17, 2), (2, 0, 3, 40, 1)]:
        assert body[cb, m, c, lane, t] == u[16 * cb + (lane & 15), 16 * m + 4 * (lane >> 4) + t, c]
    assert rp.f33(True, 150) and not rp.f33(True, 250) and not rp.f33(False, 150)


def test_generic_winograd_weight_layout():
    rng = np.random.default_rng(1)
    w = rng.standard_normal((64, 24, 3)).astype(np.float32)
    b = rng.standard_normal(64).astype(np.float32)
    assert [compiler.winograd_outputs_per_tile(n) for n in (9, 18, 36, 150, 71, 121, 250)] == [3, 3, 3, 3, 2, 2, 2]
    for length, taps, u in ((71, 4, rp.winograd_taps(w)), (18, 5, compiler.winograd_taps_f33(w))):
        packed, bias = compiler.pack_conv_winograd(w, b, length)
        assert packed.shape == (64, taps * 24) and np.array_equal(bias, b)
        for o, c, comp in [(0, 0, 0), (5, 9, 2), (63, 23, taps - 1), (17, 16, 1)]:
            assert packed[o, (c // 8) * 8 * taps + comp * 8 + c % 8] == u[o, c, comp]     # [cin/8][component][8]


def test_fused_blob_sizes_and_flags():
    for cfg, extra in (("single_tech", 0), ("single_tech_addendum", 2)):
        spec = ns.build(cfg)
        state = weights.synth_state(spec, seed=1)
        for wino in (True, False):
            prog = compiler.compile_model(spec, state, winograd=wino)
            op = next(o for o in prog.ops if o.kind == compiler.OP_READCONV_FUSED)
            assert bool(op.flags & compiler.FLAG_WINOGRAD) == wino and op.k == extra and prog.winograd == wino
            kt = 4 if wino else 3
            kt64 = 5 if wino else 3                                 # identity-shortcut residual blocks: F(3,3) at 150 bp
            n64 = 6 + 2 * extra                                     # the blocks' convs (+ the strided block's second conv)
            want = (6 * (2 * kt64 * 2 * 256 + 32) + (6144 + 64) + (2048 + 64) + (4 * kt64 * 4 * 256 + 64)
                    + n64 * (4 * kt64 * 4 * 256 + 64) + (384 + 16) + (kt * 256 + 16) + (2 * kt * 256 + 32))
            nodes = spec.nets["read_convolver0"]
            assert rp.pack(nodes, weights.fold(spec, state), 6, winograd=wino).size == want
            convs = [o for o in prog.ops if o.kind == compiler.OP_CONV1D]
            assert any(o.flags & compiler.FLAG_WINOGRAD for o in convs) == wino
            for o in convs:                                         # only k3/s1/p1 convs with cout % 64 == 0 qualify
                if o.flags & compiler.FLAG_WINOGRAD:
                    assert (o.k, o.stride, o.pad) == (3, 1, 1) and o.cout % 64 == 0 and o.cin % 8 == 0
                    m = compiler.winograd_outputs_per_tile(o.lout)
                    assert o.exec_macs_per_row == -(-o.lout // m) * (m + 2) * o.cin * o.cout < o.macs_per_row


def test_executed_macs_match_the_kernel_schedule():
    assert rp.executed_macs_per_read(False) == 5052 * 1024          # MFMAs per wave and group of 4 reads (ISA count)
    # stem: conv1 60 + conv2 19 tiles x 16 / 4 waves = 76 (120 direct) + conv3/pool 5 tiles x 32 per read = 160 (264)
    # residual blocks in F(3,3) form: 3 tiles x 4 (2) input groups x 20 MFMAs = 240 (120) per wave (F(2,3): 320 / 144)
    assert rp.executed_macs_per_read(True) == (300 + 6 * 120 + 216 + 72 + 240 + 6 * 240) * 1024
    assert rp.executed_macs_per_read(True, 2) - rp.executed_macs_per_read(True) == 4 * 240 * 1024


def test_window_geometry_matches_the_reference_layer_arithmetic():
    """readconv_pack.geometry mirrors Cfg<G, NW, WINDOW> of readconv_fused.hip; the lengths are what the
    reference's layer lists produce (3 valid k3 convs, MaxPool(3,2), one stride-2 block)."""
    for window, want in ((150, (4, 71, 72, 36, 36, 11)), (250, (2, 121, 122, 61, 62, 18))):
        assert rp.geometry(window) == want
        nodes = ns.read_convolver("x")
        assert ns.out_length(nodes[:4], window) == want[1] and ns.out_length(nodes, window) == want[3]
    # executed MACs of the 250 bp kernel: fewer than the direct-form algorithmic count, more than 2/3 of it
    algo = ns.macs(ns.read_convolver("x"), 250)
    assert 0.67 * algo < rp.executed_macs_per_read(True, 0, 250) < algo


def test_sw3_swizzle_is_conflict_free_for_the_f33_operand_reads():
    """readconv_fused.hip's SW_3 image swizzle (64 channels: chunk ^ 2*((row/3)&7)): the five ds_read_b128 a lane
    issues per F(3,3) step -- rows 3j + i of its tile, chunk 4m + q, lane = 16 q + j -- must hit 64 distinct banks
    within each of the hardware's four 16-lane groups (MI355X_MICROARCH.md, LDS: bank = dword address mod 64)."""
    def img_off(row, chunk):
        return row * 64 + 4 * (chunk ^ (2 * ((row // 3) & 7)))

    groups = [list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28)),
              list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32)),
              list(range(32, 36)) + list(range(44, 48)) + list(range(52, 60)),
              list(range(36, 44)) + list(range(48, 52)) + list(range(60, 64))]
    for k in range(3):                       # tile
        for m in range(4):                   # input group
            for i in range(5):               # tap row
                for g in groups:
                    banks = set()
                    for lane in g:
                        j, q = lane & 15, lane >> 4
                        a = img_off(48 * k + 3 * j + i, 4 * m + q)
                        banks.update((a + d) % 64 for d in range(4))
                    assert len(banks) == 64, (k, m, i)
    # the stores (rows 3j + 1 + u of the wave's 16-channel block, 8 consecutive lanes per LDS cycle, 32 banks) are
    # at most 2-way: 16 LDS-array cycles against the 13 the instruction's data transfer takes anyway
    for u in range(3):
        for cb in range(4):
            for base in range(0, 64, 8):
                banks = set()
                for lane in range(base, base + 8):
                    j, q = lane & 15, lane >> 4
                    a = img_off(3 * j + 1 + u, 4 * cb + q)
                    banks.update((a + d) % 32 for d in range(4))          # stores bank on 32 dwords
                assert len(banks) >= 16, (u, cb, base)
