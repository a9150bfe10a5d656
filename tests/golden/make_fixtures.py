"""Generate golden vectors by running the REFERENCE itself (build container only).

Imports /root/reference/python/{NNTools,MixtureOfExpertsAdvanced}.py, instantiates the reference
model for each configuration, pushes this package's seeded synthetic state dict into it with
``load_state_dict`` and records the reference's outputs on seeded synthetic sites.  Only data is
written: inputs (uint8 pileups + counts), the weight seed, and the expected outputs.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_fixtures.py

The reference cannot travel to the GPU box; the committed ``*.npz`` files are what tests replay.
"""
import hashlib
import importlib
import os
import sys
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference/python")
warnings.filterwarnings("ignore")

import torch  # noqa: E402

torch.set_num_threads(1)
import NNTools  # noqa: E402,F401  (reference; registers its layer types on torch.nn)
import MixtureOfExpertsAdvanced as REF  # noqa: E402  (reference)

from hello_amd import netspec as ns  # noqa: E402
from hello_amd import synth, weights  # noqa: E402


def state_digest(state):
    h = hashlib.sha256()
    for k in sorted(state):
        h.update(k.encode())
        h.update(np.ascontiguousarray(state[k]).tobytes())
    return h.hexdigest()[:16]


def reference_merged(config_name):
    """The older MoEMergedAdvanced family through the reference's own factory
    (MixtureOfExpertsAdvanced.py:614-654) and its "*Deeper" layer modules, weight-normed where the layer
    module has the switch; combiners and meta stay BatchNorm, as the factory leaves them."""
    hybrid = config_name == "merged_hybrid"
    cfg = {
        "weight_norm": True,
        "readConvNGS": "MoEReadConvolverDeeper",
        "alleleConvSingleNGS": "ExpertAlleleConvolverDeeper",
        "graphConvSingleNGS": "ExpertGraphConvolverDeeper",
        "kwargs": {"useAdditive": True},
    }
    if hybrid:
        cfg.update({
            "readConvTGS": "MoEReadConvolverDeeper",
            "alleleConvSingleTGS": "ExpertAlleleConvolverDeeper",
            "graphConvSingleTGS": "ExpertGraphConvolverDeeper",
            "graphConvHybrid": "ExpertGraphConvolverDeeper",
            "alleleConvCombiner": "ConvCombinerResNetDeeper",
            "siteConvCombiner": "ConvCombinerResNetDeeper",
            "meta": "MetaCombinerDeeper",
        })
    if config_name == "merged_hybrid_250":
        cfg = {
            "readConvNGS": "MoEReadConvolver250FeatureMap", "readConvTGS": "MoEReadConvolver250FeatureMap",
            "alleleConvSingleNGS": "ExpertAlleleConvolver250FeatureMap",
            "alleleConvSingleTGS": "ExpertAlleleConvolver250FeatureMap",
            "graphConvSingleNGS": "ExpertGraphConvolver250FeatureMap",
            "graphConvSingleTGS": "ExpertGraphConvolver250FeatureMap",
            "graphConvHybrid": "ExpertGraphConvolver250FeatureMap",
            "alleleConvCombiner": "ConvCombiner250FeatureMap",
            "meta": "MetaCombiner250FeatureMap",
            "kwargs": {"useAdditive": True},
        }
    moe = REF.createMoEFullMergedAdvancedModel(cfg)
    wrapper = REF.createMoEFullMergedAdvancedModelWrapper(moe)
    wrapper.eval()
    return wrapper


class _Holder:
    pass


def _build_on_top(moe, addendum_cfg):
    """MixtureOfExpertsDNNFastXferLearning.py:494-502: addendum layer lists -> Networks -> build_on_top."""
    import MixtureOfExpertsAdvancedXferLearning as XF
    orig = _Holder()
    orig.module = _Holder()
    orig.module.dnn = moe
    params = {k: XF.make_network(addendum_cfg, k) for k in addendum_cfg}
    new_moe, _ = XF.build_on_top(orig, **params)
    wrapper = XF.createMoEFullMergedAdvancedModelWrapper(new_moe)
    wrapper.eval()
    return wrapper


def reference_addendum(config_name):
    """Transfer-learning models: the base model + the reference's own *_addendum config through
    MixtureOfExpertsAdvancedXferLearning.build_on_top."""
    import MixtureOfExpertsAdvancedXferLearning as XF
    base = config_name[:-len("_addendum")]
    base_cfg = importlib.reload(importlib.import_module(ns.REFERENCE_CONFIG_MODULE[base])).configDict
    add_cfg = importlib.reload(importlib.import_module(ns.REFERENCE_CONFIG_MODULE[base] + "_addendum")).configDict
    return _build_on_top(XF.create_moe_attention_model(base_cfg), add_cfg)


def reference_model(config_name, norm):
    if config_name.startswith("merged"):
        return reference_merged(config_name)
    if config_name.endswith("_addendum"):
        return reference_addendum(config_name)
    if config_name == "single_tech_softplus":
        # the config module rewrites globals of the shared architecture modules: restore them afterwards
        import architectures.read_convolver as rc
        import architectures.compressor_conv_small as cc
        import architectures.xattn_subtract as xs
        try:
            for m in (rc, cc, xs):
                m.weight_norm = False           # as in a fresh interpreter (other configs switch it on)
            cfg = importlib.reload(importlib.import_module(ns.REFERENCE_CONFIG_MODULE[config_name])).configDict
            wrapper = REF.createMoEFullMergedAdvancedModelWrapper(REF.create_moe_attention_model(cfg))
        finally:
            for m in (rc, cc, xs):
                m.norm_type, m.activation = "BatchNorm1d", "ReLU"
                m.gen_config()
        wrapper.eval()
        return wrapper
    modname = ns.REFERENCE_CONFIG_MODULE[config_name]
    module = importlib.import_module(modname)
    module = importlib.reload(module)
    if norm == "bn":
        # same architecture with BatchNorm1d instead of weight norm: regenerate the layer lists
        import architectures.read_convolver as rc
        import architectures.compressor_conv_small as cc
        import architectures.xattn_subtract as xs
        for m in (rc, cc, xs):
            m.weight_norm = False
            m.gen_config()
        cfg = {"read_conv0": rc.config, "compressor0": cc.config, "xattn0": xs.config}
    else:
        cfg = module.configDict
    moe = REF.create_moe_attention_model(cfg)
    wrapper = REF.createMoEFullMergedAdvancedModelWrapper(moe)
    wrapper.eval()
    return wrapper


def load_state(wrapper, state):
    ref_keys = set(wrapper.state_dict().keys())
    assert ref_keys == set(state.keys()), (sorted(ref_keys ^ set(state.keys()))[:8])
    wrapper.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in state.items()})


def run_batched(wrapper, batch):
    t0 = torch.from_numpy(np.ascontiguousarray(np.transpose(batch.reads0, (0, 2, 1))))
    t1 = None
    rpa1 = None
    if batch.reads1 is not None:
        t1 = torch.from_numpy(np.ascontiguousarray(np.transpose(batch.reads1, (0, 2, 1))))
        rpa1 = batch.reads_per_allele1.tolist()
    seg = torch.from_numpy(batch.ref_onehot).float()
    captured = {}
    read_conv = getattr(wrapper.moeMerged, "read_convolver0", None) or wrapper.moeMerged.readConv0
    hook = read_conv.register_forward_hook(
        lambda m, i, o: captured.__setitem__("rc0", o.detach().numpy().copy()))
    with torch.no_grad():
        out = wrapper.moeMerged((t0, t1), batch.alleles_per_site.tolist(),
                                (batch.reads_per_allele0.tolist(), rpa1), seg)
    hook.remove()
    res = {}
    if isinstance(out, tuple):
        experts, meta = out
        res["logits"] = np.stack([e.numpy()[:, 0] for e in experts], axis=0)
        res["meta"] = meta.numpy()
    else:
        res["logits"] = out.numpy()[:, 0][None, :]
    rc0 = torch.from_numpy(captured["rc0"])
    res["frames0"] = REF.reduceSlots(rc0, batch.reads_per_allele0.tolist()).numpy()
    return res


def run_wrapper(wrapper, batch, names):
    """Per-site plug-in surface: exactly what caller_calling.scoreSite does (its lines 631-652)."""
    wrapper.providePredictions = True
    out = {}
    aoff = np.concatenate([[0], np.cumsum(batch.alleles_per_site)])
    r0off = np.concatenate([[0], np.cumsum(batch.reads_per_allele0)])
    r1off = None if batch.reads1 is None else np.concatenate([[0], np.cumsum(batch.reads_per_allele1)])
    for s in range(batch.n_sites):
        fd = {}
        for j, a in enumerate(range(aoff[s], aoff[s + 1])):
            t0 = torch.Tensor(batch.reads0[r0off[a]:r0off[a + 1]])
            t1 = None if r1off is None else torch.Tensor(batch.reads1[r1off[a]:r1off[a + 1]])
            fd[names[s][j]] = (t0, t1)
        seg = torch.from_numpy(batch.ref_onehot[s:s + 1]).float()
        with torch.no_grad():
            mix, e0, e1, e2, meta = wrapper(fd, seg)
        keys = list(mix.keys())
        out[f"site{s}_pairs"] = np.array(["|".join(k) for k in keys])
        out[f"site{s}_mix"] = np.array([float(mix[k]) for k in keys], dtype=np.float32)
        out[f"site{s}_e0"] = np.array([float(e0[k]) for k in keys], dtype=np.float32)
        out[f"site{s}_e1"] = np.array([float(e1[k]) for k in keys], dtype=np.float32)
        out[f"site{s}_e2"] = np.array([float(e2[k]) for k in keys], dtype=np.float32)
        out[f"site{s}_meta"] = meta.numpy().astype(np.float32)
    return out


def force_shapes(batch_kwargs, n_sites, seed, need):
    """Search seeds until the batch holds the edge cases named in ``need``: a 1-allele site ("one"),
    a >=3-allele site ("multi"), an unsupported allele carrying the all-zero dummy read ("dummy")."""
    for s in range(seed, seed + 500):
        b = synth.make_sites(n_sites, seed=s, **batch_kwargs)
        have = {
            "one": (b.alleles_per_site == 1).any(),
            "multi": (b.alleles_per_site >= 3).any(),
            "dummy": (b.reads0.reshape(b.reads0.shape[0], -1).max(axis=1) == 0).any(),
        }
        if all(have[n] for n in need):
            return b, s
    raise RuntimeError("no seed found")


CASES = [
    # name, config, norm, n_sites, weight_seed, make_sites kwargs, with_wrapper, keep_frames
    ("single_tech_batched", "single_tech", "wn", 6, 11, dict(coverage=30), True, True, ("one", "multi", "dummy")),
    ("single_tech_bn", "single_tech", "bn", 3, 12, dict(coverage=20), False, False, ("multi",)),
    ("single_tech_hp", "single_tech_hp", "wn", 4, 13, dict(coverage=(20, 80), channels=7, tech="pacbio"),
     True, False, ("one", "multi")),
    ("single_tech_deep", "single_tech", "wn", 3, 14, dict(coverage=(90, 128), tech="pacbio"), False, False, ("multi",)),
    ("hybrid_no_ensemble", "hybrid_no_ensemble", "wn", 4, 15, dict(coverage=30, hybrid_coverage=15),
     True, False, ("one", "multi", "dummy")),
    ("hybrid_full", "hybrid_full", "wn", 3, 16, dict(coverage=25, hybrid_coverage=12), True, False, ("multi",)),
    ("hybrid_ensemble2", "hybrid_ensemble2", "wn", 3, 17, dict(coverage=25, hybrid_coverage=12),
     True, False, ("one", "multi")),
    ("merged_single", "merged_single", "wn", 4, 18, dict(coverage=25), True, False, ("one", "multi", "dummy")),
    ("merged_hybrid", "merged_hybrid", "wn", 3, 19, dict(coverage=20, hybrid_coverage=10), True, False,
     ("one", "multi")),
    ("hybrid_no_ensemble_wide", "hybrid_no_ensemble_wide", "wn", 3, 24, dict(coverage=20, hybrid_coverage=10), True,
     False, ("multi",)),
    ("single_tech_softplus", "single_tech_softplus", "wn", 4, 23, dict(coverage=20), True, False, ("one", "multi")),
    ("single_tech_addendum", "single_tech_addendum", "wn", 3, 21, dict(coverage=20), True, False, ("multi",)),
    ("hybrid_no_ensemble_addendum", "hybrid_no_ensemble_addendum", "wn", 3, 22, dict(coverage=20, hybrid_coverage=10),
     True, False, ("multi",)),
    ("merged_hybrid_250", "merged_hybrid_250", "wn", 3, 20, dict(coverage=12, hybrid_coverage=6, window=250), True,
     False, ("multi",)),
]


def sanity_known_answer():
    """default-initialised reference, torch.manual_seed(1234)."""
    torch.manual_seed(1234)
    w = reference_model("single_tech", "wn")
    g = torch.Generator().manual_seed(99)
    fd = {}
    for name, r in (("A", 18), ("AT", 14), ("ATT", 3)):
        fd[name] = (torch.randint(0, 255, (r, 150, 6), generator=g).float(), None)
    with torch.no_grad():
        out = w(fd, torch.zeros(1, 150, 5))
    got = float(out[("A", "A")])
    assert abs(got - 0.31904656) < 2e-6, got
    print("reference import sanity OK:", got)


def make_pickle_fixture():
    """A REAL reference pickle (torch.save of a whole MoEMergedWrapperAdvanced, exactly what
    create_model_wrapper.py:7-10 writes), of a deliberately small architecture assembled from the
    reference's own layer generators, plus the reference's outputs on a seeded batch.  It pins the
    loader surface (hello_amd/loader.py) without needing a 6 MB model in the repository."""
    wn = dict(use_weight_norm=True)
    rb = dict(kernelSizes=[3, 3], paddings=[1, 1], dilations=[1, 1])
    read_conv = NNTools.SingleConvLayer(6, 8, 3, 0, 1, 1, **wn)
    read_conv.append({"type": "MaxPool1d", "kwargs": {"kernel_size": 3, "stride": 2, "padding": 0}})
    read_conv += [NNTools.ResidualBlockFTShortcut(8, 8, strides=[1, 1], **rb, **wn),
                  NNTools.ResidualBlockConvShortcut(8, 16, strides=[2, 1, 2], **rb, **wn)]
    comp = NNTools.SingleConvLayer(16, 16, 1, 0, 1, 1, **wn)
    comp += [NNTools.ResidualBlockConvShortcut(16, 32, strides=[2, 1, 2], **rb, **wn)]
    xattn = [{"type": "Fork", "kwargs": {"net_args": [[{"type": "Noop", "kwargs": {}}],
                                                       [{"type": "SelectArgument", "kwargs": {"select": 1}}]]}},
             {"type": "LinearCombination", "kwargs": {"coefficients": [2, -1]}}]
    xattn += NNTools.SingleConvLayer(32, 32, 1, 0, 1, 1, **wn)
    xattn += [NNTools.ResidualBlockConvShortcut(32, 64, strides=[2, 1, 2], **rb, **wn)]
    xattn += NNTools.terminus(64, 1, use_weight_norm=True)
    torch.manual_seed(4321)
    moe = REF.create_moe_attention_model({"read_conv0": read_conv, "compressor0": comp, "xattn0": xattn})
    wrapper = REF.createMoEFullMergedAdvancedModelWrapper(moe)
    wrapper.eval()
    with torch.no_grad():      # make g != ||v|| and scale the raw-byte input layer down
        for name, p in wrapper.named_parameters():
            if name.endswith("weight_g"):
                p.mul_(1.0 + 0.25 * torch.rand_like(p))
            if name == "moeMerged.read_convolver0.network.0.conv1d.weight_g":
                p.div_(128.0)
    path = os.path.join(HERE, "mini_reference.wrapper.dnn")
    torch.save(wrapper, path)
    batch = synth.make_sites(5, seed=321, coverage=12)
    res = run_batched(wrapper, batch)
    res.pop("frames0")
    res.update(run_wrapper(wrapper, batch, synth.allele_names(batch)))
    wrapper.providePredictions = False
    payload = dict(reads0=batch.reads0, reads_per_allele0=batch.reads_per_allele0,
                   alleles_per_site=batch.alleles_per_site, ref_onehot=batch.ref_onehot)
    payload.update({"exp_" + k: v for k, v in res.items()})
    np.savez_compressed(os.path.join(HERE, "mini_reference.npz"), **payload)
    print(f"mini_reference: pickle {os.path.getsize(path) / 1024:.0f} KB, logits "
          f"[{res['logits'].min():.3f},{res['logits'].max():.3f}]")


def make_merged_pickle_fixture():
    """A real reference pickle of the older family: a small hybrid MoEMergedAdvanced (additive, both
    combiners, meta) built by the reference factory from layer lists, torch.save'd whole."""
    wn = dict(use_weight_norm=True)
    rb = dict(kernelSizes=[3, 3], paddings=[1, 1], dilations=[1, 1])

    def read_conv():
        c = NNTools.SingleConvLayer(6, 8, 3, 0, 1, 1, **wn)
        c.append({"type": "MaxPool1d", "kwargs": {"kernel_size": 3, "stride": 2, "padding": 0}})
        return c + [NNTools.ResidualBlockFTShortcut(8, 8, strides=[1, 1], **rb, **wn),
                    NNTools.ResidualBlockConvShortcut(8, 16, strides=[2, 1, 2], **rb, **wn)]

    def allele_conv():
        return NNTools.SingleConvLayer(16, 16, 1, 0, 1, 1, **wn) + \
            [NNTools.ResidualBlockConvShortcut(16, 32, strides=[2, 1, 2], **rb, **wn)]

    def graph_conv(norm_kw, outputs=1):
        c = NNTools.SingleConvLayer(32, 32, 1, 0, 1, 1, **norm_kw)
        c += [NNTools.ResidualBlockConvShortcut(32, 64, strides=[2, 1, 2], **rb, **norm_kw)]
        return c + NNTools.terminus(64, outputs, **norm_kw)

    def combiner():
        return NNTools.SingleConvLayer(64, 48, 3, 1, 1, 1) + NNTools.SingleConvLayer(48, 32, 1, 0, 1, 1)

    torch.manual_seed(8765)
    moe = REF.createMoEFullMergedAdvancedModel({
        "readConvNGS": read_conv(), "readConvTGS": read_conv(),
        "alleleConvSingleNGS": allele_conv(), "alleleConvSingleTGS": allele_conv(),
        "graphConvSingleNGS": graph_conv(wn), "graphConvSingleTGS": graph_conv(wn), "graphConvHybrid": graph_conv(wn),
        "alleleConvCombiner": combiner(), "siteConvCombiner": combiner(), "meta": graph_conv({}, 3),
        "kwargs": {"useAdditive": True},
    })
    wrapper = REF.createMoEFullMergedAdvancedModelWrapper(moe)
    wrapper.eval()
    with torch.no_grad():
        for name, p in wrapper.named_parameters():
            if name.endswith("weight_g"):
                p.mul_(1.0 + 0.25 * torch.rand_like(p))
            if name in ("moeMerged.readConv0.network.0.conv1d.weight_g", "moeMerged.readConv1.network.0.conv1d.weight_g"):
                p.div_(128.0)
        for name, b in wrapper.named_buffers():      # non-trivial BatchNorm statistics
            if name.endswith("running_mean"):
                b.copy_(0.2 * torch.rand_like(b) - 0.1)
            if name.endswith("running_var"):
                b.copy_(0.5 + torch.rand_like(b))
    path = os.path.join(HERE, "mini_merged.wrapper.dnn")
    torch.save(wrapper, path)
    batch = synth.make_sites(4, seed=654, coverage=12, hybrid_coverage=8)
    res = run_batched(wrapper, batch)
    res.pop("frames0")
    res.update(run_wrapper(wrapper, batch, synth.allele_names(batch)))
    wrapper.providePredictions = False
    payload = dict(reads0=batch.reads0, reads_per_allele0=batch.reads_per_allele0,
                   reads1=batch.reads1, reads_per_allele1=batch.reads_per_allele1,
                   alleles_per_site=batch.alleles_per_site, ref_onehot=batch.ref_onehot)
    payload.update({"exp_" + k: v for k, v in res.items()})
    np.savez_compressed(os.path.join(HERE, "mini_merged.npz"), **payload)
    print(f"mini_merged: pickle {os.path.getsize(path) / 1024:.0f} KB, logits "
          f"[{res['logits'].min():.3f},{res['logits'].max():.3f}] meta {res['meta'][0]}")


def make_addendum_pickle_fixture():
    """A real pickle of a transfer-learning model: the small single-tech architecture of mini_reference with
    two more residual blocks on every sub-network, assembled by the reference's build_on_top."""
    import MixtureOfExpertsAdvancedXferLearning as XF
    wn = dict(use_weight_norm=True)
    rb = dict(kernelSizes=[3, 3], paddings=[1, 1], dilations=[1, 1])
    read_conv = NNTools.SingleConvLayer(6, 8, 3, 0, 1, 1, **wn)
    read_conv.append({"type": "MaxPool1d", "kwargs": {"kernel_size": 3, "stride": 2, "padding": 0}})
    read_conv += [NNTools.ResidualBlockConvShortcut(8, 16, strides=[2, 1, 2], **rb, **wn)]
    comp = NNTools.SingleConvLayer(16, 16, 1, 0, 1, 1, **wn)
    comp += [NNTools.ResidualBlockConvShortcut(16, 32, strides=[2, 1, 2], **rb, **wn)]
    xattn = [{"type": "Fork", "kwargs": {"net_args": [[{"type": "Noop", "kwargs": {}}],
                                                       [{"type": "SelectArgument", "kwargs": {"select": 1}}]]}},
             {"type": "LinearCombination", "kwargs": {"coefficients": [2, -1]}}]
    xattn += NNTools.SingleConvLayer(32, 32, 1, 0, 1, 1, **wn)
    xattn += [NNTools.ResidualBlockConvShortcut(32, 64, strides=[2, 1, 2], **rb, **wn)]
    xattn += NNTools.terminus(64, 1, use_weight_norm=True)
    more = lambda c: [NNTools.ResidualBlockFTShortcut(c, c, strides=[1, 1], **rb, **wn) for _ in range(2)]   # noqa: E731
    torch.manual_seed(2468)
    moe = XF.create_moe_attention_model({"read_conv0": read_conv, "compressor0": comp, "xattn0": xattn})
    wrapper = _build_on_top(moe, {
        "read_convolver0_addendum": more(16), "compressor0_addendum": more(32),
        "xattn0_addendum": more(64) + NNTools.terminus(64, 1, use_weight_norm=True)})
    with torch.no_grad():
        for name, p in wrapper.named_parameters():
            if name.endswith("weight_g"):
                p.mul_(1.0 + 0.25 * torch.rand_like(p))
            if name == "moeMerged.read_convolver0.0.network.0.conv1d.weight_g":
                p.div_(128.0)
    path = os.path.join(HERE, "mini_addendum.wrapper.dnn")
    torch.save(wrapper, path)
    batch = synth.make_sites(4, seed=975, coverage=12)
    res = run_batched(wrapper, batch)
    res.pop("frames0")
    res.update(run_wrapper(wrapper, batch, synth.allele_names(batch)))
    wrapper.providePredictions = False
    payload = dict(reads0=batch.reads0, reads_per_allele0=batch.reads_per_allele0,
                   alleles_per_site=batch.alleles_per_site, ref_onehot=batch.ref_onehot)
    payload.update({"exp_" + k: v for k, v in res.items()})
    np.savez_compressed(os.path.join(HERE, "mini_addendum.npz"), **payload)
    print(f"mini_addendum: pickle {os.path.getsize(path) / 1024:.0f} KB, logits "
          f"[{res['logits'].min():.3f},{res['logits'].max():.3f}]")


FRAME_CASES = ["single_tech_hp", "single_tech_deep", "hybrid_no_ensemble", "merged_single", "merged_hybrid_250",
               "single_tech_softplus", "single_tech_addendum", "single_tech_bn"]


def make_frames_fixture():
    """Kernel-level pins for the fused read convolver: the reference's per-allele frames
    reduceSlots(read_convolver(x)) ([sum A, 64, L2], MixtureOfExpertsAdvanced.py:162-163) of BOTH technologies on
    the committed inputs of existing fixtures (7 channels, 90-128 reads per site, 250 bp windows, Softplus, the
    transfer-learning blocks, BatchNorm).  One file, frames.npz: <case>_frames0 / <case>_frames1."""
    sys.path.insert(0, os.path.dirname(HERE))
    from util import load_fixture
    out = {}
    for name in FRAME_CASES:
        spec, state, batch, _ = load_fixture(name)
        cfg = str(np.load(os.path.join(HERE, name + ".npz"))["config"])
        norm = str(np.load(os.path.join(HERE, name + ".npz"))["norm"])
        wrapper = reference_model(cfg, norm)
        load_state(wrapper, state)
        moe = wrapper.moeMerged
        nets = [getattr(moe, "read_convolver0", None) or moe.readConv0]
        second = getattr(moe, "read_convolver1", None) or getattr(moe, "readConv1", None)
        if batch.reads1 is not None and second is not None:
            nets.append(second)
        for tech, net in enumerate(nets):
            reads = batch.reads0 if tech == 0 else batch.reads1
            rpa = batch.reads_per_allele0 if tech == 0 else batch.reads_per_allele1
            x = torch.from_numpy(np.ascontiguousarray(np.transpose(reads, (0, 2, 1)))).float()
            with torch.no_grad():
                frames = REF.reduceSlots(net(x), rpa.tolist()).numpy()
            out[f"{name}_frames{tech}"] = frames.astype(np.float32)
            print(f"frames {name} tech {tech}: {frames.shape}, |max| {np.abs(frames).max():.3f}")
    path = os.path.join(HERE, "frames.npz")
    np.savez_compressed(path, **out)
    print(f"frames.npz: {os.path.getsize(path) / 1024:.0f} KB")


def main():
    only = set(sys.argv[1:])          # optional: regenerate just the named fixtures
    sanity_known_answer()
    if not only or "frames" in only:
        make_frames_fixture()
        if only == {"frames"}:
            return
    if not only or "mini_reference" in only:
        make_pickle_fixture()
    if not only or "mini_merged" in only:
        make_merged_pickle_fixture()
    if not only or "mini_addendum" in only:
        make_addendum_pickle_fixture()
    for name, cfg, norm, n_sites, wseed, kw, with_wrapper, keep_frames, need in CASES:
        if only and name not in only:
            continue
        spec = ns.build(cfg, norm=norm) if norm != "wn" else ns.build(cfg)
        state = weights.synth_state(spec, seed=wseed)
        wrapper = reference_model(cfg, norm)
        load_state(wrapper, state)
        batch, iseed = force_shapes(kw, n_sites, 100 + wseed, need)
        res = run_batched(wrapper, batch)
        if not keep_frames:
            res.pop("frames0")
        if with_wrapper:
            res.update(run_wrapper(wrapper, batch, synth.allele_names(batch)))
        payload = dict(
            config=np.array(cfg), norm=np.array(norm), weight_seed=np.array(wseed),
            state_digest=np.array(state_digest(state)), input_seed=np.array(iseed),
            reads0=batch.reads0, reads_per_allele0=batch.reads_per_allele0,
            alleles_per_site=batch.alleles_per_site, ref_onehot=batch.ref_onehot,
        )
        if batch.reads1 is not None:
            payload.update(reads1=batch.reads1, reads_per_allele1=batch.reads_per_allele1)
        payload.update({"exp_" + k: v for k, v in res.items()})
        path = os.path.join(HERE, name + ".npz")
        np.savez_compressed(path, **payload)
        print(f"{name}: S={batch.n_sites} A={batch.n_alleles} R0={batch.reads0.shape[0]} "
              f"logits[{res['logits'].min():.3f},{res['logits'].max():.3f}] -> "
              f"{os.path.getsize(path) / 1024:.0f} KB")


if __name__ == "__main__":
    main()
