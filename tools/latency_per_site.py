#!/usr/bin/env python3
"""Latency of the per-site plug-in call network(featureDict, ref_segment) and of small batches."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hello_amd import netspec as ns, synth, weights
from hello_amd.wrapper import ScoringNetwork

spec = ns.build("single_tech")
net = ScoringNetwork(spec, weights.synth_state(spec, seed=1), providePredictions=True)
batch = synth.make_sites(256, seed=2, coverage=30)
names = synth.allele_names(batch)
aoff = np.concatenate([[0], np.cumsum(batch.alleles_per_site)])
roff = np.concatenate([[0], np.cumsum(batch.reads_per_allele0)])
sites = []
for s in range(batch.n_sites):
    fd = {names[s][j]: (torch.Tensor(batch.reads0[roff[a]:roff[a + 1]]), None) for j, a in enumerate(range(aoff[s], aoff[s + 1]))}
    sites.append((fd, torch.zeros(1, 150, 5)))
for fd, seg in sites[:20]:
    net(fd, seg)
t0 = time.perf_counter()
for fd, seg in sites:
    net(fd, seg)
dt = time.perf_counter() - t0
print(f"per-site call: {1e3 * dt / len(sites):.3f} ms/site  ({len(sites) / dt:.0f} sites/s)")
for n in (16, 256):
    net.score_sites(sites[:n])
    t0 = time.perf_counter()
    for _ in range(5):
        net.score_sites(sites[:n])
    dt = (time.perf_counter() - t0) / 5
    print(f"score_sites({n}): {1e3 * dt:.2f} ms  ({n / dt:.0f} sites/s, host packing included)")
eng = net.engine
t0 = time.perf_counter()
for _ in range(20):
    eng.forward_batch(batch.site_slice(0, 1))
print(f"engine.forward 1 site (host arrays, no dict packing): {1e3 * (time.perf_counter() - t0) / 20:.3f} ms")
