"""Pileup-tensor producer on the GPU.

The reference turns the reads supporting each allele of a site into ``uint8 [reads, L, 6|7]`` colour
tensors in C++ (``AlleleSearcherLite.computeFeatures`` -> ``libCallability``'s
``computeFeaturesColoredSimple``, reference python/AlleleSearcherLite.py:232-251,
c++/src/AlleleSearcherLiteFiltered.cpp:1031-1180), one allele at a time on one CPU thread.  Here every read
of every allele of every site of a batch is encoded in ONE launch (``hello_engine_featurize``) straight
into the packed ``[sum R, L, C]`` buffer + reads-per-allele / alleles-per-site counts the scoring engine
consumes, so featurisation and scoring can stay on the device.

Read collection (BAM access, allele assembly, read -> allele support) remains the caller's job; this
module takes what ``AlleleSearcherLite`` holds per read: bases, base qualities, CIGAR tuples, reference
start, mapping quality, orientation, haplotag (python/AlleleSearcherLite.py:159-174).
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field
from typing import List, Optional, Sequence, Tuple

import numpy as np

from .engine import HELLO_IN_DEVICE, HELLO_OUT_DEVICE, Engine, _check


@dataclass
class AlignedRead:
    bases: str
    quals: Sequence[int]
    cigar: Sequence[Tuple[int, int]]          # pysam-style (operation, length) tuples
    ref_start: int
    mapq: int = 40
    orientation: int = 1                      # > 0 forward strand
    hp: int = 0                               # haplotag 0 | 1 | 2


@dataclass
class SiteReads:
    reference: str                            # reference window
    window_start: int                         # genome position of reference[0]
    assembly_start: int                       # allele span [start, stop) in genome coordinates
    assembly_stop: int
    alleles: List[Tuple[str, List[AlignedRead]]] = field(default_factory=list)   # (allele, supporting reads)


def pack_sites(sites: Sequence[SiteReads]):
    """Flatten sites -> the arrays of hello_engine_featurize + the batch's count arrays.  An allele without
    supporting reads contributes one read with an empty CIGAR: the all-zero dummy row of the reference
    (AlleleSearcherLiteFiltered.cpp:1037-1043)."""
    bases, quals, cigars = [], [], []
    read_off, cigar_off = [0], [0]
    ref_start, mapq, orient, hp, site_of_read = [], [], [], [], []
    ref, ref_off, wstart, a0, a1 = [], [0], [], [], []
    reads_per_allele, alleles_per_site = [], []
    for s, site in enumerate(sites):
        ref.append(site.reference.encode("ascii"))
        ref_off.append(ref_off[-1] + len(site.reference))
        wstart.append(site.window_start)
        a0.append(site.assembly_start)
        a1.append(site.assembly_stop)
        alleles_per_site.append(len(site.alleles))
        for _, reads in site.alleles:
            group = reads if reads else [AlignedRead("", [], [], 0)]
            reads_per_allele.append(len(group))
            for rd in group:
                assert len(rd.bases) == len(rd.quals)
                bases.append(rd.bases.encode("ascii"))
                quals.append(bytes(bytearray(int(q) & 0xFF for q in rd.quals)))
                read_off.append(read_off[-1] + len(rd.bases))
                for op, length in rd.cigar:
                    cigars.append((int(length) << 4) | int(op))
                cigar_off.append(cigar_off[-1] + len(rd.cigar))
                ref_start.append(rd.ref_start)
                mapq.append(min(int(rd.mapq), 255))
                orient.append(1 if rd.orientation > 0 else -1)
                hp.append(int(rd.hp))
                site_of_read.append(s)
    u8 = lambda b: np.frombuffer(b"".join(b) + b"\0", dtype=np.uint8).copy()      # noqa: E731 (never empty)
    return dict(
        bases=u8(bases), quals=u8(quals), read_off=np.asarray(read_off, np.int64),
        cigars=np.asarray(cigars + [0], np.uint32), cigar_off=np.asarray(cigar_off, np.int64),
        ref_start=np.asarray(ref_start, np.int64), mapq=np.asarray(mapq, np.uint8),
        orientation=np.asarray(orient, np.int8), hp=np.asarray(hp, np.uint8),
        site_of_read=np.asarray(site_of_read, np.int32), ref=u8(ref), ref_off=np.asarray(ref_off, np.int64),
        window_start=np.asarray(wstart, np.int64), asm_start=np.asarray(a0, np.int64),
        asm_stop=np.asarray(a1, np.int64),
        reads_per_allele=np.asarray(reads_per_allele, np.int32),
        alleles_per_site=np.asarray(alleles_per_site, np.int32))


def featurize(engine: Engine, sites: Sequence[SiteReads], feature_length: int = 150, include_hp: bool = False,
              device_output: bool = False):
    """-> (pileups uint8 [sum R, L, C], reads_per_allele int32 [A], alleles_per_site int32 [S]).  With
    ``device_output`` the pileups are a torch CUDA tensor that can go straight into ``Engine.forward``."""
    # ``sites``: SiteReads objects, or the already flat arrays of ``pack_sites`` (shards.PackedShard.featurizer_arrays)
    p = sites if isinstance(sites, dict) else pack_sites(sites)
    n_reads, n_sites = int(p["site_of_read"].shape[0]), int(p["alleles_per_site"].shape[0])
    channels = 7 if include_hp else 6
    flags = 0
    if device_output:
        import torch
        out = torch.empty((n_reads, feature_length, channels), dtype=torch.uint8, device=f"cuda:{engine.device}")
        out_ptr, stream = out.data_ptr(), None
        flags |= HELLO_OUT_DEVICE
    else:
        out = np.empty((n_reads, feature_length, channels), dtype=np.uint8)
        out_ptr, stream = out.ctypes.data, None
    fn = engine.lib.hello_engine_featurize
    fn.restype = C.c_int
    vp = C.c_void_p
    fn.argtypes = [vp] * 16 + [C.c_int64, C.c_int32, C.c_int32, C.c_int32, vp, C.c_int32, vp]
    ptr = lambda k: p[k].ctypes.data                                              # noqa: E731
    def launch(handle):
        _check(fn(engine.handle, ptr("bases"), ptr("quals"), ptr("read_off"), ptr("cigars"), ptr("cigar_off"),
                  ptr("ref_start"), ptr("mapq"), ptr("orientation"), ptr("hp"), ptr("site_of_read"), ptr("ref"),
                  ptr("ref_off"), ptr("window_start"), ptr("asm_start"), ptr("asm_stop"),
                  n_reads, n_sites, feature_length, channels, out_ptr, flags, handle))
    if device_output:
        with engine.on_stream(out.device) as handle:      # ordered with the caller's current torch stream
            launch(handle)
    else:
        launch(None)
    return out, p["reads_per_allele"], p["alleles_per_site"]


FEATURIZE_ARRAYS = ("bases", "quals", "read_off", "cigars", "cigar_off", "ref_start", "mapq", "orientation", "hp", "site_of_read",
                    "ref", "ref_off", "window_start", "asm_start", "asm_stop")          # hello_engine_featurize's order


def featurize_device(engine: Engine, pointers: dict, n_reads: int, n_sites: int, feature_length: int, channels: int,
                     out_pointer: int, stream: int):
    """``hello_engine_featurize`` on arrays ALREADY in device memory (``pointers``: name of FEATURIZE_ARRAYS -> device
    address), asynchronous on ``stream`` -- the staged form the shard pipeline uses (one H2D copy per launch carries all
    fifteen arrays)."""
    fn = engine.lib.hello_engine_featurize
    fn.restype = C.c_int
    vp = C.c_void_p
    fn.argtypes = [vp] * 16 + [C.c_int64, C.c_int32, C.c_int32, C.c_int32, vp, C.c_int32, vp]
    _check(fn(engine.handle, *[pointers[k] for k in FEATURIZE_ARRAYS], n_reads, n_sites, feature_length, channels,
              out_pointer, HELLO_IN_DEVICE | HELLO_OUT_DEVICE, stream))
