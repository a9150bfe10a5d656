"""Binding of the host-side record stage of libhello_mi355x.so (include/hello_mi355x.h: ``hello_site_records``).

One multi-threaded C call turns a whole launch's pair posteriors into what the reference's per-shard caller and final
stage write per site in Python: the shard's VCF lines (python/caller_calling.py:698-743), the ``.features`` pickle
streams (:743-754, one per shard), and the final VCF's lines from the meta-weighted mean of the experts
(python/prepareVcf.py:138-168).  ``hello_amd.vcf`` holds the same rules as readable Python (the per-site plug-in path
uses it, and tests/test_records.py holds the two to each other line for line).
"""
from __future__ import annotations

import ctypes as C
import pickle
import pickletools
from typing import Dict, Optional, Sequence

import numpy as np

from .engine import _check, load_library


class _SiteTable(C.Structure):
    _fields_ = [("n_sites", C.c_int32), ("alleles_per_site", C.c_void_p), ("allele_text", C.c_void_p),
                ("allele_text_off", C.c_void_p), ("n_chromosomes", C.c_int32), ("chromosome_text", C.c_void_p),
                ("chromosome_text_off", C.c_void_p), ("chromosome_of_site", C.c_void_p), ("start", C.c_void_p),
                ("stop", C.c_void_p), ("ref_windows", C.c_void_p), ("ref_window_off", C.c_void_p),
                ("window_start", C.c_void_p), ("genome", C.POINTER(C.c_void_p)), ("genome_len", C.c_void_p),
                ("keep", C.c_void_p)]


class _FeaturesFormat(C.Structure):
    _fields_ = [("meta_prefix", C.c_char_p), ("meta_prefix_len", C.c_int32), ("meta_suffix", C.c_char_p),
                ("meta_suffix_len", C.c_int32)]


class _View(C.Structure):
    _fields_ = [("n_sites", C.c_int32), ("n_shards", C.c_int32), ("shard_vcf", C.c_void_p), ("shard_vcf_off", C.c_void_p),
                ("mean_vcf", C.c_void_p), ("mean_vcf_off", C.c_void_p), ("mean_position", C.c_void_p),
                ("features", C.c_void_p), ("features_off", C.c_void_p), ("n_records", C.c_void_p),
                ("best_pair", C.c_void_p), ("best_p", C.c_void_p), ("qual", C.c_void_p)]


_bound = None


def _lib():
    global _bound
    if _bound is None:
        lib = load_library()
        lib.hello_site_records.argtypes = [C.POINTER(_SiteTable), C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int32,
                                           C.POINTER(_FeaturesFormat), C.c_int32, C.POINTER(C.c_void_p)]
        lib.hello_site_records.restype = C.c_int
        lib.hello_records_get.argtypes = [C.c_void_p, C.POINTER(_View)]
        lib.hello_records_get.restype = C.c_int
        lib.hello_records_destroy.argtypes = [C.c_void_p]
        lib.hello_records_destroy.restype = None
        _bound = lib
    return _bound


_meta_format = None


def meta_pickle_format():
    """-> (prefix, suffix): this NumPy's pickle of a float32 [3] array, cut around its 12 payload bytes (no memo
    operations left in it), so ``prefix + three floats + suffix`` is that array inside a larger pickle stream."""
    global _meta_format
    if _meta_format is None:
        probe = np.array([1.0000001, -2.0000002, 3.0000005], np.float32)
        stream = pickletools.optimize(pickle.dumps(probe, protocol=3))
        body = stream[2:-1]                                  # without PROTO 3 and STOP
        at = body.find(probe.tobytes())
        if at < 0 or body.find(probe.tobytes(), at + 1) >= 0:
            raise RuntimeError("cannot locate the payload in NumPy's pickle of a float32 array")
        for op, _, _ in pickletools.genops(stream):
            if op.name in ("GET", "BINGET", "LONG_BINGET", "PUT", "BINPUT", "LONG_BINPUT", "MEMOIZE"):
                raise RuntimeError("NumPy's array pickle uses the memo: the template would clash with the stream's own")
        _meta_format = (body[:at], body[at + 12:])
    return _meta_format


def text_table(strings) -> tuple:
    """A sequence / NumPy array of ASCII strings -> (uint8 blob with one byte of padding, int64 offsets [n + 1])."""
    from .shards import text_table as table
    text, off = table(strings)
    return np.concatenate([text, np.zeros(1, np.uint8)]), off


class SiteTable:
    """The per-site strings and coordinates ``hello_site_records`` reads, as flat arrays (kept alive here)."""

    def __init__(self, alleles_per_site, allele_text, allele_text_off, chromosome_names: Sequence[str], chromosome_of_site,
                 start, stop, ref_windows=None, ref_window_off=None, window_start=None,
                 genomes: Optional[Dict[str, bytes]] = None, keep=None):
        c = np.ascontiguousarray
        self.alleles_per_site = c(alleles_per_site, dtype=np.int32)
        self.allele_text, self.allele_text_off = c(allele_text, dtype=np.uint8), c(allele_text_off, dtype=np.int64)
        self.names = [str(n) for n in chromosome_names]
        self.chrom_text, self.chrom_off = text_table(np.array(self.names) if self.names else np.array([], "U1"))
        self.chromosome_of_site = c(chromosome_of_site, dtype=np.int32)
        self.start, self.stop = c(start, dtype=np.int64), c(stop, dtype=np.int64)
        self.ref_windows = c(ref_windows, dtype=np.uint8) if ref_windows is not None else None
        self.ref_window_off = c(ref_window_off, dtype=np.int64) if ref_window_off is not None else None
        self.window_start = c(window_start, dtype=np.int64) if window_start is not None else None
        self.keep = c(keep, dtype=np.uint8) if keep is not None else None
        n = int(self.alleles_per_site.shape[0])
        if int(self.alleles_per_site.sum()) + 1 != self.allele_text_off.shape[0]:
            raise ValueError("allele_text_off must hold one offset per allele plus one")
        for name in ("chromosome_of_site", "start", "stop", "window_start", "keep"):
            a = getattr(self, name)
            if a is not None and a.shape[0] != n:
                raise ValueError(f"{name} must hold one entry per site")
        if self.ref_window_off is not None and self.ref_window_off.shape[0] != n + 1:
            raise ValueError("ref_window_off must hold one offset per site plus one")
        self._genome_bytes, self._genome_ptr, self._genome_len = [], None, None
        if genomes:
            ptrs = (C.c_void_p * len(self.names))()
            lens = np.zeros(len(self.names), np.int64)
            for i, name in enumerate(self.names):
                g = genomes.get(name)
                if g is None:
                    continue
                g = g if isinstance(g, (bytes, bytearray)) else str(g).encode("ascii")
                self._genome_bytes.append(g)
                ptrs[i] = C.cast(C.c_char_p(g), C.c_void_p)
                lens[i] = len(g)
            self._genome_ptr, self._genome_len = ptrs, lens
        p = lambda a: a.ctypes.data if a is not None else None               # noqa: E731
        self.struct = _SiteTable(n, p(self.alleles_per_site), p(self.allele_text), p(self.allele_text_off), len(self.names),
                                 p(self.chrom_text), p(self.chrom_off), p(self.chromosome_of_site), p(self.start), p(self.stop),
                                 p(self.ref_windows), p(self.ref_window_off), p(self.window_start),
                                 self._genome_ptr, p(self._genome_len), p(self.keep))

    @property
    def n_sites(self):
        return int(self.alleles_per_site.shape[0])


class Records:
    """Result of one ``hello_site_records`` call.  Texts are zero-copy ``memoryview``s of library memory: use them (or
    copy what must outlive) before ``close()``."""

    def __init__(self, handle):
        self._lib = _lib()
        self.handle = handle
        v = _View()
        _check(self._lib.hello_records_get(handle, C.byref(v)))
        S, K = v.n_sites, v.n_shards
        self.n_sites, self.n_shards = S, K

        def arr(ptr, n, ctype, dtype):
            if n == 0 or not ptr:
                return np.zeros(n, dtype)
            return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(ctype)), shape=(n,))
        self.shard_vcf_off = arr(v.shard_vcf_off, S + 1, C.c_int64, np.int64)
        self.mean_vcf_off = arr(v.mean_vcf_off, S + 1, C.c_int64, np.int64)
        self.mean_position = arr(v.mean_position, S, C.c_int64, np.int64)
        self.features_off = arr(v.features_off, K + 1, C.c_int64, np.int64)
        self.n_records = arr(v.n_records, K, C.c_int32, np.int32)
        self.best_pair = arr(v.best_pair, 5 * S, C.c_int32, np.int32).reshape(5, S)
        self.best_p = arr(v.best_p, 5 * S, C.c_double, np.float64).reshape(5, S)
        self.qual = arr(v.qual, 5 * S, C.c_double, np.float64).reshape(5, S)

        def text(ptr, n):
            return memoryview((C.c_char * n).from_address(ptr)).cast("B") if n and ptr else memoryview(b"")
        self.shard_vcf = text(v.shard_vcf, int(self.shard_vcf_off[S]) if S else 0)
        self.mean_vcf = text(v.mean_vcf, int(self.mean_vcf_off[S]) if S else 0)
        self.features = text(v.features, int(self.features_off[K]) if K else 0)

    def close(self):
        if self.handle:
            self.shard_vcf = self.mean_vcf = self.features = memoryview(b"")
            self._lib.hello_records_destroy(self.handle)
            self.handle = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def site_records(table: SiteTable, posteriors: np.ndarray, meta: Optional[np.ndarray], shard_site_off=None,
                 features: bool = True, threads: int = 0) -> Records:
    """posteriors float32 [4, P] and meta float32 [S, 3] | None on the host, as ``Engine.forward`` returns them."""
    lib = _lib()
    post = np.ascontiguousarray(posteriors, dtype=np.float32)
    if post.ndim != 2 or post.shape[0] != 4:
        raise ValueError("posteriors must be float32 [4, n_pairs]")
    m = np.ascontiguousarray(meta, dtype=np.float32) if meta is not None else None
    if m is not None and m.shape != (table.n_sites, 3):
        raise ValueError("meta must be float32 [n_sites, 3]")
    off = np.ascontiguousarray(shard_site_off, dtype=np.int32) if shard_site_off is not None else None
    fmt = None
    if features:
        prefix, suffix = meta_pickle_format()
        fmt = _FeaturesFormat(prefix, len(prefix), suffix, len(suffix))
    handle = C.c_void_p()
    _check(lib.hello_site_records(C.byref(table.struct), post.ctypes.data, int(post.shape[1]),
                                  m.ctypes.data if m is not None else None,
                                  off.ctypes.data if off is not None else None, int(off.shape[0]) - 1 if off is not None else 1,
                                  C.byref(fmt) if fmt is not None else None, int(threads), C.byref(handle)))
    return Records(handle)
