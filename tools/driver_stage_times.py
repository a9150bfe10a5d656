"""Throughput of the scoring-stage driver (python -m hello_amd.call = hello_amd.shard_pipeline) from shard FILES to the
final VCF, from ONE host process on one GPU, and where its host time goes.

    python tools/driver_stage_times.py [--sites 1048576] [--shard_sites 400,4000] [--coverage 30] [--threads 16]

A synthetic shard (30 reads per site, 150-base reads, two alleles per site) is replicated into enough ``.hshard`` files
(16 distinct ones with shifted coordinates, the rest links to them; tmpfs: the page cache stands in for the upstream stage
handing shards over) for ``--sites`` sites, once as reference-sized shards (~400 sites: call.py:162, maxShards 500 per chromosome) and once as large ones, and
``call.main`` runs end to end: per-shard .vcf / .features / .mean.vcf / .log files + results.output.vcf.  Printed: sites/s
of the whole run and of the scoring loop, the loop's stage clocks (feeder waiting for readers, staging, record stage on
its thread), the process's peak RSS -- at a quarter of the shards and at all of them, which is how "memory is flat in the
number of shards" is checked -- and the record stage alone on 1 / 4 / N threads.
"""
import argparse
import logging
import os
import resource
import shutil
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from hello_amd import call as driver, loader, netspec as ns, records, shard_pipeline as sp, shards, weights  # noqa: E402

BAM_CMATCH = 0


def template_payload(rng, n_sites, coverage):
    """One synthetic shard as flat arrays, built vectorised (no Python object per read)."""
    ref_len, spacing = 520, 700
    window_start = 1000 + spacing * np.arange(n_sites, dtype=np.int64)
    ref = rng.choice(np.frombuffer(b"ACGT", np.uint8), size=n_sites * ref_len)
    start = window_start + 240 + rng.integers(0, 20, size=n_sites)
    ref_base = ref[np.arange(n_sites) * ref_len + (start - window_start)]
    alt = np.frombuffer(b"ACGT", np.uint8)[(np.searchsorted(np.frombuffer(b"ACGT", np.uint8), ref_base) + rng.integers(1, 4, size=n_sites)) % 4]
    allele_text = np.stack([ref_base, alt], axis=1).reshape(-1)
    per_site = np.maximum(2, rng.poisson(coverage, size=n_sites))
    first = np.clip(rng.binomial(per_site, 0.5), 1, per_site - 1)
    rpa = np.stack([first, per_site - first], axis=1).reshape(-1).astype(np.int32)
    R = int(rpa.sum())
    site_of_read = np.repeat(np.arange(n_sites), per_site)
    payload = dict(
        chromosome_text=np.frombuffer(b"chr1", np.uint8), chromosome_text_off=np.array([0, 4], np.int64),
        chromosome_of_site=np.zeros(n_sites, np.int32), start=start, stop=start + 1, window_start=window_start,
        ref=ref, ref_off=ref_len * np.arange(n_sites + 1, dtype=np.int64), alleles_per_site=np.full(n_sites, 2, np.int32),
        allele_text=allele_text, allele_text_off=np.arange(2 * n_sites + 1, dtype=np.int64), has_second=np.array(0),
        reads_per_allele0=rpa, bases0=rng.choice(np.frombuffer(b"ACGT", np.uint8), size=R * 150),
        quals0=rng.integers(2, 60, size=R * 150).astype(np.uint8), read_off0=150 * np.arange(R + 1, dtype=np.int64),
        cigars0=np.full(R, (150 << 4) | BAM_CMATCH, np.uint32), cigar_off0=np.arange(R + 1, dtype=np.int64),
        ref_start0=start[site_of_read] - rng.integers(20, 130, size=R), mapq0=rng.integers(0, 80, size=R).astype(np.uint8),
        orientation0=rng.choice(np.array([-1, 1], np.int8), size=R), hp0=np.zeros(R, np.uint8))
    return payload, R


def write_shards(directory, payload, n_files, span, distinct=16):
    """``n_files`` shard files: ``distinct`` real ones (coordinates shifted per file), the rest symbolic links to them in
    rotation -- the page cache then holds ``distinct`` shards however many the run reads."""
    shards.PackedShard(dict(payload))                         # the template validates
    for k in range(n_files):
        path = os.path.join(directory, f"shard{k}.hshard")
        if k >= distinct:
            os.symlink(os.path.join(directory, f"shard{k % distinct}.hshard"), path)
            continue
        moved = dict(payload)
        for name in ("start", "stop", "window_start", "ref_start0"):
            moved[name] = payload[name] + k * span
        shards.write_flat(path, moved)


def rss_mb():
    return resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1024.0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sites", type=int, default=1048576)
    ap.add_argument("--shard_sites", default="400,4000")
    ap.add_argument("--coverage", type=int, default=30)
    ap.add_argument("--threads", type=int, default=min(16, len(os.sched_getaffinity(0))))
    ap.add_argument("--config", default="single_tech")
    ap.add_argument("--arithmetic", default="fp32")
    ap.add_argument("--no-record-alone", action="store_true", help="skip the record stage's stand-alone timing at the end")
    ap.add_argument("--only-all", action="store_true", help="skip the quarter-of-the-shards run (the memory-flatness check)")
    args = ap.parse_args()
    logging.basicConfig(level=logging.WARNING)
    rng = np.random.default_rng(7)
    base = tempfile.mkdtemp(dir="/dev/shm" if os.path.isdir("/dev/shm") else None, prefix="hello_driver_")
    try:
        spec = ns.build(args.config)
        model = os.path.join(base, "model.hello.npz")
        loader.save_native(model, args.config, weights.synth_state(spec, seed=1))
        print(f"# python tools/driver_stage_times.py --sites {args.sites} --shard_sites {args.shard_sites} --threads {args.threads} "
              f"--arithmetic {args.arithmetic}   "
              f"({args.config}, {args.coverage} reads per site, {len(os.sched_getaffinity(0))} CPUs visible)")
        for shard_sites in [int(x) for x in args.shard_sites.split(",")]:
            payload, n_reads = template_payload(rng, shard_sites, args.coverage)
            n_files = max(4, args.sites // shard_sites)
            for label, count in ((("all", n_files),) if args.only_all else (("quarter", max(1, n_files // 4)), ("all", n_files))):
                sdir, work = os.path.join(base, f"shards_{shard_sites}_{label}"), os.path.join(base, f"work_{shard_sites}_{label}")
                os.makedirs(sdir)
                write_shards(sdir, payload, count, 700 * shard_sites + 10_000)
                argv = ["--network", model, "--workdir", work, "--shards", sdir, "--num_threads", str(args.threads),
                        "--arithmetic", args.arithmetic]
                captured = []
                handler = logging.Handler()
                handler.emit = lambda record, captured=captured: captured.append(record.getMessage())
                log = logging.getLogger("hello_amd.call")
                log.setLevel(logging.INFO)
                log.addHandler(handler)
                t0 = time.perf_counter()
                driver.main(driver.parser().parse_args(argv))
                dt = time.perf_counter() - t0
                log.removeHandler(handler)
                sites = count * shard_sites
                lines = sum(1 for _ in open(os.path.join(work, "results.output.vcf")))
                print(f"{shard_sites:5d} sites/shard x {count:5d} shards = {sites:8d} sites ({n_reads / shard_sites:.1f} reads/site): "
                      f"{dt:6.2f} s end to end incl. model load + final VCF = {sites / dt:9,.0f} sites/s; {lines} lines in the final VCF; "
                      f"peak RSS {rss_mb():,.0f} MB")
                for m in captured:
                    if "sites/s" in m or "Completed runs" in m:
                        print("      " + m)
                shutil.rmtree(sdir)
                shutil.rmtree(work)
        if args.no_record_alone:
            return
        # the record stage alone (host only): one launch of 8 192 sites
        payload, _ = template_payload(rng, 8192, args.coverage)
        shard = shards.PackedShard(dict(payload))
        table = sp.site_table([shard])
        post = rng.random((4, 3 * 8192)).astype(np.float32)
        for threads in sorted({1, 4, args.threads}):
            best = 1e9
            for _ in range(5):
                t0 = time.perf_counter()
                records.site_records(table, post, None, threads=threads).close()
                best = min(best, time.perf_counter() - t0)
            print(f"record stage alone, 8 192 sites, {threads:2d} threads: {best * 1e3:6.2f} ms = {8192 / best:11,.0f} sites/s")
    finally:
        shutil.rmtree(base, ignore_errors=True)


if __name__ == "__main__":
    main()
