"""What one rank of an 8-rank run has of the HOST, measured on one GPU.

At N = 8 a rank owns an eighth of the node's CPUs (hello_amd.shard.rank_cpus), not the whole host that every one-GPU number
so far was taken on.  The scaling target (>= 0.9 linear to 8 GPUs) fails on the host side first: feeder / staging / record
threads short of CPUs.  This tool runs, in fresh child processes whose CPU affinity is set BEFORE anything touches the GPU,

  (a) the headline bench   python bench.py --gpus 1 --no-secondary --no-cpu-baseline
  (b) the driver           python tools/driver_stage_times.py   (2 621 reference-sized shards of 400 sites -> VCF)

once unrestricted and once per CPU allowance: an eighth of the visible CPUs (= shard.rank_cpus(0, 8) on a node without
NUMA information), and 16 / 8 / 4 / 2 CPUs to find where the rate starts to fall.  The parent never initialises the GPU and
never execs: children are plain subprocesses.  Prints one table; ratios are against the unrestricted run of the same lease.

    python tools/one_eighth_host.py [--cpus 8th,16,8,4,2] [--driver-sites 1048576] [--steps 10]
"""
import argparse
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from hello_amd import shard  # noqa: E402   (imports numpy only)


def quota_cores():
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        return None if quota == "max" else int(quota) / int(period)
    except (OSError, ValueError):
        return None


def child(cmd, cpus):
    """Run ``cmd`` in a child whose affinity is ``cpus`` (None = as inherited) from its first instruction on."""
    pre = "" if cpus is None else f"import os; os.sched_setaffinity(0, {sorted(cpus)!r}); "
    boot = pre + "import runpy, sys; sys.argv = sys.argv[1:]; runpy.run_path(sys.argv[0], run_name='__main__')"
    return subprocess.run([sys.executable, "-c", boot] + cmd, cwd=ROOT, capture_output=True, text=True, timeout=1500)


def bench_rate(cpus, steps):
    out = child([os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", str(steps), "--warmup", "3", "--no-secondary", "--no-cpu-baseline"], cpus)
    if out.returncode != 0:
        return None, out.stderr[-600:]
    line = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    return line["value"], f"roofline.launch_ms {line['roofline']['launch_ms']}, host_cpus_of_rank0 {line['config']['host_cpus_of_rank0']}"


def driver_rate(cpus, sites, threads):
    out = child([os.path.join(ROOT, "tools", "driver_stage_times.py"), "--sites", str(sites), "--shard_sites", "400", "--threads", str(threads),
                 "--no-record-alone", "--only-all"], cpus)
    if out.returncode != 0:
        return None, None, out.stderr[-600:]
    text = out.stdout
    loop = [float(m.replace(",", "")) for m in re.findall(r"\(([\d,]+) sites/s; waiting", text)]
    e2e = [float(m.replace(",", "")) for m in re.findall(r"=\s+([\d,]+) sites/s; \d+ lines", text)]
    detail = [ln.strip() for ln in text.splitlines() if "rank 0:" in ln]
    return (loop[-1] if loop else None), (e2e[-1] if e2e else None), (detail[-1] if detail else text[-400:])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cpus", default="8th,16,8,4,2")
    ap.add_argument("--driver-sites", type=int, default=1048576)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--skip-driver", action="store_true")
    args = ap.parse_args()
    allowed = sorted(os.sched_getaffinity(0))
    eighth = shard.rank_cpus(0, 8, None)                    # the equal split of the affinity mask: what pin_rank falls back to
    print(f"# host: {len(allowed)} CPUs visible, cgroup quota {quota_cores()} cores; shard.rank_cpus(0, 8) = {len(eighth)} CPUs "
          f"({eighth[0]}-{eighth[-1]})", flush=True)
    cases = [("unrestricted", None)]
    for tok in args.cpus.split(","):
        cases.append((f"1/8 of the host ({len(eighth)} CPUs)", eighth) if tok == "8th" else (f"{int(tok)} CPUs", eighth[:int(tok)]))
    cases.append(("unrestricted (again)", None))
    base_b = base_loop = base_e2e = None
    for name, cpus in cases:
        threads = 16 if cpus is None else max(2, min(16, len(cpus)))
        b, note = bench_rate(cpus, args.steps)
        if base_b is None:
            base_b = b
        print(f"{name:32s} bench.py --gpus 1 --no-secondary : {str(b and round(b)):>9} sites/s  x{(b / base_b) if b and base_b else float('nan'):.3f}   ({note})", flush=True)
        if not args.skip_driver:
            loop, e2e, detail = driver_rate(cpus, args.driver_sites, threads)
            if base_loop is None:
                base_loop, base_e2e = loop, e2e
            print(f"{'':32s} driver, 400-site shards, --threads {threads:2d}: scoring loop {str(loop and round(loop)):>9} sites/s  "
                  f"x{(loop / base_loop) if loop and base_loop else float('nan'):.3f}; end to end {str(e2e and round(e2e)):>9}  "
                  f"x{(e2e / base_e2e) if e2e and base_e2e else float('nan'):.3f}", flush=True)
            print(f"{'':32s}   {detail}", flush=True)


if __name__ == "__main__":
    main()
