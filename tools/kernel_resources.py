"""Register / LDS / occupancy table of every kernel of one HIP source (hipcc -Rpass-analysis=kernel-resource-usage), one line each:

    python tools/kernel_resources.py hello_amd/csrc/conv_wino.hip [name filter]
"""
import re
import subprocess
import sys

src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=off", "-Ihello_amd/csrc", "-Iinclude",
       "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/dev/null"]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
rows, cur = [], None
for line in out.splitlines():
    m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\S+)", line)
    if not m:
        continue
    key, val = m.group(1).strip(), m.group(2)
    if key == "Function Name":
        cur = {"name": subprocess.run(["c++filt", val], capture_output=True, text=True).stdout.strip()}
        rows.append(cur)
    elif cur is not None:
        cur[key] = val
for r in rows:
    if flt in r["name"]:
        print(f"{r.get('VGPRs', '?'):>4} v {r.get('AGPRs', '?'):>3} a  scratch {r.get('ScratchSize', '?'):>4}  occ {r.get('Occupancy', '?'):>2}  lds {r.get('LDS Size', '?'):>6}  "
              f"{re.sub(r'^void hello::', '', r['name'])[:110]}")
