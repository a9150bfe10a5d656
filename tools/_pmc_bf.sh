#!/bin/bash
# usage: _pmc_bf.sh <lib or ""> <tag>
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export HELLO_LIB=$1
rm -rf gpurun_out/pmc_$2
rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d gpurun_out/pmc_$2 -- python3 tools/_bf_quick.py bf16x3 > gpurun_out/pmc_$2.log 2>&1
f=$(find gpurun_out/pmc_$2 -name "*counter_collection.csv" | head -1)
python3 - <<PY
import csv, collections
acc = collections.defaultdict(float); n=0
for row in csv.DictReader(open("$f")):
    if "readconv_kernel" in row["Kernel_Name"] and int(row["Grid_Size"]) > 1000000:
        acc[row["Counter_Name"]] += float(row["Counter_Value"])
        n += row["Counter_Name"] == "SQ_WAVE_CYCLES"
print("$2", n, {k: round(v / max(n,1)) for k, v in acc.items()})
PY
