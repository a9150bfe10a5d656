#!/bin/bash
# Sweep of the shared scoring server (hello_amd.shared) on one GPU: worker processes x engines x linger.
#     bash tools/per_site_shared_sweep.sh > gpurun_out/r06_per_site_shared_sweep.txt
for cfg in "2 120" "2 0" "2 250" "3 120" "4 120" "1 120"; do
  set -- $cfg
  for w in 4 8 16 32; do
    echo "== engines $1, linger $2 us, workers $w"
    HELLO_SHARED_LINGER_US=$2 python tools/per_site_multiprocess.py --shared --engines $1 --workers $w --calls 3000 2>&1 | grep -v "worker [0-9]*:\|amdgpu.ids"
    sleep 2
  done
done
