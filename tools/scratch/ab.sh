for c in hybrid_no_ensemble hybrid_full; do for m in lanes seq lanes seq; do timeout -k 10 200 python tools/scratch/ab.py $c $m 2>&1 | grep -v amdgpu.ids; done; done
