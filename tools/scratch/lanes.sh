timeout -k 10 600 python -m pytest tests/test_gpu_lanes.py -m gpu -x -q > gpurun_out/r06_lanes.log 2>&1; tail -3 gpurun_out/r06_lanes.log
for c in hybrid_no_ensemble hybrid_full hybrid_ensemble2; do timeout -k 10 120 python tools/one_site_profile.py --config $c 2>&1 | grep "one-site"; done
timeout -k 10 300 python tools/per_site_multiprocess.py --shared --workers 16 --calls 2000 --config C4 2>&1 | grep aggregate
timeout -k 10 300 python tools/per_site_multiprocess.py --shared --workers 16 --calls 2000 --config hybrid_full 2>&1 | grep aggregate
