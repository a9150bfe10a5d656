python -m pytest tests/test_gpu_parity.py tests/test_gpu_wrapper.py tests/test_gpu_shared.py tests/test_gpu_calls.py -m gpu -x -q > gpurun_out/r06_t5.log 2>&1; tail -3 gpurun_out/r06_t5.log
python tools/one_site_profile.py 2>&1 | tail -1
python tools/one_site_profile.py --config hybrid_full 2>&1 | tail -1
python tools/per_site_multiprocess.py --workers 4 --calls 2000 2>&1 | grep aggregate
python tools/per_site_multiprocess.py --shared --workers 16 --calls 3000 2>&1 | grep aggregate
