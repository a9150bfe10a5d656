python -m pytest tests/test_gpu_parity.py tests/test_gpu_wrapper.py tests/test_gpu_shared.py -m gpu -x -q > gpurun_out/r06_t4.log 2>&1; tail -3 gpurun_out/r06_t4.log
for n in 1 2 4 8 16 32 64; do python3 tools/scratch/trace.py $n 2>&1 | tail -1; done
python tools/per_site_multiprocess.py --shared --workers 16 --calls 3000 2>&1 | grep aggregate
python tools/per_site_multiprocess.py --shared --workers 32 --calls 3000 2>&1 | grep aggregate
