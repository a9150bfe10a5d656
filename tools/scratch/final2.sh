python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_shared.py -m gpu -x -q > gpurun_out/r06_t2.log 2>&1; tail -4 gpurun_out/r06_t2.log
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
HELLO_PROFILE_COMMIT=d4e50cb bash tools/profile_round.sh r06 > gpurun_out/r06_profile_round.log 2>&1; tail -3 gpurun_out/r06_profile_round.log
cp gpurun_out/profiles_r06/hbm_traffic.json profiles/hbm_traffic.json
(time python bench.py) > gpurun_out/r06_bench.json 2> gpurun_out/r06_bench.err; tail -6 gpurun_out/r06_bench.err
