rm -f gpurun_out/r06_full_launch_parity.txt
python -m pytest tests -m gpu -x -q --durations=8 > gpurun_out/r06_gputests.txt 2>&1; tail -12 gpurun_out/r06_gputests.txt
mv gpurun_out/r06_full_launch_parity.txt gpurun_out/r06_full_launch_parity_fp32.txt
HELLO_TEST_ARITHMETIC=bf16x3 python -m pytest tests/test_gpu_fullsize.py -m gpu -x -q -k every_site > gpurun_out/r06_t3b.log 2>&1; tail -2 gpurun_out/r06_t3b.log
mv gpurun_out/r06_full_launch_parity.txt gpurun_out/r06_full_launch_parity_bf16x3.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
HELLO_PROFILE_COMMIT=5867033 bash tools/profile_round.sh r06 > gpurun_out/r06_profile_round.log 2>&1; tail -2 gpurun_out/r06_profile_round.log
cp gpurun_out/profiles_r06/hbm_traffic.json profiles/hbm_traffic.json
(time python bench.py) > gpurun_out/r06_bench.json 2> gpurun_out/r06_bench.err; tail -5 gpurun_out/r06_bench.err
python tools/one_site_profile.py --per-op > gpurun_out/r06_one_site_profile.txt 2>&1
